// keygen.hip - commitment-key generation on the device.
//
// Reference: PublicParams::setup (/root/reference/src/proof/nova.rs:196-216) -> arecibo CommitmentEngine::setup(b"ck", n) ->
// DlogGroup::from_label: SHAKE256(label) squeezed 32 bytes per point, point i = pasta_curves hash_to_curve("from_uniform_bytes")
// of its bytes, all batch-normalised to affine.  Both dependencies are un-vendored; restated from their published algorithms
// (oracle/keygen_ref.py states them in Python and checks the constants mathematically; parity unpinned against upstream bytes):
//   hash_to_field   RFC 9380 expand_message_xmd with BLAKE2b-512, DST "<domain>-<curve>_XMD:BLAKE2b_SSWU_RO_", two elements
//   map_to_curve    simplified SWU on the 3-isogenous curve y^2 = x^3 + A x + 1265, Z = -13, sign of y = sgn0(u)
//   + / iso_map     the two images added on the isogenous curve, then the degree-3 isogeny onto y^2 = x^3 + 5
// The XOF stream is inherently sequential: the host squeezes it (Keccak-f[1600]); everything per point - three BLAKE2b hashes, two
// Legendre symbols, two square roots (Tonelli-Shanks, 2-adicity 32), three inversions - runs one point per lane.
#include <memory>
#include <mutex>
#include <string>

#include "common.hpp"
#include "curve.cuh"
#include "keccak.hpp"

namespace lurk {

// ---- SHAKE256 (host): Keccak-f from keccak.hpp ----------------------------------------------------------------------------------
static void shake_xof(size_t RATE, const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len) {
    uint64_t st[25] = {0};
    uint8_t* sb = reinterpret_cast<uint8_t*>(st);  // little-endian host
    size_t pos = 0;
    for (size_t i = 0; i < in_len; i++) {
        sb[pos++] ^= in[i];
        if (pos == RATE) { keccak_f(st); pos = 0; }
    }
    sb[pos] ^= 0x1f;
    sb[RATE - 1] ^= 0x80;
    keccak_f(st);
    size_t got = 0;
    while (got < out_len) {
        size_t take = out_len - got < RATE ? out_len - got : RATE;
        memcpy(out + got, sb, take);
        got += take;
        if (got < out_len) keccak_f(st);
    }
}
void shake256(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len) { shake_xof(136, in, in_len, out, out_len); }

// ---- BLAKE2b-512 (device), unkeyed, 64-byte digest ------------------------------------------------------------
// (one copy for each side: the per-point map below is __host__ __device__ - the device builds keys, the host maps the handful of points
// a probe record or a CPU test asks for, lurk_hip_ck_from_label_host)
#define KG_HD __host__ __device__
#define B2B_IV_INIT {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL, \
                     0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL}
__device__ __constant__ uint64_t B2B_IV_DEV[8] = B2B_IV_INIT;
static const uint64_t B2B_IV_HOST[8] = B2B_IV_INIT;
#define B2B_SIGMA_INIT { \
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}, \
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8}, \
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9}, \
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10}, \
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}, \
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}}
__device__ __constant__ uint8_t B2B_SIGMA_DEV[12][16] = B2B_SIGMA_INIT;
static const uint8_t B2B_SIGMA_HOST[12][16] = B2B_SIGMA_INIT;
#if defined(__HIP_DEVICE_COMPILE__)
#define B2B_IV B2B_IV_DEV
#define B2B_SIGMA B2B_SIGMA_DEV
#else
#define B2B_IV B2B_IV_HOST
#define B2B_SIGMA B2B_SIGMA_HOST
#endif

KG_HD inline uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
// h: chaining value; m: 16 message words (little-endian); t: bytes so far including this block
KG_HD inline void b2b_compress(uint64_t* h, const uint64_t* m, uint64_t t, bool last) {
    uint64_t v[16];
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = B2B_IV[i]; }
    v[12] ^= t;
    if (last) v[14] = ~v[14];
#define B2B_G(a, b, c, d, x, y)      \
    v[a] = v[a] + v[b] + (x);        \
    v[d] = rotr64(v[d] ^ v[a], 32);  \
    v[c] = v[c] + v[d];              \
    v[b] = rotr64(v[b] ^ v[c], 24);  \
    v[a] = v[a] + v[b] + (y);        \
    v[d] = rotr64(v[d] ^ v[a], 16);  \
    v[c] = v[c] + v[d];              \
    v[b] = rotr64(v[b] ^ v[c], 63);
#pragma unroll 1
    for (int r = 0; r < 12; r++) {
        const uint8_t* s = B2B_SIGMA[r];
        B2B_G(0, 4, 8, 12, m[s[0]], m[s[1]]) B2B_G(1, 5, 9, 13, m[s[2]], m[s[3]]) B2B_G(2, 6, 10, 14, m[s[4]], m[s[5]]) B2B_G(3, 7, 11, 15, m[s[6]], m[s[7]])
        B2B_G(0, 5, 10, 15, m[s[8]], m[s[9]]) B2B_G(1, 6, 11, 12, m[s[10]], m[s[11]]) B2B_G(2, 7, 8, 13, m[s[12]], m[s[13]]) B2B_G(3, 4, 9, 14, m[s[14]], m[s[15]])
    }
#undef B2B_G
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}
KG_HD inline void b2b_init(uint64_t* h) {
    for (int i = 0; i < 8; i++) h[i] = B2B_IV[i];
    h[0] ^= 0x01010040ULL;  // digest length 64, no key, fanout 1, depth 1
}
// hash of a message of len <= 256 bytes held as bytes in `buf` (zero padded to a multiple of 128)
KG_HD inline void b2b_hash(const uint8_t* buf, unsigned len, uint8_t* out64) {
    uint64_t h[8];
    b2b_init(h);
    const unsigned nblocks = len <= 128 ? 1 : 2;
    for (unsigned b = 0; b < nblocks; b++) {
        uint64_t m[16];
        for (int i = 0; i < 16; i++) {
            uint64_t w = 0;
            for (int k = 0; k < 8; k++) w |= (uint64_t)buf[b * 128 + i * 8 + k] << (8 * k);
            m[i] = w;
        }
        const bool last = b + 1 == nblocks;
        b2b_compress(h, m, last ? len : 128, last);
    }
    for (int i = 0; i < 8; i++)
        for (int k = 0; k < 8; k++) out64[i * 8 + k] = (uint8_t)(h[i] >> (8 * k));
}

// ---- per-curve constants (host-built, passed by value) ---------------------------------------------------------
template <class P>
struct KeygenConsts {
    Fe<P> a, b, z, iso[13];      // Montgomery
    Fe<P> r3;                    // R^3 mod p: the high half of a wide reduction
    Fe<P> ts_c;                  // g^T, g a non-residue: generator of the 2^32 subgroup
    uint32_t ts_exp[8];          // (T - 1) / 2, T = (p - 1) / 2^32
    uint32_t legendre_exp[8];    // (p - 1) / 2
    uint8_t dst_prime[64];       // DST || len(DST)
    uint32_t dst_len;
    uint32_t msg_len;            // bytes of XOF output per point (lurk_hip_ck_params::bytes_per_point)
};

template <class P>
KG_HD inline Fe<P> kg_from_be64(const uint8_t* b64, const Fe<P>& r3) {  // big-endian 512-bit integer mod p, Montgomery
    Fe<P> lo, hi;
    for (int i = 0; i < 8; i++) {
        uint32_t wl = 0, wh = 0;
        for (int k = 0; k < 4; k++) {
            wl |= (uint32_t)b64[63 - (i * 4 + k)] << (8 * k);
            wh |= (uint32_t)b64[31 - (i * 4 + k)] << (8 * k);
        }
        lo.l[i] = wl;
        hi.l[i] = wh;
    }
    fe_cond_sub2<P>(lo.l); fe_cond_sub<P>(lo.l);  // < 2^256 < 4p  ->  < p
    fe_cond_sub2<P>(hi.l); fe_cond_sub<P>(hi.l);
    return fe_add<P>(fe_to_mont<P>(lo), fe_mul<P>(hi, r3));  // lo R + hi R^2 = (lo + hi 2^256) R
}

template <class P>
KG_HD inline bool kg_is_square(const Fe<P>& a, const KeygenConsts<P>& K) {
    if (fe_is_zero<P>(a)) return true;
    return fe_eq<P>(fe_pow<P>(a, K.legendre_exp), fe_one<P>());
}
// a square root of a square a (Tonelli-Shanks; per-lane loop counts differ, the wave runs the longest)
template <class P>
KG_HD inline Fe<P> kg_sqrt(const Fe<P>& a, const KeygenConsts<P>& K) {
    if (fe_is_zero<P>(a)) return a;
    const Fe<P> one = fe_one<P>();
    const Fe<P> w = fe_pow<P>(a, K.ts_exp);  // a^((T-1)/2)
    Fe<P> x = fe_mul<P>(a, w);               // a^((T+1)/2)
    Fe<P> t = fe_mul<P>(x, w);               // a^T
    Fe<P> c = K.ts_c;
    int m = 32;
    while (!fe_eq<P>(t, one)) {
        int i = 0;
        Fe<P> tt = t;
        while (!fe_eq<P>(tt, one) && i < m) { tt = fe_sqr<P>(tt); i++; }
        if (i >= m) break;  // not a square: the caller never asks
        Fe<P> bb = c;
        for (int k = 0; k < m - i - 1; k++) bb = fe_sqr<P>(bb);
        m = i;
        c = fe_sqr<P>(bb);
        t = fe_mul<P>(t, c);
        x = fe_mul<P>(x, bb);
    }
    return x;
}
template <class P>
KG_HD inline bool kg_is_odd(const Fe<P>& mont) { return fe_from_mont<P>(mont).l[0] & 1u; }

// g(x) = x^3 + a x + b on the isogenous curve
template <class P>
KG_HD inline Fe<P> kg_g(const Fe<P>& x, const KeygenConsts<P>& K) {
    return fe_add<P>(fe_mul<P>(fe_add<P>(fe_sqr<P>(x), K.a), x), K.b);
}

// one point: hash_to_curve(domain)(msg), msg = K.msg_len bytes (128 + msg_len + 3 + dst_len <= 256: checked where K is built)
template <class P>
KG_HD inline Affine<P> keygen_point(const uint8_t* msg, const KeygenConsts<P>& K) {
    // ---- hash_to_field: b0 = H(0^128 || msg || 0,128,0 || DST'), b1 = H(b0 || 1 || DST'), b2 = H(b0 ^ b1 || 2 || DST')
    uint8_t buf[256], b0[64], b1[64], b2[64];
    for (int k = 0; k < 256; k++) buf[k] = 0;
    const unsigned ml = K.msg_len;
    for (unsigned k = 0; k < ml; k++) buf[128 + k] = msg[k];
    buf[128 + ml] = 0; buf[129 + ml] = 128; buf[130 + ml] = 0;
    for (unsigned k = 0; k < K.dst_len; k++) buf[131 + ml + k] = K.dst_prime[k];
    b2b_hash(buf, 131 + ml + K.dst_len, b0);
    for (int k = 0; k < 128; k++) buf[k] = 0;
    for (int k = 0; k < 64; k++) buf[k] = b0[k];
    buf[64] = 1;
    for (unsigned k = 0; k < K.dst_len; k++) buf[65 + k] = K.dst_prime[k];
    b2b_hash(buf, 65 + K.dst_len, b1);
    for (int k = 0; k < 64; k++) buf[k] = b0[k] ^ b1[k];
    buf[64] = 2;
    b2b_hash(buf, 65 + K.dst_len, b2);
    const Fe<P> u[2] = {kg_from_be64<P>(b1, K.r3), kg_from_be64<P>(b2, K.r3)};
    // ---- simplified SWU, both elements; one shared inversion for the two denominators
    Fe<P> num_x1[2], div[2], z_u2[2];
    const Fe<P> one = fe_one<P>();
    for (int e = 0; e < 2; e++) {
        z_u2[e] = fe_mul<P>(K.z, fe_sqr<P>(u[e]));
        const Fe<P> ta = fe_add<P>(fe_sqr<P>(z_u2[e]), z_u2[e]);
        num_x1[e] = fe_mul<P>(K.b, fe_add<P>(ta, one));
        div[e] = fe_mul<P>(K.a, fe_is_zero<P>(ta) ? K.z : fe_neg<P>(ta));
    }
    const Fe<P> inv01 = fe_inv<P>(fe_mul<P>(div[0], div[1]));
    const Fe<P> inv_div[2] = {fe_mul<P>(inv01, div[1]), fe_mul<P>(inv01, div[0])};
    Fe<P> qx[2], qy[2];
    for (int e = 0; e < 2; e++) {
        const Fe<P> x1 = fe_mul<P>(num_x1[e], inv_div[e]);
        const Fe<P> gx1 = kg_g<P>(x1, K);
        const bool sq = kg_is_square<P>(gx1, K);
        const Fe<P> x = sq ? x1 : fe_mul<P>(z_u2[e], x1);
        Fe<P> y = kg_sqrt<P>(sq ? gx1 : kg_g<P>(x, K), K);
        if (kg_is_odd<P>(u[e]) != kg_is_odd<P>(y)) y = fe_neg<P>(y);  // sgn0(u) == sgn0(y)
        qx[e] = x;
        qy[e] = y;
    }
    // ---- q0 + q1 on the isogenous curve (affine), then the isogeny
    Affine<P> res;
    res.x = fe_zero<P>();
    res.y = fe_zero<P>();
    Fe<P> sx, sy;
    bool identity = false;
    {
        Fe<P> num, den;
        if (fe_eq<P>(qx[0], qx[1])) {
            if (fe_is_zero<P>(fe_add<P>(qy[0], qy[1]))) identity = true;
            const Fe<P> xx = fe_sqr<P>(qx[0]);
            num = fe_add<P>(fe_add<P>(fe_dbl<P>(xx), xx), K.a);
            den = fe_dbl<P>(qy[0]);
        } else {
            num = fe_sub<P>(qy[1], qy[0]);
            den = fe_sub<P>(qx[1], qx[0]);
        }
        if (!identity) {
            const Fe<P> lam = fe_mul<P>(num, fe_inv<P>(den));
            sx = fe_sub<P>(fe_sub<P>(fe_sqr<P>(lam), qx[0]), qx[1]);
            sy = fe_sub<P>(fe_mul<P>(lam, fe_sub<P>(qx[0], sx)), qy[0]);
        }
    }
    if (!identity) {
        const Fe<P>* c = K.iso;
        const Fe<P> nx = fe_add<P>(fe_mul<P>(fe_add<P>(fe_mul<P>(fe_add<P>(fe_mul<P>(c[0], sx), c[1]), sx), c[2]), sx), c[3]);
        const Fe<P> dx = fe_add<P>(fe_mul<P>(fe_add<P>(sx, c[4]), sx), c[5]);
        const Fe<P> ny = fe_mul<P>(fe_add<P>(fe_mul<P>(fe_add<P>(fe_mul<P>(fe_add<P>(fe_mul<P>(c[6], sx), c[7]), sx), c[8]), sx), c[9]), sy);
        const Fe<P> dy = fe_add<P>(fe_mul<P>(fe_add<P>(fe_mul<P>(fe_add<P>(sx, c[10]), sx), c[11]), sx), c[12]);
        const Fe<P> dd = fe_mul<P>(dx, dy);
        if (!fe_is_zero<P>(dd)) {  // dd == 0: a kernel point of the isogeny, image = identity
            const Fe<P> inv = fe_inv<P>(dd);
            res.x = fe_mul<P>(nx, fe_mul<P>(inv, dy));
            res.y = fe_mul<P>(ny, fe_mul<P>(inv, dx));
        }
    }
    return res;
}

template <class P>
__global__ __launch_bounds__(128) void keygen_kernel(const uint8_t* __restrict__ uniform, size_t n, KeygenConsts<P> K, Affine<P>* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 128 + threadIdx.x;
    if (i >= n) return;
    out[i] = keygen_point<P>(uniform + i * K.msg_len, K);
}

// pasta_curves 0.5.0 constants (canonical integers, 4 x u64 little-endian): the isogenous curves' A, B = 1265, Z = -13 and the
// 13 isogeny coefficients per curve; validated mathematically by tests/test_oracle_keygen.py
static const uint64_t KG_A[2][4] = {{0x92bb4b0b657a014bULL, 0xb74134581a27a59fULL, 0x49be2d7258370742ULL, 0x18354a2eb0ea8c9cULL},
                                    {0xc515ad7242eaa6b1ULL, 0x9673928c7d01b212ULL, 0x81639c4d96f78773ULL, 0x267f9b2ee592271aULL}};
static const uint64_t KG_ISO[2][13][4] = {
    {{0x775f6034aaaaaaabULL, 0x4081775473d8375bULL, 0xe38e38e38e38e38eULL, 0x0e38e38e38e38e38ULL},
     {0x8cf863b02814fb76ULL, 0x0f93b82ee4b99495ULL, 0x267c7ffa51cf412aULL, 0x3509afd51872d88eULL},
     {0x0eb64faef37ea4f7ULL, 0x380af066cfeb6d69ULL, 0x98c7d7ac3d98fd13ULL, 0x17329b9ec5253753ULL},
     {0xeebec06955555580ULL, 0x8102eea8e7b06eb6ULL, 0xc71c71c71c71c71cULL, 0x1c71c71c71c71c71ULL},
     {0xc47f2ab668bcd71fULL, 0x9c434ac1c96b6980ULL, 0x5a607fcce0494a79ULL, 0x1d572e7ddc099cffULL},
     {0x2aa3af1eae5b6604ULL, 0xb4abf9fb9a1fc81cULL, 0x1d13bf2a7f22b105ULL, 0x325669becaecd5d1ULL},
     {0x5ad985b5e38e38e4ULL, 0x7642b01ad461bad2ULL, 0x4bda12f684bda12fULL, 0x1a12f684bda12f68ULL},
     {0xc67c31d8140a7dbbULL, 0x07c9dc17725cca4aULL, 0x133e3ffd28e7a095ULL, 0x1a84d7ea8c396c47ULL},
     {0x02e2be87d225b234ULL, 0x1765e924f7459378ULL, 0x303216cce1db9ff1ULL, 0x3fb98ff0d2ddcaddULL},
     {0x93e53ab371c71c4fULL, 0x0ac03e8e134eb3e4ULL, 0x7b425ed097b425edULL, 0x025ed097b425ed09ULL},
     {0x5a28279b1d1b42aeULL, 0x5941a3a4a97aa1b3ULL, 0x0790bfb3506defb6ULL, 0x0c02c5bcca0e6b7fULL},
     {0x4d90ab820b12320aULL, 0xd976bbfabbc5661dULL, 0x573b3d7f7d681310ULL, 0x17033d3c60c68173ULL},
     {0x992d30ecfffffde5ULL, 0x224698fc094cf91bULL, 0x0000000000000000ULL, 0x4000000000000000ULL}},
    {{0x43cd42c800000001ULL, 0x0205dd51cfa0961aULL, 0x8e38e38e38e38e39ULL, 0x38e38e38e38e38e3ULL},
     {0x8b95c6aaf703bcc5ULL, 0x216b8861ec72bd5dULL, 0xacecf10f5f7c09a2ULL, 0x1d935247b4473d17ULL},
     {0xaeac67bbeb586a3dULL, 0xd59d03d23b39cb11ULL, 0xed7ee4a9cdf78f8fULL, 0x18760c7f7a9ad20dULL},
     {0xfb539a6f0000002bULL, 0xe1c521a795ac8356ULL, 0x1c71c71c71c71c71ULL, 0x31c71c71c71c71c7ULL},
     {0xb7284f7eaf21a2e9ULL, 0xa3ad678129b604d3ULL, 0x1454798a5b5c56b2ULL, 0x0a2de485568125d5ULL},
     {0xf169c187d2533465ULL, 0x30cd6d53df49d235ULL, 0x0c621de8b91c242aULL, 0x14735171ee542778ULL},
     {0x6bef1642aaaaaaabULL, 0x5601f4709a8adcb3ULL, 0xda12f684bda12f68ULL, 0x12f684bda12f684bULL},
     {0x8bee58e5fb81de63ULL, 0x21d910aefb03b31dULL, 0xd6767887afbe04d1ULL, 0x2ec9a923da239e8bULL},
     {0x4986913ab4443034ULL, 0x97a3ca5c24e9ea63ULL, 0x66d1466e9de10e64ULL, 0x19b0d87e16e25788ULL},
     {0x8f64842c55555533ULL, 0x8bc32d36fb21a6a3ULL, 0x425ed097b425ed09ULL, 0x1ed097b425ed097bULL},
     {0x58dfecce86b2745eULL, 0x06a767bfc35b5bacULL, 0x9e7eb64f890a820cULL, 0x2f44d6c801c1b8bfULL},
     {0xd43d449776f99d2fULL, 0x926847fb9ddd76a1ULL, 0x252659ba2b546c7eULL, 0x3d59f455cafc7668ULL},
     {0x8c46eb20fffffde5ULL, 0x224698fc0994a8ddULL, 0x0000000000000000ULL, 0x4000000000000000ULL}}};

template <class P>
static Fe<P> kg_const(const uint64_t* c) {
    Fe<P> x;
    for (int i = 0; i < 4; i++) { x.l[2 * i] = (uint32_t)c[i]; x.l[2 * i + 1] = (uint32_t)(c[i] >> 32); }
    return fe_to_mont<P>(x);
}
// ---- run-time parameters of from_label (round 6; include/lurk_hip.h: lurk_hip_ck_params) ----------------------------------------
// Like the random oracle's (transcript.hip): every constant of DlogGroup::from_label / hash_to_curve that was recalled from memory is a
// field, process-wide, defaults = what rounds 1-5 compiled in.
static lurk_hip_ck_params ck_params_default() {
    lurk_hip_ck_params p;
    memset(&p, 0, sizeof(p));
    p.struct_size = (uint32_t)sizeof(p);
    p.xof = LURK_CK_XOF_SHAKE256;
    p.bytes_per_point = 32;
    strcpy(p.domain_prefix, "from_uniform_bytes");
    strcpy(p.curve_name_pallas, "pallas");
    strcpy(p.curve_name_vesta, "vesta");
    strcpy(p.suite, "_XMD:BLAKE2b_SSWU_RO_");
    return p;
}
static std::mutex g_ck_mu;
static lurk_hip_ck_params g_ck = ck_params_default();
static lurk_hip_ck_params ck_params() {
    std::lock_guard<std::mutex> lk(g_ck_mu);
    return g_ck;
}
static bool c_string_within(const char* s, size_t cap) { return memchr(s, 0, cap) != nullptr; }
static std::string ck_dst(const lurk_hip_ck_params& p, int curve, const char* domain_prefix) {
    return std::string(domain_prefix) + "-" + (curve == 0 ? p.curve_name_pallas : p.curve_name_vesta) + p.suite;
}
static void ck_params_validate(const lurk_hip_ck_params& p) {
    LURK_REQUIRE(p.struct_size == sizeof(lurk_hip_ck_params), "lurk_hip_ck_params: struct_size does not match this library's layout");
    LURK_REQUIRE(p.xof == LURK_CK_XOF_SHAKE256 || p.xof == LURK_CK_XOF_SHAKE128, "lurk_hip_ck_params: unknown xof");
    LURK_REQUIRE(p.bytes_per_point >= 1 && p.bytes_per_point <= 64, "lurk_hip_ck_params: bytes_per_point must be in 1..64");
    LURK_REQUIRE(c_string_within(p.domain_prefix, sizeof(p.domain_prefix)) && c_string_within(p.curve_name_pallas, sizeof(p.curve_name_pallas)) &&
                     c_string_within(p.curve_name_vesta, sizeof(p.curve_name_vesta)) && c_string_within(p.suite, sizeof(p.suite)),
                 "lurk_hip_ck_params: a string is not NUL-terminated");
    for (int curve = 0; curve < 2; curve++)
        LURK_REQUIRE(131 + p.bytes_per_point + ck_dst(p, curve, p.domain_prefix).size() + 1 <= 256 && ck_dst(p, curve, p.domain_prefix).size() < 63,
                     "lurk_hip_ck_params: message + domain-separation tag do not fit two BLAKE2b blocks");
}

template <class P>
static KeygenConsts<P> make_keygen_consts(int curve, const char* domain_prefix, const lurk_hip_ck_params& prm) {
    KeygenConsts<P> K;
    K.a = kg_const<P>(KG_A[curve]);
    K.b = fe_from_u64<P>(1265);
    K.z = fe_neg<P>(fe_from_u64<P>(13));
    for (int i = 0; i < 13; i++) K.iso[i] = kg_const<P>(KG_ISO[curve][i]);
    K.r3 = fe_mul<P>(fe_r2<P>(), fe_r2<P>());  // R^2 R^2 / R
    // p - 1 = 2^32 T
    uint32_t pm1[8], T[8];
    for (int i = 0; i < 8; i++) pm1[i] = P::mod(i);
    pm1[0] -= 1;
    for (int i = 0; i < 8; i++) T[i] = i + 1 < 8 ? pm1[i + 1] : 0;  // >> 32 (the low word of p - 1 is zero for both Pasta primes)
    for (int i = 0; i < 8; i++) K.ts_exp[i] = (T[i] >> 1) | (i + 1 < 8 ? T[i + 1] << 31 : 0);  // (T - 1) / 2 = T >> 1 (T odd)
    for (int i = 0; i < 8; i++) K.legendre_exp[i] = (pm1[i] >> 1) | (i + 1 < 8 ? pm1[i + 1] << 31 : 0);
    Fe<P> g = fe_from_u64<P>(2);
    for (uint64_t v = 2;; v++) {  // smallest non-residue
        g = fe_from_u64<P>(v);
        if (!fe_eq<P>(fe_pow<P>(g, K.legendre_exp), fe_one<P>())) break;
    }
    K.ts_c = fe_pow<P>(g, T);
    const std::string dst = ck_dst(prm, curve, domain_prefix);
    LURK_REQUIRE(dst.size() < 63, "domain prefix too long");
    LURK_REQUIRE(131 + prm.bytes_per_point + dst.size() + 1 <= 256, "message + domain-separation tag do not fit two BLAKE2b blocks");
    memset(K.dst_prime, 0, sizeof(K.dst_prime));
    memcpy(K.dst_prime, dst.data(), dst.size());
    K.dst_prime[dst.size()] = (uint8_t)dst.size();
    K.dst_len = (uint32_t)dst.size() + 1;
    K.msg_len = prm.bytes_per_point;
    return K;
}

template <class P>
static void keygen_device(int curve, const char* domain, const void* d_uniform, size_t n, void* d_out, hipStream_t s) {
    if (n == 0) return;
    static_assert(P::ID != 2, "Pasta only");
    LURK_REQUIRE(P::mod(0) == 1u, "unexpected modulus");  // the low 32 bits of p - 1 are zero: 2-adicity 32
    const KeygenConsts<P> K = make_keygen_consts<P>(curve, domain, ck_params());
    ProfScope ps("keygen", s);
    hipLaunchKernelGGL((keygen_kernel<P>), dim3(div_up(n, 128)), dim3(128), 0, s, (const uint8_t*)d_uniform, n, K, (Affine<P>*)d_out);
    LURK_HIP_CHECK(hipGetLastError());
}

static std::vector<uint8_t> ck_xof_stream(const lurk_hip_ck_params& prm, const void* label, size_t label_len, size_t n) {
    std::vector<uint8_t> stream(n * prm.bytes_per_point);
    shake_xof(prm.xof == LURK_CK_XOF_SHAKE128 ? 168 : 136, (const uint8_t*)label, label_len, stream.data(), stream.size());
    return stream;
}

void keygen_from_label_device(int curve, const void* label, size_t label_len, size_t n, void* d_out, hipStream_t s) {
    const lurk_hip_ck_params prm = ck_params();
    const std::vector<uint8_t> stream = ck_xof_stream(prm, label, label_len, n);
    DevBuf d_u(stream.size());
    if (n) LURK_HIP_CHECK(hipMemcpyAsync(d_u.p, stream.data(), stream.size(), hipMemcpyHostToDevice, s));
    if (curve == LURK_CURVE_PALLAS) keygen_device<PallasFp>(curve, prm.domain_prefix, d_u.p, n, d_out, s);
    else keygen_device<PallasFq>(curve, prm.domain_prefix, d_u.p, n, d_out, s);
    LURK_HIP_CHECK(hipStreamSynchronize(s));  // the staging buffers go out of scope
}

// the same map on the host, one point after the other (~0.3 ms each): the first points of a key for a probe record or a CPU test
template <class P>
static void keygen_from_label_host(int curve, const void* label, size_t label_len, size_t n, void* out) {
    const lurk_hip_ck_params prm = ck_params();
    const std::vector<uint8_t> stream = ck_xof_stream(prm, label, label_len, n);
    const KeygenConsts<P> K = make_keygen_consts<P>(curve, prm.domain_prefix, prm);
    for (size_t i = 0; i < n; i++) {
        const Affine<P> pt = keygen_point<P>(stream.data() + i * prm.bytes_per_point, K);
        memcpy((char*)out + 64 * i, &pt, 64);
    }
}

}  // namespace lurk

using namespace lurk;

extern "C" {

int lurk_hip_shake256(const void* in, size_t in_len, void* out, size_t out_len) {
    // pure host computation (the CPU tests compare it with hashlib)
    try {
        LURK_REQUIRE((in || in_len == 0) && (out || out_len == 0), "null buffer");
        shake256((const uint8_t*)in, in_len, (uint8_t*)out, out_len);
        set_error(0, "");
        return 0;
    } catch (const HipFailure& e) {
        set_error(e.code, e.msg);
        return e.code;
    }
}

int lurk_hip_ck_params_get(lurk_hip_ck_params* out) {
    try {
        LURK_REQUIRE(out, "null argument");
        *out = ck_params();
        set_error(0, "");
        return 0;
    } catch (const HipFailure& e) {
        set_error(e.code, e.msg);
        return e.code;
    }
}
int lurk_hip_ck_params_set(const lurk_hip_ck_params* params) {
    try {
        const lurk_hip_ck_params p = params ? *params : ck_params_default();  // NULL: back to the defaults
        ck_params_validate(p);
        std::lock_guard<std::mutex> lk(g_ck_mu);
        g_ck = p;
        set_error(0, "");
        return 0;
    } catch (const HipFailure& e) {
        set_error(e.code, e.msg);
        return e.code;
    }
}

int lurk_hip_ck_from_label_host(int curve, const void* label, size_t label_len, size_t npoints, void* out_affine64) {
    // pure host computation: no device is needed
    try {
        LURK_REQUIRE(curve == LURK_CURVE_PALLAS || curve == LURK_CURVE_VESTA, "unknown curve id");
        LURK_REQUIRE((label || label_len == 0) && (npoints == 0 || out_affine64), "null argument");
        LURK_REQUIRE(npoints <= ((size_t)1 << 16), "the host form maps at most 2^16 points: build keys with lurk_hip_ck_from_label_dev / lurk_hip_msm_ctx_from_label");
        if (curve == LURK_CURVE_PALLAS) keygen_from_label_host<PallasFp>(curve, label, label_len, npoints, out_affine64);
        else keygen_from_label_host<PallasFq>(curve, label, label_len, npoints, out_affine64);
        set_error(0, "");
        return 0;
    } catch (const HipFailure& e) {
        set_error(e.code, e.msg);
        return e.code;
    }
}

int lurk_hip_ck_hash_to_curve_dev(int curve, const char* domain_prefix, const void* d_uniform32, size_t n, void* d_out_affine64, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(curve == LURK_CURVE_PALLAS || curve == LURK_CURVE_VESTA, "unknown curve id");
        LURK_REQUIRE(domain_prefix && (n == 0 || (d_uniform32 && d_out_affine64)), "null argument");
        if (curve == LURK_CURVE_PALLAS) keygen_device<PallasFp>(curve, domain_prefix, d_uniform32, n, d_out_affine64, (hipStream_t)stream);
        else keygen_device<PallasFq>(curve, domain_prefix, d_uniform32, n, d_out_affine64, (hipStream_t)stream);
    });
}

int lurk_hip_ck_from_label_dev(int curve, const void* label, size_t label_len, size_t npoints, void* d_out_affine64, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(curve == LURK_CURVE_PALLAS || curve == LURK_CURVE_VESTA, "unknown curve id");
        LURK_REQUIRE((label || label_len == 0) && (npoints == 0 || d_out_affine64), "null argument");
        keygen_from_label_device(curve, label, label_len, npoints, d_out_affine64, (hipStream_t)stream);
    });
}
}
