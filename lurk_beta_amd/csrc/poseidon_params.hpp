// poseidon_params.hpp - host-side generation of neptune-compatible Poseidon parameters and of
// the sparse-round ("optimized") schedule the gfx950 kernel runs.  Product code (host half of
// liblurk_hip.so); independent of oracle/ - tests compare the two.
//
// What it reproduces (reference boundary: PoseidonConstants::new() in
// /root/reference/src/hash.rs:59-84, consumed by Poseidon::new_with_preimage(..).hash() at
// /root/reference/src/hash.rs:181-203):
//   * round numbers: security level 128, field size fixed at 256 bits in the inequalities,
//     +2 full rounds and x1.075 partial-round margin (neptune's published rule)
//   * round constants: Grain LFSR seeded with (field=1, sbox=1, n=F::NUM_BITS, t, R_F, R_P)
//   * MDS: Cauchy 1/(x_i + y_j), x_i = i, y_j = t + j
//   * domain tag 2^arity - 1 (HashType::MerkleTree), digest = state[1]
//
// Sparse schedule (own derivation; algebraically identical to the plain schedule, so digests are
// bit-identical):  write a dense matrix N = [[n00, nv],[nw, Nh]] as N = S * A with
//   S = [[n00, nv*Nh^-1],[nw, I]]   (2t-1 non-trivial entries)      A = diag(1, Nh).
// A commutes with the partial S-box (coordinate 0 is neither changed nor mixed by A), so working
// backwards from the last partial round each round's dense matrix is replaced by a sparse S_p and
// the leftover A_p is pushed into the previous round; what reaches the last full round of the
// first half is the dense "pre-sparse" matrix A_0*M.  Round constants of partial rounds are pushed
// forward the same way so that only coordinate 0 receives a constant inside the partial rounds.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "field.cuh"

namespace lurk {

struct RoundNumbers {
    int rf, rp;
};

inline bool round_numbers_are_secure(int t, int rf, int rp) {
    const float n = 256.0f, m = 128.0f;
    float tf = (float)t, rpf = (float)rp;
    float rf_stat = (m <= (n - 3.0f) * (tf + 1.0f)) ? 6.0f : 10.0f;
    float rf_interp = 0.43f * m + std::log2(tf) - rpf;
    float rf_grob_1 = 0.21f * n - rpf;
    float rf_grob_2 = (0.14f * n - 1.0f - rpf) / (tf - 1.0f);
    float mx = std::fmax(std::fmax(std::ceil(rf_stat), std::ceil(rf_interp)), std::fmax(std::ceil(rf_grob_1), std::ceil(rf_grob_2)));
    return (float)rf >= mx;
}

inline RoundNumbers poseidon_round_numbers(int arity) {
    int t = arity + 1;
    int best_rf = 0, best_rp = 0;
    long best = -1;
    for (int rf0 = 2; rf0 <= 1000; rf0 += 2) {
        for (int rp0 = 4; rp0 < 200; rp0++) {
            if (!round_numbers_are_secure(t, rf0, rp0)) continue;
            int rf = rf0 + 2;
            int rp = (int)std::ceil(1.075f * (float)rp0);
            long n_sboxes = (long)t * rf + rp;
            if (best < 0 || n_sboxes < best || (n_sboxes == best && rf < best_rf)) {
                best_rf = rf;
                best_rp = rp;
                best = n_sboxes;
            }
        }
    }
    return {best_rf, best_rp};
}

class GrainLfsr {
  public:
    GrainLfsr(int field_bits, int t, int rf, int rp) : field_bits_(field_bits) {
        int pos = 0;
        auto put = [&](int nbits, uint32_t v) {
            for (int i = nbits - 1; i >= 0; i--) st_[pos++] = (v >> i) & 1;
        };
        put(2, 1);  // prime field
        put(4, 1);  // sbox
        put(12, (uint32_t)field_bits);
        put(12, (uint32_t)t);
        put(10, (uint32_t)rf);
        put(10, (uint32_t)rp);
        put(30, 0x3fffffffu);
        for (int i = 0; i < 160; i++) step();
    }
    // next candidate as 8 x 32-bit LE limbs of the big-endian bit string (first partial byte first)
    void next_candidate(uint32_t* limbs) {
        for (int i = 0; i < 8; i++) limbs[i] = 0;
        int rem = field_bits_ % 8;
        int bitpos = 255;  // bit index (from lsb) where the next generated bit goes
        int first = rem ? rem : 8;
        bitpos = 31 * 8 + first - 1;
        for (int i = 0; i < first + 31 * 8; i++) {
            if (bit()) limbs[bitpos >> 5] |= 1u << (bitpos & 31);
            bitpos--;
        }
    }

  private:
    int step() {
        int b = st_[(head_ + 62) % 80] ^ st_[(head_ + 51) % 80] ^ st_[(head_ + 38) % 80] ^ st_[(head_ + 23) % 80] ^
                st_[(head_ + 13) % 80] ^ st_[head_];
        st_[head_] = (uint8_t)b;
        head_ = (head_ + 1) % 80;
        return b;
    }
    int bit() {  // shrinking generator: keep the second bit of a pair iff the first is 1
        int b = step();
        while (!b) {
            step();
            b = step();
        }
        return step();
    }
    uint8_t st_[80];
    int head_ = 0;
    int field_bits_;
};

template <class P>
struct PoseidonParams {
    int arity, t, rf, rp;
    Fe<P> domain_tag;                // Montgomery
    std::vector<Fe<P>> rc;           // (rf+rp)*t plain round constants, Montgomery
    std::vector<Fe<P>> mds;          // t*t, row-major, Montgomery
    // sparse schedule
    std::vector<Fe<P>> pre_sparse;   // t*t dense matrix used by the last full round of the first half
    std::vector<Fe<P>> sparse;       // rp * (2t-1): [n00, vhat_1..vhat_{t-1}, nw_1..nw_{t-1}] per round
    std::vector<Fe<P>> partial_k;    // rp scalars added to coordinate 0
    std::vector<Fe<P>> rc_after;     // t constants of the first full round after the partial rounds
};

template <class P>
static std::vector<Fe<P>> mat_mul(const std::vector<Fe<P>>& a, const std::vector<Fe<P>>& b, int n) {
    std::vector<Fe<P>> c(n * n, fe_zero<P>());
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            Fe<P> acc = fe_zero<P>();
            for (int k = 0; k < n; k++) acc = fe_add<P>(acc, fe_mul<P>(a[i * n + k], b[k * n + j]));
            c[i * n + j] = acc;
        }
    return c;
}
template <class P>
static std::vector<Fe<P>> mat_inv(std::vector<Fe<P>> a, int n) {  // Gauss-Jordan
    std::vector<Fe<P>> inv(n * n, fe_zero<P>());
    for (int i = 0; i < n; i++) inv[i * n + i] = fe_one<P>();
    for (int col = 0; col < n; col++) {
        int piv = col;
        while (piv < n && fe_is_zero<P>(a[piv * n + col])) piv++;
        if (piv == n) return {};
        if (piv != col)
            for (int j = 0; j < n; j++) {
                std::swap(a[piv * n + j], a[col * n + j]);
                std::swap(inv[piv * n + j], inv[col * n + j]);
            }
        Fe<P> d = fe_inv<P>(a[col * n + col]);
        for (int j = 0; j < n; j++) {
            a[col * n + j] = fe_mul<P>(a[col * n + j], d);
            inv[col * n + j] = fe_mul<P>(inv[col * n + j], d);
        }
        for (int r = 0; r < n; r++) {
            if (r == col || fe_is_zero<P>(a[r * n + col])) continue;
            Fe<P> f = a[r * n + col];
            for (int j = 0; j < n; j++) {
                a[r * n + j] = fe_sub<P>(a[r * n + j], fe_mul<P>(f, a[col * n + j]));
                inv[r * n + j] = fe_sub<P>(inv[r * n + j], fe_mul<P>(f, inv[col * n + j]));
            }
        }
    }
    return inv;
}

template <class P>
PoseidonParams<P> make_poseidon_params(int arity) {
    PoseidonParams<P> pp;
    pp.arity = arity;
    const int t = pp.t = arity + 1;
    RoundNumbers rn = poseidon_round_numbers(arity);
    pp.rf = rn.rf;
    pp.rp = rn.rp;
    pp.domain_tag = fe_from_u64<P>(((uint64_t)1 << arity) - 1);

    GrainLfsr g(P::NBITS, t, pp.rf, pp.rp);
    const int nconst = (pp.rf + pp.rp) * t;
    while ((int)pp.rc.size() < nconst) {
        Fe<P> c;
        g.next_candidate(c.l);
        if (!fe_canonical_ge_mod<P>(c.l)) pp.rc.push_back(fe_to_mont<P>(c));
    }
    pp.mds.resize(t * t);
    for (int i = 0; i < t; i++)
        for (int j = 0; j < t; j++) pp.mds[i * t + j] = fe_inv<P>(fe_from_u64<P>((uint64_t)(i + t + j)));

    // ---- sparse schedule ------------------------------------------------------------------
    const int h = pp.rf / 2, n1 = t - 1;
    pp.sparse.assign((size_t)pp.rp * (2 * t - 1), fe_zero<P>());
    std::vector<std::vector<Fe<P>>> A_hat(pp.rp);  // Nh_p, (t-1)x(t-1)
    std::vector<Fe<P>> N = pp.mds;                 // dense matrix of the round being factored
    for (int p = pp.rp - 1; p >= 0; p--) {
        std::vector<Fe<P>> Nh(n1 * n1);
        for (int i = 0; i < n1; i++)
            for (int j = 0; j < n1; j++) Nh[i * n1 + j] = N[(i + 1) * t + (j + 1)];
        std::vector<Fe<P>> Nh_inv = mat_inv<P>(Nh, n1);
        Fe<P>* sp = &pp.sparse[(size_t)p * (2 * t - 1)];
        sp[0] = N[0];
        for (int j = 0; j < n1; j++) {  // vhat = nv * Nh^-1 (row vector)
            Fe<P> acc = fe_zero<P>();
            for (int k = 0; k < n1; k++) acc = fe_add<P>(acc, fe_mul<P>(N[0 * t + (k + 1)], Nh_inv[k * n1 + j]));
            sp[1 + j] = acc;
        }
        for (int i = 0; i < n1; i++) sp[t + i] = N[(i + 1) * t + 0];
        A_hat[p] = Nh;
        // matrix of the previous round: A_p * M, A_p = diag(1, Nh)
        std::vector<Fe<P>> A(t * t, fe_zero<P>());
        A[0] = fe_one<P>();
        for (int i = 0; i < n1; i++)
            for (int j = 0; j < n1; j++) A[(i + 1) * t + (j + 1)] = Nh[i * n1 + j];
        N = mat_mul<P>(A, pp.mds, t);
    }
    pp.pre_sparse = N;  // A_0 * M

    // constants: d_p = A_p c_p ; e_p = g_p + d_p ; k_p = e_p[0] ; g_{p+1} = S_p * (e_p with coord 0 zeroed)
    std::vector<Fe<P>> gvec(t, fe_zero<P>());
    pp.partial_k.resize(pp.rp);
    for (int p = 0; p < pp.rp; p++) {
        const Fe<P>* c = &pp.rc[(size_t)(h + p) * t];
        std::vector<Fe<P>> e(t);
        e[0] = fe_add<P>(gvec[0], c[0]);
        for (int i = 0; i < n1; i++) {
            Fe<P> acc = fe_zero<P>();
            for (int j = 0; j < n1; j++) acc = fe_add<P>(acc, fe_mul<P>(A_hat[p][i * n1 + j], c[j + 1]));
            e[i + 1] = fe_add<P>(gvec[i + 1], acc);
        }
        pp.partial_k[p] = e[0];
        const Fe<P>* sp = &pp.sparse[(size_t)p * (2 * t - 1)];
        // S_p * (0, e_rest): new_0 = vhat . e_rest ; new_i = e_i
        Fe<P> acc = fe_zero<P>();
        for (int j = 0; j < n1; j++) acc = fe_add<P>(acc, fe_mul<P>(sp[1 + j], e[j + 1]));
        gvec[0] = acc;
        for (int i = 0; i < n1; i++) gvec[i + 1] = e[i + 1];
    }
    pp.rc_after.resize(t);
    for (int i = 0; i < t; i++) pp.rc_after[i] = fe_add<P>(pp.rc[(size_t)(h + pp.rp) * t + i], gvec[i]);
    return pp;
}

// neptune's `compressed_round_constants` re-expressed as ONE post-S-box key per S-box in circuit order (what
// circuit2::poseidon_hash_allocated adds to l^5 before it allocates the S-box output): S-box
// number r*t + e for element e of full round r of the first half, then one per partial round, then the second half;
// zero for the last round (no key).  Published rule (neptune preprocessing.rs): the constants of round r+1 are pulled
// back through M^-1 to act after the S-boxes of round r; through the partial rounds only coordinate 0's key stays in
// its round and the rest moves one round earlier.  The first round's own constants are added BEFORE its S-boxes (they
// are rc[0..t), already in the image).  The oracle restates the same rule independently (oracle/circuit_ref.py).
template <class P>
std::vector<Fe<P>> neptune_post_keys(const PoseidonParams<P>& pp) {
    const int t = pp.t, h = pp.rf / 2, rp = pp.rp;
    const std::vector<Fe<P>> minv = mat_inv<P>(pp.mds, t);
    auto pull = [&](const std::vector<Fe<P>>& v) {  // v * M^-1 (M is symmetric: the same as M^-1 * v)
        std::vector<Fe<P>> o(t, fe_zero<P>());
        for (int j = 0; j < t; j++)
            for (int i = 0; i < t; i++) o[j] = fe_add<P>(o[j], fe_mul<P>(v[i], minv[i * t + j]));
        return o;
    };
    auto keys = [&](int r) { return std::vector<Fe<P>>(pp.rc.begin() + (size_t)r * t, pp.rc.begin() + (size_t)(r + 1) * t); };
    std::vector<Fe<P>> post;
    for (int r = 0; r + 1 < h; r++) {
        auto k = pull(keys(r + 1));
        post.insert(post.end(), k.begin(), k.end());
    }
    std::vector<Fe<P>> partial(rp);
    std::vector<Fe<P>> acc = keys(h + rp);
    for (int i = 0; i < rp; i++) {
        auto inv = pull(acc);
        partial[rp - 1 - i] = inv[0];
        inv[0] = fe_zero<P>();
        auto prev = keys(h + rp - i - 1);
        for (int j = 0; j < t; j++) acc[j] = fe_add<P>(prev[j], inv[j]);
    }
    {
        auto k = pull(acc);  // keys of the last full round of the first half
        post.insert(post.end(), k.begin(), k.end());
    }
    post.insert(post.end(), partial.begin(), partial.end());
    for (int r = 1; r < h; r++) {
        auto k = pull(keys(h + rp + r));
        post.insert(post.end(), k.begin(), k.end());
    }
    post.resize((size_t)t * pp.rf + rp, fe_zero<P>());  // the last round adds nothing
    return post;
}

// Flat device image of the sparse schedule, as uint32 words (8 per element):
//   [0]                      domain tag
//   [1 .. 1+h*t)             round constants of full rounds 0..h-1
//   then t*t                 MDS (row-major)
//   then t*t                 pre-sparse matrix
//   then rp                  partial_k
//   then rp*(2t-1)           sparse rounds
//   then t                   rc_after
//   then (h-1)*t             round constants of the remaining full rounds
template <class P>
std::vector<uint32_t> poseidon_device_image(const PoseidonParams<P>& pp) {
    std::vector<uint32_t> img;
    auto push = [&](const Fe<P>& x) {
        for (int i = 0; i < 8; i++) img.push_back(x.l[i]);
    };
    const int t = pp.t, h = pp.rf / 2;
    push(pp.domain_tag);
    for (int i = 0; i < h * t; i++) push(pp.rc[i]);
    for (auto& x : pp.mds) push(x);
    for (auto& x : pp.pre_sparse) push(x);
    for (auto& x : pp.partial_k) push(x);
    for (auto& x : pp.sparse) push(x);
    for (auto& x : pp.rc_after) push(x);
    for (int i = (h + pp.rp + 1) * t; i < (pp.rf + pp.rp) * t; i++) push(pp.rc[i]);
    return img;
}

}  // namespace lurk
