/* msm_fast.c - the CPU baseline of the Pedersen MSM: a Pippenger in the shape of pasta-msm's (blst-derived) CPU path.
 *
 * TEST / BENCH INFRASTRUCTURE ONLY (see oracle/pyref.py's header): bench.py's `cpu_baseline` leg times it and the CPU
 * tests check it against oracle.c's Pippenger and naive double-and-add.  It is a PORT ("kind": "port"), not the reference:
 * pasta-msm (C + assembly, /root/reference/Cargo.toml:68 via arecibo) is not in /root/reference and there is no Rust / cargo
 * here.  What it restates of that path, so that the number it produces is a fair stand-in for "the reference's rayon CPU path":
 *   - 4 x 64-bit Montgomery multiplication on mulx/adcx-class code (BMI2/ADX; gcc emits mulx + adc chains for the
 *     product-scanning form below), with the Pasta moduli's zero limb skipped;
 *   - Booth-recoded signed windows (half the buckets), window size chosen from n and the thread count;
 *   - buckets in XYZZ coordinates with mixed additions (8M + 2S), running-sum bucket reduction;
 *   - work split over (window, point-chunk) tiles on all cores, as blst's pippenger tiles its work.
 * oracle.c's orc_msm_pippenger stays as the independent, slower statement (halo2 best_multiexp shape, Jacobian, unsigned
 * windows); both must agree bit for bit (tests/test_oracle_c.py). */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fe;
typedef struct { fe x, y; } aff;           /* Montgomery, identity = (0, 0) */
typedef struct { fe x, y, zz, zzz; } xyzz; /* identity: zz = 0 */
typedef struct { fe x, y, z; } jac;
typedef struct { uint64_t m[4], inv; fe one; } fld;

static const uint64_t MODS[2][4] = {
    {0x992d30ed00000001ULL, 0x224698fc094cf91bULL, 0x0ULL, 0x4000000000000000ULL}, /* Pallas Fp (Pallas base field) */
    {0x8c46eb2100000001ULL, 0x224698fc0994a8ddULL, 0x0ULL, 0x4000000000000000ULL}, /* Pallas Fq (Vesta base field) */
};
static fld FL[2];
static int fl_init_done = 0;

static inline int ge4(const uint64_t *a, const uint64_t *b) {
    for (int i = 3; i >= 0; i--) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; }
    return 1;
}
static inline void sub4(uint64_t *r, const uint64_t *a, const uint64_t *b) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - b[i] - br; r[i] = (uint64_t)d; br = (d >> 64) & 1; }
}
static inline void f_add(const fld *f, fe *r, const fe *a, const fe *b) {
    u128 c = 0; uint64_t t[4];
    for (int i = 0; i < 4; i++) { c += (u128)a->v[i] + b->v[i]; t[i] = (uint64_t)c; c >>= 64; }
    if (ge4(t, f->m)) sub4(r->v, t, f->m); else memcpy(r->v, t, 32); /* operands < p < 2^255: no carry out */
}
static inline void f_sub(const fld *f, fe *r, const fe *a, const fe *b) {
    u128 br = 0; uint64_t t[4];
    for (int i = 0; i < 4; i++) { u128 d = (u128)a->v[i] - b->v[i] - br; t[i] = (uint64_t)d; br = (d >> 64) & 1; }
    if (br) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)t[i] + f->m[i]; t[i] = (uint64_t)c; c >>= 64; } }
    memcpy(r->v, t, 32);
}
static inline int f_is0(const fe *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static inline int f_eq(const fe *a, const fe *b) { return ((a->v[0] ^ b->v[0]) | (a->v[1] ^ b->v[1]) | (a->v[2] ^ b->v[2]) | (a->v[3] ^ b->v[3])) == 0; }

/* Montgomery product (R = 2^256), coarsely integrated operand scanning; the moduli have m[2] = 0 and m[3] = 2^62 */
static inline void f_mul(const fld *f, fe *r, const fe *a, const fe *b) {
    const uint64_t a0 = a->v[0], a1 = a->v[1], a2 = a->v[2], a3 = a->v[3];
    const uint64_t p0 = f->m[0], p1 = f->m[1], inv = f->inv;
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    for (int i = 0; i < 4; i++) {
        const uint64_t bi = b->v[i];
        u128 c;
        c = (u128)a0 * bi + t0; t0 = (uint64_t)c;
        c = (u128)a1 * bi + t1 + (uint64_t)(c >> 64); t1 = (uint64_t)c;
        c = (u128)a2 * bi + t2 + (uint64_t)(c >> 64); t2 = (uint64_t)c;
        c = (u128)a3 * bi + t3 + (uint64_t)(c >> 64); t3 = (uint64_t)c;
        t4 += (uint64_t)(c >> 64);
        const uint64_t m = t0 * inv;
        c = (u128)m * p0 + t0;
        c = (u128)m * p1 + t1 + (uint64_t)(c >> 64); t0 = (uint64_t)c;
        c = (u128)t2 + (uint64_t)(c >> 64); t1 = (uint64_t)c;
        c = ((u128)m << 62) + t3 + (uint64_t)(c >> 64); t2 = (uint64_t)c;
        c = (u128)t4 + (uint64_t)(c >> 64); t3 = (uint64_t)c; t4 = (uint64_t)(c >> 64);
    }
    uint64_t t[4] = {t0, t1, t2, t3};
    if (t4 || ge4(t, f->m)) sub4(r->v, t, f->m); else memcpy(r->v, t, 32);
}
static inline void f_sqr(const fld *f, fe *r, const fe *a) { f_mul(f, r, a, a); }
static void f_inv(const fld *f, fe *r, const fe *a) { /* a^(p-2) */
    uint64_t e[4] = {f->m[0] - 2, f->m[1], f->m[2], f->m[3]};
    fe acc = f->one, base = *a;
    for (int i = 0; i < 256; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) f_mul(f, &acc, &acc, &base);
        f_sqr(f, &base, &base);
    }
    *r = acc;
}
static void fl_init(void) {
    if (fl_init_done) return;
    for (int k = 0; k < 2; k++) {
        fld *f = &FL[k];
        memcpy(f->m, MODS[k], 32);
        uint64_t x = 1;
        for (int i = 0; i < 7; i++) x *= 2 - f->m[0] * x;
        f->inv = (uint64_t)0 - x;
        fe acc = {{1, 0, 0, 0}};
        for (int i = 0; i < 256; i++) f_add(f, &acc, &acc, &acc);
        f->one = acc;
    }
    fl_init_done = 1;
}

/* ---- XYZZ group law (a = 0 curves) ---- */
static inline void x_dbl_affine(const fld *f, xyzz *r, const fe *x1, const fe *y1) { /* 2 (x1, y1) */
    fe u, v, w, s, m, t;
    f_add(f, &u, y1, y1); f_sqr(f, &v, &u); f_mul(f, &w, &u, &v); f_mul(f, &s, x1, &v);
    f_sqr(f, &m, x1); f_add(f, &t, &m, &m); f_add(f, &m, &t, &m);
    f_sqr(f, &r->x, &m); f_sub(f, &r->x, &r->x, &s); f_sub(f, &r->x, &r->x, &s);
    f_sub(f, &t, &s, &r->x); f_mul(f, &t, &m, &t); f_mul(f, &u, &w, y1); f_sub(f, &r->y, &t, &u);
    r->zz = v; r->zzz = w;
}
static inline void x_dbl(const fld *f, xyzz *r, const xyzz *p) {
    if (f_is0(&p->zz)) { *r = *p; return; }
    fe u, v, w, s, m, t;
    f_add(f, &u, &p->y, &p->y); f_sqr(f, &v, &u); f_mul(f, &w, &u, &v); f_mul(f, &s, &p->x, &v);
    f_sqr(f, &m, &p->x); f_add(f, &t, &m, &m); f_add(f, &m, &t, &m);
    fe x3, y3;
    f_sqr(f, &x3, &m); f_sub(f, &x3, &x3, &s); f_sub(f, &x3, &x3, &s);
    f_sub(f, &t, &s, &x3); f_mul(f, &t, &m, &t); f_mul(f, &u, &w, &p->y); f_sub(f, &y3, &t, &u);
    f_mul(f, &r->zz, &v, &p->zz); f_mul(f, &r->zzz, &w, &p->zzz);
    r->x = x3; r->y = y3;
}
static inline void x_madd(const fld *f, xyzz *acc, const aff *q, int neg) { /* acc += (+/-) q */
    if (f_is0(&q->x) && f_is0(&q->y)) return;
    fe qy = q->y;
    if (neg) { fe z = {{0, 0, 0, 0}}; f_sub(f, &qy, &z, &qy); }
    if (f_is0(&acc->zz)) { acc->x = q->x; acc->y = qy; acc->zz = f->one; acc->zzz = f->one; return; }
    fe u2, s2, p, r, pp, ppp, qq, t;
    f_mul(f, &u2, &q->x, &acc->zz); f_mul(f, &s2, &qy, &acc->zzz);
    f_sub(f, &p, &u2, &acc->x); f_sub(f, &r, &s2, &acc->y);
    if (f_is0(&p)) {
        if (f_is0(&r)) x_dbl_affine(f, acc, &q->x, &qy); else memset(acc, 0, sizeof(*acc));
        return;
    }
    f_sqr(f, &pp, &p); f_mul(f, &ppp, &p, &pp); f_mul(f, &qq, &acc->x, &pp);
    fe x3; f_sqr(f, &x3, &r); f_sub(f, &x3, &x3, &ppp); f_sub(f, &x3, &x3, &qq); f_sub(f, &x3, &x3, &qq);
    f_sub(f, &t, &qq, &x3); f_mul(f, &t, &r, &t); f_mul(f, &u2, &acc->y, &ppp); f_sub(f, &acc->y, &t, &u2);
    acc->x = x3;
    f_mul(f, &acc->zz, &acc->zz, &pp); f_mul(f, &acc->zzz, &acc->zzz, &ppp);
}
static inline void x_add(const fld *f, xyzz *acc, const xyzz *q) {
    if (f_is0(&q->zz)) return;
    if (f_is0(&acc->zz)) { *acc = *q; return; }
    fe u1, u2, s1, s2, p, r, pp, ppp, qq, t;
    f_mul(f, &u1, &acc->x, &q->zz); f_mul(f, &u2, &q->x, &acc->zz);
    f_mul(f, &s1, &acc->y, &q->zzz); f_mul(f, &s2, &q->y, &acc->zzz);
    f_sub(f, &p, &u2, &u1); f_sub(f, &r, &s2, &s1);
    if (f_is0(&p)) {
        if (f_is0(&r)) x_dbl(f, acc, acc); else memset(acc, 0, sizeof(*acc));
        return;
    }
    f_sqr(f, &pp, &p); f_mul(f, &ppp, &p, &pp); f_mul(f, &qq, &u1, &pp);
    fe x3; f_sqr(f, &x3, &r); f_sub(f, &x3, &x3, &ppp); f_sub(f, &x3, &x3, &qq); f_sub(f, &x3, &x3, &qq);
    f_sub(f, &t, &qq, &x3); f_mul(f, &t, &r, &t); f_mul(f, &u2, &s1, &ppp); f_sub(f, &acc->y, &t, &u2);
    acc->x = x3;
    f_mul(f, &t, &acc->zz, &q->zz); f_mul(f, &acc->zz, &t, &pp);
    f_mul(f, &t, &acc->zzz, &q->zzz); f_mul(f, &acc->zzz, &t, &ppp);
}

/* Booth digit of window w (c bits) of a canonical 256-bit scalar: value in [-2^(c-1), 2^(c-1)] */
static inline int booth_digit(const uint64_t *k, unsigned w, unsigned c) {
    /* bits [w*c - 1, w*c + c): digit = (bits >> 1 rounded) with the standard recoding (v + 1) >> 1 - sign */
    const int off = (int)(w * c) - 1;
    uint64_t v;
    if (off < 0) {
        v = k[0] << 1;  /* bit -1 is zero */
    } else if (off >= 256) {
        v = 0;
    } else {
        const int limb = off >> 6, sh = off & 63;
        v = k[limb] >> sh;
        if (sh && limb + 1 < 4) v |= k[limb + 1] << (64 - sh);
    }
    v &= ((uint64_t)2 << c) - 1;  /* c + 1 bits */
    const int sign = (int)((v >> c) & 1);
    int d = (int)((v + 1) >> 1);
    if (sign) d -= (1 << c);  /* d in [-(2^(c-1)), 2^(c-1)] */
    return d;
}

static void tile(const fld *f, const aff *bases, const uint64_t *scalars, size_t lo, size_t hi, unsigned w, unsigned c, xyzz *buckets, xyzz *out) {
    const size_t nb = (size_t)1 << (c - 1);
    memset(buckets, 0, sizeof(xyzz) * nb);
    for (size_t i = lo; i < hi; i++) {
        const int d = booth_digit(scalars + 4 * i, w, c);
        if (d > 0) x_madd(f, &buckets[d - 1], &bases[i], 0);
        else if (d < 0) x_madd(f, &buckets[-d - 1], &bases[i], 1);
    }
    xyzz run, sum;
    memset(&run, 0, sizeof(run)); memset(&sum, 0, sizeof(sum));
    for (size_t b = nb; b-- > 0;) { x_add(f, &run, &buckets[b]); x_add(f, &sum, &run); }
    *out = sum;
}

/* out: Jacobian Montgomery (x, y, z) with z = 1 (or all zero for the identity).  bases: affine Montgomery; scalars: canonical. */
void orc_msm_fast(int curve, const uint64_t *bases_u64, const uint64_t *scalars, size_t n, int nthreads, uint64_t *out, int *c_used, int *tiles_used) {
    fl_init();
    const fld *f = &FL[curve == 0 ? 0 : 1];
    const aff *bases = (const aff *)bases_u64;
    if (nthreads < 1) nthreads = 1;
    /* window size and point-chunks per window: minimise rounds x tile cost (10 mults per mixed add, 28 per bucket in the reduction) */
    unsigned best_c = 8, best_j = 1;
    double best = 1e300;
    for (unsigned c = 6; c <= 18; c++) {
        const unsigned W = (256 + c) / c;  /* Booth needs the carry bit: ceil(257 / c) */
        for (unsigned j = 1; j <= (unsigned)nthreads; j++) {
            const double item = (double)n / j * 10.0 + (double)((size_t)1 << (c - 1)) * 28.0;
            const double rounds = ceil((double)(W * j) / nthreads);
            if (rounds * item < best) { best = rounds * item; best_c = c; best_j = j; }
        }
    }
    const unsigned c = best_c, J = n < best_j ? (n ? (unsigned)n : 1) : best_j, W = (256 + c) / c;
    if (c_used) *c_used = (int)c;
    if (tiles_used) *tiles_used = (int)(W * J);
    xyzz *res = calloc((size_t)W * J, sizeof(xyzz));
    const size_t chunk = (n + J - 1) / J;
#pragma omp parallel num_threads(nthreads)
    {
        xyzz *buckets = malloc(sizeof(xyzz) << (c - 1));
#pragma omp for schedule(dynamic, 1)
        for (int t = 0; t < (int)(W * J); t++) {
            const unsigned w = (unsigned)t / J, j = (unsigned)t % J;
            const size_t lo = (size_t)j * chunk, hi = lo + chunk > n ? n : lo + chunk;
            if (lo < hi) tile(f, bases, scalars, lo, hi, w, c, buckets, &res[t]);
        }
        free(buckets);
    }
    xyzz total;
    memset(&total, 0, sizeof(total));
    for (int w = (int)W - 1; w >= 0; w--) {
        for (unsigned d = 0; d < c; d++) x_dbl(f, &total, &total);
        for (unsigned j = 0; j < J; j++) x_add(f, &total, &res[(size_t)w * J + j]);
    }
    free(res);
    jac o;
    memset(&o, 0, sizeof(o));
    if (!f_is0(&total.zz)) { /* x = X / ZZ, y = Y / ZZZ */
        fe izzz, t, izz;
        f_inv(f, &izzz, &total.zzz);
        f_mul(f, &t, &total.zz, &izzz); f_sqr(f, &izz, &t);
        f_mul(f, &o.x, &total.x, &izz); f_mul(f, &o.y, &total.y, &izzz);
        o.z = f->one;
    }
    memcpy(out, &o, sizeof(o));
}

/* single-thread field-multiplication rate (ns per Montgomery product): the per-core figure the MSM number rests on */
double orc_fast_mul_ns(int curve, int iters) {
    fl_init();
    const fld *f = &FL[curve == 0 ? 0 : 1];
    fe a = f->one, b = {{0x1234567, 0x89abcdef, 0x55, 0x1}};
    const double t0 = omp_get_wtime();
    for (int i = 0; i < iters; i++) { f_mul(f, &a, &a, &b); f_mul(f, &b, &b, &a); }
    const double dt = omp_get_wtime() - t0;
    volatile uint64_t sink = a.v[0] ^ b.v[0];
    (void)sink;
    return dt / (2.0 * iters) * 1e9;
}
