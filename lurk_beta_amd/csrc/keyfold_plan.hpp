// keyfold_plan.hpp - host side of the key fold (ipa.hip: key_fold): the fold weights' signed sub-digits, slot by slot, sorted by magnitude.
// Plain C++ (no HIP): ipa.hip builds the plan and uploads it, tests/host_harness checks it on the CPU.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace lurk {

constexpr int KEYFOLD_MAX_T = 1 << 12;

struct KeyFoldPlan {
    bool ok = true;                     // false: a weight's signed digits carry out of the windows (never for a canonical field element)
    int U = 0, groups = 0, maxmag = 0;  // sub-digit slots per window, window groups per slot, largest digit magnitude
    int off[8] = {0}, wd[8] = {0};
    int jlo[65] = {0};                  // window group g covers table windows [jlo[g], jlo[g + 1])
    std::vector<uint32_t> ord;          // per (u, g): the (j - jlo[g]) * T + b of its non-zero digits, largest magnitude first; bit 31 = negative
    std::vector<uint32_t> dstart;       // per (u, g): maxmag + 1 offsets into its part of ord: magnitude v in [dstart[maxmag - v], dstart[maxmag - v + 1])
    std::vector<size_t> ord_base;       // per (u, g): where its part of ord begins
};

// weights: T canonical 256-bit integers (4 x u64, below 2^255)
inline KeyFoldPlan keyfold_plan(const uint64_t* weights, size_t T, size_t m, int kb, int Wt) {
    KeyFoldPlan pl;
    // sub-digit split of a kb-bit chunk: U slots of ceil(kb / U) bits; cost per output ~ U (Wt T + 1.4 * 2^(wd - 1))
    double best = 0;
    for (int U = 1; U <= 8 && U <= kb; U++) {
        const int w = (kb + U - 1) / U;
        if (w > 12) continue;
        const double cost = U * ((double)Wt * (double)T + 1.4 * (double)(1u << (w - 1)));
        if (!pl.U || cost < best) {
            pl.U = U;
            best = cost;
        }
    }
    {
        int left = kb, at = 0;
        for (int u = 0; u < pl.U; u++) {
            const int w = (left + (pl.U - u) - 1) / (pl.U - u);
            pl.off[u] = at;
            pl.wd[u] = w;
            at += w;
            left -= w;
            if ((1 << (w - 1)) > pl.maxmag) pl.maxmag = 1 << (w - 1);
        }
    }
    // enough lanes to fill the chip: ~128 K; a group never holds less than one window
    pl.groups = 1;
    while (pl.groups < Wt && m * (size_t)pl.U * (size_t)pl.groups < ((size_t)1 << 17)) pl.groups++;
    for (int g = 0; g <= pl.groups; g++) pl.jlo[g] = (int)((long)Wt * g / pl.groups);
    // signed digits of every weight, slot by slot in increasing weight (a digit above 2^(wd - 1) borrows from the next slot)
    std::vector<int16_t> dig((size_t)Wt * pl.U * T);
    for (size_t b = 0; b < T; b++) {
        const uint64_t* s = weights + 4 * b;
        int carry = 0;
        for (int j = 0; j < Wt; j++)
            for (int u = 0; u < pl.U; u++) {
                const int pos = j * kb + pl.off[u], w = pl.wd[u];
                uint64_t raw = 0;
                if (pos < 256) {
                    raw = s[pos / 64] >> (pos % 64);
                    if (pos % 64 + w > 64 && pos / 64 + 1 < 4) raw |= s[pos / 64 + 1] << (64 - pos % 64);
                    raw &= ((uint64_t)1 << w) - 1;
                }
                int d = (int)raw + carry;
                carry = 0;
                if (d > (1 << (w - 1))) {
                    d -= 1 << w;
                    carry = 1;
                }
                dig[((size_t)j * pl.U + u) * T + b] = (int16_t)d;
            }
        if (carry != 0) {  // (a value close to 2^256)
            pl.ok = false;
            return pl;
        }
    }
    pl.ord_base.assign((size_t)pl.U * pl.groups + 1, 0);
    pl.dstart.assign((size_t)pl.U * pl.groups * (pl.maxmag + 1), 0);
    std::vector<uint32_t> count(pl.maxmag + 1);
    for (int u = 0; u < pl.U; u++)
        for (int g = 0; g < pl.groups; g++) {
            const size_t sg = (size_t)u * pl.groups + g;
            std::fill(count.begin(), count.end(), 0u);
            for (int j = pl.jlo[g]; j < pl.jlo[g + 1]; j++)
                for (size_t b = 0; b < T; b++) {
                    const int d = dig[((size_t)j * pl.U + u) * T + b];
                    if (d) count[d < 0 ? -d : d]++;
                }
            uint32_t* ds = pl.dstart.data() + sg * (pl.maxmag + 1);
            uint32_t at = 0;
            for (int v = pl.maxmag; v >= 1; v--) {
                ds[pl.maxmag - v] = at;
                at += count[v];
            }
            ds[pl.maxmag] = at;
            const size_t base = pl.ord.size();
            pl.ord_base[sg] = base;
            pl.ord.resize(base + at);
            std::vector<uint32_t> cur(ds, ds + pl.maxmag);  // next free position per magnitude
            for (int j = pl.jlo[g]; j < pl.jlo[g + 1]; j++)
                for (size_t b = 0; b < T; b++) {
                    const int d = dig[((size_t)j * pl.U + u) * T + b];
                    if (!d) continue;
                    const int v = d < 0 ? -d : d;
                    pl.ord[base + cur[pl.maxmag - v]++] = (uint32_t)((size_t)(j - pl.jlo[g]) * T + b) | (d < 0 ? 0x80000000u : 0u);
                }
        }
    pl.ord_base[(size_t)pl.U * pl.groups] = pl.ord.size();
    return pl;
}

}  // namespace lurk
