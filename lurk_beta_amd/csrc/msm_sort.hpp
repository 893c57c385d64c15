// msm_sort.hpp - stage 2 of the MSM pipeline (msm_core.cuh): the two-pass partitioned counting sort of the signed window digits.
// The kernels live in msm_sort.hip; msm.hip sizes the workspaces from the shape and calls msm_launch_sort.
//
// Round 5 ("one digit extraction instead of three"): a commitment used to read its scalars three times (a
// Montgomery -> canonical pass, then both sweeps of sort pass 1) and recode each scalar's W signed digits three times with a
// register walk of 8 shifts per digit.  Now
//   pass 1a  msm_hist1_kernel     reads the caller's (Montgomery) scalar once, stores the canonical form for pass 1b and counts the
//                                 coarse partitions; the digits of the two window widths the library chooses (16, 20) are cut out at
//                                 compile-time bit positions: two instructions per digit instead of a walk
//   pass 1b  msm_scatter1_kernel  recodes ONCE into W registers (the tile's counting and placement sweeps share them) and writes
//                                 5-byte records - a u32 entry (table index | sign) and, in a separate plane, the u8 low key
//   pass 2   msm_part2_kernel     counts on the key plane alone (1 byte per entry, four entries per load), then places entries read
//                                 four at a time; 6 bytes of `inter` traffic per entry where the uint2 records cost 24
#pragma once
#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>

namespace lurk {

constexpr int MSM_P_MIN = 2048;    // coarse partitions of the key space (pass 1 of the sort): 2048 up to n = 2^22, then
constexpr int MSM_P_MAX = 8192;    // doubled until a partition fits the LDS stage of pass 2 (msm_make_shape)
#ifndef LURK_SORT_BLOCK
#define LURK_SORT_BLOCK 1024       // threads per sort workgroup (A/B builds: make SORT_DEFS=-DLURK_SORT_BLOCK=512)
#endif
constexpr int MSM_SORT_BLOCK = LURK_SORT_BLOCK;
constexpr int MSM_P_PER_MAX = MSM_P_MAX / MSM_SORT_BLOCK;
constexpr int MSM_NB1 = 256 * (1024 / MSM_SORT_BLOCK);  // workgroups of pass 1 (one per CU at 1024 threads, two at 512)
constexpr int MSM_S = 64;          // sorted entries per accumulation task (the most: MsmShape::S is what a commitment uses)
constexpr int MSM_S_MIN = 8;
constexpr size_t MSM_TASK_TARGET = 131072;  // tasks that fill the chip: 256 CUs x 4 SIMDs x 64 lanes x 2 waves
constexpr size_t MSM_LDS_BYTES = 160 * 1024;  // per workgroup on gfx950
constexpr int MSM_LB_MAX = 8;      // low key bits sorted in pass 2: they travel as one byte per entry

struct MsmShape {
    int c, W, G;           // window bits, windows, key spaces
    uint32_t B, NB;        // buckets per space, total keys
    int P;                 // coarse partitions of pass 1 (power of two, MSM_P_MIN .. MSM_P_MAX)
    int tile;              // scalars per pass-1 scatter tile (1024, or less when P counters + W*tile entries exceed the LDS)
    int LB;                // low key bits sorted in pass 2 (NB >> LB == P), <= MSM_LB_MAX
    int NG;                // scan groups of MSM_GRP keys
    size_t n, stride;      // scalars in this call; table stride per window (0 in plain mode)
    int S;                 // sorted entries per accumulation task: MSM_S, less when that would leave lanes idle (small commitments)
    int sel;               // >= 0: TWO key spaces chosen by bit `sel` of the scalar's index (a pair of commitments with disjoint supports
                           // in one pass: the two halves of an inner-product-argument round); -1: off
    int low_prio;          // 1: the sort's kernels run at the lowest wave priority (a LURK_MSM_SUBMIT_FOLLOW commitment, msm.hip); 0: raised
};

inline size_t msm_scatter1_lds(int P, int W, int tile) { return (size_t)(3 * P + 32) * 4 + (size_t)W * tile * 8; }
inline size_t msm_part2_cap(int LB) { return (MSM_LDS_BYTES - (((size_t)1 << LB) + 32) * 4) / 4; }
// `inter`: W n u32 entries, then W n key bytes (+ 16 so that the four-at-a-time loads of pass 2 may run past the last entry)
inline size_t msm_inter_bytes(size_t entries) { return entries * 5 + 16; }

// c-bit windows over a key of `npoints` points (precomputed: the per-window table form), n scalars in this call
MsmShape msm_make_shape(int c, bool precomputed, size_t npoints, size_t n, int sel);

struct MsmSortBufs {
    void* inter;              // msm_inter_bytes(W n)
    uint32_t* sorted;         // W n
    uint32_t* block_hist;     // MSM_NB1 x P
    uint32_t* part_cnt;       // P
    uint32_t* part_start;     // P + 1
    uint32_t* cnt;            // NB
    uint32_t* bucket_start;   // NB
    void* canon;              // n x 32 B, only touched when the scalars arrive in Montgomery form
    uint32_t* zero[3];        // small counters of the later stages, cleared by the single-block scan launch
    int zero_n[3];
};
// enqueues the whole sort on s: sorted (table index | sign, grouped by key), cnt and bucket_start per key
template <class SF>
void msm_launch_sort(const MsmShape& sh, const void* d_scalars, int is_mont, const MsmSortBufs& b, hipStream_t s);

}  // namespace lurk
