// synth.hip - deterministic synthetic inputs generated directly in HBM (bench / tests; SURVEY.md
// section 8d): SplitMix64 in counter mode, seed 0x4C55524B ("LURK") + stream id.  Scalars come in the
// two distributions of the measurement plan (uniform, witness-like); bases are P_i = [k_i]G with a
// known discrete log k_i, which gives full-size MSM runs a cheap exact checksum:
//   sum_i s_i * P_i  ==  [sum_i s_i * k_i mod q] G.
// The definitions mirror oracle/pyref.py (uniform_fe / witness_like_fe / synth_base_scalar); the
// tests compare the two implementations element by element.
#include "common.hpp"
#include "curve.cuh"

namespace lurk {

constexpr uint64_t SYNTH_SEED = 0x4C55524BULL;

__host__ __device__ inline uint64_t splitmix_at(uint64_t stream, uint64_t index) {
    uint64_t s = SYNTH_SEED + (stream << 32) + index * 0x9E3779B97F4A7C15ULL;
    s += 0x9E3779B97F4A7C15ULL;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

template <class F>
__device__ Fe<F> synth_uniform(uint64_t stream, uint64_t i) {
    Fe<F> v;
    for (uint64_t retry = 0;; retry++) {
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint64_t x = splitmix_at(stream, (i * 4 + w) + (retry << 40));
            v.l[2 * w] = (uint32_t)x;
            v.l[2 * w + 1] = (uint32_t)(x >> 32);
        }
        v.l[7] &= (F::NBITS == 255) ? 0x7fffffffu : 0x3fffffffu;
        if (!fe_canonical_ge_mod<F>(v.l)) return v;
    }
}
template <class F>
__device__ Fe<F> synth_witness_like(uint64_t stream, uint64_t i) {
    uint64_t sel = splitmix_at(stream + 100, i) % 100, rep = splitmix_at(stream + 101, i) % 100;
    if (rep < 5) return synth_uniform<F>(stream + 102, 0);
    if (sel < 80) return synth_uniform<F>(stream, i);
    Fe<F> v = fe_zero<F>();
    uint64_t x = splitmix_at(stream + 103, i);
    v.l[0] = (uint32_t)(sel < 92 ? (x & 1) : (x & 0xFFFF));
    return v;
}

template <class F>
__global__ __launch_bounds__(256) void synth_scalars_kernel(uint64_t stream, int dist, size_t first, size_t n, Fe<F>* out, int out_mont) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Fe<F> v = dist == 0 ? synth_uniform<F>(stream, first + i) : synth_witness_like<F>(stream, first + i);
    out[i] = out_mont ? fe_to_mont<F>(v) : v;
}

// ---- fixed-base table: tab[j*256 + d] = d * 256^j * G (affine), j < 32, d < 256 ---------------
template <class P>
__global__ void synth_table_kernel(Affine<P>* tab) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 32 * 256) return;
    int j = id >> 8, d = id & 255;
    Affine<P> g;
    g.x = fe_neg<P>(fe_one<P>());  // generator (-1, 2)
    g.y = fe_dbl<P>(fe_one<P>());
    Xyzz<P> b = xyzz_from_affine<P>(g);
    for (int q = 0; q < 8 * j; q++) b = xyzz_dbl<P>(b);
    tab[id] = xyzz_to_affine<P>(xyzz_mul_small<P>(b, (uint32_t)d));
}
template <class P, class SF>
__global__ __launch_bounds__(256) void synth_bases_kernel(const Affine<P>* __restrict__ tab, size_t first, size_t n, Affine<P>* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Fe<SF> k = synth_uniform<SF>(0, first + i);
    if (fe_is_zero<SF>(k)) k.l[0] = 1;
    Xyzz<P> acc = xyzz_identity<P>();
    for (int j = 0; j < 32; j++) {
        uint32_t d = (k.l[j >> 2] >> ((j & 3) * 8)) & 0xff;
        if (d) xyzz_madd<P>(acc, tab[j * 256 + d], false);
    }
    out[i] = xyzz_to_affine<P>(acc);
}

static std::mutex g_tab_mu;
static std::map<std::pair<int, int>, DevBuf> g_tab;  // (device, curve) -> table

template <class P>
static const Affine<P>* get_table(int curve, hipStream_t s) {
    int dev = 0;
    LURK_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_tab_mu);
    auto key = std::make_pair(dev, curve);
    auto it = g_tab.find(key);
    if (it == g_tab.end()) {
        DevBuf b(32 * 256 * sizeof(Affine<P>));
        hipLaunchKernelGGL((synth_table_kernel<P>), dim3(32), dim3(256), 0, s, b.as<Affine<P>>());
        LURK_HIP_CHECK(hipGetLastError());
        LURK_HIP_CHECK(hipStreamSynchronize(s));
        it = g_tab.emplace(key, std::move(b)).first;
    }
    return it->second.as<Affine<P>>();
}

}  // namespace lurk

using namespace lurk;

extern "C" {

int lurk_hip_synth_scalars_dev(int field_id, uint64_t stream_id, int dist, size_t first, size_t n, void* d_out, int out_mont, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(field_id >= 0 && field_id <= 2, "unknown field id");
        LURK_REQUIRE(dist == 0 || dist == 1, "dist must be 0 (uniform) or 1 (witness-like)");
        if (n == 0) return;
        LURK_REQUIRE(d_out, "null buffer");
        hipStream_t s = (hipStream_t)stream;
        dim3 grid(div_up(n, 256)), block(256);
        if (field_id == 0) hipLaunchKernelGGL((synth_scalars_kernel<PallasFp>), grid, block, 0, s, stream_id, dist, first, n, (Fe<PallasFp>*)d_out, out_mont);
        else if (field_id == 1) hipLaunchKernelGGL((synth_scalars_kernel<PallasFq>), grid, block, 0, s, stream_id, dist, first, n, (Fe<PallasFq>*)d_out, out_mont);
        else hipLaunchKernelGGL((synth_scalars_kernel<Bn254Fr>), grid, block, 0, s, stream_id, dist, first, n, (Fe<Bn254Fr>*)d_out, out_mont);
        LURK_HIP_CHECK(hipGetLastError());
    });
}

int lurk_hip_synth_bases_dev(int curve, size_t first, size_t n, void* d_out, void* stream) {
    return guarded([&] {
        LURK_REQUIRE(curve == 0 || curve == 1, "unknown curve id");
        if (n == 0) return;
        LURK_REQUIRE(d_out, "null buffer");
        hipStream_t s = (hipStream_t)stream;
        dim3 grid(div_up(n, 256)), block(256);
        if (curve == 0) {
            const Affine<PallasFp>* tab = get_table<PallasFp>(0, s);
            hipLaunchKernelGGL((synth_bases_kernel<PallasFp, PallasFq>), grid, block, 0, s, tab, first, n, (Affine<PallasFp>*)d_out);
        } else {
            const Affine<PallasFq>* tab = get_table<PallasFq>(1, s);
            hipLaunchKernelGGL((synth_bases_kernel<PallasFq, PallasFp>), grid, block, 0, s, tab, first, n, (Affine<PallasFq>*)d_out);
        }
        LURK_HIP_CHECK(hipGetLastError());
    });
}
}
