// Shared helpers for the gfx950 kernels: error reporting, launch checks, XCD-aware block remap.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstdlib>
#include "../../include/openstereo_amd.h"

namespace osa {

void set_error(const char* fmt, ...);

#define OSA_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) {                                           \
            ::osa::set_error(__VA_ARGS__);                       \
            return -1;                                           \
        }                                                        \
    } while (0)

#define OSA_LAUNCH_CHECK(name)                                                  \
    do {                                                                        \
        hipError_t e__ = hipGetLastError();                                     \
        if (e__ != hipSuccess) {                                                \
            ::osa::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return -2;                                                          \
        }                                                                       \
    } while (0)

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Measurement / ablation switches (tools/bench_layers.py, tools/bench_volume.py) exist only in the
// -DOSA_EXPERIMENTS build (openstereo_amd/lib/variants/): the shipped library reads no environment
// variable on the launch path.
#ifdef OSA_EXPERIMENTS
static inline int exp_int(const char* name, int dflt) { const char* e = getenv(name); return (e && *e) ? atoi(e) : dflt; }
static inline bool exp_set(const char* name) { return getenv(name) != nullptr; }
#else
static inline int exp_int(const char*, int dflt) { return dflt; }
static inline bool exp_set(const char*) { return false; }
#endif

// Range blocks (osa_f16x3_ranges), OSA_META_FLOATS floats per tensor.  max |value| lives in 8 slots on separate
// 64-byte lines (meta[0], meta[16], ... meta[112]; a consumer takes their maximum): atomics on one address
// serialise at ~12 ns each, so producers (a) reduce inside the workgroup first -- at most ONE atomic per workgroup --,
// (b) spread over the slots by blockIdx, and (c) skip the atomic when an EARLY relaxed peek of their slot (issued
// before the epilogue / store loop, so its L2 round trip overlaps real work; possibly stale = lower, which only
// means a redundant atomic) already covers their maximum.  Values are uint bit patterns of floats >= 0, so
// integer order == float order; the atomics are fire-and-forget.
constexpr int OSA_AMAX_SLOTS = 8, OSA_AMAX_STRIDE = 16;
__device__ __forceinline__ float amax_read(const float* meta) {          // consumer side (after the producer kernel ended)
    float m = meta[0];
#pragma unroll
    for (int s = 1; s < OSA_AMAX_SLOTS; ++s) m = fmaxf(m, meta[s * OSA_AMAX_STRIDE]);
    return m;
}
__device__ __forceinline__ float* amax_slot(float* meta) { return meta + (blockIdx.x & (OSA_AMAX_SLOTS - 1)) * OSA_AMAX_STRIDE; }
__device__ __forceinline__ unsigned amax_peek(float* meta) {
    return __hip_atomic_load(reinterpret_cast<const unsigned*>(amax_slot(meta)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// All threads of the workgroup call this (it contains barriers).  `red`: >= blockDim.x / 64 floats of LDS nobody
// else touches any more.
__device__ __forceinline__ void publish_amax(float* meta, float am, unsigned seen, float* red) {
#pragma unroll
    for (int off = 32; off; off >>= 1) am = fmaxf(am, __shfl_xor(am, off));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = am;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        float m = red[0];
        for (int w = 1; w < nw; ++w) m = fmaxf(m, red[w]);
        const unsigned mb = __builtin_bit_cast(unsigned, m);
        if (mb > seen) atomicMax(reinterpret_cast<unsigned*>(amax_slot(meta)), mb);
    }
}

// 16-byte activation stores of the big producers (conv epilogues, volume builder).  -DOSA_NT_STORE=1 issues them with the non-temporal
// hint (streaming outputs of 0.2-3.2 GB per launch that the next launch reads long after L2 / Infinity Cache have turned over): an r3
// A/B experiment (tools/build_variant.sh ntstore -DOSA_NT_STORE=1), see profiles/DESIGN_rounds1-5.md 3.2.
#ifndef OSA_NT_STORE
#define OSA_NT_STORE 0
#endif
typedef float osa_f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned osa_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16(float* dst, const float4& v) {
#if OSA_NT_STORE
    osa_f32x4_t t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<osa_f32x4_t*>(dst));
#else
    *reinterpret_cast<float4*>(dst) = v;
#endif
}
__device__ __forceinline__ void store16(float* dst, const uint4& v) {
#if OSA_NT_STORE
    osa_u32x4_t t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<osa_u32x4_t*>(dst));
#else
    *reinterpret_cast<uint4*>(dst) = v;
#endif
}

// ---- f16x3 operand splitting (conv_kernel.h arithmetic modes; shared with the volume builder's split output) ----
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));
// x = hi + lo with hi, lo fp16.  hi uses the packed round-toward-zero convert (2 floats per
// instruction): any rounding is fine for hi because lo = x - float(hi) is exact in fp32 and carries
// the remainder; lo is rounded to nearest, error <= 2^-12 |lo| <= 2^-22 |x|.
// No saturation: operands are brought into range by the per-tensor power-of-two scale below (pow2_scale);
// a value that still exceeds the fp16 range becomes inf and poisons the result visibly instead of being
// clamped silently.
__device__ __forceinline__ void split_f16(const float4 v, uint2& hi, uint2& lo) {
    const float x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
    const h16x2 h01 = __builtin_amdgcn_cvt_pkrtz(x0, x1), h23 = __builtin_amdgcn_cvt_pkrtz(x2, x3);
    // lo is rounded to nearest (unbiased): its error is what remains of the split
    const f16x4 l = {(_Float16)(x0 - (float)h01[0]), (_Float16)(x1 - (float)h01[1]),
                     (_Float16)(x2 - (float)h23[0]), (_Float16)(x3 - (float)h23[1])};
    hi = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
    lo = __builtin_bit_cast(uint2, l);
}

// Power-of-two scale s with amax * s in [2^14, 2^15): the largest operand sits one binade under the fp16
// maximum and every element down to 2^-18 * amax keeps a NORMAL lo half (22 significant bits); smaller
// elements degrade gracefully (absolute error <= 2^-25 / s, i.e. 2^-39 * amax).  Exact to undo (1 / s).
// amax == 0, denormal, inf or NaN: unscaled.
__device__ __forceinline__ float pow2_scale(float amax) {
    const unsigned b = __builtin_bit_cast(unsigned, amax);
    const int eb = (int)((b >> 23) & 0xffu);
    if (eb == 0 || eb == 255) return 1.f;
    int k = 15 - (eb - 126);
    k = k < -60 ? -60 : (k > 60 ? 60 : k);
    return __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
}
__device__ __forceinline__ float4 mul4(const float4 v, const float s) { return make_float4(v.x * s, v.y * s, v.z * s, v.w * s); }

// MI355X: 8 XCDs, each with a private L2; workgroup b is observed on XCD b % 8.
// Remap so that every XCD walks a contiguous run of tile ids (neighbouring tiles
// share halos -> L2 hits).  Bijective for any grid size.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned q = nblk >> 3, r = nblk & 7u;
    const unsigned xcd = bid & 7u, idx = bid >> 3;
    const unsigned start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

}  // namespace osa
