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

// Range blocks (osa_f16x3_ranges): a producing kernel folds max |value| of its outputs into meta[0] (uint bit
// pattern of a float >= 0, so integer order == float order).  amax_peek() is issued EARLY (before the epilogue /
// store loop) so that the L2 round trip of the pre-check overlaps real work; publish_amax() then costs a wave
// reduction and, only while the maximum is still rising, one fire-and-forget atomic.  The peeked value may be
// stale (lower): that only means a redundant atomic.  Call publish_amax with all 64 lanes active.
__device__ __forceinline__ unsigned amax_peek(const float* meta) {
    return __hip_atomic_load(reinterpret_cast<const unsigned*>(meta), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void publish_amax(float* meta, float am, unsigned seen) {
#pragma unroll
    for (int off = 32; off; off >>= 1) am = fmaxf(am, __shfl_xor(am, off));
    const unsigned mb = __builtin_bit_cast(unsigned, am);
    if ((threadIdx.x & 63) == 0 && mb > seen) atomicMax(reinterpret_cast<unsigned*>(meta), mb);
}

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
