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

// Publish a wave's max |value| into a tensor's range block (meta[0], uint bit pattern of a float >= 0):
// one conditional atomic per wave; the relaxed pre-check keeps almost every wave off the atomic unit once
// the maximum has settled.  Call with all 64 lanes active.
__device__ __forceinline__ void publish_amax(float* meta, float am) {
#pragma unroll
    for (int off = 32; off; off >>= 1) am = fmaxf(am, __shfl_xor(am, off));
    if ((threadIdx.x & 63) == 0 && am > 0.f) {
        unsigned* a = reinterpret_cast<unsigned*>(meta);
        const unsigned mb = __builtin_bit_cast(unsigned, am);
        if (mb > __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a, mb);
    }
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
