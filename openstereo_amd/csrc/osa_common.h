// Shared helpers for the gfx950 kernels: error reporting, launch checks, XCD-aware block remap.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
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
