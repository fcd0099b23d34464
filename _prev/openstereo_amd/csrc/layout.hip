// Layout converters between the reference's NCDHW volumes and the engine's NDHWC volumes.
// Tiled 32x32 transpose through LDS (padded to 33 -> conflict free); both sides coalesced.
#include "osa_common.h"

namespace osa {

// x [B][C][S] -> y [B][S][yCs] (+c_off)
__global__ __launch_bounds__(256) void to_ndhwc_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       int C, long long S, int yCs, int c_off) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const long long s0 = (long long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r; const long long s = s0 + tx;
        tile[r][tx] = (c < C && s < S) ? x[((size_t)b * C + c) * S + s] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const long long s = s0 + r; const int c = c0 + tx;
        if (c < C && s < S) y[((size_t)b * S + s) * yCs + c_off + c] = tile[tx][r];
    }
}

// x [B][S][xCs] (+c_off) -> y [B][C][S]
__global__ __launch_bounds__(256) void to_ncdhw_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       int C, long long S, int xCs, int c_off) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const long long s0 = (long long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const long long s = s0 + r; const int c = c0 + tx;
        tile[r][tx] = (c < C && s < S) ? x[((size_t)b * S + s) * xCs + c_off + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r; const long long s = s0 + tx;
        if (c < C && s < S) y[((size_t)b * C + c) * S + s] = tile[tx][r];
    }
}

// max |x| of a dense fp32 tensor into a (zeroed or partially filled) range block: what `torch.linalg.vector_norm(x, inf)` did for every
// operand that reaches an f16x3 layer from a torch op (training: ~120 reductions per GwcNet step at 17 us each).  Grid-stride float4
// loads, one atomic max per workgroup, spread over the block's 8 slots (osa_common.h).
// (r4: the slots are cleared by device-scope atomic exchanges, i.e. at the same point of the memory system the producers' atomicMax
// operations execute at -- a plain store leaves a dirty zero line in ONE XCD's L2 that is only ordered against the other XCDs' atomics by
// the kernel-boundary write-back)
__global__ void amax_clear_kernel(float* meta) {
    if (threadIdx.x < OSA_AMAX_SLOTS) atomicExch(reinterpret_cast<unsigned*>(meta) + threadIdx.x * OSA_AMAX_STRIDE, 0u);
}

__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long long n, float* meta) {
    __shared__ float red[4];
    float am = 0.f;
    const long long n4 = n >> 2, stride = (long long)gridDim.x * 256;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 v = x4[i];
        am = fmaxf(am, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) am = fmaxf(am, fabsf(x[(n4 << 2) + threadIdx.x]));
    publish_amax(meta, am, 0u, red);
}

}  // namespace osa

using namespace osa;

extern "C" int osa_amax_f32(const float* x, long long n, float* meta, void* stream) {
    OSA_REQUIRE(x && meta && n > 0, "amax: NULL pointer or empty tensor");
    OSA_REQUIRE(((size_t)x & 15) == 0, "amax: x must be 16-byte aligned");
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    // overwrite semantics (like the torch reduction this replaces): the 8 maximum slots are cleared first, [1] (a split tensor's scale) is
    // left alone.  A replayed hipGraph must not depend on what the block held at the end of the previous replay.
    hipLaunchKernelGGL(amax_clear_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, meta);
    hipLaunchKernelGGL(amax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, meta);
    OSA_LAUNCH_CHECK("amax");
    return 0;
}

extern "C" int osa_ncdhw_to_ndhwc_f32(const float* x, float* y, int B, int C, long long S,
                                      int yCs, int c_off, void* stream) {
    OSA_REQUIRE(x && y, "ncdhw_to_ndhwc: NULL pointer");
    OSA_REQUIRE(B > 0 && C > 0 && S > 0 && c_off >= 0 && c_off + C <= yCs, "ncdhw_to_ndhwc: bad dims");
    OSA_REQUIRE(B <= 65535 && cdiv(C, 32) <= 65535, "ncdhw_to_ndhwc: grid too large");
    dim3 grid(cdiv(S, 32), cdiv(C, 32), B);
    hipLaunchKernelGGL(to_ndhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, C, S, yCs, c_off);
    OSA_LAUNCH_CHECK("ncdhw_to_ndhwc");
    return 0;
}

extern "C" int osa_ndhwc_to_ncdhw_f32(const float* x, float* y, int B, int C, long long S,
                                      int xCs, int c_off, void* stream) {
    OSA_REQUIRE(x && y, "ndhwc_to_ncdhw: NULL pointer");
    OSA_REQUIRE(B > 0 && C > 0 && S > 0 && c_off >= 0 && c_off + C <= xCs, "ndhwc_to_ncdhw: bad dims");
    OSA_REQUIRE(B <= 65535 && cdiv(C, 32) <= 65535, "ndhwc_to_ncdhw: grid too large");
    dim3 grid(cdiv(S, 32), cdiv(C, 32), B);
    hipLaunchKernelGGL(to_ncdhw_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, C, S, xCs, c_off);
    OSA_LAUNCH_CHECK("ndhwc_to_ncdhw");
    return 0;
}
