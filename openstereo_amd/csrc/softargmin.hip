// Disparity regression kernels for gfx950.
//
//  * softargmin          : out = sum_d d * prob[d]                      (disp_regression.py:8-12)
//  * softmax_softargmin  : softmax over D fused with the expectation    (stereobase_gru.py:163-164)
//  * upsample_softargmin : trilinear x(D/Dl, H/Hl, W/Wl) upsample of the low-res cost, softmax over
//                          D and expectation in ONE pass: the [B,D,H,W] upsampled cost, its softmax
//                          and the p*d product (3 x 401 MB in the reference,
//                          gwcnet_disp_processor.py:128-133) never exist.  6.3 MB in, 2.1 MB out.
// All are HBM/L2-bound streaming kernels: lanes run along w (coalesced), D is a serial loop.
#include "osa_common.h"

namespace osa {

__global__ __launch_bounds__(256) void softargmin_kernel(const float* __restrict__ prob, float* __restrict__ out,
                                                         int D, long long HW, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // over B*H*W
    if (i >= total) return;
    const long long b = i / HW, hw = i - b * HW;
    const float* p = prob + (size_t)b * D * HW + hw;
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < D; ++d) s = fmaf(p[(size_t)d * HW], (float)d, s);
    out[i] = s;
}

__global__ __launch_bounds__(256) void softmax_softargmin_kernel(const float* __restrict__ cost, float* __restrict__ prob,
                                                                 float* __restrict__ out, int D, long long HW, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long b = i / HW, hw = i - b * HW;
    const float* c = cost + (size_t)b * D * HW + hw;
    float m = -INFINITY;
#pragma unroll 8
    for (int d = 0; d < D; ++d) m = fmaxf(m, c[(size_t)d * HW]);
    float se = 0.f, sd = 0.f;
#pragma unroll 8
    for (int d = 0; d < D; ++d) {
        const float e = expf(c[(size_t)d * HW] - m);
        se += e;
        sd = fmaf(e, (float)d, sd);
    }
    const float inv = 1.0f / se;
    if (out) out[i] = sd * inv;
    if (prob) {
        float* pp = prob + (size_t)b * D * HW + hw;
#pragma unroll 8
        for (int d = 0; d < D; ++d) pp[(size_t)d * HW] = expf(c[(size_t)d * HW] - m) * inv;
    }
}

// PyTorch's area_pixel_compute_source_index (linear modes)
__device__ __forceinline__ void src_index(int dst, float scale, int align, int in_size, int& i0, int& i1, float& l1) {
    float s;
    if (align) s = scale * (float)dst;
    else { s = scale * ((float)dst + 0.5f) - 0.5f; s = s < 0.f ? 0.f : s; }
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
    l1 = s - (float)i0;
}

struct UpArgs {
    const float* cost; float* out;
    int B, Dl, Hl, Wl, D, H, W;
    int align;
    float sd, sh, sw;    // input/output scale per dim
};

// one thread per output pixel; its Dl bilinearly-interpolated low-res costs live in LDS
// (layout [dl][thread] -> conflict free), then a serial pass over the D upsampled samples.
__global__ __launch_bounds__(256) void upsample_softargmin_kernel(const UpArgs p) {
    extern __shared__ float cl[];   // [Dl][256]
    const int tid = threadIdx.x;
    const long long HW = (long long)p.H * p.W;
    const long long i = (long long)blockIdx.x * 256 + tid;
    const bool live = i < (long long)p.B * HW;
    const long long ii = live ? i : 0;
    const int b = (int)(ii / HW);
    const int hw = (int)(ii - (long long)b * HW);
    const int y = hw / p.W, x = hw - y * p.W;
    int y0, y1, x0, x1; float ly, lx;
    src_index(y, p.sh, p.align, p.Hl, y0, y1, ly);
    src_index(x, p.sw, p.align, p.Wl, x0, x1, lx);
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    const size_t plane = (size_t)p.Hl * p.Wl;
    const float* c = p.cost + (size_t)b * p.Dl * plane;
    const size_t o00 = (size_t)y0 * p.Wl + x0, o01 = (size_t)y0 * p.Wl + x1, o10 = (size_t)y1 * p.Wl + x0, o11 = (size_t)y1 * p.Wl + x1;
    float m = -INFINITY;
#pragma unroll 4
    for (int dl = 0; dl < p.Dl; ++dl) {
        const float* cp = c + (size_t)dl * plane;
        const float v = w00 * cp[o00] + w01 * cp[o01] + w10 * cp[o10] + w11 * cp[o11];
        cl[dl * 256 + tid] = v;
        m = fmaxf(m, v);
    }
    float se = 0.f, sdisp = 0.f;
#pragma unroll 4
    for (int d = 0; d < p.D; ++d) {
        int d0, d1; float ld;
        src_index(d, p.sd, p.align, p.Dl, d0, d1, ld);
        const float v = (1.f - ld) * cl[d0 * 256 + tid] + ld * cl[d1 * 256 + tid];
        const float e = expf(v - m);
        se += e;
        sdisp = fmaf(e, (float)d, sdisp);
    }
    if (live) p.out[i] = sdisp / se;
}

}  // namespace osa

using namespace osa;

extern "C" int osa_softargmin_f32(const float* prob, float* out, int B, int D, int H, int W, void* stream) {
    OSA_REQUIRE(prob && out, "softargmin: NULL pointer");
    OSA_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "softargmin: bad dims");
    const long long HW = (long long)H * W, total = HW * B;
    hipLaunchKernelGGL(softargmin_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, prob, out, D, HW, total);
    OSA_LAUNCH_CHECK("softargmin");
    return 0;
}

extern "C" int osa_softmax_softargmin_f32(const float* cost, float* prob, float* out,
                                          int B, int D, int H, int W, void* stream) {
    OSA_REQUIRE(cost && (out || prob), "softmax_softargmin: NULL pointer");
    OSA_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "softmax_softargmin: bad dims");
    const long long HW = (long long)H * W, total = HW * B;
    hipLaunchKernelGGL(softmax_softargmin_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       cost, prob, out, D, HW, total);
    OSA_LAUNCH_CHECK("softmax_softargmin");
    return 0;
}

static inline float lin_scale(int in, int out, int align) {
    // at::native::area_pixel_compute_scale
    if (align) return (out > 1) ? (float)(in - 1) / (float)(out - 1) : 0.f;
    return (float)in / (float)out;
}

extern "C" int osa_upsample_softargmin_f32(const float* cost_lowres, float* out,
                                           int B, int Dl, int Hl, int Wl, int D, int H, int W,
                                           int align_corners, void* stream) {
    OSA_REQUIRE(cost_lowres && out, "upsample_softargmin: NULL pointer");
    OSA_REQUIRE(B > 0 && Dl > 0 && Hl > 0 && Wl > 0 && D > 0 && H > 0 && W > 0, "upsample_softargmin: bad dims");
    const size_t lds = (size_t)Dl * 256 * sizeof(float);
    OSA_REQUIRE(lds <= 160 * 1024, "upsample_softargmin: Dl=%d too large for LDS", Dl);
    UpArgs a;
    a.cost = cost_lowres; a.out = out; a.B = B; a.Dl = Dl; a.Hl = Hl; a.Wl = Wl; a.D = D; a.H = H; a.W = W;
    a.align = align_corners ? 1 : 0;
    a.sd = lin_scale(Dl, D, a.align); a.sh = lin_scale(Hl, H, a.align); a.sw = lin_scale(Wl, W, a.align);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)upsample_softargmin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(upsample_softargmin_kernel, dim3(cdiv(total, 256)), dim3(256), lds, (hipStream_t)stream, a);
    OSA_LAUNCH_CHECK("upsample_softargmin");
    return 0;
}
