// Disparity refinement kernels for gfx950.
//
// context_upsample: convex 3x3 up-sampling of a low-resolution disparity map
//   out[b,y,x] = sum_{k<9} w[b,k,y,x] * disp_low[b, y/s + k/3 - 1, x/s + k%3 - 1]     (zero outside)
// (stereo/modeling/disp_refinement/disp_refinement.py:194-204, models/stereobase/igev_blocks.py:51-63,
//  models/igev/submodule.py:253-265: F.unfold(3x3, pad 1) -> nearest x s -> weighted sum over the 9 taps).
// The fused form also applies the softmax over the 9 weight logits and the `disp * gain` the callers
// do first (lightstereo.py:61-62, stereobase_gru.py:114-119): the 9 x H x W unfolded / up-sampled /
// soft-maxed intermediates of the reference never exist.  HBM-bound: 9 floats read + 1 written per pixel.
#include "osa_common.h"

namespace osa {

struct CtxArgs {
    const float* disp; const float* w; float* out;
    int B, h, w_, scale, H, W;
    int softmax;      // 1: w holds logits, softmax over the 9 taps is applied in-kernel
    float gain;       // disp_low is multiplied by this first (callers pass disp*4)
};

__global__ __launch_bounds__(256) void context_upsample_kernel(const CtxArgs p) {
    const long long HW = (long long)p.H * p.W;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)p.B * HW) return;
    const int b = (int)(i / HW);
    const int hw = (int)(i - (long long)b * HW);
    const int y = hw / p.W, x = hw - y * p.W;
    const int yl = y / p.scale, xl = x / p.scale;          // F.interpolate(mode='nearest') with integer scale
    const float* wp = p.w + (size_t)b * 9 * HW + hw;
    const float* dp = p.disp + (size_t)b * p.h * p.w_;
    float wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = wp[(size_t)k * HW];
    float inv = 1.f;
    if (p.softmax) {
        float m = wv[0];
#pragma unroll
        for (int k = 1; k < 9; ++k) m = fmaxf(m, wv[k]);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { wv[k] = expf(wv[k] - m); s += wv[k]; }
        inv = 1.f / s;
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = yl + k / 3 - 1, xx = xl + k % 3 - 1;
        float d = 0.f;
        if ((unsigned)yy < (unsigned)p.h && (unsigned)xx < (unsigned)p.w_) d = dp[(size_t)yy * p.w_ + xx] * p.gain;
        acc = fmaf(wv[k] * inv, d, acc);
    }
    p.out[i] = acc;
}

}  // namespace osa

using namespace osa;

extern "C" int osa_context_upsample_f32(const float* disp_low, const float* weights, float* out,
                                        int B, int h, int w, int scale, int softmax_weights, float gain,
                                        void* stream) {
    OSA_REQUIRE(disp_low && weights && out, "context_upsample: NULL pointer");
    OSA_REQUIRE(B > 0 && h > 0 && w > 0 && scale >= 1, "context_upsample: bad dims");
    CtxArgs a;
    a.disp = disp_low; a.w = weights; a.out = out; a.B = B; a.h = h; a.w_ = w; a.scale = scale;
    a.H = h * scale; a.W = w * scale; a.softmax = softmax_weights ? 1 : 0; a.gain = gain;
    const long long total = (long long)B * a.H * a.W;
    hipLaunchKernelGGL(context_upsample_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, a);
    OSA_LAUNCH_CHECK("context_upsample");
    return 0;
}
