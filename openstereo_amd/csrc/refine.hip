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
#include <cstring>

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


// ---- training form of the fused softmax + convex up-sampling (r6) ---------------------------------------------------------------------
// The reference up-samples the disparity of EVERY GRU iteration for its sequence loss (stereobase_gru.py:196-203, igev_stereo.py:198-207:
// spx_pred = F.softmax(spx_gru(...), 1); context_upsample(disp * 4, spx_pred)): 22 x (softmax, unfold, nearest x4, product, sum over 9)
// forward and their autograd backward on 9 x H x W tensors.  Forward: the kernel above with strided fp32 / fp16 logits (the transposed
// conv's channels-last output is read in place).  Backward, two launches:
//   (1) per full-resolution pixel: w = softmax(logits), out = sum_k w_k d_k;  dlogits_k = dout * w_k * (d_k - out)  (written in the
//       logits' layout and dtype);  t_k = dout * w_k  into a planar fp32 scratch [B][9][H][W];
//   (2) per low-resolution pixel (yy, xx): ddisp = gain * sum over the 3 x 3 cells (yl, xl) around it and their scale^2 pixels of
//       t_k with k = (yy - yl + 1) * 3 + (xx - xl + 1)  -- a gather in a fixed order, no atomics.
struct CtxTrainArgs {
    const float* disp; const void* logits; const float* dout;
    float* out; void* dlogits; float* scratch; float* ddisp;
    long long lsb, lsk, lsy, lsx;       // logits strides in elements (batch, tap, row, column)
    long long dsb, dsk, dsy, dsx;       // dlogits strides
    int B, h, w_, scale, H, W, f16;
    float gain;
};
__device__ __forceinline__ float ctx_ld(const void* p, int f16, long long off) {
    return f16 ? (float)reinterpret_cast<const _Float16*>(p)[off] : reinterpret_cast<const float*>(p)[off];
}
// Thread = one full-resolution pixel; the scale^2 pixels of a low-resolution cell are consecutive lanes (host: scale in {1, 2, 4, 8}: a
// power of two <= 8, so a cell is 1 / 4 / 16 / 64 lanes and never straddles a wave), which lets the backward reduce t_k over the cell with
// lane shuffles.  Logits with unit tap stride (channels-last: the transposed conv's output) are read / written as packed pairs.
template <int BWD>
__global__ __launch_bounds__(256) void context_upsample_logits_kernel(const CtxTrainArgs p) {
    const int s2 = p.scale * p.scale;
    const long long ncell = (long long)p.B * p.h * p.w_;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long cell = t / s2;
    const int sub = (int)(t - cell * s2);
    const bool live = cell < ncell;
    const long long cc = live ? cell : ncell - 1;
    const int b = (int)(cc / ((long long)p.h * p.w_));
    const int r = (int)(cc - (long long)b * p.h * p.w_);
    const int yl = r / p.w_, xl = r - yl * p.w_;
    const int y = yl * p.scale + sub / p.scale, x = xl * p.scale + sub % p.scale;
    const long long HW = (long long)p.H * p.W;
    const long long i = (long long)b * HW + (long long)y * p.W + x;
    const long long lo = (long long)b * p.lsb + (long long)y * p.lsy + (long long)x * p.lsx;
    const float* dp = p.disp + (size_t)b * p.h * p.w_;
    float wv[9], dv[9];
    const bool packed = p.f16 && p.lsk == 1 && ((lo & 1) == 0) && ((reinterpret_cast<size_t>(p.logits) & 3) == 0);
    if (packed) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const unsigned* src = reinterpret_cast<const unsigned*>(reinterpret_cast<const _Float16*>(p.logits) + lo);
#pragma unroll
        for (int k = 0; k < 4; ++k) { const h2 v = __builtin_bit_cast(h2, src[k]); wv[2 * k] = (float)v[0]; wv[2 * k + 1] = (float)v[1]; }
        wv[8] = (float)reinterpret_cast<const _Float16*>(p.logits)[lo + 8];
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) wv[k] = ctx_ld(p.logits, p.f16, lo + k * p.lsk);
    }
    float m = wv[0];
#pragma unroll
    for (int k = 1; k < 9; ++k) m = fmaxf(m, wv[k]);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { wv[k] = expf(wv[k] - m); s += wv[k]; }
    const float inv = 1.f / s;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = yl + k / 3 - 1, xx = xl + k % 3 - 1;
        float d = 0.f;
        if ((unsigned)yy < (unsigned)p.h && (unsigned)xx < (unsigned)p.w_) d = dp[(size_t)yy * p.w_ + xx] * p.gain;
        dv[k] = d;
        wv[k] *= inv;
        acc = fmaf(wv[k], d, acc);
    }
    if (!BWD) { if (live) p.out[i] = acc; return; }
    const float g = live ? p.dout[i] : 0.f;
    float tk[9], dl[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { tk[k] = g * wv[k]; dl[k] = tk[k] * (dv[k] - acc); }
    if (live) {
        const long long dlo = (long long)b * p.dsb + (long long)y * p.dsy + (long long)x * p.dsx;
        if (p.f16 && p.dsk == 1 && ((dlo & 1) == 0) && ((reinterpret_cast<size_t>(p.dlogits) & 3) == 0)) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            unsigned* dst = reinterpret_cast<unsigned*>(reinterpret_cast<_Float16*>(p.dlogits) + dlo);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const h2 v = {(_Float16)dl[2 * k], (_Float16)dl[2 * k + 1]}; dst[k] = __builtin_bit_cast(unsigned, v); }
            reinterpret_cast<_Float16*>(p.dlogits)[dlo + 8] = (_Float16)dl[8];
        } else {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                if (p.f16) reinterpret_cast<_Float16*>(p.dlogits)[dlo + k * p.dsk] = (_Float16)dl[k];
                else reinterpret_cast<float*>(p.dlogits)[dlo + k * p.dsk] = dl[k];
            }
        }
    }
    // sum of t_k over the cell's lanes (fixed butterfly order: deterministic)
    for (int mlane = s2 >> 1; mlane >= 1; mlane >>= 1) {
#pragma unroll
        for (int k = 0; k < 9; ++k) tk[k] += __shfl_xor(tk[k], mlane, 64);
    }
    if (live && sub == 0) {
        const long long hw_lo = (long long)p.h * p.w_;
        float* sc = p.scratch + (size_t)b * 9 * hw_lo + r;
#pragma unroll
        for (int k = 0; k < 9; ++k) sc[(size_t)k * hw_lo] = tk[k];
    }
}
// scratch: T[b][k][yl][xl] = sum over the scale^2 pixels of cell (yl, xl) of dout * w_k  (9 x B x h x w floats)
__global__ __launch_bounds__(256) void context_upsample_ddisp_kernel(const CtxTrainArgs p) {
    const long long hw_lo = (long long)p.h * p.w_;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)p.B * hw_lo) return;
    const int b = (int)(i / hw_lo);
    const int r = (int)(i - (long long)b * hw_lo);
    const int yy = r / p.w_, xx = r - yy * p.w_;
    const float* sc = p.scratch + (size_t)b * 9 * hw_lo;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yl = yy - (k / 3 - 1), xl = xx - (k % 3 - 1);        // the cell whose tap k reads (yy, xx)
        if ((unsigned)yl < (unsigned)p.h && (unsigned)xl < (unsigned)p.w_) acc += sc[(size_t)k * hw_lo + (size_t)yl * p.w_ + xl];
    }
    p.ddisp[i] = acc * p.gain;
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

static int ctx_train_args(CtxTrainArgs& a, const float* disp_low, const void* logits, int logits_f16, const long long* ls, int B, int h, int w, int scale, float gain) {
    OSA_REQUIRE(disp_low && logits && ls, "context_upsample_logits: NULL pointer");
    OSA_REQUIRE(B > 0 && h > 0 && w > 0 && (scale == 1 || scale == 2 || scale == 4 || scale == 8), "context_upsample_logits: bad dims (scale 1 / 2 / 4 / 8)");
    memset(&a, 0, sizeof(a));
    a.disp = disp_low; a.logits = logits; a.f16 = logits_f16 ? 1 : 0;
    a.lsb = ls[0]; a.lsk = ls[1]; a.lsy = ls[2]; a.lsx = ls[3];
    a.B = B; a.h = h; a.w_ = w; a.scale = scale; a.H = h * scale; a.W = w * scale; a.gain = gain;
    return 0;
}

extern "C" int osa_context_upsample_logits_f32(const float* disp_low, const void* logits, int logits_f16, const long long* logits_strides,
                                               float* out, int B, int h, int w, int scale, float gain, void* stream) {
    CtxTrainArgs a;
    if (int rc = ctx_train_args(a, disp_low, logits, logits_f16, logits_strides, B, h, w, scale, gain)) return rc;
    OSA_REQUIRE(out, "context_upsample_logits: NULL out");
    a.out = out;
    hipLaunchKernelGGL(context_upsample_logits_kernel<0>, dim3(cdiv((long long)B * a.H * a.W, 256)), dim3(256), 0, (hipStream_t)stream, a);
    OSA_LAUNCH_CHECK("context_upsample_logits");
    return 0;
}

extern "C" int osa_context_upsample_logits_bwd_f32(const float* disp_low, const void* logits, int logits_f16, const long long* logits_strides,
                                                   const float* dout, float* ddisp_low, void* dlogits, const long long* dlogits_strides, float* scratch,
                                                   int B, int h, int w, int scale, float gain, void* stream) {
    CtxTrainArgs a;
    if (int rc = ctx_train_args(a, disp_low, logits, logits_f16, logits_strides, B, h, w, scale, gain)) return rc;
    OSA_REQUIRE(dout && ddisp_low && dlogits && dlogits_strides && scratch, "context_upsample_logits_bwd: NULL pointer (scratch: 9 * B * h * w floats)");
    a.dsb = dlogits_strides[0]; a.dsk = dlogits_strides[1]; a.dsy = dlogits_strides[2]; a.dsx = dlogits_strides[3];
    a.dout = dout; a.ddisp = ddisp_low; a.dlogits = dlogits; a.scratch = scratch;
    hipLaunchKernelGGL(context_upsample_logits_kernel<1>, dim3(cdiv((long long)B * a.H * a.W, 256)), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(context_upsample_ddisp_kernel, dim3(cdiv((long long)B * h * w, 256)), dim3(256), 0, (hipStream_t)stream, a);
    OSA_LAUNCH_CHECK("context_upsample_logits_bwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Input pre-processing on device: RightTopPad(edge) + HWC->CHW + /255 + normalise,
// the EVALUATING transform chain of cfgs/gwcnet/gwcnet_sceneflow_uniform.yaml (stereo/datasets/
// dataset_utils/stereo_trans.py:243-267 RightTopPad, :22-29 TransposeImage, :48-56 NormalizeImage),
// for the left and the right image in one launch.  out[i, c, y, x] = ((img_i[ys, xs, c] / 255) - mean[c]) / std[c]
// with ys = clamp(y - pad_top, 0, H-1), xs = min(x, W-1).  layout 0: NCHW [2,3,Hp,Wp];
// layout 1: NHWC4 [2,Hp,Wp,4] (4th channel 0) -- what the engine's first conv consumes directly.
namespace osa {
struct PreArgs {
    const void* img[2]; float* out;
    int u8, H, W, Hp, Wp, layout;
    float mean[3], stdv[3];
};

__global__ __launch_bounds__(256) void preprocess_pair_kernel(const PreArgs p) {
    const long long n = (long long)2 * p.Hp * p.Wp;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % p.Wp); const long long r = i / p.Wp;
    const int y = (int)(r % p.Hp); const int im = (int)(r / p.Hp);
    int ys = y - (p.Hp - p.H); ys = ys < 0 ? 0 : ys;
    const int xs = x < p.W ? x : p.W - 1;
    const size_t src = ((size_t)ys * p.W + xs) * 3;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float raw = p.u8 ? (float)reinterpret_cast<const unsigned char*>(p.img[im])[src + c]
                               : reinterpret_cast<const float*>(p.img[im])[src + c];
        v[c] = (raw / 255.0f - p.mean[c]) / p.stdv[c];
    }
    if (p.layout == 0) {
        const size_t plane = (size_t)p.Hp * p.Wp;
        float* o = p.out + (size_t)im * 3 * plane + (size_t)y * p.Wp + x;
        o[0] = v[0]; o[plane] = v[1]; o[2 * plane] = v[2];
    } else {
        reinterpret_cast<float4*>(p.out)[i] = make_float4(v[0], v[1], v[2], 0.f);
    }
}
}  // namespace osa

extern "C" int osa_preprocess_pair_f32(const void* left_hwc, const void* right_hwc, int is_u8,
                                       int H, int W, int Hp, int Wp,
                                       const float* mean3, const float* std3,
                                       float* out, int layout, void* stream) {
    OSA_REQUIRE(left_hwc && right_hwc && out && mean3 && std3, "preprocess_pair: NULL pointer");
    OSA_REQUIRE(H > 0 && W > 0 && Hp >= H && Wp >= W, "preprocess_pair: padded size %dx%d smaller than image %dx%d", Hp, Wp, H, W);
    OSA_REQUIRE(layout == 0 || layout == 1, "preprocess_pair: bad layout");
    osa::PreArgs a;
    a.img[0] = left_hwc; a.img[1] = right_hwc; a.out = out; a.u8 = is_u8 ? 1 : 0;
    a.H = H; a.W = W; a.Hp = Hp; a.Wp = Wp; a.layout = layout;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.stdv[c] = std3[c]; }   // host pointers (3 floats each)
    const long long n = (long long)2 * Hp * Wp;
    hipLaunchKernelGGL(osa::preprocess_pair_kernel, dim3(osa::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a);
    OSA_LAUNCH_CHECK("preprocess_pair");
    return 0;
}


// ------------------------------------------------------------------ ConvGRU state update --
namespace osa {
// h and out may be the same buffer (the hidden state is updated in place inside its level's state buffer): element-wise, every
// thread reads its own quad before it writes it
__global__ __launch_bounds__(256) void gru_combine_kernel(const float* __restrict__ z, const float* __restrict__ q,
                                                          const float* h, float* out,
                                                          long long total, int nq, int zCs, int qCs, int hCs, int oCs, float* meta) {
    __shared__ float red[4];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float am = 0.f;
    const unsigned am_seen = meta ? amax_peek(meta) : 0u;
    if (i < total) {
    const long long px = i / nq;
    const int c = (int)(i - px * nq) * 4;
    const float4 zv = *reinterpret_cast<const float4*>(z + px * zCs + c);
    const float4 qv = *reinterpret_cast<const float4*>(q + px * qCs + c);
    const float4 hv = *reinterpret_cast<const float4*>(h + px * hCs + c);
    float4 o;
    o.x = (1.f - zv.x) * hv.x + zv.x * qv.x; o.y = (1.f - zv.y) * hv.y + zv.y * qv.y;
    o.z = (1.f - zv.z) * hv.z + zv.z * qv.z; o.w = (1.f - zv.w) * hv.w + zv.w * qv.w;
    *reinterpret_cast<float4*>(out + px * oCs + c) = o;
    am = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
    if (meta) publish_amax(meta, am, am_seen, red);
}
}  // namespace osa

extern "C" int osa_gru_combine_f32(const float* z, const float* q, const float* h, float* out,
                                   long long npix, int C, int zCs, int qCs, int hCs, int oCs, float* out_meta, void* stream) {
    OSA_REQUIRE(z && q && h && out, "gru_combine: NULL pointer");
    OSA_REQUIRE(npix > 0 && C > 0 && C % 4 == 0, "gru_combine: bad dims npix=%lld C=%d (C must be a multiple of 4)", npix, C);
    OSA_REQUIRE(zCs >= C && qCs >= C && hCs >= C && oCs >= C && ((zCs | qCs | hCs | oCs) & 3) == 0, "gru_combine: bad channel strides");
    OSA_REQUIRE((((size_t)z | (size_t)q | (size_t)h | (size_t)out) & 15) == 0, "gru_combine: pointers must be 16-byte aligned");
    const long long total = npix * (C / 4);
    OSA_REQUIRE((total + 255) / 256 < (1ll << 31), "gru_combine: grid too large");
    hipLaunchKernelGGL(osa::gru_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       z, q, h, out, total, C / 4, zCs, qCs, hCs, oCs, out_meta);
    OSA_LAUNCH_CHECK("gru_combine");
    return 0;
}


// ------------------------------------------------------------------ hidden-state resampling (update.py:99-109) --
namespace osa {
// y's range block >= x's: one thread folds max |x| (read from x's block) into slot 0 of y's
__device__ __forceinline__ void inherit_amax(const float* xmeta, float* ymeta) {
    if (xmeta && ymeta && blockIdx.x == 0 && threadIdx.x == 0)
        atomicMax(reinterpret_cast<unsigned*>(ymeta), __builtin_bit_cast(unsigned, amax_read(xmeta)));
}

// F.avg_pool2d(x, 3, stride=2, padding=1), count_include_pad=True: (sum over the window, zeros outside) / 9; one thread = one
// output pixel x 4 channels, window rows outer / columns inner like ATen's kernel
__global__ __launch_bounds__(256) void pool2x_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int Ho, int Wo,
                                                          int C4, int xCs, int yCs, long long total, const float* xmeta, float* ymeta) {
    inherit_amax(xmeta, ymeta);
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C4) * 4;
    long long px = i / C4;
    const int wo = (int)(px % Wo); px /= Wo;
    const int ho = (int)(px % Ho);
    const long long b = px / Ho;
    const float* xb = x + b * (long long)H * W * xCs + c;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int hi = ho * 2 - 1 + kh;
        if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int wi = wo * 2 - 1 + kw;
            if ((unsigned)wi >= (unsigned)W) continue;
            const float4 v = *reinterpret_cast<const float4*>(xb + ((long long)hi * W + wi) * xCs);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    *reinterpret_cast<float4*>(y + ((b * Ho + ho) * Wo + wo) * yCs + c) = make_float4(s.x / 9.f, s.y / 9.f, s.z / 9.f, s.w / 9.f);
}

// F.interpolate(mode='bilinear', align_corners=True): source index = dst * (in - 1) / (out - 1), ATen's upsample_bilinear2d
// arithmetic (h1 = (int)h1r, lambda1 = h1r - h1, lambda0 = 1 - lambda1; out = l0h * (l0w * v00 + l1w * v01) + l1h * (l0w * v10 + l1w * v11))
__global__ __launch_bounds__(256) void resize_bilinear_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int Hi, int Wi, int Ho, int Wo,
                                                                   int C4, int xCs, int yCs, float rh, float rw, long long total,
                                                                   const float* xmeta, float* ymeta) {
    inherit_amax(xmeta, ymeta);
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C4) * 4;
    long long px = i / C4;
    const int wo = (int)(px % Wo); px /= Wo;
    const int ho = (int)(px % Ho);
    const long long b = px / Ho;
    const float h1r = rh * ho, w1r = rw * wo;
    const int h1 = (int)h1r, w1 = (int)w1r;
    const int h1p = (h1 < Hi - 1) ? 1 : 0, w1p = (w1 < Wi - 1) ? 1 : 0;
    const float h1l = h1r - h1, h0l = 1.f - h1l, w1l = w1r - w1, w0l = 1.f - w1l;
    const float* xb = x + b * (long long)Hi * Wi * xCs + c;
    const float4 v00 = *reinterpret_cast<const float4*>(xb + ((long long)h1 * Wi + w1) * xCs);
    const float4 v01 = *reinterpret_cast<const float4*>(xb + ((long long)h1 * Wi + w1 + w1p) * xCs);
    const float4 v10 = *reinterpret_cast<const float4*>(xb + ((long long)(h1 + h1p) * Wi + w1) * xCs);
    const float4 v11 = *reinterpret_cast<const float4*>(xb + ((long long)(h1 + h1p) * Wi + w1 + w1p) * xCs);
    float4 o;
    o.x = h0l * (w0l * v00.x + w1l * v01.x) + h1l * (w0l * v10.x + w1l * v11.x);
    o.y = h0l * (w0l * v00.y + w1l * v01.y) + h1l * (w0l * v10.y + w1l * v11.y);
    o.z = h0l * (w0l * v00.z + w1l * v01.z) + h1l * (w0l * v10.z + w1l * v11.z);
    o.w = h0l * (w0l * v00.w + w1l * v01.w) + h1l * (w0l * v10.w + w1l * v11.w);
    *reinterpret_cast<float4*>(y + ((b * Ho + ho) * Wo + wo) * yCs + c) = o;
}
}  // namespace osa

static int check_nhwc(const char* what, const float* x, float* y, int C, int xCs, int yCs) {
    OSA_REQUIRE(x && y, "%s: NULL pointer", what);
    OSA_REQUIRE(C > 0 && C % 4 == 0 && xCs >= C && yCs >= C && ((xCs | yCs) & 3) == 0, "%s: C=%d xCs=%d yCs=%d must be multiples of 4, strides >= C", what, C, xCs, yCs);
    OSA_REQUIRE((((size_t)x | (size_t)y) & 15) == 0, "%s: pointers must be 16-byte aligned", what);
    return 0;
}

extern "C" int osa_pool2x_nhwc_f32(const float* x, float* y, int B, int H, int W, int C, int xCs, int yCs,
                                   const float* x_meta, float* y_meta, void* stream) {
    if (check_nhwc("pool2x", x, y, C, xCs, yCs)) return -1;
    OSA_REQUIRE(B > 0 && H > 0 && W > 0, "pool2x: bad dims");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)B * Ho * Wo * (C / 4);
    OSA_REQUIRE((total + 255) / 256 < (1ll << 31), "pool2x: grid too large");
    hipLaunchKernelGGL(osa::pool2x_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, y, H, W, Ho, Wo, C / 4, xCs, yCs, total, x_meta, y_meta);
    OSA_LAUNCH_CHECK("pool2x");
    return 0;
}

extern "C" int osa_resize_bilinear_nhwc_f32(const float* x, float* y, int B, int Hi, int Wi, int Ho, int Wo, int C, int xCs, int yCs,
                                            const float* x_meta, float* y_meta, void* stream) {
    if (check_nhwc("resize_bilinear", x, y, C, xCs, yCs)) return -1;
    OSA_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "resize_bilinear: bad dims");
    const float rh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, rw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    const long long total = (long long)B * Ho * Wo * (C / 4);
    OSA_REQUIRE((total + 255) / 256 < (1ll << 31), "resize_bilinear: grid too large");
    hipLaunchKernelGGL(osa::resize_bilinear_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, y, Hi, Wi, Ho, Wo, C / 4, xCs, yCs, rh, rw, total, x_meta, y_meta);
    OSA_LAUNCH_CHECK("resize_bilinear");
    return 0;
}


// ------------------------------------------------------------------ disparity update of the GRU loop (igev_stereo.py:201) --
namespace osa {
// disp += delta (delta: channel 0 of an NHWC tensor, may be NULL = 0), and the copies the next iteration's consumers read: the NCHW
// [B,1,H,W] map itself (geometry lookup), an NHWC [B,H,W,4] map with the disparity in channel 0 and zeros behind it (7x7 convd1 of the motion
// encoder) and one channel of the 1/4 GRU level's state buffer (`torch.cat([out, disp])`, update.py:92).  max |disp| is folded into the
// range blocks of the two NHWC destinations.
__global__ __launch_bounds__(256) void disp_update_kernel(float* disp, const float* __restrict__ delta, int dCs, float* __restrict__ disp4,
                                                          float* __restrict__ slot, int sCs, long long n, float* meta4, float* meta_slot) {
    __shared__ float red[4];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const unsigned seen4 = meta4 ? amax_peek(meta4) : 0u, seen_s = meta_slot ? amax_peek(meta_slot) : 0u;
    float am = 0.f;
    if (i < n) {
        const float v = disp[i] + (delta ? delta[i * dCs] : 0.f);
        disp[i] = v;
        if (disp4) *reinterpret_cast<float4*>(disp4 + i * 4) = make_float4(v, 0.f, 0.f, 0.f);
        if (slot) slot[i * sCs] = v;
        am = fabsf(v);
    }
    if (meta4) publish_amax(meta4, am, seen4, red);
    if (meta_slot) publish_amax(meta_slot, am, seen_s, red);
}
}  // namespace osa

extern "C" int osa_disp_update_f32(float* disp, const float* delta, int delta_cs, float* disp_nhwc4, float* slot, int slot_cs,
                                   long long npix, float* disp4_meta, float* slot_meta, void* stream) {
    OSA_REQUIRE(disp && npix > 0, "disp_update: bad arguments");
    OSA_REQUIRE(!disp_nhwc4 || (((size_t)disp_nhwc4) & 15) == 0, "disp_update: the NHWC map must be 16-byte aligned");
    OSA_REQUIRE((npix + 255) / 256 < (1ll << 31), "disp_update: grid too large");
    hipLaunchKernelGGL(osa::disp_update_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       disp, delta, delta_cs, disp_nhwc4, slot, slot_cs, npix, disp4_meta, slot_meta);
    OSA_LAUNCH_CHECK("disp_update");
    return 0;
}
