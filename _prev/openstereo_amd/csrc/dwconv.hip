// Depthwise 2-D convolution, NHWC (LightStereo's MobileV2Residual depthwise
// 3x3 convs and the strip convolutions 1x7 / 7x1 / 1x11 / 11x1 / 1x21 / 21x1 of AttentionModule,
// stereo/modeling/models/lightstereo/aggregation.py:63-134).
//
// One multiply-add per output element and tap: HBM/L2 bound, no matrix work.  A thread owns 4
// consecutive channels of one output pixel (float4 everywhere), neighbouring threads neighbouring
// channel quads, so every load and store of a wave is a run of full 256-byte channel rows; the
// kh*kw taps of a pixel re-read rows that other pixels of the workgroup just touched (L1/L2).
// Weights are repacked once to [tap][C] so a tap's 4 weights are one float4.
// Epilogue: y = act(acc * scale[c] + shift[c]) + add   (folded eval BatchNorm or bias; optional addend).
#include "osa_common.h"

namespace osa {

struct DwArgs {
    const float* x; const float* w; const float* scale; const float* shift; const float* add; float* y;
    float* meta;         // range block of y (max |y| folded into meta[0]) or NULL
    int B, Hi, Wi, Ho, Wo, C, xCs, yCs, aCs;
    int kh, kw, stride, pad_h, pad_w, dil_h, dil_w, act;
    long long total;     // B*Ho*Wo*(C/4)
};

__global__ __launch_bounds__(256) void dwconv2d_nhwc_kernel(const DwArgs p) {
    __shared__ float red[4];
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    float am = 0.f;
    const unsigned am_seen = p.meta ? amax_peek(p.meta) : 0u;
    if (idx < p.total) {
    const int nq = p.C >> 2;
    const int q = (int)(idx % nq);
    long long pix = idx / nq;
    const int ox = (int)(pix % p.Wo); pix /= p.Wo;
    const int oy = (int)(pix % p.Ho);
    const int b = (int)(pix / p.Ho);
    const int c = q * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* xb = p.x + (size_t)b * p.Hi * p.Wi * p.xCs + c;
    const int iy0 = oy * p.stride - p.pad_h, ix0 = ox * p.stride - p.pad_w;
    for (int ky = 0; ky < p.kh; ++ky) {
        const int iy = iy0 + ky * p.dil_h;
        if ((unsigned)iy >= (unsigned)p.Hi) continue;
        for (int kx = 0; kx < p.kw; ++kx) {
            const int ix = ix0 + kx * p.dil_w;
            if ((unsigned)ix >= (unsigned)p.Wi) continue;
            const float4 v = *reinterpret_cast<const float4*>(xb + ((size_t)iy * p.Wi + ix) * p.xCs);
            const float4 w = *reinterpret_cast<const float4*>(p.w + (size_t)(ky * p.kw + kx) * p.C + c);
            acc.x = fmaf(v.x, w.x, acc.x); acc.y = fmaf(v.y, w.y, acc.y);
            acc.z = fmaf(v.z, w.z, acc.z); acc.w = fmaf(v.w, w.w, acc.w);
        }
    }
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + c);
    if (p.shift) sh = *reinterpret_cast<const float4*>(p.shift + c);
    float o[4] = {fmaf(acc.x, sc.x, sh.x), fmaf(acc.y, sc.y, sh.y), fmaf(acc.z, sc.z, sh.z), fmaf(acc.w, sc.w, sh.w)};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (p.act == OSA_ACT_RELU) o[e] = fmaxf(o[e], 0.f);
        else if (p.act == OSA_ACT_RELU6) o[e] = fminf(fmaxf(o[e], 0.f), 6.f);
    }
    const size_t opix = ((size_t)b * p.Ho + oy) * p.Wo + ox;
    if (p.add) {
        const float4 a = *reinterpret_cast<const float4*>(p.add + opix * p.aCs + c);
        o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w;
    }
    *reinterpret_cast<float4*>(p.y + opix * p.yCs + c) = make_float4(o[0], o[1], o[2], o[3]);
    am = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
    }
    if (p.meta) publish_amax(p.meta, am, am_seen, red);
}

// r4: pixel-run form.  The kernel above issues 2 x kh x kw 16-byte loads per output quad (input + weight per tap; the column re-reads are
// L1 hits, but every one of them occupies the vector-memory path) and ran LightStereo's depthwise launches at ~1.3 TB/s of algorithmic
// traffic -- 57 % of the whole cost stage (profiles/round4/amp_workloads_kernel_tables.txt).  Here a thread owns 4 channels of NPX
// consecutive output pixels of one row: per kernel row it loads the (NPX - 1) S + KW input quads of its window ONCE into registers and
// each tap's weight quad once; 3x3 stride 1: 27 loads per 4 outputs instead of 72, 1x21: 45 instead of 168.  Same fmaf order per
// output as the kernel above (ky outer, kx inner): bit-identical results.  Unit dilation; compile-time (KW, S) for the shapes the
// models use -- 3x3 (stride 1 / 2), the strip convolutions 1x7 / 7x1 / 1x11 / 11x1 / 1x21 / 21x1 -- anything else keeps the tap-loop kernel.
template <int KW, int S, int NPX, int KC = KW>
__global__ __launch_bounds__(256) void dwconv2d_run_kernel(const DwArgs p, const int runs) {
    // KC: taps of a row handled per window (KC < KW: the long horizontal strips 1x11 / 1x21 walk their row in chunks of KC taps, so the
    // register window stays (NPX - 1) S + KC quads; the chunk loop is kept rolled)
    static_assert(KW % KC == 0, "chunks must tile the kernel row");
    constexpr int WIN = (NPX - 1) * S + KC;
    __shared__ float red[4];
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    float am = 0.f;
    const unsigned am_seen = p.meta ? amax_peek(p.meta) : 0u;
    if (idx < p.total) {
        const unsigned nq = (unsigned)(p.C >> 2);
        unsigned r = (unsigned)idx;                        // host: total < 2^31
        const unsigned q = r % nq; r /= nq;
        const unsigned run = r % (unsigned)runs; r /= (unsigned)runs;
        const int oy = (int)(r % (unsigned)p.Ho);
        const int b = (int)(r / (unsigned)p.Ho);
        const int c = (int)q * 4, ox0 = (int)run * NPX;
        float4 acc[NPX];
#pragma unroll
        for (int j = 0; j < NPX; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* xb = p.x + (size_t)b * p.Hi * p.Wi * p.xCs + c;
        const int iy0 = oy * S - p.pad_h, ix0 = ox0 * S - p.pad_w;
        for (int ky = 0; ky < p.kh; ++ky) {
            const int iy = iy0 + ky;
            if ((unsigned)iy >= (unsigned)p.Hi) continue;
            const float* row = xb + (size_t)iy * p.Wi * p.xCs;
#pragma unroll 1
            for (int kc = 0; kc < KW; kc += KC) {
                float4 win[WIN];
#pragma unroll
                for (int i = 0; i < WIN; ++i) {
                    const int ix = ix0 + kc + i;
                    win[i] = ((unsigned)ix < (unsigned)p.Wi) ? *reinterpret_cast<const float4*>(row + (size_t)ix * p.xCs) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                const float* wr = p.w + (size_t)(ky * KW + kc) * p.C + c;
#pragma unroll
                for (int kx = 0; kx < KC; ++kx) {
                    const float4 w = *reinterpret_cast<const float4*>(wr + (size_t)kx * p.C);
#pragma unroll
                    for (int j = 0; j < NPX; ++j) {
                        const float4 v = win[j * S + kx];
                        // (a column outside the image contributes nothing in the tap-loop kernel; here it contributes v = 0: fmaf(0, w, acc) == acc
                        // exactly, except for acc = -0 -> +0, which no later operation distinguishes)
                        acc[j].x = fmaf(v.x, w.x, acc[j].x); acc[j].y = fmaf(v.y, w.y, acc[j].y);
                        acc[j].z = fmaf(v.z, w.z, acc[j].z); acc[j].w = fmaf(v.w, w.w, acc[j].w);
                    }
                }
            }
        }
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + c);
        if (p.shift) sh = *reinterpret_cast<const float4*>(p.shift + c);
        const size_t opix0 = ((size_t)b * p.Ho + oy) * p.Wo + ox0;
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            if (ox0 + j >= p.Wo) break;
            float o[4] = {fmaf(acc[j].x, sc.x, sh.x), fmaf(acc[j].y, sc.y, sh.y), fmaf(acc[j].z, sc.z, sh.z), fmaf(acc[j].w, sc.w, sh.w)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (p.act == OSA_ACT_RELU) o[e] = fmaxf(o[e], 0.f);
                else if (p.act == OSA_ACT_RELU6) o[e] = fminf(fmaxf(o[e], 0.f), 6.f);
            }
            if (p.add) {
                const float4 a = *reinterpret_cast<const float4*>(p.add + (opix0 + j) * p.aCs + c);
                o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w;
            }
            *reinterpret_cast<float4*>(p.y + (opix0 + j) * p.yCs + c) = make_float4(o[0], o[1], o[2], o[3]);
            am = fmaxf(am, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
        }
    }
    if (p.meta) publish_amax(p.meta, am, am_seen, red);
}

__global__ __launch_bounds__(256) void dwconv2d_pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int T) {
    const int i = blockIdx.x * 256 + threadIdx.x;      // dst index t*C + c
    if (i >= C * T) return;
    const int t = i / C, c = i - t * C;
    dst[i] = src[(size_t)c * T + t];
}

}  // namespace osa

using namespace osa;

extern "C" int osa_dwconv2d_pack_f32(const float* w_ref, float* w_packed, int C, int kh, int kw, void* stream) {
    OSA_REQUIRE(w_ref && w_packed, "dwconv2d_pack: NULL pointer");
    OSA_REQUIRE(C > 0 && kh > 0 && kw > 0, "dwconv2d_pack: bad dims C=%d k=%dx%d", C, kh, kw);
    const int n = C * kh * kw;
    hipLaunchKernelGGL(dwconv2d_pack_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w_ref, w_packed, C, kh * kw);
    OSA_LAUNCH_CHECK("dwconv2d_pack");
    return 0;
}

extern "C" int osa_dwconv2d_nhwc_f32(const float* x, const float* w_packed,
                                     const float* scale, const float* shift, const float* add, float* y,
                                     int B, int Hi, int Wi, int C, int xCs, int yCs, int aCs,
                                     int kh, int kw, int stride, int pad_h, int pad_w, int dil_h, int dil_w,
                                     int act, float* y_meta, void* stream) {
    OSA_REQUIRE(x && w_packed && y, "dwconv2d: NULL pointer");
    OSA_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && C > 0, "dwconv2d: bad dims B=%d H=%d W=%d C=%d", B, Hi, Wi, C);
    OSA_REQUIRE(C % 4 == 0 && xCs % 4 == 0 && yCs % 4 == 0 && xCs >= C && yCs >= C,
                "dwconv2d: C=%d, strides %d/%d must be multiples of 4 with stride >= C", C, xCs, yCs);
    OSA_REQUIRE((((size_t)x | (size_t)y | (size_t)w_packed) & 15) == 0, "dwconv2d: pointers must be 16-byte aligned");
    if (add) OSA_REQUIRE(aCs % 4 == 0 && aCs >= C && ((size_t)add & 15) == 0, "dwconv2d: addend stride %d / alignment", aCs);
    if (scale) OSA_REQUIRE(((size_t)scale & 15) == 0, "dwconv2d: scale must be 16-byte aligned");
    if (shift) OSA_REQUIRE(((size_t)shift & 15) == 0, "dwconv2d: shift must be 16-byte aligned");
    OSA_REQUIRE(stride == 1 || stride == 2, "dwconv2d: stride %d unsupported", stride);
    OSA_REQUIRE(kh > 0 && kw > 0 && dil_h > 0 && dil_w > 0 && pad_h >= 0 && pad_w >= 0, "dwconv2d: bad kernel geometry");
    OSA_REQUIRE(act == OSA_ACT_NONE || act == OSA_ACT_RELU || act == OSA_ACT_RELU6, "dwconv2d: act %d unsupported", act);
    DwArgs a;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.add = add; a.y = y; a.meta = y_meta;
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.C = C; a.xCs = xCs; a.yCs = yCs; a.aCs = aCs;
    a.kh = kh; a.kw = kw; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w; a.dil_h = dil_h; a.dil_w = dil_w; a.act = act;
    a.Ho = (Hi + 2 * pad_h - dil_h * (kh - 1) - 1) / stride + 1;
    a.Wo = (Wi + 2 * pad_w - dil_w * (kw - 1) - 1) / stride + 1;
    OSA_REQUIRE(a.Ho > 0 && a.Wo > 0, "dwconv2d: empty output");
    // pixel-run form for the shapes the models use (unit dilation; see dwconv2d_run_kernel)
    {
        constexpr int NPX = 4;
        const int runs = cdiv(a.Wo, NPX);
        const long long total = (long long)B * a.Ho * runs * (C / 4);
        void (*fn)(const DwArgs, int) = nullptr;
        if (dil_h == 1 && dil_w == 1 && total < (1ll << 31) && !exp_set("OSA_DW_TAPLOOP")) {
            if (kw == 3 && stride == 1) fn = dwconv2d_run_kernel<3, 1, NPX>;
            else if (kw == 3 && stride == 2) fn = dwconv2d_run_kernel<3, 2, NPX>;
            else if (kw == 1 && stride == 1) fn = dwconv2d_run_kernel<1, 1, NPX>;
            else if (kw == 7 && stride == 1) fn = dwconv2d_run_kernel<7, 1, NPX>;
            else if (kw == 11 && stride == 1) fn = dwconv2d_run_kernel<11, 1, NPX, 11>;      // (11 is prime: one 14-quad window)
            else if (kw == 21 && stride == 1) fn = dwconv2d_run_kernel<21, 1, NPX, 7>;       // three chunks of 7 taps: 10-quad windows
        }
        if (fn) {
            a.total = total;
            hipLaunchKernelGGL(fn, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, runs);
            OSA_LAUNCH_CHECK("dwconv2d");
            return 0;
        }
    }
    a.total = (long long)B * a.Ho * a.Wo * (C / 4);
    const long long nblk = (a.total + 255) / 256;
    OSA_REQUIRE(nblk < (1ll << 31), "dwconv2d: grid too large");
    hipLaunchKernelGGL(dwconv2d_nhwc_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, a);
    OSA_LAUNCH_CHECK("dwconv2d");
    return 0;
}
