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

// exp(x) for x <= 0 on the transcendental unit with a compensated argument: t = x * log2(e) is formed as
// hi + lo (lo = the rounding error of the product plus the low bits of the constant), exp2(hi) comes from
// v_exp_f32 and the lo part is applied to first order.  ~1.5 ulp, 6 VALU operations (expf: ~20).
__device__ __forceinline__ float exp_neg(float x) {
    const float L2E = 1.44269502162933349609375f, L2E_LO = 1.92596299112661746e-8f;
    const float t = x * L2E;
    const float lo = fmaf(x, L2E, -t) + x * L2E_LO;
    const float e = __builtin_amdgcn_exp2f(t);
    return (t < -126.f) ? e : fmaf(e, lo * 0.693147182464599609375f, e);     // x = -inf: e = 0 (lo would be NaN)
}

// Generic path (any output size, align_corners either way): one thread per output pixel; its Dl bilinearly
// interpolated low-res costs live in LDS (layout [dl][thread] -> conflict free), then two serial passes over the
// D upsampled samples: their maximum (the softmax is normalised by the maximum of the SAMPLES, as the reference's
// F.softmax does -- the plane maximum can lie far above every sample when costs are large, and exp() of all of
// them would underflow), then the exponentials.
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
#pragma unroll 4
    for (int dl = 0; dl < p.Dl; ++dl) {
        const float* cp = c + (size_t)dl * plane;
        cl[dl * 256 + tid] = w00 * cp[o00] + w01 * cp[o01] + w10 * cp[o10] + w11 * cp[o11];
    }
    float m = -INFINITY;
#pragma unroll 4
    for (int d = 0; d < p.D; ++d) {
        int d0, d1; float ld;
        src_index(d, p.sd, p.align, p.Dl, d0, d1, ld);
        m = fmaxf(m, (1.f - ld) * cl[d0 * 256 + tid] + ld * cl[d1 * 256 + tid]);
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

// Fast path: exact x4 in all three dimensions, align_corners = False (GwcNet: [48,136,240] -> [192,544,960],
// gwcnet_disp_processor.py:99-133).  The four output disparities 4k .. 4k+3 depend on planes k-1, k, k+1 only, with
// the constant weights (.375,.625) (.125,.875) (.875,.125) (.625,.375), so a thread streams over the low-res planes
// with a 3-value window of bilinear samples and an ONLINE softmax (running maximum of the samples seen so far;
// one rescale + four exponentials per plane) -- no LDS, ~40 registers, 8 waves per SIMD to hide the L1/L2 latency
// of the 4 taps per plane.  Sample values are bit-identical to the generic path (same products, same order).
// r4: a workgroup is a 64 x 4 pixel tile (its four waves sample the same two or three low-res rows: L1 hits instead of four trips to L2) and
// the tiles are numbered through xcd_remap, so that every XCD walks a contiguous band of the image: with the plain linear numbering the
// four output rows that share a low-res row sat in workgroups 3.75 ids apart, i.e. on different XCDs, and every private L2 fetched the same
// cost rows again (PMC r3: 337 MB per launch for 66.9 MB of input, 5.0x).  Same samples, same order: bit-identical.
__global__ __launch_bounds__(256) void upsample4_softargmin_kernel(const UpArgs p) {
    const long long HW = (long long)p.H * p.W;
    const int tilesX = (p.W + 63) >> 6, tilesY = (p.H + 3) >> 2;
    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = bid % tilesX; bid /= tilesX;
    const int ty = bid % tilesY;
    const int b = bid / tilesY;
    const int x = tx * 64 + (threadIdx.x & 63), y = ty * 4 + (threadIdx.x >> 6);
    if (x >= p.W || y >= p.H) return;
    const long long i = (long long)b * HW + (long long)y * p.W + x;
    int y0, y1, x0, x1; float ly, lx;
    src_index(y, 0.25f, 0, p.Hl, y0, y1, ly);
    src_index(x, 0.25f, 0, p.Wl, x0, x1, lx);
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    const int plane = p.Hl * p.Wl;
    const float* c = p.cost + (size_t)b * p.Dl * plane;
    const int o00 = y0 * p.Wl + x0, o01 = y0 * p.Wl + x1, o10 = y1 * p.Wl + x0, o11 = y1 * p.Wl + x1;
    auto bil = [&](int k) {
        const float* cp = c + (size_t)k * plane;
        return w00 * cp[o00] + w01 * cp[o01] + w10 * cp[o10] + w11 * cp[o11];
    };
    float vm = 0.f, vc = bil(0), vn = (p.Dl > 1) ? bil(1) : vc;
    float m = -INFINITY, se = 0.f, sd = 0.f;
    for (int k = 0; k < p.Dl; ++k) {
        const float vnn = (k + 2 < p.Dl) ? bil(k + 2) : 0.f;          // requested one plane ahead of its use
        // samples 4k .. 4k+3 (src = k - .375, k - .125, k + .125, k + .375; clamped to 0 below plane 0, i1 = i0 above the last)
        const float s0 = (k == 0) ? 1.f * vc + 0.f * vn : 0.375f * vm + 0.625f * vc;
        const float s1 = (k == 0) ? 1.f * vc + 0.f * vn : 0.125f * vm + 0.875f * vc;
        const float up = (k + 1 < p.Dl) ? vn : vc;
        const float s2 = 0.875f * vc + 0.125f * up;
        const float s3 = 0.625f * vc + 0.375f * up;
        const float mn = fmaxf(fmaxf(m, fmaxf(s0, s1)), fmaxf(s2, s3));
        const float r = exp_neg(m - mn);                               // m = -inf at k = 0: r = 0
        const float e0 = exp_neg(s0 - mn), e1 = exp_neg(s1 - mn), e2 = exp_neg(s2 - mn), e3 = exp_neg(s3 - mn);
        const float d0 = (float)(4 * k);
        se = fmaf(se, r, (e0 + e1) + (e2 + e3));
        sd = fmaf(sd, r, fmaf(e0, d0, fmaf(e1, d0 + 1.f, fmaf(e2, d0 + 2.f, e3 * (d0 + 3.f)))));
        m = mn;
        vm = vc; vc = vn; vn = vnn;
    }
    p.out[i] = sd / se;
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
    const long long total = (long long)B * H * W;
    if (!a.align && D == 4 * Dl && H == 4 * Hl && W == 4 * Wl && (long long)Hl * Wl < (1ll << 30)) {
        const long long tiles = (long long)B * ((H + 3) / 4) * ((W + 63) / 64);
        OSA_REQUIRE(tiles < (1ll << 31), "upsample_softargmin: grid too large");
        hipLaunchKernelGGL(upsample4_softargmin_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, a);
        OSA_LAUNCH_CHECK("upsample4_softargmin");
        return 0;
    }
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)upsample_softargmin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(upsample_softargmin_kernel, dim3(cdiv(total, 256)), dim3(256), lds, (hipStream_t)stream, a);
    OSA_LAUNCH_CHECK("upsample_softargmin");
    return 0;
}
