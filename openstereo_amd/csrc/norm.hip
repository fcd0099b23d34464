// InstanceNorm2d (+ activation) on NHWC maps: the normalisation of the reference-written FPN decoders of the StereoBase / IGEV feature
// pyramids -- Conv2xUp / Conv2x_IN / BasicConv2d(norm_layer=nn.InstanceNorm2d) (models/stereobase/backbone.py:46-53,
// models/igev/extractor.py:338-341, models/lightstereo/backbone.py:57-59: affine=False, eps 1e-5, per (image, channel) statistics over
// H x W) -- which no convolution epilogue can fold (the statistics need the whole map).  HBM-bound elementwise work: two reads of the
// map (statistics, apply) and one write, float4 along the channel rows.  Deterministic: per-segment partial sums in a caller-owned
// workspace, combined in a fixed order (no float atomics).
#include "osa_common.h"

namespace osa {

constexpr int IN_QPW = 8;         // channel quads per workgroup (32 channels: one 128-byte row segment per pixel)
constexpr int IN_PPW = 32;        // pixels in flight per workgroup iteration (256 threads = 32 pixels x 8 quads)

// partial sums of (x - k) and (x - k)^2 over one pixel segment, k = the channel's value at the image's first pixel (shifted data: no
// cancellation when |mean| >> std).  ws: [B][nseg][Cq][8] floats = {sum.xyzw, sumsq.xyzw}
__global__ __launch_bounds__(256) void instnorm_partial_kernel(const float* __restrict__ x, float* __restrict__ ws, long long HW, int C, int xCs,
                                                               int nseg, long long seglen) {
    __shared__ float4 red[2][IN_PPW][IN_QPW];
    const int b = blockIdx.z, seg = blockIdx.y, q = blockIdx.x * IN_QPW + (threadIdx.x & (IN_QPW - 1)), pl = threadIdx.x / IN_QPW;
    const int Cq = (C + 3) / 4;
    const float* xb = x + (size_t)b * HW * xCs;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), ss = s;
    if (q < Cq) {
        const float4 k = *reinterpret_cast<const float4*>(xb + q * 4);
        const long long p0 = (long long)seg * seglen, p1 = (p0 + seglen < HW) ? p0 + seglen : HW;
        for (long long p = p0 + pl; p < p1; p += IN_PPW) {
            const float4 v = *reinterpret_cast<const float4*>(xb + p * xCs + q * 4);
            const float dx = v.x - k.x, dy = v.y - k.y, dz = v.z - k.z, dw = v.w - k.w;
            s.x += dx; s.y += dy; s.z += dz; s.w += dw;
            ss.x = fmaf(dx, dx, ss.x); ss.y = fmaf(dy, dy, ss.y); ss.z = fmaf(dz, dz, ss.z); ss.w = fmaf(dw, dw, ss.w);
        }
    }
    red[0][pl][threadIdx.x & (IN_QPW - 1)] = s; red[1][pl][threadIdx.x & (IN_QPW - 1)] = ss;
    __syncthreads();
    if (pl == 0 && q < Cq) {
        float4 a = red[0][0][threadIdx.x], c = red[1][0][threadIdx.x];
        for (int i = 1; i < IN_PPW; ++i) {                  // fixed order
            const float4 u = red[0][i][threadIdx.x], v = red[1][i][threadIdx.x];
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; c.x += v.x; c.y += v.y; c.z += v.z; c.w += v.w;
        }
        float* o = ws + (((size_t)b * nseg + seg) * Cq + q) * 8;
        *reinterpret_cast<float4*>(o) = a; *reinterpret_cast<float4*>(o + 4) = c;
    }
}

// stats[b][Cq] = {mean.xyzw, rstd.xyzw} (biased variance, eps inside the root: F.instance_norm / nn.InstanceNorm2d)
__global__ __launch_bounds__(64) void instnorm_finalize_kernel(const float* __restrict__ x, const float* __restrict__ ws, float* __restrict__ stats,
                                                               long long HW, int C, int xCs, int nseg, float eps) {
    const int Cq = (C + 3) / 4, b = blockIdx.y, q = blockIdx.x * 64 + threadIdx.x;
    if (q >= Cq) return;
    const float4 k = *reinterpret_cast<const float4*>(x + (size_t)b * HW * xCs + q * 4);
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    for (int g = 0; g < nseg; ++g) {
        const float* o = ws + (((size_t)b * nseg + g) * Cq + q) * 8;
        for (int e = 0; e < 4; ++e) { s[e] += o[e]; ss[e] += o[4 + e]; }
    }
    const float kk[4] = {k.x, k.y, k.z, k.w};
    float m[4], r[4];
    for (int e = 0; e < 4; ++e) {
        const double dm = s[e] / (double)HW;                 // mean of (x - k)
        double var = ss[e] / (double)HW - dm * dm;
        var = var > 0 ? var : 0;
        m[e] = (float)(kk[e] + dm);
        r[e] = (float)(1.0 / sqrt(var + (double)eps));
    }
    float* o = stats + ((size_t)b * Cq + q) * 8;
    *reinterpret_cast<float4*>(o) = make_float4(m[0], m[1], m[2], m[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(r[0], r[1], r[2], r[3]);
}

__global__ __launch_bounds__(256) void instnorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats, float* __restrict__ y,
                                                             long long HW, int C, int xCs, int yCs, int act, float slope, float* __restrict__ y_meta) {
    __shared__ float red[4];
    const int Cq = (C + 3) / 4, b = blockIdx.y;
    const long long total = HW * Cq;
    float am = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / Cq; const int q = (int)(i - p * Cq);
        const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)b * HW + p) * xCs + q * 4);
        const float* st = stats + ((size_t)b * Cq + q) * 8;
        const float4 m = *reinterpret_cast<const float4*>(st), r = *reinterpret_cast<const float4*>(st + 4);
        float o[4] = {(v.x - m.x) * r.x, (v.y - m.y) * r.y, (v.z - m.z) * r.z, (v.w - m.w) * r.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (act == OSA_ACT_RELU) o[e] = fmaxf(o[e], 0.f);
            else if (act == OSA_ACT_LEAKY) o[e] = (o[e] > 0.f) ? o[e] : o[e] * slope;
            if (q * 4 + e >= C) o[e] = 0.f;                  // padded channels stay zero
            am = fmaxf(am, fabsf(o[e]));
        }
        *reinterpret_cast<float4*>(y + ((size_t)b * HW + p) * yCs + q * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (y_meta) publish_amax(y_meta, am, 0u, red);
}

}  // namespace osa

using namespace osa;

extern "C" size_t osa_instnorm_workspace_floats(int B, long long HW, int C) {
    const int Cq = (C + 3) / 4;
    long long nseg = (HW + 2047) / 2048;
    if (nseg > 256) nseg = 256;
    if (nseg < 1) nseg = 1;
    return (size_t)B * (size_t)nseg * Cq * 8 + (size_t)B * Cq * 8;
}

extern "C" int osa_instnorm_nhwc_f32(const float* x, float* y, int B, long long HW, int C, int xCs, int yCs, float eps, int act, float slope,
                                     float* workspace, float* y_meta, void* stream) {
    OSA_REQUIRE(x && y && workspace, "instnorm: NULL pointer");
    OSA_REQUIRE(B > 0 && HW > 0 && C > 0, "instnorm: bad dims");
    OSA_REQUIRE(xCs % 4 == 0 && yCs % 4 == 0 && xCs >= C && yCs >= C && ((size_t)x & 15) == 0 && ((size_t)y & 15) == 0,
                "instnorm: channel strides must be multiples of 4 (>= C), tensors 16-byte aligned");
    OSA_REQUIRE(act == OSA_ACT_NONE || act == OSA_ACT_RELU || act == OSA_ACT_LEAKY, "instnorm: activation %d unsupported", act);
    OSA_REQUIRE((C + 3) / 4 * 4 <= xCs && (C + 3) / 4 * 4 <= yCs, "instnorm: padded channel quad exceeds the stride");
    const int Cq = (C + 3) / 4;
    long long nseg = (HW + 2047) / 2048;
    if (nseg > 256) nseg = 256;
    if (nseg < 1) nseg = 1;
    const long long seglen = (HW + nseg - 1) / nseg;
    float* stats = workspace + (size_t)B * (size_t)nseg * Cq * 8;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(instnorm_partial_kernel, dim3(cdiv(Cq, IN_QPW), (unsigned)nseg, B), dim3(256), 0, st, x, workspace, HW, C, xCs, (int)nseg, seglen);
    hipLaunchKernelGGL(instnorm_finalize_kernel, dim3(cdiv(Cq, 64), B), dim3(64), 0, st, x, workspace, stats, HW, C, xCs, (int)nseg, eps);
    long long blocks = (HW * Cq + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(instnorm_apply_kernel, dim3((unsigned)blocks, B), dim3(256), 0, st, x, stats, y, HW, C, xCs, yCs, act, slope, y_meta);
    OSA_LAUNCH_CHECK("instnorm");
    return 0;
}
