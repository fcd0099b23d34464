// InstanceNorm2d (+ activation) on NHWC maps: the normalisation of the reference-written FPN decoders of the StereoBase / IGEV feature
// pyramids -- Conv2xUp / Conv2x_IN / BasicConv2d(norm_layer=nn.InstanceNorm2d) (models/stereobase/backbone.py:46-53,
// models/igev/extractor.py:338-341, models/lightstereo/backbone.py:57-59: affine=False, eps 1e-5, per (image, channel) statistics over
// H x W) -- which no convolution epilogue can fold (the statistics need the whole map).  HBM-bound elementwise work: two reads of the
// map (statistics, apply) and one write, float4 along the channel rows.  Deterministic: per-segment partial sums in a caller-owned
// workspace, combined in a fixed order (no float atomics).
#include "osa_common.h"
#include <cstring>

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


// ---- per-channel sums over the positions of a channels-last tensor (r6, training path) ---------------------------------------------------
// sum_dy[c] = sum_p dy[p][c] and, with x, sum_dyx[c] = sum_p dy[p][c] * (x[p][c] - shift[c]); optionally dx[p][c] = dy[p][c] * scale[c] in
// the same pass.  These are the bias gradient of a convolution (autograd of nn.Conv2d(bias=True): update.py:19-26, 38-40 -- 5 biased
// convolutions x 22 GRU iterations per StereoBase step) and the whole backward of a BatchNorm in eval mode (trainer FREEZE_BN,
// trainer_template.py:83-85: dbeta = sum dy, dgamma = invstd * sum dy (x - mean), dx = dy * gamma * invstd).  torch's reduction of a
// channels-last tensor over its positions runs at ~0.2 TB/s (reduce_kernel<512, 1>: 40 us for 14720 x 256 fp16 values); this is one
// coalesced pass: a thread owns a channel quad, the threads of a workgroup that share a quad take interleaved positions, partial sums go
// through LDS and a caller-owned workspace [workgroup][2][C4] and are combined in a fixed order (deterministic, no float atomics).
struct ChannelSumsTab { const void* p[24]; };     // blockIdx.y = item of a tensor list (osa_channel_sums_multi); a single tensor is item 0
template <int DYF16, int XF16, int WITH_X, int WITH_DX>
__global__ __launch_bounds__(256) void channel_sums_partial_kernel(const ChannelSumsTab tab, const void* __restrict__ x_, void* __restrict__ dx_,
                                                                   const float* __restrict__ shift, const float* __restrict__ scale,
                                                                   float* __restrict__ ws, long long P, int C, int dyCs, int xCs, int dxCs,
                                                                   long long per_wg) {
    __shared__ float4 red[2][256];
    const void* const dy_ = tab.p[blockIdx.y];
    const int Cq = (C + 3) / 4, PL = 256 / Cq;                 // position lanes per workgroup (host: Cq <= 256)
    const int q = threadIdx.x % Cq, pl = threadIdx.x / Cq;
    const bool live = pl < PL;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), sx = s;
    const int c0 = q * 4, nc = C - c0;                           // channels of this quad that exist (>= 1)
    float4 sh = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f);
    if (live) {
        if (WITH_X && shift) { sh.x = shift[c0]; if (nc > 1) sh.y = shift[c0 + 1]; if (nc > 2) sh.z = shift[c0 + 2]; if (nc > 3) sh.w = shift[c0 + 3]; }
        if (WITH_DX) { sc.x = scale[c0]; if (nc > 1) sc.y = scale[c0 + 1]; if (nc > 2) sc.z = scale[c0 + 2]; if (nc > 3) sc.w = scale[c0 + 3]; }
    }
    auto ld = [&](const void* base, int f16, long long p, int cs) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f16) {
            const _Float16* src = reinterpret_cast<const _Float16*>(base) + p * cs + c0;
            if (nc >= 4) {
                const uint2 u = *reinterpret_cast<const uint2*>(src);
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const h2 a = __builtin_bit_cast(h2, u.x), b = __builtin_bit_cast(h2, u.y);
                v = make_float4((float)a[0], (float)a[1], (float)b[0], (float)b[1]);
            } else { v.x = (float)src[0]; if (nc > 1) v.y = (float)src[1]; if (nc > 2) v.z = (float)src[2]; }
        } else {
            const float* src = reinterpret_cast<const float*>(base) + p * cs + c0;
            if (nc >= 4) v = *reinterpret_cast<const float4*>(src);
            else { v.x = src[0]; if (nc > 1) v.y = src[1]; if (nc > 2) v.z = src[2]; }
        }
        return v;
    };
    if (live) {
        const long long p0 = (long long)blockIdx.x * per_wg, p1 = (p0 + per_wg < P) ? p0 + per_wg : P;
        for (long long p = p0 + pl; p < p1; p += PL) {
            const float4 g = ld(dy_, DYF16, p, dyCs);
            s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
            if (WITH_X) {
                const float4 v = ld(x_, XF16, p, xCs);
                sx.x = fmaf(g.x, v.x - sh.x, sx.x); sx.y = fmaf(g.y, v.y - sh.y, sx.y); sx.z = fmaf(g.z, v.z - sh.z, sx.z); sx.w = fmaf(g.w, v.w - sh.w, sx.w);
            }
            if (WITH_DX) {
                const float o[4] = {g.x * sc.x, g.y * sc.y, g.z * sc.z, g.w * sc.w};
                if (DYF16) {
                    _Float16* dst = reinterpret_cast<_Float16*>(dx_) + p * dxCs + c0;
                    if (nc >= 4) {
                        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                        const h2 a = {(_Float16)o[0], (_Float16)o[1]}, b = {(_Float16)o[2], (_Float16)o[3]};
                        *reinterpret_cast<uint2*>(dst) = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
                    } else { dst[0] = (_Float16)o[0]; if (nc > 1) dst[1] = (_Float16)o[1]; if (nc > 2) dst[2] = (_Float16)o[2]; }
                } else {
                    float* dst = reinterpret_cast<float*>(dx_) + p * dxCs + c0;
                    if (nc >= 4) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                    else { dst[0] = o[0]; if (nc > 1) dst[1] = o[1]; if (nc > 2) dst[2] = o[2]; }
                }
            }
        }
    }
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = sx;
    __syncthreads();
    if (pl == 0) {
        float4 a = red[0][q], b = red[1][q];
        for (int i = 1; i < PL; ++i) {                            // fixed order
            const float4 u = red[0][i * Cq + q], v = red[1][i * Cq + q];
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
        }
        float4* o = reinterpret_cast<float4*>(ws) + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * Cq;
        o[q] = a; o[Cq + q] = b;
    }
}

// out[which][c] = sum over the workgroups' partials: one WAVE per (which, channel quad) -- lane l adds partials l, l + 64, ... in order, then
// a butterfly over the lanes (a fixed association order: deterministic).  (The first version gave every (which, quad) ONE thread that
// walked all <= 1024 partials: 25 us per call, 6 % of the StereoBase AMP step.)
__global__ __launch_bounds__(256) void channel_sums_final_kernel(const float* __restrict__ ws, float* __restrict__ out, int C, int nwg, int nwhich) {
    const int Cq = (C + 3) / 4;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);            // (which, quad), wave-uniform
    if (i >= nwhich * Cq) return;
    const int which = i / Cq, q = i - which * Cq;
    const float4* src = reinterpret_cast<const float4*>(ws) + (size_t)which * Cq + q;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int g = lane; g < nwg; g += 64) {
        const float4 u = src[(size_t)g * 2 * Cq];
        a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        a.x += __shfl_xor(a.x, m, 64); a.y += __shfl_xor(a.y, m, 64); a.z += __shfl_xor(a.z, m, 64); a.w += __shfl_xor(a.w, m, 64);
    }
    if (lane == 0) {
        const float r[4] = {a.x, a.y, a.z, a.w};
        for (int e = 0; e < 4; ++e)
            if (q * 4 + e < C) out[(size_t)which * C + q * 4 + e] = r[e];
    }
}

// out[p][c] = u[p][c] * a[c] + (v ? v[p][c] * b[c] : 0) + c0[c]  (+ ReLU) on channels-last rows: the normalise step of a training-mode
// BatchNorm (u = x, a = gamma * invstd, c0 = beta - mean * a) and its input gradient (u = dy, v = x: dx = dy * a + x * b + c0 with the
// batch sums folded into b and c0).  One thread per (position, channel quad), 8- / 16-byte accesses.
template <int UF16, int VF16, int WITH_V>
__global__ __launch_bounds__(256) void channel_affine_kernel(const void* __restrict__ u_, const void* __restrict__ v_, void* __restrict__ out_,
                                                             const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c0,
                                                             long long P, int C, int uCs, int vCs, int oCs, int relu) {
    const int Cq = (C + 3) / 4;
    const long long total = P * Cq;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pp = i / Cq; const int q = (int)(i - pp * Cq);
        const int c = q * 4, nc = C - c;
        float uu[4] = {0.f, 0.f, 0.f, 0.f}, vv[4] = {0.f, 0.f, 0.f, 0.f};
        auto ld = [&](const void* base, int f16, int cs, float* dst) {
            if (f16) {
                const _Float16* src = reinterpret_cast<const _Float16*>(base) + pp * cs + c;
                if (nc >= 4) {
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    const uint2 w = *reinterpret_cast<const uint2*>(src);
                    const h2 x0 = __builtin_bit_cast(h2, w.x), x1 = __builtin_bit_cast(h2, w.y);
                    dst[0] = (float)x0[0]; dst[1] = (float)x0[1]; dst[2] = (float)x1[0]; dst[3] = (float)x1[1];
                } else for (int e = 0; e < nc; ++e) dst[e] = (float)src[e];
            } else {
                const float* src = reinterpret_cast<const float*>(base) + pp * cs + c;
                if (nc >= 4) { const float4 w = *reinterpret_cast<const float4*>(src); dst[0] = w.x; dst[1] = w.y; dst[2] = w.z; dst[3] = w.w; }
                else for (int e = 0; e < nc; ++e) dst[e] = src[e];
            }
        };
        ld(u_, UF16, uCs, uu);
        if (WITH_V) ld(v_, VF16, vCs, vv);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ce = (e < nc) ? c + e : c;
            float r = uu[e] * a[ce];
            if (WITH_V) r += vv[e] * b[ce];
            r += c0[ce];
            if (relu) r = fmaxf(r, 0.f);
            o[e] = r;
        }
        if (UF16) {
            _Float16* dst = reinterpret_cast<_Float16*>(out_) + pp * oCs + c;
            if (nc >= 4) {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const h2 x0 = {(_Float16)o[0], (_Float16)o[1]}, x1 = {(_Float16)o[2], (_Float16)o[3]};
                *reinterpret_cast<uint2*>(dst) = make_uint2(__builtin_bit_cast(unsigned, x0), __builtin_bit_cast(unsigned, x1));
            } else for (int e = 0; e < nc; ++e) dst[e] = (_Float16)o[e];
        } else {
            float* dst = reinterpret_cast<float*>(out_) + pp * oCs + c;
            if (nc >= 4) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            else for (int e = 0; e < nc; ++e) dst[e] = o[e];
        }
    }
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


static int channel_sums_wgs(long long P, int C) {
    const int Cq = (C + 3) / 4, PL = 256 / (Cq > 0 ? Cq : 1);
    long long n = P / ((long long)(PL > 0 ? PL : 1) * 4);       // at least ~4 positions per thread (the second stage is one wave per channel quad: partials are cheap)
    if (n > 1024) n = 1024;
    if (n < 1) n = 1;
    return (int)n;
}

extern "C" size_t osa_channel_sums_workspace_bytes(long long P, int C) {
    if (P <= 0 || C <= 0 || C > 1024) return 0;
    return (size_t)channel_sums_wgs(P, C) * 2 * ((C + 3) / 4) * 4 * sizeof(float);
}

static int channel_sums_impl(const void* const* dys, int n_items, int dy_f16, int dy_cs, const void* x, int x_f16, int x_cs, const float* x_shift,
                             const float* dx_scale, void* dx, int dx_cs, long long P, int C,
                             float* out, float* workspace, size_t workspace_bytes, void* stream) {
    OSA_REQUIRE(dys && out && workspace && n_items >= 1 && n_items <= 24, "channel_sums: NULL pointer, or %d items outside 1..24", n_items);
    OSA_REQUIRE(P > 0 && C > 0 && C <= 1024, "channel_sums: bad dims P=%lld C=%d (C <= 1024)", P, C);
    OSA_REQUIRE(dy_cs >= C && (x == nullptr || x_cs >= C) && (dx == nullptr || dx_cs >= C), "channel_sums: channel stride < C");
    // full quads are loaded / stored as 8- or 16-byte vectors
    auto vec_ok = [&](const void* t, int f16, int cs) { return cs % 4 == 0 && ((size_t)t & (f16 ? 7 : 15)) == 0; };
    ChannelSumsTab tab;
    memset(&tab, 0, sizeof(tab));
    for (int i = 0; i < n_items; ++i) {
        OSA_REQUIRE(dys[i] && vec_ok(dys[i], dy_f16, dy_cs), "channel_sums: tensors need channel strides %% 4 == 0 and 16-byte (fp16: 8-byte) alignment");
        tab.p[i] = dys[i];
    }
    OSA_REQUIRE((x == nullptr || vec_ok(x, x_f16, x_cs)) && (dx == nullptr || vec_ok(dx, dy_f16, dx_cs)),
                "channel_sums: tensors need channel strides %% 4 == 0 and 16-byte (fp16: 8-byte) alignment");
    OSA_REQUIRE((dx == nullptr) == (dx_scale == nullptr), "channel_sums: dx and dx_scale come together");
    OSA_REQUIRE(n_items == 1 || (x == nullptr && dx == nullptr), "channel_sums: a tensor list has sums of dy only");
    const size_t need = (size_t)n_items * osa_channel_sums_workspace_bytes(P, C);
    OSA_REQUIRE(workspace_bytes >= need && ((size_t)workspace & 15) == 0, "channel_sums: workspace of %zu B needed (got %zu)", need, workspace_bytes);
    const int nwg = channel_sums_wgs(P, C);
    const long long per = (P + nwg - 1) / nwg;
    hipStream_t st = (hipStream_t)stream;
    const int key = (dy_f16 ? 8 : 0) | (x ? ((x_f16 ? 4 : 0) | 2) : 0) | (dx ? 1 : 0);
#define OSA_CS(DYF, XF, WX, WD) hipLaunchKernelGGL((channel_sums_partial_kernel<DYF, XF, WX, WD>), dim3(nwg, n_items), dim3(256), 0, st, tab, x, dx, x_shift, dx_scale, workspace, P, C, dy_cs, x_cs, dx_cs, per)
    switch (key) {
        case 0: OSA_CS(0, 0, 0, 0); break;   case 1: OSA_CS(0, 0, 0, 1); break;
        case 2: OSA_CS(0, 0, 1, 0); break;   case 3: OSA_CS(0, 0, 1, 1); break;
        case 6: OSA_CS(0, 1, 1, 0); break;   case 7: OSA_CS(0, 1, 1, 1); break;
        case 8: OSA_CS(1, 0, 0, 0); break;   case 9: OSA_CS(1, 0, 0, 1); break;
        case 10: OSA_CS(1, 0, 1, 0); break;  case 11: OSA_CS(1, 0, 1, 1); break;
        case 14: OSA_CS(1, 1, 1, 0); break;  case 15: OSA_CS(1, 1, 1, 1); break;
        default: OSA_REQUIRE(false, "channel_sums: x_f16 without x");
    }
#undef OSA_CS
    const int nwhich = x ? 2 : 1;
    hipLaunchKernelGGL(channel_sums_final_kernel, dim3(cdiv(nwhich * ((C + 3) / 4), 4)), dim3(256), 0, st, workspace, out, C, nwg * n_items, nwhich);
    OSA_LAUNCH_CHECK("channel_sums");
    return 0;
}

extern "C" int osa_channel_sums(const void* dy, int dy_f16, int dy_cs, const void* x, int x_f16, int x_cs, const float* x_shift,
                                const float* dx_scale, void* dx, int dx_cs, long long P, int C,
                                float* out, float* workspace, size_t workspace_bytes, void* stream) {
    return channel_sums_impl(&dy, 1, dy_f16, dy_cs, x, x_f16, x_cs, x_shift, dx_scale, dx, dx_cs, P, C, out, workspace, workspace_bytes, stream);
}

/* sums of dy over a LIST of n_items (<= 24) equally shaped tensors of P positions each (the queued output gradients of one biased convolution);
 * workspace: n_items x osa_channel_sums_workspace_bytes(P, C) */
extern "C" int osa_channel_sums_multi(const void* const* dys, int n_items, int dy_f16, int dy_cs, long long P, int C,
                                      float* out, float* workspace, size_t workspace_bytes, void* stream) {
    return channel_sums_impl(dys, n_items, dy_f16, dy_cs, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, P, C, out, workspace, workspace_bytes, stream);
}


extern "C" int osa_channel_affine(const void* u, int u_f16, int u_cs, const void* v, int v_f16, int v_cs,
                                  const float* a, const float* b, const float* c0, void* out, int out_cs,
                                  long long P, int C, int relu, void* stream) {
    OSA_REQUIRE(u && a && c0 && out && (v == nullptr || b != nullptr), "channel_affine: NULL pointer");
    OSA_REQUIRE(P > 0 && C > 0 && u_cs >= C && out_cs >= C && (v == nullptr || v_cs >= C), "channel_affine: bad dims / channel stride < C");
    auto vec_ok = [&](const void* t, int f16, int cs) { return cs % 4 == 0 && ((size_t)t & (f16 ? 7 : 15)) == 0; };
    OSA_REQUIRE(vec_ok(u, u_f16, u_cs) && vec_ok(out, u_f16, out_cs) && (v == nullptr || vec_ok(v, v_f16, v_cs)),
                "channel_affine: tensors need channel strides %% 4 == 0 and 16-byte (fp16: 8-byte) alignment");
    const long long total = P * ((C + 3) / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipStream_t st = (hipStream_t)stream;
    const int key = (u_f16 ? 4 : 0) | (v ? ((v_f16 ? 2 : 0) | 1) : 0);
#define OSA_CA(UF, VF, WV) hipLaunchKernelGGL((channel_affine_kernel<UF, VF, WV>), dim3((unsigned)blocks), dim3(256), 0, st, u, v, out, a, b, c0, P, C, u_cs, v_cs, out_cs, relu)
    switch (key) {
        case 0: OSA_CA(0, 0, 0); break;  case 1: OSA_CA(0, 0, 1); break;  case 3: OSA_CA(0, 1, 1); break;
        case 4: OSA_CA(1, 0, 0); break;  case 5: OSA_CA(1, 0, 1); break;  case 7: OSA_CA(1, 1, 1); break;
        default: OSA_REQUIRE(false, "channel_affine: v_f16 without v");
    }
#undef OSA_CA
    OSA_LAUNCH_CHECK("channel_affine");
    return 0;
}
