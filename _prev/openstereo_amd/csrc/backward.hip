// Backward kernels of the memory-bound hot-path ops, reference NCDHW layouts.
// The reference gets these from autograd through its slice-assignment loops; here they are explicit:
//
//  volume (gwc part)   dL[b,c,h,w]  = (1/K) sum_{d<=min(w,D-1)} dV[b,g,d,h,w]   * R[b,c,h,w-d]
//                      dR[b,c,h,w'] = (1/K) sum_{d<D, w'+d<W}   dV[b,g,d,h,w'+d] * L[b,c,h,w'+d]
//  volume (concat)     dL[b,c,h,w]  = sum_{d<=w (all d if left unmasked)} dV[b,c,d,h,w]
//                      dR[b,c,h,w'] = sum_{d, w'+d<W} dV[b,C+c,d,h,w'+d]
//  soft-argmin         dprob[b,d,h,w] = d * dout[b,h,w]
//  softmax+soft-argmin dcost[b,d,h,w] = p_d * (d - disp) * dout          (p = softmax(cost))
//  upsample+softmax+soft-argmin: the same g_d at full resolution, pushed back through the transposed
//                      trilinear interpolation (8 corner weights) with float atomics into the low-res cost.
// All are streaming kernels with lanes along w.
#include "osa_common.h"

namespace osa {

struct VolBwdArgs {
    const float* dV; const float* L; const float* R; float* dL; float* dR;
    int B, C, H, W, D, G, K, VC, coff;   // gwc: channels [coff, coff+G) of dV
    int concat, mask_left;               // concat: C = per-side channels, left at coff, right at coff+C
};

__global__ __launch_bounds__(256) void volume_bwd_kernel(const VolBwdArgs p) {
    const size_t plane = (size_t)p.H * p.W;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;          // over B*C*H*W
    if (i >= (size_t)p.B * p.C * plane) return;
    const int w = i % p.W; size_t r = i / p.W;
    const int h = r % p.H; r /= p.H;
    const int c = r % p.C; const int b = r / p.C;
    const size_t hw = (size_t)h * p.W + w;
    const size_t dstride = plane;                                     // dV stride along d
    float gl = 0.f, gr = 0.f;
    if (!p.concat) {
        const int g = c / p.K;
        const float* dv = p.dV + (((size_t)b * p.VC + p.coff + g) * p.D) * plane + hw;
        const float* Lr = p.L + ((size_t)b * p.C + c) * plane + (size_t)h * p.W;
        const float* Rr = p.R + ((size_t)b * p.C + c) * plane + (size_t)h * p.W;
        for (int d = 0; d < p.D; ++d) {
            if (d <= w) gl = fmaf(dv[(size_t)d * dstride], Rr[w - d], gl);
            if (w + d < p.W) gr = fmaf(dv[(size_t)d * dstride + d], Lr[w + d], gr);
        }
        const float invK = 1.0f / (float)p.K;
        gl *= invK; gr *= invK;
    } else {
        const float* dvl = p.dV + (((size_t)b * p.VC + p.coff + c) * p.D) * plane + hw;
        const float* dvr = p.dV + (((size_t)b * p.VC + p.coff + p.C + c) * p.D) * plane + hw;
        for (int d = 0; d < p.D; ++d) {
            if (d <= w || !p.mask_left) gl += dvl[(size_t)d * dstride];
            if (w + d < p.W) gr += dvr[(size_t)d * dstride + d];
        }
    }
    p.dL[i] = gl;
    p.dR[i] = gr;
}

__global__ __launch_bounds__(256) void softargmin_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dprob,
                                                             int D, long long HW, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;    // over B*D*H*W
    if (i >= total) return;
    const long long hw = i % HW; const long long bd = i / HW;
    const int d = (int)(bd % D); const long long b = bd / D;
    dprob[i] = (float)d * dout[b * HW + hw];
}

__global__ __launch_bounds__(256) void softmax_softargmin_bwd_kernel(const float* __restrict__ cost, const float* __restrict__ dout,
                                                                     float* __restrict__ dcost, int D, long long HW, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;    // over B*H*W
    if (i >= total) return;
    const long long b = i / HW, hw = i - b * HW;
    const float* c = cost + (size_t)b * D * HW + hw;
    float m = -INFINITY;
    for (int d = 0; d < D; ++d) m = fmaxf(m, c[(size_t)d * HW]);
    float se = 0.f, sd = 0.f;
    for (int d = 0; d < D; ++d) { const float e = expf(c[(size_t)d * HW] - m); se += e; sd = fmaf(e, (float)d, sd); }
    const float inv = 1.f / se, disp = sd * inv, g = dout[i];
    float* dc = dcost + (size_t)b * D * HW + hw;
    for (int d = 0; d < D; ++d) dc[(size_t)d * HW] = expf(c[(size_t)d * HW] - m) * inv * ((float)d - disp) * g;
}

__device__ __forceinline__ void src_index_b(int dst, float scale, int align, int in_size, int& i0, int& i1, float& l1) {
    float s;
    if (align) s = scale * (float)dst;
    else { s = scale * ((float)dst + 0.5f) - 0.5f; s = s < 0.f ? 0.f : s; }
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
    l1 = s - (float)i0;
}

struct UpBwdArgs {
    const float* cost; const float* dout; float* dcost;   // dcost must be zero-initialised
    int B, Dl, Hl, Wl, D, H, W, align;
    float sd, sh, sw;
};

// one thread per output pixel: recompute its D up-sampled costs, softmax and disparity, fold
// g_d = p_d (d - disp) dout along d into the Dl low-res planes (LDS, [dl][thread]), then add the 4
// (y,x) corner contributions with float atomics.
__global__ __launch_bounds__(256) void upsample_softargmin_bwd_kernel(const UpBwdArgs p) {
    extern __shared__ float sh[];            // cl[Dl][256] then gl[Dl][256]
    float* cl = sh; float* gl = sh + (size_t)p.Dl * 256;
    const int tid = threadIdx.x;
    const long long HW = (long long)p.H * p.W;
    const long long i = (long long)blockIdx.x * 256 + tid;
    const bool live = i < (long long)p.B * HW;
    const long long ii = live ? i : 0;
    const int b = (int)(ii / HW);
    const int hw = (int)(ii - (long long)b * HW);
    const int y = hw / p.W, x = hw - y * p.W;
    int y0, y1, x0, x1; float ly, lx;
    src_index_b(y, p.sh, p.align, p.Hl, y0, y1, ly);
    src_index_b(x, p.sw, p.align, p.Wl, x0, x1, lx);
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    const size_t plane = (size_t)p.Hl * p.Wl;
    const float* c = p.cost + (size_t)b * p.Dl * plane;
    const size_t o00 = (size_t)y0 * p.Wl + x0, o01 = (size_t)y0 * p.Wl + x1, o10 = (size_t)y1 * p.Wl + x0, o11 = (size_t)y1 * p.Wl + x1;
    float m = -INFINITY;
    for (int dl = 0; dl < p.Dl; ++dl) {
        const float* cp = c + (size_t)dl * plane;
        const float v = w00 * cp[o00] + w01 * cp[o01] + w10 * cp[o10] + w11 * cp[o11];
        cl[dl * 256 + tid] = v; gl[dl * 256 + tid] = 0.f;
        m = fmaxf(m, v);
    }
    float se = 0.f, sdisp = 0.f;
    for (int d = 0; d < p.D; ++d) {
        int d0, d1; float ld;
        src_index_b(d, p.sd, p.align, p.Dl, d0, d1, ld);
        const float e = expf((1.f - ld) * cl[d0 * 256 + tid] + ld * cl[d1 * 256 + tid] - m);
        se += e; sdisp = fmaf(e, (float)d, sdisp);
    }
    const float inv = 1.f / se, disp = sdisp * inv;
    const float g = live ? p.dout[i] : 0.f;
    for (int d = 0; d < p.D; ++d) {
        int d0, d1; float ld;
        src_index_b(d, p.sd, p.align, p.Dl, d0, d1, ld);
        const float e = expf((1.f - ld) * cl[d0 * 256 + tid] + ld * cl[d1 * 256 + tid] - m);
        const float gd = e * inv * ((float)d - disp) * g;
        gl[d0 * 256 + tid] += (1.f - ld) * gd;
        gl[d1 * 256 + tid] += ld * gd;
    }
    if (!live) return;
    float* dc = p.dcost + (size_t)b * p.Dl * plane;
    for (int dl = 0; dl < p.Dl; ++dl) {
        const float gv = gl[dl * 256 + tid];
        float* dp = dc + (size_t)dl * plane;
        atomicAdd(dp + o00, w00 * gv); atomicAdd(dp + o01, w01 * gv);
        atomicAdd(dp + o10, w10 * gv); atomicAdd(dp + o11, w11 * gv);
    }
}


// ---- two-pass, atomic-free form (osa_upsample_softargmin_bwd_ws_f32) ---------------------------------------------------------------
// The one-kernel form above scatters 4 * Dl float atomics per output pixel (25 M of them for one 256x512 pair, ~25 pixels contending for
// every low-res cell): 0.75 ms per head, 6.7 % of a GwcNet training step, and a run-dependent summation order.
// pass 1 (fold): one thread per output pixel, as above, but the Dl folded gradients go to a scratch tensor G[b][dl][y][x] (coalesced).
// pass 2 (gather): one thread per low-res cell sums w_y * w_x * G over the output pixels whose bilinear footprint contains the cell, in
// a fixed order -- deterministic, no zero-fill, no atomics.
template <int NT>
__global__ __launch_bounds__(NT) void upsample_softargmin_bwd_fold_kernel(const UpBwdArgs p, float* __restrict__ G) {
    extern __shared__ float sh[];            // cl[Dl][NT] then gl[Dl][NT]
    float* cl = sh; float* gl = sh + (size_t)p.Dl * NT;
    const int tid = threadIdx.x;
    const long long HW = (long long)p.H * p.W;
    const long long i = (long long)blockIdx.x * NT + tid;
    const bool live = i < (long long)p.B * HW;
    const long long ii = live ? i : 0;
    const int b = (int)(ii / HW);
    const int hw = (int)(ii - (long long)b * HW);
    const int y = hw / p.W, x = hw - y * p.W;
    int y0, y1, x0, x1; float ly, lx;
    src_index_b(y, p.sh, p.align, p.Hl, y0, y1, ly);
    src_index_b(x, p.sw, p.align, p.Wl, x0, x1, lx);
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    const size_t plane = (size_t)p.Hl * p.Wl;
    const float* c = p.cost + (size_t)b * p.Dl * plane;
    const size_t o00 = (size_t)y0 * p.Wl + x0, o01 = (size_t)y0 * p.Wl + x1, o10 = (size_t)y1 * p.Wl + x0, o11 = (size_t)y1 * p.Wl + x1;
    float m = -INFINITY;
    for (int dl = 0; dl < p.Dl; ++dl) {
        const float* cp = c + (size_t)dl * plane;
        const float v = w00 * cp[o00] + w01 * cp[o01] + w10 * cp[o10] + w11 * cp[o11];
        cl[dl * NT + tid] = v; gl[dl * NT + tid] = 0.f;
        m = fmaxf(m, v);
    }
    float se = 0.f, sdisp = 0.f;
    for (int d = 0; d < p.D; ++d) {
        int d0, d1; float ld;
        src_index_b(d, p.sd, p.align, p.Dl, d0, d1, ld);
        const float e = expf((1.f - ld) * cl[d0 * NT + tid] + ld * cl[d1 * NT + tid] - m);
        se += e; sdisp = fmaf(e, (float)d, sdisp);
    }
    const float inv = 1.f / se, disp = sdisp * inv;
    const float g = live ? p.dout[i] : 0.f;
    for (int d = 0; d < p.D; ++d) {
        int d0, d1; float ld;
        src_index_b(d, p.sd, p.align, p.Dl, d0, d1, ld);
        const float e = expf((1.f - ld) * cl[d0 * NT + tid] + ld * cl[d1 * NT + tid] - m);
        const float gd = e * inv * ((float)d - disp) * g;
        gl[d0 * NT + tid] += (1.f - ld) * gd;
        gl[d1 * NT + tid] += ld * gd;
    }
    if (!live) return;
    float* gp = G + (size_t)b * p.Dl * HW + hw;
    for (int dl = 0; dl < p.Dl; ++dl) gp[(size_t)dl * HW] = gl[dl * NT + tid];
}

// output positions whose source interval can contain low-res index `il` (a conservative range; the exact test is src_index_b)
__device__ __forceinline__ void footprint(int il, float scale, int align, int out_size, int& lo, int& hi) {
    if (scale <= 0.f) { lo = 0; hi = out_size - 1; return; }
    const float off = align ? 0.f : 0.5f;
    const float a = ((float)il - 1.f + off) / scale - off, b = ((float)il + 1.f + off) / scale - off;
    lo = (int)floorf(a) - 1; hi = (int)ceilf(b) + 1;
    if (lo < 0) lo = 0;
    if (hi > out_size - 1) hi = out_size - 1;
}

__global__ __launch_bounds__(256) void upsample_softargmin_bwd_gather_kernel(const UpBwdArgs p, const float* __restrict__ G) {
    const long long total = (long long)p.B * p.Dl * p.Hl * p.Wl;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int xl = (int)(i % p.Wl); long long r = i / p.Wl;
    const int yl = (int)(r % p.Hl); r /= p.Hl;               // r = b * Dl + dl
    int ylo, yhi, xlo, xhi;
    footprint(yl, p.sh, p.align, p.H, ylo, yhi);
    footprint(xl, p.sw, p.align, p.W, xlo, xhi);
    const float* g = G + (size_t)r * p.H * p.W;
    float acc = 0.f;
    for (int y = ylo; y <= yhi; ++y) {
        int y0, y1; float ly;
        src_index_b(y, p.sh, p.align, p.Hl, y0, y1, ly);
        const float wy = ((y0 == yl) ? (1.f - ly) : 0.f) + ((y1 == yl) ? ly : 0.f);
        if (wy == 0.f) continue;
        const float* gr = g + (size_t)y * p.W;
        float row = 0.f;
        for (int x = xlo; x <= xhi; ++x) {
            int x0, x1; float lx;
            src_index_b(x, p.sw, p.align, p.Wl, x0, x1, lx);
            const float wx = ((x0 == xl) ? (1.f - lx) : 0.f) + ((x1 == xl) ? lx : 0.f);
            row = fmaf(wx, gr[x], row);
        }
        acc = fmaf(wy, row, acc);
    }
    p.dcost[i] = acc;
}

}  // namespace osa

using namespace osa;

extern "C" int osa_build_volume_bwd_f32(const float* dvol, const float* left, const float* right,
                                        float* dleft, float* dright,
                                        int B, int C, int H, int W, int maxdisp, int num_groups,
                                        int concat, int mask_left_concat, int vol_channels, int c_off,
                                        void* stream) {
    OSA_REQUIRE(dvol && dleft && dright, "build_volume_bwd: NULL pointer");
    OSA_REQUIRE(concat || (left && right), "build_volume_bwd: the gwc part needs the forward features");
    OSA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && maxdisp > 0, "build_volume_bwd: bad dims");
    VolBwdArgs a;
    a.dV = dvol; a.L = left; a.R = right; a.dL = dleft; a.dR = dright;
    a.B = B; a.C = C; a.H = H; a.W = W; a.D = maxdisp; a.VC = vol_channels; a.coff = c_off;
    a.concat = concat ? 1 : 0; a.mask_left = mask_left_concat ? 1 : 0;
    a.G = num_groups; a.K = 1;
    if (!concat) {
        OSA_REQUIRE(num_groups > 0 && C % num_groups == 0, "build_volume_bwd: C=%d not divisible by groups=%d", C, num_groups);
        a.K = C / num_groups;
        OSA_REQUIRE(c_off + num_groups <= vol_channels, "build_volume_bwd: channel range exceeds vol_channels");
    } else {
        OSA_REQUIRE(c_off + 2 * C <= vol_channels, "build_volume_bwd: channel range exceeds vol_channels");
    }
    const long long total = (long long)B * C * H * W;
    hipLaunchKernelGGL(volume_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, a);
    OSA_LAUNCH_CHECK("build_volume_bwd");
    return 0;
}

extern "C" int osa_softargmin_bwd_f32(const float* dout, float* dprob, int B, int D, int H, int W, void* stream) {
    OSA_REQUIRE(dout && dprob, "softargmin_bwd: NULL pointer");
    const long long HW = (long long)H * W, total = HW * B * D;
    hipLaunchKernelGGL(softargmin_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dout, dprob, D, HW, total);
    OSA_LAUNCH_CHECK("softargmin_bwd");
    return 0;
}

extern "C" int osa_softmax_softargmin_bwd_f32(const float* cost, const float* dout, float* dcost,
                                              int B, int D, int H, int W, void* stream) {
    OSA_REQUIRE(cost && dout && dcost, "softmax_softargmin_bwd: NULL pointer");
    const long long HW = (long long)H * W, total = HW * B;
    hipLaunchKernelGGL(softmax_softargmin_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       cost, dout, dcost, D, HW, total);
    OSA_LAUNCH_CHECK("softmax_softargmin_bwd");
    return 0;
}

extern "C" int osa_upsample_softargmin_bwd_f32(const float* cost_lowres, const float* dout, float* dcost_lowres,
                                               int B, int Dl, int Hl, int Wl, int D, int H, int W,
                                               int align_corners, void* stream) {
    OSA_REQUIRE(cost_lowres && dout && dcost_lowres, "upsample_softargmin_bwd: NULL pointer");
    const size_t lds = (size_t)Dl * 256 * sizeof(float) * 2;
    OSA_REQUIRE(lds <= 160 * 1024, "upsample_softargmin_bwd: Dl=%d too large for LDS", Dl);
    UpBwdArgs a;
    a.cost = cost_lowres; a.dout = dout; a.dcost = dcost_lowres;
    a.B = B; a.Dl = Dl; a.Hl = Hl; a.Wl = Wl; a.D = D; a.H = H; a.W = W; a.align = align_corners ? 1 : 0;
    auto sc = [&](int in, int out) { return a.align ? ((out > 1) ? (float)(in - 1) / (float)(out - 1) : 0.f) : (float)in / (float)out; };
    a.sd = sc(Dl, D); a.sh = sc(Hl, H); a.sw = sc(Wl, W);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(dcost_lowres, 0, (size_t)B * Dl * Hl * Wl * sizeof(float), st);
    OSA_REQUIRE(e == hipSuccess, "upsample_softargmin_bwd: memset failed: %s", hipGetErrorString(e));
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)upsample_softargmin_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(upsample_softargmin_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), lds, st, a);
    OSA_LAUNCH_CHECK("upsample_softargmin_bwd");
    return 0;
}

extern "C" size_t osa_upsample_softargmin_bwd_workspace_bytes(int B, int Dl, int H, int W) {
    if (B <= 0 || Dl <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * Dl * H * W * sizeof(float);
}

extern "C" int osa_upsample_softargmin_bwd_ws_f32(const float* cost_lowres, const float* dout, float* dcost_lowres,
                                                  int B, int Dl, int Hl, int Wl, int D, int H, int W,
                                                  int align_corners, void* workspace, size_t workspace_bytes, void* stream) {
    OSA_REQUIRE(cost_lowres && dout && dcost_lowres && workspace, "upsample_softargmin_bwd_ws: NULL pointer");
    OSA_REQUIRE(workspace_bytes >= osa_upsample_softargmin_bwd_workspace_bytes(B, Dl, H, W) && ((size_t)workspace & 15) == 0,
                "upsample_softargmin_bwd_ws: workspace too small or misaligned (osa_upsample_softargmin_bwd_workspace_bytes)");
    constexpr int NT = 128;
    const size_t lds = (size_t)Dl * NT * sizeof(float) * 2;
    OSA_REQUIRE(lds <= 160 * 1024, "upsample_softargmin_bwd_ws: Dl=%d too large for LDS", Dl);
    UpBwdArgs a;
    a.cost = cost_lowres; a.dout = dout; a.dcost = dcost_lowres;
    a.B = B; a.Dl = Dl; a.Hl = Hl; a.Wl = Wl; a.D = D; a.H = H; a.W = W; a.align = align_corners ? 1 : 0;
    auto sc = [&](int in, int out) { return a.align ? ((out > 1) ? (float)(in - 1) / (float)(out - 1) : 0.f) : (float)in / (float)out; };
    a.sd = sc(Dl, D); a.sh = sc(Hl, H); a.sw = sc(Wl, W);
    hipStream_t st = (hipStream_t)stream;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)upsample_softargmin_bwd_fold_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float* G = static_cast<float*>(workspace);
    hipLaunchKernelGGL(upsample_softargmin_bwd_fold_kernel<NT>, dim3(cdiv((long long)B * H * W, NT)), dim3(NT), lds, st, a, G);
    OSA_LAUNCH_CHECK("upsample_softargmin_bwd_ws (fold)");
    hipLaunchKernelGGL(upsample_softargmin_bwd_gather_kernel, dim3(cdiv((long long)B * Dl * Hl * Wl, 256)), dim3(256), 0, st, a, (const float*)G);
    OSA_LAUNCH_CHECK("upsample_softargmin_bwd_ws (gather)");
    return 0;
}
