// Cost-volume constructors for gfx950.
//
// One pass builds the group-wise-correlation part and the concatenation part and
// writes them straight into the (already concatenated) volume buffer, so the
// reference's torch.cat (gwcnet_cost_processor.py:65) never happens.
//
//  * NDHWC kernel (engine layout): one workgroup per (b, h, 16-pixel w tile, 16-disparity
//    chunk).  The left tile and the sliding right window are staged channel-permuted in
//    LDS so that lane g reads its K channels as float4s, conflict free; every wave store
//    instruction writes one voxel's whole channel vector (G+2Cc floats, 256 B for GwcNet).
//    HBM-write bound: 4 B written per 8 FMAs.
//  * NCDHW kernel (reference layout, drop-in functions): lanes run along w, so loads and
//    stores are coalesced rows; operands come from L1/L2.
#include "osa_common.h"

namespace osa {

// ------------------------------------------------------------------ NDHWC ----
struct VolArgs {
    const float* lg; const float* rg; const float* lc; const float* rc;
    float* vol;
    int B, C, Cc, H, W, D, G, K;
    int VC, coff;          // volume channel count / first channel written
    int gstride, cstride;  // >0: features are NHWC with this many floats per pixel (engine backbone); 0: NCHW
    int RS;                // LDS row stride (floats per pixel)
    int catbase;           // float offset of the concat channels inside an LDS row
    int nWt, nDch;         // tiles along w, chunks along d
    int mask_left;
};

constexpr int VOL_WT = 16;   // output pixels per tile
constexpr int VOL_DCH = 16;  // disparities per chunk

template <int QG>  // K/4 : float4s per group
__global__ __launch_bounds__(256) void build_volume_ndhwc_kernel(const VolArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NPXL = VOL_WT;
    constexpr int NPXR = VOL_WT + VOL_DCH - 1;
    float* Ls = smem;                    // [NPXL][RS]
    float* Rs = smem + NPXL * p.RS;      // [NPXR][RS]

    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int dch = bid % p.nDch; bid /= p.nDch;
    const int wt = bid % p.nWt;   bid /= p.nWt;
    const int h = bid % p.H;
    const int b = bid / p.H;
    const int w0 = wt * VOL_WT, d0 = dch * VOL_DCH;
    const int wr0 = w0 - (d0 + VOL_DCH - 1);  // first right pixel of the window

    const int tid = threadIdx.x;
    const size_t plane = (size_t)p.H * p.W;
    const int nq_g = QG * p.G;                     // gwc quads per pixel
    const int nq_c = (p.Cc + 3) >> 2;              // concat quads per pixel
    const int nq = nq_g + nq_c;

    // ---- stage: item = (quad, pixel), pixel fastest -> coalesced global reads along w
    auto stage = [&](float* dst, const float* fg, const float* fc, int npx, int wbase) {
        const int items = nq * npx;
        const bool chan_fast = (p.gstride != 0);   // NHWC: consecutive lanes walk the channel quads of a pixel
        for (int it = tid; it < items; it += 256) {
            int qi, px;
            if (chan_fast) { px = it / nq; qi = it - px * nq; }
            else { qi = it / npx; px = it - qi * npx; }
            const int w = wbase + px;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const bool wok = (w >= 0) && (w < p.W);
            int lpos;
            if (qi < nq_g) {
                const int q = qi / p.G, g = qi - q * p.G;
                lpos = qi * 4;
                if (wok) {
                    if (p.gstride) {
                        v = *reinterpret_cast<const float4*>(fg + (((size_t)b * p.H + h) * p.W + w) * p.gstride + g * p.K + q * 4);
                    } else {
                        const float* src = fg + ((size_t)b * p.C + (size_t)g * p.K + q * 4) * plane + (size_t)h * p.W + w;
                        v.x = src[0]; v.y = src[plane]; v.z = src[2 * plane]; v.w = src[3 * plane];
                    }
                }
            } else {
                const int qc = qi - nq_g;
                lpos = p.catbase + qc * 4;
                if (wok) {
                    const int rem = p.Cc - qc * 4;
                    const float* src; size_t st;
                    if (p.cstride) { src = fc + (((size_t)b * p.H + h) * p.W + w) * p.cstride + qc * 4; st = 1; }
                    else { src = fc + ((size_t)b * p.Cc + qc * 4) * plane + (size_t)h * p.W + w; st = plane; }
                    v.x = src[0];
                    if (rem > 1) v.y = src[st];
                    if (rem > 2) v.z = src[2 * st];
                    if (rem > 3) v.w = src[3 * st];
                }
            }
            *reinterpret_cast<float4*>(dst + (size_t)px * p.RS + lpos) = v;
        }
    };
    stage(Ls, p.lg, p.lc, NPXL, w0);
    stage(Rs, p.rg, p.rc, NPXR, wr0);
    __syncthreads();

    // ---- compute: lane = output channel, wave = pixel
    const int lane = tid & 63, wave = tid >> 6;
    const int nch = p.G + 2 * p.Cc;
    const float invK = 1.0f / (float)p.K;
    (void)invK;
    for (int c = lane; c < nch; c += 64) {
        for (int wl = wave; wl < VOL_WT; wl += 4) {
            const int w = w0 + wl;
            if (w >= p.W) break;
            const float* lrow = Ls + wl * p.RS;
            float4 lq[QG];
            float lcat = 0.f;
            int kind;  // 0 gwc, 1 left concat, 2 right concat
            if (c < p.G) {
                kind = 0;
#pragma unroll
                for (int q = 0; q < QG; ++q) lq[q] = *reinterpret_cast<const float4*>(lrow + (q * p.G + c) * 4);
            } else if (c < p.G + p.Cc) {
                kind = 1;
                lcat = lrow[p.catbase + (c - p.G)];
            } else {
                kind = 2;
            }
#pragma unroll 4
            for (int dd = 0; dd < VOL_DCH; ++dd) {
                const int d = d0 + dd;
                if (d >= p.D) break;
                const float* rrow = Rs + (wl + VOL_DCH - 1 - dd) * p.RS;
                float v = 0.f;
                const bool valid = (w >= d);
                if (kind == 0) {
                    if (valid) {
                        float s = 0.f;
#pragma unroll
                        for (int q = 0; q < QG; ++q) {
                            const float4 r = *reinterpret_cast<const float4*>(rrow + (q * p.G + c) * 4);
                            s = fmaf(lq[q].x, r.x, s); s = fmaf(lq[q].y, r.y, s);
                            s = fmaf(lq[q].z, r.z, s); s = fmaf(lq[q].w, r.w, s);
                        }
                        v = s / (float)p.K;
                    }
                } else if (kind == 1) {
                    v = (valid || !p.mask_left) ? lcat : 0.f;
                } else {
                    if (valid) v = rrow[p.catbase + (c - p.G - p.Cc)];
                }
                const size_t vox = (((size_t)b * p.D + d) * p.H + h) * p.W + w;
                p.vol[vox * p.VC + p.coff + c] = v;
            }
        }
    }
}

// ------------------------------------------------------------------ NCDHW ----
// One thread per output element, w fastest. grid.y = channel, grid.z = b*D+d.
struct VolNArgs {
    const float* lg; const float* rg; const float* lc; const float* rc;
    float* vol;
    int B, C, Cc, H, W, D, G, K;
    int VC, coff;
    int mask_left;
};

__global__ __launch_bounds__(256) void build_volume_ncdhw_kernel(const VolNArgs p) {
    const int c = blockIdx.y;
    const int bd = blockIdx.z;
    const int b = bd / p.D, d = bd - b * p.D;
    const int hw = blockIdx.x * 256 + threadIdx.x;
    if (hw >= p.H * p.W) return;
    const int h = hw / p.W, w = hw - h * p.W;
    const size_t plane = (size_t)p.H * p.W;
    float v = 0.f;
    const bool valid = (w >= d);
    if (c < p.G) {
        if (valid) {
            const float* l = p.lg + ((size_t)b * p.C + (size_t)c * p.K) * plane + hw;
            const float* r = p.rg + ((size_t)b * p.C + (size_t)c * p.K) * plane + hw - d;
            float s = 0.f;
            for (int k = 0; k < p.K; ++k) s = fmaf(l[k * plane], r[k * plane], s);
            v = s / (float)p.K;
        }
    } else if (c < p.G + p.Cc) {
        if (valid || !p.mask_left) v = p.lc[((size_t)b * p.Cc + (c - p.G)) * plane + hw];
    } else {
        if (valid) v = p.rc[((size_t)b * p.Cc + (c - p.G - p.Cc)) * plane + hw - d];
    }
    p.vol[((((size_t)b * p.VC + p.coff + c) * p.D + d) * plane) + hw] = v;
}

}  // namespace osa

using namespace osa;

static int build_volume_impl(const float* left_gwc, const float* right_gwc, int C, int num_groups, int gwc_stride,
                             const float* left_cat, const float* right_cat, int Cc, int cat_stride,
                             float* vol, int layout, int vol_channels, int c_off,
                             int B, int H, int W, int maxdisp, int mask_left_concat, void* stream);

extern "C" int osa_build_volume_f32(const float* left_gwc, const float* right_gwc, int C, int num_groups,
                                    const float* left_cat, const float* right_cat, int Cc,
                                    float* vol, int layout, int vol_channels, int c_off,
                                    int B, int H, int W, int maxdisp, int mask_left_concat,
                                    void* stream) {
    return build_volume_impl(left_gwc, right_gwc, C, num_groups, 0, left_cat, right_cat, Cc, 0, vol, layout,
                             vol_channels, c_off, B, H, W, maxdisp, mask_left_concat, stream);
}

extern "C" int osa_build_volume_nhwc_f32(const float* left_gwc, const float* right_gwc, int C, int num_groups, int gwc_stride,
                                         const float* left_cat, const float* right_cat, int Cc, int cat_stride,
                                         float* vol, int vol_channels, int c_off,
                                         int B, int H, int W, int maxdisp, int mask_left_concat, void* stream) {
    OSA_REQUIRE((C == 0 || (gwc_stride >= C && gwc_stride % 4 == 0 && ((size_t)left_gwc & 15) == 0 && ((size_t)right_gwc & 15) == 0)),
                "build_volume_nhwc: gwc features need stride >= C, stride %% 4 == 0 and 16-byte alignment");
    OSA_REQUIRE((Cc == 0 || cat_stride >= Cc), "build_volume_nhwc: concat stride %d < Cc %d", cat_stride, Cc);
    OSA_REQUIRE(C == 0 || (C / (num_groups > 0 ? num_groups : 1)) % 4 == 0, "build_volume_nhwc: channels per group must be a multiple of 4");
    return build_volume_impl(left_gwc, right_gwc, C, num_groups, gwc_stride ? gwc_stride : C, left_cat, right_cat, Cc,
                             cat_stride ? cat_stride : Cc, vol, OSA_NDHWC, vol_channels, c_off, B, H, W, maxdisp,
                             mask_left_concat, stream);
}

static int build_volume_impl(const float* left_gwc, const float* right_gwc, int C, int num_groups, int gwc_stride,
                             const float* left_cat, const float* right_cat, int Cc, int cat_stride,
                             float* vol, int layout, int vol_channels, int c_off,
                             int B, int H, int W, int maxdisp, int mask_left_concat, void* stream) {
    OSA_REQUIRE(vol != nullptr, "build_volume: vol is NULL");
    OSA_REQUIRE(B > 0 && H > 0 && W > 0 && maxdisp > 0, "build_volume: bad dims B=%d H=%d W=%d D=%d", B, H, W, maxdisp);
    OSA_REQUIRE(C >= 0 && Cc >= 0 && (C > 0 || Cc > 0), "build_volume: nothing to build (C=%d Cc=%d)", C, Cc);
    int G = 0, K = 0;
    if (C > 0) {
        OSA_REQUIRE(left_gwc && right_gwc, "build_volume: gwc features NULL");
        OSA_REQUIRE(num_groups > 0 && C % num_groups == 0,
                    "build_volume: C=%d not divisible by num_groups=%d", C, num_groups);  // cost_volume.py:61
        G = num_groups; K = C / G;
    }
    if (Cc > 0) OSA_REQUIRE(left_cat && right_cat, "build_volume: concat features NULL");
    const int nch = G + 2 * Cc;
    OSA_REQUIRE(c_off >= 0 && c_off + nch <= vol_channels,
                "build_volume: channels [%d,%d) exceed vol_channels=%d", c_off, c_off + nch, vol_channels);
    OSA_REQUIRE(layout == OSA_NCDHW || layout == OSA_NDHWC, "build_volume: bad layout %d", layout);
    hipStream_t st = (hipStream_t)stream;

    const bool fast = (layout == OSA_NDHWC) && (G == 0 || (K % 4 == 0 && K / 4 <= 4));
    if (layout == OSA_NDHWC && !fast) {
        set_error("build_volume: NDHWC layout needs channels-per-group in {4,8,12,16} (got K=%d); "
                  "build NCDHW and convert with osa_ncdhw_to_ndhwc_f32", K);
        return -1;
    }
    if (fast) {
        VolArgs a;
        a.lg = left_gwc; a.rg = right_gwc; a.lc = left_cat; a.rc = right_cat; a.vol = vol;
        a.B = B; a.C = C; a.Cc = Cc; a.H = H; a.W = W; a.D = maxdisp; a.G = G; a.K = K;
        a.VC = vol_channels; a.coff = c_off; a.mask_left = mask_left_concat;
        a.gstride = gwc_stride; a.cstride = cat_stride;
        const int QG = (G > 0) ? K / 4 : 1;
        const int gfl = (G > 0) ? QG * G * 4 : 0;
        a.catbase = gfl;
        int rs = gfl + ((Cc + 3) / 4) * 4;
        if (((rs / 4) & 1) == 0) rs += 4;   // row stride / 16B odd -> conflict-free ds_write_b128 staging
        a.RS = rs;
        a.nWt = cdiv(W, VOL_WT); a.nDch = cdiv(maxdisp, VOL_DCH);
        const size_t lds = (size_t)(VOL_WT + VOL_WT + VOL_DCH - 1) * rs * sizeof(float);
        OSA_REQUIRE(lds <= 160 * 1024, "build_volume: %zu B of LDS needed (> 160 KiB); too many channels", lds);
        const long long nblk = (long long)B * H * a.nWt * a.nDch;
        OSA_REQUIRE(nblk < (1ll << 31), "build_volume: grid too large");
        dim3 grid((unsigned)nblk), block(256);
#define OSA_VOL_LAUNCH(Q)                                                                           \
        do {                                                                                        \
            if (lds > 64 * 1024)                                                                    \
                (void)hipFuncSetAttribute((const void*)build_volume_ndhwc_kernel<Q>,                \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
            hipLaunchKernelGGL(build_volume_ndhwc_kernel<Q>, grid, block, lds, st, a);              \
        } while (0)
        switch (QG) {
            case 1: OSA_VOL_LAUNCH(1); break;
            case 2: OSA_VOL_LAUNCH(2); break;
            case 3: OSA_VOL_LAUNCH(3); break;
            default: OSA_VOL_LAUNCH(4); break;
        }
#undef OSA_VOL_LAUNCH
        OSA_LAUNCH_CHECK("build_volume_ndhwc");
        return 0;
    }
    OSA_REQUIRE(gwc_stride == 0 && cat_stride == 0, "build_volume: NHWC features need the NDHWC volume layout and K %% 4 == 0");
    VolNArgs a;
    a.lg = left_gwc; a.rg = right_gwc; a.lc = left_cat; a.rc = right_cat; a.vol = vol;
    a.B = B; a.C = C; a.Cc = Cc; a.H = H; a.W = W; a.D = maxdisp; a.G = G; a.K = K;
    a.VC = vol_channels; a.coff = c_off; a.mask_left = mask_left_concat;
    OSA_REQUIRE((long long)B * maxdisp <= 65535 && nch <= 65535, "build_volume: grid too large");
    dim3 grid(cdiv((long long)H * W, 256), nch, B * maxdisp), block(256);
    hipLaunchKernelGGL(build_volume_ncdhw_kernel, grid, block, 0, st, a);
    OSA_LAUNCH_CHECK("build_volume_ncdhw");
    return 0;
}

extern "C" int osa_corr_volume_f32(const float* left, const float* right, float* vol,
                                   int B, int C, int H, int W, int maxdisp, void* stream) {
    // correlation layer == one group over all channels, volume [B,1,D,H,W] == [B,D,H,W]
    return osa_build_volume_f32(left, right, C, 1, nullptr, nullptr, 0, vol, OSA_NCDHW, 1, 0,
                                B, H, W, maxdisp, 1, stream);
}
