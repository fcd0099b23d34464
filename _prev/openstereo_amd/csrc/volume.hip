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
#include <type_traits>

namespace osa {

// ------------------------------------------------------------------ NDHWC ----
struct VolArgs {
    const float* lg; const float* rg; const float* lc; const float* rc;
    float* vol;
    float* meta;           // range block of the volume (meta[0] = running max |value|) or NULL
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
    float am = 0.f;
    const unsigned am_seen = p.meta ? amax_peek(p.meta) : 0u;
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
                am = fmaxf(am, fabsf(v));
            }
        }
    }
    if (p.meta) publish_amax(p.meta, am, am_seen, smem);
}

// ---- NDHWC, quad lanes: every lane produces 4 consecutive output channels of one voxel ----
// Needs G % 4 == 0, Cc % 4 == 0 and (G + 2Cc) / 4 = NQ a power of two <= 64 (GwcNet: 16).  A wave
// covers 64 / NQ consecutive pixels, so one store instruction writes 64 x 16 B = 1 KB of contiguous
// NDHWC volume (4 voxels for GwcNet) instead of one 256-byte voxel.  The left features of a lane
// (its 4 groups x K channels) live in registers for the whole disparity chunk; only the sliding
// right window goes through LDS, permuted so that the NQ lanes of a voxel read consecutive 16-byte
// slots.  Same k-ordered fmaf chain and the same division by K as the per-channel kernel.
struct VolQArgs {
    VolArgs v;
    int NQ, lgNQ, DCH, RSq;
    int dbg;               // timing experiments only (OSA_VOL_DBG): 1 = no stores, 2 = no dot products, 4 = no window staging
    // d-walking form, split output (f16x3 chains): the volume is written as a split tensor -- every 16-channel chunk [16 x fp16 hi | 16 x fp16 lo],
    // the bytes of fp32 NDHWC -- scaled by a power of two derived from the FEATURES' range blocks (a bound, known before the first voxel
    // exists: |gwc| <= max|f|^2, |concat| <= max|f_cat|), so that the first aggregation layer stages it by LDS-DMA like every other layer of
    // the chain instead of splitting fp32 values through registers.  split = 0: fp32 output.
    int split;
    const float* gmeta; const float* cmeta;     // range blocks of the gwc / concat feature tensors (left and right images in one tensor)
};

// PX2: a lane owns TWO pixels, w and w + vpw.  out(w, d) and out(w + vpw, d + vpw) read the same right vector R[w - d], so walking the
// window once serves both: the kernel is bound by LDS reads (8 float4 per lane and (w, d): 983 KB per workgroup for 196 KB of output),
// and this cuts them by DCH / (DCH + vpw) * 2 = 1.7x at the price of a second set of left registers.  Stores stay 1 KB contiguous.
template <int QG, int NWV, bool PX2 = false>     // NWV waves per workgroup (4 or 8): a wider pixel tile amortises the right window
__global__ __launch_bounds__(NWV * 64) void build_volume_quads_kernel(const VolQArgs q) {
    extern __shared__ __attribute__((aligned(16))) float4 smq[];
    const VolArgs& p = q.v;
    constexpr int NTHR = NWV * 64;
    const int vpw = 64 >> q.lgNQ, WT = NWV * vpw * (PX2 ? 2 : 1);
    const int NPXR = WT + q.DCH - 1;

    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int dch = bid % p.nDch; bid /= p.nDch;
    const int wt = bid % p.nWt;   bid /= p.nWt;
    const int h = bid % p.H;
    const int b = bid / p.H;
    const int w0 = wt * WT, d0 = dch * q.DCH;
    const int wr0 = w0 - (d0 + q.DCH - 1);            // first right pixel of the window

    const int tid = threadIdx.x;
    const size_t plane = (size_t)p.H * p.W;
    const int G4 = p.G >> 2;
    const int nq_g = QG * p.G, nq_c = p.Cc >> 2, nq = nq_g + nq_c;
    const size_t rowpix = ((size_t)b * p.H + h) * p.W;

    // ---- this lane's voxel column and role; its left features are requested first so that their
    // latency overlaps the staging of the right window
    const int lane = tid & 63, wave = tid >> 6;
    const int cq = lane & (q.NQ - 1), wsub = lane >> q.lgNQ;
    const int w = w0 + wave * vpw * (PX2 ? 2 : 1) + wsub;          // PX2: the lane's first pixel; the second is w + vpw
    const bool wlive = w < p.W;
    const int role = (cq < G4) ? 0 : ((cq < G4 + nq_c) ? 1 : 2);   // gwc quad / left concat quad / right concat quad
    float4 Lr[4 * QG], Lr2[PX2 ? 4 * QG : 1];
    float4 lcat = make_float4(0.f, 0.f, 0.f, 0.f), lcat2 = lcat;
    auto load_left = [&](int wp, float4* L, float4& lc) {
#pragma unroll
        for (int i = 0; i < 4 * QG; ++i) L[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (wp < p.W && role == 0) {
#pragma unroll
            for (int gi = 0; gi < 4; ++gi)
#pragma unroll
                for (int kq = 0; kq < QG; ++kq) {
                    const int g = cq * 4 + gi;
                    if (p.gstride) L[gi * QG + kq] = *reinterpret_cast<const float4*>(p.lg + (rowpix + wp) * p.gstride + g * p.K + kq * 4);
                    else {
                        const float* src = p.lg + ((size_t)b * p.C + (size_t)g * p.K + kq * 4) * plane + (size_t)h * p.W + wp;
                        L[gi * QG + kq] = make_float4(src[0], src[plane], src[2 * plane], src[3 * plane]);
                    }
                }
        } else if (wp < p.W && role == 1) {
            const int qc = cq - G4;
            if (p.cstride) {
                const float* src = p.lc + (rowpix + wp) * p.cstride + qc * 4;
                lc = make_float4(src[0], src[1], src[2], src[3]);
            } else {
                const float* src = p.lc + ((size_t)b * p.Cc + qc * 4) * plane + (size_t)h * p.W + wp;
                lc = make_float4(src[0], src[plane], src[2 * plane], src[3 * plane]);
            }
        }
    };
    load_left(w, Lr, lcat);
    if constexpr (PX2) load_left(w + vpw, Lr2, lcat2);
    // ---- right window -> LDS.  Source quad (g, kq) goes to slot (g&3)*QG*G4 + kq*G4 + (g>>2).
    // Item = (pixel, quad); the fast index follows the feature layout (NHWC: quads of a pixel, NCHW:
    // pixels of a quad).  A thread walks its items with a carry instead of dividing, and keeps 4
    // loads in flight before the first LDS store.
    {
        const int items = (q.dbg & 4) ? 0 : nq * NPXR;
        const bool chan_fast = (p.gstride != 0);
        const int inner = chan_fast ? nq : NPXR;      // extent of the fast index
        const int step_hi = NTHR / inner, step_lo = NTHR - step_hi * inner;
        int hi = tid / inner, lo = tid - hi * inner;
        for (int it0 = tid; it0 < items; it0 += 4 * NTHR) {
            float4 v[4];
            int dst[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int qi = chan_fast ? lo : hi, px = chan_fast ? hi : lo;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                dst[u] = -1;
                if (it0 + u * NTHR < items) {
                    const int w = wr0 + px;
                    const bool wok = (w >= 0) && (w < p.W);
                    int pos;
                    if (qi < nq_g) {
                        const int g = qi / QG, kq = qi - g * QG;
                        pos = ((g & 3) * QG + kq) * G4 + (g >> 2);
                        if (wok) {
                            if (p.gstride) v[u] = *reinterpret_cast<const float4*>(p.rg + (rowpix + w) * p.gstride + g * p.K + kq * 4);
                            else {
                                const float* src = p.rg + ((size_t)b * p.C + (size_t)g * p.K + kq * 4) * plane + (size_t)h * p.W + w;
                                v[u] = make_float4(src[0], src[plane], src[2 * plane], src[3 * plane]);
                            }
                        }
                    } else {
                        const int qc = qi - nq_g;
                        pos = qi;
                        if (wok) {
                            if (p.cstride) {
                                const float* src = p.rc + (rowpix + w) * p.cstride + qc * 4;
                                v[u] = make_float4(src[0], src[1], src[2], src[3]);
                            } else {
                                const float* src = p.rc + ((size_t)b * p.Cc + qc * 4) * plane + (size_t)h * p.W + w;
                                v[u] = make_float4(src[0], src[plane], src[2 * plane], src[3 * plane]);
                            }
                        }
                    }
                    dst[u] = px * q.RSq + pos;
                }
                lo += step_lo; hi += step_hi;
                if (lo >= inner) { lo -= inner; ++hi; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (dst[u] >= 0) smq[dst[u]] = v[u];
        }
    }

    __syncthreads();

    const float Kf = (float)p.K;
    // mean over K channels: for K a power of two the multiply by 1/K is exact (== the division)
    const bool kpow2 = (p.K & (p.K - 1)) == 0;
    const float Kinv = 1.0f / Kf;
    const int rq = nq_g + (cq - G4 - nq_c);           // right-concat slot of this lane (role 2)
    float* vout = p.vol + p.coff + cq * 4;
    float am = 0.f;
    const unsigned am_seen = p.meta ? amax_peek(p.meta) : 0u;
    // one output quad of pixel wp at disparity d from the window row rrow (R[wp - d])
    auto emit = [&](int wp, int d, const float4* rrow, const float4* L, const float4& lc) {
        const bool valid = (wp >= d);
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (role == 0 && !(q.dbg & 2)) {
            float sv[4];
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                float s_ = 0.f;
#pragma unroll
                for (int kq = 0; kq < QG; ++kq) {
                    const float4 r = rrow[(gi * QG + kq) * G4 + cq];
                    const float4 l = L[gi * QG + kq];
                    s_ = fmaf(l.x, r.x, s_); s_ = fmaf(l.y, r.y, s_);
                    s_ = fmaf(l.z, r.z, s_); s_ = fmaf(l.w, r.w, s_);
                }
                sv[gi] = valid ? (kpow2 ? s_ * Kinv : s_ / Kf) : 0.f;
            }
            o = make_float4(sv[0], sv[1], sv[2], sv[3]);
        } else if (role == 1) {
            if (valid || !p.mask_left) o = lc;
        } else {
            if (valid) o = rrow[rq];
        }
        if (wp < p.W && (!(q.dbg & 1) || o.x == 12345.678f)) {
            const size_t vox = (((size_t)b * p.D + d) * p.H + h) * p.W + wp;
            store16(vout + vox * p.VC, o);
            am = fmaxf(am, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
        }
    };
    if constexpr (PX2) {
        // t walks the window rows both pixels need: row(t) = R[(w + vpw) - (d0 + t)] serves pixel w + vpw at dd = t and pixel w at dd = t - vpw
        const int c2 = wave * vpw * 2 + wsub + vpw;             // window column of the second pixel
#pragma unroll 4
        for (int t = 0; t < q.DCH + vpw; ++t) {
            const float4* rrow = smq + (size_t)(c2 + q.DCH - 1 - t) * q.RSq;
            if (t < q.DCH && d0 + t < p.D) emit(w + vpw, d0 + t, rrow, Lr2, lcat2);
            if (t >= vpw && d0 + t - vpw < p.D) emit(w, d0 + t - vpw, rrow, Lr, lcat);
        }
        if (p.meta) publish_amax(p.meta, am, am_seen, reinterpret_cast<float*>(smq));
        return;
    }
#pragma unroll 4
    for (int dd = 0; dd < q.DCH; ++dd) {
        const int d = d0 + dd;
        if (d >= p.D) break;
        const bool valid = (w >= d);
        const float4* rrow = smq + (size_t)(wave * vpw + wsub + q.DCH - 1 - dd) * q.RSq;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (role == 0 && !(q.dbg & 2)) {
            float sv[4];
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                float s = 0.f;
#pragma unroll
                for (int kq = 0; kq < QG; ++kq) {
                    const float4 r = rrow[(gi * QG + kq) * G4 + cq];
                    const float4 l = Lr[gi * QG + kq];
                    s = fmaf(l.x, r.x, s); s = fmaf(l.y, r.y, s);
                    s = fmaf(l.z, r.z, s); s = fmaf(l.w, r.w, s);
                }
                sv[gi] = valid ? (kpow2 ? s * Kinv : s / Kf) : 0.f;
            }
            o = make_float4(sv[0], sv[1], sv[2], sv[3]);
        } else if (role == 1) {
            if (valid || !p.mask_left) o = lcat;
        } else {
            if (valid) o = rrow[rq];
        }
        if (wlive && (!(q.dbg & 1) || o.x == 12345.678f)) {
            const size_t vox = (((size_t)b * p.D + d) * p.H + h) * p.W + w;
            store16(vout + vox * p.VC, o);
            am = fmaxf(am, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
        }
    }
    if (p.meta) publish_amax(p.meta, am, am_seen, reinterpret_cast<float*>(smq));
}

// ---- NDHWC, quad lanes, d-walking form (r4) ----
// The quad-lane kernel above is bound by latency at the occupancy its 70 KB right window allows (2 workgroups per CU): every workgroup
// loads its left features, stages the window, and only then starts to store -- three phases that two workgroups per CU do not overlap
// (ablation in DESIGN.md 3.1: stores 0.05 + staging 0.035 + products 0.02 + launch / left loads / loop 0.05 ms of 0.154).  Here a workgroup
// owns ONE (b, h, 32-pixel tile) and WALKS along d in steps of DS disparities:
//   * the left features of a lane stay in registers for all D disparities (loaded once instead of once per disparity chunk);
//   * the right window is a RING of NB = WT / DS + 2 blocks of DS pixels in LDS (pixel x lives in slot (x - (w0 + 1)) mod (NB * DS)): step
//     s needs the pixels [w0 + 1 - (s + 1) DS, w0 + WT - 1 - s DS] -- DS new pixels on the left per step, which a LOADER WAVE (wave NWV of the
//     workgroup, no other duty) fetches by LDS-DMA (global_load_lds_dwordx4, per-lane source = the permuted quad the slot holds) while the NWV
//     compute waves run the dot products and the stores of step s.  One barrier per step hands the block over (the loader waits for its
//     transfers first; the compute waves never wait on vmcnt, so their stores stay in flight across steps).  The block a transfer
//     overwrites was last read at least one step -- one barrier -- earlier (derivation next to the kernel).
//   * staged bytes per output byte fall from (WT + DCH - 1) / (WT DCH) to (WT + D - 1) / (WT D) pixels per voxel row (0.36 -> 0.26 of the
//     window per output for GwcNet), and there is no second pass over the left features.
// Same lanes, same k-ordered fmaf chains, same stores as the kernel above: bit-identical output (tests/test_gpu_parity.py).
// NHWC features with 16-byte aligned quads only (the engine's backbone output); everything else keeps the kernel above.
__device__ const float4 g_vol_zeros[64] = {};        // source of the ring slots outside the image / beyond a pixel's quads

// SPLIT: the volume is written as a split tensor (VolQArgs::split; a kernel of its own -- compiled into one kernel behind a run-time branch the
// fp32 loop lost 20 %: 70 -> 92 registers and two unrolled loop bodies, profiles/round4/volume_walk_after_split_support.txt)
template <int QG, int NWV, int DS, bool SPLIT = false>
__global__ __launch_bounds__((NWV + 1) * 64) void build_volume_walk_kernel(const VolQArgs q) {
    extern __shared__ __attribute__((aligned(16))) float4 smq[];
    const VolArgs& p = q.v;
    const int vpw = 64 >> q.lgNQ, WT = NWV * vpw;
    const int NB = WT / DS + 2, NRING = NB * DS;            // ring capacity in pixels (host: WT % DS == 0)
    const int BLKQ = DS * q.RSq;                            // float4 slots per block
    const int NI = (BLKQ + 63) / 64;                        // LDS-DMA instructions per block

    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int wt = bid % p.nWt;   bid /= p.nWt;
    const int h = bid % p.H;
    const int b = bid / p.H;
    const int w0 = wt * WT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G4 = p.G >> 2;
    const int nq_g = QG * p.G, nq_c = p.Cc >> 2, nq = nq_g + nq_c;
    const size_t rowpix = ((size_t)b * p.H + h) * p.W;
    const int nsteps = (p.D + DS - 1) / DS;
    float* const red = reinterpret_cast<float*>(smq + (size_t)NRING * q.RSq);      // publish_amax scratch (NWV + 1 floats) above the ring

    // ---- LDS-DMA of ring block `blk` <- right pixels [x0, x0 + DS).  Slot j of the block = (pixel j / RSq, position j % RSq); the
    // position -> source quad map is the inverse of the staging permutation of the kernel above.  Issued by `nw` waves, wave `iw` of them.
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smq;
    auto dma_block = [&](const int blk, const int x0, const int iw, const int nw) {
        for (int i = iw; i < NI; i += nw) {
            const int j = i * 64 + lane;
            const int px = j / q.RSq, pos = j - px * q.RSq;
            const int x = x0 + px;
            const char* src = reinterpret_cast<const char*>(g_vol_zeros) + lane * 16;
            if (j < BLKQ && pos < nq && x >= 0 && x < p.W) {
                if (pos < nq_g) {
                    const int a = pos / G4, ghi = pos - a * G4;
                    const int g = ghi * 4 + a / QG, kq = a - (a / QG) * QG;
                    src = reinterpret_cast<const char*>(p.rg + (rowpix + x) * p.gstride + g * p.K + kq * 4);
                } else src = reinterpret_cast<const char*>(p.rc + (rowpix + x) * p.cstride + (pos - nq_g) * 4);
            }
            const unsigned m0v = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)((blk * BLKQ + i * 64) * 16));
            if (j < BLKQ) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(m0v) : "memory");
            }
        }
    };
    // block index of the pixels [w0 + 1 + k DS, w0 + (k + 1) DS], k may be negative
    auto blk_of = [&](int k) { k %= NB; return k < 0 ? k + NB : k; };

    // ---- prologue: every wave requests its left features (compute waves), then all NWV + 1 waves share the initial window:
    // blocks 0 .. WT / DS - 1 (pixels w0 + 1 .. w0 + WT) and block -1 (step 0's left pixels w0 + 1 - DS .. w0)
    const int cq = lane & (q.NQ - 1), wsub = lane >> q.lgNQ;
    const bool compute = wave < NWV;
    const int w = w0 + wave * vpw + wsub;
    const bool wlive = compute && w < p.W;
    const int role = (cq < G4) ? 0 : ((cq < G4 + nq_c) ? 1 : 2);
    float4 Lr[4 * QG];
    float4 lcat = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4 * QG; ++i) Lr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wlive && role == 0) {
#pragma unroll
        for (int gi = 0; gi < 4; ++gi)
#pragma unroll
            for (int kq = 0; kq < QG; ++kq)
                Lr[gi * QG + kq] = *reinterpret_cast<const float4*>(p.lg + (rowpix + w) * p.gstride + (cq * 4 + gi) * p.K + kq * 4);
    } else if (wlive && role == 1) {
        const float* src = p.lc + (rowpix + w) * p.cstride + (cq - G4) * 4;
        lcat = make_float4(src[0], src[1], src[2], src[3]);
    }
    {
        const int nbi = WT / DS + 1;                      // blocks of the initial window
        for (int t = wave; t < nbi * NI; t += NWV + 1) {  // (block, instruction) pairs round-robin over the waves
            const int kb = t / NI, i = t - kb * NI;
            const int k = (kb < WT / DS) ? kb : -1;
            // one instruction: reuse dma_block's body through a 1-wave slice
            const int j = i * 64 + lane;
            const int px = j / q.RSq, pos = j - px * q.RSq;
            const int x = w0 + 1 + k * DS + px;
            const char* src = reinterpret_cast<const char*>(g_vol_zeros) + lane * 16;
            if (j < BLKQ && pos < nq && x >= 0 && x < p.W) {
                if (pos < nq_g) {
                    const int a = pos / G4, ghi = pos - a * G4;
                    const int g = ghi * 4 + a / QG, kq = a - (a / QG) * QG;
                    src = reinterpret_cast<const char*>(p.rg + (rowpix + x) * p.gstride + g * p.K + kq * 4);
                } else src = reinterpret_cast<const char*>(p.rc + (rowpix + x) * p.cstride + (pos - nq_g) * 4);
            }
            const unsigned m0v = __builtin_amdgcn_readfirstlane(smem_lds + (unsigned)((blk_of(k) * BLKQ + i * 64) * 16));
            if (j < BLKQ) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(m0v) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (also the left features: requested above, needed below)
    __syncthreads();

    float am = 0.f;
    const unsigned am_seen = (p.meta && compute) ? amax_peek(p.meta) : 0u;
    if (!compute) {
        // ================= loader wave: block of step s + 1 while the compute waves run step s =================
        for (int s = 0; s < nsteps; ++s) {
            if (s + 1 < nsteps) dma_block(blk_of(-(s + 2)), w0 + 1 - (s + 2) * DS, 0, 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else {
        // ================= compute waves =================
        const float Kf = (float)p.K;
        const bool kpow2 = (p.K & (p.K - 1)) == 0;
        const float Kinv = 1.0f / Kf;
        const int rq = nq_g + (cq - G4 - nq_c);
        float* vout = p.vol + p.coff + cq * 4;
        // split output: scale from the features' ranges (wave-uniform); lanes (cq even, cq + 1) of a voxel pair up -- the even lane stores the
        // 16 bytes of hi halves of both quads, the odd lane the 16 bytes of lo halves (one 16-byte store per lane, as for fp32)
        [[maybe_unused]] float s_out = 1.f;
        if constexpr (SPLIT) {
            const float ag = (p.G > 0 && q.gmeta) ? amax_read(q.gmeta) : 0.f, ac = (p.Cc > 0 && q.cmeta) ? amax_read(q.cmeta) : 0.f;
            s_out = pow2_scale(fmaxf(ag * ag, ac) * 1.0625f);
            if (p.meta && blockIdx.x == 0 && tid == 0) p.meta[1] = s_out;
        }
        const int cch = p.coff + cq * 4;                       // first channel of this lane's quad
        [[maybe_unused]] float* const vsplit = p.vol + (cch >> 4) * 16 + ((cq & 1) ? 8 : 0) + (((cch & 15) >> 3) * 4);
        int ri = w - (w0 + 1);                                // ring slot of pixel w - d, d = 0 (in [-1, WT - 2])
        if (ri < 0) ri += NRING;
        // (measured and NOT kept: walking the ring offset and the output pointer incrementally instead of recomputing them per disparity --
        // fewer VALU instructions, but a loop-carried chain through the unrolled body: 0.97 -> 1.17 ms at 8 pairs, profiles/round4/volume_walk_incremental_addresses.txt)
        // one disparity of this lane's voxel column
        auto emit = [&](const int d) {
            {
                const bool valid = (w >= d);
                const float4* rrow = smq + (size_t)ri * q.RSq;
                ri = (ri == 0) ? NRING - 1 : ri - 1;
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (role == 0) {
                    float sv[4];
#pragma unroll
                    for (int gi = 0; gi < 4; ++gi) {
                        float sacc = 0.f;
#pragma unroll
                        for (int kq = 0; kq < QG; ++kq) {
                            const float4 r = rrow[(gi * QG + kq) * G4 + cq];
                            const float4 l = Lr[gi * QG + kq];
                            sacc = fmaf(l.x, r.x, sacc); sacc = fmaf(l.y, r.y, sacc);
                            sacc = fmaf(l.z, r.z, sacc); sacc = fmaf(l.w, r.w, sacc);
                        }
                        sv[gi] = valid ? (kpow2 ? sacc * Kinv : sacc / Kf) : 0.f;
                    }
                    o = make_float4(sv[0], sv[1], sv[2], sv[3]);
                } else if (role == 1) {
                    if (valid || !p.mask_left) o = lcat;
                } else {
                    if (valid) o = rrow[rq];
                }
                if constexpr (SPLIT) {
                    uint2 h2, l2;
                    split_f16(mul4(o, s_out), h2, l2);
                    const bool odd = (cq & 1) != 0;
                    const uint2 send = odd ? h2 : l2;            // what the partner lane stores
                    const uint2 recv = make_uint2((unsigned)__builtin_amdgcn_mov_dpp((int)send.x, 0xB1, 0xf, 0xf, true),     // quad_perm [1, 0, 3, 2]
                                                  (unsigned)__builtin_amdgcn_mov_dpp((int)send.y, 0xB1, 0xf, 0xf, true));
                    if (wlive) {
                        const size_t vox = (((size_t)b * p.D + d) * p.H + h) * p.W + w;
                        store16(vsplit + vox * p.VC, odd ? make_uint4(recv.x, recv.y, l2.x, l2.y) : make_uint4(h2.x, h2.y, recv.x, recv.y));
                        am = fmaxf(am, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
                    }
                } else if (wlive) {
                    const size_t vox = (((size_t)b * p.D + d) * p.H + h) * p.W + w;
                    store16(vout + vox * p.VC, o);
                    am = fmaxf(am, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
                }
            }
        };
        for (int s = 0; s < nsteps; ++s) {
            if constexpr (SPLIT) {
                // (the cross-lane exchange of the split stores is a convergent operation: the unroller will not duplicate it past an early exit,
                // so full steps run a fixed-trip loop and only the last, partial step a counted one)
                const int nd = (p.D - s * DS < DS) ? p.D - s * DS : DS;
                if (nd == DS) {
#pragma unroll 4
                    for (int dd = 0; dd < DS; ++dd) emit(s * DS + dd);
                } else {
                    for (int dd = 0; dd < nd; ++dd) emit(s * DS + dd);
                }
            } else {
#pragma unroll 4
                for (int dd = 0; dd < DS; ++dd) {
                    const int d = s * DS + dd;
                    if (d >= p.D) break;
                    emit(d);
                }
            }
            // step s is read; block s + 1 has landed (the loader waited for it).  s_barrier only: no vmcnt wait, the stores stay in flight
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    if (p.meta) publish_amax(p.meta, am, am_seen, red);
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

// ------------------------------------------------------------------ dormant variants (cost_volume.py:9-29, 44-56, 108-117) ----
// No shipped config enables them (StereoBase USE_SUB_VOLUME / CoEx-style heads); plain one-thread-per-element kernels, NCDHW like the
// reference, w fastest.  mode 0: CoEx correlation (sum over the channels of a group, D = maxdisp + 1 planes);
// mode 1 / 2: difference volume of compute_volume, side left / right; mode 3: L1 "sub" volume.
struct PairArgs {
    const float* l; const float* r; float* out;
    int B, C, G, H, W, D, mode;
};
__global__ __launch_bounds__(256) void pair_volume_kernel(const PairArgs p) {
    const int oc = blockIdx.y;                              // output channel (group / channel / 0)
    const int bd = blockIdx.z;
    const int b = bd / p.D, d = bd - b * p.D;
    const int hw = blockIdx.x * 256 + threadIdx.x;
    if (hw >= p.H * p.W) return;
    const int w = hw % p.W;
    const size_t plane = (size_t)p.H * p.W;
    const float* lb = p.l + (size_t)b * p.C * plane + hw;
    const float* rb = p.r + (size_t)b * p.C * plane + hw;
    float v = 0.f;
    int OC = 1;
    if (p.mode == 0) {                                      // cost[b,g,d,h,w] = sum_k x[g,k,h,w] * y[g,k,h,w-d]
        OC = p.G;
        const int K = p.C / p.G;
        if (w >= d)
            for (int k = 0; k < K; ++k) v += lb[(size_t)(oc * K + k) * plane] * rb[(size_t)(oc * K + k) * plane - d];   // torch's sum(2): ascending k
    } else if (p.mode == 1) {                               // reference[w] - target[w-d]   (w >= d)
        OC = p.C;
        if (w >= d) v = lb[(size_t)oc * plane] - rb[(size_t)oc * plane - d];
    } else if (p.mode == 2) {                               // target[w+d] - reference[w]   (w < W-d); d = 0: reference - target
        OC = p.C;
        if (d == 0) v = lb[(size_t)oc * plane] - rb[(size_t)oc * plane];
        else if (w < p.W - d) v = rb[(size_t)oc * plane + d] - lb[(size_t)oc * plane];
    } else {                                                // w < d: sum_c |l|;  else: sum_c |l[w] - r[w-d]|
        for (int c = 0; c < p.C; ++c) {
            const float a = lb[(size_t)c * plane];
            v += (w >= d) ? fabsf(a - rb[(size_t)c * plane - d]) : fabsf(a);
        }
    }
    p.out[(((size_t)b * OC + oc) * p.D + d) * plane + hw] = v;
}

// cat_fms with arbitrary (also negative / dilated) disparity samples, psmnet_cost_processor.py:30-47: plane idx holds disparity
// i = disp_index[idx];  i >= 0: columns w >= i get (reference[w], target[w - i]);  i < 0: columns w < W + i get (reference[w], target[w - i]);
// everything else stays zero.
__global__ __launch_bounds__(256) void cat_fms_kernel(const float* __restrict__ ref, const float* __restrict__ tgt, float* __restrict__ out,
                                                      const int* __restrict__ disp_index, int C, int H, int W, int n) {
    const int c2 = blockIdx.y;
    const int b = blockIdx.z / n, idx = blockIdx.z - b * n;
    const int hw = blockIdx.x * 256 + threadIdx.x;
    if (hw >= H * W) return;
    const int w = hw % W, i = disp_index[idx];
    const size_t plane = (size_t)H * W;
    const bool ok = (i >= 0) ? (w >= i) : (w < W + i);
    float v = 0.f;
    if (ok) v = (c2 < C) ? ref[((size_t)b * C + c2) * plane + hw] : tgt[((size_t)b * C + (c2 - C)) * plane + hw - i];
    out[(((size_t)b * 2 * C + c2) * n + idx) * plane + hw] = v;
}

}  // namespace osa

using namespace osa;

extern "C" int osa_cat_fms_f32(const float* reference_fm, const float* target_fm, float* out, const int* disp_index,
                               int B, int C, int H, int W, int n_samples, void* stream) {
    OSA_REQUIRE(reference_fm && target_fm && out && disp_index, "cat_fms: NULL pointer");
    OSA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && n_samples > 0, "cat_fms: bad dims");
    OSA_REQUIRE((long long)B * n_samples <= 65535 && 2 * C <= 65535, "cat_fms: grid too large");
    hipLaunchKernelGGL(cat_fms_kernel, dim3(cdiv((long long)H * W, 256), 2 * C, B * n_samples), dim3(256), 0, (hipStream_t)stream,
                       reference_fm, target_fm, out, disp_index, C, H, W, n_samples);
    OSA_LAUNCH_CHECK("cat_fms");
    return 0;
}

extern "C" int osa_pair_volume_f32(const float* left, const float* right, float* out,
                                   int B, int C, int groups, int H, int W, int planes, int mode, void* stream) {
    OSA_REQUIRE(left && right && out, "pair_volume: NULL pointer");
    OSA_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && planes > 0, "pair_volume: bad dims");
    OSA_REQUIRE(mode >= 0 && mode <= 3, "pair_volume: mode %d (0 coex correlation, 1 / 2 difference left / right, 3 L1 sub volume)", mode);
    if (mode == 0) OSA_REQUIRE(groups > 0 && C % groups == 0, "pair_volume: C=%d not divisible by groups=%d", C, groups);
    PairArgs a{left, right, out, B, C, mode == 0 ? groups : 1, H, W, planes, mode};
    const int oc = (mode == 0) ? groups : ((mode == 3) ? 1 : C);
    OSA_REQUIRE((long long)B * planes <= 65535 && oc <= 65535, "pair_volume: grid too large");
    hipLaunchKernelGGL(pair_volume_kernel, dim3(cdiv((long long)H * W, 256), oc, B * planes), dim3(256), 0, (hipStream_t)stream, a);
    OSA_LAUNCH_CHECK("pair_volume");
    return 0;
}

static int build_volume_impl(const float* left_gwc, const float* right_gwc, int C, int num_groups, int gwc_stride,
                             const float* left_cat, const float* right_cat, int Cc, int cat_stride,
                             float* vol, int layout, int vol_channels, int c_off,
                             int B, int H, int W, int maxdisp, int mask_left_concat, void* stream, float* vol_meta = nullptr,
                             int split = 0, const float* gwc_meta = nullptr, const float* cat_meta = nullptr);

// d-walking form of the NDHWC builder (build_volume_walk_kernel): disparities per step (4 or 8), 0 = the chunked kernel
static int g_vol_walk_ds = 8;
// ... of the split-output kernel: 4 (72 registers and a 53 KB ring -> 3 workgroups per CU; 84 / 64 KB / 2 at 8).  Measured at 3 pairs per launch,
// the sub-batch of the default line: 0.466-0.506 -> 0.418-0.424 ms, whole model +0.7 % (profiles/round4/volume_walk_step_split.txt); the fp32
// kernel is indifferent at 3 pairs and 2-3 % better with 8 at 8 pairs.
static int g_vol_walk_split_ds = 4;
static bool walk_eligible(int G, int K, int Cc, int gwc_stride, int cat_stride, const float* left_cat, const float* right_cat,
                          const float* vol, int vol_channels, int c_off, int W, int maxdisp, int ds) {
    const int nch = G + 2 * Cc, nq4 = nch / 4;
    if (!(G == 0 || (K % 4 == 0 && K / 4 <= 4))) return false;
    if (!((G % 4 == 0) && (Cc % 4 == 0) && nq4 >= 1 && nq4 <= 64 && (nq4 & (nq4 - 1)) == 0 && (vol_channels % 4 == 0) && (c_off % 4 == 0) &&
          (((size_t)vol & 15) == 0))) return false;
    const int WT = 8 * (64 / nq4);
    if (W < 2 * WT) return false;                                   // 8-wave pixel tiles
    if (!(G == 0 || gwc_stride > 0)) return false;
    if (!(Cc == 0 || (cat_stride > 0 && cat_stride % 4 == 0 && ((size_t)left_cat & 15) == 0 && ((size_t)right_cat & 15) == 0))) return false;
    return ds > 0 && WT % ds == 0 && maxdisp > ds;
}
static long long g_vol_walk_launches = 0;
extern "C" int osa_volume_walk_step(int ds) {
    const int prev = g_vol_walk_ds;
    if (ds == 0 || ds == 4 || ds == 8) { g_vol_walk_ds = ds; g_vol_walk_split_ds = ds ? ds : 4; }     // (0: the fp32 output falls back to the chunked kernel; the split form exists as a walk only)
    return prev;
}
extern "C" long long osa_volume_walk_launches(void) { return g_vol_walk_launches; }

extern "C" int osa_build_volume_f32(const float* left_gwc, const float* right_gwc, int C, int num_groups,
                                    const float* left_cat, const float* right_cat, int Cc,
                                    float* vol, int layout, int vol_channels, int c_off,
                                    int B, int H, int W, int maxdisp, int mask_left_concat,
                                    float* vol_meta, void* stream) {
    return build_volume_impl(left_gwc, right_gwc, C, num_groups, 0, left_cat, right_cat, Cc, 0, vol, layout,
                             vol_channels, c_off, B, H, W, maxdisp, mask_left_concat, stream, vol_meta);
}

extern "C" int osa_build_volume_nhwc_f32(const float* left_gwc, const float* right_gwc, int C, int num_groups, int gwc_stride,
                                         const float* left_cat, const float* right_cat, int Cc, int cat_stride,
                                         float* vol, int vol_channels, int c_off,
                                         int B, int H, int W, int maxdisp, int mask_left_concat, float* vol_meta, void* stream) {
    OSA_REQUIRE((C == 0 || (gwc_stride >= C && gwc_stride % 4 == 0 && ((size_t)left_gwc & 15) == 0 && ((size_t)right_gwc & 15) == 0)),
                "build_volume_nhwc: gwc features need stride >= C, stride %% 4 == 0 and 16-byte alignment");
    OSA_REQUIRE((Cc == 0 || cat_stride >= Cc), "build_volume_nhwc: concat stride %d < Cc %d", cat_stride, Cc);
    OSA_REQUIRE(C == 0 || (C / (num_groups > 0 ? num_groups : 1)) % 4 == 0, "build_volume_nhwc: channels per group must be a multiple of 4");
    return build_volume_impl(left_gwc, right_gwc, C, num_groups, gwc_stride ? gwc_stride : C, left_cat, right_cat, Cc,
                             cat_stride ? cat_stride : Cc, vol, OSA_NDHWC, vol_channels, c_off, B, H, W, maxdisp,
                             mask_left_concat, stream, vol_meta);
}

extern "C" int osa_build_volume_nhwc_split_eligible(const float* left_cat, const float* right_cat, const float* vol, int C, int num_groups,
                                                    int gwc_stride, int Cc, int cat_stride, int vol_channels, int c_off, int W, int maxdisp) {
    const int G = (C > 0) ? num_groups : 0;
    if (C > 0 && (num_groups <= 0 || C % num_groups)) return 0;
    const int K = G ? C / G : 0;
    if (vol_channels % 16 || c_off % 16 || (G + 2 * Cc) % 16) return 0;
    return walk_eligible(G, K, Cc, gwc_stride ? gwc_stride : C, cat_stride ? cat_stride : Cc, left_cat, right_cat, vol, vol_channels, c_off, W, maxdisp,
                         g_vol_walk_split_ds) ? 1 : 0;
}

extern "C" int osa_build_volume_nhwc_split_f16x3(const float* left_gwc, const float* right_gwc, int C, int num_groups, int gwc_stride,
                                                 const float* left_cat, const float* right_cat, int Cc, int cat_stride,
                                                 float* vol, int vol_channels, int c_off,
                                                 int B, int H, int W, int maxdisp, int mask_left_concat,
                                                 const float* gwc_meta, const float* cat_meta, float* vol_meta, void* stream) {
    OSA_REQUIRE(vol_meta != nullptr && (C == 0 || gwc_meta != nullptr) && (Cc == 0 || cat_meta != nullptr),
                "build_volume_nhwc_split: the range blocks of the features and of the volume are required (the scale of the split halves is derived from them)");
    OSA_REQUIRE((C == 0 || (gwc_stride >= C && gwc_stride % 4 == 0 && ((size_t)left_gwc & 15) == 0 && ((size_t)right_gwc & 15) == 0)),
                "build_volume_nhwc_split: gwc features need stride >= C, stride %% 4 == 0 and 16-byte alignment");
    OSA_REQUIRE((Cc == 0 || cat_stride >= Cc), "build_volume_nhwc_split: concat stride %d < Cc %d", cat_stride, Cc);
    OSA_REQUIRE(C == 0 || (C / (num_groups > 0 ? num_groups : 1)) % 4 == 0, "build_volume_nhwc_split: channels per group must be a multiple of 4");
    return build_volume_impl(left_gwc, right_gwc, C, num_groups, gwc_stride ? gwc_stride : C, left_cat, right_cat, Cc,
                             cat_stride ? cat_stride : Cc, vol, OSA_NDHWC, vol_channels, c_off, B, H, W, maxdisp,
                             mask_left_concat, stream, vol_meta, 1, gwc_meta, cat_meta);
}

static int build_volume_impl(const float* left_gwc, const float* right_gwc, int C, int num_groups, int gwc_stride,
                             const float* left_cat, const float* right_cat, int Cc, int cat_stride,
                             float* vol, int layout, int vol_channels, int c_off,
                             int B, int H, int W, int maxdisp, int mask_left_concat, void* stream, float* vol_meta,
                             int split, const float* gwc_meta, const float* cat_meta) {
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
        const int nq4 = nch / 4;
        const bool quads = (G % 4 == 0) && (Cc % 4 == 0) && nq4 >= 1 && nq4 <= 64 && (nq4 & (nq4 - 1)) == 0 &&
                           (vol_channels % 4 == 0) && (c_off % 4 == 0) && (((size_t)vol & 15) == 0) &&
                           !exp_set("OSA_VOL_PERCHANNEL");
        if (quads) {
            VolQArgs qa;
            VolArgs& a = qa.v;
            a.lg = left_gwc; a.rg = right_gwc; a.lc = left_cat; a.rc = right_cat; a.vol = vol; a.meta = vol_meta;
            a.B = B; a.C = C; a.Cc = Cc; a.H = H; a.W = W; a.D = maxdisp; a.G = G; a.K = K;
            a.VC = vol_channels; a.coff = c_off; a.mask_left = mask_left_concat;
            a.gstride = gwc_stride; a.cstride = cat_stride;
            a.RS = 0; a.catbase = 0;
            const int QG = (G > 0) ? K / 4 : 1;
            qa.NQ = nq4; qa.lgNQ = 0;
            while ((1 << qa.lgNQ) < nq4) ++qa.lgNQ;
            // 8 waves per workgroup when the map is wide enough (amortises the right window over 2x the pixels)
            int nwv = (W >= 8 * (64 / nq4) * 2) ? 8 : 4;
            { const int e = exp_int("OSA_VOL_WAVES", 0); if (e == 4 || e == 8) nwv = e; }
            // two pixels per lane (PX2): the same pixel tile from half the waves
            const bool px2 = exp_int("OSA_VOL_PX2", 0) != 0 && nwv == 8;
            if (px2) nwv = 4;
            const int WT = nwv * (64 / nq4) * (px2 ? 2 : 1);
            qa.RSq = ((G > 0) ? QG * G : 0) + Cc / 4;
            if (qa.RSq % 16 > 6) qa.RSq += 16 - qa.RSq % 16;     // keeps the lanes of two neighbouring voxels on distinct 16-byte slots
            // disparity chunk: D split evenly into the fewest chunks whose right window fits ~52 KiB of
            // LDS (3 workgroups per CU); very wide feature vectors may use up to the whole 160 KiB
            size_t budget = (nwv == 8 || px2) ? 78 * 1024 : 52 * 1024;   // 2 x 8 waves (or 2 x 4 two-pixel waves) or 3 x 4 waves per CU
            { const int e = exp_int("OSA_VOL_LDS", 0); if (e > 0) budget = (size_t)e; }
            int nchunk = 1;
            while (nchunk < maxdisp && (size_t)(WT + cdiv(maxdisp, nchunk) - 1) * qa.RSq * 16 > budget) ++nchunk;
            const int dch = cdiv(maxdisp, nchunk);
            qa.DCH = dch;
            qa.dbg = exp_int("OSA_VOL_DBG", 0);
            a.nWt = cdiv(W, WT); a.nDch = cdiv(maxdisp, dch);
            const size_t lds = (size_t)(WT + dch - 1) * qa.RSq * 16;
            OSA_REQUIRE(lds <= 160 * 1024, "build_volume: %zu B of LDS needed (> 160 KiB); too many channels", lds);
            const long long nblk = (long long)B * H * a.nWt * a.nDch;
            OSA_REQUIRE(nblk < (1ll << 31), "build_volume: grid too large");
            dim3 grid((unsigned)nblk), block(nwv * 64);
#define OSA_VOLQ_LAUNCH1(Q, NWV)                                                                    \
            do {                                                                                    \
                if (lds > 64 * 1024)                                                                \
                    (void)hipFuncSetAttribute((const void*)build_volume_quads_kernel<Q, NWV>,       \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);\
                hipLaunchKernelGGL((build_volume_quads_kernel<Q, NWV>), grid, block, lds, st, qa);  \
            } while (0)
#define OSA_VOLQ_LAUNCH1P(Q, NWV)                                                                   \
            do {                                                                                    \
                if (lds > 64 * 1024)                                                                \
                    (void)hipFuncSetAttribute((const void*)build_volume_quads_kernel<Q, NWV, true>, \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);\
                hipLaunchKernelGGL((build_volume_quads_kernel<Q, NWV, true>), grid, block, lds, st, qa); \
            } while (0)
#define OSA_VOLQ_LAUNCH(Q) do { if (px2) { if (nwv == 8) OSA_VOLQ_LAUNCH1P(Q, 8); else OSA_VOLQ_LAUNCH1P(Q, 4); }    \
                                else if (nwv == 8) OSA_VOLQ_LAUNCH1(Q, 8); else OSA_VOLQ_LAUNCH1(Q, 4); } while (0)
            // d-walking form (build_volume_walk_kernel): NHWC features with 16-byte aligned quads, 8-wave pixel tiles
            const int walk_ds = split ? g_vol_walk_split_ds : g_vol_walk_ds;
            const bool walk = !px2 && nwv == 8 && walk_eligible(G, K, Cc, gwc_stride, cat_stride, left_cat, right_cat, vol, vol_channels, c_off, W, maxdisp, walk_ds);
            OSA_REQUIRE(!split || (walk && vol_channels % 16 == 0 && c_off % 16 == 0 && nch % 16 == 0),
                        "build_volume: split output needs the d-walking form and 16-channel aligned volume channels (osa_build_volume_nhwc_split_eligible)");
            qa.split = split; qa.gmeta = gwc_meta; qa.cmeta = cat_meta;
            if (walk) {
                const int DS = walk_ds;
                a.nWt = cdiv(W, WT); a.nDch = 1;
                qa.DCH = DS; qa.dbg = 0;
                const size_t wlds = (size_t)(WT / DS + 2) * DS * qa.RSq * 16 + 64;
                const long long wblk = (long long)B * H * a.nWt;
                OSA_REQUIRE(wlds <= 160 * 1024 && wblk < (1ll << 31), "build_volume: walk form does not fit (%zu B of LDS)", wlds);
#define OSA_VOLW_LAUNCH1(Q, DSV, SP)                                                                \
                do {                                                                                \
                    if (wlds > 64 * 1024)                                                           \
                        (void)hipFuncSetAttribute((const void*)build_volume_walk_kernel<Q, 8, DSV, SP>, \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds); \
                    hipLaunchKernelGGL((build_volume_walk_kernel<Q, 8, DSV, SP>), dim3((unsigned)wblk), dim3(9 * 64), wlds, st, qa); \
                } while (0)
#define OSA_VOLW_LAUNCH(Q) do { if (split) { if (DS == 4) OSA_VOLW_LAUNCH1(Q, 4, true); else OSA_VOLW_LAUNCH1(Q, 8, true); }   \
                                else { if (DS == 4) OSA_VOLW_LAUNCH1(Q, 4, false); else OSA_VOLW_LAUNCH1(Q, 8, false); } } while (0)
                switch (QG) {
                    case 1: OSA_VOLW_LAUNCH(1); break;
                    case 2: OSA_VOLW_LAUNCH(2); break;
                    case 3: OSA_VOLW_LAUNCH(3); break;
                    default: OSA_VOLW_LAUNCH(4); break;
                }
#undef OSA_VOLW_LAUNCH
#undef OSA_VOLW_LAUNCH1
                OSA_LAUNCH_CHECK("build_volume_walk");
                ++g_vol_walk_launches;
                return 0;
            }
            switch (QG) {
                case 1: OSA_VOLQ_LAUNCH(1); break;
                case 2: OSA_VOLQ_LAUNCH(2); break;
                case 3: OSA_VOLQ_LAUNCH(3); break;
                default: OSA_VOLQ_LAUNCH(4); break;
            }
#undef OSA_VOLQ_LAUNCH1
#undef OSA_VOLQ_LAUNCH1P
#undef OSA_VOLQ_LAUNCH
            OSA_LAUNCH_CHECK("build_volume_quads");
            return 0;
        }
        OSA_REQUIRE(!split, "build_volume: split output needs the quad-lane d-walking form (osa_build_volume_nhwc_split_eligible)");
        VolArgs a;
        a.lg = left_gwc; a.rg = right_gwc; a.lc = left_cat; a.rc = right_cat; a.vol = vol; a.meta = vol_meta;
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
                                B, H, W, maxdisp, 1, nullptr, stream);
}
