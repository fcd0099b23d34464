// d-marching form of the 3x3x3 stride-1 "same" convolutions with 32 output channels (f16x3 mode): the V0-resolution layers of the 3-D
// aggregation networks -- GwcNet dres0 / dres1 / classif*.0 (gwcnet_disp_processor.py:40-81), PSMNet dres0 / dres1 -- which carry 56 % of
// GwcNet's 3-D MACs.
//
// The brick kernel (conv_kernel.h) stages a 6x10x10 halo brick per 4x8x8 output brick -- 2.34x the input in staged bytes, 2.0x measured at
// the HBM side -- and every wave pulls its B (weight) fragments through the vector-memory path.  Here a workgroup owns a TH x TW pixel
// column and WALKS along d (the classifier's trick, conv3d.hip classifier_march_kernel):
//   * every input plane (TH+2) x (TW+2) x 16 channels is staged ONCE per pass and serves the three output planes d-1, d, d+1 whose
//     kd = 2, 1, 0 taps it is -- three accumulator sets per wave (96 accumulator registers, 2 waves per SIMD); staged bytes fall to
//     (TH+2)(TW+2) / (TH TW) x (dseg + 2) / dseg = 1.3-1.5x of one pass, LDS reads of A fragments per MFMA fall 3x;
//   * B operands come through an LDS ring filled by LDS-DMA, ONE fetch per workgroup and step instead of one per wave: timing-only
//     ablations of v1 (operands per wave from L2; profiles/round4/march_v1_ablation_and_f16_tests.txt) showed the tap loop bound by them
//     -- taps 2.02 ms, without the B loads 1.36 ms, loads alone 1.15 ms: the vector-memory path delivers ~37 B/clk/CU of 16-byte-per-lane
//     loads, L1 hit or not; v3 (ring) brought the taps to 1.66 ms (march_v3_ring4_ablation_ab.txt);
//   * (v4) split inputs are staged ASYNCHRONOUSLY: while the taps of pass q = (plane, 16-channel chunk) run from one plane buffer, the
//     plane-chunk of pass q + 1 lands in the other by LDS-DMA, one 1 KB piece per wave and step.  v3's ablations had staging (0.48 ms)
//     and epilogue (0.28 ms) exactly additive to the taps (1.66 ms): with two workgroups per CU nothing hides a workgroup's 6-8 us of
//     staging latency per 7 us of MFMA work.
// The packed weight stream is the brick kernel's ([chunk][tap][hi|lo][k-group][Cout][8 x fp16], tap = kd*9 + kh*3 + kw): a (chunk, kh, kw)
// step reads its three kd taps 9 tap-steps apart.  Same split arithmetic (Ahi.Blo + Alo.Bhi + Ahi.Bhi, fp32 accumulate), same operand
// ranges and the same epilogue semantics as conv_mfma_kernel; the summation ORDER differs (kd outermost), so results agree to fp32
// rounding, not bitwise.  The exact-f32 mode keeps the brick kernel (its goldens stay bit-for-bit).
#pragma once
#include "conv_kernel.h"

namespace osa {

// NWV = 4 waves per workgroup, every wave owns MT = 2 M-tiles of 32 voxels.  TW = 32: an M-tile is one row of 32 pixels; TW = 16: two rows
// of 16.  LDS image of a chunk-plane: voxels 5 slots (80 B) apart -- the 16 lanes of a ds_read_b128 group fall on 16 distinct 16-byte
// slots (mod 256 B) -- and, for TW = 16, rows a multiple of 16 slots apart (the group straddles two rows).
template <int NWV, int TW>
struct MarchGeo {
    static constexpr int MT = 2;
    static constexpr int RPT = 32 / TW;                 // rows per M-tile
    static constexpr int TH = NWV * MT * RPT;
    static constexpr int LH = TH + 2, LW = TW + 2;
    static constexpr int VQ = 5;
    static constexpr int ROWQ = (TW == 16) ? ((LW * VQ + 15) / 16 * 16) : LW * VQ;
    static constexpr int NPI = (LH * ROWQ + 63) / 64;   // LDS-DMA instructions (64 slots of 16 B each) per chunk-plane
    static constexpr int PLANEQ = NPI * 64;             // float4 slots per chunk-plane buffer
    static constexpr int NP = (NPI + NWV - 1) / NWV;    // pieces per wave and pass
    static constexpr int NTHR = NWV * 64;
    static constexpr int BRING = 4;                     // LDS ring of B (weight) steps: 6 fragments of 1 KB per (chunk, kh, kw) step
    static constexpr int BSTEPQ = 6 * 64;               // float4 slots per step
    // two plane buffers (pass q reads buffer q & 1, the epilogue's wave-private transpose tiles alias it once its taps are done) + B ring
    static constexpr size_t lds_bytes() { return (size_t)2 * PLANEQ * 16 + (size_t)BRING * BSTEPQ * 16; }
    static_assert((size_t)NWV * 32 * 36 * 4 <= (size_t)PLANEQ * 16, "epilogue tiles must fit into one plane buffer");
    static_assert(NP <= 7, "one piece per wave and step, all of them forced home by the waits of steps 2..8");
};

__device__ const float4 g_march_zeros[4] = {};       // source of the LDS-DMA lanes that fill padding / out-of-image slots

// INS = 1: the input is a split tensor (16-byte quads are the LDS image: asynchronous LDS-DMA staging); INS = 0: fp32 input, split while
// it is staged through registers at the start of every pass (GwcNet: dres0.0 only, which reads the volume builder's fp32 output).
template <int NWV, int TW, int OUTS, int INS>
__global__ __launch_bounds__(NWV * 64, 2) void conv_march_kernel(const ConvArgs p, const int dseg, const int nseg) {
    using G = MarchGeo<NWV, TW>;
    constexpr int MT = G::MT, TH = G::TH, ROWQ = G::ROWQ, VQ = G::VQ, PLANEQ = G::PLANEQ, NTHR = G::NTHR, NP = G::NP, NPI = G::NPI;
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    float4* const bring = smem + 2 * PLANEQ;                                   // B ring: [BRING steps][kd * 2 + hl][64 lanes] float4

    const int tid = threadIdx.x, lane = tid & 63;
    const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, hh = lane >> 5;

    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int twi = bid % p.tilesW; bid /= p.tilesW;
    const int thi = bid % p.tilesH; bid /= p.tilesH;
    const int seg = bid % nseg;
    const int b = (int)(bid / nseg);
    const int d0 = seg * dseg, d1 = (d0 + dseg < p.Di) ? d0 + dseg : p.Di;
    const int a0h = thi * TH, a0w = twi * TW;
    const int g0h = a0h - 1, g0w = a0w - 1;

    // ---- f16x3 operand ranges (as conv_mfma_kernel)
    float s_in = 1.f, s_res_inv = 1.f, s_out = 1.f;
    if (p.in_meta) s_in = INS ? p.in_meta[1] : pow2_scale(amax_read(p.in_meta));
    if (p.res && p.res_meta && (p.act & OSA_RES_SPLIT)) s_res_inv = 1.0f / p.res_meta[1];
    if (OUTS && p.coef && p.in_meta) {
        float bound = p.coef[0] * amax_read(p.in_meta) + p.coef[1];
        if (p.res && p.res_meta) bound += amax_read(p.res_meta);
        s_out = pow2_scale(bound * 1.0625f);
    }
    if (OUTS && p.out_meta && blockIdx.x == 0 && tid == 0) p.out_meta[1] = s_out;
    const float osc = (p.wscale_dev ? p.wscale_dev[1] : p.oscale) * (1.0f / s_in);
    float am = 0.f;
    unsigned amax_seen = 0u;
    if (p.out_meta) amax_seen = amax_peek(p.out_meta);

    int abase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int q = (wm * MT + m) * 32 + col;
        abase[m] = (q / TW) * ROWQ + (q % TW) * VQ + hh;
    }

    f32x16 acc[3][MT];                      // [0] output plane pd - 1 (completes with this plane), [1] pd, [2] pd + 1
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][m][r] = 0.f;

    // B operands: float4 index ((ch*27 + kd*9 + khw) * 2 + hl) * 2*CoP + hh*CoP + col   (CoP == 32)
    const int bstep = 2 * p.CoP;            // float4s between the hi and the lo image of a tap
    const int tstep = 2 * bstep;            // float4s per tap
    const int nch = p.nchunks;

    const int cq = (lane & 7) * 4, vsub = lane >> 3;        // fp32 output: 4 channels of 4 voxels
    const int c8 = (lane & 3) * 8, vs2 = lane >> 2;         // split output: 8 channels of 2 voxels
    const int actk = p.act & 15;
    const float act_ns = (actk == OSA_ACT_NONE) ? 1.f : ((actk == OSA_ACT_LEAKY) ? p.slope : 0.f);    // slope for v < 0 (none / relu / leaky)
    const bool act_relu = actk == OSA_ACT_RELU;
    const size_t ovox_b = (size_t)b * p.Do * p.Ho * p.Wo;
    float* const yb = p.y + ovox_b * p.yCs;
    const float* const resb = p.res ? p.res + ovox_b * p.rCs : nullptr;

    // ---- LDS-DMA.  An instruction moves 64 x 16 bytes: lane i -> LDS [M0 base + 16 i], from a per-lane global address.  Inline asm on
    // purpose: through the builtin the compiler orders EVERY later ds_read behind the transfer (s_waitcnt vmcnt(0) right after the issue),
    // which is exactly the wait the rings exist to avoid; the hardware orders nothing (MI355X_MICROARCH.md), the vmcnt / barrier protocol
    // below does.  Every instruction is issued by every wave with all lanes on (lanes without data fetch zeros): the vmcnt immediates
    // of the protocol count instructions.
    auto dma = [&](const char* src, const unsigned lds_byte) {
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_byte);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(m0v) : "memory");
    };
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned bring_lds = smem_lds + 2u * PLANEQ * 16u;
    const char* const zsrc = reinterpret_cast<const char*>(g_march_zeros);

    // B: a step's 6 KB = 384 float4 slots [fragment f = kd * 2 + hl][64 lanes], split evenly over the waves: wave w fetches slots
    // [w * 96, (w + 1) * 96) with NI = 2 instructions (the second covers 32 slots: its upper lanes are switched off by the exec mask,
    // the instruction itself is always issued).
    constexpr int PERW = 384 / NWV, NI = (PERW + 63) / 64;
    unsigned boff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int j = wm * PERW + i * 64 + lane, f = (j >> 6) < 6 ? (j >> 6) : 5;
        boff[i] = (unsigned)(((f >> 1) * 9 * tstep + (f & 1) * bstep + (j & 63)) * 16);
    }
    auto dma_b = [&](const int slot, const int ch, const int khw) {
        const char* base = reinterpret_cast<const char*>(p.w) + (size_t)(ch * 27 + khw) * tstep * 16;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if ((i + 1) * 64 <= PERW || i * 64 + lane < PERW)
                dma(base + boff[i], bring_lds + (unsigned)((slot * 384 + wm * PERW + i * 64) * 16));
    };

    // planes (INS): piece i of this wave is DMA instruction n = i * NWV + wave of the NPI that fill a chunk-plane buffer (a wave whose
    // n would exceed NPI - 1 repeats instruction NPI - 1: same bytes to the same slots).  Slot j = 64 n + lane -> (row, voxel, quad) of the
    // padded image; its source inside the (plane, chunk) slab, or the zero block for padding and pixels outside the image.
    unsigned poff[INS ? NP : 1];
    unsigned pvalid = 0u;
    if constexpr (INS) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int n = i * NWV + wm;
            n = n < NPI ? n : NPI - 1;
            const int j = n * 64 + lane;
            const int lh = j / ROWQ, rem = j - lh * ROWQ, lw = rem / VQ, c4 = rem - lw * VQ;
            const int gh = g0h + lh, gw = g0w + lw;
            const bool ok = lh < G::LH && lw < G::LW && c4 < 4 && (unsigned)gh < (unsigned)p.Hi && (unsigned)gw < (unsigned)p.Wi;
            poff[i] = ok ? (unsigned)(((gh * p.Wi + gw) * p.xCs + c4 * 4) * 4) : 0u;
            pvalid |= ok ? (1u << i) : 0u;
        }
    }
    const size_t plane_bytes = (size_t)p.Hi * p.Wi * p.xCs * 4;
    const char* const xb = reinterpret_cast<const char*>(p.x) + (size_t)b * p.Di * plane_bytes;
    // piece i of the (plane pd, chunk c) slab into plane buffer `buf`; pd < 0: nothing to fetch (zeros)
    auto dma_piece = [&](const int i, const int buf, const int pd, const int c) {
        if constexpr (INS) {
            int n = i * NWV + wm;
            n = n < NPI ? n : NPI - 1;
            const char* base = xb + (size_t)(pd < 0 ? 0 : pd) * plane_bytes + (size_t)c * (CC * 4);
            const bool ok = ((pvalid >> i) & 1u) && pd >= 0;
            dma(ok ? base + poff[i] : zsrc, smem_lds + (unsigned)((buf * PLANEQ + n * 64) * 16));
        }
    };

    // ---- epilogue of the finished output plane `od` (accumulator set 0): BN affine + residual + activation, NDHWC store
    auto epilogue = [&](const int od, float* const tb) {
        // folded-BN scale / shift of the channels this lane finalises (re-read per plane from L2: 4 registers x 4 not held across the tap loop)
        float4 sc[2], sh[2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int co = OUTS ? c8 + 4 * h2 : cq;
            sc[h2] = make_float4(osc, osc, osc, osc); sh[h2] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.scale) {
                sc[h2] = *reinterpret_cast<const float4*>(p.scale + co); sh[h2] = *reinterpret_cast<const float4*>(p.shift + co);
                sc[h2].x *= osc; sc[h2].y *= osc; sc[h2].z *= osc; sc[h2].w *= osc;
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int tq = (wm * MT + m) * 32;                      // first voxel of this M-tile inside the TH x TW pixel tile
            // registers -> LDS (tile[voxel][channel], row stride 36 floats)
#pragma unroll
            for (int r = 0; r < 16; ++r) tb[((r & 3) + 8 * (r >> 2) + 4 * hh) * 36 + col] = acc[0][m][r];
            if constexpr (!OUTS) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int q = tq + vsub + 8 * k;
                    const int oh = a0h + q / TW, ow = a0w + q % TW;
                    const bool ok = oh < p.Ho && ow < p.Wo;
                    const int vox = (od * p.Ho + oh) * p.Wo + ow;
                    const float4 a = *reinterpret_cast<const float4*>(tb + (vsub + 8 * k) * 36 + cq);
                    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (resb && ok) {
                        if (p.act & OSA_RES_SPLIT) {
                            const float* rs = resb + vox * p.rCs;
                            const uint2 h = *reinterpret_cast<const uint2*>(rs + split_off_hi(cq));
                            const uint2 l = *reinterpret_cast<const uint2*>(rs + split_off_lo(cq));
                            r = mul4(join_f16(h, l), s_res_inv);
                        } else r = *reinterpret_cast<const float4*>(resb + vox * p.rCs + cq);
                    }
                    float o[4] = {fmaf(a.x, sc[0].x, sh[0].x) + r.x, fmaf(a.y, sc[0].y, sh[0].y) + r.y,
                                  fmaf(a.z, sc[0].z, sh[0].z) + r.z, fmaf(a.w, sc[0].w, sh[0].w) + r.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (o[e] < 0.f) ? (act_relu ? 0.f : o[e] * act_ns) : o[e];    // relu(-inf) = 0 as in the brick form, not -inf * 0
                    if (ok) {
                        am = fmaxf(am, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
                        store16(yb + vox * p.yCs + cq, make_float4(o[0], o[1], o[2], o[3]));
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int q = tq + vs2 + 16 * k;
                    const int oh = a0h + q / TW, ow = a0w + q % TW;
                    const bool ok = oh < p.Ho && ow < p.Wo;
                    const int vox = (od * p.Ho + oh) * p.Wo + ow;
                    const int soff = (c8 >> 4) * 16 + ((c8 & 15) >> 3) * 4;        // float offset of this lane's 8 hi halves inside the voxel
                    float4 rh = make_float4(0.f, 0.f, 0.f, 0.f), rl = rh;
                    if (resb && ok) {                                              // (host: a split output takes a split residual)
                        rh = *reinterpret_cast<const float4*>(resb + vox * p.rCs + soff);
                        rl = *reinterpret_cast<const float4*>(resb + vox * p.rCs + soff + 8);
                    }
                    uint2 hq[2], lq[2];
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const float4 a = *reinterpret_cast<const float4*>(tb + (vs2 + 16 * k) * 36 + c8 + 4 * h2);
                        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (resb) {
                            const uint4 hb = __builtin_bit_cast(uint4, rh), lb = __builtin_bit_cast(uint4, rl);
                            r = h2 ? join_f16(make_uint2(hb.z, hb.w), make_uint2(lb.z, lb.w)) : join_f16(make_uint2(hb.x, hb.y), make_uint2(lb.x, lb.y));
                            r = mul4(r, s_res_inv);
                        }
                        float o[4] = {fmaf(a.x, sc[h2].x, sh[h2].x) + r.x, fmaf(a.y, sc[h2].y, sh[h2].y) + r.y,
                                      fmaf(a.z, sc[h2].z, sh[h2].z) + r.z, fmaf(a.w, sc[h2].w, sh[h2].w) + r.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (o[e] < 0.f) ? (act_relu ? 0.f : o[e] * act_ns) : o[e];    // relu(-inf) = 0 as in the brick form, not -inf * 0
                        if (ok) am = fmaxf(am, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
                        split_f16(make_float4(o[0] * s_out, o[1] * s_out, o[2] * s_out, o[3] * s_out), hq[h2], lq[h2]);
                    }
                    if (ok) {
                        float* ys = yb + vox * p.yCs + soff;
                        store16(ys, make_uint4(hq[0].x, hq[0].y, hq[1].x, hq[1].y));
                        store16(ys + 8, make_uint4(lq[0].x, lq[0].y, lq[1].x, lq[1].y));
                    }
                }
            }
        }
    };


    // ---- pass sequence.  Planes pf .. pl of the input are walked (those outside [0, Di) are zero: skipped); pass q = (plane pf + q / nch,
    // chunk q % nch) reads plane buffer q & 1; global step t = 9 q + s lives in ring slot t % 4 = (q + s) % 4.
    const int pf = (d0 - 1 > 0) ? d0 - 1 : 0, pl = (d1 < p.Di - 1) ? d1 : p.Di - 1;
    const int npass = (pl - pf + 1) * nch;
    // prologue: plane-chunk of pass 0 and the B operands of steps 0..2, all in flight together
    if constexpr (INS) {
#pragma unroll
        for (int i = 0; i < NP; ++i) dma_piece(i, 0, pf, 0);
    }
    dma_b(0, 0, 0); dma_b(1, 0, 1); dma_b(2, 0, 2);

    int pd = pf, c = 0;
    for (int q = 0; q < npass; ++q) {
        const int cn = (c + 1 < nch) ? c + 1 : 0;                       // chunk of pass q + 1
        const int pdn = (q + 1 < npass) ? ((c + 1 < nch) ? pd : pd + 1) : -1;   // its plane (-1: there is none -- zeros are fetched)
        const int sb = q & 3, cur = q & 1;
        if constexpr (!INS) {
            __syncthreads();                              // the previous pass's readers of buffer `cur` are done
            stage_brick<NTHR, PREC_F16X3, 1, 4>(p, smem + cur * PLANEQ, PLANEQ, b, c * CC, pd, g0h, g0w, tid, s_in);
        }
        if (!INS || q == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (INS: pass 0's plane; later passes arrive with their planes home: the waits of steps 2..8)
        __syncthreads();
        int ab[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) ab[m] = abase[m] + cur * PLANEQ;
        int bq[4];                                        // ring slot of step s + k: bq[(s + k) & 3]
#pragma unroll
        for (int k = 0; k < 4; ++k) bq[k] = (((sb + k) & 3) * 6) * 64 + lane;
        float4 A[2][MT][2], B[3][2];           // A: ping-pong per step; B: 3-deep rotation over (step, kd) micro-steps -- [hi, lo] of one kd, read 2 micro-steps ahead
        auto load_a = [&](float4 (&An)[MT][2], const int khw) {
            const int off = (khw / 3) * ROWQ + (khw % 3) * VQ;
#pragma unroll
            for (int m = 0; m < MT; ++m) { An[m][0] = smem[ab[m] + off]; An[m][1] = smem[ab[m] + off + 2]; }
        };
        auto load_b = [&](float4 (&Bn)[2], const int u) {       // micro-step u = step * 3 + kd of this pass
            const int kd = u % 3;
            Bn[0] = bring[bq[(u / 3) & 3] + (kd * 2) * 64]; Bn[1] = bring[bq[(u / 3) & 3] + (kd * 2 + 1) * 64];
        };
        load_b(B[0], 0);
        load_b(B[1], 1);
        load_a(A[0], 0);
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            if (s > 0) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }     // every wave's share of step s + 1 has landed
            // B operands of step s + 3 (of the next pass from s = 6 on), then one piece of the next pass's plane: the piece is YOUNGER than
            // the B transfer the end-of-step wait is after, so it stays in flight across two more steps
            if (s + 3 < 9) dma_b((sb + s + 3) & 3, c, s + 3);
            else dma_b((sb + s + 3) & 3, cn, s + 3 - 9);
            if (s < NP) dma_piece(s, cur ^ 1, pdn, cn);
            if (s + 1 < 9) load_a(A[(s + 1) & 1], s + 1);
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
                const int u = s * 3 + kd;
                if (u + 2 < 27) load_b(B[(u + 2) % 3], u + 2);      // (kd = 1, 2 reach into step s + 1: landed and published by barrier s)
                __builtin_amdgcn_sched_barrier(0);
                // kd = 0 -> output plane pd + 1 (acc[2]), kd = 1 -> pd (acc[1]), kd = 2 -> pd - 1 (acc[0]); small cross terms first
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const f16x8 a = __builtin_bit_cast(f16x8, A[s & 1][m][term == 1 ? 1 : 0]);
                        const f16x8 w = __builtin_bit_cast(f16x8, B[u % 3][term == 0 ? 1 : 0]);
                        acc[2 - kd][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, w, acc[2 - kd][m], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            // everything up to the B transfer of step s + 2 (issued a step ago) is home; younger: the piece of step s - 1, this step's B
            // transfer (NI instructions) and this step's piece
            constexpr int PCS = INS ? 1 : 0;
            wait_vmcnt(NI + ((s < NP) ? PCS : 0) + ((s >= 1 && s - 1 < NP) ? PCS : 0));
        }
        // ---- plane complete?  (last chunk of plane pd)
        if (c + 1 == nch) {
            const bool v2 = (pd - 1 >= d0) && (pd - 1 < d1);     // output plane pd - 1 completes with input plane pd
            if (v2) {
                __syncthreads();                              // every wave is past its taps: buffer `cur` becomes the transpose tiles
                epilogue(pd - 1, reinterpret_cast<float*>(smem + cur * PLANEQ) + wm * (32 * 36));
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {                    // rotate: the plane that was pd becomes pd - 1 of the next step
                acc[0][m] = acc[1][m]; acc[1][m] = acc[2][m];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[2][m][r] = 0.f;
            }
            if (q + 1 == npass && pd + 1 == d1 && pd + 1 >= p.Di) {
                // the segment ends at the tensor's last plane: plane Di is zero, so output Di - 1 (now accumulator set 0) is complete too
                __syncthreads();
                epilogue(pd, reinterpret_cast<float*>(smem + cur * PLANEQ) + wm * (32 * 36));
            }
            ++pd; c = 0;
        } else ++c;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the zero pieces / spare B transfers of the last pass)
    if (p.out_meta) {
        __syncthreads();
        publish_amax(p.out_meta, am, amax_seen, reinterpret_cast<float*>(smem));
    }
}

}  // namespace osa
