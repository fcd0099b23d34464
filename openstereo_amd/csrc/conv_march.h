// d-marching form of the 3x3x3 stride-1 "same" convolutions with 32 output channels (f16x3 mode): the V0-resolution layers of the 3-D
// aggregation networks -- GwcNet dres0 / dres1 / classif*.0 (gwcnet_disp_processor.py:40-81), PSMNet dres0 / dres1 -- which carry 56 % of
// GwcNet's 3-D MACs.
//
// The brick kernel (conv_kernel.h) stages a 6x10x10 halo brick per 4x8x8 output brick: 2.34x the input in staged bytes (2.0x measured at
// the HBM side, the d-halo is re-fetched by bricks that run 510 workgroups apart), and every A fragment it reads from LDS feeds one tap
// of one output voxel row.  Here a workgroup owns a TH x TW pixel column and WALKS along d (the classifier's trick, conv3d.hip
// classifier_march_kernel):
//   * every input plane (TH+2) x (TW+2) x 32 channels is staged ONCE per pass and serves the three output planes d-1, d, d+1 whose
//     kd = 2, 1, 0 taps it is -- three accumulator sets per wave (96 accumulator registers, 2 waves per SIMD);
//   * an A fragment (one ds_read_b128 pair) of in-plane tap (kh, kw) feeds 3 x 3 MFMAs instead of 3: LDS reads per MFMA fall 3x;
//   * staged bytes: (TH+2)(TW+2) / (TH TW) x (dseg + 2) / dseg = 1.4-1.6x of one pass instead of 2.34x.
// The packed weight stream is the brick kernel's ([chunk][tap][hi|lo][k-group][Cout][8 x fp16], tap = kd*9 + kh*3 + kw): a (chunk, kh, kw)
// step reads its three kd taps 9 tap-steps apart.  Same split arithmetic (Ahi.Blo + Alo.Bhi + Ahi.Bhi, fp32 accumulate), same operand
// ranges and the same epilogue semantics as conv_mfma_kernel; the summation ORDER differs (kd outermost), so results agree to fp32
// rounding, not bitwise.  The exact-f32 mode keeps the brick kernel (its goldens stay bit-for-bit).
#pragma once
#include "conv_kernel.h"

namespace osa {

// NWV waves per workgroup, every wave owns MT = 2 M-tiles of 32 voxels.  TW = 32: an M-tile is one row of 32 pixels; TW = 16: two rows of
// 16.  LDS image of a chunk-plane: voxels 5 slots (80 B) apart -- the 16 lanes of a ds_read_b128 group fall on 16 distinct 16-byte slots
// (mod 256 B) -- and, for TW = 16, rows a multiple of 16 slots apart (the group straddles two rows).
template <int NWV, int TW>
struct MarchGeo {
    static constexpr int MT = 2;
    static constexpr int RPT = 32 / TW;                 // rows per M-tile
    static constexpr int TH = NWV * MT * RPT;
    static constexpr int LH = TH + 2, LW = TW + 2;
    static constexpr int VQ = 5;
    static constexpr int ROWQ = (TW == 16) ? ((LW * VQ + 15) / 16 * 16) : LW * VQ;
    static constexpr int PLANEQ = LH * ROWQ;            // float4 slots per staged chunk-plane
    static constexpr int NTHR = NWV * 64;
    static constexpr int CPP = 2;                       // 16-channel chunks staged per pass
    static constexpr int BRING = 4;                     // LDS ring of B (weight) steps: 6 fragments of 1 KB per (chunk, kh, kw) step
    static constexpr int BSTEPQ = 6 * 64;               // float4 slots per step
    // planes of the pass + B ring; the epilogue's wave-private transpose tiles alias the planes (all waves are past the taps by then)
    static constexpr size_t lds_bytes() {
        return (size_t)CPP * PLANEQ * 16 + (size_t)BRING * BSTEPQ * 16;
    }
    static_assert((size_t)NWV * 32 * 36 * 4 <= (size_t)CPP * PLANEQ * 16, "epilogue tiles must fit into the plane buffer");
};

template <int NWV, int TW, int OUTS>
__global__ __launch_bounds__(NWV * 64, 2) void conv_march_kernel(const ConvArgs p, const int dseg, const int nseg) {
    using G = MarchGeo<NWV, TW>;
    constexpr int MT = G::MT, TH = G::TH, ROWQ = G::ROWQ, VQ = G::VQ, PLANEQ = G::PLANEQ, NTHR = G::NTHR;
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    float* const tbase = reinterpret_cast<float*>(smem);                       // wave-private transpose tiles of the epilogue: alias the planes
    float4* const bring = smem + G::CPP * PLANEQ;                              // B ring: [BRING steps][kd * 2 + hl][64 lanes] float4

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
    if (p.in_meta) s_in = (p.act & OSA_IN_SPLIT) ? p.in_meta[1] : pow2_scale(amax_read(p.in_meta));
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
    const int npass = p.nchunks / G::CPP;

    // folded-BN scale / shift of the channels this lane finalises
    const int cq = (lane & 7) * 4, vsub = lane >> 3;        // fp32 output: 4 channels of 4 voxels
    const int c8 = (lane & 3) * 8, vs2 = lane >> 2;         // split output: 8 channels of 2 voxels
    const int actk = p.act & 15;
    const float act_ns = (actk == OSA_ACT_NONE) ? 1.f : ((actk == OSA_ACT_LEAKY) ? p.slope : 0.f);    // slope for v < 0 (none / relu / leaky)
    const size_t ovox_b = (size_t)b * p.Do * p.Ho * p.Wo;
    float* const yb = p.y + ovox_b * p.yCs;
    const float* const resb = p.res ? p.res + ovox_b * p.rCs : nullptr;
    float* const tb = tbase + wm * (32 * 36);

    // ---- B operands.  A (chunk, kh, kw) step needs the 6 fragments (kd = 0..2) x (hi, lo) of 1 KB each -- the SAME for every wave of every
    // workgroup.  Loaded per wave from L2 (v1 of this kernel) they bound the tap loop: timing-only ablations at 8 pairs (profiles/round4/
    // march_v1_ablation*.txt): taps 2.02 ms, without the B loads 1.36 ms, B loads + LDS reads without MFMAs 1.15 ms -- the vector-memory
    // path delivers ~37 B/clk/CU of 16-byte-per-lane loads and every wave pulls 6 KB per 18 MFMAs through it, L1 hit or not.  Now ONE
    // wave fetches a fragment for the whole workgroup with one LDS-DMA instruction (global_load_lds_dwordx4: lane i -> 16 bytes at row + 16 i,
    // exactly the fragment's order in the packed stream), two steps ahead into a 3-deep LDS ring; every wave reads its operands from
    // there with conflict-free ds_read_b128 (LDS: 256 B/clk/CU).  One s_barrier per step publishes the landed step.
    // A step's 6 KB = 384 float4 slots [fragment f = kd * 2 + hl][64 lanes], split evenly over the waves: wave w fetches slots
    // [w * PERW, (w + 1) * PERW) with NI instructions (the last one lane-predicated when PERW is not a multiple of 64).  The LDS
    // destination of an instruction is linear (M0 base + 16 * lane), the global source per lane is the slot's place in the packed stream.
    // Inline asm on purpose: through the builtin the compiler orders EVERY later ds_read behind the transfer (s_waitcnt vmcnt(0) right
    // after the issue), which is exactly the wait this ring exists to avoid; the hardware orders nothing, the barrier protocol below does.
    constexpr int PERW = 384 / NWV, NI = (PERW + 63) / 64;
    const unsigned bring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)bring;
    unsigned boff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int j = wm * PERW + i * 64 + lane, f = j >> 6;
        boff[i] = (unsigned)(((f >> 1) * 9 * tstep + (f & 1) * bstep + (j & 63)) * 16);
    }
    auto dma_b = [&](const int buf, const int ch, const int khw) {
        const char* base = reinterpret_cast<const char*>(p.w) + (size_t)(ch * 27 + khw) * tstep * 16;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if ((i + 1) * 64 <= PERW || i * 64 + lane < PERW) {
                const char* src = base + boff[i];
                const unsigned m0v = __builtin_amdgcn_readfirstlane(bring_lds + (unsigned)((buf * 384 + wm * PERW + i * 64) * 16));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(m0v) : "memory");
            }
        }
    };
    // ---- one pass: taps of the two staged chunks [ch0, ch0 + 2) of plane pd, all three kd.  At the two ends of a D segment one or two
    // of the three output planes lie outside [d0, d1): their sums are computed and dropped (2 plane-steps of MFMAs per cut segment, 2/3
    // of one for an uncut column) -- a second code path over the accumulators (compile-time kd masks were tried) makes this compiler
    // spill 170-240 registers, which costs far more.
    // Ring protocol (BRING = 4 slots, step t lives in slot t % 4).  Step s, after barrier s: issue the transfer of step s + 3 into the slot
    // step s - 1 was read from (every wave is past barrier s, i.e. done with step s - 1); read the operands of the next micro-steps from
    // slots s and s + 1; at the end of the step wait until all but the newest transfer have landed (vmcnt(NI): step s + 2 is home, step
    // s + 3 keeps flying -- two steps of latency tolerance); barrier s + 1 then publishes step s + 2 to every wave.
    // Caller: dma_b(t, ch0, t) for t = 0, 1, 2 issued before the plane was staged (they land behind the staging barrier).
    // (OSA_M2_*: timing-only ablations, tools/r4/build_march_variant.sh -- results wrong by construction)
    auto run_pass = [&](const int ch0) {
        float4 A[2][MT][2], B[3][2];           // A: ping-pong per step; B: 3-deep rotation over (step, kd) micro-steps -- [hi, lo] of one kd, read 2 micro-steps ahead
        auto load_a = [&](float4 (&An)[MT][2], const int cl, const int khw) {
            const int off = cl * PLANEQ + (khw / 3) * ROWQ + (khw % 3) * VQ;
#pragma unroll
            for (int m = 0; m < MT; ++m) { An[m][0] = smem[abase[m] + off]; An[m][1] = smem[abase[m] + off + 2]; }
        };
        auto load_b = [&](float4 (&Bn)[2], const int u) {       // micro-step u = step * 3 + kd
#if !defined(OSA_M2_NOBREAD)
            const int buf = (u / 3) % G::BRING, kd = u % 3;
            Bn[0] = bring[(buf * 6 + kd * 2) * 64 + lane]; Bn[1] = bring[(buf * 6 + kd * 2 + 1) * 64 + lane];
#endif
        };
        load_b(B[0], 0);
        load_b(B[1], 1);
        load_a(A[0], 0, 0);
#if defined(OSA_M2_NOBREAD)
        B[0][0] = B[0][1] = B[1][0] = B[1][1] = B[2][0] = B[2][1] = smem[abase[0]];
#endif
#pragma unroll
        for (int s = 0; s < 18; ++s) {
#if !defined(OSA_M2_NOBAR)
            if (s > 0) __builtin_amdgcn_s_barrier();          // every wave's share of step s + 1 has landed
#endif
#if !defined(OSA_M2_NODMA)
            if (s + 3 < 18) dma_b((s + 3) % G::BRING, ch0 + (s + 3) / 9, (s + 3) % 9);
#endif
            if (s + 1 < 18) load_a(A[(s + 1) & 1], (s + 1) / 9, (s + 1) % 9);
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
                const int u = s * 3 + kd;
                if (u + 2 < 54) load_b(B[(u + 2) % 3], u + 2);      // (kd = 1, 2 reach into step s + 1: landed and published by barrier s)
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
#if !defined(OSA_M2_NOWAIT)
            // the transfer of step s + 2 (issued a step ago) has landed; the one just issued keeps flying
            if (s + 3 < 18) { if constexpr (NI == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else if constexpr (NI == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        }
    };

    // ---- epilogue of the finished output plane `od` (accumulator set 0): BN affine + residual + activation, NDHWC store
    auto epilogue = [&](const int od) {
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
                    for (int e = 0; e < 4; ++e) o[e] = (o[e] < 0.f) ? o[e] * act_ns : o[e];
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
                        for (int e = 0; e < 4; ++e) o[e] = (o[e] < 0.f) ? o[e] * act_ns : o[e];
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

    for (int pd = d0 - 1; pd <= d1; ++pd) {
        const bool v2 = (pd - 1 >= d0) && (pd - 1 < d1);      // output plane pd - 1 completes with this input plane
        if (pd >= 0 && pd < p.Di) {                       // (planes outside the tensor are zero: nothing to add)
            for (int pass = 0; pass < npass; ++pass) {
                __syncthreads();                          // the previous pass's readers (planes and B ring) / the epilogue's tiles are done
                dma_b(0, pass * G::CPP, 0);
                dma_b(1, pass * G::CPP, 1);
                dma_b(2, pass * G::CPP, 2);
#if !defined(OSA_M2_NOSTAGE)
                stage_brick<NTHR, PREC_F16X3, 2, 8>(p, smem, PLANEQ, b, pass * (G::CPP * CC), pd, g0h, g0w, tid, s_in);
#endif
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                run_pass(pass * G::CPP);
            }
        }
#if defined(OSA_M2_NOEPI)
        if (v2) {
            float t_ = 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) t_ += acc[0][m][r];
            if (t_ == 12345.678f) yb[0] = t_;
        }
#else
        if (v2) {
            __syncthreads();                              // every wave is past its taps: the plane buffer becomes the transpose tiles
            epilogue(pd - 1);
        }
#endif
        // rotate: the plane that was pd becomes pd - 1 of the next step
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[0][m] = acc[1][m]; acc[1][m] = acc[2][m];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[2][m][r] = 0.f;
        }
    }
    if (p.out_meta) {
        __syncthreads();
        publish_amax(p.out_meta, am, amax_seen, reinterpret_cast<float*>(smem));
    }
}

}  // namespace osa
