// d-marching form of the 3x3x3 STRIDE-2 convolutions with 64 output channels (f16x3 mode, split tensors in and out): conv1 of the GwcNet /
// PSMNet hourglasses (models/gwcnet/hourglass.py:19-24, 32 -> 64 channels, V0 -> V1 resolution; three launches per GwcNet forward).
//
// The brick kernel stages a 5 x 9 x 17 strided halo brick per 2 x 4 x 8 output brick: 2.16x the algorithmic bytes at the HBM side (r5
// counters), 151 staged bytes per MFMA, and the stride-2 A-fragment reads hit the LDS banks two ways.  The layer is HBM-bound (2.26 GB per
// 9-pair launch against 0.23 ms of matrix time), so what this form buys is traffic:
//   * a workgroup owns a 4 x 32 OUTPUT pixel column and walks along d.  Input plane 2 od feeds output plane od (kd = 1), input plane
//     2 od + 1 feeds od (kd = 2) and od + 1 (kd = 0): TWO accumulator sets, every input plane staged ONCE -- (2 TH + 1)(2 TW + 1) /
//     (4 TH TW) = 1.14x of the input, no re-read along d inside a segment;
//   * planes are staged by LDS-DMA (split input: the 16-byte quads are the LDS image) into a PARITY-PLANAR image: a row of the 65-voxel
//     footprint is stored [33 even columns | 32 odd columns], so the 32 lanes of an M-tile (one output row, 32 pixels) read CONSECUTIVE
//     voxels for every kw (even, odd, even + 1) -- the de-interleave costs nothing, it is the per-lane source address of the transfer.
//     Voxels are 64 B apart (no padding slot); the 16-byte quad index is XOR-swizzled with bits 2-3 of the voxel index, which makes any
//     16 consecutive voxels conflict-free for ds_read_b128 (4 v + (q ^ (v >> 2 & 3)) mod 16 is a bijection of v mod 16);
//   * 8 waves = 4 M-tiles (output rows) x 2 N-tiles (32 output channels each), ONE workgroup per CU: the weights of a step are fetched once
//     per CU into a 3-slot LDS ring (a step = one (chunk, kh, kd): 3 kw taps x [hi | lo] x 2 N-tiles = 12 KB, 9 MFMAs per wave, 18 per
//     SIMD between barriers), three steps = one BLOCK ahead of their use; planes are TRIPLE-buffered: pass q computes from buffer q % 3
//     while the planes of passes q + 1 and q + 2 land (an even pass is three steps, ~1 us: less than an HBM round trip under load -- with
//     two buffers the plane transfers cost 0.47 of 1.18 ms, profiles/round6/march_s2_ablation.txt).  A wave's vmcnt counts in order, so
//     the transfer roles are split: waves 0-3 fetch weights (3 fragments per step, needed a block later), waves 4-7 planes (10 pieces per
//     pass, needed two passes later) -- a long-latency piece never sits in front of a weight transfer in the same wave's queue;
//   * the unit of control is a block of three steps (kh = 0, 1, 2 of one (plane, chunk, kd)), fully unrolled: ring slot = kh, every LDS
//     address is a register computed once per pass plus an immediate, the weight transfers take their base from an SGPR pair.  The first
//     version walked a generic step loop: ~290 instructions per 9 MFMAs, and its timing-only ablation ran 0.62 ms of its 1.0 ms with
//     neither transfers nor MFMAs (profiles/round6/march_s2_ablation.txt) -- the layer was bound by instruction issue, not by bytes;
//   * fragment reads are software-pipelined: while the 9 MFMAs of step t run from one register set, the 12 ds_read_b128 of step t + 1
//     fill the next (every wave of the workgroup sits at the same barrier: without this the LDS -- 96 KB per step, 384 cycles -- and the
//     matrix pipes -- 576 cycles per SIMD and step -- take turns instead of overlapping).
// Same split arithmetic (Ahi.Blo + Alo.Bhi + Ahi.Bhi, fp32 accumulate), operand ranges and epilogue semantics as conv_mfma_kernel; the
// summation order differs (plane-major), so results agree with the brick form to fp32 rounding, not bitwise.
#pragma once
#include "conv_kernel.h"

namespace osa {

struct MarchS2Geo {
    static constexpr int NWV = 8, TW = 32, TH = 4;
    static constexpr int LH = 2 * TH + 1, LW = 2 * TW + 1, NEV = TW + 1;       // footprint rows / columns, even columns per row
    static constexpr int NVOX = LH * LW;                                       // 585 voxels, 4 quads each
    static constexpr int NPI = (NVOX * 4 + 63) / 64;                           // LDS-DMA instructions per chunk-plane (37)
    static constexpr int PLANEQ = NPI * 64;                                    // float4 slots per plane buffer
    static constexpr int NLW = NWV / 2;                                        // waves 0 .. NLW-1 issue the weight transfers, NLW .. NWV-1 the plane transfers
    static constexpr int NP = (NPI + NLW - 1) / NLW;                           // plane pieces per plane-loader wave and pass (10)
    static constexpr int NPB = 3;                                              // plane buffers: pass q computes from q % 3 while q + 1 and q + 2 land
    static constexpr int BRING = 3, BSTEPQ = 12 * 64;                          // ring slots, float4 slots per step (12 fragments of 1 KB)
    static constexpr int NIB = 12 / NLW;                                       // B transfers per weight-loader wave and step (3)
    static constexpr size_t lds_bytes() { return (size_t)NPB * PLANEQ * 16 + (size_t)BRING * BSTEPQ * 16; }
    static_assert((size_t)NPB * PLANEQ * 16 + (size_t)BRING * BSTEPQ * 16 <= 160 * 1024, "one workgroup per CU: all of its LDS");
    static_assert((size_t)NWV * 32 * 36 * 4 <= (size_t)PLANEQ * 16, "epilogue tiles must fit into one plane buffer");
};

__device__ const float4 g_march_s2_zeros[4] = {};

template <int N>
__device__ __forceinline__ void wait_vmcnt_c() {
    static_assert(N >= 0 && N <= 15, "vmcnt immediate");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else static_assert(N <= 12, "extend the table");
}

// split input, split output, no residual (conv1 of the hourglasses); oseg output planes per segment
__global__ __launch_bounds__(512, 2) void conv_march_s2_kernel(const ConvArgs p, const int oseg, const int nseg) {
    using G = MarchS2Geo;
    constexpr int PLANEQ = G::PLANEQ, NP = G::NP, NPI = G::NPI, NLW = G::NLW, NIB = G::NIB, TH = G::TH, TW = G::TW, LW = G::LW, NEV = G::NEV;
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    float4* const bring = smem + G::NPB * PLANEQ;
#ifdef OSA_EXPERIMENTS
    const int dbg = p.dbg;                                    // timing-only ablations (tools/bench_layers.py --dbgs): 1 no weight transfers, 2 no plane transfers, 4 no MFMAs, 8 no epilogue
#else
    constexpr int dbg = 0;
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv & 3, wn = wv >> 2;                      // M-tile (output row of the column), N-tile (32 output channels)
    const bool w_loader = wv < NLW;                           // this wave issues the weight transfers (the others: the plane transfers) -- a wave's
                                                              // vmcnt counts in order, so the long-latency plane pieces must not sit in front of transfers
                                                              // that are needed a step later
    const int lw = w_loader ? wv : wv - NLW;                  // index among the waves of its role
    const int col = lane & 31, hh = lane >> 5;

    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int twi = bid % p.tilesW; bid /= p.tilesW;
    const int thi = bid % p.tilesH; bid /= p.tilesH;
    const int seg = bid % nseg;
    const int b = (int)(bid / nseg);
    const int o0 = seg * oseg, o1 = (o0 + oseg < p.Do) ? o0 + oseg : p.Do;
    const int a0h = thi * TH, a0w = twi * TW;
    const int g0h = 2 * a0h - 1, g0w = 2 * a0w - 1;

    // ---- f16x3 operand ranges (as conv_mfma_kernel / conv_march_kernel with split input and output)
    float s_in = 1.f, s_out = 1.f;
    if (p.in_meta) s_in = p.in_meta[1];
    if (p.coef && p.in_meta) s_out = pow2_scale((p.coef[0] * amax_read(p.in_meta) + p.coef[1]) * 1.0625f);
    if (p.out_meta && blockIdx.x == 0 && tid == 0) p.out_meta[1] = s_out;
    const float osc = (p.wscale_dev ? p.wscale_dev[1] : p.oscale) * (1.0f / s_in);
    float am = 0.f;
    unsigned amax_seen = 0u;
    if (p.out_meta) amax_seen = amax_peek(p.out_meta);

    f32x16 acc0, acc1;                                         // output plane od (kd = 1, 2 land here), od + 1 (kd = 0)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    const int CoP = p.CoP;                                     // == 64
    const int nch = p.nchunks;
    const int c8 = (lane & 3) * 8, vs2 = lane >> 2;            // epilogue: 8 channels of 2 voxels per lane
    const int actk = p.act & 15;
    const float act_ns = (actk == OSA_ACT_NONE) ? 1.f : ((actk == OSA_ACT_LEAKY) ? p.slope : 0.f);
    const bool act_relu = actk == OSA_ACT_RELU;
    float* const yb = p.y + (size_t)b * p.Do * p.Ho * p.Wo * p.yCs;

    // ---- LDS-DMA (conv_march.h: every instruction is issued by every wave with all lanes on; the vmcnt immediates count instructions)
    auto dma = [&](const char* src, const unsigned lds_byte) {
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_byte);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(m0v) : "memory");
    };
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned bring_lds = smem_lds + (unsigned)G::NPB * PLANEQ * 16u;
    const char* const zsrc = reinterpret_cast<const char*>(g_march_s2_zeros);

    // B: a step's 12 fragments f = (kw * 2 + hl) * 2 + n, 64 lanes x 16 B each; weight-loader wave w fetches fragments 3 w .. 3 w + 2.
    // packed weights: 16-byte unit ((ch * 27 + t) * 4 + hl * 2 + kg) * CoP + co, t = kd * 9 + kh * 3 + kw  (conv3d.hip pack_weights_f16x3).
    // The step's base (ch, kd, kh) is wave-uniform: an SGPR pair; the lane's share is a 32-bit offset (saddr form of the transfer).
    unsigned boff[NIB], bdst[NIB];
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
        const int f = lw * NIB + i;
        const int kw = f >> 2, hl = (f >> 1) & 1, n = f & 1;
        boff[i] = (unsigned)((kw * 4 * CoP + hl * 2 * CoP + hh * CoP + n * 32 + col) * 16);
        bdst[i] = bring_lds + (unsigned)(f * 64 * 16);
    }
    // Weight loaders move their fragments L2 -> registers -> LDS with ordinary loads and ds_write_b128 (r6 v5): the LDS-DMA path delivers
    // ~12 B/clk per CU whatever the source (MI355X_MICROARCH.md ldsdma-fill; this kernel at 92 KB per pass ran 4.3-4.7 us per pass), and the
    // weights were 54 of those 92 KB.  Ordinary 16-byte loads run ~3x that rate, the planes keep the DMA path to themselves, and everything a
    // weight-loader wave issues is visible to the compiler (its waits are exact: these waves issue no asm transfers).
    typedef float v4f_t __attribute__((ext_vector_type(4)));
    auto load_b = [&](v4f_t (&R)[NIB], const char* stepbase) {       // stepbase: wave-uniform
#pragma unroll
        for (int i = 0; i < NIB; ++i) R[i] = *reinterpret_cast<const v4f_t*>(stepbase + boff[i]);
    };
    auto store_b = [&](const v4f_t (&R)[NIB], const int slot) {       // slot: constant after unrolling
#pragma unroll
        for (int i = 0; i < NIB; ++i)
            *reinterpret_cast<__attribute__((address_space(3))) v4f_t*>((size_t)(bdst[i] + (unsigned)(lane * 16 + slot * G::BSTEPQ * 16))) = R[i];
    };
    const size_t kh_bytes = (size_t)3 * 4 * CoP * 16;           // bytes between the kh rows of a (chunk, kd): 3 taps
    auto block_base = [&](const int ch, const int kd) { return reinterpret_cast<const char*>(p.w) + (size_t)((ch * 27 + kd * 9) * 4 * CoP) * 16; };

    // planes: piece i of plane-loader wave w is DMA instruction n = i * NLW + w (beyond NPI - 1: instruction NPI - 1 again).  LDS slot j = 64 n + lane
    // -> swizzled quad of voxel v = j >> 2 of the parity-planar image: row lh = v / 65, r = v % 65, column lw = 2 r (r < 33) or 2 (r - 33) + 1
    unsigned poff[NP];
    unsigned pvalid = 0u;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        int n = i * NLW + lw;
        n = n < NPI ? n : NPI - 1;
        const int j = n * 64 + lane;
        const int v = j >> 2, q = (j & 3) ^ ((v >> 2) & 3);
        const int lh = v / LW, r = v - lh * LW;
        const int lw = (r < NEV) ? 2 * r : 2 * (r - NEV) + 1;
        const int gh = g0h + lh, gw = g0w + lw;
        const bool ok = v < G::NVOX && (unsigned)gh < (unsigned)p.Hi && (unsigned)gw < (unsigned)p.Wi;
        poff[i] = ok ? (unsigned)(((gh * p.Wi + gw) * p.xCs + q * 4) * 4) : 0u;
        pvalid |= ok ? (1u << i) : 0u;
    }
    const size_t plane_bytes = (size_t)p.Hi * p.Wi * p.xCs * 4;
    const char* const xb = reinterpret_cast<const char*>(p.x) + (size_t)b * p.Di * plane_bytes;
    auto dma_plane = [&](const int buf, const int pd, const int c) {          // pd < 0: nothing to fetch (zeros: the instruction count stays the same)
        const char* base = xb + (size_t)(pd < 0 ? 0 : pd) * plane_bytes + (size_t)c * (CC * 4);
        if (dbg & 2) return;                                 // (ablation: no plane transfers)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int n = i * NLW + lw;
            n = n < NPI ? n : NPI - 1;
            const bool ok = ((pvalid >> i) & 1u) && pd >= 0;
            dma(ok ? base + poff[i] : zsrc, smem_lds + (unsigned)((buf * PLANEQ + n * 64) * 16));
        }
    };

    // ---- epilogue of the finished output plane od (accumulator set 0): BN affine + activation, split NDHWC store
    auto epilogue = [&](const int od, float* const tb) {
        float4 sc[2], sh[2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int co = wn * 32 + c8 + 4 * h2;
            sc[h2] = make_float4(osc, osc, osc, osc); sh[h2] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.scale) {
                sc[h2] = *reinterpret_cast<const float4*>(p.scale + co); sh[h2] = *reinterpret_cast<const float4*>(p.shift + co);
                sc[h2].x *= osc; sc[h2].y *= osc; sc[h2].z *= osc; sc[h2].w *= osc;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) tb[((r & 3) + 8 * (r >> 2) + 4 * hh) * 36 + col] = acc0[r];
        const int oh = a0h + wm;
        const int soff = wn * 32 + (c8 >> 4) * 16 + ((c8 & 15) >> 3) * 4;      // float offset of this lane's 8 hi halves inside the voxel
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int ow = a0w + vs2 + 16 * k;
            const bool ok = oh < p.Ho && ow < p.Wo;
            const int vox = (od * p.Ho + oh) * p.Wo + ow;
            uint2 hq[2], lq[2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const float4 a = *reinterpret_cast<const float4*>(tb + (vs2 + 16 * k) * 36 + c8 + 4 * h2);
                float o[4] = {fmaf(a.x, sc[h2].x, sh[h2].x), fmaf(a.y, sc[h2].y, sh[h2].y), fmaf(a.z, sc[h2].z, sh[h2].z), fmaf(a.w, sc[h2].w, sh[h2].w)};
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (o[e] < 0.f) ? (act_relu ? 0.f : o[e] * act_ns) : o[e];
                if (ok) am = fmaxf(am, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
                split_f16(make_float4(o[0] * s_out, o[1] * s_out, o[2] * s_out, o[3] * s_out), hq[h2], lq[h2]);
            }
            if (ok) {
                float* ys = yb + (size_t)vox * p.yCs + soff;
                store16(ys, make_uint4(hq[0].x, hq[0].y, hq[1].x, hq[1].y));
                store16(ys + 8, make_uint4(lq[0].x, lq[0].y, lq[1].x, lq[1].y));
            }
        }
    };

    // ---- per-lane LDS byte addresses of the A fragments in plane buffer 0: row kh, tap kw -> hi quad; the lo quad is the address ^ 32
    // (quad index q ^ 2); + PLANEQ * 16 for buffer 1.  B fragments: one base per lane + immediates.
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const v4f lds_f4;
    unsigned aaddr[3][3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int v = (2 * wm + kh) * LW + col + (kw == 1 ? NEV : (kw >> 1));
            aaddr[kh][kw] = smem_lds + (unsigned)((4 * v + (hh ^ ((v >> 2) & 3))) * 16);
        }
    const unsigned baddr = bring_lds + (unsigned)((wn * 64 + lane) * 16);
    auto lds16 = [](const unsigned a) -> v4f { return *reinterpret_cast<lds_f4*>((size_t)a); };
    struct Frags { v4f A[3][2], B[3][2]; };
    auto read_frags = [&](Frags& F, const unsigned a0, const unsigned a1, const unsigned a2, const int slot) {   // slot: constant after unrolling
        const unsigned a[3] = {a0, a1, a2};
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            F.A[kw][0] = lds16(a[kw]);
            F.A[kw][1] = lds16(a[kw] ^ 32u);
            F.B[kw][0] = lds16(baddr + (unsigned)((slot * 12 + (kw * 2 + 0) * 2) * 64 * 16));
            F.B[kw][1] = lds16(baddr + (unsigned)((slot * 12 + (kw * 2 + 1) * 2) * 64 * 16));
        }
    };
    auto mfmas = [&](f32x16& acc, const Frags& F) {
        if (dbg & 4) {                                       // (timing-only ablation, experiments build: LDS reads without the MFMAs)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) acc[kw] += F.A[kw][0].x + F.A[kw][1].x + F.B[kw][0].x + F.B[kw][1].x;
            return;
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int term = 0; term < 3; ++term) {                      // small cross terms first (as the other forms)
                const f16x8 a = __builtin_bit_cast(f16x8, F.A[kw][term == 1 ? 1 : 0]);
                const f16x8 w = __builtin_bit_cast(f16x8, F.B[kw][term == 0 ? 1 : 0]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, w, acc, 0, 0, 0);
            }
    };

    // ---- pass / block sequence.  Planes pf .. pl are walked; pass q = (plane pd, chunk c) reads plane buffer q & 1.  A BLOCK is the three
    // steps kh = 0, 1, 2 of one (pass, kd); step kh uses ring slot kh.  An even plane has one block (kd = 1 into accumulator set 0), an odd
    // plane the block kd = 2 into set 0 -- the output plane od = pd >> 1 it completes, if that plane belongs to this segment -- and the
    // block kd = 0 into set 1 (od + 1, if that one does).
    const int pf = (2 * o0 - 1 > 0) ? 2 * o0 - 1 : 0;
    const int pl = (2 * o1 - 1 < p.Di - 1) ? 2 * o1 - 1 : p.Di - 1;
    auto has0 = [&](const int pd) { return !(pd & 1) || (pd >> 1) >= o0; };              // does plane pd have a block into set 0 / set 1?
    auto has1 = [&](const int pd) { return (pd & 1) && (pd >> 1) + 1 < o1; };
    auto first_kd = [&](const int pd) { return (pd & 1) ? (has0(pd) ? 2 : 0) : 1; };

    Frags F0, F1, F2;
    v4f_t Bpend[NIB];                                             // weight loaders: the fragments requested a step ago
#pragma unroll
    for (int i = 0; i < NIB; ++i) Bpend[i] = (v4f_t){0.f, 0.f, 0.f, 0.f};
    // One block.  `wnext`: weight base of the next block (past the end: of this block again -- the instruction counts of the waits stay what
    // they are); `anext`: A addresses (row kh = 0) of the next block's first step; `first` / `lastb`: first / last block of its pass; with
    // `first` the plane-loader waves send out the plane of pass q + 2 (buffer fbuf, plane fpd, chunk fc) in step 0.
    // Waits.  Weight loaders: the compiler's (ordinary loads, stored to the ring a step after they were requested).  Plane loaders:
    // one step before a pass ends the plane of the NEXT pass is home -- its first fragments are read in the pass's last step -- (younger: the
    // NP pieces of pass q + 2, issued in this pass's step 0).
    auto block = [&](auto SEL1, const unsigned aoff, const unsigned anoff, const char* wnext, const bool first, const bool lastb,
                     const int fbuf, const int fpd, const int fc) {        // aoff / anoff: byte offset of this block's / the next block's plane buffer
        auto step = [&](auto KH, const Frags& Fc, Frags& Fn) {
            constexpr int kh = decltype(KH)::value;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this step's fragments (read a step ago from ring slot kh) are in registers BEFORE the
                                                              // barrier behind which the weight loaders overwrite that slot
            __syncthreads();                                  // every loader's share of the next step's B (and, before a pass's last step, of the next
                                                              // pass's plane) has landed: each loader waited for its own before it arrived
            if (w_loader) {
                store_b(Bpend, (kh + 2) % 3);                 // what was requested a step ago: the previous step's slot, free since the barrier above
                if (!(dbg & 1)) load_b(Bpend, wnext + (size_t)kh * kh_bytes);   // the same step of the next block
            } else if (first && kh == 0) dma_plane(fbuf, fpd, fc);
            if constexpr (kh < 2) read_frags(Fn, aaddr[kh + 1][0] + aoff, aaddr[kh + 1][1] + aoff, aaddr[kh + 1][2] + aoff, kh + 1);
            else read_frags(Fn, aaddr[0][0] + anoff, aaddr[0][1] + anoff, aaddr[0][2] + anoff, 0);   // (past the end: read and dropped)
            if constexpr (decltype(SEL1)::value) mfmas(acc1, Fc); else mfmas(acc0, Fc);
            if (!w_loader && lastb && kh == 1) wait_vmcnt_c<NP>();
        };
        step(std::integral_constant<int, 0>{}, F0, F1);
        step(std::integral_constant<int, 1>{}, F1, F2);
        step(std::integral_constant<int, 2>{}, F2, F0);
    };

    // pass q -> (plane, chunk) without divisions: the sequence is walked incrementally; planes of passes 0 and 1 go out in the prologue
    const int npass = (pl - pf + 1) * nch;
    if (!w_loader) {
        dma_plane(0, pf, 0);
        if (npass > 1) dma_plane(1, nch > 1 ? pf : pf + 1, nch > 1 ? 1 : 0); else dma_plane(1, -1, 0);
    } else {
        const char* w0 = block_base(0, first_kd(pf));
        load_b(Bpend, w0); store_b(Bpend, 0);
        load_b(Bpend, w0 + kh_bytes); store_b(Bpend, 1);
        load_b(Bpend, w0 + 2 * kh_bytes);                         // (written in step 0, read in step 1)
    }
    if (!w_loader) wait_vmcnt_c<0>();
    __syncthreads();
    read_frags(F0, aaddr[0][0], aaddr[0][1], aaddr[0][2], 0);

    int q = 0, qb = 0;                                            // pass index, q % 3
    int f2pd = pf, f2c = 0;                                       // (plane, chunk) of pass q + 2
    for (int i2 = 0; i2 < 2; ++i2) { if (++f2c == nch) { f2c = 0; ++f2pd; } }
    for (int pd = pf; pd <= pl; ++pd) {
        const bool odd = pd & 1;
        const int od = pd >> 1;
        const bool b0 = has0(pd), b1 = has1(pd);
        for (int c = 0; c < nch; ++c, ++q) {
            const int qn = qb == 2 ? 0 : qb + 1, qf = qn == 2 ? 0 : qn + 1;      // buffers of pass q + 1, q + 2
            const unsigned aoff = (unsigned)(qb * PLANEQ * 16), anoff = (unsigned)(qn * PLANEQ * 16);
            // the pass after this one
            int npd = pd, nc = c + 1;
            if (nc == nch) { nc = 0; ++npd; }
            const bool more = npd <= pl;
            const char* wfirst_next = more ? block_base(nc, first_kd(npd)) : nullptr;
            const int fpd = f2pd <= pl ? f2pd : -1;
            // blocks of this pass: even plane: (kd = 1 -> set 0); odd plane: (kd = 2 -> set 0) if b0, then (kd = 0 -> set 1) if b1
            const int nblk = odd ? (int)b0 + (int)b1 : 1;
            for (int bi = 0; bi < nblk; ++bi) {
                const bool last = bi + 1 == nblk;                  // the next block is the first block of the next pass (or there is none)
                const bool sel1 = odd && (bi == 1 || !b0);
                const int kd_here = odd ? (sel1 ? 0 : 2) : 1;
                const char* wn = last ? (more ? wfirst_next : block_base(c, kd_here)) : block_base(c, 0);
                if (sel1) block(std::true_type{}, aoff, last ? anoff : aoff, wn, bi == 0, last, qf, fpd, f2c);
                else block(std::false_type{}, aoff, last ? anoff : aoff, wn, bi == 0, last, qf, fpd, f2c);
            }
            if (c + 1 == nch) {                                   // plane pd complete: does an output plane complete with it?
                const bool fin = odd || pd == p.Di - 1;           // (an even LAST plane: plane pd + 1 lies outside the tensor)
                // buffer qb becomes the wave-private transpose tiles: its last readers (the last step's fragments, read a step earlier) had their
                // data before that step's barrier, the prefetch of the next step reads another buffer, and the transfer that refills this one is
                // issued behind the next barrier
                if (fin && od >= o0 && od < o1 && !(dbg & 8)) epilogue(od, reinterpret_cast<float*>(smem + qb * PLANEQ) + wv * (32 * 36));
                if (odd) {
                    acc0 = acc1;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
                }
            }
            qb = qn;
            if (++f2c == nch) { f2c = 0; ++f2pd; }
        }
    }
    if (!w_loader) wait_vmcnt_c<0>();                             // (the zero pieces of the passes past the end)
    if (p.out_meta) {
        __syncthreads();
        publish_amax(p.out_meta, am, amax_seen, reinterpret_cast<float*>(smem));
    }
}

}  // namespace osa
