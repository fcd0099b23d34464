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
//     SIMD between barriers); the plane of pass q + 1 lands in the second plane buffer while pass q computes.
// r6 measured [MI355X], 9 pairs per launch: HBM traffic 2.37 GB = 1.05x the algorithmic bytes (brick form: 4.89 GB = 2.16x), 1.02 ms against
// the brick form's 1.10 ms on the same box; the default bench line (three sub-batch streams) gains 0.9 %.  Four restructured versions
// (static 3-step blocks with software-pipelined fragment reads, triple-buffered planes with role-split loader waves, weights by ordinary
// loads instead of LDS-DMA: tools/experiments/conv_march_s2_v5.h) cut the instructions per step from ~290 to ~65 and did NOT run faster
// (1.13-1.16 ms): with ONE 8-wave workgroup per CU in barrier lockstep and 9 MFMAs per wave and step, a step takes ~2100 cycles whatever
// moves the bytes -- 43 % of the wave cycles issue instructions (25 non-MFMA instructions per MFMA by the SQ counters), 31 % wait at
// barriers / counters, LDS bank conflicts 1.7 %.  What would change it is more MFMAs per barrier (a second M-tile per wave needs plane
// buffers this LDS cannot hold twice) or two independent workgroups per CU.  profiles/round6/march_s2_versions.txt has every line.
// Same split arithmetic (Ahi.Blo + Alo.Bhi + Ahi.Bhi, fp32 accumulate), operand ranges and epilogue semantics as conv_mfma_kernel; the
// summation order differs (plane-major), so results agree with the brick form to fp32 rounding, not bitwise.
#pragma once
#include "conv_kernel.h"

namespace osa {

// NWV = 8: a 4 x 32 output column, ONE workgroup per CU (112 KB of LDS); NWV = 4: a 2 x 32 column, TWO independent workgroups per CU (2 x 80 KB:
// while one sits at its step barrier the other multiplies) at the price of a taller halo (1.27x instead of 1.14x) and the weights fetched twice per CU.
template <int NWV_>
struct MarchS2Geo {
    static constexpr int NWV = NWV_, TW = 32, TH = NWV_ / 2;
    static constexpr int LH = 2 * TH + 1, LW = 2 * TW + 1, NEV = TW + 1;       // footprint rows / columns, even columns per row
    static constexpr int NVOX = LH * LW;                                       // 585 voxels, 4 quads each
    static constexpr int NPI = (NVOX * 4 + 63) / 64;                           // LDS-DMA instructions per chunk-plane (37)
    static constexpr int PLANEQ = NPI * 64;                                    // float4 slots per plane buffer
    static constexpr int NP = (NPI + NWV - 1) / NWV;                           // pieces per wave and pass (5)
    static constexpr int BRING = 3, BSTEPQ = 12 * 64;                          // ring slots, float4 slots per step (12 fragments of 1 KB)
    static constexpr int NIB = NWV == 8 ? 2 : 3;                               // B transfers per wave and step (8 waves: 16 issued for 12 fragments, 4 duplicates)
    static constexpr size_t lds_bytes() { return (size_t)2 * PLANEQ * 16 + (size_t)BRING * BSTEPQ * 16; }
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
    else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
}

// split input, split output, no residual (conv1 of the hourglasses); oseg output planes per segment
template <int NWV_>
__global__ __launch_bounds__(NWV_ * 64, 2) void conv_march_s2_kernel(const ConvArgs p, const int oseg, const int nseg) {
    using G = MarchS2Geo<NWV_>;
    constexpr int PLANEQ = G::PLANEQ, NP = G::NP, NPI = G::NPI, NWV = G::NWV, NIB = G::NIB, TH = G::TH, TW = G::TW, LW = G::LW, NEV = G::NEV;
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    float4* const bring = smem + 2 * PLANEQ;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv & (TH - 1), wn = wv / TH;               // M-tile (output row of the column), N-tile (32 output channels)
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
    const unsigned bring_lds = smem_lds + 2u * PLANEQ * 16u;
    const char* const zsrc = reinterpret_cast<const char*>(g_march_s2_zeros);

    // B: a step's 12 fragments f = (kw * 2 + hl) * 2 + n, 64 lanes x 16 B each.  Wave w fetches fragment w and fragment 8 + (w & 3)
    // (waves 4..7 repeat 8..11: the same bytes to the same slots, so that every wave issues exactly NIB instructions per step).
    // packed weights: 16-byte unit ((ch * 27 + t) * 4 + hl * 2 + kg) * CoP + co, t = kd * 9 + kh * 3 + kw  (conv3d.hip pack_weights_f16x3)
    unsigned boff[NIB], bdst[NIB];
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
        const int f = NWV == 8 ? (i == 0 ? wv : 8 + (wv & 3)) : wv * 3 + i;
        const int kw = f >> 2, hl = (f >> 1) & 1, n = f & 1;
        boff[i] = (unsigned)((kw * 4 * CoP + hl * 2 * CoP + hh * CoP + n * 32 + col) * 16);
        bdst[i] = (unsigned)(f * 64 * 16);
    }
    auto dma_b = [&](const int slot, const int ch, const int kh, const int kd) {
        const char* base = reinterpret_cast<const char*>(p.w) + (size_t)((ch * 27 + kd * 9 + kh * 3) * 4 * CoP) * 16;
#pragma unroll
        for (int i = 0; i < NIB; ++i) dma(base + boff[i], bring_lds + (unsigned)(slot * G::BSTEPQ * 16) + bdst[i]);
    };

    // planes: piece i of this wave is DMA instruction n = i * NWV + wave (beyond NPI - 1: instruction NPI - 1 again).  LDS slot j = 64 n + lane
    // -> swizzled quad of voxel v = j >> 2 of the parity-planar image: row lh = v / 65, r = v % 65, column lw = 2 r (r < 33) or 2 (r - 33) + 1
    unsigned poff[NP];
    unsigned pvalid = 0u;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        int n = i * NWV + wv;
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
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int n = i * NWV + wv;
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

    // ---- one step: the 3 kw taps of (kh, kd) from plane buffer `cur`, B fragments from ring slot `slot`, 9 MFMAs
    auto taps = [&](f32x16& acc, const int cur, const int slot, const int kh) {
        const int vrow = (2 * wm + kh) * LW + col;
        float4 A[3][2], Bf[3][2];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int v = vrow + (kw == 1 ? NEV : (kw >> 1));
            const int qs = hh ^ ((v >> 2) & 3);
            A[kw][0] = smem[cur * PLANEQ + 4 * v + qs];
            A[kw][1] = smem[cur * PLANEQ + 4 * v + (qs ^ 2)];
            Bf[kw][0] = bring[slot * G::BSTEPQ + ((kw * 2 + 0) * 2 + wn) * 64 + lane];
            Bf[kw][1] = bring[slot * G::BSTEPQ + ((kw * 2 + 1) * 2 + wn) * 64 + lane];
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int term = 0; term < 3; ++term) {                      // small cross terms first (as the other forms)
                const f16x8 a = __builtin_bit_cast(f16x8, A[kw][term == 1 ? 1 : 0]);
                const f16x8 w = __builtin_bit_cast(f16x8, Bf[kw][term == 0 ? 1 : 0]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, w, acc, 0, 0, 0);
            }
    };

    // ---- pass / step sequence.  Planes pf .. pl are walked; pass = (plane pd, chunk c); an even plane has 3 steps (kh, kd = 1), an odd plane
    // 3 steps (kh, kd = 2) into the plane od = pd >> 1 it completes (if that plane belongs to this segment) + 3 steps (kh, kd = 0) into od + 1
    // (if that one does).  Global step t uses ring slot t % 3.
    const int pf = (2 * o0 - 1 > 0) ? 2 * o0 - 1 : 0;
    const int pl = (2 * o1 - 1 < p.Di - 1) ? 2 * o1 - 1 : p.Di - 1;
    auto steps_of = [&](const int pd) { const int od = pd >> 1; return (pd & 1) ? ((od >= o0 ? 3 : 0) + (od + 1 < o1 ? 3 : 0)) : 3; };
    auto kd_of = [&](const int pd, const int k) { return (pd & 1) ? ((k < 3 && (pd >> 1) >= o0) ? 2 : 0) : 1; };
    // look-ahead iterator of the B transfers (two steps ahead of the step being computed); past the end it stays on the last step
    int lpd = pf, lc = 0, lk = 0;
    bool ldone = false;
    auto issue_b = [&](const int slot) {
        dma_b(slot, lc, lk >= 3 ? lk - 3 : lk, kd_of(lpd, lk));
        if (!ldone) {
            if (++lk == steps_of(lpd)) {
                lk = 0;
                if (++lc == nch) { lc = 0; ++lpd; }
                if (lpd > pl) { ldone = true; lpd = pl; lc = nch - 1; lk = steps_of(pl) - 1; }
            }
        }
    };

    dma_plane(0, pf, 0);
    issue_b(0); issue_b(1);
    wait_vmcnt_c<0>();

    int slot = 0;                                             // ring slot of the step being computed
    int q = 0;
    for (int pd = pf; pd <= pl; ++pd) {
        const bool odd = pd & 1;
        const int od = pd >> 1;
        const bool do0 = !odd || od >= o0;
        const int ns = steps_of(pd);
        for (int c = 0; c < nch; ++c, ++q) {
            const int cur = q & 1;
            // plane-chunk of pass q + 1
            int npd = pd, nc = c + 1;
            if (nc == nch) { nc = 0; ++npd; }
            if (npd > pl) npd = -1;
            for (int k = 0; k < ns; ++k) {
                __syncthreads();                                  // every wave's share of this step's B (and, at k = 0, of this pass's plane) has landed;
                                                                  // the previous step's readers of ring slot (slot + 2) % 3 are done
                issue_b(slot >= 1 ? slot - 1 : 2);                // step t + 2 -> slot (t + 2) % 3
                if (k == 0) dma_plane(cur ^ 1, npd, nc);
                const int kh = k >= 3 ? k - 3 : k;
                if (odd && !(k < 3 && do0)) taps(acc1, cur, slot, kh);
                else taps(acc0, cur, slot, kh);
                // B of step t + 1 is home (issued a step ago); younger: this step's B transfer and -- during steps 0 and 1 -- the plane pieces
                if (k <= 1) wait_vmcnt_c<NIB + NP>(); else wait_vmcnt_c<NIB>();
                slot = slot == 2 ? 0 : slot + 1;
            }
            if (c + 1 == nch) {                                   // plane pd complete: does an output plane complete with it?
                const bool fin = odd || pd == p.Di - 1;           // (an even LAST plane: plane pd + 1 lies outside the tensor)
                if (fin && od >= o0 && od < o1) {
                    __syncthreads();                              // every wave is past its taps: buffer `cur` becomes the transpose tiles
                    epilogue(od, reinterpret_cast<float*>(smem + cur * PLANEQ) + wv * (32 * 36));
                }
                if (odd) {
                    acc0 = acc1;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
                }
            }
        }
    }
    wait_vmcnt_c<0>();                                            // (the zero pieces / spare B transfers of the last pass)
    if (p.out_meta) {
        __syncthreads();
        publish_amax(p.out_meta, am, amax_seen, reinterpret_cast<float*>(smem));
    }
}

}  // namespace osa
