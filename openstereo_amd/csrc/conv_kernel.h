// 3-D cost-aggregation convolutions for gfx950.
//
// im2col-free implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact f32):
//   M = output voxels (32 per MFMA tile), N = output channels (32 per tile),
//   K = taps x input channels.
// A workgroup owns a (TD x TH x TW) brick of "a-space" positions; the input brick it needs
// (halo included) is staged channels-last into LDS one 16-channel chunk at a time with a
// voxel stride of 20 floats (80 B: 16-byte aligned and conflict-free for ds_read_b128).  Every
// tap is then just a wave-uniform LDS offset: the A operand of 4 consecutive MFMAs is one
// ds_read_b128 per lane, the B operand one 16-byte load of the pre-packed weights
// ([chunk][tap][octet][half][Cout][4]) that all waves of all workgroups share through L2.
//
// One kernel covers every layer shape through a tap table:
//   out[a*os + oo] = sum_t  in[a*is + delta_t] . W_t
//   stride-1/2 conv : os=1, is=stride, delta = k*dil - pad
//   1x1x1           : one tap
//   transposed conv (stride 2): 8 output-parity classes, os=2, oo=parity, is=1, only the
//                     taps that hit real (non-inserted) inputs -> no zero insertion, no wasted MACs
// Epilogue (fused): folded eval-mode BatchNorm (scale/shift), residual add, ReLU/LeakyReLU.
//
// This header holds the kernel template and its device helpers; conv3d.hip instantiates the tile configurations of
// the stage -> barrier -> taps form, conv_pipe.hip the persistent LDS-DMA pipelined form (PIPE = 1).
#pragma once
#include "osa_common.h"
#include <cstdlib>
#include <cstring>

#include <type_traits>

namespace osa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Arithmetic modes of the implicit GEMM:
//   PREC_F32   v_mfma_f32_32x32x2_f32, exact fp32 products (157 TFLOP/s peak)
//   PREC_F16X3 every fp32 operand x is split x = hi + lo (two fp16, 22 significant bits) and
//              A.B ~= Ahi.Bhi + Ahi.Blo + Alo.Bhi on v_mfma_f32_32x32x16_f16 with fp32 accumulation:
//              3 MFMAs of 32 cycles per K=16 instead of 8 of 64 -> 5.3x less matrix-pipe time at
//              fp32-class accuracy (dropped term Alo.Blo ~ 2^-22 relative).  Weights are pre-scaled
//              by a power of two into the fp16 normal range (undone exactly in the epilogue);
//              activations are saturated to +-65504 when split.
//   PREC_F16   the autocast arithmetic of the reference's AMP configs (cfgs/stereobase, cfgs/lightstereo, cfgs/igev *_amp:
//              trainer_template.py:211,281 wrap every forward in torch.autocast): operands rounded to fp16 (nearest even), ONE
//              v_mfma_f32_32x32x16_f16 per product, fp32 accumulation, fp32 epilogue.  A staged chunk is 32 input channels
//              ([ch 0-15 | ch 16-31], the same 64-byte LDS image as an f16x3 chunk of 16), activations may live in HBM as fp16
//              NDHWC tensors (OSA_IN_F16 / OSA_OUT_F16 / OSA_RES_F16 = the *_SPLIT flag bits), no operand scaling: values beyond
//              65504 become inf exactly as they do under autocast.
enum { PREC_F32 = 0, PREC_F16X3 = 1, PREC_F16 = 2 };

constexpr int CC = 16;        // input channels staged per pass (packed-weight format constant)
constexpr int VS = CC + 4;    // LDS voxel stride in floats
constexpr int JO = CC / 8;    // k-octets per chunk
constexpr int MAX_TAPS = 64;   // 3x3x3 = 27; a fused k=4 transposed conv carries all 64 taps

struct ConvArgs {
    const float* x; const float4* w; const float* scale; const float* shift; const float* res; float* y;
    const float* gate; int gCs;   // optional sigmoid channel gate, NHWC logits [B][Ho][Wo][gCs]
    int B, Di, Hi, Wi, Ci, xCs;
    int Do, Ho, Wo, Co, yCs, rCs;
    int Ad, Ah, Aw;               // a-space extent of this launch
    int isd, ish, isw;            // input step per a (per dim)
    int os, ood, ooh, oow;        // output position = a*os + oo
    int T;                        // taps
    int cls_end[8];               // fused transposed conv: taps [cls_end[c-1], cls_end[c]) belong to output-parity class c
    int dmin, hmin, wmin;         // min delta per dim
    int LD, LH, LW;               // LDS brick dims (voxels)
    int RowQ, PlaneQ;             // LDS float4s per brick row (padded) / per d-plane (16-byte units keep ds_read_b128)
    int dbg;                      // debug switch (OSA_DBG): 1 = skip staging (timing experiments only)
    int tilesD, tilesH, tilesW;
    int nchunks, CoP;
    int cps;                      // channel chunks staged per pass (LDS holds cps bricks back to back)
    int act; float slope;
    float oscale;                 // f16x3: 1 / (weight pre-scale), exact power of two; 1 for f32
    const float* wscale_dev;      // f16x3, optional: {wscale, 1 / wscale} in device memory (osa_*_pack_*_auto); [1] then replaces oscale
    int VQ;                       // LDS voxel stride in 16-byte slots: 4 (compact) or 5 (padded), see finish_geometry
    // fused 1x1x1 "redir" branch of a transposed conv (GwcNet hourglass: relu(conv6(c5) + redir1(x))):
    // rx = NDHWC tensor at OUTPUT resolution (<= 32 channels), rw = its packed 1x1x1 weights (same packing,
    // T = 1), rscale / rshift = its folded BN, roscale = its f16x3 output scale.  NULL rx = not fused.
    const float* rx; const float4* rw; const float* rscale; const float* rshift; float roscale; int rxCs, rCi;
    unsigned magicW, magicHW;     // ceil(2^32/LW), ceil(2^32/(LH*LW)) : exact for operands < 2^16
    unsigned magicH;              // ceil(2^32/LH)
    int dma;                      // 1: split input + compact LDS image -> stage rows by LDS-DMA (global_load_lds_dwordx4)
    int ringQ;                    // BL kernels: first float4 slot of the B-operand ring inside the workgroup's LDS (above bricks and epilogue tiles)
    int stag_ticks, stag_n, stag_cus;   // start-up stagger: workgroups with linear id < stag_n wait (id / stag_cus) * stag_ticks 10-ns ticks (0: off)
    // f16x3 range tracking (see osa_f16x3_ranges in the header); every pointer may be NULL.  A "meta" block is
    // OSA_META_FLOATS floats of device memory per tensor: running max |value| in 8 slots (osa_common.h),
    // [1] = power-of-two scale of the stored hi/lo halves when the tensor is a split tensor.
    const float* in_meta; const float* res_meta; const float* rx_meta; float* out_meta;
    const float* coef;            // [0] max_co |bn scale| * sum|w_co|, [1] max_co |bn shift|   (output bound of this layer)
    const float* rcoef;           // same for the fused redir layer
    int toff[MAX_TAPS];           // LDS offset of every tap in float4 units (host computed -> scalar loads)
    signed char td[MAX_TAPS], th[MAX_TAPS], tw[MAX_TAPS];
};

// Stage CC channels [c0, c0+CC) of the input brick into LDS (zero outside the tensor / beyond Ci).
// Loads are issued U at a time before the first LDS write so a thread keeps U 16-byte loads in flight.

// (split_f16, pow2_scale, mul4: osa_common.h -- the volume builder writes split tensors too)

// Thread-linear item order (every lane busy on every load).  A row-wise variant with wave-uniform
// row arithmetic (2x fewer VALU instructions) was measured slower overall on MI355X: rows of 10-18
// voxels leave 40-45 % of the lanes idle, which costs more than the index arithmetic saves.
// NCL consecutive 16-channel chunks are staged in one pass (bricks back to back in LDS, brickQ apart):
// the index arithmetic of an item is shared by its NCL loads, and the two 64-byte halves of a voxel's
// 128-byte line are requested together.
// inverse of split_f16 for one channel quad: x = float(hi) + float(lo)
__device__ __forceinline__ float4 join_f16(const uint2 hi, const uint2 lo) {
    const f16x4 h = __builtin_bit_cast(f16x4, hi), l = __builtin_bit_cast(f16x4, lo);
    return make_float4((float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]);
}

// "Split" activation tensors (OSA_IN_SPLIT / OSA_OUT_SPLIT / OSA_RES_SPLIT / OSA_REDIR_SPLIT, f16x3 mode only):
// the same bytes per voxel as fp32 NDHWC, but every 16-channel chunk holds [16 x fp16 hi | 16 x fp16 lo]
// -- exactly the LDS image of a staged chunk.  Element offset (in floats) of the hi / lo halves of the
// channel quad starting at channel c (c % 4 == 0):
__device__ __forceinline__ int split_off_hi(int c) { return (c >> 4) * 16 + ((c & 15) >> 2) * 2; }
__device__ __forceinline__ int split_off_lo(int c) { return split_off_hi(c) + 8; }

#ifndef OSA_STAGE_U
#define OSA_STAGE_U 4
#endif
template <int NTHR, int PREC, int NCL, int SU = OSA_STAGE_U>
__device__ __forceinline__ void stage_brick(const ConvArgs& p, float4* smem, int brickQ, int b, int c0,
                                            int g0d, int g0h, int g0w, int tid, float s_in = 1.f) {
    constexpr int U = (NCL == 1) ? SU : SU / 2;
    const int total = p.LD * p.LH * p.LW * (CC / 4);
    const int LHW = p.LH * p.LW;
    // wave-uniform 64-bit base of batch item b / chunk c0; per-lane offsets are 32-bit (host checks < 2^31 elements)
    const float* xb = p.x + (size_t)b * p.Di * p.Hi * p.Wi * p.xCs + c0;
    for (int base = tid; base < total; base += NTHR * U) {
        float4 v[U][NCL];
        int lo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int it = base + u * NTHR;
            lo[u] = -1;
#pragma unroll
            for (int cl = 0; cl < NCL; ++cl) v[u][cl] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (it < total) {
                const int c4 = it & 3, vx = it >> 2;
                const int ld = __umulhi((unsigned)vx, p.magicHW);
                const int r = vx - ld * LHW;
                const int lh = __umulhi((unsigned)r, p.magicW);
                const int lw = r - lh * p.LW;
                const int gd = g0d + ld, gh = g0h + lh, gw = g0w + lw;
                lo[u] = ld * p.PlaneQ + lh * p.RowQ + lw * p.VQ + c4;
                if (((unsigned)gd < (unsigned)p.Di) && ((unsigned)gh < (unsigned)p.Hi) && ((unsigned)gw < (unsigned)p.Wi)) {
                    const float* src = xb + ((gd * p.Hi + gh) * p.Wi + gw) * p.xCs + c4 * 4;
#pragma unroll
                    for (int cl = 0; cl < NCL; ++cl)
                        if (c0 + cl * CC + c4 * 4 < p.Ci) v[u][cl] = *reinterpret_cast<const float4*>(src + cl * CC);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (lo[u] >= 0) {
#pragma unroll
                for (int cl = 0; cl < NCL; ++cl) {
                    float4* dst = smem + cl * brickQ;
                    if constexpr (PREC == PREC_F32 || PREC == PREC_F16) {
                        dst[lo[u]] = v[u][cl];          // (f16 mode: this function stages fp16 tensors -- a 16-byte quad = 8 channels; fp32 inputs: stage_brick_cvt16)
                    } else {
                        if (p.act & OSA_IN_SPLIT) { dst[lo[u]] = v[u][cl]; continue; }   // already [hi | lo] in HBM
                        // voxel image: [16 x fp16 hi | 16 x fp16 lo]; this quad's 4 channels -> 8 B each
                        static_assert(NTHR % 4 == 0, "channel quad of an item must not depend on u");
                        const int c4 = base & 3;        // == (base + u*NTHR) & 3
                        uint2 h2, l2;
                        split_f16(mul4(v[u][cl], s_in), h2, l2);
                        uint2* s2 = reinterpret_cast<uint2*>(dst);
                        const int vbase = (lo[u] - c4) * 2;                 // voxel start in 8-byte units
                        s2[vbase + c4] = h2;
                        s2[vbase + 4 + c4] = l2;
                    }
                }
            }
    }
}

// f16 mode, fp32 input tensor: chunk `c0 / CC` covers the 32 channels [2 c0, 2 c0 + 32); an item is (voxel, 8-channel group): two
// float4 loads, rounded to nearest-even fp16 (what `.half()` / autocast's cast does), one 16-byte LDS store.
__device__ __forceinline__ float4 cvt8_f16(const float4 a, const float4 b) {
    const f16x8 h = {(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w, (_Float16)b.x, (_Float16)b.y, (_Float16)b.z, (_Float16)b.w};
    return __builtin_bit_cast(float4, h);
}
template <int NTHR, int NCL, int SU = OSA_STAGE_U>
__device__ __forceinline__ void stage_brick_cvt16(const ConvArgs& p, float4* smem, int brickQ, int b, int c0,
                                                  int g0d, int g0h, int g0w, int tid) {
    constexpr int U = (NCL == 1) ? SU / 2 : SU / 4;
    static_assert(U >= 1, "staging depth");
    const int total = p.LD * p.LH * p.LW * (CC / 4);
    const int LHW = p.LH * p.LW;
    const float* xb = p.x + (size_t)b * p.Di * p.Hi * p.Wi * p.xCs + 2 * c0;
    for (int base = tid; base < total; base += NTHR * U) {
        float4 v[U][NCL][2];
        int lo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int it = base + u * NTHR;
            lo[u] = -1;
#pragma unroll
            for (int cl = 0; cl < NCL; ++cl) v[u][cl][0] = v[u][cl][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (it < total) {
                const int c4 = it & 3, vx = it >> 2;
                const int ld = __umulhi((unsigned)vx, p.magicHW);
                const int r = vx - ld * LHW;
                const int lh = __umulhi((unsigned)r, p.magicW);
                const int lw = r - lh * p.LW;
                const int gd = g0d + ld, gh = g0h + lh, gw = g0w + lw;
                lo[u] = ld * p.PlaneQ + lh * p.RowQ + lw * p.VQ + c4;
                if (((unsigned)gd < (unsigned)p.Di) && ((unsigned)gh < (unsigned)p.Hi) && ((unsigned)gw < (unsigned)p.Wi)) {
                    const float* src = xb + ((gd * p.Hi + gh) * p.Wi + gw) * p.xCs + c4 * 8;
#pragma unroll
                    for (int cl = 0; cl < NCL; ++cl) {
                        const int c = 2 * c0 + cl * 2 * CC + c4 * 8;
                        if (c < p.Ci) v[u][cl][0] = *reinterpret_cast<const float4*>(src + cl * 2 * CC);
                        if (c + 4 < p.Ci) v[u][cl][1] = *reinterpret_cast<const float4*>(src + cl * 2 * CC + 4);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (lo[u] >= 0) {
#pragma unroll
                for (int cl = 0; cl < NCL; ++cl) (smem + cl * brickQ)[lo[u]] = cvt8_f16(v[u][cl][0], v[u][cl][1]);
            }
    }
}

// Staging of a SPLIT input chunk (already [hi | lo] in HBM) into the COMPACT LDS image by LDS-DMA: one
// global_load_lds_dwordx4 per (d, h) row of the brick -- lane = (w, 16-byte quad), LDS destination = row
// base + lane * 16 (exactly the compact row), global source per lane.  A wave takes whole rows, so the
// row arithmetic is scalar; no VGPR round trip, no ds_write.  Lanes / rows outside the tensor are zero
// filled with ordinary LDS stores.  Requires LW * 4 <= 64 (one row per instruction).
template <int NTHR>
__device__ __forceinline__ void stage_brick_dma(const ConvArgs& p, float4* smem, int b, int c0,
                                                int g0d, int g0h, int g0w, int tid) {
    constexpr int NWV = NTHR / 64;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rows = p.LD * p.LH;
    const int lw = lane >> 2, c4 = lane & 3;
    const bool lane_in = lw < p.LW;
    const int gw = g0w + lw;
    const bool w_ok = lane_in && ((unsigned)gw < (unsigned)p.Wi) && (c0 + c4 * 4 < p.Ci);
    const int goff = gw * p.xCs + c0 + c4 * 4;                 // floats from the start of the (d, h) row
    const float* xb = p.x + (size_t)b * p.Di * p.Hi * p.Wi * p.xCs;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = wave; r < rows; r += NWV) {                   // wave-uniform
        const int ld = (p.LH == 1) ? r : (int)__umulhi((unsigned)r, p.magicH), lh = r - ld * p.LH;
        const int gd = g0d + ld, gh = g0h + lh;
        float4* row = smem + ld * p.PlaneQ + lh * p.RowQ;      // wave-uniform LDS row base
        const bool row_ok = ((unsigned)gd < (unsigned)p.Di) && ((unsigned)gh < (unsigned)p.Hi);
        if (row_ok) {
            const float* rowp = xb + ((size_t)gd * p.Hi + gh) * (size_t)p.Wi * p.xCs;
            if (w_ok)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rowp + goff),
                                                 (__attribute__((address_space(3))) void*)row, 16, 0, 0);
            else if (lane_in) row[lane] = zero;
        } else if (lane_in) row[lane] = zero;
    }
}

// s_waitcnt vmcnt(n): the immediates of the LDS-DMA protocols are instruction counts; n is a constant after unrolling, the switch folds
__device__ __forceinline__ void wait_vmcnt(const int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    }
}

// In-kernel timeline (build with -DOSA_EXPERIMENTS -DOSA_TRACE_ON, run with OSA_DBG & 256): wave `w` of every 97th workgroup stamps the 100 MHz wall clock at its
// phase boundaries into g_trace[slot][wave][event]; tools/trace_conv.py reads it back (osa_debug_trace_read).
#ifdef OSA_TRACE_ON
constexpr int TRACE_SLOTS = 32, TRACE_WAVES = 4, TRACE_EVENTS = 40;
static __device__ unsigned long long g_trace[TRACE_SLOTS * TRACE_WAVES * TRACE_EVENTS];
#define OSA_TRACE(ev) do { if ((p.dbg & 256) && (tid & 63) == 0 && blockIdx.y == 0 && blockIdx.x % 97 == 0 && blockIdx.x / 97 < TRACE_SLOTS && (tid >> 6) < TRACE_WAVES && (ev) < TRACE_EVENTS) \
    g_trace[((blockIdx.x / 97) * TRACE_WAVES + (tid >> 6)) * TRACE_EVENTS + (ev)] = wall_clock64(); } while (0)
#else
#define OSA_TRACE(ev) do {} while (0)
#endif

// CFG: MT m-tiles x NT n-tiles per wave, WM x WN waves, brick TD x TH x TW (TD derived)
// NCLS = 1: ordinary (strided / dilated / 1x1x1) convolution.
// NCLS = 8: stride-2 transposed convolution, all 8 output-parity classes in one launch: the input
//           brick is staged once, every class has its own accumulator set and its own run of taps
//           (class-major tap order, one linear B stream), outputs go to o = 2a + parity.
template <int PREC, int NCLS, int TU, int MT, int NT, int WM, int WN, int TH, int TW, int REDIR = 0, int OUTS = 0, int PIPE = 0, int KS = 1, int BL = 0>
// BL = 1 (r4): the B (weight) operands of a tap step reach the waves through an LDS ring filled by LDS-DMA -- ONE fetch per workgroup and
// step instead of one per wave (see the BL block below).  Same products in the same order: bit-identical to BL = 0.
// OUTS = 1: the output is a split tensor (OSA_OUT_SPLIT) -- separate instantiation: a lane finalises 8
// channels of 2 voxels (16-byte hi and lo stores) instead of 4 channels of 4 voxels.
// Registers: the fused transposed convs need 2 waves per SIMD; the 256-voxel x 32-channel tiles
// (MT = 2, NT = 1: the dominant 32 -> 32 layers) are held to 128 registers so that 4 workgroups
// share a CU now that their compact LDS brick is 39 KB (measured +7 % on those layers; the same
// limit costs the 64-channel tiles 5 %, so they keep the default).
// KS > 1: split-K inside the workgroup, for small maps whose few workgroups would each walk the whole K loop alone (the 1/8 and 1/16
// GRU levels of the update block: 16-24 input chunks, 16-255 workgroups).  The workgroup has KS groups of WM*WN waves; group g
// owns the contiguous run of input chunks [g*cpg, (g+1)*cpg) -- its B stream is a contiguous piece of the ordinary packed
// buffer -- stages its own chunk per pass and accumulates its own partial tiles; the partials meet in LDS and group 0 runs the
// epilogue.  Host: nchunks % KS == 0, NCLS == 1.
// PIPE = 1: persistent workgroups walking a list of bricks, the input brick double-buffered in LDS and fed by LDS-DMA
// (buffer_load ... lds) from a loader wave while the compute waves run the taps of the previous chunk, so staging
// never waits (see the PIPE block below).  Split (OSA_IN_SPLIT) inputs, compact LDS image, unit input step; 2 workgroups
// of NW + 1 waves per CU.
// Waves per SIMD every instantiation is compiled for (its accumulators set the scale: 16 registers per 32x32 tile).  Stated
// explicitly: left to itself the compiler spends registers on scheduling freedom (the straight-line fast epilogue gives it
// plenty) and silently drops a wave per SIMD, which costs more than any schedule gains.
// Stride-2 tiles (64-voxel bricks 2x4x8 of two M-tile waves: configurations 5, 6, 15): the strided input brick is 5x9x17 voxels = 63 KB per
// chunk, so two workgroups share a CU whatever the registers allow -- the launch is bound by the latency of its staging loads (12 per
// thread and chunk at 4 in flight = 3 HBM round trips per pass).  They are compiled for 2 waves per SIMD and keep a whole chunk's loads in
// flight (OSA_S2_U).
#ifndef OSA_S2_U
#define OSA_S2_U 12
#endif
#define OSA_S2TILE (NCLS == 1 && WM == 2 && TH == 4 && TW == 8 && !PIPE && KS == 1)
#define OSA_WAVES_PER_SIMD (PIPE ? 3 : ((NCLS == 8) ? 2 : ((NCLS >= 4) ? 3 : ((OSA_S2TILE && OSA_S2_U > 4) ? 2 : ((MT * NT == 1) ? 4 : ((MT * NT == 2) ? ((MT == 2) ? 4 : 3) : 2))))))
#define OSA_MIN_BLOCKS OSA_WAVES_PER_SIMD          // HIP: the second __launch_bounds__ argument is waves per SIMD (execution unit)
__global__ __launch_bounds__(WM * WN * KS * 64 + (PIPE ? 64 : 0), (KS > 1) ? (WM * WN * KS / 4) : OSA_MIN_BLOCKS) void conv_mfma_kernel(const ConvArgs p) {
    static_assert(KS == 1 || (NCLS == 1 && !PIPE && !REDIR), "split-K: plain convolutions");
    static_assert(!BL || (KS == 1 && !PIPE && TU == 1), "B ring: one B stream per workgroup, per-tap steps");
    constexpr int NW = WM * WN;
    constexpr int TD = WM * MT * 32 / (TH * TW);
    static_assert(TD * TH * TW == WM * MT * 32, "brick must hold WM*MT*32 voxels");
    // XT: "transposed accumulators" -- an r3 experiment, compiled out by default (-DOSA_XT=1 builds it: tools/build_variant.sh).  The plain
    // f16x3 convolutions that write split tensors issue their MFMAs with the operands swapped (weights as A, activations as B: the same
    // products summed in the same order -- verified bit-identical on 7 layer shapes, tools/diag_xt.py), so a lane's 16 accumulators of a
    // 32 x 32 tile are 16 CHANNELS of ONE voxel -- voxel lane & 31, channels (r & 3) + 8 (r >> 2) + 4 hh -- instead of 16 voxels of one
    // channel.  The epilogue then needs no transpose through LDS (16 ds_write_b32 + 4 ds_read_b128 and two LDS round trips per tile):
    // one v_permlane32_swap per value pair gives every lane 8 consecutive channels, i.e. the 16-byte hi and lo rows of a split tensor.
    // MEASURED: the whole model is 14 % SLOWER (155.1 vs 179.9 pairs/s, profiles/round3/ab_xt_epilogue.txt).  A lane then stores 16 B of
    // a voxel of its own, so a wave's store (and residual load) touches 32 different 128-byte lines instead of 8 whole ones: the
    // transpose through LDS is what buys coalesced rows, and it is the cheaper of the two.
#ifndef OSA_XT
#define OSA_XT 0
#endif
    constexpr bool XT = OSA_XT && (PREC == PREC_F16X3) && OUTS && NCLS == 1 && !REDIR && !PIPE && KS == 1;
    static_assert((TW & (TW - 1)) == 0 && (TH & (TH - 1)) == 0, "TH/TW powers of two");
    extern __shared__ __attribute__((aligned(16))) float4 smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = (KS > 1) ? wave_all / NW : 0;                 // K group of this wave (split-K), 0 otherwise
    const int wave = (KS > 1) ? wave_all - kg * NW : wave_all;   // wave inside its group
    const int wm = wave / WN, wn = wave % WN;
    const int col = lane & 31, hh = lane >> 5;

    // current brick: batch item, first a-space position, first input position (fixed per workgroup unless PIPE)
    int b, a0d, a0h, a0w, g0d, g0h, g0w;
    auto set_brick = [&](int b_, int tdi, int thi, int twi) {
        b = b_;
        a0d = tdi * TD; a0h = thi * TH; a0w = twi * TW;
        g0d = a0d * p.isd + p.dmin; g0h = a0h * p.ish + p.hmin; g0w = a0w * p.isw + p.wmin;
    };
    // PIPE brick order: d fastest, then 4-row strips of h (h % 4, then w, then h / 4), then batch item -- bricks that
    // run at the same time on one XCD (xcd_remap: 64 consecutive ids) share their halos in d, w and h through its L2
    auto decode_item = [&](int id, int& b_, int& tdi, int& thi, int& twi) {
        tdi = id % p.tilesD;
        const int colid = id / p.tilesD, cpb = p.tilesH * p.tilesW;
        b_ = colid / cpb;
        const int c = colid - b_ * cpb, fullrows = p.tilesH & ~3, full = fullrows * p.tilesW;
        if (c < full) { const int hb = c / (4 * p.tilesW), r = c - hb * 4 * p.tilesW; twi = r >> 2; thi = hb * 4 + (r & 3); }
        else { const int r = c - full, rem = p.tilesH - fullrows; twi = r / rem; thi = fullrows + r - twi * rem; }
    };
    if constexpr (!PIPE) {
        unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
        const int twi = bid % p.tilesW; bid /= p.tilesW;
        const int thi = bid % p.tilesH; bid /= p.tilesH;
        const int tdi = bid % p.tilesD;
        set_brick((int)(bid / p.tilesD), tdi, thi, twi);
    }
    const int n0 = blockIdx.y * (WN * NT * 32);

    // ---- f16x3 operand ranges: power-of-two scales of the input / residual / redir operands and of a split
    // output (all wave-uniform scalar loads of a few device words; all 1 when no range block was passed)
    float s_in = 1.f, s_res_inv = 1.f, s_rx = 1.f, s_out = 1.f;
    if constexpr (PREC == PREC_F16X3) {
        if (p.in_meta) s_in = (p.act & OSA_IN_SPLIT) ? p.in_meta[1] : pow2_scale(amax_read(p.in_meta));
        if (p.res && p.res_meta && (p.act & OSA_RES_SPLIT)) s_res_inv = 1.0f / p.res_meta[1];
        if (REDIR && p.rx_meta) s_rx = (p.act & OSA_REDIR_SPLIT) ? p.rx_meta[1] : pow2_scale(amax_read(p.rx_meta));
        if (OUTS && p.coef && p.in_meta) {
            // rigorous bound of |output|: sum|w| * max|x| * |bn scale| + |bn shift| (+ residual / redir branch);
            // activations only shrink it (sigmoid / tanh: 1)
            float bound = p.coef[0] * amax_read(p.in_meta) + p.coef[1];
            if (p.res && p.res_meta) bound += amax_read(p.res_meta);
            if (REDIR && p.rcoef && p.rx_meta) bound += p.rcoef[0] * amax_read(p.rx_meta) + p.rcoef[1];
            const int ak = p.act & 15;
            if (ak == OSA_ACT_SIGMOID || ak == OSA_ACT_TANH) bound = 1.f;
            s_out = pow2_scale(bound * 1.0625f);
        }
        if (OUTS && p.out_meta && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) p.out_meta[1] = s_out;
    }
    const float osc = (p.wscale_dev ? p.wscale_dev[1] : p.oscale) * (1.0f / s_in);      // undoes the weight pre-scale and the input scale (exact)
    const float rosc = p.roscale * (1.0f / s_rx);
    float am = 0.f;                                  // running max |output| of this lane (unscaled values)

    int abase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int q = (wm * MT + m) * 32 + col;
        const int tw_ = q % TW, th_ = (q / TW) % TH, td_ = q / (TW * TH);
        abase[m] = (td_ * p.isd) * p.PlaneQ + (th_ * p.ish) * p.RowQ + (tw_ * p.isw) * p.VQ + hh;
    }

    f32x16 acc[NCLS][MT][NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int c = 0; c < NCLS; ++c)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[c][m][n][r] = 0.f;
    };
    zero_acc();

    // One linear stream of B operands: [chunk][tap][octet] "tap steps" of JO*2*CoP float4 each.
    // Copy-free software pipeline: two static register sets (0/1) ping-pong.  Each half-iteration
    // first requests the A (LDS) and B (global/L2) operands of the NEXT group of up to TU taps into
    // the other set, then issues the MFMAs of the current set (sched_barrier pins that order).  A
    // half-iteration with cnt == 0 only prefetches, so every tap run (chunk, parity class) ends with
    // "set 0 holds the next group" and no register rotation is ever needed.  The packed buffer
    // carries a few tap steps of slack for the last prefetch.
    const size_t bstep = (size_t)2 * p.CoP;          // float4s per octet
    const size_t tstep = (p.dbg & 4) ? 0 : (size_t)JO * bstep;   // float4s per tap (dbg 4: stationary B stream, timing only)
    const int cpg = (KS > 1) ? p.nchunks / KS : p.nchunks;     // chunks per K group
    const float4* const wp0 = p.w + (size_t)hh * p.CoP + n0 + wn * (NT * 32) + col + (size_t)kg * cpg * p.T * tstep;
    const float4* wp = wp0;
    constexpr bool RING3 = (TU == 3);     // TU == 3 selects the 3-deep B ring (taps % 3 == 0, NCLS == 1)
    constexpr int TUA = RING3 ? 1 : TU;
    static_assert(!RING3 || NCLS == 1, "the B ring needs tap runs that are multiples of 3");
    float4 B0[TUA][JO][NT], B1[TUA][JO][NT], B2[TUA][JO][NT], A0[TUA][JO][MT], A1[TUA][JO][MT];
    auto init_b = [&]() {                 // B operands of the first tap(s) of the stream (start of a brick)
        wp = wp0;
#pragma unroll
        for (int u = 0; u < TUA; ++u)
#pragma unroll
            for (int j = 0; j < JO; ++j)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    B0[u][j][n] = wp[u * tstep + j * bstep + n * 32];
                    if constexpr (RING3) B1[u][j][n] = wp[tstep + j * bstep + n * 32];
#ifdef OSA_DBG_NOB
                    B2[u][j][n] = B0[u][j][n];
                    if constexpr (!RING3) B1[u][j][n] = B0[u][j][n];
#endif
                }
    };
    if constexpr (!BL) init_b();

    // ---- BL = 1: B operands through an LDS ring (r4).  The r4 ablations (profiles/round4/brick_kernel_without_B_loads_and_amax_retest.txt,
    // march_v1_ablation_and_f16_tests.txt) found the tap loop bound by its weight stream: every wave pulls its own 1 KB fragments through
    // the vector-memory path, which delivers ~37 B/clk/CU of 16-byte-per-lane loads whether they hit L1 or not -- a 64 x 32 wave (6 MFMAs
    // = 192 cycles per tap) asks for 42 B/clk/CU.  Here a tap step's FR = JO * WN * NT fragments are fetched ONCE per workgroup by
    // LDS-DMA (global_load_lds_dwordx4, every wave issues its share: HP 512-byte half fragments) into slot (step & 3) of a 4-slot ring,
    // three steps ahead of their use; every wave reads its operands from there with lane-contiguous (conflict-free) ds_read_b128 one
    // step ahead, into the same two register sets the A operands ping-pong through.  ds_read_b128 runs at 256 B/clk/CU, so the ring's
    // reads ride on top of the A fragments without touching the vector-memory path.  Protocol per consumed step g (conv_march.h's):
    //   s_barrier            publishes DMA(g + 1): every wave waited for its share at the end of step g - 1
    //   issue DMA(g + 3)     into slot (g + 3) & 3 = (g - 1) & 3, whose last readers finished with step g - 1 (everyone is past the barrier)
    //   ds_read A(tap + 1), B(g + 1) -> other register set;  MFMAs of step g
    //   s_waitcnt vmcnt(NIW) DMA(g + 2) is home (vmcnt retires in order), DMA(g + 3) stays in flight
    // The B stream is the ordinary packed buffer walked linearly ([chunk][tap]: g = chunk * T + tap), so the ring runs on across chunk
    // and parity-class boundaries; the 3 steps it runs past the end fall into the buffer's slack (slack_floats).  Inline asm for the
    // DMA (through the builtin this compiler waits vmcnt(0) right after the issue); the saddr form keeps the 64-bit base scalar.
    constexpr int BL_NBT = WN * NT;                          // N tiles of the workgroup
    constexpr int BL_FR = JO * BL_NBT;                       // 1 KB fragments per tap step
    constexpr int BL_HP = BL ? (2 * BL_FR) / NW : 2;         // 512-byte half fragments per wave and step
    constexpr int BL_NIW = (BL_HP + 1) / 2;                  // DMA instructions per wave and step
    constexpr int BL_SLOTQ = BL_FR * 64;                     // float4 slots per ring step
    static_assert(!BL || ((2 * BL_FR) % NW == 0 && BL_HP >= 1 && BL_NIW <= 3), "B ring: the waves split a step's fragments evenly");
    [[maybe_unused]] const float4* const bring = smem + p.ringQ;
    [[maybe_unused]] unsigned bl_voff[BL_NIW], bl_loff[BL_NIW];
    [[maybe_unused]] const float4* bl_next = p.w + n0;       // wave-uniform: start of the next step to fetch (this workgroup's N columns)
    [[maybe_unused]] int bl_gi = 0, bl_gs = 0;               // steps issued / steps consumed
    [[maybe_unused]] const unsigned bl_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)p.ringQ * 16u;
    if constexpr (BL) {
#pragma unroll
        for (int i = 0; i < BL_NIW; ++i) {
            // fragment f = j * NBT + nt of the step; HP == 1: wave w fetches half (w & 1) of fragment w >> 1 with its lower 32 lanes
            const int f = (BL_HP == 1) ? (wave >> 1) : (wave * BL_NIW + i);
            const int khalf = (BL_HP == 1) ? (wave & 1) : hh;
            bl_voff[i] = (unsigned)(((f / BL_NBT) * (2 * p.CoP) + khalf * p.CoP + (f % BL_NBT) * 32 + col) * 16);
            bl_loff[i] = (unsigned)((f * 64 + ((BL_HP == 1) ? (wave & 1) * 32 : 0)) * 16);
        }
    }
    auto bl_issue = [&]() {
        if constexpr (BL) {
            const unsigned slot = bl_lds + (unsigned)(bl_gi & 3) * (BL_SLOTQ * 16u);
#pragma unroll
            for (int i = 0; i < BL_NIW; ++i) {
                const unsigned m0v = __builtin_amdgcn_readfirstlane(slot + bl_loff[i]);
                if (BL_HP > 1 || lane < 32) {
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(bl_voff[i]), "s"(bl_next), "s"(m0v) : "memory");
                }
            }
            bl_next += (size_t)JO * 2 * p.CoP;
            ++bl_gi;
        }
    };
    // this wave's B operands of ring step g
    auto bl_read = [&](float4 (&Bn)[1][JO][NT], const int g) {
        const float4* const sl = bring + (g & 3) * BL_SLOTQ + lane;
#pragma unroll
        for (int j = 0; j < JO; ++j)
#pragma unroll
            for (int n = 0; n < NT; ++n) Bn[0][j][n] = sl[(j * BL_NBT + wn * NT + n) * 64];
    };
    if constexpr (BL) { bl_issue(); bl_issue(); bl_issue(); }     // steps 0..2 land while the first brick is staged

    const int brickQ = p.LD * p.PlaneQ;          // float4s per staged chunk
    const int Tm1 = p.T - 1;
    const float4* sm = smem;
    int t = 0;
    // tap-offset table in one VGPR (lane i holds toff[i], T <= 64): v_readlane instead of a scalar
    // memory load + lgkmcnt wait in front of every tap's LDS reads
    const int toff_v = p.toff[(lane < p.T) ? lane : 0];

    // prefetch group starting at flat tap `tn` (B: `skip` tap steps ahead of wp) into (An, Bn)
    auto prefetch = [&](float4 (&An)[TUA][JO][MT], float4 (&Bn)[TUA][JO][NT], int tn, int skip) {
#ifndef OSA_DBG_NOB          // (-DOSA_DBG_NOB: timing-only build without the per-tap B loads -- what the weight stream through the vector-memory path costs)
#pragma unroll
        for (int u = 0; u < TUA; ++u)
#pragma unroll
            for (int j = 0; j < JO; ++j)
#pragma unroll
                for (int n = 0; n < NT; ++n) Bn[u][j][n] = wp[(size_t)(skip + u) * tstep + j * bstep + n * 32];
#endif
#pragma unroll
        for (int u = 0; u < TUA; ++u) {
            const int ti = tn + u;
            const int to = __builtin_amdgcn_readlane(toff_v, (ti < Tm1) ? ti : Tm1);
#pragma unroll
            for (int j = 0; j < JO; ++j)
#pragma unroll
                for (int m = 0; m < MT; ++m) An[u][j][m] = sm[abase[m] + to + j * 2];
        }
    };
    // MFMAs of the first `cnt` taps of (Ac, Bc) into accumulator set ac
    auto compute = [&](const float4 (&Ac)[TUA][JO][MT], const float4 (&Bc)[TUA][JO][NT], f32x16 (&ac)[MT][NT], int cnt) {
#pragma unroll
        for (int u = 0; u < TUA; ++u) {
            if (u < cnt) {
                if constexpr (PREC == PREC_F32) {
#pragma unroll
                    for (int j = 0; j < JO; ++j)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int n = 0; n < NT; ++n) {
                                ac[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[u][j][m].x, Bc[u][j][n].x, ac[m][n], 0, 0, 0);
                                ac[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[u][j][m].y, Bc[u][j][n].y, ac[m][n], 0, 0, 0);
                                ac[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[u][j][m].z, Bc[u][j][n].z, ac[m][n], 0, 0, 0);
                                ac[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ac[u][j][m].w, Bc[u][j][n].w, ac[m][n], 0, 0, 0);
                            }
                } else if constexpr (PREC == PREC_F16) {
                    // [0] = channels 0-15, [1] = channels 16-31 of this 32-channel chunk: one MFMA each
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            ac[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Ac[u][0][m]), __builtin_bit_cast(f16x8, Bc[u][0][n]), ac[m][n], 0, 0, 0);
                            ac[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Ac[u][1][m]), __builtin_bit_cast(f16x8, Bc[u][1][n]), ac[m][n], 0, 0, 0);
                        }
                } else {
                    // [0] = hi halves, [1] = lo halves of the 16 channels of this chunk (K = 16 per MFMA);
                    // small cross terms first, then hi.hi
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            const f16x8 ah = __builtin_bit_cast(f16x8, Ac[u][0][m]), al = __builtin_bit_cast(f16x8, Ac[u][1][m]);
                            const f16x8 bh = __builtin_bit_cast(f16x8, Bc[u][0][n]), bl = __builtin_bit_cast(f16x8, Bc[u][1][n]);
                            if constexpr (XT) {          // D^T = W . X^T: lane = voxel, accumulators = channels
                                ac[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, ac[m][n], 0, 0, 0);
                                ac[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, ac[m][n], 0, 0, 0);
                                ac[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, ac[m][n], 0, 0, 0);
                            } else {
                                ac[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, ac[m][n], 0, 0, 0);
                                ac[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, ac[m][n], 0, 0, 0);
                                ac[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, ac[m][n], 0, 0, 0);
                            }
                        }
                }
            }
        }
    };
    // B ring step: request A of tap `ta` into An and B of stream position wp + sb tap steps into Bn,
    // then issue the MFMAs of (Ac, Bc).  B operands are requested two taps ahead of their use (the
    // L2 round trip is longer than one tap's MFMAs), A operands (LDS) one tap ahead.
    auto ring_step = [&](float4 (&An)[TUA][JO][MT], int ta, float4 (&Bn)[TUA][JO][NT], int sb,
                         const float4 (&Ac)[TUA][JO][MT], const float4 (&Bc)[TUA][JO][NT]) {
        prefetch(An, Bn, ta, sb);
        __builtin_amdgcn_sched_barrier(0);
        compute(Ac, Bc, acc[0], 1);
        __builtin_amdgcn_sched_barrier(0);
    };

    // taps of ONE staged chunk (sm points at it).  `hook` runs once per tap step between the MFMA groups: the PIPE
    // form issues one row of the next chunk's LDS-DMA there, so the transfers are spread over the whole tap loop.
    auto chunk_taps = [&](auto&& hook) {
        // A operands of the first TU taps of this chunk -> set 0 (B0 already holds their B operands)
#pragma unroll
        for (int u = 0; u < TUA; ++u) {
            const int to = __builtin_amdgcn_readlane(toff_v, (u < Tm1) ? u : Tm1);
#pragma unroll
            for (int j = 0; j < JO; ++j)
#pragma unroll
                for (int m = 0; m < MT; ++m) A0[u][j][m] = sm[abase[m] + to + j * 2];
        }
        t = 0;
        if constexpr (BL) {
            // set 0 holds the A operands of the chunk's first tap; its B operands (ring step bl_gs) were published one step ago at the
            // latest (first chunk: by the staging barrier, after the prologue's transfers were waited for)
            bl_read(B0, bl_gs);
            // BL_B1 (128 x 64 register tiles: 8 accumulator tiles = 128 registers): ONE B register set -- step g + 1's operands are read into
            // it after step g's MFMAs were issued (the barrier, the transfer and the A reads at the top of the next step cover the LDS
            // latency); with two sets the tile spills
            constexpr bool BL_B1 = (MT * NT >= 8);
            auto bl_step = [&](float4 (&An)[TUA][JO][MT], float4 (&Bn)[TUA][JO][NT], const int ta,
                               const float4 (&Ac)[TUA][JO][MT], float4 (&Bc)[TUA][JO][NT], f32x16 (&ac)[MT][NT]) {
                // (timing-only ablations, experiments build: dbg 1024 no barrier, 2048 no transfers, 4096 no end-of-step wait)
                if (!(p.dbg & 1024)) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
                if (!(p.dbg & 2048)) bl_issue();
                const int to = __builtin_amdgcn_readlane(toff_v, (ta < Tm1) ? ta : Tm1);
#pragma unroll
                for (int j = 0; j < JO; ++j)
#pragma unroll
                    for (int m = 0; m < MT; ++m) An[0][j][m] = sm[abase[m] + to + j * 2];
                if constexpr (!BL_B1) bl_read(Bn, bl_gs + 1);
                __builtin_amdgcn_sched_barrier(0);
                compute(Ac, Bc, ac, 1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (BL_B1) bl_read(Bc, bl_gs + 1);
                if (!(p.dbg & 4096)) wait_vmcnt(BL_NIW);
                ++bl_gs;
            };
#pragma unroll
            for (int c = 0; c < NCLS; ++c) {
                const int tend = (NCLS == 1) ? p.T : p.cls_end[c];
                while (t < tend) {
                    hook();
                    if constexpr (BL_B1) bl_step(A1, B0, t + 1, A0, B0, acc[c]); else bl_step(A1, B1, t + 1, A0, B0, acc[c]);
                    ++t;
                    if (t < tend) {
                        hook();
                        if constexpr (BL_B1) bl_step(A0, B0, t + 1, A1, B0, acc[c]); else bl_step(A0, B0, t + 1, A1, B1, acc[c]);
                        ++t;
                    }
                    else {
                        // odd run: the next tap's operands sit in set 1 -- read them again into set 0 (same LDS words)
                        const int to = __builtin_amdgcn_readlane(toff_v, (t < Tm1) ? t : Tm1);
#pragma unroll
                        for (int j = 0; j < JO; ++j)
#pragma unroll
                            for (int m = 0; m < MT; ++m) A0[0][j][m] = sm[abase[m] + to + j * 2];
                        bl_read(B0, bl_gs);
                    }
                }
            }
        } else if constexpr (RING3) {
            // invariant at the top: A0 = tap t, B0 = tap t, B1 = tap t+1 (stream positions wp, wp+1).
            // Three steps are one full turn of the B ring, so leaving after the first triple keeps
            // the invariant for the next chunk (whose A0 is reloaded anyway).
            for (; t < p.T; t += 6) {
                hook(); ring_step(A1, t + 1, B2, 2, A0, B0);
                hook(); ring_step(A0, t + 2, B0, 3, A1, B1);
                hook(); ring_step(A1, t + 3, B1, 4, A0, B2);
                if (t + 3 >= p.T) { wp += (size_t)3 * tstep; break; }
                hook(); ring_step(A0, t + 4, B2, 5, A1, B0);
                hook(); ring_step(A1, t + 5, B0, 6, A0, B1);
                hook(); ring_step(A0, t + 6, B1, 7, A1, B2);
                wp += (size_t)6 * tstep;
            }
        } else {
#pragma unroll
            for (int c = 0; c < NCLS; ++c) {
                const int tend = (NCLS == 1) ? p.T : p.cls_end[c];
                while (t < tend) {
                    const int cnt0 = (tend - t < TU) ? (tend - t) : TU;
                    hook();
                    prefetch(A1, B1, t + cnt0, cnt0);
                    __builtin_amdgcn_sched_barrier(0);
                    compute(A0, B0, acc[c], cnt0);
                    __builtin_amdgcn_sched_barrier(0);
                    wp += (size_t)cnt0 * tstep; t += cnt0;
                    const int cnt1 = (tend - t < TU) ? (tend - t) : TU;     // 0 when the run had an odd number of groups
                    hook();
                    prefetch(A0, B0, t + cnt1, cnt1);
                    __builtin_amdgcn_sched_barrier(0);
                    compute(A1, B1, acc[c], cnt1);
                    __builtin_amdgcn_sched_barrier(0);
                    wp += (size_t)cnt1 * tstep; t += cnt1;
                }
            }
        }
    };

    // ---- start-up stagger (de-phasing).  Every workgroup of a launch does the same work in the same time, so the workgroups that
    // share a CU -- and with them the whole chip -- march through "stage (HBM) -> taps (MFMA) -> epilogue (HBM)" in lock step: the
    // matrix pipes idle while everybody stages or stores, HBM idles while everybody runs taps, and the launch costs the SUM of
    // its phases although 2-4 workgroups per CU could overlap them (in-kernel timeline, profiles/round2/deconv_epilogue.txt).  The
    // workgroups of the first dispatch wave that land in residency slot s of their CU (dispatch order: slot = linear id / #CUs)
    // start s * stag_ticks later (100 MHz wall clock); their successors inherit the phase because every workgroup lasts equally long.
    if (p.stag_ticks) {
        const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
        if (lin < (unsigned)p.stag_n) {
            const unsigned slot = lin / (unsigned)p.stag_cus;
            if (slot) {
                const unsigned long long t0 = wall_clock64(), ticks = (unsigned long long)slot * (unsigned)p.stag_ticks;
                while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
            }
        }
    }
    OSA_TRACE(0);
    [[maybe_unused]] int trace_ev = 1;
    if constexpr (KS > 1) {
        // split-K: pass i stages chunk kg*cpg + i of every group (each group with its own NW*64 threads, into its own LDS slot)
        const int tid_g = tid - kg * (NW * 64);
        for (int i = 0; i < cpg; ++i) {
            if (i) __syncthreads();
            if (!(p.dbg & 1)) {
                if (PREC == PREC_F16 && !(p.act & OSA_IN_SPLIT)) stage_brick_cvt16<NW * 64, 1>(p, smem + kg * brickQ, brickQ, b, (kg * cpg + i) * CC, g0d, g0h, g0w, tid_g);
                else stage_brick<NW * 64, PREC, 1>(p, smem + kg * brickQ, brickQ, b, (kg * cpg + i) * CC, g0d, g0h, g0w, tid_g, s_in);
            }
            __syncthreads();
            sm = smem + kg * brickQ;
            chunk_taps([]() {});
        }
        // partial tiles of groups 1 .. KS-1 -> LDS (lane-contiguous: [group][wave][tile][register][lane]); group 0 adds them up
        __syncthreads();                                   // every group is done with the bricks
        float* const red = reinterpret_cast<float*>(smem);
        constexpr int NIK = MT * NT;
        if (kg > 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        red[(((((kg - 1) * NW + wave) * NIK + m * NT + n) * 16) + r) * 64 + lane] = acc[0][m][n][r];
        }
        __syncthreads();
        if (kg == 0) {
            for (int g = 0; g < KS - 1; ++g)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            acc[0][m][n][r] += red[((((g * NW + wave) * NIK + m * NT + n) * 16) + r) * 64 + lane];
        }
    } else if constexpr (!PIPE) {
    for (int ch0 = 0; ch0 < p.nchunks; ch0 += p.cps) {
        if (ch0) __syncthreads();
        OSA_TRACE(trace_ev); ++trace_ev;                 // pass start (after the previous pass's readers are done)
        const int ncl = (p.nchunks - ch0 < p.cps) ? (p.nchunks - ch0) : p.cps;
        if (!(p.dbg & 1) && PREC != PREC_F32 && p.dma) {
            for (int cl = 0; cl < ncl; ++cl)
                stage_brick_dma<NW * 64>(p, smem + cl * brickQ, b, (ch0 + cl) * CC, g0d, g0h, g0w, tid);
        } else if (!(p.dbg & 1) && PREC == PREC_F16 && !(p.act & OSA_IN_SPLIT)) {
            constexpr int SU = OSA_S2TILE ? OSA_S2_U : OSA_STAGE_U;
            int cl = 0;
            for (; cl + 2 <= ncl; cl += 2)
                stage_brick_cvt16<NW * 64, 2, SU>(p, smem + cl * brickQ, brickQ, b, (ch0 + cl) * CC, g0d, g0h, g0w, tid);
            if (cl < ncl)
                stage_brick_cvt16<NW * 64, 1, SU>(p, smem + cl * brickQ, brickQ, b, (ch0 + cl) * CC, g0d, g0h, g0w, tid);
        } else if (!(p.dbg & 1)) {
            constexpr int SU = OSA_S2TILE ? OSA_S2_U : OSA_STAGE_U;
            int cl = 0;
            for (; cl + 2 <= ncl; cl += 2)
                stage_brick<NW * 64, PREC, 2, SU>(p, smem + cl * brickQ, brickQ, b, (ch0 + cl) * CC, g0d, g0h, g0w, tid, s_in);
            if (cl < ncl)
                stage_brick<NW * 64, PREC, 1, SU>(p, smem + cl * brickQ, brickQ, b, (ch0 + cl) * CC, g0d, g0h, g0w, tid, s_in);
        }
        OSA_TRACE(trace_ev); ++trace_ev;                 // own staging loads issued + written
        if constexpr (BL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ring transfers issued through inline asm: invisible to the compiler's own counts
        __syncthreads();
        OSA_TRACE(trace_ev); ++trace_ev;                 // brick complete
        for (int cl = 0; cl < ncl; ++cl) {
            sm = smem + cl * brickQ;
            chunk_taps([]() {});
        }
        OSA_TRACE(trace_ev); ++trace_ev;                 // taps done
    }
    }

    unsigned amax_seen = 0u;
    auto epilogue = [&](float* tbase) {
    // ---- epilogue: BN affine + residual + activation (+ sigmoid gate), NDHWC store ----
    // MFMA result layout: lane -> output channel `col`, accumulator r -> voxel row (r&3)+8(r>>2)+4hh.
    // Each 32x32 tile is transposed through a wave-private LDS buffer (row stride 36 floats, conflict
    // free both ways) so that a lane ends up with 4 consecutive channels of one voxel: residual /
    // gate loads and output stores are float4, 8 lanes cover one voxel's 128-byte channel row and a
    // wave instruction covers 8 consecutive voxels (1 KB contiguous for a 32-channel tensor).
    // folded-BN scale / shift of this lane's channel quads: requested before the barrier so the loads
    // overlap the tail of the tap loop instead of stalling the first tile of the epilogue
    float4 scv[NT], shv[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = n0 + (wn * NT + n) * 32 + (lane & 7) * 4;
        float4 sc = make_float4(osc, osc, osc, osc), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (co < p.Co && p.scale) {
            if (co + 3 < p.Co) { sc = *reinterpret_cast<const float4*>(p.scale + co); sh = *reinterpret_cast<const float4*>(p.shift + co); }
            else {
                sc.x = p.scale[co]; sh.x = p.shift[co];
                if (co + 1 < p.Co) { sc.y = p.scale[co + 1]; sh.y = p.shift[co + 1]; }
                if (co + 2 < p.Co) { sc.z = p.scale[co + 2]; sh.z = p.shift[co + 2]; }
            }
            sc.x *= osc; sc.y *= osc; sc.z *= osc; sc.w *= osc;
        }
        scv[n] = sc; shv[n] = sh;
    }
    if (p.out_meta) amax_seen = amax_peek(p.out_meta);   // early: its latency hides behind the epilogue
    if constexpr (BL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring's run-ahead transfers: the fast path below counts its own loads only
    if constexpr (!PIPE) __syncthreads();              // everyone is done reading the input brick (PIPE: the chunk's end barrier)
    OSA_TRACE(20);
    if (KS > 1 && kg != 0) return;                     // split-K: group 0 holds the sums (the others rejoin at publish_amax)
    if (p.dbg & 8) {                                   // timing only: no epilogue (keeps the accumulators live)
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCLS; ++c)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s += acc[c][m][n][r];
        if (s == 12345.678f) p.y[0] = s;
        return;
    }
#ifndef OSA_TB
#define OSA_TB 1                                   // transpose buffers per wave (2: tile i+1's transpose may start while tile i is finalised)
#endif
    float* const tb0 = tbase + wave * (OSA_TB * 32 * 36);
    // per-batch-item base pointers (wave-uniform, 64-bit); everything per lane is a 32-bit element offset
    const size_t bvox = (size_t)b * p.Do * p.Ho * p.Wo;
    float* yb = p.y + bvox * p.yCs;
    const float* resb = p.res ? p.res + bvox * p.rCs : nullptr;
    const float* gateb = p.gate ? p.gate + (size_t)b * p.Ho * p.Wo * p.gCs : nullptr;
    const bool vec4 = ((p.yCs & 3) == 0) && ((p.Co & 3) == 0) && (((size_t)p.y & 15) == 0) &&
                      (!p.res || (((p.rCs & 3) == 0) && (((size_t)p.res & 15) == 0))) &&
                      (!p.gate || (((p.gCs & 3) == 0) && (((size_t)p.gate & 15) == 0)));
    const int vsub = lane >> 3, cq = (lane & 7) * 4;   // voxel within a group of 8, channel quad
    const int actk = p.act & 15;
    const bool gate_raw = (p.act & OSA_GATE_RAW) != 0;
    const int gate_co = ((unsigned)p.act >> 16) ? (int)((unsigned)p.act >> 16) : 0x7fffffff;   // OSA_GATE_CHANNELS(n): gate output channels < n only
    // ---- fast path.  Workgroups whose brick lies inside the tensor, with every channel of their N tiles present,
    // 16-byte aligned rows, no gate and a cheap activation (none / ReLU / LeakyReLU / ReLU6) -- i.e. nearly all workgroups
    // of a full-size layer -- finalise their tiles with STRAIGHT-LINE code (FULL = true below): no per-lane predicate, no
    // branch around a load or a store.  This is not cosmetic: vmcnt retires in order, so the wait in front of tile i's
    // residual / redir rows (requested PD tiles earlier) may leave the younger requests outstanding only if the compiler
    // knows how many there are; one conditional memory operation in between and it has to emit vmcnt(0), which also
    // waits for the stores of the previous tile and the prefetch just issued -- every tile then pays a full memory round
    // trip (measured 3.7 us per tile in the fused transposed conv, 2 waves per SIMD).  A missing residual is handled
    // without a branch: the loads go to one valid dummy address (offset masked to 0) and the value is discarded by a select.
    const bool has_res = p.res != nullptr;
    const float* const rbase = has_res ? resb : reinterpret_cast<const float*>(p.w);
    const int rmask = has_res ? -1 : 0;
    const float act_ns = (actk == OSA_ACT_NONE) ? 1.f : ((actk == OSA_ACT_LEAKY) ? p.slope : 0.f);    // slope for v < 0
    const float act_hi = (actk == OSA_ACT_RELU6) ? 6.f : __builtin_inff();
    auto act_cheap = [&](float v) { v = (v < 0.f) ? v * act_ns : v; return fminf(v, act_hi); };
    // compiled for the f16x3 instantiations except (i) the 64-channel redir variant, which has no registers to spare, and (ii) the
    // 256-voxel x 32-channel tile of the dominant 32 -> 32 layers: two tiles per wave and 4 waves per SIMD hide the waits anyway, and
    // the second code path costs it 6 registers at its 128 cap (measured -1.5 % on that launch).  The f32 mode keeps the predicated
    // code everywhere: its 4-waves-per-SIMD tiles would spill.
    constexpr bool FASTC = (PREC != PREC_F32) && (REDIR != 2) && !(NCLS == 1 && MT == 2 && NT == 1);
    const bool fast = FASTC && (a0d + TD <= p.Ad) && (a0h + TH <= p.Ah) && (a0w + TW <= p.Aw) && (n0 + WN * NT * 32 <= p.Co) &&
                      vec4 && !p.gate && actk <= OSA_ACT_RELU6 && !(p.act & OSA_RES_AFTER_ACT) && !(p.dbg & (32 | 64));
    // A wave finalises NI = MT*NCLS*NT tiles of 32 voxels x 32 channels one after the other.  The
    // residual rows of tile i+PD are requested before tile i is processed (rolling window of PD
    // tiles, static register sets), so the HBM round trip of a residual overlaps the LDS transposes,
    // arithmetic and stores of the PD-1 tiles in front of it -- the fused transposed conv has 8 tiles
    // per wave and spent half of its time waiting for them one by one.
    constexpr int NI = MT * NCLS * NT;
#ifndef OSA_PD_REDIR
#define OSA_PD_REDIR 2
#endif
    constexpr int PD = REDIR ? ((REDIR == 1) ? OSA_PD_REDIR : 2) : ((NCLS >= 4) ? ((PREC != PREC_F32 && NCLS == 8) ? 2 : 3) : ((NI < 2) ? NI : 2));
    // voxel bookkeeping of the 4 rows (vsub + 8k) this lane finalises in M tile m
    auto rows_of = [&](auto F, int m, int (&v0)[4], int (&g0)[4], bool (&vok)[4]) {
        constexpr bool FULL = decltype(F)::value;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = (wm * MT + m) * 32 + vsub + 8 * k;
            const int ad = a0d + q / (TW * TH), ah = a0h + (q / TW) % TH, aw = a0w + q % TW;
            vok[k] = FULL || (ad < p.Ad && ah < p.Ah && aw < p.Aw);
            const int os_ = (!FULL && (p.dbg & 64)) ? 1 : p.os;                // dbg 64: contiguous rows (timing only)
            v0[k] = ((ad * os_) * p.Ho + ah * os_) * p.Wo + aw * os_;          // voxel index inside batch item b (host: < 2^31 elements)
            g0[k] = (ah * p.os) * p.Wo + aw * p.os;
        }
    };
    auto class_off = [&](int c, int& coff, int& goff) {
        const int ood = (NCLS == 1) ? p.ood : ((c >> 2) & 1), ooh = (NCLS == 1) ? p.ooh : ((c >> 1) & 1),
                  oow = (NCLS == 1) ? p.oow : (c & 1);
        coff = (ood * p.Ho + ooh) * p.Wo + oow;              // supported transposed convs: Do == 2*Di
        if (p.dbg & 64) coff = ood * (p.Ad * p.Ho * p.Wo) + (ooh * 2 + oow) * TW;
        goff = ooh * p.Wo + oow;
    };
    // tile order: m outer, class, n inner
    auto load_res = [&](auto F, int i, float4 (&rv)[4]) {
        constexpr bool FULL = decltype(F)::value;
        const int n = i % NT, c = (i / NT) % NCLS, m = i / (NT * NCLS);
        int v0[4], g0[4], coff, goff; bool vok[4];
        rows_of(F, m, v0, g0, vok);
        class_off(c, coff, goff);
        const int co = n0 + (wn * NT + n) * 32 + cq;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if constexpr (FULL) {                       // fp32 rows, 16-byte aligned (fast-path conditions); unconditional
                rv[k] = *reinterpret_cast<const float4*>(rbase + (((v0[k] + coff) * p.rCs + co) & rmask));
                continue;
            }
            rv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.res && vok[k] && co < p.Co) {
                const float* rp = resb + (v0[k] + coff) * p.rCs + co;
                if (PREC == PREC_F16X3 && (p.act & OSA_RES_SPLIT)) {
                    const float* rs = resb + (v0[k] + coff) * p.rCs;
                    const uint2 h = *reinterpret_cast<const uint2*>(rs + split_off_hi(co));
                    const uint2 l = *reinterpret_cast<const uint2*>(rs + split_off_lo(co));
                    rv[k] = __builtin_bit_cast(float4, make_uint4(h.x, h.y, l.x, l.y));     // decoded in finish()
                } else if (PREC == PREC_F16 && (p.act & OSA_RES_SPLIT)) {                 // fp16 residual (rCs in float units): decoded here
                    const f16x4 h = __builtin_bit_cast(f16x4, *reinterpret_cast<const uint2*>(resb + (v0[k] + coff) * p.rCs + (co >> 1)));
                    rv[k] = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
                } else if (vec4) rv[k] = *reinterpret_cast<const float4*>(rp);
                else {
                    float* rr = &rv[k].x;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < p.Co) rr[e] = rp[e];
                }
            }
        }
    };
    auto finish = [&](auto F, int i, const float4 (&rv)[4]) {
        constexpr bool FULL = decltype(F)::value;
        const int n = i % NT, c = (i / NT) % NCLS, m = i / (NT * NCLS);
        int v0[4], g0[4], coff, goff; bool vok[4];
        rows_of(F, m, v0, g0, vok);
        class_off(c, coff, goff);
        // registers -> LDS (tile[voxel row][channel])
        float* const tb = tb0 + (i % OSA_TB) * (32 * 36);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            tb[((r & 3) + 8 * (r >> 2) + 4 * hh) * 36 + col] = acc[c][m][n][r];
        const int co = n0 + (wn * NT + n) * 32 + cq;
        const bool cok = FULL || co < p.Co;
        const float4 sc = scv[n], sh = shv[n];
        // LDS -> registers (4 voxels x 4 channels per lane); gate rows requested together
        float4 av[4], gv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            av[k] = *reinterpret_cast<const float4*>(tb + (vsub + 8 * k) * 36 + cq);
            gv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!FULL && !REDIR && p.gate && vok[k] && cok && co < gate_co) {
                const float* gp = gateb + (g0[k] + goff) * p.gCs + co;
                if (vec4) gv[k] = *reinterpret_cast<const float4*>(gp);
                else {
                    float* gg = &gv[k].x;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < p.Co) gg[e] = gp[e];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float o[4];
            const float a4[4] = {av[k].x, av[k].y, av[k].z, av[k].w};
            const float s4[4] = {sc.x, sc.y, sc.z, sc.w}, h4[4] = {sh.x, sh.y, sh.z, sh.w};
            float4 rk = rv[k];
            if constexpr (FULL) rk = make_float4(has_res ? rk.x : 0.f, has_res ? rk.y : 0.f, has_res ? rk.z : 0.f, has_res ? rk.w : 0.f);
            if (!FULL && PREC == PREC_F16X3 && (p.act & OSA_RES_SPLIT) && p.res) {
                const uint4 b4 = __builtin_bit_cast(uint4, rv[k]);
                rk = mul4(join_f16(make_uint2(b4.x, b4.y), make_uint2(b4.z, b4.w)), s_res_inv);
            }
            const float r4[4] = {rk.x, rk.y, rk.z, rk.w}, g4[4] = {gv[k].x, gv[k].y, gv[k].z, gv[k].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t_ = fmaf(a4[e], s4[e], h4[e]);
                float v = t_ + r4[e];
                if constexpr (FULL) { o[e] = act_cheap(v); continue; }
                if (p.act & OSA_RES_AFTER_ACT) {                    // relu(residual + relu(bn(conv))): RAFT-style ResidualBlock (extractor.py:48-60)
                    o[e] = fmaxf(fmaxf(t_, 0.f) + r4[e], 0.f);
                    continue;
                }
                if (actk == OSA_ACT_RELU) v = fmaxf(v, 0.f);
                else if (actk == OSA_ACT_LEAKY) v = (v > 0.f) ? v : v * p.slope;
                else if (actk == OSA_ACT_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
                else if (actk == OSA_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                else if (actk == OSA_ACT_TANH) v = tanhf(v);
                if (!REDIR && p.gate && co < gate_co) v *= gate_raw ? g4[e] : 1.0f / (1.0f + expf(-g4[e]));
                o[e] = v;
            }
            if constexpr (FULL) {
                am = fmaxf(am, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
                store16(yb + (v0[k] + coff) * p.yCs + co, make_float4(o[0], o[1], o[2], o[3]));
                continue;
            }
            if (vok[k] && cok && !(p.dbg & 32)) {             // dbg 32: no stores (timing only)
                am = fmaxf(am, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
                float* yp = yb + (v0[k] + coff) * p.yCs + co;
                if (vec4) store16(yp, make_float4(o[0], o[1], o[2], o[3]));
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < p.Co) yp[e] = o[e];
                }
            }
        }
    };
    // ---- split-output epilogue (OUTS): a lane takes 8 consecutive channels of 2 voxels of the tile, so the
    // hi halves and the lo halves of its 8 values are one 16-byte store each; a split residual is read the
    // same way (host: a split output takes a split residual, no gate).
    const int vs2 = lane >> 2, c8 = (lane & 3) * 8;
    auto rows2 = [&](auto F, int m, int (&v0)[2], bool (&vok)[2]) {
        constexpr bool FULL = decltype(F)::value;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int q = (wm * MT + m) * 32 + vs2 + 16 * k;
            const int ad = a0d + q / (TW * TH), ah = a0h + (q / TW) % TH, aw = a0w + q % TW;
            vok[k] = FULL || (ad < p.Ad && ah < p.Ah && aw < p.Aw);
            const int os_ = (!FULL && (p.dbg & 64)) ? 1 : p.os;
            v0[k] = ((ad * os_) * p.Ho + ah * os_) * p.Wo + aw * os_;
        }
    };
    auto load_res8 = [&](auto F, int i, float4 (&rv)[4]) {
        constexpr bool FULL = decltype(F)::value;
        const int n = i % NT, c = (i / NT) % NCLS, m = i / (NT * NCLS);
        int v0[2], coff, goff; bool vok[2];
        rows2(F, m, v0, vok);
        class_off(c, coff, goff);
        const int co = n0 + (wn * NT + n) * 32 + c8;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            // f16x3: 8 hi + 8 lo halves of the split residual; f16: the 8 fp16 values themselves (one 16-byte row, rCs in float units)
            const int roff = (PREC == PREC_F16) ? (co >> 1) : ((co >> 4) * 16 + ((co & 15) >> 3) * 4);
            if constexpr (FULL) {
                const float* rs = rbase + (((v0[k] + coff) * p.rCs + roff) & rmask);
                rv[2 * k] = *reinterpret_cast<const float4*>(rs);
                if constexpr (PREC != PREC_F16) rv[2 * k + 1] = *reinterpret_cast<const float4*>(rs + 8);
                continue;
            }
            rv[2 * k] = make_float4(0.f, 0.f, 0.f, 0.f); rv[2 * k + 1] = rv[2 * k];
            if (p.res && vok[k] && co < p.Co) {
                const float* rs = resb + (v0[k] + coff) * p.rCs + roff;
                rv[2 * k] = *reinterpret_cast<const float4*>(rs);            // 8 hi halves
                if constexpr (PREC != PREC_F16) rv[2 * k + 1] = *reinterpret_cast<const float4*>(rs + 8);    // 8 lo halves
            }
        }
    };
    auto finish8 = [&](auto F, int i, const float4 (&rv)[4], const float4 (&sc8)[2], const float4 (&sh8)[2]) {
        constexpr bool FULL = decltype(F)::value;
        const int n = i % NT, c = (i / NT) % NCLS, m = i / (NT * NCLS);
        int v0[2], coff, goff; bool vok[2];
        rows2(F, m, v0, vok);
        class_off(c, coff, goff);
        float* const tb = tb0 + (i % OSA_TB) * (32 * 36);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            tb[((r & 3) + 8 * (r >> 2) + 4 * hh) * 36 + col] = acc[c][m][n][r];
        const int co = n0 + (wn * NT + n) * 32 + c8;
        const bool cok = FULL || co < p.Co;
        float4 av[2][2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            av[k][0] = *reinterpret_cast<const float4*>(tb + (vs2 + 16 * k) * 36 + c8);
            av[k][1] = *reinterpret_cast<const float4*>(tb + (vs2 + 16 * k) * 36 + c8 + 4);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            uint2 hq[2], lq[2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((FULL && !REDIR) || p.res) {
                    const uint4 hb = __builtin_bit_cast(uint4, rv[2 * k]), lb = __builtin_bit_cast(uint4, rv[2 * k + 1]);
                    if constexpr (PREC == PREC_F16) {
                        const f16x4 q = __builtin_bit_cast(f16x4, h2 ? make_uint2(hb.z, hb.w) : make_uint2(hb.x, hb.y));
                        r = make_float4((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
                    } else {
                        r = h2 ? join_f16(make_uint2(hb.z, hb.w), make_uint2(lb.z, lb.w)) : join_f16(make_uint2(hb.x, hb.y), make_uint2(lb.x, lb.y));
                        r = mul4(r, s_res_inv);
                    }
                    if constexpr (FULL) r = make_float4(has_res ? r.x : 0.f, has_res ? r.y : 0.f, has_res ? r.z : 0.f, has_res ? r.w : 0.f);
                }
                const float a4[4] = {av[k][h2].x, av[k][h2].y, av[k][h2].z, av[k][h2].w};
                const float s4[4] = {sc8[h2].x, sc8[h2].y, sc8[h2].z, sc8[h2].w}, t4[4] = {sh8[h2].x, sh8[h2].y, sh8[h2].z, sh8[h2].w};
                const float r4[4] = {r.x, r.y, r.z, r.w};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = fmaf(a4[e], s4[e], t4[e]) + r4[e];
                    if constexpr (FULL) { o[e] = act_cheap(v); continue; }
                    if (actk == OSA_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (actk == OSA_ACT_LEAKY) v = (v > 0.f) ? v : v * p.slope;
                    else if (actk == OSA_ACT_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
                    else if (actk == OSA_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                    else if (actk == OSA_ACT_TANH) v = tanhf(v);
                    o[e] = v;
                }
                if (vok[k] && cok) am = fmaxf(am, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
                if constexpr (PREC == PREC_F16) {          // fp16 output tensor: round to nearest even, the lane's 8 channels are one 16-byte row
                    const f16x4 q = {(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
                    hq[h2] = __builtin_bit_cast(uint2, q); lq[h2] = hq[h2];
                } else split_f16(make_float4(o[0] * s_out, o[1] * s_out, o[2] * s_out, o[3] * s_out), hq[h2], lq[h2]);
            }
            if (FULL || (vok[k] && cok && !(p.dbg & 32))) {
                if constexpr (PREC == PREC_F16) {
                    store16(yb + (v0[k] + coff) * p.yCs + (co >> 1), make_uint4(hq[0].x, hq[0].y, hq[1].x, hq[1].y));     // yCs in float units
                } else {
                    float* ys = yb + (v0[k] + coff) * p.yCs + (co >> 4) * 16 + ((co & 15) >> 3) * 4;
                    store16(ys, make_uint4(hq[0].x, hq[0].y, hq[1].x, hq[1].y));
                    store16(ys + 8, make_uint4(lq[0].x, lq[0].y, lq[1].x, lq[1].y));
                }
            }
        }
    };
    // BN scale / shift of the lane's 8 channels in that mapping (REDIR: already applied in accumulator layout)
    auto bn8 = [&](int n, float4 (&sc8)[2], float4 (&sh8)[2]) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int co = n0 + (wn * NT + n) * 32 + c8 + 4 * h2;
            sc8[h2] = make_float4(osc, osc, osc, osc); sh8[h2] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (REDIR) { sc8[h2] = make_float4(1.f, 1.f, 1.f, 1.f); continue; }
            if (co + 3 < p.Co && p.scale) {
                sc8[h2] = *reinterpret_cast<const float4*>(p.scale + co); sh8[h2] = *reinterpret_cast<const float4*>(p.shift + co);
                sc8[h2].x *= osc; sc8[h2].y *= osc; sc8[h2].z *= osc; sc8[h2].w *= osc;
            }
        }
    };
    constexpr int RV = (REDIR == 2) ? 8 : 4;      // float4 rows held per prefetched tile
    float4 rvb[PD][RV];
    if constexpr (REDIR) {
        {
            // ---- fused redir branch: R = BN_r(W_r . x) for the 32 output voxels of every tile, on the
            // MFMA in accumulator layout (lane = channel, register = voxel row), then
            // z = fma(acc, s, t) + fma(R, s_r, t_r) replaces the accumulator and the common path below
            // runs with unit scale and no residual -- the same arithmetic, in the same order, as the
            // separate 1x1x1 launch whose output used to be read back as the residual.
            const float* rxb = p.rx + bvox * p.rxCs;
            const size_t rbstep = (size_t)2 * p.CoP, rtstep = (size_t)JO * rbstep;
            constexpr int RCH = RV / 2;                            // chunks of 16 redir input channels held per tile
            const int rch = (p.rCi + CC - 1) / CC;
            // first channel a quad load may NOT start at: rCi for plain tensors; a split tensor's chunks are whole ([16 hi | 16 lo] per 16 channels)
            const int rlim = (PREC == PREC_F16X3 && (p.act & OSA_REDIR_SPLIT)) ? rch * CC : p.rCi;
            // x rows of tile i in MFMA A-operand order: lane (col, hh) -> voxel row `col`
            auto load_x = [&](auto F, int i, float4 (&rv)[RV]) {
                constexpr bool FULL = decltype(F)::value;
                const int c = (i / NT) % NCLS, m = i / (NT * NCLS);
                const int q = (wm * MT + m) * 32 + col;
                const int ad = a0d + q / (TW * TH), ah = a0h + (q / TW) % TH, aw = a0w + q % TW;
                const bool ok = FULL || (ad < p.Ad && ah < p.Ah && aw < p.Aw);
                const int vox = ((ad * 2 + ((c >> 2) & 1)) * p.Ho + ah * 2 + ((c >> 1) & 1)) * p.Wo + aw * 2 + (c & 1);
#pragma unroll
                for (int k = 0; k < RV; ++k) {
                    rv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    const int ch = k >> 1, j = k & 1;
                    // f16x3: 8 consecutive channels 8hh..8hh+7 of the chunk (two float4s); f32: channels 8j+4hh..+3
                    int cin = ch * CC + ((PREC == PREC_F32) ? (8 * j + 4 * hh) : (8 * hh + 4 * j));
                    // split redir input: j = 0 -> the lane's 8 hi halves, j = 1 -> its 8 lo halves (16 B each)
                    if (PREC == PREC_F16X3 && (p.act & OSA_REDIR_SPLIT)) cin = ch * CC + 4 * hh + 8 * j;
                    // (cin < rCi per quad: with 8 redir channels the upper half of the chunk would be the NEXT voxel's channels -- harmless
                    // under zero weights unless it is the tensor's last voxel and the bytes behind the allocation decode as NaN / inf:
                    // NaN x 0 = NaN, ReLU(NaN) = 0 -- found in r6 as an order-dependent test failure, StereoBase / IGEV run 8-channel hourglasses)
                    if (FULL || (ok && ch < rch && cin < rlim)) rv[k] = *reinterpret_cast<const float4*>(rxb + vox * p.rxCs + cin);
                }
            };
            // Tile-invariant operands of the branch, loaded ONCE: the stores of the tiles in between may alias them as far
            // as the compiler knows, so left inside add_redir they are re-read from L2 for every tile with their latency
            // exposed (2 waves per SIMD) -- measured 2.2 us per tile, most of the fused epilogue.  NT == 1 in every redir
            // configuration.  REDIR == 2 (64 redir channels) has no registers left for the weights: it keeps the in-place loads.
            static_assert(NT == 1, "fused redir configurations have one N tile per wave");
            const float4* const rwp = p.rw + (size_t)hh * p.CoP + n0 + wn * 32 + col;
            constexpr bool HOIST_W = (REDIR == 1);
            float4 rwb[HOIST_W ? RCH : 1][2];
            if constexpr (HOIST_W) {
#pragma unroll
                for (int ch = 0; ch < RCH; ++ch) {
                    rwb[ch][0] = rwb[ch][1] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ch < rch) { rwb[ch][0] = rwp[ch * rtstep]; rwb[ch][1] = rwp[ch * rtstep + rbstep]; }
                }
            }
            // per-lane (channel `col`) BN factors of both branches
            const int cl = n0 + wn * 32 + col;
            const bool lok = cl < p.Co;
            float s6h = osc, t6h = 0.f, srh = rosc, trh = 0.f;
            if constexpr (HOIST_W) {
                s6h = (lok && p.scale) ? p.scale[cl] * osc : osc; t6h = (lok && p.shift) ? p.shift[cl] : 0.f;
                srh = (lok && p.rscale) ? p.rscale[cl] * rosc : rosc; trh = (lok && p.rshift) ? p.rshift[cl] : 0.f;
            }
            auto add_redir = [&](auto F, int i, const float4 (&rv)[RV]) {
                constexpr bool FULL = decltype(F)::value;
                const int n = i % NT, c = (i / NT) % NCLS, m = i / (NT * NCLS);
                f32x16 r;
#pragma unroll
                for (int e = 0; e < 16; ++e) r[e] = 0.f;
#pragma unroll
                for (int ch = 0; ch < RCH; ++ch) {
                    if (FULL || ch < rch) {
                        float4 b0, b1;
                        if constexpr (HOIST_W) { b0 = rwb[ch][0]; b1 = rwb[ch][1]; }
                        else { b0 = rwp[ch * rtstep]; b1 = rwp[ch * rtstep + rbstep]; }
                        if constexpr (PREC == PREC_F32) {
                            const float4 a0 = rv[2 * ch], a1 = rv[2 * ch + 1];
                            r = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, r, 0, 0, 0);
                            r = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, r, 0, 0, 0);
                            r = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, r, 0, 0, 0);
                            r = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, r, 0, 0, 0);
                            r = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, r, 0, 0, 0);
                            r = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, r, 0, 0, 0);
                            r = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, r, 0, 0, 0);
                            r = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, r, 0, 0, 0);
                        } else {
                            f16x8 ah, al;
                            if (p.act & OSA_REDIR_SPLIT) {
                                ah = __builtin_bit_cast(f16x8, rv[2 * ch]); al = __builtin_bit_cast(f16x8, rv[2 * ch + 1]);
                            } else {
                                uint2 h0, l0, h1, l1;
                                split_f16(mul4(rv[2 * ch], s_rx), h0, l0);
                                split_f16(mul4(rv[2 * ch + 1], s_rx), h1, l1);
                                ah = __builtin_bit_cast(f16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
                                al = __builtin_bit_cast(f16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
                            }
                            const f16x8 bh = __builtin_bit_cast(f16x8, b0), bl = __builtin_bit_cast(f16x8, b1);
                            r = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, r, 0, 0, 0);
                            r = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, r, 0, 0, 0);
                            r = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, r, 0, 0, 0);
                        }
                    }
                }
                const float s6 = HOIST_W ? s6h : ((lok && p.scale) ? p.scale[cl] * osc : osc), t6 = HOIST_W ? t6h : ((lok && p.shift) ? p.shift[cl] : 0.f);
                const float sr = HOIST_W ? srh : ((lok && p.rscale) ? p.rscale[cl] * rosc : rosc), tr = HOIST_W ? trh : ((lok && p.rshift) ? p.rshift[cl] : 0.f);
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[c][m][n][e] = fmaf(acc[c][m][n][e], s6, t6) + fmaf(r[e], sr, tr);
            };
#pragma unroll
            for (int n = 0; n < NT; ++n) { scv[n] = make_float4(1.f, 1.f, 1.f, 1.f); shv[n] = make_float4(0.f, 0.f, 0.f, 0.f); }
            const float4 zero4[4] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f),
                                     make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
            // fast path also needs every redir chunk present (no conditional weight / x-row loads)
            const bool fast_r = fast && rch == RCH && p.rCi == RCH * CC;
            auto run = [&](auto F) {
#pragma unroll
                for (int i = 0; i < PD; ++i) load_x(F, i, rvb[i]);
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    add_redir(F, i, rvb[i % PD]);
                    if (i + PD < NI) load_x(F, i + PD, rvb[i % PD]);
                    if constexpr (OUTS) {
                        float4 sc8[2], sh8[2];
                        bn8(i % NT, sc8, sh8);
                        finish8(F, i, zero4, sc8, sh8);
                    } else finish(F, i, zero4);
                    OSA_TRACE(21 + i);
                    __builtin_amdgcn_sched_barrier(0);       // tiles are scheduled one at a time (the prefetch depth PD is explicit): bounds live ranges
                }
            };
            if constexpr (FASTC) { if (fast_r) run(std::true_type{}); else run(std::false_type{}); }
            else run(std::false_type{});
        }
    }
    if constexpr (XT) {
        // ---- transposed-accumulator epilogue: lane (col, hh) owns voxel `col` of M tile m; per 16-channel block P of the N tile it holds
        // channels 16P + 4hh + {0..3} (accumulators 8P .. 8P+3) and 16P + 8 + 4hh + {0..3} (8P+4 .. 8P+7).  v_permlane32_swap exchanges the
        // first quad of the hh = 1 lanes with the second quad of the hh = 0 lanes: afterwards a lane holds the 8 CONSECUTIVE channels
        // 16P + 8hh .. + 7 -- one 16-byte row of hi halves and one of lo halves in the split layout, like the residual it reads.
        auto rowT = [&](auto F, int m, int& v0, bool& vok) {
            constexpr bool FULL = decltype(F)::value;
            const int q = (wm * MT + m) * 32 + col;
            const int ad = a0d + q / (TW * TH), ah = a0h + (q / TW) % TH, aw = a0w + q % TW;
            vok = FULL || (ad < p.Ad && ah < p.Ah && aw < p.Aw);
            const int os_ = (!FULL && (p.dbg & 64)) ? 1 : p.os;
            v0 = ((ad * os_) * p.Ho + ah * os_) * p.Wo + aw * os_;
        };
        auto load_res8T = [&](auto F, int i, float4 (&rv)[4]) {
            constexpr bool FULL = decltype(F)::value;
            const int n = i % NT, m = i / NT;
            int v0, coff, goff; bool vok;
            rowT(F, m, v0, vok);
            class_off(0, coff, goff);
#pragma unroll
            for (int P = 0; P < 2; ++P) {
                const int co = n0 + (wn * NT + n) * 32 + 16 * P + 8 * hh;
                const int off = (v0 + coff) * p.rCs + (co >> 4) * 16 + ((co & 15) >> 3) * 4;
                if constexpr (FULL) {
                    const float* rs = rbase + (off & rmask);
                    rv[2 * P] = *reinterpret_cast<const float4*>(rs);
                    rv[2 * P + 1] = *reinterpret_cast<const float4*>(rs + 8);
                    continue;
                }
                rv[2 * P] = make_float4(0.f, 0.f, 0.f, 0.f); rv[2 * P + 1] = rv[2 * P];
                if (p.res && vok && co < p.Co) {
                    rv[2 * P] = *reinterpret_cast<const float4*>(resb + off);            // 8 hi halves
                    rv[2 * P + 1] = *reinterpret_cast<const float4*>(resb + off + 8);    // 8 lo halves
                }
            }
        };
        auto finish8T = [&](auto F, int i, const float4 (&rv)[4], const float4 (&sc8)[2][2], const float4 (&sh8)[2][2]) {
            constexpr bool FULL = decltype(F)::value;
            const int n = i % NT, m = i / NT;
            int v0, coff, goff; bool vok;
            rowT(F, m, v0, vok);
            class_off(0, coff, goff);
#pragma unroll
            for (int P = 0; P < 2; ++P) {
                const int co = n0 + (wn * NT + n) * 32 + 16 * P + 8 * hh;
                const bool cok = FULL || co < p.Co;
                float a8[2][4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {         // executed by all 64 lanes (no lane-dependent branch around it)
                    // (__float_as_uint of a scalar copy: __builtin_bit_cast applied to the vector element itself makes this clang fold every
                    // swap of the tile onto accumulator 0 -- tools/experiments note in profiles/DESIGN_rounds1-5.md 3.2 r3)
                    const float xf = acc[0][m][n][8 * P + e], yf = acc[0][m][n][8 * P + 4 + e];
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(xf), __float_as_uint(yf), false, false);
                    a8[0][e] = __uint_as_float(sw[0]); a8[1][e] = __uint_as_float(sw[1]);
                }
                uint2 hq[2], lq[2];
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (FULL || p.res) {
                        const uint4 hb = __builtin_bit_cast(uint4, rv[2 * P]), lb = __builtin_bit_cast(uint4, rv[2 * P + 1]);
                        r = h2 ? join_f16(make_uint2(hb.z, hb.w), make_uint2(lb.z, lb.w)) : join_f16(make_uint2(hb.x, hb.y), make_uint2(lb.x, lb.y));
                        r = mul4(r, s_res_inv);
                        if constexpr (FULL) r = make_float4(has_res ? r.x : 0.f, has_res ? r.y : 0.f, has_res ? r.z : 0.f, has_res ? r.w : 0.f);
                    }
                    const float s4[4] = {sc8[P][h2].x, sc8[P][h2].y, sc8[P][h2].z, sc8[P][h2].w}, t4[4] = {sh8[P][h2].x, sh8[P][h2].y, sh8[P][h2].z, sh8[P][h2].w};
                    const float r4[4] = {r.x, r.y, r.z, r.w};
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = fmaf(a8[h2][e], s4[e], t4[e]) + r4[e];
                        if constexpr (FULL) { o[e] = act_cheap(v); continue; }
                        if (actk == OSA_ACT_RELU) v = fmaxf(v, 0.f);
                        else if (actk == OSA_ACT_LEAKY) v = (v > 0.f) ? v : v * p.slope;
                        else if (actk == OSA_ACT_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
                        else if (actk == OSA_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                        else if (actk == OSA_ACT_TANH) v = tanhf(v);
                        o[e] = v;
                    }
                    if (vok && cok) am = fmaxf(am, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
                    split_f16(make_float4(o[0] * s_out, o[1] * s_out, o[2] * s_out, o[3] * s_out), hq[h2], lq[h2]);
                }
                if (FULL || (vok && cok && !(p.dbg & 32))) {
                    float* ys = yb + (v0 + coff) * p.yCs + (co >> 4) * 16 + ((co & 15) >> 3) * 4;
                    store16(ys, make_uint4(hq[0].x, hq[0].y, hq[1].x, hq[1].y));
                    store16(ys + 8, make_uint4(lq[0].x, lq[0].y, lq[1].x, lq[1].y));
                }
            }
        };
        float4 sc8[NT][2][2], sh8[NT][2][2];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int P = 0; P < 2; ++P)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int co = n0 + (wn * NT + n) * 32 + 16 * P + 8 * hh + 4 * h2;
                    sc8[n][P][h2] = make_float4(osc, osc, osc, osc); sh8[n][P][h2] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (co + 3 < p.Co && p.scale) {
                        sc8[n][P][h2] = *reinterpret_cast<const float4*>(p.scale + co); sh8[n][P][h2] = *reinterpret_cast<const float4*>(p.shift + co);
                        sc8[n][P][h2].x *= osc; sc8[n][P][h2].y *= osc; sc8[n][P][h2].z *= osc; sc8[n][P][h2].w *= osc;
                    }
                }
        auto run = [&](auto F) {
#pragma unroll
            for (int i = 0; i < PD; ++i) load_res8T(F, i, rvb[i]);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                finish8T(F, i, rvb[i % PD], sc8[i % NT], sh8[i % NT]);
                if (i + PD < NI) load_res8T(F, i + PD, rvb[i % PD]);
                OSA_TRACE(21 + i);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (FASTC) { if (fast) run(std::true_type{}); else run(std::false_type{}); }
        else run(std::false_type{});
    }
    if constexpr (!REDIR && OUTS && !XT) {
        float4 sc8[NT][2], sh8[NT][2];
#pragma unroll
        for (int n = 0; n < NT; ++n) bn8(n, sc8[n], sh8[n]);
        auto run = [&](auto F) {
#pragma unroll
            for (int i = 0; i < PD; ++i) load_res8(F, i, rvb[i]);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                finish8(F, i, rvb[i % PD], sc8[i % NT], sh8[i % NT]);
                if (i + PD < NI) load_res8(F, i + PD, rvb[i % PD]);
                OSA_TRACE(21 + i);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (FASTC) { if (fast) run(std::true_type{}); else run(std::false_type{}); }
        else run(std::false_type{});
    }
    if constexpr (!REDIR && !OUTS) {
        auto run = [&](auto F) {
#pragma unroll
            for (int i = 0; i < PD; ++i) load_res(F, i, rvb[i]);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                finish(F, i, rvb[i % PD]);
                if (i + PD < NI) load_res(F, i + PD, rvb[i % PD]);
                OSA_TRACE(21 + i);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (FASTC) { if (fast && !(has_res && (p.act & OSA_RES_SPLIT))) run(std::true_type{}); else run(std::false_type{}); }
        else run(std::false_type{});
    }
    };   // epilogue

    if constexpr (!PIPE) {
        epilogue(reinterpret_cast<float*>(smem));
    } else {
        static_assert(NCLS == 1 && !REDIR && PREC == PREC_F16X3, "PIPE: plain f16x3 convolutions");
        // ---- persistent, LDS-DMA pipelined form ------------------------------------------------------------------
        // The workgroup (NW compute waves + ONE loader wave) walks bricks id = xcd_remap(blockIdx.x) + k * gridDim.x.  Two
        // LDS buffers hold chunk gc and chunk gc + 1 of the running (brick, 16-channel chunk) sequence.  While the compute
        // waves run the taps of chunk gc, the loader wave streams chunk gc + 1 into the other buffer: one
        // `buffer_load_dwordx4 ... lds` per (d, h) row of the brick, straight from the split tensor in HBM/L2 -- no VGPR
        // round trip, no ds_write; a per-row buffer descriptor with num_records = row bytes (0 for rows outside the
        // tensor) makes the hardware zero-fill every out-of-range voxel.  The loader is a wave of its own because vmcnt
        // retires in order: with the DMA in the compute waves' queue every wait for a B operand (an L2 hit) would also
        // wait for the DMA rows in front of it (measured: -14 %).  One barrier per chunk publishes the landed buffer (the
        // loader waits vmcnt(0) first); the epilogue transposes through the buffer that was just consumed and one more
        // barrier keeps the next DMA out of it.  Staging is off the critical path; the epilogue of one workgroup overlaps
        // the taps of the other workgroup on the CU.
        const int G = (int)gridDim.x;
        const int nitems = p.B * p.tilesD * p.tilesH * p.tilesW;
        int item = (int)xcd_remap(blockIdx.x, (unsigned)G);
        int gc = 0;
        if (wave == NW) {
            // ================= loader wave =================
            const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
            const unsigned bufbytes = (unsigned)brickQ * 16u;
            const unsigned rowbytes = (unsigned)(p.Wi * p.xCs) * 4u;
            const long long planebytes = (long long)p.Hi * rowbytes;
            const size_t itembytes = (size_t)p.Di * planebytes;
            const int lw_ = lane >> 2, c4_ = lane & 3;
            const bool lane_in = lw_ < p.LW;
            auto dma_chunk = [&](int b_, int c0, int gd0, int gh0, int gw0, unsigned ldsbuf) {
                if (p.dbg & 1) return;                                   // experiments: no staging (timing only)
                const char* base = reinterpret_cast<const char*>(p.x) + (size_t)b_ * itembytes + (size_t)c0 * 4;
                const unsigned voff = (unsigned)((gw0 + lw_) * p.xCs * 4 + c4_ * 16);    // left of / beyond the row: >= num_records -> 0
                for (int ld = 0; ld < p.LD; ++ld) {
                    const int gd = gd0 + ld;
                    const bool dok = (unsigned)gd < (unsigned)p.Di;
                    const char* rowp = base + (long long)gd * planebytes + (long long)gh0 * rowbytes;
                    unsigned ldsrow = ldsbuf + (unsigned)(ld * p.PlaneQ) * 16u;
                    for (int lh = 0; lh < p.LH; ++lh, rowp += rowbytes, ldsrow += (unsigned)p.RowQ * 16u) {
                        const bool ok = dok && ((unsigned)(gh0 + lh) < (unsigned)p.Hi);
                        const unsigned long long rp = (unsigned long long)rowp;
                        u32x4 srd;
                        srd.x = __builtin_amdgcn_readfirstlane((unsigned)rp);
                        srd.y = __builtin_amdgcn_readfirstlane((unsigned)(rp >> 32));
                        srd.z = __builtin_amdgcn_readfirstlane(ok ? rowbytes : 0u);
                        srd.w = 0x00020000u;
                        const unsigned m0v = __builtin_amdgcn_readfirstlane(ldsrow);
                        if (lane_in) {
                            unsigned keep;
                            asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                                         "buffer_load_dwordx4 %2, %1, 0 offen lds\n\ts_mov_b32 m0, %0"
                                         : "=&s"(keep) : "s"(srd), "v"(voff), "s"(m0v) : "memory");
                        }
                    }
                }
            };
            auto land = [&]() {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every row of the chunk has landed in LDS ...
                __syncthreads();                                        // ... hand it to the compute waves (their chunk-end barrier)
            };
            if (item < nitems) {
                int b_, td_, th_, tw_;
                decode_item(item, b_, td_, th_, tw_);
                set_brick(b_, td_, th_, tw_);
                dma_chunk(b, 0, g0d, g0h, g0w, lds0);
            }
            land();
            while (item < nitems) {
                const int nitem = item + G;
                for (int ch = 0; ch < p.nchunks; ++ch, ++gc) {
                    const unsigned nbuf = lds0 + (unsigned)((gc + 1) & 1) * bufbytes;
                    if (ch + 1 < p.nchunks) dma_chunk(b, (ch + 1) * CC, g0d, g0h, g0w, nbuf);
                    else if (nitem < nitems) {
                        int b_, td_, th_, tw_;
                        decode_item(nitem, b_, td_, th_, tw_);
                        dma_chunk(b_, 0, td_ * TD * p.isd + p.dmin, th_ * TH * p.ish + p.hmin, tw_ * TW * p.isw + p.wmin, nbuf);
                    }
                    land();
                }
                __syncthreads();                                        // (the compute waves' post-epilogue barrier)
                item = nitem;
                if (item < nitems) {
                    int b_, td_, th_, tw_;
                    decode_item(item, b_, td_, th_, tw_);
                    set_brick(b_, td_, th_, tw_);
                }
            }
        } else {
            // ================= compute waves =================
            if (item < nitems) {
                int b_, td_, th_, tw_;
                decode_item(item, b_, td_, th_, tw_);
                set_brick(b_, td_, th_, tw_);
            }
            __syncthreads();                                            // chunk 0 of the first brick has landed
            while (item < nitems) {
                zero_acc();
                init_b();
                for (int ch = 0; ch < p.nchunks; ++ch, ++gc) {
                    sm = smem + (size_t)(gc & 1) * brickQ;
                    if (!(p.dbg & 16)) chunk_taps([]() {});             // (experiments: dbg 16 = no taps, timing only)
                    __syncthreads();                                    // chunk gc is consumed, chunk gc + 1 has landed
                }
                epilogue(reinterpret_cast<float*>(smem + (size_t)((gc - 1) & 1) * brickQ));
                __syncthreads();                                        // the epilogue's transpose tiles are free again
                item += G;
                if (item < nitems) {
                    int b_, td_, th_, tw_;
                    decode_item(item, b_, td_, th_, tw_);
                    set_brick(b_, td_, th_, tw_);
                }
            }
        }
    }
    // ---- publish max |output| of this wave into the output's range block
    if (p.out_meta) publish_amax(p.out_meta, am, amax_seen, reinterpret_cast<float*>(smem));   // (barrier inside: every wave is past its tiles)
}

}  // namespace osa
