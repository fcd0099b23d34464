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
#include "osa_common.h"
#include <cstdlib>
#include <cstring>

namespace osa {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// Arithmetic modes of the implicit GEMM:
//   PREC_F32   v_mfma_f32_32x32x2_f32, exact fp32 products (157 TFLOP/s peak)
//   PREC_F16X3 every fp32 operand x is split x = hi + lo (two fp16, 22 significant bits) and
//              A.B ~= Ahi.Bhi + Ahi.Blo + Alo.Bhi on v_mfma_f32_32x32x16_f16 with fp32 accumulation:
//              3 MFMAs of 32 cycles per K=16 instead of 8 of 64 -> 5.3x less matrix-pipe time at
//              fp32-class accuracy (dropped term Alo.Blo ~ 2^-22 relative).  Weights are pre-scaled
//              by a power of two into the fp16 normal range (undone exactly in the epilogue);
//              activations are saturated to +-65504 when split.
enum { PREC_F32 = 0, PREC_F16X3 = 1 };

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
    int VQ;                       // LDS voxel stride in 16-byte slots: 4 (compact) or 5 (padded), see finish_geometry
    // fused 1x1x1 "redir" branch of a transposed conv (GwcNet hourglass: relu(conv6(c5) + redir1(x))):
    // rx = NDHWC tensor at OUTPUT resolution (<= 32 channels), rw = its packed 1x1x1 weights (same packing,
    // T = 1), rscale / rshift = its folded BN, roscale = its f16x3 output scale.  NULL rx = not fused.
    const float* rx; const float4* rw; const float* rscale; const float* rshift; float roscale; int rxCs, rCi;
    unsigned magicW, magicHW;     // ceil(2^32/LW), ceil(2^32/(LH*LW)) : exact for operands < 2^16
    unsigned magicH;              // ceil(2^32/LH)
    int dma;                      // 1: split input + compact LDS image -> stage rows by LDS-DMA (global_load_lds_dwordx4)
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
typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));

// x = hi + lo with hi, lo fp16.  hi uses the packed round-toward-zero convert (2 floats per
// instruction): any rounding is fine for hi because lo = x - float(hi) is exact in fp32 and carries
// the remainder; lo is rounded to nearest, error <= 2^-12 |lo| <= 2^-22 |x|.
// No saturation: operands are brought into range by the per-tensor power-of-two scale below (pow2_scale);
// a value that still exceeds the fp16 range becomes inf and poisons the result visibly instead of being
// clamped silently.
__device__ __forceinline__ void split_f16(const float4 v, uint2& hi, uint2& lo) {
    const float x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
    const h16x2 h01 = __builtin_amdgcn_cvt_pkrtz(x0, x1), h23 = __builtin_amdgcn_cvt_pkrtz(x2, x3);
    // lo is rounded to nearest (unbiased): its error is what remains of the split
    const f16x4 l = {(_Float16)(x0 - (float)h01[0]), (_Float16)(x1 - (float)h01[1]),
                     (_Float16)(x2 - (float)h23[0]), (_Float16)(x3 - (float)h23[1])};
    hi = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
    lo = __builtin_bit_cast(uint2, l);
}

// Power-of-two scale s with amax * s in [2^14, 2^15): the largest operand sits one binade under the fp16
// maximum and every element down to 2^-18 * amax keeps a NORMAL lo half (22 significant bits); smaller
// elements degrade gracefully (absolute error <= 2^-25 / s, i.e. 2^-39 * amax).  Exact to undo (1 / s).
// amax == 0, denormal, inf or NaN: unscaled.
__device__ __forceinline__ float pow2_scale(float amax) {
    const unsigned b = __builtin_bit_cast(unsigned, amax);
    const int eb = (int)((b >> 23) & 0xffu);
    if (eb == 0 || eb == 255) return 1.f;
    int k = 15 - (eb - 126);
    k = k < -60 ? -60 : (k > 60 ? 60 : k);
    return __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
}
__device__ __forceinline__ float4 mul4(const float4 v, const float s) { return make_float4(v.x * s, v.y * s, v.z * s, v.w * s); }

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

template <int NTHR, int PREC, int NCL>
__device__ __forceinline__ void stage_brick(const ConvArgs& p, float4* smem, int brickQ, int b, int c0,
                                            int g0d, int g0h, int g0w, int tid, float s_in = 1.f) {
#ifndef OSA_STAGE_U
#define OSA_STAGE_U 4
#endif
    constexpr int U = (NCL == 1) ? OSA_STAGE_U : OSA_STAGE_U / 2;
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
                    if constexpr (PREC == PREC_F32) {
                        dst[lo[u]] = v[u][cl];
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

// CFG: MT m-tiles x NT n-tiles per wave, WM x WN waves, brick TD x TH x TW (TD derived)
// NCLS = 1: ordinary (strided / dilated / 1x1x1) convolution.
// NCLS = 8: stride-2 transposed convolution, all 8 output-parity classes in one launch: the input
//           brick is staged once, every class has its own accumulator set and its own run of taps
//           (class-major tap order, one linear B stream), outputs go to o = 2a + parity.
template <int PREC, int NCLS, int TU, int MT, int NT, int WM, int WN, int TH, int TW, int REDIR = 0, int OUTS = 0>
// OUTS = 1: the output is a split tensor (OSA_OUT_SPLIT) -- separate instantiation: a lane finalises 8
// channels of 2 voxels (16-byte hi and lo stores) instead of 4 channels of 4 voxels.
// Registers: the fused transposed convs need 2 waves per SIMD; the 256-voxel x 32-channel tiles
// (MT = 2, NT = 1: the dominant 32 -> 32 layers) are held to 128 registers so that 4 workgroups
// share a CU now that their compact LDS brick is 39 KB (measured +7 % on those layers; the same
// limit costs the 64-channel tiles 5 %, so they keep the default).
#define OSA_MIN_BLOCKS ((NCLS >= 4) ? 2 : ((MT == 2 && NT == 1 && WM * WN == 4) ? 4 : 1))
__global__ __launch_bounds__(WM * WN * 64, OSA_MIN_BLOCKS) void conv_mfma_kernel(const ConvArgs p) {
    constexpr int NW = WM * WN;
    constexpr int TD = WM * MT * 32 / (TH * TW);
    static_assert(TD * TH * TW == WM * MT * 32, "brick must hold WM*MT*32 voxels");
    static_assert((TW & (TW - 1)) == 0 && (TH & (TH - 1)) == 0, "TH/TW powers of two");
    extern __shared__ __attribute__((aligned(16))) float4 smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int col = lane & 31, hh = lane >> 5;

    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int twi = bid % p.tilesW; bid /= p.tilesW;
    const int thi = bid % p.tilesH; bid /= p.tilesH;
    const int tdi = bid % p.tilesD;
    const int b = bid / p.tilesD;
    const int a0d = tdi * TD, a0h = thi * TH, a0w = twi * TW;
    const int g0d = a0d * p.isd + p.dmin, g0h = a0h * p.ish + p.hmin, g0w = a0w * p.isw + p.wmin;
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
    const float osc = p.oscale * (1.0f / s_in);      // undoes the weight pre-scale and the input scale (exact)
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
#pragma unroll
    for (int c = 0; c < NCLS; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][m][n][r] = 0.f;

    // One linear stream of B operands: [chunk][tap][octet] "tap steps" of JO*2*CoP float4 each.
    // Copy-free software pipeline: two static register sets (0/1) ping-pong.  Each half-iteration
    // first requests the A (LDS) and B (global/L2) operands of the NEXT group of up to TU taps into
    // the other set, then issues the MFMAs of the current set (sched_barrier pins that order).  A
    // half-iteration with cnt == 0 only prefetches, so every tap run (chunk, parity class) ends with
    // "set 0 holds the next group" and no register rotation is ever needed.  The packed buffer
    // carries a few tap steps of slack for the last prefetch.
    const size_t bstep = (size_t)2 * p.CoP;          // float4s per octet
    const size_t tstep = (p.dbg & 4) ? 0 : (size_t)JO * bstep;   // float4s per tap (dbg 4: stationary B stream, timing only)
    const float4* wp = p.w + (size_t)hh * p.CoP + n0 + wn * (NT * 32) + col;
    constexpr bool RING3 = (TU == 3);     // TU == 3 selects the 3-deep B ring (taps % 3 == 0, NCLS == 1)
    constexpr int TUA = RING3 ? 1 : TU;
    static_assert(!RING3 || NCLS == 1, "the B ring needs tap runs that are multiples of 3");
    float4 B0[TUA][JO][NT], B1[TUA][JO][NT], B2[TUA][JO][NT], A0[TUA][JO][MT], A1[TUA][JO][MT];
#pragma unroll
    for (int u = 0; u < TUA; ++u)
#pragma unroll
        for (int j = 0; j < JO; ++j)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                B0[u][j][n] = wp[u * tstep + j * bstep + n * 32];
                if constexpr (RING3) B1[u][j][n] = wp[tstep + j * bstep + n * 32];
            }

    const int brickQ = p.LD * p.PlaneQ;          // float4s per staged chunk
    const int Tm1 = p.T - 1;
    const float4* sm = smem;
    int t = 0;
    // tap-offset table in one VGPR (lane i holds toff[i], T <= 64): v_readlane instead of a scalar
    // memory load + lgkmcnt wait in front of every tap's LDS reads
    const int toff_v = p.toff[(lane < p.T) ? lane : 0];

    // prefetch group starting at flat tap `tn` (B: `skip` tap steps ahead of wp) into (An, Bn)
    auto prefetch = [&](float4 (&An)[TUA][JO][MT], float4 (&Bn)[TUA][JO][NT], int tn, int skip) {
#pragma unroll
        for (int u = 0; u < TUA; ++u)
#pragma unroll
            for (int j = 0; j < JO; ++j)
#pragma unroll
                for (int n = 0; n < NT; ++n) Bn[u][j][n] = wp[(size_t)(skip + u) * tstep + j * bstep + n * 32];
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
                } else {
                    // [0] = hi halves, [1] = lo halves of the 16 channels of this chunk (K = 16 per MFMA);
                    // small cross terms first, then hi.hi
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            const f16x8 ah = __builtin_bit_cast(f16x8, Ac[u][0][m]), al = __builtin_bit_cast(f16x8, Ac[u][1][m]);
                            const f16x8 bh = __builtin_bit_cast(f16x8, Bc[u][0][n]), bl = __builtin_bit_cast(f16x8, Bc[u][1][n]);
                            ac[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, ac[m][n], 0, 0, 0);
                            ac[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, ac[m][n], 0, 0, 0);
                            ac[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, ac[m][n], 0, 0, 0);
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

    for (int ch0 = 0; ch0 < p.nchunks; ch0 += p.cps) {
        if (ch0) __syncthreads();
        const int ncl = (p.nchunks - ch0 < p.cps) ? (p.nchunks - ch0) : p.cps;
        if (!(p.dbg & 1) && PREC == PREC_F16X3 && p.dma) {
            for (int cl = 0; cl < ncl; ++cl)
                stage_brick_dma<NW * 64>(p, smem + cl * brickQ, b, (ch0 + cl) * CC, g0d, g0h, g0w, tid);
        } else if (!(p.dbg & 1)) {
            int cl = 0;
            for (; cl + 2 <= ncl; cl += 2)
                stage_brick<NW * 64, PREC, 2>(p, smem + cl * brickQ, brickQ, b, (ch0 + cl) * CC, g0d, g0h, g0w, tid, s_in);
            if (cl < ncl)
                stage_brick<NW * 64, PREC, 1>(p, smem + cl * brickQ, brickQ, b, (ch0 + cl) * CC, g0d, g0h, g0w, tid, s_in);
        }
        __syncthreads();
        for (int cl = 0; cl < ncl; ++cl) {
            sm = smem + cl * brickQ;
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
            if constexpr (RING3) {
                // invariant at the top: A0 = tap t, B0 = tap t, B1 = tap t+1 (stream positions wp, wp+1).
                // Three steps are one full turn of the B ring, so leaving after the first triple keeps
                // the invariant for the next chunk (whose A0 is reloaded anyway).
                for (; t < p.T; t += 6) {
                    ring_step(A1, t + 1, B2, 2, A0, B0);
                    ring_step(A0, t + 2, B0, 3, A1, B1);
                    ring_step(A1, t + 3, B1, 4, A0, B2);
                    if (t + 3 >= p.T) { wp += (size_t)3 * tstep; break; }
                    ring_step(A0, t + 4, B2, 5, A1, B0);
                    ring_step(A1, t + 5, B0, 6, A0, B1);
                    ring_step(A0, t + 6, B1, 7, A1, B2);
                    wp += (size_t)6 * tstep;
                }
            } else {
#pragma unroll
            for (int c = 0; c < NCLS; ++c) {
                const int tend = (NCLS == 1) ? p.T : p.cls_end[c];
                while (t < tend) {
                    const int cnt0 = (tend - t < TU) ? (tend - t) : TU;
                    prefetch(A1, B1, t + cnt0, cnt0);
                    __builtin_amdgcn_sched_barrier(0);
                    compute(A0, B0, acc[c], cnt0);
                    __builtin_amdgcn_sched_barrier(0);
                    wp += (size_t)cnt0 * tstep; t += cnt0;
                    const int cnt1 = (tend - t < TU) ? (tend - t) : TU;     // 0 when the run had an odd number of groups
                    prefetch(A0, B0, t + cnt1, cnt1);
                    __builtin_amdgcn_sched_barrier(0);
                    compute(A1, B1, acc[c], cnt1);
                    __builtin_amdgcn_sched_barrier(0);
                    wp += (size_t)cnt1 * tstep; t += cnt1;
                }
            }
            }
        }
    }

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
    const unsigned amax_seen = p.out_meta ? amax_peek(p.out_meta) : 0u;   // early: its latency hides behind the epilogue
    __syncthreads();                                   // everyone is done reading the input brick
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
    float* tb = reinterpret_cast<float*>(smem) + wave * (32 * 36);
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
    // A wave finalises NI = MT*NCLS*NT tiles of 32 voxels x 32 channels one after the other.  The
    // residual rows of tile i+PD are requested before tile i is processed (rolling window of PD
    // tiles, static register sets), so the HBM round trip of a residual overlaps the LDS transposes,
    // arithmetic and stores of the PD-1 tiles in front of it -- the fused transposed conv has 8 tiles
    // per wave and spent half of its time waiting for them one by one.
    constexpr int NI = MT * NCLS * NT;
    constexpr int PD = REDIR ? 2 : ((NCLS >= 4) ? 3 : ((NI < 2) ? NI : 2));
    // voxel bookkeeping of the 4 rows (vsub + 8k) this lane finalises in M tile m
    auto rows_of = [&](int m, int (&v0)[4], int (&g0)[4], bool (&vok)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = (wm * MT + m) * 32 + vsub + 8 * k;
            const int ad = a0d + q / (TW * TH), ah = a0h + (q / TW) % TH, aw = a0w + q % TW;
            vok[k] = ad < p.Ad && ah < p.Ah && aw < p.Aw;
            v0[k] = ((ad * p.os) * p.Ho + ah * p.os) * p.Wo + aw * p.os;      // voxel index inside batch item b (host: < 2^31 elements)
            g0[k] = (ah * p.os) * p.Wo + aw * p.os;
        }
    };
    auto class_off = [&](int c, int& coff, int& goff) {
        const int ood = (NCLS == 1) ? p.ood : ((c >> 2) & 1), ooh = (NCLS == 1) ? p.ooh : ((c >> 1) & 1),
                  oow = (NCLS == 1) ? p.oow : (c & 1);
        coff = (ood * p.Ho + ooh) * p.Wo + oow;              // supported transposed convs: Do == 2*Di
        goff = ooh * p.Wo + oow;
    };
    // tile order: m outer, class, n inner
    auto load_res = [&](int i, float4 (&rv)[4]) {
        const int n = i % NT, c = (i / NT) % NCLS, m = i / (NT * NCLS);
        int v0[4], g0[4], coff, goff; bool vok[4];
        rows_of(m, v0, g0, vok);
        class_off(c, coff, goff);
        const int co = n0 + (wn * NT + n) * 32 + cq;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            rv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.res && vok[k] && co < p.Co) {
                const float* rp = resb + (v0[k] + coff) * p.rCs + co;
                if (PREC == PREC_F16X3 && (p.act & OSA_RES_SPLIT)) {
                    const float* rs = resb + (v0[k] + coff) * p.rCs;
                    const uint2 h = *reinterpret_cast<const uint2*>(rs + split_off_hi(co));
                    const uint2 l = *reinterpret_cast<const uint2*>(rs + split_off_lo(co));
                    rv[k] = __builtin_bit_cast(float4, make_uint4(h.x, h.y, l.x, l.y));     // decoded in finish()
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
    auto finish = [&](int i, const float4 (&rv)[4]) {
        const int n = i % NT, c = (i / NT) % NCLS, m = i / (NT * NCLS);
        int v0[4], g0[4], coff, goff; bool vok[4];
        rows_of(m, v0, g0, vok);
        class_off(c, coff, goff);
        // registers -> LDS (tile[voxel row][channel])
#pragma unroll
        for (int r = 0; r < 16; ++r)
            tb[((r & 3) + 8 * (r >> 2) + 4 * hh) * 36 + col] = acc[c][m][n][r];
        const int co = n0 + (wn * NT + n) * 32 + cq;
        const bool cok = co < p.Co;
        const float4 sc = scv[n], sh = shv[n];
        // LDS -> registers (4 voxels x 4 channels per lane); gate rows requested together
        float4 av[4], gv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            av[k] = *reinterpret_cast<const float4*>(tb + (vsub + 8 * k) * 36 + cq);
            gv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!REDIR && p.gate && vok[k] && cok) {
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
            if (PREC == PREC_F16X3 && (p.act & OSA_RES_SPLIT) && p.res) {
                const uint4 b4 = __builtin_bit_cast(uint4, rv[k]);
                rk = mul4(join_f16(make_uint2(b4.x, b4.y), make_uint2(b4.z, b4.w)), s_res_inv);
            }
            const float r4[4] = {rk.x, rk.y, rk.z, rk.w}, g4[4] = {gv[k].x, gv[k].y, gv[k].z, gv[k].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = fmaf(a4[e], s4[e], h4[e]) + r4[e];
                if (actk == OSA_ACT_RELU) v = fmaxf(v, 0.f);
                else if (actk == OSA_ACT_LEAKY) v = (v > 0.f) ? v : v * p.slope;
                else if (actk == OSA_ACT_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
                else if (actk == OSA_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                else if (actk == OSA_ACT_TANH) v = tanhf(v);
                if (!REDIR && p.gate) v *= gate_raw ? g4[e] : 1.0f / (1.0f + expf(-g4[e]));
                o[e] = v;
            }
            if (vok[k] && cok) {
                am = fmaxf(am, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
                float* yp = yb + (v0[k] + coff) * p.yCs + co;
                if (vec4) *reinterpret_cast<float4*>(yp) = make_float4(o[0], o[1], o[2], o[3]);
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
    auto rows2 = [&](int m, int (&v0)[2], bool (&vok)[2]) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int q = (wm * MT + m) * 32 + vs2 + 16 * k;
            const int ad = a0d + q / (TW * TH), ah = a0h + (q / TW) % TH, aw = a0w + q % TW;
            vok[k] = ad < p.Ad && ah < p.Ah && aw < p.Aw;
            v0[k] = ((ad * p.os) * p.Ho + ah * p.os) * p.Wo + aw * p.os;
        }
    };
    auto load_res8 = [&](int i, float4 (&rv)[4]) {
        const int n = i % NT, c = (i / NT) % NCLS, m = i / (NT * NCLS);
        int v0[2], coff, goff; bool vok[2];
        rows2(m, v0, vok);
        class_off(c, coff, goff);
        const int co = n0 + (wn * NT + n) * 32 + c8;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            rv[2 * k] = make_float4(0.f, 0.f, 0.f, 0.f); rv[2 * k + 1] = rv[2 * k];
            if (p.res && vok[k] && co < p.Co) {
                const float* rs = resb + (v0[k] + coff) * p.rCs + (co >> 4) * 16 + ((co & 15) >> 3) * 4;
                rv[2 * k] = *reinterpret_cast<const float4*>(rs);            // 8 hi halves
                rv[2 * k + 1] = *reinterpret_cast<const float4*>(rs + 8);    // 8 lo halves
            }
        }
    };
    auto finish8 = [&](int i, const float4 (&rv)[4], const float4 (&sc8)[2], const float4 (&sh8)[2]) {
        const int n = i % NT, c = (i / NT) % NCLS, m = i / (NT * NCLS);
        int v0[2], coff, goff; bool vok[2];
        rows2(m, v0, vok);
        class_off(c, coff, goff);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            tb[((r & 3) + 8 * (r >> 2) + 4 * hh) * 36 + col] = acc[c][m][n][r];
        const int co = n0 + (wn * NT + n) * 32 + c8;
        const bool cok = co < p.Co;
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
                if (p.res) {
                    const uint4 hb = __builtin_bit_cast(uint4, rv[2 * k]), lb = __builtin_bit_cast(uint4, rv[2 * k + 1]);
                    r = h2 ? join_f16(make_uint2(hb.z, hb.w), make_uint2(lb.z, lb.w)) : join_f16(make_uint2(hb.x, hb.y), make_uint2(lb.x, lb.y));
                    r = mul4(r, s_res_inv);
                }
                const float a4[4] = {av[k][h2].x, av[k][h2].y, av[k][h2].z, av[k][h2].w};
                const float s4[4] = {sc8[h2].x, sc8[h2].y, sc8[h2].z, sc8[h2].w}, t4[4] = {sh8[h2].x, sh8[h2].y, sh8[h2].z, sh8[h2].w};
                const float r4[4] = {r.x, r.y, r.z, r.w};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = fmaf(a4[e], s4[e], t4[e]) + r4[e];
                    if (actk == OSA_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (actk == OSA_ACT_LEAKY) v = (v > 0.f) ? v : v * p.slope;
                    else if (actk == OSA_ACT_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
                    else if (actk == OSA_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                    else if (actk == OSA_ACT_TANH) v = tanhf(v);
                    o[e] = v;
                }
                if (vok[k] && cok) am = fmaxf(am, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
                split_f16(make_float4(o[0] * s_out, o[1] * s_out, o[2] * s_out, o[3] * s_out), hq[h2], lq[h2]);
            }
            if (vok[k] && cok) {
                float* ys = yb + (v0[k] + coff) * p.yCs + (co >> 4) * 16 + ((co & 15) >> 3) * 4;
                *reinterpret_cast<uint4*>(ys) = make_uint4(hq[0].x, hq[0].y, hq[1].x, hq[1].y);
                *reinterpret_cast<uint4*>(ys + 8) = make_uint4(lq[0].x, lq[0].y, lq[1].x, lq[1].y);
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
            // x rows of tile i in MFMA A-operand order: lane (col, hh) -> voxel row `col`
            auto load_x = [&](int i, float4 (&rv)[RV]) {
                const int c = (i / NT) % NCLS, m = i / (NT * NCLS);
                const int q = (wm * MT + m) * 32 + col;
                const int ad = a0d + q / (TW * TH), ah = a0h + (q / TW) % TH, aw = a0w + q % TW;
                const bool ok = ad < p.Ad && ah < p.Ah && aw < p.Aw;
                const int vox = ((ad * 2 + ((c >> 2) & 1)) * p.Ho + ah * 2 + ((c >> 1) & 1)) * p.Wo + aw * 2 + (c & 1);
#pragma unroll
                for (int k = 0; k < RV; ++k) {
                    rv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    const int ch = k >> 1, j = k & 1;
                    // f16x3: 8 consecutive channels 8hh..8hh+7 of the chunk (two float4s); f32: channels 8j+4hh..+3
                    int cin = ch * CC + ((PREC == PREC_F32) ? (8 * j + 4 * hh) : (8 * hh + 4 * j));
                    // split redir input: j = 0 -> the lane's 8 hi halves, j = 1 -> its 8 lo halves (16 B each)
                    if (PREC == PREC_F16X3 && (p.act & OSA_REDIR_SPLIT)) cin = ch * CC + 4 * hh + 8 * j;
                    if (ok && ch < rch && ch * CC < p.rCi) rv[k] = *reinterpret_cast<const float4*>(rxb + vox * p.rxCs + cin);
                }
            };
            auto add_redir = [&](int i, const float4 (&rv)[RV]) {
                const int n = i % NT, c = (i / NT) % NCLS, m = i / (NT * NCLS);
                const float4* rwp = p.rw + (size_t)hh * p.CoP + n0 + (wn * NT + n) * 32 + col;
                f32x16 r;
#pragma unroll
                for (int e = 0; e < 16; ++e) r[e] = 0.f;
#pragma unroll
                for (int ch = 0; ch < RCH; ++ch) {
                    if (ch < rch) {
                        const float4 b0 = rwp[ch * rtstep], b1 = rwp[ch * rtstep + rbstep];
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
                // per-lane (channel `col`) BN factors of both branches
                const int cl = n0 + (wn * NT + n) * 32 + col;
                const bool lok = cl < p.Co;
                const float s6 = (lok && p.scale) ? p.scale[cl] * osc : osc, t6 = (lok && p.shift) ? p.shift[cl] : 0.f;
                const float sr = (lok && p.rscale) ? p.rscale[cl] * rosc : rosc, tr = (lok && p.rshift) ? p.rshift[cl] : 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[c][m][n][e] = fmaf(acc[c][m][n][e], s6, t6) + fmaf(r[e], sr, tr);
            };
#pragma unroll
            for (int n = 0; n < NT; ++n) { scv[n] = make_float4(1.f, 1.f, 1.f, 1.f); shv[n] = make_float4(0.f, 0.f, 0.f, 0.f); }
            const float4 zero4[4] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f),
                                     make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
#pragma unroll
            for (int i = 0; i < PD; ++i) load_x(i, rvb[i]);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                add_redir(i, rvb[i % PD]);
                if (i + PD < NI) load_x(i + PD, rvb[i % PD]);
                if constexpr (OUTS) {
                    float4 sc8[2], sh8[2];
                    bn8(i % NT, sc8, sh8);
                    finish8(i, zero4, sc8, sh8);
                } else finish(i, zero4);
            }
        }
    }
    if constexpr (!REDIR && OUTS) {
        float4 sc8[NT][2], sh8[NT][2];
#pragma unroll
        for (int n = 0; n < NT; ++n) bn8(n, sc8[n], sh8[n]);
#pragma unroll
        for (int i = 0; i < PD; ++i) load_res8(i, rvb[i]);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            finish8(i, rvb[i % PD], sc8[i % NT], sh8[i % NT]);
            if (i + PD < NI) load_res8(i + PD, rvb[i % PD]);
        }
    }
    if constexpr (!REDIR && !OUTS) {
#pragma unroll
        for (int i = 0; i < PD; ++i) load_res(i, rvb[i]);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            finish(i, rvb[i % PD]);
            if (i + PD < NI) load_res(i + PD, rvb[i % PD]);
        }
    }
    // ---- publish max |output| of this wave into the output's range block
    if (p.out_meta) publish_amax(p.out_meta, am, amax_seen, reinterpret_cast<float*>(smem));   // (barrier inside: every wave is past its tiles)
}

// ------------------------------------------------------------------ dispatch --
struct KernelCfg {
    const char* name;
    int M, N;              // voxels / channels per workgroup
    int TD, TH, TW, threads;
    void (*fn[2])(const ConvArgs);      // [PREC_F32], [PREC_F16X3]
    void (*fn3[2])(const ConvArgs);     // same, B-ring pipeline (tap count a multiple of 3); may be null
    void (*fns[2])(const ConvArgs);     // f16x3 with split output (OUTS = 1): [ping-pong], [B ring or null]
};

constexpr int TAPS_PER_ITER = 1;   // taps per half-iteration of the ping-pong pipeline
#define OSA_CFG(MT, NT, WM, WN, TH, TW)                                                      \
    { #MT "x" #NT "_" #WM "x" #WN "_" #TH "x" #TW, WM * MT * 32, WN * NT * 32,               \
      WM * MT * 32 / (TH * TW), TH, TW, WM * WN * 64,                                        \
      { conv_mfma_kernel<PREC_F32, 1, TAPS_PER_ITER, MT, NT, WM, WN, TH, TW>,                 \
        conv_mfma_kernel<PREC_F16X3, 1, TAPS_PER_ITER, MT, NT, WM, WN, TH, TW> },             \
      { conv_mfma_kernel<PREC_F32, 1, 3, MT, NT, WM, WN, TH, TW>,                             \
        conv_mfma_kernel<PREC_F16X3, 1, 3, MT, NT, WM, WN, TH, TW> },                         \
      { conv_mfma_kernel<PREC_F16X3, 1, TAPS_PER_ITER, MT, NT, WM, WN, TH, TW, 0, 1>,         \
        conv_mfma_kernel<PREC_F16X3, 1, 3, MT, NT, WM, WN, TH, TW, 0, 1> } }

// same tile without the B ring: at 4 waves per SIMD (128 registers) the ring version of the
// 256-voxel x 32-channel tile spills, the ping-pong version (116 registers) does not
#define OSA_CFG_PINGPONG(MT, NT, WM, WN, TH, TW)                                             \
    { #MT "x" #NT "_" #WM "x" #WN "_" #TH "x" #TW, WM * MT * 32, WN * NT * 32,               \
      WM * MT * 32 / (TH * TW), TH, TW, WM * WN * 64,                                        \
      { conv_mfma_kernel<PREC_F32, 1, TAPS_PER_ITER, MT, NT, WM, WN, TH, TW>,                 \
        conv_mfma_kernel<PREC_F16X3, 1, TAPS_PER_ITER, MT, NT, WM, WN, TH, TW> },             \
      { nullptr, nullptr },                                                                  \
      { conv_mfma_kernel<PREC_F16X3, 1, TAPS_PER_ITER, MT, NT, WM, WN, TH, TW, 0, 1>, nullptr } }

static const KernelCfg g_cfgs[] = {
    OSA_CFG_PINGPONG(2, 1, 4, 1, 8, 8),   // 0: 256 vox x  32 ch   brick 4x8x8
    OSA_CFG(2, 2, 4, 1, 8, 8),   // 1: 256 vox x  64 ch   brick 4x8x8
    OSA_CFG(2, 2, 2, 2, 8, 8),   // 2: 128 vox x 128 ch   brick 2x8x8
    OSA_CFG(1, 1, 4, 1, 4, 8),   // 3: 128 vox x  32 ch   brick 4x4x8
    OSA_CFG(1, 2, 4, 1, 4, 8),   // 4: 128 vox x  64 ch   brick 4x4x8
    OSA_CFG(1, 1, 2, 2, 4, 8),   // 5:  64 vox x  64 ch   brick 2x4x8   (stride-2 layers)
    OSA_CFG(1, 2, 2, 2, 4, 8),   // 6:  64 vox x 128 ch   brick 2x4x8
    OSA_CFG_PINGPONG(2, 1, 4, 1, 16, 16), // 7: 256 vox x  32 ch   brick 1x16x16 (2-D layers)
    OSA_CFG(2, 2, 4, 1, 16, 16), // 8: 256 vox x  64 ch   brick 1x16x16
    OSA_CFG(2, 2, 2, 2, 8, 16),  // 9: 128 vox x 128 ch   brick 1x8x16
    OSA_CFG(4, 1, 4, 1, 8, 8),   // 10: 512 vox x 32 ch   brick 8x8x8
    OSA_CFG(1, 1, 1, 4, 4, 8),   // 11: 32 vox x 128 ch   brick 1x4x8   (tiny layers: more workgroups)
    OSA_CFG(1, 1, 4, 1, 8, 16),  // 12: 128 vox x 32 ch   brick 1x8x16  (2-D layers, more workgroups)
    OSA_CFG(1, 2, 4, 1, 8, 16),  // 13: 128 vox x 64 ch   brick 1x8x16
    OSA_CFG(1, 1, 2, 2, 4, 16),  // 14:  64 vox x 64 ch   brick 1x4x16  (2-D stride-2 layers)
    OSA_CFG(1, 1, 2, 1, 4, 8),   // 15:  64 vox x 32 ch   brick 2x4x8   (stride-2 layers with <= 32 outputs; 2 waves)
};
constexpr int N_CFGS = sizeof(g_cfgs) / sizeof(g_cfgs[0]);

// fused transposed conv: 128 input-resolution positions x 32 channels x 8 parity classes per workgroup
static const KernelCfg g_deconv_redir_cfg = {
    "deconv8_redir_1x1_4x1_4x8", 128, 32, 4, 4, 8, 256,
    { conv_mfma_kernel<PREC_F32, 8, 1, 1, 1, 4, 1, 4, 8, 1>, conv_mfma_kernel<PREC_F16X3, 8, 1, 1, 1, 4, 1, 4, 8, 1> },
    { nullptr, nullptr }, { conv_mfma_kernel<PREC_F16X3, 8, 1, 1, 1, 4, 1, 4, 8, 1, 1>, nullptr } };
static const KernelCfg g_deconv_redir64_cfg = {
    "deconv8_redir64_1x1_4x1_4x8", 128, 32, 4, 4, 8, 256,
    { conv_mfma_kernel<PREC_F32, 8, 1, 1, 1, 4, 1, 4, 8, 2>, conv_mfma_kernel<PREC_F16X3, 8, 1, 1, 1, 4, 1, 4, 8, 2> },
    { nullptr, nullptr }, { conv_mfma_kernel<PREC_F16X3, 8, 1, 1, 1, 4, 1, 4, 8, 2, 1>, nullptr } };
static const KernelCfg g_deconv_cfg = {
    "deconv8_1x1_4x1_4x8", 128, 32, 4, 4, 8, 256,
    { conv_mfma_kernel<PREC_F32, 8, 1, 1, 1, 4, 1, 4, 8>, conv_mfma_kernel<PREC_F16X3, 8, 1, 1, 1, 4, 1, 4, 8> },
    { nullptr, nullptr }, { conv_mfma_kernel<PREC_F16X3, 8, 1, 1, 1, 4, 1, 4, 8, 0, 1>, nullptr } };
// fused 2-D transposed conv (D = 1): 128 input-resolution pixels x 32 channels x 4 parity classes
static const KernelCfg g_deconv_flat_cfg = {
    "deconv4_1x1_4x1_8x16", 128, 32, 1, 8, 16, 256,
    { conv_mfma_kernel<PREC_F32, 4, 1, 1, 1, 4, 1, 8, 16>, conv_mfma_kernel<PREC_F16X3, 4, 1, 1, 1, 4, 1, 8, 16> },
    { nullptr, nullptr }, { nullptr, nullptr } };

static int pick_cfg(const ConvArgs& a, int stride) {
    {
        const int v = exp_int("OSA_CONV_CFG", -1);
        if (v >= 0 && v < N_CFGS && a.CoP % g_cfgs[v].N == 0) return v;
    }
    const bool flat = (a.Ad == 1);
    const long long vox = (long long)a.B * a.Ad * a.Ah * a.Aw;
    if (flat) {
        // measured on MI355X (tools/bench_layers.py --set 2d, 1 and 2 pairs per step)
        if (stride == 2) return (a.CoP % 64 == 0) ? 14 : 12;
        if (a.CoP % 128 == 0) return 9;                            // 128 pixels x 128 channels, 2x2 waves
        return (a.CoP % 64 == 0) ? 13 : 12;                        // 128-pixel tiles
    }
    if (stride == 2) return (a.CoP % 128 == 0) ? 6 : ((a.CoP % 64 == 0) ? 5 : 15);
    // measured on MI355X (tools/bench_layers.py): few-tap launches (1x1x1, transposed-conv parity
    // classes) and sub-megavoxel volumes prefer the 128-voxel bricks (more workgroups in flight)
    if (a.T <= 8) return (a.CoP % 64 == 0) ? 4 : 3;
    if (a.CoP % 128 == 0) return 2;                                // 128 voxels x 128 channels, 2x2 waves
    if (vox < (1ll << 20)) return (a.CoP % 64 == 0) ? 4 : 3;
    return (a.CoP % 64 == 0) ? 1 : 0;
}

// LDS image of a staged chunk: one voxel = 64 B of operands (16 fp32, or 16 hi + 16 lo fp16).
// ds_read_b128 is serviced in 16-lane groups that mix the rows of an M tile (4 rows of 8 voxels for
// TW = 8, 2 rows of 16 for TW = 16); a group is conflict free when its 16 addresses fall into 16
// distinct 16-byte slots (mod 256 B).
//  * compact (TW = 8, unit w stride): voxels 4 slots apart -> a row's 4 lanes of a group sit on slots
//    {0,4,8,12} + const, and an ODD row stride moves the 4 rows of the tile onto the 4 residues mod 4:
//    16 distinct slots with one slot of padding per row (39 KB for the 6x10x10 brick -> 4 per CU).
//  * padded (TW = 16, strided or dilated taps): voxels 5 slots apart (conflict free within a row of
//    16), row stride a multiple of 16 slots for TW = 16 and 8 (mod 16) for TW = 8.
static void finish_geometry(ConvArgs& a, int TW, bool compact) {
    if (exp_set("OSA_NOCOMPACT")) compact = false;
    int rowq;
    if (compact) {
        a.VQ = 4;
        rowq = a.LW * 4 + 1;
    } else {
        a.VQ = 5;
        const int want = (TW == 8) ? 8 : 0;              // slots mod 16
        rowq = a.LW * 5;
        while ((rowq & 15) != want) ++rowq;
        if (exp_set("OSA_NOPAD")) rowq = a.LW * 5;
    }
    a.RowQ = rowq; a.PlaneQ = a.LH * a.RowQ;
    for (int t = 0; t < a.T; ++t)
        a.toff[t] = (a.td[t] - a.dmin) * a.PlaneQ + (a.th[t] - a.hmin) * a.RowQ + (a.tw[t] - a.wmin) * a.VQ;
    a.magicW = (unsigned)((0x100000000ull + a.LW - 1) / a.LW);
    a.magicH = (unsigned)((0x100000000ull + a.LH - 1) / a.LH);
    a.magicHW = (unsigned)((0x100000000ull + (unsigned long long)a.LH * a.LW - 1) / ((unsigned long long)a.LH * a.LW));
    a.dbg = exp_int("OSA_DBG", 0);
}

static size_t brick_bytes(ConvArgs& a, const KernelCfg& k) {
    int dmax = -128, hmax = -128, wmax = -128;
    a.dmin = a.hmin = a.wmin = 127;
    for (int t = 0; t < a.T; ++t) {
        a.dmin = a.td[t] < a.dmin ? a.td[t] : a.dmin; dmax = a.td[t] > dmax ? a.td[t] : dmax;
        a.hmin = a.th[t] < a.hmin ? a.th[t] : a.hmin; hmax = a.th[t] > hmax ? a.th[t] : hmax;
        a.wmin = a.tw[t] < a.wmin ? a.tw[t] : a.wmin; wmax = a.tw[t] > wmax ? a.tw[t] : wmax;
    }
    a.LD = (k.TD - 1) * a.isd + (dmax - a.dmin) + 1;
    a.LH = (k.TH - 1) * a.ish + (hmax - a.hmin) + 1;
    a.LW = (k.TW - 1) * a.isw + (wmax - a.wmin) + 1;
    return (size_t)a.LD * a.LH * (a.LW * VS + 64) * sizeof(float);   // upper bound incl. row padding
}

static int launch_conv(ConvArgs& a, int stride, int prec, hipStream_t st, const char* what,
                       const KernelCfg* forced = nullptr) {
    int ci = forced ? 0 : pick_cfg(a, stride);
    if (!forced && brick_bytes(a, g_cfgs[ci]) > 160 * 1024) {
        // e.g. a stride-2 3x3x3 layer whose output depth collapses to 1: fall back to the small bricks
        static const int fallback[] = {5, 15, 3, 11};
        for (int f : fallback)
            if (a.CoP % g_cfgs[f].N == 0 && brick_bytes(a, g_cfgs[f]) <= 160 * 1024) { ci = f; break; }
    }
    const KernelCfg& k = forced ? *forced : g_cfgs[ci];
    a.tilesD = cdiv(a.Ad, k.TD); a.tilesH = cdiv(a.Ah, k.TH); a.tilesW = cdiv(a.Aw, k.TW);
    (void)brick_bytes(a, k);
    OSA_REQUIRE((long long)a.LD * a.LH * a.LW < 65536, "%s: LDS brick too large", what);
    finish_geometry(a, k.TW, k.TW == 8 && a.isw == 1);
    const size_t brick = (size_t)a.LD * a.PlaneQ * sizeof(float4);
    OSA_REQUIRE(brick <= 160 * 1024, "%s: LDS brick %dx%dx%d needs %zu B (> 160 KiB)", what, a.LD, a.LH, a.LW, brick);
    // several channel chunks per staging pass (fewer barriers, more loads in flight) while the
    // workgroup stays under ~40 KiB of LDS, i.e. as long as it does not cost residency
    a.cps = 1;
    {
        const int want = exp_int("OSA_CPS", 4);
        const size_t cap = (size_t)exp_int("OSA_CPS_LDS", 40 * 1024);
        while (a.cps < want && a.cps < a.nchunks && (size_t)(a.cps + 1) * brick <= cap) ++a.cps;
    }
    size_t lds = brick * a.cps;
    const size_t epi = (size_t)(k.threads / 64) * 32 * 36 * sizeof(float);   // wave-private transpose tiles of the epilogue
    if (lds < epi) lds = epi;
    { const size_t m = (size_t)exp_int("OSA_LDS_MIN", 0); if (m > lds) lds = m; }   // experiments: cap residency
    const long long nblk = (long long)a.B * a.tilesD * a.tilesH * a.tilesW;
    OSA_REQUIRE(nblk < (1ll << 31), "%s: grid too large", what);
    {   // the epilogue addresses one batch item with 32-bit element offsets
        const long long ovox = (long long)a.Do * a.Ho * a.Wo;
        const int cs = a.yCs > a.rCs ? (a.yCs > a.gCs ? a.yCs : a.gCs) : (a.rCs > a.gCs ? a.rCs : a.gCs);
        OSA_REQUIRE(ovox * cs < (1ll << 31), "%s: one batch item of the output exceeds 2^31 elements", what);
        OSA_REQUIRE((long long)a.Di * a.Hi * a.Wi * a.xCs < (1ll << 31), "%s: one batch item of the input exceeds 2^31 elements", what);
    }
    if (a.act & (OSA_IN_SPLIT | OSA_OUT_SPLIT | OSA_RES_SPLIT | OSA_REDIR_SPLIT)) {
        OSA_REQUIRE(prec == PREC_F16X3, "%s: split activation tensors exist in the f16x3 mode only", what);
        if (a.act & OSA_IN_SPLIT) OSA_REQUIRE(a.Ci % 16 == 0, "%s: split input needs Ci %% 16 == 0 (got %d)", what, a.Ci);
        if (a.act & OSA_OUT_SPLIT) OSA_REQUIRE(a.Co % 16 == 0 && a.yCs % 16 == 0 && !a.gate && ((size_t)a.y & 15) == 0,
                                               "%s: split output needs Co, yCs %% 16 == 0 and no gate", what);
        if ((a.act & OSA_OUT_SPLIT) && a.res) OSA_REQUIRE(a.act & OSA_RES_SPLIT, "%s: a split output takes a split residual", what);
        if ((a.act & OSA_RES_SPLIT) && a.res) OSA_REQUIRE(a.Co % 16 == 0 && a.rCs % 16 == 0 && ((size_t)a.res & 15) == 0,
                                                          "%s: split residual needs Co, rCs %% 16 == 0", what);
        if ((a.act & OSA_REDIR_SPLIT) && a.rx) OSA_REQUIRE(a.rCi % 16 == 0, "%s: split redir input needs channels %% 16 == 0", what);
    }
    // LDS-DMA staging of split inputs: measured throughput-neutral at 4 workgroups per CU (block-level overlap
    // already hides the staging), so it is opt-in (OSA_DMA=1) until the tap loop is pipelined across barriers
    a.dma = (exp_int("OSA_DMA", 0) && prec == PREC_F16X3 && (a.act & OSA_IN_SPLIT) && a.VQ == 4 && a.LW * 4 <= 64) ? 1 : 0;
    // tap counts that are multiples of 3 (3x3x3, 3x3) run the B-ring pipeline
    const bool no_ring = exp_set("OSA_NORING");
    void (*fn)(const ConvArgs) = (k.fn3[prec] && a.T % 3 == 0 && !no_ring) ? k.fn3[prec] : k.fn[prec];
    if (a.act & OSA_OUT_SPLIT) {
        fn = (k.fns[1] && a.T % 3 == 0 && !no_ring) ? k.fns[1] : k.fns[0];
        OSA_REQUIRE(fn != nullptr, "%s: this tile configuration has no split-output variant", what);
    }
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((unsigned)nblk, a.CoP / k.N), block(k.threads);
    hipLaunchKernelGGL(fn, grid, block, lds, st, a);
    OSA_LAUNCH_CHECK(what);
    return 0;
}

// ------------------------------------------------------------------ packing --
// dst float index: ((((ch*T + t)*JO + j)*2 + h)*CoP + co)*4 + e   <-  W_t[ci = ch*16 + 8j + 4h + e][co]
struct PackArgs {
    const float* src; float* dst;
    int Ci, Co, CoP, kd, kh, kw, T, nchunks, transposed;
    signed char kz[MAX_TAPS], ky[MAX_TAPS], kx[MAX_TAPS];   // kernel index of every tap
};

__global__ __launch_bounds__(256) void pack_weights_kernel(const PackArgs p) {
    const size_t total = (size_t)p.nchunks * p.T * JO * 2 * p.CoP * 4;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int e = i & 3; size_t r = i >> 2;
    const int co = r % p.CoP; r /= p.CoP;
    const int h = r & 1; r >>= 1;
    const int j = r % JO; r /= JO;
    const int t = r % p.T; const int ch = r / p.T;
    const int ci = ch * CC + 8 * j + 4 * h + e;
    float v = 0.f;
    if (ci < p.Ci && co < p.Co) {
        const size_t kvol = (size_t)p.kd * p.kh * p.kw;
        const size_t kidx = ((size_t)p.kz[t] * p.kh + p.ky[t]) * p.kw + p.kx[t];
        v = p.transposed ? p.src[((size_t)ci * p.Co + co) * kvol + kidx]
                         : p.src[((size_t)co * p.Ci + ci) * kvol + kidx];
    }
    p.dst[i] = v;
}

// f16x3 image of the same buffer: 16-byte unit index ((((ch*T + t)*2 + hl)*2 + kg)*CoP + co) holds the 8 fp16
// hi (hl=0) or lo (hl=1) parts of  wscale * W_t[ci = ch*16 + 8*kg + e][co], e = 0..7.
__global__ __launch_bounds__(256) void pack_weights_f16x3_kernel(const PackArgs p, float wscale) {
    const size_t total = (size_t)p.nchunks * p.T * 2 * 2 * p.CoP * 8;      // fp16 elements
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int e = i & 7; size_t r = i >> 3;
    const int co = r % p.CoP; r /= p.CoP;
    const int kg = r & 1; r >>= 1;
    const int hl = r & 1; r >>= 1;
    const int t = r % p.T; const int ch = r / p.T;
    const int ci = ch * CC + 8 * kg + e;
    float v = 0.f;
    if (ci < p.Ci && co < p.Co) {
        const size_t kvol = (size_t)p.kd * p.kh * p.kw;
        const size_t kidx = ((size_t)p.kz[t] * p.kh + p.ky[t]) * p.kw + p.kx[t];
        v = wscale * (p.transposed ? p.src[((size_t)ci * p.Co + co) * kvol + kidx]
                                   : p.src[((size_t)co * p.Ci + ci) * kvol + kidx]);
    }
    const _Float16 hi = (_Float16)v;
    reinterpret_cast<_Float16*>(p.dst)[i] = hl ? (_Float16)(v - (float)hi) : hi;
}

static inline int pad32(int c) { return (c + 31) / 32 * 32; }
static inline int nchunks_of(int ci) { return (ci + CC - 1) / CC; }
static inline size_t packed_floats(int Ci, int Co, int T) {
    return (size_t)nchunks_of(Ci) * T * JO * 2 * pad32(Co) * 4;
}
static inline size_t slack_floats(int Co) { return (size_t)4 * JO * 2 * pad32(Co) * 4; }   // up to 4 prefetched taps

// transposed-conv parity class: taps of one dimension. o = 2a+par ; i = a + delta ; kernel index kk
static int deconv_dim_taps(int k, int pad, int par, int* delta, int* kk) {
    int n = 0;
    for (int t = 0; t < k; ++t) {
        const int num = par + pad - t;
        if (((num % 2) + 2) % 2 != 0) continue;
        delta[n] = (num >= 0) ? num / 2 : -((-num) / 2);
        kk[n] = t;
        ++n;
    }
    return n;
}

// ------------------------------------------------------------------ small Co --
// Classifier heads (32 -> 1): N is far too small for the matrix cores, so this is a VALU kernel
// on the same LDS brick: one thread per output voxel, all taps x 16-channel chunks read from LDS
// as float4, the (tiny) weight set re-ordered into LDS as [chunk][tap][co][16] and read by
// broadcast.  HBM traffic = one pass over the input; LDS-read bound.
// WG = false: weights in the reference layout, re-ordered into LDS by every workgroup.
// WG = true : weights pre-packed [chunk][tap][co][16] in global memory; the address is wave-uniform, so
//             they arrive through the scalar cache (s_load_dwordx16) and feed the FMAs as SGPR operands --
//             half the LDS instructions per FMA of the WG = false form.
template <int CO, bool WG>
__global__ __launch_bounds__(256) void conv_small_co_tiled_kernel(const ConvArgs p, const float* __restrict__ wref,
                                                                  const float* __restrict__ bias) {
    constexpr int TD = 4, TH = 8, TW = 8;
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    const int tid = threadIdx.x;
    float4* wl4 = smem + (size_t)p.LD * p.PlaneQ;             // [nchunks][T][CO][16 floats]
    float* wl = reinterpret_cast<float*>(wl4);

    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int twi = bid % p.tilesW; bid /= p.tilesW;
    const int thi = bid % p.tilesH; bid /= p.tilesH;
    const int tdi = bid % p.tilesD;
    const int b = bid / p.tilesD;
    const int a0d = tdi * TD, a0h = thi * TH, a0w = twi * TW;
    const int g0d = a0d + p.dmin, g0h = a0h + p.hmin, g0w = a0w + p.wmin;

    const int nw = WG ? 0 : p.nchunks * p.T * CO * 16;
    for (int i = tid; i < nw; i += 256) {
        const int e = i & 15; int r = i >> 4;
        const int co = r % CO; r /= CO;
        const int t = r % p.T; const int ch = r / p.T;
        const int ci = ch * CC + e;
        wl[i] = (ci < p.Ci) ? wref[((size_t)co * p.Ci + ci) * p.T + t] : 0.f;
    }
    const int tw_ = tid % TW, th_ = (tid / TW) % TH, td_ = tid / (TW * TH);
    const int abase = td_ * p.PlaneQ + th_ * p.RowQ + tw_ * p.VQ;
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = bias ? bias[o] : 0.f;

    for (int ch = 0; ch < p.nchunks; ++ch) {
        if (ch) __syncthreads();
        stage_brick<256, PREC_F32, 1>(p, smem, 0, b, ch * CC, g0d, g0h, g0w, tid);
        __syncthreads();
        for (int t = 0; t < p.T; ++t) {
            const float4* xp = smem + abase + p.toff[t];
            const float4* wq = WG ? reinterpret_cast<const float4*>(wref) + (size_t)(ch * p.T + t) * CO * 4
                                  : wl4 + (size_t)(ch * p.T + t) * CO * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 xv = xp[q];
#pragma unroll
                for (int o = 0; o < CO; ++o) {
                    const float4 wv = wq[o * 4 + q];
                    acc[o] = fmaf(xv.x, wv.x, acc[o]); acc[o] = fmaf(xv.y, wv.y, acc[o]);
                    acc[o] = fmaf(xv.z, wv.z, acc[o]); acc[o] = fmaf(xv.w, wv.w, acc[o]);
                }
            }
        }
    }
    const int ad = a0d + td_, ah = a0h + th_, aw = a0w + tw_;
    if (ad < p.Ad && ah < p.Ah && aw < p.Aw) {
        const size_t vox = (((size_t)b * p.Do + ad) * p.Ho + ah) * p.Wo + aw;
#pragma unroll
        for (int o = 0; o < CO; ++o)
            p.y[vox * p.yCs + o] = acc[o] + (p.res ? p.res[vox * p.yCs + o] : 0.f);
    }
}

}  // namespace osa

using namespace osa;

// ------------------------------------------------------------------ C ABI -----
extern "C" size_t osa_conv3d_packed_floats(int Ci, int Co, int kd, int kh, int kw) {
    return packed_floats(Ci, Co, kd * kh * kw) + slack_floats(Co);
}

static void launch_pack(const PackArgs& p, int prec, float wscale, hipStream_t st) {
    if (prec == PREC_F32) {
        const size_t total = (size_t)p.nchunks * p.T * JO * 2 * p.CoP * 4;
        if (total) hipLaunchKernelGGL(pack_weights_kernel, dim3(cdiv((long long)total, 256)), dim3(256), 0, st, p);
    } else {
        const size_t total = (size_t)p.nchunks * p.T * 2 * 2 * p.CoP * 8;
        if (total) hipLaunchKernelGGL(pack_weights_f16x3_kernel, dim3(cdiv((long long)total, 256)), dim3(256), 0, st, p, wscale);
    }
}

static int conv3d_pack_impl(const float* w_ref, float* w_packed, int Ci, int Co,
                            int kd, int kh, int kw, int prec, float wscale, void* stream,
                            int src_transposed = 0, int flip = 0) {
    OSA_REQUIRE(w_ref && w_packed, "conv3d_pack: NULL pointer");
    const int T = kd * kh * kw;
    OSA_REQUIRE(T >= 1 && T <= MAX_TAPS, "conv3d_pack: %dx%dx%d kernel has %d taps (max %d)", kd, kh, kw, T, MAX_TAPS);
    OSA_REQUIRE(Ci > 0 && Co > 0, "conv3d_pack: bad channels %d->%d", Ci, Co);
    PackArgs p;
    p.src = w_ref; p.dst = w_packed; p.Ci = Ci; p.Co = Co; p.CoP = pad32(Co);
    p.kd = kd; p.kh = kh; p.kw = kw; p.T = T; p.nchunks = nchunks_of(Ci); p.transposed = src_transposed ? 1 : 0;
    int t = 0;
    for (int z = 0; z < kd; ++z) for (int y = 0; y < kh; ++y) for (int x = 0; x < kw; ++x, ++t) {
        p.kz[t] = (signed char)(flip ? kd - 1 - z : z); p.ky[t] = (signed char)(flip ? kh - 1 - y : y);
        p.kx[t] = (signed char)(flip ? kw - 1 - x : x);
    }
    launch_pack(p, prec, wscale, (hipStream_t)stream);
    OSA_LAUNCH_CHECK("conv3d_pack");
    return 0;
}

extern "C" int osa_conv3d_pack_f32(const float* w_ref, float* w_packed, int Ci, int Co,
                                   int kd, int kh, int kw, void* stream) {
    return conv3d_pack_impl(w_ref, w_packed, Ci, Co, kd, kh, kw, PREC_F32, 1.f, stream);
}

// Generic packing for backward passes: Ci/Co are the roles of the convolution that will be EXECUTED;
// src_transposed=1 reads w_ref as [Ci][Co][k] instead of [Co][Ci][k]; flip=1 mirrors the taps.
//   data-gradient of a stride-1 conv (weight [Co][Ci][k]) = conv with roles swapped:
//       pack_ex(Ci' = Co, Co' = Ci, src_transposed = 1, flip = 1), padding' = dil*(k-1) - pad
//   data-gradient of a ConvTranspose3d (weight [Ci][Co][k]) = strided conv:
//       pack_ex(Ci' = Co, Co' = Ci, src_transposed = 0, flip = 0)
extern "C" int osa_conv3d_pack_ex(const float* w_ref, float* w_packed, int Ci, int Co,
                                  int kd, int kh, int kw, int src_transposed, int flip,
                                  int f16x3, float wscale, void* stream) {
    OSA_REQUIRE(wscale > 0.f, "conv3d_pack_ex: wscale must be a positive power of two");
    return conv3d_pack_impl(w_ref, w_packed, Ci, Co, kd, kh, kw, f16x3 ? PREC_F16X3 : PREC_F32, wscale, stream,
                            src_transposed, flip);
}

extern "C" int osa_conv3d_pack_f16x3(const float* w_ref, float* w_packed, int Ci, int Co,
                                     int kd, int kh, int kw, float wscale, void* stream) {
    OSA_REQUIRE(wscale > 0.f, "conv3d_pack_f16x3: wscale must be a positive power of two");
    return conv3d_pack_impl(w_ref, w_packed, Ci, Co, kd, kh, kw, PREC_F16X3, wscale, stream);
}

extern "C" size_t osa_deconv3d_packed_floats(int Ci, int Co, int k) {
    // every kernel tap belongs to exactly one parity class -> k^3 taps in total
    return packed_floats(Ci, Co, k * k * k) + slack_floats(Co);
}

// class-major tap list of a stride-2 transposed conv: for class c = (pd,ph,pw) all (kz,ky,kx) that hit
// real inputs, with their input offsets delta.  Returns the total tap count (k^3).
struct DeconvTaps {
    int T; int cls_end[8];
    signed char kz[MAX_TAPS], ky[MAX_TAPS], kx[MAX_TAPS], dz[MAX_TAPS], dy[MAX_TAPS], dx[MAX_TAPS];
};
static void deconv_taps(int k, int pad, DeconvTaps& d, bool flat = false) {
    int t = 0;
    for (int cls = 0; cls < 8; ++cls) {
        int dd[4], kd_[4], dh[4], kh_[4], dw[4], kw_[4];
        // flat: a 1 x k x k kernel on a D = 1 tensor -- only the 4 classes with even d parity exist
        int nd = flat ? (((cls >> 2) & 1) ? 0 : 1) : deconv_dim_taps(k, pad, (cls >> 2) & 1, dd, kd_);
        if (flat) { dd[0] = 0; kd_[0] = 0; }
        const int nh = deconv_dim_taps(k, pad, (cls >> 1) & 1, dh, kh_);
        const int nw = deconv_dim_taps(k, pad, cls & 1, dw, kw_);
        for (int a = 0; a < nd; ++a) for (int b = 0; b < nh; ++b) for (int c = 0; c < nw; ++c, ++t) {
            d.kz[t] = (signed char)kd_[a]; d.ky[t] = (signed char)kh_[b]; d.kx[t] = (signed char)kw_[c];
            d.dz[t] = (signed char)dd[a]; d.dy[t] = (signed char)dh[b]; d.dx[t] = (signed char)dw[c];
        }
        d.cls_end[cls] = t;
    }
    d.T = t;
}

static int deconv3d_pack_impl(const float* w_ref, float* w_packed, int Ci, int Co,
                              int k, int pad, int prec, float wscale, void* stream, bool flat = false) {
    OSA_REQUIRE(w_ref && w_packed, "deconv3d_pack: NULL pointer");
    OSA_REQUIRE(k == 3 || k == 4, "deconv3d_pack: kernel %d unsupported (3 or 4)", k);
    DeconvTaps d;
    deconv_taps(k, pad, d, flat);
    PackArgs p;
    p.src = w_ref; p.dst = w_packed; p.Ci = Ci; p.Co = Co; p.CoP = pad32(Co);
    p.kd = flat ? 1 : k; p.kh = k; p.kw = k; p.T = d.T; p.nchunks = nchunks_of(Ci); p.transposed = 1;
    for (int t = 0; t < d.T; ++t) { p.kz[t] = d.kz[t]; p.ky[t] = d.ky[t]; p.kx[t] = d.kx[t]; }
    launch_pack(p, prec, wscale, (hipStream_t)stream);
    OSA_LAUNCH_CHECK("deconv3d_pack");
    return 0;
}

extern "C" int osa_deconv3d_pack_f32(const float* w_ref, float* w_packed, int Ci, int Co,
                                     int k, int pad, void* stream) {
    return deconv3d_pack_impl(w_ref, w_packed, Ci, Co, k, pad, PREC_F32, 1.f, stream);
}

extern "C" int osa_deconv3d_pack_f16x3(const float* w_ref, float* w_packed, int Ci, int Co,
                                       int k, int pad, float wscale, void* stream) {
    OSA_REQUIRE(wscale > 0.f, "deconv3d_pack_f16x3: wscale must be a positive power of two");
    return deconv3d_pack_impl(w_ref, w_packed, Ci, Co, k, pad, PREC_F16X3, wscale, stream);
}

// ---- 2-D transposed conv (nn.ConvTranspose2d, stride 2): the D = 1 case, 4 parity classes
extern "C" size_t osa_deconv2d_packed_floats(int Ci, int Co, int k) {
    return packed_floats(Ci, Co, k * k) + slack_floats(Co);
}

extern "C" int osa_deconv2d_pack_f32(const float* w_ref, float* w_packed, int Ci, int Co,
                                     int k, int pad, void* stream) {
    return deconv3d_pack_impl(w_ref, w_packed, Ci, Co, k, pad, PREC_F32, 1.f, stream, true);
}

extern "C" int osa_deconv2d_pack_f16x3(const float* w_ref, float* w_packed, int Ci, int Co,
                                       int k, int pad, float wscale, void* stream) {
    OSA_REQUIRE(wscale > 0.f, "deconv2d_pack_f16x3: wscale must be a positive power of two");
    return deconv3d_pack_impl(w_ref, w_packed, Ci, Co, k, pad, PREC_F16X3, wscale, stream, true);
}

static void set_ranges(ConvArgs& a, const osa_f16x3_ranges* r) {
    if (!r) return;
    a.in_meta = r->x_meta; a.res_meta = r->residual_meta; a.rx_meta = r->redir_meta; a.out_meta = r->y_meta;
    a.coef = r->bound_coef; a.rcoef = r->redir_bound_coef;
}

static int check_common(const char* what, const float* x, const float* w, float* y,
                        int B, int Di, int Hi, int Wi, int Ci, int xCs, int Co, int yCs, int rCs,
                        const float* residual) {
    OSA_REQUIRE(x && w && y, "%s: NULL pointer", what);
    OSA_REQUIRE(B > 0 && Di > 0 && Hi > 0 && Wi > 0, "%s: bad dims", what);
    OSA_REQUIRE(Ci > 0 && Co > 0, "%s: bad channels %d->%d", what, Ci, Co);
    OSA_REQUIRE(Ci % 4 == 0 && xCs % 4 == 0 && xCs >= Ci, "%s: Ci=%d / xCs=%d must be multiples of 4, xCs>=Ci", what, Ci, xCs);
    OSA_REQUIRE(((size_t)x & 15) == 0, "%s: x not 16-byte aligned", what);
    OSA_REQUIRE(yCs >= Co, "%s: yCs=%d < Co=%d", what, yCs, Co);
    if (residual) OSA_REQUIRE(rCs >= Co, "%s: rCs=%d < Co=%d", what, rCs, Co);
    return 0;
}

static int conv3d_impl(const float* x, const float* w_packed,
                       const float* scale, const float* shift, const float* residual,
                       float* y,
                       int B, int Di, int Hi, int Wi, int Ci, int xCs,
                       int Co, int yCs, int rCs,
                       int kd, int kh, int kw, int stride,
                       int pad_d, int pad_h, int pad_w,
                       int dil_d, int dil_h, int dil_w,
                       const float* gate_logits, int gCs,
                       int act, float slope, int prec, float oscale, void* stream,
                       const osa_f16x3_ranges* rng = nullptr) {
    if (gate_logits) OSA_REQUIRE(gCs >= Co, "conv3d: gate stride %d < Co %d", gCs, Co);
    if (check_common("conv3d", x, w_packed, y, B, Di, Hi, Wi, Ci, xCs, Co, yCs, rCs, residual)) return -1;
    const int T = kd * kh * kw;
    OSA_REQUIRE(T >= 1 && T <= MAX_TAPS, "conv3d: %dx%dx%d kernel unsupported", kd, kh, kw);
    OSA_REQUIRE(stride == 1 || stride == 2, "conv3d: stride %d unsupported", stride);
    OSA_REQUIRE(dil_d >= 1 && dil_h >= 1 && dil_w >= 1, "conv3d: bad dilation");
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = reinterpret_cast<const float4*>(w_packed); a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.gate = gate_logits; a.gCs = gCs;
    a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Ci = Ci; a.xCs = xCs;
    a.isd = (Di == 1 && kd == 1) ? 1 : stride; a.ish = stride; a.isw = stride;
    a.Do = (Di + 2 * pad_d - dil_d * (kd - 1) - 1) / a.isd + 1;
    a.Ho = (Hi + 2 * pad_h - dil_h * (kh - 1) - 1) / a.ish + 1;
    a.Wo = (Wi + 2 * pad_w - dil_w * (kw - 1) - 1) / a.isw + 1;
    OSA_REQUIRE(a.Do > 0 && a.Ho > 0 && a.Wo > 0, "conv3d: empty output");
    a.Co = Co; a.yCs = yCs; a.rCs = rCs;
    a.Ad = a.Do; a.Ah = a.Ho; a.Aw = a.Wo;
    a.os = 1; a.ood = a.ooh = a.oow = 0;
    a.T = T;
    int t = 0;
    for (int z = 0; z < kd; ++z) for (int yy = 0; yy < kh; ++yy) for (int xx = 0; xx < kw; ++xx, ++t) {
        a.td[t] = (signed char)(z * dil_d - pad_d);
        a.th[t] = (signed char)(yy * dil_h - pad_h);
        a.tw[t] = (signed char)(xx * dil_w - pad_w);
    }
    a.nchunks = nchunks_of(Ci); a.CoP = pad32(Co);
    a.act = act; a.slope = slope; a.oscale = oscale;
    set_ranges(a, rng);
    return launch_conv(a, stride, prec, (hipStream_t)stream, "conv3d");
}

#define OSA_CONV_PARAMS                                                                         \
    const float* x, const float* w_packed, const float* scale, const float* shift,             \
    const float* residual, float* y, int B, int Di, int Hi, int Wi, int Ci, int xCs,           \
    int Co, int yCs, int rCs, int kd, int kh, int kw, int stride, int pad_d, int pad_h,        \
    int pad_w, int dil_d, int dil_h, int dil_w, const float* gate_logits, int gCs, int act, float slope
#define OSA_CONV_ARGS                                                                           \
    x, w_packed, scale, shift, residual, y, B, Di, Hi, Wi, Ci, xCs, Co, yCs, rCs, kd, kh, kw,   \
    stride, pad_d, pad_h, pad_w, dil_d, dil_h, dil_w, gate_logits, gCs, act, slope

extern "C" int osa_conv3d_ndhwc_f32(OSA_CONV_PARAMS, void* stream) {
    return conv3d_impl(OSA_CONV_ARGS, PREC_F32, 1.f, stream);
}

extern "C" int osa_conv3d_ndhwc_f16x3(OSA_CONV_PARAMS, float out_scale, const osa_f16x3_ranges* ranges, void* stream) {
    return conv3d_impl(OSA_CONV_ARGS, PREC_F16X3, out_scale, stream, ranges);
}

static int deconv3d_impl(const float* x, const float* w_packed,
                         const float* scale, const float* shift, const float* residual,
                         float* y,
                         int B, int Di, int Hi, int Wi, int Ci, int xCs,
                         int Co, int yCs, int rCs,
                         int k, int pad, int opad,
                         const float* gate_logits, int gCs,
                         int act, float slope, int prec, float oscale, void* stream, bool flat = false,
                         const float* rx = nullptr, int rxCs = 0, int rCi = 0, const float* rw_packed = nullptr,
                         const float* rscale = nullptr, const float* rshift = nullptr, float roscale = 1.f,
                         const osa_f16x3_ranges* rng = nullptr) {
    if (gate_logits) OSA_REQUIRE(gCs >= Co, "deconv3d: gate stride %d < Co %d", gCs, Co);
    if (check_common("deconv3d", x, w_packed, y, B, Di, Hi, Wi, Ci, xCs, Co, yCs, rCs, residual)) return -1;
    if (flat) OSA_REQUIRE(Di == 1, "deconv2d: the tensor must have D == 1 (got %d)", Di);
    OSA_REQUIRE((k == 3 && pad == 1 && opad == 1) || (k == 4 && pad == 1 && opad == 0),
                "deconv3d: only (k=3,p=1,op=1) and (k=4,p=1,op=0) with stride 2 are supported (got k=%d p=%d op=%d)", k, pad, opad);
    DeconvTaps d;
    deconv_taps(k, pad, d, flat);
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = reinterpret_cast<const float4*>(w_packed);
    a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.gate = gate_logits; a.gCs = gCs;
    a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Ci = Ci; a.xCs = xCs;
    a.Do = flat ? 1 : (Di - 1) * 2 - 2 * pad + k + opad; a.Ho = (Hi - 1) * 2 - 2 * pad + k + opad; a.Wo = (Wi - 1) * 2 - 2 * pad + k + opad;
    a.Co = Co; a.yCs = yCs; a.rCs = rCs;
    a.Ad = (a.Do + 1) / 2; a.Ah = (a.Ho + 1) / 2; a.Aw = (a.Wo + 1) / 2;     // a-space covers every output parity
    a.isd = a.ish = a.isw = 1;
    a.os = 2;
    a.T = d.T;
    for (int t = 0; t < d.T; ++t) { a.td[t] = d.dz[t]; a.th[t] = d.dy[t]; a.tw[t] = d.dx[t]; }
    for (int c = 0; c < 8; ++c) a.cls_end[c] = d.cls_end[c];
    a.nchunks = nchunks_of(Ci); a.CoP = pad32(Co);
    a.act = act; a.slope = slope; a.oscale = oscale;
    if (rx) {
        OSA_REQUIRE(!flat && !residual && !gate_logits, "deconv3d_redir: residual / gate cannot be combined with the fused redir branch");
        OSA_REQUIRE(rw_packed && rCi > 0 && rCi <= 64 && rCi % 4 == 0 && rxCs >= rCi && rxCs % 4 == 0 && ((size_t)rx & 15) == 0,
                    "deconv3d_redir: redir input needs <= 64 channels (multiple of 4), stride >= channels, 16-byte alignment (Ci=%d stride=%d)", rCi, rxCs);
        OSA_REQUIRE((long long)a.Do * a.Ho * a.Wo * rxCs < (1ll << 31), "deconv3d_redir: redir input too large");
        a.rx = rx; a.rxCs = rxCs; a.rCi = rCi; a.rw = reinterpret_cast<const float4*>(rw_packed);
        a.rscale = rscale; a.rshift = rshift; a.roscale = roscale;
    }
    set_ranges(a, rng);
    return launch_conv(a, 1, prec, (hipStream_t)stream, flat ? "deconv2d" : "deconv3d",
                       flat ? &g_deconv_flat_cfg : (rx ? (rCi > 32 ? &g_deconv_redir64_cfg : &g_deconv_redir_cfg) : &g_deconv_cfg));
}

#define OSA_DECONV_PARAMS                                                                       \
    const float* x, const float* w_packed, const float* scale, const float* shift,             \
    const float* residual, float* y, int B, int Di, int Hi, int Wi, int Ci, int xCs,           \
    int Co, int yCs, int rCs, int k, int pad, int opad, const float* gate_logits, int gCs, int act, float slope
#define OSA_DECONV_ARGS                                                                         \
    x, w_packed, scale, shift, residual, y, B, Di, Hi, Wi, Ci, xCs, Co, yCs, rCs, k, pad, opad, \
    gate_logits, gCs, act, slope

extern "C" int osa_deconv3d_ndhwc_f32(OSA_DECONV_PARAMS, void* stream) {
    return deconv3d_impl(OSA_DECONV_ARGS, PREC_F32, 1.f, stream);
}

extern "C" int osa_deconv3d_ndhwc_f16x3(OSA_DECONV_PARAMS, float out_scale, const osa_f16x3_ranges* ranges, void* stream) {
    return deconv3d_impl(OSA_DECONV_ARGS, PREC_F16X3, out_scale, stream, false, nullptr, 0, 0, nullptr, nullptr, nullptr, 1.f, ranges);
}

// transposed conv with the 1x1x1 redir branch computed in its epilogue (see ConvArgs::rx)
#define OSA_REDIR_PARAMS const float* rx, int rxCs, int rCi, const float* rw_packed, const float* rscale, const float* rshift
extern "C" int osa_deconv3d_redir_ndhwc_f32(const float* x, const float* w_packed, const float* scale, const float* shift, float* y,
                                            int B, int Di, int Hi, int Wi, int Ci, int xCs, int Co, int yCs,
                                            int k, int pad, int opad, OSA_REDIR_PARAMS, int act, float slope, void* stream) {
    return deconv3d_impl(x, w_packed, scale, shift, nullptr, y, B, Di, Hi, Wi, Ci, xCs, Co, yCs, 0, k, pad, opad, nullptr, 0,
                         act, slope, PREC_F32, 1.f, stream, false, rx, rxCs, rCi, rw_packed, rscale, rshift, 1.f);
}

extern "C" int osa_deconv3d_redir_ndhwc_f16x3(const float* x, const float* w_packed, const float* scale, const float* shift, float* y,
                                              int B, int Di, int Hi, int Wi, int Ci, int xCs, int Co, int yCs,
                                              int k, int pad, int opad, OSA_REDIR_PARAMS, float r_out_scale,
                                              int act, float slope, float out_scale, const osa_f16x3_ranges* ranges, void* stream) {
    return deconv3d_impl(x, w_packed, scale, shift, nullptr, y, B, Di, Hi, Wi, Ci, xCs, Co, yCs, 0, k, pad, opad, nullptr, 0,
                         act, slope, PREC_F16X3, out_scale, stream, false, rx, rxCs, rCi, rw_packed, rscale, rshift, r_out_scale, ranges);
}

#define OSA_DECONV2D_PARAMS                                                                     \
    const float* x, const float* w_packed, const float* scale, const float* shift,             \
    const float* residual, float* y, int B, int Hi, int Wi, int Ci, int xCs,                   \
    int Co, int yCs, int rCs, int k, int pad, int opad, const float* gate_logits, int gCs, int act, float slope
#define OSA_DECONV2D_ARGS                                                                       \
    x, w_packed, scale, shift, residual, y, B, 1, Hi, Wi, Ci, xCs, Co, yCs, rCs, k, pad, opad, \
    gate_logits, gCs, act, slope

extern "C" int osa_deconv2d_nhwc_f32(OSA_DECONV2D_PARAMS, void* stream) {
    return deconv3d_impl(OSA_DECONV2D_ARGS, PREC_F32, 1.f, stream, true);
}

extern "C" int osa_deconv2d_nhwc_f16x3(OSA_DECONV2D_PARAMS, float out_scale, const osa_f16x3_ranges* ranges, void* stream) {
    return deconv3d_impl(OSA_DECONV2D_ARGS, PREC_F16X3, out_scale, stream, true, nullptr, 0, 0, nullptr, nullptr, nullptr, 1.f, ranges);
}

__global__ __launch_bounds__(256) void small_co_pack_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            int Ci, int Co, int T, int nchunks) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // dst index ((ch*T + t)*Co + co)*16 + e
    if (i >= nchunks * T * Co * 16) return;
    const int e = i & 15; int r = i >> 4;
    const int co = r % Co; r /= Co;
    const int t = r % T; const int ch = r / T;
    const int ci = ch * CC + e;
    dst[i] = (ci < Ci) ? src[((size_t)co * Ci + ci) * T + t] : 0.f;
}

extern "C" size_t osa_conv3d_small_co_packed_floats(int Ci, int Co, int kd, int kh, int kw) {
    return (size_t)nchunks_of(Ci) * kd * kh * kw * Co * 16;
}

extern "C" int osa_conv3d_small_co_pack_f32(const float* w_ref, float* w_packed, int Ci, int Co,
                                            int kd, int kh, int kw, void* stream) {
    OSA_REQUIRE(w_ref && w_packed, "conv3d_small_co_pack: NULL pointer");
    OSA_REQUIRE(Ci > 0 && Co >= 1 && Co <= 4 && kd > 0 && kh > 0 && kw > 0, "conv3d_small_co_pack: bad dims");
    const int n = (int)osa_conv3d_small_co_packed_floats(Ci, Co, kd, kh, kw);
    hipLaunchKernelGGL(small_co_pack_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       w_ref, w_packed, Ci, Co, kd * kh * kw, nchunks_of(Ci));
    OSA_LAUNCH_CHECK("conv3d_small_co_pack");
    return 0;
}

static int small_co_impl(const float* x, const float* w, bool packed, const float* bias,
                         const float* residual, float* y,
                         int B, int D, int H, int W, int Ci, int xCs, int Co, int yCs,
                         int kd, int kh, int kw, int pad_d, int pad_h, int pad_w,
                         void* stream);

extern "C" int osa_conv3d_small_co_ndhwc_f32(const float* x, const float* w_ref, const float* bias,
                                             const float* residual, float* y,
                                             int B, int D, int H, int W, int Ci, int xCs, int Co, int yCs,
                                             int kd, int kh, int kw, int pad_d, int pad_h, int pad_w,
                                             void* stream) {
    return small_co_impl(x, w_ref, false, bias, residual, y, B, D, H, W, Ci, xCs, Co, yCs, kd, kh, kw, pad_d, pad_h, pad_w, stream);
}

extern "C" int osa_conv3d_small_co_packed_ndhwc_f32(const float* x, const float* w_packed, const float* bias,
                                                    const float* residual, float* y,
                                                    int B, int D, int H, int W, int Ci, int xCs, int Co, int yCs,
                                                    int kd, int kh, int kw, int pad_d, int pad_h, int pad_w,
                                                    void* stream) {
    OSA_REQUIRE(((size_t)w_packed & 63) == 0, "conv3d_small_co_packed: packed weights must be 64-byte aligned");
    return small_co_impl(x, w_packed, true, bias, residual, y, B, D, H, W, Ci, xCs, Co, yCs, kd, kh, kw, pad_d, pad_h, pad_w, stream);
}

static int small_co_impl(const float* x, const float* w_ref, bool packed, const float* bias,
                         const float* residual, float* y,
                         int B, int D, int H, int W, int Ci, int xCs, int Co, int yCs,
                         int kd, int kh, int kw, int pad_d, int pad_h, int pad_w,
                         void* stream) {
    OSA_REQUIRE(x && w_ref && y, "conv3d_small_co: NULL pointer");
    OSA_REQUIRE(Co >= 1 && Co <= 4, "conv3d_small_co: Co=%d unsupported (1..4)", Co);
    OSA_REQUIRE(xCs % 4 == 0 && xCs >= Ci && ((size_t)x & 15) == 0, "conv3d_small_co: x must be 16-byte aligned, xCs %% 4 == 0");
    OSA_REQUIRE(kd == 2 * pad_d + 1 && kh == 2 * pad_h + 1 && kw == 2 * pad_w + 1, "conv3d_small_co: only 'same' convolutions");
    OSA_REQUIRE(kd * kh * kw <= MAX_TAPS, "conv3d_small_co: too many taps");
    OSA_REQUIRE(yCs >= Co, "conv3d_small_co: yCs < Co");
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.res = residual; a.oscale = 1.f;
    a.B = B; a.Di = D; a.Hi = H; a.Wi = W; a.Ci = Ci; a.xCs = xCs;
    a.Do = D; a.Ho = H; a.Wo = W; a.Co = Co; a.yCs = yCs;
    a.Ad = D; a.Ah = H; a.Aw = W;
    a.isd = a.ish = a.isw = 1; a.os = 1;
    a.T = kd * kh * kw;
    int t = 0;
    for (int z = 0; z < kd; ++z) for (int yy = 0; yy < kh; ++yy) for (int xx = 0; xx < kw; ++xx, ++t) {
        a.td[t] = (signed char)(z - pad_d); a.th[t] = (signed char)(yy - pad_h); a.tw[t] = (signed char)(xx - pad_w);
    }
    a.nchunks = nchunks_of(Ci); a.CoP = Co;
    a.dmin = -pad_d; a.hmin = -pad_h; a.wmin = -pad_w;
    a.LD = 4 + 2 * pad_d; a.LH = 8 + 2 * pad_h; a.LW = 8 + 2 * pad_w;
    a.tilesD = cdiv(D, 4); a.tilesH = cdiv(H, 8); a.tilesW = cdiv(W, 8);
    finish_geometry(a, 8, true);       // thread q reads voxel q of the 4x8x8 tile: the MFMA A-operand pattern, compact image
    const size_t lds = ((size_t)a.LD * a.PlaneQ * 4 + (packed ? 0 : (size_t)a.nchunks * a.T * Co * 16)) * sizeof(float);
    OSA_REQUIRE(lds <= 160 * 1024, "conv3d_small_co: %zu B of LDS needed", lds);
    const long long nblk = (long long)B * a.tilesD * a.tilesH * a.tilesW;
    OSA_REQUIRE(nblk < (1ll << 31), "conv3d_small_co: grid too large");
    dim3 grid((unsigned)nblk), block(256);
    hipStream_t st = (hipStream_t)stream;
#define OSA_SC_LAUNCH1(CO, WG)                                                                              \
    do {                                                                                                    \
        if (lds > 64 * 1024)                                                                                \
            (void)hipFuncSetAttribute((const void*)conv_small_co_tiled_kernel<CO, WG>,                      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
        hipLaunchKernelGGL((conv_small_co_tiled_kernel<CO, WG>), grid, block, lds, st, a, w_ref, bias);     \
    } while (0)
#define OSA_SC_LAUNCH(CO) do { if (packed) OSA_SC_LAUNCH1(CO, true); else OSA_SC_LAUNCH1(CO, false); } while (0)
    switch (Co) {
        case 1: OSA_SC_LAUNCH(1); break;
        case 2: OSA_SC_LAUNCH(2); break;
        case 3: OSA_SC_LAUNCH(3); break;
        default: OSA_SC_LAUNCH(4); break;
    }
#undef OSA_SC_LAUNCH
    OSA_LAUNCH_CHECK("conv3d_small_co");
    return 0;
}
