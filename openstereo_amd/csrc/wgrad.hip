// Weight gradient of the 3-D convolutions on the fp32 matrix cores (training).
//
//   dW[a][b][t] = sum_{batch, pos}  P[pos][a] * Q[pos*s + off_t][b]
//     Conv3d          (weight [Co][Ci][k]) : P = dy (a = co) at output positions, Q = x  (b = ci), off_t = t*dil - pad
//     ConvTranspose3d (weight [Ci][Co][k]) : P = x  (a = ci) at input  positions, Q = dy (b = co), off_t = t - pad, s = 2
// i.e. one kernel: for every tap a 32x32 (a x b) GEMM whose K dimension is the voxel axis.
//
// A workgroup owns one 32x32 (a,b) tile pair, a group of taps (one kd plane) and a strip of position
// bricks.  Per brick it stages the P tile [pos][32] and the Q brick (halo included) [vox][32] into LDS;
// each of the 4 waves takes a quarter of the positions, keeps its P operands in registers and streams
// Q operands per tap: v_mfma_f32_32x32x2_f32 with K = 2 positions per instruction, one accumulator
// tile per tap kept in registers across ALL bricks of the strip, so only one round of float atomics
// per workgroup reaches the [a][b][t] gradient buffer.
//
// Transposed convs (q = 2*pos + off_t) run in "class mode": a tap group is one output-parity class (pd, ph, pw) -- all taps whose
// off_t has that parity, at most 2 x 2 x 2 = 8 -- and on the sub-lattice of that parity the access is unit stride,
// q = 2*(pos + delta_t) + par, delta in {dmin, dmin + 1}.  The workgroup stages only the (TD+1) x (TH+1) x (TW+1) brick of that
// sub-lattice (compact in LDS, stride-2 gather from HBM) instead of the full strided brick for every group of 9 taps: the k = 4
// layers of the StereoBase / IGEV hourglasses staged 93 KB per 32 positions and 8 tap groups, i.e. 8x more bytes than they used
// (3.65 ms per launch at the 320x736 training crop).
#include "osa_common.h"
#include <cstring>
#include <cstddef>

namespace osa {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_TAPS = 9;        // taps per workgroup (register accumulators: 9 x 16)
constexpr int WG_PS = 36;         // LDS row stride (floats) of a 32-channel voxel: conflict-free column reads

struct WgradArgs {
    const float* P; const float* Q; float* dW;
    float* ws;                    // partial tiles [workgroup][WG_TAPS][16][64] (two-stage reduction) or NULL (float atomics)
    int B;
    int Pd, Ph, Pw, PC, PCs;      // P tensor dims (positions) / channels / voxel stride
    int Qd, Qh, Qw, QC, QCs;      // Q tensor
    int s;                        // Q position = pos*s + off
    int T;                        // total taps (kd*kh*kw)
    int kh, kw, kvol;
    int tgroups;                  // ceil(T / WG_TAPS)
    int tilesD, tilesH, tilesW, strip;   // position bricks; `strip` consecutive w-bricks per workgroup
    int LD, LH, LW;               // Q brick dims
    int dmin, hmin, wmin;
    int A, Bc;                    // number of a / b channels (dW is [A][Bc][kvol])
    signed char od[64], oh[64], ow[64];  // per-tap Q offsets (k = 4 transposed convs have 64 taps); class mode: delta_t on the class sub-lattice
    // class mode (transposed convs): tap arrays are class-major, tapid maps back to the kernel index
    int cls;
    signed char tapid[64];
    signed char g_t0[16], g_nt[16], g_par[16];   // first tap / tap count / parity bits (pd<<2 | ph<<1 | pw) of every tap group (fp32 form: the 8 classes)
    // f16x3 form: a tap group is the taps of ONE d offset (of one parity class in class mode): up to 16 groups
    signed char g_od[16];                      // its d offset (class mode: delta on the sub-lattice)
    signed char g_slot[16][9];                 // accumulator slot dh * 3 + dw -> tap index inside the group, or -1
    const float* Pmeta; const float* Qmeta;   // f16x3 form: range blocks (max |.|) of the P and Q tensors
    int Pf16, Qf16;                            // native f16 form (r5): the P / Q tensor holds fp16 elements (channel strides PCs / QCs in elements)
    // r6: the batch may be a LIST of equally shaped tensors (the queued (x, dy) pairs of a weight applied several times in one step): item
    // i = batch entries [i * bper, (i + 1) * bper) lives at Ptab[i] / Qtab[i]; ntab = 0: one tensor each at P / Q
    const void* Ptab[24]; const void* Qtab[24];
    int ntab, bper;
    int dbg;                                  // experiments build (OSA_WG_DBG): timing-only ablations of wgrad_f16x3_kernel -- 1 no global loads, 2 no LDS commit, 4 no MFMA phase, 8 no hand-over
};

// entry i of a pointer table inside the kernel argument (the ONLY kernel parameter: offset 0 of the kernarg segment) -- read through the
// kernarg pointer with a scalar load: indexing the by-value struct dynamically makes the compiler copy all of it (1 KB) to scratch
__device__ __forceinline__ const void* wg_tab(size_t table_offset, int i) {
    typedef const char __attribute__((address_space(4))) kchar;
    kchar* ka = (kchar*)__builtin_amdgcn_kernarg_segment_ptr();
    return *reinterpret_cast<const void* const __attribute__((address_space(4)))*>(ka + table_offset + (size_t)i * sizeof(void*));
}

template <int TD, int TH, int TW>     // position brick, TD*TH*TW = 256 (64 per wave) or 64 (16 per wave)
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs p) {
    constexpr int NPOS = TD * TH * TW;
    constexpr int PER_WAVE = NPOS / 4;
    constexpr int KSTEPS = PER_WAVE / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ps = smem;                          // [NPOS][WG_PS]
    float* Qs = smem + NPOS * WG_PS;           // [LD*LH*LW][WG_PS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, hh = lane >> 5;

    // blockIdx.x -> (strip index over bricks, tap group); blockIdx.y -> (a tile, b tile).  (Walking the ids XCD-contiguously -- xcd_remap,
    // so that the tap groups of one strip share an L2 -- was measured: the fp32 form unchanged, the f16x3 form 12-27 % SLOWER.)
    const unsigned bx = blockIdx.x;
    const int tg = bx % p.tgroups;
    int sidx = bx / p.tgroups;
    const int atiles = (p.A + 31) / 32;
    const int a0 = (blockIdx.y % atiles) * 32, b0 = (blockIdx.y / atiles) * 32;
    const int t0 = p.cls ? p.g_t0[tg] : tg * WG_TAPS;
    const int nt = p.cls ? p.g_nt[tg] : ((p.T - t0 < WG_TAPS) ? (p.T - t0) : WG_TAPS);
    // class mode: parity of the class and first delta per dimension (par 0: delta in {0, 1}; par 1: {-1, 0}); LDS brick unit stride
    const int par = p.cls ? p.g_par[tg] : 0;
    const int pard = (par >> 2) & 1, parh = (par >> 1) & 1, parw = par & 1;
    const int dmin = p.cls ? -pard : p.dmin, hmin = p.cls ? -parh : p.hmin, wmin = p.cls ? -parw : p.wmin;
    const int ls = p.cls ? 1 : p.s;                  // position step inside the LDS brick
    const int nstripsW = (p.tilesW + p.strip - 1) / p.strip;
    const int sw = sidx % nstripsW; sidx /= nstripsW;
    const int thi = sidx % p.tilesH; sidx /= p.tilesH;
    const int tdi = sidx % p.tilesD; const int b = sidx / p.tilesD;
    const int bl = p.ntab ? b % p.bper : b;
    const float* const Pbase = p.ntab ? static_cast<const float*>(wg_tab(offsetof(WgradArgs, Ptab), b / p.bper)) : p.P;
    const float* const Qbase = p.ntab ? static_cast<const float*>(wg_tab(offsetof(WgradArgs, Qtab), b / p.bper)) : p.Q;

    f32x16 acc[WG_TAPS];
#pragma unroll
    for (int t = 0; t < WG_TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nQ = p.LD * p.LH * p.LW;
    for (int twi = sw * p.strip; twi < (sw + 1) * p.strip && twi < p.tilesW; ++twi) {
        const int p0d = tdi * TD, p0h = thi * TH, p0w = twi * TW;
        __syncthreads();
        // ---- stage P tile: NPOS positions x 32 channels (zero outside the tensor / channel range)
        for (int it = tid; it < NPOS * 8; it += 256) {
            const int c4 = it & 7, q = it >> 3;
            const int pw = q % TW, ph = (q / TW) % TH, pd = q / (TW * TH);
            const int gd = p0d + pd, gh = p0h + ph, gw = p0w + pw;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gd < p.Pd && gh < p.Ph && gw < p.Pw) {
                const float* src = Pbase + ((((size_t)bl * p.Pd + gd) * p.Ph + gh) * p.Pw + gw) * p.PCs + a0 + c4 * 4;
                if (a0 + c4 * 4 + 3 < p.PC) v = *reinterpret_cast<const float4*>(src);
                else { if (a0 + c4 * 4 < p.PC) v.x = src[0]; if (a0 + c4 * 4 + 1 < p.PC) v.y = src[1]; if (a0 + c4 * 4 + 2 < p.PC) v.z = src[2]; }
            }
            *reinterpret_cast<float4*>(Ps + q * WG_PS + c4 * 4) = v;
        }
        // ---- stage Q brick
        const int q0d = p0d * ls + dmin, q0h = p0h * ls + hmin, q0w = p0w * ls + wmin;
        for (int it = tid; it < nQ * 8; it += 256) {
            const int c4 = it & 7, v_ = it >> 3;
            const int lw = v_ % p.LW, lh = (v_ / p.LW) % p.LH, ld = v_ / (p.LW * p.LH);
            int gd = q0d + ld, gh = q0h + lh, gw = q0w + lw;
            if (p.cls) { gd = 2 * gd + pard; gh = 2 * gh + parh; gw = 2 * gw + parw; }      // sub-lattice -> tensor coordinates
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)gd < (unsigned)p.Qd && (unsigned)gh < (unsigned)p.Qh && (unsigned)gw < (unsigned)p.Qw) {
                const float* src = Qbase + ((((size_t)bl * p.Qd + gd) * p.Qh + gh) * p.Qw + gw) * p.QCs + b0 + c4 * 4;
                if (b0 + c4 * 4 + 3 < p.QC) v = *reinterpret_cast<const float4*>(src);
                else { if (b0 + c4 * 4 < p.QC) v.x = src[0]; if (b0 + c4 * 4 + 1 < p.QC) v.y = src[1]; if (b0 + c4 * 4 + 2 < p.QC) v.z = src[2]; }
            }
            *reinterpret_cast<float4*>(Qs + v_ * WG_PS + c4 * 4) = v;
        }
        __syncthreads();
        // ---- this wave's positions: K steps of 2 positions (k = hh selects the position of the pair)
        float areg[KSTEPS];
        int qbase[KSTEPS];
#pragma unroll
        for (int k = 0; k < KSTEPS; ++k) {
            const int q = wave * PER_WAVE + 2 * k + hh;
            areg[k] = Ps[q * WG_PS + col];
            const int pw = q % TW, ph = (q / TW) % TH, pd = q / (TW * TH);
            qbase[k] = (((pd * ls) * p.LH + ph * ls) * p.LW + pw * ls) * WG_PS + col;
        }
#pragma unroll
        for (int t = 0; t < WG_TAPS; ++t) {
            if (t < nt) {
                const int tt = t0 + t;
                const int toff = (((p.od[tt] - dmin) * p.LH + (p.oh[tt] - hmin)) * p.LW + (p.ow[tt] - wmin)) * WG_PS;
#pragma unroll
                for (int k = 0; k < KSTEPS; ++k)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[k], Qs[qbase[k] + toff], acc[t], 0, 0, 0);
            }
        }
    }
    if (p.ws) {
        // ---- partial tiles to the workspace (lane-contiguous, plain stores); wgrad_reduce_kernel adds them up.  Float atomics on the
        // 27-64 K words of dW from every workgroup serialise in L2 (measured 4.5 G atomics/s: 0.5 ms per workgroup once a few thousand
        // of them contend) and make the gradient depend on the arrival order; this path is deterministic.
        float* dst = p.ws + ((size_t)blockIdx.y * gridDim.x + bx) * (WG_TAPS * 1024) + lane;
        // every wave holds a partial sum over its quarter of the positions: add the 4 waves up through LDS first, tap by tap
        // (3 x 16 x 64 floats = 12 KB at a time)
        __syncthreads();
        float* red = smem;
#pragma unroll
        for (int t = 0; t < WG_TAPS; ++t) {
            if (t < nt) {
                if (wave > 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[t][r];
                }
                __syncthreads();
                if (wave == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        dst[(t * 16 + r) * 64] = acc[t][r] + red[r * 64 + lane] + red[(16 + r) * 64 + lane] + red[(32 + r) * 64 + lane];
                }
                __syncthreads();
            }
        }
        return;
    }
    // ---- one round of atomics: dW[a][b][tap]
#pragma unroll
    for (int t = 0; t < WG_TAPS; ++t) {
        if (t < nt) {
            const int tt = p.cls ? p.tapid[t0 + t] : t0 + t;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int a = a0 + (r & 3) + 8 * (r >> 2) + 4 * hh, bb = b0 + col;
                if (a < p.A && bb < p.Bc) atomicAdd(p.dW + ((size_t)a * p.Bc + bb) * p.kvol + tt, acc[t][r]);
            }
        }
    }
}


// ---- split-precision (f16x3) form --------------------------------------------------------------------------------------------------
// dW = sum_pos P[pos][a] Q[pos + off][b] on v_mfma_f32_32x32x16_f16: K = 16 POSITIONS per instruction (8x the fp32 instruction's 2, at
// half its cycles), three instructions per product (P_hi Q_lo + P_lo Q_hi + P_hi Q_hi, fp32 accumulate), operands pre-scaled by
// powers of two from their tensors' range blocks.  An MFMA operand lane holds 8 consecutive K values of ONE channel, so the LDS images are
// channel-major with the position axis contiguous -- [channel][row][w] in fp16 hi and lo planes -- and a K half-block is one w-row of 8
// positions.  Staging transposes: a thread takes (two w-neighbours, four channels), converts, and writes 32-bit hi / lo pairs.
// A tap's Q operand is the same row shifted by dh rows and dw = 0..2 halves; rows are padded to 16-byte multiples, one ds_read_b128 +
// one ds_read_b32 per row and plane serve all three dw (dw = 0: as read, dw = 1: v_alignbit by 16, dw = 2: the next dwords).
// Unit stride, unit dilation, tap groups = one kd plane (<= 9 taps); 128-position bricks (2x8x8, flat 1x8x16): 59 / 49 KB of LDS, 2 / 3
// workgroups per CU.  Same accumulator layout, workspace records and reduce kernel as the fp32 form.
typedef _Float16 wf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 wf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float wg_pow2_scale(float amax) {              // = pow2_scale of conv_kernel.h: amax * s in [2^14, 2^15)
    const unsigned b = __builtin_bit_cast(unsigned, amax);
    const int eb = (int)((b >> 23) & 0xffu);
    if (eb == 0 || eb == 255) return 1.f;
    int k = 15 - (eb - 126);
    k = k < -60 ? -60 : (k > 60 ? 60 : k);
    return __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
}
// two scaled values -> packed fp16 hi pair (round toward zero) and lo pair (x - hi, rounded to nearest); low half = first value
// two values -> packed fp16 pair, each rounded to nearest even (what autocast's cast does); low half = first value
__device__ __forceinline__ unsigned wg_round2(float x0, float x1) {
    const wf16x2 h = {(_Float16)x0, (_Float16)x1};
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ void wg_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    const wf16x2 h = __builtin_bit_cast(wf16x2, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    const wf16x2 l = {(_Float16)(x0 - (float)h[0]), (_Float16)(x1 - (float)h[1])};
    hi = __builtin_bit_cast(unsigned, h); lo = __builtin_bit_cast(unsigned, l);
}

// SPLIT = 1: the f16x3 form above.  SPLIT = 0 (r5): the native f16 form -- the arithmetic of the reference's AMP training
// (trainer_template.py:211-226: autocast + GradScaler; the weight gradient of an autocast convolution multiplies fp16 activations by
// fp16 output gradients and accumulates in fp32): operands rounded to fp16 (nearest even) when they are staged, ONE MFMA per product, no
// lo planes in LDS (half the footprint).  Range blocks are optional there (NULL: no scaling, like autocast -- GradScaler owns the range).
// PF16 / QF16 (SPLIT = 0 only): the P / Q tensor holds fp16 elements -- compile-time, so that the load sequence of every variant is
// straight-line (a run-time dtype branch inside the back-to-back loads cost the fp32-tensor variant 25 % of its speed).
template <int TD, int TH, int TW, int SPLIT, int PF16 = 0, int QF16 = 0>     // 128 positions: 2x8x8 or (flat) 1x8x16
__global__ __launch_bounds__(256) void wgrad_f16x3_kernel(const WgradArgs p) {
    static_assert(!SPLIT || (!PF16 && !QF16), "fp16 tensors exist in the native f16 form only");
    static_assert(TD * TH * TW == 128 && (TW == 8 || TW == 16), "128-position bricks");
    constexpr int ROWH = (TW == 8) ? 16 : 24;        // halves per Q row in LDS (LW <= TW + 2, padded to a 16-byte multiple)
    constexpr int LHM = TH + 2;                      // Q rows per plane in LDS (host: at most 2 halo rows)
    constexpr int CHS_P = 136;                       // halves per channel (128 positions + 8: 16-byte rows of consecutive channels on distinct slots)
    constexpr int CHS_Q = TD * LHM * ROWH + 8;       // 328 / 248
    static_assert((CHS_P / 8) % 2 == 1 && (CHS_Q / 8) % 2 == 1, "odd number of 16-byte slots per channel");
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
    unsigned short* const Ph = smem_h;               // [32][CHS_P] hi halves of the P tile
    unsigned short* const Pl = Ph + 32 * CHS_P;      // (SPLIT = 0: no lo planes -- Pl / Ql are never touched and Qh follows Ph)
    unsigned short* const Qh = SPLIT ? Pl + 32 * CHS_P : Ph + 32 * CHS_P;      // [32][CHS_Q] hi halves of the Q brick
    unsigned short* const Ql = Qh + 32 * CHS_Q;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, hh = lane >> 5;
    const unsigned bx = blockIdx.x;
    const int tg = bx % p.tgroups;
    int sidx = bx / p.tgroups;
    const int atiles = (p.A + 31) / 32;
    const int a0 = (blockIdx.y % atiles) * 32, b0 = (blockIdx.y / atiles) * 32;
    // tap group = the taps of one d offset (host tables); class mode (stride-2 / transposed layers): Q is read on the parity sub-lattice
    // q = 2 * (pos + delta) + par of the group's class, where it is unit stride
    const int od = p.g_od[tg];
    const int par = p.cls ? p.g_par[tg] : 0;
    const int pard = (par >> 2) & 1, parh = (par >> 1) & 1, parw = par & 1;
    const int qs = p.cls ? 2 : 1;
    const int ghmin = p.cls ? -parh : p.hmin, gwmin = p.cls ? -parw : p.wmin;
    const int nstripsW = (p.tilesW + p.strip - 1) / p.strip;
    const int sw = sidx % nstripsW; sidx /= nstripsW;
    const int thi = sidx % p.tilesH; sidx /= p.tilesH;
    const int tdi = sidx % p.tilesD; const int b = sidx / p.tilesD;
    const int bl = p.ntab ? b % p.bper : b;
    const float* const Pbase = p.ntab ? static_cast<const float*>(wg_tab(offsetof(WgradArgs, Ptab), b / p.bper)) : p.P;
    const float* const Qbase = p.ntab ? static_cast<const float*>(wg_tab(offsetof(WgradArgs, Qtab), b / p.bper)) : p.Q;
    const float sP = p.Pmeta ? wg_pow2_scale(amax_read(p.Pmeta)) : 1.f, sQ = p.Qmeta ? wg_pow2_scale(amax_read(p.Qmeta)) : 1.f;
    const float inv = (1.0f / sP) * (1.0f / sQ);

    f32x16 acc[WG_TAPS];
#pragma unroll
    for (int t = 0; t < WG_TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int npq = (p.LW + 1) >> 1;                 // voxel pairs per Q row
    const int nQ = TD * p.LH * npq * 8;              // staging items of the Q brick: (row, pair, channel quad)
    // Staging items of this thread (the same for every brick): P: 64 position pairs x 8 channel quads = 2 per thread; Q: (row, voxel
    // pair, channel quad), up to QIT per thread.  All global loads of a brick are issued back to back into registers (a loop of
    // load -> convert -> ds_write per item costs one HBM / L2 round trip PER ITEM: measured 11.5 us per brick), and the NEXT brick's loads
    // are issued before the MFMA phase of the current one, so their latency hides behind it.
    constexpr int PIT = 2, QIT = (TD * LHM * ((TW + 3) / 2) * 8 + 255) / 256;
    int p_src[PIT], p_dst[PIT], q_src[QIT], q_dst[QIT];          // element offsets inside the batch item (relative to the brick origin) / LDS halves
    unsigned p_ok[PIT], q_ok[QIT];                                // bit 0: item exists; the coordinates are re-checked per brick
    int p_d[PIT], p_h[PIT], p_w[PIT], q_d[QIT], q_h[QIT], q_w[QIT];
#pragma unroll
    for (int k = 0; k < PIT; ++k) {
        const int it = tid + 256 * k;
        const int c4 = it & 7, q0 = (it >> 3) * 2;
        p_w[k] = q0 % TW; p_h[k] = (q0 / TW) % TH; p_d[k] = q0 / (TW * TH);
        p_src[k] = c4 * 4; p_dst[k] = (c4 * 4) * CHS_P + q0; p_ok[k] = 1u;
    }
#pragma unroll
    for (int k = 0; k < QIT; ++k) {
        const int it = tid + 256 * k;
        const int c4 = it & 7; int r = it >> 3;
        const int pr = r % npq; r /= npq;
        q_h[k] = r % p.LH; q_d[k] = r / p.LH; q_w[k] = pr * 2;
        q_src[k] = c4 * 4; q_dst[k] = (c4 * 4) * CHS_Q + (q_d[k] * LHM + q_h[k]) * ROWH + q_w[k];
        q_ok[k] = (it < nQ) ? 1u : 0u;
    }
    float4 pv[PIT][2], qv[QIT][2];
#pragma unroll
    for (int k = 0; k < PIT; ++k) pv[k][0] = pv[k][1] = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
    for (int k = 0; k < QIT; ++k) qv[k][0] = qv[k][1] = make_float4(1.f, 1.f, 1.f, 1.f);
    // four channels of one position at ELEMENT offset `off` of a tensor of fp32 or (f16 = 1: native f16 form, channel stride % 4 == 0) fp16 elements
    auto load4f = [&](const float* src, int nc, bool ok) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            if (nc >= 4) v = *reinterpret_cast<const float4*>(src);
            else { if (nc > 0) v.x = src[0]; if (nc > 1) v.y = src[1]; if (nc > 2) v.z = src[2]; }
        }
        return v;
    };
    auto load4h = [&](const _Float16* src, int nc, bool ok) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            if (nc >= 4) {
                const uint2 u = *reinterpret_cast<const uint2*>(src);
                const wf16x2 a = __builtin_bit_cast(wf16x2, u.x), c = __builtin_bit_cast(wf16x2, u.y);
                v = make_float4((float)a[0], (float)a[1], (float)c[0], (float)c[1]);
            } else { if (nc > 0) v.x = (float)src[0]; if (nc > 1) v.y = (float)src[1]; if (nc > 2) v.z = (float)src[2]; }
        }
        return v;
    };
    auto issue_loads = [&](int twi) {
        if (p.dbg & 1) return;
        const int p0d = tdi * TD, p0h = thi * TH, p0w = twi * TW;
#pragma unroll
        for (int k = 0; k < PIT; ++k) {
            const int gd = p0d + p_d[k], gh = p0h + p_h[k], gw = p0w + p_w[k];
            const bool row = gd < p.Pd && gh < p.Ph;
            const size_t off = ((((size_t)bl * p.Pd + (row ? gd : 0)) * p.Ph + (row ? gh : 0)) * p.Pw + gw) * p.PCs + a0 + p_src[k];
            const int nc = p.PC - (a0 + p_src[k]);
            if constexpr (PF16) {
                const _Float16* src = reinterpret_cast<const _Float16*>(Pbase) + off;
                pv[k][0] = load4h(src, nc, row && gw < p.Pw);
                pv[k][1] = load4h(src + p.PCs, nc, row && gw + 1 < p.Pw);
            } else {
                const float* src = Pbase + off;
                pv[k][0] = load4f(src, nc, row && gw < p.Pw);
                pv[k][1] = load4f(src + p.PCs, nc, row && gw + 1 < p.Pw);
            }
        }
        const int q0d = p0d + od, q0h = p0h + ghmin, q0w = p0w + gwmin;
#pragma unroll
        for (int k = 0; k < QIT; ++k) {
            const int gd = qs * (q0d + q_d[k]) + pard, gh = qs * (q0h + q_h[k]) + parh, gw = qs * (q0w + q_w[k]) + parw;   // tensor coordinates
            const bool row = q_ok[k] && (unsigned)gd < (unsigned)p.Qd && (unsigned)gh < (unsigned)p.Qh;
            const size_t off = ((((size_t)bl * p.Qd + (row ? gd : 0)) * p.Qh + (row ? gh : 0)) * p.Qw + gw) * p.QCs + b0 + q_src[k];
            const int nc = p.QC - (b0 + q_src[k]);
            if constexpr (QF16) {
                const _Float16* src = reinterpret_cast<const _Float16*>(Qbase) + off;
                qv[k][0] = load4h(src, nc, row && (unsigned)gw < (unsigned)p.Qw);
                qv[k][1] = load4h(src + qs * p.QCs, nc, row && (unsigned)(gw + qs) < (unsigned)p.Qw);
            } else {
                const float* src = Qbase + off;
                qv[k][0] = load4f(src, nc, row && (unsigned)gw < (unsigned)p.Qw);
                qv[k][1] = load4f(src + qs * p.QCs, nc, row && (unsigned)(gw + qs) < (unsigned)p.Qw);
            }
        }
    };
    auto commit = [&]() {                           // registers -> scaled fp16 hi / lo pairs in LDS
        if (p.dbg & 2) return;
#pragma unroll
        for (int k = 0; k < PIT; ++k) {
            const float x0[4] = {pv[k][0].x, pv[k][0].y, pv[k][0].z, pv[k][0].w}, x1[4] = {pv[k][1].x, pv[k][1].y, pv[k][1].z, pv[k][1].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (SPLIT) {
                    unsigned h, l;
                    wg_split2(x0[j] * sP, x1[j] * sP, h, l);
                    *reinterpret_cast<unsigned*>(Ph + p_dst[k] + j * CHS_P) = h;
                    *reinterpret_cast<unsigned*>(Pl + p_dst[k] + j * CHS_P) = l;
                } else *reinterpret_cast<unsigned*>(Ph + p_dst[k] + j * CHS_P) = wg_round2(x0[j] * sP, x1[j] * sP);
            }
        }
#pragma unroll
        for (int k = 0; k < QIT; ++k) {
            if (q_ok[k]) {
                const float x0[4] = {qv[k][0].x, qv[k][0].y, qv[k][0].z, qv[k][0].w}, x1[4] = {qv[k][1].x, qv[k][1].y, qv[k][1].z, qv[k][1].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (SPLIT) {
                        unsigned h, l;
                        wg_split2(x0[j] * sQ, x1[j] * sQ, h, l);
                        *reinterpret_cast<unsigned*>(Qh + q_dst[k] + j * CHS_Q) = h;
                        *reinterpret_cast<unsigned*>(Ql + q_dst[k] + j * CHS_Q) = l;
                    } else *reinterpret_cast<unsigned*>(Qh + q_dst[k] + j * CHS_Q) = wg_round2(x0[j] * sQ, x1[j] * sQ);
                }
            }
        }
    };
    const int tw_first = sw * p.strip, tw_end = ((sw + 1) * p.strip < p.tilesW) ? (sw + 1) * p.strip : p.tilesW;
    issue_loads(tw_first);
    for (int twi = tw_first; twi < tw_end; ++twi) {
        __syncthreads();                             // the previous brick's fragment reads are done
        commit();
        __syncthreads();
        if (twi + 1 < tw_end) issue_loads(twi + 1);  // in flight during the MFMA phase below
        if (p.dbg & 4) continue;
        // ---- this wave's two K blocks of 16 positions: half-block hb = 8 consecutive w positions of one row
        uint4 ah[2], al[2];
        int qoff[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hb = (wave * 2 + i) * 2 + hh;
            ah[i] = *reinterpret_cast<const uint4*>(Ph + col * CHS_P + hb * 8);
            if constexpr (SPLIT) al[i] = *reinterpret_cast<const uint4*>(Pl + col * CHS_P + hb * 8);
            else al[i] = ah[i];
            // Q row of the same positions at dh = 0: TW = 8: hb = pd * TH + ph; TW = 16: hb = 2 * ph + (w half)
            const int prow = (TW == 8) ? hb : (hb >> 1);
            const int pd = prow / TH, ph = prow % TH;
            qoff[i] = col * CHS_Q + (pd * LHM + ph) * ROWH + ((TW == 16) ? (hb & 1) * 8 : 0);
        }
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            if (p.g_slot[tg][dh * 3] >= 0 || p.g_slot[tg][dh * 3 + 1] >= 0 || p.g_slot[tg][dh * 3 + 2] >= 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned short* qh = Qh + qoff[i] + dh * ROWH;
                    const unsigned short* ql = Ql + qoff[i] + dh * ROWH;
                    const uint4 vh = *reinterpret_cast<const uint4*>(qh), vl = SPLIT ? *reinterpret_cast<const uint4*>(ql) : vh;
                    const unsigned eh = *reinterpret_cast<const unsigned*>(qh + 8), el = SPLIT ? *reinterpret_cast<const unsigned*>(ql + 8) : eh;
                    const wf16x8 pa_h = __builtin_bit_cast(wf16x8, ah[i]), pa_l = __builtin_bit_cast(wf16x8, al[i]);
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw) {
                        if (p.g_slot[tg][dh * 3 + dw] >= 0) {
                            uint4 bh, bl;
                            if (dw == 0) { bh = vh; bl = vl; }
                            else if (dw == 1) {
                                bh = make_uint4(__builtin_amdgcn_alignbit(vh.y, vh.x, 16), __builtin_amdgcn_alignbit(vh.z, vh.y, 16),
                                                __builtin_amdgcn_alignbit(vh.w, vh.z, 16), __builtin_amdgcn_alignbit(eh, vh.w, 16));
                                bl = make_uint4(__builtin_amdgcn_alignbit(vl.y, vl.x, 16), __builtin_amdgcn_alignbit(vl.z, vl.y, 16),
                                                __builtin_amdgcn_alignbit(vl.w, vl.z, 16), __builtin_amdgcn_alignbit(el, vl.w, 16));
                            } else { bh = make_uint4(vh.y, vh.z, vh.w, eh); bl = make_uint4(vl.y, vl.z, vl.w, el); }
                            const wf16x8 qb_h = __builtin_bit_cast(wf16x8, bh), qb_l = __builtin_bit_cast(wf16x8, bl);
                            f32x16& c = acc[dh * 3 + dw];
                            if constexpr (SPLIT) {
                                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa_h, qb_l, c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa_l, qb_h, c, 0, 0, 0);
                            }
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa_h, qb_h, c, 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    // ---- partial tiles to the workspace, slot dh * 3 + dw -> tap j of the group (records and reduce kernel of the fp32 form)
    if ((p.dbg & 8) && acc[0][0] != 12345.678f) return;
    float* dst = p.ws + ((size_t)blockIdx.y * gridDim.x + bx) * (WG_TAPS * 1024) + lane;
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_h);
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
            const int j = p.g_slot[tg][dh * 3 + dw];    // uniform over the workgroup
            if (j >= 0) {
                const int t = dh * 3 + dw;
                if (wave > 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[t][r];
                }
                __syncthreads();
                if (wave == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        dst[(j * 16 + r) * 64] = (acc[t][r] + red[r * 64 + lane] + red[(16 + r) * 64 + lane] + red[(32 + r) * 64 + lane]) * inv;
                }
                __syncthreads();
            }
        }
}

// ---- multi-tile form of the f16x3 / f16 kernel (r6) ----------------------------------------------------------------------------------
// The kernel above gives a workgroup ONE 32x32 (a, b) tile and splits the brick's positions over its four waves: a 384 -> 256 layer
// runs 96 tile columns, each of which re-reads all positions of dy and x (4 GB through L2 for the batched 22-iteration gradient of the
// update block, against 0.25 GB of tensors), and every workgroup hands over after one row of bricks (a 36 KB record per ~12 bricks).
// Here a workgroup owns NA x NB tiles (128 x 64 channels at 4 x 2), one WAVE per tile, every wave multiplying all 128 positions of the
// brick (8 K blocks x 9 taps = 72 MFMAs per wave and brick), the P tile [128 pos][NA * 32] and the Q brick [.][NB * 32] staged once for
// all of them: 2.7x fewer bytes per MAC at 4 x 2, no cross-wave reduction (a wave stores its own record), and a workgroup walks a
// contiguous range of bricks over (batch, d, h, w) -- `strip` bricks, chosen so that the launch is ~2 workgroups per CU -- before it
// hands over.  Same LDS images, operand reads, records and reduce kernel as above.  Unit-stride layers only (class mode keeps the
// single-tile kernel).
// NG = 3 (3 x 3 x 3 layers with few channel tiles: the 32 -> 32 / 64 -> 32 layers of GwcNet / PSMNet, r5's "single-workgroup 27-tap" ask): the
// three tap groups (kd planes) are three groups of WAVES of ONE workgroup instead of three workgroups -- the P tile is staged once instead of
// three times and the Q brick once with a d halo (TD + 2 planes) instead of three times TD planes; wave (tile, kd) reads its plane offset.
template <int TD, int TH, int TW, int SPLIT, int PF16, int QF16, int NA, int NB, int NG = 1>
__global__ __launch_bounds__(64 * NA * NB * NG) void wgrad_mt_kernel(const WgradArgs p) {
    static_assert(!SPLIT || (!PF16 && !QF16), "fp16 tensors exist in the native f16 form only");
    static_assert(TD * TH * TW == 128 && (TW == 8 || TW == 16), "128-position bricks");
    static_assert(NG == 1 || NG == 3, "tap groups per workgroup");
    constexpr int NT = 64 * NA * NB * NG;
    constexpr int ROWH = (TW == 8) ? 16 : 24;
    constexpr int LHM = TH + 2;
    constexpr int CHS_P = 136;
    constexpr int TDQ = TD + (NG == 3 ? 2 : 0);        // Q planes in LDS
    constexpr int CHS_Q = TDQ * LHM * ROWH + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_h[];
    unsigned short* const Ph = smem_h;                              // [NA * 32][CHS_P]
    unsigned short* const Pl = Ph + NA * 32 * CHS_P;
    unsigned short* const Qh = SPLIT ? Pl + NA * 32 * CHS_P : Pl;   // [NB * 32][CHS_Q]
    unsigned short* const Ql = Qh + NB * 32 * CHS_Q;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wt = wave % (NA * NB), wgp = wave / (NA * NB);      // tile of this wave, tap group of this wave (NG = 3)
    const int wa = wt % NA, wb = wt / NA;
    const int col = lane & 31, hh = lane >> 5;
    const unsigned bx = blockIdx.x;
    const int tg = (NG == 1) ? (int)(bx % p.tgroups) : wgp;
    const int sidx = (NG == 1) ? (int)(bx / p.tgroups) : (int)bx;
    const int atiles = (p.A + 31) / 32, btiles = (p.Bc + 31) / 32;
    const int nablk = (atiles + NA - 1) / NA;
    const int ablk = blockIdx.y % nablk, bblk = blockIdx.y / nablk;
    const int a0 = ablk * NA * 32, b0 = bblk * NB * 32;
    const int od = (NG == 1) ? p.g_od[tg] : p.dmin;                // first Q plane of the staged brick (NG = 3: host guarantees g_od[g] = dmin + g)
    const float sP = p.Pmeta ? wg_pow2_scale(amax_read(p.Pmeta)) : 1.f, sQ = p.Qmeta ? wg_pow2_scale(amax_read(p.Qmeta)) : 1.f;
    const float inv = (1.0f / sP) * (1.0f / sQ);
    const int nbricks = p.B * p.tilesD * p.tilesH * p.tilesW;
    const int bi0 = sidx * p.strip, bi1 = (bi0 + p.strip < nbricks) ? bi0 + p.strip : nbricks;

    f32x16 acc[WG_TAPS];
#pragma unroll
    for (int t = 0; t < WG_TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int npq = (p.LW + 1) >> 1;                 // voxel pairs per Q row
    const int nRP = TDQ * p.LH * npq;                // (row, pair) items of the Q brick per channel quad
    // staging items: a wave's 64 consecutive items are 8 (position pairs | row pairs) x 8 channel quads of ONE 32-channel group, as in the
    // single-tile kernel (its LDS write pattern: 32-bit writes of 8 consecutive pairs, channel rows on distinct 16-byte slots)
    constexpr int PIT = (512 * NA + NT - 1) / NT, QIT = (NB * TDQ * LHM * ((TW + 3) / 2) * 8 + NT - 1) / NT;
    int p_src[PIT], p_dst[PIT], q_src[QIT], q_dst[QIT];
    unsigned q_ok[QIT], p_ok[PIT];
    int p_d[PIT], p_h[PIT], p_w[PIT], q_d[QIT], q_h[QIT], q_w[QIT];
#pragma unroll
    for (int k = 0; k < PIT; ++k) {
        const int it = tid + NT * k;
        const int c4 = it & 7, pair = (it >> 3) & 63, cg = it >> 9;
        const int q0 = pair * 2, ch = cg * 32 + c4 * 4;
        p_w[k] = q0 % TW; p_h[k] = (q0 / TW) % TH; p_d[k] = q0 / (TW * TH);
        p_src[k] = ch; p_dst[k] = ch * CHS_P + q0;
        p_ok[k] = (it < 512 * NA) ? 1u : 0u;
    }
#pragma unroll
    for (int k = 0; k < QIT; ++k) {
        const int it = tid + NT * k;
        const int c4 = it & 7; int r = it >> 3;
        const int rp = r % nRP, cg = r / nRP;
        const int pr = rp % npq; r = rp / npq;
        const int ch = cg * 32 + c4 * 4;
        q_h[k] = r % p.LH; q_d[k] = r / p.LH; q_w[k] = pr * 2;
        q_src[k] = ch; q_dst[k] = ch * CHS_Q + (q_d[k] * LHM + q_h[k]) * ROWH + q_w[k];
        q_ok[k] = (cg < NB) ? 1u : 0u;
    }
    float4 pv[PIT][2], qv[QIT][2];
    auto load4f = [&](const float* src, int nc, bool ok) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            if (nc >= 4) v = *reinterpret_cast<const float4*>(src);
            else { if (nc > 0) v.x = src[0]; if (nc > 1) v.y = src[1]; if (nc > 2) v.z = src[2]; }
        }
        return v;
    };
    auto load4h = [&](const _Float16* src, int nc, bool ok) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            if (nc >= 4) {
                const uint2 u = *reinterpret_cast<const uint2*>(src);
                const wf16x2 a = __builtin_bit_cast(wf16x2, u.x), c = __builtin_bit_cast(wf16x2, u.y);
                v = make_float4((float)a[0], (float)a[1], (float)c[0], (float)c[1]);
            } else { if (nc > 0) v.x = (float)src[0]; if (nc > 1) v.y = (float)src[1]; if (nc > 2) v.z = (float)src[2]; }
        }
        return v;
    };
    auto issue_loads = [&](int bi) {
        const int twi = bi % p.tilesW; int t = bi / p.tilesW;
        const int thi = t % p.tilesH; t /= p.tilesH;
        const int tdi = t % p.tilesD; const int b = t / p.tilesD;
        const int bl = p.ntab ? b % p.bper : b;
        const float* const Pbase = p.ntab ? static_cast<const float*>(wg_tab(offsetof(WgradArgs, Ptab), b / p.bper)) : p.P;
        const float* const Qbase = p.ntab ? static_cast<const float*>(wg_tab(offsetof(WgradArgs, Qtab), b / p.bper)) : p.Q;
        const int p0d = tdi * TD, p0h = thi * TH, p0w = twi * TW;
#pragma unroll
        for (int k = 0; k < PIT; ++k) {
            const int gd = p0d + p_d[k], gh = p0h + p_h[k], gw = p0w + p_w[k];
            const bool row = p_ok[k] && gd < p.Pd && gh < p.Ph;
            const size_t off = ((((size_t)bl * p.Pd + (row ? gd : 0)) * p.Ph + (row ? gh : 0)) * p.Pw + gw) * p.PCs + a0 + p_src[k];
            const int nc = p.PC - (a0 + p_src[k]);
            if constexpr (PF16) {
                const _Float16* src = reinterpret_cast<const _Float16*>(Pbase) + off;
                pv[k][0] = load4h(src, nc, row && gw < p.Pw);
                pv[k][1] = load4h(src + p.PCs, nc, row && gw + 1 < p.Pw);
            } else {
                const float* src = Pbase + off;
                pv[k][0] = load4f(src, nc, row && gw < p.Pw);
                pv[k][1] = load4f(src + p.PCs, nc, row && gw + 1 < p.Pw);
            }
        }
        const int q0d = p0d + od, q0h = p0h + p.hmin, q0w = p0w + p.wmin;
#pragma unroll
        for (int k = 0; k < QIT; ++k) {
            const int gd = q0d + q_d[k], gh = q0h + q_h[k], gw = q0w + q_w[k];
            const bool row = q_ok[k] && (unsigned)gd < (unsigned)p.Qd && (unsigned)gh < (unsigned)p.Qh;
            const size_t off = ((((size_t)bl * p.Qd + (row ? gd : 0)) * p.Qh + (row ? gh : 0)) * p.Qw + gw) * p.QCs + b0 + q_src[k];
            const int nc = p.QC - (b0 + q_src[k]);
            if constexpr (QF16) {
                const _Float16* src = reinterpret_cast<const _Float16*>(Qbase) + off;
                qv[k][0] = load4h(src, nc, row && (unsigned)gw < (unsigned)p.Qw);
                qv[k][1] = load4h(src + p.QCs, nc, row && (unsigned)(gw + 1) < (unsigned)p.Qw);
            } else {
                const float* src = Qbase + off;
                qv[k][0] = load4f(src, nc, row && (unsigned)gw < (unsigned)p.Qw);
                qv[k][1] = load4f(src + p.QCs, nc, row && (unsigned)(gw + 1) < (unsigned)p.Qw);
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < PIT; ++k) {
            if (!p_ok[k]) continue;
            const float x0[4] = {pv[k][0].x, pv[k][0].y, pv[k][0].z, pv[k][0].w}, x1[4] = {pv[k][1].x, pv[k][1].y, pv[k][1].z, pv[k][1].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (SPLIT) {
                    unsigned h, l;
                    wg_split2(x0[j] * sP, x1[j] * sP, h, l);
                    *reinterpret_cast<unsigned*>(Ph + p_dst[k] + j * CHS_P) = h;
                    *reinterpret_cast<unsigned*>(Pl + p_dst[k] + j * CHS_P) = l;
                } else *reinterpret_cast<unsigned*>(Ph + p_dst[k] + j * CHS_P) = wg_round2(x0[j] * sP, x1[j] * sP);
            }
        }
#pragma unroll
        for (int k = 0; k < QIT; ++k) {
            if (q_ok[k]) {
                const float x0[4] = {qv[k][0].x, qv[k][0].y, qv[k][0].z, qv[k][0].w}, x1[4] = {qv[k][1].x, qv[k][1].y, qv[k][1].z, qv[k][1].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (SPLIT) {
                        unsigned h, l;
                        wg_split2(x0[j] * sQ, x1[j] * sQ, h, l);
                        *reinterpret_cast<unsigned*>(Qh + q_dst[k] + j * CHS_Q) = h;
                        *reinterpret_cast<unsigned*>(Ql + q_dst[k] + j * CHS_Q) = l;
                    } else *reinterpret_cast<unsigned*>(Qh + q_dst[k] + j * CHS_Q) = wg_round2(x0[j] * sQ, x1[j] * sQ);
                }
            }
        }
    };
    if (bi0 < bi1) issue_loads(bi0);
    const unsigned short* const Pa_h = Ph + (wa * 32 + col) * CHS_P;
    const unsigned short* const Pa_l = Pl + (wa * 32 + col) * CHS_P;
    const int qcol = (wb * 32 + col) * CHS_Q;
    for (int bi = bi0; bi < bi1; ++bi) {
        __syncthreads();                             // the previous brick's fragment reads are done
        commit();
        __syncthreads();
        if (bi + 1 < bi1) issue_loads(bi + 1);       // in flight during the MFMA phase below
#pragma unroll 2
        for (int kb = 0; kb < 8; ++kb) {             // K block = 16 positions; this lane's half-block hb = 8 consecutive w positions of one row
            const int hb = kb * 2 + hh;
            const uint4 ah = *reinterpret_cast<const uint4*>(Pa_h + hb * 8);
            uint4 al = ah;
            if constexpr (SPLIT) al = *reinterpret_cast<const uint4*>(Pa_l + hb * 8);
            const wf16x8 pa_h = __builtin_bit_cast(wf16x8, ah), pa_l = __builtin_bit_cast(wf16x8, al);
            const int prow = (TW == 8) ? hb : (hb >> 1);
            const int pd = prow / TH, ph = prow % TH;
            const int qoff = qcol + ((pd + (NG == 3 ? wgp : 0)) * LHM + ph) * ROWH + ((TW == 16) ? (hb & 1) * 8 : 0);
#pragma unroll
            for (int dh = 0; dh < 3; ++dh) {
                if (p.g_slot[tg][dh * 3] >= 0 || p.g_slot[tg][dh * 3 + 1] >= 0 || p.g_slot[tg][dh * 3 + 2] >= 0) {
                    const unsigned short* qh = Qh + qoff + dh * ROWH;
                    const unsigned short* ql = Ql + qoff + dh * ROWH;
                    const uint4 vh = *reinterpret_cast<const uint4*>(qh), vl = SPLIT ? *reinterpret_cast<const uint4*>(ql) : vh;
                    const unsigned eh = *reinterpret_cast<const unsigned*>(qh + 8), el = SPLIT ? *reinterpret_cast<const unsigned*>(ql + 8) : eh;
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw) {
                        if (p.g_slot[tg][dh * 3 + dw] >= 0) {
                            uint4 bh, bl;
                            if (dw == 0) { bh = vh; bl = vl; }
                            else if (dw == 1) {
                                bh = make_uint4(__builtin_amdgcn_alignbit(vh.y, vh.x, 16), __builtin_amdgcn_alignbit(vh.z, vh.y, 16),
                                                __builtin_amdgcn_alignbit(vh.w, vh.z, 16), __builtin_amdgcn_alignbit(eh, vh.w, 16));
                                bl = make_uint4(__builtin_amdgcn_alignbit(vl.y, vl.x, 16), __builtin_amdgcn_alignbit(vl.z, vl.y, 16),
                                                __builtin_amdgcn_alignbit(vl.w, vl.z, 16), __builtin_amdgcn_alignbit(el, vl.w, 16));
                            } else { bh = make_uint4(vh.y, vh.z, vh.w, eh); bl = make_uint4(vl.y, vl.z, vl.w, el); }
                            const wf16x8 qb_h = __builtin_bit_cast(wf16x8, bh), qb_l = __builtin_bit_cast(wf16x8, bl);
                            f32x16& c = acc[dh * 3 + dw];
                            if constexpr (SPLIT) {
                                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa_h, qb_l, c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa_l, qb_h, c, 0, 0, 0);
                            }
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa_h, qb_h, c, 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    // ---- this wave's tile record, slot dh * 3 + dw -> tap j of the group
    const int a_tile = ablk * NA + wa, b_tile = bblk * NB + wb;
    if (a_tile < atiles && b_tile < btiles) {
        // record index as in the single-tile kernels: workgroup column = strip * tgroups + tap group
        const size_t gx = (NG == 1) ? gridDim.x : (size_t)gridDim.x * NG, bxr = (NG == 1) ? bx : (size_t)sidx * NG + tg;
        float* dst = p.ws + ((size_t)(b_tile * atiles + a_tile) * gx + bxr) * (WG_TAPS * 1024) + lane;
#pragma unroll
        for (int t = 0; t < WG_TAPS; ++t) {
            const int j = p.g_slot[tg][t];
            if (j >= 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(j * 16 + r) * 64] = acc[t][r] * inv;
            }
        }
    }
}

// Second stage: dW[a][b][tap] = sum over the workgroups (position strips) of one (tile, tap group).  One thread per (tap, tile element).
struct WgradReduceArgs {
    const float* ws; float* dW;
    int gx, tgroups, nstrips;     // workspace layout: workgroup = y * gx + strip * tgroups + tg
    int A, Bc, kvol, atiles, T, cls;
    signed char tapid[64], g_t0[16], g_nt[16];
};
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradReduceArgs p) {
    const int e = blockIdx.x * 256 + threadIdx.x;            // (t, r, lane) inside one workgroup record
    const int tg = blockIdx.y, y = blockIdx.z;
    const int t = e >> 10, r = (e >> 6) & 15, lane = e & 63;
    const int t0 = p.cls ? p.g_t0[tg] : tg * WG_TAPS;
    const int nt = p.cls ? p.g_nt[tg] : ((p.T - t0 < WG_TAPS) ? (p.T - t0) : WG_TAPS);
    if (t >= nt) return;
    const int a = (y % p.atiles) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), b = (y / p.atiles) * 32 + (lane & 31);
    if (a >= p.A || b >= p.Bc) return;
    const float* src = p.ws + ((size_t)y * p.gx + tg) * (WG_TAPS * 1024) + e;
    const size_t step = (size_t)p.tgroups * (WG_TAPS * 1024);
    // fixed association order: deterministic.  Eight independent partial sums = eight loads in flight per thread (the strips are 36 KB
    // apart: with four the kernel was latency-bound, 12.6 us for ~40 strips)
    float sp[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int i = 0;
    for (; i + 8 <= p.nstrips; i += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(i + j) * step];
#pragma unroll
        for (int j = 0; j < 8; ++j) sp[j] += v[j];
    }
    for (int j = 0; i < p.nstrips; ++i, ++j) sp[j] += src[(size_t)i * step];
    const int tt = p.cls ? p.tapid[t0 + t] : t0 + t;
    p.dW[((size_t)a * p.Bc + b) * p.kvol + tt] = ((sp[0] + sp[1]) + (sp[2] + sp[3])) + ((sp[4] + sp[5]) + (sp[6] + sp[7]));
}

}  // namespace osa

using namespace osa;

// multi-tile form of the f16x3 / f16 weight gradient (wgrad_mt_kernel); osa_conv_b_ring_mask bit 27 switches it for A/B runs and tests
static int g_wgrad_mt = 1;
namespace osa { void wgrad_set_multi_tile(int on) { g_wgrad_mt = on ? 1 : 0; } }

// conv:   P = dy [B,Do,Ho,Wo,Co]  Q = x  [B,Di,Hi,Wi,Ci]  dW [Co][Ci][k]   (transposed = 0)
// deconv: P = x  [B,Di,Hi,Wi,Ci]  Q = dy [B,Do,Ho,Wo,Co]  dW [Ci][Co][k]   (transposed = 1, stride 2, off = t - pad)
// query != nullptr: only compute the workspace size of the two-stage form for these dimensions (no pointer is touched)
static int wgrad_impl(const float* x, const float* dy, float* dw,
                      int B, int Di, int Hi, int Wi, int Ci, int xCs,
                      int Do, int Ho, int Wo, int Co, int dyCs,
                      int kd, int kh, int kw, int stride,
                      int pad_d, int pad_h, int pad_w, int dil_d, int dil_h, int dil_w,
                      int transposed, float* ws, size_t ws_bytes, size_t* query, void* stream,
                      int f16x3 = 0, const float* x_meta = nullptr, const float* dy_meta = nullptr, int x_f16 = 0, int dy_f16 = 0,
                      const void* const* x_tab = nullptr, const void* const* dy_tab = nullptr, int n_tab = 0) {
    if (!query) OSA_REQUIRE(x && dy && dw, "conv3d_wgrad: NULL pointer");
    if (x_f16 || dy_f16) OSA_REQUIRE(f16x3 == 2, "conv3d_wgrad: fp16 tensors exist in the native f16 form only");
    const int T = kd * kh * kw;
    OSA_REQUIRE(T >= 1 && T <= 64, "conv3d_wgrad: %d taps unsupported", T);
    OSA_REQUIRE(stride == 1 || stride == 2, "conv3d_wgrad: stride %d unsupported", stride);
    if (!query) OSA_REQUIRE(xCs % 4 == 0 && dyCs % 4 == 0 && ((size_t)x & (x_f16 ? 7 : 15)) == 0 && ((size_t)dy & (dy_f16 ? 7 : 15)) == 0,
                            "conv3d_wgrad: tensors must be 16-byte (fp16: 8-byte) aligned with voxel strides %% 4 == 0");
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.T = T; a.kh = kh; a.kw = kw; a.kvol = T;
    a.s = (transposed ? 2 : stride);
    const int sd = (!transposed && Di == 1 && kd == 1) ? 1 : a.s;
    OSA_REQUIRE(sd == a.s, "conv3d_wgrad: flat (D=1) strided layers are not supported yet");
    if (!transposed) {
        a.P = dy; a.Pd = Do; a.Ph = Ho; a.Pw = Wo; a.PC = Co; a.PCs = dyCs;
        a.Q = x; a.Qd = Di; a.Qh = Hi; a.Qw = Wi; a.QC = Ci; a.QCs = xCs;
        a.A = Co; a.Bc = Ci;
    } else {
        OSA_REQUIRE(stride == 2, "conv3d_wgrad: transposed convs are stride 2");
        a.P = x; a.Pd = Di; a.Ph = Hi; a.Pw = Wi; a.PC = Ci; a.PCs = xCs;
        a.Q = dy; a.Qd = Do; a.Qh = Ho; a.Qw = Wo; a.QC = Co; a.QCs = dyCs;
        a.A = Ci; a.Bc = Co;
    }
    a.dW = dw;
    if (n_tab > 0) {
        OSA_REQUIRE(n_tab <= 24 && B % n_tab == 0 && x_tab && dy_tab, "conv3d_wgrad: a tensor list holds at most 24 equally shaped items (got %d, batch %d)", n_tab, B);
        a.ntab = n_tab; a.bper = B / n_tab;
        for (int i = 0; i < n_tab; ++i) {
            OSA_REQUIRE(x_tab[i] && dy_tab[i] && ((size_t)x_tab[i] & (x_f16 ? 7 : 15)) == 0 && ((size_t)dy_tab[i] & (dy_f16 ? 7 : 15)) == 0, "conv3d_wgrad: list item %d NULL or misaligned", i);
            a.Ptab[i] = transposed ? x_tab[i] : dy_tab[i];
            a.Qtab[i] = transposed ? dy_tab[i] : x_tab[i];
        }
    }
    int t = 0, dmax = -128, hmax = -128, wmax = -128;
    a.dmin = a.hmin = a.wmin = 127;
    for (int z = 0; z < kd; ++z) for (int y = 0; y < kh; ++y) for (int xx = 0; xx < kw; ++xx, ++t) {
        const int od = transposed ? z - pad_d : z * dil_d - pad_d;
        const int oh = transposed ? y - pad_h : y * dil_h - pad_h;
        const int ow = transposed ? xx - pad_w : xx * dil_w - pad_w;
        a.od[t] = (signed char)od; a.oh[t] = (signed char)oh; a.ow[t] = (signed char)ow;
        a.dmin = od < a.dmin ? od : a.dmin; dmax = od > dmax ? od : dmax;
        a.hmin = oh < a.hmin ? oh : a.hmin; hmax = oh > hmax ? oh : hmax;
        a.wmin = ow < a.wmin ? ow : a.wmin; wmax = ow > wmax ? ow : wmax;
    }
    a.tgroups = cdiv(T, WG_TAPS);
    if (a.s == 2) {
        // class-major tap list (transposed convs AND stride-2 convs: both read Q at 2*pos + off): for parity (pd, ph, pw) every kernel
        // index whose offset has that parity; delta = floor(off / 2)
        a.cls = 1;
        int n = 0;
        for (int c = 0; c < 8; ++c) {
            const int par[3] = {(c >> 2) & 1, (c >> 1) & 1, c & 1};
            const int kk[3] = {kd, kh, kw}, pd3[3] = {pad_d, pad_h, pad_w}, dl3[3] = {transposed ? 1 : dil_d, transposed ? 1 : dil_h, transposed ? 1 : dil_w};
            int idx[3][4], del[3][4], cnt[3];
            for (int dim = 0; dim < 3; ++dim) {
                cnt[dim] = 0;
                for (int k = 0; k < kk[dim]; ++k) {
                    const int off = k * dl3[dim] - pd3[dim];
                    if (((off % 2) + 2) % 2 != par[dim]) continue;
                    idx[dim][cnt[dim]] = k; del[dim][cnt[dim]] = (off - par[dim]) / 2; ++cnt[dim];      // off - par is even
                }
                for (int i = 0; i < cnt[dim]; ++i)
                    OSA_REQUIRE(del[dim][i] == -par[dim] || del[dim][i] == 1 - par[dim], "conv3d_wgrad: transposed kernel %d / pad %d out of the supported range", kk[dim], pd3[dim]);
            }
            a.g_t0[c] = (signed char)n; a.g_par[c] = (signed char)c;
            for (int i = 0; i < cnt[0]; ++i) for (int j = 0; j < cnt[1]; ++j) for (int l = 0; l < cnt[2]; ++l, ++n) {
                a.od[n] = (signed char)del[0][i]; a.oh[n] = (signed char)del[1][j]; a.ow[n] = (signed char)del[2][l];
                a.tapid[n] = (signed char)((idx[0][i] * kh + idx[1][j]) * kw + idx[2][l]);
            }
            a.g_nt[c] = (signed char)(n - a.g_t0[c]);
            OSA_REQUIRE(a.g_nt[c] <= WG_TAPS, "conv3d_wgrad: %d taps in one parity class", (int)a.g_nt[c]);
        }
        OSA_REQUIRE(n == T, "conv3d_wgrad: class decomposition covers %d of %d taps", n, T);
        a.tgroups = 8;
    }
    hipStream_t st = (hipStream_t)stream;
    if (f16x3) {
        // split-precision form (wgrad_f16x3_kernel): unit dilation; unit stride with <= 3x3 planes, or the stride-2 / transposed layers in
        // class mode (the parity classes built above).  A tap group = the taps of one d offset (of one class).  Not eligible -> *query = 0 /
        // error: the caller keeps the fp32 form.
        bool ok = dil_d == 1 && dil_h == 1 && dil_w == 1;
        if (a.s == 1) ok = ok && !transposed && kh <= 3 && kw <= 3;
        if (query && !ok) { *query = 0; return 0; }
        OSA_REQUIRE(ok, "conv3d_wgrad_f16x3: layer not eligible (unit dilation; unit-stride layers: kh, kw <= 3)");
        const bool flat16 = (a.Pd == 1 && kd == 1);
        const int TD = flat16 ? 1 : 2, TH = 8, TW = flat16 ? 16 : 8;
        int ng = 0;
        memset(a.g_slot, -1, sizeof(a.g_slot));
        if (a.cls) {
            // split every class's (d-major) tap run by its d delta
            signed char c_t0[8], c_nt[8];
            memcpy(c_t0, a.g_t0, 8); memcpy(c_nt, a.g_nt, 8);
            for (int c = 0; c < 8; ++c) {
                const int parh = (c >> 1) & 1, parw = c & 1;
                int i = c_t0[c];
                const int e = c_t0[c] + c_nt[c];
                while (i < e) {
                    int j = i;
                    while (j < e && a.od[j] == a.od[i]) ++j;
                    OSA_REQUIRE(ng < 16, "conv3d_wgrad_f16x3: more than 16 tap groups");
                    a.g_t0[ng] = (signed char)i; a.g_nt[ng] = (signed char)(j - i); a.g_par[ng] = (signed char)c; a.g_od[ng] = a.od[i];
                    for (int t2 = i; t2 < j; ++t2) {
                        const int dh = a.oh[t2] + parh, dw = a.ow[t2] + parw;            // delta - (-par) in {0, 1}
                        OSA_REQUIRE(dh >= 0 && dh <= 2 && dw >= 0 && dw <= 2 && a.g_slot[ng][dh * 3 + dw] < 0, "conv3d_wgrad_f16x3: tap offsets out of range");
                        a.g_slot[ng][dh * 3 + dw] = (signed char)(t2 - i);
                    }
                    ++ng; i = j;
                }
            }
            a.LH = TH + 1; a.LW = TW + 1;
        } else {
            // taps are (z, y, x)-major: group = one z
            const int khw = kh * kw;
            for (int z = 0; z < kd; ++z) {
                a.g_t0[ng] = (signed char)(z * khw); a.g_nt[ng] = (signed char)khw; a.g_par[ng] = 0; a.g_od[ng] = a.od[z * khw];
                for (int y = 0; y < kh; ++y) for (int xx = 0; xx < kw; ++xx) a.g_slot[ng][y * 3 + xx] = (signed char)(y * kw + xx);
                ++ng;
            }
            a.LH = TH + (hmax - a.hmin); a.LW = TW + (wmax - a.wmin);
        }
        a.LD = TD;
        a.tgroups = ng;
        a.tilesD = cdiv(a.Pd, TD); a.tilesH = cdiv(a.Ph, TH); a.tilesW = cdiv(a.Pw, TW);
        const int chs_q = TD * (TH + 2) * (flat16 ? 24 : 16) + 8;
        const int planes = (f16x3 == 2) ? 1 : 2;              // hi (+ lo) planes of P and Q
        const size_t lds_ops = (size_t)(planes * 32 * 136 + planes * 32 * chs_q) * sizeof(unsigned short);
        const size_t lds = lds_ops > (size_t)3 * 16 * 64 * sizeof(float) ? lds_ops : (size_t)3 * 16 * 64 * sizeof(float);   // (the hand-over of the partial tiles reuses it: 3 waves x 16 x 64 floats)
        const int gy = cdiv(a.A, 32) * cdiv(a.Bc, 32);
        if (!a.cls && g_wgrad_mt && !flat16 && kd == 3 && a.tgroups == 3 && cdiv(a.A, 32) * cdiv(a.Bc, 32) == 1 &&
            a.g_od[0] == a.dmin && a.g_od[1] == a.dmin + 1 && a.g_od[2] == a.dmin + 2) {
            // few channel tiles, 3 x 3 x 3 taps: ONE workgroup per (tiles, brick range) with the three kd planes as three groups of waves (NG = 3)
            const int atl = cdiv(a.A, 32), btl = cdiv(a.Bc, 32);
            const int NA = 1, NB = 1;                                       // (two-tile layers spill at 6 waves / 256 registers: they keep the single-tile kernel)
            const long long nbricks = (long long)B * a.tilesD * a.tilesH * a.tilesW;
            OSA_REQUIRE(nbricks < (1ll << 30), "conv3d_wgrad_f16x3: too many position bricks");
            long long nstr = 2 * 256ll;
            if (nstr > cdiv((int)nbricks, 4)) nstr = cdiv((int)nbricks, 4);
            if (nstr < 1) nstr = 1;
            a.strip = exp_int("OSA_WGRAD_STRIP", (int)cdiv((int)nbricks, (int)nstr));
            if (a.strip < 1) a.strip = 1;
            nstr = cdiv((int)nbricks, a.strip);
            const long long gx = nstr * 3;
            const size_t need = (size_t)gx * gy * WG_TAPS * 1024 * sizeof(float);
            if (query) { *query = need; return 0; }
            OSA_REQUIRE(ws && ws_bytes >= need && ((size_t)ws & 15) == 0, "conv3d_wgrad_f16x3: workspace of %zu B needed (got %zu)", need, ws_bytes);
            if (f16x3 == 1) OSA_REQUIRE(x_meta && dy_meta, "conv3d_wgrad_f16x3: range blocks of x and dy required");
            a.ws = ws;
            a.Pmeta = dy_meta; a.Qmeta = x_meta;
            a.Pf16 = dy_f16; a.Qf16 = x_f16;
            const int chs_q3 = (TD + 2) * (TH + 2) * 16 + 8;
            const size_t lds_mt = (size_t)planes * (NA * 32 * 136 + NB * 32 * chs_q3) * sizeof(unsigned short);
            OSA_REQUIRE(lds_mt <= 160 * 1024, "conv3d_wgrad_f16x3: %zu B of LDS", lds_mt);
            dim3 grid((unsigned)nstr, 1), block(64 * NA * NB * 3);
#define OSA_WG_G3_LAUNCH(...)                                                                                                         \
            do { (void)hipFuncSetAttribute((const void*)wgrad_mt_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mt); \
                 hipLaunchKernelGGL((wgrad_mt_kernel<__VA_ARGS__>), grid, block, lds_mt, st, a); } while (0)
#define OSA_WG_G3_T(NA_, NB_)                                                                                                         \
            if (f16x3 == 1) OSA_WG_G3_LAUNCH(2, 8, 8, 1, 0, 0, NA_, NB_, 3);                                                           \
            else switch (a.Pf16 * 2 + a.Qf16) {                                                                                       \
                case 0: OSA_WG_G3_LAUNCH(2, 8, 8, 0, 0, 0, NA_, NB_, 3); break;                                                       \
                case 1: OSA_WG_G3_LAUNCH(2, 8, 8, 0, 0, 1, NA_, NB_, 3); break;                                                       \
                case 2: OSA_WG_G3_LAUNCH(2, 8, 8, 0, 1, 0, NA_, NB_, 3); break;                                                       \
                default: OSA_WG_G3_LAUNCH(2, 8, 8, 0, 1, 1, NA_, NB_, 3); break;                                                      \
            }
            OSA_WG_G3_T(1, 1)
#undef OSA_WG_G3_T
#undef OSA_WG_G3_LAUNCH
            OSA_LAUNCH_CHECK("conv3d_wgrad_mt_g3");
            WgradReduceArgs r;
            memset(&r, 0, sizeof(r));
            r.ws = ws; r.dW = dw; r.gx = (int)gx; r.tgroups = 3; r.nstrips = (int)nstr;
            r.A = a.A; r.Bc = a.Bc; r.kvol = T; r.atiles = atl; r.T = T;
            r.cls = 1;
            for (int t2 = 0; t2 < T; ++t2) r.tapid[t2] = (signed char)t2;
            memcpy(r.g_t0, a.g_t0, sizeof(r.g_t0)); memcpy(r.g_nt, a.g_nt, sizeof(r.g_nt));
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(WG_TAPS * 1024 / 256, 3, gy), dim3(256), 0, st, r);
            OSA_LAUNCH_CHECK("conv3d_wgrad_mt_g3_reduce");
            return 0;
        }
        if (!a.cls && g_wgrad_mt && cdiv(a.A, 32) * cdiv(a.Bc, 32) >= 4) {
            // multi-tile form (wgrad_mt_kernel): NA x NB tiles per workgroup, a contiguous range of `strip` bricks per workgroup
            const int atl = cdiv(a.A, 32), btl = cdiv(a.Bc, 32);
            const int NA = (atl >= 3) ? 4 : 2, NB = 2;
            const int gyb = cdiv(atl, NA) * cdiv(btl, NB);
            const long long nbricks = (long long)B * a.tilesD * a.tilesH * a.tilesW;
            OSA_REQUIRE(nbricks < (1ll << 30), "conv3d_wgrad_f16x3: too many position bricks");
            // ~2 workgroups per CU over the whole launch, at least 4 bricks each (a hand-over is a 36 KB record per wave)
            long long nstr = (2 * 256ll) / ((long long)gyb * a.tgroups);
            if (nstr < 1) nstr = 1;
            if (nstr > cdiv((int)nbricks, 4)) nstr = cdiv((int)nbricks, 4);
            a.strip = exp_int("OSA_WGRAD_STRIP", (int)cdiv((int)nbricks, (int)nstr));
            if (a.strip < 1) a.strip = 1;
            nstr = cdiv((int)nbricks, a.strip);
            const long long gx = nstr * a.tgroups;
            const size_t need = (size_t)gx * gy * WG_TAPS * 1024 * sizeof(float);
            if (query) { *query = need; return 0; }
            OSA_REQUIRE(ws && ws_bytes >= need && ((size_t)ws & 15) == 0, "conv3d_wgrad_f16x3: workspace of %zu B needed (got %zu)", need, ws_bytes);
            if (f16x3 == 1) OSA_REQUIRE(x_meta && dy_meta, "conv3d_wgrad_f16x3: range blocks of x and dy required");
            a.ws = ws;
            a.Pmeta = dy_meta; a.Qmeta = x_meta;              // (unit-stride conv: P = dy, Q = x)
            a.Pf16 = dy_f16; a.Qf16 = x_f16;
            const size_t lds_mt = (size_t)planes * (NA * 32 * 136 + NB * 32 * chs_q) * sizeof(unsigned short);
            OSA_REQUIRE(lds_mt <= 160 * 1024, "conv3d_wgrad_f16x3: %zu B of LDS", lds_mt);
            dim3 grid((unsigned)gx, gyb), block(64 * NA * NB);
#define OSA_WG_MT_LAUNCH(...)                                                                                                         \
            do { (void)hipFuncSetAttribute((const void*)wgrad_mt_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mt); \
                 hipLaunchKernelGGL((wgrad_mt_kernel<__VA_ARGS__>), grid, block, lds_mt, st, a); } while (0)
#define OSA_WG_MT_T(TD_, TH_, TW_, NA_)                                                                                               \
            if (f16x3 == 1) OSA_WG_MT_LAUNCH(TD_, TH_, TW_, 1, 0, 0, NA_, 2);                                                          \
            else switch (a.Pf16 * 2 + a.Qf16) {                                                                                       \
                case 0: OSA_WG_MT_LAUNCH(TD_, TH_, TW_, 0, 0, 0, NA_, 2); break;                                                      \
                case 1: OSA_WG_MT_LAUNCH(TD_, TH_, TW_, 0, 0, 1, NA_, 2); break;                                                      \
                case 2: OSA_WG_MT_LAUNCH(TD_, TH_, TW_, 0, 1, 0, NA_, 2); break;                                                      \
                default: OSA_WG_MT_LAUNCH(TD_, TH_, TW_, 0, 1, 1, NA_, 2); break;                                                     \
            }
            if (flat16) { if (NA == 4) { OSA_WG_MT_T(1, 8, 16, 4) } else { OSA_WG_MT_T(1, 8, 16, 2) } }
            else        { if (NA == 4) { OSA_WG_MT_T(2, 8, 8, 4) } else { OSA_WG_MT_T(2, 8, 8, 2) } }
#undef OSA_WG_MT_T
#undef OSA_WG_MT_LAUNCH
            OSA_LAUNCH_CHECK("conv3d_wgrad_mt");
            WgradReduceArgs r;
            memset(&r, 0, sizeof(r));
            r.ws = ws; r.dW = dw; r.gx = (int)gx; r.tgroups = a.tgroups; r.nstrips = (int)nstr;
            r.A = a.A; r.Bc = a.Bc; r.kvol = T; r.atiles = atl; r.T = T;
            r.cls = 1;
            for (int t2 = 0; t2 < T; ++t2) r.tapid[t2] = (signed char)t2;
            memcpy(r.g_t0, a.g_t0, sizeof(r.g_t0)); memcpy(r.g_nt, a.g_nt, sizeof(r.g_nt));
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(WG_TAPS * 1024 / 256, a.tgroups, gy), dim3(256), 0, st, r);
            OSA_LAUNCH_CHECK("conv3d_wgrad_mt_reduce");
            return 0;
        }
        const long long rows = (long long)B * a.tilesD * a.tilesH * a.tgroups * gy;
        const long long slots = 256ll * 2;                    // (registers: 9 accumulator sets keep both forms at 2 workgroups per CU)
        long long best = -1; int best_strip = a.tilesW;
        for (int strip = a.tilesW; strip >= 1; --strip) {
            const long long wgs = rows * cdiv(a.tilesW, strip);
            const long long cost = (long long)cdiv(wgs, slots) * (strip + 1);
            if (best < 0 || cost < best) { best = cost; best_strip = strip; }
        }
        a.strip = exp_int("OSA_WGRAD_STRIP", best_strip);
        if (a.strip < 1 || a.strip > a.tilesW) a.strip = best_strip;
        const long long gx = (long long)B * a.tilesD * a.tilesH * cdiv(a.tilesW, a.strip) * a.tgroups;
        OSA_REQUIRE(gx < (1ll << 31) && gy <= 65535, "conv3d_wgrad_f16x3: grid too large");
        const size_t need = (size_t)gx * gy * WG_TAPS * 1024 * sizeof(float);
        if (query) { *query = need; return 0; }
        OSA_REQUIRE(ws && ws_bytes >= need && ((size_t)ws & 15) == 0, "conv3d_wgrad_f16x3: workspace of %zu B needed (got %zu)", need, ws_bytes);
        if (f16x3 == 1) OSA_REQUIRE(x_meta && dy_meta, "conv3d_wgrad_f16x3: range blocks of x and dy required");
        a.ws = ws;
        a.Pmeta = transposed ? x_meta : dy_meta; a.Qmeta = transposed ? dy_meta : x_meta;       // conv: P = dy, Q = x; transposed: P = x, Q = dy
        a.Pf16 = transposed ? x_f16 : dy_f16; a.Qf16 = transposed ? dy_f16 : x_f16;
        a.dbg = exp_int("OSA_WG_DBG", 0);
        dim3 grid((unsigned)gx, gy), block(256);
        if (f16x3 == 2) {
#define OSA_WG_F16(TD_, TH_, TW_)                                                                                                    \
            switch (a.Pf16 * 2 + a.Qf16) {                                                                                           \
                case 0: hipLaunchKernelGGL((wgrad_f16x3_kernel<TD_, TH_, TW_, 0, 0, 0>), grid, block, lds, st, a); break;              \
                case 1: hipLaunchKernelGGL((wgrad_f16x3_kernel<TD_, TH_, TW_, 0, 0, 1>), grid, block, lds, st, a); break;              \
                case 2: hipLaunchKernelGGL((wgrad_f16x3_kernel<TD_, TH_, TW_, 0, 1, 0>), grid, block, lds, st, a); break;              \
                default: hipLaunchKernelGGL((wgrad_f16x3_kernel<TD_, TH_, TW_, 0, 1, 1>), grid, block, lds, st, a); break;             \
            }
            if (flat16) { OSA_WG_F16(1, 8, 16) } else { OSA_WG_F16(2, 8, 8) }
#undef OSA_WG_F16
        } else {
            if (flat16) hipLaunchKernelGGL((wgrad_f16x3_kernel<1, 8, 16, 1>), grid, block, lds, st, a);
            else hipLaunchKernelGGL((wgrad_f16x3_kernel<2, 8, 8, 1>), grid, block, lds, st, a);
        }
        OSA_LAUNCH_CHECK("conv3d_wgrad_f16x3");
        WgradReduceArgs r;
        memset(&r, 0, sizeof(r));
        r.ws = ws; r.dW = dw; r.gx = (int)gx; r.tgroups = a.tgroups; r.nstrips = (int)(gx / a.tgroups);
        r.A = a.A; r.Bc = a.Bc; r.kvol = T; r.atiles = cdiv(a.A, 32); r.T = T;
        // the reduce kernel's class mode reads (first tap, count) per group and maps class-major taps back through tapid: used for BOTH modes
        // here (unit stride: identity tapid), because a group is one z plane of kh * kw taps, not a run of WG_TAPS
        r.cls = 1;
        if (a.cls) memcpy(r.tapid, a.tapid, sizeof(r.tapid));
        else for (int t2 = 0; t2 < T; ++t2) r.tapid[t2] = (signed char)t2;
        memcpy(r.g_t0, a.g_t0, sizeof(r.g_t0)); memcpy(r.g_nt, a.g_nt, sizeof(r.g_nt));
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(WG_TAPS * 1024 / 256, a.tgroups, gy), dim3(256), 0, st, r);
        OSA_LAUNCH_CHECK("conv3d_wgrad_f16x3_reduce");
        return 0;
    }
    const bool two_stage = query || ws;
    if (!two_stage) {
        hipError_t e = hipMemsetAsync(dw, 0, (size_t)a.A * a.Bc * T * sizeof(float), st);
        OSA_REQUIRE(e == hipSuccess, "conv3d_wgrad: memset failed: %s", hipGetErrorString(e));
    }
    // Position brick: 4x8x8 (256 positions, 1 workgroup per CU with its ~95-122 KB of LDS) or 2x8x8 (128 positions, 2 per CU: the
    // staging of one overlaps the MFMAs of the other).  `OSA_WGRAD_TD` (experiments build) forces one.
    // Flat (D = 1) layers -- the 2-D convs of the update block and of the upsampling heads -- take a 1x16x16 brick: a 4x8x8 one would
    // spend three of its four planes on padding.
    const bool flat = (a.Pd == 1 && kd == 1);
    const int TW = flat ? 16 : 8, TH = flat ? 16 : 8;
    int TD = flat ? 1 : exp_int("OSA_WGRAD_TD", 4);
    if (!flat && TD != 2 && TD != 4) TD = 4;
    a.LD = (TD - 1) * a.s + (dmax - a.dmin) + 1;
    a.LH = (TH - 1) * a.s + (hmax - a.hmin) + 1;
    a.LW = (TW - 1) * a.s + (wmax - a.wmin) + 1;
    if (a.cls) { a.LD = TD + 1; a.LH = TH + 1; a.LW = TW + 1; }          // class sub-lattice: delta in {dmin, dmin + 1}
    a.tilesD = cdiv(a.Pd, TD); a.tilesH = cdiv(a.Ph, TH); a.tilesW = cdiv(a.Pw, TW);
    const size_t lds = ((size_t)TD * TH * TW + (size_t)a.LD * a.LH * a.LW) * WG_PS * sizeof(float);
    OSA_REQUIRE(lds <= 160 * 1024, "conv3d_wgrad: %zu B of LDS needed", lds);
    const int gy = cdiv(a.A, 32) * cdiv(a.Bc, 32);
    // Strip = consecutive w-bricks one workgroup accumulates before it hands its partial tiles over.  Atomics form: a whole row per
    // workgroup (every extra workgroup costs a round of ~9 K contended float atomics).  Two-stage form: the hand-over is 36 KB of plain
    // stores, so the strip is chosen for load balance -- the one that minimises rounds x (strip + 1), the serial brick count of the
    // busiest CU (a whole row left e.g. 288 workgroups for 256 CUs: two rounds, the second nearly empty).
    a.strip = a.tilesW;
    if (two_stage) {
        const long long rows = (long long)B * a.tilesD * a.tilesH * a.tgroups * gy;
        const long long slots = 256ll * ((lds <= 80 * 1024) ? 2 : 1);
        long long best = -1; int best_strip = a.tilesW;
        for (int strip = a.tilesW; strip >= 1; --strip) {
            const long long wgs = rows * cdiv(a.tilesW, strip);
            const long long cost = (long long)cdiv(wgs, slots) * (strip + 1);    // + 1: the atomics round of a workgroup costs about one brick
            if (best < 0 || cost < best) { best = cost; best_strip = strip; }
        }
        a.strip = exp_int("OSA_WGRAD_STRIP", best_strip);
        if (a.strip < 1 || a.strip > a.tilesW) a.strip = best_strip;
    }
    const long long gx = (long long)B * a.tilesD * a.tilesH * cdiv(a.tilesW, a.strip) * a.tgroups;
    OSA_REQUIRE(gx < (1ll << 31) && gy <= 65535, "conv3d_wgrad: grid too large");
    const size_t need = (size_t)gx * gy * WG_TAPS * 1024 * sizeof(float);
    if (query) { *query = need; return 0; }
    if (ws) OSA_REQUIRE(ws_bytes >= need && ((size_t)ws & 15) == 0, "conv3d_wgrad: workspace of %zu B needed (got %zu)", need, ws_bytes);
    a.ws = ws;
    dim3 grid((unsigned)gx, gy), block(256);
    if (flat) {
        (void)hipFuncSetAttribute((const void*)wgrad_kernel<1, 16, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((wgrad_kernel<1, 16, 16>), grid, block, lds, st, a);
    } else if (TD == 2) {
        (void)hipFuncSetAttribute((const void*)wgrad_kernel<2, 8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((wgrad_kernel<2, 8, 8>), grid, block, lds, st, a);
    } else {
        (void)hipFuncSetAttribute((const void*)wgrad_kernel<4, 8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((wgrad_kernel<4, 8, 8>), grid, block, lds, st, a);
    }
    OSA_LAUNCH_CHECK("conv3d_wgrad");
    if (ws) {
        WgradReduceArgs r;
        memset(&r, 0, sizeof(r));
        r.ws = ws; r.dW = dw; r.gx = (int)gx; r.tgroups = a.tgroups; r.nstrips = (int)(gx / a.tgroups);
        r.A = a.A; r.Bc = a.Bc; r.kvol = T; r.atiles = cdiv(a.A, 32); r.T = T; r.cls = a.cls;
        memcpy(r.tapid, a.tapid, sizeof(r.tapid)); memcpy(r.g_t0, a.g_t0, sizeof(r.g_t0)); memcpy(r.g_nt, a.g_nt, sizeof(r.g_nt));
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(WG_TAPS * 1024 / 256, a.tgroups, gy), dim3(256), 0, st, r);
        OSA_LAUNCH_CHECK("conv3d_wgrad_reduce");
    }
    return 0;
}

extern "C" int osa_conv3d_wgrad_f32(const float* x, const float* dy, float* dw,
                                    int B, int Di, int Hi, int Wi, int Ci, int xCs,
                                    int Do, int Ho, int Wo, int Co, int dyCs,
                                    int kd, int kh, int kw, int stride,
                                    int pad_d, int pad_h, int pad_w, int dil_d, int dil_h, int dil_w,
                                    int transposed, void* stream) {
    return wgrad_impl(x, dy, dw, B, Di, Hi, Wi, Ci, xCs, Do, Ho, Wo, Co, dyCs, kd, kh, kw, stride, pad_d, pad_h, pad_w,
                      dil_d, dil_h, dil_w, transposed, nullptr, 0, nullptr, stream);
}

extern "C" size_t osa_conv3d_wgrad_workspace_bytes(int B, int Di, int Hi, int Wi, int Ci, int Do, int Ho, int Wo, int Co,
                                                   int kd, int kh, int kw, int stride, int pad_d, int pad_h, int pad_w,
                                                   int dil_d, int dil_h, int dil_w, int transposed) {
    size_t need = 0;
    if (wgrad_impl(nullptr, nullptr, nullptr, B, Di, Hi, Wi, Ci, 4, Do, Ho, Wo, Co, 4, kd, kh, kw, stride, pad_d, pad_h, pad_w,
                   dil_d, dil_h, dil_w, transposed, nullptr, 0, &need, nullptr)) return 0;
    return need;
}

extern "C" int osa_conv3d_wgrad_ws_f32(const float* x, const float* dy, float* dw,
                                       int B, int Di, int Hi, int Wi, int Ci, int xCs,
                                       int Do, int Ho, int Wo, int Co, int dyCs,
                                       int kd, int kh, int kw, int stride,
                                       int pad_d, int pad_h, int pad_w, int dil_d, int dil_h, int dil_w,
                                       int transposed, float* workspace, size_t workspace_bytes, void* stream) {
    OSA_REQUIRE(workspace, "conv3d_wgrad_ws: NULL workspace (osa_conv3d_wgrad_workspace_bytes gives its size)");
    return wgrad_impl(x, dy, dw, B, Di, Hi, Wi, Ci, xCs, Do, Ho, Wo, Co, dyCs, kd, kh, kw, stride, pad_d, pad_h, pad_w,
                      dil_d, dil_h, dil_w, transposed, workspace, workspace_bytes, nullptr, stream);
}

/* split-precision (f16x3) weight gradient, two-stage form only.  workspace_bytes query returns 0 for layers the form does not cover
 * (strided / dilated / transposed layers, kernels wider than 3): the caller then uses osa_conv3d_wgrad_ws_f32. */
extern "C" size_t osa_conv3d_wgrad_f16x3_workspace_bytes(int B, int Di, int Hi, int Wi, int Ci, int Do, int Ho, int Wo, int Co,
                                                         int kd, int kh, int kw, int stride, int pad_d, int pad_h, int pad_w,
                                                         int dil_d, int dil_h, int dil_w, int transposed) {
    size_t need = 0;
    if (wgrad_impl(nullptr, nullptr, nullptr, B, Di, Hi, Wi, Ci, 4, Do, Ho, Wo, Co, 4, kd, kh, kw, stride, pad_d, pad_h, pad_w,
                   dil_d, dil_h, dil_w, transposed, nullptr, 0, &need, nullptr, 1)) return 0;
    return need;
}

extern "C" int osa_conv3d_wgrad_ws_f16x3(const float* x, const float* dy, float* dw,
                                         int B, int Di, int Hi, int Wi, int Ci, int xCs,
                                         int Do, int Ho, int Wo, int Co, int dyCs,
                                         int kd, int kh, int kw, int stride,
                                         int pad_d, int pad_h, int pad_w, int dil_d, int dil_h, int dil_w,
                                         int transposed, const float* x_meta, const float* dy_meta,
                                         float* workspace, size_t workspace_bytes, void* stream) {
    OSA_REQUIRE(workspace, "conv3d_wgrad_ws_f16x3: NULL workspace (osa_conv3d_wgrad_f16x3_workspace_bytes gives its size)");
    return wgrad_impl(x, dy, dw, B, Di, Hi, Wi, Ci, xCs, Do, Ho, Wo, Co, dyCs, kd, kh, kw, stride, pad_d, pad_h, pad_w,
                      dil_d, dil_h, dil_w, transposed, workspace, workspace_bytes, nullptr, stream, 1, x_meta, dy_meta);
}

/* native f16 weight gradient (r5): the arithmetic of the reference's AMP training -- fp16 operands (rounded to nearest even when staged),
 * one MFMA per product, fp32 accumulation.  Same layers, workspace size and two-stage reduction as the f16x3 form; x_meta / dy_meta may be
 * NULL (no operand scaling, as under autocast: GradScaler owns the range) or range blocks (power-of-two scaling, undone exactly). */
extern "C" int osa_conv3d_wgrad_ws_f16(const void* x, const void* dy, float* dw,
                                       int B, int Di, int Hi, int Wi, int Ci, int xCs,
                                       int Do, int Ho, int Wo, int Co, int dyCs,
                                       int kd, int kh, int kw, int stride,
                                       int pad_d, int pad_h, int pad_w, int dil_d, int dil_h, int dil_w,
                                       int transposed, const float* x_meta, const float* dy_meta, int x_f16, int dy_f16,
                                       float* workspace, size_t workspace_bytes, void* stream) {
    OSA_REQUIRE(workspace, "conv3d_wgrad_ws_f16: NULL workspace (osa_conv3d_wgrad_f16x3_workspace_bytes gives its size)");
    return wgrad_impl(static_cast<const float*>(x), static_cast<const float*>(dy), dw, B, Di, Hi, Wi, Ci, xCs, Do, Ho, Wo, Co, dyCs, kd, kh, kw, stride,
                      pad_d, pad_h, pad_w, dil_d, dil_h, dil_w, transposed, workspace, workspace_bytes, nullptr, stream, 2, x_meta, dy_meta,
                      x_f16 ? 1 : 0, dy_f16 ? 1 : 0);
}

/* The weight gradient over a LIST of equally shaped (x, dy) tensor pairs -- the uses of one weight within a training step (the update
 * block applies every convolution once per GRU iteration: 22 pairs) -- in ONE launch, without concatenating them: batch entry b of the
 * launch is entry b % (B / n_items) of item b / (B / n_items).  form: 0 = exact fp32 kernel, 1 = f16x3 (range blocks required), 2 = native
 * f16 (x_f16 / dy_f16 as in osa_conv3d_wgrad_ws_f16).  B = total batch over all items; workspace as for one tensor of batch B. */
extern "C" int osa_conv3d_wgrad_ws_multi(int form, const void* const* xs, const void* const* dys, int n_items, float* dw,
                                         int B, int Di, int Hi, int Wi, int Ci, int xCs,
                                         int Do, int Ho, int Wo, int Co, int dyCs,
                                         int kd, int kh, int kw, int stride,
                                         int pad_d, int pad_h, int pad_w, int dil_d, int dil_h, int dil_w,
                                         int transposed, const float* x_meta, const float* dy_meta, int x_f16, int dy_f16,
                                         float* workspace, size_t workspace_bytes, void* stream) {
    OSA_REQUIRE(workspace && xs && dys && n_items >= 1 && n_items <= 24, "conv3d_wgrad_ws_multi: NULL workspace / lists, or n_items %d outside 1..24", n_items);
    OSA_REQUIRE(form >= 0 && form <= 2 && (form == 2 || (!x_f16 && !dy_f16)), "conv3d_wgrad_ws_multi: form %d (fp16 tensors exist in form 2 only)", form);
    return wgrad_impl(static_cast<const float*>(xs[0]), static_cast<const float*>(dys[0]), dw, B, Di, Hi, Wi, Ci, xCs, Do, Ho, Wo, Co, dyCs, kd, kh, kw, stride,
                      pad_d, pad_h, pad_w, dil_d, dil_h, dil_w, transposed, workspace, workspace_bytes, nullptr, stream, form, x_meta, dy_meta,
                      x_f16 ? 1 : 0, dy_f16 ? 1 : 0, xs, dys, n_items);
}
