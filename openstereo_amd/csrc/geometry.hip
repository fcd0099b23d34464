// Geometry-encoding volume of StereoBase / IGEV's GRU loop.
//   reference: models/stereobase/gru_blocks.py:170-229 (CombinedGeoEncodingVolume),
//              models/igev/geometry.py:7-66 (Combined_Geo_Encoding_Volume)
//  * allpairs_corr : corr[b,h,w1,w2] = sum_c f1[b,c,h,w1] * f2[b,c,h,w2]          (einsum 'aijk,aijh->ajkh')
//  * geo_rows      : NDHWC geometry volume -> per-pixel rows [B,H,W,C,D] (the reference's
//                    permute(0,3,4,1,2).reshape(b*h*w, c, 1, d)), lookup-friendly: one row = D contiguous floats
//  * avgpool_rows  : F.avg_pool2d(x, [1,2], stride=[1,2]) along the last axis -> pyramid level i+1
//  * geo_lookup    : for every pixel, level and channel row: 2r+1 taps at x = pos/2^i + (k - r), 1-D linear
//                    interpolation with zero padding (grid_sample, align_corners=True, H == 1), for the C geometry
//                    rows (pos = disp) and the correlation row (pos = coords - disp); writes the reference's
//                    [B, (C+1)*(2r+1)*levels, H, W] tensor in one pass instead of 4 grid_sample calls + cats per iteration.
// All memory-bound; lanes run along w (outputs) or along the contiguous row axis (transposes).
#include "osa_common.h"
#include <cstring>

namespace osa {

// ---- all-pairs correlation: block = (b, h, 16 left pixels); thread = right pixel
__global__ __launch_bounds__(256) void allpairs_corr_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                            float* __restrict__ corr, int C, int H, int W1, int W2) {
    extern __shared__ float f1s[];                 // [C][16]
    const int tiles = (W1 + 15) / 16;
    int bid = blockIdx.x;
    const int t = bid % tiles; bid /= tiles;
    const int h = bid % H; const int b = bid / H;
    const int w10 = t * 16;
    const size_t plane1 = (size_t)H * W1, plane2 = (size_t)H * W2;
    for (int i = threadIdx.x; i < C * 16; i += 256) {
        const int c = i >> 4, j = i & 15;
        f1s[i] = (w10 + j < W1) ? f1[((size_t)b * C + c) * plane1 + (size_t)h * W1 + w10 + j] : 0.f;
    }
    __syncthreads();
    for (int w2 = threadIdx.x; w2 < W2; w2 += 256) {
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        const float* r = f2 + (size_t)b * C * plane2 + (size_t)h * W2 + w2;
        for (int c = 0; c < C; ++c) {
            const float v = r[(size_t)c * plane2];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = fmaf(f1s[c * 16 + j], v, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (w10 + j < W1) corr[(((size_t)b * H + h) * W1 + w10 + j) * W2 + w2] = acc[j];
    }
}

// ---- NDHWC volume [B,D,H,W,Cs] -> rows [B,H,W,C,D]; block = (b, h, 8 pixels)
__global__ __launch_bounds__(256) void geo_rows_kernel(const float* __restrict__ vol, float* __restrict__ rows,
                                                       int D, int H, int W, int C, int Cs) {
    extern __shared__ float tile[];                // [8][D][C+1]
    const int tiles = (W + 7) / 8;
    int bid = blockIdx.x;
    const int t = bid % tiles; bid /= tiles;
    const int h = bid % H; const int b = bid / H;
    const int w0 = t * 8, CP = C + 1;
    const int n_in = 8 * D * C;
    for (int i = threadIdx.x; i < n_in; i += 256) {       // c fastest: coalesced C-vectors
        const int c = i % C; int r = i / C;
        const int px = r % 8; const int d = r / 8;
        float v = 0.f;
        if (w0 + px < W) v = vol[((((size_t)b * D + d) * H + h) * W + w0 + px) * Cs + c];
        tile[(px * D + d) * CP + c] = v;
    }
    __syncthreads();
    const int n_out = 8 * C * D;
    for (int i = threadIdx.x; i < n_out; i += 256) {      // d fastest: one pixel's [C][D] block is contiguous
        const int d = i % D; int r = i / D;
        const int c = r % C; const int px = r / C;
        if (w0 + px < W) rows[((((size_t)b * H + h) * W + w0 + px) * C + c) * D + d] = tile[(px * D + d) * CP + c];
    }
}

__global__ __launch_bounds__(256) void avgpool_rows_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           long long rows, int n) {
    const int no = n / 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * no) return;
    const long long r = i / no; const int j = (int)(i - r * no);
    const float* p = x + r * n + 2 * j;
    y[i] = (p[0] + p[1]) * 0.5f;
}

struct LookupArgs {
    const float* geo[4]; const float* corr[4];      // per level: rows [B,H,W,C,Dl] / [B,H,W,W2l]
    const float* disp; const float* coords; float* out;
    int B, H, W, C, levels, radius;
    int Dl[4], Wl[4];
};

// Sampling position of one tap, with the reference's own float roundings: bilinear_sampler (igev/utils.py:61-79) maps the
// pixel coordinate x to xgrid = 2 * x / (n - 1) - 1 and F.grid_sample(align_corners=True) maps it back with
// ((xgrid + 1) / 2) * (n - 1); at n = 240 the round trip moves x by up to ~2e-5, which shifts the interpolation weights.
// Reproducing the round trip keeps the lookup within a few 1e-6 of the reference at any width (-ffp-contract=off: no fma).
struct Tap { int x0; float w0, w1; };
__device__ __forceinline__ Tap tap_of(float x, int n) {
    const float nm1 = (float)(n - 1);
    const float g = 2.f * x / nm1 - 1.f;
    const float ix = ((g + 1.f) / 2.f) * nm1;
    const float xf = floorf(ix);
    Tap t;
    t.x0 = (int)xf;
    t.w1 = ix - xf;
    t.w0 = (xf + 1.f) - ix;
    return t;
}
__device__ __forceinline__ float sample_row(const float* __restrict__ row, int n, const Tap t) {
    const float a = (t.x0 >= 0 && t.x0 < n) ? row[t.x0] : 0.f;
    const float b = (t.x0 + 1 >= 0 && t.x0 + 1 < n) ? row[t.x0 + 1] : 0.f;
    return a * t.w0 + b * t.w1;
}

__global__ __launch_bounds__(256) void geo_lookup_kernel(const LookupArgs p) {
    const long long HW = (long long)p.H * p.W;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // pixel
    if (i >= (long long)p.B * HW) return;
    const long long b = i / HW, hw = i - b * HW;
    const int taps = 2 * p.radius + 1;
    const int per_level = (p.C + 1) * taps;
    const float d = p.disp[i], cx = p.coords[i];
    float* o = p.out + (size_t)b * per_level * p.levels * HW + hw;
    float scale = 1.f;
    for (int l = 0; l < p.levels; ++l, scale *= 0.5f) {
        const float* g = p.geo[l] + (size_t)i * p.C * p.Dl[l];
        const float* crow = p.corr[l] + (size_t)i * p.Wl[l];
        const float xg = d * scale, xc = cx * scale - d * scale;
        for (int k = 0; k < taps; ++k) {                                    // tap position once, shared by the C volume rows
            const float dx = (float)(k - p.radius);
            const Tap tg = tap_of(dx + xg, p.Dl[l]);                        // geometry.py:36  x0 = dx + disp / 2^i
            for (int c = 0; c < p.C; ++c)
                o[((size_t)l * per_level + c * taps + k) * HW] = sample_row(g + (size_t)c * p.Dl[l], p.Dl[l], tg);
            const Tap tc = tap_of(xc + dx, p.Wl[l]);                        // geometry.py:44  coords / 2^i - disp / 2^i + dx
            o[((size_t)l * per_level + p.C * taps + k) * HW] = sample_row(crow, p.Wl[l], tc);
        }
    }
}

// The same NCHW lookup with one thread per (row, pixel) (r5): the per-pixel kernel above has B*H*W threads -- 14720 at the StereoBase training
// map (80 x 184), under one wave per SIMD of the chip, each walking 324 scattered loads.  Here a thread produces the 2r + 1 taps of ONE row
// (geometry row c of level l, or that level's correlation row) of one pixel; consecutive threads are consecutive pixels of the same row, so
// every store instruction of a wave writes consecutive floats of one output plane.  Same tap_of / sample_row: same values.
__global__ __launch_bounds__(256) void geo_lookup_rows_kernel(const LookupArgs p) {
    const long long HW = (long long)p.H * p.W, npix = (long long)p.B * HW;
    const int rows_per_px = p.levels * (p.C + 1);
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= npix * rows_per_px) return;
    const int j = (int)(t / npix);                                  // row (uniform over a workgroup unless it straddles a row boundary)
    const long long i = t - (long long)j * npix;                    // pixel
    const long long b = i / HW, hw = i - b * HW;
    const int l = j / (p.C + 1), c = j - l * (p.C + 1);
    const int taps = 2 * p.radius + 1;
    const int per_level = (p.C + 1) * taps;
    float scale = 1.f;
    for (int q = 0; q < l; ++q) scale *= 0.5f;
    const float d = p.disp[i], cx = p.coords[i];
    const bool is_corr = (c == p.C);
    const int n = is_corr ? p.Wl[l] : p.Dl[l];
    const float* row = is_corr ? p.corr[l] + (size_t)i * p.Wl[l] : p.geo[l] + ((size_t)i * p.C + c) * p.Dl[l];
    const float xg = d * scale, xc = cx * scale - d * scale;
    float* o = p.out + (size_t)b * per_level * p.levels * HW + hw + ((size_t)l * per_level + (size_t)c * taps) * HW;
    for (int k = 0; k < taps; ++k) {
        const float dx = (float)(k - p.radius);
        o[(size_t)k * HW] = sample_row(row, n, is_corr ? tap_of(xc + dx, n) : tap_of(dx + xg, n));      // operand order of geometry.py:36 / :44
    }
}

// Channels-last form for the engine's GRU loop: out [B,H,W,Cs] (Cs >= channels, the padding zero-filled), i.e. what the update
// block's 1x1 convc1 reads -- the NCHW result of the kernel above had to be transposed every iteration (85 MB at 4 pairs).  One thread =
// one (pixel, row): row j < levels * (C + 1) is geometry row c of level l or that level's correlation row, and produces the 2r + 1 taps of
// its row = 2r + 1 consecutive output channels, so the threads of a pixel write its channel vector contiguously; 18x the threads of the
// per-pixel kernel and 18 loads per thread instead of 324.  Same tap arithmetic (tap_of / sample_row), same values.
template <int TAPS>      // TAPS = 2 * radius + 1 known at compile time (9 in every shipped config), 0 = generic
__global__ __launch_bounds__(256) void geo_lookup_nhwc_kernel(const LookupArgs p, int Cs) {
    const int rows_per_px = p.levels * (p.C + 1);
    const unsigned npix = (unsigned)p.B * p.H * p.W;                // host: B*H*W*rows < 2^31
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= npix * (unsigned)rows_per_px) return;
    const unsigned i = t / (unsigned)rows_per_px;
    const int j = (int)(t - i * (unsigned)rows_per_px);
    const int l = j / (p.C + 1), c = j - l * (p.C + 1);
    const int taps = TAPS ? TAPS : 2 * p.radius + 1;
    const float scale = 1.f / (float)(1 << l);                     // exact: the per-pixel kernel multiplies 0.5f l times
    const float d = p.disp[i], cx = p.coords[i];
    float* o = p.out + (size_t)i * Cs + (size_t)j * taps;
    const bool is_corr = (c == p.C);
    const int n = is_corr ? p.Wl[l] : p.Dl[l];
    const float* row = is_corr ? p.corr[l] + (size_t)i * p.Wl[l] : p.geo[l] + ((size_t)i * p.C + c) * p.Dl[l];
    const float x = is_corr ? (cx * scale - d * scale) : d * scale;
    if constexpr (TAPS > 0) {
        // all loads first, unconditionally (index clamped into the row, the value dropped by a select): 2 * TAPS independent requests
        // in flight per thread instead of a branch and a wait per tap
        Tap tp[TAPS]; float a[TAPS], b[TAPS];
#pragma unroll
        for (int k = 0; k < TAPS; ++k) {
            const float dx = (float)(k - TAPS / 2);
            tp[k] = tap_of(is_corr ? x + dx : dx + x, n);           // operand order of geometry.py:36 / :44
            const int i0 = tp[k].x0 < 0 ? 0 : (tp[k].x0 > n - 1 ? n - 1 : tp[k].x0);
            const int i1 = tp[k].x0 + 1 < 0 ? 0 : (tp[k].x0 + 1 > n - 1 ? n - 1 : tp[k].x0 + 1);
            a[k] = row[i0]; b[k] = row[i1];
        }
#pragma unroll
        for (int k = 0; k < TAPS; ++k) {
            const float av = (tp[k].x0 >= 0 && tp[k].x0 < n) ? a[k] : 0.f;
            const float bv = (tp[k].x0 + 1 >= 0 && tp[k].x0 + 1 < n) ? b[k] : 0.f;
            o[k] = av * tp[k].w0 + bv * tp[k].w1;                   // == sample_row
        }
    } else {
        for (int k = 0; k < taps; ++k) {
            const float dx = (float)(k - p.radius);
            o[k] = sample_row(row, n, tap_of(is_corr ? x + dx : dx + x, n));
        }
    }
    if (j == rows_per_px - 1)
        for (int k = rows_per_px * taps; k < Cs; ++k) p.out[(size_t)i * Cs + k] = 0.f;
}

// Backward of the lookup w.r.t. the pyramid levels (the disparity is detached in the reference, igev_stereo.py:190): every pixel owns
// its rows of every level, so a thread adds its taps into its own (zero-filled) rows -- no atomics, deterministic.
struct LookupBwdArgs {
    float* dgeo[4]; float* dcorr[4];
    const float* disp; const float* coords; const float* dout;
    int B, H, W, C, levels, radius;
    int Dl[4], Wl[4];
};
__device__ __forceinline__ void scatter_row(float* __restrict__ row, int n, const Tap t, float g) {
    if (t.x0 >= 0 && t.x0 < n) row[t.x0] += g * t.w0;
    if (t.x0 + 1 >= 0 && t.x0 + 1 < n) row[t.x0 + 1] += g * t.w1;
}
__global__ __launch_bounds__(256) void geo_lookup_bwd_kernel(const LookupBwdArgs p) {
    const long long HW = (long long)p.H * p.W;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)p.B * HW) return;
    const long long b = i / HW, hw = i - b * HW;
    const int taps = 2 * p.radius + 1;
    const int per_level = (p.C + 1) * taps;
    const float d = p.disp[i], cx = p.coords[i];
    const float* o = p.dout + (size_t)b * per_level * p.levels * HW + hw;
    float scale = 1.f;
    for (int l = 0; l < p.levels; ++l, scale *= 0.5f) {
        float* g = p.dgeo[l] + (size_t)i * p.C * p.Dl[l];
        float* crow = p.dcorr[l] + (size_t)i * p.Wl[l];
        const float xg = d * scale, xc = cx * scale - d * scale;
        for (int k = 0; k < taps; ++k) {
            const float dx = (float)(k - p.radius);
            const Tap tg = tap_of(dx + xg, p.Dl[l]);
            for (int c = 0; c < p.C; ++c)
                scatter_row(g + (size_t)c * p.Dl[l], p.Dl[l], tg, o[((size_t)l * per_level + c * taps + k) * HW]);
            const Tap tc = tap_of(xc + dx, p.Wl[l]);
            scatter_row(crow, p.Wl[l], tc, o[((size_t)l * per_level + p.C * taps + k) * HW]);
        }
    }
}

// r5: the same gradient in GATHER form.  The scatter kernel above gives every pixel ONE thread that read-modify-writes 324 scattered floats
// of its (memset) rows: 0.54 ms per call at the 80 x 184 training map, 22 calls per StereoBase step (12 ms, the largest single item of the AMP
// step's kernel census).  Here a thread owns one OUTPUT element (pixel, level, row, position j) and sums the taps that land on
// it: tap k contributes dout_k * w0_k if x0_k == j and dout_k * w1_k if x0_k + 1 == j.  The taps are evaluated with the same tap_of() and
// visited in the same order k = 0 .. 2r as the scatter kernel adds them, so the result is bit-identical; every element is written (zeros
// included): no memset, no read-modify-write, consecutive threads write consecutive floats.
struct LookupBwdGatherArgs {
    float* dgeo[4]; float* dcorr[4];
    const float* disp; const float* coords; const float* dout;
    int B, H, W, C, levels, radius;
    int Dl[4], Wl[4];
    long long seg_end[8];            // running end of segment 2 l (geo rows of level l) / 2 l + 1 (corr rows) in the flat element index
};
__global__ __launch_bounds__(256) void geo_lookup_bwd_gather_kernel(const LookupBwdGatherArgs p) {
    const long long HW = (long long)p.H * p.W;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int nseg = 2 * p.levels;
    if (t >= p.seg_end[nseg - 1]) return;
    int sgm = 0;
    while (t >= p.seg_end[sgm]) ++sgm;
    const long long e = t - (sgm ? p.seg_end[sgm - 1] : 0);
    const int l = sgm >> 1;
    const bool is_corr = sgm & 1;
    const int n = is_corr ? p.Wl[l] : p.Dl[l];
    const int rows = is_corr ? 1 : p.C;
    const int j = (int)(e % n);
    const long long r = e / n;
    const int c = is_corr ? p.C : (int)(r % rows);
    const long long i = is_corr ? r : r / rows;                        // pixel
    const long long b = i / HW, hw = i - b * HW;
    const int taps = 2 * p.radius + 1;
    const int per_level = (p.C + 1) * taps;
    float scale = 1.f;
    for (int q = 0; q < l; ++q) scale *= 0.5f;
    const float d = p.disp[i], cx = p.coords[i];
    const float xg = d * scale, xc = cx * scale - d * scale;
    const float* o = p.dout + (size_t)b * per_level * p.levels * HW + hw + ((size_t)l * per_level + (size_t)c * taps) * HW;
    float v = 0.f;
    for (int k = 0; k < taps; ++k) {
        const float dx = (float)(k - p.radius);
        const Tap tp = is_corr ? tap_of(xc + dx, n) : tap_of(dx + xg, n);
        if (tp.x0 == j) v += o[(size_t)k * HW] * tp.w0;
        if (tp.x0 + 1 == j) v += o[(size_t)k * HW] * tp.w1;
    }
    float* dst = is_corr ? p.dcorr[l] : p.dgeo[l];
    dst[e] = v;
}

// Gather form, one WAVE per (pixel, level) (r5, second version): the 2r + 1 tap positions of the level's geometry rows and of its correlation row
// are evaluated once per wave (they depend on the pixel only), a lane owns output positions j = lane, lane + 64, ... of every row, the
// upstream gradients of a row's taps are wave-uniform loads, and the lanes of a wave store consecutive floats.  The thread-per-element form
// above evaluates the same 9 taps for each of the C * D + W elements of the pixel.  Same taps, same order of additions: bit-identical.
template <int MAXT>
__global__ __launch_bounds__(256) void geo_lookup_bwd_rows_kernel(const LookupBwdGatherArgs p) {
    const long long HW = (long long)p.H * p.W;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long i = (long long)blockIdx.x * 4 + wv;                // pixel (wave-uniform)
    const int l = blockIdx.y;
    if (i >= (long long)p.B * HW) return;
    const int lane = threadIdx.x & 63;
    const long long b = i / HW, hw = i - b * HW;
    const int taps = 2 * p.radius + 1;                                  // <= MAXT (host)
    const int per_level = (p.C + 1) * taps;
    float scale = 1.f;
    for (int q = 0; q < l; ++q) scale *= 0.5f;
    const float d = p.disp[i], cx = p.coords[i];
    const float xg = d * scale, xc = cx * scale - d * scale;
    const int Dl = p.Dl[l], Wl = p.Wl[l];
    int gx0[MAXT], cx0[MAXT]; float gw0[MAXT], gw1[MAXT], cw0[MAXT], cw1[MAXT];
#pragma unroll
    for (int k = 0; k < MAXT; ++k) {
        const float dx = (float)(k - p.radius);
        const Tap tg = tap_of(dx + xg, Dl), tc = tap_of(xc + dx, Wl);
        const bool on = k < taps;
        gx0[k] = on ? tg.x0 : -4; gw0[k] = tg.w0; gw1[k] = tg.w1;       // (-4: matches no position j >= 0, nor j - 1)
        cx0[k] = on ? tc.x0 : -4; cw0[k] = tc.w0; cw1[k] = tc.w1;
    }
    // every upstream gradient of this (pixel, level) -- (C + 1) rows x taps, 81 in the shipped configs, each in its own output plane -- with ONE
    // round of loads (each lane holds up to four of them; host: (C + 1) * taps <= 256); the rows then take theirs by readlane.  (The first
    // version loaded a row's taps as wave-uniform values inside the row loop: C + 1 dependent memory round trips per wave, 237 us inside the
    // training step although the kernel moves 60 MB.)
    const float* o = p.dout + (size_t)b * per_level * p.levels * HW + hw + (size_t)l * per_level * HW;
    float gq[4];                                                         // lane t holds elements t, t + 64, t + 128, t + 192 (host: per_level <= 256)
#pragma unroll
    for (int v = 0; v < 4; ++v) gq[v] = lane + 64 * v < per_level ? o[(size_t)(lane + 64 * v) * HW] : 0.f;
    auto take = [&](int idx) -> float {                                  // idx wave-uniform
        const int s_ = __builtin_amdgcn_readfirstlane(idx);
        const int q = s_ >> 6;
        const float src = q == 0 ? gq[0] : (q == 1 ? gq[1] : (q == 2 ? gq[2] : gq[3]));
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(src), s_ & 63));
    };
    float* g = p.dgeo[l] + (size_t)i * p.C * Dl;
    for (int c = 0; c < p.C; ++c) {
        float ov[MAXT];
#pragma unroll
        for (int k = 0; k < MAXT; ++k) ov[k] = k < taps ? take(c * taps + k) : 0.f;
        for (int j = lane; j < Dl; j += 64) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < MAXT; ++k) {
                if (gx0[k] == j) v += ov[k] * gw0[k];
                if (gx0[k] + 1 == j) v += ov[k] * gw1[k];
            }
            g[(size_t)c * Dl + j] = v;
        }
    }
    float ov[MAXT];
#pragma unroll
    for (int k = 0; k < MAXT; ++k) ov[k] = k < taps ? take(p.C * taps + k) : 0.f;
    float* crow = p.dcorr[l] + (size_t)i * Wl;
    for (int j = lane; j < Wl; j += 64) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < MAXT; ++k) {
            if (cx0[k] == j) v += ov[k] * cw0[k];
            if (cx0[k] + 1 == j) v += ov[k] * cw1[k];
        }
        crow[j] = v;
    }
}


// Accumulating form (r6): the reference's loop looks the SAME pyramid up once per GRU iteration (igev_stereo.py:181-203: 22 iterations in
// training), so autograd used to receive 22 dense gradients per level (50 MB each at the 320x736 crop, 98 % zeros) and add them up.  Here
// the level gradients are accumulators the caller zero-fills once per step; a lookup's backward touches only the (C + 1) x (taps + 1)
// entries per (pixel, level) its taps reach: one thread per entry, a read-modify-write without atomics (every pixel owns its rows, and
// the entries of one row are distinct positions) -- 2.6 M entries instead of 12.5 M written and 12.5 M added per iteration.
__global__ __launch_bounds__(256) void geo_lookup_bwd_acc_kernel(const LookupBwdGatherArgs p) {
    const long long HW = (long long)p.H * p.W;
    const int taps = 2 * p.radius + 1;
    const int per_px = (p.C + 1) * (taps + 1);
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int l = blockIdx.y;
    if (t >= (long long)p.B * HW * per_px) return;
    const long long i = t / per_px;                                      // pixel
    const int e = (int)(t - i * per_px);
    const int c = e / (taps + 1), jj = e - c * (taps + 1);               // row (c == C: the correlation row), entry of the tap window
    const long long b = i / HW, hw = i - b * HW;
    float scale = 1.f;
    for (int q = 0; q < l; ++q) scale *= 0.5f;
    const float d = p.disp[i], cx = p.coords[i];
    const bool geo = c < p.C;
    const float x = geo ? d * scale : cx * scale - d * scale;
    const int n = geo ? p.Dl[l] : p.Wl[l];
    const int per_level = (p.C + 1) * taps;
    const float* o = p.dout + (size_t)b * per_level * p.levels * HW + hw + ((size_t)l * per_level + (size_t)c * taps) * HW;
    // entry jj is position x0(tap jj) (= x0(tap jj - 1) + 1): tap jj - 1 reaches it with its w1, tap jj with its w0 -- the order in which
    // the dense kernels add them
    float v = 0.f;
    int j;
    if (jj < taps) {
        const Tap tk = tap_of((float)(jj - p.radius) + x, n);
        j = tk.x0;
        if (jj >= 1) {
            const Tap tp = tap_of((float)(jj - 1 - p.radius) + x, n);
            if (tp.x0 + 1 == j) v += o[(size_t)(jj - 1) * HW] * tp.w1;
        }
        v += o[(size_t)jj * HW] * tk.w0;
    } else {
        const Tap tp = tap_of((float)(jj - 1 - p.radius) + x, n);
        j = tp.x0 + 1;
        v += o[(size_t)(jj - 1) * HW] * tp.w1;
    }
    if (j < 0 || j >= n) return;
    float* dst = geo ? p.dgeo[l] + ((size_t)i * p.C + c) * n + j : p.dcorr[l] + (size_t)i * n + j;
    *dst += v;
}

}  // namespace osa

using namespace osa;

extern "C" int osa_geo_lookup_bwd_f32(float* const* dgeo_levels, float* const* dcorr_levels,
                                      const int* geo_len, const int* corr_len, int levels,
                                      const float* disp, const float* coords_x, const float* dout,
                                      int B, int H, int W, int C, int radius, void* stream) {
    OSA_REQUIRE(dgeo_levels && dcorr_levels && geo_len && corr_len && disp && coords_x && dout, "geo_lookup_bwd: NULL pointer");
    OSA_REQUIRE(levels >= 1 && levels <= 4, "geo_lookup_bwd: %d levels unsupported (1..4)", levels);
    const long long total = (long long)B * H * W;
    if (!exp_int("OSA_GEO_BWD_SCATTER", 0)) {                              // gather form (r5): one thread per output element, no memset
        LookupBwdGatherArgs g;
        long long end = 0;
        for (int l = 0; l < levels; ++l) {
            OSA_REQUIRE(dgeo_levels[l] && dcorr_levels[l] && geo_len[l] > 0 && corr_len[l] > 0, "geo_lookup_bwd: level %d missing", l);
            g.dgeo[l] = dgeo_levels[l]; g.dcorr[l] = dcorr_levels[l]; g.Dl[l] = geo_len[l]; g.Wl[l] = corr_len[l];
            end += total * C * geo_len[l]; g.seg_end[2 * l] = end;
            end += total * corr_len[l]; g.seg_end[2 * l + 1] = end;
        }
        for (int q = 2 * levels; q < 8; ++q) g.seg_end[q] = end;
        g.disp = disp; g.coords = coords_x; g.dout = dout;
        g.B = B; g.H = H; g.W = W; g.C = C; g.levels = levels; g.radius = radius;
        OSA_REQUIRE((end + 255) / 256 < (1ll << 31), "geo_lookup_bwd: grid too large");
        if (2 * radius + 1 <= 9 && (C + 1) * (2 * radius + 1) <= 256 && exp_int("OSA_GEO_BWD_FORM", 2) == 2) {     // one wave per (pixel, level): every shipped config has radius 4
            hipLaunchKernelGGL(geo_lookup_bwd_rows_kernel<9>, dim3((unsigned)((total + 3) / 4), levels), dim3(256), 0, (hipStream_t)stream, g);
            OSA_LAUNCH_CHECK("geo_lookup_bwd (rows)");
            return 0;
        }
        hipLaunchKernelGGL(geo_lookup_bwd_gather_kernel, dim3((unsigned)((end + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g);
        OSA_LAUNCH_CHECK("geo_lookup_bwd (gather)");
        return 0;
    }
    LookupBwdArgs a;
    for (int l = 0; l < levels; ++l) {
        OSA_REQUIRE(dgeo_levels[l] && dcorr_levels[l] && geo_len[l] > 0 && corr_len[l] > 0, "geo_lookup_bwd: level %d missing", l);
        a.dgeo[l] = dgeo_levels[l]; a.dcorr[l] = dcorr_levels[l]; a.Dl[l] = geo_len[l]; a.Wl[l] = corr_len[l];
        hipError_t e = hipMemsetAsync(a.dgeo[l], 0, (size_t)total * C * geo_len[l] * sizeof(float), (hipStream_t)stream);
        if (e == hipSuccess) e = hipMemsetAsync(a.dcorr[l], 0, (size_t)total * corr_len[l] * sizeof(float), (hipStream_t)stream);
        OSA_REQUIRE(e == hipSuccess, "geo_lookup_bwd: memset failed: %s", hipGetErrorString(e));
    }
    a.disp = disp; a.coords = coords_x; a.dout = dout;
    a.B = B; a.H = H; a.W = W; a.C = C; a.levels = levels; a.radius = radius;
    hipLaunchKernelGGL(geo_lookup_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, a);
    OSA_LAUNCH_CHECK("geo_lookup_bwd");
    return 0;
}

extern "C" int osa_geo_lookup_bwd_acc_f32(float* const* dgeo_levels, float* const* dcorr_levels,
                                          const int* geo_len, const int* corr_len, int levels,
                                          const float* disp, const float* coords_x, const float* dout,
                                          int B, int H, int W, int C, int radius, void* stream) {
    OSA_REQUIRE(dgeo_levels && dcorr_levels && geo_len && corr_len && disp && coords_x && dout, "geo_lookup_bwd_acc: NULL pointer");
    OSA_REQUIRE(levels >= 1 && levels <= 4 && radius >= 0 && C >= 1, "geo_lookup_bwd_acc: %d levels / radius %d unsupported", levels, radius);
    LookupBwdGatherArgs g;
    memset(&g, 0, sizeof(g));
    for (int l = 0; l < levels; ++l) {
        OSA_REQUIRE(dgeo_levels[l] && dcorr_levels[l] && geo_len[l] > 0 && corr_len[l] > 0, "geo_lookup_bwd_acc: level %d missing", l);
        g.dgeo[l] = dgeo_levels[l]; g.dcorr[l] = dcorr_levels[l]; g.Dl[l] = geo_len[l]; g.Wl[l] = corr_len[l];
    }
    g.disp = disp; g.coords = coords_x; g.dout = dout;
    g.B = B; g.H = H; g.W = W; g.C = C; g.levels = levels; g.radius = radius;
    const long long n = (long long)B * H * W * (C + 1) * (2 * radius + 2);
    OSA_REQUIRE((n + 255) / 256 < (1ll << 31), "geo_lookup_bwd_acc: grid too large");
    hipLaunchKernelGGL(geo_lookup_bwd_acc_kernel, dim3((unsigned)((n + 255) / 256), levels), dim3(256), 0, (hipStream_t)stream, g);
    OSA_LAUNCH_CHECK("geo_lookup_bwd_acc");
    return 0;
}

extern "C" int osa_allpairs_corr_f32(const float* fmap1, const float* fmap2, float* corr,
                                     int B, int C, int H, int W1, int W2, void* stream) {
    OSA_REQUIRE(fmap1 && fmap2 && corr, "allpairs_corr: NULL pointer");
    OSA_REQUIRE(B > 0 && C > 0 && H > 0 && W1 > 0 && W2 > 0, "allpairs_corr: bad dims");
    const size_t lds = (size_t)C * 16 * sizeof(float);
    OSA_REQUIRE(lds <= 64 * 1024, "allpairs_corr: C=%d too large", C);
    const long long nblk = (long long)B * H * cdiv(W1, 16);
    hipLaunchKernelGGL(allpairs_corr_kernel, dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, fmap1, fmap2, corr, C, H, W1, W2);
    OSA_LAUNCH_CHECK("allpairs_corr");
    return 0;
}

extern "C" int osa_geo_rows_f32(const float* vol_ndhwc, float* rows, int B, int D, int H, int W, int C, int Cs, void* stream) {
    OSA_REQUIRE(vol_ndhwc && rows, "geo_rows: NULL pointer");
    OSA_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && Cs >= C, "geo_rows: bad dims");
    const size_t lds = (size_t)8 * D * (C + 1) * sizeof(float);
    OSA_REQUIRE(lds <= 160 * 1024, "geo_rows: D*C too large for LDS (%zu B)", lds);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)geo_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const long long nblk = (long long)B * H * cdiv(W, 8);
    hipLaunchKernelGGL(geo_rows_kernel, dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, vol_ndhwc, rows, D, H, W, C, Cs);
    OSA_LAUNCH_CHECK("geo_rows");
    return 0;
}

extern "C" int osa_avgpool_rows_f32(const float* x, float* y, long long rows, int n, void* stream) {
    OSA_REQUIRE(x && y && rows > 0 && n >= 2, "avgpool_rows: bad arguments");
    const long long total = rows * (n / 2);
    hipLaunchKernelGGL(avgpool_rows_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, rows, n);
    OSA_LAUNCH_CHECK("avgpool_rows");
    return 0;
}

extern "C" int osa_geo_lookup_nhwc_f32(const float* const* geo_levels, const float* const* corr_levels,
                                       const int* geo_len, const int* corr_len, int levels,
                                       const float* disp, const float* coords_x, float* out, int out_cs,
                                       int B, int H, int W, int C, int radius, void* stream) {
    OSA_REQUIRE(geo_levels && corr_levels && geo_len && corr_len && disp && coords_x && out, "geo_lookup_nhwc: NULL pointer");
    OSA_REQUIRE(levels >= 1 && levels <= 4, "geo_lookup_nhwc: %d levels unsupported (1..4)", levels);
    OSA_REQUIRE(out_cs >= levels * (C + 1) * (2 * radius + 1), "geo_lookup_nhwc: channel stride %d < %d channels", out_cs, levels * (C + 1) * (2 * radius + 1));
    LookupArgs a;
    for (int l = 0; l < levels; ++l) {
        OSA_REQUIRE(geo_levels[l] && corr_levels[l] && geo_len[l] > 0 && corr_len[l] > 0, "geo_lookup_nhwc: level %d missing", l);
        a.geo[l] = geo_levels[l]; a.corr[l] = corr_levels[l]; a.Dl[l] = geo_len[l]; a.Wl[l] = corr_len[l];
    }
    a.disp = disp; a.coords = coords_x; a.out = out;
    a.B = B; a.H = H; a.W = W; a.C = C; a.levels = levels; a.radius = radius;
    const long long total = (long long)B * H * W * levels * (C + 1);
    OSA_REQUIRE(total < (1ll << 31), "geo_lookup_nhwc: more than 2^31 (pixel, row) pairs");
    if (radius == 4) hipLaunchKernelGGL(geo_lookup_nhwc_kernel<9>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, out_cs);
    else hipLaunchKernelGGL(geo_lookup_nhwc_kernel<0>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, out_cs);
    OSA_LAUNCH_CHECK("geo_lookup_nhwc");
    return 0;
}

extern "C" int osa_geo_lookup_f32(const float* const* geo_levels, const float* const* corr_levels,
                                  const int* geo_len, const int* corr_len, int levels,
                                  const float* disp, const float* coords_x, float* out,
                                  int B, int H, int W, int C, int radius, void* stream) {
    OSA_REQUIRE(geo_levels && corr_levels && geo_len && corr_len && disp && coords_x && out, "geo_lookup: NULL pointer");
    OSA_REQUIRE(levels >= 1 && levels <= 4, "geo_lookup: %d levels unsupported (1..4)", levels);
    LookupArgs a;
    for (int l = 0; l < levels; ++l) {
        OSA_REQUIRE(geo_levels[l] && corr_levels[l] && geo_len[l] > 0 && corr_len[l] > 0, "geo_lookup: level %d missing", l);
        a.geo[l] = geo_levels[l]; a.corr[l] = corr_levels[l]; a.Dl[l] = geo_len[l]; a.Wl[l] = corr_len[l];
    }
    a.disp = disp; a.coords = coords_x; a.out = out;
    a.B = B; a.H = H; a.W = W; a.C = C; a.levels = levels; a.radius = radius;
    const long long total = (long long)B * H * W;
    if (exp_int("OSA_GEO_FWD_PIXEL", 0)) {                           // the r2 form: one thread per pixel (experiments build only)
        hipLaunchKernelGGL(geo_lookup_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, a);
    } else {
        const long long nthr = total * levels * (C + 1);
        OSA_REQUIRE(cdiv(nthr, 256) > 0, "geo_lookup: grid too large");
        hipLaunchKernelGGL(geo_lookup_rows_kernel, dim3(cdiv(nthr, 256)), dim3(256), 0, (hipStream_t)stream, a);
    }
    OSA_LAUNCH_CHECK("geo_lookup");
    return 0;
}
