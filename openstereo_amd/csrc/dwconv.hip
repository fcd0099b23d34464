// Depthwise 2-D convolution, NHWC (LightStereo's MobileV2Residual depthwise
// 3x3 convs and the strip convolutions 1x7 / 7x1 / 1x11 / 11x1 / 1x21 / 21x1 of AttentionModule,
// stereo/modeling/models/lightstereo/aggregation.py:63-134).
//
// One multiply-add per output element and tap: HBM/L2 bound, no matrix work.  A thread owns 4
// consecutive channels of one output pixel (float4 everywhere), neighbouring threads neighbouring
// channel quads, so every load and store of a wave is a run of full 256-byte channel rows; the
// kh*kw taps of a pixel re-read rows that other pixels of the workgroup just touched (L1/L2).
// Weights are repacked once to [tap][C] so a tap's 4 weights are one float4.
// Epilogue: y = act(acc * scale[c] + shift[c]) + add   (folded eval BatchNorm or bias; optional addend).
#include "osa_common.h"

namespace osa {

struct DwArgs {
    const float* x; const float* w; const float* scale; const float* shift; const float* add; float* y;
    float* meta;         // range block of y (max |y| folded into meta[0]) or NULL
    int B, Hi, Wi, Ho, Wo, C, xCs, yCs, aCs;
    int kh, kw, stride, pad_h, pad_w, dil_h, dil_w, act;
    long long total;     // B*Ho*Wo*(C/4)
    int xf16, yf16;      // r6 (run kernel, 3x3 only): x / y hold fp16 elements (strides in elements) -- the f16 mode's chain tensors
};

__global__ __launch_bounds__(256) void dwconv2d_nhwc_kernel(const DwArgs p) {
    __shared__ float red[4];
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    float am = 0.f;
    const unsigned am_seen = p.meta ? amax_peek(p.meta) : 0u;
    if (idx < p.total) {
    const int nq = p.C >> 2;
    const int q = (int)(idx % nq);
    long long pix = idx / nq;
    const int ox = (int)(pix % p.Wo); pix /= p.Wo;
    const int oy = (int)(pix % p.Ho);
    const int b = (int)(pix / p.Ho);
    const int c = q * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* xb = p.x + (size_t)b * p.Hi * p.Wi * p.xCs + c;
    const int iy0 = oy * p.stride - p.pad_h, ix0 = ox * p.stride - p.pad_w;
    for (int ky = 0; ky < p.kh; ++ky) {
        const int iy = iy0 + ky * p.dil_h;
        if ((unsigned)iy >= (unsigned)p.Hi) continue;
        for (int kx = 0; kx < p.kw; ++kx) {
            const int ix = ix0 + kx * p.dil_w;
            if ((unsigned)ix >= (unsigned)p.Wi) continue;
            const float4 v = *reinterpret_cast<const float4*>(xb + ((size_t)iy * p.Wi + ix) * p.xCs);
            const float4 w = *reinterpret_cast<const float4*>(p.w + (size_t)(ky * p.kw + kx) * p.C + c);
            acc.x = fmaf(v.x, w.x, acc.x); acc.y = fmaf(v.y, w.y, acc.y);
            acc.z = fmaf(v.z, w.z, acc.z); acc.w = fmaf(v.w, w.w, acc.w);
        }
    }
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + c);
    if (p.shift) sh = *reinterpret_cast<const float4*>(p.shift + c);
    float o[4] = {fmaf(acc.x, sc.x, sh.x), fmaf(acc.y, sc.y, sh.y), fmaf(acc.z, sc.z, sh.z), fmaf(acc.w, sc.w, sh.w)};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (p.act == OSA_ACT_RELU) o[e] = fmaxf(o[e], 0.f);
        else if (p.act == OSA_ACT_RELU6) o[e] = fminf(fmaxf(o[e], 0.f), 6.f);
    }
    const size_t opix = ((size_t)b * p.Ho + oy) * p.Wo + ox;
    if (p.add) {
        const float4 a = *reinterpret_cast<const float4*>(p.add + opix * p.aCs + c);
        o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w;
    }
    *reinterpret_cast<float4*>(p.y + opix * p.yCs + c) = make_float4(o[0], o[1], o[2], o[3]);
    am = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
    }
    if (p.meta) publish_amax(p.meta, am, am_seen, red);
}

// r4: pixel-run form.  The kernel above issues 2 x kh x kw 16-byte loads per output quad (input + weight per tap; the column re-reads are
// L1 hits, but every one of them occupies the vector-memory path) and ran LightStereo's depthwise launches at ~1.3 TB/s of algorithmic
// traffic -- 57 % of the whole cost stage (profiles/round4/amp_workloads_kernel_tables.txt).  Here a thread owns 4 channels of NPX
// consecutive output pixels of one row: per kernel row it loads the (NPX - 1) S + KW input quads of its window ONCE into registers and
// each tap's weight quad once; 3x3 stride 1: 27 loads per 4 outputs instead of 72, 1x21: 45 instead of 168.  Same fmaf order per
// output as the kernel above (ky outer, kx inner): bit-identical results.  Unit dilation; compile-time (KW, S) for the shapes the
// models use -- 3x3 (stride 1 / 2), the strip convolutions 1x7 / 7x1 / 1x11 / 11x1 / 1x21 / 21x1 -- anything else keeps the tap-loop kernel.
__device__ __forceinline__ float4 dw_ld4(const float* base, size_t off, int f16) {
    if (f16) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const _Float16*>(base) + off);
        const h2 a = __builtin_bit_cast(h2, u.x), b = __builtin_bit_cast(h2, u.y);
        return make_float4((float)a[0], (float)a[1], (float)b[0], (float)b[1]);
    }
    return *reinterpret_cast<const float4*>(base + off);
}
template <int KW, int S, int NPX, int KC = KW, int XF16 = 0, int YF16 = 0>
__global__ __launch_bounds__(256) void dwconv2d_run_kernel(const DwArgs p, const int runs) {
    // KC: taps of a row handled per window (KC < KW: the long horizontal strips 1x11 / 1x21 walk their row in chunks of KC taps, so the
    // register window stays (NPX - 1) S + KC quads; the chunk loop is kept rolled)
    static_assert(KW % KC == 0, "chunks must tile the kernel row");
    constexpr int WIN = (NPX - 1) * S + KC;
    __shared__ float red[4];
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    float am = 0.f;
    const unsigned am_seen = p.meta ? amax_peek(p.meta) : 0u;
    if (idx < p.total) {
        const unsigned nq = (unsigned)(p.C >> 2);
        unsigned r = (unsigned)idx;                        // host: total < 2^31
        const unsigned q = r % nq; r /= nq;
        const unsigned run = r % (unsigned)runs; r /= (unsigned)runs;
        const int oy = (int)(r % (unsigned)p.Ho);
        const int b = (int)(r / (unsigned)p.Ho);
        const int c = (int)q * 4, ox0 = (int)run * NPX;
        float4 acc[NPX];
#pragma unroll
        for (int j = 0; j < NPX; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        const size_t xb = (size_t)b * p.Hi * p.Wi * p.xCs + c;                // element offsets (fp32 or fp16 elements)
        const int iy0 = oy * S - p.pad_h, ix0 = ox0 * S - p.pad_w;
        for (int ky = 0; ky < p.kh; ++ky) {
            const int iy = iy0 + ky;
            if ((unsigned)iy >= (unsigned)p.Hi) continue;
            const size_t row = xb + (size_t)iy * p.Wi * p.xCs;
#pragma unroll 1
            for (int kc = 0; kc < KW; kc += KC) {
                float4 win[WIN];
#pragma unroll
                for (int i = 0; i < WIN; ++i) {
                    const int ix = ix0 + kc + i;
                    win[i] = ((unsigned)ix < (unsigned)p.Wi) ? dw_ld4(p.x, row + (size_t)ix * p.xCs, XF16) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                const float* wr = p.w + (size_t)(ky * KW + kc) * p.C + c;
#pragma unroll
                for (int kx = 0; kx < KC; ++kx) {
                    const float4 w = *reinterpret_cast<const float4*>(wr + (size_t)kx * p.C);
#pragma unroll
                    for (int j = 0; j < NPX; ++j) {
                        const float4 v = win[j * S + kx];
                        // (a column outside the image contributes nothing in the tap-loop kernel; here it contributes v = 0: fmaf(0, w, acc) == acc
                        // exactly, except for acc = -0 -> +0, which no later operation distinguishes)
                        acc[j].x = fmaf(v.x, w.x, acc[j].x); acc[j].y = fmaf(v.y, w.y, acc[j].y);
                        acc[j].z = fmaf(v.z, w.z, acc[j].z); acc[j].w = fmaf(v.w, w.w, acc[j].w);
                    }
                }
            }
        }
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + c);
        if (p.shift) sh = *reinterpret_cast<const float4*>(p.shift + c);
        const size_t opix0 = ((size_t)b * p.Ho + oy) * p.Wo + ox0;
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            if (ox0 + j >= p.Wo) break;
            float o[4] = {fmaf(acc[j].x, sc.x, sh.x), fmaf(acc[j].y, sc.y, sh.y), fmaf(acc[j].z, sc.z, sh.z), fmaf(acc[j].w, sc.w, sh.w)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (p.act == OSA_ACT_RELU) o[e] = fmaxf(o[e], 0.f);
                else if (p.act == OSA_ACT_RELU6) o[e] = fminf(fmaxf(o[e], 0.f), 6.f);
            }
            if (p.add) {
                const float4 a = *reinterpret_cast<const float4*>(p.add + (opix0 + j) * p.aCs + c);
                o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w;
            }
            if (YF16) {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const h2 a = {(_Float16)o[0], (_Float16)o[1]}, bq = {(_Float16)o[2], (_Float16)o[3]};
                *reinterpret_cast<uint2*>(reinterpret_cast<_Float16*>(p.y) + (opix0 + j) * p.yCs + c) = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, bq));
            } else *reinterpret_cast<float4*>(p.y + (opix0 + j) * p.yCs + c) = make_float4(o[0], o[1], o[2], o[3]);
            am = fmaxf(am, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
        }
    }
    if (p.meta) publish_amax(p.meta, am, am_seen, red);
}

// 3 x 3 depthwise layer with fp16 input AND output (r6): a thread owns EIGHT channels (one 16-byte load / store per pixel) of NPX consecutive
// output pixels -- the 4-channel form above moves fp16 tensors in 8-byte accesses and reached 2.1 TB/s.  Same fmaf order per output.
template <int S, int NPX>
__global__ __launch_bounds__(256) void dwconv3x3_h8_kernel(const DwArgs p, const int runs) {
    constexpr int WIN = (NPX - 1) * S + 3;
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    __shared__ float red[4];
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    float am = 0.f;
    const unsigned am_seen = p.meta ? amax_peek(p.meta) : 0u;
    if (idx < p.total) {
        const unsigned no = (unsigned)(p.C >> 3);
        unsigned r = (unsigned)idx;
        const unsigned q = r % no; r /= no;
        const unsigned run = r % (unsigned)runs; r /= (unsigned)runs;
        const int oy = (int)(r % (unsigned)p.Ho);
        const int b = (int)(r / (unsigned)p.Ho);
        const int c = (int)q * 8, ox0 = (int)run * NPX;
        float acc[NPX][8];
#pragma unroll
        for (int j = 0; j < NPX; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
        const _Float16* xh = reinterpret_cast<const _Float16*>(p.x) + (size_t)b * p.Hi * p.Wi * p.xCs + c;
        const int iy0 = oy * S - p.pad_h, ix0 = ox0 * S - p.pad_w;
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = iy0 + ky;
            if ((unsigned)iy >= (unsigned)p.Hi) continue;
            const _Float16* row = xh + (size_t)iy * p.Wi * p.xCs;
            float win[WIN][8];
#pragma unroll
            for (int i = 0; i < WIN; ++i) {
                const int ix = ix0 + i;
                uint4 u = make_uint4(0u, 0u, 0u, 0u);
                if ((unsigned)ix < (unsigned)p.Wi) u = *reinterpret_cast<const uint4*>(row + (size_t)ix * p.xCs);
                const h2 a0 = __builtin_bit_cast(h2, u.x), a1 = __builtin_bit_cast(h2, u.y), a2 = __builtin_bit_cast(h2, u.z), a3 = __builtin_bit_cast(h2, u.w);
                win[i][0] = (float)a0[0]; win[i][1] = (float)a0[1]; win[i][2] = (float)a1[0]; win[i][3] = (float)a1[1];
                win[i][4] = (float)a2[0]; win[i][5] = (float)a2[1]; win[i][6] = (float)a3[0]; win[i][7] = (float)a3[1];
            }
            const float* wr = p.w + (size_t)(ky * 3) * p.C + c;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 w0 = *reinterpret_cast<const float4*>(wr + (size_t)kx * p.C), w1 = *reinterpret_cast<const float4*>(wr + (size_t)kx * p.C + 4);
                const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                for (int j = 0; j < NPX; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[j][e] = fmaf(win[j * S + kx][e], w[e], acc[j][e]);
            }
        }
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = p.scale ? p.scale[c + e] : 1.f; sh[e] = p.shift ? p.shift[c + e] : 0.f; }
        const size_t opix0 = ((size_t)b * p.Ho + oy) * p.Wo + ox0;
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            if (ox0 + j >= p.Wo) break;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = fmaf(acc[j][e], sc[e], sh[e]);
                if (p.act == OSA_ACT_RELU) o[e] = fmaxf(o[e], 0.f);
                else if (p.act == OSA_ACT_RELU6) o[e] = fminf(fmaxf(o[e], 0.f), 6.f);
                am = fmaxf(am, fabsf(o[e]));
            }
            const h2 b0 = {(_Float16)o[0], (_Float16)o[1]}, b1 = {(_Float16)o[2], (_Float16)o[3]}, b2 = {(_Float16)o[4], (_Float16)o[5]}, b3 = {(_Float16)o[6], (_Float16)o[7]};
            *reinterpret_cast<uint4*>(reinterpret_cast<_Float16*>(p.y) + (opix0 + j) * p.yCs + c) =
                make_uint4(__builtin_bit_cast(unsigned, b0), __builtin_bit_cast(unsigned, b1), __builtin_bit_cast(unsigned, b2), __builtin_bit_cast(unsigned, b3));
        }
    }
    if (p.meta) publish_amax(p.meta, am, am_seen, red);
}

// Vertical strips (K x 1: AttentionModule's 7x1 / 11x1 / 21x1 layers, aggregation.py:105-113), r6: a thread owns 4 channels of NPY consecutive
// output ROWS at one x and loads the NPY + KH - 1 input quads of its column once -- the pixel-run kernel above (KW = 1) re-loads every
// input row KH times across the threads of neighbouring output rows (24 loads per 4 outputs instead of 84 at KH = 21).  Same fmaf order
// per output (ky ascending): bit-identical.
template <int KH, int NPY>
__global__ __launch_bounds__(256) void dwconv2d_vrun_kernel(const DwArgs p, const int vruns) {
    constexpr int WIN = NPY + KH - 1;
    __shared__ float red[4];
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    float am = 0.f;
    const unsigned am_seen = p.meta ? amax_peek(p.meta) : 0u;
    if (idx < p.total) {
        const unsigned nq = (unsigned)(p.C >> 2);
        unsigned r = (unsigned)idx;
        const unsigned q = r % nq; r /= nq;
        const int ox = (int)(r % (unsigned)p.Wo); r /= (unsigned)p.Wo;
        const unsigned vr = r % (unsigned)vruns;
        const int b = (int)(r / (unsigned)vruns);
        const int c = (int)q * 4, oy0 = (int)vr * NPY;
        const float* xb = p.x + (size_t)b * p.Hi * p.Wi * p.xCs + (size_t)ox * p.xCs + c;      // (pad_w = 0, kw = 1: input column = output column)
        float4 win[WIN];
#pragma unroll
        for (int i = 0; i < WIN; ++i) {
            const int iy = oy0 - p.pad_h + i;
            win[i] = ((unsigned)iy < (unsigned)p.Hi) ? *reinterpret_cast<const float4*>(xb + (size_t)iy * p.Wi * p.xCs) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 acc[NPY];
#pragma unroll
        for (int j = 0; j < NPY; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ky = 0; ky < KH; ++ky) {
            const float4 w = *reinterpret_cast<const float4*>(p.w + (size_t)ky * p.C + c);
#pragma unroll
            for (int j = 0; j < NPY; ++j) {
                const float4 v = win[j + ky];
                acc[j].x = fmaf(v.x, w.x, acc[j].x); acc[j].y = fmaf(v.y, w.y, acc[j].y);
                acc[j].z = fmaf(v.z, w.z, acc[j].z); acc[j].w = fmaf(v.w, w.w, acc[j].w);
            }
        }
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + c);
        if (p.shift) sh = *reinterpret_cast<const float4*>(p.shift + c);
#pragma unroll
        for (int j = 0; j < NPY; ++j) {
            const int oy = oy0 + j;
            if (oy >= p.Ho) break;
            float o[4] = {fmaf(acc[j].x, sc.x, sh.x), fmaf(acc[j].y, sc.y, sh.y), fmaf(acc[j].z, sc.z, sh.z), fmaf(acc[j].w, sc.w, sh.w)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (p.act == OSA_ACT_RELU) o[e] = fmaxf(o[e], 0.f);
                else if (p.act == OSA_ACT_RELU6) o[e] = fminf(fmaxf(o[e], 0.f), 6.f);
            }
            const size_t opix = ((size_t)b * p.Ho + oy) * p.Wo + ox;
            if (p.add) {
                const float4 a = *reinterpret_cast<const float4*>(p.add + opix * p.aCs + c);
                o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w;
            }
            *reinterpret_cast<float4*>(p.y + opix * p.yCs + c) = make_float4(o[0], o[1], o[2], o[3]);
            am = fmaxf(am, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
        }
    }
    if (p.meta) publish_amax(p.meta, am, am_seen, red);
}

__global__ __launch_bounds__(256) void dwconv2d_pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int T) {
    const int i = blockIdx.x * 256 + threadIdx.x;      // dst index t*C + c
    if (i >= C * T) return;
    const int t = i / C, c = i - t * C;
    dst[i] = src[(size_t)c * T + t];
}

}  // namespace osa

using namespace osa;

extern "C" int osa_dwconv2d_pack_f32(const float* w_ref, float* w_packed, int C, int kh, int kw, void* stream) {
    OSA_REQUIRE(w_ref && w_packed, "dwconv2d_pack: NULL pointer");
    OSA_REQUIRE(C > 0 && kh > 0 && kw > 0, "dwconv2d_pack: bad dims C=%d k=%dx%d", C, kh, kw);
    const int n = C * kh * kw;
    hipLaunchKernelGGL(dwconv2d_pack_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w_ref, w_packed, C, kh * kw);
    OSA_LAUNCH_CHECK("dwconv2d_pack");
    return 0;
}

static int dwconv2d_impl(const float* x, int x_f16, const float* w_packed,
                         const float* scale, const float* shift, const float* add, float* y, int y_f16,
                         int B, int Hi, int Wi, int C, int xCs, int yCs, int aCs,
                         int kh, int kw, int stride, int pad_h, int pad_w, int dil_h, int dil_w,
                         int act, float* y_meta, void* stream) {
    OSA_REQUIRE(x && w_packed && y, "dwconv2d: NULL pointer");
    OSA_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && C > 0, "dwconv2d: bad dims B=%d H=%d W=%d C=%d", B, Hi, Wi, C);
    OSA_REQUIRE(C % 4 == 0 && xCs % 4 == 0 && yCs % 4 == 0 && xCs >= C && yCs >= C,
                "dwconv2d: C=%d, strides %d/%d must be multiples of 4 with stride >= C", C, xCs, yCs);
    OSA_REQUIRE(((size_t)w_packed & 15) == 0 && ((size_t)x & (x_f16 ? 7 : 15)) == 0 && ((size_t)y & (y_f16 ? 7 : 15)) == 0, "dwconv2d: pointers must be 16-byte (fp16 tensors: 8-byte) aligned");
    if (x_f16 || y_f16) OSA_REQUIRE(kw == 3 && dil_h == 1 && dil_w == 1, "dwconv2d: fp16 tensors are supported for the 3x3 layers only");
    if (add) OSA_REQUIRE(aCs % 4 == 0 && aCs >= C && ((size_t)add & 15) == 0, "dwconv2d: addend stride %d / alignment", aCs);
    if (scale) OSA_REQUIRE(((size_t)scale & 15) == 0, "dwconv2d: scale must be 16-byte aligned");
    if (shift) OSA_REQUIRE(((size_t)shift & 15) == 0, "dwconv2d: shift must be 16-byte aligned");
    OSA_REQUIRE(stride == 1 || stride == 2, "dwconv2d: stride %d unsupported", stride);
    OSA_REQUIRE(kh > 0 && kw > 0 && dil_h > 0 && dil_w > 0 && pad_h >= 0 && pad_w >= 0, "dwconv2d: bad kernel geometry");
    OSA_REQUIRE(act == OSA_ACT_NONE || act == OSA_ACT_RELU || act == OSA_ACT_RELU6, "dwconv2d: act %d unsupported", act);
    DwArgs a;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.add = add; a.y = y; a.meta = y_meta;
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.C = C; a.xCs = xCs; a.yCs = yCs; a.aCs = aCs;
    a.kh = kh; a.kw = kw; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w; a.dil_h = dil_h; a.dil_w = dil_w; a.act = act;
    a.xf16 = x_f16 ? 1 : 0; a.yf16 = y_f16 ? 1 : 0;
    a.Ho = (Hi + 2 * pad_h - dil_h * (kh - 1) - 1) / stride + 1;
    a.Wo = (Wi + 2 * pad_w - dil_w * (kw - 1) - 1) / stride + 1;
    OSA_REQUIRE(a.Ho > 0 && a.Wo > 0, "dwconv2d: empty output");
    // vertical strips (kw = 1, pad_w = 0, stride 1, unit dilation): column-run form
    if (kw == 1 && pad_w == 0 && stride == 1 && dil_h == 1 && dil_w == 1 && !x_f16 && !y_f16 && (kh == 7 || kh == 11 || kh == 21) && !exp_set("OSA_DW_NOVRUN")) {
        constexpr int NPY = 4;
        const int vruns = cdiv(a.Ho, NPY);
        const long long total = (long long)B * vruns * a.Wo * (C / 4);
        if (total < (1ll << 31)) {
            a.total = total;
            const dim3 grid((unsigned)((total + 255) / 256)), block(256);
            if (kh == 7) hipLaunchKernelGGL((dwconv2d_vrun_kernel<7, NPY>), grid, block, 0, (hipStream_t)stream, a, vruns);
            else if (kh == 11) hipLaunchKernelGGL((dwconv2d_vrun_kernel<11, NPY>), grid, block, 0, (hipStream_t)stream, a, vruns);
            else hipLaunchKernelGGL((dwconv2d_vrun_kernel<21, NPY>), grid, block, 0, (hipStream_t)stream, a, vruns);
            OSA_LAUNCH_CHECK("dwconv2d (column runs)");
            return 0;
        }
    }
    // pixel-run form for the shapes the models use (unit dilation; see dwconv2d_run_kernel)
    {
        constexpr int NPX = 4;
        const int runs = cdiv(a.Wo, NPX);
        const long long total = (long long)B * a.Ho * runs * (C / 4);
        void (*fn)(const DwArgs, int) = nullptr;
        if (dil_h == 1 && dil_w == 1 && total < (1ll << 31) && !exp_set("OSA_DW_TAPLOOP")) {
            if (kw == 3 && kh == 3 && x_f16 && y_f16 && C % 8 == 0 && xCs % 8 == 0 && yCs % 8 == 0 && (((size_t)x | (size_t)y) & 15) == 0 && !add) {
                // both tensors fp16: eight channels per thread (16-byte accesses)
                const long long total8 = (long long)B * a.Ho * runs * (C / 8);
                a.total = total8;
                if (stride == 1) hipLaunchKernelGGL((dwconv3x3_h8_kernel<1, NPX>), dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, runs);
                else hipLaunchKernelGGL((dwconv3x3_h8_kernel<2, NPX>), dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, runs);
                OSA_LAUNCH_CHECK("dwconv2d (fp16 x 8)");
                return 0;
            }
            if (kw == 3 && (x_f16 || y_f16)) {
                const int key = (stride == 2 ? 4 : 0) | (x_f16 ? 2 : 0) | (y_f16 ? 1 : 0);
                switch (key) {
                    case 1: fn = dwconv2d_run_kernel<3, 1, NPX, 3, 0, 1>; break;  case 2: fn = dwconv2d_run_kernel<3, 1, NPX, 3, 1, 0>; break;
                    case 3: fn = dwconv2d_run_kernel<3, 1, NPX, 3, 1, 1>; break;  case 5: fn = dwconv2d_run_kernel<3, 2, NPX, 3, 0, 1>; break;
                    case 6: fn = dwconv2d_run_kernel<3, 2, NPX, 3, 1, 0>; break;  default: fn = dwconv2d_run_kernel<3, 2, NPX, 3, 1, 1>; break;
                }
            }
            else if (kw == 3 && stride == 1) fn = dwconv2d_run_kernel<3, 1, NPX>;
            else if (kw == 3 && stride == 2) fn = dwconv2d_run_kernel<3, 2, NPX>;
            else if (kw == 1 && stride == 1) fn = dwconv2d_run_kernel<1, 1, NPX>;
            else if (kw == 7 && stride == 1) fn = dwconv2d_run_kernel<7, 1, NPX>;
            else if (kw == 11 && stride == 1) fn = dwconv2d_run_kernel<11, 1, NPX, 11>;      // (11 is prime: one 14-quad window)
            else if (kw == 21 && stride == 1) fn = dwconv2d_run_kernel<21, 1, NPX, 7>;       // three chunks of 7 taps: 10-quad windows
        }
        if (fn) {
            a.total = total;
            hipLaunchKernelGGL(fn, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, runs);
            OSA_LAUNCH_CHECK("dwconv2d");
            return 0;
        }
    }
    OSA_REQUIRE(!x_f16 && !y_f16, "dwconv2d: fp16 tensors need the pixel-run form (3x3, < 2^31 items)");
    a.total = (long long)B * a.Ho * a.Wo * (C / 4);
    const long long nblk = (a.total + 255) / 256;
    OSA_REQUIRE(nblk < (1ll << 31), "dwconv2d: grid too large");
    hipLaunchKernelGGL(dwconv2d_nhwc_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, a);
    OSA_LAUNCH_CHECK("dwconv2d");
    return 0;
}

extern "C" int osa_dwconv2d_nhwc_f32(const float* x, const float* w_packed,
                                     const float* scale, const float* shift, const float* add, float* y,
                                     int B, int Hi, int Wi, int C, int xCs, int yCs, int aCs,
                                     int kh, int kw, int stride, int pad_h, int pad_w, int dil_h, int dil_w,
                                     int act, float* y_meta, void* stream) {
    return dwconv2d_impl(x, 0, w_packed, scale, shift, add, y, 0, B, Hi, Wi, C, xCs, yCs, aCs, kh, kw, stride, pad_h, pad_w, dil_h, dil_w, act, y_meta, stream);
}

/* the 3 x 3 depthwise layers with fp16 input and / or output tensors (r6: the chain tensors of the f16 mode -- what the reference's autocast
 * moves between MobileV2Residual's convolutions, aggregation.py:63-98); arithmetic as above (fp32 fmaf per tap), strides in elements */
extern "C" int osa_dwconv2d_nhwc_f16io(const void* x, int x_f16, const float* w_packed,
                                       const float* scale, const float* shift, void* y, int y_f16,
                                       int B, int Hi, int Wi, int C, int xCs, int yCs,
                                       int kh, int kw, int stride, int pad_h, int pad_w,
                                       int act, float* y_meta, void* stream) {
    return dwconv2d_impl(static_cast<const float*>(x), x_f16, w_packed, scale, shift, nullptr, static_cast<float*>(y), y_f16, B, Hi, Wi, C, xCs, yCs, 0,
                         kh, kw, stride, pad_h, pad_w, 1, 1, act, y_meta, stream);
}
