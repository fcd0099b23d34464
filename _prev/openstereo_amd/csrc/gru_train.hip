// ConvGRU gate arithmetic of the TRAINING path, fused (r5; VERDICT r4 missing #3 / next #3c).
//
// Reference: stereo/modeling/models/igev/update.py:36-45 == models/stereobase/gru_blocks.py:261-268
//     z = sigmoid(convz(hx) + cz);  r = sigmoid(convr(hx) + cr);  q = tanh(convq(cat([r * h, x])) + cq);  h' = (1 - z) * h + z * q
// In inference these live in the conv epilogues (OSA_ACT_SIGMOID / OSA_ACT_TANH + osa_gru_combine_f32).  In training the three convolutions
// run through the autograd Functions and everything else used to be ~13 torch elementwise launches forward and ~20 backward per cell --
// 66 cells per StereoBase step, i.e. the bulk of the ~6400 graph nodes that bound the AMP step (profiles/round5).  Two launches forward and
// two backward now:
//     rz:  (pre = [convz | convr](hx) without bias, bz, br, cz, cr, h)  ->  z, r * h
//     q :  (z, qpre = convq([r*h, x]) without bias, bq, cq, h)          ->  h'
// and their gradients (sigmoid / tanh recomputed from the saved pre-activations: nothing but the inputs is kept for backward).
// Tensors are NHWC with a per-tensor channel stride, fp32 or fp16 each (flag per operand: under autocast the context features cz / cr /
// cq and the hidden state arrive in fp16, the conv results in fp32); arithmetic in fp32, one rounding at the store.  HBM-bound elementwise
// work: 16-byte (fp32) / 8-byte (fp16) accesses of 4 consecutive channels per thread, consecutive threads on consecutive channel quads.
#include "osa_common.h"
#include "../../include/openstereo_amd.h"
#include <hip/hip_fp16.h>
#include <cstring>

namespace osa {

struct Ref { const void* p; int cs; int f16; };
struct MRef { void* p; int cs; int f16; };

__device__ __forceinline__ float4 ld4(const Ref& r, long long pix, int c) {
    if (r.f16) {
        const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const __half*>(r.p) + pix * r.cs + c);
        const __half2 a = __builtin_bit_cast(__half2, v.x), b = __builtin_bit_cast(__half2, v.y);
        const float2 fa = __half22float2(a), fb = __half22float2(b);
        return make_float4(fa.x, fa.y, fb.x, fb.y);
    }
    return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(r.p) + pix * r.cs + c);
}
__device__ __forceinline__ void st4(const MRef& r, long long pix, int c, const float4 v) {
    if (r.f16) {
        const __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
        *reinterpret_cast<uint2*>(reinterpret_cast<__half*>(r.p) + pix * r.cs + c) = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
    } else *reinterpret_cast<float4*>(reinterpret_cast<float*>(r.p) + pix * r.cs + c) = v;
}
__device__ __forceinline__ float4 ldb(const float* b, int c) { return b ? *reinterpret_cast<const float4*>(b + c) : make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }
#define OSA_V4(expr) make_float4(expr(x), expr(y), expr(z), expr(w))

struct GruRZ {
    Ref pre, cz, cr, h;           // pre: [z | r] pre-activations, 2C channels
    const float* bz; const float* br;
    MRef z, rh;                   // forward outputs
    Ref dz, drh;                  // backward inputs
    MRef dpre, dh;                // backward outputs (dpre: 2C channels = [dz_pre | dr_pre]; its halves are the gradients of cz / cr too)
    long long total; int C4;
};

__global__ __launch_bounds__(256) void gru_rz_fwd_kernel(const GruRZ a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.total) return;
    const int c = (int)(i % a.C4) * 4; const long long p = i / a.C4;
    const int C = a.C4 * 4;
    const float4 pz = ld4(a.pre, p, c), pr = ld4(a.pre, p, C + c), cz = ld4(a.cz, p, c), cr = ld4(a.cr, p, c), h = ld4(a.h, p, c);
    const float4 bz = ldb(a.bz, c), br = ldb(a.br, c);
#define ZF(k) sigm(pz.k + bz.k + cz.k)
#define RF(k) (sigm(pr.k + br.k + cr.k) * h.k)
    st4(a.z, p, c, OSA_V4(ZF));
    st4(a.rh, p, c, OSA_V4(RF));
#undef ZF
#undef RF
}

__global__ __launch_bounds__(256) void gru_rz_bwd_kernel(const GruRZ a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.total) return;
    const int c = (int)(i % a.C4) * 4; const long long p = i / a.C4;
    const int C = a.C4 * 4;
    const float4 pz = ld4(a.pre, p, c), pr = ld4(a.pre, p, C + c), cz = ld4(a.cz, p, c), cr = ld4(a.cr, p, c), h = ld4(a.h, p, c);
    const float4 bz = ldb(a.bz, c), br = ldb(a.br, c);
    const float4 dz = ld4(a.dz, p, c), drh = ld4(a.drh, p, c);
    float4 gz, gr, gh;
#define ONE(k) { const float z = sigm(pz.k + bz.k + cz.k), r = sigm(pr.k + br.k + cr.k);                 \
                 gz.k = dz.k * z * (1.0f - z); gr.k = drh.k * h.k * r * (1.0f - r); gh.k = drh.k * r; }
    ONE(x) ONE(y) ONE(z) ONE(w)
#undef ONE
    st4(a.dpre, p, c, gz);
    st4(a.dpre, p, C + c, gr);
    st4(a.dh, p, c, gh);
}

struct GruQ {
    Ref z, qpre, cq, h;
    const float* bq;
    MRef out;                     // forward output h'
    Ref dout;                     // backward input
    MRef dz, dqpre, dh;           // backward outputs (dqpre is the gradient of cq too)
    long long total; int C4;
};

__global__ __launch_bounds__(256) void gru_q_fwd_kernel(const GruQ a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.total) return;
    const int c = (int)(i % a.C4) * 4; const long long p = i / a.C4;
    const float4 z = ld4(a.z, p, c), qp = ld4(a.qpre, p, c), cq = ld4(a.cq, p, c), h = ld4(a.h, p, c), bq = ldb(a.bq, c);
#define HF(k) ((1.0f - z.k) * h.k + z.k * tanhf(qp.k + bq.k + cq.k))
    st4(a.out, p, c, OSA_V4(HF));
#undef HF
}

__global__ __launch_bounds__(256) void gru_q_bwd_kernel(const GruQ a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.total) return;
    const int c = (int)(i % a.C4) * 4; const long long p = i / a.C4;
    const float4 z = ld4(a.z, p, c), qp = ld4(a.qpre, p, c), cq = ld4(a.cq, p, c), h = ld4(a.h, p, c), bq = ldb(a.bq, c), g = ld4(a.dout, p, c);
    float4 gz, gq, gh;
#define ONE(k) { const float q = tanhf(qp.k + bq.k + cq.k); gz.k = g.k * (q - h.k); gq.k = g.k * z.k * (1.0f - q * q); gh.k = g.k * (1.0f - z.k); }
    ONE(x) ONE(y) ONE(z) ONE(w)
#undef ONE
    st4(a.dz, p, c, gz);
    st4(a.dqpre, p, c, gq);
    st4(a.dh, p, c, gh);
}

static int check_ref(const char* what, const char* name, const osa_nhwc_ref* r, int C, bool may_be_null = false) {
    if (may_be_null && (!r || !r->ptr)) return 0;
    OSA_REQUIRE(r && r->ptr, "%s: %s is NULL", what, name);
    OSA_REQUIRE(r->cs >= C && r->cs % 4 == 0, "%s: %s has channel stride %d for %d channels (must be >= C and a multiple of 4)", what, name, r->cs, C);
    OSA_REQUIRE(((size_t)r->ptr & (r->f16 ? 7 : 15)) == 0, "%s: %s is not %d-byte aligned", what, name, r->f16 ? 8 : 16);
    return 0;
}
static Ref cref(const osa_nhwc_ref* r) { Ref o; o.p = r->ptr; o.cs = r->cs; o.f16 = r->f16 ? 1 : 0; return o; }
static MRef mref(const osa_nhwc_ref* r) { MRef o; o.p = r->ptr; o.cs = r->cs; o.f16 = r->f16 ? 1 : 0; return o; }

}  // namespace osa

using namespace osa;

#define OSA_GRU_COMMON(what)                                                                                              \
    OSA_REQUIRE(npix > 0 && C > 0 && C % 4 == 0, what ": bad dims npix=%lld C=%d (C must be a multiple of 4)", npix, C);  \
    const long long total = npix * (C / 4);                                                                              \
    OSA_REQUIRE((total + 255) / 256 < (1ll << 31), what ": grid too large");                                             \
    OSA_REQUIRE((((size_t)bias_a | (size_t)bias_b) & 15) == 0, what ": bias vectors must be 16-byte aligned");

extern "C" int osa_gru_gates_rz_fwd(const osa_nhwc_ref* pre, const float* bias_z, const float* bias_r, const osa_nhwc_ref* cz,
                                    const osa_nhwc_ref* cr, const osa_nhwc_ref* h, const osa_nhwc_ref* z_out, const osa_nhwc_ref* rh_out,
                                    long long npix, int C, void* stream) {
    const float* bias_a = bias_z; const float* bias_b = bias_r;
    OSA_GRU_COMMON("gru_gates_rz_fwd")
    if (check_ref("gru_gates_rz_fwd", "pre", pre, 2 * C) || check_ref("gru_gates_rz_fwd", "cz", cz, C) || check_ref("gru_gates_rz_fwd", "cr", cr, C) ||
        check_ref("gru_gates_rz_fwd", "h", h, C) || check_ref("gru_gates_rz_fwd", "z_out", z_out, C) || check_ref("gru_gates_rz_fwd", "rh_out", rh_out, C)) return -1;
    GruRZ a;
    memset(&a, 0, sizeof(a));
    a.pre = cref(pre); a.cz = cref(cz); a.cr = cref(cr); a.h = cref(h); a.bz = bias_z; a.br = bias_r; a.z = mref(z_out); a.rh = mref(rh_out);
    a.total = total; a.C4 = C / 4;
    hipLaunchKernelGGL(gru_rz_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    OSA_LAUNCH_CHECK("gru_gates_rz_fwd");
    return 0;
}

extern "C" int osa_gru_gates_rz_bwd(const osa_nhwc_ref* pre, const float* bias_z, const float* bias_r, const osa_nhwc_ref* cz,
                                    const osa_nhwc_ref* cr, const osa_nhwc_ref* h, const osa_nhwc_ref* dz, const osa_nhwc_ref* drh,
                                    const osa_nhwc_ref* dpre_out, const osa_nhwc_ref* dh_out, long long npix, int C, void* stream) {
    const float* bias_a = bias_z; const float* bias_b = bias_r;
    OSA_GRU_COMMON("gru_gates_rz_bwd")
    if (check_ref("gru_gates_rz_bwd", "pre", pre, 2 * C) || check_ref("gru_gates_rz_bwd", "cz", cz, C) || check_ref("gru_gates_rz_bwd", "cr", cr, C) ||
        check_ref("gru_gates_rz_bwd", "h", h, C) || check_ref("gru_gates_rz_bwd", "dz", dz, C) || check_ref("gru_gates_rz_bwd", "drh", drh, C) ||
        check_ref("gru_gates_rz_bwd", "dpre_out", dpre_out, 2 * C) || check_ref("gru_gates_rz_bwd", "dh_out", dh_out, C)) return -1;
    GruRZ a;
    memset(&a, 0, sizeof(a));
    a.pre = cref(pre); a.cz = cref(cz); a.cr = cref(cr); a.h = cref(h); a.bz = bias_z; a.br = bias_r;
    a.dz = cref(dz); a.drh = cref(drh); a.dpre = mref(dpre_out); a.dh = mref(dh_out);
    a.total = total; a.C4 = C / 4;
    hipLaunchKernelGGL(gru_rz_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    OSA_LAUNCH_CHECK("gru_gates_rz_bwd");
    return 0;
}

extern "C" int osa_gru_gates_q_fwd(const osa_nhwc_ref* z, const osa_nhwc_ref* qpre, const float* bias_q, const osa_nhwc_ref* cq,
                                   const osa_nhwc_ref* h, const osa_nhwc_ref* out, long long npix, int C, void* stream) {
    const float* bias_a = bias_q; const float* bias_b = nullptr;
    OSA_GRU_COMMON("gru_gates_q_fwd")
    if (check_ref("gru_gates_q_fwd", "z", z, C) || check_ref("gru_gates_q_fwd", "qpre", qpre, C) || check_ref("gru_gates_q_fwd", "cq", cq, C) ||
        check_ref("gru_gates_q_fwd", "h", h, C) || check_ref("gru_gates_q_fwd", "out", out, C)) return -1;
    GruQ a;
    memset(&a, 0, sizeof(a));
    a.z = cref(z); a.qpre = cref(qpre); a.cq = cref(cq); a.h = cref(h); a.bq = bias_q; a.out = mref(out);
    a.total = total; a.C4 = C / 4;
    hipLaunchKernelGGL(gru_q_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    OSA_LAUNCH_CHECK("gru_gates_q_fwd");
    return 0;
}

extern "C" int osa_gru_gates_q_bwd(const osa_nhwc_ref* z, const osa_nhwc_ref* qpre, const float* bias_q, const osa_nhwc_ref* cq,
                                   const osa_nhwc_ref* h, const osa_nhwc_ref* dout, const osa_nhwc_ref* dz_out, const osa_nhwc_ref* dqpre_out,
                                   const osa_nhwc_ref* dh_out, long long npix, int C, void* stream) {
    const float* bias_a = bias_q; const float* bias_b = nullptr;
    OSA_GRU_COMMON("gru_gates_q_bwd")
    if (check_ref("gru_gates_q_bwd", "z", z, C) || check_ref("gru_gates_q_bwd", "qpre", qpre, C) || check_ref("gru_gates_q_bwd", "cq", cq, C) ||
        check_ref("gru_gates_q_bwd", "h", h, C) || check_ref("gru_gates_q_bwd", "dout", dout, C) || check_ref("gru_gates_q_bwd", "dz_out", dz_out, C) ||
        check_ref("gru_gates_q_bwd", "dqpre_out", dqpre_out, C) || check_ref("gru_gates_q_bwd", "dh_out", dh_out, C)) return -1;
    GruQ a;
    memset(&a, 0, sizeof(a));
    a.z = cref(z); a.qpre = cref(qpre); a.cq = cref(cq); a.h = cref(h); a.bq = bias_q; a.dout = cref(dout);
    a.dz = mref(dz_out); a.dqpre = mref(dqpre_out); a.dh = mref(dh_out);
    a.total = total; a.C4 = C / 4;
    hipLaunchKernelGGL(gru_q_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    OSA_LAUNCH_CHECK("gru_gates_q_bwd");
    return 0;
}
