/*
 * CPU ORACLE (plain C) -- test infrastructure only, never linked into the product.
 *
 * Independent restatement (scalar loops, reference NCDHW layouts, fp32 storage, the accumulation
 * order of a naive loop nest) of the OpenStereo hot-path arithmetic.  It is checked against the
 * real reference's golden vectors (tests/golden) and against oracle/torch_ref.py by
 * tests/test_oracle_c.py, and exists so that the parity of the HIP engine does not rest on torch's
 * CPU kernels alone (e.g. ConvTranspose3d / trilinear index conventions are re-derived here).
 *
 * Reference lines followed (stereo/modeling/...):
 *   ora_gwc_volume            cost_volume/cost_volume.py:59-78, models/gwcnet/gwcnet_cost_processor.py:13-39
 *   ora_concat_volume         cost_volume/cost_volume.py:81-92, models/psmnet/psmnet_cost_processor.py:9-50,
 *                             models/igev/submodule.py:216-227 (mask_left=0)
 *   ora_conv3d / ora_deconv3d nn.Conv3d / nn.ConvTranspose3d as used in models/gwcnet/gwcnet_disp_processor.py:8-81,
 *                             models/gwcnet/hourglass.py:19-56
 *   ora_bn_act                nn.BatchNorm3d (eval) + ReLU/LeakyReLU of the same files
 *   ora_softargmin            disp_pred/disp_regression.py:8-12
 *   ora_upsample_softargmin   models/gwcnet/gwcnet_disp_processor.py:128-133 (F.interpolate trilinear,
 *                             softmax over D, expectation); align_corners=1: psmnet_cost_processor.py:201-214
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define IDX4(b, c, h, w, C, H, W) ((((size_t)(b) * (C) + (c)) * (H) + (h)) * (W) + (w))
#define IDX5(b, c, d, h, w, C, D, H, W) (((((size_t)(b) * (C) + (c)) * (D) + (d)) * (H) + (h)) * (W) + (w))

void ora_gwc_volume(const float* L, const float* R, float* vol, int B, int C, int H, int W, int D, int G) {
    const int K = C / G;
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
        for (int g = 0; g < G; ++g)
            for (int d = 0; d < D; ++d)
                for (int h = 0; h < H; ++h)
                    for (int w = 0; w < W; ++w) {
                        float v = 0.f;
                        if (w >= d) {
                            float s = 0.f;
                            for (int k = 0; k < K; ++k)
                                s += L[IDX4(b, g * K + k, h, w, C, H, W)] * R[IDX4(b, g * K + k, h, w - d, C, H, W)];
                            v = s / (float)K;
                        }
                        vol[IDX5(b, g, d, h, w, G, D, H, W)] = v;
                    }
}

void ora_concat_volume(const float* L, const float* R, float* vol, int B, int C, int H, int W, int D, int mask_left) {
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int d = 0; d < D; ++d)
                for (int h = 0; h < H; ++h)
                    for (int w = 0; w < W; ++w) {
                        const int ok = (w >= d);
                        vol[IDX5(b, c, d, h, w, 2 * C, D, H, W)] = (ok || !mask_left) ? L[IDX4(b, c, h, w, C, H, W)] : 0.f;
                        vol[IDX5(b, C + c, d, h, w, 2 * C, D, H, W)] = ok ? R[IDX4(b, c, h, w - d, C, H, W)] : 0.f;
                    }
}

/* x [B,Ci,Di,Hi,Wi], w [Co,Ci,kd,kh,kw] -> y [B,Co,Do,Ho,Wo] */
void ora_conv3d(const float* x, const float* w, float* y, int B, int Ci, int Di, int Hi, int Wi, int Co,
                int kd, int kh, int kw, int stride, int pd, int ph, int pw, int dd, int dh, int dw) {
    const int Do = (Di + 2 * pd - dd * (kd - 1) - 1) / stride + 1;
    const int Ho = (Hi + 2 * ph - dh * (kh - 1) - 1) / stride + 1;
    const int Wo = (Wi + 2 * pw - dw * (kw - 1) - 1) / stride + 1;
#pragma omp parallel for collapse(3)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Co; ++co)
            for (int od = 0; od < Do; ++od)
                for (int oh = 0; oh < Ho; ++oh)
                    for (int ow = 0; ow < Wo; ++ow) {
                        float s = 0.f;
                        for (int ci = 0; ci < Ci; ++ci)
                            for (int z = 0; z < kd; ++z) {
                                const int id = od * stride - pd + z * dd;
                                if (id < 0 || id >= Di) continue;
                                for (int yy = 0; yy < kh; ++yy) {
                                    const int ih = oh * stride - ph + yy * dh;
                                    if (ih < 0 || ih >= Hi) continue;
                                    for (int xx = 0; xx < kw; ++xx) {
                                        const int iw = ow * stride - pw + xx * dw;
                                        if (iw < 0 || iw >= Wi) continue;
                                        s += x[IDX5(b, ci, id, ih, iw, Ci, Di, Hi, Wi)] *
                                             w[((((size_t)co * Ci + ci) * kd + z) * kh + yy) * kw + xx];
                                    }
                                }
                            }
                        y[IDX5(b, co, od, oh, ow, Co, Do, Ho, Wo)] = s;
                    }
}

/* scatter definition of ConvTranspose3d: x [B,Ci,Di,Hi,Wi], w [Ci,Co,k,k,k] -> y [B,Co,Do,Ho,Wo],
 * o = i*stride - pad + t,  Do = (Di-1)*stride - 2*pad + k + opad */
void ora_deconv3d(const float* x, const float* w, float* y, int B, int Ci, int Di, int Hi, int Wi, int Co,
                  int k, int stride, int pad, int opad) {
    const int Do = (Di - 1) * stride - 2 * pad + k + opad;
    const int Ho = (Hi - 1) * stride - 2 * pad + k + opad;
    const int Wo = (Wi - 1) * stride - 2 * pad + k + opad;
    memset(y, 0, sizeof(float) * (size_t)B * Co * Do * Ho * Wo);
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Co; ++co)
            for (int ci = 0; ci < Ci; ++ci)
                for (int id = 0; id < Di; ++id)
                    for (int ih = 0; ih < Hi; ++ih)
                        for (int iw = 0; iw < Wi; ++iw) {
                            const float xv = x[IDX5(b, ci, id, ih, iw, Ci, Di, Hi, Wi)];
                            for (int z = 0; z < k; ++z) {
                                const int od = id * stride - pad + z;
                                if (od < 0 || od >= Do) continue;
                                for (int yy = 0; yy < k; ++yy) {
                                    const int oh = ih * stride - pad + yy;
                                    if (oh < 0 || oh >= Ho) continue;
                                    for (int xx = 0; xx < k; ++xx) {
                                        const int ow = iw * stride - pad + xx;
                                        if (ow < 0 || ow >= Wo) continue;
                                        y[IDX5(b, co, od, oh, ow, Co, Do, Ho, Wo)] +=
                                            xv * w[((((size_t)ci * Co + co) * k + z) * k + yy) * k + xx];
                                    }
                                }
                            }
                        }
}

/* y = act( (y - mean)/sqrt(var+eps)*gamma + beta + res ), act: 0 none, 1 relu, 2 leaky(slope). in place. */
void ora_bn_act(float* y, const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                const float* res, int B, int C, size_t S, int act, float slope) {
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            const float a = gamma[c] / sqrtf(var[c] + eps);
            const float bb = beta[c] - mean[c] * a;
            float* p = y + ((size_t)b * C + c) * S;
            const float* r = res ? res + ((size_t)b * C + c) * S : NULL;
            for (size_t i = 0; i < S; ++i) {
                float v = p[i] * a + bb;
                if (r) v += r[i];
                if (act == 1) v = v > 0.f ? v : 0.f;
                else if (act == 2) v = v > 0.f ? v : v * slope;
                p[i] = v;
            }
        }
}

void ora_softargmin(const float* prob, float* out, int B, int D, int H, int W) {
    const size_t HW = (size_t)H * W;
#pragma omp parallel for
    for (int b = 0; b < B; ++b)
        for (size_t i = 0; i < HW; ++i) {
            float s = 0.f;
            for (int d = 0; d < D; ++d) s += prob[((size_t)b * D + d) * HW + i] * (float)d;
            out[(size_t)b * HW + i] = s;
        }
}

static void src_idx(int dst, int in, int out, int align, int* i0, int* i1, float* l1) {
    float s;
    if (align) s = (out > 1) ? (float)dst * ((float)(in - 1) / (float)(out - 1)) : 0.f;
    else {
        s = ((float)in / (float)out) * ((float)dst + 0.5f) - 0.5f;
        if (s < 0.f) s = 0.f;
    }
    *i0 = (int)s;
    if (*i0 > in - 1) *i0 = in - 1;
    *i1 = *i0 + ((*i0 < in - 1) ? 1 : 0);
    *l1 = s - (float)*i0;
}

/* cost [B,Dl,Hl,Wl] -> out [B,H,W]: trilinear upsample to [D,H,W], softmax over D, sum_d d*p_d */
void ora_upsample_softargmin(const float* cost, float* out, int B, int Dl, int Hl, int Wl, int D, int H, int W,
                             int align) {
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y) {
            float* col = (float*)malloc(sizeof(float) * D);
            for (int x = 0; x < W; ++x) {
                int y0, y1, x0, x1; float ly, lx;
                src_idx(y, Hl, H, align, &y0, &y1, &ly);
                src_idx(x, Wl, W, align, &x0, &x1, &lx);
                float m = -INFINITY;
                for (int d = 0; d < D; ++d) {
                    int d0, d1; float ld;
                    src_idx(d, Dl, D, align, &d0, &d1, &ld);
                    const float* c0 = cost + ((size_t)b * Dl + d0) * Hl * Wl;
                    const float* c1 = cost + ((size_t)b * Dl + d1) * Hl * Wl;
                    const float p0 = (1.f - ly) * ((1.f - lx) * c0[y0 * Wl + x0] + lx * c0[y0 * Wl + x1]) +
                                     ly * ((1.f - lx) * c0[y1 * Wl + x0] + lx * c0[y1 * Wl + x1]);
                    const float p1 = (1.f - ly) * ((1.f - lx) * c1[y0 * Wl + x0] + lx * c1[y0 * Wl + x1]) +
                                     ly * ((1.f - lx) * c1[y1 * Wl + x0] + lx * c1[y1 * Wl + x1]);
                    col[d] = (1.f - ld) * p0 + ld * p1;
                    if (col[d] > m) m = col[d];
                }
                double se = 0.0, sd = 0.0;
                for (int d = 0; d < D; ++d) {
                    const double e = exp((double)(col[d] - m));
                    se += e; sd += e * d;
                }
                out[((size_t)b * H + y) * W + x] = (float)(sd / se);
            }
            free(col);
        }
}
