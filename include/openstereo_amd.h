/*
 * openstereo_amd.h -- C ABI of the MI355X (gfx950) cost-volume engine.
 *
 * Every entry point takes plain device pointers + sizes and a HIP stream
 * (`void* stream` == hipStream_t, NULL = default stream).  No torch types.
 * All tensors are fp32 and live in HBM.  Return value: 0 on success, <0 on
 * error; the message is available from osa_last_error() (thread local).
 * Nothing here ever falls back to a CPU path.
 *
 * Two HBM layouts are understood for 5-D volumes:
 *   OSA_NCDHW  reference layout  [B][C][D][H][W]        (what OpenStereo returns)
 *   OSA_NDHWC  engine layout     [B][D][H][W][C]        (channels innermost; what
 *              the MFMA aggregation kernels consume: one voxel = one contiguous
 *              channel vector, 16-byte aligned).
 *
 * Reference interfaces replaced (paths relative to the OpenStereo tree):
 *   osa_build_volume_f32        stereo/modeling/cost_volume/cost_volume.py:59-92
 *                               models/gwcnet/gwcnet_cost_processor.py:13-68
 *                               models/psmnet/psmnet_cost_processor.py:9-50
 *                               models/igev/submodule.py:158-177,216-227
 *   osa_corr_volume_f32         stereo/modeling/cost_volume/cost_volume.py:32-41,95-105
 *   osa_conv3d_* / osa_deconv3d_*  nn.Conv3d/ConvTranspose3d + BatchNorm3d(eval) + act of
 *                               models/gwcnet/gwcnet_disp_processor.py:8-81,
 *                               models/gwcnet/hourglass.py:5-56,
 *                               models/psmnet/psmnet_cost_processor.py:53-221,
 *                               stereo/modeling/common/basic_block_3d.py:5-38
 *   osa_softargmin_f32          stereo/modeling/disp_pred/disp_regression.py:8-12,
 *                               models/gwcnet/gwcnet_disp_processor.py:22-26,
 *                               models/psmnet/psmnet_disp_processor.py:6-74
 *   osa_softmax_softargmin_f32  F.softmax(dim=1) + the above (stereobase_gru.py:163-164,
 *                               lightstereo.py:55-56, igev_stereo.py:164-165)
 *   osa_upsample_softargmin_f32 F.interpolate(trilinear)+softmax+regression,
 *                               models/gwcnet/gwcnet_disp_processor.py:99-133,
 *                               models/psmnet/psmnet_cost_processor.py:201-214
 */
#ifndef OPENSTEREO_AMD_H
#define OPENSTEREO_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSA_ABI_VERSION 7
#define OSA_META_FLOATS 128   /* floats per range block (osa_f16x3_ranges) */

enum { OSA_NCDHW = 0, OSA_NDHWC = 1 };
enum { OSA_ACT_NONE = 0, OSA_ACT_RELU = 1, OSA_ACT_LEAKY = 2, OSA_ACT_RELU6 = 3, OSA_ACT_SIGMOID = 4, OSA_ACT_TANH = 5 };
/* OR'ed into `act`: the gate tensor is a plain multiplier (LightStereo AttentionModule, attn * cost,
 * models/lightstereo/aggregation.py:134) instead of logits passed through a sigmoid */
enum { OSA_GATE_RAW = 16 };
/* OR'ed into `act`: the gate multiplies output channels [0, n) only (n a multiple of 4; 0 = all of them).  One launch can then
 * produce [r * h | z] for ConvGRU (update.py:38-39: z and r convolve the same [h | x]): the weights of convr and convz
 * concatenated, the hidden state as raw gate of the first half. */
#define OSA_GATE_CHANNELS(n) ((int)((unsigned)(n) << 16))
/* OR'ed into `act` of the f16x3 conv / deconv calls: "split" activation tensors.  Same bytes per voxel as
 * fp32 NDHWC, but every 16-channel chunk is stored as [16 x fp16 hi | 16 x fp16 lo] (x = hi + lo, the
 * image the kernels stage into LDS), so a consumer copies instead of splitting and a producer splits each
 * value once.  Channel counts must be multiples of 16.  Internal to chains of engine layers. */
enum { OSA_IN_SPLIT = 32, OSA_OUT_SPLIT = 64, OSA_RES_SPLIT = 128, OSA_REDIR_SPLIT = 256 };
/* OR'ed into `act` (with OSA_ACT_RELU): y = relu(residual + relu(bn(conv(x)))) instead of relu(bn(conv(x)) + residual) -- the
 * ResidualBlock of MultiBasicEncoder (models/igev/extractor.py:48-60, models/stereobase/gru_blocks.py:48-60) applies the ReLU to the
 * branch before the sum.  Plain fp32 outputs only (no split output, no gate). */
enum { OSA_RES_AFTER_ACT = 512 };

/* ---- misc ------------------------------------------------------------- */
int         osa_abi_version(void);
const char* osa_last_error(void);
/* device the library was built for ("gfx950") */
const char* osa_target_arch(void);

/* ---- cost-volume constructors (a1-a4) ----------------------- */
/*
 * Fused group-wise-correlation + concatenation volume.
 *   gwc part   : vol[b, c_off+g, d, h, w]        = (1/K) sum_k L[b,gK+k,h,w] * R[b,gK+k,h,w-d]
 *   concat part: vol[b, c_off+G+c, d, h, w]      = Lc[b,c,h,w]
 *                vol[b, c_off+G+Cc+c, d, h, w]   = Rc[b,c,h,w-d]
 *   everything is 0 where w < d (mask_left_concat=0 reproduces the IGEV copy
 *   that leaves the left half unmasked).
 * Either part may be absent: C==0 (no gwc part) or Cc==0 (no concat part).
 * left/right feature maps are NCHW contiguous.  `vol` has `vol_channels`
 * channels in `layout`; this call writes channels [c_off, c_off+G+2*Cc).
 * vol_meta: NULL, or the volume's range block (see osa_f16x3_ranges): the NDHWC kernels fold max |value|
 * into it so that an f16x3 consumer can scale its operands.
 */
int osa_build_volume_f32(const float* left_gwc, const float* right_gwc, int C, int num_groups,
                         const float* left_cat, const float* right_cat, int Cc,
                         float* vol, int layout, int vol_channels, int c_off,
                         int B, int H, int W, int maxdisp, int mask_left_concat,
                         float* vol_meta, void* stream);

/* Same, for channels-last feature maps [B][H][W][stride] (what the engine's own 2-D backbone
 * produces; *_stride = floats per pixel, >= channel count).  Always writes the NDHWC volume. */
int osa_build_volume_nhwc_f32(const float* left_gwc, const float* right_gwc, int C, int num_groups, int gwc_stride,
                              const float* left_cat, const float* right_cat, int Cc, int cat_stride,
                              float* vol, int vol_channels, int c_off,
                              int B, int H, int W, int maxdisp, int mask_left_concat, float* vol_meta, void* stream);
/* Dormant volume variants of cost_volume.py (no shipped config enables them), NCHW in, NCDHW out:
 *   mode 0  CoExCostVolume.forward (:17-29): out[B,groups,planes,H,W] = sum over the group's channels of x[w] * y[w-d], planes = maxdisp + 1
 *   mode 1  compute_volume(side='left')  (:44-56): out[B,C,planes,H,W] = reference[w] - target[w-d]   (w >= d, else 0)
 *   mode 2  compute_volume(side='right'):          out[B,C,planes,H,W] = target[w+d] - reference[w]   (w < W-d; plane 0: reference - target)
 *   mode 3  build_sub_volume (:108-117):           out[B,planes,H,W]   = sum_c |l[w] - r[w-d]| (w >= d), sum_c |l[w]| (w < d) */
/* PSMNet cat_fms with a general disparity sampling (psmnet_cost_processor.py:9-50: start_disp, dilation; cfgs/psmnet uses 0 / 1, which
 * osa_build_volume_f32 covers): out [B, 2C, n_samples, H, W]; disp_index: DEVICE array of the n_samples integer disparities
 * (int(torch.linspace(start, start + max_disp - 1, n)) as the reference computes them; negative values sample to the left). */
int osa_cat_fms_f32(const float* reference_fm, const float* target_fm, float* out, const int* disp_index,
                    int B, int C, int H, int W, int n_samples, void* stream);
int osa_pair_volume_f32(const float* left, const float* right, float* out,
                        int B, int C, int groups, int H, int W, int planes, int mode, void* stream);

/* correlation layer: vol[b,d,h,w] = mean_c L[b,c,h,w]*R[b,c,h,w-d], 0 for w<d. vol is [B,D,H,W]. */
int osa_corr_volume_f32(const float* left, const float* right, float* vol,
                        int B, int C, int H, int W, int maxdisp, void* stream);

/* ---- layout helpers ---------------------------------------------------- */
/* x [B][C][S] -> y [B][S][yCs] channels [c_off, c_off+C)   (S = D*H*W) */
int osa_ncdhw_to_ndhwc_f32(const float* x, float* y, int B, int C, long long S,
                           int yCs, int c_off, void* stream);
/* x [B][S][xCs] channels [c_off, c_off+C) -> y [B][C][S] */
int osa_ndhwc_to_ncdhw_f32(const float* x, float* y, int B, int C, long long S,
                           int xCs, int c_off, void* stream);

/* ---- 3-D aggregation convolutions (a6-a8) ------------------- */
/*
 * Weights are packed once per layer into the MFMA operand order
 *   [cin-chunk][tap][k-octet][half][CoutPadded][4]   (fp32)
 * by osa_conv3d_pack_f32 (ordinary conv, reference layout [Co][Ci][kd][kh][kw])
 * or osa_deconv3d_pack_f32 (transposed conv, reference layout [Ci][Co][kd][kh][kw];
 * stride 2; the 8 output-parity classes are packed back to back).
 * osa_*_packed_floats gives the size of the packed buffer in floats.
 */
size_t osa_conv3d_packed_floats(int Ci, int Co, int kd, int kh, int kw);
int    osa_conv3d_pack_f32(const float* w_ref, float* w_packed,
                           int Ci, int Co, int kd, int kh, int kw, void* stream);
size_t osa_deconv3d_packed_floats(int Ci, int Co, int k);
int    osa_deconv3d_pack_f32(const float* w_ref, float* w_packed,
                             int Ci, int Co, int k, int pad, void* stream);

/*
 * y = act( conv3d(x, w) * scale[co] + shift[co] + residual ) [* sigmoid(gate)]
 *   x        : NDHWC, voxel stride xCs floats (>= Ci, multiple of 4), reads channels [0,Ci)
 *   y        : NDHWC, voxel stride yCs floats, writes channels [0,Co)
 *   residual : NDHWC with voxel stride rCs, same spatial size as y, or NULL
 *   scale/shift: per-output-channel (folded eval-mode BatchNorm; NULL = 1 / 0)
 *   gate_logits: NULL, or NHWC [B][Ho][Wo][gCs] logits: the result is multiplied by
 *                sigmoid(gate[b,h,w,co]) broadcast over D -- the FeatureAtt channel gating of
 *                models/stereobase/igev_blocks.py:35-48 / models/igev/submodule.py (cv = sigmoid(att) * cv)
 *   stride is isotropic (1 or 2) -- a dimension of size 1 with kernel 1 is not strided;
 *   pad_* / dil_* per dimension.  act: OSA_ACT_*; slope for leaky.
 *   Output dims follow the PyTorch formula.
 */
int osa_conv3d_ndhwc_f32(const float* x, const float* w_packed,
                         const float* scale, const float* shift, const float* residual,
                         float* y,
                         int B, int Di, int Hi, int Wi, int Ci, int xCs,
                         int Co, int yCs, int rCs,
                         int kd, int kh, int kw, int stride,
                         int pad_d, int pad_h, int pad_w,
                         int dil_d, int dil_h, int dil_w,
                         const float* gate_logits, int gCs,
                         int act, float slope, void* stream);

/*
 * y = act( conv_transpose3d(x, w, stride=2, padding=pad, output_padding=opad) * scale + shift + residual )
 * k = 3 (pad 1, opad 1: GwcNet/PSMNet) or k = 4 (pad 1, opad 0: StereoBase/IGEV); Do = 2*Di.
 * Implemented as 8 output-parity sub-convolutions, no zero insertion.
 */
int osa_deconv3d_ndhwc_f32(const float* x, const float* w_packed,
                           const float* scale, const float* shift, const float* residual,
                           float* y,
                           int B, int Di, int Hi, int Wi, int Ci, int xCs,
                           int Co, int yCs, int rCs,
                           int k, int pad, int opad,
                           const float* gate_logits, int gCs,
                           int act, float slope, void* stream);

/*
 * Split-precision mode ("f16x3"): every fp32 operand is split into two fp16 parts (hi + lo, 22
 * significant bits) and each product is evaluated as Ahi*Bhi + Ahi*Blo + Alo*Bhi on the fp16
 * matrix cores with fp32 accumulation -- fp32-class accuracy at a fraction of the fp32-MFMA time.
 * Tensors in HBM stay fp32.  Weights are packed by the *_pack_f16x3 calls (same buffer size as the
 * f32 packing) after multiplication by `wscale`, a power of two that moves them into the fp16
 * normal range; pass out_scale = 1/wscale to the matching conv call.  Activations: osa_f16x3_ranges below.
 */
/*
 * Operand ranges of the f16x3 mode.  fp16 halves have a 5-bit exponent, so every operand tensor is brought
 * into range by a per-tensor POWER-OF-TWO scale (exact to apply and to undo): the largest |value| lands in
 * [2^14, 2^15), everything down to 2^-18 of it keeps 22 significant bits, smaller values degrade gracefully
 * and nothing is clamped.  The scale is derived ON THE DEVICE from a "range block" of OSA_META_FLOATS floats
 * (512 bytes) that travels with each activation tensor (no host synchronisation, hipGraph friendly):
 *     meta[16*s], s = 0..7   running max |value| of the tensor, in 8 slots on separate cache lines: every producing
 *              workgroup folds its outputs into one slot with (at most) one atomic max; max|x| = max over the slots
 *              (zero the block before the first producer runs; a host that knows max|x| may write it to meta[0])
 *     meta[1]  split tensors only: the scale their stored hi/lo halves carry (written by the producer)
 *     others   reserved (work-queue words of the persistent kernels)
 * A plain fp32 input is scaled by pow2(max|x|) while it is staged; a split OUTPUT must choose its scale
 * before the maximum is known, from the rigorous bound
 *     max|y| <= bound_coef[0] * max|x| + bound_coef[1] (+ max|residual|) (+ redir_bound_coef[0] * max|rx| + redir_bound_coef[1])
 * with bound_coef = { max_co |bn_scale[co]| * sum|w[co]|, max_co |bn_shift[co]| } (device pointer, 2 floats).
 * Any pointer may be NULL (and `ranges` itself): that operand is then used unscaled / not tracked.  A value
 * that leaves the fp16 range becomes inf/NaN in the output -- never a silently clamped number.
 */
typedef struct osa_f16x3_ranges {
    const float* x_meta;            /* range block of x */
    const float* residual_meta;     /* range block of the residual */
    const float* redir_meta;        /* range block of the fused redir input (osa_deconv3d_redir_*) */
    float*       y_meta;            /* range block of y (updated) */
    const float* bound_coef;        /* 2 floats, see above */
    const float* redir_bound_coef;  /* 2 floats of the fused redir layer */
    const float* weight_scale;      /* {wscale, 1 / wscale} written by an osa_*_pack_*_auto call (device memory): when non-NULL,
                                       weight_scale[1] replaces the `out_scale` argument of the conv call (ABI v3) */
} osa_f16x3_ranges;

/* Packing with the power-of-two weight pre-scale derived ON THE DEVICE (training: the weights change every optimizer step, and a
 * host-side scale would cost a synchronisation per layer, role and step and rule out hipGraph capture of the step):
 * w_amax = max |w| (1 float, device), scale_out = {wscale, 1 / wscale} (2 floats, device, written by the call) -> pass it to the
 * conv / deconv call through osa_f16x3_ranges.weight_scale.  Same packed image as the *_f16x3 calls given that scale. */
int osa_conv3d_pack_ex_auto(const float* w_ref, float* w_packed, int Ci, int Co, int kd, int kh, int kw,
                            int src_transposed, int flip, const float* w_amax, float* scale_out, void* stream);
int osa_deconv3d_pack_f16x3_auto(const float* w_ref, float* w_packed, int Ci, int Co, int k, int pad,
                                 const float* w_amax, float* scale_out, void* stream);
int osa_deconv2d_pack_f16x3_auto(const float* w_ref, float* w_packed, int Ci, int Co, int k, int pad,
                                 const float* w_amax, float* scale_out, void* stream);

int osa_conv3d_pack_f16x3(const float* w_ref, float* w_packed,
                          int Ci, int Co, int kd, int kh, int kw, float wscale, void* stream);
int osa_deconv3d_pack_f16x3(const float* w_ref, float* w_packed,
                            int Ci, int Co, int k, int pad, float wscale, void* stream);
int osa_conv3d_ndhwc_f16x3(const float* x, const float* w_packed,
                           const float* scale, const float* shift, const float* residual,
                           float* y,
                           int B, int Di, int Hi, int Wi, int Ci, int xCs,
                           int Co, int yCs, int rCs,
                           int kd, int kh, int kw, int stride,
                           int pad_d, int pad_h, int pad_w,
                           int dil_d, int dil_h, int dil_w,
                           const float* gate_logits, int gCs,
                           int act, float slope, float out_scale, const osa_f16x3_ranges* ranges, void* stream);
int osa_deconv3d_ndhwc_f16x3(const float* x, const float* w_packed,
                             const float* scale, const float* shift, const float* residual,
                             float* y,
                             int B, int Di, int Hi, int Wi, int Ci, int xCs,
                             int Co, int yCs, int rCs,
                             int k, int pad, int opad,
                             const float* gate_logits, int gCs,
                             int act, float slope, float out_scale, const osa_f16x3_ranges* ranges, void* stream);

/*
 * Transposed conv with a fused 1x1x1 "redir" branch: GwcNet Hourglass,
 *   conv6 = relu(conv6(conv5) + redir1(x))        models/gwcnet/hourglass.py:54 (redir1 = Conv3d 1x1x1 + BN, :43)
 *   y = act( BN(deconv3d(x, w)) + BN_r(conv1x1x1(rx, rw)) )
 * rx is an NDHWC tensor at OUTPUT resolution with rCi <= 64 channels (stride rxCs); rw_packed comes from
 * osa_conv3d_pack_{f32,f16x3}(w_r, ..., Ci = rCi, Co, 1, 1, 1); rscale / rshift are its folded BN.  The 1x1x1
 * product runs on the MFMA inside the epilogue (x rows loaded straight into A-operand order), so the redir
 * tensor is never written or read back: 400 MB of HBM traffic less per GwcNet hourglass at 544x960.  Same
 * arithmetic in the same order as the two separate launches (bit-identical results).
 */
int osa_deconv3d_redir_ndhwc_f32(const float* x, const float* w_packed, const float* scale, const float* shift, float* y,
                                 int B, int Di, int Hi, int Wi, int Ci, int xCs, int Co, int yCs,
                                 int k, int pad, int opad,
                                 const float* rx, int rxCs, int rCi, const float* rw_packed,
                                 const float* rscale, const float* rshift,
                                 int act, float slope, void* stream);
int osa_deconv3d_redir_ndhwc_f16x3(const float* x, const float* w_packed, const float* scale, const float* shift, float* y,
                                   int B, int Di, int Hi, int Wi, int Ci, int xCs, int Co, int yCs,
                                   int k, int pad, int opad,
                                   const float* rx, int rxCs, int rCi, const float* rw_packed,
                                   const float* rscale, const float* rshift, float r_out_scale,
                                   int act, float slope, float out_scale, const osa_f16x3_ranges* ranges, void* stream);

/*
 * 2-D transposed convolution (nn.ConvTranspose2d, stride 2) on an NHWC map: the D = 1 case of the
 * fused parity-class kernel (4 classes).  LightStereo Aggregation.conv5 / conv6
 * (stereo/modeling/models/lightstereo/aggregation.py:29-35): k = 3, pad 1, opad 1, + BatchNorm2d,
 * with the redir residual and ReLU of :57-58 fused.  Weight layout of the reference: [Ci][Co][kh][kw].
 */
size_t osa_deconv2d_packed_floats(int Ci, int Co, int k);
int    osa_deconv2d_pack_f32(const float* w_ref, float* w_packed, int Ci, int Co, int k, int pad, void* stream);
int    osa_deconv2d_pack_f16x3(const float* w_ref, float* w_packed, int Ci, int Co, int k, int pad,
                               float wscale, void* stream);
int osa_deconv2d_nhwc_f32(const float* x, const float* w_packed,
                          const float* scale, const float* shift, const float* residual,
                          float* y,
                          int B, int Hi, int Wi, int Ci, int xCs,
                          int Co, int yCs, int rCs,
                          int k, int pad, int opad,
                          const float* gate_logits, int gCs,
                          int act, float slope, void* stream);
int osa_deconv2d_nhwc_f16x3(const float* x, const float* w_packed,
                            const float* scale, const float* shift, const float* residual,
                            float* y,
                            int B, int Hi, int Wi, int Ci, int xCs,
                            int Co, int yCs, int rCs,
                            int k, int pad, int opad,
                            const float* gate_logits, int gCs,
                            int act, float slope, float out_scale, const osa_f16x3_ranges* ranges, void* stream);

/*
 * Depthwise 2-D convolution on an NHWC map (groups == channels): LightStereo MobileV2Residual.dwconv
 * (aggregation.py:79-83, 3x3, stride 1/2, + BatchNorm2d + ReLU6) and the strip convolutions of
 * AttentionModule (aggregation.py:105-113: 1x7, 7x1, 1x11, 11x1, 1x21, 21x1, with bias).
 *   y = act( dwconv(x, w) * scale[c] + shift[c] ) + add
 *   w_packed: [kh*kw][C], made by osa_dwconv2d_pack_f32 from the reference layout [C][1][kh][kw]
 *   scale/shift: folded eval BatchNorm or (1, bias); NULL = 1 / 0.   add: NHWC addend (stride aCs) or NULL
 *   act: OSA_ACT_NONE / OSA_ACT_RELU / OSA_ACT_RELU6.   C and all strides multiples of 4, fp32 exact (fmaf per tap).
 *   y_meta: NULL or y's range block (max |y| is folded into it for f16x3 consumers, see osa_f16x3_ranges).
 */
int osa_dwconv2d_pack_f32(const float* w_ref, float* w_packed, int C, int kh, int kw, void* stream);
int osa_dwconv2d_nhwc_f32(const float* x, const float* w_packed,
                          const float* scale, const float* shift, const float* add, float* y,
                          int B, int Hi, int Wi, int C, int xCs, int yCs, int aCs,
                          int kh, int kw, int stride, int pad_h, int pad_w, int dil_h, int dil_w,
                          int act, float* y_meta, void* stream);
/* The 3 x 3 depthwise layers with fp16 input and / or output tensors (r6): the chain tensors of the f16 mode between MobileV2Residual's
 * 1 x 1 expansion, depthwise and 1 x 1 projection convolutions (aggregation.py:63-98; under the reference's autocast these are fp16
 * tensors too).  Same arithmetic (fp32 fmaf per tap, folded BN, activation); channel strides in elements, fp16 tensors 8-byte aligned. */
int osa_dwconv2d_nhwc_f16io(const void* x, int x_f16, const float* w_packed,
                            const float* scale, const float* shift, void* y, int y_f16,
                            int B, int Hi, int Wi, int C, int xCs, int yCs,
                            int kh, int kw, int stride, int pad_h, int pad_w,
                            int act, float* y_meta, void* stream);

/*
 * ConvGRU state update (models/igev/update.py:42, models/stereobase/gru_blocks.py ConvGRU):
 *   out[p][c] = (1 - z[p][c]) * h[p][c] + z[p][c] * q[p][c]       for npix pixels x C channels, NHWC with
 * per-tensor channel strides.  z = sigmoid(convz(hx) + cz) and q = tanh(convq([r*h, x]) + cq) come out of
 * the conv epilogues (OSA_ACT_SIGMOID / OSA_ACT_TANH, residual = cz / cq, r*h = sigmoid(...) with h as raw gate).
 */
int osa_gru_combine_f32(const float* z, const float* q, const float* h, float* out,
                        long long npix, int C, int zCs, int qCs, int hCs, int oCs, float* out_meta, void* stream);

/* ConvGRU gate arithmetic of the TRAINING path, fused (r5, ABI v5; csrc/gru_train.hip).  Reference: models/igev/update.py:36-45 ==
 * models/stereobase/gru_blocks.py:261-268
 *     z = sigmoid(convz(hx) + cz);  r = sigmoid(convr(hx) + cr);  q = tanh(convq(cat([r * h, x])) + cq);  h' = (1 - z) * h + z * q.
 * The three convolutions run through the conv / dgrad / wgrad entry points; these four calls replace the ~13 forward and ~20 backward
 * elementwise launches per cell of the torch composition (66 cells per StereoBase training step):
 *   rz_fwd: pre = [convz(hx) | convr(hx)] WITHOUT bias (2C channels), bias_z / bias_r (C floats each, or NULL), cz, cr, h -> z, r * h
 *   rz_bwd: the same inputs + dz, d(r*h) -> dpre = [d pre_z | d pre_r] (2C channels; its halves are also the gradients of cz and cr, and
 *           summed over the pixels those of the biases) and the part of dh that flows through r * h
 *   q_fwd : z, qpre = convq([r*h, x]) without bias, bias_q, cq, h -> h'
 *   q_bwd : the same inputs + dh' -> dz, dqpre (= dcq; summed: dbias_q), the direct part of dh
 * sigmoid / tanh are recomputed from the saved pre-activations in the backward calls.  Every tensor is NHWC over npix pixels with its own
 * channel stride (elements, % 4 == 0) and element type: fp32 (16-byte aligned) or fp16 (f16 = 1, 8-byte aligned) -- under autocast the
 * context features and the hidden state arrive in fp16 next to fp32 conv results.  fp32 arithmetic, one rounding at the store.  C % 4 == 0. */
typedef struct osa_nhwc_ref { void* ptr; int cs; int f16; } osa_nhwc_ref;
int osa_gru_gates_rz_fwd(const osa_nhwc_ref* pre, const float* bias_z, const float* bias_r, const osa_nhwc_ref* cz, const osa_nhwc_ref* cr,
                         const osa_nhwc_ref* h, const osa_nhwc_ref* z_out, const osa_nhwc_ref* rh_out, long long npix, int C, void* stream);
int osa_gru_gates_rz_bwd(const osa_nhwc_ref* pre, const float* bias_z, const float* bias_r, const osa_nhwc_ref* cz, const osa_nhwc_ref* cr,
                         const osa_nhwc_ref* h, const osa_nhwc_ref* dz, const osa_nhwc_ref* drh, const osa_nhwc_ref* dpre_out,
                         const osa_nhwc_ref* dh_out, long long npix, int C, void* stream);
int osa_gru_gates_q_fwd(const osa_nhwc_ref* z, const osa_nhwc_ref* qpre, const float* bias_q, const osa_nhwc_ref* cq, const osa_nhwc_ref* h,
                        const osa_nhwc_ref* out, long long npix, int C, void* stream);
int osa_gru_gates_q_bwd(const osa_nhwc_ref* z, const osa_nhwc_ref* qpre, const float* bias_q, const osa_nhwc_ref* cq, const osa_nhwc_ref* h,
                        const osa_nhwc_ref* dout, const osa_nhwc_ref* dz_out, const osa_nhwc_ref* dqpre_out, const osa_nhwc_ref* dh_out,
                        long long npix, int C, void* stream);

/* Disparity update of the GRU loop (igev_stereo.py:201 `disp = disp + delta_disp`) with the copies its consumers read:
 *   disp [npix] += delta[p * delta_cs] (delta NULL: unchanged);  disp_nhwc4 [npix][4] = (disp, 0, 0, 0) (input of the motion encoder's
 *   7x7 convd1, update.py:87);  slot[p * slot_cs] = disp (the `torch.cat([out, disp])` channel of the 1/4 level, update.py:92).
 *   Optional range blocks of the two NHWC destinations receive max |disp|.  Any of disp_nhwc4 / slot may be NULL. */
int osa_disp_update_f32(float* disp, const float* delta, int delta_cs, float* disp_nhwc4, float* slot, int slot_cs,
                        long long npix, float* disp4_meta, float* slot_meta, void* stream);

/*
 * Resampling of the GRU hidden states between the 1/4, 1/8 and 1/16 levels (models/igev/update.py:99-109,
 * models/stereobase/gru_blocks.py pool2x / interp), NHWC with channel strides so that source and destination can be
 * channel slices of the per-level [h | x | r*h | z] state buffers (no torch.cat, no copies):
 *   osa_pool2x_nhwc_f32          F.avg_pool2d(x, 3, stride=2, padding=1)  (count_include_pad: the sum of the 3x3 window / 9)
 *                                -> [B, floor((H - 1) / 2) + 1, floor((W - 1) / 2) + 1, C]
 *   osa_resize_bilinear_nhwc_f32 F.interpolate(x, (Ho, Wo), mode='bilinear', align_corners=True)
 * C, xCs, yCs multiples of 4.  x_meta / y_meta (optional): f16x3 range blocks -- both results are convex combinations of
 * inputs (or of inputs and zeros), so max |x| is folded into y's block without a reduction over the data.
 */
int osa_pool2x_nhwc_f32(const float* x, float* y, int B, int H, int W, int C, int xCs, int yCs,
                        const float* x_meta, float* y_meta, void* stream);
int osa_resize_bilinear_nhwc_f32(const float* x, float* y, int B, int Hi, int Wi, int Ho, int Wo, int C, int xCs, int yCs,
                                 const float* x_meta, float* y_meta, void* stream);

/* Packing for backward passes.  Ci/Co are the roles of the convolution that will be EXECUTED with the
 * packed buffer; src_transposed=1 reads w_ref as [Ci][Co][k] (instead of [Co][Ci][k]); flip=1 mirrors taps.
 *   d(input) of a stride-1 Conv3d   : pack_ex(Ci'=Co, Co'=Ci, src_transposed=1, flip=1) + conv, pad' = dil*(k-1)-pad
 *   d(input) of a stride-2 Conv3d   : osa_deconv3d_pack_*(w, Ci'=Co, Co'=Ci) + osa_deconv3d_ndhwc_* (k=3,p=1,op=1)
 *   d(input) of a ConvTranspose3d   : pack_ex(Ci'=Co, Co'=Ci, 0, 0) + stride-2 conv */
int osa_conv3d_pack_ex(const float* w_ref, float* w_packed, int Ci, int Co,
                       int kd, int kh, int kw, int src_transposed, int flip,
                       int f16x3, float wscale, void* stream);
/* Weight gradient (fp32 matrix cores, exact): dw [Co][Ci][k] for a Conv3d, [Ci][Co][k] for a stride-2
 * ConvTranspose3d (transposed=1).  x, dy NDHWC.  dw is zero-filled, then accumulated atomically. */
int osa_conv3d_wgrad_f32(const float* x, const float* dy, float* dw,
                         int B, int Di, int Hi, int Wi, int Ci, int xCs,
                         int Do, int Ho, int Wo, int Co, int dyCs,
                         int kd, int kh, int kw, int stride,
                         int pad_d, int pad_h, int pad_w, int dil_d, int dil_h, int dil_w,
                         int transposed, void* stream);
/* Two-stage form of the same gradient: every workgroup stores its partial tiles to `workspace` (plain stores), a second kernel adds
 * them in a fixed order.  Deterministic (bit-identical from run to run) and 2-4x faster than the atomics form, whose contended float
 * atomics serialise in L2.  The workspace is caller-owned scratch of osa_conv3d_wgrad_workspace_bytes(...) bytes (same dimensions),
 * 16-byte aligned; its contents are undefined afterwards. */
size_t osa_conv3d_wgrad_workspace_bytes(int B, int Di, int Hi, int Wi, int Ci, int Do, int Ho, int Wo, int Co,
                                        int kd, int kh, int kw, int stride, int pad_d, int pad_h, int pad_w,
                                        int dil_d, int dil_h, int dil_w, int transposed);
int osa_conv3d_wgrad_ws_f32(const float* x, const float* dy, float* dw,
                            int B, int Di, int Hi, int Wi, int Ci, int xCs,
                            int Do, int Ho, int Wo, int Co, int dyCs,
                            int kd, int kh, int kw, int stride,
                            int pad_d, int pad_h, int pad_w, int dil_d, int dil_h, int dil_w,
                            int transposed, float* workspace, size_t workspace_bytes, void* stream);
/* Split-precision (f16x3) weight gradient (r3): the same gradient on v_mfma_f32_32x32x16_f16 -- 16 positions per instruction, operands
 * split into fp16 hi + lo and pre-scaled by powers of two from the tensors' range blocks (x_meta, dy_meta: osa_f16x3_ranges layout,
 * max |.| in the 8 slots), fp32 accumulation, same two-stage deterministic reduction.  Covers unit-stride, unit-dilation Conv3d / Conv2d
 * layers with kh, kw <= 3 (3x3 planes, or kd == 1); the workspace query returns 0 for anything else (use osa_conv3d_wgrad_ws_f32). */
size_t osa_conv3d_wgrad_f16x3_workspace_bytes(int B, int Di, int Hi, int Wi, int Ci, int Do, int Ho, int Wo, int Co,
                                              int kd, int kh, int kw, int stride, int pad_d, int pad_h, int pad_w,
                                              int dil_d, int dil_h, int dil_w, int transposed);
int osa_conv3d_wgrad_ws_f16x3(const float* x, const float* dy, float* dw,
                              int B, int Di, int Hi, int Wi, int Ci, int xCs,
                              int Do, int Ho, int Wo, int Co, int dyCs,
                              int kd, int kh, int kw, int stride,
                              int pad_d, int pad_h, int pad_w, int dil_d, int dil_h, int dil_w,
                              int transposed, const float* x_meta, const float* dy_meta,
                              float* workspace, size_t workspace_bytes, void* stream);
/* Native f16 weight gradient (r5, ABI v5): the arithmetic of the reference's AMP training (stereo/modeling/trainer_template.py:211-226,
 * autocast + GradScaler: the weight gradient of an autocast convolution multiplies fp16 activations by fp16 output gradients and
 * accumulates in fp32) -- operands rounded to fp16 (nearest even) when they are staged, ONE v_mfma_f32_32x32x16_f16 per product, no lo
 * planes.  Same layers, workspace (osa_conv3d_wgrad_f16x3_workspace_bytes) and deterministic two-stage reduction as the f16x3 form.
 * x_meta / dy_meta: NULL (no operand scaling, exactly like autocast: GradScaler owns the range) or range blocks (power-of-two scaling).
 * x_f16 / dy_f16 = 1: that tensor holds fp16 elements (NDHWC, channel stride in elements, 8-byte aligned) -- the activations an AMP step
 * saves and the gradients it propagates are fp16 tensors; dw stays fp32. */
int osa_conv3d_wgrad_ws_f16(const void* x, const void* dy, float* dw,
                            int B, int Di, int Hi, int Wi, int Ci, int xCs,
                            int Do, int Ho, int Wo, int Co, int dyCs,
                            int kd, int kh, int kw, int stride,
                            int pad_d, int pad_h, int pad_w, int dil_d, int dil_h, int dil_w,
                            int transposed, const float* x_meta, const float* dy_meta, int x_f16, int dy_f16,
                            float* workspace, size_t workspace_bytes, void* stream);

/* The same weight gradients over a LIST of equally shaped (x, dy) tensor pairs in one launch, without concatenating them (r6): the uses
 * of ONE weight within a training step -- the update block applies each of its convolutions once per GRU iteration
 * (stereo/modeling/models/igev/update.py:97-150, 22 iterations in training: igev_stereo.py:181-203), and autograd would accumulate 22
 * separate weight gradients.  Batch entry b of the launch is entry b % (B / n_items) of item b / (B / n_items); xs / dys are HOST arrays of
 * n_items (<= 24) device pointers.  form: 0 = osa_conv3d_wgrad_ws_f32, 1 = ..._f16x3 (range blocks over ALL items required), 2 = ..._f16
 * (x_f16 / dy_f16 as there).  B = total batch over all items; workspace: the form's query for a single tensor of batch B. */
int osa_conv3d_wgrad_ws_multi(int form, const void* const* xs, const void* const* dys, int n_items, float* dw,
                              int B, int Di, int Hi, int Wi, int Ci, int xCs,
                              int Do, int Ho, int Wo, int Co, int dyCs,
                              int kd, int kh, int kw, int stride,
                              int pad_d, int pad_h, int pad_w, int dil_d, int dil_h, int dil_w,
                              int transposed, const float* x_meta, const float* dy_meta, int x_f16, int dy_f16,
                              float* workspace, size_t workspace_bytes, void* stream);


/* small-Cout 'same' convolution (Co <= 4, e.g. the 32->1 classifier heads). Reference weight
 * layout [Co][Ci][kd][kh][kw] is consumed directly (device pointer).
 * y = conv(x) + bias[co] + residual ; residual (or NULL) has y's layout (voxel stride yCs). */
int osa_conv3d_small_co_ndhwc_f32(const float* x, const float* w_ref, const float* bias,
                                  const float* residual, float* y,
                                  int B, int D, int H, int W, int Ci, int xCs, int Co, int yCs,
                                  int kd, int kh, int kw, int pad_d, int pad_h, int pad_w,
                                  void* stream);

/* Same convolution with the weights packed once ([cin-chunk][tap][Co][16], 64-byte aligned) by
 * osa_conv3d_small_co_pack_f32: the workgroups then read them through the scalar cache instead of
 * re-ordering the reference layout into LDS. */
size_t osa_conv3d_small_co_packed_floats(int Ci, int Co, int kd, int kh, int kw);
int    osa_conv3d_small_co_pack_f32(const float* w_ref, float* w_packed, int Ci, int Co,
                                    int kd, int kh, int kw, void* stream);
int    osa_conv3d_small_co_packed_ndhwc_f32(const float* x, const float* w_packed, const float* bias,
                                            const float* residual, float* y,
                                            int B, int D, int H, int W, int Ci, int xCs, int Co, int yCs,
                                            int kd, int kh, int kw, int pad_d, int pad_h, int pad_w,
                                            void* stream);

/* ---- disparity regression (a10-a12) ------------------------- */
/* out[b,h,w] = sum_d d * prob[b,d,h,w] */
int osa_softargmin_f32(const float* prob, float* out, int B, int D, int H, int W, void* stream);
/* out[b,h,w] = sum_d d * softmax_d(cost[b,:,h,w]) ; optionally also writes prob (may be NULL) */
int osa_softmax_softargmin_f32(const float* cost, float* prob, float* out,
                               int B, int D, int H, int W, void* stream);
/* fused trilinear upsample [B,Dl,Hl,Wl] -> [B,D,H,W] + softmax over D + expectation */
int osa_upsample_softargmin_f32(const float* cost_lowres, float* out,
                                int B, int Dl, int Hl, int Wl, int D, int H, int W,
                                int align_corners, void* stream);

/* ---- backward of the memory-bound ops (training) ------- */
/* d(build_gwc_volume) (concat=0: left/right = forward features [B,C,H,W]) or d(build_concat_volume)
 * (concat=1: C = channels per side).  dvol is NCDHW with vol_channels channels; this op's channels
 * start at c_off.  Writes dleft/dright [B,C,H,W]. */
int osa_build_volume_bwd_f32(const float* dvol, const float* left, const float* right,
                             float* dleft, float* dright,
                             int B, int C, int H, int W, int maxdisp, int num_groups,
                             int concat, int mask_left_concat, int vol_channels, int c_off,
                             void* stream);
/* dprob[b,d,h,w] = d * dout[b,h,w] */
int osa_softargmin_bwd_f32(const float* dout, float* dprob, int B, int D, int H, int W, void* stream);
/* dcost = softmax(cost) * (d - disp) * dout */
int osa_softmax_softargmin_bwd_f32(const float* cost, const float* dout, float* dcost,
                                   int B, int D, int H, int W, void* stream);
/* backward of osa_upsample_softargmin_f32 w.r.t. the low-res cost (zero-fills, then atomically accumulates) */
int osa_upsample_softargmin_bwd_f32(const float* cost_lowres, const float* dout, float* dcost_lowres,
                                    int B, int Dl, int Hl, int Wl, int D, int H, int W,
                                    int align_corners, void* stream);
/* The same gradient without atomics (r3): pass 1 folds every output pixel's D gradients into Dl values of a scratch tensor
 * [B,Dl,H,W] (workspace, 16-byte aligned, osa_upsample_softargmin_bwd_workspace_bytes), pass 2 gathers each low-res cell's bilinear
 * footprint in a fixed order -- deterministic, no zero-fill; what autograd of the reference's F.interpolate + softmax + regression
 * (gwcnet_disp_processor.py:98-108) computes.  4x faster than the atomic form on a 256x512 training crop. */
size_t osa_upsample_softargmin_bwd_workspace_bytes(int B, int Dl, int H, int W);
int osa_upsample_softargmin_bwd_ws_f32(const float* cost_lowres, const float* dout, float* dcost_lowres,
                                       int B, int Dl, int Hl, int Wl, int D, int H, int W,
                                       int align_corners, void* workspace, size_t workspace_bytes, void* stream);

/* max |x| over n contiguous floats (16-byte aligned) folded into the range block `meta` (osa_f16x3_ranges layout: 8 slots, atomic max) --
 * the device-side operand range of a tensor that reaches an f16x3 layer from outside the engine (what torch.linalg.vector_norm(x, inf)
 * computed before r3; no reference counterpart: the reference has no split-precision arithmetic). */
int osa_amax_f32(const float* x, long long n, float* meta, void* stream);

/* ---- disparity refinement ----------------------------- */
/* convex 3x3 up-sampling: out[b,y,x] = sum_k W[b,k,y,x] * (gain*disp_low)[b, y/scale + k/3-1, x/scale + k%3-1]
 * disp_low [B,1,h,w], weights [B,9,h*scale,w*scale], out [B,h*scale,w*scale].
 * softmax_weights=1: `weights` are logits and softmax over the 9 taps is fused.
 * (disp_refinement/disp_refinement.py:194-204, stereobase/igev_blocks.py:51-63, igev/submodule.py:253-265) */
int osa_context_upsample_f32(const float* disp_low, const float* weights, float* out,
                             int B, int h, int w, int scale, int softmax_weights, float gain,
                             void* stream);
/* The training form (r6): weights = softmax of `logits` [B,9,H,W] given with their element strides {batch, tap, row, column} (the
 * transposed conv's channels-last output is read in place), fp32 (logits_f16 0) or fp16 (1); out = the map above with the softmax and the
 * gain fused.  The reference does this once per GRU iteration for its sequence loss (stereobase_gru.py:196-203, igev_stereo.py:198-207:
 * F.softmax + context_upsample on 9 x H x W tensors, forward and autograd backward).
 * _bwd: ddisp_low [B,1,h,w] and dlogits (element type of `logits`, its own element strides) from dout [B,H,W]; scratch: 9 * B * h * w floats
 * (per-cell tap sums).  scale: 1, 2, 4 or 8.  Deterministic (lane-shuffle sums and a gather in a fixed order, no atomics). */
int osa_context_upsample_logits_f32(const float* disp_low, const void* logits, int logits_f16, const long long* logits_strides,
                                    float* out, int B, int h, int w, int scale, float gain, void* stream);
int osa_context_upsample_logits_bwd_f32(const float* disp_low, const void* logits, int logits_f16, const long long* logits_strides,
                                        const float* dout, float* ddisp_low, void* dlogits, const long long* dlogits_strides, float* scratch,
                                        int B, int h, int w, int scale, float gain, void* stream);

/* ---- geometry-encoding volume of the GRU loop ---------- */
/* models/stereobase/gru_blocks.py:170-229, models/igev/geometry.py:7-66 */
/* corr[b,h,w1,w2] = sum_c fmap1[b,c,h,w1] * fmap2[b,c,h,w2]; fmaps NCHW, corr [B,H,W1,W2] */
int osa_allpairs_corr_f32(const float* fmap1, const float* fmap2, float* corr,
                          int B, int C, int H, int W1, int W2, void* stream);
/* NDHWC geometry volume [B,D,H,W,Cs] (first C channels) -> per-pixel rows [B,H,W,C,D] */
int osa_geo_rows_f32(const float* vol_ndhwc, float* rows, int B, int D, int H, int W, int C, int Cs, void* stream);
/* y[r, j] = (x[r,2j] + x[r,2j+1]) / 2, j < n/2   (F.avg_pool2d(.,[1,2],stride=[1,2]) along the row axis) */
int osa_avgpool_rows_f32(const float* x, float* y, long long rows, int n, void* stream);
/* One GRU-iteration lookup: out [B,(C+1)*(2r+1)*levels,H,W]; geo_levels[l] rows [B,H,W,C,geo_len[l]],
 * corr_levels[l] rows [B,H,W,corr_len[l]] (HOST arrays of device pointers / lengths); disp, coords_x [B,H,W]. */
int osa_geo_lookup_f32(const float* const* geo_levels, const float* const* corr_levels,
                       const int* geo_len, const int* corr_len, int levels,
                       const float* disp, const float* coords_x, float* out,
                       int B, int H, int W, int C, int radius, void* stream);
/* The same lookup, channels-last: out [B,H,W,out_cs], channel index as above, channels beyond (C+1)*(2r+1)*levels zero-filled --
 * the layout the update block's motion encoder (1x1 convc1, update.py:85) reads, so the GRU loop needs no transpose per iteration. */
int osa_geo_lookup_nhwc_f32(const float* const* geo_levels, const float* const* corr_levels,
                            const int* geo_len, const int* corr_len, int levels,
                            const float* disp, const float* coords_x, float* out, int out_cs,
                            int B, int H, int W, int C, int radius, void* stream);
/* Gradient of osa_geo_lookup_f32 with respect to the pyramid levels (the disparity is detached in the reference's loop,
 * models/igev/igev_stereo.py:190): dgeo_levels / dcorr_levels have the shapes of geo_levels / corr_levels, are zero-filled and then
 * accumulated without atomics (every pixel owns its rows).  dout: [B,(C+1)*(2r+1)*levels,H,W]. */
int osa_geo_lookup_bwd_f32(float* const* dgeo_levels, float* const* dcorr_levels,
                           const int* geo_len, const int* corr_len, int levels,
                           const float* disp, const float* coords_x, const float* dout,
                           int B, int H, int W, int C, int radius, void* stream);
/* The same gradient ACCUMULATED into dgeo_levels / dcorr_levels (r6): the caller zero-fills them once per training step and calls this for
 * every lookup of the step (one per GRU iteration, igev_stereo.py:181-203); only the (C+1) x (2r+2) entries per (pixel, level) a lookup's
 * taps reach are touched (read-modify-write, no atomics: a pixel owns its rows) -- instead of 22 dense gradients that autograd adds up. */
int osa_geo_lookup_bwd_acc_f32(float* const* dgeo_levels, float* const* dcorr_levels,
                               const int* geo_len, const int* corr_len, int levels,
                               const float* disp, const float* coords_x, const float* dout,
                               int B, int H, int W, int C, int radius, void* stream);

/* ---- input pre-processing on device ------------------------- */
/* RightTopPad(edge) + HWC->CHW + /255 + (x-mean)/std for the left and right image in one launch
 * (stereo_trans.py:243-267, :22-29, :48-56).  left/right: device HWC images [H][W][3], uint8 (is_u8=1)
 * or float32; mean3/std3: HOST pointers to 3 floats.  out: layout 0 = NCHW [2,3,Hp,Wp] (left first),
 * layout 1 = NHWC4 [2,Hp,Wp,4] (4th channel zero; the engine backbone's input layout). */
int osa_preprocess_pair_f32(const void* left_hwc, const void* right_hwc, int is_u8,
                            int H, int W, int Hp, int Wp,
                            const float* mean3, const float* std3,
                            float* out, int layout, void* stream);

/* ---- f16 arithmetic mode (r4): the autocast arithmetic of the reference's AMP configs ----
 * cfgs/stereobase/stereobase_sceneflow.yaml:50, cfgs/lightstereo/lightstereo_s_sceneflow.yaml:36, cfgs/igev/igev_sceneflow_amp.yaml:40 set
 * AMP: true and stereo/trainer/trainer_template.py:211,281 wraps every forward in torch.autocast: convolutions then multiply fp16 operands
 * and accumulate in fp32.  These entry points do the same on the MFMA engine: operands rounded to fp16 (nearest even), ONE
 * v_mfma_f32_32x32x16_f16 per product (the f16x3 mode spends three), fp32 accumulation, fp32 BN / bias / activation epilogue.
 * Tensors are fp32 NDHWC, or fp16 NDHWC where the flags below (OR'ed into `act`) say so; channel counts and strides are in ELEMENTS of
 * each tensor, fp16 tensors need them % 8 == 0.  An fp16 output takes an fp16 residual.  No operand scaling and no range blocks: values
 * beyond 65504 become inf exactly as under autocast.  Same argument lists as the *_f32 calls otherwise; weights packed with *_pack_f16
 * (buffer sizes: the *_packed_floats functions, an upper bound here). */
enum { OSA_IN_F16 = 32, OSA_OUT_F16 = 64, OSA_RES_F16 = 128 };
int osa_conv3d_pack_f16(const float* w_ref, float* w_packed, int Ci, int Co, int kd, int kh, int kw, void* stream);
int osa_deconv3d_pack_f16(const float* w_ref, float* w_packed, int Ci, int Co, int k, int pad, void* stream);
int osa_deconv2d_pack_f16(const float* w_ref, float* w_packed, int Ci, int Co, int k, int pad, void* stream);
int osa_conv3d_ndhwc_f16(const void* x, const float* w_packed, const float* scale, const float* shift, const void* residual, void* y,
                         int B, int Di, int Hi, int Wi, int Ci, int xCs, int Co, int yCs, int rCs, int kd, int kh, int kw, int stride,
                         int pad_d, int pad_h, int pad_w, int dil_d, int dil_h, int dil_w, const float* gate_logits, int gCs,
                         int act, float slope, void* stream);
int osa_deconv3d_ndhwc_f16(const void* x, const float* w_packed, const float* scale, const float* shift, const void* residual, void* y,
                           int B, int Di, int Hi, int Wi, int Ci, int xCs, int Co, int yCs, int rCs, int k, int pad, int opad,
                           const float* gate_logits, int gCs, int act, float slope, void* stream);
int osa_deconv2d_nhwc_f16(const void* x, const float* w_packed, const float* scale, const float* shift, const void* residual, void* y,
                          int B, int Hi, int Wi, int Ci, int xCs, int Co, int yCs, int rCs, int k, int pad, int opad,
                          const float* gate_logits, int gCs, int act, float slope, void* stream);

/* Per-channel sums over the P positions of a channels-last tensor (training path, r6): out[0..C) = sum_p dy[p][c] and, with x != NULL,
 * out[C..2C) = sum_p dy[p][c] * (x[p][c] - x_shift[c]) (x_shift may be NULL = 0); with dx != NULL the same pass also writes
 * dx[p][c] = dy[p][c] * dx_scale[c] in dy's element type.  One coalesced read, deterministic two-stage sum.  Replaces, on the
 * reference's training path, the bias-gradient reductions autograd runs for every nn.Conv2d(bias=True) of the update block
 * (stereo/modeling/models/igev/update.py:19-26,38-40; stereobase/gru_blocks.py:233-328) and the backward of BatchNorm layers in eval
 * mode (FREEZE_BN: stereo/trainer/trainer_template.py:83-85 -- dbeta = out[0..C), dgamma = invstd * out[C..2C) with x_shift =
 * running_mean, dx = dy * gamma * invstd).  dy / x: fp32 (f16 flag 0) or fp16 (1) elements, channel strides in elements, % 4 == 0, base
 * pointers 16-byte (fp16: 8-byte) aligned; C <= 1024.  workspace: osa_channel_sums_workspace_bytes(P, C) bytes (0 = unsupported). */
size_t osa_channel_sums_workspace_bytes(long long P, int C);
int osa_channel_sums(const void* dy, int dy_f16, int dy_cs, const void* x, int x_f16, int x_cs, const float* x_shift,
                     const float* dx_scale, void* dx, int dx_cs, long long P, int C,
                     float* out, float* workspace, size_t workspace_bytes, void* stream);
/* ... sums of dy over a LIST of n_items (<= 24) equally shaped tensors of P positions each, in one launch: the queued output gradients of
 * one biased convolution applied once per GRU iteration (HOST array of device pointers); workspace: n_items x the query above. */
int osa_channel_sums_multi(const void* const* dys, int n_items, int dy_f16, int dy_cs, long long P, int C,
                           float* out, float* workspace, size_t workspace_bytes, void* stream);
/* out[p][c] = u[p][c] * a[c] + (v != NULL ? v[p][c] * b[c] : 0) + c0[c], optionally ReLU, on channels-last rows; out has u's element type.
 * With osa_channel_sums this is a BatchNorm in TRAINING mode on the engine's layouts (r6; GwcNet / PSMNet train with batch statistics:
 * stereo/modeling/models/gwcnet/gwcnet_disp_processor.py:8-19, cfgs/gwcnet/gwcnet_sceneflow.yaml): forward = sums of x and x (x - pivot)
 * -> mean / invstd -> y = x * (gamma invstd) + (beta - mean gamma invstd); backward = sums of dy and dy (x - mean) ->
 * dx = dy * a + x * b + c0 with a = gamma invstd, b = -gamma invstd^3 S2 / N, c0 = -a S1 / N - b mean. */
int osa_channel_affine(const void* u, int u_f16, int u_cs, const void* v, int v_f16, int v_cs,
                       const float* a, const float* b, const float* c0, void* out, int out_cs,
                       long long P, int C, int relu, void* stream);

/* ---- InstanceNorm2d (+ activation) on NHWC maps (r4, csrc/norm.hip) ----
 * The normalisation of the reference-written FPN decoders of the feature pyramids: Conv2xUp / BasicConv2d(norm_layer=nn.InstanceNorm2d)
 * (models/stereobase/backbone.py:46-53), Conv2x_IN / BasicConv_IN (models/igev/extractor.py:338-341, models/igev/submodule.py:78-108) and
 * LightStereo's out_conv (models/lightstereo/backbone.py:57-59): affine = False, per (image, channel) mean / biased variance over H x W,
 * y = act((x - mean) / sqrt(var + eps)).  x: [B][HW][xCs], y: [B][HW][yCs] (a channel slice of a concat buffer when the caller offsets the
 * pointer), C channels; act: OSA_ACT_NONE | OSA_ACT_RELU | OSA_ACT_LEAKY.  workspace: osa_instnorm_workspace_floats(B, HW, C) floats;
 * deterministic (no float atomics).  y_meta: NULL or y's range block. */
size_t osa_instnorm_workspace_floats(int B, long long HW, int C);
int osa_instnorm_nhwc_f32(const float* x, float* y, int B, long long HW, int C, int xCs, int yCs, float eps, int act, float slope,
                          float* workspace, float* y_meta, void* stream);

/* ---- d-marching form of the 3x3x3 stride-1 convolutions with 32 output channels (r4, csrc/conv_march.h) ----
 * osa_conv3d_ndhwc_f16x3 runs eligible layers (3x3x3, stride 1, padding 1, Ci % 16 == 0, Co == 32, no gate: GwcNet / PSMNet dres0,
 * dres1, classif*.0 -- gwcnet_disp_processor.py:40-81) as workgroups that own a pixel column and walk along d, each staged input plane
 * feeding three output planes.  Same arguments, same semantics; results agree with the brick form to fp32 rounding (different summation
 * order).  This counter tells how many calls of this process took that form (tests assert that the intended layers do). */
long long osa_conv3d_march_launches(void);

/* ---- d-marching form of the 3x3x3 STRIDE-2 convolutions with 64 output channels (r6, csrc/conv_march_s2.h) ----
 * osa_conv3d_ndhwc_f16x3 runs eligible layers (3x3x3, stride 2, padding 1, Ci % 16 == 0, Co == 64, split input and output, no residual /
 * gate: conv1 of the GwcNet / PSMNet hourglasses -- models/gwcnet/hourglass.py:19-24) as workgroups that own a 4 x 32 output pixel column
 * and walk along d: even input planes feed one output plane, odd ones two; every input plane is staged once, by LDS-DMA into a
 * parity-planar LDS image.  Same arguments, same semantics; results agree with the brick form to fp32 rounding.  Bit 29 of
 * osa_conv_b_ring_mask switches the form (A/B runs, parity tests); this counter tells how many calls took it. */
long long osa_conv3d_march_s2_launches(void);

/* ---- B (weight) operands through an LDS ring (r4, csrc/conv_kernel.h BL = 1; f16x3 and f16 modes) ----
 * Every convolution / transposed convolution entry point above (the MFMA tiles behind nn.Conv3d / nn.Conv2d / nn.ConvTranspose3d of
 * gwcnet/hourglass.py:19-56, gwcnet_disp_processor.py:40-81, gwcnet_backbone.py:38-91) fetches a tap step's weight fragments once per
 * workgroup by LDS-DMA instead of once per wave.  Same products, same order: results are bit-identical either way.  The mask selects the
 * tile configurations that take the ring (bit i = entry i of csrc/conv_cfgs.def, bit 30 = the fused transposed convs; default: the
 * tiles where it measured faster, csrc/conv3d.hip g_b_ring_mask; -1 = every tile that has the form, 0 = none); it exists for A/B
 * measurements and for the parity test that proves the two forms identical.  Returns the previous mask.
 * osa_conv_b_ring_launches: launches of this process that took the ring form. */
int osa_conv_b_ring_mask(int mask);
long long osa_conv_b_ring_launches(void);

/* ---- d-walking form of the fused NDHWC volume builder (r4, csrc/volume.hip build_volume_walk_kernel) ----
 * osa_build_volume_nhwc_f32 (build_gwc_volume + build_concat_volume + torch.cat, cost_volume.py:59-105, gwcnet_cost_processor.py:65) runs
 * eligible calls (NHWC features with 16-byte aligned channel quads, quad-lane channel counts, maps at least 2 x 32 pixels wide) as
 * workgroups that own a (row, 32-pixel tile) and walk along d: left features stay in registers for all disparities, the right window is a
 * ring in LDS that a loader wave refills by LDS-DMA `step` pixels at a time while the other waves compute and store.  Bit-identical output.
 * osa_volume_walk_step(step): 8 or 4 disparities per step for both output forms, 0 = the chunked kernel (fp32 output); returns the previous
 * fp32-form value (A/B runs and the parity test).  Defaults: 8 for the fp32 output, 4 for the split output (csrc/volume.hip).  osa_volume_walk_launches: calls of this process that took the walking form. */
int osa_volume_walk_step(int step);
long long osa_volume_walk_launches(void);

/* ---- the fused volume written as a split tensor (r4, f16x3 chains) ----
 * Same construction and arguments as osa_build_volume_nhwc_f32 (build_gwc_volume + build_concat_volume + torch.cat: cost_volume.py:59-105,
 * gwcnet_cost_processor.py:41-65), but the NDHWC volume is stored in the split activation format of the f16x3 convolution chain (every
 * 16-channel chunk [16 x fp16 hi | 16 x fp16 lo], the bytes of fp32; see osa_f16x3_ranges) so that the first aggregation layer
 * (dres0, gwcnet_disp_processor.py:40-47) stages it by LDS-DMA like every later layer instead of splitting fp32 values through registers.
 * The power-of-two scale of the halves must be known before the first voxel exists, so it comes from a BOUND: |gwc| <= max|f|^2 and
 * |concat| <= max|f_cat| with the maxima read from the features' range blocks gwc_meta / cat_meta (one block per feature tensor, left and
 * right images together); it is stored in vol_meta[1], the volume's running maximum in vol_meta's slots as usual.  Decoded values
 * (float(hi) + float(lo)) / scale equal the fp32 volume to 2^-22 relative (elements below 2^-18 of the bound: 2^-25 / scale absolute).
 * Only the d-walking form writes this layout: osa_build_volume_nhwc_split_eligible(...) == 1 tells whether a call qualifies (NHWC features
 * with 16-byte aligned quads, quad-lane channel counts, vol_channels / c_off / G + 2 Cc multiples of 16, map at least two 8-wave tiles wide,
 * maxdisp > osa_volume_walk_step); otherwise build the fp32 volume. */
int osa_build_volume_nhwc_split_eligible(const float* left_cat, const float* right_cat, const float* vol, int C, int num_groups,
                                         int gwc_stride, int Cc, int cat_stride, int vol_channels, int c_off, int W, int maxdisp);
int osa_build_volume_nhwc_split_f16x3(const float* left_gwc, const float* right_gwc, int C, int num_groups, int gwc_stride,
                                      const float* left_cat, const float* right_cat, int Cc, int cat_stride,
                                      float* vol, int vol_channels, int c_off,
                                      int B, int H, int W, int maxdisp, int mask_left_concat,
                                      const float* gwc_meta, const float* cat_meta, float* vol_meta, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OPENSTEREO_AMD_H */
