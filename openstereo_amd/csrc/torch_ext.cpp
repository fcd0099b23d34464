// PyTorch-ROCm C++ extension over the C ABI (include/openstereo_amd.h): the dispatch layer north_star names -- "exposed to Python through a
// PyTorch-ROCm C++/HIP extension" (SURVEY 8b last row: TORCH_LIBRARY, at::Tensor in / out, the current HIP stream, TORCH_CHECK ->
// RuntimeError).  Host code only: every op validates its tensors, allocates the result with ATen and calls ONE entry point of
// libopenstereo_amd.so on c10::hip::getCurrentHIPStream().  The C ABI stays the boundary a non-torch host binds (tests/test_abi_cpu.py);
// this file is the torch binding of the same entry points, replacing the ctypes marshalling on the hot launch path (openstereo_amd/_ext.py
// loads it; openstereo_amd/ops.py and engine.PackedConv3d route through `torch.ops.osa_native.*` when it is present).
//
// Reference interfaces (stereo/modeling/...): gwc_volume / concat_volume -- cost_volume/cost_volume.py:59-92; corr_volume -- :32-41;
// softargmin -- disp_pred/disp_regression.py:8-12; softmax_softargmin -- stereobase_gru.py:163-164; upsample_softargmin --
// models/gwcnet/gwcnet_disp_processor.py:99-133; context_upsample -- models/stereobase/igev_blocks.py:51-70; conv_ndhwc -- the
// nn.Conv3d / ConvTranspose3d + BatchNorm3d(eval) + activation units of gwcnet_disp_processor.py:8-81, hourglass.py:5-56.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>
#include <torch/csrc/autograd/custom_function.h>

#include "../../include/openstereo_amd.h"

namespace {

inline void* cur_stream() { return static_cast<void*>(c10::hip::getCurrentHIPStream().stream()); }

#define OSA_CALL(expr)                                                                     \
    do {                                                                                   \
        const int rc__ = (expr);                                                           \
        TORCH_CHECK(rc__ == 0, "openstereo_amd: ", #expr, " failed (", rc__, "): ", osa_last_error()); \
    } while (0)

inline const at::Tensor& gpu_f32(const at::Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda(), "openstereo_amd: ", name, " is on ", t.device(), " -- the gfx950 engine has no CPU path");
    TORCH_CHECK(t.scalar_type() == at::kFloat, "openstereo_amd: ", name, " must be float32");
    return t;
}
inline const float* fp(const at::Tensor& t) { return t.data_ptr<float>(); }
inline const float* fpo(const c10::optional<at::Tensor>& t) { return (t.has_value() && t->defined()) ? t->data_ptr<float>() : nullptr; }
// raw pointer of a tensor of either float dtype (the f16 mode hands fp16 tensors through the same arguments), plus an element offset
inline void* vp(const at::Tensor& t, int64_t off) { return static_cast<char*>(t.data_ptr()) + off * t.element_size(); }

// ---- volumes -------------------------------------------------------------------------------------------------------------------
at::Tensor gwc_volume(const at::Tensor& left, const at::Tensor& right, int64_t maxdisp, int64_t groups) {
    gpu_f32(left, "left"); gpu_f32(right, "right");
    TORCH_CHECK(left.dim() == 4 && left.sizes() == right.sizes(), "gwc_volume: features must be [B,C,H,W] of equal shape");
    const auto l = left.contiguous(), r = right.contiguous();
    const int64_t B = l.size(0), C = l.size(1), H = l.size(2), W = l.size(3);
    TORCH_CHECK(groups > 0 && C % groups == 0, "gwc_volume: ", C, " channels are not divisible by ", groups, " groups");   // cost_volume.py:61
    auto vol = at::empty({B, groups, maxdisp, H, W}, l.options());
    OSA_CALL(osa_build_volume_f32(fp(l), fp(r), (int)C, (int)groups, nullptr, nullptr, 0, vol.data_ptr<float>(), OSA_NCDHW, (int)groups, 0,
                                  (int)B, (int)H, (int)W, (int)maxdisp, 1, nullptr, cur_stream()));
    return vol;
}

at::Tensor concat_volume(const at::Tensor& left, const at::Tensor& right, int64_t maxdisp, bool mask_left) {
    gpu_f32(left, "left"); gpu_f32(right, "right");
    TORCH_CHECK(left.dim() == 4 && left.sizes() == right.sizes(), "concat_volume: features must be [B,C,H,W] of equal shape");
    const auto l = left.contiguous(), r = right.contiguous();
    const int64_t B = l.size(0), C = l.size(1), H = l.size(2), W = l.size(3);
    auto vol = at::empty({B, 2 * C, maxdisp, H, W}, l.options());
    OSA_CALL(osa_build_volume_f32(nullptr, nullptr, 0, 0, fp(l), fp(r), (int)C, vol.data_ptr<float>(), OSA_NCDHW, (int)(2 * C), 0,
                                  (int)B, (int)H, (int)W, (int)maxdisp, mask_left ? 1 : 0, nullptr, cur_stream()));
    return vol;
}

at::Tensor corr_volume(const at::Tensor& left, const at::Tensor& right, int64_t maxdisp) {
    gpu_f32(left, "left"); gpu_f32(right, "right");
    TORCH_CHECK(left.dim() == 4 && left.sizes() == right.sizes(), "corr_volume: features must be [B,C,H,W] of equal shape");
    const auto l = left.contiguous(), r = right.contiguous();
    auto vol = at::empty({l.size(0), maxdisp, l.size(2), l.size(3)}, l.options());
    OSA_CALL(osa_corr_volume_f32(fp(l), fp(r), vol.data_ptr<float>(), (int)l.size(0), (int)l.size(1), (int)l.size(2), (int)l.size(3), (int)maxdisp,
                                 cur_stream()));
    return vol;
}

// ---- regression heads ----------------------------------------------------------------------------------------------------------
at::Tensor softargmin(const at::Tensor& prob) {
    gpu_f32(prob, "prob");
    TORCH_CHECK(prob.dim() == 4, "softargmin: prob must be [B,D,H,W]");
    const auto p = prob.contiguous();
    auto out = at::empty({p.size(0), p.size(2), p.size(3)}, p.options());
    OSA_CALL(osa_softargmin_f32(fp(p), out.data_ptr<float>(), (int)p.size(0), (int)p.size(1), (int)p.size(2), (int)p.size(3), cur_stream()));
    return out;
}

std::tuple<at::Tensor, at::Tensor> softmax_softargmin(const at::Tensor& cost, bool return_prob) {
    gpu_f32(cost, "cost");
    TORCH_CHECK(cost.dim() == 4, "softmax_softargmin: cost must be [B,D,H,W]");
    const auto c = cost.contiguous();
    auto out = at::empty({c.size(0), c.size(2), c.size(3)}, c.options());
    at::Tensor prob = return_prob ? at::empty_like(c) : at::empty({0}, c.options());
    OSA_CALL(osa_softmax_softargmin_f32(fp(c), return_prob ? prob.data_ptr<float>() : nullptr, out.data_ptr<float>(), (int)c.size(0), (int)c.size(1),
                                        (int)c.size(2), (int)c.size(3), cur_stream()));
    return {out, prob};
}

at::Tensor upsample_softargmin(const at::Tensor& cost_lowres, int64_t maxdisp, int64_t h, int64_t w, bool align_corners) {
    gpu_f32(cost_lowres, "cost_lowres");
    TORCH_CHECK(cost_lowres.dim() == 4, "upsample_softargmin: cost must be [B,Dl,Hl,Wl]");
    const auto c = cost_lowres.contiguous();
    auto out = at::empty({c.size(0), h, w}, c.options());
    OSA_CALL(osa_upsample_softargmin_f32(fp(c), out.data_ptr<float>(), (int)c.size(0), (int)c.size(1), (int)c.size(2), (int)c.size(3), (int)maxdisp,
                                         (int)h, (int)w, align_corners ? 1 : 0, cur_stream()));
    return out;
}

at::Tensor context_upsample(const at::Tensor& disp_low, const at::Tensor& weights, int64_t scale, bool softmax_weights, double gain) {
    gpu_f32(disp_low, "disp_low"); gpu_f32(weights, "up_weights");
    TORCH_CHECK(disp_low.dim() == 4 && disp_low.size(1) == 1 && weights.dim() == 4 && weights.size(1) == 9, "context_upsample: disp [B,1,h,w], weights [B,9,s*h,s*w]");
    const auto d = disp_low.contiguous(), wt = weights.contiguous();
    const int64_t B = d.size(0), h = d.size(2), w = d.size(3);
    TORCH_CHECK(wt.size(2) == scale * h && wt.size(3) == scale * w, "context_upsample: weights must be at ", scale, "x the disparity's resolution");
    auto out = at::empty({B, scale * h, scale * w}, d.options());
    OSA_CALL(osa_context_upsample_f32(fp(d), fp(wt), out.data_ptr<float>(), (int)B, (int)h, (int)w, (int)scale, softmax_weights ? 1 : 0, (float)gain,
                                      cur_stream()));
    return out;
}

// ---- the convolution launch of engine.PackedConv3d (all four families, all three arithmetic modes) -----------------------------------
// dims: [B, D, H, W, Ci, xCs, Co, yCs, rCs, gCs]; geom: conv -> [kd, kh, kw, stride, pad_d, pad_h, pad_w, dil_d, dil_h, dil_w],
// transposed conv -> [k, pad, opad].  family: 0 conv3d, 1 deconv3d, 2 deconv2d (flat).  prec: 0 f32, 1 f16x3, 2 f16.  metas: the f16x3
// range blocks [x, residual, redir, y, bound_coef, redir_bound_coef, weight_scale] (undefined = NULL).  Element offsets select channel
// slices.  Writes `out` in place and returns nothing: allocation policy stays with the Python layer classes.
void conv_ndhwc(const at::Tensor& x, int64_t x_off, const at::Tensor& packed, const c10::optional<at::Tensor>& scale, const c10::optional<at::Tensor>& shift,
                const c10::optional<at::Tensor>& residual, int64_t res_off, at::Tensor out, int64_t out_off, const c10::optional<at::Tensor>& gate,
                at::IntArrayRef dims, at::IntArrayRef geom, int64_t family, int64_t prec, int64_t act, double slope, double out_scale,
                at::TensorList metas) {
    TORCH_CHECK(x.is_cuda() && out.is_cuda() && packed.is_cuda(), "conv_ndhwc: the gfx950 engine has no CPU path");
    TORCH_CHECK(dims.size() == 10, "conv_ndhwc: dims = [B, D, H, W, Ci, xCs, Co, yCs, rCs, gCs]");
    const int B = (int)dims[0], D = (int)dims[1], H = (int)dims[2], W = (int)dims[3], Ci = (int)dims[4], xCs = (int)dims[5], Co = (int)dims[6],
              yCs = (int)dims[7], rCs = (int)dims[8], gCs = (int)dims[9];
    const void* xp = vp(x, x_off);
    void* yp = vp(out, out_off);
    const void* rp = (residual.has_value() && residual->defined()) ? vp(*residual, res_off) : nullptr;
    const float* gp = fpo(gate);
    osa_f16x3_ranges rng{};
    const osa_f16x3_ranges* rngp = nullptr;
    if (prec == 1 && metas.size() == 7) {
        auto mp = [&](size_t i) -> float* { return metas[i].defined() && metas[i].numel() ? metas[i].data_ptr<float>() : nullptr; };
        rng.x_meta = mp(0); rng.residual_meta = mp(1); rng.redir_meta = mp(2); rng.y_meta = mp(3); rng.bound_coef = mp(4); rng.redir_bound_coef = mp(5);
        rng.weight_scale = mp(6);
        rngp = &rng;
    }
    void* st = cur_stream();
    const float* w = packed.data_ptr<float>();
    if (family == 0) {
        TORCH_CHECK(geom.size() == 10, "conv_ndhwc: conv geometry = [kd, kh, kw, stride, pad_d, pad_h, pad_w, dil_d, dil_h, dil_w]");
        const int g[10] = {(int)geom[0], (int)geom[1], (int)geom[2], (int)geom[3], (int)geom[4], (int)geom[5], (int)geom[6], (int)geom[7], (int)geom[8], (int)geom[9]};
        if (prec == 0)
            OSA_CALL(osa_conv3d_ndhwc_f32((const float*)xp, w, fpo(scale), fpo(shift), (const float*)rp, (float*)yp, B, D, H, W, Ci, xCs, Co, yCs, rCs,
                                          g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[9], gp, gCs, (int)act, (float)slope, st));
        else if (prec == 1)
            OSA_CALL(osa_conv3d_ndhwc_f16x3((const float*)xp, w, fpo(scale), fpo(shift), (const float*)rp, (float*)yp, B, D, H, W, Ci, xCs, Co, yCs, rCs,
                                            g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[9], gp, gCs, (int)act, (float)slope, (float)out_scale, rngp, st));
        else
            OSA_CALL(osa_conv3d_ndhwc_f16(xp, w, fpo(scale), fpo(shift), rp, yp, B, D, H, W, Ci, xCs, Co, yCs, rCs,
                                          g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[9], gp, gCs, (int)act, (float)slope, st));
        return;
    }
    TORCH_CHECK(geom.size() == 3, "conv_ndhwc: transposed-conv geometry = [k, pad, opad]");
    const int k = (int)geom[0], pad = (int)geom[1], opad = (int)geom[2];
    if (family == 1) {
        if (prec == 0)
            OSA_CALL(osa_deconv3d_ndhwc_f32((const float*)xp, w, fpo(scale), fpo(shift), (const float*)rp, (float*)yp, B, D, H, W, Ci, xCs, Co, yCs, rCs, k, pad, opad,
                                            gp, gCs, (int)act, (float)slope, st));
        else if (prec == 1)
            OSA_CALL(osa_deconv3d_ndhwc_f16x3((const float*)xp, w, fpo(scale), fpo(shift), (const float*)rp, (float*)yp, B, D, H, W, Ci, xCs, Co, yCs, rCs, k, pad, opad,
                                              gp, gCs, (int)act, (float)slope, (float)out_scale, rngp, st));
        else
            OSA_CALL(osa_deconv3d_ndhwc_f16(xp, w, fpo(scale), fpo(shift), rp, yp, B, D, H, W, Ci, xCs, Co, yCs, rCs, k, pad, opad, gp, gCs, (int)act, (float)slope, st));
        return;
    }
    TORCH_CHECK(family == 2 && D == 1, "conv_ndhwc: family 2 is the 2-D transposed conv (D == 1)");
    if (prec == 0)
        OSA_CALL(osa_deconv2d_nhwc_f32((const float*)xp, w, fpo(scale), fpo(shift), (const float*)rp, (float*)yp, B, H, W, Ci, xCs, Co, yCs, rCs, k, pad, opad,
                                       gp, gCs, (int)act, (float)slope, st));
    else if (prec == 1)
        OSA_CALL(osa_deconv2d_nhwc_f16x3((const float*)xp, w, fpo(scale), fpo(shift), (const float*)rp, (float*)yp, B, H, W, Ci, xCs, Co, yCs, rCs, k, pad, opad,
                                         gp, gCs, (int)act, (float)slope, (float)out_scale, rngp, st));
    else
        OSA_CALL(osa_deconv2d_nhwc_f16(xp, w, fpo(scale), fpo(shift), rp, yp, B, H, W, Ci, xCs, Co, yCs, rCs, k, pad, opad, gp, gCs, (int)act, (float)slope, st));
}

// ---- backward kernels of the memory-bound ops (r5: SURVEY 8b last row "*_fwd / *_bwd") --------------------------------------------------
// d(gwc_volume) / d(corr_volume: groups = 1, dvol [B,1,D,H,W]) need the forward features; d(concat_volume) needs only their shape.
std::tuple<at::Tensor, at::Tensor> volume_bwd(const at::Tensor& dvol, const c10::optional<at::Tensor>& left, const c10::optional<at::Tensor>& right,
                                              at::IntArrayRef shape, int64_t maxdisp, int64_t groups, bool concat, bool mask_left) {
    gpu_f32(dvol, "dvol");
    TORCH_CHECK(shape.size() == 4, "volume_bwd: shape = [B, C, H, W] of one feature map");
    const int64_t B = shape[0], C = shape[1], H = shape[2], W = shape[3];
    const auto dv = dvol.contiguous();
    auto dl = at::empty({B, C, H, W}, dv.options()), dr = at::empty({B, C, H, W}, dv.options());
    if (concat) {
        TORCH_CHECK(dv.dim() == 5 && dv.size(1) == 2 * C && dv.size(2) == maxdisp, "volume_bwd: dvol must be [B, 2C, maxdisp, H, W]");
        OSA_CALL(osa_build_volume_bwd_f32(fp(dv), nullptr, nullptr, dl.data_ptr<float>(), dr.data_ptr<float>(), (int)B, (int)C, (int)H, (int)W, (int)maxdisp, 0, 1,
                                          mask_left ? 1 : 0, (int)(2 * C), 0, cur_stream()));
    } else {
        TORCH_CHECK(left.has_value() && right.has_value(), "volume_bwd: the group-wise correlation gradient needs the forward features");
        TORCH_CHECK(dv.dim() == 5 && dv.size(1) == groups && dv.size(2) == maxdisp, "volume_bwd: dvol must be [B, groups, maxdisp, H, W]");
        const auto l = gpu_f32(*left, "left").contiguous(), r = gpu_f32(*right, "right").contiguous();
        OSA_CALL(osa_build_volume_bwd_f32(fp(dv), fp(l), fp(r), dl.data_ptr<float>(), dr.data_ptr<float>(), (int)B, (int)C, (int)H, (int)W, (int)maxdisp, (int)groups, 0, 1,
                                          (int)groups, 0, cur_stream()));
    }
    return {dl, dr};
}

at::Tensor softargmin_bwd(const at::Tensor& dout, int64_t D) {
    gpu_f32(dout, "dout");
    TORCH_CHECK(dout.dim() == 3, "softargmin_bwd: dout must be [B,H,W]");
    const auto g = dout.contiguous();
    auto dp = at::empty({g.size(0), D, g.size(1), g.size(2)}, g.options());
    OSA_CALL(osa_softargmin_bwd_f32(fp(g), dp.data_ptr<float>(), (int)g.size(0), (int)D, (int)g.size(1), (int)g.size(2), cur_stream()));
    return dp;
}

at::Tensor softmax_softargmin_bwd(const at::Tensor& cost, const at::Tensor& dout) {
    gpu_f32(cost, "cost"); gpu_f32(dout, "dout");
    TORCH_CHECK(cost.dim() == 4 && dout.dim() == 3, "softmax_softargmin_bwd: cost [B,D,H,W], dout [B,H,W]");
    const auto c = cost.contiguous(), g = dout.contiguous();
    auto dc = at::empty_like(c);
    OSA_CALL(osa_softmax_softargmin_bwd_f32(fp(c), fp(g), dc.data_ptr<float>(), (int)c.size(0), (int)c.size(1), (int)c.size(2), (int)c.size(3), cur_stream()));
    return dc;
}

// deterministic two-pass form (fold per output pixel, fixed-order gather per low-res cell); the scratch tensor comes from the caching allocator
at::Tensor upsample_softargmin_bwd(const at::Tensor& cost_lowres, const at::Tensor& dout, int64_t maxdisp, int64_t h, int64_t w, bool align_corners) {
    gpu_f32(cost_lowres, "cost_lowres"); gpu_f32(dout, "dout");
    TORCH_CHECK(cost_lowres.dim() == 4 && dout.dim() == 3, "upsample_softargmin_bwd: cost [B,Dl,Hl,Wl], dout [B,h,w]");
    const auto c = cost_lowres.contiguous(), g = dout.contiguous();
    auto dc = at::empty_like(c);
    const size_t need = osa_upsample_softargmin_bwd_workspace_bytes((int)c.size(0), (int)c.size(1), (int)h, (int)w);
    auto ws = at::empty({(int64_t)((need + 3) / 4)}, c.options());
    OSA_CALL(osa_upsample_softargmin_bwd_ws_f32(fp(c), fp(g), dc.data_ptr<float>(), (int)c.size(0), (int)c.size(1), (int)c.size(2), (int)c.size(3), (int)maxdisp,
                                                (int)h, (int)w, align_corners ? 1 : 0, ws.data_ptr<float>(), need, cur_stream()));
    return dc;
}

// ---- the fused NDHWC cost-volume builder the engine models run (ops.build_cost_volume_from_cl) -----------------------------------------
// gwc_feat / cat_feat: NHWC feature maps of the 2B images (left images first), logical [2B, Cs, 1, H, W] with channels_last_3d strides;
// gwc_channels / cat_channels: how many of their channels enter the volume (-1: all of them from gwc_off on / all).
// Returns the [B, G + 2 Cc, maxdisp, H, W] NDHWC volume; out_split asks for the f16x3 chain's split format (taken when the call is
// eligible and both range blocks are there: the second result says which format was written).  out_meta: the volume's range block.
std::tuple<at::Tensor, bool> cost_volume_cl(const at::Tensor& gwc_feat, const c10::optional<at::Tensor>& cat_feat, int64_t B, int64_t num_groups, int64_t maxdisp,
                                            int64_t gwc_channels, int64_t cat_channels, int64_t gwc_off, bool mask_left, bool out_split,
                                            const c10::optional<at::Tensor>& gwc_meta, const c10::optional<at::Tensor>& cat_meta, const at::Tensor& out_meta) {
    gpu_f32(gwc_feat, "gwc_feat"); gpu_f32(out_meta, "out_meta");
    TORCH_CHECK(gwc_feat.dim() == 5 && gwc_feat.size(0) == 2 * B && gwc_feat.size(2) == 1 && gwc_feat.stride(1) == 1, "cost_volume_cl: gwc_feat must be NHWC [2B, Cs, 1, H, W]");
    const int64_t Gs = gwc_feat.size(1), H = gwc_feat.size(3), W = gwc_feat.size(4);
    const int64_t C = gwc_channels >= 0 ? gwc_channels : Gs - gwc_off;        // (0 = no group-wise part: the concat-only volumes of PSMNet)
    const float* lg = fp(gwc_feat) + gwc_off;
    const float* rg = lg + B * H * W * Gs;
    const float *lc = nullptr, *rc = nullptr;
    int64_t Cc = 0, cs = 0;
    if (cat_feat.has_value() && cat_feat->defined()) {
        gpu_f32(*cat_feat, "cat_feat");
        TORCH_CHECK(cat_feat->dim() == 5 && cat_feat->size(0) == 2 * B && cat_feat->size(3) == H && cat_feat->size(4) == W && cat_feat->stride(1) == 1,
                    "cost_volume_cl: cat_feat must be NHWC [2B, cs, 1, H, W] at the gwc features' resolution");
        cs = cat_feat->size(1); Cc = cat_channels >= 0 ? cat_channels : cs;
        lc = fp(*cat_feat); rc = lc + B * H * W * cs;
    }
    const int64_t nch = num_groups + 2 * Cc, VC = (nch + 3) / 4 * 4;
    auto out = at::empty({B, maxdisp, H, W, VC}, gwc_feat.options()).permute({0, 4, 1, 2, 3});        // logical NCDHW, NDHWC in memory
    if (VC != nch) out.zero_();
    const float* gm = fpo(gwc_meta); const float* cm = fpo(cat_meta);
    const bool split = out_split && VC == nch && (C == 0 || gm) && (Cc == 0 || cm) &&
        osa_build_volume_nhwc_split_eligible(lc, rc, out.data_ptr<float>(), (int)C, (int)num_groups, (int)Gs, (int)Cc, (int)cs, (int)VC, 0, (int)W, (int)maxdisp) == 1;
    if (split)
        OSA_CALL(osa_build_volume_nhwc_split_f16x3(lg, rg, (int)C, (int)num_groups, (int)Gs, lc, rc, (int)Cc, (int)cs, out.data_ptr<float>(), (int)VC, 0, (int)B, (int)H, (int)W,
                                                   (int)maxdisp, mask_left ? 1 : 0, gm, cm, out_meta.data_ptr<float>(), cur_stream()));
    else
        OSA_CALL(osa_build_volume_nhwc_f32(lg, rg, (int)C, (int)num_groups, (int)Gs, lc, rc, (int)Cc, (int)cs, out.data_ptr<float>(), (int)VC, 0, (int)B, (int)H, (int)W,
                                           (int)maxdisp, mask_left ? 1 : 0, out_meta.data_ptr<float>(), cur_stream()));
    return {out, split};
}

// ---- weight gradient of the training path (autograd._wgrad): two-stage deterministic form, workspace from the caching allocator -----------
// dims: [B, D, H, W, Ci, xCs, Do, Ho, Wo, Co, dyCs, kd, kh, kw, stride, pad_d, pad_h, pad_w, dil_d, dil_h, dil_w, transposed].  prec: 0 exact
// fp32, 1 f16x3 (x_meta / dy_meta required), 2 native f16 (metas optional).  Returns false when the split-precision forms do not cover the layer
// (the caller then asks for prec 0); dw is written in place ([Co][Ci][k] or, transposed, [Ci][Co][k]).
bool conv_wgrad(const at::Tensor& x, const at::Tensor& dy, at::Tensor dw, at::IntArrayRef dims, int64_t prec, const c10::optional<at::Tensor>& x_meta,
                const c10::optional<at::Tensor>& dy_meta) {
    gpu_f32(dw, "dw");
    const bool xh = x.scalar_type() == at::kHalf, dyh = dy.scalar_type() == at::kHalf;
    TORCH_CHECK(x.is_cuda() && dy.is_cuda() && (xh || x.scalar_type() == at::kFloat) && (dyh || dy.scalar_type() == at::kFloat), "conv_wgrad: x / dy must be CUDA fp32 or fp16 tensors");
    TORCH_CHECK(prec == 2 || (!xh && !dyh), "conv_wgrad: fp16 tensors exist in the native f16 form only (prec 2)");
    TORCH_CHECK(dims.size() == 22, "conv_wgrad: dims = [B, D, H, W, Ci, xCs, Do, Ho, Wo, Co, dyCs, kd, kh, kw, stride, pad x3, dil x3, transposed]");
    int d[22];
    for (int i = 0; i < 22; ++i) d[i] = (int)dims[i];
    const size_t need = prec == 0
        ? osa_conv3d_wgrad_workspace_bytes(d[0], d[1], d[2], d[3], d[4], d[6], d[7], d[8], d[9], d[11], d[12], d[13], d[14], d[15], d[16], d[17], d[18], d[19], d[20], d[21])
        : osa_conv3d_wgrad_f16x3_workspace_bytes(d[0], d[1], d[2], d[3], d[4], d[6], d[7], d[8], d[9], d[11], d[12], d[13], d[14], d[15], d[16], d[17], d[18], d[19], d[20], d[21]);
    if (need == 0) {
        TORCH_CHECK(prec != 0, "conv_wgrad: unsupported layer");
        return false;
    }
    auto ws = at::empty({(int64_t)((need + 3) / 4)}, dw.options());
    void* st = cur_stream();
#define OSA_WG_ARGS static_cast<const float*>(x.data_ptr()), static_cast<const float*>(dy.data_ptr()), dw.data_ptr<float>(), d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], d[10], d[11], d[12], d[13], d[14], d[15], d[16], d[17], d[18], d[19], d[20], d[21]
    if (prec == 0) OSA_CALL(osa_conv3d_wgrad_ws_f32(OSA_WG_ARGS, ws.data_ptr<float>(), need, st));
    else if (prec == 1) OSA_CALL(osa_conv3d_wgrad_ws_f16x3(OSA_WG_ARGS, fpo(x_meta), fpo(dy_meta), ws.data_ptr<float>(), need, st));
    else OSA_CALL(osa_conv3d_wgrad_ws_f16(OSA_WG_ARGS, fpo(x_meta), fpo(dy_meta), xh ? 1 : 0, dyh ? 1 : 0, ws.data_ptr<float>(), need, st));
#undef OSA_WG_ARGS
    return true;
}

// the same over a LIST of equally shaped (x, dy) pairs in one launch (osa_conv3d_wgrad_ws_multi): dims[0] = the TOTAL batch over all items
bool conv_wgrad_multi(at::TensorList xs, at::TensorList dys, at::Tensor dw, at::IntArrayRef dims, int64_t prec, const c10::optional<at::Tensor>& x_meta,
                      const c10::optional<at::Tensor>& dy_meta) {
    gpu_f32(dw, "dw");
    TORCH_CHECK(xs.size() >= 1 && xs.size() <= 24 && xs.size() == dys.size(), "conv_wgrad_multi: 1..24 (x, dy) pairs");
    const bool xh = xs[0].scalar_type() == at::kHalf, dyh = dys[0].scalar_type() == at::kHalf;
    std::vector<const void*> xp, dp;
    for (size_t i = 0; i < xs.size(); ++i) {
        TORCH_CHECK(xs[i].is_cuda() && dys[i].is_cuda() && xs[i].scalar_type() == xs[0].scalar_type() && dys[i].scalar_type() == dys[0].scalar_type() &&
                    xs[i].sizes() == xs[0].sizes() && xs[i].strides() == xs[0].strides() && dys[i].sizes() == dys[0].sizes() && dys[i].strides() == dys[0].strides(),
                    "conv_wgrad_multi: the items must agree in device, dtype, shape and strides");
        xp.push_back(xs[i].data_ptr()); dp.push_back(dys[i].data_ptr());
    }
    TORCH_CHECK((xh || xs[0].scalar_type() == at::kFloat) && (dyh || dys[0].scalar_type() == at::kFloat), "conv_wgrad_multi: fp32 or fp16 tensors");
    TORCH_CHECK(prec == 2 || (!xh && !dyh), "conv_wgrad_multi: fp16 tensors exist in the native f16 form only (prec 2)");
    TORCH_CHECK(dims.size() == 22, "conv_wgrad_multi: dims = [B total, D, H, W, Ci, xCs, Do, Ho, Wo, Co, dyCs, kd, kh, kw, stride, pad x3, dil x3, transposed]");
    int d[22];
    for (int i = 0; i < 22; ++i) d[i] = (int)dims[i];
    const size_t need = prec == 0
        ? osa_conv3d_wgrad_workspace_bytes(d[0], d[1], d[2], d[3], d[4], d[6], d[7], d[8], d[9], d[11], d[12], d[13], d[14], d[15], d[16], d[17], d[18], d[19], d[20], d[21])
        : osa_conv3d_wgrad_f16x3_workspace_bytes(d[0], d[1], d[2], d[3], d[4], d[6], d[7], d[8], d[9], d[11], d[12], d[13], d[14], d[15], d[16], d[17], d[18], d[19], d[20], d[21]);
    if (need == 0) {
        TORCH_CHECK(prec != 0, "conv_wgrad_multi: unsupported layer");
        return false;
    }
    auto ws = at::empty({(int64_t)((need + 3) / 4)}, dw.options());
    OSA_CALL(osa_conv3d_wgrad_ws_multi((int)prec, xp.data(), dp.data(), (int)xs.size(), dw.data_ptr<float>(), d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], d[10],
                                       d[11], d[12], d[13], d[14], d[15], d[16], d[17], d[18], d[19], d[20], d[21], fpo(x_meta), fpo(dy_meta), xh ? 1 : 0, dyh ? 1 : 0,
                                       ws.data_ptr<float>(), need, cur_stream()));
    return true;
}

// ---- layout, packing, ConvGRU gates, geometry-encoding lookup: the remaining per-step launches of the training path --------------------
// x [B, C, ...S] contiguous NCDHW -> channels [c_off, c_off + C) of the NDHWC tensor y (voxel stride = y's channel stride)
void to_cl(const at::Tensor& x, at::Tensor y, int64_t C, int64_t S, int64_t c_off) {
    gpu_f32(x, "x"); gpu_f32(y, "y");
    TORCH_CHECK(x.is_contiguous() && y.dim() == 5 && y.stride(1) == 1, "to_cl: x contiguous NCDHW, y NDHWC");
    OSA_CALL(osa_ncdhw_to_ndhwc_f32(fp(x), y.data_ptr<float>(), (int)x.size(0), (int)C, (long long)S, (int)y.size(1), (int)c_off, cur_stream()));
}
void to_ncdhw(const at::Tensor& x, at::Tensor y, int64_t C, int64_t S, int64_t c_off) {
    gpu_f32(x, "x"); gpu_f32(y, "y");
    TORCH_CHECK(y.is_contiguous() && x.dim() == 5 && x.stride(1) == 1, "to_ncdhw: x NDHWC, y contiguous NCDHW");
    OSA_CALL(osa_ndhwc_to_ncdhw_f32(fp(x), y.data_ptr<float>(), (int)x.size(0), (int)C, (long long)S, (int)x.size(1), (int)c_off, cur_stream()));
}

// conv weights -> MFMA operand order.  geom = [Ci, Co, kd, kh, kw, src_transposed, flip]; prec 0 f32 / 1 f16x3 / 2 f16.  f16x3: the power-of-two
// weight scale is derived on the device from w_amax (1 float) and written to scale_out (2 floats) -- no host synchronisation.
void conv_pack(const at::Tensor& w, at::Tensor packed, at::IntArrayRef geom, int64_t prec, const c10::optional<at::Tensor>& w_amax, const c10::optional<at::Tensor>& scale_out) {
    gpu_f32(w, "w"); gpu_f32(packed, "packed");
    TORCH_CHECK(geom.size() == 7 && w.is_contiguous(), "conv_pack: geom = [Ci, Co, kd, kh, kw, src_transposed, flip], w contiguous");
    const int g[7] = {(int)geom[0], (int)geom[1], (int)geom[2], (int)geom[3], (int)geom[4], (int)geom[5], (int)geom[6]};
    if (prec == 1) {
        TORCH_CHECK(w_amax.has_value() && scale_out.has_value(), "conv_pack: the f16x3 mode needs w_amax and scale_out (device tensors)");
        OSA_CALL(osa_conv3d_pack_ex_auto(fp(w), packed.data_ptr<float>(), g[0], g[1], g[2], g[3], g[4], g[5], g[6], fpo(w_amax), const_cast<float*>(fpo(scale_out)), cur_stream()));
    } else OSA_CALL(osa_conv3d_pack_ex(fp(w), packed.data_ptr<float>(), g[0], g[1], g[2], g[3], g[4], g[5], g[6], prec == 2 ? 2 : 0, 1.0f, cur_stream()));
}
// transposed-conv weights [Ci][Co][k..] -> parity-class packing; geom = [Ci, Co, k, pad, flat (1 = 2-D)]
void deconv_pack(const at::Tensor& w, at::Tensor packed, at::IntArrayRef geom, int64_t prec, const c10::optional<at::Tensor>& w_amax, const c10::optional<at::Tensor>& scale_out) {
    gpu_f32(w, "w"); gpu_f32(packed, "packed");
    TORCH_CHECK(geom.size() == 5 && w.is_contiguous(), "deconv_pack: geom = [Ci, Co, k, pad, flat], w contiguous");
    const int Ci = (int)geom[0], Co = (int)geom[1], k = (int)geom[2], pad = (int)geom[3];
    const bool flat = geom[4] != 0;
    float* dst = packed.data_ptr<float>();
    void* st = cur_stream();
    if (prec == 1) {
        TORCH_CHECK(w_amax.has_value() && scale_out.has_value(), "deconv_pack: the f16x3 mode needs w_amax and scale_out (device tensors)");
        float* so = const_cast<float*>(fpo(scale_out));
        if (flat) OSA_CALL(osa_deconv2d_pack_f16x3_auto(fp(w), dst, Ci, Co, k, pad, fpo(w_amax), so, st));
        else OSA_CALL(osa_deconv3d_pack_f16x3_auto(fp(w), dst, Ci, Co, k, pad, fpo(w_amax), so, st));
    } else if (prec == 2) {
        if (flat) OSA_CALL(osa_deconv2d_pack_f16(fp(w), dst, Ci, Co, k, pad, st));
        else OSA_CALL(osa_deconv3d_pack_f16(fp(w), dst, Ci, Co, k, pad, st));
    } else {
        if (flat) OSA_CALL(osa_deconv2d_pack_f32(fp(w), dst, Ci, Co, k, pad, st));
        else OSA_CALL(osa_deconv3d_pack_f32(fp(w), dst, Ci, Co, k, pad, st));
    }
}

// ConvGRU gate arithmetic of the training path (csrc/gru_train.hip).  Every tensor: logical [B, C', H, W] read / written as NHWC with its own
// channel stride (stride(1) == 1, stride(3) = channel stride), fp32 or fp16.
inline osa_nhwc_ref nref(const at::Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda() && t.dim() == 4 && t.stride(1) == 1 && (t.scalar_type() == at::kFloat || t.scalar_type() == at::kHalf),
                "openstereo_amd: ", name, " must be a CUDA fp32 / fp16 tensor [B,C,H,W] in NHWC memory order");
    TORCH_CHECK(t.stride(2) == t.size(3) * t.stride(3) && t.stride(0) == t.size(2) * t.stride(2), "openstereo_amd: ", name, " is not dense over its pixels");
    osa_nhwc_ref r;
    r.ptr = t.data_ptr(); r.cs = (int)t.stride(3); r.f16 = t.scalar_type() == at::kHalf ? 1 : 0;
    return r;
}
void gru_gates_rz_fwd(const at::Tensor& pre, const c10::optional<at::Tensor>& bz, const c10::optional<at::Tensor>& br, const at::Tensor& cz, const at::Tensor& cr,
                      const at::Tensor& h, at::Tensor z, at::Tensor rh) {
    const auto a = nref(pre, "pre"), b = nref(cz, "cz"), c = nref(cr, "cr"), d = nref(h, "h"), e = nref(z, "z"), f = nref(rh, "rh");
    OSA_CALL(osa_gru_gates_rz_fwd(&a, fpo(bz), fpo(br), &b, &c, &d, &e, &f, (long long)h.size(0) * h.size(2) * h.size(3), (int)h.size(1), cur_stream()));
}
void gru_gates_rz_bwd(const at::Tensor& pre, const c10::optional<at::Tensor>& bz, const c10::optional<at::Tensor>& br, const at::Tensor& cz, const at::Tensor& cr,
                      const at::Tensor& h, const at::Tensor& dz, const at::Tensor& drh, at::Tensor dpre, at::Tensor dh) {
    const auto a = nref(pre, "pre"), b = nref(cz, "cz"), c = nref(cr, "cr"), d = nref(h, "h"), e = nref(dz, "dz"), f = nref(drh, "drh"), g = nref(dpre, "dpre"), i = nref(dh, "dh");
    OSA_CALL(osa_gru_gates_rz_bwd(&a, fpo(bz), fpo(br), &b, &c, &d, &e, &f, &g, &i, (long long)h.size(0) * h.size(2) * h.size(3), (int)h.size(1), cur_stream()));
}
void gru_gates_q_fwd(const at::Tensor& z, const at::Tensor& qpre, const c10::optional<at::Tensor>& bq, const at::Tensor& cq, const at::Tensor& h, at::Tensor out) {
    const auto a = nref(z, "z"), b = nref(qpre, "qpre"), c = nref(cq, "cq"), d = nref(h, "h"), e = nref(out, "out");
    OSA_CALL(osa_gru_gates_q_fwd(&a, &b, fpo(bq), &c, &d, &e, (long long)h.size(0) * h.size(2) * h.size(3), (int)h.size(1), cur_stream()));
}
void gru_gates_q_bwd(const at::Tensor& z, const at::Tensor& qpre, const c10::optional<at::Tensor>& bq, const at::Tensor& cq, const at::Tensor& h, const at::Tensor& dout,
                     at::Tensor dz, at::Tensor dqpre, at::Tensor dh) {
    const auto a = nref(z, "z"), b = nref(qpre, "qpre"), c = nref(cq, "cq"), d = nref(h, "h"), e = nref(dout, "dout"), f = nref(dz, "dz"), g = nref(dqpre, "dqpre"), i = nref(dh, "dh");
    OSA_CALL(osa_gru_gates_q_bwd(&a, &b, fpo(bq), &c, &d, &e, &f, &g, &i, (long long)h.size(0) * h.size(2) * h.size(3), (int)h.size(1), cur_stream()));
}

// geometry-encoding lookup of the GRU loop (igev/geometry.py, stereobase/gru_blocks.py:170-231): `levels` = the geo pyramid followed by the corr pyramid
// (rows of [.., len_l] floats); forward writes out [B, (C + 1)(2r + 1) L, H, W], backward writes the gradients of every level (same shapes).
void geo_lookup(at::TensorList levels, const at::Tensor& disp, const at::Tensor& coords_x, at::Tensor out, int64_t C, int64_t radius) {
    const size_t L = levels.size() / 2;
    TORCH_CHECK(L >= 1 && L <= 4 && levels.size() == 2 * L, "geo_lookup: levels = geo pyramid + corr pyramid, 1..4 levels each");
    gpu_f32(disp, "disp"); gpu_f32(coords_x, "coords_x"); gpu_f32(out, "out");
    const float* gp[4]; const float* cp[4]; int gl[4], cl[4];
    for (size_t l = 0; l < L; ++l) {
        gp[l] = fp(gpu_f32(levels[l], "geo level")); cp[l] = fp(gpu_f32(levels[L + l], "corr level"));
        gl[l] = (int)levels[l].size(-1); cl[l] = (int)levels[L + l].size(-1);
    }
    OSA_CALL(osa_geo_lookup_f32(gp, cp, gl, cl, (int)L, fp(disp), fp(coords_x), out.data_ptr<float>(), (int)disp.size(0), (int)disp.size(1), (int)disp.size(2), (int)C, (int)radius, cur_stream()));
}
static void geo_lookup_bwd_any(bool acc, at::TensorList dlevels, const at::Tensor& disp, const at::Tensor& coords_x, const at::Tensor& dout, int64_t C, int64_t radius) {
    const size_t L = dlevels.size() / 2;
    TORCH_CHECK(L >= 1 && L <= 4 && dlevels.size() == 2 * L, "geo_lookup_bwd: dlevels = gradients of the geo pyramid + of the corr pyramid");
    gpu_f32(disp, "disp"); gpu_f32(coords_x, "coords_x"); gpu_f32(dout, "dout");
    float* gp[4]; float* cp[4]; int gl[4], cl[4];
    for (size_t l = 0; l < L; ++l) {
        gp[l] = gpu_f32(dlevels[l], "dgeo level").data_ptr<float>(); cp[l] = gpu_f32(dlevels[L + l], "dcorr level").data_ptr<float>();
        gl[l] = (int)dlevels[l].size(-1); cl[l] = (int)dlevels[L + l].size(-1);
    }
    if (acc) OSA_CALL(osa_geo_lookup_bwd_acc_f32(gp, cp, gl, cl, (int)L, fp(disp), fp(coords_x), fp(dout), (int)disp.size(0), (int)disp.size(1), (int)disp.size(2), (int)C, (int)radius, cur_stream()));
    else OSA_CALL(osa_geo_lookup_bwd_f32(gp, cp, gl, cl, (int)L, fp(disp), fp(coords_x), fp(dout), (int)disp.size(0), (int)disp.size(1), (int)disp.size(2), (int)C, (int)radius, cur_stream()));
}
void geo_lookup_bwd(at::TensorList dlevels, const at::Tensor& disp, const at::Tensor& coords_x, const at::Tensor& dout, int64_t C, int64_t radius) {
    geo_lookup_bwd_any(false, dlevels, disp, coords_x, dout, C, radius);
}
void geo_lookup_bwd_acc(at::TensorList dlevels, const at::Tensor& disp, const at::Tensor& coords_x, const at::Tensor& dout, int64_t C, int64_t radius) {
    geo_lookup_bwd_any(true, dlevels, disp, coords_x, dout, C, radius);     // accumulates into dlevels (zero-filled once per step by the caller)
}

// ---- r5, second batch: the rest of the launch path (inference helpers of the GRU loop, LightStereo's depthwise layers, weight packing, the
// fused redir branch, preprocessing), so that a process with the extension loaded issues NO launch through ctypes (tools/ctypes_census.py).
// These are in-place engine launches on caller-allocated buffers (allocation policy and range-block bookkeeping stay with the Python layer
// classes), hence the Tensor(a!) schemas and no Meta / Autograd registrations: inference-only, nothing to trace or differentiate.
inline float* mfp(const c10::optional<at::Tensor>& t) { return (t.has_value() && t->defined() && t->numel()) ? t->data_ptr<float>() : nullptr; }
inline const at::Tensor* opt(const c10::optional<at::Tensor>& t) { return (t.has_value() && t->defined()) ? &*t : nullptr; }

// osa_build_volume_f32: group-wise correlation and / or concatenation volume into `out` (layout 0 NCDHW / 1 NDHWC with VC channel stride, c_off)
void build_volume(const c10::optional<at::Tensor>& lg, const c10::optional<at::Tensor>& rg, int64_t G, const c10::optional<at::Tensor>& lc,
                  const c10::optional<at::Tensor>& rc, at::Tensor out, int64_t layout, int64_t VC, int64_t c_off, int64_t maxdisp, bool mask_left,
                  const c10::optional<at::Tensor>& meta) {
    const at::Tensor* g0 = opt(lg); const at::Tensor* c0 = opt(lc);
    TORCH_CHECK(g0 || c0, "build_volume: no features");
    const at::Tensor& ref = g0 ? *g0 : *c0;
    gpu_f32(ref, "features"); gpu_f32(out, "out");
    TORCH_CHECK(ref.dim() == 4, "build_volume: features are [B,C,H,W]");
    for (const auto* t : {opt(lg), opt(rg), opt(lc), opt(rc)})
        if (t) TORCH_CHECK(gpu_f32(*t, "features").is_contiguous() && t->size(0) == ref.size(0) && t->size(2) == ref.size(2) && t->size(3) == ref.size(3),
                           "build_volume: contiguous fp32 feature maps of one spatial shape");
    TORCH_CHECK((g0 != nullptr) == (opt(rg) != nullptr) && (c0 != nullptr) == (opt(rc) != nullptr), "build_volume: left and right maps come in pairs");
    OSA_CALL(osa_build_volume_f32(fpo(lg), fpo(rg), g0 ? (int)g0->size(1) : 0, (int)G, fpo(lc), fpo(rc), c0 ? (int)c0->size(1) : 0, out.data_ptr<float>(),
                                  (int)layout, (int)VC, (int)c_off, (int)ref.size(0), (int)ref.size(2), (int)ref.size(3), (int)maxdisp, mask_left ? 1 : 0, mfp(meta),
                                  cur_stream()));
}

// transposed conv + fused 1x1x1 redir branch (hourglass.py:52-56: relu(conv6(c5) + redir1(x))); dims = [B, D, H, W, Ci, xCs, Co, yCs],
// geom = [k, pad, opad], rdims = [rxCs, rCi]; prec 0 f32 / 1 f16x3 (metas as in conv_ndhwc)
void deconv_redir(const at::Tensor& x, int64_t x_off, const at::Tensor& packed, const c10::optional<at::Tensor>& scale, const c10::optional<at::Tensor>& shift,
                  at::Tensor out, int64_t out_off, at::IntArrayRef dims, at::IntArrayRef geom, const at::Tensor& rx, at::IntArrayRef rdims, const at::Tensor& rpacked,
                  const c10::optional<at::Tensor>& rscale, const c10::optional<at::Tensor>& rshift, double r_out_scale, int64_t prec, int64_t act, double slope,
                  double out_scale, at::TensorList metas) {
    gpu_f32(x, "x"); gpu_f32(out, "out"); gpu_f32(rx, "redir input"); gpu_f32(packed, "packed"); gpu_f32(rpacked, "redir packed");
    TORCH_CHECK(dims.size() == 8 && geom.size() == 3 && rdims.size() == 2, "deconv_redir: dims = [B, D, H, W, Ci, xCs, Co, yCs], geom = [k, pad, opad], rdims = [rxCs, rCi]");
    TORCH_CHECK(prec == 0 || prec == 1, "deconv_redir: f32 and f16x3 only (the f16 mode runs the 1x1x1 layer separately)");
    const int d[8] = {(int)dims[0], (int)dims[1], (int)dims[2], (int)dims[3], (int)dims[4], (int)dims[5], (int)dims[6], (int)dims[7]};
    const float* xp = (const float*)vp(x, x_off);
    float* yp = (float*)vp(out, out_off);
    if (prec == 0) {
        OSA_CALL(osa_deconv3d_redir_ndhwc_f32(xp, fp(packed), fpo(scale), fpo(shift), yp, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], (int)geom[0], (int)geom[1], (int)geom[2],
                                              fp(rx), (int)rdims[0], (int)rdims[1], fp(rpacked), fpo(rscale), fpo(rshift), (int)act, (float)slope, cur_stream()));
        return;
    }
    TORCH_CHECK(metas.size() == 7, "deconv_redir: the f16x3 mode takes the 7 range blocks of conv_ndhwc");
    auto mp = [&](size_t i) -> float* { return metas[i].defined() && metas[i].numel() ? metas[i].data_ptr<float>() : nullptr; };
    osa_f16x3_ranges rng{};
    rng.x_meta = mp(0); rng.residual_meta = mp(1); rng.redir_meta = mp(2); rng.y_meta = mp(3); rng.bound_coef = mp(4); rng.redir_bound_coef = mp(5); rng.weight_scale = mp(6);
    OSA_CALL(osa_deconv3d_redir_ndhwc_f16x3(xp, fp(packed), fpo(scale), fpo(shift), yp, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], (int)geom[0], (int)geom[1], (int)geom[2],
                                            fp(rx), (int)rdims[0], (int)rdims[1], fp(rpacked), fpo(rscale), fpo(rshift), (float)r_out_scale, (int)act, (float)slope,
                                            (float)out_scale, &rng, cur_stream()));
}

// classification conv with Co <= 4 (gwcnet_disp_processor.py:64-81 classif*: 32 -> 1): dims = [B, D, H, W, Ci, xCs, Co, yCs], geom = [kd, kh, kw, pd, ph, pw]
void small_co_conv(const at::Tensor& x, const at::Tensor& packed, const c10::optional<at::Tensor>& bias, const c10::optional<at::Tensor>& residual, at::Tensor y,
                   at::IntArrayRef dims, at::IntArrayRef geom) {
    gpu_f32(x, "x"); gpu_f32(packed, "packed"); gpu_f32(y, "y");
    TORCH_CHECK(dims.size() == 8 && geom.size() == 6, "small_co_conv: dims = [B, D, H, W, Ci, xCs, Co, yCs], geom = [kd, kh, kw, pd, ph, pw]");
    OSA_CALL(osa_conv3d_small_co_packed_ndhwc_f32(fp(x), fp(packed), fpo(bias), fpo(residual), y.data_ptr<float>(), (int)dims[0], (int)dims[1], (int)dims[2], (int)dims[3],
                                                  (int)dims[4], (int)dims[5], (int)dims[6], (int)dims[7], (int)geom[0], (int)geom[1], (int)geom[2], (int)geom[3], (int)geom[4],
                                                  (int)geom[5], cur_stream()));
}

// depthwise 2-D conv + folded norm + activation (+ add) on NHWC maps (lightstereo aggregation.py:79-113): dims = [B, Hi, Wi, C, xCs, yCs, aCs],
// geom = [kh, kw, stride, pad_h, pad_w, dil_h, dil_w]
void dwconv2d(const at::Tensor& x, const at::Tensor& packed, const c10::optional<at::Tensor>& scale, const c10::optional<at::Tensor>& shift,
              const c10::optional<at::Tensor>& add, at::Tensor y, at::IntArrayRef dims, at::IntArrayRef geom, int64_t act, const c10::optional<at::Tensor>& y_meta) {
    gpu_f32(x, "x"); gpu_f32(packed, "packed"); gpu_f32(y, "y");
    TORCH_CHECK(dims.size() == 7 && geom.size() == 7, "dwconv2d: dims = [B, Hi, Wi, C, xCs, yCs, aCs], geom = [kh, kw, stride, pad_h, pad_w, dil_h, dil_w]");
    OSA_CALL(osa_dwconv2d_nhwc_f32(fp(x), fp(packed), fpo(scale), fpo(shift), fpo(add), y.data_ptr<float>(), (int)dims[0], (int)dims[1], (int)dims[2], (int)dims[3], (int)dims[4],
                                   (int)dims[5], (int)dims[6], (int)geom[0], (int)geom[1], (int)geom[2], (int)geom[3], (int)geom[4], (int)geom[5], (int)geom[6], (int)act,
                                   mfp(y_meta), cur_stream()));
}

// the 3 x 3 depthwise layers with fp16 input and / or output tensors (the f16 mode's chain tensors): dims = [B, Hi, Wi, C, xCs, yCs], geom = [kh, kw, stride, pad_h, pad_w]
void dwconv2d_f16io(const at::Tensor& x, const at::Tensor& packed, const c10::optional<at::Tensor>& scale, const c10::optional<at::Tensor>& shift,
                    at::Tensor y, at::IntArrayRef dims, at::IntArrayRef geom, int64_t act, const c10::optional<at::Tensor>& y_meta) {
    gpu_f32(packed, "packed");
    const bool xh = x.scalar_type() == at::kHalf, yh = y.scalar_type() == at::kHalf;
    TORCH_CHECK(x.is_cuda() && y.is_cuda() && (xh || x.scalar_type() == at::kFloat) && (yh || y.scalar_type() == at::kFloat), "dwconv2d_f16io: CUDA fp32 / fp16 tensors");
    TORCH_CHECK(dims.size() == 6 && geom.size() == 5, "dwconv2d_f16io: dims = [B, Hi, Wi, C, xCs, yCs], geom = [kh, kw, stride, pad_h, pad_w]");
    OSA_CALL(osa_dwconv2d_nhwc_f16io(x.data_ptr(), xh ? 1 : 0, fp(packed), fpo(scale), fpo(shift), y.data_ptr(), yh ? 1 : 0, (int)dims[0], (int)dims[1], (int)dims[2], (int)dims[3],
                                     (int)dims[4], (int)dims[5], (int)geom[0], (int)geom[1], (int)geom[2], (int)geom[3], (int)geom[4], (int)act, mfp(y_meta), cur_stream()));
}

// h' = (1 - z) h + z q of the inference GRU loop (igev/update.py:45) on channel slices of the level buffer: dims = [npix, C, zCs, qCs, hCs, oCs]
void gru_combine(const at::Tensor& z, int64_t z_off, const at::Tensor& q, const at::Tensor& h, at::Tensor out, at::IntArrayRef dims, const c10::optional<at::Tensor>& out_meta) {
    gpu_f32(z, "z"); gpu_f32(q, "q"); gpu_f32(h, "h"); gpu_f32(out, "out");
    TORCH_CHECK(dims.size() == 6, "gru_combine: dims = [npix, C, zCs, qCs, hCs, oCs]");
    OSA_CALL(osa_gru_combine_f32(fp(z) + z_off, fp(q), fp(h), out.data_ptr<float>(), (long long)dims[0], (int)dims[1], (int)dims[2], (int)dims[3], (int)dims[4], (int)dims[5],
                                 mfp(out_meta), cur_stream()));
}

// pool2x (update.py:99-100; kind 0, dims = [B, H, W, C, xCs, yCs]) / bilinear align_corners resize (:107-109; kind 1, dims = [B, Hi, Wi, Ho, Wo, C, xCs, yCs])
// of channels [0, C) of x into channels [y_off, y_off + C) of y
void resample_nhwc(const at::Tensor& x, at::Tensor y, int64_t y_off, int64_t kind, at::IntArrayRef dims, const c10::optional<at::Tensor>& x_meta,
                   const c10::optional<at::Tensor>& y_meta) {
    gpu_f32(x, "x"); gpu_f32(y, "y");
    float* yp = y.data_ptr<float>() + y_off;
    if (kind == 0) {
        TORCH_CHECK(dims.size() == 6, "resample_nhwc: pool dims = [B, H, W, C, xCs, yCs]");
        OSA_CALL(osa_pool2x_nhwc_f32(fp(x), yp, (int)dims[0], (int)dims[1], (int)dims[2], (int)dims[3], (int)dims[4], (int)dims[5], fpo(x_meta), mfp(y_meta), cur_stream()));
    } else {
        TORCH_CHECK(dims.size() == 8, "resample_nhwc: resize dims = [B, Hi, Wi, Ho, Wo, C, xCs, yCs]");
        OSA_CALL(osa_resize_bilinear_nhwc_f32(fp(x), yp, (int)dims[0], (int)dims[1], (int)dims[2], (int)dims[3], (int)dims[4], (int)dims[5], (int)dims[6], (int)dims[7],
                                              fpo(x_meta), mfp(y_meta), cur_stream()));
    }
}

// disp += delta and its two other homes (NHWC [disp,0,0,0] map, channel `slot_off` of the 1/4 level's x slot): igev_stereo.py:196-199
void disp_update(at::Tensor disp, const c10::optional<at::Tensor>& delta, int64_t delta_cs, at::Tensor disp4, at::Tensor slot, int64_t slot_off, int64_t slot_cs,
                 int64_t npix, const c10::optional<at::Tensor>& disp4_meta, const c10::optional<at::Tensor>& slot_meta) {
    gpu_f32(disp, "disp"); gpu_f32(disp4, "disp4"); gpu_f32(slot, "slot");
    OSA_CALL(osa_disp_update_f32(disp.data_ptr<float>(), fpo(delta), (int)delta_cs, disp4.data_ptr<float>(), slot.data_ptr<float>() + slot_off, (int)slot_cs, (long long)npix,
                                 mfp(disp4_meta), mfp(slot_meta), cur_stream()));
}

// the lookup of the inference loop: NHWC output with channel stride out_cs (geometry.py lookup_cl); bhw = [B, H, W]
void geo_lookup_nhwc(at::TensorList levels, const at::Tensor& disp, const at::Tensor& coords_x, at::Tensor out, int64_t out_cs, at::IntArrayRef bhw, int64_t C, int64_t radius) {
    const size_t L = levels.size() / 2;
    TORCH_CHECK(L >= 1 && L <= 4 && levels.size() == 2 * L && bhw.size() == 3, "geo_lookup_nhwc: levels = geo pyramid + corr pyramid (1..4 levels each), bhw = [B, H, W]");
    gpu_f32(disp, "disp"); gpu_f32(coords_x, "coords_x"); gpu_f32(out, "out");
    const float* gp[4]; const float* cp[4]; int gl[4], cl[4];
    for (size_t l = 0; l < L; ++l) {
        gp[l] = fp(gpu_f32(levels[l], "geo level")); cp[l] = fp(gpu_f32(levels[L + l], "corr level"));
        gl[l] = (int)levels[l].size(-1); cl[l] = (int)levels[L + l].size(-1);
    }
    OSA_CALL(osa_geo_lookup_nhwc_f32(gp, cp, gl, cl, (int)L, fp(disp), fp(coords_x), out.data_ptr<float>(), (int)out_cs, (int)bhw[0], (int)bhw[1], (int)bhw[2], (int)C, (int)radius,
                                     cur_stream()));
}

// pyramid construction (igev/geometry.py:8-31): all-pairs correlation, volume -> per-pixel rows, row-wise average pooling
void allpairs_corr(const at::Tensor& f1, const at::Tensor& f2, at::Tensor corr) {
    gpu_f32(f1, "fmap1"); gpu_f32(f2, "fmap2"); gpu_f32(corr, "corr");
    TORCH_CHECK(f1.dim() == 4 && f2.dim() == 4 && f1.is_contiguous() && f2.is_contiguous(), "allpairs_corr: contiguous [B,C,H,W] feature maps");
    OSA_CALL(osa_allpairs_corr_f32(fp(f1), fp(f2), corr.data_ptr<float>(), (int)f1.size(0), (int)f1.size(1), (int)f1.size(2), (int)f1.size(3), (int)f2.size(3), cur_stream()));
}
void geo_rows(const at::Tensor& vol, at::Tensor rows, int64_t C) {
    gpu_f32(vol, "volume"); gpu_f32(rows, "rows");
    TORCH_CHECK(vol.dim() == 5, "geo_rows: volume is logical [B,Cs,D,H,W] in NDHWC memory order");
    OSA_CALL(osa_geo_rows_f32(fp(vol), rows.data_ptr<float>(), (int)vol.size(0), (int)vol.size(2), (int)vol.size(3), (int)vol.size(4), (int)C, (int)vol.size(1), cur_stream()));
}
void avgpool_rows(const at::Tensor& x, at::Tensor y) {
    gpu_f32(x, "x"); gpu_f32(y, "y");
    TORCH_CHECK(x.is_contiguous() && x.dim() >= 1 && x.size(-1) > 0, "avgpool_rows: contiguous rows");
    OSA_CALL(osa_avgpool_rows_f32(fp(x), y.data_ptr<float>(), (long long)(x.numel() / x.size(-1)), (int)x.size(-1), cur_stream()));
}

// weight packing of the inference layer classes (host float weight scale).  family 0 conv3d geom = [Ci, Co, kd, kh, kw]; 1 deconv3d / 2 deconv2d
// geom = [Ci, Co, k, pad]; 3 depthwise 2-D geom = [C, kh, kw]; 4 small-Co conv geom = [Ci, Co, kd, kh, kw].  prec 0 f32 / 1 f16x3 / 2 f16.
void weight_pack(const at::Tensor& w, at::Tensor packed, int64_t family, int64_t prec, at::IntArrayRef geom, double wscale) {
    gpu_f32(w, "w"); gpu_f32(packed, "packed");
    TORCH_CHECK(w.is_contiguous(), "weight_pack: contiguous reference-layout weights");
    const float* src = fp(w); float* dst = packed.data_ptr<float>(); void* st = cur_stream();
    auto g = [&](size_t i) { return (int)geom[i]; };
    if (family == 0) {
        TORCH_CHECK(geom.size() == 5, "weight_pack: conv geometry = [Ci, Co, kd, kh, kw]");
        if (prec == 1) OSA_CALL(osa_conv3d_pack_f16x3(src, dst, g(0), g(1), g(2), g(3), g(4), (float)wscale, st));
        else if (prec == 2) OSA_CALL(osa_conv3d_pack_f16(src, dst, g(0), g(1), g(2), g(3), g(4), st));
        else OSA_CALL(osa_conv3d_pack_f32(src, dst, g(0), g(1), g(2), g(3), g(4), st));
    } else if (family == 1 || family == 2) {
        TORCH_CHECK(geom.size() == 4, "weight_pack: transposed-conv geometry = [Ci, Co, k, pad]");
        const bool flat = family == 2;
        if (prec == 1) OSA_CALL((flat ? osa_deconv2d_pack_f16x3 : osa_deconv3d_pack_f16x3)(src, dst, g(0), g(1), g(2), g(3), (float)wscale, st));
        else if (prec == 2) OSA_CALL((flat ? osa_deconv2d_pack_f16 : osa_deconv3d_pack_f16)(src, dst, g(0), g(1), g(2), g(3), st));
        else OSA_CALL((flat ? osa_deconv2d_pack_f32 : osa_deconv3d_pack_f32)(src, dst, g(0), g(1), g(2), g(3), st));
    } else if (family == 3) {
        TORCH_CHECK(geom.size() == 3, "weight_pack: depthwise geometry = [C, kh, kw]");
        OSA_CALL(osa_dwconv2d_pack_f32(src, dst, g(0), g(1), g(2), st));
    } else {
        TORCH_CHECK(family == 4 && geom.size() == 5, "weight_pack: small-Co geometry = [Ci, Co, kd, kh, kw]");
        OSA_CALL(osa_conv3d_small_co_pack_f32(src, dst, g(0), g(1), g(2), g(3), g(4), st));
    }
}

// remaining single-launch helpers: PSMNet's cat_fms volume, the pairwise volumes of the smaller model families, instance norm on NHWC maps, the
// fused preprocessing of one stereo pair (mean / std: 3 host floats each), max |t| into a range block
void cat_fms(const at::Tensor& ref, const at::Tensor& tgt, at::Tensor out, const at::Tensor& disp_index) {
    gpu_f32(ref, "reference_fm"); gpu_f32(tgt, "target_fm"); gpu_f32(out, "out");
    TORCH_CHECK(ref.dim() == 4 && ref.is_contiguous() && tgt.is_contiguous() && disp_index.is_cuda() && disp_index.scalar_type() == at::kInt, "cat_fms: contiguous [B,C,H,W] maps, int32 indices");
    OSA_CALL(osa_cat_fms_f32(fp(ref), fp(tgt), out.data_ptr<float>(), disp_index.data_ptr<int>(), (int)ref.size(0), (int)ref.size(1), (int)ref.size(2), (int)ref.size(3),
                             (int)disp_index.numel(), cur_stream()));
}
void pair_volume(const at::Tensor& l, const at::Tensor& r, at::Tensor out, int64_t groups, int64_t planes, int64_t mode) {
    gpu_f32(l, "left"); gpu_f32(r, "right"); gpu_f32(out, "out");
    TORCH_CHECK(l.dim() == 4 && l.is_contiguous() && r.is_contiguous() && l.sizes() == r.sizes(), "pair_volume: contiguous [B,C,H,W] maps of equal shape");
    OSA_CALL(osa_pair_volume_f32(fp(l), fp(r), out.data_ptr<float>(), (int)l.size(0), (int)l.size(1), (int)groups, (int)l.size(2), (int)l.size(3), (int)planes, (int)mode, cur_stream()));
}
void instnorm_nhwc(const at::Tensor& x, at::Tensor out, int64_t out_off, at::IntArrayRef dims, double eps, int64_t act, double slope, at::Tensor workspace,
                   const c10::optional<at::Tensor>& y_meta) {
    gpu_f32(x, "x"); gpu_f32(out, "out"); gpu_f32(workspace, "workspace");
    TORCH_CHECK(dims.size() == 5, "instnorm_nhwc: dims = [B, HW, C, xCs, yCs]");
    OSA_CALL(osa_instnorm_nhwc_f32(fp(x), out.data_ptr<float>() + out_off, (int)dims[0], (long long)dims[1], (int)dims[2], (int)dims[3], (int)dims[4], (float)eps, (int)act,
                                   (float)slope, workspace.data_ptr<float>(), mfp(y_meta), cur_stream()));
}
void preprocess_pair(const at::Tensor& l, const at::Tensor& r, at::Tensor out, at::IntArrayRef pad_size, at::ArrayRef<double> mean, at::ArrayRef<double> stdv, bool channels_last) {
    TORCH_CHECK(l.is_cuda() && r.is_cuda() && l.dim() == 3 && l.size(2) == 3 && l.sizes() == r.sizes() && l.scalar_type() == r.scalar_type() && l.is_contiguous() && r.is_contiguous(),
                "preprocess_pair: two contiguous [H,W,3] CUDA images of one dtype");
    TORCH_CHECK(l.scalar_type() == at::kByte || l.scalar_type() == at::kFloat, "preprocess_pair: uint8 or float32 images");
    TORCH_CHECK(pad_size.size() == 2 && mean.size() == 3 && stdv.size() == 3, "preprocess_pair: pad_size = [Hp, Wp], mean / std of 3 channels");
    gpu_f32(out, "out");
    const float m3[3] = {(float)mean[0], (float)mean[1], (float)mean[2]}, s3[3] = {(float)stdv[0], (float)stdv[1], (float)stdv[2]};
    OSA_CALL(osa_preprocess_pair_f32(l.data_ptr(), r.data_ptr(), l.scalar_type() == at::kByte ? 1 : 0, (int)l.size(0), (int)l.size(1), (int)pad_size[0], (int)pad_size[1], m3, s3,
                                     out.data_ptr<float>(), channels_last ? 1 : 0, cur_stream()));
}
// per-channel sums over the positions of a channels-last tensor (bias gradients; eval-mode BatchNorm backward): functional, allocates
// the sums [1 or 2][C], the optional dx (dy's dtype and strides) and the workspace
std::tuple<at::Tensor, at::Tensor> channel_sums(const at::Tensor& dy, const c10::optional<at::Tensor>& x, const c10::optional<at::Tensor>& x_shift,
                                                const c10::optional<at::Tensor>& dx_scale, int64_t P, int64_t C, int64_t dy_cs, int64_t x_cs) {
    const bool dyh = dy.scalar_type() == at::kHalf;
    TORCH_CHECK(dy.is_cuda() && (dyh || dy.scalar_type() == at::kFloat), "channel_sums: dy must be a CUDA fp32 / fp16 tensor");
    const bool hx = x.has_value() && x->defined();
    const bool xh = hx && x->scalar_type() == at::kHalf;
    if (hx) TORCH_CHECK(x->is_cuda() && (xh || x->scalar_type() == at::kFloat), "channel_sums: x must be a CUDA fp32 / fp16 tensor");
    const bool hd = dx_scale.has_value() && dx_scale->defined();
    const size_t need = osa_channel_sums_workspace_bytes((long long)P, (int)C);
    TORCH_CHECK(need != 0, "channel_sums: unsupported dims");
    at::Tensor out = at::empty({hx ? 2 : 1, C}, dy.options().dtype(at::kFloat));
    at::Tensor ws = at::empty({(int64_t)((need + 3) / 4)}, dy.options().dtype(at::kFloat));
    at::Tensor dx = hd ? at::empty_strided(dy.sizes(), dy.strides(), dy.options()) : at::Tensor();
    OSA_CALL(osa_channel_sums(dy.data_ptr(), dyh ? 1 : 0, (int)dy_cs, hx ? x->data_ptr() : nullptr, xh ? 1 : 0, (int)x_cs,
                              (x_shift.has_value() && x_shift->defined()) ? x_shift->data_ptr<float>() : nullptr,
                              hd ? dx_scale->data_ptr<float>() : nullptr, hd ? dx.data_ptr() : nullptr, (int)dy_cs, (long long)P, (int)C,
                              out.data_ptr<float>(), ws.data_ptr<float>(), need, cur_stream()));
    return std::make_tuple(out, hd ? dx : at::empty({0}, dy.options()));
}
std::tuple<at::Tensor, at::Tensor> channel_sums_meta(const at::Tensor& dy, const c10::optional<at::Tensor>& x, const c10::optional<at::Tensor>& x_shift,
                                                     const c10::optional<at::Tensor>& dx_scale, int64_t P, int64_t C, int64_t dy_cs, int64_t x_cs) {
    const bool hx = x.has_value() && x->defined(), hd = dx_scale.has_value() && dx_scale->defined();
    return std::make_tuple(at::empty({hx ? 2 : 1, C}, dy.options().dtype(at::kFloat)), hd ? at::empty_strided(dy.sizes(), dy.strides(), dy.options()) : at::empty({0}, dy.options()));
}

at::Tensor channel_sums_multi(at::TensorList dys, int64_t P, int64_t C, int64_t dy_cs) {
    TORCH_CHECK(dys.size() >= 1 && dys.size() <= 24, "channel_sums_multi: 1..24 tensors");
    const bool dyh = dys[0].scalar_type() == at::kHalf;
    std::vector<const void*> ptr;
    for (const auto& t : dys) {
        TORCH_CHECK(t.is_cuda() && t.scalar_type() == dys[0].scalar_type() && t.sizes() == dys[0].sizes() && t.strides() == dys[0].strides() && (dyh || t.scalar_type() == at::kFloat),
                    "channel_sums_multi: CUDA fp32 / fp16 tensors of one dtype, shape and stride set");
        ptr.push_back(t.data_ptr());
    }
    const size_t need = dys.size() * osa_channel_sums_workspace_bytes((long long)P, (int)C);
    TORCH_CHECK(need != 0, "channel_sums_multi: unsupported dims");
    at::Tensor out = at::empty({1, C}, dys[0].options().dtype(at::kFloat));
    at::Tensor ws = at::empty({(int64_t)((need + 3) / 4)}, dys[0].options().dtype(at::kFloat));
    OSA_CALL(osa_channel_sums_multi(ptr.data(), (int)dys.size(), dyh ? 1 : 0, (int)dy_cs, (long long)P, (int)C, out.data_ptr<float>(), ws.data_ptr<float>(), need, cur_stream()));
    return out;
}
at::Tensor channel_sums_multi_meta(at::TensorList dys, int64_t P, int64_t C, int64_t dy_cs) {
    return at::empty({1, C}, dys[0].options().dtype(at::kFloat));
}

// training form of the fused softmax + convex up-sampling: logits [B,9,H,W] of any strides, fp32 / fp16
at::Tensor context_upsample_logits(const at::Tensor& disp_low, const at::Tensor& logits, int64_t scale, double gain) {
    at::Tensor d = gpu_f32(disp_low, "disp_low").contiguous();
    TORCH_CHECK(d.dim() == 4 && d.size(1) == 1 && logits.dim() == 4 && logits.size(1) == 9 && logits.is_cuda(), "context_upsample_logits: disp [B,1,h,w], logits [B,9,s*h,s*w]");
    const bool lh = logits.scalar_type() == at::kHalf;
    TORCH_CHECK(lh || logits.scalar_type() == at::kFloat, "context_upsample_logits: fp32 / fp16 logits");
    const int64_t B = d.size(0), h = d.size(2), w = d.size(3);
    TORCH_CHECK(logits.size(0) == B && logits.size(2) == scale * h && logits.size(3) == scale * w, "context_upsample_logits: logits must be at ", scale, "x the disparity's resolution");
    at::Tensor out = at::empty({B, scale * h, scale * w}, d.options());
    const long long ls[4] = {(long long)logits.stride(0), (long long)logits.stride(1), (long long)logits.stride(2), (long long)logits.stride(3)};
    OSA_CALL(osa_context_upsample_logits_f32(fp(d), logits.data_ptr(), lh ? 1 : 0, ls, out.data_ptr<float>(), (int)B, (int)h, (int)w, (int)scale, (float)gain, cur_stream()));
    return out;
}
std::tuple<at::Tensor, at::Tensor> context_upsample_logits_bwd(const at::Tensor& disp_low, const at::Tensor& logits, const at::Tensor& dout, int64_t scale, double gain) {
    at::Tensor d = gpu_f32(disp_low, "disp_low").contiguous();
    at::Tensor g = gpu_f32(dout, "dout").contiguous();
    const bool lh = logits.scalar_type() == at::kHalf;
    const int64_t B = d.size(0), h = d.size(2), w = d.size(3);
    TORCH_CHECK(logits.dim() == 4 && logits.size(1) == 9 && g.numel() == B * scale * h * scale * w, "context_upsample_logits_bwd: shapes");
    at::Tensor dd = at::empty_like(d);
    // dlogits: contiguous NCHW -- what the torch composition's softmax backward hands on (a 9-of-12-channel NHWC slice would reach the next
    // backward nodes as a 9-channel channels-last tensor after the autocast cast, which torch reduces at 97 us per call)
    at::Tensor dl = at::empty(logits.sizes(), logits.options());
    at::Tensor sc = at::empty({B, 9, h, w}, d.options());
    const long long ls[4] = {(long long)logits.stride(0), (long long)logits.stride(1), (long long)logits.stride(2), (long long)logits.stride(3)};
    const long long ds[4] = {(long long)dl.stride(0), (long long)dl.stride(1), (long long)dl.stride(2), (long long)dl.stride(3)};
    OSA_CALL(osa_context_upsample_logits_bwd_f32(fp(d), logits.data_ptr(), lh ? 1 : 0, ls, fp(g), dd.data_ptr<float>(), dl.data_ptr(), ds, sc.data_ptr<float>(),
                                                 (int)B, (int)h, (int)w, (int)scale, (float)gain, cur_stream()));
    return std::make_tuple(dd, dl);
}
at::Tensor context_upsample_logits_meta(const at::Tensor& d, const at::Tensor& logits, int64_t scale, double) {
    return at::empty({d.size(0), scale * d.size(2), scale * d.size(3)}, d.options().dtype(at::kFloat));
}
std::tuple<at::Tensor, at::Tensor> context_upsample_logits_bwd_meta(const at::Tensor& d, const at::Tensor& logits, const at::Tensor& dout, int64_t, double) {
    return std::make_tuple(at::empty(d.sizes(), d.options().dtype(at::kFloat)), at::empty(logits.sizes(), logits.options()));
}

// out = u * a[c] + (v * b[c]) + c0[c] (+ ReLU) on channels-last rows; out: u's dtype and strides
at::Tensor channel_affine(const at::Tensor& u, const c10::optional<at::Tensor>& v, const at::Tensor& a, const c10::optional<at::Tensor>& b, const at::Tensor& c0,
                          int64_t P, int64_t C, int64_t u_cs, int64_t v_cs, bool relu) {
    const bool uh = u.scalar_type() == at::kHalf;
    TORCH_CHECK(u.is_cuda() && (uh || u.scalar_type() == at::kFloat), "channel_affine: u must be a CUDA fp32 / fp16 tensor");
    const bool hv = v.has_value() && v->defined();
    const bool vh = hv && v->scalar_type() == at::kHalf;
    if (hv) TORCH_CHECK(v->is_cuda() && (vh || v->scalar_type() == at::kFloat) && b.has_value() && b->defined(), "channel_affine: v must be a CUDA fp32 / fp16 tensor and come with b");
    gpu_f32(a, "a"); gpu_f32(c0, "c0");
    at::Tensor out = at::empty_strided(u.sizes(), u.strides(), u.options());
    OSA_CALL(osa_channel_affine(u.data_ptr(), uh ? 1 : 0, (int)u_cs, hv ? v->data_ptr() : nullptr, vh ? 1 : 0, (int)v_cs, a.data_ptr<float>(),
                                hv ? b->data_ptr<float>() : nullptr, c0.data_ptr<float>(), out.data_ptr(), (int)u_cs, (long long)P, (int)C, relu ? 1 : 0, cur_stream()));
    return out;
}
at::Tensor channel_affine_meta(const at::Tensor& u, const c10::optional<at::Tensor>& v, const at::Tensor& a, const c10::optional<at::Tensor>& b, const at::Tensor& c0,
                               int64_t P, int64_t C, int64_t u_cs, int64_t v_cs, bool relu) {
    return at::empty_strided(u.sizes(), u.strides(), u.options());
}

void amax_into(const at::Tensor& t, at::Tensor meta) {
    gpu_f32(t, "t"); gpu_f32(meta, "meta");
    OSA_CALL(osa_amax_f32(fp(t), (long long)t.numel(), meta.data_ptr<float>(), cur_stream()));
}

// ---- Meta kernels (shape / dtype inference in C++: FakeTensor tracing, torch.export, torch.compile need no Python shim) --------------------
at::Tensor gwc_volume_meta(const at::Tensor& l, const at::Tensor& r, int64_t maxdisp, int64_t groups) {
    TORCH_CHECK(l.dim() == 4 && l.sizes() == r.sizes() && groups > 0 && l.size(1) % groups == 0, "gwc_volume: [B,C,H,W] features of equal shape, C % groups == 0");
    return at::empty({l.size(0), groups, maxdisp, l.size(2), l.size(3)}, l.options());
}
at::Tensor concat_volume_meta(const at::Tensor& l, const at::Tensor& r, int64_t maxdisp, bool) {
    TORCH_CHECK(l.dim() == 4 && l.sizes() == r.sizes(), "concat_volume: [B,C,H,W] features of equal shape");
    return at::empty({l.size(0), 2 * l.size(1), maxdisp, l.size(2), l.size(3)}, l.options());
}
at::Tensor corr_volume_meta(const at::Tensor& l, const at::Tensor& r, int64_t maxdisp) {
    TORCH_CHECK(l.dim() == 4 && l.sizes() == r.sizes(), "corr_volume: [B,C,H,W] features of equal shape");
    return at::empty({l.size(0), maxdisp, l.size(2), l.size(3)}, l.options());
}
at::Tensor softargmin_meta(const at::Tensor& p) { TORCH_CHECK(p.dim() == 4, "softargmin: [B,D,H,W]"); return at::empty({p.size(0), p.size(2), p.size(3)}, p.options()); }
std::tuple<at::Tensor, at::Tensor> softmax_softargmin_meta(const at::Tensor& c, bool return_prob) {
    TORCH_CHECK(c.dim() == 4, "softmax_softargmin: [B,D,H,W]");
    return {at::empty({c.size(0), c.size(2), c.size(3)}, c.options()), return_prob ? at::empty_like(c) : at::empty({0}, c.options())};
}
at::Tensor upsample_softargmin_meta(const at::Tensor& c, int64_t, int64_t h, int64_t w, bool) {
    TORCH_CHECK(c.dim() == 4, "upsample_softargmin: [B,Dl,Hl,Wl]");
    return at::empty({c.size(0), h, w}, c.options());
}
at::Tensor context_upsample_meta(const at::Tensor& d, const at::Tensor& wt, int64_t scale, bool, double) {
    TORCH_CHECK(d.dim() == 4 && d.size(1) == 1 && wt.dim() == 4 && wt.size(1) == 9, "context_upsample: disp [B,1,h,w], weights [B,9,s*h,s*w]");
    return at::empty({d.size(0), scale * d.size(2), scale * d.size(3)}, d.options());
}
std::tuple<at::Tensor, at::Tensor> volume_bwd_meta(const at::Tensor& dvol, const c10::optional<at::Tensor>&, const c10::optional<at::Tensor>&, at::IntArrayRef shape, int64_t,
                                                   int64_t, bool, bool) {
    return {at::empty(shape, dvol.options()), at::empty(shape, dvol.options())};
}
at::Tensor softargmin_bwd_meta(const at::Tensor& g, int64_t D) { return at::empty({g.size(0), D, g.size(1), g.size(2)}, g.options()); }
at::Tensor softmax_softargmin_bwd_meta(const at::Tensor& c, const at::Tensor&) { return at::empty_like(c); }
at::Tensor upsample_softargmin_bwd_meta(const at::Tensor& c, const at::Tensor&, int64_t, int64_t, int64_t, bool) { return at::empty_like(c); }

// ---- autograd in C++ (TORCH_LIBRARY_IMPL(osa_native, Autograd, ...)): torch.ops.osa_native.* are differentiable without any Python ------
// Each Function redispatches below the Autograd key for its forward and calls the *_bwd op in backward (so double tracing sees ops, too).
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;
struct GwcVolumeFn : torch::autograd::Function<GwcVolumeFn> {
    static at::Tensor forward(AutogradContext* ctx, const at::Tensor& l, const at::Tensor& r, int64_t maxdisp, int64_t groups) {
        at::AutoDispatchBelowADInplaceOrView g;
        ctx->save_for_backward({l, r});
        ctx->saved_data["maxdisp"] = maxdisp; ctx->saved_data["groups"] = groups;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("osa_native::gwc_volume", "").typed<at::Tensor(const at::Tensor&, const at::Tensor&, int64_t, int64_t)>();
        return op.call(l, r, maxdisp, groups);
    }
    static variable_list backward(AutogradContext* ctx, variable_list gy) {
        const auto sv = ctx->get_saved_variables();
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("osa_native::volume_bwd", "")
            .typed<std::tuple<at::Tensor, at::Tensor>(const at::Tensor&, const c10::optional<at::Tensor>&, const c10::optional<at::Tensor>&, at::IntArrayRef, int64_t, int64_t, bool, bool)>();
        auto [dl, dr] = op.call(gy[0].to(at::kFloat), sv[0].to(at::kFloat), sv[1].to(at::kFloat), sv[0].sizes(), ctx->saved_data["maxdisp"].toInt(), ctx->saved_data["groups"].toInt(), false, true);
        return {dl.to(sv[0].scalar_type()), dr.to(sv[1].scalar_type()), at::Tensor(), at::Tensor()};
    }
};
at::Tensor gwc_volume_autograd(const at::Tensor& l, const at::Tensor& r, int64_t maxdisp, int64_t groups) { return GwcVolumeFn::apply(l, r, maxdisp, groups); }

struct ConcatVolumeFn : torch::autograd::Function<ConcatVolumeFn> {
    static at::Tensor forward(AutogradContext* ctx, const at::Tensor& l, const at::Tensor& r, int64_t maxdisp, bool mask_left) {
        at::AutoDispatchBelowADInplaceOrView g;
        ctx->saved_data["shape"] = l.sizes().vec(); ctx->saved_data["maxdisp"] = maxdisp; ctx->saved_data["mask_left"] = mask_left;
        ctx->saved_data["dtype"] = (int64_t)l.scalar_type();
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("osa_native::concat_volume", "").typed<at::Tensor(const at::Tensor&, const at::Tensor&, int64_t, bool)>();
        return op.call(l, r, maxdisp, mask_left);
    }
    static variable_list backward(AutogradContext* ctx, variable_list gy) {
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("osa_native::volume_bwd", "")
            .typed<std::tuple<at::Tensor, at::Tensor>(const at::Tensor&, const c10::optional<at::Tensor>&, const c10::optional<at::Tensor>&, at::IntArrayRef, int64_t, int64_t, bool, bool)>();
        const auto shape = ctx->saved_data["shape"].toIntVector();
        auto [dl, dr] = op.call(gy[0].to(at::kFloat), c10::nullopt, c10::nullopt, shape, ctx->saved_data["maxdisp"].toInt(), 0, true, ctx->saved_data["mask_left"].toBool());
        const auto dt = (at::ScalarType)ctx->saved_data["dtype"].toInt();
        return {dl.to(dt), dr.to(dt), at::Tensor(), at::Tensor()};
    }
};
at::Tensor concat_volume_autograd(const at::Tensor& l, const at::Tensor& r, int64_t maxdisp, bool mask_left) { return ConcatVolumeFn::apply(l, r, maxdisp, mask_left); }

struct CorrVolumeFn : torch::autograd::Function<CorrVolumeFn> {
    static at::Tensor forward(AutogradContext* ctx, const at::Tensor& l, const at::Tensor& r, int64_t maxdisp) {
        at::AutoDispatchBelowADInplaceOrView g;
        ctx->save_for_backward({l, r});
        ctx->saved_data["maxdisp"] = maxdisp;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("osa_native::corr_volume", "").typed<at::Tensor(const at::Tensor&, const at::Tensor&, int64_t)>();
        return op.call(l, r, maxdisp);
    }
    static variable_list backward(AutogradContext* ctx, variable_list gy) {
        const auto sv = ctx->get_saved_variables();
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("osa_native::volume_bwd", "")
            .typed<std::tuple<at::Tensor, at::Tensor>(const at::Tensor&, const c10::optional<at::Tensor>&, const c10::optional<at::Tensor>&, at::IntArrayRef, int64_t, int64_t, bool, bool)>();
        auto [dl, dr] = op.call(gy[0].to(at::kFloat).unsqueeze(1), sv[0].to(at::kFloat), sv[1].to(at::kFloat), sv[0].sizes(), ctx->saved_data["maxdisp"].toInt(), 1, false, true);
        return {dl.to(sv[0].scalar_type()), dr.to(sv[1].scalar_type()), at::Tensor()};
    }
};
at::Tensor corr_volume_autograd(const at::Tensor& l, const at::Tensor& r, int64_t maxdisp) { return CorrVolumeFn::apply(l, r, maxdisp); }

struct SoftargminFn : torch::autograd::Function<SoftargminFn> {
    static at::Tensor forward(AutogradContext* ctx, const at::Tensor& p) {
        at::AutoDispatchBelowADInplaceOrView g;
        ctx->saved_data["D"] = p.size(1); ctx->saved_data["dtype"] = (int64_t)p.scalar_type();
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("osa_native::softargmin", "").typed<at::Tensor(const at::Tensor&)>();
        return op.call(p);
    }
    static variable_list backward(AutogradContext* ctx, variable_list gy) {
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("osa_native::softargmin_bwd", "").typed<at::Tensor(const at::Tensor&, int64_t)>();
        return {op.call(gy[0].to(at::kFloat), ctx->saved_data["D"].toInt()).to((at::ScalarType)ctx->saved_data["dtype"].toInt())};
    }
};
at::Tensor softargmin_autograd(const at::Tensor& p) { return SoftargminFn::apply(p); }

struct SoftmaxSoftargminFn : torch::autograd::Function<SoftmaxSoftargminFn> {
    static variable_list forward(AutogradContext* ctx, const at::Tensor& c, bool return_prob) {
        at::AutoDispatchBelowADInplaceOrView g;
        ctx->save_for_backward({c});
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("osa_native::softmax_softargmin", "").typed<std::tuple<at::Tensor, at::Tensor>(const at::Tensor&, bool)>();
        auto [out, prob] = op.call(c, return_prob);
        ctx->mark_non_differentiable({prob});                 // (the probability volume is a by-product for inspection: gradients flow through `out`)
        return {out, prob};
    }
    static variable_list backward(AutogradContext* ctx, variable_list gy) {
        const auto c = ctx->get_saved_variables()[0];
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("osa_native::softmax_softargmin_bwd", "").typed<at::Tensor(const at::Tensor&, const at::Tensor&)>();
        return {op.call(c.to(at::kFloat), gy[0].to(at::kFloat)).to(c.scalar_type()), at::Tensor()};
    }
};
std::tuple<at::Tensor, at::Tensor> softmax_softargmin_autograd(const at::Tensor& c, bool return_prob) {
    auto r = SoftmaxSoftargminFn::apply(c, return_prob);
    return {r[0], r[1]};
}

struct UpsampleSoftargminFn : torch::autograd::Function<UpsampleSoftargminFn> {
    static at::Tensor forward(AutogradContext* ctx, const at::Tensor& c, int64_t maxdisp, int64_t h, int64_t w, bool align) {
        at::AutoDispatchBelowADInplaceOrView g;
        ctx->save_for_backward({c});
        ctx->saved_data["maxdisp"] = maxdisp; ctx->saved_data["h"] = h; ctx->saved_data["w"] = w; ctx->saved_data["align"] = align;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("osa_native::upsample_softargmin", "").typed<at::Tensor(const at::Tensor&, int64_t, int64_t, int64_t, bool)>();
        return op.call(c, maxdisp, h, w, align);
    }
    static variable_list backward(AutogradContext* ctx, variable_list gy) {
        const auto c = ctx->get_saved_variables()[0];
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("osa_native::upsample_softargmin_bwd", "")
            .typed<at::Tensor(const at::Tensor&, const at::Tensor&, int64_t, int64_t, int64_t, bool)>();
        auto dc = op.call(c.to(at::kFloat), gy[0].to(at::kFloat), ctx->saved_data["maxdisp"].toInt(), ctx->saved_data["h"].toInt(), ctx->saved_data["w"].toInt(),
                          ctx->saved_data["align"].toBool());
        return {dc.to(c.scalar_type()), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};
at::Tensor upsample_softargmin_autograd(const at::Tensor& c, int64_t maxdisp, int64_t h, int64_t w, bool align) { return UpsampleSoftargminFn::apply(c, maxdisp, h, w, align); }

int64_t abi_version() { return osa_abi_version(); }

// ---- Meta (FakeTensor / torch.export / torch.compile) kernels of the launch ops (r6) ------------------------------------------------------
// The in-place launch ops (`Tensor(a!) ... -> ()`) compute nothing a shape pass needs: their outputs are allocated by the caller.  One boxed
// kernel serves all of them: pop the arguments, push nothing.
void noop_boxed(const c10::OperatorHandle& op, torch::jit::Stack* stack) {
    TORCH_INTERNAL_ASSERT(op.schema().returns().empty(), "noop_boxed serves ops without results only: ", op.schema().name());
    torch::jit::drop(*stack, op.schema().arguments().size());
}
// cost_volume_cl: the volume's shape and the split decision (the eligibility test is host arithmetic on sizes and alignments: device
// allocations are at least 256-byte aligned, so aligned stand-in addresses give the answer the CUDA kernel gives)
std::tuple<at::Tensor, bool> cost_volume_cl_meta(const at::Tensor& gwc_feat, const c10::optional<at::Tensor>& cat_feat, int64_t B, int64_t num_groups, int64_t maxdisp,
                                                 int64_t gwc_channels, int64_t cat_channels, int64_t gwc_off, bool mask_left, bool out_split,
                                                 const c10::optional<at::Tensor>& gwc_meta, const c10::optional<at::Tensor>& cat_meta, const at::Tensor& out_meta) {
    TORCH_CHECK(gwc_feat.dim() == 5 && gwc_feat.size(0) == 2 * B && gwc_feat.size(2) == 1, "cost_volume_cl: gwc_feat must be NHWC [2B, Cs, 1, H, W]");
    const int64_t Gs = gwc_feat.size(1), H = gwc_feat.size(3), W = gwc_feat.size(4);
    const int64_t C = gwc_channels >= 0 ? gwc_channels : Gs - gwc_off;
    int64_t Cc = 0, cs = 0;
    const bool has_cat = cat_feat.has_value() && cat_feat->defined();
    if (has_cat) { cs = cat_feat->size(1); Cc = cat_channels >= 0 ? cat_channels : cs; }
    const int64_t nch = num_groups + 2 * Cc, VC = (nch + 3) / 4 * 4;
    auto out = at::empty({B, maxdisp, H, W, VC}, gwc_feat.options()).permute({0, 4, 1, 2, 3});
    const float* al = reinterpret_cast<const float*>(static_cast<uintptr_t>(4096));
    float* ao = reinterpret_cast<float*>(static_cast<uintptr_t>(8192));
    const bool gm = gwc_meta.has_value() && gwc_meta->defined(), cm = cat_meta.has_value() && cat_meta->defined();
    const bool split = out_split && VC == nch && (C == 0 || gm) && (Cc == 0 || cm) &&
        osa_build_volume_nhwc_split_eligible(has_cat ? al : nullptr, has_cat ? al : nullptr, ao, (int)C, (int)num_groups, (int)Gs, (int)Cc, (int)cs, (int)VC, 0, (int)W, (int)maxdisp) == 1;
    (void)mask_left; (void)out_meta;
    return {out, split};
}
// conv_wgrad: whether the requested form covers the layer is the workspace query's answer (host arithmetic on dims)
bool conv_wgrad_meta(const at::Tensor& x, const at::Tensor& dy, at::Tensor dw, at::IntArrayRef dims, int64_t prec, const c10::optional<at::Tensor>& x_meta,
                     const c10::optional<at::Tensor>& dy_meta) {
    TORCH_CHECK(dims.size() == 22, "conv_wgrad: dims = [B, D, H, W, Ci, xCs, Do, Ho, Wo, Co, dyCs, kd, kh, kw, stride, pad x3, dil x3, transposed]");
    int d[22];
    for (int i = 0; i < 22; ++i) d[i] = (int)dims[i];
    const size_t need = prec == 0
        ? osa_conv3d_wgrad_workspace_bytes(d[0], d[1], d[2], d[3], d[4], d[6], d[7], d[8], d[9], d[11], d[12], d[13], d[14], d[15], d[16], d[17], d[18], d[19], d[20], d[21])
        : osa_conv3d_wgrad_f16x3_workspace_bytes(d[0], d[1], d[2], d[3], d[4], d[6], d[7], d[8], d[9], d[11], d[12], d[13], d[14], d[15], d[16], d[17], d[18], d[19], d[20], d[21]);
    TORCH_CHECK(need != 0 || prec != 0, "conv_wgrad: unsupported layer");
    (void)x; (void)dy; (void)dw; (void)x_meta; (void)dy_meta;
    return need != 0;
}
bool conv_wgrad_multi_meta(at::TensorList xs, at::TensorList dys, at::Tensor dw, at::IntArrayRef dims, int64_t prec, const c10::optional<at::Tensor>& x_meta,
                           const c10::optional<at::Tensor>& dy_meta) {
    TORCH_CHECK(dims.size() == 22 && xs.size() >= 1 && xs.size() == dys.size(), "conv_wgrad_multi: dims of 22 entries, equally long lists");
    return conv_wgrad_meta(xs[0], dys[0], dw, dims, prec, x_meta, dy_meta);
}

}  // namespace

TORCH_LIBRARY(osa_native, m) {
    m.def("abi_version() -> int", &abi_version);
    m.def("gwc_volume(Tensor left, Tensor right, int maxdisp, int groups) -> Tensor");
    m.def("concat_volume(Tensor left, Tensor right, int maxdisp, bool mask_left=True) -> Tensor");
    m.def("corr_volume(Tensor left, Tensor right, int maxdisp) -> Tensor");
    m.def("softargmin(Tensor prob) -> Tensor");
    m.def("softmax_softargmin(Tensor cost, bool return_prob=False) -> (Tensor, Tensor)");
    m.def("upsample_softargmin(Tensor cost_lowres, int maxdisp, int h, int w, bool align_corners=False) -> Tensor");
    m.def("context_upsample(Tensor disp_low, Tensor up_weights, int scale=4, bool softmax_weights=False, float gain=1.0) -> Tensor");
    m.def("conv_ndhwc(Tensor x, int x_off, Tensor packed, Tensor? scale, Tensor? shift, Tensor? residual, int res_off, Tensor(a!) out, int out_off, "
          "Tensor? gate, int[] dims, int[] geom, int family, int prec, int act, float slope, float out_scale, Tensor[] metas) -> ()");
    // r5: backward kernels, the fused NDHWC builder, the weight gradient
    m.def("volume_bwd(Tensor dvol, Tensor? left, Tensor? right, int[] shape, int maxdisp, int groups, bool concat, bool mask_left) -> (Tensor, Tensor)");
    m.def("softargmin_bwd(Tensor dout, int D) -> Tensor");
    m.def("softmax_softargmin_bwd(Tensor cost, Tensor dout) -> Tensor");
    m.def("upsample_softargmin_bwd(Tensor cost_lowres, Tensor dout, int maxdisp, int h, int w, bool align_corners=False) -> Tensor");
    m.def("cost_volume_cl(Tensor gwc_feat, Tensor? cat_feat, int B, int num_groups, int maxdisp, int gwc_channels, int cat_channels, int gwc_off, bool mask_left, "
          "bool out_split, Tensor? gwc_meta, Tensor? cat_meta, Tensor(a!) out_meta) -> (Tensor, bool)");
    m.def("conv_wgrad(Tensor x, Tensor dy, Tensor(a!) dw, int[] dims, int prec, Tensor? x_meta, Tensor? dy_meta) -> bool");
    m.def("conv_wgrad_multi(Tensor[] xs, Tensor[] dys, Tensor(a!) dw, int[] dims, int prec, Tensor? x_meta, Tensor? dy_meta) -> bool");
    m.def("to_cl(Tensor x, Tensor(a!) y, int C, int S, int c_off) -> ()");
    m.def("to_ncdhw(Tensor x, Tensor(a!) y, int C, int S, int c_off) -> ()");
    m.def("conv_pack(Tensor w, Tensor(a!) packed, int[] geom, int prec, Tensor? w_amax, Tensor(b!)? scale_out) -> ()");
    m.def("deconv_pack(Tensor w, Tensor(a!) packed, int[] geom, int prec, Tensor? w_amax, Tensor(b!)? scale_out) -> ()");
    m.def("gru_gates_rz_fwd(Tensor pre, Tensor? bias_z, Tensor? bias_r, Tensor cz, Tensor cr, Tensor h, Tensor(a!) z, Tensor(b!) rh) -> ()");
    m.def("gru_gates_rz_bwd(Tensor pre, Tensor? bias_z, Tensor? bias_r, Tensor cz, Tensor cr, Tensor h, Tensor dz, Tensor drh, Tensor(a!) dpre, Tensor(b!) dh) -> ()");
    m.def("gru_gates_q_fwd(Tensor z, Tensor qpre, Tensor? bias_q, Tensor cq, Tensor h, Tensor(a!) out) -> ()");
    m.def("gru_gates_q_bwd(Tensor z, Tensor qpre, Tensor? bias_q, Tensor cq, Tensor h, Tensor dout, Tensor(a!) dz, Tensor(b!) dqpre, Tensor(c!) dh) -> ()");
    m.def("geo_lookup(Tensor[] levels, Tensor disp, Tensor coords_x, Tensor(a!) out, int C, int radius) -> ()");
    m.def("geo_lookup_bwd(Tensor(a!)[] dlevels, Tensor disp, Tensor coords_x, Tensor dout, int C, int radius) -> ()");
    m.def("geo_lookup_bwd_acc(Tensor(a!)[] dlevels, Tensor disp, Tensor coords_x, Tensor dout, int C, int radius) -> ()");
    m.def("build_volume(Tensor? left_gwc, Tensor? right_gwc, int groups, Tensor? left_cat, Tensor? right_cat, Tensor(a!) out, int layout, int vol_channels, int c_off, "
          "int maxdisp, bool mask_left, Tensor(b!)? meta) -> ()");
    m.def("deconv_redir(Tensor x, int x_off, Tensor packed, Tensor? scale, Tensor? shift, Tensor(a!) out, int out_off, int[] dims, int[] geom, Tensor rx, int[] rdims, "
          "Tensor rpacked, Tensor? rscale, Tensor? rshift, float r_out_scale, int prec, int act, float slope, float out_scale, Tensor[] metas) -> ()");
    m.def("small_co_conv(Tensor x, Tensor packed, Tensor? bias, Tensor? residual, Tensor(a!) y, int[] dims, int[] geom) -> ()");
    m.def("dwconv2d(Tensor x, Tensor packed, Tensor? scale, Tensor? shift, Tensor? add, Tensor(a!) y, int[] dims, int[] geom, int act, Tensor(b!)? y_meta) -> ()");
    m.def("dwconv2d_f16io(Tensor x, Tensor packed, Tensor? scale, Tensor? shift, Tensor(a!) y, int[] dims, int[] geom, int act, Tensor(b!)? y_meta) -> ()");
    m.def("gru_combine(Tensor z, int z_off, Tensor q, Tensor h, Tensor(a!) out, int[] dims, Tensor(b!)? out_meta) -> ()");
    m.def("resample_nhwc(Tensor x, Tensor(a!) y, int y_off, int kind, int[] dims, Tensor? x_meta, Tensor(b!)? y_meta) -> ()");
    m.def("disp_update(Tensor(a!) disp, Tensor? delta, int delta_cs, Tensor(b!) disp4, Tensor(c!) slot, int slot_off, int slot_cs, int npix, Tensor(d!)? disp4_meta, "
          "Tensor(e!)? slot_meta) -> ()");
    m.def("geo_lookup_nhwc(Tensor[] levels, Tensor disp, Tensor coords_x, Tensor(a!) out, int out_cs, int[] bhw, int C, int radius) -> ()");
    m.def("allpairs_corr(Tensor fmap1, Tensor fmap2, Tensor(a!) corr) -> ()");
    m.def("geo_rows(Tensor volume, Tensor(a!) rows, int C) -> ()");
    m.def("avgpool_rows(Tensor x, Tensor(a!) y) -> ()");
    m.def("weight_pack(Tensor w, Tensor(a!) packed, int family, int prec, int[] geom, float wscale) -> ()");
    m.def("cat_fms(Tensor reference_fm, Tensor target_fm, Tensor(a!) out, Tensor disp_index) -> ()");
    m.def("pair_volume(Tensor left, Tensor right, Tensor(a!) out, int groups, int planes, int mode) -> ()");
    m.def("channel_sums(Tensor dy, Tensor? x, Tensor? x_shift, Tensor? dx_scale, int P, int C, int dy_cs, int x_cs) -> (Tensor, Tensor)");
    m.def("context_upsample_logits(Tensor disp_low, Tensor logits, int scale, float gain) -> Tensor");
    m.def("context_upsample_logits_bwd(Tensor disp_low, Tensor logits, Tensor dout, int scale, float gain) -> (Tensor, Tensor)");
    m.def("channel_affine(Tensor u, Tensor? v, Tensor a, Tensor? b, Tensor c0, int P, int C, int u_cs, int v_cs, bool relu) -> Tensor");
    m.def("channel_sums_multi(Tensor[] dys, int P, int C, int dy_cs) -> Tensor");
    m.def("instnorm_nhwc(Tensor x, Tensor(a!) out, int out_off, int[] dims, float eps, int act, float slope, Tensor(b!) workspace, Tensor(c!)? y_meta) -> ()");
    m.def("preprocess_pair(Tensor left_hwc, Tensor right_hwc, Tensor(a!) out, int[] pad_size, float[] mean, float[] std, bool channels_last) -> ()");
    m.def("amax_into(Tensor t, Tensor(a!) meta) -> ()");
}

TORCH_LIBRARY_IMPL(osa_native, CUDA, m) {        // (the HIP backend registers under PyTorch's CUDA dispatch key)
    m.impl("gwc_volume", &gwc_volume);
    m.impl("concat_volume", &concat_volume);
    m.impl("corr_volume", &corr_volume);
    m.impl("softargmin", &softargmin);
    m.impl("softmax_softargmin", &softmax_softargmin);
    m.impl("upsample_softargmin", &upsample_softargmin);
    m.impl("context_upsample", &context_upsample);
    m.impl("conv_ndhwc", &conv_ndhwc);
    m.impl("volume_bwd", &volume_bwd);
    m.impl("softargmin_bwd", &softargmin_bwd);
    m.impl("softmax_softargmin_bwd", &softmax_softargmin_bwd);
    m.impl("upsample_softargmin_bwd", &upsample_softargmin_bwd);
    m.impl("cost_volume_cl", &cost_volume_cl);
    m.impl("conv_wgrad", &conv_wgrad);
    m.impl("conv_wgrad_multi", &conv_wgrad_multi);
    m.impl("to_cl", &to_cl);
    m.impl("to_ncdhw", &to_ncdhw);
    m.impl("conv_pack", &conv_pack);
    m.impl("deconv_pack", &deconv_pack);
    m.impl("gru_gates_rz_fwd", &gru_gates_rz_fwd);
    m.impl("gru_gates_rz_bwd", &gru_gates_rz_bwd);
    m.impl("gru_gates_q_fwd", &gru_gates_q_fwd);
    m.impl("gru_gates_q_bwd", &gru_gates_q_bwd);
    m.impl("geo_lookup", &geo_lookup);
    m.impl("geo_lookup_bwd", &geo_lookup_bwd);
    m.impl("geo_lookup_bwd_acc", &geo_lookup_bwd_acc);
    m.impl("build_volume", &build_volume);
    m.impl("deconv_redir", &deconv_redir);
    m.impl("small_co_conv", &small_co_conv);
    m.impl("dwconv2d", &dwconv2d);
    m.impl("dwconv2d_f16io", &dwconv2d_f16io);
    m.impl("gru_combine", &gru_combine);
    m.impl("resample_nhwc", &resample_nhwc);
    m.impl("disp_update", &disp_update);
    m.impl("geo_lookup_nhwc", &geo_lookup_nhwc);
    m.impl("allpairs_corr", &allpairs_corr);
    m.impl("geo_rows", &geo_rows);
    m.impl("avgpool_rows", &avgpool_rows);
    m.impl("weight_pack", &weight_pack);
    m.impl("cat_fms", &cat_fms);
    m.impl("pair_volume", &pair_volume);
    m.impl("instnorm_nhwc", &instnorm_nhwc);
    m.impl("channel_sums", &channel_sums);
    m.impl("channel_sums_multi", &channel_sums_multi);
    m.impl("channel_affine", &channel_affine);
    m.impl("context_upsample_logits", &context_upsample_logits);
    m.impl("context_upsample_logits_bwd", &context_upsample_logits_bwd);
    m.impl("preprocess_pair", &preprocess_pair);
    m.impl("amax_into", &amax_into);
}

TORCH_LIBRARY_IMPL(osa_native, Meta, m) {        // shape / dtype inference without a device: FakeTensor, torch.export, torch.compile
    m.impl("gwc_volume", &gwc_volume_meta);
    m.impl("concat_volume", &concat_volume_meta);
    m.impl("corr_volume", &corr_volume_meta);
    m.impl("softargmin", &softargmin_meta);
    m.impl("softmax_softargmin", &softmax_softargmin_meta);
    m.impl("upsample_softargmin", &upsample_softargmin_meta);
    m.impl("context_upsample", &context_upsample_meta);
    m.impl("volume_bwd", &volume_bwd_meta);
    m.impl("softargmin_bwd", &softargmin_bwd_meta);
    m.impl("softmax_softargmin_bwd", &softmax_softargmin_bwd_meta);
    m.impl("upsample_softargmin_bwd", &upsample_softargmin_bwd_meta);
    // r6: every launch op -- an engine model traces under FakeTensorMode / make_fx without a kernel running (tests/test_gpu_fake_trace.py)
    m.impl("cost_volume_cl", &cost_volume_cl_meta);
    m.impl("conv_wgrad", &conv_wgrad_meta);
    m.impl("conv_wgrad_multi", &conv_wgrad_multi_meta);
    m.impl("channel_sums", &channel_sums_meta);
    m.impl("channel_sums_multi", &channel_sums_multi_meta);
    m.impl("channel_affine", &channel_affine_meta);
    m.impl("context_upsample_logits", &context_upsample_logits_meta);
    m.impl("context_upsample_logits_bwd", &context_upsample_logits_bwd_meta);
    for (const char* name : {"conv_ndhwc", "to_cl", "to_ncdhw", "conv_pack", "deconv_pack", "gru_gates_rz_fwd", "gru_gates_rz_bwd", "gru_gates_q_fwd", "gru_gates_q_bwd",
                             "geo_lookup", "geo_lookup_bwd", "geo_lookup_bwd_acc", "build_volume", "deconv_redir", "small_co_conv", "dwconv2d", "dwconv2d_f16io", "gru_combine", "resample_nhwc", "disp_update",
                             "geo_lookup_nhwc", "allpairs_corr", "geo_rows", "avgpool_rows", "weight_pack", "cat_fms", "pair_volume", "instnorm_nhwc", "preprocess_pair",
                             "amax_into"})
        m.impl(name, torch::CppFunction::makeFromBoxedFunction<&noop_boxed>());
}

TORCH_LIBRARY_IMPL(osa_native, Autograd, m) {    // differentiable in C++: backward = the engine's *_bwd kernels (no Python autograd.Function involved)
    m.impl("gwc_volume", &gwc_volume_autograd);
    m.impl("concat_volume", &concat_volume_autograd);
    m.impl("corr_volume", &corr_volume_autograd);
    m.impl("softargmin", &softargmin_autograd);
    m.impl("softmax_softargmin", &softmax_softargmin_autograd);
    m.impl("upsample_softargmin", &upsample_softargmin_autograd);
}
