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

int64_t abi_version() { return osa_abi_version(); }

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
}
