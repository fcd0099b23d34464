"""torch.autocast contract of the drop-in surface (reference: trainer_template.py:205-230,281 run every forward under
`torch.cuda.amp.autocast(enabled=cfgs.OPTIMIZATION.AMP)`; cfgs/igev/igev_sceneflow_amp.yaml is the AMP config BASELINE configs[4] names).

Arithmetic: inside an fp16 autocast region (inference, no_grad) the engine layers run the native f16 mode (r4, engine.effective_precision:
fp16 operands, one MFMA per product, fp32 accumulate -- what the reference's autocast convolutions compute, a third of the matrix work of
the f16x3 mode); bf16 regions, training and OSA_AUTOCAST_NATIVE=0 keep the global fp32-class mode.  Either way, what a caller of the
reference's functions / modules relies on under autocast is the DTYPE of what comes back (the next torch op promotes against it) and
values within low-precision tolerance of the eager composition.  Rules, derived from PyTorch's CUDA autocast op lists applied to the
reference's code:

  * convolutions (`conv2d/3d`, `conv_transpose2d/3d`) are on the lower-precision list: a module whose forward ends in a convolution
    (+ BatchNorm / activation / residual add, all dtype-preserving) returns the autocast dtype  ->  rule "cast";
  * `softmax`, `sum` are on the fp32 list: `disparity_regression` (torch.sum, disp_regression.py:8-12), the softmax heads and
    `context_upsample` (`.sum(1)`, disp_refinement.py:194-204) return fp32  ->  rule "keep";
  * element-wise products / `mean` / `new_zeros` fills (cost_volume.py:32-105) keep the input dtype  ->  volumes follow their features;
  * `torch.cat` and binary ops promote to the widest input: BasicMotionEncoder's `cat([out, disp])` (update.py:91-92), ConvGRU's
    `(1 - z) * h + z * q` (:44), MobileV2Residual's `x + feat` and AttentionModule's `attn * cost` (lightstereo/aggregation.py:97,45 --
    and through its `redir1` skip block the whole Aggregation) follow the disparity / hidden-state / cost dtype
    ->  rules "enc", "gru", "res", "update".

`contract(rule)` decorates the mirrors' `forward` methods (openstereo_amd/models/*); attach.patch_reference_modules() grafts those
decorated forwards onto the reference's classes, so both surfaces obey it.  Outside an autocast region the wrappers cost one
`torch.is_autocast_enabled` call and change nothing.  tests/test_gpu_autocast.py checks dtype and values against the oracle
restatement run under the same autocast on the same GPU, and backward under `torch.amp.GradScaler`.
"""
from __future__ import annotations

import functools

import torch


def autocast_dtype():
    """dtype of the active CUDA autocast region, None outside one."""
    return torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else None


def _map(o, fn):
    if isinstance(o, torch.Tensor):
        return fn(o)
    if isinstance(o, dict):
        return {k: _map(v, fn) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_map(v, fn) for v in o)
    return o


def _to(dt):
    return lambda t: t.to(dt) if t.is_floating_point() and t.dtype != dt else t


def _promote(o, *dts):
    dt = dts[0]
    for d in dts[1:]:
        dt = torch.promote_types(dt, d)
    return _map(o, _to(dt))


def _rule_cast(dt, mod, args, out):
    return _map(out, _to(dt))


def _rule_gru(dt, mod, args, out):           # ConvGRU.forward(h, cz, cr, cq, *x): (1 - z) * h + z * q, z / q in the autocast dtype
    return _promote(out, dt, args[0].dtype)


def _rule_enc(dt, mod, args, out):           # BasicMotionEncoder.forward(disp, corr): cat([conv out, disp])
    return _promote(out, dt, args[0].dtype)


def _rule_res(dt, mod, args, out):           # MobileV2Residual.forward(x): x + feat when the block has a skip (aggregation.py:97-98)
    return _promote(out, dt, args[0].dtype) if getattr(mod, "use_res_connect", False) else out.to(dt)


def _rule_update(dt, mod, args, out):        # BasicMultiUpdateBlock.forward(net, inp, ...) -> net | (net, mask, delta)
    net_in = args[0]
    fix = lambda lst: [_promote(t, dt, r.dtype) for t, r in zip(lst, net_in)]
    if isinstance(out, tuple):
        n, mask, delta = out
        return fix(n), mask.to(dt), delta.to(dt)
    return fix(out)


def _rule_volume(dt, mod, args, out):            # GwcVolumeCostProcessor.forward(inputs): `refimg_fea.new_zeros` -> the features' dtype
    fd = args[0]["ref_feature"]["gwc_feature"].dtype
    return _map(out, _to(fd))


_RULES = {"cast": _rule_cast, "gru": _rule_gru, "enc": _rule_enc, "res": _rule_res, "update": _rule_update, "volume": _rule_volume}


def contract(rule="cast"):
    fn = _RULES[rule]

    def deco(fwd):
        @functools.wraps(fwd)
        def forward(self, *a, **k):
            out = fwd(self, *a, **k)
            dt = autocast_dtype()
            return out if dt is None else fn(dt, self, a, out)
        forward._osa_autocast_rule = rule
        return forward
    return deco


def conv_out_dtype(x):
    """dtype a torch convolution module returns for input x: the autocast dtype inside a region, x's own outside."""
    return autocast_dtype() or x.dtype
