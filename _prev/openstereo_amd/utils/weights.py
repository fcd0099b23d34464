"""Deterministic synthetic parameters keyed by parameter name.

No checkpoints exist offline and default torch init gives a uniform softmax
(disparity == (D-1)/2 everywhere), which would make every parity test vacuous.  This generator
produces a "sharpened" but well-conditioned weight set: He-normal conv kernels, randomised
BatchNorm affine + running statistics, and scaled-up classifier heads so the soft-argmin spans
the disparity range.  Values depend only on (seed, parameter name, shape) through numpy's
PCG64, so the reference model (golden generation), this package's model and the CPU oracle all
see bit-identical parameters on any machine.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def synth_tensor(name: str, shape, seed: int = 0, head_gain: float = 12.0, gain: float = 1.2,
                 res_gain: float = 0.35) -> torch.Tensor:
    shape = tuple(shape)
    r = _rng(seed, name)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_mean":
        a = r.normal(0.0, 0.1, shape)
    elif leaf == "running_var":
        a = r.uniform(0.5, 1.5, shape)
    elif len(shape) == 1 and leaf == "weight":      # BatchNorm gamma
        a = r.uniform(0.6, 1.4, shape)
    elif len(shape) == 1 and leaf == "bias":        # BatchNorm beta / conv bias
        a = r.normal(0.0, 0.1, shape)
    else:                                           # conv / deconv kernels
        fan_in = int(np.prod(shape[1:]))
        a = r.normal(0.0, gain * np.sqrt(1.0 / max(fan_in, 1)), shape)
        if len(shape) == 5 and 1 in shape[:2]:      # single-channel classifier head -> sharpen
            a = a * head_gain
        elif _is_residual_tail(name):               # last conv of a residual branch: keep the sum tame
            a = a * res_gain
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _is_residual_tail(name: str) -> bool:
    """Convs whose output is ADDED to a skip path without normalisation of the sum
    (BasicBlock.conv2, dres1's second conv, hourglass conv5/conv6 + redir*): scaled down so the
    un-normalised residual sums do not blow activations up layer after layer."""
    parts = name.split(".")
    if "aggregator" in parts:      # PSMNet: dres1's 2nd conv, hourglass conv2/conv5/conv6 all feed un-normalised sums
        return ".dres1.1." in name or any(p in ("conv2", "conv5", "conv6") for p in parts)
    return ("conv2" in parts and any(p.startswith("layer") for p in parts)) or ".dres1.2." in name \
        or any(p in ("conv5", "conv6", "redir1", "redir2") for p in parts)


def synth_state_dict(model_or_shapes, seed: int = 0, head_gain: float = 12.0, gain: float = 1.2,
                     res_gain: float = 0.35) -> dict:
    """model_or_shapes: an nn.Module (its state_dict() gives names+shapes) or {name: shape}."""
    if hasattr(model_or_shapes, "state_dict"):
        shapes = {k: tuple(v.shape) for k, v in model_or_shapes.state_dict().items()}
    else:
        shapes = dict(model_or_shapes)
    out = {}
    for k, shp in shapes.items():
        if k.endswith("disp_regression.weight"):    # PSMNet's frozen linspace kernel: keep as is
            continue
        out[k] = synth_tensor(k, shp, seed, head_gain, gain, res_gain)
    return out


def synth_images(B: int, H: int, W: int, seed: int = 1, max_shift: float = 48.0):
    """SceneFlow-shaped synthetic pair: smooth random texture, right = left shifted by a smooth
    disparity field, ImageNet-normalised statistics (roughly N(0,1)).  Returns (left, right) fp32
    [B,3,H,W] torch CPU tensors, deterministic in (seed, B, H, W)."""
    r = np.random.default_rng([seed, B, H, W])
    base = r.normal(0.0, 1.0, (B, 3, H, W + int(max_shift) + 8)).astype(np.float32)
    # light smoothing along x so neighbouring disparities correlate
    k = np.array([0.25, 0.5, 0.25], dtype=np.float32)
    sm = base.copy()
    sm[..., 1:-1] = k[0] * base[..., :-2] + k[1] * base[..., 1:-1] + k[2] * base[..., 2:]
    ys = np.linspace(0, 1, H, dtype=np.float32)[:, None]
    xs = np.linspace(0, 1, W, dtype=np.float32)[None, :]
    disp = (0.5 + 0.5 * np.sin(2 * np.pi * (xs * 1.5 + ys))) * max_shift * 0.9
    xi = np.arange(W, dtype=np.float32)[None, :] + np.zeros((H, 1), np.float32)
    left = sm[..., :W]
    src = np.clip(xi + disp, 0, sm.shape[-1] - 2)
    i0 = np.floor(src).astype(np.int64)
    f = (src - i0).astype(np.float32)
    right = np.take_along_axis(sm, np.broadcast_to(i0, (B, 3, H, W)), axis=-1) * (1 - f) + \
        np.take_along_axis(sm, np.broadcast_to(i0 + 1, (B, 3, H, W)), axis=-1) * f
    return torch.from_numpy(np.ascontiguousarray(left)), torch.from_numpy(np.ascontiguousarray(right.astype(np.float32)))
