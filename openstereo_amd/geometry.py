"""CombinedGeoEncodingVolume on the engine.

Same constructor / call contract as models/stereobase/gru_blocks.py:170-229 and
models/igev/geometry.py:7-66: built once per forward from the matching features and the aggregated
geometry volume, then called once per GRU iteration with the current disparity.  The engine keeps the
volume as per-pixel rows ([B,H,W,C,D], D contiguous) plus an averaged pyramid, and one fused kernel per
iteration produces the reference's [B,(C+1)*(2r+1)*levels,H,W] tensor (replacing 2*levels grid_sample
calls, their coordinate tensors and the concatenations)."""
from __future__ import annotations

import ctypes

import torch

from . import _ext, _lib, ops, timing
from .ops import _f32c, _stream, is_cl


import os

ACC_LOOKUP_BWD = os.environ.get("OSA_LOOKUP_BWD_ACC", "1") != "0"


class _LevelAcc:
    """gradient accumulators of one pyramid (one CombinedGeoEncodingVolume = one forward): zero-filled by the first lookup backward of a
    backward pass, added to by every lookup backward (osa_geo_lookup_bwd_acc_f32), handed to autograd ONCE by _LevelJoin.backward"""
    __slots__ = ("acc",)

    def __init__(self):
        self.acc = None


class _LevelJoin(torch.autograd.Function):
    """Identity on the pyramid levels, applied once when the pyramid is built -- i.e. BEFORE every lookup, so its backward node runs after
    all lookup backward nodes of the running backward pass (autograd's own dependency count: also right for a backward pass that reaches
    only some of the iterations).  The lookups return no level gradients of their own; this node returns the accumulated ones."""

    @staticmethod
    def forward(ctx, state, *levels):
        ctx.state = state
        ctx.set_materialize_grads(False)
        return tuple(t.view_as(t) for t in levels)

    @staticmethod
    def backward(ctx, *grads):
        acc, ctx.state.acc = ctx.state.acc, None
        if acc is None:
            return (None, *grads)
        return (None, *[a if g is None else a + g for a, g in zip(acc, grads)])


class _Lookup(torch.autograd.Function):
    """out = lookup(disp, coords; geo levels, corr levels) with gradients for the levels (osa_geo_lookup_bwd_f32).  The disparity is
    detached in the reference's loop (igev_stereo.py:190, stereobase_gru.py:186), so it gets none.
    The kernels take fp32 rows: inside an autocast region the levels arrive cast to fp32 (custom_fwd) and every tensor is passed through
    `_f32c` -- handing the raw pointer of an fp16 pyramid level to the kernel reads past its end (found by tests/test_gpu_autocast.py)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, d, cx, C, radius, state, *levels):
        d, cx, levels = _f32c(d), _f32c(cx), tuple(_f32c(t) for t in levels)
        ctx.state = state
        L = len(levels) // 2
        geo, corr = levels[:L], levels[L:]
        B, H, W = d.shape
        out = torch.empty((B, (C + 1) * (2 * radius + 1) * L, H, W), device=d.device, dtype=torch.float32)
        ext = _ext.load()
        if ext is not None:                               # PyTorch-ROCm C++ extension: the pyramid as a tensor list
            ext.geo_lookup(list(levels), d, cx, out, C, radius)
            ctx.save_for_backward(d, cx)
            ctx.meta = (C, radius, L, [tuple(t.shape) for t in levels])
            return out
        gp = (ctypes.c_void_p * L)(*[t.data_ptr() for t in geo])
        cp = (ctypes.c_void_p * L)(*[t.data_ptr() for t in corr])
        gl = (ctypes.c_int * L)(*[t.shape[-1] for t in geo])
        cl = (ctypes.c_int * L)(*[t.shape[-1] for t in corr])
        _lib.call("osa_geo_lookup_f32", gp, cp, gl, cl, L, d.data_ptr(), cx.data_ptr(), out.data_ptr(), B, H, W, C, radius, _stream())
        ctx.save_for_backward(d, cx)
        ctx.meta = (C, radius, L, [tuple(t.shape) for t in levels])
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dout):
        d, cx = ctx.saved_tensors
        C, radius, L, shapes = ctx.meta
        B, H, W = d.shape
        ext = _ext.load()
        st = ctx.state
        if st is not None:
            # accumulate into the pyramid's gradient buffers; _LevelJoin delivers them (r6)
            if st.acc is None:
                st.acc = [torch.zeros(s, device=d.device, dtype=torch.float32) for s in shapes]
            acc = st.acc
            if ext is not None:
                ext.geo_lookup_bwd_acc(acc, d, cx, _f32c(dout), C, radius)
            else:
                _lib.call("osa_geo_lookup_bwd_acc_f32", (ctypes.c_void_p * L)(*[t.data_ptr() for t in acc[:L]]), (ctypes.c_void_p * L)(*[t.data_ptr() for t in acc[L:]]),
                          (ctypes.c_int * L)(*[s[-1] for s in shapes[:L]]), (ctypes.c_int * L)(*[s[-1] for s in shapes[L:]]), L,
                          d.data_ptr(), cx.data_ptr(), _f32c(dout).data_ptr(), B, H, W, C, radius, _stream())
            return (None, None, None, None, None, *([None] * len(shapes)))
        grads = [torch.empty(s, device=d.device, dtype=torch.float32) for s in shapes]
        if ext is not None:
            ext.geo_lookup_bwd(grads, d, cx, _f32c(dout), C, radius)
            return (None, None, None, None, None, *grads)
        gp = (ctypes.c_void_p * L)(*[t.data_ptr() for t in grads[:L]])
        cp = (ctypes.c_void_p * L)(*[t.data_ptr() for t in grads[L:]])
        gl = (ctypes.c_int * L)(*[s[-1] for s in shapes[:L]])
        cl = (ctypes.c_int * L)(*[s[-1] for s in shapes[L:]])
        _lib.call("osa_geo_lookup_bwd_f32", gp, cp, gl, cl, L, d.data_ptr(), cx.data_ptr(), _f32c(dout).data_ptr(),
                  B, H, W, C, radius, _stream())
        return (None, None, None, None, None, *grads)


class CombinedGeoEncodingVolume:
    def __init__(self, init_fmap1, init_fmap2, geo_volume, num_levels=2, radius=4):
        assert 1 <= num_levels <= 4
        self.num_levels, self.radius = num_levels, radius
        self.train_path = torch.is_grad_enabled() and any(t.requires_grad for t in (init_fmap1, init_fmap2, geo_volume))
        if self.train_path:
            # Training: the pyramid is built from differentiable torch ops (a permutation, an einsum, two average pools: once per
            # forward), the per-iteration lookup and its gradient run on the engine (_Lookup).
            # fp32 throughout, whatever autocast region surrounds the call: the reference casts its inputs to fp32 here too
            # (`match_left.float()`, igev_stereo.py:184) but its einsum / pooling still run in the autocast dtype; the engine keeps the
            # pyramid in fp32 -- no fp16 overflow of the 96-term correlation sums, and the levels are what _Lookup's kernels read.
            # The pooling is written as a strided add (exactly F.avg_pool2d(x, [1, 2], stride=[1, 2]): (a + b) / 2, a trailing odd
            # element dropped) on fp32 tensors; a memory-access fault first blamed on F.avg_pool1d with fp16 rows turned out to be _Lookup's kernels
            # being handed fp16 levels (no custom_fwd cast at the time; tools/diag_lookup_ac.py) -- the fp32 pyramid is what fixed it.
            with torch.autocast("cuda", enabled=False):
                f1, f2, gv = init_fmap1.float(), init_fmap2.float(), geo_volume.float()
                B, C, D, H, W1 = gv.shape
                self.C, self.shape = C, (B, H, W1)
                rows = gv.permute(0, 3, 4, 1, 2).contiguous()                                 # [B,H,W,C,D]
                corr = torch.einsum("aijk,aijh->ajkh", f1, f2).contiguous()                   # [B,H,W1,W2]
                self.geo_volume_pyramid, self.init_corr_pyramid = [rows], [corr]
                half = lambda t: ((t[..., 0:t.shape[-1] // 2 * 2:2] + t[..., 1::2]) * 0.5).contiguous()
                for _ in range(num_levels - 1):
                    self.geo_volume_pyramid.append(half(self.geo_volume_pyramid[-1]))
                    self.init_corr_pyramid.append(half(self.init_corr_pyramid[-1]))
            self._acc = None
            if ACC_LOOKUP_BWD:
                self._acc = _LevelAcc()
                joined = _LevelJoin.apply(self._acc, *self.geo_volume_pyramid, *self.init_corr_pyramid)
                self.geo_volume_pyramid, self.init_corr_pyramid = list(joined[:num_levels]), list(joined[num_levels:])
            from .ranges import new_meta
            self.meta = new_meta(rows.device)
            self.meta[0:1] = torch.maximum(rows.detach().abs().amax(), corr.detach().abs().amax()).reshape(1)
            return
        f1, f2 = _f32c(init_fmap1), _f32c(init_fmap2)
        B, Cf, H, W1 = f1.shape
        W2 = f2.shape[3]
        dev = f1.device
        corr = torch.empty((B, H, W1, W2), device=dev, dtype=torch.float32)
        ext = _ext.load()
        if ext is not None:
            ext.allpairs_corr(f1, f2, corr)
        else:
            _lib.call("osa_allpairs_corr_f32", f1.data_ptr(), f2.data_ptr(), corr.data_ptr(), B, Cf, H, W1, W2, _stream())
        gv = geo_volume if is_cl(geo_volume) and geo_volume.dtype == torch.float32 else ops.to_cl(geo_volume.float(), pad_to=1)
        _, Cs, D, Hg, Wg = gv.shape
        C = geo_volume.shape[1] if not is_cl(geo_volume) else Cs
        self.C = C
        assert (Hg, Wg) == (H, W1)
        rows = torch.empty((B, H, W1, C, D), device=dev, dtype=torch.float32)
        if ext is not None:
            ext.geo_rows(gv, rows, C)
        else:
            _lib.call("osa_geo_rows_f32", gv.data_ptr(), rows.data_ptr(), B, D, H, W1, C, Cs, _stream())
        self.geo_volume_pyramid, self.init_corr_pyramid = [rows], [corr]
        for _ in range(num_levels - 1):
            g, c = self.geo_volume_pyramid[-1], self.init_corr_pyramid[-1]
            g2 = torch.empty(g.shape[:-1] + (g.shape[-1] // 2,), device=dev, dtype=torch.float32)
            c2 = torch.empty(c.shape[:-1] + (c.shape[-1] // 2,), device=dev, dtype=torch.float32)
            if ext is not None:
                ext.avgpool_rows(g, g2); ext.avgpool_rows(c, c2)
            else:
                _lib.call("osa_avgpool_rows_f32", g.data_ptr(), g2.data_ptr(), g.numel() // g.shape[-1], g.shape[-1], _stream())
                _lib.call("osa_avgpool_rows_f32", c.data_ptr(), c2.data_ptr(), c.numel() // c.shape[-1], c.shape[-1], _stream())
            self.geo_volume_pyramid.append(g2); self.init_corr_pyramid.append(c2)
        L = num_levels
        self._gp = (ctypes.c_void_p * L)(*[t.data_ptr() for t in self.geo_volume_pyramid])
        self._cp = (ctypes.c_void_p * L)(*[t.data_ptr() for t in self.init_corr_pyramid])
        self._gl = (ctypes.c_int * L)(*[t.shape[-1] for t in self.geo_volume_pyramid])
        self._cl = (ctypes.c_int * L)(*[t.shape[-1] for t in self.init_corr_pyramid])
        self.shape = (B, H, W1)
        # range block of every lookup result (f16x3 consumers): taps are convex combinations of volume entries or zero,
        # the coarser pyramid levels are averages -> bounded by max |level 0|; measured once here, not once per iteration
        from .ranges import new_meta
        self.meta = new_meta(dev)
        self.meta[0:1] = torch.maximum(rows.abs().amax(), corr.abs().amax()).reshape(1)

    def __call__(self, disp, coords):
        """disp [B,1,H,W] (quarter-res disparity), coords [B,H,W,1] (x coordinate grid) ->
        [B,(C+1)*(2r+1)*levels,H,W] float32."""
        B, H, W = self.shape
        d, cx = _f32c(disp).reshape(B, H, W), _f32c(coords).reshape(B, H, W)
        if self.train_path:
            return _Lookup.apply(d.detach(), cx.detach(), self.C, self.radius, self._acc, *self.geo_volume_pyramid, *self.init_corr_pyramid)
        out = torch.empty((B, (self.C + 1) * (2 * self.radius + 1) * self.num_levels, H, W), device=d.device, dtype=torch.float32)
        with timing.span("geo_lookup", self.C, self.num_levels, self.radius, H, W):
            _lib.call("osa_geo_lookup_f32", self._gp, self._cp, self._gl, self._cl, self.num_levels,
                      d.data_ptr(), cx.data_ptr(), out.data_ptr(), B, H, W, self.C, self.radius, _stream())
        return out

    def lookup_cl(self, disp, coords):
        """The same lookup as an NHWC engine tensor [B, Cpad4, 1, H, W] (padding channels zero) for the engine's GRU loop
        (osa_geo_lookup_nhwc_f32): what the motion encoder's convc1 reads, without the per-iteration NCHW -> NHWC transpose.
        disp / coords: fp32, [B,1,H,W] / [B,H,W,1] (or any shape with B*H*W elements), contiguous.  Inference path only."""
        assert not self.train_path
        B, H, W = self.shape
        nch = (self.C + 1) * (2 * self.radius + 1) * self.num_levels
        out = ops.empty_cl(B, (nch + 3) // 4 * 4, 1, H, W, disp.device)
        d, cx = _f32c(disp), _f32c(coords)
        assert d.numel() == B * H * W and cx.numel() == B * H * W
        with timing.span("geo_lookup", self.C, self.num_levels, self.radius, H, W):
            ext = _ext.load()
            if ext is not None:
                ext.geo_lookup_nhwc(self.geo_volume_pyramid + self.init_corr_pyramid, d, cx, out, out.shape[1], [B, H, W], self.C, self.radius)
            else:
                _lib.call("osa_geo_lookup_nhwc_f32", self._gp, self._cp, self._gl, self._cl, self.num_levels,
                          d.data_ptr(), cx.data_ptr(), out.data_ptr(), out.shape[1], B, H, W, self.C, self.radius, _stream())
        out._osa_meta = self.meta                  # taps interpolate / zero-pad the volumes: bounded by their max |.|
        return out

    @staticmethod
    def corr(fmap1, fmap2):
        """einsum('aijk,aijh->ajkh') -> [B,H,W1,1,W2] (gru_blocks.py:221-229)."""
        f1, f2 = _f32c(fmap1), _f32c(fmap2)
        B, Cf, H, W1 = f1.shape
        W2 = f2.shape[3]
        out = torch.empty((B, H, W1, 1, W2), device=f1.device, dtype=torch.float32)
        ext = _ext.load()
        if ext is not None:
            ext.allpairs_corr(f1, f2, out)
        else:
            _lib.call("osa_allpairs_corr_f32", f1.data_ptr(), f2.data_ptr(), out.data_ptr(), B, Cf, H, W1, W2, _stream())
        return out


Combined_Geo_Encoding_Volume = CombinedGeoEncodingVolume      # IGEV's name (models/igev/geometry.py:7)
