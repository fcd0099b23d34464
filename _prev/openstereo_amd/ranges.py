"""f16x3 operand ranges: the per-tensor "range blocks" the split-precision conv kernels scale their operands by.

Layout and arithmetic are documented with `osa_f16x3_ranges` in include/openstereo_amd.h."""
from __future__ import annotations

import os

import torch

# ----------------------------------------------------------------------------- f16x3 operand ranges
# Every activation tensor that f16x3 layers touch carries a 16-float "range block" in device memory
# (`t._osa_meta`; layout in include/openstereo_amd.h, osa_f16x3_ranges): running max |value| in 8 slots, folded in
# by each producing workgroup with an atomic max, [1] = the power-of-two scale of a split tensor's halves.  The
# consumer derives its operand scale from it ON THE DEVICE, so nothing here synchronises with the host and
# the whole chain can be captured in a hipGraph.  Blocks are slices of zero-filled arenas; an arena is never
# reused (a slot is handed out once), and stream capture gets an arena of its own so that the captured
# zero-fill is replayed with the graph.
META_FLOATS = 128      # OSA_META_FLOATS: max |value| in 8 slots at [0], [16], ... [112]; [1] = scale of a split tensor
_ARENA_SLOTS = 256
# Arenas are keyed by (device, raw stream handle): a slot is only ever handed to work on the stream its arena's zero-fill
# was issued on, so the memset is ordered before every producer / consumer of the block by stream order alone -- also inside a captured
# graph with forked branches (SubBatchStreams: the memset node of a branch's arena sits in that branch).  r3 shared one arena between
# streams and only re-examined the capture state on a stream change: an arena zeroed on sub-stream 1 could hand slots to sub-stream 2
# with no dependency on the memset.
_arenas = {}         # (device, stream handle) -> [tensor, next free slot, allocated during stream capture?]
_last = None         # (stream handle, device, arena) of the previous call: the common case is one dict-free comparison


def new_meta(device, stream=None) -> torch.Tensor:
    """A fresh zeroed range block on `device` for work on `stream` (raw handle; default: the current stream).  Stream capture runs on
    streams of its own and gets arenas of its own (allocated from the graph's pool, zero-filled by a captured memset that replays with
    the graph); an arena made outside capture is never handed out inside one and vice versa.  (A block from a pre-capture arena baked
    into a graph would never be re-zeroed: its maximum would cover every replay so far.)"""
    global _last
    if stream is None:
        stream = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
    # (the capture state is re-examined on every call: SubBatchStreams uses the SAME side streams in its eager warm-up and inside the
    # capture, so a stream handle alone does not tell)
    cap = torch.cuda.is_current_stream_capturing() if device.type == "cuda" else False
    l = _last
    if l is not None and l[0] == stream and l[1] == device and l[2][1] < _ARENA_SLOTS and l[2][2] == cap:
        a = l[2]
    else:
        key = (device, stream)
        a = _arenas.get(key)
        if a is None or a[1] >= _ARENA_SLOTS or a[2] != cap:
            a = _arenas[key] = [torch.zeros(_ARENA_SLOTS * META_FLOATS, device=device, dtype=torch.float32), 0, cap]
        _last = (stream, device, a)
    i = a[1]
    a[1] = i + 1
    return a[0][i * META_FLOATS:(i + 1) * META_FLOATS]


def reset_arenas():
    """Forget every arena (handed-out blocks stay alive through their tensors).  Call between a warm-up and a stream capture, or after a
    captured graph is destroyed, to drop arenas that belong to dead streams / pools."""
    global _last
    _arenas.clear()
    _last = None


# osa_amax_f32 (csrc/layout.hip) is exact and ~1.7x faster than torch's infinity norm, also inside replayed hipGraphs in isolation
# (tools/diag_amax_graph.py) -- but with it on FORWARD tensors the captured GwcNet training step replays NaN from its second replay on
# (eager steps, StereoBase's captured step and backward-only use are fine; running torch's reduction next to it cures it; cause not found:
# tools/diag_train_nan.py, profiles/round3/diag/amax_kernel_training_graph.txt).  So the default stays torch's reduction;
# OSA_ENGINE_AMAX=1 (or ranges.ENGINE_AMAX = True) selects the kernel.
ENGINE_AMAX = bool(os.environ.get("OSA_ENGINE_AMAX"))


def _dense(t) -> bool:
    """storage of `t` is exactly numel() elements from data_ptr() (any permutation of a contiguous layout)"""
    if t.is_contiguous():
        return True
    try:
        from torch._prims_common import is_non_overlapping_and_dense
        return bool(is_non_overlapping_and_dense(t))
    except Exception:
        return False


def _amax_into(m, t):
    """max |t| into the (fresh) range block `m` by ONE reduction, no temporaries: torch's infinity norm, or -- ENGINE_AMAX -- the engine's
    own kernel for dense fp32 CUDA tensors (float4 grid-stride loads, one atomic per workgroup, the block's 8 slots)."""
    t = t.detach()
    if ENGINE_AMAX and t.is_cuda and t.dtype == torch.float32 and t.numel() > 0 and (t.data_ptr() & 15) == 0 and _dense(t):
        from . import _ext, _lib
        ext = _ext.load()
        if ext is not None:
            ext.amax_into(t, m)
        else:
            _lib.call("osa_amax_f32", t.data_ptr(), t.numel(), m.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return
    torch.linalg.vector_norm(t, float("inf"), dtype=torch.float32 if t.dtype != torch.float32 else None, out=m[0])   # all dims, no reshape (no copy of strided tensors)


def meta_of(t):
    return None if t is None else getattr(t, "_osa_meta", None)


def input_meta(t) -> torch.Tensor:
    """Range block for an operand of an engine call.  Engine-produced tensors carry theirs; for anything else
    (torch ops, user input) max |t| is computed NOW by torch (device side, no host sync) into a fresh block that is
    NOT cached on the tensor -- the caller may overwrite the tensor in place before the next call (static hipGraph
    inputs), and a captured graph must contain the reduction."""
    m = getattr(t, "_osa_meta", None)
    if m is None:
        m = new_meta(t.device)
        _amax_into(m, t)
    return m


def ensure_meta(t) -> torch.Tensor:
    """Like input_meta, but the block is attached to `t`: for tensors the caller has just created and will not
    modify (several engine layers then share one reduction)."""
    m = getattr(t, "_osa_meta", None)
    if m is None:
        m = new_meta(t.device)
        _amax_into(m, t)
        t._osa_meta = m
    return m


def fold_amax(t, values):
    """`values` were written into the engine buffer `t` by torch ops (slice assignment): fold their max |.| into
    t's range block, as an engine producer would have done."""
    m = getattr(t, "_osa_meta", None)
    if m is None:
        m = new_meta(t.device)
        t._osa_meta = m
    m[0:1] = torch.maximum(m[0:1], values.detach().abs().amax().reshape(1).float())
    return t


def amax_of(meta) -> torch.Tensor:
    """max |value| recorded in a range block (device tensor, shape [1])."""
    return meta[0:META_FLOATS:16].amax().reshape(1)


def attach_meta(t, stream=None):
    """Give an engine-allocated output buffer a fresh (zero) range block if it has none."""
    m = getattr(t, "_osa_meta", None)
    if m is None:
        m = new_meta(t.device, stream)
        t._osa_meta = m
    return m


def combine_meta(*metas) -> torch.Tensor:
    """Range block covering several tensors (a concatenation): slot-wise maximum -- one tiny elementwise kernel per pair,
    no reduction over the data."""
    out = metas[0]
    for m in metas[1:]:
        out = torch.maximum(out, m)
    return out


def inherit_meta(dst, src):
    """`dst` holds convex combinations / copies of `src`'s values (pooling, bilinear resampling, layout change, clone):
    max |dst| <= max |src|, so src's block is a valid (shared, read-only) range block for dst."""
    m = getattr(src, "_osa_meta", None)
    if m is not None:
        dst._osa_meta = m
    return dst
