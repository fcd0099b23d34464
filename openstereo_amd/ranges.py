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
_arena = None        # [tensor, next free slot, allocated during stream capture?, stream handle it was made on]


def new_meta(device, stream=None) -> torch.Tensor:
    """A fresh zeroed range block on `device`.  `stream` (raw handle of the current stream, when the caller has it
    anyway) keys the arena: stream capture runs on a stream of its own, so a change of stream is when the capture
    state is re-examined -- the per-call cost is then one integer comparison.  (A block from a pre-capture arena
    baked into a graph would merely never be re-zeroed: its maximum then covers every replay so far -- still a
    valid bound, the results just stop being independent of history.)"""
    global _arena
    a = _arena
    if a is None or a[1] >= _ARENA_SLOTS or a[0].device != device or stream is None or a[3] != stream:
        cap = torch.cuda.is_current_stream_capturing()
        if a is None or a[1] >= _ARENA_SLOTS or a[0].device != device or a[2] != cap:
            a = _arena = [torch.zeros(_ARENA_SLOTS * META_FLOATS, device=device, dtype=torch.float32), 0, cap, stream]
        else:
            a[3] = stream
    i = a[1]
    a[1] = i + 1
    return a[0][i * META_FLOATS:(i + 1) * META_FLOATS]


# osa_amax_f32 (csrc/layout.hip) is exact and ~1.7x faster than torch's infinity norm, also inside replayed hipGraphs in isolation
# (tools/diag_amax_graph.py) -- but with it on FORWARD tensors the captured GwcNet training step replays NaN from its second replay on
# (eager steps, StereoBase's captured step and backward-only use are fine; running torch's reduction next to it cures it; cause not found:
# tools/diag_train_nan.py, profiles/round3/diag/amax_kernel_training_graph.txt).  So the default stays torch's reduction;
# OSA_ENGINE_AMAX=1 (or ranges.ENGINE_AMAX = True) selects the kernel.
ENGINE_AMAX = bool(os.environ.get("OSA_ENGINE_AMAX"))
DIAG = {"count": 0, "lo": 0, "hi": 1 << 30, "log": None}


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
    if ENGINE_AMAX:
        mode = os.environ.get("OSA_AMAX_MODE", "")          # diagnostics: "fwd" / "bwd" = kernel only outside / inside autograd's backward
        in_bwd = torch._C._current_graph_task_id() != -1
        use = not ((mode == "fwd" and in_bwd) or (mode == "bwd" and not in_bwd))
        if mode == "idx":                                     # diagnostics (tools/diag_train_nan2.py): kernel for forward calls [lo, hi) of a step only
            use = False
            if not in_bwd:
                i = DIAG["count"]; DIAG["count"] = i + 1
                use = DIAG["lo"] <= i < DIAG["hi"]
                if DIAG["log"] is not None:
                    DIAG["log"].append((i, tuple(t.shape), t.stride(), t.data_ptr() % 4096))
        if use and t.is_cuda and t.dtype == torch.float32 and t.numel() > 0 and (t.data_ptr() & 15) == 0 and _dense(t):
            from . import _lib
            if mode == "both":                                    # diagnostics: torch's reduction into a scratch block as well
                torch.linalg.vector_norm(t, float("inf"), out=new_meta(t.device)[0])
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
