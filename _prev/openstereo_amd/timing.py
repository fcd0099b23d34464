"""Opt-in per-stage timing with HIP events on the launch stream (used by bench.py's roofline leg).
Disabled by default: the hot path then pays one `is None` test per call."""
from __future__ import annotations

import torch

_rec = None
work = {}          # key -> (algorithmic flops, algorithmic bytes) of ONE launch with that key (filled while timing is enabled)


def enable():
    global _rec
    _rec = []
    return _rec


def disable():
    global _rec
    _rec = None


class span:
    __slots__ = ("key", "e0")

    def __init__(self, *key, flops=None, nbytes=None):
        self.key = key
        if _rec is not None and flops is not None:
            work[key] = (flops, nbytes)

    def __enter__(self):
        if _rec is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()            # current stream == the stream the kernels are launched on
        return self

    def __exit__(self, *exc):
        if _rec is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _rec.append((self.key, self.e0, e1))
        return False


def collect(rec):
    """-> {key: [ms, ...]} ; call after torch.cuda.synchronize()."""
    out = {}
    for key, e0, e1 in rec:
        out.setdefault(key, []).append(e0.elapsed_time(e1))
    disable()
    return out
