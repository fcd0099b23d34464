"""Parallel plumbing for the hot path.  Across GPUs: inference shards by independent stereo pairs -- one process per GPU, no data-path
collective; the only collective is the timing MAX.  Inside a GPU: independent sub-batches on concurrent HIP streams."""
from __future__ import annotations

import torch


def shard_pairs(n_pairs: int, rank: int, world: int) -> list[int]:
    """Strided split of a list of pairs (what DistributedSampler(shuffle=False) does,
    stereo/datasets/__init__.py:64-65)."""
    return list(range(rank, n_pairs, world))


def reduce_step_time(local_seconds: float, device: torch.device) -> float:
    """Max over ranks of the timed region (RCCL on GPUs, gloo in the CPU tests)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(local_seconds)
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_rate(pairs_per_rank_per_step: int, steps: int, world: int, seconds: float) -> float:
    return world * pairs_per_rank_per_step * steps / seconds


class SubBatchStreams:
    """Run a batch as `n` independent sub-batches on concurrent HIP streams (fork from / join into the current stream, so the whole thing
    is capturable in one hipGraph).  Every launch of the hot path ends in a tail where the last workgroups leave most CUs idle (a 64-channel
    backbone layer at 4 pairs is only ~5 rounds of resident workgroups); with two sub-batches in flight the tail of one launch overlaps
    the head of the other stream's.  Measured on GwcNet 544x960, 8 pairs per step: 1 stream 176.2-177.3, 2 streams 180.2-180.6, 4
    streams ~177 pairs/s (profiles/round3/ab_substreams_*.txt); starting sub-batch i + 1 when sub-batch i leaves its 2-D backbone, so that
    an HBM-leaning stage always runs beside an MFMA-bound one, gains nothing (the step is energy-bound, profiles/DESIGN_rounds1-5.md 3.2c)."""

    def __init__(self, n: int):
        self.n = max(1, int(n))
        self.streams = [torch.cuda.Stream() for _ in range(self.n)] if self.n > 1 else []
        self._warm = False          # the first call runs its sub-batches chained (see __call__); rearm() asks for another chained call
        if self.n > 1:
            from . import engine
            engine.MULTI_STREAM = True      # packed forms built on one stream are event-ordered before their use on another (engine.cached_pack)

    def rearm(self):
        """Run the next call chained again (after swapping the model / its weights for objects that build their engine state lazily)."""
        self._warm = False

    def __call__(self, fn, *batched):
        """fn(*sub_batch_tensors) -> tensor; `batched`: tensors with the pairs on dim 0 (size divisible by n).  Returns the concatenation.

        Process-global engine state that is built lazily on first use (packed weights, f16x3 range arenas, workspaces) is created on
        whatever stream gets there first.  Range arenas are per stream (ranges.new_meta) and packed forms carry an event
        (engine.cached_pack); on top of that the FIRST call of this object (and the first after rearm()) runs its sub-batches one after the other (stream i + 1 waits
        for stream i), so anything else a model builds on first use is complete before a second stream touches it.  Later calls --
        and the hipGraph captured from them -- run the sub-batches concurrently."""
        if self.n == 1:
            return fn(*batched)
        B = batched[0].shape[0]
        assert B % self.n == 0, f"batch of {B} pairs does not split into {self.n} sub-batches"
        per, cur, outs = B // self.n, torch.cuda.current_stream(), []
        chain = not self._warm
        prev = cur
        for i, st in enumerate(self.streams):
            st.wait_stream(prev if chain else cur)
            with torch.cuda.stream(st):
                outs.append(fn(*[t[i * per:(i + 1) * per] for t in batched]))
            prev = st
        for st in self.streams:
            cur.wait_stream(st)
        self._warm = True
        return torch.cat(outs, 0)
