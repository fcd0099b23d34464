"""CPU, 2 processes over gloo: the multi-GPU path of bench.py (pair sharding, barrier, max-over-ranks
timing, whole-job throughput) with the GPU forward replaced by a stub.  Inference has no data-path
collective, so this is the whole N>1 logic."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from openstereo_amd.parallel import shard_pairs, reduce_step_time, whole_job_rate


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_pairs(10, rank, world)
    # every pair is owned by exactly one rank
    owned = [torch.zeros(10, dtype=torch.int64) for _ in range(world)]
    flags = torch.zeros(10, dtype=torch.int64)
    flags[mine] = 1
    dist.all_gather(owned, flags)
    total = torch.stack(owned).sum(0)
    local_dt = 0.010 * (rank + 1)                     # rank 1 is the slow one
    dt = reduce_step_time(local_dt, torch.device("cpu"))
    q.put((rank, mine, total.tolist(), dt, whole_job_rate(pairs_per_rank_per_step=3, steps=4, world=world, seconds=dt)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    assert res[0][1] == [0, 2, 4, 6, 8] and res[1][1] == [1, 3, 5, 7, 9]
    assert res[0][2] == [1] * 10                      # disjoint cover
    for r in res:
        assert abs(r[3] - 0.020) < 1e-9               # MAX over ranks
        assert abs(r[4] - 2 * 3 * 4 / 0.020) < 1e-6   # whole-job pairs/s


def test_single_process_degenerates():
    assert shard_pairs(5, 0, 1) == [0, 1, 2, 3, 4]
    assert reduce_step_time(0.5, torch.device("cpu")) == 0.5
    assert whole_job_rate(2, 10, 1, 4.0) == 5.0
