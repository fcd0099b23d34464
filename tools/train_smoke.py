"""DDP training smoke for the engine's training path (BASELINE configs[2] shape of work: data-parallel
training, one process per GPU, stock DistributedDataParallel -> bucketed RCCL all-reduce over xGMI).

    python tools/train_smoke.py --steps 3                                   # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           --master-port 29511 tools/train_smoke.py --steps 3               # 8 GPUs

GwcNet (the model whose full graph is available offline), synthetic 256x512 crops (the reference's
training crop, cfgs/gwcnet/gwcnet_sceneflow.yaml:13), RMSprop lr 1e-3 as in the config.  Prints one
JSON line on rank 0 with pairs/s and the loss trajectory.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    args = ap.parse_args()
    rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    from openstereo_amd.models.gwcnet import GwcNet
    from openstereo_amd.utils.weights import synth_state_dict, synth_images
    net = GwcNet()
    net.load_state_dict(synth_state_dict(net, seed=0))
    net = net.to(dev).train()
    model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[local]) if world > 1 else net
    opt = torch.optim.RMSprop(model.parameters(), lr=1e-3)
    L, R = synth_images(args.batch, args.height, args.width, seed=10 + rank)
    L, R = L.to(dev), R.to(dev)
    gt = torch.from_numpy(np.random.default_rng(rank).uniform(1, 100, (args.batch, args.height, args.width)).astype(np.float32)).to(dev)
    losses = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        opt.zero_grad(set_to_none=True)
        out = model({"left": L, "right": R})
        loss, _ = net.get_loss(out, {"disp": gt})
        loss.backward()                       # DDP all-reduces gradient buckets over RCCL here
        opt.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"train_pairs_per_s": round(world * args.batch * args.steps / dt, 3), "n_gpus": world,
                          "steps": args.steps, "crop": [args.height, args.width], "losses": [round(x, 4) for x in losses]}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
