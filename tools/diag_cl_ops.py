"""Diagnostic (GPU): backward of stock PyTorch-ROCm ops for channels-last inputs / incoming gradients vs the same op on contiguous tensors.
(PyTorch 2.10 + ROCm 7.0: avg_pool2d backward was found wrong for channels-last inputs; this sweeps the other ops the training path uses.)"""
import torch, torch.nn.functional as F
torch.manual_seed(0)
ops = {
    "avg_pool2d 3/2/1": lambda x: F.avg_pool2d(x, 3, stride=2, padding=1),
    "avg_pool2d 2/2": lambda x: F.avg_pool2d(x, 2, stride=2),
    "interp bilinear ac=True x2": lambda x: F.interpolate(x, size=(x.shape[2] * 2, x.shape[3] * 2), mode="bilinear", align_corners=True),
    "interp bilinear ac=False x2": lambda x: F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False),
    "interp bilinear ac=False x4": lambda x: F.interpolate(x, scale_factor=4, mode="bilinear", align_corners=False),
    "interp nearest x2": lambda x: F.interpolate(x, scale_factor=2, mode="nearest"),
    "instance_norm": lambda x: F.instance_norm(x),
    "batch_norm eval": lambda x: F.batch_norm(x, torch.zeros(x.shape[1], device=x.device), torch.ones(x.shape[1], device=x.device) * 2, training=False),
    "batch_norm train": lambda x: F.batch_norm(x, None, None, training=True),
    "max_pool2d 3/2/1": lambda x: F.max_pool2d(x, 3, stride=2, padding=1),
    "softmax dim1": lambda x: F.softmax(x, 1),
    "leaky_relu": lambda x: F.leaky_relu(x, 0.2),
    "tanh*sigmoid": lambda x: torch.tanh(x) * torch.sigmoid(x),
    "unfold 3x3": lambda x: F.unfold(x, 3, padding=1),
    "pad replicate": lambda x: F.pad(x, (1, 1, 1, 1), mode="replicate"),
    "grid-free slice": lambda x: x[:, 3:20, 1:, :-1] * 2,
    "cat": lambda x: torch.cat([x, x * 2], 1),
    "mean hw": lambda x: x.mean((2, 3), keepdim=True) * x,
    "pixel_shuffle": lambda x: F.pixel_shuffle(x, 2),
}
for shape in ((1, 128, 8, 16), (2, 96, 16, 32), (1, 32, 33, 61)):
    print("shape", shape)
    for name, f in ops.items():
        x0 = torch.randn(*shape, device="cuda")
        y0 = f(x0)
        dy0 = torch.randn_like(y0)
        res = {}
        for xf in ("nchw", "cl"):
            for gf in ("nchw", "cl"):
                x = x0.clone()
                if xf == "cl":
                    x = x.contiguous(memory_format=torch.channels_last)
                x.requires_grad_(True)
                y = f(x)
                dy = dy0.clone()
                if gf == "cl" and dy.dim() == 4:
                    dy = dy.contiguous(memory_format=torch.channels_last)
                (g,) = torch.autograd.grad(y, x, dy)
                res[(xf, gf)] = (y.detach(), g)
        ref_y, ref_g = res[("nchw", "nchw")]
        bad = []
        for k, (y, g) in res.items():
            ey = float((y - ref_y).abs().max() / (ref_y.abs().max() + 1e-30)); eg = float((g - ref_g).abs().max() / (ref_g.abs().max() + 1e-30))
            if ey > 1e-5 or eg > 1e-5:
                bad.append(f"x={k[0]} dy={k[1]}: fwd {ey:.1e} bwd {eg:.1e}")
        print(f"   {name:30s}", "ok" if not bad else "  ***  " + "; ".join(bad))
