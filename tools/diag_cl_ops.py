"""Diagnostic (GPU): do PyTorch-ROCm's pooling / resampling / elementwise backward kernels agree between NCHW-contiguous and
channels-last inputs?  (The engine's conv outputs are channels-last; the torch-conv run of the same model sees NCHW.)"""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
dev = "cuda"


def check(name, fn, shape, extra=None):
    x = torch.randn(*shape, device=dev)
    outs = []
    for cl in (False, True):
        xx = x.clone()
        if cl:
            xx = xx.to(memory_format=torch.channels_last)
        xx.requires_grad_()
        y = fn(xx)
        g = torch.randn(y.shape, device=dev, generator=torch.Generator(dev).manual_seed(1))
        if cl:
            g = g.to(memory_format=torch.channels_last)
        y.backward(g)
        outs.append((y.detach().contiguous(), xx.grad.contiguous()))
    ey = float((outs[0][0] - outs[1][0]).abs().max()); eg = float((outs[0][1] - outs[1][1]).abs().max())
    print(f"{name:50s} fwd diff {ey:.2e}  grad diff {eg:.2e}  (|grad| max {float(outs[0][1].abs().max()):.2e})")


for shp in ((1, 128, 16, 32), (1, 128, 8, 16), (1, 128, 4, 8), (2, 128, 34, 60)):
    check(f"avg_pool2d 3 s2 p1 {shp}", lambda t: F.avg_pool2d(t, 3, stride=2, padding=1), shp)
    check(f"interp bilinear ac=True x2 {shp}", lambda t: F.interpolate(t, (shp[2] * 2, shp[3] * 2), mode="bilinear", align_corners=True), shp)
    check(f"interp nearest x2 {shp}", lambda t: F.interpolate(t, (shp[2] * 2, shp[3] * 2), mode="nearest"), shp)
    check(f"sigmoid*tanh {shp}", lambda t: torch.sigmoid(t) * torch.tanh(t) + (1 - torch.sigmoid(t)) * t, shp)
    check(f"instance_norm {shp}", lambda t: F.instance_norm(t), shp)
    check(f"cat+relu {shp}", lambda t: F.relu(torch.cat([t, t * 2], 1)), shp)
    check(f"unfold {shp}", lambda t: F.unfold(t[:, :1], 3, 1, 1), shp)
