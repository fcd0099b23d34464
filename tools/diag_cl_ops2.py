import torch, torch.nn.functional as F
torch.manual_seed(0)
x = torch.randn(1, 128, 16, 32); g = torch.randn(1, 128, 8, 16)
xc = x.clone().requires_grad_(); F.avg_pool2d(xc, 3, stride=2, padding=1).backward(g); want = xc.grad
cl = lambda t: t.to(memory_format=torch.channels_last)
for name, fx, fg in (("x nchw, g nchw", lambda t: t, lambda t: t), ("x cl, g nchw", cl, lambda t: t), ("x nchw, g cl", lambda t: t, cl), ("x cl, g cl", cl, cl)):
    xx = fx(x.cuda()).requires_grad_()
    y = F.avg_pool2d(xx, 3, stride=2, padding=1)
    y.backward(fg(g.cuda()))
    print(f"{name:16s} y strides {y.stride()}  grad err vs CPU {float((xx.grad.cpu() - want).abs().max()):.2e}")
xx = cl(x.cuda()).requires_grad_()
y = F.avg_pool2d(xx.contiguous(), 3, stride=2, padding=1); y.backward(cl(g.cuda()))
print("x cl -> .contiguous() first, g cl:", float((xx.grad.cpu() - want).abs().max()))
gx = torch.ops.aten.avg_pool2d_backward(g.cuda().contiguous(), x.cuda().contiguous(), [3, 3], [2, 2], [1, 1], False, True, None)
print("aten.avg_pool2d_backward contiguous args:", float((gx.cpu() - want).abs().max()))
gx = torch.ops.aten.avg_pool2d_backward(cl(g.cuda()), cl(x.cuda()), [3, 3], [2, 2], [1, 1], False, True, None)
print("aten.avg_pool2d_backward cl args:", float((gx.cpu() - want).abs().max()), gx.stride())
