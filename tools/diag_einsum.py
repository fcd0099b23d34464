import torch, sys
torch.manual_seed(0)
def p(*a):
    print(*a, flush=True)
for (B, Cn, H, W) in ((1, 96, 16, 32), (1, 96, 136, 240), (2, 96, 80, 184)):
    f1 = torch.randn(B, Cn, H, W, device="cuda"); f2 = torch.randn(B, Cn, H, W, device="cuda")
    ref = torch.einsum("aijk,aijh->ajkh", f1, f2)
    torch.cuda.synchronize(); p("fp32 einsum ok", tuple(ref.shape))
    for dt in (torch.float16, torch.bfloat16):
        out = torch.einsum("aijk,aijh->ajkh", f1.to(dt), f2.to(dt))
        torch.cuda.synchronize(); p(" ", dt, "direct: err", float((out.float() - ref).abs().max()), "max", float(out.float().abs().max()))
        with torch.autocast("cuda", dtype=dt):
            out = torch.einsum("aijk,aijh->ajkh", f1, f2)
        torch.cuda.synchronize(); p(" ", dt, "autocast: dtype", out.dtype, "err", float((out.float() - ref).abs().max()), "contig", out.is_contiguous(), out.stride())
        # the same contraction as an explicit bmm
        a = f1.permute(0, 2, 3, 1).reshape(B * H, W, Cn).to(dt); b = f2.permute(0, 2, 1, 3).reshape(B * H, Cn, W).to(dt)
        out2 = torch.bmm(a, b); torch.cuda.synchronize(); p(" ", dt, "bmm: err", float((out2.float().reshape(B, H, W, W) - ref).abs().max()))
