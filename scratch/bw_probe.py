"""HBM copy / fill rates (torch kernels) for reference against the HBM-bound conv layers."""
import torch
def t(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (103, 411, 1024):
    x = torch.empty(mb * 1024 * 1024 // 2, dtype=torch.bfloat16, device='cuda').normal_()
    y = torch.empty_like(x)
    s = t(lambda: y.copy_(x)); print(f"{mb} MB copy: {2*x.numel()*2/s/1e9:.0f} GB/s (r+w)")
    s = t(lambda: y.fill_(1.0)); print(f"{mb} MB fill: {x.numel()*2/s/1e9:.0f} GB/s (w)")
    s = t(lambda: x.sum()); print(f"{mb} MB sum: {x.numel()*2/s/1e9:.0f} GB/s (r)")
    q = x[: x.numel() // 5]
    s = t(lambda: torch.cat([q, q, q, q], out=y[: 4 * q.numel()])); print(f"{mb} MB 1r:4w: {5*q.numel()*2/s/1e9:.0f} GB/s")
