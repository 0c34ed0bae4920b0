"""Times one fused LSTM layer step at the C2 decode shape (1280 beam rows, hidden 512)."""
import sys, torch
sys.path.insert(0, '.')
from deephumor_amd import hip
rows, hh = 1280, 512
for e in (256, 512):
    w = (torch.randn(4 * hh, e + hh, device='cuda') * 0.05).bfloat16(); b = torch.zeros(4 * hh, device='cuda')
    x = torch.randn(rows, e, device='cuda').bfloat16()
    hp, cp = torch.randn(rows, hh, device='cuda').bfloat16(), torch.randn(rows, hh, device='cuda')
    par = torch.randint(0, rows, (rows,), device='cuda', dtype=torch.int32)
    hn, cn, ho = torch.empty_like(hp), torch.empty_like(cp), torch.empty_like(hp)
    f = lambda: hip.lstm_layer_fused(x, 1, None, None, 0, hp, cp, par, hn, cn, ho, w, b, rows, 1, e, hh)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"E={e}: {us:.1f} us  {2.0 * rows * 4 * hh * (e + hh) / us / 1e6:.0f} TF")
