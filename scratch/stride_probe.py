"""Does a non-power-of-two row stride of the operands change the classifier GEMM time? (L2 channel conflicts)"""
import sys, torch
sys.path.insert(0, '.')
from deephumor_amd import hip
M, V, K = 1280, 36541, 512
b = torch.zeros(V, device='cuda')
logits = torch.empty(M, (V + 63) // 64 * 64, device='cuda')[:, :V]; gm = torch.empty(M, hip.n_groups(V), device='cuda')
for pad_a, pad_w in ((0, 0), (8, 8), (64, 64), (0, 64), (64, 0), (32, 32), (72, 72)):
    a = torch.randn(M, K + pad_a, device='cuda').bfloat16()[:, :K]
    w = (torch.randn(V, K + pad_w, device='cuda') * 0.05).bfloat16()[:, :K]
    for _ in range(3): hip.vocab_logits(a, w, b, logits, gm)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): hip.vocab_logits(a, w, b, logits, gm)
    e1.record(); torch.cuda.synchronize()
    print(f"pad a {pad_a:3d} w {pad_w:3d}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
