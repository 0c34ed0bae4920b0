import sys, torch
sys.path.insert(0, '.')
from deephumor_amd import hip
M, V, K = 1280, 36541, 512
a = (torch.randn(M, K, device='cuda') * 1).bfloat16()
w = (torch.randn(V, K, device='cuda') * 0.05).bfloat16()
b = torch.zeros(V, device='cuda')
ld = (V + 63) // 64 * 64
logits = torch.empty(M, ld, device='cuda')[:, :V]
gm = torch.empty(M, hip.n_groups(V), device='cuda')
for _ in range(3): hip.vocab_logits(a, w, b, logits, gm)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): hip.vocab_logits(a, w, b, logits, gm)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print('vocab', ms * 1e3, 'us', 2.0 * M * V * K / ms / 1e9, 'TF')
# a long-K square-ish GEMM for reference
A2 = torch.randn(4096, 4096, device='cuda').bfloat16(); W2 = torch.randn(4096, 4096, device='cuda').bfloat16()
for _ in range(2): o = hip.linear(A2, W2)
e0.record()
for _ in range(5): o = hip.linear(A2, W2)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print('4096^3', ms * 1e3, 'us', 2.0 * 4096**3 / ms / 1e9, 'TF')
