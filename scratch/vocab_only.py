"""Only the vocabulary GEMM at the C2 decode shape (for rocprofv3 --pmc passes)."""
import sys, torch
sys.path.insert(0, '.')
from deephumor_amd import hip
M, V, K = 1280, 36541, 512
a = torch.randn(M, K, device='cuda').bfloat16(); w = (torch.randn(V, K, device='cuda') * 0.05).bfloat16()
b = torch.zeros(V, device='cuda')
logits = torch.empty(M, (V + 63) // 64 * 64, device='cuda')[:, :V]; gm = torch.empty(M, hip.n_groups(V), device='cuda')
for _ in range(10): hip.vocab_logits(a, w, b, logits, gm)
torch.cuda.synchronize()
