import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephumor_amd import hip
hip.load()
torch.manual_seed(0)
rows, v, k = 1280, 36541, 512
a = torch.randn(rows, k).bfloat16().cuda(); w = (torch.randn(v, k) * 0.1).bfloat16().cuda(); b = torch.randn(v).cuda()
ref = hip.linear(a, w, b, out_dtype=torch.float32)
ld = (v + 127) // 128 * 128
logits = torch.full((rows, ld), float("nan"), device="cuda")[:, :v]
gmax = torch.full((rows, hip.n_groups(v)), float("nan"), device="cuda")
hip.vocab_logits(a, w, b, logits, gmax)
bad = (logits != ref)
print("bad elements", int(bad.sum()), "of", bad.numel())
if bad.any():
    idx = bad.nonzero()
    print("first", idx[:10].tolist())
    r, c = idx[:, 0], idx[:, 1]
    print("rows mod 256 hist", torch.bincount(r % 256, minlength=256).nonzero().flatten()[:40].tolist())
    print("rows mod 32", torch.bincount(r % 32, minlength=32).tolist())
    print("cols mod 128", torch.bincount(c % 128, minlength=128).tolist())
    print("panels", torch.bincount(c // 128).nonzero().flatten()[:20].tolist(), int((torch.bincount(c // 128) > 0).sum()))
    i0 = idx[0]
    print(float(logits[i0[0], i0[1]]), float(ref[i0[0], i0[1]]), "isnan", int(torch.isnan(logits).sum()))
    d = (logits - ref).abs()
    print("max abs diff", float(d[~torch.isnan(d)].max()))
