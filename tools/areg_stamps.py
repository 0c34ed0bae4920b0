"""Developer probe: phase cycle breakdown of the A-stationary classifier kernel (DH_VOCAB_AREG=2)."""
import ctypes, os, sys, torch
os.environ["DH_VOCAB_AREG"] = "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephumor_amd import hip
lib = hip.load()
M, V, K = 1280, 36541, 512
a = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(V, K, device="cuda") * K ** -0.5).bfloat16()
b = torch.zeros(V, device="cuda")
logits = torch.empty(M, (V + 3) // 4 * 4, device="cuda")[:, :V]
gm = torch.empty(M, hip.n_groups(V), device="cuda")
for mode, lg in (("logits", logits), ("gmax-only", None)):
    for _ in range(5):
        hip.vocab_logits(a, w, b, lg, gm)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (256 * 8))()
    assert lib.dh_debug_areg_stamps(buf, 256 * 8) == 0
    rows = [[buf[i * 8 + k] for k in range(7)] for i in range(256)]
    rows = [r for r in rows if r[6] > 0]
    names = ["vmcnt wait", "barrier", "reads+DMA issue", "lgkm wait", "MFMAs(+tile tail)", "epilogue"]
    tot = [sum(r[k] for r in rows) / len(rows) for k in range(6)]
    tiles = sum(r[6] for r in rows) / len(rows)
    print(mode, f"{len(rows)} workgroups, {tiles:.1f} tiles each; s_memtime ticks (100 MHz) per workgroup:")
    for n_, t in zip(names, tot):
        print(f"   {n_:22s} {t:10.0f}  ({100 * t / sum(tot):5.1f} %)  per slab {t / (tiles * 8):7.1f}")
