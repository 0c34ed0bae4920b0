"""Developer probe: per-stage times of the logits-free decode classifier (dh_vocab_topk_sample) against the dense pair
(dh_vocab_logits + dh_beam_row_sample_groups) at the decode shape; KB_HOT=1 biases 64 tokens so that every row lists the same group."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephumor_amd import hip
hip.load()
rows, v, k = int(os.environ.get("KB_M", 1280)), 36541, 512
a = torch.randn(rows, k, device="cuda").bfloat16(); w = (torch.randn(v, k, device="cuda") * 0.05).bfloat16(); b = torch.zeros(v, device="cuda")
if os.environ.get("KB_HOT"):
    b[640:704] += 8.0
    b[6400:6420] += 8.0
buf = hip.TopkBuffers(rows, v, "cuda")
pi = torch.empty(rows, 5, dtype=torch.int32, device="cuda"); pv = torch.empty(rows, 5, device="cuda"); err = torch.zeros(1, dtype=torch.int32, device="cuda")
logits = torch.empty(rows, 36544, device="cuda")[:, :v]; gm = torch.empty(rows, hip.n_groups(v), device="cuda")
for i in range(5): hip.vocab_topk_sample(a, w, b, buf, 5, 5, 50, 1.0, 1, None, 0, 0, i, pi, pv, err)
torch.cuda.synchronize()
with hip.profile() as prof:
    for i in range(30): hip.vocab_topk_sample(a, w, b, buf, 5, 5, 50, 1.0, 1, None, 0, 0, i, pi, pv, err)
    torch.cuda.synchronize()
tot = 0
for k_, v_ in prof.summary().items():
    print("topk ", k_, round(v_["ms"] / v_["calls"] * 1e3, 1)); tot += v_["ms"] / v_["calls"] * 1e3
print("topk  total", round(tot, 1))
p1 = pi.clone()
with hip.profile() as prof:
    for i in range(30):
        hip.vocab_logits(a, w, b, logits, gm)
        hip.beam_row_sample_groups(logits, v, gm, rows, 5, 5, 50, 1.0, 1, None, 0, 0, i, pi, pv, err)
    torch.cuda.synchronize()
tot = 0
for k_, v_ in prof.summary().items():
    print("dense", k_, round(v_["ms"] / v_["calls"] * 1e3, 1)); tot += v_["ms"] / v_["calls"] * 1e3
print("dense total", round(tot, 1), "| same picks:", bool(torch.equal(p1, pi)), "err", int(err.item()), "mean groups/row", float(buf.cand_n.float().mean()))
