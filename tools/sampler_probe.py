"""Developer probe: dh_beam_row_sample_groups (the decode sampler) on real classifier outputs, by row count -- latency-bound (time flat in
the row count) or throughput-bound (time proportional to it)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephumor_amd import hip
hip.load()
v, k = 36541, 512
w = (torch.randn(v, k, device="cuda") * 0.1).bfloat16(); b = torch.randn(v, device="cuda")
ld = (v + 127) // 128 * 128
def timeit(fn, iters=30, warm=5):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    with hip.profile() as prof:
        for i in range(iters): fn(i)
        torch.cuda.synchronize()
    return {k: round(v["ms"] / v["calls"] * 1e3, 1) for k, v in prof.summary().items()}
for rows in (256, 512, 1280, 2560):
    a = torch.randn(rows, k, device="cuda").bfloat16()
    logits = torch.empty(rows, ld, device="cuda")[:, :v]
    gmax = torch.empty(rows, hip.n_groups(v), device="cuda")
    hip.vocab_logits(a, w, b, logits, gmax)
    pi = torch.empty(rows, 5, dtype=torch.int32, device="cuda"); pv = torch.empty(rows, 5, device="cuda"); err = torch.zeros(1, dtype=torch.int32, device="cuda")
    junk = torch.empty(96 * 1024 * 1024, device="cuda")
    def f(i):
        junk.fill_(float(i))          # the sampler reads logits the classifier wrote long ago (not L2-resident), as in the decode step
        hip.beam_row_sample_groups(logits, v, gmax, rows, 5, 5, 50, 1.0, 1, None, 1, 0, i, pi, pv, err)
    print(rows, "rows:", timeit(f))
