"""Developer tool: where does the HOST time of one C3 step go?  (batch 16: the GPU work is negligible, wall = host time)"""
import cProfile
import pstats
import sys
import time
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
model, sd, hp = bench.build_model(wl, dev, "bf16")
from deephumor_amd.synth import synth_images
imgs = synth_images(n, seed=0).to(dev)
with torch.no_grad():
    for s in range(3):
        bench.one_step(model, imgs, 0, n, seed=s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(5):
        bench.one_step(model, imgs, 0, n, seed=10 + s)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{wl} batch {n}: host issue {t_issue / 5 * 1e3:.2f} ms/step, wall {t_all / 5 * 1e3:.2f} ms/step")
    pr = cProfile.Profile()
    pr.enable()
    for s in range(3):
        bench.one_step(model, imgs, 0, n, seed=20 + s)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
