"""Stress of ``CaptionPipeline`` (copy / encode / decode on three HIP streams, rotating pinned output buffers, staging slots guarded
by events) on the GPU box: random numbers of batches of random sizes (so staging buffers are re-allocated and re-used across shapes),
uint8 HWC and fp32 NCHW inputs in pinned or pageable host memory or already on the device, overlap on and off, one pipeline object
re-used across runs, a consumer that keeps the previous result alive for one more iteration (the documented limit) -- every batch's
ids must equal ``generate_batch`` of the same images under the same seed, run sequentially on the default stream.  TEST INFRASTRUCTURE.

    python tools/fuzz_pipeline.py --trials 40 > gpurun_out/fuzz_pipe.jsonl
"""
import argparse
import json
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import deephumor_amd.models as M                                            # noqa: E402
from deephumor_amd.pipeline import CaptionPipeline, u8_preprocess          # noqa: E402
from deephumor_amd.synth import load_synthetic                             # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args(argv)
    rng = random.Random(args.seed)
    models = {}
    for kind in ("CaptioningLSTM", "CaptioningTransformer"):
        for dt in (torch.float32, torch.bfloat16):
            models[(kind, dt)] = load_synthetic(getattr(M, kind)(1000).eval(), seed=1234).cuda().to(dt)
    pipes = {}
    bad = 0
    for t in range(args.trials):
        kind, dt = rng.choice(list(models))
        model = models[(kind, dt)]
        fmt = rng.choice(["u8", "f32"])
        overlap = rng.random() < 0.8
        where = rng.choice(["pinned", "pageable", "device"])
        kw = dict(max_len=rng.randint(2, 8), beam_size=rng.choice([1, 3, 5]), top_k=20, temperature=1.1)
        key = (kind, dt, fmt, overlap, tuple(sorted(kw.items())))
        pipe = pipes.get(key)
        if pipe is None or rng.random() < 0.3:
            pipe = pipes[key] = CaptionPipeline(model, overlap=overlap, preprocess=u8_preprocess(model) if fmt == "u8" else None, **kw)
        nb = rng.randint(1, 7)
        g = torch.Generator().manual_seed(500 + t)
        batches, seeds = [], []
        for b in range(nb):
            n = rng.choice([1, 2, rng.randint(3, 24), 8])
            if fmt == "u8":
                hw = rng.choice([(224, 224), (224, 224), (rng.randint(100, 400), rng.randint(100, 400))])
                x = torch.randint(0, 256, (n, hw[0], hw[1], 3), generator=g, dtype=torch.uint8)
            else:
                x = torch.randn(n, 3, 224, 224, generator=g)
            x = x.pin_memory() if where == "pinned" else (x.cuda() if where == "device" else x)
            batches.append((x,))
            seeds.append(rng.randint(0, 10 ** 6))
        got, prev = [], None
        for toks, lens in pipe.run(batches, seeds=seeds):
            if prev is not None:                       # the previous pair is still valid here (kept for ONE further iteration)
                got.append((prev[0].clone(), prev[1].clone()))
            prev = (toks, lens)
        got.append((prev[0].clone(), prev[1].clone()))
        torch.cuda.synchronize()
        ok = len(got) == nb
        with torch.no_grad():
            for (x,), s, (toks, lens) in zip(batches, seeds, got):
                xd = x.cuda()
                inp = u8_preprocess(model)(xd)[0] if fmt == "u8" else xd
                wt, wl = model.generate_batch(inp, seed=s, **kw)
                ok = ok and bool(torch.equal(wt.cpu(), toks.cpu()) and torch.equal(wl.cpu(), lens.cpu()))
        bad += (not ok)
        print(json.dumps(dict(t=t, ok=bool(ok), kind=kind, dt=str(dt)[6:], fmt=fmt, overlap=overlap, where=where, batches=[int(b[0].shape[0]) for b in batches], **kw)), flush=True)
    print(json.dumps({"trials": args.trials, "failures": bad}), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
