"""Randomised sweep of ``BeamSearchHelper``'s METHOD surface (reference beam.py:32-108) on the GPU box against the oracle's
``BeamBook`` / plain torch statements: ``filter_top_k`` (ties at the threshold, -inf entries, <unk> inside / outside the top-k,
V from 2 to 40,000), ``sample_k_indices`` (2-D and 1-D, under replayed noise), ``filter_by_indices``, ``process_logits`` with random
ended flags.  Ids / masks bit-exact, values <= 1e-6.  TEST INFRASTRUCTURE (imports the oracle).

    python tools/fuzz_beam_methods.py --trials 500 > gpurun_out/fuzz_beam.jsonl
"""
import argparse
import json
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from deephumor_amd.models import BeamSearchHelper                 # noqa: E402
from oracle import ref_path as R                                   # noqa: E402


def replay(kind, call, shape):
    return torch.empty(shape).exponential_(1)


def rand_logits(rng, g, n, v):
    x = torch.randn(n, v, generator=g) * rng.choice([0.5, 2.5, 6.0])
    mode = rng.random()
    if mode < 0.25:                                               # exact ties (quantised values)
        x = (x * rng.choice([1, 2, 4])).round() / rng.choice([1, 2, 4])
    if rng.random() < 0.3:                                        # some entries already -inf
        x[torch.rand(n, v, generator=g) < rng.choice([0.05, 0.5])] = float("-inf")
        x[:, rng.randrange(v)] = 0.5                              # at least one finite entry per row
    if rng.random() < 0.3 and v > 1:
        x[:, 1] = x.max() + 1.0                                   # <unk> on top
    return x


def one_trial(rng, idx):
    g = torch.Generator().manual_seed(9000 + idx)
    v = rng.choice([rng.randint(2, 64), rng.randint(65, 700), rng.randint(701, 5000), rng.choice([36541, 40000])])
    beam = min(rng.choice([1, 2, 3, 5, 7, 10, 16]), max(1, v - 1))
    top_k = rng.randint(beam, min(v, rng.choice([beam + 1, 20, 50, 100, 300])) if min(v, 300) >= beam else beam)
    top_k = max(top_k, beam)
    temp = rng.choice([1.0, 1.3, 0.7, rng.uniform(0.4, 2.5)])
    rec = dict(V=v, beam=beam, top_k=top_k, T=round(temp, 4))
    book = R.BeamBook(temp, beam, top_k)
    # filter_top_k on n rows
    n = rng.randint(1, 6)
    x = rand_logits(rng, g, n, v)
    want = book.keep_top_k(x.clone())
    h = BeamSearchHelper(temperature=temp, beam_size=beam, top_k=top_k, device="cuda", noise_source=replay)
    got = h.filter_top_k(x.clone().cuda()).cpu()
    rec["filter_ok"] = bool(torch.equal(got, want))
    # sample_k_indices 2-D (only rows with >= k positive entries, otherwise both sides are in undefined territory)
    k = rng.randint(1, beam)
    alive = (want > float("-inf")).sum(-1)
    if int(alive.min()) >= k:
        torch.manual_seed(100 + idx)
        nz = torch.empty(want.shape).exponential_(1)
        ref = torch.topk(torch.softmax(want / temp, -1) / nz, k, dim=-1).indices
        torch.manual_seed(100 + idx)
        mine = h.sample_k_indices(want.cuda(), k=k).cpu()
        rec["sample2d_ok"] = bool(torch.equal(mine, ref))
        rec["gather_ok"] = bool(torch.equal(BeamSearchHelper.filter_by_indices(want.cuda(), mine.cuda()).cpu(), torch.gather(want, 1, ref)))
    # process_logits: beam rows, random ended flags
    logits = rand_logits(rng, g, beam, v)
    ended = torch.tensor([rng.random() < 0.3 for _ in range(beam)])
    tlen = rng.randint(1, 9)
    seqs = torch.randint(0, v, (beam, tlen), generator=g)
    vals = -torch.rand(beam, 1, generator=g) * 5
    kept = book.keep_top_k(logits.clone())
    if int((kept > float("-inf")).sum(-1).min()) >= beam:
        book.ended = ended.clone()
        torch.manual_seed(200 + idx)
        ps, pv, ni, nv, _ = book.expand(logits.clone(), seqs, vals)
        h2 = BeamSearchHelper(temperature=temp, beam_size=beam, top_k=top_k, device="cuda", noise_source=replay)
        h2.has_ended = ended.clone().cuda()
        torch.manual_seed(200 + idx)
        (gps, gpv), (gni, gnv) = h2.process_logits(logits.clone().cuda(), seqs.cuda(), vals.cuda())
        rec["process_ok"] = bool(torch.equal(gps.cpu(), ps) and torch.equal(gni.cpu(), ni) and torch.equal(h2.has_ended.cpu(), book.ended)
                                 and float((gpv.cpu().flatten() - pv.flatten()).abs().max()) <= 1e-6
                                 and float((gnv.cpu() - nv).abs().max()) <= 1e-5)
        # the 1-D candidate draw
        cand = (pv.flatten() + nv)
        kk = min(beam, int((cand > float("-inf")).sum()))
        if kk >= 1:
            torch.manual_seed(300 + idx)
            nz = torch.empty(cand.shape).exponential_(1)
            ref = torch.topk(torch.softmax(cand / temp, -1) / nz, kk).indices
            torch.manual_seed(300 + idx)
            rec["sample1d_ok"] = bool(torch.equal(h2.sample_k_indices(cand.cuda(), k=kk).cpu(), ref))
    return rec


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args(argv)
    bad = 0
    for i in range(args.trials):
        rng = random.Random(args.seed * 100003 + i)
        try:
            rec = one_trial(rng, i)
            ok = all(v for k, v in rec.items() if k.endswith("_ok"))
        except Exception as e:
            rec, ok = {"error": f"{type(e).__name__}: {e}"[:400]}, False
        bad += (not ok)
        print(json.dumps(dict(i=i, ok=ok, **rec)), flush=True)
    print(json.dumps({"trials": args.trials, "failures": bad}), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
