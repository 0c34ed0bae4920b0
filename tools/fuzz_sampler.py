"""Randomised sweep of the row samplers on the GPU box: ``dh_beam_row_sample`` (full-row top-k + draw; what the fp32 path runs, pinned to
the oracle by tools/fuzz_generate.py) against ``dh_beam_row_sample_groups`` (the 16-bit paths' sampler, which reads only the 64-column
groups whose maxima can hold a top-k logit) on the same fp32 logits, group maxima and noise -- picks and values must be identical --
and both against the torch statement of beam.py:32-48 (threshold with strict ``<``, <unk> dropped, softmax / T, k winners of p / E).
Random V (2 - 40,000), rows, beams, top_k, temperatures, exact ties at the threshold, -inf entries, <unk> on top, padded row strides;
Philox mode: both kernels again identical.  TEST INFRASTRUCTURE.

    python tools/fuzz_sampler.py --trials 400 > gpurun_out/fuzz_sampler.jsonl
"""
import argparse
import json
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from deephumor_amd import hip          # noqa: E402


def one_trial(rng, idx):
    g = torch.Generator().manual_seed(90000 + idx)
    v = rng.choice([rng.randint(2, 64), rng.randint(65, 700), rng.randint(701, 6000), rng.randint(6001, 40000), 36541, 71])
    beam = min(rng.choice([1, 2, 3, 5, 7, 10, 16]), max(1, v - 1))
    n_img = rng.choice([1, 2, rng.randint(3, 40)])
    first = rng.random() < 0.3
    rpi = 1 if first else beam
    rows = n_img * rpi
    top_k = rng.randint(beam, max(beam, min(v, rng.choice([beam + 1, 20, 50, 100, 300]))))
    temp = rng.choice([1.0, 1.3, 0.7, rng.uniform(0.4, 2.5)])
    ld = rng.choice([(v + 127) // 128 * 128, (v + 63) // 64 * 64, v, v + 5])
    x = torch.randn(rows, v, generator=g) * rng.choice([0.5, 2.5, 6.0])
    if rng.random() < 0.3:
        q = rng.choice([1, 2, 4])
        x = (x * q).round() / q                                   # exact ties, also at the top-k threshold
    if rng.random() < 0.25:
        x[torch.rand(rows, v, generator=g) < rng.choice([0.05, 0.6])] = float("-inf")
    if rng.random() < 0.12:                                       # flat rows: (almost) every logit ties at the threshold -- more
        flat = torch.rand(rows) < 0.7                             # survivors than the kernels' 1,024-entry candidate buffers when V allows
        x[flat] = float(rng.choice([0.0, -3.5, 2.0]))
        if rng.random() < 0.5:
            x[flat, :: rng.randint(2, 9)] += 1.0                  # two plateaus
    if rng.random() < 0.25 and v > 1:
        x[:, 1] = x[torch.isfinite(x)].max() + 1.0                # <unk> on top
    rec = dict(V=v, beam=beam, n_img=n_img, first=first, top_k=top_k, T=round(temp, 4), ld=ld)
    # torch statement; rows that end with fewer than `beam` positive-probability tokens or too many threshold ties are error cases
    kth = x.topk(top_k, dim=-1).values[:, -1:]
    kept = x.clone()
    kept[x < kth] = float("-inf")
    kept[:, 1] = float("-inf")
    alive = torch.isfinite(kept).sum(-1)
    if int(alive.min()) < beam:
        return dict(rec, ok=True, skipped="fewer survivors than beams (dead beams; covered by tools/fuzz_generate.py)")
    rec["max_survivors"] = int(alive.max())
    noise = torch.empty(rows, v).exponential_(1, generator=g)
    want = torch.topk(torch.softmax(kept / temp, -1) / noise, beam, dim=-1).indices
    want_val = torch.gather(kept, 1, want)
    buf = torch.full((rows, ld), 777.0)
    buf[:, :v] = x
    logits = buf.cuda()[:, :v]
    nbuf = torch.ones(rows, ld)
    nbuf[:, :v] = noise
    nz = nbuf.cuda()
    ng = hip.n_groups(v)
    pad = torch.full((rows, ng * 64), float("-inf"))
    pad[:, :v] = x
    gmax = pad.view(rows, ng, 64).max(-1).values.cuda()

    def run(kind, nsrc, seed=0):
        pi = torch.full((rows, beam), -7, dtype=torch.int32, device="cuda")
        pv = torch.full((rows, beam), -7.0, device="cuda")
        err = torch.zeros(1, dtype=torch.int32, device="cuda")
        if kind == "full":      # more than 1,024 survivors: the general sampler (what the models repeat such a batch with)
            hip.beam_row_sample(logits, v, rows, rpi, beam, top_k, temp, 1, nsrc, seed, 3, 5, pi, pv, err, exact=rec["max_survivors"] > 1024)
        else:
            hip.beam_row_sample_groups(logits, v, gmax, rows, rpi, beam, top_k, temp, 1, nsrc, seed, 3, 5, pi, pv, err)
        return pi.cpu().long(), pv.cpu(), int(err.item())

    def run_fast_err():
        pi = torch.full((rows, beam), -7, dtype=torch.int32, device="cuda")
        pv = torch.full((rows, beam), -7.0, device="cuda")
        err = torch.zeros(1, dtype=torch.int32, device="cuda")
        hip.beam_row_sample(logits, v, rows, rpi, beam, top_k, temp, 1, nz, 0, 3, 5, pi, pv, err)
        return int(err.item())

    fi, fv, fe = run("full", nz)
    rec["full_vs_torch"] = bool(torch.equal(fi, want) and fe == 0)
    ok = rec["full_vs_torch"]
    if not ok:
        r = int((fi != want).any(-1).nonzero()[0]) if bool((fi != want).any()) else 0
        qq = (torch.softmax(kept / temp, -1) / noise)[r]
        rec.update(err=fe, row=r, want=want[r].tolist(), got=fi[r].tolist(), q_want=[float(qq[j]) for j in want[r]],
                   q_got=[float(qq[j]) if 0 <= j < v else None for j in fi[r].tolist()], alive=int(alive[r]),
                   ties_at_threshold=int((x[r] == kth[r]).sum()))
    if rec["max_survivors"] > 1024:                               # the pre-filtered samplers must FLAG such rows (never answer wrongly)
        if top_k <= 256 and v <= 65536:                           # (otherwise dh_beam_row_sample is the general kernel already)
            rec["fast_flags_overflow"] = bool(run_fast_err() & hip.ERR_OVERFLOW)
            ok = ok and rec["fast_flags_overflow"]
    elif top_k <= ng:                                             # the engine's condition for the group-guided sampler
        gi, gv, ge = run("groups", nz)
        # (the two kernels sum the picks' log-softmax in different orders: values to 1e-6, ids exactly)
        rec["groups_vs_full"] = bool(torch.equal(gi, fi) and torch.allclose(gv, fv, atol=1e-6, rtol=1e-6) and ge == fe)
        if not rec["groups_vs_full"]:
            bad_rows = (gi != fi).any(-1).nonzero().flatten().tolist()
            r = bad_rows[0] if bad_rows else 0
            rec.update(err_full=fe, err_groups=ge, n_bad_rows=len(bad_rows), row=r, full=fi[r].tolist(), groups=gi[r].tolist(), alive=int(alive[r]),
                       ties_at_threshold=int((x[r] == kth[r]).sum()), groups_with_survivors=int(torch.isfinite(kept[r]).view(-1)[:0].numel()))
        pi1, pv1, e1 = run("full", None, seed=1234 + idx)
        pi2, pv2, e2 = run("groups", None, seed=1234 + idx)
        rec["philox_equal"] = bool(torch.equal(pi1, pi2) and torch.allclose(pv1, pv2, atol=1e-6, rtol=1e-6) and e1 == e2)
        ok = ok and rec["groups_vs_full"] and rec["philox_equal"]
    # the values are the kept logits of the picks (the log-softmax over the picks happens in dh_beam_select)
    rec["ok"] = bool(ok)
    return rec


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args(argv)
    bad = 0
    for i in range(args.trials):
        rng = random.Random(args.seed * 100003 + i)
        try:
            rec = one_trial(rng, i)
        except Exception as e:
            rec = {"ok": False, "error": f"{type(e).__name__}: {e}"[:400]}
        bad += (not rec["ok"])
        print(json.dumps(dict(i=i, **rec)), flush=True)
    print(json.dumps({"trials": args.trials, "failures": bad}), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
