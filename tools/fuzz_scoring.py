"""Randomised sweep of the teacher-forced scoring kernels on the GPU box: ``sequence_perplexity`` (dh_token_logprob +
dh_seq_perplexity) against the reference's formula (metrics.py:4-9: log-softmax, gather, divide by length, mask ``target ==
pad_index``, exp of the negated sum) on random logits / targets / lengths -- pads also INSIDE sequences and lengths that do not match
the pad pattern, because the formula masks by token value and divides by the given length -- and ``sequence_perplexity_from_hidden``
(dh_vocab_logprob: classifier fused with the log-sum-exp, 16-bit) against the same formula on the same rounded operands.
TEST INFRASTRUCTURE.

    python tools/fuzz_scoring.py --trials 300 > gpurun_out/fuzz_score.jsonl
"""
import argparse
import json
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from deephumor_amd.experiments.metrics import sequence_perplexity, sequence_perplexity_from_hidden      # noqa: E402


def formula(logits, targets, lengths, pad):
    logp = logits.double().log_softmax(-1).gather(-1, targets.unsqueeze(-1)).squeeze(-1)
    logp = logp / lengths.unsqueeze(1)
    logp = logp.masked_fill(targets == pad, 0.)
    return (-logp.sum(dim=-1)).exp()


def one_trial(rng, idx):
    g = torch.Generator().manual_seed(12000 + idx)
    bs, tl = rng.randint(1, 40), rng.randint(1, 40)
    v = rng.choice([rng.randint(2, 64), rng.randint(65, 1000), rng.randint(1001, 6000), 36541])
    if v > 6000:
        bs, tl = min(bs, 6), min(tl, 12)
    pad = rng.choice([0, 0, 0, rng.randrange(v)])
    logits = torch.randn(bs, tl, v, generator=g) * rng.choice([0.5, 2.5, 8.0])
    targets = torch.randint(0, v, (bs, tl), generator=g)
    lengths = torch.tensor([rng.randint(1, tl) for _ in range(bs)])
    if rng.random() < 0.7:
        for r in range(bs):
            targets[r, int(lengths[r]):] = pad
    rec = dict(bs=bs, L=tl, V=v, pad=pad)
    want = formula(logits, targets, lengths, pad)
    got = sequence_perplexity(logits.cuda(), targets.cuda(), lengths, pad).double().cpu()
    fin = want < 1e30                                   # beyond that the fp32 exp of the reference overflows to inf as well
    rel = float(((got[fin] - want[fin]).abs() / want[fin].abs().clamp_min(1e-30)).max()) if bool(fin.any()) else 0.0
    rec["fp32_rel"] = rel
    ok = rel < 2e-5 and bool((got[~fin] > 1e30).all())
    d = 64 * rng.randint(2, 8)                          # dh_vocab_logprob: K % 64 == 0, K >= 128 (score_captions falls back otherwise)
    for name, dt, tol in (("bf16", torch.bfloat16, 2e-2), ("f16", torch.float16, 3e-3)):
        hid = (torch.randn(bs, tl, d, generator=g)).to(dt)
        w = (torch.randn(v, d, generator=g) * (2.5 / d ** 0.5)).to(dt)
        b = torch.randn(v, generator=g) * 0.5
        lg = torch.nn.functional.linear(hid.float(), w.float(), b)
        want16 = formula(lg, targets, lengths, pad)
        got16 = sequence_perplexity_from_hidden(hid.cuda(), w.cuda(), b.cuda(), targets.cuda(), lengths, pad).double().cpu()
        # compare in log space (perplexities of random models span many decades)
        rel16 = float((got16[fin16].log() - want16[fin16].log()).abs().max()) if bool((fin16 := want16 < 1e30).any()) else 0.0
        rec[f"{name}_d"] = d
        rec[f"{name}_dlog"] = rel16
        ok = ok and rel16 < tol
    rec["ok"] = bool(ok)
    return rec


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args(argv)
    bad, worst = 0, {}
    for i in range(args.trials):
        rng = random.Random(args.seed * 100003 + i)
        try:
            rec = one_trial(rng, i)
        except Exception as e:
            rec = {"ok": False, "error": f"{type(e).__name__}: {e}"[:400]}
        bad += (not rec["ok"])
        for k in ("fp32_rel", "bf16_dlog", "f16_dlog"):
            if rec.get(k) is not None:
                worst[k] = max(worst.get(k, 0.0), rec[k])
        print(json.dumps(dict(i=i, **rec)), flush=True)
    print(json.dumps({"trials": args.trials, "failures": bad, "worst": worst}), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
