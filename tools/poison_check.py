"""Does any kernel read memory it did not write?  Every trial decodes the same batch three times, each time after the caching allocator
has been handed back a few GB of blocks filled with a different byte pattern (0x00, 0xFF = NaN in every floating type and -1 in the
integer ones, 0x7B = large finite values), so that the ``torch.empty`` scratch, cache and output tensors of the decode come out of poisoned
blocks -- and, on a second pass, with fresh model plans (weights repacked into poisoned blocks as well).  Tokens, lengths and (for the
teacher-forced forward) logits must not depend on the pattern.  All five model classes, fp32 / bf16 / fp16, random batch sizes and decode
settings; per-caption perplexities of the scoring path (``experiments.scoring.score_captions``) included.  TEST INFRASTRUCTURE; runs on the GPU box:

    python tools/poison_check.py --trials 60 > gpurun_out/poison.jsonl
"""
import argparse
import json
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import deephumor_amd.models as M                          # noqa: E402
from deephumor_amd import hip                             # noqa: E402
from deephumor_amd.synth import load_synthetic, synth_images           # noqa: E402
from deephumor_amd.experiments.scoring import score_captions           # noqa: E402

KINDS = ("CaptioningLSTM", "CaptioningLSTMWithLabels", "CaptioningTransformerBase", "CaptioningTransformer", "CaptioningTransformerWithLabels")


def poison(byte, gib=3.0):
    """Fill ``gib`` GiB of allocator blocks of assorted sizes with ``byte`` and give them back to the caching allocator."""
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    blocks = []
    sizes = [1 << 30, 1 << 28, 1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 14, 1 << 12, 512]
    left = int(gib * (1 << 30))
    for sz in sizes:
        reps = max(1, min(24, left // (len(sizes) * sz)))
        for _ in range(reps):
            blocks.append(torch.full((sz,), byte, dtype=torch.uint8, device="cuda"))
    torch.cuda.synchronize()
    del blocks
    # the mechanism works only if later allocations come out of the poisoned blocks: probe a few sizes
    hit = []
    for sz in (1 << 12, 1 << 18, 1 << 22, 1 << 27):
        probe = torch.empty((sz,), dtype=torch.uint8, device="cuda")
        hit.append(float((probe == byte).float().mean()))
        del probe
    return min(hit)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args(argv)
    rng = random.Random(args.seed)
    bad = 0
    for t in range(args.trials):
        kind = rng.choice(KINDS)
        dt = rng.choice([torch.float32, torch.bfloat16, torch.float16])
        v = rng.choice([71, 300, 1000, 36541 if rng.random() < 0.3 else 1000])
        n = rng.choice([1, 3, 8, rng.randint(9, 70), 256 if v > 1000 else 40])
        model = load_synthetic(getattr(M, kind)(v).eval(), seed=1234).cuda().to(dt)
        g = torch.Generator().manual_seed(900 + t)
        images = synth_images(n, seed=t).cuda()
        extra = (torch.randint(4, v, (n, rng.randint(1, 6)), generator=g).cuda(),) if "WithLabels" in kind else ()
        beam = rng.choice([1, 3, 5, 10])
        kw = dict(max_len=rng.randint(2, 12), beam_size=beam, top_k=min(v, max(beam + 1, rng.choice([5, 20, 50]))), temperature=rng.choice([1.0, 1.3]),
                  seed=rng.randint(0, 10 ** 6))
        cap = torch.randint(4, v, (n, rng.randint(2, 9)), generator=g).cuda()
        lengths = torch.randint(1, cap.shape[1] + 1, (n,), generator=g)
        cap_pad = cap.clone()
        cap_pad[(torch.arange(cap.shape[1])[None, :] >= lengths[:, None]).cuda()] = 0          # (padding as pad_collate leaves it)
        tidx = torch.randint(0, n, (n,), generator=g)
        # the opt-in kernel selections
        opts = {k: 1 for k in ("vocab_wreg_transformer", "encoder_generic") if rng.random() < 0.3}
        rec = dict(t=t, options=opts, kind=kind, dt=str(dt)[6:], V=v, N=n, **{k: x for k, x in kw.items()})
        try:
            outs = []
            for byte, fresh_plan in ((0x00, False), (0xFF, False), (0x7B, False), (0xFF, True)):
                if fresh_plan:                                   # weights repacked into poisoned blocks too
                    for mod in model.modules():
                        for attr in ("_plan", "_plans", "_plan_cache", "_graphs"):
                            if attr in mod.__dict__:
                                mod.__dict__.pop(attr)
                rec.setdefault("poisoned_fraction_of_fresh_blocks", []).append(round(poison(byte), 3))
                with torch.no_grad(), hip.option_scope(**opts):
                    toks, lens = model.generate_batch(images, *extra, **kw)
                    logits = model(images, cap, lengths, *extra).float()
                    ppl = score_captions(model, images, tidx, cap_pad, lengths, labels=extra[0] if extra else None, batch_size=rng.choice([7, 64]))
                torch.cuda.synchronize()
                outs.append((toks.cpu(), lens.cpu(), logits.cpu(), ppl.float().cpu()))
            same = [bool(torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1])) for o in outs[1:]]
            same_lg = [bool(torch.equal(o[2], outs[0][2]) and torch.equal(o[3], outs[0][3])) for o in outs[1:]]
            rec["tokens_same"], rec["logits_same"], rec["finite"] = same, same_lg, bool(torch.isfinite(outs[1][2]).all())
            ok = all(same) and all(same_lg) and rec["finite"]
        except Exception as e:                                # noqa: BLE001 -- a raising trial is a failing trial
            rec["error"], ok = f"{type(e).__name__}: {e}"[:300], False
        rec["ok"] = bool(ok)
        bad += (not ok)
        print(json.dumps(rec), flush=True)
        del model
    print(json.dumps({"trials": args.trials, "failures": bad}), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
