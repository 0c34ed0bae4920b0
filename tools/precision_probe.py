#!/usr/bin/env python3
"""What 16-bit storage costs in greedy tokens, and which tensor's precision buys what.

    python tools/precision_probe.py [--workload c2|c3] [--images 256] [--out gpurun_out/precision_c2.json]

Reference of every row: the fp32 HIP path (bit-exact greedy ids vs the CPU oracle on rows {0, 77, 255}, tests/test_fullsize_gpu.py)
on the same 256 synthetic bench images, greedy (beam 1, top_k 1), 32 tokens, V = 36,541.  Rows:
  * the 16-bit product paths (bf16, fp16) as they are;
  * the fp32 ARITHMETIC with only the weights rounded to a 16-bit grid (all / encoder only / decoder only / classifier only /
    recurrent or layer weights only): what storage rounding alone costs, without any 16-bit activation or state;
Metrics: token match (positional, over max(len) positions), share of captions identical, mean first-divergence position, and the
step-0 logit error (max / mean |delta| over all 256 x V logits) -- the size of the perturbation that flips thin arg-max margins.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


greedy, compare = bench.greedy_all, bench.compare_greedy


def rounded_sd(sd, dtype, pred):
    return {k: (v.to(dtype).float() if v.is_floating_point() and v.dim() >= 1 and pred(k) else v) for k, v in sd.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2", choices=["c2", "c3"])
    ap.add_argument("--images", type=int, default=256)
    ap.add_argument("--out", default=None)
    ap.add_argument("--skip-weight-rows", action="store_true")
    args = ap.parse_args()
    from deephumor_amd.synth import synth_images
    dev = torch.device("cuda", 0)
    images = synth_images(args.images, seed=0).to(dev)
    m32, sd, hp = bench.build_model(args.workload, dev, "f32")
    ref = greedy(m32, images)
    rows = {"fp32 HIP path (reference of this table)": compare(ref, ref)}
    rows["fp32 HIP path"] = dict(rows.pop("fp32 HIP path (reference of this table)"), mean_len=float(ref[1].float().mean()))
    if not args.skip_weight_rows:
        dec_core = (lambda k: "lstm" in k) if args.workload == "c2" else (lambda k: k.startswith("decoder.layers"))
        groups = {"all weights": lambda k: True, "encoder weights only": lambda k: k.startswith("encoder."),
                  "decoder weights only": lambda k: k.startswith("decoder."),
                  "classifier weight only": lambda k: k == "decoder.classifier.weight",
                  ("LSTM weights only" if args.workload == "c2" else "decoder layer weights only"): dec_core,
                  "token embedding only": lambda k: "embedding" in k}
        for dt, name in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
            for gname, pred in groups.items():
                if dt == torch.float16 and gname not in ("all weights",):
                    continue
                m32.load_state_dict(rounded_sd(sd, dt, pred))
                rows[f"fp32 arithmetic, {gname} rounded to {name}"] = compare(ref, greedy(m32, images))
                print(gname, name, rows[f"fp32 arithmetic, {gname} rounded to {name}"], flush=True)
    del m32
    torch.cuda.empty_cache()
    for dt in ("bf16", "f16"):
        m16 = bench.build_model(args.workload, dev, dt)[0]
        rows[f"{dt} product path"] = compare(ref, greedy(m16, images))
        print(dt, rows[f"{dt} product path"], flush=True)
        del m16
        torch.cuda.empty_cache()
    out = {"workload": bench.workload_name(args.workload), "images": args.images, "max_len": bench.MAX_LEN, "vocab": bench.V_WORD,
           "decode": "greedy (beam_size=1, top_k=1)", "commit": bench.git_head(), "env": {k: v for k, v in os.environ.items() if k.startswith("DH_")},
           "rows": rows}
    text = json.dumps(out, indent=1)
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(text)


if __name__ == "__main__":
    main()
