"""Debug helper: re-runs one --big trial of tools/fuzz_forward.py and prints where the 16-bit logits part from the fp32 HIP logits."""
import random
import sys

import torch

import fuzz_forward as F

seed, i = int(sys.argv[1]), int(sys.argv[2])
F.BIG = True
rng = random.Random(seed * 100003 + i)
keep = {}
orig = F.R.transformer_forward
import deephumor_amd.models.transformers as tm
og = tm.TransformerDecoder.forward


def fwd(self, *a, **k):
    out = og(self, *a, **k)
    keep.setdefault(str(next(self.parameters()).dtype), out.float().cpu())
    return out


tm.TransformerDecoder.forward = fwd
rec = F.one_trial(rng, i)
print({k: v for k, v in rec.items() if k != "lengths"})
f32, b16, f16 = keep["torch.float32"], keep["torch.bfloat16"], keep["torch.float16"]
for name, t in (("bf16", b16), ("f16", f16)):
    d = (t - f32).abs()
    per_img = d.amax(dim=(1, 2))
    per_pos = d.amax(dim=(0, 2))
    top = per_img.topk(8)
    print(name, "max", float(d.max()), "mean", float(d.mean()), "per position", [round(float(x), 3) for x in per_pos])
    print("   worst images", top.indices.tolist(), [round(float(x), 3) for x in top.values], " median per-image max", float(per_img.median()))
