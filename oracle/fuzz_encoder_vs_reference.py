"""Randomised pinning of the oracle's ENCODERS (``oracle/ref_path.py``: ``image_encoder``, ``label_encoder``, ``image_label_encoder``)
against the reference's modules (``/root/reference/deephumor/models/encoders.py`` on the torchvision stand-in of ``oracle/_standin``) in
the BUILD CONTAINER: random image sizes (odd, not multiples of 32), batch sizes, embedding widths, +/- spatial features, random label
lengths -- outputs within 1e-5 relative.  TEST INFRASTRUCTURE ONLY.

    python oracle/fuzz_encoder_vs_reference.py --trials 40
"""
import argparse
import json
import os
import random
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "_standin"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

from deephumor.models.encoders import ImageEncoder, ImageLabelEncoder        # noqa: E402  (the reference's)
from deephumor_amd.synth import synth_state_dict                              # noqa: E402
from oracle import ref_path as R                                              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    torch.set_num_threads(8)
    rng = random.Random(args.seed)
    bad = 0
    for t in range(args.trials):
        g = torch.Generator().manual_seed(100 + t)
        emb = rng.choice([256, 512, rng.randint(1, 300)])
        h, w = rng.randint(33, 260), rng.randint(33, 260)
        n = rng.randint(1, 3)
        x = torch.randn(n, 3, h, w, generator=g)
        kind = rng.choice(["image", "spatial", "labels"])
        if kind == "labels":
            v = rng.randint(5, 400)
            labels = torch.randint(0, v, (n, rng.randint(1, 10)), generator=g)
            mod = ImageLabelEncoder(num_tokens=v, emb_dim=emb, dropout=0.3).eval()
        else:
            mod = ImageEncoder(emb, 0.3, spatial_features=kind == "spatial").eval()
        sd = synth_state_dict(mod.state_dict(), seed=t)
        mod.load_state_dict(sd)
        osd = {"encoder." + k: v_.clone() for k, v_ in sd.items()}
        with torch.no_grad():
            if kind == "labels":
                want, got = (mod(images=x, labels=labels),), (R.image_label_encoder(osd, "encoder", x, labels),)
            elif kind == "spatial":
                want, got = mod(x), R.image_encoder(osd, "encoder", x, True)
            else:
                want, got = (mod(x),), (R.image_encoder(osd, "encoder", x, False),)
        ok = all(tuple(a.shape) == tuple(b.shape) and float((a - b).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max())) for a, b in zip(want, got))
        bad += (not ok)
        if not ok:
            print(json.dumps(dict(t=t, kind=kind, emb=emb, hw=[h, w], n=n)), flush=True)
    print(json.dumps({"trials": args.trials, "mismatches": bad}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
