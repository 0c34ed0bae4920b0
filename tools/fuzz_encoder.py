"""Randomised sweep of the image side on the GPU box:

  resize : ``resize_images`` (dh_resize_u8_hwc) against Pillow's ``Image.resize(..., BILINEAR)`` on random source / target sizes
           (up- and down-scaling, extreme aspect ratios, 1-pixel sides) -- bit-exact;
  encoder: ``ImageEncoder`` (ResNet-50 trunk + Linear + BatchNorm1d, +/- spatial features) at random image sizes (odd, not multiples
           of 32) and batch sizes -- fp32 HIP vs the oracle (bar 1e-4 relative to the largest |value|), bf16 / fp16 vs the fp32 HIP
           path (reported), ``LabelEncoder`` / ``ImageLabelEncoder`` with random label lengths.
TEST INFRASTRUCTURE (imports the oracle).

    python tools/fuzz_encoder.py --trials 60 --seed 1 > gpurun_out/fuzz_enc.jsonl
"""
import argparse
import json
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import deephumor_amd.models as M                                                    # noqa: E402
from deephumor_amd.experiments.inference import resize_images                       # noqa: E402
from deephumor_amd.synth import synth_state_dict                                    # noqa: E402
from oracle import ref_path as R                                                    # noqa: E402


def resize_trial(rng, idx):
    from PIL import Image
    pick = lambda: rng.choice([rng.randint(1, 40), rng.randint(41, 400), rng.randint(401, 1300), 224])
    h, w = pick(), pick()
    th, tw = rng.choice([(224, 224), (224, 224), (rng.randint(1, 300), rng.randint(1, 300))])
    n = rng.randint(1, 3)
    g = np.random.default_rng(3000 + idx)
    src = g.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)
    if rng.random() < 0.3:                                  # smooth content next to noise
        src = (np.linspace(0, 255, w)[None, None, :, None] * np.ones((n, h, 1, 3))).astype(np.uint8)
    want = np.stack([np.asarray(Image.fromarray(src[i]).resize((tw, th), Image.BILINEAR)) for i in range(n)])
    got = resize_images(torch.from_numpy(src).cuda(), (th, tw)).cpu().numpy()
    ok = got.shape == want.shape and bool((got == want).all())
    rec = dict(kind="resize", src=[h, w], dst=[th, tw], n=n, ok=ok)
    if not ok and got.shape == want.shape:
        d = np.abs(got.astype(int) - want.astype(int))
        rec.update(max_diff=int(d.max()), n_diff=int((d > 0).sum()))
    return rec


def encoder_trial(rng, idx):
    spatial = rng.random() < 0.5
    emb = rng.choice([256, 512, 8 * rng.randint(1, 64)])
    h, w = (rng.randint(33, 330), rng.randint(33, 330)) if rng.random() < 0.8 else (224, 224)
    n = rng.randint(1, 3)
    with_labels = (not spatial) and rng.random() < 0.4
    g = torch.Generator().manual_seed(4000 + idx)
    x = torch.randn(n, 3, h, w, generator=g)
    rec = dict(kind="encoder", emb=emb, spatial=spatial, hw=[h, w], n=n, labels=with_labels)
    if with_labels:
        v = rng.randint(5, 500)
        labels = torch.randint(0, v, (n, rng.randint(1, 12)), generator=g)
        enc = M.ImageLabelEncoder(num_tokens=v, emb_dim=emb, dropout=0.3).eval()
    else:
        enc = M.ImageEncoder(emb, 0.3, spatial_features=spatial).eval()
    sd = synth_state_dict(enc.state_dict(), seed=55 + idx)
    enc.load_state_dict(sd)
    osd = {"encoder." + k: t.clone() for k, t in sd.items()}
    with torch.no_grad():
        if with_labels:
            want = (R.image_label_encoder(osd, "encoder", x, labels),)
            run = lambda m: (m(images=x.cuda(), labels=labels.cuda()),)
        elif spatial:
            want = R.image_encoder(osd, "encoder", x, True)
            run = lambda m: m(x.cuda())
        else:
            want = (R.image_encoder(osd, "encoder", x, False),)
            run = lambda m: (m(x.cuda()),)
        got = run(enc.cuda())
        rec["shape_ok"] = all(tuple(a.shape) == tuple(b.shape) for a, b in zip(got, want))
        if rec["shape_ok"]:
            rec["fp32_rel"] = max(float((a.float().cpu() - b).abs().max()) / max(1.0, float(b.abs().max())) for a, b in zip(got, want))
            for name, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
                e16 = (M.ImageLabelEncoder(num_tokens=v, emb_dim=emb, dropout=0.3) if with_labels
                       else M.ImageEncoder(emb, 0.3, spatial_features=spatial)).eval()
                e16.load_state_dict(sd)
                g16 = run(e16.cuda().to(dt))
                rec[f"{name}_rel"] = max(float((a.float() - b.float()).abs().max()) / max(1.0, float(b.abs().max())) for a, b in zip(g16, got))
    rec["ok"] = bool(rec["shape_ok"] and rec["fp32_rel"] < 1e-4)
    return rec


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", choices=["resize", "encoder"], default=None)
    args = ap.parse_args(argv)
    bad, worst = 0, {}
    for i in range(args.trials):
        for fn in (resize_trial, encoder_trial):
            if args.only and args.only not in fn.__name__:
                continue
            rng = random.Random(args.seed * 100003 + i)
            try:
                rec = fn(rng, i)
            except Exception as e:
                rec = {"kind": fn.__name__, "ok": False, "error": f"{type(e).__name__}: {e}"[:400]}
            bad += (not rec["ok"])
            for k in ("fp32_rel", "bf16_rel", "f16_rel"):
                if rec.get(k) is not None:
                    worst[k] = max(worst.get(k, 0.0), rec[k])
            print(json.dumps(dict(i=i, **rec)), flush=True)
    print(json.dumps({"trials": args.trials, "failures": bad, "worst": worst}), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
