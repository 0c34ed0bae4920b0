"""Shared test helpers: model construction on synthetic weights, golden loading."""
import json
import os

import numpy as np
import torch

from deephumor_amd.synth import synth_state_dict, synth_images  # noqa: F401

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEED = 1234


def meta():
    with open(os.path.join(GOLDEN, "golden_meta.json")) as f:
        return json.load(f)


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def shapes_to_sd(kind, m=None):
    """A zero state dict with the reference's recorded key layout (golden_meta.json), with the
    fixed ``scale`` parameters filled in as the reference constructs them."""
    m = m or meta()
    rec = m["models"][kind]
    hp = rec["hp"] or {"num_tokens": 1000, "hid_dim": 512, "n_layers": 6, "n_heads": 8, "pf_dim": 2048,
                       "enc_dropout": 0.3, "dec_dropout": 0.1, "pad_index": 0, "max_len": 128}
    sd = {}
    for k, shp in rec["keys"].items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros(shp, dtype=torch.int64)
        elif k.endswith("_attn.scale"):
            sd[k] = torch.sqrt(torch.tensor(float(hp["hid_dim"] // hp["n_heads"])))
        elif k == "decoder.scale":
            sd[k] = torch.sqrt(torch.tensor(float(hp["hid_dim"])))
        else:
            sd[k] = torch.zeros(shp)
    return sd, hp


def synthetic_sd(kind, v=None):
    """Synthetic fp32 CPU state dict for ``kind`` at vocabulary ``v`` (default: the small golden vocab)."""
    sd, hp = shapes_to_sd(kind)
    if v is not None:
        old = hp.get("num_tokens", 1000) if hp else 1000
        for k, t in list(sd.items()):
            if t.dim() >= 1 and t.shape[0] == old and ("embedding" in k or "classifier" in k):
                sd[k] = torch.zeros((v,) + tuple(t.shape[1:]))
        hp = dict(hp, num_tokens=v)
    return synth_state_dict(sd, seed=SEED), hp


def captions_and_lengths(v=1000):
    g = np.random.Generator(np.random.Philox(key=[SEED, 77]))
    cap = torch.from_numpy(g.integers(6, v, size=(4, 31)).astype(np.int64))
    lengths = torch.tensor([32, 20, 32, 11])
    for r, n in enumerate(lengths.tolist()):
        cap[r, n - 1:] = 0
    labels = torch.from_numpy(g.integers(6, v, size=(4, 3)).astype(np.int64))
    return cap, lengths, labels


KINDS = ("CaptioningLSTM", "CaptioningLSTMWithLabels", "CaptioningTransformerBase",
         "CaptioningTransformer", "CaptioningTransformerWithLabels")
PREFIX = torch.tensor([[17, 230, 45]])
