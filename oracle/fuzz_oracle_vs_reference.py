"""Randomised pinning of the ORACLE (``oracle/ref_path.py``) against the REAL reference decoders
(``/root/reference/deephumor/models``: ``LSTMDecoder``, ``TransformerDecoder``, ``SelfAttentionTransformerDecoder``) in the BUILD
CONTAINER: random vocabularies, widths, depths, heads, encoder lengths, beam sizes, top_k, temperatures, prefixes (incl. the
no-decode-step edge), ``pad_index`` 0 and >= 2 -- ``generate`` under the same ``torch.manual_seed`` must return the SAME tensor (shape
and ids), teacher-forced ``forward`` logits must agree to 1e-5.  The committed goldens pin the oracle at a handful of configurations;
this sweep pins it across the configuration space the GPU sweeps (tools/fuzz_*.py) then use it on.  TEST INFRASTRUCTURE ONLY.

    python oracle/fuzz_oracle_vs_reference.py --trials 400
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

from deephumor.models.rnn_models import LSTMDecoder                                                 # noqa: E402  (the reference's)
from deephumor.models.transformers import SelfAttentionTransformerDecoder, TransformerDecoder      # noqa: E402
from deephumor_amd.synth import synth_state_dict                                                     # noqa: E402
from oracle import ref_path as R                                                                     # noqa: E402


def outcome(fn, seed):
    """A result tensor, or the exception type the call raises (e.g. torch.multinomial on a row whose only top-k token is <unk>)."""
    torch.manual_seed(seed)
    try:
        return fn()
    except Exception as e:
        return type(e).__name__


def same(a, b):
    if isinstance(a, str) or isinstance(b, str):
        return a == b
    return tuple(a.shape) == tuple(b.shape) and a.reshape(-1).tolist() == b.reshape(-1).tolist()


def one_trial(rng, idx):
    kind = rng.choice(["lstm", "tfm", "tfm_self"])
    v = rng.choice([rng.randint(5, 70), rng.randint(71, 700), rng.randint(701, 3000)])
    beam = min(rng.choice([1, 2, 3, 5, 7, 10, 16, rng.randint(1, 24)]), v)
    top_k = rng.randint(beam, max(beam, min(v, rng.choice([beam + 1, 20, 50, 100, 300]))))
    if top_k == beam and top_k < v:
        top_k += 1                                  # beam == top_k with <unk> in the top-k: zero-probability picks, order undefined
    temp = rng.choice([1.0, 1.3, 0.7, rng.uniform(0.4, 2.5)])
    max_len = rng.randint(1, 24)
    prefix = min(rng.choice([0, 0, rng.randint(1, max(1, max_len - 1)), max_len - 1]), max_len - 1)
    prefix = max(prefix, 0)
    pad = rng.choice([0, 0, 0, rng.randint(2, 4)])
    cfg = dict(kind=kind, V=v, beam=beam, top_k=top_k, T=round(temp, 4), max_len=max_len, prefix=prefix, pad=pad)
    g = torch.Generator().manual_seed(1000 + idx)
    cap = torch.randint(4, v, (1, prefix), generator=g) if prefix and v > 4 else None
    bs, cl = rng.randint(1, 4), rng.randint(1, 20)
    fcap = torch.randint(4, max(v, 5), (bs, cl), generator=g).clamp_(max=v - 1)
    lengths = torch.tensor([rng.randint(1, cl + 1) for _ in range(bs)])
    if kind == "lstm":
        e, h, nl = rng.randint(1, 96), rng.randint(1, 160), rng.randint(1, 3)
        cfg.update(emb=e, hidden=h, layers=nl)
        dec = LSTMDecoder(v, emb_dim=e, hidden_size=h, num_layers=nl, dropout=0.0)
    else:
        heads = rng.choice([1, 2, 3, 4, 8])
        hid = heads * rng.randint(1, 24)
        nl, pf = rng.randint(1, 3), rng.randint(1, 200)
        cfg.update(hid=hid, heads=heads, layers=nl, pf=pf)
        cls = TransformerDecoder if kind == "tfm" else SelfAttentionTransformerDecoder
        dec = cls(v, hid_dim=hid, n_layers=nl, n_heads=heads, pf_dim=pf, dropout=0.0, pad_index=pad, max_len=max(max_len + 1, 64))
    sd = synth_state_dict(dec.state_dict(), seed=77 + idx, logit_std=rng.choice([2.5, 1.0, 4.0]))
    dec.load_state_dict(sd)
    dec.eval()
    osd = {"decoder." + k: t.clone() for k, t in sd.items()}
    kw = dict(caption=cap, max_len=max_len, temperature=temp, beam_size=beam, top_k=top_k)
    seed = 5000 + idx
    with torch.no_grad():
        if kind == "lstm":
            emb = torch.randn(1, 1, cfg["emb"], generator=g)
            want = outcome(lambda: dec.generate(emb, **kw), seed)
            got = outcome(lambda: R.lstm_decoder_generate(osd, "decoder", emb, **kw), seed)
            femb = torch.randn(bs, cfg["emb"], generator=g)
            fw, fg = dec(femb, fcap, lengths), R.lstm_decoder_forward(osd, "decoder", femb, fcap, lengths)
        else:
            start = torch.randn(1, cfg["hid"], generator=g)
            s_len = rng.choice([49, 49, rng.randint(1, 60)])
            enc = torch.randn(1, s_len, cfg["hid"], generator=g) if kind == "tfm" else None
            cfg["enc_len"] = s_len if enc is not None else 0
            args = (start, enc) if kind == "tfm" else (start,)
            want = outcome(lambda: dec.generate(*args, **kw), seed)
            got = outcome(lambda: R.transformer_generate(osd, "decoder", start, enc, pad, cfg["heads"], **kw), seed)
            fstart = torch.randn(bs, cfg["hid"], generator=g)
            fenc = torch.randn(bs, s_len, cfg["hid"], generator=g) if kind == "tfm" else None
            fcap2 = fcap.clone()
            for b in range(bs):
                fcap2[b, max(int(lengths[b]) - 1, 0):] = pad
            fw = dec(fcap2, fenc, fstart) if kind == "tfm" else dec(fcap2, fstart)
            fg = R.transformer_forward(osd, "decoder", fcap2, fenc, fstart, pad, cfg["heads"])
    cfg["gen_ok"] = same(want, got)
    cfg["raised"] = want if isinstance(want, str) else None
    cfg["fwd_ok"] = tuple(fw.shape) == tuple(fg.shape) and float((fw - fg).abs().max()) <= 1e-5 * max(1.0, float(fw.abs().max()))
    if not cfg["gen_ok"]:
        show = lambda t: t if isinstance(t, str) else (list(t.shape), t.reshape(-1).tolist()[:40])
        cfg.update(want=show(want), got=show(got))
    return cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    torch.set_num_threads(4)
    bad = raised = 0
    for i in range(args.trials):
        rng = random.Random(args.seed * 100003 + i)
        try:
            rec = one_trial(rng, i)
            ok = rec["gen_ok"] and rec["fwd_ok"]
        except Exception as e:
            rec, ok = {"error": f"{type(e).__name__}: {e}"[:300]}, False
        raised += bool(rec.get("raised"))
        if not ok:
            bad += 1
            print(json.dumps(dict(i=i, **rec)), flush=True)
    print(json.dumps({"trials": args.trials, "mismatches_or_errors": bad, "both_raised_the_same_error": raised}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
