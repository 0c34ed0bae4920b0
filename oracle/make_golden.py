"""Generates ``tests/golden/*`` by running the REAL reference (``/root/reference``) in the build
container.  TEST INFRASTRUCTURE ONLY; cannot run on the GPU box (no reference there) and is not
needed there: the fixtures it writes are committed.

    python oracle/make_golden.py            # rewrites tests/golden/
    python oracle/make_golden.py r3         # only the round-3 fixtures (g10 char-level, g11 demo decode settings)
    python oracle/make_golden.py r4         # only the round-4 fixtures (g14 V=36,541 logits, g15 bench-shape beam, g16 beam 24, g17 pad_index 1)
    python oracle/make_golden.py r6         # only the round-6 fixture (g18: 16 images of the bench batch, greedy ids + step-0 logit samples)

The reference imports torchvision (encoders.py:4), which is absent here, so
``oracle/_standin`` (our own ResNet-50 definition, torchvision naming) is put on ``sys.path``
first.  Weights and images come from ``deephumor_amd.synth`` (pure functions of seed + name).
Fixtures are data only: inputs are regenerated from seeds, expected outputs are stored.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "_standin"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

from deephumor.models import (CaptioningLSTM, CaptioningLSTMWithLabels, CaptioningTransformer,   # noqa: E402
                              CaptioningTransformerBase, ImageEncoder, TransformerDecoder)
from deephumor.models.beam import BeamSearchHelper                                              # noqa: E402
from deephumor.models.encoders import LabelEncoder                                              # noqa: E402
from deephumor_amd.synth import load_synthetic, synth_images                                    # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 1234
V_SMALL = 1000
V_WORD = 36541          # deephumor_demo.ipynb:524
V_CHAR = 71             # char-level vocabulary of four of the eight released models (deephumor_demo.ipynb:1307-1309, 1393-1395)


class _WithLabelsTransformer(torch.nn.Module):
    """BASELINE config 5 composition, assembled from reference modules."""

    def __init__(self, num_tokens, hid_dim=512, n_layers=6, n_heads=8, pf_dim=2048, max_len=128):
        super().__init__()
        enc = torch.nn.Module()
        enc.image_encoder = ImageEncoder(hid_dim, 0.3, spatial_features=True)
        enc.label_encoder = LabelEncoder(num_tokens, hid_dim, 0.3)
        enc.linear = torch.nn.Linear(2 * hid_dim, hid_dim)
        self.encoder = enc
        self.decoder = TransformerDecoder(num_tokens, hid_dim, n_layers, n_heads, pf_dim, 0.1, 0, max_len)

    def _start(self, images, labels):
        emb, spatial = self.encoder.image_encoder(images)
        start = self.encoder.linear(torch.cat([emb, self.encoder.label_encoder(labels)], dim=1))
        return start, spatial

    def forward(self, images, captions, lengths, labels):
        start, spatial = self._start(images, labels)
        return self.decoder(captions, enc_out=spatial, start_emb=start)

    def generate(self, image, label, **kw):
        start, spatial = self._start(image, label)
        return self.decoder.generate(start, spatial, **kw)


def build(kind, v):
    cls = {"CaptioningLSTM": CaptioningLSTM, "CaptioningLSTMWithLabels": CaptioningLSTMWithLabels,
           "CaptioningTransformerBase": CaptioningTransformerBase, "CaptioningTransformer": CaptioningTransformer,
           "CaptioningTransformerWithLabels": _WithLabelsTransformer}[kind]
    return load_synthetic(cls(v).eval(), seed=SEED)


def captions_and_lengths(v):
    g = np.random.Generator(np.random.Philox(key=[SEED, 77]))
    cap = torch.from_numpy(g.integers(6, v, size=(4, 31)).astype(np.int64))
    lengths = torch.tensor([32, 20, 32, 11])          # valid inputs incl. the image slot
    for r, n in enumerate(lengths.tolist()):
        cap[r, n - 1:] = 0
    labels = torch.from_numpy(g.integers(6, v, size=(4, 3)).astype(np.int64))
    return cap, lengths, labels


class _LogitTap:
    """Captures the classifier output of every decoder call (pre-filter: cloned before
    BeamSearchHelper.filter_top_k mutates it in place, beam.py:36)."""

    def __init__(self, model):
        self.rows = []
        self.handle = model.decoder.classifier.register_forward_hook(lambda m, i, o: self.rows.append(o.detach().clone()))

    def close(self):
        self.handle.remove()


def greedy_with_margins(model, kind, image, label, caption, max_len=32):
    tap = _LogitTap(model)
    kw = dict(caption=caption, max_len=max_len, beam_size=1, top_k=1)
    with torch.no_grad():
        ids = model.generate(image, label, **kw) if "WithLabels" in kind else model.generate(image, **kw)
    tap.close()
    pos0 = 0 if caption is None else caption.size(1)
    margins, top1 = [], []
    for step, out in enumerate(tap.rows):
        row = out[0] if out.dim() == 2 else out[0, pos0 + step]
        t = torch.topk(row, 2)
        margins.append(float(t.values[0] - t.values[1]))
        top1.append(int(t.indices[0]))
    return ids.reshape(-1).numpy(), np.array(margins, np.float32), np.array(top1, np.int64)


def text_and_metric_goldens():
    """G8: host text helpers (inference.py:11-89, vocab.py, tokenizers.py) and the perplexity metric (metrics.py:4-9),
    recorded from the reference.  ``deephumor.experiments`` cannot be imported as a package here (tensorboard is
    absent), so its two dependency-free files are loaded by path."""
    import importlib.util
    from deephumor.data import WordPunctTokenizer, CharTokenizer
    from deephumor.data.vocab import build_vocab

    def by_path(name):
        spec = importlib.util.spec_from_file_location("ref_" + name, f"/root/reference/deephumor/experiments/{name}.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    inf, met = by_path("inference"), by_path("metrics")
    docs = ["One does not simply <sep> walk into Mordor !", "y u no <sep> use the gpu ?!",
            "i don't always test <sep> but when i do , it's in production...", "such wow <emp> <sep> much kernel",
            "not sure if fast <sep> or just cached", "brace yourselves <sep> the benchmarks are coming"] * 2
    wt, ct = WordPunctTokenizer(), CharTokenizer()
    wv, cv = build_vocab(docs, wt, min_df=2), build_vocab(docs, ct, min_df=2)
    out = {"docs": docs, "word_vocab": wv.tokens, "char_vocab": cv.tokens, "cases": []}
    for c in ("One does not simply walk into MORDOR!!", "y u no compile, bro?", "unknownword <sep> it's fine..."):
        ws, cs = inf.text_to_seq(c, wv, wt), inf.text_to_seq(c, cv, ct)
        out["cases"].append({"text": c, "word_tokens": wt.tokenize(c.lower()), "char_tokens": ct.tokenize(c.lower()),
                             "word_seq": ws[0].tolist(), "char_seq": cs[0].tolist(),
                             "word_text": inf.seq_to_text(torch.cat([ws[0], torch.tensor([3, 7])]), wv),
                             "char_text": inf.seq_to_text(cs[0], cv, delimiter='')})
    splits = ["one does not simply <sep> walk into mordor !", "<bos> top text , with comma <sep>  bottom ... <eos>",
              "no separator here", "a <sep> b <sep> c ?"]
    out["splits"] = [{"text": t, "all": inf.split_caption(t), "two": inf.split_caption(t, 2),
                      "three": inf.split_caption(t, 3)} for t in splits]
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(4, 9, 50, generator=g) * 2
    targets = torch.randint(6, 50, (4, 9), generator=g)
    lengths = torch.tensor([9, 5, 7, 2])
    for r, n in enumerate(lengths.tolist()):
        targets[r, n:] = 0
    out["perplexity"] = {"seed": 5, "value": float(met.perplexity(logits.clone(), targets, lengths, 0)),
                         "lengths": lengths.tolist()}
    with open(os.path.join(OUT, "g8_text_and_metrics.json"), "w") as f:
        json.dump(out, f, indent=1)


def beam_ids(model, image, seed, **kw):
    torch.manual_seed(seed)
    with torch.no_grad():
        return model.generate(image, **kw).reshape(-1).numpy()


def round3_goldens():
    """G10: the char-level configuration (V = 71, captions of up to 127 tokens, beam 7 / top_k 50 / T 1.1 --
    deephumor_demo.ipynb:1132, 1307-1309, 1393-1395): greedy ids + margins for 2 images, and the stochastic beam under
    torch.manual_seed (RNG replay).  max_len = 127 because generate() needs max_len < the Transformer's 128 position slots.
    G11: the demo's word-level decode settings at V = 36,541 (LSTM: beam 10, top_k 100, T 1.3 -- :1264-1266; Transformer:
    beam 10, top_k 70, T 1.0 -- :1350-1352), stochastic beam under torch.manual_seed, max_len 12 to keep it small."""
    torch.set_num_threads(8)
    images = synth_images(2, seed=0)
    for kind in ("CaptioningLSTM", "CaptioningTransformer"):
        model = build(kind, V_CHAR)
        rec = {}
        for i in range(2):
            ids, margins, top1 = greedy_with_margins(model, kind, images[i:i + 1], None, None, max_len=127)
            rec[f"greedy_{i}"], rec[f"greedy_margin_{i}"], rec[f"greedy_top1_{i}"] = ids, margins, top1
            rec[f"beam_{i}"] = beam_ids(model, images[i:i + 1], 200 + i, max_len=127, beam_size=7, top_k=50, temperature=1.1)
        # <eos> made likely: beams end at different steps of a long caption (ended-beam bookkeeping over ~100 positions)
        with torch.no_grad():
            model.decoder.classifier.bias[3] += 2.5
        rec["beam_eos_0"] = beam_ids(model, images[:1], 300, max_len=127, beam_size=7, top_k=50, temperature=1.1)
        np.savez_compressed(os.path.join(OUT, f"g10_char_{kind}.npz"), **rec)
        print(kind, "V=71 greedy0", rec["greedy_0"][:10], "len", len(rec["greedy_0"]), "min margin",
              min(float(rec[f"greedy_margin_{i}"].min()) for i in range(2)), "beam lens", len(rec["beam_0"]), len(rec["beam_eos_0"]))
    for kind, kw in (("CaptioningLSTM", dict(beam_size=10, top_k=100, temperature=1.3)),
                     ("CaptioningTransformer", dict(beam_size=10, top_k=70, temperature=1.0))):
        model = build(kind, V_WORD)
        rec = {"beam_size": np.array(kw["beam_size"]), "top_k": np.array(kw["top_k"]), "temperature": np.float32(kw["temperature"])}
        for i in range(2):
            rec[f"beam_{i}"] = beam_ids(model, images[i:i + 1], 400 + i, max_len=12, **kw)
        np.savez_compressed(os.path.join(OUT, f"g11_demo_{kind}.npz"), **rec)
        print(kind, "V=36541 demo settings", kw, rec["beam_0"])


def pad_index_goldens():
    """G12: a Transformer built with pad_index = 7 (transformers.py:393-394 accepts any value): the encoder-row mask then hides
    nothing (the 0/1 row flags never equal 7, :480-481), <pad> positions are ids 7, an ended beam's 0 tokens are ordinary keys, and
    once max_len + 1 exceeds the 49 image patches the zero rows the reference pads enc_out with (:452) take part in cross-attention
    (K = V = the projection biases).  Greedy ids at max_len 32 (no padded rows) and 60 (12 padded rows), RNG-replay beam 3, and
    teacher-forced logits on captions padded with 7."""
    torch.set_num_threads(8)
    images = synth_images(2, seed=0)
    rec = {}
    for kind in ("CaptioningTransformer", "CaptioningTransformerBase"):
        model = load_synthetic({"CaptioningTransformer": CaptioningTransformer, "CaptioningTransformerBase": CaptioningTransformerBase}[kind](
            V_SMALL, pad_index=7).eval(), seed=SEED)
        with torch.no_grad():
            for ml in (32, 60):
                for i in range(2):
                    rec[f"{kind}_greedy{ml}_{i}"] = model.generate(images[i:i + 1], max_len=ml, beam_size=1, top_k=1).reshape(-1).numpy()
            rec[f"{kind}_beam_0"] = beam_ids(model, images[:1], 500, max_len=60, beam_size=3, top_k=20, temperature=1.3)
            cap, lengths, _ = captions_and_lengths(V_SMALL)
            cap = cap.clone()
            cap[cap == 0] = 7
            logits = model(images, cap[:2], lengths[:2])
            rec[f"{kind}_forward_logits"] = logits.numpy()
        print(kind, "pad_index=7", rec[f"{kind}_greedy60_0"][:8], len(rec[f"{kind}_greedy60_0"]), logits.shape)
    np.savez_compressed(os.path.join(OUT, "g12_pad_index.npz"), **rec)


def round4_goldens():
    """G14: logits at the BASELINE vocabulary (V = 36,541; SURVEY 8(c) G2 "checksums at V=36541"): teacher-forced ``forward()`` of
    2 images -- row sums, arg-max and a 512-column slice (the first 256 and the LAST 256 columns: the partial 128-column panel at the
    end of the vocabulary) of every position -- and the pre-filter logits of ``generate``'s first step (same slice + row sum).
    G15: stochastic beam search at the BASELINE decode settings (beam 5, top_k 50, T 1.0, 32 tokens, V = 36,541) for images 0 and
    255 of the 256-image bench batch under ``torch.manual_seed(700 + index)`` -- what ``generate(..., rng="torch")`` must return.
    G16: beam_size 24 > 16 (beam.py:7-9 allows any beam_size <= top_k), top_k 50, V = 1,000, under ``torch.manual_seed(800 + i)``.
    G17: a Transformer built with pad_index = 1 (transformers.py:393-394): the image slot's stand-in id 1 (:474) then counts as
    padding -- greedy ids, RNG-replay beam 3, teacher-forced logits."""
    torch.set_num_threads(8)
    images = synth_images(4, seed=0)
    cols = np.r_[0:256, V_WORD - 256:V_WORD]
    cap, lengths, _ = captions_and_lengths(V_WORD)
    for kind in ("CaptioningLSTM", "CaptioningTransformer"):
        model = build(kind, V_WORD)
        rec = {"cols": cols}
        with torch.no_grad():
            logits = model(images[:2], cap[:2], lengths[:2])
        rec["forward_shape"] = np.array(logits.shape)
        rec["forward_slice"] = logits[:, :, cols].numpy()
        rec["forward_rowsum"] = logits.double().sum(-1).numpy()
        rec["forward_argmax"] = logits.argmax(-1).numpy()
        for i in range(2):
            tap = _LogitTap(model)
            with torch.no_grad():
                model.generate(images[i:i + 1], max_len=2, beam_size=1, top_k=1)
            tap.close()
            out = tap.rows[0]
            row = out[0] if out.dim() == 2 else out[0, 0]
            rec[f"step0_slice_{i}"] = row[cols].numpy()
            rec[f"step0_rowsum_{i}"] = np.float64(row.double().sum())
            rec[f"step0_argmax_{i}"] = np.int64(row.argmax())
        np.savez_compressed(os.path.join(OUT, f"g14_word_logits_{kind}.npz"), **rec)
        print(kind, "V=36541 forward", tuple(logits.shape), "rowsum[0,:3]", rec["forward_rowsum"][0, :3])
        rec = {}
        for idx in (0, 255):
            img = synth_images(1, seed=0, first=idx)
            rec[f"beam_{idx}"] = beam_ids(model, img, 700 + idx, max_len=32, beam_size=5, top_k=50, temperature=1.0)
            print(kind, "bench image", idx, "beam 5:", rec[f"beam_{idx}"][:10], len(rec[f"beam_{idx}"]))
        np.savez_compressed(os.path.join(OUT, f"g15_bench_beam_{kind}.npz"), **rec)
    for kind in ("CaptioningLSTM", "CaptioningTransformer"):
        model = build(kind, V_SMALL)
        rec = {}
        for i in range(2):
            rec[f"beam_{i}"] = beam_ids(model, images[i:i + 1], 800 + i, max_len=12, beam_size=24, top_k=50, temperature=1.0)
        np.savez_compressed(os.path.join(OUT, f"g16_beam24_{kind}.npz"), **rec)
        print(kind, "beam 24", rec["beam_0"])
    rec = {}
    for kind in ("CaptioningTransformer", "CaptioningTransformerBase"):
        model = load_synthetic({"CaptioningTransformer": CaptioningTransformer, "CaptioningTransformerBase": CaptioningTransformerBase}[kind](
            V_SMALL, pad_index=1).eval(), seed=SEED)
        with torch.no_grad():
            for ml in (32, 60):
                for i in range(2):
                    rec[f"{kind}_greedy{ml}_{i}"] = model.generate(images[i:i + 1], max_len=ml, beam_size=1, top_k=1).reshape(-1).numpy()
            rec[f"{kind}_beam_0"] = beam_ids(model, images[:1], 500, max_len=60, beam_size=3, top_k=20, temperature=1.3)
            c1, l1, _ = captions_and_lengths(V_SMALL)
            c1 = c1.clone()
            c1[c1 == 0] = 1
            logits = model(images[:2], c1[:2], l1[:2])
            rec[f"{kind}_forward_logits"] = logits.numpy()
        print(kind, "pad_index=1", rec[f"{kind}_greedy60_0"][:8], len(rec[f"{kind}_greedy60_0"]), logits.shape)
    np.savez_compressed(os.path.join(OUT, "g17_pad_index_1.npz"), **rec)


G18_IMAGES = [17 * i for i in range(16)]          # 0, 17, ..., 255: sixteen images of the 256-image bench batch
G18_COLS = 4096


def round6_goldens():
    """G18: the 16-bit gates at the BASELINE shape need more than three oracle rows, and the GPU run should not pay
    for them -- recorded here from the REAL reference, V = 36,541, synthetic weights (seed 1234), images ``G18_IMAGES`` of
    ``synth_images(256, seed=0)``: the greedy caption (32 tokens), the per-step top-1 / top-2 margin, and of the pre-filter logits of the
    first decode step a fixed random sample of 4,096 columns + the top-8 (index, value) pairs + the row sum."""
    torch.set_num_threads(8)
    g = np.random.Generator(np.random.Philox(key=[SEED, 18]))
    cols = np.sort(g.choice(V_WORD, size=G18_COLS, replace=False)).astype(np.int64)
    for kind in ("CaptioningLSTM", "CaptioningTransformer"):
        model = build(kind, V_WORD)
        rec = {"images": np.array(G18_IMAGES, np.int64), "cols": cols}
        for idx in G18_IMAGES:
            img = synth_images(1, seed=0, first=idx)
            tap = _LogitTap(model)
            with torch.no_grad():
                ids = model.generate(img, max_len=32, beam_size=1, top_k=1)
            tap.close()
            out = tap.rows[0]
            row = (out[0] if out.dim() == 2 else out[0, 0]).float()
            t8 = torch.topk(row, 8)
            margins = []
            for step, o in enumerate(tap.rows):
                r = o[0] if o.dim() == 2 else o[0, step]
                t2 = torch.topk(r, 2)
                margins.append(float(t2.values[0] - t2.values[1]))
            rec[f"greedy_{idx}"] = ids.reshape(-1).numpy().astype(np.int64)
            rec[f"margins_{idx}"] = np.array(margins, np.float32)
            rec[f"step0_cols_{idx}"] = row[torch.from_numpy(cols)].numpy()
            rec[f"step0_top8_idx_{idx}"] = t8.indices.numpy().astype(np.int64)
            rec[f"step0_top8_val_{idx}"] = t8.values.numpy()
            rec[f"step0_rowsum_{idx}"] = np.float64(row.double().sum())
            print(kind, "image", idx, "greedy", rec[f"greedy_{idx}"][:6], len(rec[f"greedy_{idx}"]), "min margin", min(margins), flush=True)
        if kind == "CaptioningTransformer":
            # BASELINE config C4's 2,048-image global batch: two images far outside the first 256 (tests/test_dist_gpu.py)
            for idx in (1000, 2047):
                with torch.no_grad():
                    rec[f"greedy_far_{idx}"] = model.generate(synth_images(1, seed=0, first=idx), max_len=32, beam_size=1, top_k=1).reshape(-1).numpy().astype(np.int64)
                print(kind, "C4 image", idx, rec[f"greedy_far_{idx}"][:6], flush=True)
        np.savez_compressed(os.path.join(OUT, f"g18_bench_rows_{kind}.npz"), **rec)



def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    images = synth_images(4, seed=0)
    meta = {"seed": SEED, "v_small": V_SMALL, "v_word": V_WORD, "torch": torch.__version__, "models": {}}

    # ---- G1: encoder ------------------------------------------------------------------
    g1 = {}
    for e in (256, 512):
        enc = ImageEncoder(e, 0.3, spatial_features=(e == 512)).eval()
        holder = torch.nn.Module()
        holder.encoder = enc
        load_synthetic(holder, seed=SEED)
        with torch.no_grad():
            x = images
            for i, mod in enumerate(enc.resnet):
                x = mod(x)
                if i >= 3:
                    g1[f"e{e}_stage{i}_mean"] = np.float32(x.mean())
                    g1[f"e{e}_stage{i}_absmean"] = np.float32(x.abs().mean())
            g1[f"e{e}_features_slice"] = x[:, :64].numpy().copy()
            out = enc(images)
        if e == 512:
            g1["e512_emb"], g1["e512_spatial"] = out[0].numpy(), out[1].numpy()
        else:
            g1["e256_emb"] = out.numpy()
    np.savez_compressed(os.path.join(OUT, "g1_encoder.npz"), **g1)

    cap, lengths, labels = captions_and_lengths(V_SMALL)
    prefix = torch.tensor([[17, 230, 45]])
    for kind in ("CaptioningLSTM", "CaptioningLSTMWithLabels", "CaptioningTransformerBase",
                 "CaptioningTransformer", "CaptioningTransformerWithLabels"):
        model = build(kind, V_SMALL)
        sd = model.state_dict()
        meta["models"][kind] = {"hp": getattr(model, "_hp", None),
                                "keys": {k: list(v.shape) for k, v in sd.items()}}
        rec = {}
        # ---- G2: teacher-forced forward logits (first two images stored in full) ---------
        with torch.no_grad():
            if "WithLabels" in kind:
                logits = model(images, cap, lengths, labels)
            else:
                logits = model(images, cap, lengths)
        rec["forward_shape"] = np.array(logits.shape)
        rec["forward_logits01"] = logits[:2].numpy()
        rec["forward_rowsum"] = logits.sum(-1).numpy()
        rec["forward_argmax"] = logits.argmax(-1).numpy()
        # ---- G3: greedy ids + margins -----------------------------------------------------
        for i in range(4):
            lab = labels[i:i + 1] if "WithLabels" in kind else None
            for tag, pre in (("", None), ("_prefix", prefix)):
                ids, margins, top1 = greedy_with_margins(model, kind, images[i:i + 1], lab, pre)
                rec[f"greedy{tag}_{i}"], rec[f"greedy{tag}_margin_{i}"], rec[f"greedy{tag}_top1_{i}"] = ids, margins, top1
        # ---- G5: stochastic beam replay (torch CPU RNG stream) ----------------------------
        for i in range(2):
            lab = labels[i:i + 1] if "WithLabels" in kind else None
            torch.manual_seed(100 + i)
            kw = dict(max_len=12, beam_size=3, top_k=20, temperature=1.3)
            with torch.no_grad():
                ids = model.generate(images[i:i + 1], lab, **kw) if "WithLabels" in kind else model.generate(images[i:i + 1], **kw)
            rec[f"beam_{i}"] = ids.reshape(-1).numpy()
        # ---- EOS behaviour: force EOS as the arg-max token ---------------------------------
        with torch.no_grad():
            model.decoder.classifier.bias[3] += 100.0
            lab = labels[:1] if "WithLabels" in kind else None
            kw = dict(max_len=8, beam_size=1, top_k=1)
            ids = model.generate(images[:1], lab, **kw) if "WithLabels" in kind else model.generate(images[:1], **kw)
            rec["forced_eos"] = ids.reshape(-1).numpy()
            rec["forced_eos_ndim"] = np.array(ids.dim())
            model.decoder.classifier.bias[3] -= 100.0
        np.savez_compressed(os.path.join(OUT, f"g2g3_{kind}.npz"), **rec)
        print(kind, "forward", tuple(logits.shape), "greedy0", rec["greedy_0"][:10],
              "min margin", min(float(rec[f'greedy_margin_{i}'].min()) for i in range(4)))

    # ---- G3 at the word vocabulary (BASELINE configs C1-C3) -----------------------------
    for kind in ("CaptioningLSTM", "CaptioningTransformer"):
        model = build(kind, V_WORD)
        rec = {}
        for i in range(4):
            ids, margins, top1 = greedy_with_margins(model, kind, images[i:i + 1], None, None)
            rec[f"greedy_{i}"], rec[f"greedy_margin_{i}"], rec[f"greedy_top1_{i}"] = ids, margins, top1
        np.savez_compressed(os.path.join(OUT, f"g3_word_{kind}.npz"), **rec)
        print(kind, "V=36541 greedy0", rec["greedy_0"][:10],
              "min margin", min(float(rec[f'greedy_margin_{i}'].min()) for i in range(4)))

    # ---- G4: BeamSearchHelper unit vectors --------------------------------------------------
    g4 = {}
    h = BeamSearchHelper(temperature=1.0, beam_size=3, top_k=4, device="cpu")
    lg = torch.tensor([[0.5, 9.0, 3.0, 3.0, 2.0, 3.0, -1.0, 2.5],       # unk(1) is the max; tie at the 4th value
                       [4.0, 0.0, 4.0, 1.0, 4.0, 4.0, 4.0, -2.0],       # 5-way tie at the threshold
                       [-3.0, -2.0, -1.0, 0.0, 1.0, 2.0, 3.0, 4.0]])
    g4["filter_in"] = lg.numpy().copy()
    g4["filter_out"] = h.filter_top_k(lg.clone()).numpy()
    g = np.random.Generator(np.random.Philox(key=[SEED, 99]))
    lg = torch.from_numpy(g.standard_normal(size=(3, 40)).astype(np.float32) * 2)
    lg[2, 3] = 30.0                                  # beam 2 will pick EOS first
    seqs = torch.tensor([[7, 8], [9, 3], [11, 12]])
    vals = torch.tensor([-0.5, -1.0, -2.0])
    h = BeamSearchHelper(temperature=0.7, beam_size=3, top_k=5, device="cpu")
    h.has_ended = torch.tensor([False, True, False])
    torch.manual_seed(7)
    (ps, pv), (ni, nv) = h.process_logits(lg.clone(), seqs, vals)
    g4.update(pl_logits=lg.numpy(), pl_seqs=seqs.numpy(), pl_vals=vals.numpy(), pl_prev_seqs=ps.numpy(),
              pl_prev_vals=pv.numpy(), pl_new_ind=ni.numpy(), pl_new_val=nv.numpy(), pl_has_ended=h.has_ended.numpy())
    # multinomial == top-k of p / Exp(1) noise drawn with the same generator state
    p = torch.softmax(torch.from_numpy(g.standard_normal(size=(3, 50)).astype(np.float32)), -1)
    torch.manual_seed(11)
    picks = torch.multinomial(p, 4)
    torch.manual_seed(11)
    q = torch.empty_like(p).exponential_(1)
    g4.update(mn_p=p.numpy(), mn_picks=picks.numpy(), mn_noise=q.numpy())
    np.savez_compressed(os.path.join(OUT, "g4_beam_helper.npz"), **g4)

    text_and_metric_goldens()
    round3_goldens()
    pad_index_goldens()
    round4_goldens()
    with open(os.path.join(OUT, "golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", OUT)


def edge_goldens():
    """G13: ``generate`` when the prefix already fills ``max_len - 1`` positions.  LSTMDecoder.generate's loop
    (rnn_models.py:103) then runs zero times and the final draw (:140-141) is taken on the [beam, 1] scores of the first step:
    a [beam, 1] index tensor of zeros, so ``sample_seq[ind, :].squeeze()`` is ``beam_size`` copies of beam 0's row -- a 2-D
    result (1-D for beam_size = 1 or a single column).  The Transformer's loop (transformers.py:546) always runs once more
    (its last step is discarded), so it returns the usual 1-D caption.  Found by tools/fuzz_generate.py."""
    torch.set_num_threads(8)
    images = synth_images(1, seed=0)
    cap, _, _ = captions_and_lengths(V_SMALL)
    rec = {}
    for kind in ("CaptioningLSTM", "CaptioningTransformer"):
        model = build(kind, V_SMALL)
        for name, kw in (("prefix5_len6_beam3", dict(caption=cap[:1, :5], max_len=6, beam_size=3, top_k=20, temperature=1.3)),
                         ("prefix5_len6_beam1", dict(caption=cap[:1, :5], max_len=6, beam_size=1, top_k=20, temperature=1.3)),
                         ("noprefix_len1_beam3", dict(max_len=1, beam_size=3, top_k=20, temperature=1.3)),
                         ("prefix1_len2_beam5", dict(caption=cap[:1, :1], max_len=2, beam_size=5, top_k=5, temperature=0.8))):
            torch.manual_seed(600)
            with torch.no_grad():
                out = model.generate(images, **kw)
            rec[f"{kind}_{name}"] = out.numpy()
            print(kind, name, tuple(out.shape), out.reshape(-1)[:12].tolist())
    np.savez_compressed(os.path.join(OUT, "g13_no_decode_step.npz"), **rec)


if __name__ == "__main__":
    if sys.argv[1:] == ["edge"]:
        edge_goldens()
    elif sys.argv[1:] == ["r3"]:
        round3_goldens()
        pad_index_goldens()
    elif sys.argv[1:] == ["r4"]:
        round4_goldens()
    elif sys.argv[1:] == ["pad"]:
        pad_index_goldens()
    elif sys.argv[1:] == ["r6"]:
        round6_goldens()
    else:
        main()
