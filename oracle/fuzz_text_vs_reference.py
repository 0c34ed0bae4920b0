"""Randomised comparison of the host text step against the REAL reference
(``/root/reference/deephumor/data/{tokenizers,vocab}.py``, ``experiments/inference.py`` loaded by path), in the BUILD
CONTAINER only (no reference on the GPU box).  Random strings over a meme-like alphabet (letters, digits, punctuation runs, the
``<sep>`` / ``<emp>`` markers, unicode, whitespace runs), random corpora and ``min_df``, random id sequences incl. special tokens.
TEST INFRASTRUCTURE ONLY.

    python oracle/fuzz_text_vs_reference.py --trials 2000
"""
import argparse
import importlib.util
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

from deephumor.data import CharTokenizer as RChar, WordPunctTokenizer as RWord            # noqa: E402
from deephumor.data.vocab import build_vocab as r_build_vocab                               # noqa: E402
from deephumor_amd.data import CharTokenizer, WordPunctTokenizer, build_vocab              # noqa: E402
from deephumor_amd.experiments import inference as inf                     # noqa: E402


def by_path(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, f"/root/reference/deephumor/experiments/{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


PIECES = ["one", "does", "not", "simply", "walk", "into", "mordor", "y", "u", "no", "i", "don't", "it's", "can't", "gpu", "1337", "3.14",
          "<sep>", "<emp>", "<sep>", "<bos>", "<eos>", "<unk>", "<pad>", "!", "?", "?!", "...", ",", ".", ";", ":", "-", "--", "'", "\"", "(", ")",
          "#", "@", "&", "$5", "100%", "naïve", "über", "日本", "😀", "a_b", "x-y", "C++", "\t", "\n", "  ", "'s", "<", ">", "<sep", "sep>", "<SEP>"]


def rand_text(rng):
    n = rng.randint(0, 14)
    parts = []
    for _ in range(n):
        p = rng.choice(PIECES)
        if rng.random() < 0.15:
            p = p.upper()
        if rng.random() < 0.1:
            p = "".join(rng.choice("abcxyz019!?.,'<> ") for _ in range(rng.randint(1, 6)))
        parts.append(p)
    return rng.choice([" ", " ", "", "  "]).join(parts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rinf = by_path("inference")
    rng = random.Random(args.seed)
    bad = []

    def check(what, a, b, ctx):
        if a != b:
            bad.append({"what": what, "ours": repr(a)[:300], "reference": repr(b)[:300], "ctx": repr(ctx)[:300]})

    pairs = ((WordPunctTokenizer(), RWord()), (CharTokenizer(), RChar()))
    for t in range(args.trials):
        text = rand_text(rng)
        for ours, ref in pairs:
            check("tokenize", ours.tokenize(text), ref.tokenize(text), text)
            check("tokenize.lower", ours.tokenize(text.lower()), ref.tokenize(text.lower()), text)
        for nb in (None, 1, 2, 3):
            try:
                want = ("ok", rinf.split_caption(text, nb))
            except Exception as e:
                want = ("raise", type(e).__name__)
            try:
                got = ("ok", inf.split_caption(text, nb))
            except Exception as e:
                got = ("raise", type(e).__name__)
            check(f"split_caption[{nb}]", got, want, text)
        if t % 10 == 0:                                           # corpora, vocabularies, text <-> ids
            docs = [rand_text(rng) for _ in range(rng.randint(1, 30))] * rng.randint(1, 3)
            min_df = rng.randint(1, 4)
            for ours, ref in pairs:
                v1, v2 = build_vocab(docs, ours, min_df=min_df), r_build_vocab(docs, ref, min_df=min_df)
                check("vocab.tokens", list(v1.tokens), list(v2.tokens), (docs[:3], min_df))
                check("vocab.stoi", dict(v1.stoi), dict(v2.stoi), min_df)
                if list(v1.tokens) != list(v2.tokens):
                    continue
                for _ in range(5):
                    q = rand_text(rng)
                    a, b = inf.text_to_seq(q, v1, ours), rinf.text_to_seq(q, v2, ref)
                    check("text_to_seq", (tuple(a.shape), a.tolist()), (tuple(b.shape), b.tolist()), q)
                    seq = torch.tensor([rng.randrange(len(v1)) for _ in range(rng.randint(0, 12))], dtype=torch.long)
                    for delim in (" ", ""):
                        check("seq_to_text", inf.seq_to_text(seq, v1, delimiter=delim), rinf.seq_to_text(seq, v2, delimiter=delim), seq.tolist())
    print(json.dumps({"trials": args.trials, "mismatches": len(bad)}))
    for b in bad[:25]:
        print(json.dumps(b, ensure_ascii=False))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
