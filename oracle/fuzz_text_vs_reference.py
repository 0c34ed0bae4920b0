"""TEST INFRASTRUCTURE, build container only (imports the real reference from /root/reference, as oracle/make_golden.py does): the host
text step of this package -- tokenizers, ``build_vocab``, ``text_to_seq``, ``seq_to_text``, ``split_caption`` (SURVEY 8(f) rank 3;
reference deephumor/data/tokenizers.py, vocab.py, experiments/inference.py) -- against the reference's own functions on 400 random
strings (unicode, special tokens, empty and whitespace-only texts, both tokenizers, three ``min_df``): 0 differences on the round-5 tree.

    python oracle/fuzz_text_vs_reference.py
"""
import sys, random, json, importlib
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deephumor_amd.data.tokenizers as T
import deephumor_amd.data.vocab as Vm
import deephumor_amd.experiments.inference as I
sys.path.insert(0, "/root/reference")
rT = importlib.import_module("deephumor.data.tokenizers")
rV = importlib.import_module("deephumor.data.vocab")
import importlib.util
try:
    spec = importlib.util.spec_from_file_location("ref_inference", "/root/reference/deephumor/experiments/inference.py")
    rI = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rI)
except Exception as e:
    rI = None
    print("reference inference not importable:", repr(e)[:200])
rng = random.Random(5)
alphabet = "abc XYZ  \t\n.,!?'\"-_<>|#@0123456789éüñ😀中文" + "<sep> <emp> <unk> <pad> <eos> <bos> "
def rand_text():
    n = rng.choice([0, 1, 2, 5, 20, 80])
    parts = []
    for _ in range(n):
        r = rng.random()
        if r < 0.2: parts.append(rng.choice(["<sep>", "<emp>", "<unk>", "<pad>", "<eos>", " ", "  ", "\n"]))
        else: parts.append("".join(rng.choice(alphabet) for _ in range(rng.randint(1, 6))))
    return rng.choice(["", " "]).join(parts)
bad = 0
docs = [rand_text() for _ in range(400)]
for tk_name in ("WordPunctTokenizer", "CharTokenizer"):
    a, b = getattr(T, tk_name)(), getattr(rT, tk_name)()
    for d in docs:
        if a.tokenize(d) != b.tokenize(d):
            bad += 1; print("tokenize differs", tk_name, repr(d)[:80])
    for min_df in (1, 2, 7):
        va, vb = Vm.build_vocab(docs, a, min_df=min_df), rV.build_vocab(docs, b, min_df=min_df)
        if list(va.stoi.items()) != list(vb.stoi.items()) or va.itos != vb.itos:
            bad += 1; print("vocab differs", tk_name, min_df, len(va.stoi), len(vb.stoi))
        if rI is not None:
            for d in docs[:200]:
                sa, sb = I.text_to_seq(d, va, a), rI.text_to_seq(d, vb, b)
                if tuple(sa.shape) != tuple(sb.shape) or sa.reshape(-1).tolist() != sb.reshape(-1).tolist():
                    bad += 1; print("text_to_seq differs", repr(d)[:60])
                ids = [rng.randrange(len(va.itos)) for _ in range(rng.randint(0, 30))]
                import torch
                for delim in (" ", ""):
                    ta, tb = I.seq_to_text(torch.tensor(ids), va, delim), rI.seq_to_text(torch.tensor(ids), vb, delim)
                    if ta != tb:
                        bad += 1; print("seq_to_text differs", ids[:10], repr(ta)[:50], repr(tb)[:50])
            for d in docs[:300]:
                for nb in (None, 1, 2, 3):
                    try: ra = ("ok", I.split_caption(d, nb))
                    except Exception as e: ra = ("exc", type(e).__name__)
                    try: rb = ("ok", rI.split_caption(d, nb))
                    except Exception as e: rb = ("exc", type(e).__name__)
                    if ra != rb:
                        bad += 1; print("split_caption differs", repr(d)[:60], nb, str(ra)[:80], str(rb)[:80])
print(json.dumps({"docs": len(docs), "differences": bad, "reference_inference_imported": rI is not None}))
# a difference, or a reference module that did not import (its three checks were skipped), is a failure of this script
sys.exit(1 if (bad or rI is None) else 0)
