"""Host text helpers against golden vectors recorded from the reference's
inference.py / vocab.py / tokenizers.py, and the oracle's perplexity against the reference's value."""
import json
import os

import torch

from deephumor_amd.data import CharTokenizer, Vocab, WordPunctTokenizer, build_vocab, SPECIAL_TOKENS
from deephumor_amd.experiments.inference import seq_to_text, split_caption, text_to_seq
from helpers import GOLDEN

G = json.load(open(os.path.join(GOLDEN, "g8_text_and_metrics.json")))


def test_vocab_and_tokenizers():
    wt, ct = WordPunctTokenizer(), CharTokenizer()
    wv, cv = build_vocab(G["docs"], wt, min_df=2), build_vocab(G["docs"], ct, min_df=2)
    assert wv.tokens == G["word_vocab"] and cv.tokens == G["char_vocab"]
    assert [wv.stoi[SPECIAL_TOKENS[k]] for k in ("PAD", "UNK", "BOS", "EOS", "SEP", "EMPTY")] == [0, 1, 2, 3, 4, 5]
    assert len(Vocab(wv.tokens)) == len(wv)
    for c in G["cases"]:
        assert wt.tokenize(c["text"].lower()) == c["word_tokens"] and ct.tokenize(c["text"].lower()) == c["char_tokens"]
        ws, cs = text_to_seq(c["text"], wv, wt), text_to_seq(c["text"], cv, ct)
        assert ws.shape[0] == 1 and ws[0].tolist() == c["word_seq"] and cs[0].tolist() == c["char_seq"]
        assert seq_to_text(torch.cat([ws[0], torch.tensor([3, 7])]), wv) == c["word_text"]
        assert seq_to_text(cs[0], cv, delimiter='') == c["char_text"]


def test_vocab_file_round_trip(tmp_path):
    v = Vocab(G["word_vocab"])
    v.save(str(tmp_path / "v.txt"))
    assert Vocab.load(str(tmp_path / "v.txt")).tokens == v.tokens


def test_split_caption():
    for s in G["splits"]:
        assert split_caption(s["text"]) == s["all"]
        assert split_caption(s["text"], 2) == s["two"] and split_caption(s["text"], 3) == s["three"]


def test_oracle_perplexity_matches_reference():
    from oracle.ref_path import perplexity
    g = torch.Generator().manual_seed(G["perplexity"]["seed"])
    logits = torch.randn(4, 9, 50, generator=g) * 2
    targets = torch.randint(6, 50, (4, 9), generator=g)
    lengths = torch.tensor(G["perplexity"]["lengths"])
    for r, n in enumerate(lengths.tolist()):
        targets[r, n:] = 0
    assert abs(float(perplexity(logits, targets, lengths)) - G["perplexity"]["value"]) < 1e-3 * G["perplexity"]["value"]


def test_resize_oracle_and_host_coefficients_match_pillow_golden():
    """G9: outputs of the real Pillow (Image.resize(..., BILINEAR) = torchvision Resize on PIL images) recorded in the build
    container; the numpy restatement reproduces them bit for bit, and the product's host-side coefficient tables equal the
    oracle's."""
    import hashlib
    import numpy as np
    from helpers import golden
    from oracle.make_resize_golden import CASES, image
    from oracle.resize_ref import coefficients, resize_bilinear_u8
    from deephumor_amd.experiments.inference import resize_coefficients
    g = golden("g9_resize.npz")
    for name, (h, w, oh, ow) in CASES.items():
        res = resize_bilinear_u8(image(name, h, w), oh, ow)
        assert hashlib.sha256(res.tobytes()).digest() == g[f"{name}_sha"].tobytes(), name
        assert (res[:24, :24] == g[f"{name}_corner"]).all() and (res[-8:, -8:] == g[f"{name}_tail"]).all()
        for a, b in ((w, ow), (h, oh)):
            ob, ok = coefficients(a, b)
            pb, pk = resize_coefficients(a, b)
            assert (ob == pb).all() and (ok == pk).all()
    assert (resize_bilinear_u8(image("small", 37, 53), 24, 24) == g["small_full"]).all()
