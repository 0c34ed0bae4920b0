"""Pins the CPU oracle (oracle/ref_path.py) against golden vectors recorded from the REAL
reference by oracle/make_golden.py.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import ref_path as R
from helpers import KINDS, PREFIX, captions_and_lengths, golden, synthetic_sd, synth_images

TOL = 1e-5     # same ops in (almost) the same order as the reference


@pytest.fixture(scope="module")
def images():
    return synth_images(4, seed=0)


def test_g1_encoder(images):
    g = golden("g1_encoder.npz")
    sd, _ = synthetic_sd("CaptioningLSTM")
    with torch.no_grad():
        taps = {}
        feats = R.resnet50_trunk(sd, "encoder.resnet", images, taps)
        for idx in (4, 5, 6, 7):
            assert abs(float(taps[f"stage{idx}"].mean()) - float(g[f"e256_stage{idx}_mean"])) < TOL
            assert abs(float(taps[f"stage{idx}"].abs().mean()) - float(g[f"e256_stage{idx}_absmean"])) < TOL
        np.testing.assert_allclose(feats[:, :64].numpy(), g["e256_features_slice"], atol=1e-4, rtol=1e-5)
        emb = R.image_encoder(sd, "encoder", images, False)
        np.testing.assert_allclose(emb.numpy(), g["e256_emb"], atol=TOL, rtol=1e-5)
        # G1 was recorded on a bare ImageEncoder(512, spatial) holder: encoder.* keys only
        from helpers import shapes_to_sd, synth_state_dict
        full, _ = shapes_to_sd("CaptioningTransformer")
        sd = synth_state_dict({k: v for k, v in full.items() if k.startswith("encoder.")}, seed=1234)
        emb, spatial = R.image_encoder(sd, "encoder", images, True)
        np.testing.assert_allclose(emb.numpy(), g["e512_emb"], atol=1e-4, rtol=1e-5)
        np.testing.assert_allclose(spatial.numpy(), g["e512_spatial"], atol=TOL, rtol=1e-5)


@pytest.mark.parametrize("kind", KINDS)
def test_g2_forward_logits(kind, images):
    g = golden(f"g2g3_{kind}.npz")
    sd, hp = synthetic_sd(kind)
    cap, lengths, labels = captions_and_lengths()
    out = R.model_forward(kind, sd, hp, images, cap, lengths, labels if "WithLabels" in kind else None)
    assert tuple(out.shape) == tuple(g["forward_shape"])
    np.testing.assert_allclose(out[:2].numpy(), g["forward_logits01"], atol=2e-4, rtol=1e-5)  # explicit LSTM cell vs mkldnn: 2e-5 on |logit|<=30
    np.testing.assert_allclose(out.sum(-1).numpy(), g["forward_rowsum"], atol=5e-3, rtol=1e-5)
    assert (out.argmax(-1).numpy() == g["forward_argmax"]).mean() > 0.999


@pytest.mark.parametrize("kind", KINDS)
def test_g3_greedy_ids(kind, images):
    g = golden(f"g2g3_{kind}.npz")
    sd, hp = synthetic_sd(kind)
    _, _, labels = captions_and_lengths()
    for i in range(4):
        lab = labels[i:i + 1] if "WithLabels" in kind else None
        for tag, pre in (("", None), ("_prefix", PREFIX)):
            trace = []
            ids = R.model_generate(kind, sd, hp, images[i:i + 1], label=lab, caption=pre, max_len=32,
                                   beam_size=1, top_k=1, trace=trace)
            assert ids.reshape(-1).tolist() == g[f"greedy{tag}_{i}"].tolist()
            margins = np.array([float(t["top2_val"][0, 0] - t["top2_val"][0, 1]) for t in trace])
            np.testing.assert_allclose(margins, g[f"greedy{tag}_margin_{i}"], atol=1e-4)


@pytest.mark.parametrize("kind", KINDS)
def test_g5_beam_rng_replay(kind, images):
    """Stochastic beam search consumes the global CPU generator in the reference's order."""
    g = golden(f"g2g3_{kind}.npz")
    sd, hp = synthetic_sd(kind)
    _, _, labels = captions_and_lengths()
    for i in range(2):
        lab = labels[i:i + 1] if "WithLabels" in kind else None
        torch.manual_seed(100 + i)
        ids = R.model_generate(kind, sd, hp, images[i:i + 1], label=lab, max_len=12, beam_size=3, top_k=20,
                               temperature=1.3)
        assert ids.reshape(-1).tolist() == g[f"beam_{i}"].tolist()


@pytest.mark.parametrize("kind", KINDS)
def test_forced_eos(kind, images):
    """LSTM output includes EOS + one trailing 0; Transformer output excludes the terminator and a
    first-step EOS does not end the beam (SURVEY.md 8(a) G-LSTM step 4 / G-TR steps 1,3)."""
    g = golden(f"g2g3_{kind}.npz")
    sd, hp = synthetic_sd(kind)
    sd["decoder.classifier.bias"] = sd["decoder.classifier.bias"].clone()
    sd["decoder.classifier.bias"][3] += 100.0
    _, _, labels = captions_and_lengths()
    ids = R.model_generate(kind, sd, hp, images[:1], label=labels[:1] if "WithLabels" in kind else None,
                           max_len=8, beam_size=1, top_k=1)
    assert ids.dim() == int(g["forced_eos_ndim"])
    assert ids.reshape(-1).tolist() == g["forced_eos"].tolist()


def test_g4_beam_helper():
    g = golden("g4_beam_helper.npz")
    book = R.BeamBook(1.0, 3, 4)
    out = book.keep_top_k(torch.from_numpy(g["filter_in"].copy()))
    np.testing.assert_array_equal(out.numpy(), g["filter_out"])
    book = R.BeamBook(0.7, 3, 5)
    book.ended = torch.tensor([False, True, False])
    torch.manual_seed(7)
    ps, pv, ni, nv, parent = book.expand(torch.from_numpy(g["pl_logits"].copy()), torch.from_numpy(g["pl_seqs"]),
                                         torch.from_numpy(g["pl_vals"]))
    np.testing.assert_array_equal(ps.numpy(), g["pl_prev_seqs"])
    np.testing.assert_allclose(pv.numpy(), g["pl_prev_vals"].reshape(-1))
    np.testing.assert_array_equal(ni.numpy(), g["pl_new_ind"])
    np.testing.assert_allclose(nv.numpy(), g["pl_new_val"], atol=1e-6)
    np.testing.assert_array_equal(book.ended.numpy(), g["pl_has_ended"])
    assert parent.tolist() == [0, 0, 0, 1, 2, 2, 2]
    # multinomial(p, k) without replacement == top-k of p / Exp(1) noise
    q = torch.from_numpy(g["mn_p"]) / torch.from_numpy(g["mn_noise"])
    np.testing.assert_array_equal(torch.topk(q, 4, dim=-1).indices.numpy(), g["mn_picks"])


@pytest.mark.parametrize("kind", ("CaptioningLSTM", "CaptioningTransformer"))
def test_g3_word_vocab(kind, images):
    g = golden(f"g3_word_{kind}.npz")
    sd, hp = synthetic_sd(kind, v=36541)
    for i in range(2):
        ids = R.model_generate(kind, sd, hp, images[i:i + 1], max_len=32, beam_size=1, top_k=1)
        assert ids.reshape(-1).tolist() == g[f"greedy_{i}"].tolist()


@pytest.mark.parametrize("kind", ("CaptioningLSTM", "CaptioningTransformer"))
def test_g10_char_level(kind):
    """Char-level configuration (V = 71, 127-token captions, beam 7 / top_k 50 / T 1.1: deephumor_demo.ipynb:1307-1309,
    1393-1395): the oracle's greedy ids + margins and its stochastic beam under torch.manual_seed equal the reference's."""
    g = golden(f"g10_char_{kind}.npz")
    sd, hp = synthetic_sd(kind, v=71)
    images = synth_images(2, seed=0)
    trace = []
    ids = R.model_generate(kind, sd, hp, images[:1], max_len=127, beam_size=1, top_k=1, trace=trace)
    assert ids.reshape(-1).tolist() == g["greedy_0"].tolist()
    margins = np.array([float(t["top2_val"][0, 0] - t["top2_val"][0, 1]) for t in trace])
    np.testing.assert_allclose(margins, g["greedy_margin_0"], atol=1e-4)
    if kind == "CaptioningLSTM":                  # (the Transformer oracle re-forwards 7 x 127 positions per token: one is enough here)
        torch.manual_seed(201)
        ids = R.model_generate(kind, sd, hp, images[1:2], max_len=127, beam_size=7, top_k=50, temperature=1.1)
        assert ids.reshape(-1).tolist() == g["beam_1"].tolist()
    sd["decoder.classifier.bias"] = sd["decoder.classifier.bias"].clone()
    sd["decoder.classifier.bias"][3] += 2.5
    torch.manual_seed(300)
    ids = R.model_generate(kind, sd, hp, images[:1], max_len=127, beam_size=7, top_k=50, temperature=1.1)
    assert ids.reshape(-1).tolist() == g["beam_eos_0"].tolist()


@pytest.mark.parametrize("kind", ("CaptioningLSTM", "CaptioningTransformer"))
def test_g11_demo_decode_settings(kind):
    """The demo's word-level decode settings at V = 36,541 (deephumor_demo.ipynb:1264-1266, 1350-1352)."""
    g = golden(f"g11_demo_{kind}.npz")
    sd, hp = synthetic_sd(kind, v=36541)
    images = synth_images(2, seed=0)
    torch.manual_seed(400)
    ids = R.model_generate(kind, sd, hp, images[:1], max_len=12, beam_size=int(g["beam_size"]), top_k=int(g["top_k"]),
                           temperature=float(g["temperature"]))
    assert ids.reshape(-1).tolist() == g["beam_0"].tolist()


@pytest.mark.parametrize("kind", ("CaptioningTransformer", "CaptioningTransformerBase"))
def test_g12_pad_index(kind):
    """pad_index = 7 (transformers.py:393-394 accepts any value): nothing of the encoder is masked, and past 49 positions the zero
    rows the reference pads enc_out with are real keys -- greedy ids at max_len 32 / 60, RNG replay, teacher-forced logits."""
    g = golden("g12_pad_index.npz")
    sd, hp = synthetic_sd(kind)
    hp = dict(hp, pad_index=7)
    images = synth_images(2, seed=0)
    for ml in (32, 60):
        ids = R.model_generate(kind, sd, hp, images[:1], max_len=ml, beam_size=1, top_k=1)
        assert ids.reshape(-1).tolist() == g[f"{kind}_greedy{ml}_0"].tolist()
    torch.manual_seed(500)
    ids = R.model_generate(kind, sd, hp, images[:1], max_len=60, beam_size=3, top_k=20, temperature=1.3)
    assert ids.reshape(-1).tolist() == g[f"{kind}_beam_0"].tolist()
    cap, lengths, _ = captions_and_lengths()
    cap = cap.clone()
    cap[cap == 0] = 7
    out = R.model_forward(kind, sd, hp, images, cap[:2], lengths[:2])
    np.testing.assert_allclose(out.numpy(), g[f"{kind}_forward_logits"], atol=2e-4, rtol=1e-5)


G13_CASES = (("prefix5_len6_beam3", dict(prefix=5, max_len=6, beam_size=3, top_k=20, temperature=1.3)),
             ("prefix5_len6_beam1", dict(prefix=5, max_len=6, beam_size=1, top_k=20, temperature=1.3)),
             ("noprefix_len1_beam3", dict(prefix=0, max_len=1, beam_size=3, top_k=20, temperature=1.3)),
             ("prefix1_len2_beam5", dict(prefix=1, max_len=2, beam_size=5, top_k=5, temperature=0.8)))


@pytest.mark.parametrize("kind", ("CaptioningLSTM", "CaptioningTransformer"))
def test_g13_no_decode_step(kind):
    """The prefix fills max_len - 1 positions: LSTMDecoder.generate's loop runs zero times and its final draw on the [beam, 1]
    first-step scores returns beam_size COPIES of beam 0's row (rnn_models.py:103, 140-141) -- shapes and values recorded from
    the reference; the Transformer runs its (discarded) extra step and returns the usual 1-D caption."""
    g = golden("g13_no_decode_step.npz")
    sd, hp = synthetic_sd(kind)
    images = synth_images(1, seed=0)
    cap, _, _ = captions_and_lengths()
    for name, kw in G13_CASES:
        kw = dict(kw)
        p = kw.pop("prefix")
        torch.manual_seed(600)
        ids = R.model_generate(kind, sd, hp, images, caption=cap[:1, :p] if p else None, **kw)
        want = g[f"{kind}_{name}"]
        assert tuple(ids.shape) == want.shape and ids.reshape(-1).tolist() == want.reshape(-1).tolist(), (kind, name)


@pytest.mark.parametrize("kind", ("CaptioningLSTM", "CaptioningTransformer"))
def test_g14_logits_at_the_word_vocabulary(kind):
    """SURVEY 8(c) G2 "checksums at V=36541": the oracle's teacher-forced logits at the BASELINE vocabulary against the slice
    (first 256 + last 256 columns of every position), row sums and arg-max recorded from the reference; and the pre-filter
    logits of generate's first step."""
    g = golden(f"g14_word_logits_{kind}.npz")
    sd, hp = synthetic_sd(kind, v=36541)
    images = synth_images(2, seed=0)
    cap, lengths, _ = captions_and_lengths(36541)
    out = R.model_forward(kind, sd, hp, images, cap[:2], lengths[:2])
    assert tuple(out.shape) == tuple(g["forward_shape"])
    cols = torch.from_numpy(g["cols"])
    np.testing.assert_allclose(out[:, :, cols].numpy(), g["forward_slice"], atol=2e-4, rtol=1e-5)
    np.testing.assert_allclose(out.double().sum(-1).numpy(), g["forward_rowsum"], atol=5e-2, rtol=1e-5)
    assert (out.argmax(-1).numpy() == g["forward_argmax"]).mean() > 0.999
    trace = []
    R.model_generate(kind, sd, hp, images[:1], max_len=2, beam_size=1, top_k=1, trace=trace)
    assert int(trace[0]["top2_idx"][0, 0]) == int(g["step0_argmax_0"])


@pytest.mark.parametrize("kind", ("CaptioningLSTM", "CaptioningTransformer"))
def test_g15_bench_shape_beam(kind):
    """Stochastic beam search at the BASELINE decode settings (beam 5, top_k 50, 32 tokens, V = 36,541) on bench image 255 under
    torch.manual_seed(700 + 255): the oracle returns the reference's caption (image 0 is checked on the GPU box, where the HIP
    path is compared with both)."""
    g = golden(f"g15_bench_beam_{kind}.npz")
    sd, hp = synthetic_sd(kind, v=36541)
    img = synth_images(1, seed=0, first=255)
    torch.manual_seed(700 + 255)
    ids = R.model_generate(kind, sd, hp, img, max_len=32, beam_size=5, top_k=50, temperature=1.0)
    assert ids.reshape(-1).tolist() == g["beam_255"].tolist()


@pytest.mark.parametrize("kind", ("CaptioningLSTM", "CaptioningTransformer"))
def test_g18_bench_rows(kind):
    """Golden G18 (round 6): sixteen images of the 256-image bench batch at V = 36,541, recorded from the reference for the 16-bit
    gates of ``tests/test_fullsize_gpu.py`` -- the ORACLE is pinned on two of them here: its greedy caption token for token, and its
    first-step logits on the recorded 4,096-column sample + top-8."""
    g = golden(f"g18_bench_rows_{kind}.npz")
    sd, hp = synthetic_sd(kind, v=36541)
    cols = torch.from_numpy(g["cols"])
    for idx in (17, 255):
        img = synth_images(1, seed=0, first=idx)
        ids = R.model_generate(kind, sd, hp, img, max_len=32, beam_size=1, top_k=1)
        assert ids.reshape(-1).tolist() == g[f"greedy_{idx}"].tolist(), (kind, idx)
        row = R.model_forward(kind, sd, hp, img, torch.zeros((1, 0), dtype=torch.int64))[0, 0]
        np.testing.assert_allclose(row[cols].numpy(), g[f"step0_cols_{idx}"], atol=2e-4, rtol=1e-5)
        t8 = torch.from_numpy(g[f"step0_top8_idx_{idx}"])
        np.testing.assert_allclose(row[t8].numpy(), g[f"step0_top8_val_{idx}"], atol=2e-4, rtol=1e-5)
        assert abs(float(row.double().sum()) - float(g[f"step0_rowsum_{idx}"])) < 5e-2
    if kind == "CaptioningTransformer":      # an image of BASELINE config C4's 2,048-image batch (what tests/test_dist_gpu.py compares the HIP path with)
        ids = R.model_generate(kind, sd, hp, synth_images(1, seed=0, first=2047), max_len=32, beam_size=1, top_k=1)
        assert ids.reshape(-1).tolist() == g["greedy_far_2047"].tolist()


@pytest.mark.parametrize("kind", ("CaptioningLSTM", "CaptioningTransformer"))
def test_g16_beam_24(kind):
    """beam_size 24 (any beam_size <= top_k is valid, beam.py:7-9)."""
    g = golden(f"g16_beam24_{kind}.npz")
    sd, hp = synthetic_sd(kind)
    images = synth_images(2, seed=0)
    for i in range(2):
        torch.manual_seed(800 + i)
        ids = R.model_generate(kind, sd, hp, images[i:i + 1], max_len=12, beam_size=24, top_k=50, temperature=1.0)
        assert ids.reshape(-1).tolist() == g[f"beam_{i}"].tolist()


@pytest.mark.parametrize("kind", ("CaptioningTransformer", "CaptioningTransformerBase"))
def test_g17_pad_index_one(kind):
    """pad_index = 1: the image slot's stand-in id 1 (transformers.py:474) is itself padding -- position 0 attends uniformly to
    every position, the real patch rows are the masked encoder keys (:480-481).  Greedy ids, RNG replay, teacher-forced logits."""
    g = golden("g17_pad_index_1.npz")
    sd, hp = synthetic_sd(kind)
    hp = dict(hp, pad_index=1)
    images = synth_images(2, seed=0)
    for ml in (32, 60):
        ids = R.model_generate(kind, sd, hp, images[:1], max_len=ml, beam_size=1, top_k=1)
        assert ids.reshape(-1).tolist() == g[f"{kind}_greedy{ml}_0"].tolist()
    torch.manual_seed(500)
    ids = R.model_generate(kind, sd, hp, images[:1], max_len=60, beam_size=3, top_k=20, temperature=1.3)
    assert ids.reshape(-1).tolist() == g[f"{kind}_beam_0"].tolist()
    cap, lengths, _ = captions_and_lengths()
    cap = cap.clone()
    cap[cap == 0] = 1
    out = R.model_forward(kind, sd, hp, images, cap[:2], lengths[:2])
    np.testing.assert_allclose(out.numpy(), g[f"{kind}_forward_logits"], atol=2e-4, rtol=1e-5)
