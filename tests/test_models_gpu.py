"""End-to-end parity of the HIP path on a real MI355X against (a) golden vectors recorded from the
real reference and (b) the CPU oracle on the same seeded inputs.  Bar (BASELINE.json north_star):
greedy token ids bit-exact, logits within 1e-3 (fp32)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import KINDS, PREFIX, captions_and_lengths, golden, synthetic_sd, synth_images  # noqa: E402

LOGIT_TOL = 1e-3


def build(kind, v=None):
    import deephumor_amd.models as M
    sd, hp = synthetic_sd(kind, v)
    model = getattr(M, kind)(**hp).eval()
    model.load_state_dict(sd)
    return model.cuda(), sd, hp


@pytest.fixture(scope="module")
def images():
    return synth_images(4, seed=0)


def test_native_library_is_loaded():
    from deephumor_amd import hip
    lib = hip.load()
    assert lib.dh_abi_version() == hip.ABI_VERSION
    with open("/proc/self/maps") as f:
        assert "libdeephumor_hip.so" in f.read()


def test_encoder_matches_reference(images):
    g = golden("g1_encoder.npz")
    model, _, _ = build("CaptioningLSTM")
    with torch.no_grad():
        feats = model.encoder.features(images.cuda())
        np.testing.assert_allclose(feats[:, :64].cpu().numpy(), g["e256_features_slice"], atol=1e-3, rtol=1e-4)
        np.testing.assert_allclose(model.encoder(images.cuda()).cpu().numpy(), g["e256_emb"], atol=1e-4, rtol=1e-4)
    import deephumor_amd.models as M
    from helpers import shapes_to_sd, synth_state_dict
    full, _ = shapes_to_sd("CaptioningTransformer")
    sd = synth_state_dict({k: v for k, v in full.items() if k.startswith("encoder.")}, seed=1234)
    enc = M.ImageEncoder(512, 0.3, spatial_features=True).eval()
    enc.load_state_dict({k[len("encoder."):]: v for k, v in sd.items()})
    with torch.no_grad():
        emb, spatial = enc.cuda()(images.cuda())
    np.testing.assert_allclose(emb.cpu().numpy(), g["e512_emb"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(spatial.cpu().numpy(), g["e512_spatial"], atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("kind", KINDS)
def test_forward_logits(kind, images):
    """Teacher-forced forward() (trainer.py:69-73 call shape) vs the reference's logits."""
    g = golden(f"g2g3_{kind}.npz")
    model, _, _ = build(kind)
    cap, lengths, labels = captions_and_lengths()
    with torch.no_grad():
        if "WithLabels" in kind:
            out = model(images.cuda(), cap.cuda(), lengths, labels.cuda())
        else:
            out = model(images.cuda(), cap.cuda(), lengths)
    assert tuple(out.shape) == tuple(g["forward_shape"])
    out = out.cpu()
    np.testing.assert_allclose(out[:2].numpy(), g["forward_logits01"], atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(out.sum(-1).numpy(), g["forward_rowsum"], atol=2e-2, rtol=1e-4)


@pytest.mark.parametrize("kind", KINDS)
def test_greedy_ids_bit_exact(kind, images):
    """generate(beam_size=1, top_k=1) token ids == the reference's, per image and batched."""
    g = golden(f"g2g3_{kind}.npz")
    model, _, _ = build(kind)
    _, _, labels = captions_and_lengths()
    for tag, pre in (("", None), ("_prefix", PREFIX)):
        kw = dict(max_len=32, beam_size=1, top_k=1)
        captured = {}
        with torch.no_grad():
            args = (images.cuda(), labels.cuda()) if "WithLabels" in kind else (images.cuda(),)
            cap = None if pre is None else pre.repeat(4, 1).cuda()
            toks, lens = model.generate_batch(*args, caption=cap, logits_hook=lambda i, lg: captured.setdefault(i, lg[:4].cpu().clone()), **kw)
        toks, lens = toks.cpu(), lens.cpu()
        for i in range(4):
            want = g[f"greedy{tag}_{i}"].tolist()
            assert toks[i, :int(lens[i])].tolist() == want, (kind, tag, i)
            assert toks[i, int(lens[i]):].abs().sum() == 0
        # logits of the first sampled position: top-1 and margin as recorded from the reference
        pos0 = 0 if pre is None else pre.shape[1]
        top = torch.topk(captured[pos0], 2, dim=-1)
        for i in range(4):
            assert int(top.indices[i, 0]) == int(g[f"greedy{tag}_top1_{i}"][0])
            assert abs(float(top.values[i, 0] - top.values[i, 1]) - float(g[f"greedy{tag}_margin_{i}"][0])) < LOGIT_TOL
    # reference single-image API and return shape
    with torch.no_grad():
        args = (images[:1].cuda(), labels[:1].cuda()) if "WithLabels" in kind else (images[:1].cuda(),)
        one = model.generate(*args, max_len=32, beam_size=1, top_k=1)
    assert one.dtype == torch.int64 and one.is_cuda and one.cpu().tolist() == g["greedy_0"].tolist()


@pytest.mark.parametrize("kind", KINDS)
def test_forced_eos_shapes(kind, images):
    g = golden(f"g2g3_{kind}.npz")
    model, _, _ = build(kind)
    _, _, labels = captions_and_lengths()
    with torch.no_grad():
        model.decoder.classifier.bias[3] += 100.0
        args = (images[:1].cuda(), labels[:1].cuda()) if "WithLabels" in kind else (images[:1].cuda(),)
        ids = model.generate(*args, max_len=8, beam_size=1, top_k=1)
    assert ids.dim() == int(g["forced_eos_ndim"])
    assert ids.reshape(-1).cpu().tolist() == g["forced_eos"].tolist()


class _Replay:
    """Feeds the kernels the Exp(1) noise torch.multinomial would draw from the CPU generator, in the
    reference's order and shapes (RNG draw schedule)."""

    def __init__(self, helper_getter):
        self.get = helper_getter

    def __call__(self, kind, step, shape):
        h = self.get()
        b = h.beam_size
        if kind == "row":
            if bool(h.done.cpu()[0]):
                return torch.ones(shape)
            return torch.empty(shape).exponential_(1)
        if kind == "cand":
            if bool(h.done.cpu()[0]):
                return torch.ones(shape)
            n_cand = int(sum(1 if e else b for e in h.has_ended.cpu().tolist()))
            out = torch.ones(shape)
            out[0, :n_cand] = torch.empty(n_cand).exponential_(1)
            return out
        return torch.empty(shape[1]).exponential_(1)[None]


@pytest.mark.parametrize("kind", KINDS)
def test_beam_rng_replay(kind, images):
    """Stochastic beam search (beam=3, top_k=20, T=1.3) reproduces the reference token-for-token when the
    kernels are fed the CPU generator's noise."""
    import deephumor_amd.models.beam as beam_mod
    g = golden(f"g2g3_{kind}.npz")
    model, _, _ = build(kind)
    _, _, labels = captions_and_lengths()
    made = []
    orig = beam_mod.BeamSearchHelper.__init__

    def spy(self, *a, **k):
        orig(self, *a, **k)
        made.append(self)

    beam_mod.BeamSearchHelper.__init__ = spy
    try:
        for i in range(2):
            torch.manual_seed(100 + i)
            args = (images[i:i + 1].cuda(), labels[i:i + 1].cuda()) if "WithLabels" in kind else (images[i:i + 1].cuda(),)
            with torch.no_grad():
                ids = model.generate(*args, max_len=12, beam_size=3, top_k=20, temperature=1.3,
                                     noise_source=_Replay(lambda: made[-1]))
            assert ids.reshape(-1).cpu().tolist() == g[f"beam_{i}"].tolist(), (kind, i)
    finally:
        beam_mod.BeamSearchHelper.__init__ = orig


@pytest.mark.parametrize("kind", ("CaptioningLSTM", "CaptioningTransformer"))
def test_word_vocab_greedy(kind, images):
    """BASELINE configs C1-C3 vocabulary (V=36,541): greedy ids equal the reference's."""
    g = golden(f"g3_word_{kind}.npz")
    model, _, _ = build(kind, v=36541)
    top2 = []
    with torch.no_grad():
        toks, lens = model.generate_batch(images.cuda(), max_len=32, beam_size=1, top_k=1,
                                          logits_hook=lambda i, lg: top2.append(torch.topk(lg[:4].float(), 2, dim=-1)))
    for i in range(4):
        assert toks[i, :int(lens[i])].cpu().tolist() == g[f"greedy_{i}"].tolist()
        # the recorded top-1 ids and top-1 / top-2 margins of every step the reference ran (pre-filter logits)
        n = len(g[f"greedy_margin_{i}"])
        got_top1 = [int(t.indices[i, 0]) for t in top2[:n]]
        got_margin = np.array([float(t.values[i, 0] - t.values[i, 1]) for t in top2[:n]])
        assert got_top1 == g[f"greedy_top1_{i}"].tolist()
        np.testing.assert_allclose(got_margin, g[f"greedy_margin_{i}"], atol=2 * LOGIT_TOL, rtol=0)


@pytest.mark.parametrize("kind", ("CaptioningLSTM", "CaptioningTransformer"))
def test_word_vocab_logits(kind):
    """"Logits within 1e-3 fp32" AT the BASELINE vocabulary (V = 36,541; SURVEY 8(c) G2, golden G14 recorded from the reference):
    teacher-forced ``forward()`` -- the first 256 and LAST 256 columns of every position (the partial 128-column panel at the end
    of the vocabulary), row sums, arg-max -- and the pre-filter logits of ``generate``'s first step."""
    g = golden(f"g14_word_logits_{kind}.npz")
    model, _, _ = build(kind, v=36541)
    imgs = synth_images(2, seed=0)
    cap, lengths, _ = captions_and_lengths(36541)
    with torch.no_grad():
        out = model(imgs.cuda(), cap[:2].cuda(), lengths[:2]).cpu()
    assert tuple(out.shape) == tuple(g["forward_shape"])
    cols = torch.from_numpy(g["cols"])
    np.testing.assert_allclose(out[:, :, cols].numpy(), g["forward_slice"], atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(out.double().sum(-1).numpy(), g["forward_rowsum"], atol=0.5, rtol=0)      # 36,541 terms of |err| << 1e-3
    assert (out.argmax(-1).numpy() == g["forward_argmax"]).mean() > 0.999
    step0 = []
    with torch.no_grad():
        model.generate_batch(imgs.cuda(), max_len=2, beam_size=1, top_k=1,
                             logits_hook=lambda i, lg: step0.append(lg[:2].float().cpu().clone()))
    for i in range(2):
        np.testing.assert_allclose(step0[0][i, cols].numpy(), g[f"step0_slice_{i}"], atol=LOGIT_TOL, rtol=0)
        assert int(step0[0][i].argmax()) == int(g[f"step0_argmax_{i}"])
        assert abs(float(step0[0][i].double().sum()) - float(g[f"step0_rowsum_{i}"])) < 0.5


@pytest.mark.parametrize("kind", ("CaptioningLSTM", "CaptioningTransformer"))
def test_batched_beam_equals_per_image_and_oracle_properties(kind, images):
    """Philox noise is keyed by the GLOBAL image index: a batch of 4 gives the same captions as four
    single-image calls (img0 = index), and as a 2+2 split -- the property multi-GPU sharding relies on."""
    model, _, _ = build(kind)
    kw = dict(max_len=16, beam_size=5, top_k=50, temperature=1.0, seed=42)
    with torch.no_grad():
        toks, lens = model.generate_batch(images.cuda(), **kw)
        for i in range(4):
            t1, l1 = model.generate_batch(images[i:i + 1].cuda(), img0=i, **kw)
            assert t1[0].tolist() == toks[i].tolist() and int(l1[0]) == int(lens[i])
        t2, l2 = model.generate_batch(images[2:].cuda(), img0=2, **kw)
        assert t2.tolist() == toks[2:].tolist()
        # sub-batches decoded concurrently on separate HIP streams give the same captions
        for k in (2, 3):
            ts, ls = model.generate_batch(images.cuda(), streams=k, **kw)
            assert ts.tolist() == toks.tolist() and ls.tolist() == lens.tolist()
        other, _ = model.generate_batch(images.cuda(), **dict(kw, seed=43))
    assert other.tolist() != toks.tolist()
    assert int(lens.max()) <= 16 and int(toks.max()) < 1000 and not bool((toks == 1).any())     # <unk> never sampled


def test_cpu_tensors_fail_loudly():
    model, _, _ = build("CaptioningLSTM")
    with pytest.raises(RuntimeError, match="no CPU"):
        model.cpu()(synth_images(1), torch.zeros(1, 4, dtype=torch.long))


@pytest.mark.parametrize("kind", ("CaptioningLSTM", "CaptioningTransformer"))
def test_hipgraph_replay_equals_eager(kind, images):
    """The captured graph, replayed with other images and other seeds, gives exactly the eager captions."""
    model, _, _ = build(kind)
    kw = dict(max_len=10, beam_size=3, top_k=20, temperature=1.2)
    with torch.no_grad():
        for seed, imgs in ((5, images), (9, images.flip(0)), (5, images)):
            want_t, want_l = model.generate_batch(imgs.cuda(), seed=seed, **kw)
            got_t, got_l = model.generate_batch_graphed(imgs.cuda(), seed=seed, **kw)
            assert got_t.tolist() == want_t.tolist() and got_l.tolist() == want_l.tolist()
    assert len(model._graphs) == 1


@pytest.mark.parametrize("kind,hp", [
    ("CaptioningLSTM", dict(num_tokens=1000, emb_dim=512, hidden_size=512, num_layers=3)),
    ("CaptioningTransformer", dict(num_tokens=1000, hid_dim=512, n_layers=3, n_heads=8, pf_dim=2048, max_len=128)),
])
def test_released_checkpoint_hyperparameters(kind, hp, images, tmp_path):
    """the released checkpoints' shapes (LSTM 512/512/3, 3-layer Transformer) through
    ``save`` / ``from_pretrained`` ({'model', 'hp'} files): greedy ids of the fp32 HIP path == the CPU oracle's,
    and the bf16 path (3-layer fused LSTM step with E = 512) agrees on the first token."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import deephumor_amd.models as M
    from deephumor_amd.synth import load_synthetic
    from oracle import ref_path as R
    model = load_synthetic(getattr(M, kind)(**hp).eval(), seed=77)
    path = str(tmp_path / "released.pth")
    model.save(path)
    again = getattr(M, kind).from_pretrained(path).eval()
    assert again._hp == model._hp and set(again.state_dict()) == set(model.state_dict())
    sd = {k: v.clone() for k, v in again.state_dict().items()}
    with torch.no_grad():
        toks, lens = again.cuda().generate_batch(images[:2].cuda(), max_len=20, beam_size=1, top_k=1)
    for i in range(2):
        want = R.model_generate(kind, sd, again._hp, images[i:i + 1], max_len=20, beam_size=1, top_k=1).reshape(-1).tolist()
        assert toks[i, :int(lens[i])].cpu().tolist() == want, (kind, i)
    with torch.no_grad():
        tb, _ = again.bfloat16().generate_batch(images[:2].cuda(), max_len=20, beam_size=1, top_k=1)
    assert tb[:, 0].cpu().tolist() == toks[:, 0].cpu().tolist()


@pytest.mark.parametrize("kind,bf16", [("CaptioningTransformerBase", False), ("CaptioningTransformerWithLabels", False),
                                       ("CaptioningTransformer", True), ("CaptioningTransformerBase", True)])
def test_forward_prefill_equals_position_by_position(kind, bf16, images):
    """forward() in prefill form (batched GEMMs, one causal-attention launch per layer, cross-attention in position
    chunks) against the same forward run position by position on the decode engine (KV cache): same logits."""
    model, _, _ = build(kind)
    if bf16:
        model = model.bfloat16()
    cap, lengths, labels = captions_and_lengths()
    args = (images.cuda(), cap.cuda(), lengths) + ((labels.cuda(),) if "WithLabels" in kind else ())
    dec = model.decoder
    with torch.no_grad():
        assert dec._prefill_ok(dec._get_plan(), cap.shape[1] + 1 if kind != "CaptioningTransformer" else 49)
        fast = model(*args)
        dec._prefill_ok = lambda plan, seq: False
        slow = model(*args)
    assert fast.shape == slow.shape
    tol = 2e-2 if bf16 else 2e-5
    np.testing.assert_allclose(fast.cpu().numpy(), slow.cpu().numpy(), atol=tol, rtol=0)


@pytest.mark.parametrize("kind", ["CaptioningTransformerBase", "CaptioningTransformer"])
def test_long_caption_uses_the_long_history_kernels(kind, images):
    """max_len = 64 (> the 40 / 56 keys the register-resident attention kernels hold): greedy ids of the fp32 path
    still equal the CPU oracle's, and the bf16 path runs a longer caption still (100).  (The oracle re-runs the whole sequence per
    token: 64 positions cost a third of what 100 did.)"""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import ref_path as R
    model, sd, hp = build(kind)
    with torch.no_grad():
        toks, lens = model.generate_batch(images[:2].cuda(), max_len=64, beam_size=1, top_k=1)
    want = R.model_generate(kind, sd, hp, images[:1], max_len=64, beam_size=1, top_k=1).reshape(-1).tolist()
    assert toks[0, :int(lens[0])].cpu().tolist() == want
    with torch.no_grad():
        tb, lb = model.bfloat16().generate_batch(images[:2].cuda(), max_len=100, beam_size=3, top_k=10, seed=1)
    assert tuple(tb.shape) == (2, 100) and int(lb.max()) <= 100


@pytest.mark.parametrize("kind", ["CaptioningLSTM", "CaptioningTransformer"])
def test_early_stop_gives_the_same_captions(kind, images):
    """``early_stop_every``: decoding stops once every image has finished (the reference's all_ended() break); the
    captions equal those of the full-length loop.  <eos> is made likely so that all images finish early."""
    model, _, _ = build(kind)
    kw = dict(max_len=32, beam_size=3, top_k=10, seed=5)
    with torch.no_grad():
        model.decoder.classifier.bias[3] += 9.0
        full = model.generate_batch(images.cuda(), **kw)
        early = model.generate_batch(images.cuda(), early_stop_every=2, **kw)
    assert torch.equal(full[0], early[0]) and torch.equal(full[1], early[1])
    assert int(full[1].min()) < 32          # <eos> was sampled: images do finish before max_len


def _replay_generate(model, image, seed, **kw):
    """model.generate under torch.manual_seed(seed) with the kernels fed the CPU generator's Exp(1) noise in the reference's
    draw order (the _Replay schedule above)."""
    import deephumor_amd.models.beam as beam_mod
    made = []
    orig = beam_mod.BeamSearchHelper.__init__

    def spy(self, *a, **k):
        orig(self, *a, **k)
        made.append(self)

    beam_mod.BeamSearchHelper.__init__ = spy
    try:
        torch.manual_seed(seed)
        with torch.no_grad():
            return model.generate(image, noise_source=_Replay(lambda: made[-1]), **kw).reshape(-1).cpu().tolist()
    finally:
        beam_mod.BeamSearchHelper.__init__ = orig


@pytest.mark.parametrize("kind", ("CaptioningLSTM", "CaptioningTransformer"))
def test_char_level_configuration(kind):
    """Char-level models (V = 71 < one 128-column GEMM tile, top_k 50 > the 2 column groups, 127-position histories;
    deephumor_demo.ipynb:1307-1309, 1393-1395), goldens recorded from the reference: fp32 greedy ids bit-exact, stochastic beam
    7 / top_k 50 / T 1.1 token for token under RNG replay (incl. beams ending at different steps); both 16-bit paths run the
    same shapes, repeatably, and agree with fp32 on the first greedy token."""
    g = golden(f"g10_char_{kind}.npz")
    model, sd, _ = build(kind, v=71)
    images = synth_images(2, seed=0)
    with torch.no_grad():
        toks, lens = model.generate_batch(images.cuda(), max_len=127, beam_size=1, top_k=1)
    for i in range(2):
        assert toks[i, :int(lens[i])].cpu().tolist() == g[f"greedy_{i}"].tolist(), (kind, i)
    kw = dict(max_len=127, beam_size=7, top_k=50, temperature=1.1)
    for i in range(2):
        assert _replay_generate(model, images[i:i + 1].cuda(), 200 + i, **kw) == g[f"beam_{i}"].tolist(), (kind, i)
    with torch.no_grad():
        model.decoder.classifier.bias[3] += 2.5
    assert _replay_generate(model, images[:1].cuda(), 300, **kw) == g["beam_eos_0"].tolist()
    with torch.no_grad():
        model.decoder.classifier.bias[3] -= 2.5
    first = toks[:, 0].cpu().tolist()
    for dt in (torch.bfloat16, torch.float16):
        m16 = build(kind, v=71)[0].to(dt)
        with torch.no_grad():
            tg, lg = m16.generate_batch(images.cuda(), max_len=127, beam_size=1, top_k=1)
            t1, l1 = m16.generate_batch(images.cuda(), seed=3, **kw)
            t2, l2 = m16.generate_batch(images.cuda(), seed=3, **kw)
        assert tg[:, 0].cpu().tolist() == first
        assert torch.equal(t1, t2) and torch.equal(l1, l2) and tuple(t1.shape) == (2, 127)
        assert int(t1.min()) >= 0 and int(t1.max()) < 71 and not bool((t1 == 1).any())


@pytest.mark.parametrize("kind", ("CaptioningLSTM", "CaptioningTransformer"))
def test_demo_decode_settings_word_vocab(kind):
    """The notebook's word-level settings at V = 36,541 (LSTM beam 10 / top_k 100 / T 1.3, Transformer beam 10 / top_k 70 /
    T 1.0: deephumor_demo.ipynb:1264-1266, 1350-1352): fp32 token for token under RNG replay; the 16-bit paths run them
    (top_k above the 56 the group-guided sampler was tested with) repeatably, never sampling <unk>."""
    g = golden(f"g11_demo_{kind}.npz")
    model, _, _ = build(kind, v=36541)
    images = synth_images(2, seed=0)
    kw = dict(max_len=12, beam_size=int(g["beam_size"]), top_k=int(g["top_k"]), temperature=float(g["temperature"]))
    for i in range(2):
        assert _replay_generate(model, images[i:i + 1].cuda(), 400 + i, **kw) == g[f"beam_{i}"].tolist(), (kind, i)
    for dt in (torch.bfloat16, torch.float16):
        m16 = build(kind, v=36541)[0].to(dt)
        with torch.no_grad():
            t1, l1 = m16.generate_batch(images.cuda(), seed=9, **kw)
            t2, l2 = m16.generate_batch(images.cuda(), seed=9, **kw)
        assert torch.equal(t1, t2) and torch.equal(l1, l2) and not bool((t1 == 1).any()) and int(t1.max()) < 36541


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_u8_pipeline_equals_fp32_image_path(dtype):
    """Host-inclusive path for DECODED images (uint8 HWC in pinned memory -> CaptionPipeline with u8_preprocess -> ids in pinned
    memory): the captions equal generate_batch on the fp32 NCHW batch the notebook's ToTensor + Normalize produces from the same
    pixels -- on the fp32 model and on the bf16 model (whose stem reads the packed 16-bit layout instead of an fp32 tensor);
    results of consecutive batches live in different pinned buffers (no aliasing across one iteration)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from deephumor_amd.experiments.inference import images_to_tensor
    from deephumor_amd.pipeline import CaptionPipeline, u8_preprocess
    model, _, _ = build("CaptioningLSTM")
    model = model.to(dtype)
    u8 = bench.synth_images_u8(6, seed=0)
    batches = [(u8[:3].pin_memory(),), (u8[3:].pin_memory(),)]
    kw = dict(max_len=12, beam_size=3, top_k=20, temperature=1.0)
    with torch.no_grad():
        want = [model.generate_batch(images_to_tensor(b[0].cuda()), seed=7 + i, **kw) for i, b in enumerate(batches)]
    for overlap in (False, True):
        pipe = CaptionPipeline(model, overlap=overlap, preprocess=u8_preprocess(model), **kw)
        got = list(pipe.run(batches, seeds=[7, 8]))
        assert got[0][0].data_ptr() != got[1][0].data_ptr() and not got[0][0].is_cuda
        for (wt, wl), (gt, gl) in zip(want, got):
            assert torch.equal(wt.cpu(), gt) and torch.equal(wl.cpu(), gl)


@pytest.mark.parametrize("kind", ("CaptioningLSTM", "CaptioningTransformer"))
def test_pipeline_schedules_and_deferred_error_word(kind):
    """CaptionPipeline queues decodes asynchronously (the beam engine's error word is read at hand-over): four batches
    give ``generate_batch``'s captions in both schedules; in the default schedule result i is handed over once batch i+2 has been
    fetched, with ``low_latency=True`` before anything beyond batch i+1 is fetched; flat logits (BeamOverflow, detected at
    hand-over) still give the exact sampler's captions."""
    from deephumor_amd.pipeline import CaptionPipeline
    model, _, _ = build(kind)
    model = model.bfloat16()
    imgs = synth_images(8, seed=3)
    batches = [(imgs[2 * i:2 * i + 2].pin_memory(),) for i in range(4)]
    kw = dict(max_len=10, beam_size=3, top_k=20, temperature=1.0)
    with torch.no_grad():
        want = [model.generate_batch(b[0].cuda(), seed=20 + i, **kw) for i, b in enumerate(batches)]
    for low_latency, ahead in ((False, 2), (True, 1)):
        fetched = []

        def feed():
            for i, b in enumerate(batches):
                fetched.append(i)
                yield b
        pipe = CaptionPipeline(model, **kw)
        for i, (gt, gl) in enumerate(pipe.run(feed(), seeds=[20, 21, 22, 23], low_latency=low_latency)):
            assert max(fetched) == min(i + ahead, 3), (low_latency, i, fetched)
            assert torch.equal(want[i][0].cpu(), gt) and torch.equal(want[i][1].cpu(), gl)
    # flat logits: every row ties at its top-k threshold -> the pre-filtered samplers overflow; the pipeline repeats the batch exact
    with torch.no_grad():
        model.decoder.classifier.weight.zero_()
        model.decoder.classifier.bias.zero_()
        flat = model.generate_batch(batches[0][0].cuda(), seed=5, max_len=6, beam_size=3, top_k=20)
        pipe = CaptionPipeline(model, max_len=6, beam_size=3, top_k=20)
        got = [(t.clone(), l.clone()) for t, l in pipe.run(batches[:2], seeds=[5, 6])]
        flat1 = model.generate_batch(batches[1][0].cuda(), seed=6, max_len=6, beam_size=3, top_k=20)
        #  the repeated batch lands in the pinned pair it already owns: the PREVIOUS result, which the consumer may hold
        # across one further iteration, is not overwritten by it
        pipe = CaptionPipeline(model, max_len=6, beam_size=3, top_k=20)
        held = None
        for i, (t, l) in enumerate(pipe.run(batches[:3], seeds=[5, 6, 5])):
            if held is not None:
                assert torch.equal(held[0], held[2]) and held[0].data_ptr() != t.data_ptr(), i
            held = (t, l, t.clone())
    assert torch.equal(flat[0].cpu(), got[0][0]) and torch.equal(flat[1].cpu(), got[0][1])
    assert torch.equal(flat1[0].cpu(), got[1][0]) and torch.equal(flat1[1].cpu(), got[1][1])


@pytest.mark.parametrize("kind", ("CaptioningLSTM", "CaptioningTransformer"))
def test_pipeline_with_decode_streams(kind):
    """ ``CaptionPipeline(model, streams=2)``: the decode sessions run as two interleaved sub-batches AND defer their error
    words -- same captions as ``generate_batch``; flat logits are still caught (the sub-batches' error words are OR-ed)."""
    from deephumor_amd.pipeline import CaptionPipeline
    model, _, _ = build(kind)
    model = model.bfloat16()
    imgs = synth_images(9, seed=4)
    batches = [(imgs[3 * i:3 * i + 3].pin_memory(),) for i in range(3)]
    kw = dict(max_len=8, beam_size=3, top_k=20, temperature=1.0)
    with torch.no_grad():
        want = [model.generate_batch(b[0].cuda(), seed=30 + i, **kw) for i, b in enumerate(batches)]
        pipe = CaptionPipeline(model, streams=2, **kw)
        for i, (gt, gl) in enumerate(pipe.run(batches, seeds=[30, 31, 32])):
            assert torch.equal(want[i][0].cpu(), gt) and torch.equal(want[i][1].cpu(), gl), i
        model.decoder.classifier.weight.zero_()
        model.decoder.classifier.bias.zero_()
        flat = model.generate_batch(batches[0][0].cuda(), seed=5, **kw)
        got = [(t.clone(), l.clone()) for t, l in CaptionPipeline(model, streams=2, **kw).run(batches[:1], seeds=[5])]
    assert torch.equal(flat[0].cpu(), got[0][0]) and torch.equal(flat[1].cpu(), got[0][1])


@pytest.mark.parametrize("kind", ("CaptioningTransformer", "CaptioningTransformerBase"))
def test_pad_index_other_than_zero(kind):
    """``pad_index = 7`` (the reference's constructor accepts any value, transformers.py:393-394), goldens recorded from the reference:
    fp32 greedy ids at max_len 32 and 60 (past 49 positions the reference's zero-padded encoder rows become real cross-attention keys),
    RNG-replay beam 3, teacher-forced logits within 1e-3; the 16-bit path runs the same shapes."""
    import deephumor_amd.models as M
    g = golden("g12_pad_index.npz")
    sd, hp = synthetic_sd(kind)
    hp = dict(hp, pad_index=7)
    model = getattr(M, kind)(**hp).eval()
    model.load_state_dict(sd)
    model = model.cuda()
    images = synth_images(2, seed=0)
    for ml in (32, 60):
        with torch.no_grad():
            toks, lens = model.generate_batch(images.cuda(), max_len=ml, beam_size=1, top_k=1)
        for i in range(2):
            assert toks[i, :int(lens[i])].cpu().tolist() == g[f"{kind}_greedy{ml}_{i}"].tolist(), (kind, ml, i)
            assert bool((toks[i, int(lens[i]):] == 7).all())                  # padded with pad_index
    assert _replay_generate(model, images[:1].cuda(), 500, max_len=60, beam_size=3, top_k=20, temperature=1.3) == g[f"{kind}_beam_0"].tolist()
    cap, lengths, _ = captions_and_lengths()
    cap = cap.clone()
    cap[cap == 0] = 7
    with torch.no_grad():
        out = model(images.cuda(), cap[:2].cuda(), lengths[:2])
    np.testing.assert_allclose(out.cpu().numpy(), g[f"{kind}_forward_logits"], atol=LOGIT_TOL, rtol=0)
    with torch.no_grad():
        t16, l16 = model.bfloat16().generate_batch(images.cuda(), max_len=60, beam_size=3, top_k=20, seed=1)
    assert tuple(t16.shape) == (2, 60) and int(t16.max()) < 1000


@pytest.mark.parametrize("kind", ("CaptioningTransformer", "CaptioningTransformerBase"))
def test_pad_index_one(kind):
    """``pad_index = 1`` (golden G17 recorded from the reference): the image slot's stand-in id 1 (transformers.py:474) is itself
    padding, so position 0 attends uniformly to EVERY position and the real patch rows are the masked encoder keys -- decoded by
    full re-forward on the module-API kernels (no KV cache applies).  fp32 greedy ids at max_len 32 / 60, the stochastic beam
    through ``rng="torch"``, teacher-forced logits within 1e-3; the bf16 path runs the same shapes."""
    import deephumor_amd.models as M
    g = golden("g17_pad_index_1.npz")
    sd, hp = synthetic_sd(kind)
    hp = dict(hp, pad_index=1)
    model = getattr(M, kind)(**hp).eval()
    model.load_state_dict(sd)
    model = model.cuda()
    images = synth_images(2, seed=0)
    for ml in (32, 60):
        with torch.no_grad():
            toks, lens = model.generate_batch(images.cuda(), max_len=ml, beam_size=1, top_k=1)
        for i in range(2):
            assert toks[i, :int(lens[i])].cpu().tolist() == g[f"{kind}_greedy{ml}_{i}"].tolist(), (kind, ml, i)
    torch.manual_seed(500)
    with torch.no_grad():
        ids = model.generate(images[:1].cuda(), max_len=60, beam_size=3, top_k=20, temperature=1.3, rng="torch")
    assert ids.reshape(-1).cpu().tolist() == g[f"{kind}_beam_0"].tolist()
    cap, lengths, _ = captions_and_lengths()
    cap = cap.clone()
    cap[cap == 0] = 1
    with torch.no_grad():
        out = model(images.cuda(), cap[:2].cuda(), lengths[:2])
    np.testing.assert_allclose(out.cpu().numpy(), g[f"{kind}_forward_logits"], atol=LOGIT_TOL, rtol=0)
    with torch.no_grad():
        t16, l16 = model.bfloat16().generate_batch(images.cuda(), max_len=20, beam_size=3, top_k=20, seed=1)
    assert tuple(t16.shape) == (2, 20) and int(t16.max()) < 1000


@pytest.mark.parametrize("kind", KINDS)
def test_rng_torch_returns_the_reference_sampled_caption(kind, images):
    """The product option ``rng="torch"``: ``torch.manual_seed(s); model.generate(image, rng="torch")`` is the reference's own
    call sequence and returns its sampled caption (golden G5: beam 3, top_k 20, T 1.3); in a batch, image i replays
    ``torch.manual_seed(seed + i)``."""
    g = golden(f"g2g3_{kind}.npz")
    model, _, _ = build(kind)
    _, _, labels = captions_and_lengths()
    kw = dict(max_len=12, beam_size=3, top_k=20, temperature=1.3)
    wl = "WithLabels" in kind
    for i in range(2):
        torch.manual_seed(100 + i)
        args = (images[i:i + 1].cuda(), labels[i:i + 1].cuda()) if wl else (images[i:i + 1].cuda(),)
        with torch.no_grad():
            ids = model.generate(*args, rng="torch", **kw)
        assert ids.reshape(-1).cpu().tolist() == g[f"beam_{i}"].tolist(), (kind, i)
    args = (images[:2].cuda(), labels[:2].cuda()) if wl else (images[:2].cuda(),)
    with torch.no_grad():
        toks, lens = model.generate_batch(*args, seed=100, rng="torch", **kw)
        t2, l2 = model.generate_batch(*args, seed=100, rng="torch", streams=2, **kw)
    for i in range(2):
        assert toks[i, :int(lens[i])].cpu().tolist() == g[f"beam_{i}"].tolist(), (kind, i)
    assert torch.equal(toks, t2) and torch.equal(lens, l2)
    with pytest.raises(ValueError):
        model.generate_batch(*args, rng="torch", **kw)          # the default generator cannot serve two images in the reference's order


@pytest.mark.parametrize("kind", ("CaptioningLSTM", "CaptioningTransformer"))
def test_beam_size_above_16(kind, images):
    """beam_size 24 (the reference takes any beam_size <= top_k, beam.py:7-9; golden G16 recorded from it): fp32 token for token
    through ``rng="torch"``; batch of 2 == singles; both 16-bit paths run beam 24 and beam 40 repeatably."""
    g = golden(f"g16_beam24_{kind}.npz")
    model, _, _ = build(kind)
    kw = dict(max_len=12, beam_size=24, top_k=50, temperature=1.0)
    with torch.no_grad():
        toks, lens = model.generate_batch(images[:2].cuda(), seed=800, rng="torch", **kw)
    for i in range(2):
        assert toks[i, :int(lens[i])].cpu().tolist() == g[f"beam_{i}"].tolist(), (kind, i)
    for dt in (torch.bfloat16, torch.float16):
        m16 = build(kind)[0].to(dt)
        for beam in (24, 40):
            with torch.no_grad():
                t1, l1 = m16.generate_batch(images.cuda(), max_len=12, beam_size=beam, top_k=50, seed=3)
                t2, l2 = m16.generate_batch(images.cuda(), max_len=12, beam_size=beam, top_k=50, seed=3)
                ta, la = m16.generate_batch(images[:2].cuda(), max_len=12, beam_size=beam, top_k=50, seed=3)
            assert torch.equal(t1, t2) and torch.equal(l1, l2) and torch.equal(t1[:2], ta) and torch.equal(l1[:2], la)
            assert int(t1.min()) >= 0 and int(t1.max()) < 1000 and not bool((t1 == 1).any())
    with pytest.raises(ValueError):
        model.generate_batch(images[:1].cuda(), max_len=4, beam_size=65, top_k=70)


@pytest.mark.parametrize("kind", ("CaptioningLSTM", "CaptioningTransformer"))
def test_no_decode_step_edge(kind):
    """Prefix of max_len - 1 tokens (found by tools/fuzz_generate.py): the reference's LSTM ``generate`` returns beam_size copies of
    beam 0's row, 2-D (rnn_models.py:103, 140-141); the Transformer its usual 1-D caption.  Shapes and ids vs the reference-recorded
    G13, fp32 under RNG replay; ``generate_batch`` gives that row; the 16-bit path returns the same shapes."""
    from test_oracle_golden import G13_CASES
    import deephumor_amd.models.beam as beam_mod
    g = golden("g13_no_decode_step.npz")
    model, _, _ = build(kind)
    images = synth_images(1, seed=0).cuda()
    cap, _, _ = captions_and_lengths()
    for name, kw in G13_CASES:
        kw = dict(kw)
        p = kw.pop("prefix")
        kw["caption"] = cap[:1, :p].cuda() if p else None
        made = []
        orig = beam_mod.BeamSearchHelper.__init__

        def spy(self, *a, **k):
            orig(self, *a, **k)
            made.append(self)

        beam_mod.BeamSearchHelper.__init__ = spy
        try:
            torch.manual_seed(600)
            with torch.no_grad():
                ids = model.generate(images, noise_source=_Replay(lambda: made[-1]), **kw)
        finally:
            beam_mod.BeamSearchHelper.__init__ = orig
        want = g[f"{kind}_{name}"]
        assert tuple(ids.shape) == want.shape and ids.reshape(-1).cpu().tolist() == want.reshape(-1).tolist(), (kind, name)
        with torch.no_grad():
            toks, lens = model.generate_batch(images, seed=3, **kw)
            half = build(kind)[0].to(torch.bfloat16).generate(images, seed=3, **kw)
        assert int(lens[0]) == kw["max_len"] and tuple(toks.shape) == (1, kw["max_len"])
        assert tuple(half.shape) == want.shape


@pytest.mark.parametrize("dec_kind", ("lstm", "tfm"))
def test_beam_size_equal_to_vocabulary_matches_the_reference(dec_kind):
    """beam_size == top_k == num_tokens: after <unk> is dropped fewer positive-probability tokens than beams remain.  Current
    torch.multinomial (hence the reference) does not raise -- the last pick is a zero-probability token, a dead beam -- and neither does
    the engine (dead beam = <pad> at -inf): the captions equal the oracle's token for token under RNG replay (twelve random cases
    of tools/fuzz_generate.py; two of them pinned here)."""
    from deephumor_amd.models import LSTMDecoder, SelfAttentionTransformerDecoder
    from helpers import synth_state_dict
    from oracle import ref_path as R
    v, max_len, temp = 7, 9, 0.7
    g = torch.Generator().manual_seed(29)
    if dec_kind == "lstm":
        dec = LSTMDecoder(v, emb_dim=112, hidden_size=408, num_layers=1, dropout=0.0)
        first = torch.randn(1, 1, 112, generator=g)
    else:
        dec = SelfAttentionTransformerDecoder(v, hid_dim=16, n_layers=3, n_heads=1, pf_dim=32, dropout=0.0, pad_index=0, max_len=64)
        first = torch.randn(1, 16, generator=g)
    sd = synth_state_dict(dec.state_dict(), seed=106, logit_std=4.0)
    dec.load_state_dict(sd)
    osd = {"decoder." + k: t.clone() for k, t in sd.items()}
    kw = dict(max_len=max_len, temperature=temp, beam_size=v, top_k=v)
    torch.manual_seed(77)
    with torch.no_grad():
        want = (R.lstm_decoder_generate(osd, "decoder", first, **kw) if dec_kind == "lstm"
                else R.transformer_generate(osd, "decoder", first, None, 0, 1, **kw)).reshape(-1).tolist()
    got = _replay_generate(dec.cuda().eval(), first.cuda(), 77, **kw)
    assert got == want



@pytest.mark.parametrize("dec_kind", ("lstm", "tfm"))
def test_constant_logits_decode_like_the_reference(dec_kind):
    """A classifier that outputs the same logit for every token (zero weights): all 3,000 tokens tie at every top-k threshold, which the
    pre-filtered samplers cannot hold -- the decoders catch the overflow and repeat the batch through the general sampler.  Token for
    token the oracle's caption under RNG replay (fp32); the 16-bit path runs, repeatably."""
    from deephumor_amd.models import LSTMDecoder, SelfAttentionTransformerDecoder
    from helpers import synth_state_dict
    from oracle import ref_path as R
    v = 3000
    g = torch.Generator().manual_seed(3)
    if dec_kind == "lstm":
        dec = LSTMDecoder(v, emb_dim=32, hidden_size=64, num_layers=1, dropout=0.0)
        first = torch.randn(1, 1, 32, generator=g)
    else:
        dec = SelfAttentionTransformerDecoder(v, hid_dim=64, n_layers=1, n_heads=1, pf_dim=64, dropout=0.0, pad_index=0, max_len=64)
        first = torch.randn(1, 64, generator=g)
    sd = synth_state_dict(dec.state_dict(), seed=9)
    sd["classifier.weight"] = torch.zeros_like(sd["classifier.weight"])
    sd["classifier.bias"] = torch.full_like(sd["classifier.bias"], 0.5)
    dec.load_state_dict(sd)
    osd = {"decoder." + k: t.clone() for k, t in sd.items()}
    kw = dict(max_len=7, temperature=1.1, beam_size=3, top_k=20)
    torch.manual_seed(5)
    with torch.no_grad():
        want = (R.lstm_decoder_generate(osd, "decoder", first, **kw) if dec_kind == "lstm"
                else R.transformer_generate(osd, "decoder", first, None, 0, 1, **kw)).reshape(-1).tolist()
    # (RNG replay draws from the CPU generator as it goes, so it cannot be repeated: the general sampler from the start)
    assert _replay_generate(dec.cuda().eval(), first.cuda(), 5, exact=True, **kw) == want
    with torch.no_grad():                                           # Philox noise: the automatic repeat == exact from the start
        auto = dec.generate_batch(first.cuda().reshape(1, -1), seed=4, **kw)
        forced = dec.generate_batch(first.cuda().reshape(1, -1), seed=4, exact=True, **kw)
    assert torch.equal(auto[0], forced[0]) and torch.equal(auto[1], forced[1])
    half = dec.to(torch.bfloat16)
    with torch.no_grad():
        a = half.generate_batch(first.cuda().to(torch.bfloat16).reshape(1, -1), seed=4, **kw)
        b = half.generate_batch(first.cuda().to(torch.bfloat16).reshape(1, -1), seed=4, **kw)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and int(a[0].max()) < v and not bool((a[0] == 1).any())


@pytest.mark.parametrize("what", ["nan_bias", "inf_bias", "nan_weight", "nan_hidden"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("kind", ["CaptioningLSTM", "CaptioningTransformer"])
def test_nonfinite_logits_raise_like_the_reference(kind, dtype, what, images):
    """A NaN or +inf among a row's logits makes the reference's ``torch.multinomial`` raise (beam.py:46: softmax of such a row is NaN).
    Round 5 found the engine returning float bit patterns as token ids instead (NaN scores -> no lane equals the maximum -> row -1 of the
    token buffer; uninitialised survivor slots in the candidate draw): the samplers flag the row (DH_BEAM_ERR_NONFINITE) and hand on a finite
    dummy pick, the 16-bit paths check the classifier operands once per plan -- RuntimeError with torch's message, for greedy and beam
    search, whether the NaN comes from the classifier, from +inf, or from the decoder's activations (every logit NaN)."""
    model, sd, hp = build(kind)
    with torch.no_grad():
        if what == "nan_bias":
            model.decoder.classifier.bias[17] = float("nan")
        elif what == "inf_bias":
            model.decoder.classifier.bias[17] = float("inf")
        elif what == "nan_weight":
            model.decoder.classifier.weight[23, 5] = float("nan")
    if dtype != torch.float32:
        model = model.to(dtype)
    imgs = images.clone().cuda()
    if what == "nan_hidden":                              # every logit of every row NaN (a NaN pixel would not do: INTEGRATION.md, the
        with torch.no_grad():                            # encoder's ReLU is a hardware maximum, which drops NaN)
            p0 = next(p for n, p in model.decoder.named_parameters() if n.endswith("weight_hh_l0") or n.endswith("fc_o.weight"))
            p0.view(-1)[3] = float("nan")                 # (no ReLU between these and the classifier: the NaN reaches every logit)
    for beam, top_k in ((1, 1), (3, 20)):
        with torch.no_grad(), pytest.raises(RuntimeError, match="probability tensor contains"):
            model.generate_batch(imgs, max_len=6, beam_size=beam, top_k=top_k, seed=1)
    # ... and the engine is usable afterwards (the error word is per call; nothing indexed out of bounds)
    clean, _, _ = build(kind)
    if dtype != torch.float32:
        clean = clean.to(dtype)
    with torch.no_grad():
        toks, lens = clean.generate_batch(images.cuda(), max_len=6, beam_size=3, top_k=20, seed=1)
    assert int(toks.max()) < hp["num_tokens"] and int(toks.min()) >= 0


def test_reference_raises_on_nonfinite_logits(images):
    """The behaviour the test above mirrors, from the CPU oracle (the reference's own sequence of torch calls)."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import ref_path as R
    _, sd, hp = build("CaptioningLSTM")
    sd = dict(sd)
    sd["decoder.classifier.bias"] = sd["decoder.classifier.bias"].clone()
    sd["decoder.classifier.bias"][17] = float("nan")
    with pytest.raises(RuntimeError, match="probability tensor contains"):
        R.model_generate("CaptioningLSTM", sd, hp, images[:1], max_len=4, beam_size=3, top_k=20)


@pytest.mark.parametrize("kind", ["CaptioningLSTM", "CaptioningTransformer", "CaptioningTransformerWithLabels", "CaptioningLSTMWithLabels"])
def test_out_of_range_ids_raise_like_nn_embedding(kind, images):
    """Token and label ids outside their table raise ``IndexError`` as the reference's ``nn.Embedding`` lookups do, a length outside the
    padded sequence ``RuntimeError`` as ``pack_padded_sequence`` does -- BEFORE any kernel gathers with them (round 5: a label of -3 or a
    token of 2^31 + 5 was a GPU memory fault, V + 100 silent garbage).  Good inputs still run afterwards."""
    model, _, hp = build(kind)
    v = hp["num_tokens"]
    imgs = images[:3].cuda()
    g = torch.Generator().manual_seed(3)
    lab = (torch.randint(4, v, (3, 2), generator=g).cuda(),) if "WithLabels" in kind else ()
    good = torch.randint(4, v, (3, 5), generator=g).cuda()
    lengths = torch.tensor([5, 4, 2])
    bad = []
    for r, c, val in ((1, 2, v + 100), (0, 1, -1), (2, 0, 2 ** 31 + 5), (2, 4, v)):
        t = good.clone()
        t[r, c] = val
        bad.append(t)
    gen_kw = dict(max_len=6, beam_size=3, top_k=20, seed=1)
    with torch.no_grad():
        for t in bad:
            with pytest.raises(IndexError, match="index out of range"):
                model(imgs, t, lengths, *lab)
            with pytest.raises(IndexError, match="index out of range"):
                model.generate_batch(imgs, *lab, caption=t, **gen_kw)
            with pytest.raises(IndexError, match="index out of range"):
                model.generate_batch_graphed(imgs, *lab, caption=t, **gen_kw)
        if lab:
            for val in (v + 7, -3):
                bl = lab[0].clone()
                bl[1, 1] = val
                with pytest.raises(IndexError, match="index out of range"):
                    model.generate_batch(imgs, bl, **gen_kw)
                with pytest.raises(IndexError, match="index out of range"):
                    model.generate_batch_graphed(imgs, bl, **gen_kw)
                with pytest.raises(IndexError, match="index out of range"):
                    model(imgs, good, lengths, bl)
        if "LSTM" in kind:
            with pytest.raises(RuntimeError, match="sequence length"):
                model(imgs, good, torch.tensor([9, 4, 2]), *lab)
            with pytest.raises(RuntimeError, match="greater than 0"):
                model(imgs, good, torch.tensor([5, 0, 2]), *lab)
        out = model(imgs, good, lengths, *lab)
        toks, _ = model.generate_batch(imgs, *lab, caption=good[:, :2], **gen_kw)
        tg, _ = model.generate_batch_graphed(imgs, *lab, caption=good[:, :2], **gen_kw)
    assert bool(torch.isfinite(out).all()) and int(toks.max()) < v and int(toks.min()) >= 0 and torch.equal(toks, tg)


@pytest.mark.parametrize("half", [False, True], ids=["f32", "bf16"])
def test_images_torch_conv2d_would_refuse_are_refused(half, images):
    """A tensor that is not [N, 3, H, W] (or the packed layout of ``preprocess_images``) raises as ``conv2d`` does in the reference; round 5:
    the 16-bit stem read a 1-channel batch as 3 channels (foreign memory) and captioned it."""
    model, _, _ = build("CaptioningTransformer")
    if half:
        model = model.bfloat16()
    x = images.cuda()
    with torch.no_grad():
        for bad in (x[:, :1], torch.cat([x, x[:, :1]], 1)):
            with pytest.raises(RuntimeError, match="to have 3 channels"):
                model.generate_batch(bad, max_len=3, beam_size=1, top_k=1)
        with pytest.raises(RuntimeError, match="4D"):
            model.generate_batch(x[0], max_len=3, beam_size=1, top_k=1)
        for bad in (x.double(), (x * 40 + 128).clamp(0, 255).to(torch.uint8)):
            with pytest.raises(TypeError, match="dtype"):
                model.generate_batch(bad, max_len=3, beam_size=1, top_k=1)
        toks, _ = model.generate_batch(x.permute(0, 1, 3, 2), max_len=3, beam_size=1, top_k=1)      # a non-contiguous view is fine
    assert tuple(toks.shape) == (4, 3)


def test_lstm_prefix_as_long_as_max_len_is_not_truncated(images):
    """The reference's LSTM ``generate`` makes its first draw whatever the prefix length and never truncates to ``max_len``
    (rnn_models.py:82-101): a prefix of L >= max_len tokens comes back as L + 1 tokens.  (Round 5 edge-shape sweep: the engine returned
    max_len tokens for L == max_len and raised for L > max_len.)  Greedy ids against the CPU oracle."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import ref_path as R
    model, sd, hp = build("CaptioningLSTM")
    g = torch.Generator().manual_seed(1)
    cap = torch.randint(4, hp["num_tokens"], (1, 6), generator=g)
    for L, max_len in ((4, 4), (5, 3), (3, 4), (6, 1)):
        kw = dict(caption=cap[:, :L], max_len=max_len, beam_size=1, top_k=1)
        want = R.model_generate("CaptioningLSTM", sd, hp, images[:1], **kw).reshape(-1).tolist()
        with torch.no_grad():
            toks, lens = model.generate_batch(images[:1].cuda(), caption=cap[:, :L].cuda(), max_len=max_len, beam_size=1, top_k=1)
            one = model.generate(images[:1].cuda(), caption=cap[:, :L].cuda(), max_len=max_len, beam_size=1, top_k=1)
        assert toks[0, :int(lens[0])].cpu().tolist() == want and len(want) == max(L + 1, min(max_len, L + 1)), (L, max_len, want)
        assert one.reshape(-1).cpu().tolist() == want


@pytest.mark.parametrize("kind,half", [("CaptioningLSTM", True), ("CaptioningTransformer", True), ("CaptioningTransformer", False)])
def test_concurrent_generate_calls_from_threads(kind, half, images):
    """One model, several host threads decoding different batches at once (each on its own HIP stream, or all on the default one):
    every call returns what it returns alone -- plans are shared read-only, decode state, scratch and beam buffers are per call."""
    import threading
    from deephumor_amd.synth import synth_images as si
    model, _, _ = build(kind)
    if half:
        model = model.bfloat16()
    jobs = [(si(n, seed=j).cuda(), dict(max_len=10, beam_size=b, top_k=20, seed=100 + j)) for j, (n, b) in enumerate(((3, 3), (17, 5), (40, 1), (8, 10), (5, 3)))]
    with torch.no_grad():
        want = [model.generate_batch(i, **kw) for i, kw in jobs]
    torch.cuda.synchronize()
    for own_stream in (False, True):
        got, errs = [None] * len(jobs), []

        def work(k):
            try:
                s = torch.cuda.Stream() if own_stream else torch.cuda.current_stream()
                with torch.no_grad(), torch.cuda.stream(s):
                    got[k] = model.generate_batch(jobs[k][0], **jobs[k][1])
                s.synchronize()
            except Exception as e:                        # noqa: BLE001
                errs.append(repr(e))
        ths = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
        [t.start() for t in ths]
        [t.join() for t in ths]
        torch.cuda.synchronize()
        assert not errs, errs[:2]
        for k in range(len(jobs)):
            assert torch.equal(got[k][0], want[k][0]) and torch.equal(got[k][1], want[k][1]), (own_stream, k)
