"""Module-level API of ``deephumor.models.transformers`` that the captioning models themselves never call but
users of the reference can: ``get_pad_mask`` / ``get_autoregressive_mask`` (transformers.py:12-40),
``MultiHeadAttentionLayer.forward`` (:82-129), ``PositionwiseFeedforwardLayer.forward`` (:151-165),
``DecoderLayer.forward`` (:343-377), ``SelfAttentionDecoderLayer.forward`` (:612-636),
``TransformerDecoder.forward(x, enc_out, start_emb=None)`` (:432) -- each against the oracle restatement; plus the
RNG contract of ``generate`` (beam.py:46 draws from torch's global generator), hipGraph invalidation on weight
change, and the pre-filtered samplers' exact fallback."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from helpers import synthetic_sd, synth_images  # noqa: E402


def close(a, b, atol, rtol=1e-4):
    np.testing.assert_allclose(a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy(), atol=atol, rtol=rtol)


@pytest.fixture(scope="module")
def tr():
    import deephumor_amd.models as M
    sd, hp = synthetic_sd("CaptioningTransformer")
    model = M.CaptioningTransformer(**hp).eval()
    model.load_state_dict(sd)
    return model.cuda(), sd, hp


def test_mask_helpers_match_reference_formulas():
    from deephumor_amd.models.transformers import get_autoregressive_mask, get_pad_mask
    g = torch.Generator().manual_seed(0)
    q = torch.randint(0, 4, (3, 7), generator=g)
    k = torch.randint(0, 4, (3, 11), generator=g)
    for pad in (0, 2):
        got = get_pad_mask(q.cuda(), k.cuda(), pad_index=pad)
        want = (k == pad).unsqueeze(1).expand(3, 7, 11)                       # transformers.py:24-26
        assert got.dtype == torch.bool and torch.equal(got.cpu(), want)
    got = get_autoregressive_mask(q.cuda())
    want = torch.triu(torch.ones([3, 7, 7]), 1).bool()                        # transformers.py:39-40
    assert got.dtype == torch.bool and torch.equal(got.cpu(), want)


@pytest.mark.parametrize("with_mask", [True, False])
def test_multi_head_attention_forward(tr, with_mask):
    from oracle import ref_path as R
    model, sd, hp = tr
    layer = model.decoder.layers[1].enc_attn
    g = torch.Generator().manual_seed(3)
    bs, L, d = 3, 13, hp["hid_dim"]
    q, k, v = (torch.randn(bs, L, d, generator=g) for _ in range(3))
    mask = (torch.rand(bs, L, L, generator=g) < 0.3) if with_mask else None
    if with_mask:
        mask[1, 4, :] = True                                                  # a fully masked query row -> uniform weights
    with torch.no_grad():
        got = layer(q.cuda(), k.cuda(), v.cuda(), mask=None if mask is None else mask.cuda())
    want = R.mha(sd, "decoder.layers.1.enc_attn", q, k, v, mask, hp["n_heads"])
    assert tuple(got.shape) == (bs, L, d)
    close(got, want, atol=2e-5)
    with pytest.raises(RuntimeError):                                         # reference: k.view(bs, seq_len_q, ...) fails
        layer(q.cuda(), k[:, :5].cuda(), v[:, :5].cuda())


def test_feedforward_and_decoder_layer_forward(tr):
    from oracle import ref_path as R
    model, sd, hp = tr
    g = torch.Generator().manual_seed(5)
    bs, L, d, nh = 2, 9, hp["hid_dim"], hp["n_heads"]
    x, enc = torch.randn(bs, L, d, generator=g), torch.randn(bs, L, d, generator=g)
    lp = "decoder.layers.0"
    layer = model.decoder.layers[0]
    with torch.no_grad():
        ff = layer.pf(x.cuda())
    want_ff = F.linear(torch.relu(F.linear(x, sd[lp + ".pf.fc_1.weight"], sd[lp + ".pf.fc_1.bias"])),
                       sd[lp + ".pf.fc_2.weight"], sd[lp + ".pf.fc_2.bias"])
    close(ff, want_ff, atol=5e-5)
    causal = torch.triu(torch.ones(bs, L, L), 1).bool()
    enc_mask = torch.zeros(bs, L, L, dtype=torch.bool)
    enc_mask[:, :, 2] = True
    with torch.no_grad():
        got = layer(x.cuda(), enc.cuda(), input_mask=causal.cuda(), enc_mask=enc_mask.cuda())
    h = R._ln(sd, lp + ".self_attn_ln", x + R.mha(sd, lp + ".self_attn", x, x, x, causal, nh))
    h = R._ln(sd, lp + ".enc_attn_ln", h + R.mha(sd, lp + ".enc_attn", h, enc, enc, enc_mask, nh))
    ffw = F.linear(torch.relu(F.linear(h, sd[lp + ".pf.fc_1.weight"], sd[lp + ".pf.fc_1.bias"])),
                   sd[lp + ".pf.fc_2.weight"], sd[lp + ".pf.fc_2.bias"])
    want = R._ln(sd, lp + ".pf_ln", h + ffw)
    close(got, want, atol=1e-4)


def test_self_attention_decoder_layer_forward():
    from oracle import ref_path as R
    import deephumor_amd.models as M
    sd, hp = synthetic_sd("CaptioningTransformerBase")
    model = M.CaptioningTransformerBase(**hp).eval()
    model.load_state_dict(sd)
    model = model.cuda()
    g = torch.Generator().manual_seed(7)
    bs, L, d, nh = 2, 6, hp["hid_dim"], hp["n_heads"]
    x = torch.randn(bs, L, d, generator=g)
    causal = torch.triu(torch.ones(bs, L, L), 1).bool()
    lp = "decoder.layers.2"
    with torch.no_grad():
        got = model.decoder.layers[2](x.cuda(), input_mask=causal.cuda())
    h = R._ln(sd, lp + ".self_attn_ln", x + R.mha(sd, lp + ".self_attn", x, x, x, causal, nh))
    ffw = F.linear(torch.relu(F.linear(h, sd[lp + ".pf.fc_1.weight"], sd[lp + ".pf.fc_1.bias"])),
                   sd[lp + ".pf.fc_2.weight"], sd[lp + ".pf.fc_2.bias"])
    close(got, R._ln(sd, lp + ".pf_ln", h + ffw), atol=1e-4)


def test_transformer_decoder_forward_without_start_emb(tr):
    """transformers.py:432: ``start_emb=None`` -- no image slot, ids are not shifted, the sequence is padded to the 49
    patches; pad tokens (also at position 0) are masked as keys."""
    from oracle import ref_path as R
    model, sd, hp = tr
    g = torch.Generator().manual_seed(9)
    bs = 3
    x = torch.randint(6, hp["num_tokens"], (bs, 12), generator=g)
    x[1, 8:] = 0
    x[2, 0] = 0
    enc = torch.randn(bs, 49, hp["hid_dim"], generator=g)
    enc[0, 5, 7] = 0.0                                                        # masked patch (transformers.py:480-481)
    with torch.no_grad():
        got = model.decoder(x.cuda(), enc.cuda())
    want = R.transformer_forward(sd, "decoder", x, enc, None, hp["pad_index"], hp["n_heads"])
    assert tuple(got.shape) == tuple(want.shape) == (bs, 49, hp["num_tokens"])
    close(got, want, atol=1e-3)


@pytest.mark.parametrize("kind", ["CaptioningLSTM", "CaptioningTransformer"])
def test_generate_draws_its_seed_from_the_torch_generator(kind):
    """beam.py:46: the reference samples with torch's global generator, so ``torch.manual_seed`` reproduces a caption and
    successive calls differ ("generate another meme", deephumor_demo.ipynb:1264-1266)."""
    import deephumor_amd.models as M
    sd, hp = synthetic_sd(kind)
    model = getattr(M, kind)(**hp).eval()
    model.load_state_dict(sd)
    model = model.cuda()
    img = synth_images(1, seed=4).cuda()
    kw = dict(max_len=12, beam_size=5, top_k=50, temperature=1.3)
    with torch.no_grad():
        torch.manual_seed(123)
        a = model.generate(img, **kw)
        b = model.generate(img, **kw)
        torch.manual_seed(123)
        a2 = model.generate(img, **kw)
        b2 = model.generate(img, **kw)
        c = [model.generate(img, **kw) for _ in range(3)]
    assert torch.equal(a, a2) and torch.equal(b, b2)
    assert any(not torch.equal(a, t) for t in [b] + c)            # 4 further stochastic captions: not all identical to a
    assert a.dtype == torch.int64 and a.dim() == 1


def test_graph_replay_follows_weight_updates():
    """A captured hipGraph holds raw pointers to weights and plan tensors: after load_state_dict / .to() the cached
    graph must be re-captured, not replayed against stale memory."""
    import deephumor_amd.models as M
    sd, hp = synthetic_sd("CaptioningLSTM")
    from deephumor_amd.synth import synth_state_dict
    model = M.CaptioningLSTM(**hp).eval()
    model.load_state_dict(sd)
    model = model.cuda().bfloat16()
    imgs = synth_images(3, seed=1).cuda()
    kw = dict(max_len=8, beam_size=3, top_k=10)
    with torch.no_grad():
        t0, _ = model.generate_batch_graphed(imgs, seed=5, **kw)
        e0, _ = model.generate_batch(imgs, seed=5, **kw)
        assert torch.equal(t0, e0)
        sd2 = synth_state_dict(sd, seed=99)
        model.load_state_dict(sd2)
        e1, _ = model.generate_batch(imgs, seed=5, **kw)
        t1, _ = model.generate_batch_graphed(imgs, seed=5, **kw)
        assert torch.equal(t1, e1) and not torch.equal(e1, e0)
        model = model.float().bfloat16()                         # new parameter storage, same values
        t2, _ = model.generate_batch_graphed(imgs, seed=5, **kw)
        assert torch.equal(t2, e1)


def _expected_picks(logits, noise, beam, top_k, temp):
    from oracle.ref_path import BeamBook
    book = BeamBook(temp, beam, top_k)
    filt = book.keep_top_k(logits.clone())
    picks = torch.topk(torch.softmax(filt / temp, -1) / noise, beam, dim=-1).indices
    return picks, torch.gather(filt, 1, picks).log_softmax(-1)


def test_prefiltered_samplers_fall_back_to_the_exact_select():
    """Structured logits that let far more than DH_BEAM_MAX_SURVIVORS (1024) values through the cheap pre-filters
    (per-thread maxima / 16-bit bucket of the group maxima) without any tie at the threshold: the kernels re-derive the
    exact candidate set in place (radix select) instead of reporting an overflow."""
    from deephumor_amd import hip
    v, rows, beam, top_k, temp = 36541, 4, 5, 50, 1.0
    i = torch.arange(v)
    base = -(i % 1024).float() + 1e-4 * (i // 1024).float()      # 49 threads x 36 values pass the thread-maxima bound
    logits = torch.stack([base, base.flip(0), base * 0.5, 1.0 + 0.007 * torch.rand(v, generator=torch.Generator().manual_seed(1))])
    ld = (v + 63) // 64 * 64
    noise = torch.ones(rows, ld)
    noise[:, :v] = torch.empty(rows, v).exponential_(1, generator=torch.Generator().manual_seed(2))
    want_i, want_v = _expected_picks(logits, noise[:, :v], beam, top_k, temp)
    lg = torch.zeros(rows, ld)
    lg[:, :v] = logits
    lg = lg.cuda()[:, :v]
    pi = torch.empty(rows, beam, dtype=torch.int32, device="cuda")
    pv = torch.empty(rows, beam, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    hip.beam_row_sample(lg, v, rows, 1, beam, top_k, temp, 1, noise.cuda(), 0, 0, 0, pi, pv, err)
    assert int(err.item()) == 0
    assert pi.cpu().tolist() == want_i.tolist()
    close(pv, want_v, atol=2e-6)
    # the group-guided sampler: every value of row 3 lies in ONE 16-bit key bucket -> all 36,541 pass its bound
    ng = hip.n_groups(v)
    pad = torch.full((rows, ng * 64 - v), float("-inf"))
    gmax = torch.cat([logits, pad], 1).view(rows, ng, 64).max(-1).values.cuda()
    pi.zero_()
    hip.beam_row_sample_groups(lg, v, gmax, rows, 1, beam, top_k, temp, 1, noise.cuda(), 0, 0, 0, pi, pv, err)
    assert int(err.item()) == 0
    assert pi.cpu().tolist() == want_i.tolist()
    # a genuine overflow: more than 1024 logits tie AT the threshold -> error bit, as documented
    flat = torch.zeros(1, ld).cuda()[:, :v]
    hip.beam_row_sample(flat, v, 1, 1, beam, top_k, temp, 1, noise[:1].cuda(), 0, 0, 0, pi[:1], pv[:1], err)
    assert int(err.item()) & hip.ERR_OVERFLOW
