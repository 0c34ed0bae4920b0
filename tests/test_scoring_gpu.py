"""Teacher-forced scoring: perplexity kernels vs the reference's value and the oracle,
corpus scoring with per-template encoder caching vs the oracle's per-pair forward."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import GOLDEN, synthetic_sd, synth_images  # noqa: E402


def test_perplexity_kernels():
    from deephumor_amd.experiments import perplexity
    from oracle.ref_path import perplexity as ref_pp
    gold = json.load(open(os.path.join(GOLDEN, "g8_text_and_metrics.json")))["perplexity"]
    g = torch.Generator().manual_seed(gold["seed"])
    logits = torch.randn(4, 9, 50, generator=g) * 2
    targets = torch.randint(6, 50, (4, 9), generator=g)
    lengths = torch.tensor(gold["lengths"])
    for r, n in enumerate(lengths.tolist()):
        targets[r, n:] = 0
    pp = perplexity(logits.cuda(), targets.cuda(), lengths.cuda())
    assert abs(float(pp) - gold["value"]) < 1e-4 * gold["value"]
    big = torch.randn(3, 7, 36541, generator=g) * 3
    tg = torch.randint(6, 36541, (3, 7), generator=g)
    ln = torch.tensor([7, 7, 4])
    tg[2, 4:] = 0
    want = float(ref_pp(big, tg, ln))
    assert abs(float(perplexity(big.cuda(), tg.cuda(), ln.cuda())) - want) < 1e-4 * want


@pytest.mark.parametrize("kind", ("CaptioningLSTM", "CaptioningTransformer"))
def test_score_captions_with_template_caching(kind):
    import deephumor_amd.models as M
    from deephumor_amd.experiments import score_captions
    from oracle import ref_path as R
    sd, hp = synthetic_sd(kind)
    model = getattr(M, kind)(**hp).eval()
    model.load_state_dict(sd)
    model = model.cuda()
    templates = synth_images(3, seed=0)
    g = torch.Generator().manual_seed(3)
    n, l = 7, 12
    caps = torch.randint(6, 1000, (n, l), generator=g)
    lengths = torch.tensor([12, 9, 12, 5, 7, 12, 3])
    for r, k in enumerate(lengths.tolist()):
        caps[r, k - 1] = 3                       # <eos>
        caps[r, k:] = 0
    tidx = torch.tensor([0, 1, 2, 2, 1, 0, 0])
    got = score_captions(model, templates.cuda(), tidx.cuda(), caps.cuda(), lengths.cuda(), batch_size=4).cpu()
    for i in range(n):
        logits = R.model_forward(kind, sd, hp, templates[tidx[i]:tidx[i] + 1], caps[i:i + 1, :-1])[:, :l]
        want = float(R.perplexity(logits, caps[i:i + 1], lengths[i:i + 1]))
        assert abs(float(got[i]) - want) < 2e-3 * want, (i, float(got[i]), want)


@pytest.mark.parametrize("kind", ("CaptioningLSTM", "CaptioningTransformer", "CaptioningTransformerBase"))
def test_bf16_scoring_from_hidden_states_equals_logits_path(kind):
    """bf16 ``score_captions`` (hidden states -> dh_vocab_logprob, no logits in memory) against the perplexity of the
    same bf16 model's materialised ``forward()`` logits; and close to the fp32 reference path's perplexity."""
    import deephumor_amd.models as M
    from deephumor_amd.experiments import score_captions
    from deephumor_amd.experiments.metrics import sequence_perplexity
    sd, hp = synthetic_sd(kind)
    model = getattr(M, kind)(**hp).eval()
    model.load_state_dict(sd)
    templates = synth_images(3, seed=0).cuda()
    g = torch.Generator().manual_seed(4)
    n, l = 9, 14
    caps = torch.randint(6, 1000, (n, l), generator=g)
    lengths = torch.tensor([14, 9, 14, 5, 7, 14, 3, 11, 14])
    for r, k in enumerate(lengths.tolist()):
        caps[r, k - 1] = 3
        caps[r, k:] = 0
    tidx = torch.tensor([0, 1, 2, 2, 1, 0, 0, 1, 2])
    with torch.no_grad():
        ref32 = score_captions(model.cuda(), templates, tidx.cuda(), caps.cuda(), lengths.cuda(), batch_size=4).cpu()
        m16 = model.bfloat16()
        got = score_captions(m16, templates, tidx.cuda(), caps.cuda(), lengths.cuda(), batch_size=4).cpu()
        logits = m16(templates[tidx.cuda()], caps[:, :-1].cuda(), None)[:, :l].contiguous()
        want = sequence_perplexity(logits, caps.cuda(), lengths.cuda()).cpu()
    assert torch.allclose(got, want, rtol=2e-3), (got, want)
    assert torch.allclose(got, ref32, rtol=0.25), (got, ref32)


@pytest.mark.parametrize("v", (9, 71, 125, 129, 1552, 3451, 36541 - 189 + 5))
@pytest.mark.parametrize("dt", (torch.bfloat16, torch.float16))
def test_vocab_logprob_any_vocabulary_size(v, dt):
    """dh_vocab_logprob at 600 rows (the 256-column-tile form) for vocabularies whose last tile's second half lies past V
    (V % 256 in 1..128, V < 128: the char-level V = 71) -- found by tools/fuzz_scoring.py: the tile's non-existent groups were
    written into the next row's slots.  Against fp32 math on the same 16-bit operands, and equal to the 128-column-tile form
    (fewer than 512 rows) on the rows both see."""
    from deephumor_amd import hip
    g = torch.Generator().manual_seed(v)
    m, k = 600, 192
    a = torch.randn(m, k, generator=g).to(dt)
    w = (torch.randn(v, k, generator=g) * (2.5 / k ** 0.5)).to(dt)
    b = torch.randn(v, generator=g) * 0.5
    t = torch.randint(0, v, (m,), generator=g)
    want = torch.nn.functional.linear(a.float(), w.float(), b).double().log_softmax(-1).gather(-1, t[:, None])[:, 0]
    got = hip.vocab_logprob(a.cuda(), w.cuda(), b.cuda(), t.cuda()).double().cpu()
    assert bool(torch.isfinite(got).all())
    assert float((got - want).abs().max()) < (2e-2 if dt == torch.bfloat16 else 3e-3)
    small = hip.vocab_logprob(a[:300].cuda(), w.cuda(), b.cuda(), t[:300].cuda()).double().cpu()
    assert float((small - got[:300]).abs().max()) < 1e-5


def test_bf16_scoring_falls_back_for_widths_outside_the_fused_kernels_contract():
    """A 16-bit decoder whose classifier input is not a multiple of 64 (dh_vocab_logprob: K % 64 == 0, K >= 128) is scored
    through its materialised logits instead of raising."""
    import deephumor_amd.models as M
    from deephumor_amd.experiments import score_captions
    from deephumor_amd.synth import load_synthetic
    model = load_synthetic(M.CaptioningLSTM(300, emb_dim=64, hidden_size=72, num_layers=1).eval(), seed=5).cuda()
    templates = synth_images(2, seed=0).cuda()
    g = torch.Generator().manual_seed(1)
    caps = torch.randint(6, 300, (5, 8), generator=g)
    lengths = torch.tensor([8, 4, 6, 8, 2])
    for r, k in enumerate(lengths.tolist()):
        caps[r, k - 1] = 3
        caps[r, k:] = 0
    tidx = torch.tensor([0, 1, 1, 0, 1]).cuda()
    with torch.no_grad():
        ref32 = score_captions(model, templates, tidx, caps.cuda(), lengths.cuda()).cpu()
        got = score_captions(model.bfloat16(), templates, tidx, caps.cuda(), lengths.cuda()).cpu()
    assert torch.allclose(got, ref32, rtol=0.25), (got, ref32)
