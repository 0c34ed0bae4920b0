"""16-bit throughput paths (BASELINE configs C2-C4 name bf16, C5 fp16): bf16 / fp16 storage, 16-bit MFMA operands,
fp32 accumulation.  Every test of this module runs once per type (fixture ``half``).  Bit-exact greedy ids are only
promised by the fp32 path; here the bar is closeness to the reference's fp32 logits at the
type's resolution (|logit| std ~2.7; bf16 has 8 significant bits -> errors of a few 1e-2, fp16 has 11 -> a few 1e-3,
gated 6x tighter) and agreement of teacher-forced arg-max tokens."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from helpers import KINDS, captions_and_lengths, golden, synthetic_sd, synth_images  # noqa: E402


@pytest.fixture(scope="module")
def hip():
    from deephumor_amd import hip as h
    h.load()
    return h


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed + sum(shape)))


HALF = torch.bfloat16


@pytest.fixture(params=[torch.bfloat16, torch.float16], ids=["bf16", "f16"], autouse=True)
def half(request):
    """Storage / MFMA operand type of the test (the module-level ``bf`` helper rounds operands to it)."""
    global HALF
    HALF = request.param
    yield request.param
    HALF = torch.bfloat16


def bf(x):
    return x.to(HALF)


def test_rowops_bf16(hip):
    d = 512
    x, y, g, b = bf(rnd(37, d, seed=1)), bf(rnd(37, d, seed=2)), rnd(d, seed=3), rnd(d, seed=4)
    out = hip.add_layernorm(x.cuda(), y.cuda(), g.cuda(), b.cuda())
    ref = F.layer_norm(x.float() + y.float(), (d,), g, b, 1e-5)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), atol=3e-2, rtol=1e-2)
    tok, pos, start = bf(rnd(50, d, seed=5)), bf(rnd(20, d, seed=6)), bf(rnd(3, d, seed=7))
    tokens = torch.randint(0, 50, (6, 8), dtype=torch.int32)
    o = torch.empty(6, d, device="cuda", dtype=HALF)
    hip.embed_rows(tok.cuda(), pos.cuda(), start.cuda(), tokens.cuda(), o, 6, 2, 1, 4, 22.625)
    np.testing.assert_allclose(o.float().cpu().numpy(), (tok.float()[tokens[:, 3].long()] / 22.625 + pos.float()[4]).numpy(),
                               atol=2e-2, rtol=1e-2)
    e = bf(rnd(10, d, seed=8))
    e[3, 5] = 0.0
    assert hip.enc_key_mask(e.cuda()).cpu().tolist() == [0, 0, 0, 1, 0, 0, 0, 0, 0, 0]


@pytest.mark.parametrize("t", [0, 5, 32])
def test_attention_bf16(hip, t):
    n_img, beam, d, h, tmax = 3, 4, 512, 8, 40
    r, dh = n_img * beam, 64
    qkv = bf(rnd(r, 3 * d, seed=t))
    kc, vc = bf(rnd(tmax + 1, r, d, seed=1)), bf(rnd(tmax + 1, r, d, seed=2))
    g = torch.Generator().manual_seed(t)
    src = (torch.arange(r)[:, None] // beam * beam + torch.randint(0, beam, (r, tmax + 1), generator=g)).int()
    tokens = torch.randint(0, 5, (r, tmax), generator=g, dtype=torch.int32)
    out = torch.empty(r, d, device="cuda", dtype=HALF)
    kcd, vcd = kc.cuda(), vc.cuda()
    hip.attn_self_decode(qkv.cuda(), kcd, vcd, src.cuda(), tokens.cuda(), out, n_img, beam, 1, r, t, d, h, 8.0, 0)
    q32, kc32, vc32 = qkv.float(), kc.float(), vc.float()
    for row in range(r):
        keys = torch.stack([kc32[j, src[row, j]] for j in range(t)] + [q32[row, d:2 * d]]).view(t + 1, h, dh)
        vals = torch.stack([vc32[j, src[row, j]] for j in range(t)] + [q32[row, 2 * d:]]).view(t + 1, h, dh)
        masked = torch.tensor([False] + [bool(tokens[row, j - 1] == 0) for j in range(1, t + 1)])
        energy = (torch.einsum("hd,lhd->hl", q32[row, :d].view(h, dh), keys) / 8.0).masked_fill(masked[None], -1e8)
        ref = torch.einsum("hl,lhd->hd", torch.softmax(energy, -1), vals).reshape(-1)
        np.testing.assert_allclose(out[row].float().cpu().numpy(), ref.numpy(), atol=2e-2, rtol=1e-2)
    assert torch.equal(kcd[t].cpu(), qkv[:, d:2 * d]) and torch.equal(vcd[t].cpu(), qkv[:, 2 * d:])
    # cross attention
    s = 49
    q, kv = bf(rnd(r, d, seed=11)), bf(rnd(n_img * s, 2 * d, seed=12))
    mask = torch.zeros(n_img * s, dtype=torch.uint8)
    mask[5] = 1
    hip.attn_cross_decode(q.cuda(), kv.cuda(), mask.cuda(), out, n_img, beam, s, d, h, 8.0)
    for row in range(r):
        i = row // beam
        keys, vals = kv.float()[i * s:(i + 1) * s, :d].reshape(s, h, dh), kv.float()[i * s:(i + 1) * s, d:].reshape(s, h, dh)
        energy = (torch.einsum("hd,lhd->hl", q.float()[row].view(h, dh), keys) / 8.0).masked_fill(mask[i * s:(i + 1) * s].bool()[None], -1e8)
        ref = torch.einsum("hl,lhd->hd", torch.softmax(energy, -1), vals).reshape(-1)
        np.testing.assert_allclose(out[row].float().cpu().numpy(), ref.numpy(), atol=2e-2, rtol=1e-2)


def test_lstm_and_pools_bf16(hip):
    n_img, beam, e, hh, nl, v = 3, 2, 256, 512, 2, 40
    r = n_img * beam
    emb, h_prev, c_prev = bf(rnd(v, e, seed=1)), bf(rnd(nl, r, hh, seed=3)), rnd(nl, r, hh, seed=4)
    tokens = torch.randint(0, v, (r, 6), dtype=torch.int32)
    hpar = torch.tensor([1, 0, 3, 3, 4, 5], dtype=torch.int32)
    xcat0 = torch.zeros(r, e + hh, device="cuda", dtype=HALF)
    xcatl = torch.zeros(nl - 1, r, 2 * hh, device="cuda", dtype=HALF)
    c_cur = torch.zeros(nl, r, hh, device="cuda")
    hip.lstm_prepare(emb.cuda(), None, tokens.cuda(), 2, hpar.cuda(), h_prev.cuda(), c_prev.cuda(), xcat0, xcatl, c_cur,
                     r, beam, 1, r, nl, e, hh)
    assert torch.equal(xcat0[:, :e].cpu(), emb[tokens[:, 2].long()]) and torch.equal(xcat0[:, e:].cpu(), h_prev[0][hpar.long()])
    assert torch.equal(xcatl[0][:, hh:].cpu(), h_prev[1][hpar.long()]) and torch.equal(c_cur.cpu(), c_prev[:, hpar.long()])
    gates, c0 = rnd(r, 4 * hh, seed=5) * 2, rnd(r, hh, seed=6)
    h_new, c_new = torch.zeros(r, hh, device="cuda", dtype=HALF), torch.zeros(r, hh, device="cuda")
    h_out = torch.zeros(r, hh, device="cuda", dtype=HALF)
    hip.lstm_cell(gates.cuda(), c0.cuda(), h_new, c_new, h_out, hh, r, 1, hh)
    gi, gf, gg, go = gates.chunk(4, 1)
    c1 = torch.sigmoid(gf) * c0 + torch.sigmoid(gi) * torch.tanh(gg)
    np.testing.assert_allclose(c_new.cpu().numpy(), c1.numpy(), atol=2e-6)
    np.testing.assert_allclose(h_new.float().cpu().numpy(), (torch.sigmoid(go) * torch.tanh(c1)).numpy(), atol=5e-3)
    assert torch.equal(h_new, h_out)
    # stem + channels-last pools
    x, w = rnd(2, 3, 64, 64, seed=7), rnd(64, 3, 7, 7, seed=8) * 0.1
    sc, sh = rnd(64, seed=9).abs() + 0.5, rnd(64, seed=10)
    y = hip.stem_conv_nhwc(x.cuda(), w.cuda(), sc.cuda(), sh.cuda(), out_dtype=HALF)
    ref = torch.relu(F.conv2d(x, w, stride=2, padding=3) * sc[None, :, None, None] + sh[None, :, None, None])
    np.testing.assert_allclose(y.float().cpu().permute(0, 3, 1, 2).numpy(), ref.numpy(), atol=3e-2, rtol=1e-2)
    p = hip.maxpool3x3s2_nhwc(y)
    assert torch.equal(p.float().cpu().permute(0, 3, 1, 2), F.max_pool2d(y.float().cpu().permute(0, 3, 1, 2), 3, 2, 1))
    f = bf(rnd(5, 7, 7, 2048, seed=11))
    np.testing.assert_allclose(hip.avgpool_nhwc(f.cuda()).float().cpu().numpy(), f.float().mean(dim=(1, 2)).numpy(), atol=1e-2)


@pytest.mark.parametrize("kind", KINDS)
def test_models_bf16_close_to_reference(kind):
    import deephumor_amd.models as M
    g = golden(f"g2g3_{kind}.npz")
    sd, hp = synthetic_sd(kind)
    model = getattr(M, kind)(**hp).eval()
    model.load_state_dict(sd)
    model = model.cuda().to(HALF)
    images = synth_images(4, seed=0)
    cap, lengths, labels = captions_and_lengths()
    with torch.no_grad():
        args = (images.cuda(), cap.cuda(), lengths) + ((labels.cuda(),) if "WithLabels" in kind else ())
        out = model(*args)
        assert out.dtype == torch.float32 and tuple(out.shape) == tuple(g["forward_shape"])
        ref = torch.from_numpy(g["forward_logits01"])
        err = (out[:2].cpu() - ref).abs()
        if HALF == torch.bfloat16:
            assert float(err.max()) < 0.6 and float(err.mean()) < 0.08          # logits std ~2.7, bf16 ~ 2^-8
            assert float((out[:2].cpu().argmax(-1) == ref.argmax(-1)).float().mean()) > 0.93
        else:                                                                   # fp16: 3 more mantissa bits
            assert float(err.max()) < 0.1 and float(err.mean()) < 0.012
            assert float((out[:2].cpu().argmax(-1) == ref.argmax(-1)).float().mean()) > 0.98
        gargs = (images.cuda(), labels.cuda()) if "WithLabels" in kind else (images.cuda(),)
        toks, lens = model.generate_batch(*gargs, max_len=32, beam_size=1, top_k=1)
        first_ok = sum(int(toks[i, 0]) == int(g[f"greedy_{i}"][0]) for i in range(4))
        assert first_ok >= (3 if HALF == torch.bfloat16 else 4)
        toks, lens = model.generate_batch(*gargs, max_len=32, beam_size=5, top_k=50, seed=3)
        assert tuple(toks.shape) == (4, 32) and int(toks.max()) < 1000 and not bool((toks == 1).any())


@pytest.mark.parametrize("padded", [False, True])
@pytest.mark.parametrize("rows", [37, 300])
@pytest.mark.parametrize("v,beam,top_k", [(36541, 5, 50), (1000, 3, 16), (4000, 7, 50)])
def test_vocab_logits_group_max_and_guided_sampling(hip, v, beam, top_k, rows, padded):
    """dh_vocab_logits = dh_linear(out fp32) + per-row maxima of every 64-column group; the group-guided row
    sampler picks exactly what the full-row sampler picks."""
    k = 512
    a, w, b = bf(rnd(rows, k, seed=1)), bf(rnd(v, k, seed=2) * 0.12), rnd(v, seed=3)
    logits = torch.empty(rows, (v + 63) // 64 * 64 if padded else v, device="cuda")[:, :v]   # decoders pad the row stride
    ng = hip.n_groups(v)
    gmax = torch.full((rows, ng), float("nan"), device="cuda")
    hip.vocab_logits(a.cuda(), w.cuda(), b.cuda(), logits, gmax)
    ref = hip.linear(a.cuda(), w.cuda(), b.cuda(), out_dtype=torch.float32)
    assert torch.equal(logits, ref)
    pad = torch.full((rows, ng * 64 - v), float("-inf"), device="cuda")
    want = torch.cat([logits, pad], 1).view(rows, ng, 64).max(-1).values
    assert torch.equal(gmax, want)
    noise = torch.ones(rows, logits.stride(0))                       # explicit noise shares the logits row stride
    noise[:, :v] = torch.empty(rows, v).exponential_(1, generator=torch.Generator().manual_seed(5))
    noise = noise.cuda()
    out = []
    for guided in (False, True):
        pi = torch.empty(rows, beam, dtype=torch.int32, device="cuda")
        pv = torch.empty(rows, beam, device="cuda")
        err = torch.zeros(1, dtype=torch.int32, device="cuda")
        if guided:
            hip.beam_row_sample_groups(logits, v, gmax, rows, 1, beam, top_k, 1.1, 1, noise, 0, 0, 0, pi, pv, err)
        else:
            hip.beam_row_sample(logits, v, rows, 1, beam, top_k, 1.1, 1, noise, 0, 0, 0, pi, pv, err)
        assert int(err.item()) == 0
        out.append((pi.cpu(), pv.cpu()))
    assert torch.equal(out[0][0], out[1][0])
    np.testing.assert_allclose(out[0][1].numpy(), out[1][1].numpy(), atol=1e-6)


def test_vocab_logits_without_bias(hip):
    """bias = NULL through every classifier kernel that stages a bias strip by LDS-DMA (A-stationary K = 512 default, the tile
    kernel for K != 512, the group-maxima-only form): the strip comes from the zero page, never from address 0."""
    for rows, v, k in ((1280, 4000, 512), (300, 1000, 256), (130, 36541, 512)):
        a, w = bf(rnd(rows, k, seed=21)).cuda(), bf(rnd(v, k, seed=22) * 0.1).cuda()
        ref = hip.linear(a, w, None, out_dtype=torch.float32)
        logits = torch.full((rows, v), float("nan"), device="cuda")
        gmax = torch.full((rows, hip.n_groups(v)), float("nan"), device="cuda")
        hip.vocab_logits(a, w, None, logits, gmax)
        assert torch.equal(logits, ref)
        g2 = torch.full_like(gmax, float("nan"))
        hip.vocab_logits(a, w, None, None, g2)
        assert torch.equal(g2, gmax)


def test_vocab_logits_full_size_repeatable(hip):
    """BASELINE-size classifier (1280 beam rows x 36,541 tokens, K = 512): the persistent kernel (several tiles per
    workgroup, ring never drained) against the one-tile-per-workgroup GEMM, bit for bit, over repeated launches."""
    rows, v, k = 1280, 36541, 512
    a, w, b = bf(rnd(rows, k, seed=11)).cuda(), (bf(rnd(v, k, seed=12) * 0.1)).cuda(), rnd(v, seed=13).cuda()
    ref = hip.linear(a, w, b, out_dtype=torch.float32)
    ng = hip.n_groups(v)
    pad = torch.full((rows, ng * 64 - v), float("-inf"), device="cuda")
    want_g = torch.cat([ref, pad], 1).view(rows, ng, 64).max(-1).values
    for rep in range(6):
        # row stride of whole 128-column panels (what the decoders allocate): the 256-row A-stationary kernel; a stride that ends
        # inside the last panel: the 128-row kernel with its element-wise edge tiles
        for ld in ((v + 127) // 128 * 128, (v + 63) // 64 * 64):
            logits = torch.full((rows, ld), float("nan"), device="cuda")[:, :v]
            gmax = torch.full((rows, ng), float("nan"), device="cuda")
            hip.vocab_logits(a, w, b, logits, gmax)
            assert torch.equal(logits, ref), (rep, ld)
            assert torch.equal(gmax, want_g), (rep, ld)


@pytest.mark.parametrize("rows,v", [(512, 5000), (768, 36541), (2560, 4100), (1280, 8192)])
def test_vocab_logits_256_row_tiles(hip, rows, v):
    """vocab_areg256_kernel (A-stationary, 256-row tiles, no edge path): bit-equal to dh_linear for row counts that are
    multiples of 256, vocabularies that end inside a panel / inside a 64-column group (clamped weight rows must not disturb
    the boundary group's maximum) and exact multiples of 128; padding columns of the row stride are the only bytes it may
    write besides the logits."""
    k = 512
    a, w, b = bf(rnd(rows, k, seed=31)).cuda(), bf(rnd(v, k, seed=32) * 0.1).cuda(), rnd(v, seed=33).cuda()
    ref = hip.linear(a, w, b, out_dtype=torch.float32)
    ng = hip.n_groups(v)
    pad = torch.full((rows, ng * 64 - v), float("-inf"), device="cuda")
    want_g = torch.cat([ref, pad], 1).view(rows, ng, 64).max(-1).values
    ld = (v + 127) // 128 * 128
    buf = torch.full((rows + 1, ld), float("nan"), device="cuda")
    gbuf = torch.full((rows + 1, ng + 3), float("nan"), device="cuda")       # (n_groups counts whole 128-column panels: a group past V holds -inf)
    hip.vocab_logits(a, w, b, buf[:rows, :v], gbuf[:rows, :ng])
    assert torch.equal(buf[:rows, :v], ref)
    assert torch.equal(gbuf[:rows, :ng], want_g)
    assert bool(torch.isnan(buf[rows]).all()) and bool(torch.isnan(gbuf[rows]).all()) and bool(torch.isnan(gbuf[:rows, ng:]).all())


@pytest.mark.parametrize("rows,v,with_bias,with_logits", [(1280, 36541, True, True), (80, 300, True, True), (640, 8192, False, True),
                                                           (320, 4100, True, False), (2560, 1000, True, True), (160, 36541, True, True),
                                                           (380, 36541, True, True), (37, 3001, True, True), (5, 1000, False, True), (600, 5000, True, False),
                                                           (81, 2000, True, True), (370, 36541, True, True)])
def test_vocab_logits_wreg_equals_vocab_logits(hip, rows, v, with_bias, with_logits):
    """dh_vocab_logits_wreg (weights from L2 into registers in fragment order, 80-row activation blocks resident in LDS, no barrier)
    against dh_vocab_logits on the same operands, bit for bit: logits on every column of the padded row stride the old kernel
    defines (copies of logit[V - 1] past V where both write), group maxima incl. the boundary group and the -inf groups past V;
    vocabularies that end inside a chunk / a group / on a chunk boundary, with and without bias, group maxima only; nothing
    written outside the buffers; bf16 and fp16."""
    k = 512
    a, w = bf(rnd(rows, k, seed=41)).cuda(), bf(rnd(v, k, seed=42) * 0.1).cuda()
    b = rnd(v, seed=43).cuda() if with_bias else None
    vpad = (v + 255) // 256 * 256
    ng = vpad // 64
    assert hip.vocab_logits_wreg_supported(rows, v, k, vpad if with_logits else 0, ng)
    # (round 5: ANY row count up to 640 -- the small-shard regime: 380 = C5's 38-template shard x beam 10 -- with idle padding row blocks
    #  and a masked last block; above 640 rows only 80 x a power of two)
    assert hip.vocab_logits_wreg_supported(rows + 16, v, k, vpad, ng) == (rows + 16 <= 640) and not hip.vocab_logits_wreg_supported(rows, v, 256, vpad, ng)
    assert not hip.vocab_logits_wreg_supported(rows, v, k, vpad - 4, ng) and not hip.vocab_logits_wreg_supported(rows, v, k, vpad, ng - 1)
    wp, bp = hip.pack_vocab_weights(w, b)
    ref_l = torch.full((rows + 1, vpad), float("nan"), device="cuda")
    ref_g = torch.full((rows + 1, ng), float("nan"), device="cuda")
    hip.vocab_logits(a, w, b, ref_l[:rows, :v] if with_logits else None, ref_g[:rows])
    got_l = torch.full((rows + 1, vpad), float("nan"), device="cuda")
    got_g = torch.full((rows + 1, ng + 2), float("nan"), device="cuda")
    hip.vocab_logits_wreg(a, wp, bp, v, got_l[:rows] if with_logits else None, got_g[:rows, :ng])
    old_groups = hip.n_groups(v)                        # the old kernel defines whole 128-column panels of groups
    assert torch.equal(got_g[:rows, :old_groups], ref_g[:rows, :old_groups])
    assert bool((got_g[:rows, old_groups:ng] == float("-inf")).all()) and bool(torch.isnan(got_g[rows]).all()) and bool(torch.isnan(got_g[:rows, ng:]).all())
    if with_logits:
        assert torch.equal(got_l[:rows, :v], ref_l[:rows, :v])
        assert bool((got_l[:rows, v:] == got_l[:rows, v - 1:v]).all())      # padding columns: copies of logit[V - 1]
        assert bool(torch.isnan(got_l[rows]).all())
    else:
        assert bool(torch.isnan(got_l).all())


@pytest.mark.parametrize("rows,row_mult,e,hh,use_tokens,with_state", [
    (37, 1, 64, 128, True, True), (8, 5, 256, 512, False, False), (130, 1, 512, 512, False, True),
    (64, 1, 256, 512, True, True)])
def test_lstm_layer_fused_matches_cell_math(hip, rows, row_mult, e, hh, use_tokens, with_state):
    """dh_lstm_layer_fused (operand gather + gate GEMM + cell update in one launch, gate-interleaved weights) against
    the LSTM cell equations in fp32 on the same bf16-rounded operands, incl. the beam-parent gather of the state."""
    g = torch.Generator().manual_seed(rows * 7 + e)
    rows_total = rows * row_mult
    w = bf(torch.randn(4 * hh, e + hh, generator=g) * 0.05)
    b = torch.randn(4 * hh, generator=g) * 0.1
    w_il = w.view(4, hh, -1).permute(1, 0, 2).reshape(4 * hh, -1).contiguous().cuda()
    b_il = b.view(4, hh).t().reshape(-1).contiguous().cuda()
    emb = bf(torch.randn(50, e, generator=g))
    tokens = torch.randint(0, 50, (rows_total, 6), generator=g, dtype=torch.int32)
    x_rows = bf(torch.randn(rows, e, generator=g))
    h_prev = bf(torch.randn(rows_total, hh, generator=g) * 0.5)
    c_prev = torch.randn(rows_total, hh, generator=g)
    hparent = torch.randint(0, rows_total, (rows_total,), generator=g, dtype=torch.int32)
    h_next = torch.full((rows_total, hh), 9.0).to(HALF).cuda()
    c_next = torch.full((rows_total, hh), 9.0).cuda()
    h_out = torch.zeros(rows, hh + 8).to(HALF).cuda()[:, :hh]
    hip.lstm_layer_fused(None if use_tokens else x_rows.cuda(), 1, emb.cuda() if use_tokens else None,
                         tokens.cuda() if use_tokens else None, 3, h_prev.cuda() if with_state else None,
                         c_prev.cuda() if with_state else None, hparent.cuda() if with_state else None, h_next, c_next, h_out,
                         w_il, b_il, rows, row_mult, e, hh)
    rl = torch.arange(rows) * row_mult
    x = emb.float()[tokens[rl, 3].long()] if use_tokens else x_rows.float()
    hp = hparent[rl].long()
    h0 = h_prev.float()[hp] if with_state else torch.zeros(rows, hh)
    c0 = c_prev[hp] if with_state else torch.zeros(rows, hh)
    gates = torch.cat([x, h0], 1) @ w.float().t() + b
    i, f, gg, o = gates.split(hh, dim=1)
    c1 = torch.sigmoid(f) * c0 + torch.sigmoid(i) * torch.tanh(gg)
    h1 = torch.sigmoid(o) * torch.tanh(c1)
    np.testing.assert_allclose(c_next.cpu()[rl].numpy(), c1.numpy(), atol=2e-3, rtol=2e-3)
    np.testing.assert_allclose(h_next.cpu().float()[rl].numpy(), h1.numpy(), atol=1e-2, rtol=1e-2)
    assert torch.equal(h_out.cpu(), h_next.cpu()[rl])
    if row_mult > 1:                        # rows between the logical rows are untouched
        assert float(c_next.cpu()[1].min()) == 9.0


@pytest.mark.parametrize("rows,row_mult,e,hh,use_tokens,with_state", [
    (1280, 1, 256, 512, True, True), (1280, 1, 512, 512, False, True), (256, 5, 256, 512, False, False),
    (333, 1, 512, 512, True, True), (77, 3, 256, 512, True, True)])
def test_lstm_layer_wreg_equals_fused(hip, rows, row_mult, e, hh, use_tokens, with_state):
    """dh_lstm_layer_wreg (gate weights stationary in registers, the whole [80 rows x K] activation block in LDS, one barrier) against
    dh_lstm_layer_fused on the same operands: h / c / h_out bit for bit -- decode shapes (K = 768 and 1024), the zero-state first
    step (256 compact rows, row_mult 5), row counts that do not fill the last 80-row block."""
    g = torch.Generator().manual_seed(rows * 11 + e)
    rows_total = rows * row_mult
    w = bf(torch.randn(4 * hh, e + hh, generator=g) * 0.05)
    b = torch.randn(4 * hh, generator=g) * 0.1
    w_il = w.view(4, hh, -1).permute(1, 0, 2).reshape(4 * hh, -1).contiguous().cuda()
    b_il = b.view(4, hh).t().reshape(-1).contiguous().cuda()
    w_pk = hip.pack_mfma_fragments(w_il)
    emb = bf(torch.randn(500, e, generator=g)).cuda()
    tokens = torch.randint(0, 500, (rows_total, 6), generator=g, dtype=torch.int32).cuda()
    x_rows = bf(torch.randn(rows, e, generator=g)).cuda()
    h_prev = bf(torch.randn(rows_total, hh, generator=g) * 0.5).cuda()
    c_prev = torch.randn(rows_total, hh, generator=g).cuda()
    hparent = torch.randint(0, rows_total, (rows_total,), generator=g, dtype=torch.int32).cuda()
    outs = []
    for fn, wt in ((hip.lstm_layer_fused, w_il), (hip.lstm_layer_wreg, w_pk)):
        h_next = torch.full((rows_total, hh), 9.0).to(HALF).cuda()
        c_next = torch.full((rows_total, hh), 9.0).cuda()
        h_out = torch.zeros(rows, hh + 8).to(HALF).cuda()[:, :hh]
        fn(None if use_tokens else x_rows, 1, emb if use_tokens else None, tokens if use_tokens else None, 3,
           h_prev if with_state else None, c_prev if with_state else None, hparent if with_state else None, h_next, c_next, h_out,
           wt, b_il, rows, row_mult, e, hh)
        outs.append((h_next, c_next, h_out.clone()))
    for a, b_ in zip(outs[0], outs[1]):
        assert torch.equal(a, b_)


@pytest.mark.parametrize("rows,v", [(300, 36541), (1280, 36541), (37, 1000), (200, 4000)])
def test_vocab_logprob_matches_log_softmax(hip, rows, v):
    """dh_vocab_logprob (log-sum-exp partials in the classifier GEMM epilogue, logits never written) against
    log_softmax of the materialised fp32 logits of the same GEMM."""
    k = 512
    a, w, b = bf(rnd(rows, k, seed=21)).cuda(), (bf(rnd(v, k, seed=22) * 0.1)).cuda(), rnd(v, seed=23).cuda()
    targets = torch.randint(0, v, (rows,), generator=torch.Generator().manual_seed(5))
    targets[0], targets[1] = v - 1, 0
    ref = torch.log_softmax(hip.linear(a, w, b, out_dtype=torch.float32), dim=-1).cpu()
    want = ref[torch.arange(rows), targets]
    got = hip.vocab_logprob(a, w, b, targets.cuda()).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=2e-4, rtol=0)


@pytest.mark.parametrize("kind", ["CaptioningLSTM", "CaptioningTransformer", "CaptioningTransformerWithLabels"])
@pytest.mark.parametrize("n_img,beam,top_k,max_len", [(1, 1, 1, 5), (3, 2, 7, 9), (7, 7, 10, 6), (5, 16, 16, 4)])
def test_bf16_generate_odd_shapes(kind, n_img, beam, top_k, max_len):
    """bf16 path on shapes that are not multiples of any tile (1 / 3 / 5 / 7 images, beam 1..16): tokens are valid,
    two runs agree bit for bit, and a split of the batch (global image offset ``img0``) reproduces the batched result --
    exercises the edge tiles of the fused convolution, LSTM-step and classifier kernels."""
    import deephumor_amd.models as M
    sd, hp = synthetic_sd(kind)
    model = getattr(M, kind)(**hp).eval()
    model.load_state_dict(sd)
    model = model.cuda().to(HALF)
    imgs = synth_images(n_img, seed=5).cuda()
    labels = torch.randint(6, 1000, (n_img, 3), generator=torch.Generator().manual_seed(1)).cuda()
    args = (lambda lo, hi: (imgs[lo:hi], labels[lo:hi])) if "WithLabels" in kind else (lambda lo, hi: (imgs[lo:hi],))
    kw = dict(max_len=max_len, beam_size=beam, top_k=top_k, temperature=0.9, seed=11)
    with torch.no_grad():
        t1, l1 = model.generate_batch(*args(0, n_img), **kw)
        t2, l2 = model.generate_batch(*args(0, n_img), **kw)
        assert torch.equal(t1, t2) and torch.equal(l1, l2)
        assert tuple(t1.shape) == (n_img, max_len) and int(t1.max()) < 1000 and int(t1.min()) >= 0
        assert not bool((t1 == 1).any()) and int(l1.max()) <= max_len and int(l1.min()) >= 1
        if n_img > 1:
            k = n_img // 2
            ta, la = model.generate_batch(*args(0, k), img0=0, **kw)
            tb, lb = model.generate_batch(*args(k, n_img), img0=k, **kw)
            # bf16 GEMM tiles differ with the batch size only in which rows share a tile, not in any row's arithmetic
            assert torch.equal(torch.cat([ta, tb]), t1) and torch.equal(torch.cat([la, lb]), l1)


@pytest.mark.parametrize("n_img,beam,s", [(3, 5, 49), (2, 16, 64), (5, 1, 10), (7, 10, 49)])
def test_cross_attention_on_matrix_cores(hip, n_img, beam, s):
    """dh_attn_cross_pack + dh_attn_cross_decode_packed (one wave per (image, head), operands straight into MFMA
    fragments) against masked softmax attention in fp32 on the same rounded operands, and against the LDS kernel."""
    d, h, dh = 512, 8, 64
    r = n_img * beam
    q, kv = bf(rnd(r, d, seed=31)), bf(rnd(n_img * s, 2 * d, seed=32))
    mask = torch.zeros(n_img * s, dtype=torch.uint8)
    mask[3] = 1
    if n_img > 1:
        mask[s:2 * s] = 1                                   # an image with every key masked -> uniform weights
    kp, vt = hip.attn_cross_pack(kv.cuda(), n_img, s, d, h)
    out = torch.full((r, d), 7.0, device="cuda", dtype=HALF)
    hip.attn_cross_decode_packed(q.cuda(), kp, vt, mask.cuda(), out, n_img, beam, s, d, h, 8.0)
    old = torch.empty_like(out)
    hip.attn_cross_decode(q.cuda(), kv.cuda(), mask.cuda(), old, n_img, beam, s, d, h, 8.0)
    for row in range(r):
        i = row // beam
        keys = kv.float()[i * s:(i + 1) * s, :d].reshape(s, h, dh)
        vals = kv.float()[i * s:(i + 1) * s, d:].reshape(s, h, dh)
        energy = (torch.einsum("hd,lhd->hl", q.float()[row].view(h, dh), keys) / 8.0).masked_fill(mask[i * s:(i + 1) * s].bool()[None], -1e8)
        ref = torch.einsum("hl,lhd->hd", torch.softmax(energy, -1), vals).reshape(-1)
        np.testing.assert_allclose(out[row].float().cpu().numpy(), ref.numpy(), atol=2e-2, rtol=1e-2)
    np.testing.assert_allclose(out.float().cpu().numpy(), old.float().cpu().numpy(), atol=2e-2, rtol=1e-2)
    # prefill form: 21 positions per image in chunks of 16
    n_pos = 21
    qp = bf(rnd(n_img * n_pos, d, seed=33))
    got = hip.attn_cross_prefill_packed(qp.cuda(), kp, vt, mask.cuda(), n_img, n_pos, s, d, h, 8.0)
    want = hip.attn_cross_prefill(qp.cuda(), kv.cuda(), mask.cuda(), n_img, n_pos, s, d, h, 8.0)
    np.testing.assert_allclose(got.float().cpu().numpy(), want.float().cpu().numpy(), atol=2e-2, rtol=1e-2)


def _tile_stats(y):
    """Reference partial statistics: per (row, 64-column tile) mean and sum of squared deviations."""
    t = y.view(y.shape[0], -1, 64)
    mean = t.mean(-1)
    return torch.stack([mean, ((t - mean[..., None]) ** 2).sum(-1)], -1)


@pytest.mark.parametrize("m", [37, 1280, 256])
def test_linear_with_deferred_layernorm(hip, m):
    """dh_linear_ln: LayerNorm of the A rows applied on the accumulators (gamma / beta folded into weight / bias),
    LayerNorm of the residual rows applied in the epilogue, partial statistics of the output rows -- against
    F.layer_norm + F.linear in fp32 on the same rounded operands."""
    d, n = 512, 512
    g = torch.Generator().manual_seed(m)
    y = bf(torch.randn(m, d, generator=g) * 1.7 + 0.3)                       # pre-LayerNorm rows (non-zero mean)
    gamma, beta = torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g) * 0.2
    w, b = bf(torch.randn(n, d, generator=g) / d ** 0.5), torch.randn(n, generator=g) * 0.1
    stats = _tile_stats(y.float())
    # (a) output = LN(y) W^T + b via folding
    wf = bf(w.float() * gamma[None, :])
    bfold = b + (w.float() * beta[None, :]).sum(1)
    colsum = wf.float().sum(1)
    out = hip.linear_ln(y.cuda(), wf.cuda(), bfold.cuda(), a_ln=(stats.cuda(), 1e-5, colsum.cuda()))
    ln = F.layer_norm(y.float(), (d,), gamma, beta, 1e-5)
    want = F.linear(ln, w.float(), b)
    np.testing.assert_allclose(out.float().cpu().numpy(), want.numpy(), atol=6e-2 if HALF == torch.bfloat16 else 1.5e-2, rtol=2e-2)
    # (b) output = LN(y) + a W2^T + b2 with statistics of the output
    a = bf(torch.randn(m, 256, generator=g))
    w2, b2 = bf(torch.randn(n, 256, generator=g) / 16.0), torch.randn(n, generator=g) * 0.1
    out2, st2 = hip.linear_ln(a.cuda(), w2.cuda(), b2.cuda(), residual=y.cuda(), want_stats=True,
                              r_ln=(stats.cuda(), 1e-5, gamma.cuda(), beta.cuda()))
    want2 = ln + F.linear(a.float(), w2.float(), b2)
    np.testing.assert_allclose(out2.float().cpu().numpy(), want2.numpy(), atol=5e-2 if HALF == torch.bfloat16 else 8e-3, rtol=1e-2)
    ref_st = _tile_stats(out2.float().cpu())                                 # statistics of the ROUNDED output
    np.testing.assert_allclose(st2.cpu()[..., 0].numpy(), ref_st[..., 0].numpy(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(st2.cpu()[..., 1].numpy(), ref_st[..., 1].numpy(), atol=2e-3, rtol=1e-4)
    # (c) chained: the statistics reproduce the LayerNorm of the next consumer
    out3 = hip.linear_ln(out2, wf.cuda(), bfold.cuda(), a_ln=(st2, 1e-5, colsum.cuda()))
    want3 = F.linear(F.layer_norm(out2.float().cpu(), (d,), gamma, beta, 1e-5), w.float(), b)
    np.testing.assert_allclose(out3.float().cpu().numpy(), want3.numpy(), atol=6e-2 if HALF == torch.bfloat16 else 1.5e-2, rtol=2e-2)
    # plain form == dh_linear
    assert torch.equal(hip.linear_ln(a.cuda(), w2.cuda(), b2.cuda()), hip.linear(a.cuda(), w2.cuda(), b2.cuda()))


@pytest.mark.parametrize("n,h,w", [(2, 224, 224), (3, 64, 96), (1, 36, 28)])
def test_stem_conv_with_fused_maxpool(hip, n, h, w):
    """dh_conv2d_nhwc_bn_relu_maxpool == dh_conv2d_nhwc_bn_act + dh_maxpool3x3s2_nhwc bit for bit (partial 7 x 7 blocks at
    the image edges included), and close to F.conv2d + max_pool2d in fp32."""
    x = bf(rnd(n, h, w, 8, seed=41))
    x[..., 3:] = 0
    wgt = bf(rnd(64, 7, 7, 8, seed=42) * 0.08)
    sc, sh = rnd(64, seed=43).abs() + 0.5, rnd(64, seed=44) * 0.3
    fused = hip.conv2d_nhwc_bn_relu_maxpool(x.cuda(), wgt.cuda(), sc.cuda(), sh.cuda(), stride=2, pad=3)
    conv = hip.conv2d_nhwc_bn_act(x.cuda(), wgt.cuda(), sc.cuda(), sh.cuda(), relu=True, stride=2, pad=3)
    ref = hip.maxpool3x3s2_nhwc(conv)
    assert tuple(fused.shape) == tuple(ref.shape) and torch.equal(fused, ref)
    want = F.max_pool2d(torch.relu(F.conv2d(x.float().permute(0, 3, 1, 2), wgt.float().permute(0, 3, 1, 2), stride=2, padding=3)
                                   * sc[None, :, None, None] + sh[None, :, None, None]), 3, 2, 1)
    np.testing.assert_allclose(fused.float().cpu().permute(0, 3, 1, 2).numpy(), want.numpy(), atol=4e-2, rtol=2e-2)


@pytest.mark.parametrize("n,h,w,c", [(2, 56, 56, 64), (3, 8, 56, 64), (2, 28, 28, 128), (5, 4, 28, 128), (1, 4, 28, 128), (3, 28, 28, 128)])
def test_direct_conv3x3(hip, n, h, w, c):
    """dh_conv3x3_direct_nhwc (patch-resident direct convolution, stages 1-2 of the ResNet) against the implicit-GEMM
    dh_conv2d_nhwc_bn_act and against fp32 F.conv2d on the same 16-bit operands."""
    assert hip.conv3x3_direct_supported(h, w, c, c) and not hip.conv3x3_direct_supported(h, w + 4, c, c)
    x = bf(rnd(n, h, w, c, seed=61))
    wgt = bf(rnd(c, 3, 3, c, seed=62) * (9 * c) ** -0.5)
    sc, sh = rnd(c, seed=63).abs() + 0.5, rnd(c, seed=64) * 0.3
    got = hip.conv3x3_direct_nhwc(x.cuda(), wgt.cuda(), sc.cuda(), sh.cuda())
    old = hip.conv2d_nhwc_bn_act(x.cuda(), wgt.cuda(), sc.cuda(), sh.cuda(), relu=True, stride=1, pad=1)
    want = torch.relu(F.conv2d(x.float().permute(0, 3, 1, 2), wgt.float().permute(0, 3, 1, 2), padding=1) * sc[None, :, None, None]
                      + sh[None, :, None, None]).permute(0, 2, 3, 1)
    tol = dict(atol=3e-2, rtol=2e-2) if HALF == torch.bfloat16 else dict(atol=4e-3, rtol=3e-3)
    np.testing.assert_allclose(got.float().cpu().numpy(), want.numpy(), **tol)
    np.testing.assert_allclose(got.float().cpu().numpy(), old.float().cpu().numpy(), **tol)


@pytest.mark.parametrize("n,h,w,c", [(2, 56, 56, 64), (3, 8, 56, 64), (2, 28, 28, 128), (5, 4, 28, 128), (1, 4, 28, 128), (3, 28, 28, 128)])
def test_fused_bottleneck_tail(hip, n, h, w, c):
    """dh_bottleneck_tail_nhwc (3x3 conv2 + 1x1 conv3 + residual in one launch, the conv2 tile resident in LDS) against the
    two-launch route, bit for bit, and against fp32 math."""
    y1 = bf(rnd(n, h, w, c, seed=71)).cuda()
    w2 = bf(rnd(c, 3, 3, c, seed=72) * (9 * c) ** -0.5).cuda()
    w3 = bf(rnd(4 * c, 1, 1, c, seed=73) * c ** -0.5).cuda()
    s2, h2 = (rnd(c, seed=74).abs() + 0.5).cuda(), (rnd(c, seed=75) * 0.3).cuda()
    s3, h3 = (rnd(4 * c, seed=76).abs() + 0.5).cuda(), (rnd(4 * c, seed=77) * 0.3).cuda()
    res = bf(rnd(n, h, w, 4 * c, seed=78)).cuda()
    got = hip.bottleneck_tail_nhwc(y1, w2, s2, h2, w3, s3, h3, res)
    y2 = hip.conv3x3_direct_nhwc(y1, w2, s2, h2)
    two = hip.conv2d_nhwc_bn_act(y2, w3, s3, h3, residual=res, relu=True, stride=1, pad=0)
    assert torch.equal(got, two)
    y2f = torch.relu(F.conv2d(y1.float().cpu().permute(0, 3, 1, 2), w2.float().cpu().permute(0, 3, 1, 2), padding=1)
                     * s2.cpu()[None, :, None, None] + h2.cpu()[None, :, None, None])
    want = torch.relu(F.conv2d(bf(y2f).float(), w3.float().cpu().permute(0, 3, 1, 2)) * s3.cpu()[None, :, None, None]
                      + h3.cpu()[None, :, None, None] + res.float().cpu().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    tol = dict(atol=6e-2, rtol=3e-2) if HALF == torch.bfloat16 else dict(atol=8e-3, rtol=4e-3)
    np.testing.assert_allclose(got.float().cpu().numpy(), want.numpy(), **tol)


@pytest.mark.parametrize("h,w", [(224, 224), (192, 256), (256, 256), (160, 224)])
def test_encoder_16bit_other_image_sizes(h, w):
    """The 16-bit encoder at image sizes where the shape-specialised kernels do NOT apply (stage 3 is not 14 x 14: no one-image
    tail; stages 1-2 are not 56 / 28 wide: no patch-resident 3x3) falls back to the general kernels: embedding close to the fp32
    path's, and at 224 x 224 identical with and without the specialised kernels switched off (they are bit-identical by design)."""
    import os
    import deephumor_amd.models as M
    from deephumor_amd.synth import synth_state_dict
    enc = M.ImageEncoder(256, spatial_features=True).eval()
    enc.load_state_dict(synth_state_dict(enc.state_dict(), seed=1234))
    x = rnd(3, 3, h, w, seed=h + w).cuda()
    with torch.no_grad():
        emb32, sp32 = enc.cuda()(x)
        e16 = enc.to(HALF)
        emb16, sp16 = e16(x)
        assert emb16.shape == emb32.shape and sp16.shape == sp32.shape == (3, (h // 32) * (w // 32), 256)
        tol = 0.12 if HALF == torch.bfloat16 else 0.02
        assert float((emb16.float() - emb32).abs().max()) < tol * max(1.0, float(emb32.abs().max()))
        if (h, w) == (224, 224):
            from deephumor_amd import hip as H
            with H.option_scope(encoder_generic=2):     # every bottleneck through the implicit-GEMM tile kernel
                emb_g, sp_g = e16(x)
            assert torch.equal(emb_g, emb16) and torch.equal(sp_g, sp16)


@pytest.mark.parametrize("n", [1, 3, 9])
def test_stage3_bottleneck_tail(hip, n):
    """dh_bottleneck_tail_s3_nhwc (14 x 14 x 256 -> 1024: one image per workgroup, patch-resident 3x3, weights streamed into
    registers in MFMA fragment order, 1x1 expansion + residual in the same launch) against the two implicit-GEMM launches bit for
    bit -- fused and conv2-only forms -- and against fp32 math; dh_pack_mfma_fragments against its definition."""
    c, hw = 256, 14
    y1 = bf(rnd(n, hw, hw, c, seed=81)).cuda()
    w2 = bf(rnd(c, 3, 3, c, seed=82) * (9 * c) ** -0.5).cuda()
    w3 = bf(rnd(4 * c, 1, 1, c, seed=83) * c ** -0.5).cuda()
    s2, h2 = (rnd(c, seed=84).abs() + 0.5).cuda(), (rnd(c, seed=85) * 0.3).cuda()
    s3, h3 = (rnd(4 * c, seed=86).abs() + 0.5).cuda(), (rnd(4 * c, seed=87) * 0.3).cuda()
    res = bf(rnd(n, hw, hw, 4 * c, seed=88)).cuda()
    w2p, w3p = hip.pack_mfma_fragments(w2), hip.pack_mfma_fragments(w3.reshape(4 * c, c).contiguous())
    # fragment (s, rt, lane): 8 values k = 32 s + 8 (lane >> 4) .. of row 16 rt + (lane & 15)
    k2 = 9 * c
    want_p = w2.reshape(c // 16, 16, k2 // 32, 4, 8).permute(2, 0, 3, 1, 4).reshape(-1)
    assert torch.equal(w2p, want_p)
    y2_ref = hip.conv2d_nhwc_bn_act(y1, w2, s2, h2, relu=True, stride=1, pad=1)
    y2 = hip.bottleneck_tail_s3_nhwc(y1, w2p, s2, h2)
    assert torch.equal(y2, y2_ref)
    got = hip.bottleneck_tail_s3_nhwc(y1, w2p, s2, h2, w3p, s3, h3, res)
    two = hip.conv2d_nhwc_bn_act(y2_ref, w3, s3, h3, residual=res, relu=True, stride=1, pad=0)
    assert torch.equal(got, two)
    y2f = torch.relu(F.conv2d(y1.float().cpu().permute(0, 3, 1, 2), w2.float().cpu().permute(0, 3, 1, 2), padding=1)
                     * s2.cpu()[None, :, None, None] + h2.cpu()[None, :, None, None])
    want = torch.relu(F.conv2d(bf(y2f).float(), w3.float().cpu().permute(0, 3, 1, 2)) * s3.cpu()[None, :, None, None]
                      + h3.cpu()[None, :, None, None] + res.float().cpu().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    tol = dict(atol=6e-2, rtol=3e-2) if HALF == torch.bfloat16 else dict(atol=8e-3, rtol=4e-3)
    np.testing.assert_allclose(got.float().cpu().numpy(), want.numpy(), **tol)


@pytest.mark.parametrize("n,h,w", [(2, 224, 224), (3, 64, 96), (1, 36, 28), (1, 8, 4), (70, 60, 64)])
def test_direct_stem_convolution(hip, n, h, w):
    """dh_stem_conv7_bn_relu_maxpool (direct 7x7/2 convolution + BN + ReLU + maxpool, one launch) against fp32
    F.conv2d + max_pool2d on the same 16-bit-rounded operands -- partial 7 x 7 pooled blocks at the image edges and more
    patches than workgroups (the persistent loop) included -- and its two input formats against each other, bit for bit."""
    x = bf(rnd(n, 3, h, w, seed=51)).float()                      # fp32 values that are exact in the 16-bit type
    wgt = bf(rnd(64, 3, 7, 7, seed=52) * 0.08)
    sc, sh = rnd(64, seed=53).abs() + 0.5, rnd(64, seed=54) * 0.3
    wpk = hip.pack_stem_weight(wgt.cuda(), HALF)
    assert torch.equal(wpk[:, :, :7, :3].cpu(), wgt.permute(0, 2, 3, 1)) and float(wpk[:, :, 7].abs().max()) == 0 and float(wpk[..., 3].abs().max()) == 0
    got = hip.stem_conv7_bn_relu_maxpool(x.cuda(), wpk, sc.cuda(), sh.cuda())
    want = F.max_pool2d(torch.relu(F.conv2d(x, wgt.float(), stride=2, padding=3) * sc[None, :, None, None] + sh[None, :, None, None]), 3, 2, 1)
    assert tuple(got.shape) == (n, want.shape[2], want.shape[3], 64)
    np.testing.assert_allclose(got.float().cpu().permute(0, 3, 1, 2).numpy(), want.numpy(), atol=4e-2 if HALF == torch.bfloat16 else 6e-3, rtol=2e-2 if HALF == torch.bfloat16 else 3e-3)
    packed = hip.pack_nchw_to_nhwc8(x.cuda(), out_dtype=HALF)
    assert torch.equal(hip.stem_conv7_bn_relu_maxpool(packed, wpk, sc.cuda(), sh.cuda()), got)
    # against the implicit-GEMM stem (different summation order: close, not bit-equal)
    w8 = torch.zeros(64, 7, 7, 8, dtype=HALF)
    w8[..., :3] = wgt.permute(0, 2, 3, 1)
    old = hip.conv2d_nhwc_bn_relu_maxpool(packed, w8.cuda(), sc.cuda(), sh.cuda(), stride=2, pad=3)
    np.testing.assert_allclose(got.float().cpu().numpy(), old.float().cpu().numpy(), atol=4e-2 if HALF == torch.bfloat16 else 6e-3, rtol=2e-2)


def test_round16_keep_nonzero_and_fp16_spatial_features(hip):
    """dh_round16_keep_nonzero: round to nearest even, but a non-zero fp32 value never becomes zero (smallest subnormal of its sign
    instead); exact zeros stay zero.  ImageEncoder's fp16 spatial features go through it: the reference's zero-row test
    (transformers.py:480) must not fire on a feature that merely underflowed in fp16."""
    x = torch.tensor([0.0, -0.0, 1e-9, -1e-9, 2.9e-8, 3.1e-8, 1.0, -65504.0, 1e-40, 7.3])
    y = hip.round16_keep_nonzero(x.cuda(), HALF).float().cpu()
    want = x.to(HALF).float()
    tiny = (want == 0) & (x != 0)
    sub = 2.0 ** -24 if HALF == torch.float16 else 2.0 ** -133
    assert bool((y[~tiny] == want[~tiny]).all()) and bool((y[tiny].abs() == sub).all()) and bool((torch.sign(y[tiny]) == torch.sign(x[tiny])).all())
    big = torch.randn(100_003, generator=torch.Generator().manual_seed(1)) * 1e-7
    yb = hip.round16_keep_nonzero(big.cuda(), HALF).float().cpu()
    assert bool((yb != 0).all()) and float((yb - big).abs().max()) <= 6e-8
    import deephumor_amd.models as M
    from deephumor_amd.synth import synth_state_dict
    enc = M.ImageEncoder(256, spatial_features=True).eval()
    enc.load_state_dict(synth_state_dict(enc.state_dict(), seed=1234))
    calls = []
    real = hip.round16_keep_nonzero
    hip.round16_keep_nonzero = lambda t, dt: (calls.append(tuple(t.shape)), real(t, dt))[1]
    try:
        with torch.no_grad():
            e16 = enc.cuda().to(HALF)
            _, sp = e16(rnd(2, 3, 64, 64, seed=3).cuda())
    finally:
        hip.round16_keep_nonzero = real
    assert sp.dtype == HALF and tuple(sp.shape) == (2, 4, 256)
    assert calls == ([(8, 256)] if HALF == torch.float16 else [])       # the fp16 path rounds its spatial features through it


@pytest.mark.parametrize("m", [1280, 37, 640, 3000, 81])
def test_linear_ln_wreg_equals_tile_kernels(hip, m):
    """dh_linear_ln_wreg (weights stationary in registers, one round of <= 256 workgroups) against dh_linear_ln's tile kernels on
    the same operands: every form the decode chain uses (plain, deferred LayerNorm on the A rows + ReLU, residual +- its LayerNorm
    + output statistics; K = 512 and 2,048), bit for bit -- outputs AND statistics."""
    d, pf = 512, 2048
    g = torch.Generator().manual_seed(m)
    y = bf(torch.randn(m, d, generator=g) * 1.7 + 0.3).cuda()
    stats = _tile_stats(y.float().cpu()).cuda()
    gamma, beta = (torch.rand(d, generator=g) + 0.5).cuda(), (torch.randn(d, generator=g) * 0.2).cuda()
    for n in (3 * d, pf, 128, d, 192):
        w = bf(torch.randn(n, d, generator=g) / d ** 0.5).cuda()
        b, cs = (torch.randn(n, generator=g) * 0.1).cuda(), (torch.randn(n, generator=g)).cuda()
        wp = hip.pack_mfma_fragments(w)
        assert hip.linear_ln_wreg_supported(n, d, False)
        for relu in (False, True):
            assert torch.equal(hip.linear_ln_wreg(y, wp, n, b, relu=relu), hip.linear_ln(y, w, b, relu=relu))
            assert torch.equal(hip.linear_ln_wreg(y, wp, n, b, relu=relu, a_ln=(stats, 1e-5, cs)),
                               hip.linear_ln(y, w, b, relu=relu, a_ln=(stats, 1e-5, cs)))
    for k in (d, pf):
        a = bf(torch.randn(m, k, generator=g)).cuda()
        w = bf(torch.randn(d, k, generator=g) / k ** 0.5).cuda()
        b = (torch.randn(d, generator=g) * 0.1).cuda()
        wp = hip.pack_mfma_fragments(w)
        assert hip.linear_ln_wreg_supported(d, k, True)
        for r_ln in (None, (stats, 1e-5, gamma, beta)):
            got, gst = hip.linear_ln_wreg(a, wp, d, b, residual=y, r_ln=r_ln)
            want, wst = hip.linear_ln(a, w, b, residual=y, r_ln=r_ln, want_stats=True)
            assert torch.equal(got, want) and torch.equal(gst, wst)
    assert not hip.linear_ln_wreg_supported(d, 256, False) and not hip.linear_ln_wreg_supported(96, d, True)


@pytest.mark.parametrize("kind", ["CaptioningTransformer", "CaptioningTransformerBase"])
def test_decode_chain_wreg_equals_tile_chain(kind):
    """The decode chain on the register-stationary GEMMs against the same chain on the tile kernels (plan built with option
    decode_wreg_plan = 0): 64 images x beam 5 (320 rows: the wreg threshold) -- same tokens, same lengths, bit for bit."""
    import deephumor_amd.models as M
    from deephumor_amd.synth import synth_state_dict
    model = getattr(M, kind)(1000, hid_dim=512, n_layers=2).eval()
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=4321))
    model = model.to(HALF).cuda()
    imgs = synth_images(64, seed=11).cuda()
    with torch.no_grad():
        t1, l1 = model.generate_batch(imgs, max_len=10, beam_size=5, top_k=20, seed=5)
        assert "wo_pk" in model.decoder._get_plan()["layers"][0]
        from deephumor_amd import hip as H
        with H.option_scope(decode_wreg=0):               # (a changed option re-keys the plan: it is rebuilt at the next call)
            assert "wo_pk" not in model.decoder._get_plan()["layers"][0]
            t2, l2 = model.generate_batch(imgs, max_len=10, beam_size=5, top_k=20, seed=5)
        assert "wo_pk" in model.decoder._get_plan()["layers"][0]
    assert torch.equal(t1, t2) and torch.equal(l1, l2)


@pytest.mark.parametrize("kind", ["CaptioningLSTM", "CaptioningTransformer"])
def test_decode_with_vocab_wreg_equals_vocab_areg(kind):
    """Beam search with the register-streamed classifier (dh_vocab_logits_wreg: 64 images x beam 5 = 320 rows = 4 row blocks) against
    the same decode on dh_vocab_logits (plan built with option vocab_wreg_plan = 0): a vocabulary that ends inside a chunk -- same tokens, same
    lengths, bit for bit; a batch whose row count the kernel does not take (3 images) falls back and is split-invariant."""
    import deephumor_amd.models as M
    from deephumor_amd.synth import synth_state_dict
    kw = dict(hid_dim=512, n_layers=1) if "Transformer" in kind else dict(hidden_size=512)
    from deephumor_amd import hip as H
    model = getattr(M, kind)(3001, **kw).eval()
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=987))
    model = model.to(HALF).cuda()
    imgs = synth_images(64, seed=12).cuda()
    with torch.no_grad(), H.option_scope(vocab_wreg_transformer=1):   # (opt-in for the Transformer decoder, default for the LSTM decoder)
        t1, l1 = model.generate_batch(imgs, max_len=9, beam_size=5, top_k=20, seed=6)
        s1 = model.generate_batch(imgs[:3], max_len=9, beam_size=5, top_k=20, seed=6)
        assert "cls_w_pk" in model.decoder._get_plan()
        with H.option_scope(vocab_wreg=0):
            assert "cls_w_pk" not in model.decoder._get_plan()
            t2, l2 = model.generate_batch(imgs, max_len=9, beam_size=5, top_k=20, seed=6)
    assert torch.equal(t1, t2) and torch.equal(l1, l2)
    assert torch.equal(s1[0], t1[:3]) and torch.equal(s1[1], l1[:3])


@pytest.mark.parametrize("n,hw,cin,cout,relu", [(16, 28, 512, 128, True), (64, 14, 1024, 256, True), (11, 28, 512, 256, False), (43, 14, 1024, 512, True),
                                                (256, 7, 512, 2048, True), (3, 56, 1024, 128, True), (5, 56, 256, 128, True), (9, 31, 256, 256, False)])
def test_conv1x1_wreg_equals_tile_gemm(hip, n, hw, cin, cout, relu):
    """dh_conv1x1_wreg_nhwc (weights stationary in registers, pixels streamed through two 64 KB LDS buffers, persistent workgroups)
    against the implicit-GEMM tile kernel on the same operands, bit for bit -- pixel counts that are not a multiple of the block (a
    partial last block), 1 to 16 column blocks, with and without ReLU."""
    g = torch.Generator().manual_seed(n * 7 + cout)
    x = bf(torch.randn(n, hw, hw, cin, generator=g)).cuda()
    w = bf(torch.randn(cout, 1, 1, cin, generator=g) / cin ** 0.5).cuda()
    scale, shift = (torch.rand(cout, generator=g) + 0.5).cuda(), (torch.randn(cout, generator=g) * 0.3).cuda()
    assert hip.conv1x1_wreg_supported(n * hw * hw, cin, cout)
    want = hip.conv2d_nhwc_bn_act(x, w, scale, shift, relu=relu)
    got = hip.conv1x1_wreg_nhwc(x, hip.pack_mfma_fragments(w.view(cout, cin)), cout, scale, shift, relu=relu)
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
    assert not hip.conv1x1_wreg_supported(4096, cin, cout) and not hip.conv1x1_wreg_supported(n * hw * hw, 128, cout)
    if cin == 512:                                       # + residual before the ReLU (conv3 of the stage-4 bottlenecks)
        res = bf(torch.randn(n, hw, hw, cout, generator=g)).cuda()
        want = hip.conv2d_nhwc_bn_act(x, w, scale, shift, residual=res, relu=relu)
        got = hip.conv1x1_wreg_nhwc(x, hip.pack_mfma_fragments(w.view(cout, cin)), cout, scale, shift, relu=relu, residual=res)
        assert torch.equal(got, want), float((got.float() - want.float()).abs().max())


@pytest.mark.parametrize("n,ho,c1,c2,cout,stride", [(3, 56, 64, 64, 256, 1), (13, 28, 128, 256, 512, 2), (5, 57, 64, 64, 512, 1),
                                                    (21, 27, 128, 256, 256, 2), (12, 28, 64, 320, 1024, 2), (47, 14, 256, 512, 1024, 2), (50, 13, 256, 512, 256, 2), (2, 83, 64, 64, 256, 1)])
def test_conv1x1_dual_wreg_equals_tile_gemm(hip, n, ho, c1, c2, cout, stride):
    """dh_conv1x1_dual_wreg_nhwc (conv3 + strided downsample of a stage's first bottleneck as one streamed GEMM over [y | x at the strided
    pixels], weights stationary in registers) against dh_conv1x1_dual_nhwc on the same operands, bit for bit: both strides, odd input
    sizes (the strided pixel map), partial last blocks, 1 to 4 column blocks."""
    g = torch.Generator().manual_seed(n * 11 + cout)
    h = (ho - 1) * stride + 1 + (n % 2)                  # an input grid the strided walk does not end on, every other case
    y = bf(torch.randn(n, ho, ho, c1, generator=g)).cuda()
    x = bf(torch.randn(n, h, h, c2, generator=g)).cuda()
    w = bf(torch.randn(cout, c1 + c2, generator=g) / (c1 + c2) ** 0.5).cuda()
    shift = (torch.randn(cout, generator=g) * 0.3).cuda()
    assert hip.conv1x1_dual_wreg_supported(y.shape, x.shape, cout)
    want = hip.conv1x1_dual_nhwc(y, x, w, shift, stride, relu=True)
    got = hip.conv1x1_dual_wreg_nhwc(y, x, hip.pack_mfma_fragments(w), cout, shift, stride, relu=True)
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
    assert not hip.conv1x1_dual_wreg_supported((n, ho, ho, 512), (n, h, h, 512), cout) and not hip.conv1x1_dual_wreg_supported((1, 64, 64, c1), (1, 64, 64, c2), cout)


@pytest.mark.parametrize("n", [1, 2, 7, 64])
def test_conv3x3_s4_equals_tile_gemm(hip, n):
    """dh_conv3x3_s4_nhwc (stage-4 conv2: two images x half of the channels per workgroup, halo-free patch in LDS, weights from L2 into
    registers) against the implicit-GEMM tile kernel on the same operands, bit for bit -- odd image counts (a one-image last pair),
    image counts that leave XCD slots without a pair."""
    g = torch.Generator().manual_seed(40 + n)
    x = bf(torch.randn(n, 7, 7, 512, generator=g)).cuda()
    w = bf(torch.randn(512, 3, 3, 512, generator=g) / 4608 ** 0.5).cuda()
    scale, shift = (torch.rand(512, generator=g) + 0.5).cuda(), (torch.randn(512, generator=g) * 0.3).cuda()
    assert hip.conv3x3_s4_supported(7, 7, 512) and not hip.conv3x3_s4_supported(14, 14, 512) and not hip.conv3x3_s4_supported(7, 7, 256)
    want = hip.conv2d_nhwc_bn_act(x, w, scale, shift, relu=True, stride=1, pad=1)
    got = hip.conv3x3_s4_nhwc(x, hip.pack_mfma_fragments(w), scale, shift)
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())


@pytest.mark.parametrize("n", [1, 3, 32])
def test_bottleneck_tail_s2_equals_fused_tail(hip, n):
    """dh_bottleneck_tail_s2_nhwc (stage-2 tail: 4-row strips, three workgroups per CU, weights from L2 into registers, no barrier)
    against dh_bottleneck_tail_nhwc and against the two implicit GEMMs, bit for bit."""
    c, hw = 128, 28
    g = torch.Generator().manual_seed(70 + n)
    y1 = bf(torch.randn(n, hw, hw, c, generator=g)).cuda()
    w2 = bf(torch.randn(c, 3, 3, c, generator=g) / (9 * c) ** 0.5).cuda()
    s2, h2 = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.3).cuda()
    w3 = bf(torch.randn(4 * c, 1, 1, c, generator=g) / c ** 0.5).cuda()
    s3, h3 = (torch.rand(4 * c, generator=g) + 0.5).cuda(), (torch.randn(4 * c, generator=g) * 0.3).cuda()
    res = bf(torch.randn(n, hw, hw, 4 * c, generator=g)).cuda()
    assert hip.bottleneck_tail_s2_supported(hw, hw, c) and not hip.bottleneck_tail_s2_supported(56, 56, 64)
    want = hip.bottleneck_tail_nhwc(y1, w2, s2, h2, w3, s3, h3, res)
    y2 = hip.conv2d_nhwc_bn_act(y1, w2, s2, h2, None, relu=True, stride=1, pad=1)
    want2 = hip.conv2d_nhwc_bn_act(y2, w3, s3, h3, res, relu=True, stride=1, pad=0)
    got = hip.bottleneck_tail_s2_nhwc(y1, hip.pack_mfma_fragments(w2), s2, h2, hip.pack_mfma_fragments(w3.view(4 * c, c)), s3, h3, res)
    assert torch.equal(want, want2)
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
    # + the NEXT bottleneck's conv1 (512 -> 128) on the output chunks while they are in LDS: both outputs against the two launches
    w1 = bf(torch.randn(128, 1, 1, 4 * c, generator=g) / (4 * c) ** 0.5).cuda()
    s1, h1 = (torch.rand(128, generator=g) + 0.5).cuda(), (torch.randn(128, generator=g) * 0.3).cuda()
    want1 = hip.conv2d_nhwc_bn_act(want, w1, s1, h1, relu=True)
    gotf, got1 = hip.bottleneck_tail_s2_nhwc(y1, hip.pack_mfma_fragments(w2), s2, h2, hip.pack_mfma_fragments(w3.view(4 * c, c)), s3, h3, res,
                                             hip.pack_mfma_fragments(w1.view(128, 4 * c)), s1, h1, 128)
    assert torch.equal(gotf, want) and torch.equal(got1, want1), float((got1.float() - want1.float()).abs().max())


@pytest.mark.parametrize("n,fuse", [(1, False), (2, True), (5, True), (16, False)])
def test_bottleneck_tail_s1_equals_fused_tail_and_next_conv1(hip, n, fuse):
    """dh_bottleneck_tail_s1_nhwc (stage-1 tail on 4-row strips, weights from L2 into registers) against dh_bottleneck_tail_nhwc, and
    its fused form -- the NEXT bottleneck's conv1 + bn1 + relu on the output tile while it is in LDS -- against the stand-alone 1x1
    launch on the stored output: bit for bit."""
    c, hw, n1 = 64, 56, 64
    g = torch.Generator().manual_seed(90 + n)
    y1 = bf(torch.randn(n, hw, hw, c, generator=g)).cuda()
    w2 = bf(torch.randn(c, 3, 3, c, generator=g) / (9 * c) ** 0.5).cuda()
    s2, h2 = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.3).cuda()
    w3 = bf(torch.randn(4 * c, 1, 1, c, generator=g) / c ** 0.5).cuda()
    s3, h3 = (torch.rand(4 * c, generator=g) + 0.5).cuda(), (torch.randn(4 * c, generator=g) * 0.3).cuda()
    res = bf(torch.randn(n, hw, hw, 4 * c, generator=g)).cuda()
    w1 = bf(torch.randn(n1, 1, 1, 4 * c, generator=g) / (4 * c) ** 0.5).cuda()
    s1, h1 = (torch.rand(n1, generator=g) + 0.5).cuda(), (torch.randn(n1, generator=g) * 0.3).cuda()
    assert hip.bottleneck_tail_s1_supported(hw, hw, c, n1) and not hip.bottleneck_tail_s1_supported(28, 28, 128) and not hip.bottleneck_tail_s1_supported(hw, hw, c, 128)
    want = hip.bottleneck_tail_nhwc(y1, w2, s2, h2, w3, s3, h3, res)
    w2p, w3p = hip.pack_mfma_fragments(w2), hip.pack_mfma_fragments(w3.view(4 * c, c))
    if not fuse:
        got = hip.bottleneck_tail_s1_nhwc(y1, w2p, s2, h2, w3p, s3, h3, res)
        assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
        return
    want1 = hip.conv2d_nhwc_bn_act(want, w1, s1, h1, relu=True)
    got, got1 = hip.bottleneck_tail_s1_nhwc(y1, w2p, s2, h2, w3p, s3, h3, res, hip.pack_mfma_fragments(w1.view(n1, 4 * c)), s1, h1, n1)
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
    assert torch.equal(got1, want1), float((got1.float() - want1.float()).abs().max())


@pytest.mark.parametrize("n", [3, 16])
def test_conv1x1_dual_wreg_with_next_conv1(hip, n):
    """The stage-1 dual launch (conv3 + downsample of layer1.0) with layer1.1's conv1 + bn1 + relu behind it on the outputs while they
    are in LDS: both outputs against the two separate launches, bit for bit (a partial last block included)."""
    g = torch.Generator().manual_seed(300 + n)
    ho = 56
    y = bf(torch.randn(n, ho, ho, 64, generator=g)).cuda()
    x = bf(torch.randn(n, ho, ho, 64, generator=g)).cuda()
    w = bf(torch.randn(256, 128, generator=g) / 128 ** 0.5).cuda()
    shift = (torch.randn(256, generator=g) * 0.3).cuda()
    w1 = bf(torch.randn(64, 1, 1, 256, generator=g) / 16).cuda()
    s1, h1 = (torch.rand(64, generator=g) + 0.5).cuda(), (torch.randn(64, generator=g) * 0.3).cuda()
    want = hip.conv1x1_dual_nhwc(y, x, w, shift, 1, relu=True)
    want1 = hip.conv2d_nhwc_bn_act(want, w1, s1, h1, relu=True)
    got, got1 = hip.conv1x1_dual_wreg_nhwc(y, x, hip.pack_mfma_fragments(w), 256, shift, 1, relu=True, w1p=hip.pack_mfma_fragments(w1.view(64, 256)),
                                           scale1=s1, shift1=h1, n1=64)
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
    assert torch.equal(got1, want1), float((got1.float() - want1.float()).abs().max())


@pytest.mark.parametrize("n", [4, 48])
def test_encoder_round4_kernels_equal_the_kernels_they_replace(n):
    """The whole 16-bit encoder with the round-4 kernels (streaming 1x1 / dual layers, the register-streamed stage-1 / 2 / 4 kernels, the
    next bottleneck's conv1 fused behind the stage-1 dual launch and the stage-1 tail) against the same encoder with every one of them
    switched off: embeddings and spatial features bit for bit, at a batch where only some of them engage (4 images) and at one where
    all do (48)."""
    import deephumor_amd.models as M
    from deephumor_amd.synth import synth_state_dict
    enc = M.ImageEncoder(256, spatial_features=True).eval()
    enc.load_state_dict(synth_state_dict(enc.state_dict(), seed=4242))
    enc = enc.cuda().to(HALF)
    x = synth_images(n, seed=77).cuda()
    with torch.no_grad():
        emb, sp = enc(x)
        from deephumor_amd import hip as H
        with H.option_scope(encoder_generic=1):       # round 3's kernel set: tile GEMM 1x1 layers, ring tails, no conv1 fusion
            emb0, sp0 = enc(x)
        with H.option_scope(encoder_generic=2):       # every bottleneck convolution through the implicit-GEMM tile kernel
            emb1, sp1 = enc(x)
    assert torch.equal(emb, emb0) and torch.equal(sp, sp0)
    assert torch.equal(emb, emb1) and torch.equal(sp, sp1)


@pytest.mark.parametrize("n", [385, 520])
def test_encoder_batches_above_the_chunk_limit_equal_their_parts(n):
    """ImageEncoder on more than TRUNK_MAX_IMAGES = 384 images runs the trunk in chunks of 256 (encoders.py: the 16-bit kernels are
    tuned for <= 256 images per launch): embeddings and spatial features equal those of the same images encoded as two separate
    batches cut at another place, bit for bit -- an image's arithmetic does not depend on its batch."""
    import deephumor_amd.models as M
    from deephumor_amd.synth import synth_state_dict
    enc = M.ImageEncoder(256, spatial_features=True).eval()
    enc.load_state_dict(synth_state_dict(enc.state_dict(), seed=4242))
    enc = enc.cuda().to(HALF)
    x = synth_images(n, seed=5).cuda()
    cut = 200
    with torch.no_grad():
        emb, sp = enc(x)
        ea, sa = enc(x[:cut])
        eb, sb = enc(x[cut:])
    assert tuple(emb.shape)[0] == n
    assert torch.equal(emb, torch.cat([ea, eb])) and torch.equal(sp, torch.cat([sa, sb]))
