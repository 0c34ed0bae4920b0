"""Per-kernel parity on a real MI355X: every C-ABI entry point against a plain fp32 torch-CPU
statement of the same operation (floating-point kernels -> torch fp32 reference, tolerance in
each test), beam kernels against the oracle's BeamBook / the reference's golden vectors."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from helpers import golden  # noqa: E402


@pytest.fixture(scope="module")
def hip():
    from deephumor_amd import hip as h
    h.load()
    assert torch.cuda.is_available()
    return h


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


def close(a, b, atol, rtol=1e-4):
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), atol=atol, rtol=rtol)


@pytest.mark.parametrize("m,n,k", [(4, 1000, 512), (1280, 512, 512), (300, 36541 // 8, 768), (77, 130, 2048),
                                   (256, 2048, 512), (1, 64, 4), (129, 129, 36)])
def test_linear(hip, m, n, k):
    a, w, b = rnd(m, k, seed=1), rnd(n, k, seed=2) / k ** 0.5, rnd(n, seed=3)
    out = hip.linear(a.cuda(), w.cuda(), b.cuda())
    close(out, F.linear(a, w, b), atol=2e-5 * max(1.0, k ** 0.5 / 8))
    sc, sh = rnd(n, seed=4).abs() + 0.5, rnd(n, seed=5)
    out = hip.linear(a.cuda(), w.cuda(), b.cuda(), scale=sc.cuda(), shift=sh.cuda(), relu=True)
    close(out, torch.relu(F.linear(a, w, b) * sc + sh), atol=5e-5 * max(1.0, k ** 0.5 / 8))


def test_linear_strided(hip):
    """lda > K (LSTM [x|h] operand slices) and ldc > N (writing into a wider buffer)."""
    a_full, w = rnd(50, 96, seed=7), rnd(40, 64, seed=8)
    a = a_full.cuda()[:, 32:96]
    out_full = torch.zeros(50, 100, device="cuda")
    hip.linear(a, w.cuda(), None, out=out_full[:, 20:60])
    close(out_full[:, 20:60], F.linear(a_full[:, 32:96], w), atol=5e-5)
    assert float(out_full[:, :20].abs().max()) == 0 and float(out_full[:, 60:].abs().max()) == 0


@pytest.mark.parametrize("cin,cout,hw,ks,stride,pad,n", [
    (3, 64, 32, 7, 2, 3, 3), (64, 64, 14, 1, 1, 0, 2), (64, 256, 14, 1, 1, 0, 2), (256, 128, 14, 1, 2, 0, 2),
    (128, 128, 14, 3, 2, 1, 3), (64, 64, 9, 3, 1, 1, 2), (512, 2048, 7, 1, 1, 0, 3), (512, 512, 7, 3, 1, 1, 3),
    (1024, 2048, 14, 1, 2, 0, 2)])
def test_conv_bn_act(hip, cin, cout, hw, ks, stride, pad, n):
    x, w = rnd(n, cin, hw, hw, seed=1), rnd(cout, cin, ks, ks, seed=2) * (2.0 / (cin * ks * ks)) ** 0.5
    sc, sh = rnd(cout, seed=3).abs() + 0.5, rnd(cout, seed=4)
    ref = F.conv2d(x, w, stride=stride, padding=pad) * sc[None, :, None, None] + sh[None, :, None, None]
    out = hip.conv2d_bn_act(x.cuda(), w.cuda(), sc.cuda(), sh.cuda(), relu=False, stride=stride, pad=pad)
    close(out, ref, atol=1e-4)
    res = rnd(*ref.shape, seed=5)
    out = hip.conv2d_bn_act(x.cuda(), w.cuda(), sc.cuda(), sh.cuda(), residual=res.cuda(), relu=True,
                            stride=stride, pad=pad)
    close(out, torch.relu(ref + res), atol=1e-4)


def test_pools_and_layout(hip):
    x = rnd(3, 64, 30, 30)
    close(hip.maxpool3x3s2(x.cuda()), F.max_pool2d(x, 3, 2, 1), atol=0)
    x = rnd(2, 16, 15, 15)
    close(hip.maxpool3x3s2(x.cuda()), F.max_pool2d(x, 3, 2, 1), atol=0)
    f = rnd(5, 2048, 7, 7)
    close(hip.avgpool_rows(f.cuda()), f.mean(dim=(2, 3)), atol=1e-6)
    close(hip.nchw_to_rows(f.cuda()), f.reshape(5, 2048, 49).transpose(2, 1), atol=0)
    emb = rnd(100, 256)
    labels = torch.randint(0, 100, (7, 3))
    out = torch.zeros(7, 512, device="cuda")
    hip.label_mean(emb.cuda(), labels.cuda(), out[:, 256:])
    close(out[:, 256:], emb[labels].mean(dim=1), atol=1e-6)


@pytest.mark.parametrize("d", [512, 256, 1024])
def test_add_layernorm_embed_mask(hip, d):
    x, y, g, b = rnd(37, d, seed=1), rnd(37, d, seed=2), rnd(d, seed=3), rnd(d, seed=4)
    close(hip.add_layernorm(x.cuda(), y.cuda(), g.cuda(), b.cuda()), F.layer_norm(x + y, (d,), g, b, 1e-5), atol=5e-6)
    xin = x.cuda()
    hip.add_layernorm(xin, y.cuda(), g.cuda(), b.cuda(), out=xin)            # in place
    close(xin, F.layer_norm(x + y, (d,), g, b, 1e-5), atol=5e-6)
    # embedding rows: 3 images x 2 beams, logical rows = compact rows (row_mult 1)
    tok, pos, start = rnd(50, d, seed=5), rnd(20, d, seed=6), rnd(3, d, seed=7)
    tokens = torch.randint(0, 50, (6, 8), dtype=torch.int32)
    out = torch.empty(6, d, device="cuda")
    scale = float(d) ** 0.5
    hip.embed_rows(tok.cuda(), pos.cuda(), start.cuda(), tokens.cuda(), out, 6, 2, 1, 4, scale)
    close(out, tok[tokens[:, 3].long()] / scale + pos[4], atol=1e-6)
    hip.embed_rows(tok.cuda(), pos.cuda(), start.cuda(), tokens.cuda(), out, 6, 2, 1, 0, scale)
    close(out, start.repeat_interleave(2, 0) / scale + pos[0], atol=1e-6)
    out3 = torch.empty(3, d, device="cuda")                                   # one row per image, row_mult = beam
    hip.embed_rows(tok.cuda(), pos.cuda(), start.cuda(), tokens.cuda(), out3, 3, 1, 2, 2, scale)
    close(out3, tok[tokens[::2, 1].long()] / scale + pos[2], atol=1e-6)
    e = rnd(10, d, seed=8)
    e[3, 5] = 0.0
    e[7] = 0.0
    assert hip.enc_key_mask(e.cuda()).cpu().tolist() == [0, 0, 0, 1, 0, 0, 0, 1, 0, 0]


def _attn_ref(q, keys, vals, masked, scale):
    """q [H,dh]; keys/vals [L,H,dh]; masked [L] bool -> [H*dh] following transformers.py:106-120."""
    energy = torch.einsum("hd,lhd->hl", q, keys) / scale
    energy = energy.masked_fill(masked[None, :], -1e8)
    return torch.einsum("hl,lhd->hd", torch.softmax(energy, -1), vals).reshape(-1)


@pytest.mark.parametrize("t", [0, 1, 5, 15, 16, 20, 23, 24, 30, 31, 32, 39, 40, 100])
def test_attn_self_decode(hip, t):
    n_img, beam, d, h, tmax = 3, 4, 512, 8, 128
    r = n_img * beam
    dh = d // h
    qkv = rnd(r, 3 * d, seed=t)
    kc, vc = rnd(tmax + 1, r, d, seed=1), rnd(tmax + 1, r, d, seed=2)
    g = torch.Generator().manual_seed(t)
    src = (torch.arange(r)[:, None] // beam * beam + torch.randint(0, beam, (r, tmax + 1), generator=g)).int()
    tokens = torch.randint(0, 5, (r, tmax), generator=g, dtype=torch.int32)   # plenty of pad (0) keys
    out = torch.empty(r, d, device="cuda")
    kcd, vcd = kc.cuda(), vc.cuda()
    hip.attn_self_decode(qkv.cuda(), kcd, vcd, src.cuda(), tokens.cuda(), out, n_img, beam, 1, r, t, d, h, 8.0, 0)
    ref = torch.empty(r, d)
    for row in range(r):
        keys = torch.stack([kc[j, src[row, j]] for j in range(t)] + [qkv[row, d:2 * d]]).view(t + 1, h, dh)
        vals = torch.stack([vc[j, src[row, j]] for j in range(t)] + [qkv[row, 2 * d:]]).view(t + 1, h, dh)
        masked = torch.tensor([False] + [bool(tokens[row, j - 1] == 0) for j in range(1, t + 1)])
        ref[row] = _attn_ref(qkv[row, :d].view(h, dh), keys, vals, masked, 8.0)
    close(out, ref, atol=2e-5)
    close(kcd[t], qkv[:, d:2 * d], atol=0)       # this position appended at each row's own slot
    close(vcd[t], qkv[:, 2 * d:], atol=0)
    # one row per image (before the first draw): compact row i <-> logical row i*beam
    out1 = torch.empty(n_img, d, device="cuda")
    kcd, vcd = kc.cuda(), vc.cuda()
    hip.attn_self_decode(qkv[:n_img].cuda(), kcd, vcd, src.cuda(), tokens.cuda(), out1, n_img, 1, beam, r, t, d, h,
                         8.0, 0)
    for i in range(n_img):
        rl = i * beam
        keys = torch.stack([kc[j, src[rl, j]] for j in range(t)] + [qkv[i, d:2 * d]]).view(t + 1, h, dh)
        vals = torch.stack([vc[j, src[rl, j]] for j in range(t)] + [qkv[i, 2 * d:]]).view(t + 1, h, dh)
        masked = torch.tensor([False] + [bool(tokens[rl, j - 1] == 0) for j in range(1, t + 1)])
        close(out1[i], _attn_ref(qkv[i, :d].view(h, dh), keys, vals, masked, 8.0), atol=2e-5)
        close(kcd[t, rl], qkv[i, d:2 * d], atol=0)


@pytest.mark.parametrize("s", [49, 64, 100])
def test_attn_cross_decode(hip, s):
    n_img, beam, d, h = 3, 5, 512, 8
    r, dh = n_img * beam, 512 // 8
    q, kv = rnd(r, d, seed=1), rnd(n_img * s, 2 * d, seed=2)
    mask = torch.zeros(n_img * s, dtype=torch.uint8)
    mask[3] = 1
    mask[s:2 * s] = 1                      # image 1: every key masked -> uniform attention (softmax of equal -1e8)
    out = torch.empty(r, d, device="cuda")
    hip.attn_cross_decode(q.cuda(), kv.cuda(), mask.cuda(), out, n_img, beam, s, d, h, 8.0)
    for row in range(r):
        i = row // beam
        keys = kv[i * s:(i + 1) * s, :d].reshape(s, h, dh)
        vals = kv[i * s:(i + 1) * s, d:].reshape(s, h, dh)
        close(out[row], _attn_ref(q[row].view(h, dh), keys, vals, mask[i * s:(i + 1) * s].bool(), 8.0), atol=2e-5)


def test_lstm_step(hip):
    n_img, beam, e, hh, nl, v = 3, 2, 256, 512, 2, 40
    r = n_img * beam
    emb, img = rnd(v, e, seed=1), rnd(n_img, e, seed=2)
    h_prev, c_prev = rnd(nl, r, hh, seed=3), rnd(nl, r, hh, seed=4)
    tokens = torch.randint(0, v, (r, 6), dtype=torch.int32)
    hpar = torch.tensor([1, 0, 3, 3, 4, 5], dtype=torch.int32)
    xcat0, xcatl = torch.zeros(r, e + hh, device="cuda"), torch.zeros(nl - 1, r, 2 * hh, device="cuda")
    c_cur = torch.zeros(nl, r, hh, device="cuda")
    hip.lstm_prepare(emb.cuda(), None, tokens.cuda(), 2, hpar.cuda(), h_prev.cuda(), c_prev.cuda(), xcat0, xcatl,
                     c_cur, r, beam, 1, r, nl, e, hh)
    close(xcat0[:, :e], emb[tokens[:, 2].long()], atol=0)
    close(xcat0[:, e:], h_prev[0][hpar.long()], atol=0)
    close(xcatl[0][:, hh:], h_prev[1][hpar.long()], atol=0)
    close(c_cur, c_prev[:, hpar.long()], atol=0)
    # image step, zero state, one row per image stored at logical row img*beam
    x0 = torch.full((n_img, e + hh), 7.0, device="cuda")
    xl, cc = torch.full((1, n_img, 2 * hh), 7.0, device="cuda"), torch.full((nl, n_img, hh), 7.0, device="cuda")
    hip.lstm_prepare(None, img.cuda(), None, 0, None, None, None, x0, xl, cc, n_img, 1, beam, r, nl, e, hh)
    close(x0[:, :e], img, atol=0)
    assert float(x0[:, e:].abs().max()) == 0 and float(cc.abs().max()) == 0 and float(xl[0][:, hh:].abs().max()) == 0
    gates, c0 = rnd(n_img, 4 * hh, seed=5) * 2, rnd(n_img, hh, seed=6)
    h_new, c_new = torch.zeros(r, hh, device="cuda"), torch.zeros(r, hh, device="cuda")
    h_out = torch.zeros(n_img, 2 * hh, device="cuda")
    hip.lstm_cell(gates.cuda(), c0.cuda(), h_new, c_new, h_out, 2 * hh, n_img, beam, hh)
    gi, gf, gg, go = gates.chunk(4, 1)
    c1 = torch.sigmoid(gf) * c0 + torch.sigmoid(gi) * torch.tanh(gg)
    h1 = torch.sigmoid(go) * torch.tanh(c1)
    close(c_new[::beam], c1, atol=2e-6)
    close(h_new[::beam], h1, atol=2e-6)
    close(h_out[:, :hh], h1, atol=2e-6)
    assert float(h_new[1::beam].abs().max()) == 0


def _race(p, noise, k):
    return torch.topk(p / noise, k, dim=-1).indices


@pytest.mark.parametrize("v,beam,top_k,temp", [(1000, 3, 20, 1.3), (36541, 5, 50, 1.0), (71, 7, 50, 1.1), (500, 1, 1, 1.0)])
def test_beam_row_sample(hip, v, beam, top_k, temp):
    """beam.py:32-48,78-79 with caller-supplied Exp(1) noise == torch.multinomial's race."""
    from oracle.ref_path import BeamBook
    rows = 6
    logits = rnd(rows, v, seed=v) * 2.5
    if top_k > 2:
        logits[0, 1] = 50.0                              # <unk> is the arg-max of row 0: always dropped
        logits[2, 10:10 + top_k + 3] = 30.0              # ties straddling the threshold are all kept
    noise = torch.empty(rows, v).exponential_(1, generator=torch.Generator().manual_seed(3))
    book = BeamBook(temp, beam, top_k)
    filt = book.keep_top_k(logits.clone())
    picks = _race(torch.softmax(filt / temp, -1), noise, beam)
    vals = torch.gather(filt, 1, picks).log_softmax(-1)
    pi = torch.empty(rows, beam, dtype=torch.int32, device="cuda")
    pv = torch.empty(rows, beam, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    hip.beam_row_sample(logits.cuda(), v, rows, 3, beam, top_k, temp, 1, noise.cuda(), 0, 0, 0, pi, pv, err)
    assert int(err.item()) == 0
    assert pi.cpu().tolist() == picks.tolist()
    close(pv, vals, atol=2e-6)
    # Philox mode: picks are survivors, distinct, deterministic and depend on the seed
    a, b = torch.empty_like(pi), torch.empty_like(pi)
    hip.beam_row_sample(logits.cuda(), v, rows, 3, beam, top_k, temp, 1, None, 11, 5, 2, a, pv, err)
    hip.beam_row_sample(logits.cuda(), v, rows, 3, beam, top_k, temp, 1, None, 11, 5, 2, b, pv, err)
    assert a.cpu().tolist() == b.cpu().tolist()
    for row in range(rows):
        ids = a[row].cpu().tolist()
        assert len(set(ids)) == beam and all(torch.isfinite(filt[row, i]) for i in ids)


def test_beam_row_sample_all_filtered(hip):
    """top_k=1 with <unk> as arg-max: the reference raises (beam.py:46); the kernel sets the error bit."""
    logits = rnd(2, 100)
    logits[1, 1] = 99.0
    pi = torch.empty(2, 1, dtype=torch.int32, device="cuda")
    pv = torch.empty(2, 1, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    hip.beam_row_sample(logits.cuda(), 100, 2, 1, 1, 1, 1.0, 1, None, 0, 0, 0, pi, pv, err)
    assert int(err.item()) & hip.ERR_ALL_FILTERED
    assert int(pi[0, 0]) == int(logits[0].argmax())


def test_beam_select_matches_reference_process_logits(hip):
    """G4 golden: BeamSearchHelper.process_logits recorded from the real reference (torch.manual_seed(7)),
    replayed through both kernels with the same Exp(1) noise."""
    g = golden("g4_beam_helper.npz")
    logits = torch.from_numpy(g["pl_logits"])
    b, v = 3, logits.shape[1]
    torch.manual_seed(7)
    noise_row = torch.empty(b, v).exponential_(1)        # what multinomial consumed inside process_logits
    pi = torch.empty(b, b, dtype=torch.int32, device="cuda")
    pv = torch.empty(b, b, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    hip.beam_row_sample(logits.cuda(), v, b, b, b, 5, 0.7, 1, noise_row.cuda(), 0, 0, 0, pi, pv, err)
    tokens = torch.zeros(b, 4, dtype=torch.int32)
    tokens[:, :2] = torch.from_numpy(g["pl_seqs"]).int()
    tokens, vals = tokens.cuda(), torch.from_numpy(g["pl_vals"]).cuda()
    ended = torch.tensor([0, 1, 0], dtype=torch.uint8).cuda()
    parent, hparent = torch.zeros(b, dtype=torch.int32).cuda(), torch.zeros(b, dtype=torch.int32).cuda()
    done, end_step = torch.zeros(1, dtype=torch.uint8).cuda(), torch.zeros(1, dtype=torch.int32).cuda()
    # candidate order of the reference: [b0 x3, b1 x1, b2 x3]; give candidates 4, 3, 0 the smallest noise
    n_cand = len(g["pl_new_ind"])
    cand_val = torch.from_numpy(g["pl_prev_vals"].reshape(-1) + g["pl_new_val"])
    noise_c = torch.ones(1, b * b)
    noise_c[0, :n_cand] = torch.tensor([1.0, 50.0, 60.0, 0.5, 0.1, 70.0, 80.0])
    keep = _race(torch.softmax(cand_val / 0.7, -1), noise_c[0, :n_cand], b)
    hip.beam_select(pi, pv, tokens, vals, ended, None, parent, hparent, done, end_step, 1, b, False, False, 2, 0, 2,
                    0.7, 3, noise_c.cuda(), 0, 0)
    exp_seq = torch.cat([torch.from_numpy(g["pl_prev_seqs"]), torch.from_numpy(g["pl_new_ind"])[:, None]], 1)[keep]
    assert tokens[:, :3].cpu().tolist() == exp_seq.tolist()
    close(vals, cand_val[keep], atol=2e-6)
    assert ended.cpu().bool().tolist() == torch.from_numpy(g["pl_has_ended"])[keep].tolist()
    cand_parent = torch.tensor([0, 0, 0, 1, 2, 2, 2])
    assert parent.cpu().tolist() == cand_parent[keep].tolist()
    assert hparent.cpu().tolist() == (keep // b).tolist()       # rnn_models.py:135-137 dense-layout quirk
    assert int(done.item()) == int(bool(torch.from_numpy(g["pl_has_ended"])[keep].all()))


# ------------------------------------------------------------------------------------------------
# bf16 throughput path: bf16 storage, bf16 MFMA, fp32 accumulate.  References are computed in fp32
# from the SAME bf16-rounded operands, so the only differences are accumulation order and the final
# rounding of the output to bf16 (<= 2^-8 relative).
# ------------------------------------------------------------------------------------------------
def bf(x):
    return x.to(torch.bfloat16)


@pytest.mark.parametrize("m,n,k", [(4, 1000, 512), (1280, 512, 512), (300, 4567, 768), (77, 130, 2048), (1, 64, 8),
                                   (1280, 2048, 512),     # 640 workgroups of 64 x 64: 3-slab ring, three per CU
                                   (65, 25000, 512),      # 782 workgroups: 2-slab ring, five per CU
                                   (1280, 3072, 520)])    # 128 x 128 tiles, K not a multiple of 64 (K tail, no pointer stepping)
def test_linear_bf16(hip, m, n, k):
    a, w, b = bf(rnd(m, k, seed=1)), bf(rnd(n, k, seed=2) / k ** 0.5), rnd(n, seed=3)
    ref = F.linear(a.float(), w.float(), b)
    out32 = hip.linear(a.cuda(), w.cuda(), b.cuda(), out_dtype=torch.float32)
    assert out32.dtype == torch.float32
    close(out32, ref, atol=1e-4 * max(1.0, k ** 0.5 / 8), rtol=1e-4)
    out = hip.linear(a.cuda(), w.cuda(), b.cuda())
    assert out.dtype == torch.bfloat16
    close(out.float(), ref, atol=2e-2, rtol=1e-2)
    sc, sh, res = rnd(n, seed=4).abs() + 0.5, rnd(n, seed=5), bf(rnd(m, n, seed=6))
    out = hip.linear(a.cuda(), w.cuda(), b.cuda(), scale=sc.cuda(), shift=sh.cuda(), residual=res.cuda(), relu=True)
    close(out.float(), torch.relu(ref * sc + sh + res.float()), atol=3e-2, rtol=1e-2)


def test_linear_f32_residual(hip):
    a, w, b, res = rnd(70, 64, seed=1), rnd(50, 64, seed=2), rnd(50, seed=3), rnd(70, 50, seed=4)
    out = hip.linear(a.cuda(), w.cuda(), b.cuda(), residual=res.cuda())
    close(out, F.linear(a, w, b) + res, atol=5e-5)


@pytest.mark.parametrize("cin,cout,hw,ks,stride,pad,n", [
    (64, 64, 14, 1, 1, 0, 2), (64, 256, 14, 1, 1, 0, 2), (256, 128, 14, 1, 2, 0, 2), (128, 128, 14, 3, 2, 1, 3),
    (64, 64, 9, 3, 1, 1, 2), (512, 2048, 7, 1, 1, 0, 3), (512, 512, 7, 3, 1, 1, 3), (1024, 2048, 14, 1, 2, 0, 2),
    (64, 64, 56, 3, 1, 1, 5),
    # >= 131,072 output pixels with 64 output channels: the 256 x 64 (3x3) and 128 x 64 (dense 1x1) tile configurations
    (64, 64, 56, 3, 1, 1, 43), (256, 64, 56, 1, 1, 0, 43), (64, 256, 56, 1, 1, 0, 11)])
def test_conv_nhwc_bf16(hip, cin, cout, hw, ks, stride, pad, n):
    x, w = bf(rnd(n, cin, hw, hw, seed=1)), bf(rnd(cout, cin, ks, ks, seed=2) * (2.0 / (cin * ks * ks)) ** 0.5)
    sc, sh = rnd(cout, seed=3).abs() + 0.5, rnd(cout, seed=4)
    ref = F.conv2d(x.float(), w.float(), stride=stride, padding=pad) * sc[None, :, None, None] + sh[None, :, None, None]
    res = bf(rnd(*ref.shape, seed=5))
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().cuda()
    w_k = w.permute(0, 2, 3, 1).contiguous().cuda()                 # [Cout, KS, KS, Cin]
    out = hip.conv2d_nhwc_bn_act(x_nhwc, w_k, sc.cuda(), sh.cuda(), relu=False, stride=stride, pad=pad)
    close(out.float().permute(0, 3, 1, 2), ref, atol=3e-2, rtol=1e-2)
    out = hip.conv2d_nhwc_bn_act(x_nhwc, w_k, sc.cuda(), sh.cuda(), residual=res.permute(0, 2, 3, 1).contiguous().cuda(),
                                 relu=True, stride=stride, pad=pad)
    close(out.float().permute(0, 3, 1, 2), torch.relu(ref + res.float()), atol=3e-2, rtol=1e-2)


@pytest.mark.parametrize("d,h", [(128, 8), (512, 4), (256, 8)])
def test_attn_other_head_dims(hip, d, h):
    """head_dim 16 -> generic kernel; 128 and 32 -> the (key slot, chunk) kernel."""
    n_img, beam, t, tmax = 2, 3, 9, 16
    r, dh = n_img * beam, d // h
    qkv = rnd(r, 3 * d, seed=d)
    kc, vc = rnd(tmax + 1, r, d, seed=1), rnd(tmax + 1, r, d, seed=2)
    src = (torch.arange(r)[:, None] // beam * beam).expand(r, tmax + 1).contiguous().int()
    tokens = torch.randint(0, 3, (r, tmax), dtype=torch.int32)
    out = torch.empty(r, d, device="cuda")
    hip.attn_self_decode(qkv.cuda(), kc.cuda(), vc.cuda(), src.cuda(), tokens.cuda(), out, n_img, beam, 1, r, t, d, h,
                         float(dh) ** 0.5, 0)
    for row in range(r):
        keys = torch.stack([kc[j, src[row, j]] for j in range(t)] + [qkv[row, d:2 * d]]).view(t + 1, h, dh)
        vals = torch.stack([vc[j, src[row, j]] for j in range(t)] + [qkv[row, 2 * d:]]).view(t + 1, h, dh)
        masked = torch.tensor([False] + [bool(tokens[row, j - 1] == 0) for j in range(1, t + 1)])
        close(out[row], _attn_ref(qkv[row, :d].view(h, dh), keys, vals, masked, float(dh) ** 0.5), atol=2e-5)


def test_normalize_u8_matches_totensor_normalize():
    """dh_normalize_u8_hwc == ToTensor().div(255) followed by Normalize's sub_().div_(), bit for bit."""
    from deephumor_amd.experiments.inference import images_to_tensor, IMAGENET_MEAN, IMAGENET_STD
    g = torch.Generator().manual_seed(3)
    u8 = torch.randint(0, 256, (3, 37, 53, 3), generator=g, dtype=torch.uint8)
    want = u8.permute(0, 3, 1, 2).float().div(255)
    want = want.sub(torch.tensor(IMAGENET_MEAN)[None, :, None, None]).div(torch.tensor(IMAGENET_STD)[None, :, None, None])
    got = images_to_tensor(u8.cuda()).cpu()
    assert torch.equal(got, want)


@pytest.mark.parametrize("c1,c2,cout,hw,stride,n", [(64, 64, 256, 14, 1, 3), (128, 256, 512, 14, 2, 2), (256, 512, 1024, 8, 2, 2),
                                                   (64, 64, 256, 56, 1, 9)])
def test_conv1x1_dual_matches_conv3_plus_downsample(hip, c1, c2, cout, hw, stride, n):
    """dh_conv1x1_dual_nhwc == relu(bn3(conv3(y)) + bn_d(downsample(x))) in fp32 on the same bf16-rounded tensors (the
    BatchNorm scales are folded into the bf16 weights of the fused form, hence the bf16-level tolerance)."""
    ho = (hw - 1) // stride + 1
    y, x = bf(rnd(n, c1, ho, ho, seed=1)), bf(rnd(n, c2, hw, hw, seed=2))
    w3, wd = bf(rnd(cout, c1, 1, 1, seed=3) * (2.0 / c1) ** 0.5), bf(rnd(cout, c2, 1, 1, seed=4) * (2.0 / c2) ** 0.5)
    s3, sd = rnd(cout, seed=5).abs() * 0.3 + 0.2, rnd(cout, seed=6).abs() * 0.3 + 0.5
    b3, bd = rnd(cout, seed=7), rnd(cout, seed=8)
    ref = torch.relu(F.conv2d(y.float(), w3.float()) * s3[None, :, None, None] + b3[None, :, None, None]
                     + F.conv2d(x.float(), wd.float(), stride=stride) * sd[None, :, None, None] + bd[None, :, None, None])
    w_cat = torch.cat([w3.float().view(cout, c1) * s3[:, None], wd.float().view(cout, c2) * sd[:, None]], 1).to(torch.bfloat16)
    out = hip.conv1x1_dual_nhwc(y.permute(0, 2, 3, 1).contiguous().cuda(), x.permute(0, 2, 3, 1).contiguous().cuda(),
                                w_cat.contiguous().cuda(), (b3 + bd).cuda(), stride)
    close(out.float().permute(0, 3, 1, 2), ref, atol=4e-2, rtol=2e-2)


def test_device_resize_is_bit_exact_vs_pillow_golden(hip):
    """dh_resize_u8_hwc against the G9 golden (real Pillow outputs) and the numpy oracle, incl. one-pass cases."""
    import hashlib
    from oracle.make_resize_golden import CASES, image
    from oracle.resize_ref import resize_bilinear_u8
    from deephumor_amd.experiments.inference import resize_images
    g = golden("g9_resize.npz")
    for name, (h, w, oh, ow) in CASES.items():
        img = image(name, h, w)
        batch = torch.from_numpy(np.stack([img, img[::-1].copy()])).cuda()           # two images per launch
        out = resize_images(batch, (oh, ow)).cpu().numpy()
        assert hashlib.sha256(out[0].tobytes()).digest() == g[f"{name}_sha"].tobytes(), name
        assert (out[1] == resize_bilinear_u8(img[::-1].copy(), oh, ow)).all(), name


def test_fused_normalize_pack_feeds_the_stem_identically():
    """preprocess_images(dtype=bf16) (resize -> normalise + pack fused, no fp32 tensor in front of conv1) gives the encoder
    exactly what the reference-shaped route (fp32 NCHW batch -> pack) gives."""
    import deephumor_amd.models as M
    from deephumor_amd.experiments.inference import preprocess_images
    from helpers import synthetic_sd
    g = torch.Generator().manual_seed(4)
    u8 = torch.randint(0, 256, (3, 300, 260, 3), generator=g, dtype=torch.uint8).cuda()
    x32 = preprocess_images(u8, dtype=torch.float32)
    assert tuple(x32.shape) == (3, 3, 224, 224) and x32.dtype == torch.float32
    sd, hp = synthetic_sd("CaptioningTransformer")
    model = M.CaptioningTransformer(**hp).eval()
    model.load_state_dict(sd)
    for dt in (torch.bfloat16, torch.float16):
        m16 = model.cuda().to(dt)
        packed = preprocess_images(u8, dtype=dt)
        assert tuple(packed.shape) == (3, 224, 224, 8) and packed.dtype == dt
        with torch.no_grad():
            e1, s1 = m16.encoder(x32)
            e2, s2 = m16.encoder(packed)
        assert torch.equal(e1, e2) and torch.equal(s1, s2)


# ------------------------------------------------------------------------------------------------
# BeamSearchHelper's METHOD surface (reference beam.py:32-108), driven the way the reference drives it
# ------------------------------------------------------------------------------------------------
def _replay_noise(kind, call, shape):
    """What torch.multinomial consumes from the CPU generator for a probability tensor of this shape."""
    assert kind == "multinomial"
    return torch.empty(shape).exponential_(1)


def test_helper_methods_replay_reference_golden(hip):
    """G4 (recorded from the REAL reference): filter_top_k unit vectors, then process_logits under torch.manual_seed(7) through
    the METHODS -- ids / masks bit-exact, values <= 1e-6."""
    from deephumor_amd.models import BeamSearchHelper
    g = golden("g4_beam_helper.npz")
    h = BeamSearchHelper(temperature=1.0, beam_size=3, top_k=4, device="cuda")
    x = torch.from_numpy(g["filter_in"]).cuda()
    out = h.filter_top_k(x)
    assert out.data_ptr() == x.data_ptr()                                   # in place, like the reference (beam.py:36)
    np.testing.assert_array_equal(x.cpu().numpy(), g["filter_out"])
    h = BeamSearchHelper(temperature=0.7, beam_size=3, top_k=5, device="cuda", noise_source=_replay_noise)
    h.has_ended = torch.tensor([False, True, False]).cuda()
    logits = torch.from_numpy(g["pl_logits"]).cuda()
    torch.manual_seed(7)
    (ps, pv), (ni, nv) = h.process_logits(logits, torch.from_numpy(g["pl_seqs"]).cuda(), torch.from_numpy(g["pl_vals"]).cuda())
    assert ps.dtype == torch.int64 and ni.dtype == torch.int64 and h.has_ended.dtype == torch.bool
    np.testing.assert_array_equal(ps.cpu().numpy(), g["pl_prev_seqs"])
    np.testing.assert_array_equal(ni.cpu().numpy(), g["pl_new_ind"])
    np.testing.assert_array_equal(h.has_ended.cpu().numpy(), g["pl_has_ended"])
    assert pv.shape == g["pl_prev_vals"].shape and nv.shape == g["pl_new_val"].shape
    close(pv, torch.from_numpy(g["pl_prev_vals"]), atol=1e-6, rtol=0)
    close(nv, torch.from_numpy(g["pl_new_val"]), atol=1e-6, rtol=0)
    assert bool(torch.isinf(logits).any())                                  # process_logits filtered its argument in place
    assert not h.all_ended()


def test_helper_sample_k_and_gather(hip):
    """sample_k_indices == torch.multinomial's race (2-D and the 1-D candidate form), golden mn_picks from the reference;
    filter_by_indices == torch.gather; torch's error cases raise."""
    from deephumor_amd.models import BeamSearchHelper
    g = golden("g4_beam_helper.npz")
    noise = torch.from_numpy(g["mn_noise"])
    h = BeamSearchHelper(temperature=1.0, beam_size=4, top_k=50, device="cuda", noise_source=lambda k, c, s: noise)
    # softmax(log p) == p up to rounding: the recorded picks come back
    picks = h.sample_k_indices(torch.from_numpy(g["mn_p"]).log().cuda())
    assert picks.dtype == torch.int64 and picks.cpu().tolist() == g["mn_picks"].tolist()
    x = rnd(5, 36541, seed=3) * 3
    x[0, 5:400] = float("-inf")
    nz = torch.empty(5, 36541).exponential_(1, generator=torch.Generator().manual_seed(9))
    h = BeamSearchHelper(temperature=1.3, beam_size=7, top_k=50, device="cuda", noise_source=lambda k, c, s: nz[:s[0]])
    want = torch.topk(torch.softmax(x / 1.3, -1) / nz, 7, dim=-1).indices
    got = h.sample_k_indices(x.cuda())
    assert got.cpu().tolist() == want.tolist()
    one = h.sample_k_indices(x[2].cuda(), k=3)                              # 1-D in -> 1-D out (rnn_models.py:120,140)
    assert one.shape == (3,) and one.cpu().tolist() == torch.topk(torch.softmax(x[2] / 1.3, -1) / nz[0], 3).indices.tolist()
    vals = BeamSearchHelper.filter_by_indices(x.cuda(), got)
    assert torch.equal(vals.cpu(), torch.gather(x, 1, want))
    # Philox mode: deterministic per (seed, draw), distinct, never a zero-probability entry
    h = BeamSearchHelper(temperature=1.0, beam_size=5, top_k=50, device="cuda", seed=11)
    a = h.sample_k_indices(x.cuda())
    h2 = BeamSearchHelper(temperature=1.0, beam_size=5, top_k=50, device="cuda", seed=11)
    assert torch.equal(a, h2.sample_k_indices(x.cuda())) and not torch.equal(a, h.sample_k_indices(x.cuda()))
    assert all(len(set(r)) == 5 for r in a.cpu().tolist()) and not bool(((a[0] >= 5) & (a[0] < 400)).any())
    with pytest.raises(RuntimeError):                                       # all -inf row: softmax is NaN, torch raises
        h.sample_k_indices(torch.full((2, 30), float("-inf")).cuda())
    few = torch.full((1, 30), float("-inf"))
    few[0, 4:7] = 1.0
    got = h.sample_k_indices(few.cuda()).cpu()[0].tolist()                  # 3 positive categories, 5 draws: no error (current torch)
    assert sorted(got[:3]) == [4, 5, 6] and len(set(got)) == 5 and int(h.err.item()) & hip.ERR_TOO_FEW


@pytest.mark.parametrize("v,beam,top_k,temp,seed", [(1000, 3, 20, 1.3, 100), (71, 7, 50, 1.1, 5), (1000, 5, 6, 1.0, 3)])
def test_helper_drives_reference_lstm_loop(hip, v, beam, top_k, temp, seed):
    """The reference's own LSTMDecoder.generate loop (rnn_models.py:80-141) written against the helper's METHODS -- logits from
    the oracle's LSTM on CPU, every helper call on the GPU -- reproduces the oracle's caption token for token under RNG replay
    (incl. ended beams: EOS is made likely, and the h/c re-indexing uses the helper's filter indices)."""
    import torch.nn.functional as Fn
    from deephumor_amd.models import BeamSearchHelper
    from helpers import synthetic_sd
    from oracle import ref_path as R
    sd, hp = synthetic_sd("CaptioningLSTM", v)
    sd = dict(sd)
    sd["decoder.classifier.bias"] = sd["decoder.classifier.bias"].clone()
    sd["decoder.classifier.bias"][3] += 4.0                                 # beams end at different steps
    p = "decoder"
    emb_w, cls_w, cls_b = sd[p + ".embedding.weight"], sd[p + ".classifier.weight"], sd[p + ".classifier.bias"]
    image_emb = rnd(1, 1, emb_w.shape[1], seed=seed)
    max_len = 10
    torch.manual_seed(seed)
    want = R.lstm_decoder_generate(sd, p, image_emb, max_len=max_len, temperature=temp, beam_size=beam, top_k=top_k)

    torch.manual_seed(seed)
    helper = BeamSearchHelper(temperature=temp, beam_size=beam, top_k=top_k, eos_index=3, device="cuda", noise_source=_replay_noise)
    out, (hh, cc) = R.lstm_run(sd, p + ".lstm", image_emb)
    logits = Fn.linear(out[:, -1], cls_w, cls_b).cuda()
    hh, cc = hh.repeat(1, beam, 1), cc.repeat(1, beam, 1)
    logits = helper.filter_top_k(logits)
    sample_ind = helper.sample_k_indices(logits, k=beam)
    sample_val = helper.filter_by_indices(logits, sample_ind).log_softmax(-1)
    sample_ind, sample_val = sample_ind.T, sample_val.T
    sample_seq = sample_ind.clone()
    helper.has_ended = (sample_ind == 3).view(-1)
    for _ in range(sample_seq.size(1), max_len):
        out, (hh, cc) = R.lstm_run(sd, p + ".lstm", emb_w[sample_ind.cpu()], (hh, cc))
        logits = Fn.linear(out[:, -1], cls_w, cls_b).cuda()
        (prev_seqs, prev_vals), (new_ind, new_val) = helper.process_logits(logits, sample_seq, sample_val)
        cand_seq = torch.cat((prev_seqs, new_ind.unsqueeze(0).T), -1)
        cand_val = prev_vals.flatten() + new_val
        filter_ind = helper.sample_k_indices(cand_val, k=beam)
        sample_val, sample_seq = cand_val[filter_ind], cand_seq[filter_ind]
        sample_ind = sample_seq[:, -1].unsqueeze(-1)
        helper.has_ended = helper.has_ended[filter_ind]
        if helper.all_ended():
            break
        keep = filter_ind.cpu()
        hh = torch.repeat_interleave(hh, beam, dim=1)[:, keep]
        cc = torch.repeat_interleave(cc, beam, dim=1)[:, keep]
    ind = helper.sample_k_indices(sample_val, k=1)
    got = sample_seq[ind, :].squeeze()
    assert got.cpu().tolist() == want.tolist()


@pytest.mark.parametrize("v", (61, 5000))
def test_row_samplers_treat_signed_zeros_as_ties(hip, v):
    """A row whose top-k threshold is a zero: ``logits < threshold`` (beam.py:34) keeps the zeros of BOTH signs (-0.0 == +0.0); the
    samplers' integer keys used to order -0.0 below +0.0 and dropped them (found by tools/fuzz_sampler.py)."""
    g = torch.Generator().manual_seed(v)
    rows, beam, top_k, temp = 4, 3, 10, 1.3
    x = -1.0 - torch.rand(rows, v, generator=g)                       # everything else below zero
    x[:, 2:10] = torch.arange(8, 0, -1).float()                       # eight values above
    zeros = torch.tensor([0.0, -0.0, 0.0, -0.0, -0.0, 0.0])
    cols = torch.tensor([11, 17, 23, 29, 40, 55])
    x[:, cols] = zeros                                                # the 9th..14th largest: top_k = 10 cuts through them (at a +0.0 in key order)
    noise = torch.empty(rows, v).exponential_(1, generator=g)
    noise[:, cols[1]] = 1e-6                                          # a -0.0 tie must win the race
    kept = x.clone()
    kept[x < x.topk(top_k, dim=-1).values[:, -1:]] = float("-inf")
    kept[:, 1] = float("-inf")
    assert int(torch.isfinite(kept).sum(-1).min()) == 14             # 8 + all six zeros
    want = torch.topk(torch.softmax(kept / temp, -1) / noise, beam, dim=-1).indices
    assert bool((want[:, 0] == cols[1]).all())
    ng = hip.n_groups(v)
    pad = torch.full((rows, ng * 64), float("-inf"))
    pad[:, :v] = x
    gmax = pad.view(rows, ng, 64).max(-1).values.cuda()
    for kind in ("full", "groups"):
        if kind == "groups" and top_k > ng:
            continue
        pi = torch.empty((rows, beam), dtype=torch.int32, device="cuda")
        pv = torch.empty((rows, beam), device="cuda")
        err = torch.zeros(1, dtype=torch.int32, device="cuda")
        if kind == "full":
            hip.beam_row_sample(x.cuda(), v, rows, beam, beam, top_k, temp, 1, noise.cuda(), 0, 0, 0, pi, pv, err)
        else:
            hip.beam_row_sample_groups(x.cuda(), v, gmax, rows, beam, beam, top_k, temp, 1, noise.cuda(), 0, 0, 0, pi, pv, err)
        assert int(err.item()) == 0 and pi.cpu().long().tolist() == want.tolist(), kind


@pytest.mark.parametrize("v,top_k", [(5000, 50), (36541, 20), (1500, 300)])
def test_row_samplers_on_flat_logits(hip, v, top_k):
    """Constant logits: EVERY token ties at the top-k threshold and survives (beam.py:34 keeps ties), far more than the pre-filtered
    samplers' 1,024-entry candidate buffers: those flag ERR_OVERFLOW, and dh_beam_row_sample_exact (what the models then repeat the
    batch with) draws over the row itself -- the `beam` largest 1 / Exp(1) noise values (without <unk>), each with value log(1 / beam);
    replayed noise and Philox."""
    g = torch.Generator().manual_seed(v)
    rows, beam, temp = 3, 5, 1.3
    x = torch.full((rows, v), 0.25)
    x[1, ::2] += 1.0                                                  # a row with two plateaus: the upper one (v / 2 tokens) is the top-k
    noise = torch.empty(rows, v).exponential_(1, generator=g)
    kept = x.clone()
    kept[x < x.topk(top_k, dim=-1).values[:, -1:]] = float("-inf")
    kept[:, 1] = float("-inf")
    want = torch.topk(torch.softmax(kept / temp, -1) / noise, beam, dim=-1).indices
    ng = hip.n_groups(v)
    pad = torch.full((rows, ng * 64), float("-inf"))
    pad[:, :v] = x
    gmax = pad.view(rows, ng, 64).max(-1).values.cuda()

    def run(kind, nz):
        pi = torch.empty((rows, beam), dtype=torch.int32, device="cuda")
        pv = torch.empty((rows, beam), device="cuda")
        err = torch.zeros(1, dtype=torch.int32, device="cuda")
        if kind == "groups":
            hip.beam_row_sample_groups(x.cuda(), v, gmax, rows, beam, beam, top_k, temp, 1, nz, 5, 0, 2, pi, pv, err)
        else:
            hip.beam_row_sample(x.cuda(), v, rows, beam, beam, top_k, temp, 1, nz, 5, 0, 2, pi, pv, err, exact=kind == "exact")
        return pi.cpu().long(), pv.cpu(), int(err.item())

    fi, _, fe = run("fast", noise.cuda())          # top_k > 256 already dispatches to the general kernel: right answer, no flag
    assert (fe & hip.ERR_OVERFLOW) if top_k <= 256 else (fe == 0 and fi.tolist() == want.tolist())
    if top_k <= ng and top_k <= 256:
        assert run("groups", noise.cuda())[2] & hip.ERR_OVERFLOW
    for nz in (noise.cuda(), None):
        got, pv, e = run("exact", nz)
        assert e == 0
        if nz is not None:
            assert got.tolist() == want.tolist()
        assert bool((got != 1).all()) and all(len(set(r)) == beam for r in got.tolist())
        assert bool(torch.isfinite(kept.gather(1, got)).all())                          # only surviving tokens
        close(pv, torch.full((rows, beam), float(np.log(1.0 / beam))), atol=1e-5, rtol=0)
