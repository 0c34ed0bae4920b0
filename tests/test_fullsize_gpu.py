"""BASELINE-size runs inside ``-m gpu``: the workloads bench.py times, checked -- not only timed.
  * fp32, 256 images, V = 36,541, greedy: sixteen rows of the batch equal the captions the reference produced for them (golden G18)
  * bf16 C2 / C3 at 256 images x beam 5: run-to-run equality and 128 + 128 split (``img0``) equality at the tile
    configurations the full batch selects (gemm_bf16.hip picks tiles by workgroup count)
  * the C5 shape: fp16, 300 templates x beam 10 with labels: run-to-run equality and a 150 + 150 split
Sized to stay under about a minute in all."""
import pytest
import torch

pytestmark = pytest.mark.gpu

V = 36541


def _model(kind, dtype):
    import deephumor_amd.models as M
    from deephumor_amd.synth import synth_state_dict
    model = getattr(M, kind)(V).eval()
    sd = synth_state_dict(model.state_dict(), seed=1234)
    model.load_state_dict(sd)
    model = model.cuda()
    if dtype != torch.float32:
        model = model.to(dtype)
    return model, sd


@pytest.mark.parametrize("kind", ["CaptioningLSTM", "CaptioningTransformer"])
def test_fp32_full_batch_greedy_rows_equal_the_reference(kind):
    """N = 256, V = 36,541, fp32: the greedy captions of sixteen images inside the batch are, token for token, the captions the REFERENCE
    produced for them (golden G18, recorded by oracle/make_golden.py from /root/reference; rounds 1 - 5 ran the CPU oracle on rows
    {0, 77, 255} here: 20 s of host time for three rows)."""
    import os
    import numpy as np
    from deephumor_amd.synth import synth_images
    g18 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"g18_bench_rows_{kind}.npz"))
    model, sd = _model(kind, torch.float32)
    imgs = synth_images(256, seed=0)
    with torch.no_grad():
        toks, lens = model.generate_batch(imgs.cuda(), max_len=32, beam_size=1, top_k=1)
    for i in (int(j) for j in g18["images"]):
        assert toks[i, :int(lens[i])].cpu().tolist() == g18[f"greedy_{i}"].tolist(), (kind, i)


_ORACLE_BEAM = {}




# What the 16-bit paths must stay within of the REFERENCE at the BASELINE shape.  Round 6: sixteen images instead of
# three, from golden G18 (recorded from the real reference by oracle/make_golden.py r6, so the GPU run does not pay for the rows), and
# the bounds follow what the committed kernels give -- step-0 logits 1.25 x the worst observed over the sixteen images, greedy floors =
# observed - 0.05 -- so a kernel that loses a further bit of precision turns the run red (checked with a deliberately degraded LSTM
# step).  Observed on the round-6 tree (scripts/r6_gpu_calls/r6_call02.sh, DH_GATE_RECORD=1; gpurun_out/r6/oracle_gate_*.json):
G18_OBSERVED = {  # (max |dlogit|, mean |dlogit| of the worst image) at step 0 over 16 images x (4,096 sampled columns + top-8); greedy token match over 16 captions
    ("CaptioningLSTM", torch.bfloat16): (0.1545, 0.03064, 0.6035), ("CaptioningLSTM", torch.float16): (0.0153, 0.00303, 0.9922),
    ("CaptioningTransformer", torch.bfloat16): (0.1198, 0.02475, 0.8926), ("CaptioningTransformer", torch.float16): (0.0156, 0.00304, 0.9160)}
LOGIT_MARGIN, GREEDY_SLACK = 1.25, 0.05


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("kind", ["CaptioningLSTM", "CaptioningTransformer"])
def test_16bit_big_batch_kernels_against_the_oracle(kind, dtype):
    """The kernels only a BASELINE-size batch selects (256 images x beam 5 = 1,280 rows, V = 36,541: ``vocab_wreg`` / ``vocab_areg256``,
    ``linear_wreg``, ``lstm_wreg``, the 256-workgroup encoder kernels) tied to the REFERENCE directly, not through bit-equality with
    older kernels: (a) the step-0 logits the beam-5 decode itself computes (256 rows) for the sixteen G18 images against the logits
    the real reference recorded (4,096 sampled columns + its top-8), within 1.25 x the worst error the committed kernels show, the
    arg-max inside the reference's near-top set; (b) step 1 (1,280 rows, each row teacher-forced by the token the engine drew for it)
    for three rows against ``oracle.ref_path.model_forward``; (c) greedy ids of the sixteen images at N = 256 against the reference's
    captions at or above (observed - 0.05)."""
    import json
    import os
    import numpy as np
    from oracle import ref_path as R
    from deephumor_amd.synth import synth_images
    g18 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"g18_bench_rows_{kind}.npz"))
    sel = [int(i) for i in g18["images"]]
    cols = torch.from_numpy(g18["cols"])
    model, sd = _model(kind, dtype)
    imgs = synth_images(256, seed=0)
    picks = {0: sel, 1: [0 * 5 + 0, 77 * 5 + 2, 255 * 5 + 4]}
    cap = {}

    def hook(i, lg, tokens):
        if i in picks:
            rows = torch.tensor(picks[i], device=lg.device)
            cap[i] = (lg[rows].float().cpu(), tokens[rows].cpu())
    hook.with_tokens = True
    with torch.no_grad():
        model.generate_batch(imgs.cuda(), max_len=32, beam_size=5, top_k=50, temperature=1.0, seed=42, logits_hook=hook)
    obs_max, obs_mean, obs_greedy = G18_OBSERVED[(kind, dtype)]
    tol_max, tol_mean = LOGIT_MARGIN * obs_max, LOGIT_MARGIN * obs_mean
    if os.environ.get("DH_GATE_RECORD"):                              # (the run that sets G18_OBSERVED: record, do not judge)
        tol_max = tol_mean = 1e9
        obs_greedy = 0.0
    seen = {}
    # (a) step 0 against the recorded reference rows
    got0 = cap[0][0]
    worst_max = worst_mean = 0.0
    for j, img in enumerate(sel):
        want_cols = torch.from_numpy(g18[f"step0_cols_{img}"])
        t8i, t8v = torch.from_numpy(g18[f"step0_top8_idx_{img}"]), torch.from_numpy(g18[f"step0_top8_val_{img}"])
        d = torch.cat([(got0[j][cols] - want_cols).abs(), (got0[j][t8i] - t8v).abs()])
        worst_max, worst_mean = max(worst_max, float(d.max())), max(worst_mean, float(d.mean()))
        assert float(d.max()) < tol_max and float(d.mean()) < tol_mean, (kind, dtype, img, float(d.max()), float(d.mean()), tol_max, tol_mean)
        # the engine's arg-max is one of the reference's near-top tokens (its top-8 reach further down than 2 x the tolerance)
        am, band = int(got0[j].argmax()), float(t8v[0]) - 2 * tol_max
        if float(t8v[7]) < band:                                      # (else the recorded top-8 do not reach below the band: undecidable)
            assert am in t8i.tolist() and float(t8v[t8i.tolist().index(am)]) >= band, (kind, dtype, img, am)
    seen["step0_worst_max_mean_16_images"] = [round(worst_max, 4), round(worst_mean, 5)]
    # (b) step 1: 1,280 rows, teacher-forced by the engine's own draw
    got, toks = cap[1]
    for j, r in enumerate(picks[1]):
        prefix = toks[j, :1].long()[None]
        want = R.model_forward(kind, sd, model._hp, imgs[r // 5:r // 5 + 1], prefix)[0, 1]
        d = (got[j] - want).abs()
        seen[f"step1_row{r}"] = [round(float(d.max()), 4), round(float(d.mean()), 5)]
        # (one more layer stack of rounding than step 0 and the whole vocabulary instead of a sample: 1.5 x step 0's bound)
        assert float(d.max()) < 1.5 * tol_max and float(d.mean()) < 1.5 * tol_mean, (kind, dtype, r, float(d.max()), float(d.mean()))
        assert float(want[int(got[j].argmax())]) >= float(want.max()) - 2 * tol_max, (kind, dtype, r)
    # (c) greedy captions of the sixteen images inside the 256-image batch
    with torch.no_grad():
        toks, lens = model.generate_batch(imgs.cuda(), max_len=32, beam_size=1, top_k=1)
    same = total = 0
    for i in sel:
        want = g18[f"greedy_{i}"].tolist()
        g = toks[i, :int(lens[i])].cpu().tolist()
        total += max(len(want), len(g))
        same += sum(int(a == b) for a, b in zip(want, g))
    seen["greedy_token_match_16_images"] = round(same / total, 4)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):                                            # (kept next to the run's other records)
        with open(os.path.join(out, f"oracle_gate_{kind}_{str(dtype).split('.')[-1]}.json"), "w") as f:
            json.dump(seen, f)
    print(kind, dtype, seen)
    assert same / total >= obs_greedy - GREEDY_SLACK, (kind, dtype, seen)


# bench.py's precision_vs_fp32_hip as an assertion: greedy tokens of the 16-bit paths against the fp32 HIP path (which
# is bit-exact vs the CPU oracle: the tests above) over ALL 256 bench images.  Floors: C2 / C3.
MATCH_FLOOR = {("CaptioningLSTM", torch.bfloat16): 0.55, ("CaptioningTransformer", torch.bfloat16): 0.80,
               ("CaptioningLSTM", torch.float16): 0.93, ("CaptioningTransformer", torch.float16): 0.95}


@pytest.mark.parametrize("kind", ["CaptioningLSTM", "CaptioningTransformer"])
def test_16bit_greedy_token_match_against_the_fp32_path_over_all_bench_images(kind):
    import bench
    from deephumor_amd.synth import synth_images
    imgs = synth_images(256, seed=0).cuda()
    m32, _ = _model(kind, torch.float32)
    ref = bench.greedy_all(m32, imgs)
    del m32
    torch.cuda.empty_cache()
    for dtype in (torch.bfloat16, torch.float16):
        model, _ = _model(kind, dtype)
        cmp_ = bench.compare_greedy(ref, bench.greedy_all(model, imgs))
        print(kind, dtype, cmp_)
        assert cmp_["token_match"] >= MATCH_FLOOR[(kind, dtype)], (kind, dtype, cmp_)
        # step-0 logits: the same bound as the oracle gate's maximum (all 256 images x the whole vocabulary instead of a sample)
        assert cmp_["step0_logit_max_abs_err"] < 1.5 * LOGIT_MARGIN * G18_OBSERVED[(kind, dtype)][0], (kind, dtype, cmp_)
        del model
        torch.cuda.empty_cache()


@pytest.mark.parametrize("kind", ["CaptioningLSTM", "CaptioningTransformer"])
def test_bf16_full_batch_is_repeatable_and_split_invariant(kind):
    from deephumor_amd.synth import synth_images
    model, _ = _model(kind, torch.bfloat16)
    imgs = synth_images(256, seed=0).cuda()
    kw = dict(max_len=32, beam_size=5, top_k=50, temperature=1.0, seed=42)
    with torch.no_grad():
        t1, l1 = model.generate_batch(imgs, **kw)
        t2, l2 = model.generate_batch(imgs, **kw)
        assert torch.equal(t1, t2) and torch.equal(l1, l2)
        ta, la = model.generate_batch(imgs[:128], img0=0, **kw)
        tb, lb = model.generate_batch(imgs[128:], img0=128, **kw)
    assert tuple(t1.shape) == (256, 32) and int(t1.max()) < V and not bool((t1 == 1).any())
    # a row's arithmetic does not depend on which rows share its tile: the 128-image halves reproduce the batch
    assert torch.equal(torch.cat([ta, tb]), t1) and torch.equal(torch.cat([la, lb]), l1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("kind", ["CaptioningLSTM", "CaptioningTransformer"])
def test_bit_identical_kernel_options_at_the_baseline_shape(kind, dtype):
    """Every run-time option documented as selecting between BIT-IDENTICAL kernels, one at a time and all together, at the shape where
    the big-batch kernels run (256 images x beam 5, V = 36,541), in both 16-bit types: same tokens, same lengths
    (tools/fuzz_variants.py sweeps random combinations at small batches, this is the 1,280-row leg)."""
    from deephumor_amd import hip as H
    from deephumor_amd.synth import synth_images
    model, _ = _model(kind, dtype)
    imgs = synth_images(256, seed=3).cuda()
    kw = dict(max_len=12, beam_size=5, top_k=50, temperature=1.0, seed=21)
    with torch.no_grad():
        base = model.generate_batch(imgs, **kw)
    if kind == "CaptioningTransformer":
        sets = [dict(decode_wreg_min_rows=100000), dict(vocab_wreg_transformer=1), dict(vocab_areg=0),
                dict(decode_wreg=0, vocab_wreg_transformer=1)]
    else:
        sets = [dict(lstm_wreg_min_rows=100000), dict(vocab_wreg=0), dict(lstm_wreg=0, vocab_wreg=0, vocab_areg=0)]
    sets += [dict(encoder_generic=2)]
    for opts in sets:
        with torch.no_grad(), H.option_scope(**opts):
            got = model.generate_batch(imgs, **kw)
        assert torch.equal(got[0], base[0]) and torch.equal(got[1], base[1]), (kind, dtype, opts)


def test_c5_shape_fp16_beam10_300_templates():
    import numpy as np
    from deephumor_amd.synth import synth_images
    model, _ = _model("CaptioningTransformerWithLabels", torch.float16)
    imgs = synth_images(300, seed=2).cuda()
    g = np.random.Generator(np.random.Philox(key=[1, 0]))
    labels = torch.from_numpy(g.integers(6, V, size=(300, 3)).astype(np.int64)).cuda()
    kw = dict(max_len=32, beam_size=10, top_k=50, temperature=1.0, seed=7)
    with torch.no_grad():
        t1, l1 = model.generate_batch(imgs, labels, **kw)
        t2, l2 = model.generate_batch(imgs, labels, **kw)
        assert torch.equal(t1, t2) and torch.equal(l1, l2)
        ta, la = model.generate_batch(imgs[:150], labels[:150], img0=0, **kw)
        tb, lb = model.generate_batch(imgs[150:], labels[150:], img0=150, **kw)
    assert tuple(t1.shape) == (300, 32) and int(t1.max()) < V and not bool((t1 == 1).any()) and int(l1.min()) >= 1
    # the bit-identical kernel options at THIS shape (3,000 rows per position: the classifier's row split, three-round GEMMs) and at the
    # 38-template shard of an 8-rank run (380 rows: the register-streamed classifier, one-round GEMMs)
    from deephumor_amd import hip as H
    short = dict(kw, max_len=10)
    for n in (300, 38):
        with torch.no_grad():
            base = model.generate_batch(imgs[:n], labels[:n], **short)
            for opts in (dict(decode_wreg_min_rows=100000), dict(vocab_areg=0), dict(vocab_wreg_transformer_max_rows=0),
                         dict(decode_wreg=0, vocab_wreg_transformer=1)):
                with H.option_scope(**opts):
                    got = model.generate_batch(imgs[:n], labels[:n], **short)
                assert torch.equal(got[0], base[0]) and torch.equal(got[1], base[1]), (n, opts)
    assert torch.equal(torch.cat([ta, tb]), t1) and torch.equal(torch.cat([la, lb]), l1)


@pytest.mark.parametrize("kind", ["CaptioningLSTM", "CaptioningTransformer"])
def test_fp32_full_batch_sampled_beam_rows_equal_the_reference(kind):
    """Stochastic beam search AT THE BASELINE SHAPE (256 images x beam 5, top_k 50, 32 tokens, V = 36,541 -- the shape where the
    big-batch kernels are selected), fp32, through the product option ``rng="torch"`` (image i replays ``torch.manual_seed(700 + i)``):
    rows 0 and 255 token for token against (a) the captions recorded from the REAL reference under those seeds (golden G15) and
    (b) the CPU oracle run here under the same seeds."""
    import os
    import numpy as np
    from oracle import ref_path as R
    from deephumor_amd.synth import synth_images
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"g15_bench_beam_{kind}.npz"))
    model, sd = _model(kind, torch.float32)
    imgs = synth_images(256, seed=0)
    kw = dict(max_len=32, beam_size=5, top_k=50, temperature=1.0)
    with torch.no_grad():
        toks, lens = model.generate_batch(imgs.cuda(), seed=700, rng="torch", **kw)
    for i in (0, 255):
        got = toks[i, :int(lens[i])].cpu().tolist()
        assert got == g[f"beam_{i}"].tolist(), (kind, i, "vs the reference-recorded caption")
    # (the oracle reproduces G15's image 255 in tests/test_oracle_golden.py; image 0 here, next to the HIP path -- once per process:
    #  tests/test_f32x_gpu.py re-runs this test on the split-operand path against the same oracle caption)
    if kind not in _ORACLE_BEAM:
        torch.manual_seed(700)
        _ORACLE_BEAM[kind] = R.model_generate(kind, sd, model._hp, imgs[:1], **kw).reshape(-1).tolist()
    assert toks[0, :int(lens[0])].cpu().tolist() == _ORACLE_BEAM[kind], (kind, "vs the oracle")
    # the single-image reference call sequence: torch.manual_seed(s); model.generate(image, rng="torch")
    torch.manual_seed(700 + 255)
    with torch.no_grad():
        one = model.generate(imgs[255:256].cuda(), rng="torch", **kw)
    assert one.reshape(-1).cpu().tolist() == g["beam_255"].tolist()


@pytest.mark.parametrize("kind,dtype,beam,n", [("CaptioningTransformer", torch.bfloat16, 5, 1024), ("CaptioningLSTM", torch.float16, 5, 2048)])
def test_one_batch_of_thousands_of_images_equals_its_256_image_shards(kind, dtype, beam, n):
    """C4's global batch (and half of it) as ONE generate_batch call on one GPU -- 5,120 / 10,240 rows per position: the encoder in
    chunks of 256, the classifier in several launches, multi-round decode GEMMs -- against the same images as 256-image shards with
    ``img0``: same tokens, same lengths (what an 8-rank run produces)."""
    from deephumor_amd.synth import synth_images
    model, _ = _model(kind, dtype)
    imgs = synth_images(n, seed=4).cuda()
    kw = dict(max_len=8, beam_size=beam, top_k=50, temperature=1.0, seed=3)
    with torch.no_grad():
        big = model.generate_batch(imgs, **kw)
        parts = [model.generate_batch(imgs[i:i + 256], img0=i, **kw) for i in range(0, n, 256)]
    assert torch.equal(big[0], torch.cat([p[0] for p in parts])) and torch.equal(big[1], torch.cat([p[1] for p in parts]))
