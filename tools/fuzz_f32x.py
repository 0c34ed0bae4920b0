"""Randomised sweep of the split-operand fp32 entry points (csrc/gemm_f32x.hip) on the GPU box, across the shapes where the dispatcher
switches tiles (128 x 128, 128 x 64, 64 x 64, the deep-ring 64 x 64 kernel) and loaders (dense, convolution with a wave-uniform tap,
the generic per-lane tap decode):

  linear : ``dh_linear_f32x`` (a [M, K] fp32 @ planes(w [N, K])^T, optional bias / scale + shift / residual / ReLU, strided operands and
           output) against the fp64 product of the same fp32 operands -- error within 1.2e-6 of sum |a||w| per element (the analytic bound:
           each operand is represented to 2^-22, the lo x lo term dropped is 2^-22 of the product: <= 5 x 2^-22 in the worst case; typical
           errors are 1e-7);
  conv   : ``dh_conv2d_nhwc_f32x`` (KS 1 / 3 / 5 / 7, stride 1 / 2, any padding, Cin % 4 == 0, residual / ReLU) against fp64 ``conv2d``;
  planes : (round 6) the kernels that take operands STORED as fp16 planes -- ``dh_linear_f32xp`` (+ its 64-column group maxima),
           ``dh_linear_f32xp_wreg`` (fp32 / planes out), ``dh_conv2d_nhwc_f32xp`` / ``dh_conv2d_nhwc_f32x_planes_out``,
           ``dh_conv1x1_f32x_stream`` -- against ``dh_linear_f32x`` / ``dh_conv2d_nhwc_f32x`` on the same values: BITWISE equality;
every output sits in a canary-guarded buffer: an out-of-bounds or row-padding write fails the trial.  TEST INFRASTRUCTURE.

    python tools/fuzz_f32x.py --trials 200 > gpurun_out/fuzz_f32x.jsonl
"""
import argparse
import json
import os
import random
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from deephumor_amd import hip, f32xp   # noqa: E402

CANARY = 12345.0


def guarded(rows, cols, ld, margin=256):
    buf = torch.full((margin + rows * ld + margin,), CANARY, dtype=torch.float32, device="cuda")
    view = buf[margin:margin + rows * ld].view(rows, ld)[:, :cols]

    def check():
        ok = bool((buf[:margin] == CANARY).all()) and bool((buf[margin + rows * ld:] == CANARY).all())
        if ld > cols:
            ok = ok and bool((buf[margin:margin + rows * ld].view(rows, ld)[:, cols:] == CANARY).all())
        return ok
    return view, check


def linear_trial(rng, idx):
    g = torch.Generator().manual_seed(50000 + idx)
    m = rng.choice([rng.randint(1, 64), rng.randint(65, 700), rng.randint(701, 3000), 1280, 256 * rng.randint(1, 8), rng.randint(3001, 20000)])
    n = rng.choice([rng.randint(1, 64), 64, rng.randint(65, 700), rng.randint(701, 5000), 512, 2048])
    k = 4 * rng.choice([rng.randint(1, 16), rng.randint(17, 128), 128, 192, 512, 8 * rng.randint(1, 64)])
    while m * n > (1 << 25) or m * k > (1 << 24) or n * k > (1 << 24):
        m = max(1, m // 2)
    lda = k + 4 * rng.choice([0, 0, 1, 3])
    scale_a = rng.choice([1.0, 1e-3, 50.0])
    a = (torch.randn(m, lda, generator=g) * scale_a).cuda()[:, :k]
    w = (torch.randn(n, k, generator=g) / k ** 0.5).cuda()
    bias = torch.randn(n, generator=g).cuda() if rng.random() < 0.7 else None
    affine = rng.random() < 0.3
    sc, sh = ((torch.rand(n, generator=g) + 0.5).cuda(), torch.randn(n, generator=g).cuda()) if affine else (None, None)
    ldr = n + 4 * rng.choice([0, 1])
    res = torch.randn(m, ldr, generator=g).cuda()[:, :n] if rng.random() < 0.4 else None
    relu = rng.random() < 0.5
    out, check = guarded(m, n, n + rng.choice([0, 0, 3, 4, 64]))
    hip.linear_f32x(a, hip.split_f32x(w), bias, scale=sc, shift=sh, residual=res, relu=relu, out=out)
    want = a.double() @ w.double().t()
    mag = a.double().abs() @ w.double().abs().t()
    if bias is not None:
        want = want + bias.double()
        mag = mag + bias.double().abs()
    if affine:
        want = want * sc.double() + sh.double()
        mag = mag * sc.double() + sh.double().abs()
    if res is not None:
        want = want + res.double()
        mag = mag + res.double().abs()
    if relu:
        want = torch.relu(want)
    err = float(((out.double() - want).abs() / mag.clamp_min(1e-30)).max())
    ok = err < 1.2e-6 and check() and bool(torch.isfinite(out).all())
    return dict(kind="linear", m=m, n=n, k=k, lda=lda, bias=bias is not None, affine=affine, residual=res is not None, relu=relu, rel_err=err, ok=ok)


def conv_trial(rng, idx):
    g = torch.Generator().manual_seed(60000 + idx)
    ks = rng.choice([1, 1, 3, 3, 5, 7])
    stride = rng.choice([1, 1, 2])
    pad = rng.choice([0, ks // 2, ks // 2, 1])
    cin = 4 * rng.choice([1, 2, 8, 16, 16, 32, 64, rng.randint(1, 40)])
    cout = rng.choice([rng.randint(1, 64), 64, 64, 128, 256, rng.randint(65, 600)])
    h, w_ = rng.randint(max(1, ks - 2 * pad), 40), rng.randint(max(1, ks - 2 * pad), 40)
    n = rng.randint(1, 6)
    x = torch.randn(n, cin, h, w_, generator=g).cuda()
    wt = (torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5).cuda()
    sc, sh = (torch.rand(cout, generator=g) + 0.5).cuda(), torch.randn(cout, generator=g).cuda()
    relu = rng.random() < 0.6
    want = F.conv2d(x.double(), wt.double(), stride=stride, padding=pad) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
    mag = F.conv2d(x.double().abs(), wt.double().abs(), stride=stride, padding=pad) * sc.double()[None, :, None, None] + sh.double().abs()[None, :, None, None]
    res = None
    if rng.random() < 0.4:
        res = torch.randn(want.shape, generator=g).cuda()
        want, mag = want + res.double(), mag + res.double().abs()
    if relu:
        want = torch.relu(want)
    got = hip.conv2d_nhwc_f32x(x.permute(0, 2, 3, 1).contiguous(), hip.split_f32x(wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()), ks, sc, sh,
                               residual=None if res is None else res.permute(0, 2, 3, 1).contiguous(), relu=relu, stride=stride, pad=pad)
    err = float(((got.permute(0, 3, 1, 2).double() - want).abs() / mag.clamp_min(1e-30)).max())
    ok = err < 1.2e-6 and tuple(got.shape) == (n, want.shape[2], want.shape[3], cout) and bool(torch.isfinite(got).all())
    return dict(kind="conv", n=n, h=h, w=w_, cin=cin, cout=cout, ks=ks, stride=stride, pad=pad, residual=res is not None, relu=relu, rel_err=err, ok=ok)


def planes_trial(rng, idx):
    g = torch.Generator().manual_seed(70000 + idx)
    kind = rng.choice(["linear", "linear", "wreg", "conv", "conv", "stream"])
    if kind == "linear":
        m = rng.choice([rng.randint(1, 300), 1280, rng.randint(301, 4000)])
        n = rng.choice([rng.randint(1, 200), rng.randint(201, 6000), 36541 if m <= 1280 else 4096])
        k = 32 * rng.randint(1, 24)
        a = torch.randn(m, k, generator=g).cuda()
        w = (torch.randn(n, k, generator=g) / k ** 0.5).cuda()
        b = torch.randn(n, generator=g).cuda() if rng.random() < 0.8 else None
        relu = rng.random() < 0.3
        wp, ap = hip.split_f32x(w), f32xp.split_act(a)
        want = hip.linear_f32x(a, wp, b, relu=relu)
        out, check = guarded(m, n, (n + 255) // 256 * 256)
        gm = torch.zeros(m, hip.n_groups(n), device="cuda")
        f32xp.linear(ap, wp, b, relu=relu, out=out, group_max=gm)
        pad = torch.full((m, gm.shape[1] * 64 - n), float("-inf"), device="cuda")
        ok = torch.equal(out, want) and check() and torch.equal(gm, torch.cat([want, pad], 1).view(m, -1, 64).amax(-1))
        return dict(kind="planes_linear", m=m, n=n, k=k, bias=b is not None, relu=relu, ok=bool(ok))
    if kind == "wreg":
        n, k = rng.choice([(512, 512), (1536, 512), (2048, 512), (512, 2048), (2048, 768), (2048, 1024), (64 * rng.randint(1, 8), 512), (256, 384)])
        m = rng.choice([rng.randint(1, 100), rng.randint(101, 1300), 1280, 3000])
        if not hip.load().dh_linear_f32x_wreg_supported(m, n, k):
            return dict(kind="planes_wreg", m=m, n=n, k=k, skipped=True, ok=True)
        a = (torch.randn(m, k, generator=g) * rng.choice([1.0, 30.0])).cuda()
        w = (torch.randn(n, k, generator=g) / k ** 0.5).cuda()
        b = torch.randn(n, generator=g).cuda()
        relu = rng.random() < 0.5
        res = torch.randn(m, n, generator=g).cuda() if rng.random() < 0.4 else None
        wp = hip.split_f32x(w)
        want = hip.linear_f32x(a, wp, b, relu=relu, residual=res)
        got, gp = f32xp.linear_wreg(f32xp.split_act(a), hip.pack_f32x_fragments(wp), b, relu=relu, residual=res, want="both")
        ok = torch.equal(got, want) and torch.equal(gp, f32xp.split_act(want))
        return dict(kind="planes_wreg", m=m, n=n, k=k, relu=relu, residual=res is not None, ok=bool(ok))
    if kind == "conv":
        ks, stride = rng.choice([(3, 1), (3, 1), (3, 2), (1, 1), (1, 2)])
        pad = ks // 2
        cin, cout = 32 * rng.randint(1, 10), 4 * rng.choice([16, 32, 33, 64, 128, rng.randint(32, 160)])
        n, hw = rng.randint(1, 40), rng.randint(3, 30)
        x = torch.randn(n, hw, hw, cin, generator=g).cuda()
        w = (torch.randn(cout, ks * ks * cin, generator=g) / (ks * ks * cin) ** 0.5).cuda()
        sc, sh = (torch.rand(cout, generator=g) + 0.5).cuda(), torch.randn(cout, generator=g).cuda()
        relu = rng.random() < 0.6
        wp = hip.split_f32x(w)
        ho = (hw + 2 * pad - ks) // stride + 1
        res = torch.randn(n, ho, ho, cout, generator=g).cuda() if rng.random() < 0.4 else None
        want = hip.conv2d_nhwc_f32x(x, wp, ks, sc, sh, residual=res, relu=relu, stride=stride, pad=pad)
        xp = f32xp.split_act(x.view(-1, cin)).view(2, n, hw, hw, cin)
        y, yp = f32xp.conv2d_nhwc(xp, wp, ks, sc, sh, residual=res, relu=relu, stride=stride, pad=pad, want="both")
        ok = torch.equal(y, want) and torch.equal(yp, f32xp.split_act(want.view(-1, cout))[:, :, :cout].reshape(yp.shape))
        if res is None:
            ok = ok and torch.equal(f32xp.conv2d_nhwc_planes_out(x, wp, ks, sc, sh, relu=relu, stride=stride, pad=pad), yp)
        return dict(kind="planes_conv", n=n, hw=hw, cin=cin, cout=cout, ks=ks, stride=stride, residual=res is not None, relu=relu, ok=bool(ok))
    cin = rng.choice([64, 128, 256])
    cout = 64 * rng.choice([2, 4, 8, 16])
    wpc = {64: 8, 128: 4, 256: 2}[cin]
    need = 8 * 8 * (32 * wpc // (cout // 64)) * 32                  # rows the persistent grid wants (dh_conv1x1_f32x_stream_supported)
    hw = rng.choice([7, 14, 28])
    n = need // (hw * hw) + rng.randint(1, 4)
    x = torch.randn(n, hw, hw, cin, generator=g).cuda()
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).cuda()
    sc, sh = (torch.rand(cout, generator=g) + 0.5).cuda(), torch.randn(cout, generator=g).cuda()
    relu = rng.random() < 0.7
    res = torch.randn(n, hw, hw, cout, generator=g).cuda() if rng.random() < 0.6 else None
    wp = hip.split_f32x(w)
    pk = f32xp.pack_conv1x1(wp)
    if pk is None or not f32xp.conv1x1_stream_supported(n * hw * hw, cin, cout):
        return dict(kind="planes_stream", cin=cin, cout=cout, rows=n * hw * hw, ok=False, error="unexpectedly unsupported")
    ok = torch.equal(f32xp.conv1x1_stream(x, pk, sc, sh, residual=res, relu=relu), hip.conv2d_nhwc_f32x(x, wp, 1, sc, sh, residual=res, relu=relu))
    return dict(kind="planes_stream", cin=cin, cout=cout, rows=n * hw * hw, residual=res is not None, relu=relu, ok=bool(ok))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args(argv)
    rng = random.Random(args.seed)
    bad, worst = 0, 0.0
    for i in range(args.trials):
        u = rng.random()
        fn = linear_trial if u < 0.35 else conv_trial if u < 0.6 else planes_trial
        try:
            rec = fn(rng, args.seed * 100000 + i)
        except Exception as e:                                    # noqa: BLE001 -- a raising trial is a failing trial
            rec = dict(kind=fn.__name__, ok=False, error=repr(e)[:300])
        bad += not rec["ok"]
        worst = max(worst, rec.get("rel_err", 0.0))
        print(json.dumps(rec), flush=True)
    hip.f32x_take_overflow()
    print(json.dumps({"trials": args.trials, "failures": bad, "worst_rel_err_vs_fp64": worst}), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
