"""Randomised sweep of the convolution entry points on the GPU box against ``torch.nn.functional.conv2d`` (fp32 math on the same
rounded operands), outputs in canary-guarded buffers (an out-of-bounds write fails the trial):

  nhwc   : ``dh_conv2d_nhwc_bn_act`` 16-bit channels-last implicit GEMM -- KS 1 / 3 / 7, stride 1 / 2, any padding, BatchNorm scale /
           shift, ReLU, residual, batch 1-48, odd spatial sizes, channel counts that are multiples of 8;
  direct : ``dh_conv3x3_direct_nhwc`` / ``dh_bottleneck_tail_nhwc`` / ``dh_bottleneck_tail_s3_nhwc`` at the shapes their ``*_supported``
           predicates accept, random batch, against the implicit-GEMM launches (bit-equal by design) and fp32 math;
  stem   : ``dh_stem_conv7_bn_relu_maxpool`` (both input formats) at random image sizes against conv + BN + ReLU + max_pool2d;
  fp32   : ``dh_conv2d_bn_act`` (NCHW fp32 parity path) against F.conv2d.
TEST INFRASTRUCTURE.

    python tools/fuzz_conv.py --trials 200 > gpurun_out/fuzz_conv.jsonl
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

from deephumor_amd import hip          # noqa: E402

CANARY = 12345.0


def guarded(shape, dtype, margin=512):
    n = 1
    for s in shape:
        n *= s
    buf = torch.full((margin + n + margin,), CANARY, dtype=dtype, device="cuda")
    view = buf[margin:margin + n].view(*shape)
    return view, lambda: bool((buf[:margin] == CANARY).all()) and bool((buf[margin + n:] == CANARY).all())


def tol(dt, want):
    return (4e-2 if dt == torch.bfloat16 else 5e-3) * max(1.0, float(want.abs().max()))


def nhwc_trial(rng, idx):
    g = torch.Generator().manual_seed(40000 + idx)
    dt = rng.choice([torch.bfloat16, torch.float16])
    ks = rng.choice([1, 1, 3, 3, 7])
    stride = rng.choice([1, 1, 2])
    pad = rng.choice([ks // 2, ks // 2, 0, 1])
    n = rng.choice([1, 2, rng.randint(3, 48)])
    h, w = rng.randint(max(ks - 2 * pad, 1), 40), rng.randint(max(ks - 2 * pad, 1), 40)
    cin = 8 * rng.choice([1, rng.randint(1, 16), 8, 16, 32, 64])
    cout = 8 * rng.choice([rng.randint(1, 16), 8, 16, 32, 64, 128])
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    rec = dict(kind="nhwc", dt=str(dt)[6:], N=n, H=h, W=w, Cin=cin, Cout=cout, KS=ks, stride=stride, pad=pad)
    if ho < 1 or wo < 1:
        return dict(rec, ok=True, skipped=True)
    x = torch.randn(n, h, w, cin, generator=g).to(dt).cuda()
    wt = (torch.randn(cout, ks, ks, cin, generator=g) / (ks * ks * cin) ** 0.5).to(dt).cuda()
    scale, shift = (torch.rand(cout, generator=g) + 0.5).cuda(), torch.randn(cout, generator=g).cuda()
    relu = rng.random() < 0.6
    res = torch.randn(n, ho, wo, cout, generator=g).to(dt).cuda() if rng.random() < 0.4 else None
    out, check = guarded((n, ho, wo, cout), dt)
    hip._launch("dh_conv2d_nhwc_bn_act", hip._ptr(x), hip._ptr(wt), hip._ptr(scale), hip._ptr(shift), hip._ptr(res), hip._ptr(out),
                n, h, w, cin, cout, ks, stride, pad, int(relu), hip._dt(x), hip._stream())
    torch.cuda.synchronize()
    want = F.conv2d(x.float().permute(0, 3, 1, 2), wt.float().permute(0, 3, 1, 2), stride=stride, padding=pad)
    want = want * scale[None, :, None, None] + shift[None, :, None, None]
    if res is not None:
        want = want + res.float().permute(0, 3, 1, 2)
    if relu:
        want = want.relu()
    err = float((out.float().permute(0, 3, 1, 2) - want).abs().max())
    rec.update(relu=relu, res=res is not None, err=err, canary_ok=check(), ok=bool(err <= tol(dt, want) and check()))
    return rec


def direct_trial(rng, idx):
    g = torch.Generator().manual_seed(50000 + idx)
    dt = rng.choice([torch.bfloat16, torch.float16])
    which = rng.choice(["c3_56", "c3_28", "tail_56", "tail_56_s1", "tail_56_s1f", "tail_28", "tail_28_s2", "tail_28_s2f", "s3", "s3_conv2"])
    n = rng.choice([1, 2, 3, rng.randint(4, 40)])
    rec = dict(kind="direct", which=which, dt=str(dt)[6:], N=n)
    hw, c = {"c3_56": (56, 64), "tail_56": (56, 64), "tail_56_s1": (56, 64), "tail_56_s1f": (56, 64), "c3_28": (28, 128), "tail_28": (28, 128), "tail_28_s2": (28, 128), "tail_28_s2f": (28, 128), "s3": (14, 256), "s3_conv2": (14, 256)}[which]
    y1 = torch.randn(n, hw, hw, c, generator=g).to(dt).cuda()
    w2 = (torch.randn(c, 3, 3, c, generator=g) / (9 * c) ** 0.5).to(dt).cuda()
    s2, h2 = (torch.rand(c, generator=g) + 0.5).cuda(), torch.randn(c, generator=g).cuda()
    w3 = (torch.randn(4 * c, 1, 1, c, generator=g) / c ** 0.5).to(dt).cuda()
    s3, h3 = (torch.rand(4 * c, generator=g) + 0.5).cuda(), torch.randn(4 * c, generator=g).cuda()
    res = torch.randn(n, hw, hw, 4 * c, generator=g).to(dt).cuda()
    y2 = hip.conv2d_nhwc_bn_act(y1, w2, s2, h2, None, relu=True, stride=1, pad=1)
    if which.startswith("c3"):
        assert hip.conv3x3_direct_supported(hw, hw, c, c)
        got, ref = hip.conv3x3_direct_nhwc(y1, w2, s2, h2), y2
    elif which == "s3_conv2":
        assert hip.bottleneck_tail_s3_supported(hw, hw, c)
        got, ref = hip.bottleneck_tail_s3_nhwc(y1, hip.pack_mfma_fragments(w2), s2, h2), y2
    else:
        ref = hip.conv2d_nhwc_bn_act(y2, w3, s3, h3, res, relu=True, stride=1, pad=0)
        if which == "s3":
            got = hip.bottleneck_tail_s3_nhwc(y1, hip.pack_mfma_fragments(w2), s2, h2, hip.pack_mfma_fragments(w3.view(4 * c, c)), s3, h3, res)
        elif which == "tail_56_s1":
            got = hip.bottleneck_tail_s1_nhwc(y1, hip.pack_mfma_fragments(w2), s2, h2, hip.pack_mfma_fragments(w3.view(4 * c, c)), s3, h3, res)
        elif which == "tail_56_s1f":                  # + the next bottleneck's conv1 on the output tile (both outputs compared)
            w1 = (torch.randn(64, 1, 1, 4 * c, generator=g) / (4 * c) ** 0.5).to(dt).cuda()
            s1, h1 = (torch.rand(64, generator=g) + 0.5).cuda(), torch.randn(64, generator=g).cuda()
            got, got1 = hip.bottleneck_tail_s1_nhwc(y1, hip.pack_mfma_fragments(w2), s2, h2, hip.pack_mfma_fragments(w3.view(4 * c, c)), s3, h3, res,
                                                    hip.pack_mfma_fragments(w1.view(64, 4 * c)), s1, h1, 64)
            ref1 = hip.conv2d_nhwc_bn_act(ref, w1, s1, h1, relu=True)
            got, ref = torch.cat([got.reshape(-1), got1.reshape(-1)]), torch.cat([ref.reshape(-1), ref1.reshape(-1)])
        elif which == "tail_28_s2f":                  # + the next bottleneck's conv1 (512 -> 128) on the output chunks (both outputs compared)
            w1 = (torch.randn(128, 1, 1, 4 * c, generator=g) / (4 * c) ** 0.5).to(dt).cuda()
            s1, h1 = (torch.rand(128, generator=g) + 0.5).cuda(), torch.randn(128, generator=g).cuda()
            got, got1 = hip.bottleneck_tail_s2_nhwc(y1, hip.pack_mfma_fragments(w2), s2, h2, hip.pack_mfma_fragments(w3.view(4 * c, c)), s3, h3, res,
                                                    hip.pack_mfma_fragments(w1.view(128, 4 * c)), s1, h1, 128)
            ref1 = hip.conv2d_nhwc_bn_act(ref, w1, s1, h1, relu=True)
            got, ref = torch.cat([got.reshape(-1), got1.reshape(-1)]), torch.cat([ref.reshape(-1), ref1.reshape(-1)])
        elif which == "tail_28_s2":
            got = hip.bottleneck_tail_s2_nhwc(y1, hip.pack_mfma_fragments(w2), s2, h2, hip.pack_mfma_fragments(w3.view(4 * c, c)), s3, h3, res)
        else:
            got = hip.bottleneck_tail_nhwc(y1, w2, s2, h2, w3, s3, h3, res)
    torch.cuda.synchronize()
    rec.update(bit_equal=bool(torch.equal(got, ref)), ok=bool(torch.equal(got, ref)))
    return rec


def wreg_trial(rng, idx):
    """The round-4 streaming / patch-resident kernels against the tile GEMM on random shapes, bit for bit, with canaries around the
    outputs: dh_conv1x1_wreg_nhwc, dh_conv1x1_dual_wreg_nhwc (both strides, odd input grids), dh_conv3x3_s4_nhwc (odd image counts)."""
    g = torch.Generator().manual_seed(70000 + idx)
    dt = rng.choice([torch.bfloat16, torch.float16])
    which = rng.choice(["c1", "c1", "dual", "dual", "s4"])
    rec = dict(kind="wreg", which=which, dt=str(dt)[6:])
    if which == "c1":
        cin, cout = rng.choice([256, 512, 1024]), 128 * rng.choice([1, 2, 4, 8, 16])
        n, hw = rng.randint(1, 24), rng.choice([7, 14, 28, 31, 56])
        while n * hw * hw < 8192:
            n += 1
        relu = rng.random() < 0.7
        x = torch.randn(n, hw, hw, cin, generator=g).to(dt).cuda()
        w = (torch.randn(cout, 1, 1, cin, generator=g) / cin ** 0.5).to(dt).cuda()
        scale, shift = (torch.rand(cout, generator=g) + 0.5).cuda(), torch.randn(cout, generator=g).cuda()
        rec.update(N=n, HW=hw, Cin=cin, Cout=cout, relu=relu)
        if not hip.conv1x1_wreg_supported(n * hw * hw, cin, cout):
            return dict(rec, ok=True, skipped=True)
        res = torch.randn(n, hw, hw, cout, generator=g).to(dt).cuda() if (cin == 512 and rng.random() < 0.5) else None
        rec["res"] = res is not None
        ref = hip.conv2d_nhwc_bn_act(x, w, scale, shift, residual=res, relu=relu)
        out, check = guarded((n, hw, hw, cout), dt)
        hip._launch("dh_conv1x1_wreg_nhwc", hip._ptr(x), hip._ptr(hip.pack_mfma_fragments(w.view(cout, cin))), hip._ptr(scale), hip._ptr(shift),
                    hip._ptr(res), hip._ptr(out), n * hw * hw, cin, cout, int(relu), hip._dt(x), hip._stream())
    elif which == "dual":
        c1, c2 = rng.choice([(64, 64), (128, 256), (256, 512), (64, 320), (192, 192)])
        cout, stride = 256 * rng.choice([1, 2, 4]), rng.choice([1, 2])
        ho = rng.choice([14, 27, 28, 56, 61])
        n = rng.randint(1, 12)
        while n * ho * ho < 8192:
            n += 1
        h = (ho - 1) * stride + 1 + rng.randint(0, 1)
        y = torch.randn(n, ho, ho, c1, generator=g).to(dt).cuda()
        x = torch.randn(n, h, h, c2, generator=g).to(dt).cuda()
        w = (torch.randn(cout, c1 + c2, generator=g) / (c1 + c2) ** 0.5).to(dt).cuda()
        shift = torch.randn(cout, generator=g).cuda()
        rec.update(N=n, Ho=ho, H=h, C1=c1, C2=c2, Cout=cout, stride=stride)
        if not hip.conv1x1_dual_wreg_supported(y.shape, x.shape, cout):
            return dict(rec, ok=True, skipped=True)
        ref = hip.conv1x1_dual_nhwc(y, x, w, shift, stride, relu=True)
        out, check = guarded((n, ho, ho, cout), dt)
        hip._launch("dh_conv1x1_dual_wreg_nhwc", hip._ptr(y), hip._ptr(x), hip._ptr(hip.pack_mfma_fragments(w)), hip._ptr(shift), hip._ptr(out),
                    n, ho, ho, c1, h, h, c2, stride, cout, 1, None, None, None, None, 0, hip._dt(y), hip._stream())
    else:
        n = rng.choice([1, 2, 3, rng.randint(4, 70)])
        x = torch.randn(n, 7, 7, 512, generator=g).to(dt).cuda()
        w = (torch.randn(512, 3, 3, 512, generator=g) / 4608 ** 0.5).to(dt).cuda()
        scale, shift = (torch.rand(512, generator=g) + 0.5).cuda(), torch.randn(512, generator=g).cuda()
        rec.update(N=n)
        ref = hip.conv2d_nhwc_bn_act(x, w, scale, shift, relu=True, stride=1, pad=1)
        out, check = guarded((n, 7, 7, 512), dt)
        hip._launch("dh_conv3x3_s4_nhwc", hip._ptr(x), hip._ptr(hip.pack_mfma_fragments(w)), hip._ptr(scale), hip._ptr(shift), hip._ptr(out),
                    n, 7, 7, 512, hip._dt(x), hip._stream())
    torch.cuda.synchronize()
    eq = bool(torch.equal(out, ref))
    rec.update(bit_equal=eq, canary_ok=check(), ok=bool(eq and check()))
    return rec


def stem_trial(rng, idx):
    g = torch.Generator().manual_seed(60000 + idx)
    dt = rng.choice([torch.bfloat16, torch.float16])
    n = rng.choice([1, 2, rng.randint(3, 12)])
    h, w = rng.choice([(224, 224), (rng.randint(8, 300), rng.randint(8, 300))])
    rec = dict(kind="stem", dt=str(dt)[6:], N=n, H=h, W=w)
    x = torch.randn(n, 3, h, w, generator=g).cuda()
    wt = (torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5).cuda()
    scale, shift = (torch.rand(64, generator=g) + 0.5).cuda(), torch.randn(64, generator=g).cuda()
    if ((h - 1) // 2 + 1) % 2 or ((w - 1) // 2 + 1) % 2:      # the direct stem takes even convolution outputs (encoders.py: else the general path)
        h, w = h + 2 * (((h - 1) // 2 + 1) % 2), w + 2 * (((w - 1) // 2 + 1) % 2)
        x = torch.randn(n, 3, h, w, generator=g).cuda()
        rec.update(H=h, W=w)
    wpk = hip.pack_stem_weight(wt, dt)
    a = hip.stem_conv7_bn_relu_maxpool(x, wpk, scale, shift)
    b = hip.stem_conv7_bn_relu_maxpool(hip.pack_nchw_to_nhwc8(x, out_dtype=dt), wpk, scale, shift)
    torch.cuda.synchronize()
    xr, wr = x.to(dt).float(), wt.to(dt).float()
    want = F.conv2d(xr, wr, stride=2, padding=3) * scale[None, :, None, None] + shift[None, :, None, None]
    want = F.max_pool2d(want.relu(), 3, 2, 1)
    ok_shape = tuple(a.shape) == (n, want.shape[2], want.shape[3], 64)
    err = float((a.float().permute(0, 3, 1, 2) - want).abs().max()) if ok_shape else float("inf")
    rec.update(shape_ok=ok_shape, err=err, formats_equal=bool(torch.equal(a, b)), ok=bool(ok_shape and err <= tol(dt, want) and torch.equal(a, b)))
    return rec


def fp32_trial(rng, idx):
    g = torch.Generator().manual_seed(70000 + idx)
    ks = rng.choice([1, 3, 7])
    stride = rng.choice([1, 2])
    pad = rng.choice([ks // 2, 0])
    n, cin, cout = rng.randint(1, 6), rng.randint(1, 70), rng.randint(1, 70)
    h, w = rng.randint(ks, 40), rng.randint(ks, 40)
    x = torch.randn(n, cin, h, w, generator=g).cuda()
    wt = (torch.randn(cout, cin, ks, ks, generator=g) / (ks * ks * cin) ** 0.5).cuda()
    scale, shift = (torch.rand(cout, generator=g) + 0.5).cuda(), torch.randn(cout, generator=g).cuda()
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    relu = rng.random() < 0.5
    res = torch.randn(n, cout, ho, wo, generator=g).cuda() if rng.random() < 0.4 else None
    out, check = guarded((n, cout, ho, wo), torch.float32)
    hip.conv2d_bn_act(x, wt, scale, shift, res, relu=relu, stride=stride, pad=pad, out=out)
    torch.cuda.synchronize()
    want = F.conv2d(x.cpu(), wt.cpu(), stride=stride, padding=pad) * scale.cpu()[None, :, None, None] + shift.cpu()[None, :, None, None]
    if res is not None:
        want = want + res.cpu()
    if relu:
        want = want.relu()
    err = float((out.cpu() - want).abs().max())
    return dict(kind="fp32", N=n, Cin=cin, Cout=cout, H=h, W=w, KS=ks, stride=stride, pad=pad, err=err, canary_ok=check(),
                ok=bool(err <= 2e-4 * max(1.0, float(want.abs().max())) and check()))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", default=None)
    args = ap.parse_args(argv)
    torch.backends.cudnn.allow_tf32 = False
    bad = 0
    for i in range(args.trials):
        for fn in (nhwc_trial, direct_trial, wreg_trial, stem_trial, fp32_trial):
            if args.only and args.only not in fn.__name__:
                continue
            if (fn in (direct_trial, stem_trial) and i % 3) or (fn is wreg_trial and i % 2):
                continue
            rng = random.Random(args.seed * 100003 + i)
            try:
                rec = fn(rng, i)
            except Exception as e:
                rec = {"kind": fn.__name__, "ok": False, "error": f"{type(e).__name__}: {e}"[:400]}
            bad += (not rec["ok"])
            print(json.dumps(dict(i=i, **rec)), flush=True)
    print(json.dumps({"trials": args.trials, "failures": bad}), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
