"""Randomised sweep of the 16-bit GEMM entry points on the GPU box, across the shape ranges where the dispatchers switch kernels:

  linear : ``dh_linear`` (a [M,K] @ w[N,K]^T, optional bias / scale+shift / ReLU / residual, 16-bit or fp32 output, strided operands)
           against fp32 math on the same rounded operands;
  vocab  : ``dh_vocab_logits`` (fp32 logits + 64-column group maxima; the A-stationary 128 / 256-row kernels, the tile kernels, with
           and without bias, padded and unpadded row strides) -- logits bit-equal to ``dh_linear``'s fp32 output, group maxima equal
           to the maxima of those logits;
every output sits in a buffer with canary margins on both sides and between rows (row stride > N): a single out-of-bounds or
row-padding write fails the trial (the padding of a logits row is scratch by contract and only reported).  TEST INFRASTRUCTURE.

    python tools/fuzz_gemm.py --trials 300 > gpurun_out/fuzz_gemm.jsonl
"""
import argparse
import json
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from deephumor_amd import hip          # noqa: E402

CANARY = 12345.0


def guarded(rows, cols, ld, dtype, margin=256):
    """A [rows, cols] view with row stride ``ld`` inside a canary-filled buffer; returns (view, check())."""
    buf = torch.full((margin + rows * ld + margin,), CANARY, dtype=dtype, device="cuda")
    view = buf[margin:margin + rows * ld].view(rows, ld)[:, :cols]

    def check(padding=True):
        ok = bool((buf[:margin] == CANARY).all()) and bool((buf[margin + rows * ld:] == CANARY).all())
        if ld > cols and padding:
            ok = ok and bool((buf[margin:margin + rows * ld].view(rows, ld)[:, cols:] == CANARY).all())
        return ok
    return view, check


def pick_dims(rng):
    m = rng.choice([rng.randint(1, 64), rng.randint(65, 700), rng.randint(701, 3000), rng.randint(3001, 60000), 1280, 256 * rng.randint(1, 8)])
    n = rng.choice([rng.randint(1, 64), rng.randint(65, 700), rng.randint(701, 5000), 8 * rng.randint(1, 300), 512, 2048])
    k = 8 * rng.choice([rng.randint(1, 16), rng.randint(17, 128), 64, 128, 256, 288, 8 * rng.randint(1, 40)])
    while m * n > (1 << 26) or m * k > (1 << 25):
        m = max(1, m // 2)
    return m, n, k


def linear_trial(rng, idx):
    g = torch.Generator().manual_seed(20000 + idx)
    dt = rng.choice([torch.bfloat16, torch.float16])
    m, n, k = pick_dims(rng)
    out_f32 = rng.random() < 0.3
    if not out_f32:
        n = max(8, n // 8 * 8)                     # 16-bit outputs: rows of whole 16-byte chunks (the models' layers)
    lda = k + 8 * rng.choice([0, 0, 1, 5])
    a = torch.randn(m, lda, generator=g).to(dt).cuda()[:, :k]
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dt).cuda()
    bias = torch.randn(n, generator=g).cuda() if rng.random() < 0.6 else None
    bn = rng.random() < 0.3 and bias is None
    scale = (torch.rand(n, generator=g) + 0.5).cuda() if bn else None
    shift = torch.randn(n, generator=g).cuda() if bn else None
    relu = rng.random() < 0.4
    res = torch.randn(m, n, generator=g).to(dt).cuda() if (rng.random() < 0.3 and not out_f32) else None
    ldc = n + (rng.choice([0, 0, 8, 64]) if not out_f32 else rng.choice([0, 0, 4, 64 - n % 64]))
    out, check = guarded(m, n, ldc, torch.float32 if out_f32 else dt)
    rec = dict(kind="linear", M=m, N=n, K=k, dt=str(dt)[6:], f32=out_f32, bias=bias is not None, bn=bn, relu=relu, res=res is not None,
               lda=lda, ldc=ldc)
    hip.linear(a, w, bias, scale, shift, relu=relu, out=out, residual=res)
    torch.cuda.synchronize()
    want = a.float() @ w.float().T
    if scale is not None:
        want = want * scale + shift
    if bias is not None:
        want = want + bias
    if res is not None:
        want = want + res.float()
    if relu:
        want = want.relu()
    err = float((out.float() - want).abs().max())
    tol = (3e-2 if dt == torch.bfloat16 else 4e-3) * max(1.0, float(want.abs().max())) if not out_f32 else 2e-3
    rec.update(err=err, canary_ok=check(), ok=bool(err <= tol and check() and bool(torch.isfinite(out.float()).all())))
    return rec


def vocab_trial(rng, idx):
    g = torch.Generator().manual_seed(30000 + idx)
    dt = rng.choice([torch.bfloat16, torch.float16])
    m = rng.choice([rng.randint(1, 300), 256 * rng.randint(1, 8), 1280, rng.randint(301, 2600), 5 * rng.randint(1, 300), 256 * rng.randint(9, 44)])
    v = rng.choice([rng.randint(2, 200), rng.randint(201, 3000), rng.randint(3001, 12000), rng.randint(12001, 40000), 36541])
    k = rng.choice([512, 512, 512, 256, 64 * rng.randint(2, 10)])
    while m * v > (1 << 26):
        m = max(1, m // 2)
    a = torch.randn(m, k, generator=g).to(dt).cuda()
    w = (torch.randn(v, k, generator=g) * (2.5 / k ** 0.5)).to(dt).cuda()
    bias = torch.randn(v, generator=g).cuda() if rng.random() < 0.8 else None
    ldl = rng.choice([(v + 127) // 128 * 128, (v + 127) // 128 * 128, (v + 63) // 64 * 64, (v + 3) // 4 * 4, v])
    ng = hip.n_groups(v)
    logits, check_l = guarded(m, v, ldl, torch.float32)
    ldg = ng + rng.choice([0, 0, 3])
    gmax, check_g = guarded(m, ng, ldg, torch.float32)
    rec = dict(kind="vocab", M=m, V=v, K=k, dt=str(dt)[6:], bias=bias is not None, ldl=ldl, ldg=ldg)
    hip.vocab_logits(a, w, bias, logits, gmax)
    torch.cuda.synchronize()
    ref = hip.linear(a, w, bias, out_dtype=torch.float32)
    bit = bool(torch.equal(logits, ref))
    pad = torch.full((m, ng * 64), float("-inf"), device="cuda")
    pad[:, :v] = ref
    gm_want = pad.view(m, ng, 64).max(-1).values
    gm_ok = bool(torch.equal(gmax, gm_want))
    # (the padding of a logits row is scratch by contract -- include/deephumor_hip.h; the margins and the group rows are not)
    rec.update(bit_equal=bit, gmax_ok=gm_ok, canary_ok=check_l(padding=False) and check_g(), logits_padding_untouched=check_l(),
               ok=bool(bit and gm_ok and check_l(padding=False) and check_g()))
    if not bit:
        rec["max_diff"] = float((logits - ref).abs().max())
    return rec


def vocab_wreg_trial(rng, idx):
    """dh_vocab_logits_wreg (round 4: weights streamed from L2 into registers) against dh_linear on the same operands: logits bit-equal,
    group maxima exact, padding columns = copies of logit[V - 1], nothing outside the buffers -- every row count it takes, random
    vocabularies, strides wider than the chunk padding."""
    g = torch.Generator().manual_seed(35000 + idx)
    dt = rng.choice([torch.bfloat16, torch.float16])
    m = 80 * rng.choice([1, 2, 4, 8, 16, 32])
    v = rng.choice([rng.randint(2, 300), rng.randint(301, 3000), rng.randint(3001, 12000), rng.randint(12001, 40000), 36541])
    while m * v > (1 << 26):
        m //= 2
    k = 512
    a = torch.randn(m, k, generator=g).to(dt).cuda()
    w = (torch.randn(v, k, generator=g) * (2.5 / k ** 0.5)).to(dt).cuda()
    bias = torch.randn(v, generator=g).cuda() if rng.random() < 0.8 else None
    vpad = (v + 255) // 256 * 256
    ldl, ng = vpad + rng.choice([0, 0, 4, 256]), vpad // 64
    logits, check_l = guarded(m, vpad, ldl, torch.float32)
    ldg = ng + rng.choice([0, 0, 3])
    gmax, check_g = guarded(m, ng, ldg, torch.float32)
    rec = dict(kind="vocab_wreg", M=m, V=v, dt=str(dt)[6:], bias=bias is not None, ldl=ldl, ldg=ldg)
    if not hip.vocab_logits_wreg_supported(m, v, k, ldl, ldg):
        return dict(rec, ok=False, error="supported() refused a shape of its contract")
    wp, bp = hip.pack_vocab_weights(w, bias)
    hip.vocab_logits_wreg(a, wp, bp, v, logits, gmax)
    torch.cuda.synchronize()
    ref = hip.linear(a, w, bias, out_dtype=torch.float32)
    bit = bool(torch.equal(logits[:, :v], ref)) and bool((logits[:, v:] == ref[:, v - 1:v]).all())
    pad = torch.full((m, ng * 64), float("-inf"), device="cuda")
    pad[:, :v] = ref
    gm_ok = bool(torch.equal(gmax, pad.view(m, ng, 64).max(-1).values))
    rec.update(bit_equal=bit, gmax_ok=gm_ok, canary_ok=check_l() and check_g(), ok=bool(bit and gm_ok and check_l() and check_g()))
    return rec


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=150)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", choices=["linear", "vocab", "wreg"], default=None)
    args = ap.parse_args(argv)
    bad = 0
    for i in range(args.trials):
        for fn in (linear_trial, vocab_trial, vocab_wreg_trial):
            if args.only and args.only not in fn.__name__:
                continue
            if fn is vocab_wreg_trial and i % 3:
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
