"""Stand-alone timing of dh_decode_gemm_chain (1 - 4 phases) against the same GEMMs as separate dh_linear_ln_wreg launches, six rotating
weight sets as the six layers of a position.  NOTE: the separate launches are issued from Python at ~10 us per call, so their column is
HOST-bound here (their GPU time: profiles/r5/c3_bf16_kernel_stats.csv: 5.7 + 8.6 + 11.1 + 7.7 us at 1,280 rows); the chain is one call,
its column is GPU time above ~10 us.  Round-5 result (one box): chain 33.9 / 43.8 / 54.0 us at 160 / 380 / 1,280 rows; with the GEMM blocks
compiled out the protocol alone 15.5 / 18.1 / 24.3 us (1.7 - 2 us per phase on top of the launch)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephumor_amd import hip  # noqa: E402


def main():
    d, pf, dev = 512, 2048, "cuda"
    r = lambda *s, sc=1.0: (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)
    for m in (160, 380, 1280):
        att, x = r(m, d), r(m, d)
        sets = []
        for _ in range(6):
            w_o, w_1, w_2, w_q = r(d, d, sc=d ** -0.5), r(pf, d, sc=d ** -0.5), r(d, pf, sc=pf ** -0.5), r(3 * d, d, sc=d ** -0.5)
            sets.append(dict(pk={k: hip.pack_mfma_fragments(v) for k, v in (("o", w_o), ("1", w_1), ("2", w_2), ("q", w_q))},
                             cs1=w_1.float().sum(1).contiguous(), csq=w_q.float().sum(1).contiguous()))
        b_o, b_1, b_2, b_q = [torch.zeros(n, device=dev) for n in (d, pf, d, 3 * d)]
        gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
        st_x = torch.stack([torch.zeros(m, 8, device=dev), torch.full((m, 8), 64.0, device=dev)], -1).contiguous()
        o, ff, xq, xo = torch.empty_like(x), torch.empty((m, pf), dtype=x.dtype, device=dev), torch.empty((m, 3 * d), dtype=x.dtype, device=dev), torch.empty_like(x)
        st_o, st_2 = torch.empty((m, 8, 2), device=dev), torch.empty((m, 8, 2), device=dev)
        sync = torch.zeros(80, dtype=torch.int32, device=dev)

        def steps(i):
            s = sets[i % 6]
            return [dict(a=att, w_packed=s["pk"]["o"], n=d, bias=b_o, out=o, residual=x, r_ln=(st_x, 1e-5, gam, bet), o_stats=st_o),
                    dict(a=o, w_packed=s["pk"]["1"], n=pf, bias=b_1, out=ff, relu=True, a_ln=(st_o, 1e-5, s["cs1"])),
                    dict(a=ff, w_packed=s["pk"]["2"], n=d, bias=b_2, out=xo, residual=o, r_ln=(st_o, 1e-5, gam, bet), o_stats=st_2),
                    dict(a=xo, w_packed=s["pk"]["q"], n=3 * d, bias=b_q, out=xq, a_ln=(st_2, 1e-5, s["csq"]))]

        def fused(i):
            hip.decode_gemm_chain(steps(i), m, sync)

        def separate(i):
            for s_ in steps(i):
                if s_.get("residual") is not None:
                    hip.linear_ln_wreg(s_["a"], s_["w_packed"], s_["n"], s_["bias"], out=s_["out"], residual=s_["residual"], r_ln=s_["r_ln"])
                else:
                    hip.linear_ln_wreg(s_["a"], s_["w_packed"], s_["n"], s_["bias"], out=s_["out"], relu=bool(s_.get("relu")), a_ln=s_["a_ln"])

        def timeit(fn, n=60):
            for i in range(6):
                fn(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n * 1e3
        print(f"rows {m}: separate {timeit(separate):.1f} us, chain {timeit(fused):.1f} us", flush=True)
        for k in (1, 2, 3):
            print(f"   first {k} step(s): separate {timeit(lambda i: [None for _ in [0]] and separate_k(i, k, steps)):.1f} us, chain {timeit(lambda i: hip.decode_gemm_chain(steps(i)[:k], m, sync)):.1f} us", flush=True)


def separate_k(i, k, steps):
    for s_ in steps(i)[:k]:
        if s_.get("residual") is not None:
            hip.linear_ln_wreg(s_["a"], s_["w_packed"], s_["n"], s_["bias"], out=s_["out"], residual=s_["residual"], r_ln=s_["r_ln"])
        else:
            hip.linear_ln_wreg(s_["a"], s_["w_packed"], s_["n"], s_["bias"], out=s_["out"], relu=bool(s_.get("relu")), a_ln=s_["a_ln"])


if __name__ == "__main__":
    main()
