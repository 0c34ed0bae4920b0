#!/usr/bin/env python3
"""Stand-alone timings of the decode-position kernels at the C3 shapes (1280 rows, D 512, PF 2048, 8 heads, 49 patches),
through the C-ABI, with six rotating weight sets (as the six layers of a position) so the weights are not L2-resident.
Developer tool (A/B of kernel variants inside ONE gpurun call -- MI355X boxes differ by ~10 %); not part of the product."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephumor_amd import hip  # noqa: E402

DT = {"bf16": torch.bfloat16, "f16": torch.float16}[os.environ.get("KB_DTYPE", "bf16")]
R, D, PF, H, S, NIMG, BEAM = 1280, 512, 2048, 8, 49, 256, 5
dev = "cuda"


def rnd(*shape, scale=1.0, dtype=DT):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)


def timeit(fn, iters=60, warm=6):
    """GPU-side time per launch: HIP events recorded INSIDE the library around every launch (not host-bound)."""
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    with hip.profile() as prof:
        for i in range(iters):
            fn(i)
        torch.cuda.synchronize()
    summ = prof.summary()
    return sum(v["ms"] for v in summ.values()) / iters * 1e3


def main():
    hip.load()
    res = {}
    x = [rnd(R, D) for _ in range(6)]
    ff = [rnd(R, PF) for _ in range(6)]
    st = [torch.stack([torch.zeros(R, 8, device=dev), torch.full((R, 8), 64.0, device=dev)], -1).contiguous() for _ in range(6)]
    gamma, beta = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    for name, n, k in (("qkv", 3 * D, D), ("proj", D, D), ("ffn1", PF, D), ("ffn2", D, PF)):
        w = [rnd(n, k, scale=k ** -0.5) for _ in range(6)]
        b = torch.zeros(n, device=dev)
        cs = torch.zeros(n, device=dev)
        a = x if k == D else ff
        out = torch.empty(R, n, device=dev, dtype=DT)
        res[f"{name}:linear"] = timeit(lambda i: hip.linear(a[i % 6], w[i % 6], b, out=out, relu=(name == "ffn1")))
        res[f"{name}:linear_ln(plain)"] = timeit(lambda i: hip.linear_ln(a[i % 6], w[i % 6], b, out=out))
        if k == D:
            res[f"{name}:linear_ln(a_ln)"] = timeit(lambda i: hip.linear_ln(a[i % 6], w[i % 6], b, out=out, a_ln=(st[i % 6], 1e-5, cs)))
            if hip.linear_ln_wreg_supported(n, k, False):
                wp = [hip.pack_mfma_fragments(t) for t in w]
                res[f"{name}:WREG(a_ln)"] = timeit(lambda i: hip.linear_ln_wreg(a[i % 6], wp[i % 6], n, b, out=out, a_ln=(st[i % 6], 1e-5, cs), relu=(name == "ffn1")))
        if n == D:
            res[f"{name}:linear(+res)"] = timeit(lambda i: hip.linear(a[i % 6], w[i % 6], b, out=out, residual=x[(i + 1) % 6]))
            stats_out = torch.empty(R, 8, 2, device=dev)

            def f(i):
                fl = hip.LnFold()
                fl.r_stats, fl.r_tiles, fl.r_eps, fl.r_gamma, fl.r_beta = st[i % 6].data_ptr(), 8, 1e-5, gamma.data_ptr(), beta.data_ptr()
                fl.o_stats = stats_out.data_ptr()
                hip._launch("dh_linear_ln", a[i % 6].data_ptr(), k, w[i % 6].data_ptr(), k, b.data_ptr(), x[(i + 1) % 6].data_ptr(), D,
                            out.data_ptr(), n, R, n, k, 0, hip._c.byref(fl), hip._dt(out), hip._stream())
            res[f"{name}:linear_ln(r_ln+stats)"] = timeit(f)
            wp2 = [hip.pack_mfma_fragments(t) for t in w]
            res[f"{name}:WREG(r_ln+stats)"] = timeit(lambda i: hip.linear_ln_wreg(a[i % 6], wp2[i % 6], n, b, out=out, residual=x[(i + 1) % 6],
                                                                                  r_ln=(st[i % 6], 1e-5, gamma, beta)))
    # padded row strides (power-of-two row strides of 1 KB / 4 KB put the 8 rows of an LDS-DMA piece on few L2 channels?)
    for name, n, k in (("proj", D, D), ("ffn2", D, PF), ("ffn1", PF, D)):
        for pad in (0, 64, 32):
            a = [torch.randn(R, k + pad, device=dev).to(DT)[:, :k] for _ in range(6)]
            w = [(torch.randn(n, k + pad, device=dev) * k ** -0.5).to(DT)[:, :k] for _ in range(6)]
            b = torch.zeros(n, device=dev)
            out = torch.empty(R, n, device=dev, dtype=DT)
            res[f"{name}:linear ld=K+{pad}"] = timeit(lambda i: hip.linear(a[i % 6], w[i % 6], b, out=out))
    o = torch.empty(R, D, device=dev, dtype=DT)
    res["add_layernorm"] = timeit(lambda i: hip.add_layernorm(x[i % 6], x[(i + 1) % 6], gamma, beta, out=o))
    # cross attention
    kv = [rnd(NIMG * S, 2 * D) for _ in range(6)]
    mask = torch.zeros(NIMG * S, dtype=torch.uint8, device=dev)
    packed = [hip.attn_cross_pack(t, NIMG, S, D, H) for t in kv]
    res["cross:lds"] = timeit(lambda i: hip.attn_cross_decode(x[i % 6], kv[i % 6], mask, o, NIMG, BEAM, S, D, H, 8.0))
    res["cross:mfma"] = timeit(lambda i: hip.attn_cross_decode_packed(x[i % 6], packed[i % 6][0], packed[i % 6][1], mask, o, NIMG, BEAM, S, D, H, 8.0))
    packed_d = [hip.attn_cross_pack(t, NIMG, S, D, H, dperm=True) for t in kv]
    res["cross:mfma (dperm slots)"] = timeit(lambda i: hip.attn_cross_decode_packed(x[i % 6], packed_d[i % 6][0], packed_d[i % 6][1], mask, o, NIMG, BEAM, S, D, H,
                                                                                  8.0, dperm=True))
    # self attention at t = 16 and 31
    qkv = [rnd(R, 3 * D) for _ in range(6)]
    kc = [rnd(33, R, D) for _ in range(6)]
    vc = [rnd(33, R, D) for _ in range(6)]
    src = (torch.arange(R, device=dev, dtype=torch.int32) // BEAM * BEAM)[:, None].expand(R, 33).contiguous()
    tokens = torch.full((R, 32), 7, dtype=torch.int32, device=dev)
    for t in (8, 16, 31):
        res[f"self:t={t}"] = timeit(lambda i: hip.attn_self_decode(qkv[i % 6], kc[i % 6], vc[i % 6], src, tokens, o, NIMG, BEAM, 1, R, t, D, H, 8.0, 0))
    for k_, v_ in res.items():
        print(f"{k_:34s} {v_:8.2f} us")


if __name__ == "__main__":
    main()
