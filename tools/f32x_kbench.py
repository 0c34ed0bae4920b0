"""Round 6: dh_linear_f32x (64 x 64 / 128 x 128 tiles, one round trip per 32-k slab) against dh_linear_f32x_wreg (split weights
stationary in registers) at the shapes of one decode position, back-to-back launches on six rotating operand sets, us per launch."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephumor_amd import hip  # noqa: E402


def timeit(fn, n=120):
    for i in range(12):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = "cuda"
    for m in (1280, 160, 380, 3000):
        for name, n, k in (("fc_o/fc_q", 512, 512), ("qkv", 1536, 512), ("fc_1", 2048, 512), ("fc_2", 512, 2048), ("lstm l0", 2048, 768), ("lstm l1", 2048, 1024)):
            a = [torch.randn(m, k, device=dev) for _ in range(6)]
            w = [torch.randn(n, k, device=dev) * k ** -0.5 for _ in range(6)]
            b = torch.zeros(n, device=dev)
            planes = [hip.split_f32x(x) for x in w]
            packed = [hip.pack_f32x_fragments(x) for x in planes]
            out = torch.empty(m, n, device=dev)
            t_tile = timeit(lambda i: hip.linear_f32x(a[i % 6], planes[i % 6], b, out=out))
            t_wreg = timeit(lambda i: hip.linear_f32x_wreg(a[i % 6], packed[i % 6], b, out=out))
            t_f32 = timeit(lambda i: hip.linear(a[i % 6], w[i % 6], b, out=out))
            print(f"rows {m:5d} {name:10s} {n:5d} x {k:5d}: exact fp32 {t_f32:7.1f} us | f32x tile {t_tile:7.1f} us | f32x wreg {t_wreg:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
