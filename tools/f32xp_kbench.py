"""Round 6: the split-operand classifier on an activation stored as planes (csrc/gemm_f32xp.hip: both operands by LDS-DMA, three slabs
deep, two wave groups a phase apart) against dh_linear_f32x (fp32 activation split in registers, one slab of look-ahead) -- bit-equality
and us per launch.  (The convolution form of the same kernel was measured with this tool and not kept: profiles/r6/f32xp_kbench_*.txt.)"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephumor_amd import hip, f32xp  # noqa: E402


def timeit(fn, n=20):
    for i in range(3):
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
    torch.manual_seed(0)
    bad = 0
    # ---- the classifier: 1,280 rows x V = 36,541 x 512 -------------------------------------------------------------------------------
    for m, v, k in ((1280, 36541, 512), (256, 36541, 512), (1280, 4096, 512), (300, 1000, 96)):
        a = torch.randn(m, k, device=dev)
        w = torch.randn(v, k, device=dev) * k ** -0.5
        b = torch.randn(v, device=dev)
        wp = hip.split_f32x(w)
        ld = (v + 255) // 256 * 256
        out0 = torch.empty(m, ld, device=dev)[:, :v]
        out1 = torch.zeros(m, ld, device=dev)[:, :v]
        gm = torch.zeros(m, hip.n_groups(v), device=dev)
        ap = f32xp.split_act(a)
        hip.linear_f32x(a, wp, b, out=out0)
        f32xp.linear(ap, wp, b, out=out1, group_max=gm)
        same = torch.equal(out0, out1)
        pad = torch.full((m, gm.shape[1] * 64 - v), float("-inf"), device=dev)
        gm_ref = torch.cat([out0, pad], 1).view(m, -1, 64).amax(-1)
        same_g = torch.equal(gm, gm_ref)
        bad += (not same) + (not same_g)
        t0 = timeit(lambda i: hip.linear_f32x(a, wp, b, out=out0))
        t1 = timeit(lambda i: f32xp.linear(ap, wp, b, out=out1, group_max=gm))
        t2 = timeit(lambda i: f32xp.split_act(a))
        flop = 3 * 2.0 * m * v * k
        print(f"linear {m:5d} x {v:6d} x {k:4d}: f32x {t0:7.1f} us | planes {t1:7.1f} us ({flop / t1 / 1e6:6.0f} TF of MFMA work) + split {t2:5.1f} us"
              f" | equal {same} groups {same_g}", flush=True)
    print("overflow word", hip.f32x_take_overflow())
    print("MISMATCHES", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
