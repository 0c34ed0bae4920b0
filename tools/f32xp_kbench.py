"""Round 6: the split-operand kernels on activations stored as planes (csrc/gemm_f32xp.hip: both operands by LDS-DMA, three slabs deep)
against dh_linear_f32x / dh_conv2d_nhwc_f32x (fp32 activations split in registers, one slab of look-ahead) -- bit-equality and us per
launch at the classifier's shape and at the encoder's convolution shapes (256 images)."""
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
    # ---- the encoder's convolutions, 256 images ------------------------------------------------------------------------------------------
    n = 256
    shapes = (("l1 conv1", 56, 256, 64, 1, 1, 0), ("l1 conv2", 56, 64, 64, 3, 1, 1), ("l1 conv3", 56, 64, 256, 1, 1, 0),
              ("l2.0 conv2 s2", 56, 128, 128, 3, 2, 1), ("l2 down", 56, 256, 512, 1, 2, 0),
              ("l2 conv1", 28, 512, 128, 1, 1, 0), ("l2 conv2", 28, 128, 128, 3, 1, 1), ("l2 conv3", 28, 128, 512, 1, 1, 0),
              ("l3 conv1", 14, 1024, 256, 1, 1, 0), ("l3 conv2", 14, 256, 256, 3, 1, 1), ("l3 conv3", 14, 256, 1024, 1, 1, 0),
              ("l4 conv1", 7, 2048, 512, 1, 1, 0), ("l4 conv2", 7, 512, 512, 3, 1, 1), ("l4 conv3", 7, 512, 2048, 1, 1, 0))
    tot0 = tot1 = 0.0
    for name, hw, cin, cout, ks, stride, pad in shapes:
        x = torch.randn(n, hw, hw, cin, device=dev).relu_()
        w = torch.randn(cout, ks * ks * cin, device=dev) * (ks * ks * cin) ** -0.5
        sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
        wp = hip.split_f32x(w)
        xp = f32xp.split_act(x.view(-1, cin)).view(2, n, hw, hw, cin)
        ho = (hw + 2 * pad - ks) // stride + 1
        res = torch.randn(n, ho, ho, cout, device=dev) if "conv3" in name else None
        y0 = hip.conv2d_nhwc_f32x(x, wp, ks, sc, sh, residual=res, stride=stride, pad=pad)
        y1, y1p = f32xp.conv2d_nhwc(xp, wp, ks, sc, sh, residual=res, stride=stride, pad=pad, want="both")
        y0p = f32xp.split_act(y0.view(-1, cout)).view_as(y1p)
        same = torch.equal(y0, y1) and torch.equal(y0p, y1p)
        bad += not same
        t0 = timeit(lambda i: hip.conv2d_nhwc_f32x(x, wp, ks, sc, sh, residual=res, stride=stride, pad=pad), n=8)
        t1 = timeit(lambda i: f32xp.conv2d_nhwc(xp, wp, ks, sc, sh, residual=res, stride=stride, pad=pad, want="planes" if res is None else "both"), n=8)
        flop = 3 * 2.0 * n * ho * ho * cout * ks * ks * cin
        tot0 += t0
        tot1 += t1
        print(f"{name:14s} {hw:3d}^2 x {cin:4d} -> {cout:4d} k{ks} s{stride}: f32x {t0:7.1f} us | planes {t1:7.1f} us ({flop / t1 / 1e6:6.0f} TF) | equal {same}",
              flush=True)
        del x, xp, y0, y1, y1p, y0p
    print(f"sum of the shapes: f32x {tot0:.0f} us | planes {tot1:.0f} us")
    xm = torch.randn(n, 112, 112, 64, device=dev)
    same = torch.equal(f32xp.maxpool3x3s2_nhwc(xm), f32xp.split_act(hip.maxpool3x3s2_nhwc_f32(xm).view(-1, 64)).view(2, n, 56, 56, 64))
    bad += not same
    print("maxpool -> planes equal", same, "| overflow word", hip.f32x_take_overflow())
    print("MISMATCHES", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
