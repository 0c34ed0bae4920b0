"""Round 6: the wide 1 x 1 layers of the fp32 trunk on the split-operand path -- dh_conv2d_nhwc_f32x (128 x 128 tiles) against
dh_conv1x1_f32x_stream (persistent, weights in registers, double-buffered activation blocks, residual prefetched): equality and us per
launch at 256 images.  DH_F32X_DIAG (tile kernel only; wrong results, timing only): 1 = no output stores, 2 = no residual loads,
4 = no residual prefetch -- the phase decomposition quoted in csrc/conv1x1_f32x.hip."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephumor_amd import hip, f32xp  # noqa: E402


def timeit(fn, n=8):
    for i in range(2):
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
    n, bad, t_tile, t_stream = 256, 0, 0.0, 0.0
    diag = os.environ.get("DH_F32X_DIAG", "0")
    for name, hw, cin, cout, res, count in (("l1 conv3", 56, 64, 256, True, 3), ("l1.0 downsample", 56, 64, 256, False, 1), ("l2 conv3", 28, 128, 512, True, 4),
                                             ("l3 conv3", 14, 256, 1024, True, 6), ("l1 conv1", 56, 256, 64, False, 2), ("l2 conv1", 28, 512, 128, False, 3)):
        x = torch.randn(n, hw, hw, cin, device="cuda").relu_()
        w = torch.randn(cout, cin, device="cuda") * cin ** -0.5
        sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda") * 0.1
        wp = hip.split_f32x(w)
        r = torch.randn(n, hw, hw, cout, device="cuda") if res else None
        t0 = timeit(lambda i: hip.conv2d_nhwc_f32x(x, wp, 1, sc, sh, residual=r))
        gb = (x.numel() + n * hw * hw * cout * (2 if res else 1)) * 4 / 1e9
        line = f"diag={diag} {name:16s} {cin:4d} -> {cout:4d}: tiles {t0:7.1f} us ({gb / t0 * 1e3:.2f} TB/s of {gb:.2f} GB)"
        pk = f32xp.pack_conv1x1(wp)
        if pk is not None and f32xp.conv1x1_stream_supported(n * hw * hw, cin, cout):
            same = torch.equal(f32xp.conv1x1_stream(x, pk, sc, sh, residual=r), hip.conv2d_nhwc_f32x(x, wp, 1, sc, sh, residual=r)) if diag == "0" else None
            t1 = timeit(lambda i: f32xp.conv1x1_stream(x, pk, sc, sh, residual=r))
            bad += same is False
            t_tile += count * t0
            t_stream += count * t1
            line += f" | streaming {t1:7.1f} us ({gb / t1 * 1e3:.2f} TB/s) | equal {same}"
        print(line, flush=True)
        del x, r
    print(f"the {3 + 1 + 4 + 6} launches of a ResNet-50 trunk the streaming kernel takes: tiles {t_tile:.0f} us | streaming {t_stream:.0f} us")
    print("MISMATCHES", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
