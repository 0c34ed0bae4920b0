"""Developer probe: the round-4 encoder kernels stand-alone at their 256-image shapes (for tools/pmc_sq.sh): the streaming 1x1 / dual
layers (conv1x1_wreg.hip), the stage-1 tail + next conv1 (conv_s1.hip), the stage-2 tail (conv_s2.hip), the stage-4 3x3 (conv_s4.hip),
next to the kernels they replace."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephumor_amd import hip
hip.load()
dt, dev, n = torch.bfloat16, "cuda", 256
def r(*shape, s=1.0): return (torch.randn(*shape, device=dev) * s).to(dt)
def bn(c): return torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.3
def timeit(fn, iters=20, warm=3):
    for i in range(warm): fn()
    torch.cuda.synchronize()
    with hip.profile() as prof:
        for i in range(iters): fn()
        torch.cuda.synchronize()
    return {k: round(v["ms"] / v["calls"] * 1e3, 1) for k, v in prof.summary().items()}
# conv1 of stage 3 (K = 1024 -> 256) and of stage 2 (512 -> 128): streaming vs tile
for hw, cin, cout in ((14, 1024, 256), (28, 512, 128)):
    x, w = r(n, hw, hw, cin), r(cout, 1, 1, cin, s=cin ** -0.5); sc, sh = bn(cout); wp = hip.pack_mfma_fragments(w.view(cout, cin))
    print("conv1x1 wreg", hw, cin, cout, timeit(lambda: hip.conv1x1_wreg_nhwc(x, wp, cout, sc, sh)), "tile", timeit(lambda: hip.conv2d_nhwc_bn_act(x, w, sc, sh, relu=True)))
# stage-1 dual (+ next conv1) and stage-1 tail (+ next conv1)
y, x0, wd, shd = r(n, 56, 56, 64), r(n, 56, 56, 64), r(256, 128, s=128 ** -0.5), torch.randn(256, device=dev) * 0.3
w1, (s1, h1) = r(64, 1, 1, 256, s=1 / 16), bn(64)
wdp, w1p = hip.pack_mfma_fragments(wd), hip.pack_mfma_fragments(w1.view(64, 256))
print("dual wreg    ", timeit(lambda: hip.conv1x1_dual_wreg_nhwc(y, x0, wdp, 256, shd, 1)), "+conv1", timeit(lambda: hip.conv1x1_dual_wreg_nhwc(y, x0, wdp, 256, shd, 1, w1p=w1p, scale1=s1, shift1=h1, n1=64)),
      "tile", timeit(lambda: hip.conv1x1_dual_nhwc(y, x0, wd, shd, 1)))
for hw, c, name in ((56, 64, "s1"), (28, 128, "s2")):
    y1, w2, w3, res = r(n, hw, hw, c), r(c, 3, 3, c, s=(9 * c) ** -0.5), r(4 * c, 1, 1, c, s=c ** -0.5), r(n, hw, hw, 4 * c)
    (s2, h2), (s3, h3) = bn(c), bn(4 * c)
    w2p, w3p = hip.pack_mfma_fragments(w2), hip.pack_mfma_fragments(w3.view(4 * c, c))
    print(name, "ring tail", timeit(lambda: hip.bottleneck_tail_nhwc(y1, w2, s2, h2, w3, s3, h3, res)))
    if name == "s1":
        print(name, "strip tail + next conv1", timeit(lambda: hip.bottleneck_tail_s1_nhwc(y1, w2p, s2, h2, w3p, s3, h3, res, w1p, s1, h1, 64)))
    else:
        print(name, "strip tail", timeit(lambda: hip.bottleneck_tail_s2_nhwc(y1, w2p, s2, h2, w3p, s3, h3, res)))
x4, w4 = r(n, 7, 7, 512), r(512, 3, 3, 512, s=4608 ** -0.5); s4, h4 = bn(512); w4p = hip.pack_mfma_fragments(w4)
print("s4 3x3", timeit(lambda: hip.conv3x3_s4_nhwc(x4, w4p, s4, h4)), "tile", timeit(lambda: hip.conv2d_nhwc_bn_act(x4, w4, s4, h4, relu=True, stride=1, pad=1)))
