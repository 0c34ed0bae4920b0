import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephumor_amd import hip
hip.load()
n, c, hw = 256, 256, 14
dt = torch.bfloat16
y1 = [torch.randn(n, hw, hw, c, device="cuda").to(dt) for _ in range(3)]
w2 = (torch.randn(c, 3, 3, c, device="cuda") * (9 * c) ** -0.5).to(dt)
w3 = (torch.randn(4 * c, 1, 1, c, device="cuda") * c ** -0.5).to(dt)
s2, h2 = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda") * 0.3
s3, h3 = torch.rand(4 * c, device="cuda") + 0.5, torch.randn(4 * c, device="cuda") * 0.3
res = [torch.randn(n, hw, hw, 4 * c, device="cuda").to(dt) for _ in range(3)]
w2p, w3p = hip.pack_mfma_fragments(w2), hip.pack_mfma_fragments(w3.reshape(4 * c, c).contiguous())
def timeit(fn, iters=30, warm=5):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    with hip.profile() as prof:
        for i in range(iters): fn(i)
        torch.cuda.synchronize()
    return {k: round(v["ms"] / v["calls"] * 1e3, 1) for k, v in prof.summary().items()}
print("s3 conv2 only ", timeit(lambda i: hip.bottleneck_tail_s3_nhwc(y1[i % 3], w2p, s2, h2)))
print("s3 fused tail ", timeit(lambda i: hip.bottleneck_tail_s3_nhwc(y1[i % 3], w2p, s2, h2, w3p, s3, h3, res[i % 3])))
print("implicit 3x3  ", timeit(lambda i: hip.conv2d_nhwc_bn_act(y1[i % 3], w2, s2, h2, relu=True, stride=1, pad=1)))
y2 = hip.conv2d_nhwc_bn_act(y1[0], w2, s2, h2, relu=True, stride=1, pad=1)
print("implicit 1x1  ", timeit(lambda i: hip.conv2d_nhwc_bn_act(y2, w3, s3, h3, residual=res[i % 3], relu=True, stride=1, pad=0)))
