import sys, torch
sys.path.insert(0, '.')
from deephumor_amd import hip
N = 256
cfgs = []  # (name, H, Cin, Cout, ks, stride)
inpl, H = 64, 56
for planes, blocks, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)):
    for b in range(blocks):
        s = stride if b == 0 else 1
        if b == 0:
            cfgs.append((f"down{planes}", H, inpl, planes * 4, 1, s, 1))
        cfgs.append((f"c1_{planes}_{b}", H, inpl, planes, 1, 1, 1))
        cfgs.append((f"c2_{planes}_{b}", H, planes, planes, 3, s, 1))
        H2 = H // s
        cfgs.append((f"c3_{planes}_{b}", H2, planes, planes * 4, 1, 1, 1))
        inpl, H = planes * 4, H2
seen = {}
tot = 0
for name, h, cin, cout, ks, s, _ in cfgs:
    key = (h, cin, cout, ks, s)
    if key in seen:
        tot += seen[key]; continue
    x = torch.randn(N, h, h, cin, device='cuda').bfloat16()
    w = (torch.randn(cout, ks, ks, cin, device='cuda') * 0.05).bfloat16()
    sc, sh = torch.ones(cout, device='cuda'), torch.zeros(cout, device='cuda')
    pad = 1 if ks == 3 else 0
    res = torch.randn(N, h // s, h // s, cout, device='cuda').bfloat16() if name.startswith('c3') else None
    for _ in range(2): y = hip.conv2d_nhwc_bn_act(x, w, sc, sh, residual=res, stride=s, pad=pad)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): y = hip.conv2d_nhwc_bn_act(x, w, sc, sh, residual=res, stride=s, pad=pad)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    ho = y.shape[1]
    fl = 2.0 * N * ho * ho * cout * cin * ks * ks
    by = 2.0 * (x.numel() + y.numel() * (2 if res is not None else 1) + w.numel())
    seen[key] = ms; tot += ms
    print(f"{name:12s} H{h:3d} {cin:4d}->{cout:4d} k{ks} s{s}: {ms*1e3:7.1f} us  {fl/ms/1e9:7.1f} TF  {by/ms/1e6:7.1f} GB/s  M={N*ho*ho} K={cin*ks*ks}")
print("total ms", tot)
