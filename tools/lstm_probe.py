"""Developer probe: the fused LSTM layer step at the decode shape (1280 beam rows, Hh 512, E 256 / 512) -- the 64 x 64 tile kernel
(dh_lstm_layer_fused) against the register-stationary one (dh_lstm_layer_wreg), token gather + parent gather as in the decode loop."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephumor_amd import hip
hip.load()
dt = torch.bfloat16
rows, hh = 1280, 512
def timeit(fn, iters=40, warm=5):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    with hip.profile() as prof:
        for i in range(iters): fn(i)
        torch.cuda.synchronize()
    return {k: round(v["ms"] / v["calls"] * 1e3, 1) for k, v in prof.summary().items()}
for e in (256, 512):
    g = torch.Generator().manual_seed(e)
    w = (torch.randn(4 * hh, e + hh, generator=g) * 0.05).to(dt)
    w_il = w.view(4, hh, -1).permute(1, 0, 2).reshape(4 * hh, -1).contiguous().cuda()
    b_il = (torch.randn(4 * hh, generator=g) * 0.1).cuda()
    w_pk = hip.pack_mfma_fragments(w_il)
    emb = torch.randn(36541, e, generator=g).to(dt).cuda()
    tokens = torch.randint(0, 36541, (rows, 32), generator=g, dtype=torch.int32).cuda()
    x_rows = torch.randn(rows, e, generator=g).to(dt).cuda()
    hs = [(torch.randn(rows, hh, generator=g) * 0.5).to(dt).cuda() for _ in range(2)]
    cs = [torch.randn(rows, hh, generator=g).cuda() for _ in range(2)]
    hparent = ((torch.arange(rows) // 5) * 5 + torch.randint(0, 5, (rows,), generator=g)).to(torch.int32).cuda()
    h_out = torch.zeros(rows, hh, dtype=dt, device="cuda")
    tok = e == 256
    for name, fn, wt in (("fused", hip.lstm_layer_fused, w_il), ("wreg ", hip.lstm_layer_wreg, w_pk)):
        r = timeit(lambda i: fn(None if tok else x_rows, 1, emb if tok else None, tokens if tok else None, 3 + i % 8,
                                hs[i % 2], cs[i % 2], hparent, hs[1 - i % 2], cs[1 - i % 2], h_out, wt, b_il, rows, 1, e, hh))
        print(f"E={e} {name}", r)
