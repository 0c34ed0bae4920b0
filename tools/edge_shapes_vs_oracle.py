"""Edge shapes through the fp32 HIP path and through the CPU oracle (TEST INFRASTRUCTURE; GPU box): teacher-forced forward with empty /
one-token / full captions and every kind of ``lengths``, generate with max_len 1 / 2 / 30, prefixes as long as or longer than max_len, a
prefix holding <eos> -- same result, or the same kind of exception?  One JSON line per case ("same" / "both raise" / "DIFFERENT" /
"MISMATCH").  Round 5: everything equal except the LSTM prefix >= max_len case (fixed); the "topk == V" lines sample with different
random streams and are not comparable.

    python tools/edge_shapes_vs_oracle.py
"""
import sys, os, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_path as R
import deephumor_amd.models as M
from deephumor_amd.synth import synth_state_dict, synth_images
def both(label, f_hip, f_ref, cmp):
    res = {}
    for name, f in (("hip", f_hip), ("ref", f_ref)):
        try:
            with torch.no_grad():
                res[name] = ("ok", f())
        except Exception as e:
            res[name] = ("exc", type(e).__name__ + ": " + str(e)[:70])
    if res["hip"][0] == "ok" and res["ref"][0] == "ok":
        verdict = "same" if cmp(res["hip"][1], res["ref"][1]) else "DIFFERENT"
        detail = None if verdict == "same" else (str(res["hip"][1])[:80], str(res["ref"][1])[:80])
    elif res["hip"][0] == "exc" and res["ref"][0] == "exc":
        verdict, detail = "both raise", (res["hip"][1], res["ref"][1])
    else:
        verdict, detail = "MISMATCH", (res["hip"][0] + " " + str(res["hip"][1])[:90], res["ref"][0] + " " + str(res["ref"][1])[:90])
    print(json.dumps(dict(case=label, verdict=verdict, detail=detail)), flush=True)
for kind in ("CaptioningLSTM", "CaptioningTransformer", "CaptioningTransformerBase"):
    V = 60
    model = getattr(M, kind)(V).eval()
    sd = synth_state_dict(model.state_dict(), seed=1234)
    model.load_state_dict(sd)
    hp = model._hp
    model = model.cuda()
    imgs = synth_images(2, seed=3)
    g = torch.Generator().manual_seed(1)
    cap = torch.randint(4, V, (2, 5), generator=g)
    close = lambda a, b: tuple(a.shape) == tuple(b.shape) and float((a.cpu().float() - b.float()).abs().max()) < 2e-3
    ids = lambda a, b: [int(x) for x in a] == [int(x) for x in b]
    fw = lambda c, l: both(f"{kind} forward cap{list(c.shape)} len={l}", lambda: model(imgs.cuda(), c.cuda(), None if l is None else torch.tensor(l)),
                           lambda: R.model_forward(kind, sd, hp, imgs, c, None if l is None else torch.tensor(l)), close)
    fw(cap, [5, 3]); fw(cap, None); fw(cap[:, :1], [1, 1]); fw(cap[:, :0], None); fw(cap, [6, 6]) ; fw(cap, [1, 1]); fw(cap, [2, 5])
    def gen(label, **kw):
        def h():
            t, l = model.generate_batch(imgs[:1].cuda(), **kw)
            return t[0, :int(l[0])].cpu().tolist()
        both(f"{kind} generate {label}", h, lambda: R.model_generate(kind, sd, hp, imgs[:1], **kw).reshape(-1).tolist(), ids)
    gen("greedy len1", max_len=1, beam_size=1, top_k=1)
    gen("greedy len2", max_len=2, beam_size=1, top_k=1)
    gen("greedy len30", max_len=30, beam_size=1, top_k=1)
    gen("prefix len==max", caption=cap[:1, :4], max_len=4, beam_size=1, top_k=1)
    gen("prefix longer than max", caption=cap[:1, :5], max_len=3, beam_size=1, top_k=1)
    gen("prefix with eos", caption=torch.tensor([[7, 3, 9]]), max_len=6, beam_size=1, top_k=1)
    gen("topk == V", max_len=4, beam_size=1, top_k=V)
    gen("beam == topk == 1 eos likely", max_len=8, beam_size=1, top_k=1, eos_index=int(cap[0, 0]))
