"""What happens with ids a user must not pass (TEST INFRASTRUCTURE; GPU box): caption-prefix / teacher-forced tokens and labels outside
their table, LSTM lengths outside the padded sequence -- every case in its OWN process (round 5: some of them were GPU memory faults).
The reference raises IndexError / RuntimeError; since round 5 so does this package (tests/test_models_gpu.py).  One JSON line per case.

    python tools/bad_inputs_probe.py
"""
import sys, os, json, subprocess
CASES = ["gen_prefix_big", "gen_prefix_neg", "fwd_big", "fwd_neg", "fwd_huge", "label_big", "label_neg", "fwd_len_too_long", "fwd_len_zero", "gen_ok"]
CHILD = r'''
import sys, os, json
import torch
sys.path.insert(0, %(root)r)
import deephumor_amd.models as M
from deephumor_amd.synth import load_synthetic, synth_images
kind, case = %(kind)r, %(case)r
V = 300
model = load_synthetic(getattr(M, kind)(V).eval(), seed=1234).cuda()
images = synth_images(3, seed=1).cuda()
lab = (torch.randint(4, V, (3, 2)).cuda(),) if "WithLabels" in kind else ()
good = torch.randint(4, V, (3, 5)).cuda()
lengths = torch.tensor([5, 4, 2])
big = good.clone(); big[1, 2] = V + 100
neg = good.clone(); neg[0, 1] = -1
huge = good.clone(); huge[2, 0] = 2 ** 31 + 5
def run():
    if case == "gen_prefix_big": return model.generate_batch(images, *lab, caption=big[:, :3], max_len=6, beam_size=3, top_k=20, seed=1)
    if case == "gen_prefix_neg": return model.generate_batch(images, *lab, caption=neg[:, :3], max_len=6, beam_size=3, top_k=20, seed=1)
    if case == "fwd_big": return model(images, big, lengths, *lab)
    if case == "fwd_neg": return model(images, neg, lengths, *lab)
    if case == "fwd_huge": return model(images, huge, lengths, *lab)
    if case == "label_big":
        bl = lab[0].clone(); bl[0, 0] = V + 7
        return model.generate_batch(images, bl, max_len=4, beam_size=1, top_k=1, seed=1)
    if case == "label_neg":
        nl = lab[0].clone(); nl[1, 1] = -3
        return model.generate_batch(images, nl, max_len=4, beam_size=1, top_k=1, seed=1)
    if case == "fwd_len_too_long": return model(images, good, torch.tensor([9, 4, 2]), *lab)
    if case == "fwd_len_zero": return model(images, good, torch.tensor([5, 0, 2]), *lab)
    return model.generate_batch(images, *lab, max_len=4, beam_size=3, top_k=20, seed=1)
try:
    with torch.no_grad():
        out = run()
    torch.cuda.synchronize()
    if isinstance(out, tuple): print("RES returned max_tok=%%d min_tok=%%d" %% (int(out[0].max()), int(out[0].min())))
    else: print("RES returned finite=%%s shape=%%s" %% (bool(torch.isfinite(out.float()).all()), list(out.shape)))
except Exception as e:
    print("RES raised " + type(e).__name__ + ": " + str(e)[:100])
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for kind in ("CaptioningLSTM", "CaptioningTransformerWithLabels"):
    for case in CASES:
        if case.startswith("label") and "WithLabels" not in kind:
            continue
        p = subprocess.run([sys.executable, "-c", CHILD % dict(root=root, kind=kind, case=case)], capture_output=True, text=True, timeout=300)
        res = [l for l in p.stdout.splitlines() if l.startswith("RES ")]
        fault = "Memory access fault" in p.stderr
        print(json.dumps(dict(kind=kind, case=case, rc=p.returncode, res=res[-1][4:] if res else None, gpu_fault=fault)), flush=True)
