import sys, torch, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from helpers import KINDS, captions_and_lengths, golden, synthetic_sd, synth_images
import deephumor_amd.models as M
images = synth_images(4, seed=0)
cap, lengths, labels = captions_and_lengths()
for kind in KINDS:
    g = golden(f"g2g3_{kind}.npz")
    sd, hp = synthetic_sd(kind)
    model = getattr(M, kind)(**hp).eval(); model.load_state_dict(sd); model = model.cuda().bfloat16()
    with torch.no_grad():
        args = (images.cuda(), cap.cuda(), lengths) + ((labels.cuda(),) if "WithLabels" in kind else ())
        out = model(*args).float().cpu()
        ref = torch.from_numpy(g["forward_logits01"])
        d = (out[:2] - ref).abs()
        top_ref = ref.argmax(-1); top = out[:2].argmax(-1)
        gargs = (images.cuda(), labels.cuda()) if "WithLabels" in kind else (images.cuda(),)
        toks, lens = model.generate_batch(*gargs, max_len=32, beam_size=1, top_k=1)
        match = []
        for i in range(4):
            want = g[f"greedy_{i}"].tolist(); got = toks[i, :int(lens[i])].cpu().tolist()
            n = 0
            for a, b in zip(want, got):
                if a != b: break
                n += 1
            match.append(n)
        bt, bl = model.generate_batch(*gargs, max_len=32, beam_size=5, top_k=50, seed=1)
    print(kind, "logit maxabs %.3f mean %.4f ref_std %.2f argmax agree %.3f" % (d.max(), d.mean(), ref.std(), (top == top_ref).float().mean()),
          "greedy common prefix", match, "beam ok", tuple(bt.shape), int(bl.min()))
