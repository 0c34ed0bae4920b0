"""Does running the ResNet trunk in image sub-batches (activations resident in L2 / Infinity Cache) beat one big batch?"""
import sys, torch
sys.path.insert(0, '.')
from deephumor_amd.models.encoders import ImageEncoder
from deephumor_amd.synth import load_synthetic, synth_images
enc = ImageEncoder(256).eval()
load_synthetic(enc, seed=0)
enc = enc.cuda().bfloat16()
imgs = synth_images(256, seed=2).cuda()
def run(chunk):
    outs = [enc(imgs[i:i + chunk]) for i in range(0, 256, chunk)]
    return torch.cat(outs)
ref = run(256)
for chunk in (256, 128, 64, 32, 16, 8):
    for _ in range(2): o = run(chunk)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): o = run(chunk)
    e1.record(); torch.cuda.synchronize()
    print(f"chunk {chunk:4d}: {e0.elapsed_time(e1)/5:7.3f} ms  maxdiff {float((o.float()-ref.float()).abs().max()):.3g}")
