"""Teacher-forced forward() logits throughput (bf16 models): rows/s through the classifier."""
import sys, time, torch
sys.path.insert(0, '.')
import deephumor_amd.models as M
from deephumor_amd.synth import load_synthetic, synth_images
m = load_synthetic(M.CaptioningLSTM(36541).eval(), seed=1).cuda().bfloat16()
imgs = synth_images(256, seed=0).cuda()
caps = torch.randint(6, 36541, (256, 31)).cuda()
with torch.no_grad():
    for _ in range(2): out = m(imgs, caps, None)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): out = m(imgs, caps, None)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print("forward 256 x 32 positions:", dt * 1e3, "ms", tuple(out.shape), out.stride())
