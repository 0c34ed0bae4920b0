"""Round 6: per-phase time stamps (s_memrealtime, DH_DL_DEBUG=2) of layer 0 of the persistent decoder-layer kernel (csrc/decode_layers.hip) at the
BASELINE C3 shape, last position of a 256-image beam-5 decode: where the 143 us per layer go."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from deephumor_amd import hip
from deephumor_amd.synth import synth_images
import deephumor_amd.models.transformers as T
dev = torch.device("cuda", 0)
hip.set_option("decode_layers", 1)
model = bench.build_model("c3", dev, "bf16")[0]
imgs = synth_images(256, seed=0).to(dev)
runs = []
orig = T._IncrementalDecoder._Run.__init__
def patched(self, *a, **k):
    orig(self, *a, **k); runs.append(self)
T._IncrementalDecoder._Run.__init__ = patched
with torch.no_grad():
    for s in range(3):
        model.generate_batch(imgs, max_len=32, beam_size=5, top_k=50, seed=s)
torch.cuda.synchronize()
st = runs[-1].layers_sync[400:416].cpu().tolist()
st = [x & 0xFFFFFFFF for x in st]
names = ["start", "qkv gemm done", "self-attn done", "B1", "fc_o done", "B2", "fc_q+cross done", "B3", "enc fc_o done", "B4", "fc_1 done", "B5", "fc_2 done", "B6", "fc_q gemm done(14)"]
print("stamps (10 ns units) of layer 0, last position:")
base = st[0]
for i, n in enumerate(names):
    print(f"  {n:24s} {((st[i] - base) & 0xFFFFFFFF) * 0.01:8.2f} us")
