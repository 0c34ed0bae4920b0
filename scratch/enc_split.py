import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephumor_amd.models import ImageEncoder
from deephumor_amd.synth import synth_images, synth_state_dict
enc = ImageEncoder(256).eval()
enc.load_state_dict(synth_state_dict(enc.state_dict(), seed=1234))
enc = enc.cuda().bfloat16()
imgs = synth_images(256, seed=0).cuda()
def run_full():
    return enc(imgs)
streams = [torch.cuda.Stream() for _ in range(4)]
def run_split(k):
    cur = torch.cuda.current_stream()
    outs = []
    n = 256 // k
    for i in range(k):
        streams[i].wait_stream(cur)
        with torch.cuda.stream(streams[i]):
            outs.append(enc(imgs[i * n:(i + 1) * n]))
    for i in range(k):
        cur.wait_stream(streams[i])
    return torch.cat(outs)
with torch.no_grad():
    ref = run_full()
    for name, fn in (("full", run_full), ("2 streams", lambda: run_split(2)), ("4 streams", lambda: run_split(4)), ("full", run_full), ("2 streams", lambda: run_split(2))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"{name:10s} median {ts[5]*1e3:.3f} ms  min {ts[0]*1e3:.3f}  equal {bool(torch.equal(out, ref))}")
