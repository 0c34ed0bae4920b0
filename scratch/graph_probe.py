import sys, time, torch
sys.path.insert(0, '.')
import bench
from deephumor_amd.synth import synth_images
dev = torch.device('cuda')
for wl in ('c3', 'c2'):
    model, sd, hp = bench.build_model(wl, dev, 'bf16')
    images = synth_images(256, seed=0).to(dev)
    def run():
        return model.generate_batch(images, max_len=32, beam_size=5, top_k=50, temperature=1.0, seed=1)
    with torch.no_grad():
        for _ in range(2): toks, lens = run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): toks, lens = run()
        torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 3
        import deephumor_amd.models.beam as B
        orig = B.BeamSearchHelper.check
        B.BeamSearchHelper.check = lambda self: None
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            run()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        t0 = time.perf_counter()
        with torch.cuda.graph(g):
            gt, gl = run()
        torch.cuda.synchronize(); cap = time.perf_counter() - t0
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): g.replay()
        torch.cuda.synchronize(); graphed = (time.perf_counter() - t0) / 3
        B.BeamSearchHelper.check = orig
        print(wl, 'eager ms', eager * 1e3, 'graph ms', graphed * 1e3, 'capture s', cap, 'same tokens', bool((gt == toks).all()))
    del model
