"""Developer probe: C3 decode as ONE captured graph over 256 images vs TWO graphs over 128 images each replayed
concurrently on two HIP streams (do two half-size latency-bound chains overlap?)."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from deephumor_amd.synth import synth_images
dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
model, sd, hp = bench.build_model(wl, dev, "bf16")
imgs = synth_images(256, seed=0).to(dev)
kw = dict(max_len=32, beam_size=5, top_k=50, temperature=1.0)


def graph_for(x, slot):
    saved = model.__dict__.get("_graphs")
    model.__dict__["_graphs"] = {}
    model.generate_batch_graphed(x, seed=1, **kw)
    (state,) = model.__dict__["_graphs"].values()
    model.__dict__["_graphs"] = saved if saved is not None else {}
    return state


def timeit(fn, n=6):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


full = graph_for(imgs, 0)
print(wl, "one graph, 256 images:", round(timeit(lambda: full[0].replay()), 2), "ms")
for parts in (2, 4):
    n = 256 // parts
    states = [graph_for(imgs[i * n:(i + 1) * n].contiguous(), i) for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]

    def both():
        for st, s in zip(states, streams):
            with torch.cuda.stream(s):
                st[0].replay()
    print(wl, f"{parts} graphs of {n} images on {parts} streams:", round(timeit(both), 2), "ms")
    print(wl, f"{parts} graphs of {n} images, one stream:", round(timeit(lambda: [st[0].replay() for st in states]), 2), "ms")
    del states
