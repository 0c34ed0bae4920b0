import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from deephumor_amd.pipeline import CaptionPipeline
from deephumor_amd.synth import synth_images
dev = torch.device("cuda", 0)
for wl in ("c3", "c2"):
    model, sd, hp = bench.build_model(wl, dev, "bf16")
    imgs = synth_images(256, seed=0)
    dimgs = imgs.to(dev)
    pinned = imgs.pin_memory()
    kw = dict(max_len=32, beam_size=5, top_k=50, temperature=1.0)
    with torch.no_grad():
        ref, _ = model.generate_batch(dimgs, seed=7, **kw)
    for name, overlap, src, to_host in (("seq dev", False, dimgs, False), ("ovl dev", True, dimgs, False), ("seq host", False, pinned, True), ("ovl host", True, pinned, True)):
        pipe = CaptionPipeline(model, overlap=overlap, **kw)
        outs = [(t.clone(), l.clone()) for t, l in pipe.run([(src,)] * 3, seeds=[7, 8, 9], to_host=to_host)]
        assert torch.equal(outs[0][0].to(dev), ref), name
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for toks, lens in pipe.run([(src,)] * n, seeds=range(100, 100 + n), to_host=to_host):
            pass
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(wl, name, round(dt * 1e3, 2), "ms/step", round(256 / dt), "captions/s")
