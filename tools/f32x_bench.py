"""fp32 models at the BASELINE shape (256 images, beam 5, V = 36,541): ms per step of the exact-fp32 path and of the split-operand
matrix-core path (option f32_split), with the in-library per-kernel breakdown of the latter."""
import json
import os
import re
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from deephumor_amd import hip  # noqa: E402
from deephumor_amd.synth import synth_images  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    out = {}
    imgs = synth_images(256, seed=0).to(dev)
    for wl in sys.argv[1:] or ["c2", "c3"]:
        # exact fp32 | split operands on the tile kernels | + the register-stationary decode layers | + the operands stored split (planes)
        for split, wreg, planes in ((0, 1, 0), (1, 0, 0), (1, 1, 0), (1, 1, 1)):
            with hip.option_scope(f32_split=split, decode_wreg=wreg, f32_planes=planes), torch.no_grad():
                model = bench.build_model(wl, dev, "f32")[0]
                bench.one_step(model, imgs, 0, 256, seed=0)
                with hip.profile() as prof:
                    bench.one_step(model, imgs, 0, 256, seed=0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for s in range(3):
                    bench.one_step(model, imgs, 0, 256, seed=1 + s)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / 3 * 1e3
                by = {}
                for k, v in prof.summary().items():
                    base = re.sub(r"\{.*\}$", "", k)
                    e = by.setdefault(base, [0, 0.0])
                    e[0] += v["calls"]; e[1] += v["ms"]
                top = {k: [c, round(m, 2)] for k, (c, m) in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]}
                enc = bench.encoder_table(prof.summary(), "bf16" if split else "f32")[:12] if split else None
                out[f"{wl}_split{split}" + ("" if wreg else "_tile_kernels") + ("_planes" if planes else "")] = {"ms_per_step": round(ms, 2), "captions_per_s": round(256 / ms * 1e3, 1), "event_timed_ms": top, "encoder_rows": enc}
                del model
                torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
