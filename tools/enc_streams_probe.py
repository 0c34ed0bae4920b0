#!/usr/bin/env python3
"""Round-6 probe: does the 16-bit encoder gain from running its batch as 2 / 4 image sub-batches on separate HIP streams?  (Its stage
tails run an MFMA-bound 3x3 phase and an HBM-bound 1x1 phase in lockstep across all workgroups; independent sub-batches de-phase them.)
Wall time per 256 images, median of 12, one process: 1 stream x 256, 2 x 128, 4 x 64, and 2 x 128 back to back on ONE stream."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from deephumor_amd.models import ImageEncoder  # noqa: E402
from deephumor_amd.synth import synth_images, synth_state_dict  # noqa: E402


def main():
    enc = ImageEncoder(256, spatial_features=False).eval()
    enc.load_state_dict(synth_state_dict(enc.state_dict(), seed=1234))
    enc = enc.cuda().bfloat16()
    imgs = synth_images(256, seed=0).cuda()
    streams = [torch.cuda.Stream() for _ in range(4)]
    cur = torch.cuda.current_stream()

    def run(parts, serial=False):
        n = 256 // parts
        if serial or parts == 1:
            return [enc(imgs[i * n:(i + 1) * n]) for i in range(parts)]
        outs = []
        for i in range(parts):
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                outs.append(enc(imgs[i * n:(i + 1) * n]))
        for i in range(parts):
            cur.wait_stream(streams[i])
        return outs

    with torch.no_grad():
        ref = torch.cat(run(1))
        for parts, serial in ((1, False), (2, False), (4, False), (2, True), (1, False)):
            for _ in range(3):
                out = run(parts, serial)
            torch.cuda.synchronize()
            same = bool(torch.equal(torch.cat(out), ref))
            ts = []
            for _ in range(12):
                t0 = time.perf_counter()
                run(parts, serial)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            ts.sort()
            print(f"encoder 256 images as {parts} x {256 // parts}{' (one stream)' if serial else ''}: median {ts[6] * 1e3:.3f} ms, min {ts[0] * 1e3:.3f} ms, "
                  f"bit-identical to one batch: {same}", flush=True)


if __name__ == "__main__":
    main()
