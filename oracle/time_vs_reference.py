"""Timing of the oracle (``oracle/ref_path.py``, what ``bench.py``'s ``cpu_baseline`` leg runs on the GPU box's host)
against the REAL reference (``/root/reference``, imported through the torchvision stand-in) on the same inputs, weights
and thread count, in the BUILD CONTAINER ("restatement-vs-reference timing ratio").
TEST INFRASTRUCTURE ONLY; cannot run on the GPU box (no reference there).

    python oracle/time_vs_reference.py      # writes profiles/r3/oracle_vs_reference_timing.json
"""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "_standin"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

from deephumor.models import CaptioningLSTM, CaptioningTransformer      # noqa: E402
from deephumor_amd.synth import load_synthetic, synth_images           # noqa: E402
from oracle import ref_path as R                                        # noqa: E402

V, MAX_LEN, BEAM, TOP_K = 36541, 32, 5, 50


def best_of_interleaved(fa, fb, rounds=5):
    """Minimum over ``rounds`` of each callable, the two measured alternately (the build container's 8 vCPUs are shared: a
    burst of foreign load then hits both, not one of them)."""
    fa(), fb()
    ta, tb = [], []
    for _ in range(rounds):
        t0 = time.perf_counter(); fa(); ta.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); fb(); tb.append(time.perf_counter() - t0)
    return min(ta), min(tb)


def main():
    torch.set_num_threads(8)
    imgs = synth_images(1, seed=0)
    out = {"host": f"build container, {os.cpu_count()} vCPU, torch {torch.__version__}, {torch.get_num_threads()} threads",
           "settings": f"V={V}, max_len={MAX_LEN}, beam={BEAM}, top_k={TOP_K}, 1 image per generate, warm-up + best of 5, reference and oracle alternating"}
    for kind, cls in (("CaptioningLSTM", CaptioningLSTM), ("CaptioningTransformer", CaptioningTransformer)):
        ref = load_synthetic(cls(V).eval(), seed=1234)
        sd = {k: v.clone() for k, v in ref.state_dict().items()}
        hp = ref._hp
        with torch.no_grad():
            t_ref, t_orc = best_of_interleaved(lambda: ref.generate(imgs, max_len=MAX_LEN, beam_size=BEAM, top_k=TOP_K),
                                               lambda: R.model_generate(kind, sd, hp, imgs, max_len=MAX_LEN, beam_size=BEAM, top_k=TOP_K))
            torch.manual_seed(3)
            a = ref.generate(imgs, max_len=MAX_LEN, beam_size=BEAM, top_k=TOP_K)
            torch.manual_seed(3)
            b = R.model_generate(kind, sd, hp, imgs, max_len=MAX_LEN, beam_size=BEAM, top_k=TOP_K)
        out[kind] = {"reference_s_per_caption": round(t_ref, 3), "oracle_s_per_caption": round(t_orc, 3),
                     "oracle_over_reference": round(t_orc / t_ref, 3), "same_ids_under_same_seed": bool(torch.equal(a.reshape(-1), b.reshape(-1)))}
        print(kind, out[kind])
    path = os.path.join(ROOT, "profiles", "r3", "oracle_vs_reference_timing.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
