"""The RCCL side of the sharded path on ONE GPU (the pool has no multi-GPU node): a one-rank ``nccl`` process group through which
``deephumor_amd.dist`` and ``bench.timed_region`` run their collectives.  Each case runs in a child process (a process group is
per-process state)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = r"""
import datetime, json, os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
import socket
with socket.socket() as _s:                      # a free port per run: two concurrent runs on one box must not collide
    _s.bind(("127.0.0.1", 0))
    _port = _s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT") or str(_port))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
from deephumor_amd.dist import generate_sharded, score_sharded
from deephumor_amd.models import CaptioningLSTM
from deephumor_amd.synth import load_synthetic, synth_images
model = load_synthetic(CaptioningLSTM(1000), seed=7).to(dev).eval()
images = synth_images(6, seed=0).to(dev)
kw = dict(max_len=10, beam_size=3, top_k=20, seed=11)
with torch.no_grad():
    want = model.generate_batch(images, img0=0, **kw)
    got = generate_sharded(lambda lo, hi: model.generate_batch(images[lo:hi], img0=lo, **kw), 6, always=True)
    rows = torch.randn(6, 5, device=dev)
    back = score_sharded(lambda lo, hi: rows[lo:hi], 6, always=True)
import bench
from deephumor_amd import hip
hip.set_option("dist_always", 1)
dt, out = bench.timed_region(lambda s: s + 1, 3, 1, dev)
print("RESULT " + json.dumps({"backend": dist.get_backend(), "ids": bool(torch.equal(want[0], got[0]) and torch.equal(want[1], got[1])),
                              "rows": bool(torch.equal(rows, back)), "timed_out": out, "dt_ok": dt > 0,
                              "rank_times": len(bench.RANK_TIMES) == 1 and abs(bench.RANK_TIMES[0] - dt) < 1e-9}))
dist.barrier()
dist.destroy_process_group()
"""


def test_sharded_generate_through_one_rank_rccl():
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[7:])
    assert res == {"backend": "nccl", "ids": True, "rows": True, "timed_out": 3, "dt_ok": True, "rank_times": True}


def test_bench_rccl_single_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rccl-single", "--quick", "--workload", "c2", "--batch", "32",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])      # the LAST stdout line (RCCL's banner must not follow it)
    assert line["n_ranks_seen"] == 1 and "rccl_single_rank" in line and line["value"] > 0
    assert line["per_rank_ms_per_step"]["ranks"] == 1 and line["per_rank_ms_per_step"]["max"] == pytest.approx(line["ms_per_step"])
