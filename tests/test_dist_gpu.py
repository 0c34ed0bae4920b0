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


GRAPH_CHILD = r"""
import datetime, json, os, socket, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
with socket.socket() as _s:
    _s.bind(("127.0.0.1", 0))
    _port = _s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT") or str(_port))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
from deephumor_amd.dist import gather_captions_async
from deephumor_amd.models import CaptioningTransformer
from deephumor_amd.synth import load_synthetic, synth_images
# (a 6-layer Transformer at 30 positions: ~1,600 launches per capture = a capture window of tens of ms; the watchdog polls every 100 ms)
model = load_synthetic(CaptioningTransformer(300), seed=7).to(dev).eval()
images = synth_images(3, seed=0).to(dev)
kw = dict(max_len=30, beam_size=3, top_k=20)
ok, captures = True, 0
with torch.no_grad():
    want = model.generate_batch(images, seed=11, **kw)
    for it in range(40):
        # works the watchdog thread keeps polling (hipEventQuery) while the next capture runs
        pend = [gather_captions_async(want[0], want[1], 3, always=True) for _ in range(3)]
        model.__dict__.pop("_graphs", None)                       # a fresh capture every iteration
        got = model.generate_batch_graphed(images, seed=11, **kw)
        captures += 1
        ok = ok and bool(torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]))
        for h in pend:
            t, l = h.wait()
            ok = ok and bool(torch.equal(t, want[0]))
print("RESULT " + json.dumps({"ok": ok, "captures": captures}))
dist.barrier()
dist.destroy_process_group()
"""


def test_hipgraph_capture_next_to_the_rccl_watchdog():
    """Capturing a decode graph while c10d's watchdog thread polls pending RCCL works: in HIP's default (global) capture mode one
    ``hipEventQuery`` of that thread inside the capture window fails the capture and aborts the process ("operation not permitted when
    stream is capturing"; ``bench.py --rccl-single`` died in 1 run of 25, a bare capture loop next to pending works within a few
    iterations).  ``generate_batch_graphed`` captures in thread-local mode: 40 captures of ~1,600 launches each with works pending, all equal to eager
    (with the default mode this test aborts)."""
    p = subprocess.run([sys.executable, "-c", GRAPH_CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res == {"ok": True, "captures": 40}


C4_CHILD = r"""
import datetime, json, os, socket, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
with socket.socket() as _s:
    _s.bind(("127.0.0.1", 0))
    _port = _s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT") or str(_port))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=300))
import deephumor_amd.models as M
from deephumor_amd import hip
from deephumor_amd.dist import generate_micro_sharded
from deephumor_amd.synth import synth_images, synth_state_dict
from oracle import ref_path as R
hip.set_option("dist_always", 1)                 # the one-rank group runs its all_gather per shard
V, N, SH = 36541, 2048, 8
model = M.CaptioningTransformer(V).eval()
sd = synth_state_dict(model.state_dict(), seed=1234)
model.load_state_dict(sd)
model = model.to(dev)
res = {"backend": dist.get_backend()}
spans = []
def shard(m, kw):
    def fn(lo, hi):
        spans.append((lo, hi))
        return m.generate_batch(synth_images(hi - lo, seed=0, first=lo).to(dev), img0=lo, **kw)
    return fn
with torch.no_grad():
    # fp32: the 2,048-image global batch as 8 sequential 256-image shards (img0 = 256 r), greedy, rows {0, 1000, 2047} vs the oracle
    toks, lens = generate_micro_sharded(shard(model, dict(max_len=32, beam_size=1, top_k=1)), N, SH)
    res["shape"] = list(toks.shape)
    res["spans"] = spans == [(256 * r, 256 * (r + 1)) for r in range(SH)]
    # rows {0, 1000, 2047} against the captions the REAL reference gave for those images (golden G18, oracle/make_golden.py r6; the
    # CPU oracle is pinned on the same file in tests/test_oracle_golden.py -- running it here cost 12 s of the GPU box's time)
    import numpy as np
    g18 = np.load(os.path.join(%(root)r, "tests", "golden", "g18_bench_rows_CaptioningTransformer.npz"))
    ok = True
    for i, key in ((0, "greedy_0"), (1000, "greedy_far_1000"), (2047, "greedy_far_2047")):
        ok = ok and toks[i, :int(lens[i])].cpu().tolist() == g18[key].tolist()
    res["fp32_rows_equal_oracle"] = ok
    # bf16, beam 5 (Philox keyed by the global image index): the sharded batch is repeatable, and shard 3 (images 768..1023) equals
    # the same images decoded as two 128-image halves -- a row does not depend on which shard / tile it sits in
    m16 = model.bfloat16()
    kw = dict(max_len=32, beam_size=5, top_k=50, temperature=1.0, seed=42)
    t1, l1 = generate_micro_sharded(shard(m16, kw), N, SH)
    t2, l2 = generate_micro_sharded(shard(m16, kw), N, SH)
    res["bf16_repeatable"] = bool(torch.equal(t1, t2) and torch.equal(l1, l2))
    ha = m16.generate_batch(synth_images(128, seed=0, first=768).to(dev), img0=768, **kw)
    hb = m16.generate_batch(synth_images(128, seed=0, first=896).to(dev), img0=896, **kw)
    res["bf16_shard_invariant"] = bool(torch.equal(torch.cat([ha[0], hb[0]]), t1[768:1024]) and torch.equal(torch.cat([ha[1], hb[1]]), l1[768:1024]))
    res["bf16_valid"] = bool(int(t1.max()) < V and not bool((t1 == 1).any()) and int(l1.min()) >= 1 and tuple(t1.shape) == (N, 32))
    # distinct images give distinct captions: shards are not copies of each other
    res["bf16_shards_differ"] = bool(not torch.equal(t1[:256], t1[256:512]))
print("RESULT " + json.dumps(res))
dist.barrier()
dist.destroy_process_group()
"""


def test_c4_global_batch_as_eight_sequential_shards():
    """BASELINE config C4 (CaptioningTransformer, batch 2,048 image-sharded over 8 ranks) on the one GPU this pool has: the 8 shards
    (``img0 = 256 r``) one after another through ``generate_micro_sharded`` and the one-rank RCCL group's per-shard all_gather --
    fp32 rows {0, 1000, 2047} equal the CPU oracle's greedy ids, the bf16 beam-5 batch is repeatable and shard-invariant."""
    p = subprocess.run([sys.executable, "-c", C4_CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res == {"backend": "nccl", "shape": [2048, 32], "spans": True, "fp32_rows_equal_oracle": True, "bf16_repeatable": True,
                   "bf16_shard_invariant": True, "bf16_valid": True, "bf16_shards_differ": True}


C5_CHILD = r"""
import sys, os, json, socket, datetime
sys.path.insert(0, %(root)r)
import numpy as np
import torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
with socket.socket() as _s:
    _s.bind(("127.0.0.1", 0))
    _port = _s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT") or str(_port))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=300))
import deephumor_amd.models as M
from deephumor_amd import hip
from deephumor_amd.dist import generate_micro_sharded, shard_range
from deephumor_amd.synth import synth_images, synth_state_dict
hip.set_option("dist_always", 1)                 # the one-rank group runs its all_gather per shard
V, N, SH = 36541, 300, 8
model = M.CaptioningTransformerWithLabels(V).eval()
model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234))
model = model.to(dev).half()
g = np.random.Generator(np.random.Philox(key=[1, 0]))
labels = torch.from_numpy(g.integers(6, V, size=(N, 3)).astype(np.int64)).to(dev)
images = synth_images(N, seed=2).to(dev)
kw = dict(max_len=32, beam_size=10, top_k=50, temperature=1.0, seed=7)
spans = []
def fn(lo, hi):
    spans.append((lo, hi))
    return model.generate_batch(images[lo:hi], labels[lo:hi], img0=lo, **kw)
res = {"backend": dist.get_backend()}
with torch.no_grad():
    toks, lens = generate_micro_sharded(fn, N, SH)
    one, one_l = model.generate_batch(images, labels, **kw)            # the whole sweep as ONE batch on this GPU
res["shape"] = list(toks.shape)
res["spans"] = spans == [shard_range(N, r, SH) for r in range(SH)]
res["shard_sizes"] = [b - a for a, b in spans]
res["rows_0_150_299_equal_single_batch"] = all(bool(torch.equal(toks[i], one[i]) and int(lens[i]) == int(one_l[i])) for i in (0, 150, 299))
res["all_rows_equal_single_batch"] = bool(torch.equal(toks, one) and torch.equal(lens, one_l))
print("RESULT " + json.dumps(res))
dist.barrier()
dist.destroy_process_group()
"""


def test_c5_sweep_as_eight_uneven_shards():
    """BASELINE config C5 (ImageLabelEncoder + CaptioningTransformer, fp16, beam 10, 300 templates over 8 ranks) on the one GPU this
    pool has: the eight UNEVEN shards (38 x 4 + 37 x 4, ``img0`` = the shard's first template) one after another
    through ``generate_micro_sharded`` and the one-rank RCCL group's per-shard all_gather (padded to the largest shard) -- rows
    {0, 150, 299}, and every other row, equal the 300-template single batch."""
    p = subprocess.run([sys.executable, "-c", C5_CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res == {"backend": "nccl", "shape": [300, 32], "spans": True, "shard_sizes": [38, 38, 38, 38, 37, 37, 37, 37],
                   "rows_0_150_299_equal_single_batch": True, "all_rows_equal_single_batch": True}
