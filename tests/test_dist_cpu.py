"""world_size-2 gloo test of the image-sharding path (no GPU): shard arithmetic and the single
all_gather of results reproduce the unsharded order, including uneven shards."""
import datetime
import os
import socket

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deephumor_amd.dist import generate_sharded, shard_range


def test_shard_range_covers_everything():
    for n, w in [(300, 8), (2048, 8), (5, 2), (3, 4), (256, 1)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
    assert [hi - lo for lo, hi in (shard_range(300, r, 8) for r in range(8))] == [38] * 4 + [37] * 4


def _fake_generate(lo, hi, t=6):
    """Stands in for model.generate_batch: captions are a pure function of the GLOBAL image index."""
    idx = torch.arange(lo, hi)
    toks = (idx[:, None] * 10 + torch.arange(t)[None, :]) % 97
    lens = (idx % t) + 1
    return toks, lens


def _worker(rank, world, port, n_total, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    toks, lens = generate_sharded(_fake_generate, n_total)
    ret[rank] = (toks.clone(), lens.clone())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_gather_in_global_order():
    for n_total in (8, 7):
        with mp.Manager() as mgr:
            ret = mgr.dict()
            mp.spawn(_worker, args=(2, _free_port(), n_total, ret), nprocs=2, join=True)
            want_t, want_l = _fake_generate(0, n_total)
            for r in range(2):
                toks, lens = ret[r]
                assert torch.equal(toks, want_t) and torch.equal(lens, want_l)


def _micro_worker(rank, world, port, n_total, n_shards, ret):
    from deephumor_amd.dist import generate_micro_sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    seen = []

    def gen(lo, hi):
        seen.append((lo, hi))
        return _fake_generate(lo, hi)
    toks, lens = generate_micro_sharded(gen, n_total, n_shards)
    ret[rank] = (toks.clone(), lens.clone(), seen)
    dist.barrier()
    dist.destroy_process_group()


def test_more_shards_than_ranks_gather_in_global_order():
    """generate_micro_sharded: C4's global batch as 8 shards on 2 ranks (4 rounds of one all_gather each), even and uneven shards;
    and with no process group at all (one GPU: the shards one after another)."""
    from deephumor_amd.dist import generate_micro_sharded
    for n_total, n_shards in ((16, 8), (13, 4)):
        with mp.Manager() as mgr:
            ret = mgr.dict()
            mp.spawn(_micro_worker, args=(2, _free_port(), n_total, n_shards, ret), nprocs=2, join=True)
            want_t, want_l = _fake_generate(0, n_total)
            spans = [shard_range(n_total, s, n_shards) for s in range(n_shards)]
            for r in range(2):
                toks, lens, seen = ret[r]
                assert torch.equal(toks, want_t) and torch.equal(lens, want_l)
                assert [tuple(x) for x in seen] == spans[r * n_shards // 2:(r + 1) * n_shards // 2]
    toks, lens = generate_micro_sharded(_fake_generate, 13, 8)
    want_t, want_l = _fake_generate(0, 13)
    assert torch.equal(toks, want_t) and torch.equal(lens, want_l)


def _async_worker(rank, world, port, n_total, ret):
    from deephumor_amd.dist import gather_captions_async
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    lo, hi = shard_range(n_total, rank, world)
    handles = []
    for s in range(3):                                  # three batches in flight before the first wait
        toks, lens = _fake_generate(lo, hi)
        handles.append(gather_captions_async(toks + s, lens, n_total))
    ret[rank] = [tuple(x.clone() for x in h.wait()) for h in handles]
    dist.barrier()
    dist.destroy_process_group()


def test_async_gather_waited_later():
    """gather_captions_async: exchanges issued for several batches, waited for afterwards (bench.py waits one step later) --
    every handle returns its own batch in global order, even and uneven shards."""
    for n_total in (8, 7):
        with mp.Manager() as mgr:
            ret = mgr.dict()
            mp.spawn(_async_worker, args=(2, _free_port(), n_total, ret), nprocs=2, join=True)
            want_t, want_l = _fake_generate(0, n_total)
            for r in range(2):
                for s, (toks, lens) in enumerate(ret[r]):
                    assert torch.equal(toks, want_t + s) and torch.equal(lens, want_l)


def _score_worker(rank, world, port, n_total, ret):
    from deephumor_amd.dist import score_sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    full = score_sharded(lambda lo, hi: torch.arange(lo, hi, dtype=torch.float32)[:, None] * torch.tensor([1.0, 0.5, 0.25]), n_total)
    ret[rank] = full.clone()
    dist.barrier()
    dist.destroy_process_group()


def test_score_rows_gather_in_global_order():
    """gather_rows / score_sharded (the scoring path's all-gather of float rows), even and uneven shards."""
    for n_total in (10, 9):
        with mp.Manager() as mgr:
            ret = mgr.dict()
            mp.spawn(_score_worker, args=(2, _free_port(), n_total, ret), nprocs=2, join=True)
            want = torch.arange(n_total, dtype=torch.float32)[:, None] * torch.tensor([1.0, 0.5, 0.25])
            for r in range(2):
                assert torch.equal(ret[r], want)


def _bench_worker(rank, world, port, n_total, ret):
    """bench.py's multi-rank control flow on CPU: barrier -> K steps -> barrier -> MAX all-reduce of the wall time, with
    the model stubbed (captions = a function of the global image index) and the C5-style uneven shards going through
    gather_captions."""
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from deephumor_amd.dist import gather_captions
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    lo, hi = shard_range(n_total, rank, world)
    calls = []

    def step(s):
        calls.append(s)
        time.sleep(0.05 * (rank + 1))                    # rank 1 is slower: the reported time must be ITS time
        toks, lens = _fake_generate(lo, hi)
        return gather_captions(toks, lens, n_total)

    dt, (toks, lens) = bench.timed_region(step, 3, world, "cpu")
    ret[rank] = (dt, calls, toks.clone(), lens.clone(), dist.get_world_size())
    dist.barrier()
    dist.destroy_process_group()


def test_bench_timed_region_over_two_ranks():
    n_total = 7                                           # uneven: 4 + 3 images
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_bench_worker, args=(2, _free_port(), n_total, ret), nprocs=2, join=True)
        want_t, want_l = _fake_generate(0, n_total)
        for r in range(2):
            dt, calls, toks, lens, seen = ret[r]
            assert calls == [0, 1, 2] and seen == 2                         # EXACTLY K steps on every rank
            assert torch.equal(toks, want_t) and torch.equal(lens, want_l)  # whole batch, global order, on every rank
            assert dt >= 3 * 0.1 - 0.02                                     # max over ranks (rank 1 sleeps 0.1 s per step)
        assert abs(ret[0][0] - ret[1][0]) < 1e-9                            # the same number on every rank


def test_bench_gpus_n_launches_its_own_ranks(monkeypatch, capfd):
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset): main() spawns the ranks as a CHILD
    torch.distributed.run before touching any GPU, relays rank 0's line and returns the child's rc.  Model stubbed
    (--stub: CPU / gloo), everything else -- argument relay, rendezvous on 127.0.0.1, shards, timed_region, all_gather,
    n_ranks_seen -- is the code the 8-GPU run takes."""
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "DH_BENCH_LAUNCHED_BY"):
        monkeypatch.delenv(k, raising=False)
    rc = bench.main(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "5", "--stub"])
    out = capfd.readouterr().out.strip().splitlines()
    assert rc == 0, out
    line = json.loads(out[-1])                                             # the LAST stdout line is the result line
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["ranks_launched_by"].startswith("bench.py self-launch")
    assert line["steps_run"] == [0, 1, 2] and line["gathered"] == 10 and line["gather_in_global_order"]
    assert line["data"] == "stub"                                          # can never be mistaken for a measurement
    assert line["ms_per_step"] >= 20 - 2                                   # rank 1 sleeps 20 ms per step: MAX over ranks
    pr = line["per_rank_ms_per_step"]                                      # every rank's own time (one all_gather of a float): the skew is visible
    assert pr["ranks"] == 2 and pr["max"] == pytest.approx(line["ms_per_step"]) and pr["min"] <= pr["max"]


def test_bench_c5_stub_over_eight_ranks_with_uneven_shards(monkeypatch, capfd):
    """BASELINE config 5's partitioning through the launcher path: ``bench.py --gpus 8 --workload c5 --stub`` -- 300
    templates in the uneven shards 38 x 4 + 37 x 4, every step's all_gather issued asynchronously and waited for one step later, the
    gathered batch in global order on rank 0, per-rank step times and their skew reported.  Eight gloo ranks on CPU; the model is stubbed,
    the rendezvous / shard / exchange / timing code is what the 8-GPU run takes unchanged."""
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "DH_BENCH_LAUNCHED_BY"):
        monkeypatch.delenv(k, raising=False)
    rc = bench.main(["--gpus", "8", "--steps", "3", "--warmup", "1", "--workload", "c5", "--stub"])
    out = capfd.readouterr().out.strip().splitlines()
    assert rc == 0, out
    line = json.loads(out[-1])
    assert line["n_gpus"] == 8 and line["n_ranks_seen"] == 8 and line["scaling"] == "strong" and line["data"] == "stub"
    assert line["shard_sizes"] == [38, 38, 38, 38, 37, 37, 37, 37] and line["shard"] == [0, 38]
    assert line["gathered"] == 300 and line["gather_in_global_order"] and line["steps_run"] == [0, 1, 2]
    pr = line["per_rank_ms_per_step"]
    assert pr["ranks"] == 8 and pr["max"] == pytest.approx(line["ms_per_step"]) and pr["skew_ms"] == pytest.approx(pr["max"] - pr["min"])
    assert line["ms_per_step"] >= 80 - 5                                   # rank 7 sleeps 80 ms per step: MAX over ranks


def test_bench_rejects_mismatched_world_size(monkeypatch, capfd):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "0")
    assert bench.main(["--gpus", "2", "--stub"]) == 2
    assert "WORLD_SIZE=4" in capfd.readouterr().err


def _single_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120))
    calls = []
    real = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    t0, l0 = generate_sharded(_fake_generate, 5)                       # one rank: nothing to exchange, no collective
    n0 = len(calls)
    t1, l1 = generate_sharded(_fake_generate, 5, always=True)          # the collective runs anyway (bench.py --rccl-single)
    ret[0] = (n0, len(calls), torch.equal(t0, t1) and torch.equal(l0, l1))
    dist.destroy_process_group()


def test_single_rank_collective_only_on_request():
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_single_worker, args=(1, _free_port(), ret), nprocs=1, join=True)
        assert ret[0] == (0, 1, True)
