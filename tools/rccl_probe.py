"""One-rank RCCL smoke test for the GPU box (the pool has no multi-GPU node): initialises the ``nccl`` (= RCCL) process group
the way ``bench.py`` does for N > 1 and runs the two collectives of the sharded path (``deephumor_amd/dist.py``:
``all_gather_into_tensor`` of the padded int64 caption rows; ``barrier``) plus the max-reduction of ``bench.timed_region``.

    timeout 180 python tools/rccl_probe.py
"""
import datetime
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    t0 = time.perf_counter()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
    t_init = time.perf_counter() - t0
    rows = torch.arange(256 * 33, dtype=torch.int64, device=dev).reshape(256, 33)
    out = torch.empty_like(rows)
    dist.all_gather_into_tensor(out, rows)
    torch.cuda.synchronize()
    ok = bool(torch.equal(out, rows))
    t = torch.tensor([1.5], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20):
        dist.all_gather_into_tensor(out, rows)
    ev1.record()
    torch.cuda.synchronize()
    print(json.dumps({"backend": dist.get_backend(), "world_size": dist.get_world_size(), "init_s": round(t_init, 2),
                      "all_gather_int64_256x33_equal": ok, "all_reduce_max": float(t.item()),
                      "all_gather_us": round(ev0.elapsed_time(ev1) * 1000 / 20, 1),
                      "nccl_version": ".".join(map(str, torch.cuda.nccl.version()))}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
