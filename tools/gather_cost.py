"""What the per-batch exchange of the sharded path costs on one rank (--rccl-single read +0.5 ms per C2 step where
round 3 measured +0.06 ms): host time and stream time of ``gather_captions`` through a one-rank RCCL group, of its parts (the packing
torch ops, the collective alone), and of the asynchronous form (``gather_captions_async``)."""
import json
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from deephumor_amd import dist as D
    toks = torch.randint(0, 30000, (256, 32), device=dev)
    lens = torch.randint(1, 32, (256,), device=dev)
    out = {}

    def timeit(name, fn, n=200):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        host = (time.perf_counter() - t0) / n * 1e6
        torch.cuda.synchronize()
        out[name] = {"host_us": round(host, 1), "stream_us": round(e0.elapsed_time(e1) / n * 1e3, 1)}

    timeit("gather_captions(always=True)", lambda: D.gather_captions(toks, lens, 256, always=True))
    packed = torch.zeros((256, 33), dtype=torch.int64, device=dev)
    recv = torch.empty((256, 33), dtype=torch.int64, device=dev)
    timeit("all_gather_into_tensor alone", lambda: dist.all_gather_into_tensor(recv, packed))

    def pack_only():
        p = torch.zeros((256, 33), dtype=torch.int64, device=dev)
        p[:, :32] = toks
        p[:, 32] = lens
        full = torch.cat([p[:256]], 0)
        return full[:, :32].contiguous(), full[:, 32].contiguous()
    timeit("packing + unpacking torch ops alone", pack_only)
    if hasattr(D, "gather_captions_async"):
        timeit("gather_captions_async(...).wait()", lambda: D.gather_captions_async(toks, lens, 256, always=True).wait())
        hs = []

        def deferred():
            hs.append(D.gather_captions_async(toks, lens, 256, always=True))
            if len(hs) > 1:
                hs.pop(0).wait()
        timeit("gather_captions_async, waited one call later", deferred)
    print(json.dumps(out, indent=1))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
