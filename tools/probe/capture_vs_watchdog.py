"""hipGraph captures next to pending RCCL works of a one-rank process group: c10d's watchdog thread polls hipEventQuery every 100 ms, and
in HIP's global capture mode one such call inside a capture window fails the capture and aborts the process (round 5).

    python tools/probe/capture_vs_watchdog.py global 400        # dies within a few iterations
    python tools/probe/capture_vs_watchdog.py thread_local 400  # {"failures": 0}
"""
import os, sys, json, datetime, socket
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mode = sys.argv[1]
n_iter = int(sys.argv[2])
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=120))
x = torch.ones(4096, device="cuda")
out = torch.empty(4096, device="cuda")
fails = 0
first_err = None
a = torch.randn(256, 256, device="cuda")
b = a @ a * 0.01
torch.cuda.synchronize()
for i in range(n_iter):
    works = [dist.all_gather_into_tensor(out, x, async_op=True) for _ in range(3)]      # pending works for the watchdog to poll
    g = torch.cuda.CUDAGraph()
    try:
        kw = {} if mode == "global" else {"capture_error_mode": mode}
        with torch.cuda.graph(g, **kw):
            b = a
            for _ in range(200):
                b = b @ a * 0.01
        g.replay()
        torch.cuda.synchronize()
    except Exception as e:                                     # noqa: BLE001
        fails += 1
        first_err = first_err or repr(e)[:200]
        torch.cuda.synchronize()
    for w in works:
        w.wait()
print(json.dumps(dict(mode=mode, captures=n_iter, failures=fails, first_err=first_err)), flush=True)
dist.destroy_process_group()
