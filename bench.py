#!/usr/bin/env python3
"""Benchmark of the image -> caption hot path on MI355X (contract: see the task brief / DESIGN.md).

    python bench.py --gpus 1 --steps K --warmup W                 # one rank
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic images that are already resident
in HBM: ResNet-50 encoder -> decoder -> beam-search generate (beam=5, top_k=50, 32 tokens) for
``--batch`` images per rank, followed -- when N > 1 -- by the single all_gather of token ids.
``value`` = captions finished by all ranks / wall time (max over ranks), weak scaling (256 images
per GPU).  Workloads (BASELINE.json configs): ``c2`` = CaptioningLSTM (default, configs[1]);
``c3`` = CaptioningTransformer 6-layer/8-head.  The default run reports c2 as the contract line and
attaches a short c3 measurement (with the decoder self-attention roofline) under ``"c3"``.

Only the ``cpu_baseline`` leg and the greedy parity check import ``oracle/`` (the CPU restatement of
the reference); the measured path is the HIP library only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

V_WORD = 36541          # deephumor_demo.ipynb:524
MAX_LEN = 32            # deephumor_demo.ipynb:1127
BEAM, TOP_K, TEMP = 5, 50, 1.0
STREAMS = int(os.environ.get("DH_DECODE_STREAMS", "1"))
GRAPH = False        # set by --graph   # image sub-batches decoded concurrently (HIP streams)
PEAK_HBM_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PEAK_F32_TFLOPS = 157.3  # fp32 vector == fp32 MFMA peak
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak


_SD_CACHE = {}


def build_model(workload, dev, dtype="bf16"):
    import deephumor_amd.models as M
    from deephumor_amd.synth import synth_state_dict
    cls = {"c2": M.CaptioningLSTM, "c3": M.CaptioningTransformer, "c5": M.CaptioningTransformerWithLabels}[workload]
    model = cls(V_WORD).eval()
    if workload not in _SD_CACHE:
        _SD_CACHE[workload] = synth_state_dict(model.state_dict(), seed=1234)
    sd = _SD_CACHE[workload]
    model.load_state_dict(sd)
    model = model.to(dev)
    if dtype == "bf16":
        model = model.bfloat16()
    elif dtype == "f16":
        model = model.half()
    return model, sd, model._hp


def one_step(model, images, img0, n_total, seed, eager=False, labels=None, beam=None):
    from deephumor_amd.dist import gather_captions
    if labels is not None:      # config C5: ImageLabelEncoder (+ spatial features) + CaptioningTransformer, beam 10
        toks, lens = model.generate_batch(images, labels, max_len=MAX_LEN, beam_size=beam, top_k=TOP_K, temperature=TEMP,
                                          seed=seed, img0=img0)
        return gather_captions(toks, lens, n_total)
    if GRAPH and not eager:      # whole step replayed from a captured hipGraph (profiling passes run eagerly: events need real launches)
        toks, lens = model.generate_batch_graphed(images, max_len=MAX_LEN, beam_size=BEAM, top_k=TOP_K, temperature=TEMP,
                                                  seed=seed, img0=img0)
    else:
        toks, lens = model.generate_batch(images, max_len=MAX_LEN, beam_size=BEAM, top_k=TOP_K, temperature=TEMP,
                                          seed=seed, img0=img0, streams=STREAMS)
    return gather_captions(toks, lens, n_total)


def kind_of(workload):
    return {"c2": "CaptioningLSTM", "c3": "CaptioningTransformer", "c5": "CaptioningTransformerWithLabels"}[workload]


def run_score(args, rank, world, dev, dtype, kind):
    """teacher-forced scoring (per-caption perplexity) of a caption corpus with per-template
    encoder-feature caching: 300 templates x 30 captions of 32 tokens, captions sharded over the ranks, one
    all-gather of the perplexities."""
    import numpy as np
    from deephumor_amd.dist import score_sharded
    from deephumor_amd.experiments.scoring import score_captions
    from deephumor_amd.synth import synth_images
    model, sd, hp = build_model(kind, dev, dtype)
    n_tpl, per = 300, 30
    n_total = n_tpl * per
    images = synth_images(n_tpl, seed=3).to(dev)
    g = np.random.Generator(np.random.Philox(key=[7, 0]))
    lengths = torch.from_numpy(g.integers(8, 33, size=(n_total,)).astype(np.int64))
    caps = torch.from_numpy(g.integers(6, V_WORD, size=(n_total, 32)).astype(np.int64))
    for r in range(n_total):
        caps[r, lengths[r] - 1] = 3                 # <eos>
        caps[r, lengths[r]:] = 0
    tpl = torch.arange(n_total, dtype=torch.int64) // per
    caps, lengths_d = caps.to(dev), lengths.to(dev)

    def run():
        return score_sharded(lambda lo, hi: score_captions(model, images, tpl[lo:hi], caps[lo:hi], lengths_d[lo:hi], batch_size=args.score_batch), n_total)

    pp = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = max(1, args.steps)
    for _ in range(reps):
        pp = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    return {"workload": f"teacher-forced scoring, {kind_of(kind)}, 300 templates x 30 captions x 32 tokens, V={V_WORD}",
            "value": n_total * reps / dt, "unit": "captions scored/s", "passes": reps, "ms_per_pass": dt / reps * 1e3,
            "captions": n_total, "mean_perplexity": float(pp.float().mean()), "dtype": dtype}


def run_c5(args, rank, world, dev, dtype):
    """BASELINE config 5: the full 300-template sweep (ImageLabelEncoder + CaptioningTransformer, beam 10), templates
    sharded 38/38/38/38/37/37/37/37 over 8 ranks (all 300 on one), uneven shards gathered with one padded all_gather."""
    import torch.distributed as dist
    from deephumor_amd.dist import shard_range
    from deephumor_amd.synth import synth_images
    import numpy as np
    model, sd, hp = build_model("c5", dev, dtype)
    n_total = 300
    lo, hi = shard_range(n_total, rank, world)
    images = synth_images(hi - lo, seed=2, first=lo).to(dev)
    g = np.random.Generator(np.random.Philox(key=[1, 0]))
    labels = torch.from_numpy(g.integers(6, V_WORD, size=(n_total, 3)).astype(np.int64))[lo:hi].to(dev)
    with torch.no_grad():
        one_step(model, images, lo, n_total, 0, labels=labels, beam=10)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        reps = max(1, args.steps)
        for s in range(reps):
            toks, lens = one_step(model, images, lo, n_total, 100 + s, labels=labels, beam=10)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    return {"workload": "C5 ImageLabelEncoder + CaptioningTransformer (spatial feats), beam=10, 300-template sweep",
            "value": n_total * reps / dt, "unit": "captions/s", "sweeps": reps, "ms_per_sweep": dt / reps * 1e3,
            "templates": n_total, "gathered": int(toks.shape[0]), "dtype": dtype}


def cpu_baseline(workload, sd, hp, n_sample):
    """The oracle (CPU restatement of the reference: per-image generate, full re-forward per token,
    fp32, torch CPU threads) timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import ref_path as R
    from deephumor_amd.synth import synth_images
    imgs = synth_images(n_sample, seed=0)
    torch.manual_seed(0)
    R.model_generate(kind_of(workload), sd, hp, imgs[:1], max_len=4, beam_size=BEAM, top_k=TOP_K)   # warm-up
    t0 = time.perf_counter()
    n_tok = 0
    for i in range(n_sample):
        ids = R.model_generate(kind_of(workload), sd, hp, imgs[i:i + 1], max_len=MAX_LEN, temperature=TEMP,
                               beam_size=BEAM, top_k=TOP_K)
        n_tok += ids.numel()
    dt = time.perf_counter() - t0
    return {"value": n_sample / dt, "unit": "captions/s", "cores": torch.get_num_threads(), "kind": "port",
            "host_cpus": os.cpu_count(),
            "sample": f"{n_sample} images, {kind_of(workload)} V={V_WORD}, beam={BEAM}, top_k={TOP_K}, "
                      f"max_len={MAX_LEN}, per-image generate with full re-forward per token (reference algorithm), "
                      f"{dt:.1f} s, mean length {n_tok / n_sample:.1f}"}


def greedy_match(workload, model, sd, hp, n_check):
    """Greedy decode (beam_size=1, top_k=1) token match of the HIP path vs the CPU oracle."""
    from oracle import ref_path as R
    from deephumor_amd.synth import synth_images
    imgs = synth_images(n_check, seed=0)
    with torch.no_grad():
        toks, lens = model.generate_batch(imgs.to(next(model.parameters()).device), max_len=MAX_LEN, beam_size=1, top_k=1)
    same = total = 0
    for i in range(n_check):
        want = R.model_generate(kind_of(workload), sd, hp, imgs[i:i + 1], max_len=MAX_LEN, beam_size=1, top_k=1).reshape(-1).tolist()
        got = toks[i, :int(lens[i])].cpu().tolist()
        total += max(len(want), len(got))
        same += sum(int(a == b) for a, b in zip(want, got))
    return same / max(total, 1)


def roofline_from(summary, prefer=None, dtype="bf16"):
    """Picks the dominant kernel (by measured time) -- or ``prefer`` -- and prices it against its roofline."""
    if not summary:
        return None
    key = prefer if prefer in summary else max(summary, key=lambda k: summary[k]["ms"])
    d = summary[key]
    sec = d["ms"] / 1e3 / max(d["calls"], 1)
    if key.startswith("dh_attn"):
        ach = d["bytes"] / d["calls"] / sec / 1e9
        return {"kernel": key, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": ach / PEAK_HBM_GBS, "traffic": None, "avg_launch_us": sec * 1e6, "launches": d["calls"],
                "algorithmic_bytes_per_launch": d["bytes"] / d["calls"]}
    peak = PEAK_F32_TFLOPS if (dtype == "f32" or key.startswith("dh_stem")) else PEAK_BF16_TFLOPS
    # the roofline that bounds these launches: algorithmic intensity (flop per algorithmic byte) against the
    # ridge peak_flops / peak_bytes -- e.g. the 1x1 convolutions with K <= 256 are HBM-bound, not MFMA-bound
    intensity = d["flops"] / d["bytes"] if d["bytes"] else float("inf")
    ridge = peak * 1e12 / (PEAK_HBM_GBS * 1e9)
    if intensity < ridge:
        ach = d["bytes"] / d["calls"] / sec / 1e9
        return {"kernel": key, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": ach / PEAK_HBM_GBS, "traffic": None, "avg_launch_us": sec * 1e6, "launches": d["calls"],
                "algorithmic_bytes_per_launch": d["bytes"] / d["calls"],
                "algorithmic_flops_per_launch": d["flops"] / d["calls"], "flop_per_byte": intensity,
                "tflops": d["flops"] / d["calls"] / sec / 1e12}
    ach = d["flops"] / d["calls"] / sec / 1e12
    return {"kernel": key, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
            "frac": ach / peak, "traffic": None, "avg_launch_us": sec * 1e6, "launches": d["calls"],
            "algorithmic_flops_per_launch": d["flops"] / d["calls"], "flop_per_byte": intensity}


def run_workload(workload, args, rank, world, dev, steps, warmup, with_cpu, dtype="bf16", main_line=True):
    import torch.distributed as dist
    from deephumor_amd import hip
    from deephumor_amd.synth import synth_images
    model, sd, hp = build_model(workload, dev, dtype)
    n_local, n_total = args.batch, args.batch * world
    images = synth_images(n_local, seed=0, first=rank * n_local).to(dev)     # resident in HBM before timing
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    lens = None
    with torch.no_grad():
        # untimed: a cold pass (code-object loading, allocator growth), then the per-kernel breakdown pass
        # (HIP events around every launch made through the library)
        one_step(model, images, rank * n_local, n_total, seed=0)
        with hip.profile() as prof0:
            _, lens = one_step(model, images, rank * n_local, n_total, seed=0, eager=True)
        breakdown = prof0.summary()
        dominant = max(breakdown, key=lambda k: breakdown[k]["ms"])
        # watch the dominant "entry[tag]" only (plus the decoder self-attention, the north star's roofline
        # target, when this workload is the contract line): a few dozen event pairs per step
        watch = {dominant} | ({"dh_attn_self_decode"} if (workload == "c3" and main_line) else set())
        for w in range(1, warmup):
            _, lens = one_step(model, images, rank * n_local, n_total, seed=w)
        barrier()
        t0 = time.perf_counter()
        # timed region: HIP events (on the launch stream) only around the dominant entry point and the
        # attention kernels, so the roofline line is measured over exactly the steps `value` is
        if main_line and not GRAPH:
            with hip.profile(watch=watch, stride=4) as prof:     # every 4th launch: <0.3 ms of events per step
                for s in range(steps):
                    _, lens = one_step(model, images, rank * n_local, n_total, seed=100 + s)
                torch.cuda.synchronize()
                barrier()
                dt = time.perf_counter() - t0
            summary = prof.summary()
        else:
            # secondary workload: clean timed steps for its captions/s, then one separately profiled step for
            # its rooflines (so ~600 event pairs per step do not perturb the number)
            for s in range(steps):
                _, lens = one_step(model, images, rank * n_local, n_total, seed=100 + s)
            torch.cuda.synchronize()
            barrier()
            dt = time.perf_counter() - t0
            with hip.profile(watch={dominant, "dh_attn_self_decode", "dh_attn_cross_decode"}) as prof:
                one_step(model, images, rank * n_local, n_total, seed=999, eager=True)
            summary = prof.summary()
    t = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    graph_res = None
    if main_line and not GRAPH and world == 1:
        # the same K steps replayed from a captured hipGraph (reported next to the contract number, which is eager
        # because the roofline events need real launches)
        with torch.no_grad():
            kw = dict(max_len=MAX_LEN, beam_size=BEAM, top_k=TOP_K, temperature=TEMP)
            model.generate_batch_graphed(images, seed=1, **kw)
            torch.cuda.synchronize()
            tg = time.perf_counter()
            for s in range(steps):
                model.generate_batch_graphed(images, seed=100 + s, **kw)
            torch.cuda.synchronize()
            tg = time.perf_counter() - tg
        graph_res = {"value": n_total * steps / tg, "unit": "captions/s", "ms_per_step": tg / steps * 1e3}
    res = {"value": n_total * steps / dt, "ms_per_step": dt / steps * 1e3, "hipgraph_replay": graph_res,
           "mean_caption_len": float(lens.float().mean()) if lens is not None else None}
    total_ms = sum(d["ms"] for d in breakdown.values())
    res["kernel_breakdown_ms_per_step"] = {k: round(v["ms"], 3) for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1]["ms"])}
    res["kernel_ms_sum"] = round(total_ms, 3)
    res["roofline"] = roofline_from(summary, prefer=dominant, dtype=dtype)
    if workload == "c3":
        res["roofline_self_attention"] = roofline_from(summary, prefer="dh_attn_self_decode", dtype=dtype)
        res["roofline_cross_attention"] = roofline_from(summary, prefer="dh_attn_cross_decode", dtype=dtype)
    try:     # HBM traffic of those kernels from the committed PMC passes (rocprofv3 cannot run inside this process)
        pmc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1", "pmc_hbm_traffic.json")))
        for key in ("roofline", "roofline_self_attention", "roofline_cross_attention"):
            rl = res.get(key)
            ent = pmc.get(workload, {}).get(rl["kernel"]) if (rl and dtype == "bf16") else None
            if ent:
                rl["traffic"] = ent["traffic_bytes_per_launch"]
                rl["traffic_source"] = "profiles/r1/pmc_hbm_traffic.json (" + ent["note"] + ")"
    except (OSError, ValueError, KeyError, TypeError):
        pass
    if rank == 0 and with_cpu:
        # the parity gate is the fp32 path: bit-exact greedy ids vs the CPU reference path
        m32 = model if dtype == "f32" else build_model(workload, dev, "f32")[0]
        res["greedy_token_match_vs_cpu_ref"] = greedy_match(workload, m32, sd, hp, 2)
        if dtype != "f32":
            res["greedy_token_match_bf16_vs_cpu_ref"] = greedy_match(workload, model, sd, hp, 2)
            del m32
        res["cpu_baseline"] = cpu_baseline(workload, sd, hp, args.cpu_sample if workload == "c2" else max(2, args.cpu_sample // 4))
        res["speedup_vs_cpu"] = res["value"] / res["cpu_baseline"]["value"]
    del model
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU (BASELINE configs: 256)")
    ap.add_argument("--workload", choices=["c2", "c3", "both", "c5", "score-c2", "score-c3"], default="both")
    ap.add_argument("--cpu-sample", type=int, default=8, help="images for the CPU baseline leg (rank 0, N=1 only)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--score-batch", type=int, default=1024, help="captions per teacher-forced batch (score-* workloads)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (eager by default: "
                    "the in-library event profiler of the roofline line needs real launches)")
    ap.add_argument("--dtype", choices=["bf16", "f16", "f32"], default="bf16",
                    help="storage/MFMA operand type of the measured path (BASELINE configs C2/C3: bf16)")
    args = ap.parse_args()

    global GRAPH
    GRAPH = args.graph
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.workload.startswith("score-"):
        res = run_score(args, rank, world, dev, args.dtype, args.workload.split("-")[1])
        if rank == 0:
            print(json.dumps(dict(res, metric="captions scored/sec (teacher-forced perplexity, 32 tokens)", n_gpus=world,
                                  higher_is_better=True, scaling="strong", vs_baseline=None, data="synthetic")))
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.workload == "c5":
        res = run_c5(args, rank, world, dev, args.dtype)
        if rank == 0:
            print(json.dumps(dict(res, metric="captions/sec (224x224, 32-tok, beam=10, 300-template sweep)", n_gpus=world,
                                  higher_is_better=True, scaling="strong", vs_baseline=None, data="synthetic")))
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return
    main_wl = "c2" if args.workload in ("c2", "both") else "c3"
    with_cpu = (world == 1) and not args.no_cpu
    res = run_workload(main_wl, args, rank, world, dev, args.steps, args.warmup, with_cpu, args.dtype)
    line = {
        "metric": "captions/sec (224x224, 32-tok, beam=5)", "value": res["value"], "unit": "captions/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": ("C2 CaptioningLSTM + ImageEncoder (emb 256, hidden 512, 2 layers)" if main_wl == "c2" else
                                "C3 CaptioningTransformer 6-layer/8-head (spatial feats)"),
                   "images_per_gpu": args.batch, "global_batch": args.batch * world, "vocab": V_WORD,
                   "max_len": MAX_LEN, "beam_size": BEAM, "top_k": TOP_K, "temperature": TEMP,
                   "parallelism": f"image-sharded x{world}, one all_gather of token ids per batch",
                   "weights": "synthetic name-keyed (seed 1234)", "encoder_in_timed_region": True},
        "roofline": res["roofline"], "cpu_baseline": res.get("cpu_baseline"),
        "greedy_token_match_vs_cpu_ref": res.get("greedy_token_match_vs_cpu_ref"),
        "greedy_token_match_bf16_vs_cpu_ref": res.get("greedy_token_match_bf16_vs_cpu_ref"),
        "speedup_vs_cpu": res.get("speedup_vs_cpu"), "mean_caption_len": res["mean_caption_len"],
        "hipgraph_replay": res.get("hipgraph_replay"),
        "kernel_breakdown_ms_per_step": res["kernel_breakdown_ms_per_step"],
    }
    if main_wl == "c3":
        line["roofline_self_attention"] = res.get("roofline_self_attention")
        line["roofline_cross_attention"] = res.get("roofline_cross_attention")
    if args.workload == "both":
        r3 = run_workload("c3", args, rank, world, dev, max(2, args.steps // 2), 1, with_cpu, args.dtype, main_line=False)
        line["c3"] = {"workload": "C3 CaptioningTransformer 6-layer/8-head (spatial feats), same batch/beam settings",
                      "value": r3["value"], "unit": "captions/s", "ms_per_step": r3["ms_per_step"],
                      "roofline": r3["roofline"], "roofline_self_attention": r3.get("roofline_self_attention"),
                      "roofline_cross_attention": r3.get("roofline_cross_attention"),
                      "cpu_baseline": r3.get("cpu_baseline"), "speedup_vs_cpu": r3.get("speedup_vs_cpu"),
                      "greedy_token_match_vs_cpu_ref": r3.get("greedy_token_match_vs_cpu_ref"),
                      "kernel_breakdown_ms_per_step": r3["kernel_breakdown_ms_per_step"]}
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
