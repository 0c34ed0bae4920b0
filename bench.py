#!/usr/bin/env python3
"""Benchmark of the image -> caption hot path on MI355X (contract: see the task brief / DESIGN.md section 6b).

    python bench.py --gpus 1 --steps K --warmup W                 # one rank, in-process
    python bench.py --gpus N --steps K --warmup W                 # N > 1 without a launcher: starts its N ranks itself (child
                                                                  # torch.distributed.run, before anything touches a GPU)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W     # the driver's form

A "step" is one pass of the hot path over one batch of synthetic images that are already resident in HBM:
ResNet-50 encoder -> decoder -> beam-search generate (beam=5, top_k=50, 32 tokens) for ``--batch`` images per rank,
followed -- when N > 1 -- by the single all_gather of token ids.  ``value`` = captions finished by all ranks / wall time
(max over ranks), weak scaling (256 images per GPU).  Schedule (``--schedule``, round 5): the K steps of the timed region run as ONE
stream of batches (``CaptionPipeline``: batch i + 1's encoder on its own HIP stream under batch i's decode, the batch's all_gather
waited for one step later) -- every step still encodes, decodes and exchanges exactly one batch and all K are complete when the
region closes; the one-batch-at-a-time rate of rounds 1-4 is measured in the same run and reported as ``value_sequential``.  Workloads (BASELINE.json configs): ``c2`` = CaptioningLSTM
(default, configs[1]); ``c3`` = CaptioningTransformer 6-layer/8-head; ``c5`` = the 300-template fp16 beam-10 sweep.
The default run reports c2 as the contract line and attaches a c3 measurement under ``"c3"``.

What the line carries besides the contract fields:
  roofline           ONE kernel template at ONE shape -- the launch key (entry[role]{MxNxK}) with the largest share of the
                     step -- timed with HIP events recorded inside the library on the launch stream, every launch of it,
                     inside the timed region; priced against MFMA peak or HBM bandwidth by its arithmetic intensity
  encoder_layers     per-shape table of the encoder's convolution launches (from an untimed, fully instrumented pass)
  cpu_baseline       the oracle (CPU restatement of the reference) on this box's host cores: thread-count sweep,
                     warm-up + best of 3 (rank 0, N = 1 only)
  fp32_parity_path   captions/s of the fp32 path -- the one whose greedy ids are bit-exact vs the CPU reference
  host_inclusive     pinned host images in -> token ids in pinned host memory out: sequential, and software-pipelined
                     (copy + encoder of batch i+1 on their own streams while batch i decodes; deephumor_amd/pipeline.py), for
                     decoded uint8 HWC images (value_host_inclusive at the top level) and for the fp32 NCHW batch
  precision_vs_fp32_hip   greedy token match / step-0 logit error of the bf16 AND fp16 paths against the fp32 HIP path over all
                     bench images (what 16-bit storage costs; the table behind it: profiles/r3/precision_*.json)

Only the ``cpu_baseline`` leg and the greedy parity check import ``oracle/``; the measured path is the HIP library only.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

V_WORD = 36541          # deephumor_demo.ipynb:524
MAX_LEN = 32            # deephumor_demo.ipynb:1127
BEAM, TOP_K, TEMP = 5, 50, 1.0
N_CHECK = 4             # images of the greedy parity check against the CPU oracle
STREAMS = int(os.environ.get("DH_DECODE_STREAMS", "1"))      # (option "decode_streams"; read before the library is loaded)
PEAK_HBM_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PEAK_F32_TFLOPS = 157.3  # fp32 vector == fp32 MFMA peak
PEAK_16_TFLOPS = 2500.0  # dense bf16 / fp16 MFMA peak
PMC_ROUND = next((r for r in ("r6", "r5") if os.path.exists(os.path.join(ROOT, "profiles", r, "pmc_hbm_traffic.json"))), "r6")
PMC_FILE = os.path.join(ROOT, "profiles", PMC_ROUND, "pmc_hbm_traffic.json")
TORCH_DTYPE = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}

_SD_CACHE = {}
RANK_TIMES = []         # wall time of the last timed region on every rank (seconds), filled by timed_region


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
    if dtype != "f32":
        model = model.to(TORCH_DTYPE[dtype])
    return model, sd, model._hp


def kind_of(workload):
    return {"c2": "CaptioningLSTM", "c3": "CaptioningTransformer", "c5": "CaptioningTransformerWithLabels"}[workload]


def workload_name(workload):
    return {"c2": "C2 CaptioningLSTM + ImageEncoder (emb 256, hidden 512, 2 layers)",
            "c3": "C3 CaptioningTransformer 6-layer/8-head (spatial feats)",
            "c5": "C5 ImageLabelEncoder + CaptioningTransformer (spatial feats), beam=10, 300-template sweep"}[workload]


def one_step(model, images, img0, n_total, seed, graph=False, pending=None):
    """One batch: encoder + decode + the batch's single all_gather.  With ``pending`` (a one-element list) the all_gather is issued
    asynchronously and the PREVIOUS step's is waited for here, after this step's launches are queued: every step still makes
    exactly one exchange, its latency sits behind the next batch's decode (``drain_pending`` ends the last one)."""
    from deephumor_amd.dist import gather_captions, gather_captions_async
    kw = dict(max_len=MAX_LEN, beam_size=BEAM, top_k=TOP_K, temperature=TEMP, seed=seed, img0=img0)
    if graph:      # whole step replayed from a captured hipGraph
        toks, lens = model.generate_batch_graphed(images, **kw)
    else:
        toks, lens = model.generate_batch(images, streams=STREAMS, **kw)
    if pending is None:
        return gather_captions(toks, lens, n_total)
    prev, pending[0] = pending[0], gather_captions_async(toks, lens, n_total)
    return prev.wait() if prev is not None else (toks, lens)


def drain_pending(pending):
    if pending and pending[0] is not None:
        out, pending[0] = pending[0].wait(), None
        return out
    return None


# ---- rooflines ----------------------------------------------------------------------------------------------------
def price(key, d, dtype):
    """One launch key (calls, ms, algorithmic flops / bytes) against the roofline that bounds it."""
    sec = d["ms"] / 1e3 / max(d["calls"], 1)
    fl, by = d["flops"] / max(d["calls"], 1), d["bytes"] / max(d["calls"], 1)
    peak = PEAK_F32_TFLOPS if dtype == "f32" else PEAK_16_TFLOPS
    if "_f32x" in key:          # split-operand fp32 kernels: three fp16 MFMAs per algorithmic product
        peak = PEAK_16_TFLOPS / 3.0
    out = {"kernel": key, "launches": d["calls"], "avg_launch_us": sec * 1e6, "traffic": None}
    if "_f32x" in key:
        out["peak_note"] = "dense fp16 MFMA peak / 3 (hi*hi + hi*lo + lo*hi per product)"
    if by:
        out["algorithmic_bytes_per_launch"] = by
    if fl:
        out["algorithmic_flops_per_launch"] = fl
    if not by and not fl:
        return dict(out, bound=None, achieved=None, peak=None, unit=None, frac=None)
    intensity = fl / by if by else float("inf")
    if fl and by:
        out["flop_per_byte"] = intensity
    ridge = peak * 1e12 / (PEAK_HBM_GBS * 1e9)
    if not fl or intensity < ridge:
        ach = by / sec / 1e9
        out.update(bound="hbm", achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s", frac=ach / PEAK_HBM_GBS)
        if fl:
            out["tflops"] = fl / sec / 1e12
    else:
        ach = fl / sec / 1e12
        out.update(bound="mfma", achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak)
    return out


def attach_traffic(rl, workload, dtype):
    """HBM traffic per launch of the roofline kernel from the committed PMC passes of THIS tree (rocprofv3 --pmc cannot
    run inside this process); the file names the commit it was measured on."""
    if rl is None or dtype != "bf16":
        return rl
    try:
        pmc = json.load(open(PMC_FILE))
        ent = pmc.get(workload, {}).get(rl["kernel"])
        if ent:
            rl["traffic"] = ent["traffic_bytes_per_launch"]
            rl["traffic_over_algorithmic"] = ent["traffic_bytes_per_launch"] / rl["algorithmic_bytes_per_launch"]
            rl["traffic_source"] = f"profiles/{PMC_ROUND}/pmc_hbm_traffic.json @ {pmc.get('commit', '?')} ({ent.get('note', '')})"
            if ent.get("kernel_only_us"):
                # the same kernel's average duration in the rocprofv3 kernel trace of that tree (no event pair around the launch): the
                # event-timed `avg_launch_us` above carries ~2 us of instrumentation on a 10 us launch
                us = ent["kernel_only_us"]
                rl["kernel_only_us"] = us
                rl["kernel_only_source"] = f"profiles/{PMC_ROUND}/{workload}_bf16_kernel_stats.csv (rocprofv3 --kernel-trace, {ent.get('kernel_only_launches', '?')} launches)"
                if rl.get("bound") == "hbm":
                    rl["frac_kernel_only"] = rl["algorithmic_bytes_per_launch"] / (us * 1e-6) / 1e9 / PEAK_HBM_GBS
                    # HBM-side: what the chip really moved per launch (PMC) over the same time -- below the effective fraction when
                    # sibling beams share ancestor rows in L2 (self-attention), above it when padding / re-reads are fetched
                    rl["frac_hbm_side_kernel_only"] = ent["traffic_bytes_per_launch"] / (us * 1e-6) / 1e9 / PEAK_HBM_GBS
                elif rl.get("bound") == "mfma":
                    rl["frac_kernel_only"] = rl["algorithmic_flops_per_launch"] / (us * 1e-6) / 1e12 / rl["peak"]
    except (OSError, ValueError, KeyError, TypeError):
        pass
    return rl


def merge_keys(summary, pred):
    """Sum of all launch keys selected by ``pred`` (e.g. every history depth of the self-attention kernel)."""
    sel = [v for k, v in summary.items() if pred(k)]
    if not sel:
        return None
    return {f: sum(v[f] for v in sel) for f in ("calls", "ms", "flops", "bytes")}


def encoder_table(breakdown, dtype):
    rows = []
    for k, d in breakdown.items():
        if not k.startswith("dh_conv2d"):
            continue
        r = price(k, d, dtype)
        rows.append({"key": k, "launches": r["launches"], "us": round(r["avg_launch_us"], 1), "bound": r["bound"],
                     "frac": round(r["frac"], 3), "gflop": round(r.get("algorithmic_flops_per_launch", 0) / 1e9, 2),
                     "mbytes": round(r.get("algorithmic_bytes_per_launch", 0) / 1e6, 1)})
    rows.sort(key=lambda r: -r["us"] * r["launches"])
    return rows


# ---- CPU legs -----------------------------------------------------------------------------------------------------
def cpu_baseline(workload, sd, hp):
    """The oracle (CPU restatement of the reference: per-image generate, full re-forward per token, fp32) timed on this
    box's host cores on a bounded sample of the same workload: torch thread-count sweep on short captions, then warm-up +
    best of 3 full-length captions at the best thread count."""
    from oracle import ref_path as R
    from deephumor_amd.synth import synth_images
    imgs = synth_images(24, seed=0)
    kind = kind_of(workload)
    gen = lambda i, n: R.model_generate(kind, sd, hp, imgs[i:i + 1], max_len=n, temperature=TEMP, beam_size=BEAM, top_k=TOP_K)
    t_all = time.perf_counter()
    torch.manual_seed(0)
    saved = torch.get_num_threads()
    sweep = {}
    short = 6
    cands = sorted({t for t in (8, 16, 32, 64, 128) if t <= (os.cpu_count() or 8)} | {min(8, os.cpu_count() or 8)})
    for th in cands:
        torch.set_num_threads(th)
        gen(0, 2)                                         # warm-up at this thread count
        t0 = time.perf_counter()
        gen(0, short)
        sweep[th] = time.perf_counter() - t0
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    gen(0, MAX_LEN)                                       # warm-up at full length; sizes the sample to ~12 s of CPU work
    t1 = time.perf_counter() - t0
    n_img = int(max(3, min(24, -(-12.0 // (3 * t1)))))
    times, n_tok = [], 0
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(n_img):
            n_tok += gen(i, MAX_LEN).numel()
        times.append(time.perf_counter() - t0)
    torch.set_num_threads(saved)
    return {"value": n_img / min(times), "unit": "captions/s", "cores": best, "kind": "port", "host_cpus": os.cpu_count(),
            "thread_sweep_s_per_short_caption": {str(k): round(v, 3) for k, v in sweep.items()},
            "pass_times_s": [round(t, 3) for t in times],
            "sample": f"{n_img} images per pass, best of 3 passes, {kind} V={V_WORD}, beam={BEAM}, top_k={TOP_K}, max_len={MAX_LEN}, "
                      f"per-image generate with full re-forward per token (reference algorithm), threads swept over {cands} on "
                      f"{short}-token captions, {time.perf_counter() - t_all:.1f} s of CPU work in all, mean length {n_tok / (3 * n_img):.1f}",
            "oracle_vs_reference": "profiles/r3/oracle_vs_reference_timing.json (build container, 8 vCPU)"}


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


def greedy_all(model, images, chunk=256):
    """Greedy decode (beam 1, top_k 1) of every bench image: (tokens [N, 32], lengths [N], step-0 logits [N, V] fp32)."""
    toks, lens, lg0 = [], [], []
    with torch.no_grad():
        for lo in range(0, images.shape[0], chunk):
            cap = {}
            t, l = model.generate_batch(images[lo:lo + chunk], max_len=MAX_LEN, beam_size=1, top_k=1,
                                        logits_hook=lambda i, lg: cap.__setitem__(0, lg.float().clone()) if i == 0 else None)
            toks.append(t), lens.append(l), lg0.append(cap[0][:t.shape[0]])
    return torch.cat(toks), torch.cat(lens), torch.cat(lg0)


def compare_greedy(ref, got):
    """Token agreement of two greedy runs over the same images + the step-0 logit error (the perturbation that flips thin
    arg-max margins; greedy decoding of synthetic random weights is chaotic: one flip changes the rest of the caption)."""
    (rt, rl, r0), (gt, gl, g0) = ref, got
    n, t = rt.shape
    pos = torch.arange(t, device=rt.device)[None, :]
    valid = pos < torch.maximum(rl, gl)[:, None]
    same = (rt == gt) & valid & (pos < torch.minimum(rl, gl)[:, None])
    first = torch.where(valid & ~same, pos, torch.full_like(pos, t)).min(1).values
    d = (g0 - r0).abs()
    top2 = torch.topk(r0, 2, dim=-1).values
    return {"images": n, "token_match": float(same.sum() / valid.sum()), "captions_identical": float((first == t).float().mean()),
            "mean_first_divergence_pos": float(first.float().mean()), "first_token_match": float((rt[:, 0] == gt[:, 0]).float().mean()),
            "step0_logit_max_abs_err": float(d.max()), "step0_logit_mean_abs_err": float(d.mean()),
            "step0_ref_margin_median": float((top2[:, 0] - top2[:, 1]).median())}


def dist_always():
    """Option "dist_always" (environment default DH_DIST_ALWAYS): a one-rank process group still runs its collectives."""
    try:
        from deephumor_amd import hip
        return bool(hip.option("dist_always"))
    except Exception:                                     # (--stub on a box without the library)
        return os.environ.get("DH_DIST_ALWAYS", "0") not in ("", "0")


# ---- the timed region ----------------------------------------------------------------------------------------------
def timed_region(step_fn, steps, world, device, drain=None):
    """Contract of the brief: barrier + device synchronize, EXACTLY ``steps`` calls of ``step_fn(s)``, device synchronize +
    barrier, wall time = MAX over ranks (one all_reduce).  Returns ``(seconds, last result of step_fn)``.  Used by every
    workload; ``tests/test_dist_cpu.py`` drives it over gloo with a stubbed step.  ``drain``: called once after the last step,
    INSIDE the timed region (finishes an exchange the last step issued asynchronously); its result replaces the last step's."""
    import torch.distributed as dist
    # (a one-rank process group counts when DH_DIST_ALWAYS is set: --rccl-single runs the barriers / the max-reduction / the
    #  all_gather on RCCL with one rank, the only RCCL run a one-GPU box allows)
    multi = (world > 1 or dist_always()) and dist.is_available() and dist.is_initialized()
    cuda = torch.device(device).type == "cuda"

    def fence():
        if cuda:
            torch.cuda.synchronize()
        if multi:
            dist.barrier()
        if cuda:
            torch.cuda.synchronize()

    out = None
    fence()
    t0 = time.perf_counter()
    for s in range(steps):
        out = step_fn(s)
    if drain is not None:
        last = drain()
        out = out if last is None else last
    fence()
    dt = time.perf_counter() - t0
    RANK_TIMES[:] = [dt]
    if multi:
        # one extra all_gather of a float: every rank's own wall time, so the line shows skew between ranks next to the MAX
        mine = torch.tensor([dt], dtype=torch.float64, device=device)
        every = torch.empty((dist.get_world_size(),), dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(every, mine)
        RANK_TIMES[:] = [float(x) for x in every.cpu().tolist()]
        dt = max(RANK_TIMES)
    return dt, out


# ---- legs on the GPU ----------------------------------------------------------------------------------------------
def pipelined_stepper(model, images, img0, n_total, steps, pending, first_seed=100):
    """``step(s)`` for ``timed_region`` on the production schedule (``deephumor_amd.pipeline.CaptionPipeline``): the K batches of the
    timed region as ONE stream of batches -- the encoder of batch i + 1 runs on its own HIP stream while batch i decodes (the decode
    chains of both workloads leave most CUs idle; the encoder is throughput-bound) -- every step still encodes, decodes and exchanges
    exactly one batch, and all K are complete when the region closes (the pipeline is fed exactly K batches; call s returns batch
    s's captions)."""
    from deephumor_amd.dist import gather_captions_async
    from deephumor_amd.pipeline import CaptionPipeline
    kw = dict(max_len=MAX_LEN, beam_size=BEAM, top_k=TOP_K, temperature=TEMP)
    if STREAMS > 1:
        kw["streams"] = STREAMS
    pipe = CaptionPipeline(model, overlap=True, **kw)
    gen = []

    def step(s):
        if not gen:
            gen.append(pipe.run([(images,)] * steps, seeds=range(first_seed, first_seed + steps), img0=img0, to_host=False))
        toks, lens = next(gen[0])
        if toks.is_cuda:                                   # produced on the pipeline's decode stream, consumed on this one
            toks.record_stream(torch.cuda.current_stream())
            lens.record_stream(torch.cuda.current_stream())
        prev, pending[0] = pending[0], gather_captions_async(toks, lens, n_total)
        return prev.wait() if prev is not None else (toks, lens)
    return step


def timed_steps(model, images, rank, n_local, n_total, steps, barrier, watch=None, graph=False, stride=1, pipelined=False):
    from deephumor_amd import hip
    world = n_total // n_local
    pending = [None]          # the batch's all_gather is waited for one step later (inside the timed region: drain_pending)
    if pipelined:
        with torch.no_grad():                             # untimed: the pipeline's streams and buffers exist before the region opens
            warm = pipelined_stepper(model, images, rank * n_local, n_total, 2, pending, first_seed=1)
            warm(0), warm(1)
            drain_pending(pending)
        step = pipelined_stepper(model, images, rank * n_local, n_total, steps, pending)
    else:
        step = lambda s: one_step(model, images, rank * n_local, n_total, seed=100 + s, graph=graph, pending=pending)
    drain = lambda: drain_pending(pending)
    if watch:
        # HIP events around the launches of the roofline kernel INSIDE the timed region.  Each event pair costs ~5 us of stream
        # time: a key launched hundreds of times per step (the C3 decode GEMMs: 192-576) is sampled every `stride`-th launch so
        # that the instrumentation stays below ~0.2 ms per step (an unbiased sample of the same launches)
        with hip.profile(watch=watch, stride=stride) as prof:
            dt, out = timed_region(step, steps, world, images.device, drain=drain)
        return dt, out[1], prof.summary()
    dt, out = timed_region(step, steps, world, images.device, drain=drain)
    return dt, out[1], {}


def synth_images_u8(n, seed=0):
    """The synthetic bench images as DECODED pictures: uint8 [N, 224, 224, 3] whose ToTensor + Normalize image is the randn batch
    quantised to the 8-bit grid (deephumor_demo.ipynb:565-567)."""
    from deephumor_amd.experiments.inference import IMAGENET_MEAN, IMAGENET_STD
    from deephumor_amd.synth import synth_images
    x = synth_images(n, seed=seed)
    m, s = torch.tensor(IMAGENET_MEAN)[None, :, None, None], torch.tensor(IMAGENET_STD)[None, :, None, None]
    return ((x * s + m) * 255.0).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()


def host_inclusive(model, n_local, steps):
    """Pinned host images -> token ids in pinned host memory (SURVEY 8(d)'s metric definition), sequential and pipelined, for the
    two forms a caller can hand images over in: decoded uint8 HWC (38.5 MB per 256 images; ToTensor + Normalize on the device,
    fused with the stem's input packing) and the reference's fp32 NCHW batch (154 MB)."""
    from deephumor_amd.pipeline import CaptionPipeline, u8_preprocess
    from deephumor_amd.synth import synth_images
    kw = dict(max_len=MAX_LEN, beam_size=BEAM, top_k=TOP_K, temperature=TEMP)
    out = {}
    for form, pinned, pre in (("u8_hwc", synth_images_u8(n_local, seed=0).pin_memory(), u8_preprocess(model)),
                              ("f32_nchw", synth_images(n_local, seed=0).pin_memory(), None)):
        res = {"host_bytes_per_step": pinned.numel() * pinned.element_size()}
        for name, overlap in (("sequential", False), ("pipelined", True)):
            pipe = CaptionPipeline(model, overlap=overlap, preprocess=pre, **kw)
            for _ in pipe.run([(pinned,)] * 2, seeds=[1, 2]):
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in pipe.run([(pinned,)] * steps, seeds=range(100, 100 + steps)):
                pass
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[name] = {"value": n_local * steps / dt, "unit": "captions/s", "ms_per_step": dt / steps * 1e3, "batches": steps}
        out[form] = res
    out["note"] = ("images start in pinned host memory, ids end in pinned host memory; pipelined = H2D copy and encoder of batch "
                   "i+1 on their own HIP streams while batch i decodes")
    return out


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

    graph = args.graph
    stride = 1
    with torch.no_grad():
        # untimed: a cold pass (code-object loading, allocator growth), then one fully instrumented pass (HIP events around
        # every launch made through the library) for the per-kernel breakdown and the choice of the roofline kernel
        one_step(model, images, rank * n_local, n_total, seed=0)
        with hip.profile() as prof0:
            one_step(model, images, rank * n_local, n_total, seed=0)
        breakdown = prof0.summary()
        dominant = max(breakdown, key=lambda k: breakdown[k]["ms"])          # ONE kernel template at ONE shape
        for w in range(1, warmup):
            one_step(model, images, rank * n_local, n_total, seed=w, graph=graph)
        pipelined = args.schedule == "pipelined" and not graph
        seq = seq_rank_times = None
        # events only around the launches of the dominant key (a few dozen pairs per step)
        stride = max(1, breakdown[dominant]["calls"] // 32)
        watch = {dominant} if main_line and not graph else None
        if pipelined:
            # the same K steps one batch at a time first (rounds 1-4's headline; kept as value_sequential) -- the roofline kernel's
            # launches are event-timed in THIS region, where the kernel has the chip to itself (in the pipelined region it shares the
            # CUs with the next batch's encoder: its elapsed time there says nothing about the kernel) -- then the production schedule
            ts, _, summary = timed_steps(model, images, rank, n_local, n_total, steps, barrier, watch=watch, stride=stride)
            seq = {"value": n_total * steps / ts, "unit": "captions/s", "ms_per_step": ts / steps * 1e3}
            seq_rank_times = list(RANK_TIMES)             # every rank's wall time of THIS region (the one `value` comes from)
            dt, lens, _ = timed_steps(model, images, rank, n_local, n_total, steps, barrier, pipelined=True)
        elif watch:
            dt, lens, summary = timed_steps(model, images, rank, n_local, n_total, steps, barrier, watch=watch, stride=stride)
        else:
            dt, lens, summary = timed_steps(model, images, rank, n_local, n_total, steps, barrier, graph=graph)
    # `value` is ONE BATCH AT A TIME (rounds 1-4's definition: a step = one pass of the hot path over one batch, nothing of the next
    # batch under it) -- the region the roofline kernel is event-timed in; the production schedule's rate (the same K batches as one
    # pipelined stream, round 5's `value`) is reported next to it as `pipelined` / value_pipelined
    pipe = {"value": n_total * steps / dt, "unit": "captions/s", "ms_per_step": dt / steps * 1e3,
            "schedule": "the K batches of the timed region as one stream of batches (CaptionPipeline): encoder of batch i + 1 on its own HIP "
                        "stream under the decode of batch i; every step = one batch encoded + decoded + exchanged, all K complete inside the region"}
    if seq is None:
        seq, pipe = {"value": n_total * steps / dt, "unit": "captions/s", "ms_per_step": dt / steps * 1e3}, None
    res = {"value": seq["value"], "ms_per_step": seq["ms_per_step"],
           "mean_caption_len": float(lens.float().mean()) if lens is not None else None}
    res["schedule"] = "sequential: one batch at a time" + (" (hipGraph replay)" if graph else "")
    res["sequential"] = seq
    if pipe is not None:
        res["pipelined"] = pipe
    # whole-step fraction of the dense 16-bit MFMA peak: the algorithmic flops of every launch of one step (the library attaches them to
    # each launch; instrumented pass) over the sequential step time
    step_flops = sum(v["flops"] for v in breakdown.values())
    peak = PEAK_F32_TFLOPS if dtype == "f32" else PEAK_16_TFLOPS
    res["whole_step"] = {"algorithmic_tflop": step_flops / 1e12, "launches": sum(v["calls"] for v in breakdown.values()),
                         "achieved_tflops": step_flops / 1e12 / (seq["ms_per_step"] / 1e3), "peak_tflops": peak,
                         "mfma_frac": step_flops / 1e12 / (seq["ms_per_step"] / 1e3) / peak}
    by_entry = {}
    for k, v in breakdown.items():
        base = re.sub(r"\{.*\}$", "", k)
        by_entry[base] = by_entry.get(base, 0.0) + v["ms"]
    res["kernel_breakdown_ms_per_step"] = {k: round(v, 3) for k, v in sorted(by_entry.items(), key=lambda kv: -kv[1])}
    res["kernel_breakdown_note"] = ("untimed pass with HIP events around every launch: the events add ~15 % to a chain of "
                                    "small launches; per-kernel shares of the clean run: profiles/r5/*_kernel_stats.csv (rocprofv3)")
    src = summary if dominant in summary else breakdown
    res["roofline"] = attach_traffic(price(dominant, src[dominant], dtype), workload, dtype)
    res["roofline"]["measured"] = ((f"timed region, every {stride}th launch" if main_line and not graph and stride > 1 else "timed region, every launch")
                                   if dominant in summary else "instrumented pass")
    if pipe is not None and dominant in summary:
        res["roofline"]["measured"] += " of the timed region `value` comes from (one batch at a time: the kernel alone on the chip)"
    if workload in ("c3", "c5"):
        res["roofline"]["note"] = ("the Transformer step is ~53 dependent launches per position; its largest launch KEY is a 5 - 6 us decode GEMM "
                                   "at the dependent-launch floor (DESIGN section 12), so this fraction prices launch latency, not bandwidth -- the "
                                   "kernels north_star names are priced in roofline_self_attention / roofline_cross_attention / "
                                   "roofline_decoder_attention_combined")
    rt = seq_rank_times or list(RANK_TIMES)
    res["per_rank_ms_per_step"] = {"min": min(rt) / steps * 1e3, "max": max(rt) / steps * 1e3, "ranks": len(rt), "skew_ms": (max(rt) - min(rt)) / steps * 1e3}
    res["encoder_layers"] = encoder_table(breakdown, dtype)
    if workload in ("c3", "c5"):
        for name, entry in (("roofline_self_attention", "dh_attn_self_decode"), ("roofline_cross_attention", "dh_attn_cross_decode")):
            m = merge_keys(breakdown, lambda k: k.startswith(entry))
            if m:
                res[name] = attach_traffic(price(entry, m, dtype), workload, dtype)
                res[name]["measured"] = "instrumented pass (events around every launch; rocprofv3 kernel-only times are ~2 us lower)"
                if name == "roofline_cross_attention":
                    res[name]["launch"] = "the packed cross-attention launch alone; fc_q is its own register-stationary GEMM in front of it"
        sa, ca = res.get("roofline_self_attention"), res.get("roofline_cross_attention")
        if sa and ca:
            tot_b = sa["algorithmic_bytes_per_launch"] * sa["launches"] + ca["algorithmic_bytes_per_launch"] * ca["launches"]
            tot_s = (sa["avg_launch_us"] * sa["launches"] + ca["avg_launch_us"] * ca["launches"]) * 1e-6
            res["roofline_decoder_attention_combined"] = {"bound": "hbm", "achieved": tot_b / tot_s / 1e9, "peak": PEAK_HBM_GBS,
                                                          "unit": "GB/s", "frac": tot_b / tot_s / 1e9 / PEAK_HBM_GBS}
    if world == 1 and not graph:
        with torch.no_grad():
            one_step(model, images, 0, n_total, seed=1, graph=True)
            tg, _, _ = timed_steps(model, images, 0, n_local, n_total, steps, barrier, graph=True)
        res["hipgraph_replay"] = {"value": n_total * steps / tg, "unit": "captions/s", "ms_per_step": tg / steps * 1e3}
    if world == 1 and not args.quick:
        with torch.no_grad():
            res["host_inclusive"] = host_inclusive(model, n_local, max(10, steps))    # enough batches for the pipeline's steady state
    if world == 1 and not args.quick and args.schedule != "pipelined":
        with torch.no_grad():
            # the same device-resident batches as `value`, but as a STREAM of batches through CaptionPipeline: the encoder of batch
            # i+1 runs on its own HIP stream while batch i decodes (reported next to `value`, which times one batch at a time)
            from deephumor_amd.pipeline import CaptionPipeline
            pipe = CaptionPipeline(model, overlap=True, max_len=MAX_LEN, beam_size=BEAM, top_k=TOP_K, temperature=TEMP)
            nb = max(10, steps)
            for _ in pipe.run([(images,)] * 2, seeds=[1, 2], to_host=False):
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in pipe.run([(images,)] * nb, seeds=range(100, 100 + nb), to_host=False):
                pass
            torch.cuda.synchronize()
            tp = time.perf_counter() - t0
            res["pipelined_device_resident"] = {"value": n_local * nb / tp, "unit": "captions/s", "ms_per_step": tp / nb * 1e3, "batches": nb}
    if rank == 0 and with_cpu and dtype != "f32":
        # the other 16-bit storage type, timed straight after the main legs (the same schedule as `value`; behind the fp32 legs below
        # the same measurement read 24 - 37 k captions/s from run to run -- whatever state those long legs leave the part in)
        other = "f16" if dtype == "bf16" else "bf16"
        m16 = build_model(workload, dev, other)[0]
        with torch.no_grad():
            for w in range(3):
                one_step(m16, images, 0, n_total, seed=w)
            t16, _, _ = timed_steps(m16, images, 0, n_local, n_total, 10, barrier)
        res[f"{other}_path"] = {"value": n_total * 10 / t16, "unit": "captions/s", "ms_per_step": t16 / 10 * 1e3, "steps": 10,
                                "schedule": "sequential (as `value`)"}
        del m16
        torch.cuda.empty_cache()
    if rank == 0 and with_cpu:
        # the parity gate is the fp32 path: bit-exact greedy ids vs the CPU reference path -- timed here too
        if dtype == "f32":
            m32 = model
        else:
            del model
            torch.cuda.empty_cache()
            m32 = build_model(workload, dev, "f32")[0]
        res["greedy_token_match_vs_cpu_ref"] = greedy_match(workload, m32, sd, hp, N_CHECK)
        if dtype != "f32":
            ref = greedy_all(m32, images)                 # all bench images on the bit-exact path: the reference of `precision`
            with torch.no_grad():
                one_step(m32, images, 0, n_total, seed=0)
                t32, _, _ = timed_steps(m32, images, 0, n_local, n_total, 3, barrier)
            res["fp32_parity_path"] = {"value": n_total * 3 / t32, "unit": "captions/s", "ms_per_step": t32 / 3 * 1e3, "steps": 3,
                                       "note": "same step, fp32 storage + exact-fp32 arithmetic: the path whose greedy ids are bit-exact (sequential schedule, as "
                                               "parity_grade_path: compare these two with each other and with value_sequential)"}
            del m32
            torch.cuda.empty_cache()
            # the parity-grade matrix-core path: the same fp32 model with option f32_split (every dense layer / convolution as three
            # fp16 MFMAs on split operands, csrc/gemm_f32x.hip): fp32-class results -- its greedy ids are held to the same gates as
            # the exact-fp32 path's (tests/test_f32x_gpu.py) -- at a multiple of the exact path's rate
            from deephumor_amd import hip as _hip
            with _hip.option_scope(f32_split=1):
                mx = build_model(workload, dev, "f32")[0]
                gx = greedy_match(workload, mx, sd, hp, N_CHECK)
                cmpx = compare_greedy(ref, greedy_all(mx, images))
                with torch.no_grad():
                    one_step(mx, images, 0, n_total, seed=0)
                    tx, _, _ = timed_steps(mx, images, 0, n_local, n_total, 5, barrier)
                del mx
            torch.cuda.empty_cache()
            res["parity_grade_path"] = {"value": n_total * 5 / tx, "unit": "captions/s", "ms_per_step": tx / 5 * 1e3, "steps": 5,
                                        "greedy_token_match_vs_cpu_ref": gx, "vs_exact_fp32_hip_all_images": cmpx,
                                        "speedup_over_exact_fp32_path": t32 / 3 / (tx / 5),
                                        "note": "fp32 storage, dense layers + convolutions as split-operand fp16 MFMAs (hi*hi + hi*lo + lo*hi, fp32 "
                                                "accumulation): logits within 1e-3 and greedy ids bit-exact vs the reference in tests/test_f32x_gpu.py"}
            # what 16-bit storage costs in greedy tokens: both 16-bit paths against the fp32 HIP path (bit-exact vs the CPU
            # oracle on the checked images above and vs the reference's captions on sixteen rows in tests/test_fullsize_gpu.py) over ALL bench images
            res["precision_vs_fp32_hip"] = {"reference": "fp32 HIP path, greedy (beam 1, top_k 1), all bench images",
                                            "table": "profiles/r3/precision_c2.json, precision_c3.json (which tensor's precision buys what; round 3)"}
            for dt in [dtype] + [d for d in ("bf16", "f16") if d != dtype]:
                model = build_model(workload, dev, dt)[0]
                res["precision_vs_fp32_hip"][dt] = compare_greedy(ref, greedy_all(model, images))
                if dt == dtype:
                    res[f"greedy_token_match_{dtype}_vs_cpu_ref"] = greedy_match(workload, model, sd, hp, N_CHECK)
                del model
                torch.cuda.empty_cache()
            del ref
            model = None
        res["cpu_baseline"] = cpu_baseline(workload, sd, hp)
        res["speedup_vs_cpu"] = res["value"] / res["cpu_baseline"]["value"]
    torch.cuda.empty_cache()
    return res


def run_c5(args, rank, world, dev, dtype):
    """BASELINE config 5: the full 300-template sweep (ImageLabelEncoder + CaptioningTransformer, beam 10), templates
    sharded 38/38/38/38/37/37/37/37 over 8 ranks (all 300 on one), uneven shards gathered with one padded all_gather."""
    import numpy as np
    import torch.distributed as dist
    from deephumor_amd.dist import gather_captions, shard_range
    from deephumor_amd.synth import synth_images
    model, sd, hp = build_model("c5", dev, dtype)
    n_total = 300
    lo, hi = shard_range(n_total, rank, world)
    images = synth_images(hi - lo, seed=2, first=lo).to(dev)
    g = np.random.Generator(np.random.Philox(key=[1, 0]))
    labels = torch.from_numpy(g.integers(6, V_WORD, size=(n_total, 3)).astype(np.int64))[lo:hi].to(dev)

    def sweep(seed):
        toks, lens = model.generate_batch(images, labels, max_len=MAX_LEN, beam_size=10, top_k=TOP_K, temperature=TEMP,
                                          seed=seed, img0=lo)
        return gather_captions(toks, lens, n_total)

    with torch.no_grad():
        sweep(0)
        reps = max(1, args.steps)
        dt, (toks, lens) = timed_region(lambda s: sweep(100 + s), reps, world, dev)
    return {"workload": workload_name("c5"), "value": n_total * reps / dt, "unit": "captions/s", "sweeps": reps,
            "ms_per_sweep": dt / reps * 1e3, "templates": n_total, "gathered": int(toks.shape[0]), "dtype": dtype,
            "mean_caption_len": float(lens.float().mean())}


def run_shard(args, dev, dtype):
    """``--shard-of W [--shard-rank R]``: ONE GPU times exactly what rank R of a W-rank run would do -- its image shard (C2 / C3:
    ``--batch`` images of a W x batch global batch, global indices from R x batch; C5: ``shard_range(300, R, W)`` = 38 or 37
    templates x beam 10), followed by the per-batch all_gather of the ``[cap, T + 1]`` payload a rank contributes, through the
    one-rank RCCL group when ``--rccl-single`` is given.  Printed next to the measured shard time: the same workload's FULL batch on
    this one GPU (C5: all 300 templates) and ``projected_value`` = n_total / shard time -- a PROJECTION (W identical GPUs, no
    straggler, an all_gather no slower than the one-rank one), labelled as such; it is not a multi-GPU measurement."""
    import numpy as np
    import torch.distributed as dist
    from deephumor_amd import hip
    from deephumor_amd.dist import shard_range
    from deephumor_amd.synth import synth_images
    wl = {"both": "c2"}.get(args.workload, args.workload)
    W, R = args.shard_of, args.shard_rank
    model, sd, hp = build_model(wl, dev, dtype)
    labels_all = None
    if wl == "c5":
        n_total, beam, seed_img = 300, 10, 2
        lo, hi = shard_range(n_total, R, W)
        g = np.random.Generator(np.random.Philox(key=[1, 0]))
        labels_all = torch.from_numpy(g.integers(6, V_WORD, size=(n_total, 3)).astype(np.int64)).to(dev)
    else:
        n_total, beam, seed_img = args.batch * W, BEAM, 0
        lo, hi = R * args.batch, (R + 1) * args.batch
    kw = dict(max_len=MAX_LEN, beam_size=beam, top_k=TOP_K, temperature=TEMP)
    cap = -(-n_total // W)
    use_pg = dist.is_available() and dist.is_initialized()

    def exchange(toks, lens):
        """What gather_captions does on a W-rank group, for this rank's payload: pad to the largest shard, one all_gather."""
        packed = torch.zeros((max(cap, toks.shape[0]), toks.shape[1] + 1), dtype=torch.int64, device=toks.device)
        packed[:toks.shape[0], :-1] = toks
        packed[:toks.shape[0], -1] = lens
        if use_pg:
            out = torch.empty((dist.get_world_size() * packed.shape[0], packed.shape[1]), dtype=torch.int64, device=toks.device)
            dist.all_gather_into_tensor(out, packed)
            return out
        return packed

    def timed(a, b, steps, warm):
        imgs = synth_images(b - a, seed=seed_img, first=a).to(dev)
        lab = () if labels_all is None else (labels_all[a:b],)
        step = lambda s: exchange(*model.generate_batch(imgs, *lab, seed=100 + s, img0=a, **kw))
        with torch.no_grad():
            for w in range(max(1, warm)):
                step(-1 - w)
            with hip.profile() as prof:
                step(0)
            dt, _ = timed_region(step, steps, 1, dev)
        bd = prof.summary()
        by_entry = {}
        for k, v in bd.items():
            base = re.sub(r"\{.*\}$", "", k)
            e = by_entry.setdefault(base, [0, 0.0])
            e[0] += v["calls"]; e[1] += v["ms"]
        launches = sum(v["calls"] for v in bd.values())
        return dt / steps, {"launches_per_step": launches,
                            "event_timed_ms_per_step": {k: [c, round(ms, 3)] for k, (c, ms) in sorted(by_entry.items(), key=lambda kv: -kv[1][1])}}

    t_shard, bd_shard = timed(lo, hi, args.steps, args.warmup)
    full_n = n_total if wl == "c5" else args.batch            # C2 / C3 are weak scaling: the 1-GPU workload IS one shard
    if wl == "c5" and not args.shard_only:
        t_full, bd_full = timed(0, n_total, max(2, args.steps // 2), 1)
    else:
        t_full, bd_full = t_shard, None
    res = {"workload": workload_name(wl), "dtype": dtype,
           "shard": {"of": W, "rank": R, "images": hi - lo, "rows_per_position": (hi - lo) * beam, "img0": lo, "shard_ms": t_shard * 1e3,
                     "exchange": (f"one-rank RCCL all_gather_into_tensor of the [{cap}, {MAX_LEN + 1}] int64 payload per batch" if use_pg else
                                  "payload packed, no process group (run with --rccl-single for the collective)"),
                     "breakdown": bd_shard},
           "one_gpu": {"images": full_n, "ms": t_full * 1e3, "value": full_n / t_full, "unit": "captions/s"},
           "projection": {"label": f"PROJECTION, not a measurement: n_total / shard time of rank {R} (the largest shard for rank 0), assuming {W} "
                                   "identical GPUs, no straggler and an all_gather no slower than the one-rank one",
                          "n_total": n_total, f"projected_{W}gpu_value": n_total / t_shard, "unit": "captions/s",
                          f"projected_speedup_over_one_gpu": (n_total / t_shard) / (full_n / t_full),
                          "scaling": "strong" if wl == "c5" else "weak"}}
    if bd_full is not None:
        res["one_gpu"]["breakdown"] = bd_full
    res["options"] = {k: v[0] for k, v in hip.options().items()}
    return res


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

    from deephumor_amd import hip
    pp = run()
    with hip.profile() as prof0:                          # untimed instrumented pass: which launch key dominates the pass
        run()
    breakdown = prof0.summary()
    dominant = max(breakdown, key=lambda k: breakdown[k]["ms"])
    reps = max(1, args.steps)
    stride = max(1, breakdown[dominant]["calls"] // 32)
    with hip.profile(watch={dominant}, stride=stride) as prof:       # events around the dominant key's launches inside the timed region
        dt, pp = timed_region(lambda s: run(), reps, world, dev)
    summary = prof.summary()
    res = {"workload": f"teacher-forced scoring, {kind_of(kind)}, 300 templates x 30 captions x 32 tokens, V={V_WORD}",
           "value": n_total * reps / dt, "unit": "captions scored/s", "passes": reps, "ms_per_pass": dt / reps * 1e3,
           "captions": n_total, "mean_perplexity": float(pp.float().mean()), "dtype": dtype}
    res["roofline"] = price(dominant, (summary if dominant in summary else breakdown)[dominant], dtype)
    res["roofline"]["measured"] = f"timed region, every {stride}th launch" if dominant in summary else "instrumented pass"
    by_entry = {}
    for k, v in breakdown.items():
        base = re.sub(r"\{.*\}$", "", k)
        by_entry[base] = by_entry.get(base, 0.0) + v["ms"]
    res["kernel_breakdown_ms_per_pass"] = {k: round(v, 3) for k, v in sorted(by_entry.items(), key=lambda kv: -kv[1])}
    if rank == 0 and world == 1 and not args.no_cpu and not args.quick:
        res["cpu_baseline"] = cpu_baseline_score(kind, sd, hp, images.cpu(), tpl, caps.cpu(), lengths)
        res["speedup_vs_cpu"] = res["value"] / res["cpu_baseline"]["value"]
    return res


def cpu_baseline_score(kind, sd, hp, images, tpl, caps, lengths):
    """The oracle's teacher-forced scoring (the reference's forward + perplexity per (template, caption) pair, trainer.py:63-81 /
    metrics.py:4-9; the encoder re-run per pair as the reference's evaluation loop does) on a bounded sample of the same corpus:
    torch thread-count sweep on 2 pairs, then about 12 s of pairs at the best thread count."""
    from oracle import ref_path as R
    k = kind_of(kind)

    def score(i):
        logits = R.model_forward(k, sd, hp, images[tpl[i]:tpl[i] + 1], caps[i:i + 1, :-1])[:, :caps.shape[1]]
        return float(R.perplexity(logits, caps[i:i + 1], lengths[i:i + 1]))

    t_all = time.perf_counter()
    saved = torch.get_num_threads()
    sweep = {}
    cands = sorted({t for t in (8, 16, 32, 64) if t <= (os.cpu_count() or 8)} | {min(8, os.cpu_count() or 8)})
    for th in cands:
        torch.set_num_threads(th)
        score(0)
        t0 = time.perf_counter()
        score(1), score(2)
        sweep[th] = (time.perf_counter() - t0) / 2
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    n = int(max(4, min(200, 12.0 // sweep[best])))
    t0 = time.perf_counter()
    pps = [score(3 + i) for i in range(n)]
    dt = time.perf_counter() - t0
    torch.set_num_threads(saved)
    return {"value": n / dt, "unit": "captions scored/s", "cores": best, "kind": "port", "host_cpus": os.cpu_count(),
            "thread_sweep_s_per_pair": {str(k_): round(v, 3) for k_, v in sweep.items()}, "mean_perplexity_of_sample": sum(pps) / n,
            "sample": f"{n} (template, caption) pairs of the same corpus, one pass, {k} V={V_WORD}, per-pair encoder + teacher-forced forward + "
                      f"perplexity (reference evaluation loop), {time.perf_counter() - t_all:.1f} s of CPU work in all"}


def git_head():
    """Commit of the tree being measured: from git where the checkout has one, else the stamp ``deephumor_amd/_build.py`` left next
    to the library when it was built (the GPU box receives a snapshot without ``.git``)."""
    try:
        out = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=5).stdout.strip()
        if out:
            return out
    except Exception:
        pass
    try:
        with open(os.path.join(ROOT, "deephumor_amd", "lib", "BUILD_COMMIT")) as f:
            return f.read().strip() or None
    except OSError:
        return None


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """``python bench.py --gpus N`` without a launcher: THIS process (which has only imported torch -- no HIP call, no
    ``torch.cuda.is_available()``) builds the library once, then starts ``python -m torch.distributed.run`` as a CHILD with one
    rank per GPU, relays rank 0's JSON line as its own last stdout line and returns the child's exit code.  Nothing that has
    touched a GPU ever re-execs, and the parent never initialises HIP."""
    if not args.stub:
        from deephumor_amd import _build
        _build.build()                                    # ranks only load the prebuilt .so (no compile under the launcher)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               DH_BENCH_LAUNCHED_BY="bench.py self-launch (child torch.distributed.run)")
    port = int(os.environ.get("MASTER_PORT", 0)) or free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    line = None
    for out in child.stdout:                              # the ranks' stderr passes straight through
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        else:
            print(out, flush=True)
    rc = child.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 3
        print(f"bench.py: {args.gpus} ranks exited 0 but rank 0 printed no result line", file=sys.stderr)
    return rc


def stub_workload(args, rank, world, dev):
    """``--stub`` (tests/test_dist_cpu.py): the whole control flow of a multi-rank run -- launcher, process group, image shards,
    barrier / K steps / barrier, MAX over ranks, one all_gather per step, rank 0's line -- with the model replaced by a pure
    function of the GLOBAL image index, on CPU over gloo.  Its line says ``"data": "stub"``: it is never a measurement."""
    from deephumor_amd.dist import gather_captions_async, shard_range
    if args.workload == "c5":
        # BASELINE config 5's partitioning: 300 templates in contiguous, UNEVEN shards (8 ranks: 38 x 4 + 37 x 4), strong scaling
        n_total = 300
        lo, hi = shard_range(n_total, rank, world)
    else:
        n_local, n_total = args.batch, args.batch * world
        lo, hi = rank * n_local, (rank + 1) * n_local
    idx = torch.arange(lo, hi)
    calls = []
    pending = [None]

    def step(s):          # as one_step(pending=...): this step's exchange is issued asynchronously, the previous one's waited for
        calls.append(s)
        time.sleep(0.01 * (rank + 1))
        toks = (idx[:, None] * 10 + torch.arange(MAX_LEN)[None, :] + s) % 97
        prev, pending[0] = pending[0], gather_captions_async(toks, (idx % MAX_LEN) + 1, n_total)
        return prev.wait() if prev is not None else None

    for w in range(args.warmup):
        step(-1 - w)
    drain_pending(pending)
    del calls[:]
    dt, (toks, lens) = timed_region(step, args.steps, world, dev, drain=lambda: drain_pending(pending))
    want = (torch.arange(n_total)[:, None] * 10 + torch.arange(MAX_LEN)[None, :] + args.steps - 1) % 97
    return {"value": n_total * args.steps / dt, "ms_per_step": dt / args.steps * 1e3, "steps_run": calls,
            "per_rank_ms_per_step": {"min": min(RANK_TIMES) / args.steps * 1e3, "max": max(RANK_TIMES) / args.steps * 1e3, "ranks": len(RANK_TIMES),
                                     "skew_ms": (max(RANK_TIMES) - min(RANK_TIMES)) / args.steps * 1e3},
            "shard": [lo, hi], "shard_sizes": [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
            if args.workload == "c5" else [args.batch] * world,
            "gathered": int(toks.shape[0]), "gather_in_global_order": bool(torch.equal(toks, want)),
            "mean_caption_len": float(lens.float().mean())}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU (BASELINE configs: 256)")
    ap.add_argument("--workload", choices=["c2", "c3", "both", "c5", "score-c2", "score-c3"], default="both")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline / fp32 parity legs")
    ap.add_argument("--quick", action="store_true", help="only the timed steps and the roofline (no host-inclusive / CPU legs)")
    ap.add_argument("--score-batch", type=int, default=1024, help="captions per teacher-forced batch (score-* workloads)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (the in-library event "
                    "profiler of the roofline line needs real launches, so the roofline then comes from the instrumented pass)")
    ap.add_argument("--dtype", choices=["bf16", "f16", "f32"], default=None,
                    help="storage/MFMA operand type of the measured path (BASELINE configs C2-C4: bf16, C5: fp16)")
    ap.add_argument("--rccl-single", action="store_true", help="with --gpus 1: create the RCCL process group anyway (one rank) and run the "
                    "barriers, the max-reduction and the per-batch all_gather of the N > 1 path through it")
    ap.add_argument("--shard-of", type=int, default=0, help="with --gpus 1: time the shard rank --shard-rank of a W-rank run would decode "
                    "(C5: 38 / 37 of the 300 templates; C2 / C3: --batch images of a W x batch global batch) and print a labelled projection")
    ap.add_argument("--shard-rank", type=int, default=0)
    ap.add_argument("--shard-only", action="store_true", help="with --shard-of: skip the full-batch leg (profiling runs)")
    ap.add_argument("--schedule", choices=["pipelined", "sequential"], default="pipelined",
                    help="pipelined (default): the timed region's K batches as one stream of batches, the next batch's encoder overlapping the "
                         "current batch's decode (deephumor_amd.pipeline.CaptionPipeline); sequential: one batch at a time (the rounds 1-4 headline)")
    ap.add_argument("--stub", action="store_true", help="CPU/gloo control-flow test of the multi-rank path (model stubbed; not a measurement)")
    args = ap.parse_args(argv)
    if args.dtype is None:
        args.dtype = "f16" if args.workload == "c5" else "bf16"

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, argv)                   # before anything touches a GPU
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        return 2
    launched_by = os.environ.get("DH_BENCH_LAUNCHED_BY") or ("external launcher (torch.distributed.run env)" if "WORLD_SIZE" in os.environ
                                                             else "in-process (single rank)")
    if args.stub:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    n_ranks_seen = 1
    single_pg = args.rccl_single and world == 1 and not args.stub
    if single_pg:
        os.environ["DH_DIST_ALWAYS"] = "1"             # default source of option "dist_always" (the library is loaded later)
        os.environ.setdefault("MASTER_PORT", str(free_port()))
    if world > 1 or single_pg:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.stub:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        n_ranks_seen = dist.get_world_size()

    def finish(line):
        line["n_ranks_seen"] = n_ranks_seen          # what the process group itself reports (RCCL saw N ranks)
        line["ranks_launched_by"] = launched_by
        if single_pg:
            line["rccl_single_rank"] = "process group 'nccl' (RCCL) with one rank: barrier, all_reduce(MAX) and all_gather_into_tensor ran through it"
        if rank == 0:
            # RCCL writes a version banner through C stdio at communicator creation; piped, that buffer would only be flushed at
            # exit -- AFTER the JSON line.  Flush it first so that the JSON line is the last line of stdout.
            import ctypes
            ctypes.CDLL(None).fflush(None)
            print(json.dumps(line), flush=True)
        if world > 1 or single_pg:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return 0

    if args.stub:
        res = stub_workload(args, rank, world, dev)
        c5 = args.workload == "c5"
        return finish(dict(res, metric="captions/sec (224x224, 32-tok, beam=10, 300-template sweep)" if c5 else "captions/sec (224x224, 32-tok, beam=5)",
                           unit="captions/s", n_gpus=world, steps=args.steps, warmup=args.warmup, higher_is_better=True,
                           scaling="strong" if c5 else "weak", vs_baseline=None, dtype="none", data="stub",
                           config={"workload": "STUB: control flow only, no model, CPU/gloo" + (" (C5 partitioning: 300 templates, uneven shards)" if c5 else "")}))

    if args.shard_of:
        if world != 1 or not 0 <= args.shard_rank < args.shard_of or args.workload.startswith("score-"):
            print("bench.py: --shard-of needs --gpus 1, 0 <= --shard-rank < --shard-of and a decode workload", file=sys.stderr)
            return 2
        res = run_shard(args, dev, args.dtype)
        wl = {"both": "c2"}.get(args.workload, args.workload)
        return finish(dict(res, metric="captions/sec of ONE rank's shard on one GPU + projected whole-job rate (see projection.label)",
                           value=res["shard"]["images"] / (res["shard"]["shard_ms"] / 1e3), unit="captions/s", n_gpus=1, steps=args.steps,
                           warmup=args.warmup, ms_per_step=res["shard"]["shard_ms"], higher_is_better=True, scaling=res["projection"]["scaling"],
                           vs_baseline=None, data="synthetic",
                           config={"workload": workload_name(wl), "shard_of": args.shard_of, "shard_rank": args.shard_rank,
                                   "vocab": V_WORD, "max_len": MAX_LEN, "top_k": TOP_K, "commit": git_head()}))
    if args.workload.startswith("score-"):
        res = run_score(args, rank, world, dev, args.dtype, args.workload.split("-")[1])
        return finish(dict(res, metric="captions scored/sec (teacher-forced perplexity, 32 tokens)", n_gpus=world,
                           higher_is_better=True, scaling="strong", vs_baseline=None, data="synthetic"))
    if args.workload == "c5":
        res = run_c5(args, rank, world, dev, args.dtype)
        return finish(dict(res, metric="captions/sec (224x224, 32-tok, beam=10, 300-template sweep)", n_gpus=world,
                           higher_is_better=True, scaling="strong", vs_baseline=None, data="synthetic",
                           config={"workload": workload_name("c5"), "beam_size": 10, "top_k": TOP_K, "vocab": V_WORD,
                                   "parallelism": f"templates sharded x{world} (uneven shards), one padded all_gather of ids"}))
    main_wl = "c2" if args.workload in ("c2", "both") else "c3"
    with_cpu = (world == 1) and not args.no_cpu and not args.quick
    res = run_workload(main_wl, args, rank, world, dev, args.steps, args.warmup, with_cpu, args.dtype)
    line = {
        "metric": "captions/sec (224x224, 32-tok, beam=5)", "value": res["value"], "unit": "captions/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": workload_name(main_wl),
                   "images_per_gpu": args.batch, "global_batch": args.batch * world, "vocab": V_WORD,
                   "max_len": MAX_LEN, "beam_size": BEAM, "top_k": TOP_K, "temperature": TEMP,
                   "parallelism": f"image-sharded x{world}, one all_gather of token ids per batch",
                   "weights": "synthetic name-keyed (seed 1234)", "encoder_in_timed_region": True,
                   "inputs": "device-resident fp32 NCHW images (host-inclusive rates under host_inclusive)",
                   "hipgraph": bool(args.graph), "commit": git_head()},
        "roofline": res["roofline"], "cpu_baseline": res.get("cpu_baseline"),
        # which rate `value` is: the bench contract's (inputs already resident in HBM when the timed region starts; a PCIe-inclusive rate
        # is never `value`).  SURVEY 8(d) / BASELINE.md section 3 define the metric host images -> host ids: that one is
        # `value_host_inclusive` (pipelined, decoded uint8 images in pinned memory) with its ratio to `value` next to it
        "value_def": "device-resident fp32 NCHW images -> token ids on the device (bench contract), ONE BATCH AT A TIME (= value_sequential, the "
                     "definition of rounds 1-4 and of the timed region the roofline kernel is measured in); the same K batches as one pipelined "
                     "stream (round 5's `value`): value_pipelined; host -> host (SURVEY 8(d)): value_host_inclusive.  speedup_vs_cpu = value / "
                     "cpu_baseline.value.  Top-level workload = BASELINE.json configs[1] (C2, the configuration the metric is quoted on that fits "
                     "one GPU -- the bench contract's rule); C3, the Transformer configuration north_star's roofline target names, is the nested "
                     "`c3` object with the same keys",
    }
    line["config"]["schedule"] = res.get("schedule")
    line["value_sequential"] = res["sequential"]["value"]
    if "pipelined" in res:
        line["value_pipelined"] = res["pipelined"]["value"]
        line["ms_per_step_pipelined"] = res["pipelined"]["ms_per_step"]
        line["pipelined_schedule"] = res["pipelined"]["schedule"]
    line["whole_step"] = res["whole_step"]
    line["whole_step_mfma_frac"] = res["whole_step"]["mfma_frac"]
    for k in ("greedy_token_match_vs_cpu_ref", f"greedy_token_match_{args.dtype}_vs_cpu_ref", "precision_vs_fp32_hip", "speedup_vs_cpu", "pipelined_device_resident",
              "mean_caption_len", "fp32_parity_path", "parity_grade_path", "f16_path", "bf16_path", "host_inclusive", "hipgraph_replay", "kernel_breakdown_ms_per_step", "kernel_breakdown_note",
              "encoder_layers", "roofline_self_attention", "roofline_cross_attention", "roofline_decoder_attention_combined"):
        if k in res:
            line[k] = res[k]
    for k in ("per_rank_ms_per_step",):
        if k in res:
            line[k] = res[k]
    # the second half of the metric ("greedy-decode token match vs CPU ref") next to `value`: the fp32 path's match against the CPU
    # oracle (bit-exact: 1.0), and what each 16-bit storage type keeps of the fp32 path's greedy tokens over all bench images
    pv = res.get("precision_vs_fp32_hip")
    if pv:
        line["greedy_token_match_vs_fp32_hip"] = {dt: pv[dt]["token_match"] for dt in ("bf16", "f16") if dt in pv}
        line["greedy_captions_identical_vs_fp32_hip"] = {dt: pv[dt]["captions_identical"] for dt in ("bf16", "f16") if dt in pv}
    for dt in ("bf16", "f16"):
        if f"{dt}_path" in res:
            line[f"value_{dt}"] = res[f"{dt}_path"]["value"]
    line[f"value_{args.dtype}"] = res["value"]
    if "parity_grade_path" in res:
        line["value_parity_grade"] = res["parity_grade_path"]["value"]
    if "hipgraph_replay" in res:
        line["value_hipgraph_replay"] = res["hipgraph_replay"]["value"]
    hi = res.get("host_inclusive")
    if hi:
        # SURVEY 8(d)'s definition of the metric (images in host memory -> ids in host memory), next to `value` (inputs resident
        # in HBM, as the bench contract prescribes): decoded uint8 images in pinned memory, copy + preprocessing + encoder of
        # batch i+1 overlapped with the decode of batch i
        line["value_host_inclusive"] = hi["u8_hwc"]["pipelined"]["value"]
        line["value_host_inclusive_over_value"] = hi["u8_hwc"]["pipelined"]["value"] / res["value"]
        line["value_host_inclusive_def"] = ("uint8 HWC 224x224 images in pinned host memory -> token ids in pinned host memory, pipelined "
                                            "(CaptionPipeline + u8_preprocess); the fp32 NCHW form is under host_inclusive.f32_nchw")
    if args.workload == "both":
        r3 = run_workload("c3", args, rank, world, dev, max(3, args.steps // 2), 1, with_cpu, args.dtype, main_line=False)
        r3.pop("encoder_layers", None)
        line["c3"] = dict(r3, workload=workload_name("c3") + ", same batch/beam settings", unit="captions/s")
    return finish(line)


if __name__ == "__main__":
    sys.exit(main())
