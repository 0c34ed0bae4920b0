#!/usr/bin/env python3
"""Reduce the per-kernel PMC CSVs of tools/collect_profiles.sh to profiles/<round>/pmc_hbm_traffic.json, the file bench.py reads
for `roofline.traffic` (rocprofv3 --pmc cannot run inside the bench process).

traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE, both counters in KB: on gfx950 FETCH_SIZE tallies the 128-byte requests of
wide coalesced reads at 64 bytes and has to be doubled, WRITE_SIZE is exact for 16-byte-per-lane stores
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section).  Separate passes per counter (they cannot share one on gfx950).

    python tools/make_pmc_json.py profiles/r4 [commit]
"""
import csv
import json
import subprocess
import sys
from pathlib import Path

# bench.py roofline key -> (workload, kernel-name fragment, workgroups or None, note)
KEYS = [
    ("c2", "dh_linear[vocab]{1280x36541x512}", "vocab_wreg_kernel", 256,
     "classifier + bias with the weights streamed from L2 into registers (fragment-packed, 16 row-block workgroups per weight chunk on "
     "one XCD) and 80-row activation blocks in LDS, fp32 logits out (non-temporal) into a row stride of whole 256-column chunks"),
    ("c3", "dh_linear[vocab]{1280x36541x512}", "vocab_areg256_kernel", 256, "same kernel, same shape (256 images x 5 beams)"),
    ("c3", "dh_attn_self_decode", "attn_decode_reg_kernel", 2048,
     "launch-weighted mean over the history depths of one sweep (2..22 keys per row)"),
    ("c3", "dh_attn_cross_decode", "attn_cross_mfma_kernel", 512,
     "the packed cross-attention launch (fc_q is its own GEMM in front of it)"),
    ("c3", "dh_linear[ffn]{1280x512x2048}", "linear_wreg_kernelIDF16bLi4ELi40ELi4ELi1E", 256,
     "fc_2 of the feed-forward layer on the register-stationary kernel (64 columns x 40 rows per workgroup, K = 2,048)"),
    ("c3", "dh_linear[ffn]{1280x2048x512}", "linear_wreg_kernelIDF16bLi8ELi80ELi1ELi0E", 256,
     "fc_1 of the feed-forward layer on the register-stationary kernel (128 columns x 80 rows per workgroup)"),
]


def rows(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def mean_kb(rs, frag, wgs):
    sel = [r for r in rs if frag in r["name"] and (wgs is None or int(r["workgroups"]) == wgs)]
    n = sum(int(r["launches"]) for r in sel)
    return (sum(float(r["total"]) for r in sel) / n, n) if n else (None, 0)


def kernel_only_us(d, wl, frag, wgs):
    """Launch-weighted mean duration of the same kernel in the rocprofv3 kernel trace of the same tree (`<wl>_bf16_kernel_stats.csv`)."""
    try:
        sel = [r for r in rows(d / f"{wl}_bf16_kernel_stats.csv") if frag in r["name"] and (wgs is None or int(r["workgroups"]) == wgs)]
    except OSError:
        return None, 0
    n = sum(int(r["calls"]) for r in sel)
    return (sum(float(r["total_ns"]) for r in sel) / n / 1e3, n) if n else (None, 0)


def main():
    d = Path(sys.argv[1] if len(sys.argv) > 1 else "profiles/r6")
    commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.run(
        ["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    out = {"commit": commit,
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes with --kernel-trace; traffic = "
                     "(2 * FETCH_SIZE + WRITE_SIZE) KB * 1024 per launch (gfx950 FETCH_SIZE correction of the micro-arch guide)"}
    for wl, key, frag, wgs, note in KEYS:
        f, nf = mean_kb(rows(d / f"pmc_{wl}_FETCH_SIZE.csv"), frag, wgs)
        w, nw = mean_kb(rows(d / f"pmc_{wl}_WRITE_SIZE.csv"), frag, wgs)
        if f is None or w is None:
            continue
        out.setdefault(wl, {})[key] = {
            "traffic_bytes_per_launch": (2 * f + w) * 1024, "fetch_size_kb_raw": f, "write_size_kb": w,
            "launches_fetch_pass": nf, "launches_write_pass": nw, "note": note}
        us, n_us = kernel_only_us(d, wl, frag, wgs)
        if us is not None:
            out[wl][key].update(kernel_only_us=us, kernel_only_launches=n_us)
    json.dump(out, open(d / "pmc_hbm_traffic.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
