#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 result database (rocpd SQLite, what `rocprofv3 --kernel-trace` writes on ROCm 7):
    python tools/rocpd_stats.py results.db [--by-grid] [--csv out.csv]
Name, calls, total / average / min / max duration (ns) and -- with --by-grid -- one line per (kernel, grid, block) shape;
also the mean gap between consecutive dispatches.  Used to produce the summaries under profiles/."""
import argparse
import csv
import sqlite3
import sys
from collections import defaultdict


def load(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = (f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x, d.group_segment_size, "
         f"s.arch_vgpr_count, s.accum_vgpr_count from {kd} d join {ks} s on d.kernel_id = s.id order by d.start")
    return list(c.execute(q))


def load_pmc(path):
    """-> {counter name: {(kernel, workgroups, block): [values per dispatch]}} from a `rocprofv3 --pmc ... --kernel-trace` database."""
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    ip = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    q = (f"select i.name, s.kernel_name, d.grid_size_x * d.grid_size_y * d.grid_size_z / d.workgroup_size_x, d.workgroup_size_x, e.value "
         f"from {pe} e join {ip} i on e.pmc_id = i.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id")
    out = defaultdict(lambda: defaultdict(list))
    for cname, kname, wgs, blk, val in c.execute(q):
        out[cname][(kname, wgs, blk)].append(val)
    return out


def main_pmc(a):
    data = load_pmc(a.db)
    rows = []
    for cname, per in data.items():
        for (kname, wgs, blk), vals in per.items():
            rows.append(dict(counter=cname, name=kname, workgroups=wgs, block=blk, launches=len(vals), avg=sum(vals) / len(vals),
                             total=sum(vals)))
    rows.sort(key=lambda r: (r["counter"], -r["total"]))
    if a.csv:
        with open(a.csv, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
    for r in rows[:a.top]:
        nm = r["name"] if len(r["name"]) < 80 else r["name"][:77] + "..."
        print(f"{r['counter']:28s} avg {r['avg']:14.1f}  x{r['launches']:<5} wg={r['workgroups']:<6} {nm}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--pmc", action="store_true", help="per-kernel averages of the hardware counters of a --pmc run")
    ap.add_argument("--by-grid", action="store_true")
    ap.add_argument("--csv")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--sequence", type=int, default=0, help="also print the LAST N dispatches in launch order (name, workgroups, us, gap to the previous one)")
    a = ap.parse_args()
    if a.pmc:
        return main_pmc(a)
    rows = load(a.db)
    if a.sequence:
        seq = rows[-a.sequence:]
        for i, (name, st, en, gx, gy, gz, wx, lds, vg, ag) in enumerate(seq):
            gap = (st - seq[i - 1][2]) / 1e3 if i else 0.0
            short = name.split("(")[0]
            short = short if len(short) < 70 else short[:67] + "..."
            print(f"SEQ {i:5d} {(en - st) / 1e3:8.2f} us  gap {gap:7.2f}  wg={gx * gy * gz // max(wx, 1):<6} {short}")
    agg = defaultdict(list)
    for name, st, en, gx, gy, gz, wx, lds, vg, ag in rows:
        key = (name, gx * gy * gz // max(wx, 1), wx, lds) if a.by_grid else (name,)
        agg[key].append(en - st)
    total = sum(sum(v) for v in agg.values())
    out = []
    for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        out.append(dict(name=key[0], workgroups=key[1] if a.by_grid else "", block=key[2] if a.by_grid else "",
                        lds=key[3] if a.by_grid else "", calls=len(v), total_ns=sum(v), avg_ns=round(sum(v) / len(v), 1),
                        min_ns=min(v), max_ns=max(v), pct=round(100.0 * sum(v) / max(total, 1), 2)))
    gaps = [rows[i + 1][1] - rows[i][2] for i in range(len(rows) - 1)]
    gaps = [g for g in gaps if 0 <= g < 50_000]
    if a.csv:
        with open(a.csv, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(out[0].keys()))
            w.writeheader()
            w.writerows(out)
    for r in out[:a.top]:
        nm = r["name"] if len(r["name"]) < 90 else r["name"][:87] + "..."
        extra = f" wg={r['workgroups']:<6} blk={r['block']:<4} lds={r['lds']:<6}" if a.by_grid else ""
        print(f"{r['avg_ns'] / 1e3:8.2f} us avg {r['min_ns'] / 1e3:7.2f} min  x{r['calls']:<5} {r['pct']:5.1f}%{extra}  {nm}")
    print(f"kernel time {total / 1e6:.3f} ms over {len(rows)} dispatches; mean gap between dispatches {sum(gaps) / max(len(gaps), 1) / 1e3:.2f} us", file=sys.stderr)


if __name__ == "__main__":
    main()
