#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd`
writes DIR/NAME_results.db on ROCm 7.2).  Prints / writes the same table `--stats` would: calls, total, average,
min, max duration per kernel, plus PMC counter sums per kernel when the run collected counters (`--pmc ...`).

    python tools/rocpd_summary.py gpurun_out/prof/r1_results.db [-o profiles/r1_kernel_stats.md] [--json out.json]
"""
import argparse
import json
import sqlite3
import sys


def kernel_stats(db, by_grid=False):
    c = sqlite3.connect(db)
    if by_grid:   # one row per launch shape AND queue: a run that launches a kernel over different batch sizes keeps them apart, and so does one
        # that launches the same shape from different streams (the replay lanes' overlapped launches against bench.py's isolated roofline passes)
        rows = c.execute("select name || ' [grid ' || (grid_x / workgroup_x) || 'x' || grid_y || 'x' || grid_z || ' wg] [stream ' || stream_id || ']', count(*), "
                         "sum(end-start), avg(end-start), min(end-start), max(end-start) "
                         "from kernels group by name, grid_x, grid_y, grid_z, stream_id order by name, sum(end-start) desc").fetchall()
    else:
        rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                         "from kernels group by name order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = [{"kernel": r[0], "calls": r[1], "total_ns": int(r[2]), "avg_ns": float(r[3]), "min_ns": int(r[4]),
            "max_ns": int(r[5]), "pct": 100.0 * r[2] / total} for r in rows]
    pmc = {}
    try:
        q = ("select k.name, p.counter_name, sum(p.value), count(*) from pmc_events p "
             "join kernels k on p.dispatch_id = k.dispatch_id group by k.name, p.counter_name")
        for name, counter, val, n in c.execute(q):
            pmc.setdefault(name, {})[counter] = {"sum": float(val), "dispatches": int(n)}
    except sqlite3.Error:
        try:
            cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
            kcol = "kernel_name" if "kernel_name" in cols else "name"
            for name, counter, val, n in c.execute(
                    f"select {kcol}, counter_name, sum(value), count(*) from counters_collection group by {kcol}, counter_name"):
                pmc.setdefault(name, {})[counter] = {"sum": float(val), "dispatches": int(n)}
        except sqlite3.Error:
            pass
    return out, pmc


def short(name, n=70):
    tail = name[name.find(" [grid"):] if " [grid" in name else ""
    name = name.split("(")[0]
    if tail and not name.endswith(tail):
        name += tail
    return name if len(name) <= n else name[:n - 3] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("-o", "--out")
    ap.add_argument("--json")
    ap.add_argument("--title", default="")
    ap.add_argument("--by-grid", action="store_true", help="one row per (kernel, grid size)")
    a = ap.parse_args()
    stats, pmc = kernel_stats(a.db, a.by_grid)
    lines = []
    if a.title:
        lines += [f"# {a.title}", ""]
    lines += ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for s in stats:
        lines.append(f"| {short(s['kernel'])} | {s['calls']} | {s['total_ns'] / 1e6:.3f} | {s['avg_ns'] / 1e3:.2f} | "
                     f"{s['min_ns'] / 1e3:.2f} | {s['max_ns'] / 1e3:.2f} | {s['pct']:.1f} |")
    if pmc:
        lines += ["", "| kernel | counter | sum over dispatches | dispatches | per dispatch |", "|---|---|---:|---:|---:|"]
        for k, cs in pmc.items():
            for cn, v in sorted(cs.items()):
                lines.append(f"| {short(k)} | {cn} | {v['sum']:.0f} | {v['dispatches']} | {v['sum'] / max(v['dispatches'], 1):.1f} |")
    text = "\n".join(lines) + "\n"
    if a.out:
        open(a.out, "w").write(text)
    else:
        sys.stdout.write(text)
    if a.json:
        json.dump({"kernels": stats, "pmc": pmc}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
