#!/usr/bin/env python3
"""Condenses a tools/profile_bench.sh output directory (rocprofv3 CSVs) into the small summaries
committed under profiles/: kernel stats, per-kernel PMC averages and the HBM traffic per
hot-path call (FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md section HBM prescribes).

usage: tools/pmc_summary.py gpurun_out/prof_<tag> profiles/<name> [workload]
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    workload = sys.argv[3] if len(sys.argv) > 3 else "pendulum"
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    out = {"source": src, "workload": workload, "kernels": {}, "pmc": {}}
    # (a bench run is several processes -- the multi-device check runs in a child -- and a directory may hold several
    # runs: per pass, the file with the most library dispatches is the bench's own main process of the latest run)
    def main_file(pattern, count):
        best, bestn = None, -1
        for f in sorted(glob.glob(pattern), key=os.path.getmtime):
            n = count(f)
            if n >= bestn:
                best, bestn = f, n
        return best

    def stat_calls(f):
        return sum(int(r["Calls"]) for r in csv.DictReader(open(f)) if "lmpc::" in r["Name"])

    def pmc_calls(f):
        return sum(1 for r in csv.DictReader(open(f)) if "lmpc::" in r["Kernel_Name"])

    f = main_file(os.path.join(src, "trace", "*", "*_kernel_stats.csv"), stat_calls)
    if f:
        for r in csv.DictReader(open(f)):
            out["kernels"][r["Name"].split("(")[0]] = {
                "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]), "pct": float(r["Percentage"])}
    agg = collections.defaultdict(list)
    for d in glob.glob(os.path.join(src, "pmc_*")):
        f = main_file(os.path.join(d, "*", "*_counter_collection.csv"), pmc_calls)
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            if "lmpc::" in name:
                agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        out["pmc"].setdefault(k, {})[c] = sum(v) / len(v)
    fetch_kb = sum(d.get("FETCH_SIZE", 0.0) for d in out["pmc"].values())
    write_kb = sum(d.get("WRITE_SIZE", 0.0) for d in out["pmc"].values())
    out["hbm_bytes_per_launch"] = (2.0 * fetch_kb + write_kb) * 1024.0
    out["note"] = ("per hot-path call = screening kernel + iterating kernel; FETCH_SIZE/WRITE_SIZE are KB per "
                   "dispatch from separate --pmc passes, FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B)")
    json.dump(out, open(dst + ".json", "w"), indent=1)
    with open(dst + ".md", "w") as fh:
        fh.write(f"# rocprofv3 summary ({workload}) from {src}\n\n## kernel-trace --stats\n\n")
        fh.write("| kernel | calls | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|\n")
        for k, d in out["kernels"].items():
            fh.write(f"| `{k}` | {d['calls']} | {d['avg_ns']:.0f} | {d['min_ns']:.0f} | {d['max_ns']:.0f} | {d['pct']:.1f} |\n")
        fh.write("\n## PMC (average per dispatch)\n\n")
        for k, d in out["pmc"].items():
            fh.write(f"### `{k}`\n\n| counter | value |\n|---|---|\n")
            for c, v in d.items():
                fh.write(f"| {c} | {v:.1f} |\n")
            fh.write("\n")
        fh.write(f"HBM bytes per hot-path call (2*FETCH_SIZE + WRITE_SIZE, KB->B): {out['hbm_bytes_per_launch']:.0f}\n")
    print(json.dumps({k: out[k] for k in ("hbm_bytes_per_launch",)}))


if __name__ == "__main__":
    main()
