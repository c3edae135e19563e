#!/usr/bin/env python3
"""Distils the rocprofv3 CSVs that profiles/collect.sh left under gpurun_out/ into the small, committed
summaries under profiles/:  <tag>_kernel_stats.csv (copy of --stats), <tag>_summary.md and pmc_summary.json
(read by bench.py for roofline.traffic).

HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md section HBM:
  bytes = (FETCH_SIZE_corrected + WRITE_SIZE) * 1024,  FETCH_SIZE and WRITE_SIZE collected in SEPARATE passes;
  on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read, so
  fetch is doubled when the kernel's input loads are 16-B-per-lane streams (`--fetch-x2`, stated in the summary).
usage: python profiles/summarize.py <tag> <kernel-substring> <variant: x_only|factor> [--fetch-x2] [--fetch-add-kb=N]
  --fetch-add-kb: kernels that mix 16-B-per-lane loads (half-counted) with narrower ones (fully counted) cannot use one factor;
                  N = the deficit of the wide loads, i.e. their raw count measured in a pass of the variant that has only them.
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(path):
    with open(path, newline="") as f:
        return list(csv.DictReader(f))


def mean_counter(path, kernel_sub, counter):
    vals = [float(r["Counter_Value"]) for r in rows(path) if kernel_sub in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    tag, ksub, variant = sys.argv[1], sys.argv[2], sys.argv[3]
    x2 = "--fetch-x2" in sys.argv
    add_kb = 0.0
    for a in sys.argv:
        if a.startswith("--fetch-add-kb="):
            add_kb = float(a.split("=", 1)[1])
    g = os.path.join(ROOT, "gpurun_out")
    stats_src = os.path.join(g, f"prof_stats_{tag}", "stats_kernel_stats.csv")
    shutil.copy(stats_src, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    st = [r for r in rows(stats_src)]
    k = next(r for r in st if ksub in r["Name"])
    trace = rows(os.path.join(g, f"prof_stats_{tag}", "stats_kernel_trace.csv"))
    kt = next(r for r in trace if ksub in r["Kernel_Name"])
    fetch, nf = mean_counter(os.path.join(g, f"prof_fetch_{tag}", "fetch_counter_collection.csv"), ksub, "FETCH_SIZE")
    write, nw = mean_counter(os.path.join(g, f"prof_write_{tag}", "write_counter_collection.csv"), ksub, "WRITE_SIZE")
    fetch_corr = None if fetch is None else fetch * (2.0 if x2 else 1.0) + add_kb
    hbm = None if fetch is None or write is None else (fetch_corr + write) * 1024.0
    summary = {
        "tag": tag, "kernel": k["Name"], "calls": int(k["Calls"]), "avg_ns": float(k["AverageNs"]), "min_ns": float(k["MinNs"]),
        "max_ns": float(k["MaxNs"]), "pct_of_gpu_time": float(k["Percentage"]),
        "vgpr": int(kt["VGPR_Count"]), "agpr": int(kt["Accum_VGPR_Count"]), "sgpr": int(kt["SGPR_Count"]), "lds_bytes": int(kt["LDS_Block_Size"]),
        "workgroup": int(kt["Workgroup_Size_X"]), "grid": int(kt["Grid_Size_X"]),
        "FETCH_SIZE_KB_raw": fetch, "FETCH_SIZE_x2_applied": x2, "FETCH_SIZE_added_KB": add_kb, "WRITE_SIZE_KB": write, "pmc_dispatches": [nf, nw],
        "hbm_bytes_per_launch": hbm,
    }
    pj = os.path.join(ROOT, "profiles", "pmc_summary.json")
    allsum = json.load(open(pj)) if os.path.exists(pj) else {}
    allsum[variant] = summary
    json.dump(allsum, open(pj, "w"), indent=1)
    with open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w") as f:
        f.write(f"# rocprofv3 summary `{tag}` ({variant})\n\n")
        f.write("Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 100 --warmup 10` "
                "(+ two `--pmc` passes: FETCH_SIZE, WRITE_SIZE), see profiles/collect.sh.\n\n")
        f.write("| kernel | calls | avg (us) | min (us) | max (us) | % GPU time | VGPR | AGPR | SGPR | LDS (B) | WG | grid |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        f.write(f"| `{k['Name']}` | {k['Calls']} | {float(k['AverageNs'])/1e3:.2f} | {float(k['MinNs'])/1e3:.2f} | {float(k['MaxNs'])/1e3:.2f} | "
                f"{k['Percentage']} | {kt['VGPR_Count']} | {kt['Accum_VGPR_Count']} | {kt['SGPR_Count']} | {kt['LDS_Block_Size']} | {kt['Workgroup_Size_X']} | {kt['Grid_Size_X']} |\n\n")
        f.write(f"PMC (mean per dispatch of that kernel): FETCH_SIZE = {fetch} KB raw"
                f"{' (x2 gfx950 wide-stream correction applied)' if x2 else ' (no x2 correction applied)'}{f' (+{add_kb} KB: half-counted wide loads, see summarize.py)' if add_kb else ''}, WRITE_SIZE = {write} KB  ->  "
                f"HBM bytes per launch = {hbm}\n\nAll kernels (--stats):\n\n```\n")
        for r in st[:8]:
            f.write(f"{r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_ns={r['AverageNs']:>14s} pct={r['Percentage']}\n")
        f.write("```\n")
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
