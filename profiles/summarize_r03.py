#!/usr/bin/env python3
"""Distils the rocprofv3 CSVs of profiles/collect_r03.sh (under gpurun_out/) into the committed summaries:
profiles/<tag>_kernel_stats.csv, <tag>_summary.md, <tag>_bench.json and the `x_only` entry of pmc_summary.json (read by bench.py).

HBM traffic (MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE come from separate passes, in KB.  gfx950 tallies a
16-byte-per-lane load at half its bytes; every global load of this kernel is a global_load_dwordx4, so the corrected figure doubles the
raw one.  Both are written down; `roofline.traffic` in bench.py is the corrected one, `roofline.frac` uses algorithmic bytes only.
usage: python profiles/summarize_r02.py <tag> <kernel-substring>"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(path):
    with open(path, newline="") as f:
        return list(csv.DictReader(f))


def counters(path, ksub):
    acc = {}
    for r in rows(path):
        if ksub in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    tag, ksub = sys.argv[1], sys.argv[2]
    g = os.path.join(ROOT, "gpurun_out")
    stats_src = os.path.join(g, f"prof_stats_{tag}", "stats_kernel_stats.csv")
    shutil.copy(stats_src, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    st = rows(stats_src)
    k = next(r for r in st if ksub in r["Name"])
    kt = next(r for r in rows(os.path.join(g, f"prof_stats_{tag}", "stats_kernel_trace.csv")) if ksub in r["Kernel_Name"])
    fetch, _ = counters(os.path.join(g, f"prof_fetch_{tag}", "fetch_counter_collection.csv"), ksub)
    write, _ = counters(os.path.join(g, f"prof_write_{tag}", "write_counter_collection.csv"), ksub)
    sq1, n1 = counters(os.path.join(g, f"prof_sq1_{tag}", "sq_counter_collection.csv"), ksub)
    sq2, _ = counters(os.path.join(g, f"prof_sq2_{tag}", "sq_counter_collection.csv"), ksub)
    f_raw, w_raw = fetch["FETCH_SIZE"], write["WRITE_SIZE"]
    hbm = (2.0 * f_raw + w_raw) * 1024.0
    waves = sq1["SQ_WAVES"]
    summary = {
        "tag": tag, "kernel": k["Name"], "calls": int(k["Calls"]), "avg_ns": float(k["AverageNs"]), "min_ns": float(k["MinNs"]), "max_ns": float(k["MaxNs"]),
        "pct_of_gpu_time": float(k["Percentage"]),
        "rocprof_VGPR_Count": int(kt["VGPR_Count"]), "rocprof_Accum_VGPR_Count": int(kt["Accum_VGPR_Count"]), "rocprof_SGPR_Count": int(kt["SGPR_Count"]),
        "rocprof_LDS_Block_Size": int(kt["LDS_Block_Size"]), "workgroup": int(kt["Workgroup_Size_X"]), "grid": int(kt["Grid_Size_X"]),
        "FETCH_SIZE_KB_raw": f_raw, "WRITE_SIZE_KB_raw": w_raw, "FETCH_SIZE_x2_applied": True,
        "hbm_bytes_per_launch": hbm, "hbm_bytes_per_launch_uncorrected": (f_raw + w_raw) * 1024.0,
        "per_wave": {"VALU": sq1["SQ_INSTS_VALU"] / waves, "SALU": sq1["SQ_INSTS_SALU"] / waves, "LDS": sq1["SQ_INSTS_LDS"] / waves,
                     "wave_cycles": 4 * sq1["SQ_WAVE_CYCLES"] / waves, "valu_active_cycles": 4 * sq1["SQ_ACTIVE_INST_VALU"] / waves,
                     "wait_inst_any_cycles": 4 * sq1["SQ_WAIT_INST_ANY"] / waves, "wait_any_cycles": 4 * sq2["SQ_WAIT_ANY"] / waves,
                     "active_inst_any_cycles": 4 * sq2["SQ_ACTIVE_INST_ANY"] / waves, "mfma_f64_insts": sq2.get("SQ_INSTS_VALU_MFMA_F64", 0.0) / waves},
        "waves_per_dispatch": waves,
    }
    # calibration of FETCH_SIZE on known byte counts in this kernel's access patterns (scripts/ubench/loadpat.hip under the same counter)
    cal = {}
    known = {4: ("contiguous 16 B per lane, the whole 4-problem block of a wave", 4096 * 60 * 41 * 8),
             2: ("lane = column, 96-byte column segments at a 480-byte stride (level 0 of lqr_qtol)", 4096 * 4 * 12 * 41 * 8),
             0: ("48-byte pieces of columns in consecutive lane triples (levels 1.. of lqr_qtol)", 4096 * 4 * 12 * 41 * 8)}
    for m, (what, useful) in known.items():
        path = os.path.join(g, f"prof_cal_{tag}_m{m}", "cal_counter_collection.csv")
        if os.path.exists(path):
            c, _ = counters(path, "k<")
            raw = c["FETCH_SIZE"] * 1024.0
            cal[f"mode{m}"] = {"pattern": what, "useful_bytes": useful, "FETCH_SIZE_bytes_raw": raw, "raw_over_useful": raw / useful, "x2_over_useful": 2 * raw / useful}
    summary["fetch_size_calibration"] = cal
    pj = os.path.join(ROOT, "profiles", "pmc_summary.json")
    allsum = json.load(open(pj)) if os.path.exists(pj) else {}
    allsum["x_only"] = summary
    json.dump(allsum, open(pj, "w"), indent=1)
    bj = os.path.join(g, f"bench_{tag}.json")
    if os.path.exists(bj):
        shutil.copy(bj, os.path.join(ROOT, "profiles", f"{tag}_bench.json"))
    pw = summary["per_wave"]
    with open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w") as f:
        f.write(f"# rocprofv3 summary `{tag}` — `python3 bench.py` (x-only, 4 rotating resident batches; lqr_qtol, the tolerance-contract bench kernel)\n\n")
        f.write("Passes (profiles/collect_r03.sh): `--kernel-trace --stats`; `--pmc FETCH_SIZE`; `--pmc WRITE_SIZE`; two `--pmc SQ_*` passes.\n\n")
        f.write("| kernel | calls | avg (us) | min (us) | max (us) | % GPU time | WG | grid |\n|---|---|---|---|---|---|---|---|\n")
        f.write(f"| `{k['Name']}` | {k['Calls']} | {float(k['AverageNs'])/1e3:.2f} | {float(k['MinNs'])/1e3:.2f} | {float(k['MaxNs'])/1e3:.2f} | {k['Percentage']} | "
                f"{kt['Workgroup_Size_X']} | {kt['Grid_Size_X']} |\n\n")
        f.write("Register / LDS columns of the kernel trace, and how to read them on gfx950 (profiles/r02_resources.txt has the compiler's own report):\n"
                f"`VGPR_Count` = {kt['VGPR_Count']}, `Accum_VGPR_Count` = {kt['Accum_VGPR_Count']}, `SGPR_Count` = {kt['SGPR_Count']}, `LDS_Block_Size` = {kt['LDS_Block_Size']}.\n"
                "rocprofv3 decodes the descriptor's VGPR granule count with a granule of 4 registers; the gfx950 granule is 8, so the ALLOCATED architectural VGPRs are "
                "2 x `VGPR_Count` (the same factor explains round 1's `52` for a kernel the compiler reports at 102 and that is allocated 104). `LDS_Block_Size` is the STATIC "
                "LDS of the code object; these kernels take their LDS dynamically at launch (lqr_qtol: 161,792 B per 256-thread workgroup for the IK shape = 4 wavefronts x 4 problems x 10,112 B, one wavefront per SIMD), which the column does not show. "
                "lqr_qtol: 256 architectural VGPRs + 256 accumulation registers (a184..a255 hold the level-ahead pieces, managed by inline assembly; the compiler's own spill traffic uses a0..a28).\n\n")
        f.write(f"HBM traffic per launch: FETCH_SIZE = {f_raw:.1f} KB raw, WRITE_SIZE = {w_raw:.1f} KB raw. gfx950 tallies 16-B-per-lane loads at half "
                f"(MI355X_MICROARCH.md, HBM): corrected fetch = {2*f_raw:.1f} KB -> **{hbm/1e6:.1f} MB per launch** (uncorrected: {(f_raw+w_raw)*1024/1e6:.1f} MB). "
                "Algorithmic bytes per launch: 81.92 MB by SURVEY 8(d)'s figure (20,000 B x 4096); the x-only kernel itself needs 64.5 MB of input "
                "(it never reads the rows of level 4, whose columns are exhausted) + 1.3 MB of x + 0.8 MB of pivots / ranks = 66.6 MB touched.\n\n")
        if cal:
            f.write("FETCH_SIZE calibrated on known byte counts (same counter, `scripts/ubench/loadpat`, buffers beyond the Infinity Cache):\n\n| pattern | useful bytes | FETCH_SIZE raw | raw / useful | 2 x raw / useful |\n|---|---|---|---|---|\n")
            for v in cal.values():
                f.write(f"| {v['pattern']} | {v['useful_bytes']/1e6:.1f} MB | {v['FETCH_SIZE_bytes_raw']/1e6:.1f} MB | {v['raw_over_useful']:.3f} | {v['x2_over_useful']:.3f} |\n")
            f.write("\nThe contiguous pattern reads every byte once and is tallied at exactly half (the guide's figure), so the doubling applies to this kernel's 16-byte-per-lane loads; "
                    "its two strided patterns fetch 1.43-1.46 x their useful bytes (64-byte sectors around 96- and 48-byte segments).\n\n")
        f.write("SQ counters per wavefront (1024 wavefronts per launch, one per SIMD; cycle counters x4: they count quad-cycles):\n\n")
        f.write("| VALU insts | SALU insts | LDS insts | wave cycles | VALU-active cycles | issue-stall cycles (WAIT_INST_ANY) | s_waitcnt cycles (WAIT_ANY) | any-inst-active cycles | MFMA f64 insts |\n|---|---|---|---|---|---|---|---|---|\n")
        f.write(f"| {pw['VALU']:.0f} | {pw['SALU']:.0f} | {pw['LDS']:.0f} | {pw['wave_cycles']:.0f} | {pw['valu_active_cycles']:.0f} | {pw['wait_inst_any_cycles']:.0f} | "
                f"{pw['wait_any_cycles']:.0f} | {pw['active_inst_any_cycles']:.0f} | {pw['mfma_f64_insts']:.0f} |\n\n")
        f.write("All kernels (--stats):\n\n```\n")
        for r in st[:8]:
            f.write(f"{r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_ns={r['AverageNs']:>14s} pct={r['Percentage']}\n")
        f.write("```\n")
        # the large path (configs[1]) and the lock-step LexLSI batch (configs[4]): own traces of the same collection
        lg = os.path.join(g, f"prof_large_{tag}", "large_kernel_stats.csv")
        if os.path.exists(lg):
            shutil.copy(lg, os.path.join(ROOT, "profiles", f"{tag}_large_kernel_stats.csv"))
            f.write("\n## configs[1] (n = 512, 4 levels x 256 rows): `python3 scripts/time_large.py` (6 factorize+solve calls)\n\n```\n")
            for r in rows(lg)[:10]:
                f.write(f"{r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.2f} pct={r['Percentage']}\n")
            f.write("```\n")
            mf = os.path.join(g, f"prof_large_mfma_{tag}", "mfma_counter_collection.csv")
            if os.path.exists(mf):
                c, nn = counters(mf, "large_gemm_mfma")
                if c:
                    f.write(f"\nMatrix-core counters of `large_gemm_mfma` (per launch, mean over {max(nn.values())} launches; `--pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES`): "
                            + ", ".join(f"{k2} = {v2:.0f}" for k2, v2 in sorted(c.items())) + ". "
                            "One `v_mfma_f64_16x16x4_f64` = 2048 flops per wavefront instruction: the instruction count x 2048 is the trailing update's flop count; the busy-cycle share says how little of "
                            "the launch the matrix cores are occupied — the launch is bound by its LDS staging and barriers, not by the matrix cores.\n")
        ls = os.path.join(g, f"prof_lsi_{tag}", "lsi_kernel_stats.csv")
        if os.path.exists(ls):
            shutil.copy(ls, os.path.join(ROOT, "profiles", f"{tag}_lsi_kernel_stats.csv"))
            f.write("\n## configs[4] (1024 lock-step LexLSI instances): `python3 bench.py --workload lsi --steps 3 --warmup 1`\n\n```\n")
            for r in rows(ls)[:8]:
                f.write(f"{r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.2f} pct={r['Percentage']}\n")
            f.write("```\n")
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
