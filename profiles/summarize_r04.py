#!/usr/bin/env python3
"""Distils the rocprofv3 CSVs of profiles/collect_r04.sh (under gpurun_out/) into the committed summaries of round 4:
profiles/r04_kernel_stats.csv (bench.py's default command: lqr_qtol), r04m_kernel_stats.csv (the same command on the matrix-core kernel lqr_mfma,
LEXLS_KERNEL_POLICY=7), r04_large / r04_lsi kernel stats, r04_summary.md, r04_bench*.json and the `x_only` / `x_only_mfma` entries of
pmc_summary.json (bench.py reads `x_only` for roofline.traffic).

HBM traffic (MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE come from separate passes, in KB.  gfx950 tallies a 16-byte-per-lane
load at half its bytes (calibrated in the same collection on scripts/ubench/loadpat, mode 4); lqr_qtol's loads are global_load_dwordx4, lqr_mfma's
are global_load_lds_dwordx4 (LDS-DMA, 16 bytes per lane): the corrected figure doubles the raw one for both."""
import csv
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")


def rows(path):
    with open(path, newline="") as f:
        return list(csv.DictReader(f))


def counters(path, ksub):
    acc = {}
    for r in rows(path):
        if ksub in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def one(tag, ksub):
    st = rows(os.path.join(G, f"prof_stats_{tag}", "stats_kernel_stats.csv"))
    shutil.copy(os.path.join(G, f"prof_stats_{tag}", "stats_kernel_stats.csv"), os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    k = next(r for r in st if ksub in r["Name"])
    kt = next(r for r in rows(os.path.join(G, f"prof_stats_{tag}", "stats_kernel_trace.csv")) if ksub in r["Kernel_Name"])
    f_raw = counters(os.path.join(G, f"prof_fetch_{tag}", "fetch_counter_collection.csv"), ksub)["FETCH_SIZE"]
    w_raw = counters(os.path.join(G, f"prof_write_{tag}", "write_counter_collection.csv"), ksub)["WRITE_SIZE"]
    sq1 = counters(os.path.join(G, f"prof_sq1_{tag}", "sq_counter_collection.csv"), ksub)
    sq2 = counters(os.path.join(G, f"prof_sq2_{tag}", "sq_counter_collection.csv"), ksub)
    waves = sq1["SQ_WAVES"]
    simds = 1024.0
    return {
        "tag": tag, "kernel": k["Name"], "calls": int(k["Calls"]), "avg_ns": float(k["AverageNs"]), "min_ns": float(k["MinNs"]), "max_ns": float(k["MaxNs"]),
        "pct_of_gpu_time": float(k["Percentage"]), "workgroup": int(kt["Workgroup_Size_X"]), "grid": int(kt["Grid_Size_X"]),
        "rocprof_VGPR_Count": int(kt["VGPR_Count"]), "rocprof_Accum_VGPR_Count": int(kt["Accum_VGPR_Count"]), "rocprof_SGPR_Count": int(kt["SGPR_Count"]),
        "FETCH_SIZE_KB_raw": f_raw, "WRITE_SIZE_KB_raw": w_raw, "FETCH_SIZE_x2_applied": True,
        "hbm_bytes_per_launch": (2.0 * f_raw + w_raw) * 1024.0, "hbm_bytes_per_launch_uncorrected": (f_raw + w_raw) * 1024.0,
        "waves_per_dispatch": waves,
        "per_wave": {"VALU": sq1["SQ_INSTS_VALU"] / waves, "SALU": sq1["SQ_INSTS_SALU"] / waves, "LDS": sq1["SQ_INSTS_LDS"] / waves,
                     "wave_cycles": 4 * sq1["SQ_WAVE_CYCLES"] / waves, "valu_active_cycles": 4 * sq1["SQ_ACTIVE_INST_VALU"] / waves,
                     "wait_inst_any_cycles": 4 * sq1["SQ_WAIT_INST_ANY"] / waves, "wait_any_cycles": 4 * sq2["SQ_WAIT_ANY"] / waves,
                     "active_inst_any_cycles": 4 * sq2["SQ_ACTIVE_INST_ANY"] / waves, "salu_active_cycles": 4 * sq2["SQ_ACTIVE_INST_SCA"] / waves,
                     "lds_active_cycles": 4 * sq2["SQ_ACTIVE_INST_LDS"] / waves, "mfma_f64_insts": sq2.get("SQ_INSTS_VALU_MFMA_F64", 0.0) / waves},
        "per_simd": {"valu_active_cycles": 4 * sq1["SQ_ACTIVE_INST_VALU"] / simds, "mfma_busy_cycles": sq2.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simds,
                     "VALU": sq1["SQ_INSTS_VALU"] / simds},
        "lds_bank_conflict_cycles_per_cu": sq2.get("SQ_LDS_BANK_CONFLICT", 0.0) / 256.0,
    }, st, kt


KN = re.compile(r'lqr_[a-z]+_kernel<[^>]*>')


def kname(s_):
    return KN.search(s_['kernel']).group(0)


def main():
    q, stq, ktq = one("r04", "lqr_qtol_kernel")
    m, stm, ktm = one("r04m", "lqr_mfma_kernel")
    cal = {}
    known = {4: ("contiguous 16 B per lane, the whole 4-problem block of a wave", 4096 * 60 * 41 * 8),
             2: ("lane = column, 96-byte column segments at a 480-byte stride (level 0 of lqr_qtol)", 4096 * 4 * 12 * 41 * 8),
             0: ("48-byte pieces of columns in consecutive lane triples (levels 1.. of lqr_qtol)", 4096 * 4 * 12 * 41 * 8)}
    for mm, (what, useful) in known.items():
        path = os.path.join(G, f"prof_cal_r04_m{mm}", "cal_counter_collection.csv")
        if os.path.exists(path):
            raw = counters(path, "k<")["FETCH_SIZE"] * 1024.0
            cal[f"mode{mm}"] = {"pattern": what, "useful_bytes": useful, "FETCH_SIZE_bytes_raw": raw, "raw_over_useful": raw / useful, "x2_over_useful": 2 * raw / useful}
    q["fetch_size_calibration"] = cal
    pj = os.path.join(ROOT, "profiles", "pmc_summary.json")
    allsum = json.load(open(pj)) if os.path.exists(pj) else {}
    allsum["x_only"], allsum["x_only_mfma"] = q, m
    json.dump(allsum, open(pj, "w"), indent=1)
    for src, dst in (("bench_r04.json", "r04_bench.json"), ("bench_r04_driver_flags.json", "r04_bench_driver_flags.json"), ("bench_r04m.json", "r04m_bench.json")):
        if os.path.exists(os.path.join(G, src)):
            shutil.copy(os.path.join(G, src), os.path.join(ROOT, "profiles", dst))
    with open(os.path.join(ROOT, "profiles", "r04_summary.md"), "w") as f:
        f.write("# rocprofv3 summary `r04` — `python3 bench.py` (x-only, 4 rotating resident batches of 4096 IK problems)\n\n")
        f.write("Passes (profiles/collect_r04.sh), each its own rocprofv3 run: `--kernel-trace --stats`; `--pmc FETCH_SIZE`; `--pmc WRITE_SIZE`; two `--pmc SQ_*` passes — "
                "once for the default command (the bench kernel `lqr_qtol`, four problems per wavefront, one wavefront per SIMD) and once with `LEXLS_KERNEL_POLICY=7` "
                "(this round's matrix-core kernel `lqr_mfma`, two problems per wavefront, two wavefronts per SIMD, Gauss step on `v_mfma_f64_16x16x4_f64`).\n\n")
        f.write("| kernel | calls | avg (us) | min (us) | max (us) | WG | grid | waves / launch |\n|---|---|---|---|---|---|---|---|\n")
        for s_, kt in ((q, ktq), (m, ktm)):
            f.write(f"| `{s_['kernel'][:110]}` | {s_['calls']} | {s_['avg_ns']/1e3:.2f} | {s_['min_ns']/1e3:.2f} | {s_['max_ns']/1e3:.2f} | {s_['workgroup']} | {s_['grid']} | {s_['waves_per_dispatch']:.0f} |\n")
        f.write("\nRoofline of the dominant kernel (bench.py's definition: SURVEY 8(d)'s algorithmic bytes, 20,000 B x 4096 = 81.92 MB per launch, over the average launch duration, over 8 TB/s):\n\n")
        f.write("| kernel | avg launch | GB/s on algorithmic bytes | frac | frac on the 66.6 MB the x-only solve touches | fact/s |\n|---|---|---|---|---|---|\n")
        for s_ in (q, m):
            t = s_["avg_ns"] * 1e-9
            f.write(f"| `{kname(s_)}` | {s_['avg_ns']/1e3:.2f} us | {81.92e6/t/1e9:.0f} | {81.92e6/t/8e12:.4f} | {66.6e6/t/8e12:.4f} | {4096/t:.3e} |\n")
        f.write("\nHBM traffic per launch (FETCH_SIZE x 2 + WRITE_SIZE; the doubling calibrated below):\n\n| kernel | FETCH_SIZE raw (KB) | WRITE_SIZE raw (KB) | corrected MB per launch | / 81.92 MB algorithmic | / 66.6 MB touched |\n|---|---|---|---|---|---|\n")
        for s_ in (q, m):
            h = s_["hbm_bytes_per_launch"]
            f.write(f"| `{kname(s_)}` | {s_['FETCH_SIZE_KB_raw']:.1f} | {s_['WRITE_SIZE_KB_raw']:.1f} | {h/1e6:.1f} | {h/81.92e6:.3f} | {h/66.6e6:.3f} |\n")
        if cal:
            f.write("\nFETCH_SIZE calibrated on known byte counts (same counter, `scripts/ubench/loadpat`, buffers beyond the Infinity Cache):\n\n| pattern | useful bytes | FETCH_SIZE raw | raw / useful | 2 x raw / useful |\n|---|---|---|---|---|\n")
            for v in cal.values():
                f.write(f"| {v['pattern']} | {v['useful_bytes']/1e6:.1f} MB | {v['FETCH_SIZE_bytes_raw']/1e6:.1f} MB | {v['raw_over_useful']:.3f} | {v['x2_over_useful']:.3f} |\n")
        f.write("\nSQ counters (cycle counters x4: they count quad-cycles).  Per wavefront:\n\n")
        f.write("| kernel | VALU insts | SALU insts | LDS insts | MFMA f64 insts | wave cycles | VALU-active | SALU-active | LDS-active | issue stalls (WAIT_INST_ANY) | s_waitcnt (WAIT_ANY) |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
        for s_ in (q, m):
            pw = s_["per_wave"]
            f.write(f"| `{kname(s_)}` | {pw['VALU']:.0f} | {pw['SALU']:.0f} | {pw['LDS']:.0f} | {pw['mfma_f64_insts']:.0f} | {pw['wave_cycles']:.0f} | {pw['valu_active_cycles']:.0f} | "
                    f"{pw['salu_active_cycles']:.0f} | {pw['lds_active_cycles']:.0f} | {pw['wait_inst_any_cycles']:.0f} | {pw['wait_any_cycles']:.0f} |\n")
        f.write("\nPer SIMD (1024 SIMDs; lqr_qtol: one wavefront each, lqr_mfma: two):\n\n| kernel | VALU insts | VALU-active cycles | matrix-core busy cycles (SQ_VALU_MFMA_BUSY_CYCLES) | LDS bank-conflict cycles per CU |\n|---|---|---|---|---|\n")
        for s_ in (q, m):
            ps = s_["per_simd"]
            f.write(f"| `{kname(s_)}` | {ps['VALU']:.0f} | {ps['valu_active_cycles']:.0f} | {ps['mfma_busy_cycles']:.0f} | {s_['lds_bank_conflict_cycles_per_cu']:.0f} |\n")
        f.write("\nAll kernels of the default command (--stats):\n\n```\n")
        for r in stq[:8]:
            f.write(f"{r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_ns={r['AverageNs']:>14s} pct={r['Percentage']}\n")
        f.write("```\n")
        for key, title, fn in (("large", "configs[1] (n = 512, 4 levels x 256 rows): `python3 scripts/time_large.py` (6 factorize+solve calls)", "large_kernel_stats.csv"),
                               ("lsi", "configs[4] (1024 lock-step LexLSI instances): `python3 bench.py --workload lsi --steps 3 --warmup 1`", "lsi_kernel_stats.csv")):
            src = os.path.join(G, f"prof_{key}_r04", fn)
            if os.path.exists(src):
                shutil.copy(src, os.path.join(ROOT, "profiles", f"r04_{key}_kernel_stats.csv"))
                f.write(f"\n## {title}\n\n```\n")
                for r in rows(src)[:10]:
                    f.write(f"{r['Name'][:90]:90s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.2f} pct={r['Percentage']}\n")
                f.write("```\n")
    print(json.dumps({"qtol": q, "mfma": m}, indent=1)[:3000])


if __name__ == "__main__":
    main()
