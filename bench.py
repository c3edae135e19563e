#!/usr/bin/env python3
"""Headline benchmark: batched fp64 l-QR factorizations/s (one factorization = factorize()+solve() of one problem).

Workload (BASELINE.json configs[2] per GPU, configs[3] at 8 GPUs): 4096 IK-sized problems per GPU
(n=40 variables, 5 levels x 12 rows, iid N(0,1), seed 20260100 + global problem id), resident in HBM
before the timed region.  One "step" = one pass of the hot path (lexls_lse_factorize_solve, x-only
traffic variant) over ONE batch of 4096 problems; successive steps walk through --resident-batches (default 4)
DIFFERENT resident batches, 322 MB in total — more than the 256 MiB Infinity Cache, so a step's reads are HBM
reads, not re-reads of a cached batch (MI355X_MICROARCH.md, Infinity Cache).  Multi-GPU: problems are independent, so the batch is sharded
by contiguous index blocks, one process per GPU, no data-path collective (weak scaling); the process
group (RCCL) is used for the barriers around the timed region, the max-over-ranks of the elapsed time
and a final all-gather of per-shard solution checksums.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import datetime
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NVAR, DIMS, PER_GPU_BATCH, SEED0 = 40, [12] * 5, 4096, 20260100
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CUs x 4 SIMDs x 16 fp64 FMA lanes/clk x 2 flop x 2.4 GHz (the fp64 MFMA rate is the same)
MIN_WARMUP = 20  # launches before the timed region, whatever --warmup says: caches in steady state
MIN_WARMUP_SECONDS = 0.5  # ... and at least this long: the timed region of a short run (--steps 20 is 1 ms) must not fall into the clock ramp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="problems per GPU")
    ap.add_argument("--keep-factor", action="store_true", help="also write the factor to HBM (40,360 B/problem variant)")
    ap.add_argument("--resident-batches", type=int, default=4, help="distinct batches resident in HBM that the steps rotate through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlapped", action="store_true", help="N=1: also time two half batches on two streams (extra field, never `value`); off by default so that\n                    the default command launches the bench kernel only as the timed step does (profiles/ hold rocprofv3 summaries of it)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, the real thing); gloo only to rehearse the multi-rank control flow on a box with fewer GPUs than ranks")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--workload", default="lse", choices=["lse", "lsi"],
                    help="lse (default): the headline metric, BASELINE configs[2]/[3]; lsi: configs[4], a lock-step batch of 1024 LexLSI instances "
                         "warm-started to ~30 factorizations each, instance blocks sharded over the ranks (strong scaling)")
    ap.add_argument("--lsi-batch", type=int, default=1024, help="--workload lsi: instances in the whole job")
    ap.add_argument("--no-extras", action="store_true", help="N=1: skip the config1_large / config4_lsi / factor_kept side measurements (profiling runs)")
    ap.add_argument("--dry-run", action="store_true", help="control flow only (launcher, process group, barriers, max-over-ranks, rank-0 line): no GPU, no kernels; "
                                                            "what the non-GPU test of the self-launcher runs with --dist-backend gloo")
    args = ap.parse_args()

    # Started plainly with --gpus N > 1 (no launcher around it): spawn the N ranks as fresh child processes BEFORE anything touches the GPU
    # (never re-exec a process that has initialised HIP), one per GPU, rendezvous on 127.0.0.1; rank 0's JSON line is this process's output
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args.gpus)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if os.environ.get("LEXLS_BENCH_FAIL_RANK") == str(rank) and world > 1:  # test hook of the launcher's failure path (tests/test_bench_launcher.py)
        sys.stderr.write("bench.py: failing on request (LEXLS_BENCH_FAIL_RANK)\n")
        raise SystemExit(7)
    if args.dry_run:
        return dry_run(args, world, rank)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    device_index = local_rank if args.dist_backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    coll_device = "cuda" if args.dist_backend == "nccl" else "cpu"  # where the few collective payloads live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        rdv = {"init_method": os.environ["LEXLS_BENCH_RDV"], "rank": rank, "world_size": world} if os.environ.get("LEXLS_BENCH_RDV") else {}
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index), timeout=datetime.timedelta(minutes=5), **rdv)
        else:
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(minutes=5), **rdv)

    import lexls_amd
    from lexls_amd import problems as P

    if args.workload == "lsi":
        return bench_lsi(args, world, rank, device_index, coll_device, torch, dist)

    batch = args.batch
    # this rank's shard of the global batch (contiguous block of problem ids) -> HBM
    first_id = rank * batch
    lod_host = P.lse_batch(SEED0 + first_id, batch, NVAR, DIMS)  # problem id -> seed 20260100 + id
    lod_dev = torch.from_numpy(lod_host).cuda()
    # the other resident batches of this rank: same shape, fresh problem ids beyond the global batch of every rank
    nres = max(1, args.resident_batches)
    resident = [lod_dev] + [torch.from_numpy(P.lse_batch_fast(SEED0 + (j * world + rank) * batch + 7919 * j, batch, NVAR, DIMS)).cuda() for j in range(1, nres)]
    ptrs = [t.data_ptr() for t in resident]

    stream = torch.cuda.Stream()
    solver = lexls_amd.BatchedLexLSE(batch, NVAR, DIMS, device=device_index)
    solver.set_stream(stream.cuda_stream)
    solver.setProblemDevice(lod_dev.data_ptr())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step(i):
        solver.setProblemDevice(ptrs[i % nres])  # zero-copy: the next resident batch becomes the problem data
        solver.factorize_solve(keep_factor=args.keep_factor)

    with torch.cuda.stream(stream):
        warm_run, t_w = 0, time.perf_counter()
        while warm_run < max(args.warmup, MIN_WARMUP) or time.perf_counter() - t_w < MIN_WARMUP_SECONDS:
            for i in range(50):
                step(warm_run + i)
            warm_run += 50
            torch.cuda.synchronize()
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for i in range(args.steps):
            step(i)
        ev1.record(stream)
        barrier()
        elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream: average launch duration
    if (args.steps - 1) % nres != 0:  # leave the solutions of batch 0 behind for the guard and the scatter/gather check below
        with torch.cuda.stream(stream):
            step(0)
        torch.cuda.synchronize()

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness guard outside the timed region: solution checksum per shard, gathered on every rank
    x = solver.get_x()
    ranks_ok = bool((solver.getRanks()[0] == np.array([12, 12, 12, 4, 0])).all())  # every problem of the batch
    checksum = torch.tensor([float(np.abs(x).sum()), float(ranks_ok)], dtype=torch.float64, device=coll_device)
    if world > 1:
        gathered = [torch.zeros_like(checksum) for _ in range(world)]
        dist.all_gather(gathered, checksum)
        all_ok = all(bool(g[1].item()) and np.isfinite(g[0].item()) for g in gathered)
        shard_checksums = [float(g[0].item()) for g in gathered]
    else:
        all_ok = ranks_ok and np.isfinite(checksum[0].item())
        shard_checksums = [float(checksum[0].item())]
    if not all_ok:
        raise SystemExit("bench: wrong ranks / non-finite solution — refusing to report a number")

    # ---- outside the timed region, N = 1 only, on request (--overlapped): the same batch as two half batches on two streams.  Their launches overlap each
    #      other's launch / first-load / drain floor; a serving loop that keeps two batches in flight gets this rate.  Never `value`.
    overlapped = None
    if args.overlapped and world == 1 and batch % 2 == 0:
        try:
            halves = []
            for i in range(2):
                st2 = torch.cuda.Stream()
                h2 = lexls_amd.BatchedLexLSE(batch // 2, NVAR, DIMS, device=device_index)
                h2.set_stream(st2.cuda_stream)
                h2.set_kernel_policy(3)  # same kernel as the timed step: the automatic choice looks at ONE handle's batch, here two share the chip
                h2.setProblemDevice(lod_dev.data_ptr() + i * (batch // 2) * lod_host.shape[1] * lod_host.shape[2] * 8)
                halves.append((h2, st2))
            for h2, _ in halves:
                h2.factorize_solve(keep_factor=args.keep_factor)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                for h2, _ in halves:
                    h2.factorize_solve(keep_factor=args.keep_factor)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            overlapped = {"streams": 2, "value": batch * args.steps / dt, "unit": "factorizations/s", "ms_per_step": 1e3 * dt / args.steps,
                          "note": "two half batches in flight on two streams; reported next to, not as, the metric"}
            del halves
        except Exception as exc:  # noqa: BLE001
            overlapped = {"error": f"{type(exc).__name__}: {exc}"}

    # ---- outside the timed region: the one place RCCL carries data (north_star: "used only to scatter problem blocks and gather
    #      solutions").  Rank 0 builds the whole batch, scatters the blocks, every rank checks its block against the shard it generated
    #      itself, the solutions are gathered on rank 0.  Reported separately; never part of `value`.  A failure here is
    #      reported, it does not take the benchmark down.
    scatter_gather = None
    if world > 1 and not os.environ.get("LEXLS_BENCH_SKIP_SCATTER"):
        try:
            from lexls_amd import sharding
            cdev = torch.device("cuda", device_index) if args.dist_backend == "nccl" else torch.device("cpu")
            cap = lod_host.shape[2]
            root = None
            if rank == 0:
                root = torch.from_numpy(np.concatenate([lod_host] + [P.lse_batch(SEED0 + r * batch, batch, NVAR, DIMS) for r in range(1, world)])).to(cdev)
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mine = sharding.scatter_problems(root, batch * world, NVAR, cap, device=cdev)
            torch.cuda.synchronize()
            dist.barrier()
            t_scatter = time.perf_counter() - t0
            same = bool(torch.equal(mine.cpu(), torch.from_numpy(lod_host)))  # the block that arrived == the shard this rank generated itself
            t0 = time.perf_counter()
            xg = sharding.gather_solutions(torch.from_numpy(x).to(cdev), batch * world, NVAR)
            torch.cuda.synchronize()
            dist.barrier()
            t_gather = time.perf_counter() - t0
            flags = torch.tensor([float(same)], dtype=torch.float64, device=coll_device)
            dist.all_reduce(flags, op=dist.ReduceOp.MIN)
            scatter_gather = {"scatter_ms": 1e3 * t_scatter, "gather_ms": 1e3 * t_gather, "bytes_scattered": int(8 * batch * world * (NVAR + 1) * cap),
                              "blocks_verified": bool(flags.item()), "gathered_shape": list(xg.shape) if rank == 0 else None,
                              "backend": args.dist_backend}
        except Exception as exc:  # noqa: BLE001
            scatter_gather = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        total = batch * world * args.steps
        value = total / elapsed
        bytes_per = P.algorithmic_bytes(NVAR, DIMS, write_factor=args.keep_factor)
        fm = P.flop_model(NVAR, DIMS)
        flops_per = fm["total"]
        achieved = bytes_per * batch / (kernel_ms * 1e-3) / 1e9
        # what the x-only kernel actually touches / executes: the columns are exhausted inside level 3 (ranks 12,12,12,4,0), so the rows of
        # level 4 are never read and their elimination never runs (legitimate for x; the model above counts them)
        touched = P.bytes_touched_x_only(NVAR, DIMS) if not args.keep_factor else bytes_per
        executed = P.flops_executed_x_only(NVAR, DIMS) if not args.keep_factor else flops_per
        line = {
            "metric": "batched fp64 l-QR factorizations/s (factorize+solve)",
            "value": value,
            "unit": "factorizations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "warmup_run": warm_run,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"IK batch: {batch} problems/GPU x (n={NVAR}, 5 levels x 12 rows), BASELINE configs[2] (x{world} GPUs -> configs[3] at 8)",
                       "per_gpu_batch": batch, "global_batch": batch * world, "parallelism": f"batch-sharded x{world}",
                       "variant": "factor kept in HBM" if args.keep_factor else "x-only", "kernel": solver.last_kernel()},
            "gflops_fp64": value * flops_per / 1e9,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": _committed_traffic(args.keep_factor), "traffic_source": "profiles/pmc_summary.json (committed rocprofv3 PMC passes of this command, corrected as the summary states)",
                         "algorithmic_bytes_per_launch": bytes_per * batch, "kernel_ms": kernel_ms, "resident_batches": nres,
                         "resident_bytes": int(nres * lod_host.nbytes),
                         "bytes_touched_per_launch": touched * batch, "frac_on_bytes_touched": touched * batch / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "note": "frac = SURVEY 8(d)'s algorithmic bytes (20,000 B/problem) / time / peak, as bench.py's contract defines it; the x-only kernel never reads "
                                 "level 4 (its columns are exhausted): frac_on_bytes_touched is the share of the bytes it moves",
                         "contract": ("pivots / ranks exact (norms within 2^-40 of each other ordered by position), x within 1e-10 (lqr_qtol)" if solver.last_kernel().startswith("lqr_qtol")
                                      else "pivots / ranks exact (norms compared by value), x within 1e-10 (lqr_mfma)" if solver.last_kernel().startswith("lqr_mfma")
                                      else "bit-identical to the oracle")},
            # the same launch against the fp64 vector peak (the path is issue-bound, not byte-bound: DESIGN.md section 5)
            "roofline_fp64": {"bound": "fp64 vector", "achieved": flops_per * batch / (kernel_ms * 1e-3) / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": flops_per * batch / (kernel_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                              "flops_per_problem": flops_per, "flops_executed_per_problem": executed,
                              "frac_on_flops_executed": executed * batch / (kernel_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS},
        }
        if world == 1 and not args.no_extras and not args.keep_factor:
            line["factor_kept"] = side_factor_kept(solver, stream, ptrs, nres, batch, torch)
            line["config1_large"] = side_config1_large(device_index, not args.no_cpu_baseline)
            line["config4_lsi"] = side_config4_lsi(device_index, not args.no_cpu_baseline)
        line["shard_checksums"] = shard_checksums  # sum |x| of every rank's shard (all ranks' ranks were verified before this line is printed)
        if scatter_gather is not None:
            line["scatter_gather"] = scatter_gather
            line["scatter_gather_ok"] = bool("error" not in scatter_gather and scatter_gather.get("blocks_verified"))
        if world == 1 and overlapped is not None:
            line["overlapped"] = overlapped
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(lod_host, args.cpu_seconds)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.destroy_process_group()


def bench_lsi(args, world, rank, device_index, coll_device, torch, dist):
    """BASELINE configs[4]: lock-step LexLSI batch.  The job's instances (ids 20260500 + i) are split into contiguous blocks, one block
    and ONE batch object (lexls_lsi_batch_*) per rank; no rank talks to another between active-set iterations.  A step = one warm-started
    solve of every instance (working set and x of the unperturbed neighbour as the guess, right-hand sides perturbed by 0.9 N(0,1):
    ~30 factorizations per instance).  value = factorizations of all instances of all ranks / time (max over ranks): strong scaling."""
    from lexls_amd import lexlsi, sharding, problems as P
    n, dims, total = NVAR, DIMS, args.lsi_batch
    lo, hi = sharding.shard_range(total, rank, world)
    mine = hi - lo
    base = lexlsi.pack_batch(n, [P.lsi_problem(20260500 + i, n, dims) for i in range(lo, hi)])
    pert = lexlsi.pack_batch(n, [P.lsi_problem(20260500 + i, n, dims, perturb=0.9) for i in range(lo, hi)])
    srv = lexlsi.LsiBatch(n, base.dims, base.types, mine, device=device_index)
    cold = srv.run(base)
    guess = np.where(cold["active"] == 3, 0, cold["active"]).astype(np.uint8)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(2, min(args.warmup, 5))):
        r = srv.run(pert, active_guess=guess, x0=cold["x"])
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = srv.run(pert, active_guess=guess, x0=cold["x"])
    barrier()
    elapsed = time.perf_counter() - t0
    f = np.array([i["factorizations"] for i in r["info"]], np.float64)
    solved = sum(i["status"] == 0 for i in r["info"])
    t = torch.tensor([elapsed, f.sum(), float(solved), float(f.max())], dtype=torch.float64, device=coll_device)
    if world > 1:
        tm = t.clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, fsum, nsolved, fmax = float(tm[0]), float(t[1]), int(t[2]), float(tm[3])
    else:
        fsum, nsolved, fmax = float(t[1]), int(t[2]), float(t[3])
    stats = srv.stats()
    srv.close()
    if nsolved != total:
        raise SystemExit(f"bench --workload lsi: only {nsolved} of {total} instances solved — refusing to report a number")
    if rank == 0:
        extra = {}
        flops = P.flop_model(n, dims)["total"]  # per factorization of a full problem (an upper bound: working sets hold fewer rows)
        rate = fsum * args.steps / elapsed
        extra["roofline"] = {"bound": "fp64", "achieved": rate * flops / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": rate * flops / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                             "traffic": None, "note": "flops of a full 60 x 41 problem per factorization (upper bound); the path is a latency chain per instance (DESIGN.md 3.5)"}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle_ctypes as oc
            share = host_cpu_share()
            cores = max(1, int(round(share["effective_cores"])))
            nfc, tcc = 0, 0.0
            while tcc < 2.0:  # (sustained: one pass is a burst the cgroup's CPU quota does not throttle yet)
                a_, b_ = oc.lsi_time_batch(pert, guess, cold["x"], 4 * cores)
                nfc, tcc = nfc + a_, tcc + b_
            extra["cpu_baseline"] = {"value": nfc / tcc, "unit": "factorizations/s", "cores": cores, "threads": 4 * cores, "kind": "port", **share,
                                     "sample": f"the same warm-started batch of {total} instances, repeated for {tcc:.1f} s ({nfc} factorizations) through the host driver over the oracle, "
                                               f"instances dealt out to {4 * cores} std::threads inside the library"}
        print(json.dumps({
            **extra,
            "metric": "batched fp64 l-QR factorizations/s (LexLSI lock-step batch: factorize+solve+removal search per active-set iteration)",
            "value": fsum * args.steps / elapsed, "unit": "factorizations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"LexLSI lock-step batch: {total} instances x (n={n}, 5 levels x 12 rows, level 0 simple bounds), warm-started, BASELINE configs[4]",
                       "instances_per_gpu": mine, "parallelism": f"instance blocks x{world}", "mean_factorizations": fsum / total, "max_factorizations": fmax,
                       "stages_rank0": stats},
        }), flush=True)
    if world > 1:
        dist.destroy_process_group()


def host_cpu_share():
    """What this job may use of the host: CPUs in the affinity mask, the cgroup CPU quota, and the smaller of the two ("effective_cores")."""
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        aff = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    quota = float(parts[0]) / float(parts[1])
            else:
                q = float(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        quota = q / float(f.read().split()[0])
            break
        except Exception:  # noqa: BLE001
            continue
    eff = aff if quota is None else max(1.0, min(float(aff), quota))
    return {"affinity_cpus": aff, "cgroup_cpu_quota": quota, "effective_cores": eff}


def self_launch(n):
    """python bench.py --gpus N without a launcher: N child processes (one rank per GPU), 127.0.0.1 rendezvous, rank 0's stdout is ours.
    Every child is watched: the first one that exits with an error takes the others down with it (a rank that dies before the rendezvous
    would otherwise leave the rest in init_process_group until its timeout) and its stderr tail is shown; the launcher exits with that code."""
    import subprocess
    import tempfile
    rdv = tempfile.NamedTemporaryFile(prefix="lexls_bench_rdv_", delete=False)  # file:// rendezvous: no probed port that another process could take
    rdv.close()
    os.unlink(rdv.name)
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", LEXLS_BENCH_RDV="file://" + rdv.name)
        env.setdefault("MASTER_PORT", "29531")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        err = tempfile.TemporaryFile()
        errs.append(err)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=err))
    import threading
    out_chunks = []
    reader = threading.Thread(target=lambda: out_chunks.append(procs[0].stdout.read()), daemon=True)  # (rank 0's pipe must be drained while we poll)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, pr in enumerate(procs):
            if pr.poll() not in (None, 0):
                failed = r
                break
        time.sleep(0.05)
    if failed is None:
        failed = next((r for r, pr in enumerate(procs) if pr.returncode != 0), None)
    if failed is not None:
        for pr in procs:
            if pr.poll() is None:
                pr.terminate()
        for pr in procs:
            try:
                pr.wait(timeout=10)
            except Exception:  # noqa: BLE001
                pr.kill()
    reader.join(timeout=10)
    sys.stdout.write(b"".join(out_chunks).decode())
    sys.stdout.flush()
    try:
        os.unlink(rdv.name)
    except OSError:
        pass
    if failed is not None:
        errs[failed].seek(0)
        tail = errs[failed].read().decode(errors="replace")[-2000:]
        sys.stderr.write(f"bench.py: rank {failed} exited with code {procs[failed].returncode}; the other ranks were stopped\n{tail}\n")
        raise SystemExit(procs[failed].returncode or 1)


def dry_run(args, world, rank):
    """The multi-rank control flow without a GPU: process group, barriers, max-over-ranks of the elapsed time, rank 0 prints the line."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        rdv = {"init_method": os.environ["LEXLS_BENCH_RDV"], "rank": rank, "world_size": world} if os.environ.get("LEXLS_BENCH_RDV") else {}
        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(minutes=2), **rdv)
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))  # ranks finish at different times: the reported time is the slowest rank's
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64)
    ids = torch.tensor([float(rank)], dtype=torch.float64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(ids, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU work)", "dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "max_rank_seconds": float(t.item()), "rank_id_sum": float(ids.item()), "workload": args.workload}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def side_factor_kept(solver, stream, ptrs, nres, batch, torch):
    """Next to the metric, never in `value`: the same batches with the factor kept in HBM (all the work of factorize(): the last level too,
    factor / Householder scalars / pivots written), bit-identical kernels, against ITS algorithmic bytes (40,360 B per problem)."""
    from lexls_amd import problems as P
    try:
        reps = 40
        with torch.cuda.stream(stream):
            for i in range(8):
                solver.setProblemDevice(ptrs[i % nres])
                solver.factorize_solve(keep_factor=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for i in range(reps):
                solver.setProblemDevice(ptrs[i % nres])
                solver.factorize_solve(keep_factor=True)
            e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        kern = solver.last_kernel()
        b = P.algorithmic_bytes(NVAR, DIMS, write_factor=True)
        f = P.flop_model(NVAR, DIMS)["total"]
        solver.setProblemDevice(ptrs[0])
        solver.factorize_solve(keep_factor=False)  # leave the x-only solution of batch 0 behind, as the timed steps did
        torch.cuda.synchronize()
        return {"kernel": kern, "kernel_ms": ms, "value": batch / (ms * 1e-3), "unit": "factorizations/s", "algorithmic_bytes_per_problem": b,
                "roofline": {"bound": "hbm", "achieved": b * batch / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b * batch / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "roofline_fp64_frac": f * batch / (ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS, "contract": "bit-identical to the oracle"}
    except Exception as exc:  # noqa: BLE001
        return {"error": f"{type(exc).__name__}: {exc}"}


def side_config1_large(device_index, with_cpu):
    """BASELINE configs[1]: ONE equality problem, n = 512, 4 levels x 256 rows, the whole chip on it (lqr_large)."""
    import lexls_amd
    from lexls_amd import problems as P
    try:
        n, dims = 512, [256] * 4
        lod = P.lse_batch(20260001, 1, n, dims)
        s = lexls_amd.BatchedLexLSE(1, n, dims, device=device_index)
        s.setProblem(lod)
        for _ in range(3):
            s.factorize_solve(True)
        s.synchronize()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            s.factorize_solve(True)
        s.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / reps
        flops = P.flop_model(n, dims)["total"]
        out = {"workload": "single equality l-QR, n=512, 4 levels x 256 rows (BASELINE configs[1])", "kernel": s.last_kernel(), "ms": ms, "gflops": flops / ms / 1e6,
               "roofline": {"bound": "fp64", "achieved": flops / ms / 1e9, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / ms / 1e9 / FP64_VECTOR_PEAK_TFLOPS},
               "algorithmic_bytes": P.algorithmic_bytes(n, dims, write_factor=True), "contract": "pivots / ranks exact, values within 1e-10 (step-per-pivot path)"}
        if with_cpu:
            from oracle import oracle_ctypes as oc
            tc, _ = oc.lse_time(lod, dims, n, 1, 2)
            out["cpu_baseline"] = {"value": 1e3 * tc / 2, "unit": "ms per factorize+solve", "cores": 1, "kind": "port", "gflops": flops / (tc / 2) / 1e9,
                                   "sample": "2 solves of the same problem, scalar g++ -O3 restatement (oracle/), one thread"}
        return out
    except Exception as exc:  # noqa: BLE001
        return {"error": f"{type(exc).__name__}: {exc}"}


def side_config4_lsi(device_index, with_cpu, total=1024):
    """BASELINE configs[4]: lock-step LexLSI batch of 1024 instances, warm-started to ~30 factorizations each (as --workload lsi)."""
    try:
        from lexls_amd import lexlsi, problems as P
        n, dims = NVAR, DIMS
        base = lexlsi.pack_batch(n, [P.lsi_problem(20260500 + i, n, dims) for i in range(total)])
        pert = lexlsi.pack_batch(n, [P.lsi_problem(20260500 + i, n, dims, perturb=0.9) for i in range(total)])
        srv = lexlsi.LsiBatch(n, base.dims, base.types, total, device=device_index)
        cold = srv.run(base)
        guess = np.where(cold["active"] == 3, 0, cold["active"]).astype(np.uint8)
        for _ in range(2):
            r = srv.run(pert, active_guess=guess, x0=cold["x"])
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            r = srv.run(pert, active_guess=guess, x0=cold["x"])
        dt = (time.perf_counter() - t0) / reps
        f = np.array([i["factorizations"] for i in r["info"]], np.float64)
        solved = int(sum(i["status"] == 0 for i in r["info"]))
        stats = srv.stats()
        srv.close()
        flops = P.flop_model(n, dims)["total"]  # per factorization of a full problem (an upper bound: working sets hold fewer rows)
        out = {"workload": f"LexLSI lock-step batch, {total} instances x (n=40, 5 levels x 12 rows, level 0 simple bounds), warm-started (BASELINE configs[4])",
               "ms_per_batch": 1e3 * dt, "factorizations_per_s": float(f.sum()) / dt, "mean_factorizations": float(f.mean()), "max_factorizations": float(f.max()),
               "solved": solved, "stages": stats,
               "roofline": {"bound": "fp64", "achieved": float(f.sum()) * flops / dt / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": float(f.sum()) * flops / dt / 1e12 / FP64_VECTOR_PEAK_TFLOPS, "note": "flops of a full 60 x 41 problem per factorization (upper bound)"}}
        if with_cpu:
            from oracle import oracle_ctypes as oc
            # the whole warm-started batch through the oracle-backed host driver (oracle_lsi_time_batch: instances dealt out to host threads inside the
            # library, nothing of Python in the timed region), on the thread count that delivers most, and on one thread for the per-core figure
            share = host_cpu_share()
            hw = max(1, oc.hardware_threads())
            best = (0.0, 1, 0, 0.0)
            for cand in sorted({int(round(share["effective_cores"])), 2 * int(round(share["effective_cores"])), 4 * int(round(share["effective_cores"]))}):
                if cand < 1 or cand > hw:
                    continue
                nf, tc = 0, 0.0
                while tc < 2.0:  # (sustained: one pass is ~15 ms, a burst the cgroup's CPU quota does not throttle yet)
                    a, b = oc.lsi_time_batch(pert, guess, cold["x"], cand)
                    nf, tc = nf + a, tc + b
                if nf / tc > best[0]:
                    best = (nf / tc, cand, nf, tc)
            nf1, tc1 = oc.lsi_time_batch(pert, guess, cold["x"], 1)
            out["cpu_baseline"] = {"value": best[0], "unit": "factorizations/s", "cores": int(round(min(best[1], share["effective_cores"]))), "threads": best[1], "kind": "port",
                                   **share, "single_thread": nf1 / tc1,
                                   "sample": f"the same warm-started batch of {total} instances, repeated for {best[3]:.1f} s ({best[2]} factorizations on {best[1]} std::threads; one pass on one thread: {tc1:.2f} s) "
                                             f"through the host driver over the oracle, instances dealt out inside the library"}
        return out
    except Exception as exc:  # noqa: BLE001
        return {"error": f"{type(exc).__name__}: {exc}"}


def _committed_traffic(keep_factor):
    """HBM bytes per launch from the committed rocprofv3 PMC summary of this same command (profiles/), if present."""
    path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            d = json.load(f)
        return d.get("factor" if keep_factor else "x_only", {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def cpu_baseline(lod_host, target_seconds):
    """The CPU restatement of the reference algorithm (oracle/, kind 'port'; the Eigen-backed reference cannot be built:
    Eigen is absent) on this host's cores, on a bounded sample of the same workload."""
    from oracle import oracle_ctypes as oc
    hw = max(1, oc.hardware_threads())
    sample = lod_host[:min(len(lod_host), 4096)]
    # the box may grant this job fewer cores than the host has hardware threads: use the thread count that actually delivers the
    # highest rate on a short probe, and report that count as `cores`
    best_threads, best_rate = 1, 0.0
    for cand in sorted({1, 8, 16, 32, 64, 128, hw}):
        if cand > hw:
            continue
        tp, _ = oc.lse_time(sample, DIMS, NVAR, cand, 2)
        rate = 2 * len(sample) / max(tp, 1e-9)
        if rate > best_rate * 1.03:
            best_threads, best_rate = cand, rate
    threads = best_threads
    # the sustained rate is well below the two-pass probe's: the sample is run in slices until the target time is reached
    repeats, t = 0, 0.0
    slice_passes = int(max(2, min(5000, 2.0 * best_rate / len(sample))))
    while t < target_seconds and repeats < 20000:
        ts_, _ = oc.lse_time(sample, DIMS, NVAR, threads, slice_passes)
        repeats += slice_passes
        t += ts_
        slice_passes = int(max(2, min(5000, 2.0 * repeats / max(t, 1e-9))))
    ts, _ = oc.lse_time(sample[:256], DIMS, NVAR, 1, 8)  # one thread alone, for the per-thread rate without contention
    share = host_cpu_share()
    # `cores` = what the job can actually run on (affinity mask and cgroup quota), not the number of threads the probe preferred
    return {"value": len(sample) * repeats / t, "unit": "factorizations/s", "cores": int(round(min(threads, share["effective_cores"]))), "threads": threads,
            "kind": "port", **share,
            "per_thread": len(sample) * repeats / t / threads, "single_thread_alone": 256 * 8 / ts,
            "sample": f"{repeats} passes over {len(sample)} problems of the bench batch ({t:.1f} s in ~2-s slices, g++ -O3 scalar restatement, {threads} std::threads started once per slice, "
                      f"one solver object per thread; {hw} hardware threads on the host, thread count chosen by a probe)"}


if __name__ == "__main__":
    main()
