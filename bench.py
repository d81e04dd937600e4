#!/usr/bin/env python3
"""bench.py -- EM hot-path benchmark on MI355X (contract: see the task statement / DESIGN.md section 6).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2] [--kernel 0..3]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one EM round (E step + M step, EM.cpp:365-416 with frozen CSR values) over the whole
read x transcript matrix of BASELINE.json configs[1] (SingleQModel-shaped: 10 M reads, 50 k
transcripts, ~5-6 alignments per read), synthetic and seeded (tools/synth_data.py), resident in HBM
before the timed region.  value = read-alignments processed per second = nnz * K / wall.
N > 1: weak scaling -- every rank holds its own configs[1]-sized shard of reads over the same
transcriptome, theta replicated, one RCCL all-reduce of the M+1 fractional counts per round.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline_port(wl, budget_s=12.0):
    """The CPU restatement (oracle/, single thread) timed on a bounded row-subsample of the same workload."""
    from oracle import pyoracle as orc
    N1 = len(wl["row_ptr"]) - 1
    sub = min(N1, 500_000)
    rp = np.ascontiguousarray(wl["row_ptr"][:sub + 1])
    nnz = int(rp[-1])
    sid, cp, ncp = wl["sid"][:nnz], wl["conprb"][:nnz], wl["ncp"][:sub]
    theta = wl["theta0"].copy()
    N0 = wl["N0"] * sub / N1
    t0 = time.perf_counter()
    rounds = 0
    while True:
        counts = orc.em_estep(wl["M"], rp, sid, cp, ncp, theta)
        _, theta, *_ = orc.em_mstep(wl["M"], N0, counts, theta)
        rounds += 1
        el = time.perf_counter() - t0
        if el > budget_s or rounds >= 200:
            break
    return {"value": nnz * rounds / el, "unit": "read-alignments/s", "cores": 1, "kind": "port",
            "sample": "first %d reads (%d alignments) of the same workload, %d EM rounds, oracle/rsem_oracle.c"
                      % (sub, nnz, rounds)}


def cpu_baseline_reference(n_reads=1_000_000, M=20_000, limit_s=150.0):
    """The UNMODIFIED reference binary (oracle/_ref/rsem-run-em, built from /root/reference) on this host's
    cores, on a bounded SingleQModel sample written by tools/gen_temp.cpp; per-round time of the rounds
    with frozen alignment probabilities (ROUND >= 12) from the arrival times of its 'ROUND =' lines
    (EM.cpp:415).  The drop-in rsem_amd/bin/rsem-run-em is run on the same files for the wall-clock ratio."""
    import shutil
    import subprocess
    import tempfile
    gen = os.path.join(ROOT, "tools", "bin", "gen_temp")
    ref_em = os.path.join(ROOT, "oracle", "_ref", "rsem-run-em")
    ref_idx = os.path.join(ROOT, "oracle", "_ref", "rsem-build-read-index")
    new_em = os.path.join(ROOT, "rsem_amd", "bin", "rsem-run-em")
    if not all(os.path.exists(p) for p in (gen, ref_em, ref_idx, new_em)):
        return None
    d = tempfile.mkdtemp(prefix="rsem_bench_")
    try:
        out = subprocess.run([gen, d, str(n_reads), str(M), "1"], stdout=subprocess.PIPE, text=True, check=True).stdout
        nhits = int(out.split("nHits=")[1].split()[0])
        subprocess.run([ref_idx, "32", "1", "1", os.path.join(d, "temp", "s_alignable.fq")], stdout=subprocess.DEVNULL, check=True)
        args = [os.path.join(d, "ref"), "1", os.path.join(d, "s"), os.path.join(d, "temp", "s"), os.path.join(d, "stat", "s")]
        cores = min(os.cpu_count() or 1, 64)
        t0 = time.perf_counter()
        p = subprocess.Popen([ref_em] + args + ["-p", str(cores)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        stamps = []
        finished = True
        for line in p.stdout:
            if line.startswith("ROUND ="):
                stamps.append((int(line.split(",")[0].split("=")[1]), time.perf_counter()))
            if time.perf_counter() - t0 > limit_s:
                p.kill()
                finished = False
                break
        p.wait()
        ref_wall = time.perf_counter() - t0
        late = [(r, t) for r, t in stamps if r >= 12]
        if len(late) < 3:
            return None
        per_round = (late[-1][1] - late[0][1]) / (late[-1][0] - late[0][0])
        res = {"value": nhits / per_round, "unit": "read-alignments/s", "cores": cores, "kind": "reference",
               "sample": "oracle/_ref/rsem-run-em -p %d on a generated SingleQModel sample: %d reads, %d alignments, %d "
                         "transcripts; %d rounds >= 12 timed (%.3f ms/round)" % (cores, n_reads, nhits, M, len(late) - 1, per_round * 1e3),
               "reference_rounds": stamps[-1][0], "reference_finished": finished, "reference_wall_s": ref_wall,
               "host_cores_available": os.cpu_count()}
        t0 = time.perf_counter()
        r = subprocess.run([new_em] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        res["dropin_wall_s"] = time.perf_counter() - t0
        res["dropin_ok"] = r.returncode == 0
        rl = [l for l in r.stdout.split("\n") if l.startswith("ROUND")]
        res["dropin_rounds"] = int(rl[-1].split(",")[0].split("=")[1]) if rl else None
        if finished and res["dropin_ok"]:
            res["wall_clock_speedup_same_files"] = ref_wall / res["dropin_wall_s"]
        return res
    finally:
        shutil.rmtree(d, ignore_errors=True)


def ci_leg(capi, M, nCV=1000, nSpC=50):
    """rsem-calculate-credibility-intervals at the workload's M with rsem-calculate-expression's defaults
    (1000 count vectors x 50 draws): synthetic count vectors, device-side times from rsem_ci_profile."""
    try:
        rng = np.random.default_rng(3)
        mean = np.exp(rng.normal(3.0, 2.5, M + 1)) * (rng.random(M + 1) > 0.3)
        cv = rng.poisson(mean, size=(nCV, M + 1)).astype(np.int32)
        eel = np.concatenate([[0.0], rng.uniform(300, 4000, M)])
        starts = np.arange(1, M + 2, 5, dtype=np.int32)
        if starts[-1] != M + 1:
            starts = np.append(starts, M + 1).astype(np.int32)
        t0 = time.perf_counter()
        out = capi.ci_calculate(cv, nSpC, eel, np.ones(M + 1), starts, 0.95, 1.0, seed=1)
        wall = time.perf_counter() - t0
        p = out["profile"]
        return {"transcripts": M, "samples": nCV * nSpC, "wall_s": wall, "device_ms": p.total_ms,
                "gamma_draws_per_s": p.n_draws / p.sample_ms * 1e3, "keys_sorted_per_s": p.n_keys_sorted / p.sort_ms * 1e3,
                "interval_ms": p.interval_ms}
    except Exception as e:
        return {"error": str(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gibbs", action="store_true")
    ap.add_argument("--no-ci", action="store_true")
    ap.add_argument("--gibbs-sweeps", type=int, default=30)
    args = ap.parse_args()

    import torch
    from rsem_amd import build, capi
    from tools.synth_data import make_em_workload, to_gibbs_items

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: librsem_hip has no CPU path")
    if os.environ.get("BENCH_SINGLE_DEVICE"):  # debugging aid: several ranks share GPU 0 (use with BENCH_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if rank == 0:
        build.build()
    if world > 1:
        dist.barrier()

    t0 = time.perf_counter()
    wl = make_em_workload(args.config, shard=rank, scale=args.scale)
    N1, nnz, M = len(wl["row_ptr"]) - 1, len(wl["sid"]), wl["M"]
    log("[rank %d] workload %s: N1=%d nnz=%d M=%d (%.1f s)" % (rank, args.config, N1, nnz, M, time.perf_counter() - t0))
    t0 = time.perf_counter()
    ctx = capi.EmContext(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], device=local)
    ctx.set_option("kernel", args.kernel)
    log("[rank %d] upload + device layout: %.2f s" % (rank, time.perf_counter() - t0))
    alg_bytes = 12 * nnz + 16 * N1 + 16 * (M + 1)
    K, W = args.steps, args.warmup

    if world == 1:
        if W > 0:
            ctx.run(wl["theta0"], wl["N0"], min_round=W, max_round=W)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = ctx.run(wl["theta0"], wl["N0"], min_round=K, max_round=K, profile=True)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        prof = out["profile"]
        assert out["rounds"] == K
        estep_ms = prof.estep_ms_sum / max(prof.estep_launches, 1)
        theta_sum = float(out["theta"].sum())
        total_nnz = nnz
    else:
        import torch.distributed as dist
        dev = torch.device("cuda", local)
        theta = [torch.from_numpy(wl["theta0"]).to(dev), torch.zeros(M + 1, dtype=torch.float64, device=dev)]
        counts = torch.zeros(M + 1, dtype=torch.float64, device=dev)
        stats = torch.zeros(3, dtype=torch.float64, device=dev)
        N0g = float(wl["N0"] * world)
        stream = torch.cuda.current_stream().cuda_stream

        def one_round(r, ev=None):
            a, b = theta[r & 1], theta[(r + 1) & 1]
            if ev:
                ev[0].record()
            ctx.estep_device(a.data_ptr(), counts.data_ptr(), stream)
            if ev:
                ev[1].record()
            dist.all_reduce(counts)  # EM.cpp:385-389 across shards
            ctx.mstep_device(counts.data_ptr(), N0g, a.data_ptr(), b.data_ptr(), stats.data_ptr(), stream)

        for r in range(W):
            one_round(r)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(K):
            one_round(W + r, evs[r])
        torch.cuda.synchronize()
        dist.barrier()
        elapsed = time.perf_counter() - t0
        estep_ms = sum(a.elapsed_time(b) for a, b in evs) / K
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tn = torch.tensor([float(nnz)], dtype=torch.float64, device=dev)
        dist.all_reduce(tn)
        total_nnz = int(tn.item())
        theta_sum = float(theta[(W + K) & 1].sum().item())

    gibbs = None
    if not args.no_gibbs:
        # Gibbs PARALLEL sweeps on the same matrix (one chain per GPU; single reduce of the accumulators)
        try:
            irp, isid, icp = to_gibbs_items(wl)
            g = capi.GibbsContext(M, irp, isid, icp, np.zeros(M + 1, np.int32), None, 1.0, (M + 1) + wl["N0"] + N1, wl["N0"],
                                  np.full(M + 1, 1000.0), np.ones(M + 1), np.arange(1, M + 2, 50, dtype=np.int32)[: (M // 50) + 1]
                                  if (M % 50 == 0) else np.array([1, M + 1], np.int32), device=local)
            cv, acc, ms = g.run(capi.GIBBS_PARALLEL, 1 + rank, args.gibbs_sweeps - 2, 2, 1, thin=1, want_vectors=False)
            g.close()
            if world > 1:
                buf = torch.from_numpy(np.concatenate(acc)).to(dev)
                dist.reduce(buf, dst=0)  # Gibbs.cpp:372-388 across chains
            b_g = 12 * (len(isid) - N1) + 16 * N1  # conprb + sid per alignment, noise conprb + row slot per read
            gibbs = {"mode": "parallel (data-augmentation)", "chains": world, "items_per_chain": int(len(isid)),
                     "ms_per_sweep": ms, "algorithmic_GBps_per_chain": b_g / ms / 1e6 if ms > 0 else None, "sweeps_per_s_all_chains": world * 1e3 / ms if ms > 0 else None,
                     "items_per_s_all_chains": world * len(isid) * 1e3 / ms if ms > 0 else None}
        except Exception as e:  # the EM line must still be printed
            gibbs = {"error": str(e)}

    ctx.close()
    if rank == 0:
        achieved = alg_bytes / (estep_ms * 1e-3) / 1e9
        traffic = None  # PMC passes cannot run inside this process: take the committed measurement for this workload
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pm = json.load(f)
            if pm.get("workload") == args.config and args.scale == 1.0 and args.kernel in (0, 3):
                traffic = pm["traffic_bytes_per_launch"]
        except Exception:
            pass
        line = {
            "metric": "EM read-alignments/s (nnz x EM iterations per second), rsem-run-em theta-only rounds",
            "value": total_nnz * K / elapsed, "unit": "read-alignments/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed * 1e3 / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "em_iterations_per_s": K / elapsed,
            "config": {"workload": "%s: EM matrix of %d reads x %d transcripts, "
                                   "%d alignments (%.2f/read) per GPU, frozen conprb (rounds >= 12)"
                                   % ({"C2": "BASELINE configs[1] (SingleQModel-shaped)", "C3": "BASELINE configs[2] (PairedEndQModel-shaped)",
                                       "C5": "BASELINE configs[4] (multi-mapping stress)"}.get(args.config, args.config),
                                      N1, M, nnz, nnz / max(N1, 1)),
                       "synthetic_config": args.config, "kernel": args.kernel,
                       "parallelism": "1 GPU" if world == 1 else "read-sharded x%d + RCCL all-reduce(M+1 f64)/round" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "k_estep_lane (E step)", "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_ms": estep_ms},
            "checks": {"theta_sum": theta_sum},
            "gibbs": gibbs,
        }
        if not args.no_ci and world == 1:
            line["credibility_intervals"] = ci_leg(capi, M)
        if not args.no_cpu_baseline and world == 1:  # reported baseline: rank 0 at N=1 only
            cb = cpu_baseline_reference()
            if cb is None:
                cb = cpu_baseline_port(wl)
            else:
                line["cpu_baseline_port_1core"] = cpu_baseline_port(wl, budget_s=6.0)
            line["cpu_baseline"] = cb
            line["speedup_vs_cpu_baseline_rounds"] = line["value"] / cb["value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
