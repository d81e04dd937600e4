#!/usr/bin/env python3
"""bench.py -- EM hot-path benchmark on MI355X (contract: see the task statement / DESIGN.md section 6).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3] [--legs C2[,C5]] [--kernel 0..3]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one EM round (E step + M step, EM.cpp:365-416 with frozen CSR values) over the whole read x transcript
matrix of BASELINE.json configs[2] (the configuration the north-star target is quoted on: PairedEndQModel-shaped,
50 M reads, 200 k transcripts, ~10 alignments per read; `--config` C2 / C5 select configs[1] / configs[4]), synthetic
and seeded (tools/synth_data.py), resident in HBM before the timed region.  value = read-alignments processed per
second = nnz * rounds / wall.  The timed region is repeated (K rounds each time, bracketed by a barrier and a device
synchronisation on both sides) until at least 0.5 s have been measured, whatever K is; ms_per_step is the mean.
N > 1: weak scaling -- every rank holds its own config-sized shard of reads over the same transcriptome, theta
replicated, one RCCL all-reduce of the M+1 fractional counts per round, issued from C++ on the kernel stream
(rsem_em_set_comm; the communicator id travels through torch.distributed).  Rank 0 prints ONE JSON line.
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
MIN_TIMED_S = 0.5
WORKLOADS = {"C2R": "configs[1]'s size with NO gene structure (every read hits random transcripts: worst case for tuple re-use and the LDS window)",
             "C2": "BASELINE configs[1] (SingleQModel-shaped)", "C3": "BASELINE configs[2] (PairedEndQModel-shaped, the north-star target config)",
             "C5": "BASELINE configs[4] (multi-mapping stress)"}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline_port(wl, budget_s=8.0):
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


# the reference binary's input: a complete .temp directory, same shape as the bench workload at a stated fraction of
# its reads (tools/gen_temp.cpp: genes of kmin..kmax overlapping isoforms)
CPU_SAMPLE = {"C2": dict(read_type=1, frac=0.1, M=50_000, iso="4-12"), "C3": dict(read_type=3, frac=0.05, M=200_000, iso="5-16"),
              "C5": dict(read_type=1, frac=0.01, M=500_000, iso="32-64")}


def cpu_baseline_reference(config, n_full, timed_s=10.0, limit_s=240.0):
    """The UNMODIFIED reference binary (oracle/_ref/rsem-run-em, built from /root/reference) on this host's cores, on
    a generated row-subsample-sized input of the SAME shape as the bench workload (model type, transcripts, isoforms
    per gene; `frac` of its reads).  Per-round time of the rounds with frozen alignment probabilities (ROUND >= 12,
    the rounds the GPU line times) from the arrival times of its 'ROUND =' lines (EM.cpp:415); the process is stopped
    once `timed_s` seconds of such rounds have been seen.  The E step is O(alignments) (EM.cpp:199-236), so the rate
    in read-alignments/s carries over to the full size; run with -p 64 and with -p <all cores>, best one reported."""
    import shutil
    import subprocess
    import tempfile
    gen = os.path.join(ROOT, "tools", "bin", "gen_temp")
    ref_em = os.path.join(ROOT, "oracle", "_ref", "rsem-run-em")
    ref_idx = os.path.join(ROOT, "oracle", "_ref", "rsem-build-read-index")
    if not all(os.path.exists(p) for p in (gen, ref_em, ref_idx)):
        return None
    cs = CPU_SAMPLE[config]
    rt = cs["read_type"]
    n_reads = max(100_000, int(n_full * cs["frac"] / 0.95))  # gen_temp: 95 % of the reads are alignable
    d = tempfile.mkdtemp(prefix="rsem_bench_", dir="/tmp")
    try:
        t0 = time.perf_counter()
        out = subprocess.run([gen, d, str(n_reads), str(cs["M"]), str(rt), "20250925", "100", "nosam", cs["iso"]],
                             stdout=subprocess.PIPE, text=True, check=True).stdout
        nhits = int(out.split("nHits=")[1].split()[0])
        n1 = int(out.split("N1=")[1].split()[0])
        reads = ["s_alignable.fq"] if rt == 1 else ["s_alignable_1.fq", "s_alignable_2.fq"]
        subprocess.run([ref_idx, "32", "1", "1"] + [os.path.join(d, "temp", r) for r in reads], stdout=subprocess.DEVNULL, check=True)
        gen_s = time.perf_counter() - t0
        args = [os.path.join(d, "ref"), str(rt), os.path.join(d, "s"), os.path.join(d, "temp", "s"), os.path.join(d, "stat", "s")]
        ncpu = os.cpu_count() or 1
        runs = []
        for cores in sorted({min(64, ncpu), ncpu}):
            t0 = time.perf_counter()
            p = subprocess.Popen([ref_em] + args + ["-p", str(cores)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            late = []
            for line in p.stdout:
                now = time.perf_counter()
                if line.startswith("ROUND ="):
                    r = int(line.split(",")[0].split("=")[1])
                    if r >= 12:
                        late.append((r, now))
                if (len(late) >= 3 and late[-1][1] - late[0][1] >= timed_s) or now - t0 > limit_s:
                    break
            p.kill()
            p.wait()
            if len(late) >= 3:
                per_round = (late[-1][1] - late[0][1]) / (late[-1][0] - late[0][0])
                runs.append({"cores": cores, "ms_per_round": per_round * 1e3, "rounds_timed": late[-1][0] - late[0][0],
                             "value": nhits / per_round, "startup_s": late[0][1] - t0})
        if not runs:
            return None
        best = max(runs, key=lambda r: r["value"])
        return {"value": best["value"], "unit": "read-alignments/s", "cores": best["cores"], "kind": "reference",
                "sample": "oracle/_ref/rsem-run-em on a generated %s input of the bench workload's shape at %.0f %% of its reads: %d "
                          "alignable reads, %d alignments (%.2f/read), %d transcripts; rounds >= 12 timed from its ROUND lines (%d rounds, "
                          "%.2f ms/round at -p %d); rate per alignment, so it carries over to the full size (E step is O(alignments))"
                          % ({1: "SingleQModel", 3: "PairedEndQModel"}[rt], cs["frac"] * 100, n1, nhits, nhits / n1, cs["M"],
                             best["rounds_timed"], best["ms_per_round"], best["cores"]),
                "runs": runs, "host_cores_available": ncpu, "generate_s": gen_s}
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


def timed_rounds(ctx, wl, N0, K, W, sync, barrier, agree=lambda x: x):
    """W untimed rounds, then K-round regions (barrier + device sync on both sides) until MIN_TIMED_S is covered; then one
    more K-round region with HIP events around every E-step launch.  Returns (elapsed_s, rounds, reps, estep_ms, theta_sum)."""
    if W > 0:
        ctx.run(wl["theta0"], N0, min_round=W, max_round=W)
    elapsed, reps = 0.0, 0
    while True:
        barrier()
        sync()
        t0 = time.perf_counter()
        out = ctx.run(wl["theta0"], N0, min_round=K, max_round=K)
        sync()
        barrier()
        elapsed += time.perf_counter() - t0
        reps += 1
        assert out["rounds"] == K
        if agree(elapsed) >= MIN_TIMED_S or reps >= 4096:  # agree(): the same decision on every rank (max over ranks)
            break
    prof = ctx.run(wl["theta0"], N0, min_round=min(K, 64), max_round=min(K, 64), profile=True)["profile"]
    estep_ms = prof.estep_ms_sum / max(prof.estep_launches, 1)
    return elapsed, reps * K, reps, estep_ms, float(out["theta"].sum())


def q32_leg(ctx, wl, N0, K, W, sync, alg_bytes, f64_ms_per_step, f64_estep_ms, traffic=None):
    """The same context with Q32 value planes (rsem_em_set_option "value_bits" 32: 32-bit mantissas + one exponent per
    read for the reads that qualify, include/rsem_hip.h) -- same rounds, same procedure, reported BESIDE the headline,
    which stays on the doubles.  Also: theta after K rounds in both formats."""
    try:
        N1, nnz, M = len(wl["row_ptr"]) - 1, len(wl["sid"]), wl["M"]
        ref = ctx.run(wl["theta0"], N0, min_round=K, max_round=K)["theta"]
        bytes64 = ctx.info("value_plane_bytes")
        t0 = time.perf_counter()
        ctx.set_option("value_bits", 32)
        relayout_s = time.perf_counter() - t0
        bytes32, n_q32 = ctx.info("value_plane_bytes"), ctx.info("reads_q32")
        el, rounds, reps, estep_ms, ts = timed_rounds(ctx, wl, N0, K, W, sync, lambda: None)
        th = ctx.run(wl["theta0"], N0, min_round=K, max_round=K)["theta"]
        ctx.set_option("value_bits", 64)
        big = ref >= 1e-7
        own = alg_bytes - (bytes64 - bytes32) + 2 * n_q32  # what this layout stores per round instead of the doubles
        return {"value_bits": 32, "value_range_bits": ctx.info("value_range_bits"), "reads_q32_fraction": n_q32 / max(N1, 1),
                "value_plane_bytes_f64": bytes64, "value_plane_bytes_q32": bytes32, "relayout_s": relayout_s,
                "ms_per_step": el * 1e3 / rounds, "timed_rounds": rounds, "timed_region_s": el, "value": nnz * rounds / el,
                "estep_avg_launch_ms": estep_ms, "speedup_step_vs_f64": f64_ms_per_step / (el * 1e3 / rounds),
                "speedup_launch_vs_f64": f64_estep_ms / estep_ms,
                "bytes_per_launch_this_format": own, "achieved_GBps_this_format": own / (estep_ms * 1e-3) / 1e9,
                "frac_this_format": own / (estep_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "frac_by_the_f64_formula": alg_bytes / (estep_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "theta_max_rel_diff_vs_f64_after_%d_rounds" % K: float(np.max(np.abs(th - ref)[big] / ref[big])) if big.any() else 0.0,
                "theta_sum": ts, "traffic": traffic}
    except Exception as e:
        return {"error": str(e)}


def em_leg(capi, make_em_workload, config, K, W, kernel, sync, device, q32=False):
    """One extra single-GPU E-step measurement on another BASELINE config (same procedure as the headline).
    "C5@0.1" = configs[4] at a tenth of its reads (the full size takes minutes to generate with numpy)."""
    try:
        t0 = time.perf_counter()
        scale = 1.0
        if "@" in config:
            config, sc = config.split("@", 1)
            scale = float(sc)
        wl = make_em_workload(config, scale=scale)
        gen_s = time.perf_counter() - t0
        N1, nnz, M = len(wl["row_ptr"]) - 1, len(wl["sid"]), wl["M"]
        ctx = capi.EmContext(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], device=device)
        ctx.set_option("kernel", kernel)
        el, rounds, reps, estep_ms, ts = timed_rounds(ctx, wl, wl["N0"], K, W, sync, lambda: None)
        alg = 12 * nnz + 16 * N1 + 16 * (M + 1)
        ach = alg / (estep_ms * 1e-3) / 1e9
        out = {"workload": "%s%s: %d reads x %d transcripts, %d alignments" % (WORKLOADS.get(config, config), "" if scale == 1.0 else " at %g of its reads" % scale, N1, M, nnz),
               "ms_per_step": el * 1e3 / rounds, "timed_rounds": rounds, "timed_region_s": el, "value": nnz * rounds / el,
               "estep_avg_launch_ms": estep_ms, "algorithmic_bytes_per_launch": alg, "achieved_GBps": ach, "frac": ach / HBM_PEAK_GBPS,
               "theta_sum": ts, "generate_s": gen_s}
        if q32 and kernel in (0, 3):
            out["q32_value_planes"] = q32_leg(ctx, wl, wl["N0"], K, W, sync, alg, el * 1e3 / rounds, estep_ms)
        ctx.close()
        return out
    except Exception as e:
        return {"error": str(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--legs", default="C2,C2R", help="extra single-GPU E-step legs on other configs (comma list, '' for none; NAME@scale for a fraction of the reads, e.g. C5@0.1)")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--no-q32", action="store_true", help="skip the Q32 value-plane measurement beside the headline")
    ap.add_argument("--value-bits", type=int, default=64, choices=(64, 32),
                    help="32: the HEADLINE context itself streams Q32 value planes (profiling runs; the default line stays on the doubles)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gibbs", action="store_true")
    ap.add_argument("--no-ci", action="store_true")
    ap.add_argument("--no-stream", action="store_true", help="skip the device STREAM probe beside the roofline")
    ap.add_argument("--gibbs-sweeps", type=int, default=30)
    args = ap.parse_args()

    # stdout carries exactly one JSON line: libraries that print to the C-level stdout (RCCL's version banner) are sent to
    # stderr for the whole run, the line itself goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

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
    torch.cuda.set_device(local)
    comm = None
    # BENCH_FORCE_DIST=1: take the N > 1 code path with a single rank (process group, communicator id broadcast, RCCL
    # collectives issued although there is one rank) -- the only way to execute that path on a one-GPU box
    distributed = world > 1 or bool(os.environ.get("BENCH_FORCE_DIST"))
    if distributed and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ["RSEM_COMM_FORCE"] = "1"
    if distributed:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))  # "nccl" is RCCL on ROCm
    if rank == 0:
        build.build()
    if distributed:
        dist.barrier()
        # the product's own communicator (RCCL from C++, collectives on the EM stream); only its id goes through torch
        ids = [capi.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        comm = capi.Comm.create(local, rank, world, ids[0])
    sync = torch.cuda.synchronize
    barrier = dist.barrier if distributed else (lambda: None)

    t0 = time.perf_counter()
    wl = make_em_workload(args.config, shard=rank, scale=args.scale)
    N1, nnz, M = len(wl["row_ptr"]) - 1, len(wl["sid"]), wl["M"]
    log("[rank %d] workload %s: N1=%d nnz=%d M=%d (%.1f s)" % (rank, args.config, N1, nnz, M, time.perf_counter() - t0))
    t0 = time.perf_counter()
    ctx = capi.EmContext(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], device=local)
    ctx.set_option("kernel", args.kernel)
    if args.value_bits == 32:
        ctx.set_option("value_bits", 32)
    upload_s = time.perf_counter() - t0
    log("[rank %d] upload + device layout: %.2f s" % (rank, upload_s))
    alg_bytes = 12 * nnz + 16 * N1 + 16 * (M + 1)
    K, W = args.steps, args.warmup
    if comm is not None:
        ctx.set_comm(comm)
    N0g = float(wl["N0"] * world)  # every rank passes the GLOBAL N0 (rsem_em_set_comm)
    def agree(x):
        if not distributed:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=torch.device("cuda", local))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed, rounds, reps, estep_ms, theta_sum = timed_rounds(ctx, wl, N0g, K, W, sync, barrier, agree)
    total_nnz = nnz
    dist_info = None
    if distributed:
        dev = torch.device("cuda", local)
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tn = torch.tensor([float(nnz)], dtype=torch.float64, device=dev)
        dist.all_reduce(tn)
        total_nnz = int(tn.item())
        # per-rank E-step time and the cost of the per-round collective alone (same size, same communicator)
        es = torch.zeros(world, dtype=torch.float64, device=dev)
        es[rank] = estep_ms
        dist.all_reduce(es)
        buf = torch.zeros(M + 1 + 128, dtype=torch.float64, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(5):
            comm.allreduce(buf.data_ptr(), buf.numel(), st)
        sync()
        barrier()
        t1 = time.perf_counter()
        for _ in range(200):
            comm.allreduce(buf.data_ptr(), buf.numel(), st)
        sync()
        ar_ms = (time.perf_counter() - t1) / 200 * 1e3
        dist_info = {"rccl_ranks": comm.world, "estep_ms_per_rank": [float(x) for x in es.cpu()], "allreduce_ms": ar_ms,
                     "allreduce_doubles": int(buf.numel())}

    gibbs = None
    if not args.no_gibbs:
        # Gibbs on the same matrix, one context per GPU: data-augmentation sweeps (one chain fills the GPU) and the exact
        # (reference) chain, 8 chains advancing together; across GPUs the chains are independent (one reduce at the end)
        try:
            irp, isid, icp = to_gibbs_items(wl)
            grp = np.arange(1, M + 2, 50, dtype=np.int32)
            if grp[-1] != M + 1:
                grp = np.append(grp, M + 1).astype(np.int32)
            g = capi.GibbsContext(M, irp, isid, icp, np.zeros(M + 1, np.int32), None, 1.0, (M + 1) + wl["N0"] + N1, wl["N0"],
                                  np.full(M + 1, 1000.0), np.ones(M + 1), grp, device=local)
            if comm is not None:
                g.set_comm(comm)
            _, _, _, pp = g.run_chains(capi.GIBBS_PARALLEL, [1 + rank], args.gibbs_sweeps - 2, [2], 1, thin=1, want_vectors=False)
            n_exact = 8
            _, _, _, pe = g.run_chains(capi.GIBBS_EXACT, capi.gibbs_chain_seeds(7 + rank, n_exact), 1, [2] * n_exact, 1, want_vectors=False)
            g.close()
            b_g = 12 * (len(isid) - N1) + 16 * N1  # conprb + sid per alignment, noise conprb + row slot per read
            gibbs = {"items_per_chain": int(len(isid)), "gpus": world,
                     "parallel": {"mode": "data-augmentation sampler, 1 chain per GPU", "ms_per_sweep": pp.sweep_ms,
                                  "algorithmic_GBps_per_gpu": b_g / pp.sweep_ms / 1e6 if pp.sweep_ms > 0 else None,
                                  "sweeps_per_s_all_gpus": world * 1e3 / pp.sweep_ms if pp.sweep_ms > 0 else None,
                                  "items_per_s_all_gpus": world * len(isid) * 1e3 / pp.sweep_ms if pp.sweep_ms > 0 else None},
                     "exact": {"mode": "reference chain (bit-identical draws), %d chains per GPU advancing together" % n_exact,
                               "ms_per_round": pe.sweep_ms, "chains_per_gpu": n_exact,
                               "read_visits_per_s_all_gpus": world * n_exact * N1 * 1e3 / pe.sweep_ms if pe.sweep_ms > 0 else None,
                               "items_per_s_all_gpus": world * n_exact * len(isid) * 1e3 / pe.sweep_ms if pe.sweep_ms > 0 else None}}
        except Exception as e:  # the EM line must still be printed
            gibbs = {"error": str(e)}

    value_plane_bytes = ctx.info("value_plane_bytes")
    q32 = None
    if world == 1 and not distributed and not args.no_q32 and args.value_bits == 64 and args.kernel in (0, 3):
        q32_traffic = None  # the committed PMC measurement of this layout (profiles/pmc_traffic.json), as for the headline
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pm = json.load(f).get(args.config + "_q32")
            if pm and args.scale == 1.0:
                q32_traffic = pm["traffic_bytes_per_launch"]
        except Exception:
            pass
        q32 = q32_leg(ctx, wl, N0g, K, W, sync, alg_bytes, elapsed * 1e3 / rounds, estep_ms, q32_traffic)
    ctx.close()
    if rank == 0:
        achieved = alg_bytes / (estep_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None  # PMC passes cannot run inside this process: the committed measurement for this workload
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pm = json.load(f).get(args.config)
            if pm and args.scale == 1.0 and args.kernel in (0, 3):
                traffic, traffic_src = pm["traffic_bytes_per_launch"], pm.get("source")
        except Exception:
            pass
        stream = None
        if not args.no_stream:
            try:
                rd, cp = capi.stream_probe(local, 8 << 30, 5)
                stream = {"read_GBps": rd, "copy_GBps": cp, "how": "rsem_hip_stream_probe: 8 B/lane wave loads over 4 GiB (read) / 4 GiB -> 4 GiB (copy, "
                          "read + written bytes counted), best of 5 launches, HIP events, in this process right after the timed region"}
            except Exception as e:
                stream = {"error": str(e)}
        line = {
            "metric": "EM read-alignments/s (nnz x EM iterations per second), rsem-run-em theta-only rounds",
            "value": total_nnz * rounds / elapsed, "unit": "read-alignments/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed * 1e3 / rounds,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "em_iterations_per_s": rounds / elapsed, "timed_rounds": rounds, "timed_region_s": elapsed, "timed_repeats_of_steps": reps,
            "config": {"workload": "%s: EM matrix of %d reads x %d transcripts, %d alignments (%.2f/read) per GPU, frozen conprb "
                                   "(rounds >= 12)" % (WORKLOADS.get(args.config, args.config), N1, M, nnz, nnz / max(N1, 1)),
                       "synthetic_config": args.config, "kernel": args.kernel, "value_bits": args.value_bits,
                       "value_plane_bytes": value_plane_bytes,
                       "parallelism": "1 GPU" if world == 1 else "read-sharded x%d + RCCL all-reduce(M+1 f64)/round from C++ on the EM stream" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "k_estep_lane (E step)", "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_ms": estep_ms, "step_over_launch": elapsed * 1e3 / rounds / estep_ms,
                         # the same launch time against the bytes the kernel physically moved (PMC) and against what a plain
                         # streaming kernel reaches on this device (measured here), beside the 8 TB/s specification
                         "frac_of_traffic": (traffic / (estep_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                         "stream": stream,
                         "achieved_over_stream_read": (achieved / stream["read_GBps"]) if stream and stream.get("read_GBps") else None,
                         "traffic_rate_over_stream_read": (traffic / (estep_ms * 1e-3) / 1e9 / stream["read_GBps"])
                         if traffic and stream and stream.get("read_GBps") else None},
            "checks": {"theta_sum": theta_sum},
            "upload_and_layout_s": upload_s,
            "gibbs": gibbs,
        }
        if q32 is not None:
            line["q32_value_planes"] = q32
        if dist_info:
            line["distributed"] = dist_info
        if world == 1:
            legs = {}
            for cfg in [c for c in args.legs.split(",") if c and c != args.config]:
                legs[cfg] = em_leg(capi, make_em_workload, cfg, K, W, args.kernel, sync, local, q32=not args.no_q32 and cfg == "C2")
            if legs:
                line["other_configs"] = legs
            if not args.no_ci:
                line["credibility_intervals"] = ci_leg(capi, min(M, 50_000))
            try:  # end-to-end wall clock of the drop-in programs vs the reference on the same files: measured by tools/e2e_c3.sh
                with open(os.path.join(ROOT, "profiles", "e2e_wall_clock.json")) as f:
                    line["e2e_wall_clock"] = json.load(f)
            except Exception:
                pass
            if not args.no_cpu_baseline:  # reported baseline: rank 0 at N=1 only
                cb = None
                try:
                    cb = cpu_baseline_reference(args.config, N1)
                except Exception as e:
                    log("cpu_baseline_reference failed: %s" % e)
                if cb is None:
                    cb = cpu_baseline_port(wl)
                else:
                    line["cpu_baseline_port_1core"] = cpu_baseline_port(wl, budget_s=5.0)
                line["cpu_baseline"] = cb
                line["speedup_vs_cpu_baseline_rounds"] = line["value"] / cb["value"]
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if comm is not None:
        barrier()
        comm.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
