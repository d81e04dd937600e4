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
(rsem_em_set_comm; the communicator id travels through torch.distributed); the Gibbs leg deals independent chains to the
ranks and ends in ONE reduce (the split BASELINE.json's north_star names).  `python bench.py --gpus N` without a launcher
starts its own ranks (torch.distributed.run, 127.0.0.1).  Rank 0 prints ONE JSON line.
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
             "C5": "BASELINE configs[4] (multi-mapping stress: > 2^32 alignments in one context)",
             "C3X": "configs[2] with cross-gene multi-mappers: 10 % of the reads also hit 1-3 transcripts of another gene (between C3, where no read "
                    "leaves its gene, and C2R)",
             "C3X30": "configs[2] with 30 % cross-gene multi-mappers"}


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
BAM_FRAC = 0.02  # of the workload's reads, for the -b leg of the end-to-end comparison
CPU_SAMPLE = {"C2": dict(read_type=1, frac=0.1, M=50_000, iso="4-12"), "C3": dict(read_type=3, frac=0.05, M=200_000, iso="5-16"),
              "C5": dict(read_type=1, frac=0.01, M=500_000, iso="32-64"), "tiny": dict(read_type=1, frac=1.0, M=400, iso="4-12")}


def _cpu_quota_cores():
    """The CPU time the container's cgroup grants, in cores (cpu.max of cgroup v2, cfs quota of v1); None = no limit found.  The gpurun
    boxes show 256 hardware threads and grant 16 cores' worth of time (profiles/r06q_*): what `-p 64` of either program really gets."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def _theta_line(path):
    with open(path) as f:
        return np.array(f.read().split("\n")[1].split(), float)


def one_socket_cores(want=64):
    """`want` cpus of ONE package, one hardware thread per physical core (sysfs), or None if the host has no such set or no
    `taskset`.  The reference's rounds >= 12 are bound by memory bandwidth and thread hand-offs: -p 64 pinned like this was
    its fastest setting on the GPU box (11.8 ms per round at 5 % of configs[2]; unpinned 13.2; -p 128 pinned / unpinned
    15.6 / 15.2: profiles/r04p_ref_threads_probe.json, tools/ref_threads_probe.py; -p 256: 25.4, round 2)."""
    import glob
    import shutil
    if not shutil.which("taskset"):
        return None
    try:
        by_pkg = {}
        for d in glob.glob("/sys/devices/system/cpu/cpu[0-9]*"):
            cpu = int(os.path.basename(d)[3:])
            with open(d + "/topology/physical_package_id") as f:
                pkg = int(f.read())
            with open(d + "/topology/thread_siblings_list") as f:
                first = int(f.read().strip().replace("-", ",").split(",")[0])
            if first == cpu:
                by_pkg.setdefault(pkg, []).append(cpu)
        allowed = os.sched_getaffinity(0)
        for pkg in sorted(by_pkg):
            cpus = sorted(c for c in by_pkg[pkg] if c in allowed)
            if len(cpus) >= want:
                return cpus[:want]
    except (OSError, ValueError):
        pass
    return None


GIBBS_E2E = dict(burnin=20, nsamples=40, gap=1, threads=8)  # per chain: 20 + 5 rounds (the pipeline's 200 / 1000 / 1 would be 325)


def gibbs_reference_leg(root, rt, n1, pin_cpus, limit_s=150.0):
    """rsem-run-gibbs, the UNMODIFIED reference binary (oracle/_ref, its pthread chains: Gibbs.cpp:207-254) and the drop-in
    (--gibbs-mode exact: the same chains on the GPU), on the SAME imdName.ofg -- written by the drop-in's rsem-run-em --gibbs-out
    in `root` (the 5 % input of the EM comparison) --, same -p 8 --seed 1 and chain parameters; wall clock of the whole programs,
    and the count-vector files compared byte for byte (the checker: the chains must be the reference's)."""
    import filecmp
    import shutil
    import subprocess
    ref_g = os.path.join(ROOT, "oracle", "_ref", "rsem-run-gibbs")
    new_g = os.path.join(ROOT, "rsem_amd", "bin", "rsem-run-gibbs")
    new_em = os.path.join(ROOT, "rsem_amd", "bin", "rsem-run-em")
    if not all(os.path.exists(p) for p in (ref_g, new_g, new_em)):
        return None
    g = GIBBS_E2E
    P = g["threads"]
    ref, imd, stat = os.path.join(root, "ref"), os.path.join(root, "temp", "s"), os.path.join(root, "stat", "s")
    t0 = time.perf_counter()
    r = subprocess.run([new_em, ref, str(rt), os.path.join(root, "s"), imd, stat, "-p", "64", "--gibbs-out", "-q"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0 or not os.path.exists(imd + ".ofg"):
        return {"error": "rsem-run-em --gibbs-out failed: " + r.stdout[-200:]}
    ofg_s = time.perf_counter() - t0
    tref = os.path.join(root, "tref")
    os.makedirs(tref, exist_ok=True)
    os.symlink(imd + ".ofg", os.path.join(tref, "s.ofg"))
    for f in ("omit", "iso_res", "gene_res"):
        if os.path.exists(imd + "." + f):
            shutil.copy(imd + "." + f, os.path.join(tref, "s." + f))
    open(os.path.join(tref, "s.omit"), "a").close()
    args = [str(g["burnin"]), str(g["nsamples"]), str(g["gap"]), "-p", str(P), "--seed", "1"]
    pin = ["taskset", "-c", ",".join(map(str, pin_cpus[:P]))] if pin_cpus and len(pin_cpus) >= P else []
    t0 = time.perf_counter()
    try:
        rr = subprocess.run(pin + [ref_g, ref, os.path.join(tref, "s"), stat] + args + ["-q"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=limit_s)
    except subprocess.TimeoutExpired:
        return {"error": "the reference's rsem-run-gibbs did not finish in %.0f s" % limit_s}
    ref_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    rn = subprocess.run([new_g, ref, imd, stat] + args + ["--gibbs-mode", "exact"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    new_s = time.perf_counter() - t0
    if rr.returncode != 0 or rn.returncode != 0:
        return {"error": "rsem-run-gibbs failed: reference rc %d, drop-in rc %d %s" % (rr.returncode, rn.returncode, rn.stdout[-200:])}
    same = [os.path.exists(imd + ".countvectors%d" % k) and filecmp.cmp(imd + ".countvectors%d" % k, os.path.join(tref, "s.countvectors%d" % k), shallow=False)
            for k in range(P)]
    rounds = g["burnin"] + 1 + (g["nsamples"] // P - 1) * g["gap"]
    ci = None
    try:  # rsem-calculate-credibility-intervals on those count vectors: the reference (calcCI.cpp:216-284) beside the drop-in
        ref_ci = os.path.join(ROOT, "oracle", "_ref", "rsem-calculate-credibility-intervals")
        new_ci = os.path.join(ROOT, "rsem_amd", "bin", "rsem-calculate-credibility-intervals")
        if all(same) and os.path.exists(ref_ci) and os.path.exists(new_ci):
            ci_args = ["0.95", str(g["nsamples"]), "50", "1024", "-p", str(P), "--seed", "7"]
            keep = {f: open(imd + "." + f).read() for f in ("iso_res", "gene_res")}  # (the programs append their rows to these files)

            def restore(prefix):
                for f, text in keep.items():
                    open(prefix + "." + f, "w").write(text)

            def ci_rows(prefix):
                return [l for l in open(prefix + ".iso_res").read().strip().split("\n")[-6:]]
            restore(os.path.join(tref, "s"))
            t0 = time.perf_counter()
            rc_r = subprocess.run(pin + [ref_ci, ref, os.path.join(tref, "s"), stat] + ci_args + ["-q"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=limit_s).returncode
            ci_ref_s = time.perf_counter() - t0
            rows_ref = ci_rows(os.path.join(tref, "s"))
            restore(imd)
            t0 = time.perf_counter()
            rc_n = subprocess.run([new_ci, ref, imd, stat] + ci_args + ["-q"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode
            ci_new_s = time.perf_counter() - t0
            restore(imd)
            t0 = time.perf_counter()
            rc_s = subprocess.run([new_ci, ref, imd, stat] + ci_args + ["-q", "--ci-stream", "reference"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode
            ci_stream_s = time.perf_counter() - t0
            ci = {"what": "rsem-calculate-credibility-intervals 0.95 %d 50 on those count vectors: reference -p %d; drop-in (samples drawn on the GPU); "
                          "drop-in --ci-stream reference (the reference's own draws restated on %d host threads, intervals on the GPU)" % (g["nsamples"], P, P),
                  "reference_s": ci_ref_s, "dropin_s": ci_new_s, "speedup": ci_ref_s / ci_new_s if ci_new_s > 0 else None,
                  "dropin_reference_stream_s": ci_stream_s, "rows_equal_with_reference_stream": bool(rc_r == 0 and rc_s == 0 and ci_rows(imd) == rows_ref),
                  "rc": [rc_r, rc_n, rc_s]}
    except Exception as e:
        ci = {"error": str(e)[:200]}
    return {"what": "rsem-run-gibbs %s on the same imdName.ofg (%d reads, %.2f GB of text): oracle/_ref (pthread chains%s) and the drop-in (--gibbs-mode exact), whole programs"
                    % (" ".join(args), n1, os.path.getsize(imd + ".ofg") / 1e9, ", pinned to %d cores of one socket" % P if pin else ""),
            "kind": "reference", "cores": P, "reads": n1, "chains": P, "rounds_per_chain": rounds,
            "value": P * rounds * n1 / ref_s, "unit": "read visits/s (all chains, whole program incl. reading .ofg)",
            "reference_s": ref_s, "dropin_s": new_s, "speedup": ref_s / new_s, "dropin_read_visits_per_s": P * rounds * n1 / new_s,
            "count_vector_files": P, "count_vectors_identical": bool(all(same)), "write_ofg_s": ofg_s, "credibility_intervals": ci,
            "ci_reference_s": (ci or {}).get("reference_s"), "ci_dropin_s": (ci or {}).get("dropin_s"),
            "ci_rows_equal_with_reference_stream": (ci or {}).get("rows_equal_with_reference_stream")}


def reference_e2e(config, n_full, ref_limit_s=260.0, full_size=True, bam_leg=True, gibbs_leg=True):
    """The UNMODIFIED reference binary (oracle/_ref/rsem-run-em, built from /root/reference) and the drop-in
    (rsem_amd/bin/rsem-run-em) on the SAME generated .temp files of the bench workload's shape (tools/gen_temp.cpp: model
    type, transcripts, isoforms per gene; `frac` of its reads), each run to convergence, wall clock of the whole program.
    -> (cpu_baseline, e2e).  cpu_baseline = the reference's rate in the rounds the GPU line times (ROUND >= 12, frozen
    alignment probabilities; from the arrival times of its 'ROUND =' lines, EM.cpp:415): the E step is O(alignments)
    (EM.cpp:199-236), so the rate per alignment carries over to the full size.  e2e.measured = both programs' wall clock at
    that size, ROUND counts and theta compared; e2e.full_size = the drop-in alone on the full-size files (50 M read pairs,
    30 GB of text) and the reference's time EXTRAPOLATED to them (every phase of EM.cpp is linear in reads / alignments;
    the ROUND count is the drop-in's: the two programs stop at the same ROUND on the same input, checked at the small size)."""
    import shutil
    import subprocess
    import tempfile
    gen = os.path.join(ROOT, "tools", "bin", "gen_temp")
    ref_em = os.path.join(ROOT, "oracle", "_ref", "rsem-run-em")
    ref_idx = os.path.join(ROOT, "oracle", "_ref", "rsem-build-read-index")
    new_em = os.path.join(ROOT, "rsem_amd", "bin", "rsem-run-em")
    if not all(os.path.exists(p) for p in (gen, ref_em, ref_idx, new_em)):
        return None, None
    cs = CPU_SAMPLE[config]
    rt = cs["read_type"]
    ncpu = os.cpu_count() or 1
    cores = min(64, ncpu)  # -p 64 beat -p 128 and -p 256 in every run (one_socket_cores)
    pinned = one_socket_cores(cores) if cores == 64 else None
    pin = ["taskset", "-c", ",".join(map(str, pinned))] if pinned else []
    d = tempfile.mkdtemp(prefix="rsem_bench_", dir="/tmp")

    def generate(root, n_reads, sam=False):
        t0 = time.perf_counter()
        out = subprocess.run([gen, root, str(n_reads), str(cs["M"]), str(rt), "20250925", "100", "sam" if sam else "nosam", cs["iso"]],
                             stdout=subprocess.PIPE, text=True, check=True).stdout
        reads = ["s_alignable.fq"] if rt == 1 else ["s_alignable_1.fq", "s_alignable_2.fq"]
        subprocess.run([ref_idx, "32", "1", "1"] + [os.path.join(root, "temp", r) for r in reads], stdout=subprocess.DEVNULL, check=True)
        return int(out.split("nHits=")[1].split()[0]), int(out.split("N1=")[1].split()[0]), time.perf_counter() - t0

    def em_args(root):
        return [os.path.join(root, "ref"), str(rt), os.path.join(root, "s"), os.path.join(root, "temp", "s"), os.path.join(root, "stat", "s"), "-p", str(cores)]

    def run_dropin(root):
        t0 = time.perf_counter()
        r = subprocess.run([new_em] + em_args(root), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        wall = time.perf_counter() - t0
        rounds = [int(l.split(",")[0].split("=")[1]) for l in r.stdout.split("\n") if l.startswith("ROUND =")]
        if r.returncode != 0 or not rounds:
            raise RuntimeError("drop-in rsem-run-em failed: " + r.stdout[-500:])
        return wall, rounds[-1], _theta_line(os.path.join(root, "stat", "s.theta"))

    try:
        small = os.path.join(d, "small")
        n_reads = max(100_000, int(n_full * cs["frac"] / 0.95))  # gen_temp: 95 % of the reads are alignable
        nhits, n1, gen_s = generate(small, n_reads)
        # --- the reference, to convergence (or to the limit: then only its per-round rate is known)
        t0 = time.perf_counter()
        p = subprocess.Popen(pin + [ref_em] + em_args(small), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        marks = []  # (ROUND, arrival time)
        finished = False
        for line in p.stdout:
            now = time.perf_counter()
            if line.startswith("ROUND ="):
                marks.append((int(line.split(",")[0].split("=")[1]), now))
            if now - t0 > ref_limit_s:
                break
        else:
            finished = True
        if not finished:
            p.kill()
        p.wait()
        ref_wall = time.perf_counter() - t0
        finished = finished and p.returncode == 0
        late = [m for m in marks if m[0] >= 12]
        if len(late) < 3:
            return None, None
        per_round = (late[-1][1] - late[0][1]) / (late[-1][0] - late[0][0])
        startup_s = marks[0][1] - t0
        early_s = late[0][1] - marks[0][1]
        cpu = {"value": nhits / per_round, "unit": "read-alignments/s", "cores": cores, "kind": "reference",
               "sample": "oracle/_ref/rsem-run-em -p %d%s on a generated %s input of the bench workload's shape at %.0f %% of its reads: %d alignable "
                         "reads, %d alignments (%.2f/read), %d transcripts; rounds >= 12 timed from its ROUND lines (%d rounds, %.2f ms/round); rate "
                         "per alignment, so it carries over to the full size (E step is O(alignments))"
                         % (cores, " pinned to the physical cores of one socket (its fastest setting found: profiles/r04p_ref_threads_probe.json)" if pinned else "", {1: "SingleQModel", 3: "PairedEndQModel"}[rt], cs["frac"] * 100, n1, nhits, nhits / n1, cs["M"],
                            late[-1][0] - late[0][0], per_round * 1e3),
               "sample_short": "oracle/_ref/rsem-run-em -p %d%s, generated %s input at %.0f %% of the workload's reads (%d alignments), rounds >= 12 from its ROUND lines"
                               % (cores, " pinned to one socket" if pinned else "", {1: "SingleQModel", 3: "PairedEndQModel"}[rt], cs["frac"] * 100, nhits),
               "ms_per_round": per_round * 1e3, "rounds_timed": late[-1][0] - late[0][0], "startup_s": startup_s,
               "host_cores_available": ncpu, "cpu_quota_cores": _cpu_quota_cores(), "pinned_cpus": ",".join(map(str, pinned)) if pinned else None, "generate_s": gen_s}
        e2e = {"what": "whole programs on the same files, wall clock: parse the .temp files, rounds 1-11 with the model, rounds >= 12 to convergence, "
                       "expected counts, results",
               "measured": {"size": "%d alignable reads, %d alignments, %d transcripts (%.0f %% of the bench workload's reads)" % (n1, nhits, cs["M"], cs["frac"] * 100),
                            "reference_s": ref_wall if finished else None, "reference_finished": finished, "reference_rounds": marks[-1][0],
                            "reference_breakdown_s": {"startup": startup_s, "rounds_1_11": early_s, "rounds_12_on": late[-1][1] - late[0][1]},
                            "reference_threads": cores}}
        if finished:
            ref_theta = _theta_line(os.path.join(small, "stat", "s.theta"))
            new_wall, new_rounds, new_theta = run_dropin(small)
            big = ref_theta >= 1e-7
            e2e["measured"].update({"dropin_s": new_wall, "dropin_rounds": new_rounds, "speedup": ref_wall / new_wall,
                                    "same_round_count": new_rounds == marks[-1][0],
                                    "theta_max_rel_diff": float(np.max(np.abs(new_theta - ref_theta)[big] / ref_theta[big])) if big.any() else 0.0})
        if gibbs_leg and finished:
            try:
                e2e["gibbs"] = gibbs_reference_leg(small, rt, n1, pinned)
            except Exception as e:
                e2e["gibbs"] = {"error": str(e)[:300]}
        shutil.rmtree(small, ignore_errors=True)
        if bam_leg and finished:
            # -b is ON by default in rsem-calculate-expression (:61,626-632): the same comparison with the transcript.bam pass
            # (BamWriter.h: every input alignment copied with MAPQ and ZW:f set from its posterior weight), at BAM_FRAC of the
            # workload's reads -- the SAM text alone is 4.5 GB there.  Both programs get the same -p (the reference hands it to
            # htslib's compression threads, the drop-in to the threads of host/bam_io.hpp).
            try:
                bamd = os.path.join(d, "bam")
                bh, b1, bgen_s = generate(bamd, max(100_000, int(n_full * BAM_FRAC / 0.95)), sam=True)
                # BAM input, what aligners hand over (the pipeline gives rsem-run-em the BAM its parser read): the records of the
                # generated SAM text as BAM = the drop-in's own transcript.bam of a first, untimed run (both programs overwrite
                # MAPQ and ZW:f of every record)
                sam_gb = os.path.getsize(os.path.join(bamd, "aln.sam")) / 1e9
                rc0 = subprocess.run([new_em] + em_args(bamd) + ["-b", os.path.join(bamd, "aln.sam"), "0", "-q"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                if rc0.returncode != 0:
                    raise RuntimeError("the conversion run failed: rc %d" % rc0.returncode)
                os.replace(os.path.join(bamd, "s.transcript.bam"), os.path.join(bamd, "aln.bam"))
                os.remove(os.path.join(bamd, "aln.sam"))
                bargs = em_args(bamd) + ["-b", os.path.join(bamd, "aln.bam"), "0", "-q"]
                t0 = time.perf_counter()
                rr = subprocess.run(pin + [ref_em] + bargs, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=ref_limit_s)
                ref_b = time.perf_counter() - t0
                ref_size = os.path.getsize(os.path.join(bamd, "s.transcript.bam")) if rr.returncode == 0 else 0
                ref_theta_b = _theta_line(os.path.join(bamd, "stat", "s.theta"))
                t0 = time.perf_counter()
                rn = subprocess.run([new_em] + bargs, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                new_b = time.perf_counter() - t0
                if rr.returncode != 0 or rn.returncode != 0:
                    raise RuntimeError("a -b run failed: reference rc %d, drop-in rc %d %s" % (rr.returncode, rn.returncode, rn.stdout[-300:]))
                new_theta_b = _theta_line(os.path.join(bamd, "stat", "s.theta"))
                bigb = ref_theta_b >= 1e-7
                e2e["bam_on"] = {"size": "%d alignable reads, %d alignments, %d transcripts (%.0f %% of the bench workload's reads), BAM input of %.2f GB (%.2f GB as SAM text)"
                                         % (b1, bh, cs["M"], BAM_FRAC * 100, os.path.getsize(os.path.join(bamd, "aln.bam")) / 1e9, sam_gb),
                                 "input": "BAM",
                                 "what": "rsem-run-em ... -p %d -b aln.bam 0: the EM and the transcript.bam pass (the pipeline's default), wall clock" % cores,
                                 "reference_s": ref_b, "dropin_s": new_b, "speedup": ref_b / new_b, "generate_s": bgen_s,
                                 "transcript_bam_bytes": {"reference": ref_size, "dropin": os.path.getsize(os.path.join(bamd, "s.transcript.bam"))},
                                 "theta_max_rel_diff": float(np.max(np.abs(new_theta_b - ref_theta_b)[bigb] / ref_theta_b[bigb])) if bigb.any() else 0.0,
                                 "records_compared_in": "tests/test_cli_gpu.py::test_transcript_bam_matches_reference, tests/test_bam_cpu.py"}
            except Exception as e:
                e2e["bam_on"] = {"error": str(e)[:300]}
            shutil.rmtree(os.path.join(d, "bam"), ignore_errors=True)
        if full_size:
            full = os.path.join(d, "full")
            fh, f1, fgen_s = generate(full, int(n_full / 0.95))
            new_wall, new_rounds, new_theta = run_dropin(full)
            scale = fh / nhits
            ref_full = scale * (startup_s + early_s) + (new_rounds - 11) * scale * per_round
            e2e["full_size"] = {"size": "%d alignable reads, %d alignments, %d transcripts" % (f1, fh, cs["M"]), "generate_s": fgen_s,
                                "dropin_s": new_wall, "dropin_rounds": new_rounds, "theta_sum": float(new_theta.sum()),
                                "reference_s_extrapolated": ref_full,
                                "extrapolation": "reference(full) = %.2f x (startup + rounds 1-11 measured above) + (%d - 11) rounds x %.2f x %.2f ms"
                                                 % (scale, new_rounds, scale, per_round * 1e3),
                                "speedup_extrapolated": ref_full / new_wall}
            try:  # the reference itself on these very files (same generator, same seed), measured once in round 4: profiles/
                with open(os.path.join(ROOT, "profiles", "e2e_full_size_reference.json")) as f:
                    mref = json.load(f)
                if config == "C3" and ("%d alignments" % fh).replace(",", "") in mref["workload"].replace(",", ""):  # the very files of the recording
                    r = mref["reference"]
                    e2e["full_size"].update({
                        "reference_s": r["wall_s_measured"], "reference_rounds": r["rounds"], "reference_measured_in": mref["source"],
                        "reference_conditions": r["conditions"],
                        "reference_s_undisturbed": r["wall_s_if_all_late_rounds_at_the_undisturbed_rate"],
                        "same_round_count_as_the_reference": new_rounds == r["rounds"],
                        "speedup_vs_recorded_reference": r["wall_s_measured"] / new_wall,
                        "speedup_vs_recorded_reference_undisturbed": r["wall_s_if_all_late_rounds_at_the_undisturbed_rate"] / new_wall,
                        "reference_s_alone_pinned": (mref.get("reference_alone_pinned_round5") or {}).get("wall_s_measured"),
                        "speedup_vs_recorded_reference_alone_pinned": ((mref.get("reference_alone_pinned_round5") or {}).get("wall_s_measured") or 0.0) / new_wall or None,
                        "recorded_reference_note": "the reference's time is a RECORDING (round 4, another session and host state: reference_measured_in); "
                                                   "only `measured.speedup` above divides two times of this run",
                        "parity_full_size_recorded": mref["parity_full_size"]})
            except Exception:
                pass
        return cpu, e2e
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
        phys = physical(ctx, estep_ms, traffic)
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
                "theta_sum": ts, "traffic": traffic, "physical": phys, "frac_physical": phys.get("frac_physical")}
    except Exception as e:
        return {"error": str(e)}


def physical(ctx, estep_ms, pmc_traffic=None):
    """The bytes one E-step launch moves through HBM BY CONSTRUCTION of the layout this context holds right now
    (rsem_em_get_info "physical_bytes_per_launch": every value plane incl. padding, the sid planes of the slices in which a
    tuple starts, noise value + mask + unit records, theta into / counts out of every LDS window) against the launch time
    measured in this run -- the roofline fraction this run can vouch for itself.  pmc_traffic: the committed rocprofv3
    counter measurement of the same workload, when there is one (profiles/pmc_traffic.json), as a cross-check of the
    accounting."""
    try:
        b = ctx.info("physical_bytes_per_launch")
        out = {"physical_bytes_per_launch": b, "physical_GBps": b / (estep_ms * 1e-3) / 1e9,
               "frac_physical": b / (estep_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
               "parts": {"value_planes": ctx.info("value_plane_bytes"), "sid_planes_loaded": ctx.info("sid_plane_bytes_loaded"),
                         "sid_planes_stored": ctx.info("sid_plane_bytes"), "noise_values": 8 * ctx.info("slots"),
                         "masks": 8 * ctx.info("slices"), "lds_windows_theta_in_counts_out": 16 * ctx.info("window_entries")}}
        if pmc_traffic:
            out["pmc_traffic_bytes_per_launch"] = pmc_traffic
            out["physical_over_pmc"] = b / pmc_traffic
        return out
    except Exception as e:  # (a library built before the keys existed: RSEM_HIP_LIB experiments)
        return {"error": str(e)}


def pmc_traffic_of(key, scale=1.0, kernel=0):
    """(bytes per launch, source) of the committed PMC measurement for a workload key of profiles/pmc_traffic.json."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pm = json.load(f).get(key)
        if pm and scale == 1.0 and kernel in (0, 3):
            return pm["traffic_bytes_per_launch"], pm.get("source")
    except Exception:
        pass
    return None, None


def split_info(ctx):
    """Reads laid out as an in-window row plus far entries (sell_layout.hpp split rows): counts, for the leg's record."""
    try:
        return {"reads": ctx.info("split_rows"), "far_entries": ctx.info("far_entries")}
    except Exception:
        return None


def far_units(ctx):
    """Units of the hot-loop layout with an id outside their LDS window (they run the loop with the global gather / atomics)."""
    try:
        return "%d of %d" % (ctx.info("far_units"), ctx.info("units"))
    except Exception:  # (a library built before the key existed: RSEM_HIP_LIB experiments)
        return None


def one_step_parity(ctx, wl):
    """One E + M step of the context against the CPU restatement (oracle/, EM.cpp:199-236,391-398) on the WHOLE matrix --
    the checker beside the measurement, not part of it (for configs[4] this is also the > 2^32-alignments index check:
    the oracle walks 4 G alignments in a few seconds)."""
    try:
        from oracle import pyoracle as orc
        t0 = time.perf_counter()
        counts, *_ = ctx.step(wl["theta0"], wl["N0"])
        oc = orc.em_estep(wl["M"], wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], wl["theta0"])
        oc[0] += wl["N0"]
        denom = np.maximum(np.abs(oc), 1e-6)
        err = float(np.max(np.abs(counts - oc) / denom))
        return {"max_rel_diff_counts_vs_oracle": err, "ok": bool(err < 1e-9), "tolerance": 1e-9, "alignments_checked": int(len(wl["sid"])),
                "seconds": time.perf_counter() - t0}
    except Exception as e:
        return {"error": str(e)}


def em_leg(capi, make_em_workload, config, K, W, kernel, sync, device, q32=False):
    """One extra single-GPU E-step measurement on another BASELINE config (same procedure as the headline).
    "C5@0.1" = configs[4] at a tenth of its reads."""
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
        phys = physical(ctx, estep_ms, pmc_traffic_of(config, scale, kernel)[0])
        out = {"workload": "%s%s: %d reads x %d transcripts, %d alignments" % (WORKLOADS.get(config, config), "" if scale == 1.0 else " at %g of its reads" % scale, N1, M, nnz),
               "ms_per_step": el * 1e3 / rounds, "timed_rounds": rounds, "timed_region_s": el, "value": nnz * rounds / el,
               "estep_avg_launch_ms": estep_ms, "algorithmic_bytes_per_launch": alg, "achieved_algorithmic_GBps": ach, "frac_algorithmic": ach / HBM_PEAK_GBPS,
               "frac": phys.get("frac_physical"), "frac_physical": phys.get("frac_physical"), "physical": phys,
               "theta_sum": ts, "generate_s": gen_s, "parity_one_step": one_step_parity(ctx, wl),
               "units_with_ids_outside_their_window": far_units(ctx), "split_rows": split_info(ctx)}
        if q32 and kernel in (0, 3):
            out["q32_value_planes"] = q32_leg(ctx, wl, wl["N0"], K, W, sync, alg, el * 1e3 / rounds, estep_ms)
        ctx.close()
        return out
    except Exception as e:
        return {"error": str(e)}


DETAIL_NAME = "bench_detail_latest.json"
LINE_LIMIT = 6000  # bytes; the driver keeps the last 8 187 bytes of stdout: the contract line must fit in them whole


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _num(x, digits=6):
    """Shorter floats for the contract line (the side file keeps every digit)."""
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _num(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_num(v, digits) for v in x]
    return x


def _leg_summary(leg):
    if not isinstance(leg, dict) or "error" in leg:
        return {"error": str((leg or {}).get("error"))[:120]}
    par = leg.get("parity_one_step") or {}
    out = {"ms": leg.get("estep_avg_launch_ms"), "ms_per_step": leg.get("ms_per_step"), "frac": leg.get("frac"),
           "frac_algorithmic": leg.get("frac_algorithmic"), "parity_one_step": par.get("max_rel_diff_counts_vs_oracle"), "parity_ok": par.get("ok")}
    q = leg.get("q32_value_planes")
    if isinstance(q, dict) and "error" not in q:
        out["q32_ms"] = q.get("estep_avg_launch_ms")
    return out


def _gibbs_summary(g):
    if not isinstance(g, dict) or "error" in g:
        return {"error": str((g or {}).get("error"))[:160]}
    out = {"items_per_chain": g.get("items_per_chain"), "gpus": g.get("gpus")}
    p, e = g.get("parallel") or {}, g.get("exact") or {}
    many = ("ms_per_sweep_per_rank", "sweeps_per_s_per_rank", "frac_of_hbm_peak_per_rank", "ms_per_round_per_rank") if (g.get("gpus") or 1) > 1 else ()
    out["parallel"] = _pick(p, ("ms_per_sweep", "sweeps_per_s_all_gpus", "frac_of_hbm_peak_per_gpu", "final_reduce_ms") + many)
    out["exact"] = _pick(e, ("ms_per_round", "chains_per_gpu", "workgroups_per_chain", "us_per_read_visit_and_chain", "read_visits_per_s_all_gpus", "final_reduce_ms") + many)
    for k in ("exact_strong", "cpu_baseline"):
        if isinstance(g.get(k), dict):
            out[k] = {kk: vv for kk, vv in g[k].items() if not isinstance(vv, (dict, list)) and not (isinstance(vv, str) and len(vv) > 100)}
    return out


def contract_line(detail, detail_path):
    """The ONE line of stdout: the contract's keys and one-number summaries of the legs, well under LINE_LIMIT bytes.  Everything
    else (prose, conditions, recordings, per-part byte tables, per-leg records) is `detail`, written to `detail_path` by the
    same run."""
    line = _pick(detail, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data",
                          "em_iterations_per_s", "timed_rounds", "timed_region_s"))
    line["vs_baseline"] = detail.get("vs_baseline")
    line["config"] = _pick(detail["config"], ("workload", "synthetic_config", "kernel", "value_bits", "parallelism"))
    r = detail["roofline"]
    line["roofline"] = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "kernel", "avg_launch_ms", "algorithmic_bytes_per_launch",
                                 "achieved_algorithmic", "frac_algorithmic", "step_over_launch"))
    line["roofline"]["traffic"] = r.get("traffic")
    line["roofline"]["physical_bytes_per_launch"] = (r.get("physical") or {}).get("physical_bytes_per_launch")
    if isinstance(r.get("stream"), dict):
        line["roofline"]["stream_read_GBps"] = r["stream"].get("read_GBps")
    cb = detail.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "ms_per_round", "rounds_timed", "host_cores_available", "cpu_quota_cores"))
        line["cpu_baseline"]["sample"] = cb.get("sample_short") or str(cb.get("sample"))[:200]
        line["speedup_vs_cpu_baseline_rounds"] = detail.get("speedup_vs_cpu_baseline_rounds")
    par = (detail.get("checks") or {}).get("parity_one_step")
    line["checks"] = {"theta_sum": (detail.get("checks") or {}).get("theta_sum"),
                      "parity_one_step": _pick(par, ("max_rel_diff_counts_vs_oracle", "ok", "tolerance", "alignments_checked", "error")) if par else None}
    e = detail.get("e2e_wall_clock")
    if isinstance(e, dict):
        o = {}
        if isinstance(e.get("measured"), dict):
            o["measured"] = _pick(e["measured"], ("reference_s", "dropin_s", "speedup", "same_round_count", "dropin_rounds", "theta_max_rel_diff", "reference_threads", "frac_of_workload"))
        if isinstance(e.get("bam_on"), dict):
            o["bam_on"] = _pick(e["bam_on"], ("reference_s", "dropin_s", "speedup", "theta_max_rel_diff", "frac_of_workload", "input", "bam_pass_reference_s", "bam_pass_dropin_s", "error"))
        if isinstance(e.get("full_size"), dict):
            o["full_size"] = _pick(e["full_size"], ("dropin_s", "dropin_rounds", "same_round_count_as_the_reference", "reference_s_extrapolated", "speedup_extrapolated",
                                                   "reference_s", "speedup_vs_recorded_reference", "theta_max_rel_diff_vs_recorded_reference"))
        line["e2e"] = o
    if isinstance(detail.get("other_configs"), dict):
        line["legs"] = {k: _leg_summary(v) for k, v in detail["other_configs"].items()}
    q = detail.get("q32_value_planes")
    if isinstance(q, dict):
        line["q32"] = _pick(q, ("estep_avg_launch_ms", "ms_per_step", "frac_physical", "reads_q32_fraction", "error"))
    if detail.get("gibbs") is not None:
        line["gibbs"] = _gibbs_summary(detail["gibbs"])
    ci = detail.get("credibility_intervals")
    if isinstance(ci, dict):
        line["ci"] = {k: v for k, v in ci.items() if not isinstance(v, (dict, list)) and not (isinstance(v, str) and len(v) > 100)}
    if isinstance(detail.get("distributed"), dict):
        line["distributed"] = detail["distributed"]
    line["upload_and_layout_s"] = detail.get("upload_and_layout_s")
    line["detail"] = detail_path
    line = _num(line)
    s = json.dumps(line, separators=(",", ":"))
    if len(s) > LINE_LIMIT:  # never again a line the driver cannot parse: drop the summaries, largest first, keep the contract
        for k in sorted(("legs", "gibbs", "ci", "e2e", "q32", "distributed"), key=lambda k: -len(json.dumps(line.get(k)))):
            if k in line:
                line[k] = {"see": "detail"}
                s = json.dumps(line, separators=(",", ":"))
                if len(s) <= LINE_LIMIT:
                    break
    return s


def write_detail(detail):
    """profiles/bench_detail_latest.json (BENCH_DETAIL_PATH overrides: tests), and a copy under gpurun_out/ where that exists."""
    path = os.environ.get("BENCH_DETAIL_PATH") or os.path.join(ROOT, "profiles", DETAIL_NAME)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(detail, f, indent=1)
        scratch = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(scratch) and not os.environ.get("BENCH_DETAIL_PATH"):
            with open(os.path.join(scratch, DETAIL_NAME), "w") as f:
                json.dump(detail, f, indent=1)
    except OSError as e:
        log("bench detail not written: %s" % e)
        return None
    return os.path.relpath(path, ROOT) if path.startswith(ROOT) else path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--legs", default="C2,C2R,C3X,C3X30,C5", help="extra single-GPU E-step legs on other configs (comma list, '' for none; NAME@scale for a fraction of the reads, e.g. C5@0.1)")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--no-q32", action="store_true", help="skip the Q32 value-plane measurement beside the headline")
    ap.add_argument("--value-bits", type=int, default=64, choices=(64, 32),
                    help="32: the HEADLINE context itself streams Q32 value planes (profiling runs; the default line stays on the doubles)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the reference binary (CPU baseline + whole-program wall clock)")
    ap.add_argument("--no-e2e-full", action="store_true", help="skip the drop-in's full-size whole-program run (30 GB of generated text inputs)")
    ap.add_argument("--no-gibbs", action="store_true")
    ap.add_argument("--no-ci", action="store_true")
    ap.add_argument("--no-stream", action="store_true", help="skip the device STREAM probe beside the roofline")
    ap.add_argument("--gibbs-sweeps", type=int, default=30)
    ap.add_argument("--no-bam-leg", action="store_true", help="skip the -b (transcript.bam) leg of the end-to-end comparison")
    ap.add_argument("--gibbs-exact-rounds", type=int, default=6, help="rounds of the exact (reference) chain timed in the Gibbs leg (>= 2)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # called without a launcher: start one rank per GPU ourselves (the contract's launch line)
        port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 2000))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        if os.environ.get("BENCH_PRINT_LAUNCH"):  # (tests: show the launch line instead of running it)
            print(json.dumps(cmd))
            return
        os.execv(sys.executable, cmd)

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
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus N`, or under torch.distributed.run "
                         "with --nproc-per-node equal to --gpus)" % (args.gpus, world))
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
    # tensors of the timing collectives live on the GPU ("nccl" is RCCL on ROCm); BENCH_DIST_BACKEND=gloo (tests/, no GPU:
    # the C-ABI wrappers are stand-ins there) keeps them on the host so that the N > 1 control flow and the JSON line it
    # assembles run as world-size-2 CPU processes
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    tdev = torch.device("cuda", local) if backend == "nccl" else torch.device("cpu")
    if distributed:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=tdev)
        else:
            dist.init_process_group(backend)
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
        t = torch.tensor([x], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed, rounds, reps, estep_ms, theta_sum = timed_rounds(ctx, wl, N0g, K, W, sync, barrier, agree)
    total_nnz = nnz
    dist_info = None
    if distributed:
        dev = tdev
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tn = torch.tensor([float(nnz)], dtype=torch.float64, device=dev)
        dist.all_reduce(tn)
        total_nnz = int(tn.item())
        # per-rank E-step time and the cost of the per-round collective alone (same size, same communicator)
        es = torch.zeros(world, 2, dtype=torch.float64, device=dev)
        es[rank, 0] = estep_ms
        es[rank, 1] = physical(ctx, estep_ms).get("frac_physical") or 0.0   # every GPU's own layout against its own launch time
        dist.all_reduce(es)
        buf = torch.zeros(M + 1 + 128, dtype=torch.float64, device=dev)
        st = torch.cuda.current_stream().cuda_stream if backend == "nccl" else 0
        for _ in range(5):
            comm.allreduce(buf.data_ptr(), buf.numel(), st)
        sync()
        barrier()
        t1 = time.perf_counter()
        for _ in range(200):
            comm.allreduce(buf.data_ptr(), buf.numel(), st)
        sync()
        ar_ms = (time.perf_counter() - t1) / 200 * 1e3
        dist_info = {"rccl_ranks": comm.world, "estep_ms_per_rank": [float(x) for x in es[:, 0].cpu()],
                     "frac_physical_per_rank": [float(x) for x in es[:, 1].cpu()],
                     "em_iterations_per_s_per_rank": [1e3 / float(x) if x > 0 else None for x in es[:, 0].cpu()],
                     "allreduce_ms": ar_ms, "allreduce_doubles": int(buf.numel())}

    gibbs = None
    if not args.no_gibbs:
        # Gibbs on the same matrix, one context per GPU.  Chains are independent: global chain k runs on rank k % world
        # (rsem_amd.dist.gibbs_rank_chains = rsem-run-gibbs's deal), every rank's chain sums meet in ONE reduce to rank 0 at
        # the end of rsem_gibbs_run_chains (RCCL from C++ on the chain stream; release(), Gibbs.cpp:372-388).  Two samplers:
        # the data-augmentation sweeps (one chain fills a GPU) and the reference's own chain (8 chains per GPU advancing
        # together, one workgroup each).
        try:
            from rsem_amd.dist import gibbs_rank_chains
            irp, isid, icp = to_gibbs_items(wl)
            grp = np.arange(1, M + 2, 50, dtype=np.int32)
            if grp[-1] != M + 1:
                grp = np.append(grp, M + 1).astype(np.int32)
            g = capi.GibbsContext(M, irp, isid, icp, np.zeros(M + 1, np.int32), None, 1.0, (M + 1) + wl["N0"] + N1, wl["N0"],
                                  np.full(M + 1, 1000.0), np.ones(M + 1), grp, device=local)
            if comm is not None:
                g.set_comm(comm)
            par_seeds = capi.gibbs_chain_seeds(1, world)
            _, _, _, pp = g.run_chains(capi.GIBBS_PARALLEL, [par_seeds[k] for k in gibbs_rank_chains(world, world, rank)],
                                       args.gibbs_sweeps - 2, [2], 1, thin=1, want_vectors=False)
            n_exact = 8
            ex_seeds = capi.gibbs_chain_seeds(7, n_exact * world)
            _, acc_e, _, pe = g.run_chains(capi.GIBBS_EXACT, [ex_seeds[k] for k in gibbs_rank_chains(n_exact * world, world, rank)],
                                           args.gibbs_exact_rounds - 1, [2] * n_exact, 1, want_vectors=False)
            # the split north_star names: 8 chains IN ALL dealt to the GPUs (chain k on rank k % N: one chain per GPU at N = 8), a
            # team of min(64, CUs / chains of the GPU) workgroups per chain -- strong scaling of ONE rsem-run-gibbs -p 8 run.  At N = 1
            # that IS the leg above (8 chains x teams of 32); beside it one chain alone on the GPU (a team of 64) = what every GPU of
            # an 8-GPU run does, so the 1 -> 8 prediction can be read off a single-GPU line.
            strong_total = 8
            mine = list(gibbs_rank_chains(strong_total, world, rank))
            ps, p1 = pe, None
            if 1 < world <= strong_total:  # (every rank has a chain: all of them meet in the run's one reduce)
                st_seeds = capi.gibbs_chain_seeds(7, strong_total)
                _, _, _, ps = g.run_chains(capi.GIBBS_EXACT, [st_seeds[k] for k in mine], args.gibbs_exact_rounds - 1, [2] * len(mine), 1, want_vectors=False)
            if world == 1:
                _, _, _, p1 = g.run_chains(capi.GIBBS_EXACT, capi.gibbs_chain_seeds(7, 1), args.gibbs_exact_rounds - 1, [2], 1, want_vectors=False)
            g.close()
            b_g = 12 * (len(isid) - N1) + 16 * N1  # conprb + sid per alignment, noise conprb + row slot per read
            per_rank = [[pp.sweep_ms, pp.reduce_ms, pe.sweep_ms, pe.reduce_ms, ps.sweep_ms if mine else 0.0, float(ps.team if mine else 0)]]
            if distributed:
                t = torch.zeros(world, 6, dtype=torch.float64, device=tdev)
                t[rank] = torch.tensor(per_rank[0], dtype=torch.float64)
                dist.all_reduce(t)
                per_rank = t.cpu().tolist()
            sw = max(r[0] for r in per_rank)      # the slowest GPU sets the job's rate (all finish before the reduce)
            ex = max(r[2] for r in per_rank)
            gibbs = {"items_per_chain": int(len(isid)), "gpus": world, "chains_to_ranks": "chain k on rank k % world, one reduce to rank 0 at the end",
                     "parallel": {"mode": "data-augmentation sampler, 1 chain per GPU", "ms_per_sweep": sw,
                                  "ms_per_sweep_per_rank": [r[0] for r in per_rank],
                                  "sweeps_per_s_per_rank": [1e3 / r[0] if r[0] > 0 else None for r in per_rank],
                                  "frac_of_hbm_peak_per_rank": [b_g / r[0] / 1e6 / HBM_PEAK_GBPS if r[0] > 0 else None for r in per_rank],
                                  "algorithmic_GBps_per_gpu": b_g / sw / 1e6 if sw > 0 else None,
                                  "frac_of_hbm_peak_per_gpu": b_g / sw / 1e6 / HBM_PEAK_GBPS if sw > 0 else None,
                                  "sweeps_per_s_all_gpus": world * 1e3 / sw if sw > 0 else None,
                                  "items_per_s_all_gpus": world * len(isid) * 1e3 / sw if sw > 0 else None,
                                  "final_reduce_ms": max(r[1] for r in per_rank) if distributed else None},
                     "exact": {"mode": "reference chain (bit-identical draws), %d chains per GPU advancing together, a team of %d workgroups per chain "
                                       "(k_gibbs_exact_team: the tiles of a window at once)" % (n_exact, pe.team),
                               "workgroups_per_chain": pe.team,
                               "scaling_note": "chains are dealt to the GPUs (chain k on rank k % N); a chain's team grows with the compute units its GPU has "
                                               "to spare -- min(64, CUs / chains of the GPU) -- so this leg keeps 8 chains PER GPU (weak scaling); 8 chains "
                                               "in all on N GPUs would run 8 / N chains per GPU with larger teams (1 chain, 64 workgroups: 56 ms against "
                                               "97 ms per round at a fifth of this size, profiles/r05e_*)",
                               "ms_per_round": ex, "ms_per_round_per_rank": [r[2] for r in per_rank], "chains_per_gpu": n_exact,
                               "rounds_timed": args.gibbs_exact_rounds,
                               "us_per_read_visit_and_chain": ex * 1e3 / N1 if N1 else None,
                               "read_visits_per_s_all_gpus": world * n_exact * N1 * 1e3 / ex if ex > 0 else None,
                               "items_per_s_all_gpus": world * n_exact * len(isid) * 1e3 / ex if ex > 0 else None,
                               "final_reduce_ms": max(r[3] for r in per_rank) if distributed else None,
                               "sum_of_pme_c_over_samples": float(acc_e[0].sum())}}
            exs = max(r[4] for r in per_rank)
            gibbs["exact_strong"] = {"chains_total": strong_total, "chains_per_gpu": (strong_total + world - 1) // world, "gpus": world,
                                     "workgroups_per_chain": int(max(r[5] for r in per_rank)), "ms_per_round": exs,
                                     "ms_per_round_per_rank": [r[4] for r in per_rank] if world > 1 else None,
                                     "rounds_per_s": 1e3 / exs if exs > 0 else None,
                                     "read_visits_per_s_all_chains": strong_total * N1 * 1e3 / exs if exs > 0 else None}
            if p1 is not None:
                gibbs["exact_strong"].update({"one_chain_alone_ms_per_round": p1.sweep_ms, "one_chain_alone_workgroups": p1.team,
                                              "predicted_speedup_1_to_8_gpus": exs / p1.sweep_ms if p1.sweep_ms > 0 else None})
        except Exception as e:  # the EM line must still be printed
            gibbs = {"error": str(e)}

    parity = one_step_parity(ctx, wl) if (rank == 0 and world == 1) else None
    value_plane_bytes = ctx.info("value_plane_bytes")
    key_pmc = args.config + ("_q32" if args.value_bits == 32 else "")
    phys_headline = physical(ctx, estep_ms, pmc_traffic_of(key_pmc, args.scale, args.kernel)[0])
    units_far = far_units(ctx)
    q32 = None
    if world == 1 and not distributed and not args.no_q32 and args.value_bits == 64 and args.kernel in (0, 3):
        # the committed PMC measurement of this layout (profiles/pmc_traffic.json), as for the headline
        q32_traffic = pmc_traffic_of(args.config + "_q32", args.scale, args.kernel)[0]
        q32 = q32_leg(ctx, wl, N0g, K, W, sync, alg_bytes, elapsed * 1e3 / rounds, estep_ms, q32_traffic)
    ctx.close()
    if rank == 0:
        achieved = alg_bytes / (estep_ms * 1e-3) / 1e9
        # PMC passes cannot run inside this process: `traffic` is the committed counter measurement for this workload
        traffic, traffic_src = pmc_traffic_of(key_pmc, args.scale, args.kernel)
        stream = None
        if not args.no_stream:
            try:
                rd, cp = capi.stream_probe(local, 8 << 30, 5)
                stream = {"read_GBps": rd, "copy_GBps": cp, "how": "rsem_hip_stream_probe: 16 B/lane non-temporal wave loads over 4 GiB (read) / 4 GiB -> 4 GiB "
                          "(copy, read + written bytes counted), best of 5 launches, HIP events, in this process right after the timed region"}
            except Exception as e:
                stream = {"error": str(e)}
        phys_achieved = (phys_headline.get("frac_physical") or 0.0) * HBM_PEAK_GBPS
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
                       "units_with_ids_outside_their_window": units_far,
                       "parallelism": "1 GPU" if world == 1 else "read-sharded x%d + RCCL all-reduce(M+1 f64)/round from C++ on the EM stream" % world},
            "roofline": {"bound": "hbm", "achieved": phys_achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": phys_achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "k_estep_lane (E step)", "algorithmic_bytes_per_launch": alg_bytes,
                         "achieved_algorithmic": achieved, "frac_algorithmic": achieved / HBM_PEAK_GBPS,
                         "note": "achieved / frac = the bytes the layout of THIS run moves per launch by construction (physical.parts, from "
                                 "rsem_em_get_info; within 0.3 % of the rocprofv3 PMC counters on every config measured: physical_over_pmc) / this "
                                 "run's average launch time (HIP events on the kernel's stream) / peak: at most 1 by construction.  "
                                 "achieved_algorithmic / frac_algorithmic use the ALGORITHMIC bytes of SURVEY.md 8(d) (12 B per alignment + 16 B per "
                                 "read + 16 B per transcript) over the same launch time; the layout moves fewer (a tuple's transcript ids are re-used "
                                 "from registers, there are no row pointers), so that fraction can pass 1.  `traffic` = the committed rocprofv3 PMC "
                                 "measurement of the same workload (counter passes cannot run inside this process)",
                         "frac_physical": phys_headline.get("frac_physical"), "physical": phys_headline,
                         "avg_launch_ms": estep_ms, "step_over_launch": elapsed * 1e3 / rounds / estep_ms,
                         # the same launch time against the bytes the kernel physically moved (PMC) and against what a plain
                         # streaming kernel reaches on this device (measured here), beside the 8 TB/s specification
                         "frac_of_traffic": (traffic / (estep_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                         "stream": stream,
                         "achieved_over_stream_read": (achieved / stream["read_GBps"]) if stream and stream.get("read_GBps") else None,
                         "traffic_rate_over_stream_read": (traffic / (estep_ms * 1e-3) / 1e9 / stream["read_GBps"])
                         if traffic and stream and stream.get("read_GBps") else None},
            "checks": {"theta_sum": theta_sum, "parity_one_step": parity},
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
            if not args.no_cpu_baseline:  # reported baseline + whole-program wall clock: rank 0 at N=1 only
                cb, e2e = None, None
                try:
                    cb, e2e = reference_e2e(args.config, N1, full_size=not args.no_e2e_full, bam_leg=not args.no_bam_leg, gibbs_leg=not args.no_gibbs)
                except Exception as e:
                    log("reference_e2e failed: %s" % e)
                if cb is None:
                    cb = cpu_baseline_port(wl)
                else:
                    line["cpu_baseline_port_1core"] = cpu_baseline_port(wl, budget_s=5.0)
                line["cpu_baseline"] = cb
                line["speedup_vs_cpu_baseline_rounds"] = line["value"] / cb["value"]
                if e2e is not None:
                    line["e2e_wall_clock"] = e2e
                    if isinstance(e2e.get("gibbs"), dict) and isinstance(line.get("gibbs"), dict):
                        line["gibbs"]["cpu_baseline"] = e2e.pop("gibbs")  # the reference's chains timed in this run, beside the GPU's
            try:  # the builder-run record of both programs at FULL size (profiles/scripts/gpu_r04a.sh), for reference
                with open(os.path.join(ROOT, "profiles", "e2e_full_size_reference.json")) as f:
                    line["e2e_full_size_recorded_round4"] = json.load(f)
            except Exception:
                pass
        os.write(json_fd, (contract_line(line, write_detail(line)) + "\n").encode())
    if comm is not None:
        barrier()
        comm.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
