#!/usr/bin/env python3
"""Reads of a gene that also hit a few transcripts elsewhere (C3X / C3X30) and reads without a gene (C2R): the E step under the
layout policies for them, one workload generation per config, every variant checked against the CPU restatement on the whole
matrix.  Variants are environment knobs read when a context lays its reads out:
    whole      RSEM_HIP_SPLIT=0                                   every read a whole row (ids outside: gather + global atomics)
    most       (round 5's rule) only reads that are mostly outside their window split
    most_noq   RSEM_HIP_FAR_QUEUE=0                               ... the units with ids outside their window in the same launch as the others, a global atomic per far count
               (until round 6; now a launch of their own whose far counts queue up in LDS: estep_block.hpp FarQueue)
    all_1s     RSEM_HIP_SPLIT_POLICY=all RSEM_HIP_X_OVERLAP=0     every read with an id outside splits; one stream
    all        RSEM_HIP_SPLIT_POLICY=all RSEM_HIP_X_OVERLAP=1     ... the split rows' chain beside the compact units (two streams)
(Measured in round 6 and taken out again: id planes loaded in every slice of the split rows' units; the units with ids outside their
window dealt evenly over the launch order -- profiles/r06b_xrows_probe.log .. r06e_xrows_probe.log.)
    python tools/xrows_probe.py [configs=C3X,C3X30,C2R] [variants=most,all_1s,all,all_noids] [scale=1.0]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as orc  # noqa: E402  (the checker beside the measurement)
from rsem_amd import capi  # noqa: E402
from tools.synth_data import make_em_workload  # noqa: E402

VARIANTS = {"whole": {"RSEM_HIP_SPLIT": "0"}, "most": {}, "most_noq": {"RSEM_HIP_FAR_QUEUE": "0"},
            "all_1s": {"RSEM_HIP_SPLIT_POLICY": "all", "RSEM_HIP_X_OVERLAP": "0"}, "all": {"RSEM_HIP_SPLIT_POLICY": "all", "RSEM_HIP_X_OVERLAP": "1"},}
KNOBS = ("RSEM_HIP_SPLIT", "RSEM_HIP_SPLIT_POLICY", "RSEM_HIP_X_OVERLAP", "RSEM_HIP_FAR_QUEUE")

configs = (sys.argv[1] if len(sys.argv) > 1 else "C3X,C3X30,C2R").split(",")
variants = (sys.argv[2] if len(sys.argv) > 2 else "most,all_1s,all").split(",")
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
for cfg in configs:
    t0 = time.perf_counter()
    wl = make_em_workload(cfg, scale=scale)
    M, N1, nnz = wl["M"], len(wl["row_ptr"]) - 1, len(wl["sid"])
    oc = orc.em_estep(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"], wl["theta0"])
    oc[0] += wl["N0"]
    print("== %s: %d reads, %d alignments, %d transcripts (generated + oracle step: %.1f s)" % (cfg, N1, nnz, M, time.perf_counter() - t0), flush=True)
    for v in variants:
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(VARIANTS[v])
        t0 = time.perf_counter()
        ctx = capi.EmContext(M, wl["row_ptr"], wl["sid"], wl["conprb"], wl["ncp"])
        build_s = time.perf_counter() - t0
        counts, *_ = ctx.step(wl["theta0"], wl["N0"])
        err = float(np.max(np.abs(counts - oc) / np.maximum(np.abs(oc), 1e-6)))
        ctx.run(wl["theta0"], wl["N0"], min_round=5, max_round=5)
        best = 1e9
        for rep in range(3):
            p = ctx.run(wl["theta0"], wl["N0"], min_round=30, max_round=30, profile=True)["profile"]
            best = min(best, p.estep_ms_sum / max(p.estep_launches, 1))
        def info(k):
            try:
                return ctx.info(k)
            except Exception:
                return None
        print("   %-10s E step %.4f ms  | parity %.2e %s | split reads %s far entries %s | units %s far units %s | physical MB %s | build %.2f s" % (
            v, best, err, "ok" if err < 1e-9 else "FAIL", info("split_rows"), info("far_entries"), info("units"), info("far_units"),
            (info("physical_bytes_per_launch") or 0) // 1000000, build_s), flush=True)
        ctx.close()
    for k in KNOBS:
        os.environ.pop(k, None)
