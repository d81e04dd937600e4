#!/usr/bin/env python3
"""Golden vectors for rsem-calculate-credibility-intervals, made by the REFERENCE binary (oracle/_ref, built from
/root/reference by oracle/Makefile) on the committed fixtures.  Run in the build container:

    python tests/golden/make_ci_golden.py

For every fixture <fx> it writes
  <fx>/ci_pin/   nCV=40 nSpC=10 (400 samples): the reference's sample matrix `s.tmp` (M x nSamples float32, row per
                 transcript, calcCI.cpp:349-352 / Buffer.h:66-80) and the six rows it appended to iso_res / gene_res
                 [/ allele_res].  The rows are a pure function of s.tmp (+ l_bars for FPKM), which pins orc_calc_ci.
  <fx>/ci_stat/  nCV=40 nSpC=500 (20 000 samples): appended rows only; the statistical target for the GPU sampler.
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "rsem-calculate-credibility-intervals")
FIXTURES = ["se_noq", "se_q", "pe_q", "se_q_polya_rspd", "se_q_allele"]


def meta(fx):
    d = {}
    for ln in open(os.path.join(HERE, fx, "META")):
        k, *v = ln.split()
        d[k] = v
    return d


def run_ref(fx, nspc, keep_tmp, out):
    m = meta(fx)
    ncv = int(m["gibbs"][1])
    threads = int(m["gibbs_threads"][0])
    pc = int(m.get("pseudo_count_x1000", ["1000"])[0]) / 1000.0
    with tempfile.TemporaryDirectory() as td:
        w = os.path.join(td, "w")
        shutil.copytree(os.path.join(HERE, fx), w)
        before = {f: len(open(os.path.join(w, "temp", f)).read().split("\n")) for f in os.listdir(os.path.join(w, "temp")) if f.endswith("_res")}
        cmd = [REF, os.path.join(w, "ref"), os.path.join(w, "temp", "s"), os.path.join(w, "stat", "s"), "0.95", str(ncv), str(nspc),
               "1024", "-p", str(threads), "--seed", "777", "-q"] + (["--pseudo-count", str(pc)] if pc != 1.0 else [])
        subprocess.check_call(cmd)
        os.makedirs(out, exist_ok=True)
        for f in before:
            rows = open(os.path.join(w, "temp", f)).read().split("\n")
            new = rows[before[f] - 1:]
            assert len([r for r in new if r]) == 6, (f, len(new))
            open(os.path.join(out, f[2:] + ".txt"), "w").write("\n".join(r for r in new if r) + "\n")
        if keep_tmp:
            shutil.copy(os.path.join(w, "temp", "s.tmp"), os.path.join(out, "s.tmp"))
        open(os.path.join(out, "CMD"), "w").write(" ".join(["rsem-calculate-credibility-intervals"] + cmd[1:4] + cmd[4:]).replace(w + "/", "") + "\n")


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("build oracle/_ref first (make -C oracle ref)")
    for fx in FIXTURES:
        run_ref(fx, 10, True, os.path.join(HERE, fx, "ci_pin"))
        run_ref(fx, 500, False, os.path.join(HERE, fx, "ci_stat"))
        print(fx, "done")
