"""End-to-end parity of the drop-in programs rsem_amd/bin/rsem-run-em and rsem-run-gibbs against the
reference's outputs on the golden fixtures (same argv, same input files, compared output files).

EM is deterministic: theta (both lines of .theta) must agree to 1e-6 relative with the same ROUND count;
the .model tables to 1e-6; iso_res/gene_res values are printed with %.2f.  Gibbs in exact mode must
write byte-identical count vectors.
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

import rsem_files as rf

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "rsem_amd", "bin")


def _stage(name, tmp_path):
    fx = rf.fixture(name)
    dst = os.path.join(str(tmp_path), name)
    shutil.copytree(fx, dst)
    return fx, dst


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    return r.stdout


def _rel(a, b, floor):
    a, b = np.asarray(a, float), np.asarray(b, float)
    m = np.abs(b) >= floor
    return float(np.max(np.abs(a[m] - b[m]) / np.abs(b[m]))) if m.any() else 0.0


def _res_close(path_a, path_b, atol=0.011):
    A, B = rf.read_res(path_a), rf.read_res(path_b)
    assert len(A) == len(B)
    for ra, rb in zip(A, B):
        try:
            va, vb = np.array(ra, float), np.array(rb, float)
        except ValueError:
            assert ra == rb
            continue
        assert np.allclose(va, vb, atol=atol, rtol=1e-6), (ra[:5], rb[:5])


SHARDED = ["--ngpus", "2", "--devices", "0,0"]  # two row shards on GPU 0 (LOCAL communicator); on a node: RCCL, one GPU each


LEAN = ["--lean-device"]  # after the model rounds: the model context and the caller-order arrays leave the device; .ofg from the planes


@pytest.mark.parametrize("extra", [[], SHARDED, LEAN], ids=["1gpu", "2shards", "lean"])
@pytest.mark.parametrize("name", rf.FIXTURES)
def test_rsem_run_em_matches_reference(name, extra, tmp_path):
    fx, dst = _stage(name, tmp_path)
    meta = rf.read_meta(fx)
    allele = os.path.exists(os.path.join(fx, "ref.ta"))
    for f in ("stat/s.theta", "stat/s.model", "temp/s.ofg", "temp/s.iso_res", "temp/s.gene_res") + (("temp/s.allele_res",) if allele else ()):
        os.remove(os.path.join(dst, f))
    out = _run([os.path.join(BIN, "rsem-run-em"), os.path.join(dst, "ref"), str(meta["model_type"]), os.path.join(dst, "s"),
                os.path.join(dst, "temp", "s"), os.path.join(dst, "stat", "s"), "-p", "1", "--gibbs-out"] + extra)
    # same number of rounds as the reference run, and one ROUND line per round like the reference (EM.cpp:415)
    ref_log = [l for l in open(os.path.join(fx, "em.log")).read().strip().split("\n") if l.startswith("ROUND")]
    ref_rounds = int(ref_log[-1].split(",")[0].split("=")[1])
    my_log = [l for l in out.split("\n") if l.startswith("ROUND")]
    my_rounds = int(my_log[-1].split(",")[0].split("=")[1])
    assert my_rounds == ref_rounds
    assert [int(l.split(",")[0].split("=")[1]) for l in my_log] == list(range(1, ref_rounds + 1))
    for a, b in zip(my_log[11:], ref_log[11:]):  # rounds >= 12: same totNum, bChange to the printed precision
        fa, fb = a.replace(",", "").split(), b.replace(",", "").split()
        assert fa[-1] == fb[-1] and abs(float(fa[8]) - float(fb[8])) <= 2e-5 * max(float(fb[8]), 1e-3), (a, b)
        # SUM: the floating-point sum of the round's counts, as the reference prints it (EM.cpp:394-398,415) -- the same
        # number up to the order of the additions
        assert abs(float(fa[5]) - float(fb[5])) <= 1e-9 * float(fb[5]), (a, b)
    if extra == SHARDED:
        assert sum(l.startswith("GPU ") for l in out.split("\n")) == 2
    # the first 11 rounds print the same SUM / totNum lines as the reference
    ref_lines = open(os.path.join(fx, "em.log")).read().strip().split("\n")[:11]
    my_lines = [l for l in out.split("\n") if l.startswith("ROUND")][:11]
    for a, b in zip(my_lines, ref_lines):
        fa, fb = a.replace(",", "").split(), b.replace(",", "").split()
        assert fa[2] == fb[2] and abs(float(fa[5]) - float(fb[5])) < 1e-6 * float(fb[5]) and fa[-1] == fb[-1], (a, b)
    raw, pol = rf.read_theta(os.path.join(dst, "stat", "s.theta"))
    graw, gpol = rf.read_theta(os.path.join(fx, "stat", "s.theta"))
    assert _rel(raw, graw, 1e-7) < 1e-6 and np.allclose(raw, graw, rtol=1e-6, atol=1e-10)
    assert _rel(pol, gpol, 1e-7) < 1e-6 and np.allclose(pol, gpol, rtol=1e-6, atol=1e-10)
    # model file: every table
    a, b = rf.read_model(os.path.join(dst, "stat", "s.model")), rf.read_model(os.path.join(fx, "stat", "s.model"))
    assert a["type"] == b["type"] and a["gld"][:3] == b["gld"][:3]
    for key in ("qd_init", "qd_tran", "qpro", "nqpro", "pro", "npro", "rspd", "mw"):
        if key in b and b[key] is not None:
            assert np.allclose(a[key], b[key], rtol=1e-6, atol=1e-9), key
    assert np.allclose(a["gld"][3], b["gld"][3], rtol=1e-6, atol=1e-9)
    if b["mld"] is not None:
        assert a["mld"][:3] == b["mld"][:3] and np.allclose(a["mld"][3], b["mld"][3], rtol=1e-6, atol=1e-12)
    # results and the Gibbs input
    _res_close(os.path.join(dst, "temp", "s.iso_res"), os.path.join(fx, "temp", "s.iso_res.em"))
    _res_close(os.path.join(dst, "temp", "s.gene_res"), os.path.join(fx, "temp", "s.gene_res.em"))
    if allele:
        _res_close(os.path.join(dst, "temp", "s.allele_res"), os.path.join(fx, "temp", "s.allele_res.em"))
    M, N0, rp, sid, val = rf.read_ofg(os.path.join(dst, "temp", "s.ofg"))
    gM, gN0, grp, gsid, gval = rf.read_ofg(os.path.join(fx, "temp", "s.ofg"))
    assert (M, N0) == (gM, gN0) and np.array_equal(rp, grp) and np.array_equal(sid, gsid)
    assert np.allclose(val, gval, rtol=1e-6, atol=0)


@pytest.mark.parametrize("extra_dev", [[], ["--devices", "0,0"]], ids=["1gpu", "2groups"])
@pytest.mark.parametrize("name", rf.FIXTURES)
def test_rsem_run_gibbs_exact_matches_reference(name, extra_dev, tmp_path):
    fx, dst = _stage(name, tmp_path)
    meta = rf.read_meta(fx)
    b, n, g = meta["gibbs"]
    imd = os.path.join(dst, "temp", "s")
    shutil.copy(imd + ".iso_res.em", imd + ".iso_res")
    shutil.copy(imd + ".gene_res.em", imd + ".gene_res")
    allele = os.path.exists(os.path.join(fx, "ref.ta"))
    if allele:
        shutil.copy(imd + ".allele_res.em", imd + ".allele_res")
    for k in range(meta["gibbs_threads"]):
        os.remove(imd + ".countvectors%d" % k)
    extra = []
    if meta.get("pseudo_count_x1000", 1000) != 1000:
        extra = ["--pseudo-count", str(meta["pseudo_count_x1000"] / 1000.0)]
    _run([os.path.join(BIN, "rsem-run-gibbs"), os.path.join(dst, "ref"), imd, os.path.join(dst, "stat", "s"), str(b), str(n), str(g),
          "-p", str(meta["gibbs_threads"]), "--seed", str(meta["gibbs_seed"]), "-q", "--gibbs-mode", "exact"] + extra + extra_dev)
    for k in range(meta["gibbs_threads"]):
        with open(imd + ".countvectors%d" % k, "rb") as f1, open(os.path.join(fx, "temp", "s.countvectors%d" % k), "rb") as f2:
            assert f1.read() == f2.read()
    _res_close(imd + ".iso_res", os.path.join(fx, "temp", "s.iso_res"))
    _res_close(imd + ".gene_res", os.path.join(fx, "temp", "s.gene_res"))
    if allele:
        _res_close(imd + ".allele_res", os.path.join(fx, "temp", "s.allele_res"))


def test_rsem_run_gibbs_parallel_runs_and_is_close(tmp_path):
    fx, dst = _stage("pe_q", tmp_path)
    imd = os.path.join(dst, "temp", "s")
    shutil.copy(imd + ".iso_res.em", imd + ".iso_res")
    shutil.copy(imd + ".gene_res.em", imd + ".gene_res")
    _run([os.path.join(BIN, "rsem-run-gibbs"), os.path.join(dst, "ref"), imd, os.path.join(dst, "stat", "s"), "200", "1000", "1",
          "-p", "2", "--seed", "5", "-q", "--gibbs-mode", "parallel", "--gibbs-thin", "4"])
    res = rf.read_res(imd + ".iso_res")
    assert len(res) == 13
    em_counts = np.array(res[4], float)
    pme = np.array(res[8], float)
    cv = np.vstack([rf.read_countvectors(imd + ".countvectors%d" % k) for k in range(2)])
    assert cv.shape[0] == 1000 and np.all(cv.sum(1) == cv[0].sum())
    # posterior means track the ML counts (+1 pseudo count each) on well-determined transcripts
    big = em_counts > 30
    assert np.all(np.abs(pme[big] - em_counts[big]) < 0.25 * em_counts[big] + 3)


@pytest.mark.parametrize("read_type,n_reads,M,threads", [(0, 100_000, 1000, 1), (2, 50_000, 2000, 4), (1, 40000, 2000, 4), (3, 30000, 2000, 4),
                                                          (3, 1_060_000, 20000, 64)])
def test_generated_dataset_vs_reference_binary(read_type, n_reads, M, threads, tmp_path):
    """A fresh, larger synthetic .temp directory (tools/gen_temp.cpp): run the reference binary (oracle/_ref, shipped
    to the GPU box) and the drop-in on the same files; same ROUND count, theta within 1e-6 relative.  The last case is
    paired-end at > 1 M pairs over 20 k transcripts: more transcripts than one LDS window holds (2048), several
    workgroup units per shape -- the out-of-window and multi-unit paths against the REFERENCE, not only the oracle.
    The first case is BASELINE configs[0] as named: SingleModel (type 0, FASTA reads), 100 k reads, 1 k transcripts,
    the reference with -p 1 (SingleModel.h:95-146); the second the paired-end model without qualities."""
    gen = os.path.join(ROOT, "tools", "bin", "gen_temp")
    ref_em = os.path.join(ROOT, "oracle", "_ref", "rsem-run-em")
    ref_idx = os.path.join(ROOT, "oracle", "_ref", "rsem-build-read-index")
    if not (os.path.exists(gen) and os.path.exists(ref_em) and os.path.exists(ref_idx)):
        pytest.skip("generator or reference binaries not built")
    d = str(tmp_path)
    _run([gen, d, str(n_reads), str(M), str(read_type), "7", "75"])
    ext = ".fq" if read_type in (1, 3) else ".fa"
    reads = ["s_alignable" + ext] if read_type < 2 else ["s_alignable_1" + ext, "s_alignable_2" + ext]
    # rsem-build-read-index gap hasQ quiet files (buildReadIndex.cpp:72-84), as rsem-calculate-expression calls it
    _run([ref_idx, "32", "1" if read_type in (1, 3) else "0", "1"] + [os.path.join(d, "temp", r) for r in reads])
    args = [os.path.join(d, "ref"), str(read_type), os.path.join(d, "s"), os.path.join(d, "temp", "s"), os.path.join(d, "stat", "s")]
    out_ref = _run([ref_em] + args + ["-p", str(threads)])
    graw, gpol = rf.read_theta(os.path.join(d, "stat", "s.theta"))
    gres = rf.read_res(os.path.join(d, "temp", "s.iso_res"))
    os.rename(os.path.join(d, "stat", "s.theta"), os.path.join(d, "stat", "ref.theta"))
    out_new = _run([os.path.join(BIN, "rsem-run-em")] + args)
    raw, pol = rf.read_theta(os.path.join(d, "stat", "s.theta"))
    r_ref = int([l for l in out_ref.split("\n") if l.startswith("ROUND")][-1].split(",")[0].split("=")[1])
    r_new = int([l for l in out_new.split("\n") if l.startswith("ROUND")][-1].split(",")[0].split("=")[1])
    assert r_new == r_ref
    assert _rel(raw, graw, 1e-7) < 1e-6 and _rel(pol, gpol, 1e-7) < 1e-6
    res = rf.read_res(os.path.join(d, "temp", "s.iso_res"))
    assert np.allclose(np.array(res[5], float), np.array(gres[5], float), atol=0.011, rtol=1e-6)  # TPM


@pytest.mark.parametrize("name", ["se_q", "se_noq_rev_rspd_omit", "pe_q", "pe_noq", "se_q_polya_rspd"])
def test_rsem_run_em_binary_input_equals_text_input(name, tmp_path):
    """imdName.rsb/ (rsem-parse-alignments --binary, host/rsb.hpp) instead of .dat + read files: same rounds, same theta
    bits-for-tolerance, same model, same .ofg -- and the text files are really not read (they are deleted)."""
    conv = os.path.join(ROOT, "tools", "bin", "temp_to_rsb")
    if not os.path.exists(conv):
        pytest.skip("tools/bin/temp_to_rsb not built")
    fx, dst = _stage(name, tmp_path)
    meta = rf.read_meta(fx)
    rt = str(meta["model_type"])
    args = [os.path.join(dst, "ref"), rt, os.path.join(dst, "s"), os.path.join(dst, "temp", "s"), os.path.join(dst, "stat", "s"), "--gibbs-out"]
    out_t = _run([os.path.join(BIN, "rsem-run-em")] + args)
    keep = {}
    for f in ("stat/s.theta", "stat/s.model", "temp/s.ofg", "temp/s.iso_res"):
        keep[f] = open(os.path.join(dst, f)).read()
    _run([conv, os.path.join(dst, "temp", "s"), os.path.join(dst, "stat", "s"), rt])
    for f in os.listdir(os.path.join(dst, "temp")):
        if f == "s.dat" or f.endswith((".fq", ".fa")):
            os.remove(os.path.join(dst, "temp", f))
    out_b = _run([os.path.join(BIN, "rsem-run-em")] + args)
    lt = [l for l in out_t.split("\n") if l.startswith("ROUND")]
    lb = [l for l in out_b.split("\n") if l.startswith("ROUND")]
    assert len(lt) == len(lb) and lt[:11] == lb[:11]
    ta, tb = rf.read_theta(os.path.join(dst, "stat", "s.theta")), None
    open(os.path.join(dst, "stat", "t.theta"), "w").write(keep["stat/s.theta"])
    tb = rf.read_theta(os.path.join(dst, "stat", "t.theta"))
    assert np.allclose(ta[0], tb[0], rtol=1e-9, atol=1e-15) and np.allclose(ta[1], tb[1], rtol=1e-9, atol=1e-15)
    assert open(os.path.join(dst, "stat", "s.model")).read().split() [:50] == keep["stat/s.model"].split()[:50]
    open(os.path.join(dst, "temp", "t.ofg"), "w").write(keep["temp/s.ofg"])
    oa, ob = rf.read_ofg(os.path.join(dst, "temp", "s.ofg")), rf.read_ofg(os.path.join(dst, "temp", "t.ofg"))
    assert oa[:2] == ob[:2] and np.array_equal(oa[2], ob[2]) and np.array_equal(oa[3], ob[3]) and np.allclose(oa[4], ob[4], rtol=1e-9, atol=0)


def _bam_records(path):
    """Decompress a BAM (BGZF = concatenated gzip members) and split it into (header bytes, [record bytes])."""
    import gzip
    import struct
    raw = gzip.open(path, "rb").read()
    assert raw[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, p)[0]
    p += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", raw, p)[0]
        p += 4 + l_name + 4
    header = raw[:p]
    recs = []
    while p < len(raw):
        bs = struct.unpack_from("<i", raw, p)[0]
        recs.append(raw[p + 4:p + 4 + bs])
        p += 4 + bs
    return header, recs


def _assert_bam_equal(mine, gold):
    import struct
    h1, r1 = _bam_records(mine)
    h2, r2 = _bam_records(gold)
    assert h1 == h2
    assert len(r1) == len(r2)
    n_diff = 0
    for a, b in zip(r1, r2):
        if a == b:
            continue
        # only MAPQ (byte 9) and the trailing ZW:f value may differ, and only by rounding of the posterior weight
        assert len(a) == len(b) and a[:9] == b[:9] and a[10:-4] == b[10:-4]
        assert abs(a[9] - b[9]) <= 1
        fa, fb = struct.unpack("<f", a[-4:])[0], struct.unpack("<f", b[-4:])[0]
        assert abs(fa - fb) <= 1e-6 * max(abs(fb), 1e-30) + 1e-12
        n_diff += 1
    assert n_diff <= max(2, len(r1) // 200)


@pytest.mark.parametrize("name", ["se_q", "pe_q"])
def test_transcript_bam_matches_reference(name, tmp_path):
    """-b: <sample>.transcript.bam (MAPQ + ZW:f from the posteriors) from SAM input, and from BAM input with
    --sampling --seed (one alignment per read drawn with the reference's MT19937 stream)."""
    fx, dst = _stage(name, tmp_path)
    meta = rf.read_meta(fx)
    args = [os.path.join(dst, "ref"), str(meta["model_type"]), os.path.join(dst, "s"), os.path.join(dst, "temp", "s"),
            os.path.join(dst, "stat", "s"), "-p", "1"]
    _run([os.path.join(BIN, "rsem-run-em")] + args + ["-b", os.path.join(dst, "aln.sam"), "0"])
    _assert_bam_equal(os.path.join(dst, "s.transcript.bam"), os.path.join(fx, "golden.transcript.bam"))
    _run([os.path.join(BIN, "rsem-run-em")] + args + ["-b", os.path.join(fx, "golden.transcript.bam"), "0", "--sampling", "--seed", "77"])
    _assert_bam_equal(os.path.join(dst, "s.transcript.bam"), os.path.join(fx, "golden.sampled.transcript.bam"))


@pytest.mark.parametrize("name", ["se_noq", "se_q", "pe_q", "se_q_polya_rspd", "se_q_allele"])
def test_credibility_intervals_with_the_reference_stream_are_the_reference_rows(name, tmp_path):
    """--ci-stream reference: the reference's own draws (per-thread MT19937 seeded by the engine factory, boost's gamma
    distribution, calcCI.cpp:93-164, sampling.h:19-44; restated in csrc/host/ci_stream.hpp) for the same --seed and -p, the
    intervals on the GPU: the six appended rows are the rows the REFERENCE BINARY appended (ci_stat golden, same argv) -- not
    statistically, but value for value as printed with %.6g (a last printed digit may differ where the host's libm rounds a
    transcendental differently from the machine the golden was made on; the fixtures were made with this image's)."""
    fx, dst = _stage(name, tmp_path)
    meta = rf.read_meta(fx)
    nCV = int(meta["gibbs"][1])
    threads = int(meta["gibbs_threads"])
    pc = meta.get("pseudo_count_x1000", 1000) / 1000.0
    imd = os.path.join(dst, "temp", "s")
    res_files = [f for f in ("iso_res", "gene_res", "allele_res") if os.path.exists(imd + "." + f)]
    before = {f: open(imd + "." + f).read() for f in res_files}
    cmd = [os.path.join(BIN, "rsem-calculate-credibility-intervals"), os.path.join(dst, "ref"), imd, os.path.join(dst, "stat", "s"),
           "0.95", str(nCV), "500", "1024", "-p", str(threads), "--seed", "777", "-q", "--ci-stream", "reference"]
    if pc != 1.0:
        cmd += ["--pseudo-count", str(pc)]
    _run(cmd)
    n_fields = n_same = 0
    for f in res_files:
        after = open(imd + "." + f).read()
        assert after.startswith(before[f])
        new = after[len(before[f]):].strip("\n").split("\n")
        gold = open(os.path.join(fx, "ci_stat", f + ".txt")).read().strip().split("\n")
        assert len(new) == 6 and len(gold) == 6
        if f == "iso_res" and "allele_res" in res_files:
            continue  # the reference's rows here carry its accumulator quirk (calcCI.cpp:308-311: never reset between transcripts; tests/test_ci_cpu.py)
        for a, b in zip(new, gold):
            fa, fb = a.split("\t"), b.split("\t")
            assert len(fa) == len(fb)
            n_fields += len(fa)
            n_same += sum(x == y for x, y in zip(fa, fb))
            assert np.allclose(np.array(fa, float), np.array(fb, float), rtol=2e-5, atol=1e-9)
    assert n_same >= 0.995 * n_fields, (n_same, n_fields)


@pytest.mark.parametrize("name", ["se_noq", "se_q", "pe_q", "se_q_polya_rspd", "se_q_allele"])
def test_rsem_calculate_credibility_intervals_cli(name, tmp_path):
    """Drop-in rsem-calculate-credibility-intervals on the fixture's own count vectors: six "%.6g" rows appended to
    every result file, statistically equal to the rows the reference binary appended (ci_stat golden, same argv;
    tolerances as tests/test_ci_gpu.py), nothing else in the files touched, no imd.tmp left behind."""
    fx, dst = _stage(name, tmp_path)
    meta = rf.read_meta(fx)
    nCV = int(meta["gibbs"][1])
    threads = int(meta["gibbs_threads"])
    pc = meta.get("pseudo_count_x1000", 1000) / 1000.0
    imd = os.path.join(dst, "temp", "s")
    res_files = [f for f in ("iso_res", "gene_res", "allele_res") if os.path.exists(imd + "." + f)]
    before = {f: open(imd + "." + f).read() for f in res_files}
    cmd = [os.path.join(BIN, "rsem-calculate-credibility-intervals"), os.path.join(dst, "ref"), imd, os.path.join(dst, "stat", "s"),
           "0.95", str(nCV), "500", "1024", "-p", str(threads), "--seed", "777", "-q"]
    if pc != 1.0:
        cmd += ["--pseudo-count", str(pc)]
    _run(cmd)
    assert not os.path.exists(imd + ".tmp")
    for f in res_files:
        after = open(imd + "." + f).read()
        assert after.startswith(before[f])
        new = after[len(before[f]):].strip("\n").split("\n")
        assert len(new) == 6
        got = np.array([[float(x) for x in l.split("\t")] for l in new])
        for l in new:  # "%.6g"
            assert all(("%.6g" % float(x)) == x for x in l.split("\t"))
        ref = np.array([[float(x) for x in l.split("\t")] for l in open(os.path.join(fx, "ci_stat", f + ".txt")).read().strip().split("\n")])
        assert got.shape == ref.shape
        if f == "iso_res" and "allele_res" in res_files:
            continue  # the reference's rows here carry its accumulator quirk (tests/test_ci_cpu.py)
        for k in (0, 3):
            tol = 0.10 * (ref[k + 1] - ref[k]) + 1e-3 * np.abs(ref[k + 1]) + 1e-6
            assert (np.abs(got[k] - ref[k]) <= tol).all() and (np.abs(got[k + 1] - ref[k + 1]) <= tol).all(), (f, k)
            assert (np.abs(got[k + 2] - ref[k + 2]) <= 0.02).all(), (f, k)
    # same seed -> same rows
    first = {f: open(imd + "." + f).read() for f in res_files}
    for f in res_files:
        open(imd + "." + f, "w").write(before[f])
    _run(cmd)
    for f in res_files:
        assert open(imd + "." + f).read() == first[f]


@pytest.mark.parametrize("name", ["se_q", "pe_noq"])
def test_gibbs_binary_handoff_between_the_two_programs(name, tmp_path):
    """rsem-run-em --gibbs-out with RSEM_HIP_BINARY=both writes imd.ofg AND imd.ofb/ (host/ofb.hpp); rsem-run-gibbs on
    the arrays alone draws the chains it draws from the text: count-vector files byte-equal (and equal to the reference's,
    when the EM run reproduces the golden .ofg)."""
    fx, dst = _stage(name, tmp_path)
    meta = rf.read_meta(fx)
    rt = str(meta["model_type"])
    imd = os.path.join(dst, "temp", "s")
    env = dict(os.environ, RSEM_HIP_BINARY="both")
    r = subprocess.run([os.path.join(BIN, "rsem-run-em"), os.path.join(dst, "ref"), rt, os.path.join(dst, "s"), imd, os.path.join(dst, "stat", "s"), "-p", "1",
                        "--gibbs-out"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert os.path.exists(imd + ".ofb/hdr") and os.path.exists(imd + ".ofg")
    # the header is written after the text file is closed: rsem-run-gibbs must find the arrays not older than the text
    assert os.stat(imd + ".ofb/hdr").st_mtime_ns >= os.stat(imd + ".ofg").st_mtime_ns
    M, N0, rp, sid, val = rf.read_ofg(imd + ".ofg")
    brp = np.fromfile(imd + ".ofb/row_ptr", np.uint64)
    assert np.array_equal(brp, rp) and np.array_equal(np.fromfile(imd + ".ofb/sid", np.int32), sid)
    assert np.array_equal(np.fromfile(imd + ".ofb/val", np.float64), val)  # bit for bit what the text parses to
    b, n, g = meta["gibbs"]
    args = [os.path.join(BIN, "rsem-run-gibbs"), os.path.join(dst, "ref"), imd, os.path.join(dst, "stat", "s"), str(b), str(n), str(g),
            "-p", str(meta["gibbs_threads"]), "--seed", str(meta["gibbs_seed"]), "-q"]
    for f in ("iso_res", "gene_res"):
        shutil.copy(imd + "." + f, imd + "." + f + ".keep")  # (rsem-run-gibbs appends its columns to these files)
    def chains(tag):
        for f in ("iso_res", "gene_res"):
            shutil.copy(imd + "." + f + ".keep", imd + "." + f)
        log = _run(args)
        assert ("newer than" in log) == (tag == "stale"), log[-1500:]  # the warning of ofb_present() = the text was chosen
        out = [open(imd + ".countvectors%d" % k, "rb").read() for k in range(meta["gibbs_threads"])]
        return out
    from_both = chains("both")       # .ofb present and not older than .ofg: the arrays are used (no warning)
    os.utime(imd + ".ofg")           # a text file written AFTER the arrays wins, and says so
    assert chains("stale") == from_both
    os.remove(imd + ".ofg")
    from_arrays = chains("arrays")   # the arrays alone
    shutil.rmtree(imd + ".ofb")
    with open(imd + ".ofg", "w") as f:  # the text alone, rebuilt from what was parsed above
        f.write("%d %d\n" % (M, N0))
        for i in range(len(rp) - 1):
            a, e = int(rp[i]), int(rp[i + 1])
            f.write("".join("%d %.15g " % (sid[j], val[j]) for j in range(a, e)) + "\n")
    from_text = chains("text")
    assert from_both == from_arrays == from_text
