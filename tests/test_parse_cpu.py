"""rsem-parse-alignments drop-in (rsem_amd/csrc/host/parse_alignments.cpp) against the reference's outputs.

Host-only stage, so these run without a GPU.  Golden: the fixtures' .dat/.cnt/.omit/read files were written by the
reference's parser (parseIt.cpp) from the committed aln.sam.  Where oracle/_ref/rsem-parse-alignments is present, more
variants (FASTA read types, BAM input, the -tag filter, reverse-strand heavy input) are compared with it byte for byte.
"""
import filecmp
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
NEW = os.path.join(ROOT, "rsem_amd", "bin", "rsem-parse-alignments")
REF = os.path.join(ROOT, "oracle", "_ref", "rsem-parse-alignments")


def _need_new():
    if not os.path.exists(NEW):
        from rsem_amd import build
        build.build()
    assert os.path.exists(NEW), "rsem-parse-alignments was not built"


def _run(exe, ref, out, aln, read_type, extra=()):
    os.makedirs(os.path.join(out, "temp"), exist_ok=True)
    os.makedirs(os.path.join(out, "stat"), exist_ok=True)
    cmd = [exe, ref, os.path.join(out, "temp", "s"), os.path.join(out, "stat", "s"), aln, str(read_type), "-q"] + list(extra)
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def _files(d):
    out = []
    for sub in ("temp", "stat"):
        for f in sorted(os.listdir(os.path.join(d, sub))):
            out.append(os.path.join(sub, f))
    return out


def _same_tree(a, b):
    fa, fb = _files(a), _files(b)
    assert fa == fb, (fa, fb)
    for f in fa:
        assert filecmp.cmp(os.path.join(a, f), os.path.join(b, f), shallow=False), f


@pytest.mark.parametrize("name,read_type", [("se_q", 1), ("pe_q", 3)])
def test_parse_matches_golden(name, read_type, tmp_path):
    _need_new()
    fx = os.path.join(GOLD, name)
    r = _run(NEW, os.path.join(fx, "ref"), str(tmp_path), os.path.join(fx, "aln.sam"), read_type)
    assert r.returncode == 0, r.stderr
    produced = _files(str(tmp_path))
    assert "temp/s.dat" in produced and "stat/s.cnt" in produced and "temp/s.omit" in produced
    for f in produced:
        assert filecmp.cmp(os.path.join(str(tmp_path), f), os.path.join(fx, f), shallow=False), f
    # and nothing the reference's parser wrote is missing
    for f in os.listdir(os.path.join(fx, "temp")):
        if f.startswith("s_") and (f.endswith(".fq") or f.endswith(".fa")):
            assert os.path.join("temp", f) in produced, f


def _tagged_sam(src, dst, paired):
    """Mark every third unaligned read (pair) with ZT:i:2 (what `-tag ZT` treats as "filtered", SamParser.h:61-82)."""
    k = 0
    with open(src) as fi, open(dst, "w") as fo:
        lines = fi.readlines()
        i = 0
        while i < len(lines):
            ln = lines[i]
            if ln.startswith("@"):
                fo.write(ln); i += 1; continue
            grp = lines[i:i + (2 if paired else 1)]
            i += len(grp)
            unal = int(grp[0].split("\t")[1]) & 4
            if unal:
                k += 1
                if k % 3 == 0:
                    grp = [grp[0].rstrip("\n") + "\tZT:i:2\n"] + [g.rstrip("\n") + "\tZT:i:0\n" for g in grp[1:]]
                elif k % 3 == 1 and paired:
                    grp = [grp[0], grp[1].rstrip("\n") + "\tXS:Z:abc\tZT:i:300\n"]
            fo.writelines(grp)


VARIANTS = [
    ("se_q", 1, "aln.sam", ()), ("se_q", 0, "aln.sam", ()), ("pe_q", 3, "aln.sam", ()), ("pe_q", 2, "aln.sam", ()),
    ("se_q", 1, "golden.transcript.bam", ()), ("pe_q", 3, "golden.transcript.bam", ()),
    # tiny waves / odd thread counts: reads straddle wave and chunk boundaries (the reference ignores these options)
    ("se_q", 1, "aln.sam", ("--wave-bytes", "3000", "-p", "3")), ("pe_q", 3, "aln.sam", ("--wave-bytes", "5000", "-p", "7")),
    ("se_q", 1, "golden.transcript.bam", ("--wave-bytes", "3000", "-p", "5")),
    ("pe_q", 3, "golden.transcript.bam", ("--wave-bytes", "4000", "-p", "2")), ("pe_q", 2, "aln.sam", ("-p", "1")),
    ("se_q", 1, "tagged", ("-tag", "ZT")), ("pe_q", 3, "tagged", ("-tag", "ZT")), ("pe_q", 2, "tagged", ("-tag", "ZT")),
]


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/rsem-parse-alignments not built")
@pytest.mark.parametrize("name,read_type,aln,extra", VARIANTS)
def test_parse_matches_reference_binary(name, read_type, aln, extra, tmp_path):
    _need_new()
    fx = os.path.join(GOLD, name)
    if aln == "tagged":
        alnf = str(tmp_path / "tagged.sam")
        _tagged_sam(os.path.join(fx, "aln.sam"), alnf, read_type >= 2)
    else:
        alnf = os.path.join(fx, aln)
    a, b = str(tmp_path / "new"), str(tmp_path / "ref")
    r1 = _run(NEW, os.path.join(fx, "ref"), a, alnf, read_type, extra)
    r2 = _run(REF, os.path.join(fx, "ref"), b, alnf, read_type, extra)
    assert r2.returncode == 0, r2.stderr
    assert r1.returncode == 0, r1.stderr
    _same_tree(a, b)
    if "-tag" in extra:
        assert any(f.startswith("temp/s_max") for f in _files(a))  # the filter really fired


def test_parse_error_paths(tmp_path):
    """Error behaviour of SamParser.h:121-141: message on stderr, non-zero exit."""
    _need_new()
    fx = os.path.join(GOLD, "se_q")
    # paired-end records handed to a single-end parse
    r = _run(NEW, os.path.join(GOLD, "pe_q", "ref"), str(tmp_path / "a"), os.path.join(GOLD, "pe_q", "aln.sam"), 1)
    assert r.returncode != 0 and "paired end read" in r.stderr
    # gapped alignment
    bad = str(tmp_path / "gap.sam")
    done = False
    with open(os.path.join(fx, "aln.sam")) as fi, open(bad, "w") as fo:
        for ln in fi:
            f = ln.split("\t")
            if not done and not ln.startswith("@") and not (int(f[1]) & 4):
                L = len(f[9])
                f[5] = "%dM1I%dM" % (L // 2, L - L // 2 - 1)
                ln = "\t".join(f)
                done = True
            fo.write(ln)
    r = _run(NEW, os.path.join(fx, "ref"), str(tmp_path / "b"), bad, 1)
    assert r.returncode != 0 and "gapped alignments" in r.stderr
    # unknown reference name
    bad2 = str(tmp_path / "sq.sam")
    with open(os.path.join(fx, "aln.sam")) as fi, open(bad2, "w") as fo:
        first = True
        for ln in fi:
            if first and ln.startswith("@SQ"):
                ln = ln.replace("SN:", "SN:zz_", 1)
                first = False
            fo.write(ln)
    r = _run(NEW, os.path.join(fx, "ref"), str(tmp_path / "c"), bad2, 1)
    assert r.returncode != 0 and "can not recognize reference sequence name" in r.stderr


@pytest.mark.parametrize("read_type", [1, 3])
def test_parse_round_trip_generated(read_type, tmp_path):
    """tools/gen_temp.cpp writes a .temp directory AND the SAM those files would have come from: parsing the SAM must
    reproduce .dat / .omit / read files byte for byte (size-independent property; 30 k reads, ~200 k alignments,
    reverse-strand and multi-isoform hits, several waves)."""
    _need_new()
    gen = os.path.join(ROOT, "tools", "bin", "gen_temp")
    if not os.path.exists(gen):
        os.makedirs(os.path.dirname(gen), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", gen, os.path.join(ROOT, "tools", "gen_temp.cpp")])
    d = str(tmp_path / "g")
    subprocess.check_call([gen, d, "30000", "600", str(read_type), "11", "75", "sam"], stdout=subprocess.DEVNULL)
    out = str(tmp_path / "p")
    r = _run(NEW, os.path.join(d, "ref"), out, os.path.join(d, "aln.sam"), read_type, ("--wave-bytes", "2000000", "-p", "4"))
    assert r.returncode == 0, r.stderr
    names = ["temp/s.dat", "temp/s.omit"] + ["temp/" + f for f in os.listdir(os.path.join(d, "temp")) if f.endswith(".fq")]
    assert len(names) == (6 if read_type == 3 else 4)
    for f in names:
        assert filecmp.cmp(os.path.join(d, f), os.path.join(out, f), shallow=False), f
    cnt = open(os.path.join(out, "stat", "s.cnt")).read().split("\n")
    assert cnt[0].split() == ["1500", "28500", "0", "30000"]
    hist = dict(l.split("\t") for l in cnt[3:] if "\t" in l)
    assert sum(int(v) for k, v in hist.items() if k not in ("0", "Inf")) == 28500


def _edit_sam(src, dst, fn):
    with open(src) as fi, open(dst, "w", newline="") as fo:
        body = [l for l in fi]
        fo.writelines(fn(body))


EDGE_CASES = {
    # name: (fixture, read_type, transformation of the SAM lines)
    "no_trailing_newline": ("se_q", 1, lambda L: L[:-1] + [L[-1].rstrip("\n")]),
    "crlf_line_ends": ("se_q", 1, lambda L: [l.rstrip("\n") + "\r\n" for l in L]),
    "lower_case_bases": ("se_q", 1, lambda L: [l if l.startswith("@") else "\t".join(f.lower() if i == 9 else f for i, f in enumerate(l.split("\t"))) for l in L]),
    "qname_with_spaces": ("se_q", 1, lambda L: [l if l.startswith("@") else l.replace("\t", " extra words\t", 1) for l in L]),
    "missing_qualities": ("se_q", 1, lambda L: [l if l.startswith("@") else "\t".join("*" if i == 10 else f for i, f in enumerate(l.rstrip("\n").split("\t"))) + "\n" for l in L]),
    "eq_and_x_cigar": ("se_q", 1, lambda L: [l if l.startswith("@") or l.split("\t")[5] == "*" else "\t".join((f[:-1] + ("=" if n % 2 else "X")) if i == 5 else f for i, f in enumerate(l.split("\t"))) for n, l in enumerate(L)]),
    "header_only": ("se_q", 1, lambda L: [l for l in L if l.startswith("@")]),
    "pe_mates_swapped": ("pe_q", 3, lambda L: [l for l in L if l.startswith("@")] + [x for a, b in zip(*[iter([l for l in L if not l.startswith("@")])] * 2) for x in (b, a)]),
    "pe_mate_names_differ": ("pe_q", 3, lambda L: [l if l.startswith("@") or not (int(l.split("\t")[1]) & 128) else "m2_" + l for l in L]),
}


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/rsem-parse-alignments not built")
@pytest.mark.parametrize("case", sorted(EDGE_CASES))
def test_parse_edge_cases_match_reference_binary(case, tmp_path):
    """Inputs htslib accepts in more than one spelling: the drop-in's own SAM decoder must land on the same bytes."""
    _need_new()
    name, read_type, fn = EDGE_CASES[case]
    fx = os.path.join(GOLD, name)
    alnf = str(tmp_path / "in.sam")
    _edit_sam(os.path.join(fx, "aln.sam"), alnf, fn)
    a, b = str(tmp_path / "new"), str(tmp_path / "ref")
    r2 = _run(REF, os.path.join(fx, "ref"), b, alnf, read_type)
    for extra in ((), ("--wave-bytes", "2500", "-p", "3")):
        r1 = _run(NEW, os.path.join(fx, "ref"), a, alnf, read_type, extra)
        assert r2.returncode == 0, r2.stderr
        assert r1.returncode == 0, r1.stderr
        _same_tree(a, b)
        if case == "pe_mate_names_differ":
            assert "two mates have different names" in r1.stderr and "two mates have different names" in r2.stderr
        import shutil
        shutil.rmtree(a)


def test_parse_more_error_paths(tmp_path):
    _need_new()
    fx = os.path.join(GOLD, "se_q")
    src = os.path.join(fx, "aln.sam")
    lines = open(src).read().split("\n")
    body0 = next(i for i, l in enumerate(lines) if l and not l.startswith("@"))
    first_aligned = next(i for i, l in enumerate(lines) if l and not l.startswith("@") and not (int(l.split("\t")[1]) & 4))

    def write(name, L):
        p = str(tmp_path / name)
        open(p, "w").write("\n".join(L))
        return p

    # an unaligned record followed by an alignment of the same read name (parseIt.cpp:97)
    f = lines[first_aligned].split("\t")
    un = "\t".join([f[0], "4", "*", "0", "0", "*", "*", "0", "0", f[9], f[10]])
    p = write("both.sam", lines[:first_aligned] + [un] + lines[first_aligned:])
    r = _run(NEW, os.path.join(fx, "ref"), str(tmp_path / "a"), p, 1)
    assert r.returncode != 0 and "both unalignable and alignable" in r.stderr
    # two alignments of one read with different read lengths (SamParser.h:133)
    g = list(f)
    g[9], g[10], g[5] = g[9][:-1], g[10][:-1], "%dM" % (len(g[9]) - 1)
    p = write("len.sam", lines[:first_aligned + 1] + ["\t".join(g)] + lines[first_aligned + 1:])
    r = _run(NEW, os.path.join(fx, "ref"), str(tmp_path / "b"), p, 1)
    assert r.returncode != 0 and "inconsistent read lengths" in r.stderr
    # an ambiguity code the reference asserts on (sam_utils.h:88-96)
    g = list(f)
    g[9] = "R" + g[9][1:]
    p = write("iupac.sam", lines[:first_aligned] + ["\t".join(g)] + lines[first_aligned + 1:])
    r = _run(NEW, os.path.join(fx, "ref"), str(tmp_path / "c"), p, 1)
    assert r.returncode != 0 and "base other than A, C, G, T, N" in r.stderr
    # missing file / not enough arguments
    r = _run(NEW, os.path.join(fx, "ref"), str(tmp_path / "d"), str(tmp_path / "nope.sam"), 1)
    assert r.returncode != 0 and "Cannot open" in r.stderr
    r = subprocess.run([NEW, "x"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode != 0 and "Usage" in r.stdout
    assert body0 > 0


BIN_VARIANTS = [("se_q", 1, "aln.sam", ()), ("se_q", 0, "aln.sam", ()), ("pe_q", 3, "aln.sam", ("--wave-bytes", "5000", "-p", "7")),
                ("pe_q", 2, "golden.transcript.bam", ()), ("pe_q", 3, "tagged", ("-tag", "ZT"))]


@pytest.mark.parametrize("name,read_type,aln,extra", BIN_VARIANTS)
def test_binary_handoff_equals_the_text_files(name, read_type, aln, extra, tmp_path):
    """--binary: imdName.rsb/ holds, as arrays, exactly what the text files hold.  Checked by
    converting the parser's own text output with tools/temp_to_rsb (which goes through the text READERS of rsem-run-em):
    every file of the two directories must be byte-identical.  --binary alone writes no .dat / read files, --binary
    --text writes both, .cnt / .omit are always there and identical."""
    _need_new()
    conv = os.path.join(ROOT, "tools", "bin", "temp_to_rsb")
    if not os.path.exists(conv):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", conv, os.path.join(ROOT, "tools", "temp_to_rsb.cpp")])
    fx = os.path.join(GOLD, name)
    src = os.path.join(fx, aln)
    if aln == "tagged":
        src = str(tmp_path / "tagged.sam")
        _tagged_sam(os.path.join(fx, "aln.sam"), src, read_type >= 2)
    t, b, both = str(tmp_path / "text"), str(tmp_path / "bin"), str(tmp_path / "both")
    for d, fl in ((t, ()), (b, ("--binary",)), (both, ("--binary", "--text"))):
        r = _run(NEW, os.path.join(fx, "ref"), d, src, read_type, tuple(extra) + fl)
        assert r.returncode == 0, r.stderr
    assert not os.path.exists(os.path.join(t, "temp", "s.rsb"))
    assert sorted(os.listdir(os.path.join(b, "temp"))) == ["s.omit", "s.rsb"]  # no .dat, no read files
    for f in ("temp/s.omit", "stat/s.cnt"):
        assert filecmp.cmp(os.path.join(t, f), os.path.join(b, f), shallow=False), f
    # --text keeps the reference's files next to the binary ones
    for f in _files(t):
        assert filecmp.cmp(os.path.join(t, f), os.path.join(both, f), shallow=False), f
    # the conversion of the text output = the binary output
    subprocess.check_call([conv, os.path.join(t, "temp", "s"), os.path.join(t, "stat", "s"), str(read_type)], stdout=subprocess.DEVNULL)
    ra, rb = os.path.join(t, "temp", "s.rsb"), os.path.join(b, "temp", "s.rsb")
    assert sorted(os.listdir(ra)) == sorted(os.listdir(rb)) and "hdr" in os.listdir(rb)
    for f in os.listdir(ra):
        assert filecmp.cmp(os.path.join(ra, f), os.path.join(rb, f), shallow=False), f
    for f in os.listdir(ra):
        assert filecmp.cmp(os.path.join(ra, f), os.path.join(both, "temp", "s.rsb", f), shallow=False), f


def test_binary_handoff_via_environment(tmp_path):
    """The Perl driver builds the parser's command line; RSEM_HIP_BINARY switches the hand-off without touching it."""
    _need_new()
    fx = os.path.join(GOLD, "se_q")
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "temp")); os.makedirs(os.path.join(d, "stat"))
    cmd = [NEW, os.path.join(fx, "ref"), os.path.join(d, "temp", "s"), os.path.join(d, "stat", "s"), os.path.join(fx, "aln.sam"), "1", "-q"]
    subprocess.check_call(cmd, env=dict(os.environ, RSEM_HIP_BINARY="1"))
    assert os.path.exists(os.path.join(d, "temp", "s.rsb", "hdr")) and not os.path.exists(os.path.join(d, "temp", "s.dat"))
    # ... but the driver's next command is the reference's rsem-build-read-index on the alignable read files
    # (rsem-calculate-expression:597-604): they are still there, identical to the text run's, and the index builds
    ali = os.path.join(d, "temp", "s_alignable.fq")
    assert filecmp.cmp(ali, os.path.join(fx, "temp", "s_alignable.fq"), shallow=False)
    assert not os.path.exists(os.path.join(d, "temp", "s_un.fq"))
    ref_idx = os.path.join(ROOT, "oracle", "_ref", "rsem-build-read-index")
    if os.path.exists(ref_idx):
        subprocess.check_call([ref_idx, "32", "1", "1", ali])
        assert os.path.exists(ali + ".ridx")
    # the flag on the command line (no driver in between) writes no text at all
    d2 = os.path.join(d, "cli")
    os.makedirs(os.path.join(d2, "temp")); os.makedirs(os.path.join(d2, "stat"))
    subprocess.check_call([NEW, os.path.join(fx, "ref"), os.path.join(d2, "temp", "s"), os.path.join(d2, "stat", "s"), os.path.join(fx, "aln.sam"), "1", "-q", "--binary"])
    assert sorted(os.listdir(os.path.join(d2, "temp"))) == ["s.omit", "s.rsb"]
    subprocess.check_call(cmd, env=dict(os.environ, RSEM_HIP_BINARY="both"))
    assert os.path.exists(os.path.join(d, "temp", "s.rsb", "hdr")) and os.path.exists(os.path.join(d, "temp", "s.dat"))
    # a later TEXT run on the same sample.temp (kept intermediate files) removes the arrays: rsem-run-em must not find stale ones
    subprocess.check_call(cmd)
    assert os.path.exists(os.path.join(d, "temp", "s.dat")) and not os.path.exists(os.path.join(d, "temp", "s.rsb"))


def test_text_parsers_are_independent_of_the_thread_count(tmp_path):
    """rsem-run-em's parsers of imd.dat and the read files (host/reads.hpp: three scans, every thread a run of whole records
    written straight into place) on files with CRLF line ends, a last line without a newline, stray empty lines, reads of very
    different lengths: 2, 3, 7 and 16 threads give the arrays one thread gives (tests/parse_chunks_check.cpp; the split
    threshold is lowered so that these small files are split at all)."""
    exe = str(tmp_path / "parse_chunks_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", os.path.join(ROOT, "tests", "parse_chunks_check.cpp"), "-o", exe])
    d = tmp_path / "files"
    d.mkdir()
    r = subprocess.run([exe, str(d)], env=dict(os.environ, RSEM_HIP_PARSE_SPLIT_BYTES="64"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:]
