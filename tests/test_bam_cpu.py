"""The -b pass of rsem-run-em (rsem_amd/csrc/host/bam_io.hpp: SAM / BAM input -> <sample>.transcript.bam with MAPQ and ZW:f set from
the posterior weights, BamWriter.h:39-146) without a GPU: tests/bam_write_check.cpp takes the weights from the ZW:f tags of the
transcript.bam the REFERENCE wrote for the fixture and runs the writer on the fixture's alignment file.  The records must be the
reference's (same bytes; MAPQ may differ by one where it is recomputed from a float), and the decompressed output must be the same
stream of bytes for every thread count, every super-chunk size (down to chunks shorter than a line, so that pieces, mate pairs and
records straddle every kind of boundary) and for SAM and BAM input alike."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CXX = shutil.which("g++")
pytestmark = pytest.mark.skipif(CXX is None, reason="needs g++")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = os.path.join(str(tmp_path_factory.mktemp("bam_write_check")), "bam_write_check")
    subprocess.check_call([CXX, "-O2", "-std=c++17", os.path.join(ROOT, "tests", "bam_write_check.cpp"), "-o", exe, "-lpthread", "-lz"])
    return exe


def _run(exe, fx, inp, paired, threads, chunk, out):
    env = dict(os.environ)
    env.pop("BAM_CHECK_REPEAT", None)
    if chunk:
        env["RSEM_HIP_BAM_CHUNK"] = str(chunk)
    else:
        env.pop("RSEM_HIP_BAM_CHUNK", None)
    g = os.path.join(ROOT, "tests", "golden", fx)
    r = subprocess.run([exe, os.path.join(g, "ref.ti"), os.path.join(g, inp), os.path.join(g, "golden.transcript.bam"), out, str(int(paired)), str(threads)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    import re
    m = re.search(r"(\d+) guessed segments taken, (\d+) stretches walked again", r.stdout)
    text = "\n".join(l for l in r.stdout.split("\n") if not l.startswith("[timing]"))
    f = dict(zip(text.split()[0::2], text.split()[1::2]))
    if m:
        f["_taken"], f["_again"] = m.group(1), m.group(2)
    m = re.search(r"blocks inflated by inflate_fast.hpp (\d+), by zlib (\d+)", r.stdout)
    if m:
        f["_fast_blocks"], f["_zlib_blocks"] = m.group(1), m.group(2)
    assert f["header_equal"] == "1" and int(f["identical"]) + int(f["mapq_off_by_one"]) == int(f["records"])
    return f


@pytest.mark.parametrize("fx,paired", [("se_q", False), ("pe_q", True)])
def test_transcript_bam_is_the_same_for_every_thread_count_and_chunk_size(checker, fx, paired, tmp_path):
    out = os.path.join(str(tmp_path), "o.bam")
    base = _run(checker, fx, "aln.sam", paired, 1, 0, out)
    assert int(base["records"]) > 3000
    for inp in ("aln.sam", "golden.transcript.bam"):
        for threads, chunk in [(3, 0), (8, 0), (8, 5000), (5, 777), (6, 200), (4, 64)]:
            f = _run(checker, fx, inp, paired, threads, chunk, out)
            assert (f["fnv"], f["stream_bytes"], f["records"]) == (base["fnv"], base["stream_bytes"], base["records"]), (inp, threads, chunk)


@pytest.mark.parametrize("fx,paired", [("se_q", False), ("pe_q", True)])
def test_bam_input_framed_in_segments_from_guessed_record_starts(checker, fx, paired, tmp_path, monkeypatch):
    """BAM input: a super-chunk's records are found by several walks at once, all but the first from a GUESSED record start
    (bam_io.hpp, frame_chunk).  Segments of a few hundred bytes (every guess lands in another record), guesses forced wrong
    (one record late / none: the stretch is walked again from the verified position) -- the same stream of bytes every time."""
    out = os.path.join(str(tmp_path), "o.bam")
    base = _run(checker, fx, "golden.transcript.bam", paired, 1, 0, out)
    for seg, bad, threads, chunk in [(300, False, 8, 0), (1000, False, 16, 0), (5000, False, 64, 0), (300, False, 8, 20000), (700, True, 8, 0),
                                     (4000, True, 32, 50000), (1 << 22, True, 8, 0)]:
        monkeypatch.setenv("RSEM_HIP_BAM_FRAME_SEG", str(seg))
        monkeypatch.setenv("RSEM_HIP_BAM_FRAME_THREADS", str(threads))
        monkeypatch.setenv("RSEM_HIP_TIMING", "1")
        if bad:
            monkeypatch.setenv("RSEM_HIP_BAM_BAD_GUESS", "1")
        else:
            monkeypatch.delenv("RSEM_HIP_BAM_BAD_GUESS", raising=False)
        f = _run(checker, fx, "golden.transcript.bam", paired, threads, chunk, out)
        assert (f["fnv"], f["stream_bytes"], f["records"]) == (base["fnv"], base["stream_bytes"], base["records"]), (seg, bad, threads, chunk)
        taken, again = int(f["_taken"]), int(f["_again"])
        assert (again > 0 and taken == 0) if bad else (taken > 0 and again == 0), (seg, bad, threads, chunk, taken, again)


def test_bam_input_inflated_by_the_repositorys_decoder_or_by_zlib(checker, tmp_path, monkeypatch):
    """BAM input: every BGZF block is inflated by host/inflate_fast.hpp and believed only if the block's CRC-32 agrees; RSEM_HIP_INFLATE_ZLIB
    sends every block to zlib.  Same output stream either way; the timing line says who inflated how many."""
    out = os.path.join(str(tmp_path), "o.bam")
    monkeypatch.setenv("RSEM_HIP_TIMING", "1")
    for fx, paired in (("se_q", False), ("pe_q", True)):
        a = _run(checker, fx, "golden.transcript.bam", paired, 4, 0, out)
        assert int(a["_fast_blocks"]) > 0 and int(a["_zlib_blocks"]) == 0
        monkeypatch.setenv("RSEM_HIP_INFLATE_ZLIB", "1")
        b = _run(checker, fx, "golden.transcript.bam", paired, 4, 0, out)
        monkeypatch.delenv("RSEM_HIP_INFLATE_ZLIB")
        assert int(b["_fast_blocks"]) == 0 and int(b["_zlib_blocks"]) > 0
        assert (a["fnv"], a["stream_bytes"], a["records"]) == (b["fnv"], b["stream_bytes"], b["records"])


def test_bam_input_with_a_wrong_block_checksum_is_refused(checker, tmp_path):
    """A BGZF block whose CRC-32 does not match its data: the repository's decoder is not believed, zlib inflates the block, and the checksum
    is compared again -- the pass ends with an error like htslib's reader (until round 6 the input's checksums were not looked at)."""
    import struct
    g = os.path.join(ROOT, "tests", "golden", "pe_q")
    d = bytearray(open(os.path.join(g, "golden.transcript.bam"), "rb").read())
    blocks, i = [], 0
    while i < len(d):
        bs = struct.unpack("<H", d[i + 16:i + 18])[0] + 1
        blocks.append((i, bs))
        i += bs
    assert len(blocks) >= 3
    off, bs = blocks[-2]  # the last block that holds records (the file ends with the empty EOF block)
    d[off + bs - 8] ^= 0x55  # its CRC-32
    bad = os.path.join(str(tmp_path), "bad.bam")
    open(bad, "wb").write(d)
    r = subprocess.run([checker, os.path.join(g, "ref.ti"), bad, os.path.join(g, "golden.transcript.bam"), os.path.join(str(tmp_path), "o.bam"), "1", "4"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode != 0 and "checksum" in r.stdout, r.stdout[-500:]
