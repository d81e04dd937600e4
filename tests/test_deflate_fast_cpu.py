"""rsem_amd/csrc/host/deflate_fast.hpp -- the DEFLATE encoder behind the BGZF blocks of the drop-in's transcript.bam -- held to zlib's
inflate by tests/deflate_fast_check.cpp: every block it writes must inflate to its input, whole, with nothing left over.  Inputs: the
record streams of the fixtures' transcript.bam files (what the encoder is made for) and their SAM text, cut into blocks of every
size class (1 byte ... 65 280, the largest a BGZF block takes); generated blocks of nine kinds (noise, one byte, four letters,
repeated records with small changes, near-periodic, sparse, repeats beyond DEFLATE's 32 KB window, tiny alphabets, runs) with lengths
at and around the limits; the same under AddressSanitizer + UBSan.  No GPU involved."""
import glob
import os
import shutil
import struct
import subprocess
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CXX = shutil.which("g++")
pytestmark = pytest.mark.skipif(CXX is None, reason="needs g++")


def _build(tmp_path_factory, name, flags):
    exe = os.path.join(str(tmp_path_factory.mktemp(name)), name)
    subprocess.check_call([CXX, "-std=c++17"] + flags + [os.path.join(ROOT, "tests", "deflate_fast_check.cpp"), "-o", exe, "-lz"])
    return exe


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    return _build(tmp_path_factory, "deflate_fast_check", ["-O2"])


@pytest.fixture(scope="module")
def checker_san(tmp_path_factory):
    try:
        return _build(tmp_path_factory, "deflate_fast_check_san", ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"])
    except subprocess.CalledProcessError:
        pytest.skip("this g++ cannot build with -fsanitize=address,undefined")


def _bam_stream(path):
    d = open(path, "rb").read()
    out, i = bytearray(), 0
    while i < len(d):
        bs = struct.unpack("<H", d[i + 16:i + 18])[0] + 1
        out += zlib.decompress(d[i + 18:i + bs - 8], -15)
        i += bs
    return bytes(out)


def _ok(exe, *args):
    r = subprocess.run([exe] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout[-2000:]
    return r.stdout


@pytest.mark.parametrize("seed", [1, 2])
def test_generated_blocks_inflate_to_their_input(checker, seed):
    _ok(checker, "fuzz", seed, 4000)


def test_record_streams_and_text_in_blocks_of_every_size(checker, tmp_path):
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*", "golden*.transcript.bam")))
    assert len(files) >= 2
    raw = os.path.join(str(tmp_path), "stream.bin")
    smaller = 0
    for k, f in enumerate(files):
        with open(raw, "wb") as g:
            g.write(_bam_stream(f))
        for blk in (65280, 65279, 40000, 4097, 256, 255, 5, 4, 3, 2, 1)[: (11 if k == 0 else 2)]:
            out = _ok(checker, "file", raw, blk)
            if blk == 65280:  # (not a ratio test -- a sanity check that matches are found at all: half of zlib's level 6 at worst)
                mine, ref = int(out.split(" bytes in, ")[1].split()[0]), int(out.split("zlib level 6: ")[1].split()[0])
                assert mine < 1.25 * ref, out
                smaller += 1
    assert smaller == len(files)
    _ok(checker, "file", os.path.join(ROOT, "tests", "golden", "pe_q", "aln.sam"))
    _ok(checker, "file", os.path.join(ROOT, "tests", "golden", "pe_q", "aln.sam"), 1000)


def test_under_address_and_undefined_behaviour_sanitizers(checker_san, tmp_path):
    _ok(checker_san, "fuzz", 11, 1500)
    raw = os.path.join(str(tmp_path), "stream.bin")
    with open(raw, "wb") as g:
        g.write(_bam_stream(os.path.join(ROOT, "tests", "golden", "pe_q", "golden.transcript.bam")))
    _ok(checker_san, "file", raw)
    _ok(checker_san, "file", raw, 777)


def test_crc32_by_carry_less_multiplication_is_zlibs(tmp_path):
    """The BGZF blocks' checksum (rsem_amd/csrc/host/crc32_fold.hpp) against zlib's crc32: every length 0 .. 400, random triples."""
    exe = os.path.join(str(tmp_path), "crc32_fold_check")
    subprocess.check_call([CXX, "-O2", "-std=c++17", os.path.join(ROOT, "tests", "crc32_fold_check.cpp"), "-o", exe, "-lz"])
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "bad 0" in r.stdout, r.stdout


# ---- the decoder (rsem_amd/csrc/host/inflate_fast.hpp): BAM input's BGZF blocks ----------------------------------------------------------

def _build_inflate(tmp_path_factory, name, flags):
    exe = os.path.join(str(tmp_path_factory.mktemp(name)), name)
    subprocess.check_call([CXX, "-std=c++17"] + flags + [os.path.join(ROOT, "tests", "inflate_fast_check.cpp"), "-o", exe, "-lz"])
    return exe


@pytest.fixture(scope="module")
def inflate_checker(tmp_path_factory):
    return _build_inflate(tmp_path_factory, "inflate_fast_check", ["-O2"])


@pytest.mark.parametrize("seed", [1, 2])
def test_decoder_on_generated_streams_good_and_damaged(inflate_checker, seed):
    """zlib's streams at every level x strategy (default, filtered, Huffman only, RLE, fixed codes) and deflate_fast.hpp's inflate to their
    input; truncated and bit-flipped copies are refused or inflate to the right length (the caller's CRC decides) -- never a byte outside
    the output, which the checker brackets with guard bytes."""
    out = _ok(inflate_checker, "fuzz", seed, 1500)
    assert "rejected" in out


def test_decoder_on_the_fixtures_streams(inflate_checker, tmp_path):
    raw = os.path.join(str(tmp_path), "stream.bin")
    with open(raw, "wb") as g:
        g.write(_bam_stream(os.path.join(ROOT, "tests", "golden", "pe_q", "golden.transcript.bam")))
    for blk in (65280, 4097):
        _ok(inflate_checker, "file", raw, blk)
    small = os.path.join(str(tmp_path), "prefix.bin")
    with open(small, "wb") as g:
        g.write(open(raw, "rb").read()[:6000])
    for blk in (300, 17, 1):  # (51 streams per block)
        _ok(inflate_checker, "file", small, blk)
    _ok(inflate_checker, "file", os.path.join(ROOT, "tests", "golden", "pe_q", "aln.sam"))


def test_decoder_under_address_and_undefined_behaviour_sanitizers(tmp_path_factory, tmp_path):
    try:
        exe = _build_inflate(tmp_path_factory, "inflate_fast_check_san", ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"])
    except subprocess.CalledProcessError:
        pytest.skip("this g++ cannot build with -fsanitize=address,undefined")
    _ok(exe, "fuzz", 21, 600)
    raw = os.path.join(str(tmp_path), "stream.bin")
    with open(raw, "wb") as g:
        g.write(_bam_stream(os.path.join(ROOT, "tests", "golden", "se_q", "golden.transcript.bam")))
    _ok(exe, "file", raw, 20000)
