"""rsem-calculate-expression (the reference's Perl driver, copied by oracle/Makefile into oracle/_ref next to the reference
binaries) run on a fixture's SAM file, through an installation directory made by tools/make_overlay.sh."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
FIXTURES_WITH_SAM = {"se_q": [], "pe_q": ["--paired-end"]}


def available():
    return (shutil.which("perl") is not None and os.path.exists(os.path.join(REF_DIR, "rsem-calculate-expression"))
            and os.path.exists(os.path.join(REF_DIR, "rsem-run-em")))


def run_pipeline(tmp, tag, fixture, programs, extra=(), env=None):
    """-> directory holding s.isoforms.results, s.genes.results, s.stat/.  programs: drop-ins to take from rsem_amd/bin
    ([] = the reference's pipeline untouched)."""
    ov = os.path.join(tmp, "install_" + tag)
    subprocess.check_call([os.path.join(ROOT, "tools", "make_overlay.sh"), REF_DIR, ov] + (list(programs) or ["--none"]),
                          stdout=subprocess.DEVNULL)
    work = os.path.join(tmp, "run_" + tag)
    os.makedirs(work)
    fx = os.path.join(ROOT, "tests", "golden", fixture)
    for f in os.listdir(fx):
        if f.startswith("ref.") or f == "aln.sam":
            shutil.copy(os.path.join(fx, f), work)
    cmd = [os.path.join(ov, "rsem-calculate-expression"), "--alignments"] + FIXTURES_WITH_SAM[fixture] + \
          ["-p", "2", "--no-bam-output", "--seed", "7"] + list(extra) + ["aln.sam", "ref", "s"]
    r = subprocess.run(cmd, cwd=work, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout[-3000:]
    return work, r.stdout


def read_results(path):
    rows = [l.rstrip("\n").split("\t") for l in open(path)]
    return rows[0], rows[1:]
