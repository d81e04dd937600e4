"""The four drop-ins inside the reference's own pipeline on a GPU box: rsem-calculate-expression (Perl, unmodified) --calc-pme
with rsem-parse-alignments, rsem-run-em, rsem-run-gibbs (exact mode = the reference's chains) from this repo, against the
untouched pipeline on the same SAM file: expected counts / TPM / FPKM to the printed 0.01, posterior mean counts likewise
(the exact sampler draws the reference's chains)."""
import os

import numpy as np
import pytest

import pipeline_util as pu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not pu.available(), reason="needs perl and oracle/_ref")]


@pytest.mark.parametrize("handoff", ["text", "binary"])
@pytest.mark.parametrize("fixture", sorted(pu.FIXTURES_WITH_SAM))
def test_all_drop_ins_inside_the_perl_pipeline(fixture, handoff, tmp_path):
    """handoff = binary: RSEM_HIP_BINARY=1 in the driver's environment -- the parser hands arrays to rsem-run-em
    (imdName.rsb/, alignable read files kept for the driver's rsem-build-read-index step) and rsem-run-em hands arrays to
    rsem-run-gibbs (imdName.ofb/): same results."""
    extra = ["--calc-pme", "--gibbs-burnin", "20", "--gibbs-number-of-samples", "40"]
    ref, _ = pu.run_pipeline(str(tmp_path), "ref", fixture, [], extra)
    mine, log = pu.run_pipeline(str(tmp_path), "dropin", fixture, ["rsem-parse-alignments", "rsem-run-em", "rsem-run-gibbs"], extra + ["--keep-intermediate-files"],
                                env={"RSEM_HIP_BINARY": "1"} if handoff == "binary" else None)
    if handoff == "binary":
        assert os.path.exists(os.path.join(mine, "s.temp", "s.rsb", "hdr")) and os.path.exists(os.path.join(mine, "s.temp", "s.ofb", "hdr"))
        assert not os.path.exists(os.path.join(mine, "s.temp", "s.dat")) and not os.path.exists(os.path.join(mine, "s.temp", "s.ofg"))
    for f in ("s.isoforms.results", "s.genes.results"):
        ha, ra = pu.read_results(os.path.join(ref, f))
        hb, rb = pu.read_results(os.path.join(mine, f))
        assert ha == hb and len(ra) == len(rb)
        for a, b in zip(ra, rb):
            assert a[:2] == b[:2]
            va, vb = np.array(a[2:], float), np.array(b[2:], float)
            assert np.allclose(va, vb, rtol=1e-6, atol=0.011), (f, a, b)
