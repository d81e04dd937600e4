"""The four drop-ins inside the reference's own pipeline on a GPU box: rsem-calculate-expression (Perl, unmodified) --calc-pme
with rsem-parse-alignments, rsem-run-em, rsem-run-gibbs (exact mode = the reference's chains) from this repo, against the
untouched pipeline on the same SAM file: expected counts / TPM / FPKM to the printed 0.01, posterior mean counts likewise
(the exact sampler draws the reference's chains)."""
import os

import numpy as np
import pytest

import pipeline_util as pu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not pu.available(), reason="needs perl and oracle/_ref")]


@pytest.mark.parametrize("fixture", sorted(pu.FIXTURES_WITH_SAM))
def test_all_drop_ins_inside_the_perl_pipeline(fixture, tmp_path):
    extra = ["--calc-pme", "--gibbs-burnin", "20", "--gibbs-number-of-samples", "40"]
    ref, _ = pu.run_pipeline(str(tmp_path), "ref", fixture, [], extra)
    mine, log = pu.run_pipeline(str(tmp_path), "dropin", fixture, ["rsem-parse-alignments", "rsem-run-em", "rsem-run-gibbs"], extra)
    for f in ("s.isoforms.results", "s.genes.results"):
        ha, ra = pu.read_results(os.path.join(ref, f))
        hb, rb = pu.read_results(os.path.join(mine, f))
        assert ha == hb and len(ra) == len(rb)
        for a, b in zip(ra, rb):
            assert a[:2] == b[:2]
            va, vb = np.array(a[2:], float), np.array(b[2:], float)
            assert np.allclose(va, vb, rtol=1e-6, atol=0.011), (f, a, b)
