"""The drop-in inside the reference's OWN pipeline, on the CPU: rsem-calculate-expression (Perl, unmodified) with this
repo's rsem-parse-alignments in place of the reference's (the stage of the path that needs no GPU), every other program
the reference's -- the result files must be the same bytes as the untouched pipeline's.  Also pins how the driver finds
its programs (its own directory first: INTEGRATION.md section A, tools/make_overlay.sh)."""
import filecmp
import os

import pytest

import pipeline_util as pu

pytestmark = pytest.mark.skipif(not pu.available(), reason="needs perl and oracle/_ref (make -C oracle ref where /root/reference exists)")


@pytest.mark.parametrize("fixture", sorted(pu.FIXTURES_WITH_SAM))
def test_parse_alignments_drop_in_inside_the_perl_pipeline(fixture, tmp_path):
    ref, ref_log = pu.run_pipeline(str(tmp_path), "ref", fixture, [])
    mix, mix_log = pu.run_pipeline(str(tmp_path), "mixed", fixture, ["rsem-parse-alignments"])
    # the driver ran the program next to it: ours in the mixed installation (it announces itself), the reference's otherwise
    assert "rsem-parse-alignments ref s.temp/s s.stat/s aln.sam" in mix_log
    assert os.path.realpath(os.path.join(str(tmp_path), "install_mixed", "rsem-parse-alignments")).startswith(os.path.join(pu.ROOT, "rsem_amd", "bin"))
    for f in ("s.isoforms.results", "s.genes.results", os.path.join("s.stat", "s.cnt"), os.path.join("s.stat", "s.theta"),
              os.path.join("s.stat", "s.model")):
        assert filecmp.cmp(os.path.join(ref, f), os.path.join(mix, f), shallow=False), f
    head, rows = pu.read_results(os.path.join(mix, "s.isoforms.results"))
    assert head[:5] == ["transcript_id", "gene_id", "length", "effective_length", "expected_count"] and len(rows) > 20
