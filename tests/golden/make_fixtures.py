#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/<name>/ with the UNMODIFIED reference programs.

TEST INFRASTRUCTURE.  Run in the build container only (needs oracle/_ref, i.e. /root/reference):

    make -C oracle ref && python tests/golden/make_fixtures.py

Pipeline per fixture (nothing here is copied from the reference):
  1. a tiny random transcriptome with multi-isoform genes  -> rsem-synthesis-reference-transcripts,
     rsem-preref  (what rsem-prepare-reference runs, rsem-prepare-reference:151-164)
  2. a hand-written simulation model (model_file_description.txt) + TPM table
     -> rsem-simulate-reads --seed
  3. ground-truth SAM: every read is "aligned" to its true origin and to every isoform of the same
     gene containing the identical reference substring (names carry rid_dir_sid_pos[_insertL],
     simulation.cpp:158-165) -> rsem-parse-alignments, rsem-build-read-index
  4. golden outputs: rsem-run-em ... --gibbs-out  and  rsem-run-gibbs ... --seed
The committed files are the reference's inputs (.temp/.stat/ref files) and outputs
(.theta .model .ofg .iso_res .gene_res .countvectors*).
"""
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFBIN = os.path.join(REPO, "oracle", "_ref")

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def rc(s):
    return "".join(COMP[c] for c in reversed(s))


def run(cmd, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("command failed: " + " ".join(cmd))
    return r.stdout


def make_transcriptome(rng, n_genes, max_iso):
    """genes = ordered exon lists; isoforms = ordered exon subsets sharing sequence."""
    txs = []  # (tname, gname, seq)
    for g in range(n_genes):
        n_ex = int(rng.integers(3, 7))
        exons = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(120, 360)))) for _ in range(n_ex)]
        n_iso = int(rng.integers(1, max_iso + 1))
        seen = set()
        for k in range(n_iso):
            for _ in range(20):
                keep = rng.random(n_ex) < 0.7
                keep[0] = True if k == 0 else keep[0]
                if keep.sum() >= 2 and tuple(keep) not in seen:
                    break
            else:
                continue
            seen.add(tuple(keep))
            seq = "".join(e for e, kp in zip(exons, keep) if kp)
            txs.append(("g%02d_t%d" % (g, k), "g%02d" % g, seq))
    return txs


def fmt(v):
    return " ".join("%.10g" % x for x in v)


def write_sim_model(path, model_type, rng, read_len_lo, read_len_hi, frag_lo=120, frag_hi=320, probF=0.5):
    """A plausible hand-written model in the reference's .model layout (model_file_description.txt)."""
    out = [str(model_type), "", "%.10g" % probF, ""]
    if model_type < 2:
        span = read_len_hi - (read_len_lo - 1)
        p = np.linspace(1.0, 3.0, span)
        p /= p.sum()
        out += ["%d %d %d" % (read_len_lo - 1, read_len_hi, span), fmt(p), "", "0", ""]
    else:
        span = frag_hi - (frag_lo - 1)
        x = np.arange(frag_lo, frag_hi + 1)
        p = np.exp(-0.5 * ((x - 0.5 * (frag_lo + frag_hi)) / 35.0) ** 2)
        p /= p.sum()
        out += ["%d %d %d" % (frag_lo - 1, frag_hi, span), fmt(p), ""]
        mspan = read_len_hi - (read_len_lo - 1)
        mp = np.linspace(1.0, 2.0, mspan)
        mp /= mp.sum()
        out += ["%d %d %d" % (read_len_lo - 1, read_len_hi, mspan), fmt(mp), ""]
    out += ["0", ""]  # RSPD: uniform
    if model_type in (1, 3):
        size = 100
        init = np.zeros(size)
        init[25:41] = np.linspace(1, 4, 16)
        init /= init.sum()
        tran = np.zeros((size, size))
        for a in range(size):
            lo, hi = max(2, a - 6), min(40, a + 3)
            if a < 2 or a > 40:
                lo, hi = 20, 40
            w = np.ones(hi - lo + 1)
            w[-min(4, len(w)):] += 2.0
            tran[a, lo:hi + 1] = w / w.sum()
        out += [str(size), fmt(init)] + [fmt(tran[a]) for a in range(size)] + [""]
        out += ["%d 5" % size]
        for q in range(size):
            e = min(0.75, 10 ** (-q / 10.0))
            pn = 1e-4
            for r in range(5):
                row = np.zeros(5)
                if r < 4:
                    row[:4] = e / 3.0 * (1 - pn)
                    row[r] = (1 - e) * (1 - pn)
                    row[4] = pn
                else:
                    row[:4] = (1 - pn) / 4
                    row[4] = pn
                out.append(fmt(row))
            if q < size - 1:
                out.append("")
        out += ["", "%d 5" % size]
        for q in range(size):
            out.append(fmt([0.28, 0.22, 0.22, 0.2799, 0.0001]))
    else:
        L = read_len_hi
        out += ["%d 5" % L]
        for i in range(L):
            e = 0.004 + 0.03 * i / L
            pn = 1e-4
            for r in range(5):
                row = np.zeros(5)
                if r < 4:
                    row[:4] = e / 3.0 * (1 - pn)
                    row[r] = (1 - e) * (1 - pn)
                    row[4] = pn
                else:
                    row[:4] = (1 - pn) / 4
                    row[4] = pn
                out.append(fmt(row))
            if i < L - 1:
                out.append("")
        out += ["", "5", fmt([0.28, 0.22, 0.22, 0.2799, 0.0001])]
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")


def read_fasta_or_fastq(path, fastq):
    recs = []
    with open(path) as f:
        lines = f.read().split("\n")
    step = 4 if fastq else 2
    for i in range(0, len(lines) - 1, step):
        name = lines[i][1:]
        seq = lines[i + 1]
        qual = lines[i + 3] if fastq else None
        recs.append((name, seq, qual))
    return recs


def find_all(hay, needle):
    res, i = [], hay.find(needle)
    while i >= 0:
        res.append(i)
        i = hay.find(needle, i + 1)
    return res


def make_sam(path, model_type, txs, sim_prefix, omit_last=0):
    """txs: list of (name, gene, fullseq_incl_polyA) in reference (internal id) order.
    omit_last: leave the last transcripts out of the @SQ header (-> imd.omit, Transcripts.h:135-142); alignments to them are dropped."""
    n_keep = len(txs) - omit_last
    fastq = model_type in (1, 3)
    paired = model_type >= 2
    ext = "fq" if fastq else "fa"
    genes = {}
    for i, (_, g, _) in enumerate(txs):
        genes.setdefault(g, []).append(i)
    with open(path, "w") as sam:
        sam.write("@HD\tVN:1.0\tSO:unsorted\n")
        for name, _, seq in txs[:n_keep]:
            sam.write("@SQ\tSN:%s\tLN:%d\n" % (name, len(seq)))
        if not paired:
            reads = read_fasta_or_fastq("%s.%s" % (sim_prefix, ext), fastq)
            for name, seq, qual in reads:
                rid, d, sid, pos = [int(x) for x in name.split("_")]
                q = qual if fastq else "*"
                if sid == 0:
                    sam.write("%s\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n" % (name, seq, q))
                    continue
                L = len(seq)
                tseq = txs[sid - 1][2]
                fpos = pos if d == 0 else len(tseq) - pos - L
                sub = tseq[fpos:fpos + L]
                hits = []
                for j in genes[txs[sid - 1][1]]:
                    for p in find_all(txs[j][2], sub):
                        hits.append((j, p))
                assert (sid - 1, fpos) in hits
                hits = [(j, p) for j, p in hits if j < n_keep]
                if not hits:
                    sam.write("%s\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n" % (name, seq, q))
                for j, p in hits:
                    if d == 0:
                        sam.write("%s\t0\t%s\t%d\t255\t%dM\t*\t0\t0\t%s\t%s\n" % (name, txs[j][0], p + 1, L, seq, q))
                    else:
                        sam.write("%s\t16\t%s\t%d\t255\t%dM\t*\t0\t0\t%s\t%s\n" % (
                            name, txs[j][0], p + 1, L, rc(seq), q[::-1] if fastq else "*"))
        else:
            r1 = read_fasta_or_fastq("%s_1.%s" % (sim_prefix, ext), fastq)
            r2 = read_fasta_or_fastq("%s_2.%s" % (sim_prefix, ext), fastq)
            for (n1, s1, q1), (n2, s2, q2) in zip(r1, r2):
                name = n1[:-2]
                rid, d, sid, pos, ins = [int(x) for x in name.split("_")]
                qq1 = q1 if fastq else "*"
                qq2 = q2 if fastq else "*"
                if sid == 0:
                    sam.write("%s\t77\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n" % (name, s1, qq1))
                    sam.write("%s\t141\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n" % (name, s2, qq2))
                    continue
                tseq = txs[sid - 1][2]
                fpos = pos if d == 0 else len(tseq) - pos - ins
                sub = tseq[fpos:fpos + ins]
                hits = []
                for j in genes[txs[sid - 1][1]]:
                    for p in find_all(txs[j][2], sub):
                        hits.append((j, p))
                assert (sid - 1, fpos) in hits
                L1, L2 = len(s1), len(s2)
                for j, p in hits:
                    tn = txs[j][0]
                    if d == 0:  # mate1 forward at p, mate2 reverse ending at p+ins
                        p2 = p + ins - L2
                        sam.write("%s\t99\t%s\t%d\t255\t%dM\t=\t%d\t%d\t%s\t%s\n" % (name, tn, p + 1, L1, p2 + 1, ins, s1, qq1))
                        sam.write("%s\t147\t%s\t%d\t255\t%dM\t=\t%d\t%d\t%s\t%s\n" % (
                            name, tn, p2 + 1, L2, p + 1, -ins, rc(s2), qq2[::-1] if fastq else "*"))
                    else:  # mate1 reverse ending at p+ins, mate2 forward at p
                        p1 = p + ins - L1
                        sam.write("%s\t83\t%s\t%d\t255\t%dM\t=\t%d\t%d\t%s\t%s\n" % (
                            name, tn, p1 + 1, L1, p + 1, -ins, rc(s1), qq1[::-1] if fastq else "*"))
                        sam.write("%s\t163\t%s\t%d\t255\t%dM\t=\t%d\t%d\t%s\t%s\n" % (name, tn, p + 1, L2, p1 + 1, ins, s2, qq2))


def load_ref_seq(path):
    """ref.seq (RefSeq.h:108-138): 4 lines per transcript: 'fullLen totLen' / name / seq / mask words."""
    out = []
    with open(path) as f:
        lines = f.read().split("\n")
    for i in range(0, len(lines) - 3, 4):
        out.append((lines[i + 1], lines[i + 2]))
    return out


def build_fixture(name, model_type, n_reads, seed, n_genes=14, max_iso=4, polyA=False, estRSPD=0,
                  read_len=(36, 50), gibbs=(20, 40, 1), gibbs_threads=2, theta0=0.06, probF=0.5, frag_mean=None,
                  omit_last=0, pseudo_count=None, allele=False, bam=False):
    rng = np.random.default_rng(seed)
    out = os.path.join(HERE, name)
    work = os.path.join("/tmp", "rsem_fixture_" + name)
    shutil.rmtree(out, ignore_errors=True)
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    os.makedirs(os.path.join(out, "temp"))
    os.makedirs(os.path.join(out, "stat"))

    txs = make_transcriptome(rng, n_genes, max_iso)
    if allele:  # two alleles per isoform differing by a few SNPs (allele-specific reference: ref.ta / ref.gt)
        al = []
        for t, g, s in txs:
            b = list(s)
            for p in rng.choice(len(b), size=max(2, len(b) // 150), replace=False):
                b[p] = "ACGT"[("ACGT".index(b[p]) + 1 + int(rng.integers(0, 3))) % 4]
            al.append((t + "_a", g, s, t))
            al.append((t + "_b", g, "".join(b), t))
        with open(os.path.join(work, "amap.txt"), "w") as f:
            for an, g, s, t in al:
                f.write("%s\t%s\t%s\n" % (g, t, an))
        txs = [(an, g, s) for an, g, s, t in al]
    with open(os.path.join(work, "tx.fa"), "w") as f:
        for t, g, s in txs:
            f.write(">%s\n%s\n" % (t, s))
    with open(os.path.join(work, "t2g.txt"), "w") as f:
        for t, g, s in txs:
            f.write("%s\t%s\n" % (g, t))
    ref = os.path.join(work, "ref")
    if allele:
        run([os.path.join(REFBIN, "rsem-synthesis-reference-transcripts"), ref, "1", "2",
             os.path.join(work, "amap.txt"), os.path.join(work, "tx.fa")])
    else:
        run([os.path.join(REFBIN, "rsem-synthesis-reference-transcripts"), ref, "1", "1",
             os.path.join(work, "t2g.txt"), os.path.join(work, "tx.fa")])
    cmd = [os.path.join(REFBIN, "rsem-preref"), ref + ".transcripts.fa", "0" if polyA else "1", ref, "-q"]
    if polyA:
        cmd += ["-l", "125"]
    run(cmd)
    refseqs = load_ref_seq(ref + ".seq")
    t2g = {t: g for t, g, _ in txs}
    M = len(refseqs)
    ordered = [(n, t2g[n], s) for n, s in refseqs]

    sim_model = os.path.join(work, "sim.model")
    write_sim_model(sim_model, model_type, rng, read_len[0], read_len[1], probF=probF)
    tpm = np.exp(rng.normal(0, 2, M))
    tpm[rng.random(M) < 0.3] = 0.0
    tpm = tpm / tpm.sum() * 1e6
    with open(os.path.join(work, "sim.isoforms.results"), "w") as f:
        if allele:  # alleles.results layout: TPM is the 7th column (simulation.cpp:189 OFFSITE = 6)
            f.write("allele_id\ttranscript_id\tgene_id\tlength\teffective_length\texpected_count\tTPM\tFPKM\tAlleleIsoPct\tAlleleGenePct\n")
            for (n, g, s), v in zip(ordered, tpm):
                f.write("%s\t%s\t%s\t%d\t0\t0\t%.4f\t0\t0\t0\n" % (n, n[:-2], g, len(s), v))
        else:
            f.write("transcript_id\tgene_id\tlength\teffective_length\texpected_count\tTPM\tFPKM\tIsoPct\n")
            for (n, g, s), v in zip(ordered, tpm):
                f.write("%s\t%s\t%d\t0\t0\t%.4f\t0\t0\n" % (n, g, len(s), v))
    sim = os.path.join(work, "sim")
    run([os.path.join(REFBIN, "rsem-simulate-reads"), ref, sim_model, os.path.join(work, "sim.isoforms.results"),
         str(theta0), str(n_reads), sim, "--seed", str(seed), "-q"])
    samf = os.path.join(work, "x.sam")
    make_sam(samf, model_type, ordered, sim, omit_last=omit_last)

    imd = os.path.join(out, "temp", "s")
    stat = os.path.join(out, "stat", "s")
    for ext in ("seq", "ti", "grp") + (("ta", "gt") if allele else ()):
        shutil.copy(ref + "." + ext, os.path.join(out, "ref." + ext))
    oref = os.path.join(out, "ref")
    run([os.path.join(REFBIN, "rsem-parse-alignments"), oref, imd, stat, samf, str(model_type), "-q"])
    fastq = "1" if model_type in (1, 3) else "0"
    ext = "fq" if model_type in (1, 3) else "fa"
    if model_type < 2:
        idx = [imd + "_alignable." + ext]
    else:
        idx = [imd + "_alignable_1." + ext, imd + "_alignable_2." + ext]
    run([os.path.join(REFBIN, "rsem-build-read-index"), "32", fastq, "1"] + idx)
    with open(imd + ".mparams", "w") as f:
        f.write("1 1000\n%g\n%d\n20\n1 1000\n%s\n25\n" % (probF, estRSPD, "%g %g" % frag_mean if frag_mean else "-1 0"))

    bam_args = []
    if bam:  # keep the alignments: golden <sample>.transcript.bam from the reference's BamWriter (SAM in, then BAM in + sampling)
        shutil.copy(samf, os.path.join(out, "aln.sam"))
        bam_args = ["-b", os.path.join(out, "aln.sam"), "0"]
    log = run([os.path.join(REFBIN, "rsem-run-em"), oref, str(model_type), os.path.join(out, "s"), imd, stat,
               "-p", "1", "--gibbs-out"] + bam_args)
    if bam:
        os.rename(os.path.join(out, "s.transcript.bam"), os.path.join(out, "golden.transcript.bam"))
        run([os.path.join(REFBIN, "rsem-run-em"), oref, str(model_type), os.path.join(out, "s"), imd, stat, "-p", "1", "--gibbs-out",
             "-b", os.path.join(out, "golden.transcript.bam"), "0", "--sampling", "--seed", "77"])
        os.rename(os.path.join(out, "s.transcript.bam"), os.path.join(out, "golden.sampled.transcript.bam"))
    with open(os.path.join(out, "em.log"), "w") as f:
        f.write("\n".join(l for l in log.split("\n") if l.startswith("ROUND")) + "\n")
    # keep pre-Gibbs result files
    shutil.copy(imd + ".iso_res", imd + ".iso_res.em")
    shutil.copy(imd + ".gene_res", imd + ".gene_res.em")
    if allele:
        shutil.copy(imd + ".allele_res", imd + ".allele_res.em")
    b, n, g = gibbs
    run([os.path.join(REFBIN, "rsem-run-gibbs"), oref, imd, stat, str(b), str(n), str(g),
         "-p", str(gibbs_threads), "--seed", "12345", "-q"] + (["--pseudo-count", str(pseudo_count)] if pseudo_count else []))
    # the read index is only needed by the reference binary; regenerated on demand by tests
    for p in idx:
        os.remove(p + ".ridx")
    with open(os.path.join(out, "META"), "w") as f:
        f.write("model_type %d\nM %d\nn_reads %d\nseed %d\npolyA %d\nestRSPD %d\ngibbs %d %d %d\ngibbs_threads %d\ngibbs_seed 12345\n"
                % (model_type, M, n_reads, seed, polyA, estRSPD, b, n, g, gibbs_threads))
        f.write("pseudo_count_x1000 %d\n" % int(round((pseudo_count or 1.0) * 1000)))
    shutil.rmtree(work, ignore_errors=True)
    sz = sum(os.path.getsize(os.path.join(dp, fn)) for dp, _, fns in os.walk(out) for fn in fns)
    print("fixture %-10s type %d M=%d size=%.1f KB" % (name, model_type, M, sz / 1024))


if __name__ == "__main__":
    only = set(sys.argv[1:])
    specs = [
        dict(name="se_noq", model_type=0, n_reads=1500, seed=11),
        dict(name="se_q", model_type=1, n_reads=1500, seed=12, bam=True),
        dict(name="se_q_polya_rspd", model_type=1, n_reads=1500, seed=15, polyA=True, estRSPD=1),
        dict(name="pe_noq", model_type=2, n_reads=1200, seed=13),
        dict(name="pe_q", model_type=3, n_reads=1200, seed=14, bam=True),
        dict(name="pe_q_polya_rspd", model_type=3, n_reads=1200, seed=16, polyA=True, estRSPD=1),
        # --fragment-length-mean/sd with single-end reads (mld != NULL paths, LenDist::setAsNormal)
        dict(name="se_q_fragmean", model_type=1, n_reads=1200, seed=17, frag_mean=(140, 25)),
        # reverse-stranded protocol + RSPD (the probF < 0.1 && dir == 1 branch of update), transcripts missing from
        # the alignment header (imd.omit -> counts = -1 in Gibbs), single-cell pseudo count
        dict(name="se_q_allele", model_type=1, n_reads=1500, seed=19, allele=True, n_genes=8, max_iso=3),
        dict(name="se_noq_rev_rspd_omit", model_type=0, n_reads=1200, seed=18, probF=0.0, estRSPD=1, omit_last=3, pseudo_count=0.1),
    ]
    for sp in specs:
        if not only or sp["name"] in only:
            build_fixture(**sp)
