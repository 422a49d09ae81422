import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_CLI = os.environ.get("MP_ORACLE_CLI") or os.path.join(ROOT, "oracle", "_build", "oracle_cli")   # (tools/run_sanitized.sh: the sanitizer build)

# (test dir, bam, vcf, gtf, mini fasta, expected stem) - the reference's live somatic tests (tests/lib.rs:211-342)
SOMATIC_FIXTURES = {
    "test_forward": ("test_forward", "forward_test.bam", "forward_test.vcf", "forward_test.gtf", "chr14.mini.fa", "forward_test"),
    "test_reverse": ("test_reverse", "reverse_test.bam", "reverse_test.vcf", "reverse_test.gtf", "chr1.mini.fa", "reverse_test"),
    "splice_forward_test": ("splice_forward_test", "INSIG1.test.bam", "INSIG1.test.vcf", "INSIG1.test.gtf", "chr7.mini.fa", "splice_forward_test"),
    "splice_reverse_test": ("splice_reverse_test", "MMS22L.test.bam", "MMS22L.test.vcf", "MMS22L.test.gtf", "chr6.mini.fa", "splice_reverse_test"),
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def fixture_paths(name):
    d, bam, vcf, gtf, fa, stem = SOMATIC_FIXTURES[name]
    base = os.path.join(GOLDEN, d)
    return dict(bam=os.path.join(base, bam), vcf=os.path.join(base, vcf), gtf=os.path.join(base, gtf),
                fasta=os.path.join(base, fa), expected=os.path.join(base, "expected_output", stem))


def read_expected(prefix):
    out = {}
    for ext in ("fa", "normal.fa", "tsv"):
        with open(prefix + "." + ext, "rb") as f:
            out[ext] = f.read()
    return out


@pytest.fixture(scope="session")
def built():
    """Build the oracle (gcc) and the HIP library (hipcc cross-compiles without a GPU) once per session."""
    import __graft_entry__ as g
    g.build()
    return True


def run_oracle_files(paths, tmp, window_len=27, extra=()):
    """Run the CPU oracle CLI on BAM/VCF/GTF/FASTA files; returns dict(fa, normal.fa, tsv) bytes."""
    tsv = os.path.join(tmp, "o.tsv")
    nfa = os.path.join(tmp, "o.normal.fa")
    with open(paths["gtf"], "rb") as gtf:
        r = subprocess.run([ORACLE_CLI, "somatic", paths["bam"], "--variants", paths["vcf"], "--ref", paths["fasta"],
                            "--tsv", tsv, "--normal-output", nfa, "-w", str(window_len), *extra], stdin=gtf, capture_output=True)
    if r.returncode != 0:
        raise RuntimeError("oracle failed: " + r.stderr.decode())
    return {"fa": r.stdout, "normal.fa": open(nfa, "rb").read(), "tsv": open(tsv, "rb").read()}


# `microphaser normal` fixtures (reference tests/lib.rs:237-249, :273-285): bam, vcf, gtf, fasta, expected FASTA
NORMAL_FIXTURES = {
    "test_forward": ("forward_test.bam", "forward_test.germline.vcf", "forward_test.gtf", "chr14.mini.fa", "forward_test.germline.fa"),
    "splice_forward_test": ("INSIG1.test.bam", "INSIG1.test.germline.vcf", "INSIG1.test.gtf", "chr7.mini.fa", "splice_forward_test.germline.fa"),
}


# --- reference fixtures whose upstream tests are DISABLED (tests/lib.rs:309-320, :384-408) ------------------------------
def fasta_records(data):
    """[(id, sequence)] of a one-line-per-record FASTA (bytes)."""
    lines = data.decode().split("\n")
    return [(lines[i][1:], lines[i + 1]) for i in range(0, len(lines) - 1, 2)]


REVERSE_GERMLINE = dict(
    dir=os.path.join(GOLDEN, "test_reverse"), bam="reverse_test.bam", vcf="reverse_test.germline.vcf", fasta="chr1.mini.fa",
    gtf="reverse_test_germline.gtf",                    # the GTF the disabled upstream test uses
    last_exon_gtf="reverse_test_germline.last_exon.gtf",  # derived by make_golden.py: the last CDS exon alone, same frame
    expected="expected_output/reverse_test.germline.fa")


def check_reverse_germline(whole_fa, last_exon_fa):
    """`normal` on the '-' strand against the stale upstream expectation (180 records made by an older revision):

    whole transcript: the current source (src/normal_microphasing.rs:493-507, :1119-1134) skips a window whose last codon is a
    reverse-strand stop and then closes the main ORF, so it ends after 61 records (the 62nd expected record ends in TCA); those 61
    are the first 61 expected ones, ids included, except the two windows that touch an exon end (records 14, 15): there the
    current source hashes the sequence INCLUDING the rest / splice-gap bases (:509-517, the very rule the live forward fixture
    pins with its 351st record) while the expectation hashes the 27 printed bases.
    last exon alone (derived GTF): the two germline SNVs of the fixture lie in that exon, which the whole-transcript run no
    longer reaches; 49 expected records (2-3 haplotypes per window where the SNVs are, soft-masked reference) byte for byte."""
    exp = fasta_records(open(os.path.join(REVERSE_GERMLINE["dir"], REVERSE_GERMLINE["expected"]), "rb").read())
    got = fasta_records(whole_fa)
    assert len(exp) == 180 and len(got) == 61
    assert [s for _, s in got] == [s for _, s in exp[:61]]
    assert [k for k in range(61) if got[k][0] != exp[k][0]] == [14, 15]
    assert exp[61][1].endswith("TCA")
    tail = fasta_records(last_exon_fa)
    assert tail[:49] == exp[129:178]
    assert sum(1 for k in range(48) if tail[k][1].lower() != tail[k][1] and "A" in tail[k][1]) >= 10  # the germline alt (upper-case A in soft-masked reference)


# frameshift_test / three_way_splice: expectations in the OLD 20-column TSV layout (no `frame` column, 60-nt windows) -> report only
DISABLED_SOMATIC = {
    "frameshift_test": dict(bam="frameshift_test.bam", vcf="frameshift_test.vcf", gtf="frameshift_test.gtf", fasta="11.mini.fa",
                            stem="frameshift_test", window_len=60),
    "three_way_splice": dict(bam="three_way_splice.bam", vcf="three_way_splice.vcf", gtf="three_way_splice.contig19.gtf", fasta="19.mini.fa",
                             stem="three_way_splice", window_len=60),
}


def disabled_paths(name):
    fx = DISABLED_SOMATIC[name]
    d = os.path.join(GOLDEN, name)
    return dict(bam=os.path.join(d, fx["bam"]), vcf=os.path.join(d, fx["vcf"]), gtf=os.path.join(d, fx["gtf"]),
                fasta=os.path.join(d, fx["fasta"]), expected=os.path.join(d, "expected_output", fx["stem"])), fx["window_len"]


def stale_report(name, tsv_bytes):
    """Rows of the stale upstream TSV that the given TSV reproduces, by key; returns a dict of counts (printed by the tests)."""
    def rows(data):
        lines = data.decode().rstrip("\n").split("\n")
        if not lines or not lines[0]:
            return []
        hdr = lines[0].split("\t")
        return [dict(zip(hdr, l.split("\t"))) for l in lines[1:]]
    p, _w = disabled_paths(name)
    exp = rows(open(p["expected"] + ".tsv", "rb").read())
    got = rows(tsv_bytes)
    rep = {"expected_rows": len(exp), "rows": len(got)}
    for label, key in (("mutant_sequence", lambda r: r["mutant_sequence"].upper()),
                       ("mutant_sequence+freq", lambda r: (r["mutant_sequence"].upper(), r["freq"])),
                       ("mutant_sequence+freq+depth+nvar", lambda r: (r["mutant_sequence"].upper(), r["freq"], r["depth"], r["nvar"], r["nsomatic"])),
                       ("id", lambda r: r["id"])):
        have = {key(r) for r in got}
        rep["matched_by_" + label] = sum(1 for r in exp if key(r) in have)
    return rep


def tsv_rows(data):
    lines = data.decode().rstrip("\n").split("\n")
    if not lines or not lines[0]:
        return [], []
    hdr = lines[0].split("\t")
    return hdr, [dict(zip(hdr, l.split("\t"))) for l in lines[1:]]


def check_frameshift_fixture(got):
    """frameshift_test (tests/lib.rs:396-407, disabled upstream; expectation from an older revision: 0-based positions, no `frame`
    column, normal_sequence left empty). The current source ends the shifted ORF at its first stop codon (the 46th window ends
    in TGA: has_stop_codon -> remove_peptide, src/microphasing.rs:694-718), the old one went on for 353 rows. Every row the current
    semantics produce must be the expected row: id (sha1 of sequence + transcript + offset), freq (frameshift frequency algebra of
    :604-631, all rows are frame 1), depth, variant counts, aa changes and the 60-nt mutant sequence, positions shifted by the
    1-based printing; the tumor FASTA is the expectation's first 46 records."""
    p, _w = disabled_paths("frameshift_test")
    _he, exp = tsv_rows(open(p["expected"] + ".tsv", "rb").read())
    _hg, rows = tsv_rows(got["tsv"])
    assert len(exp) == 353 and len(rows) == 46
    same = ["id", "transcript", "gene_id", "gene_name", "chrom", "freq", "depth", "nvar", "nsomatic", "nvariant_sites", "nsomvariant_sites",
            "strand", "somatic_aa_change", "germline_aa_change", "mutant_sequence"]
    plus1 = lambda v: "|".join(str(int(x) + 1) for x in v.split("|")) if v else v
    for g, e in zip(rows, exp):
        assert [g[c] for c in same] == [e[c] for c in same]
        assert g["frame"] == "1"
        for c in ("offset", "variant_sites", "somatic_positions", "germline_positions"):
            assert g[c] == plus1(e[c])
    assert rows[-1]["mutant_sequence"].endswith("TGA")   # the stop that closes the shifted ORF: emitted (frame > 0), nothing after it
    exp_fa = open(p["expected"] + ".mt.fa", "rb").read().split(b"\n")
    assert got["fa"] == b"\n".join(exp_fa[:92]) + b"\n"
