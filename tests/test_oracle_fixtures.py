"""The CPU oracle against the reference's own golden outputs (no GPU).

This is what pins the oracle: the expected .fa / .normal.fa / .tsv files are the reference's
(tests/lib.rs:106-342), the inputs are its BAM / VCF / GTF fixtures plus the mini reference FASTA
rebuilt by tests/golden/make_golden.py.
"""
import os
import subprocess

import pytest

from conftest import NORMAL_FIXTURES, GOLDEN, ORACLE_CLI, SOMATIC_FIXTURES, fixture_paths, read_expected, run_oracle_files


@pytest.mark.parametrize("name", sorted(SOMATIC_FIXTURES))
def test_oracle_matches_reference_expected_output(built, tmp_path, name):
    p = fixture_paths(name)
    got = run_oracle_files(p, str(tmp_path))
    exp = read_expected(p["expected"])
    for ext in ("fa", "normal.fa", "tsv"):
        assert got[ext] == exp[ext], "%s.%s differs from the reference's expected output" % (name, ext)


def test_oracle_empty_vcf_gives_three_empty_files(built, tmp_path):
    # tests/lib.rs:106-130 (test_empty)
    p = fixture_paths("test_forward")
    p["vcf"] = os.path.join(GOLDEN, "test_empty", "empty_test.vcf")
    got = run_oracle_files(p, str(tmp_path))
    assert got == {"fa": b"", "normal.fa": b"", "tsv": b""}


def test_oracle_unsorted_gtf_fails(built, tmp_path):
    # tests/lib.rs:344-382 (unsorted_gtf_test): unsorted must fail, sorted must succeed
    d = os.path.join(GOLDEN, "test_unsorted_gtf")
    base = [ORACLE_CLI, "somatic", os.path.join(d, "forward_test.bam"), "--variants", os.path.join(d, "empty.vcf"),
            "--ref", os.path.join(GOLDEN, "test_forward", "chr14.mini.fa"), "--tsv", str(tmp_path / "t.tsv"),
            "--normal-output", str(tmp_path / "n.fa")]
    with open(os.path.join(d, "chr14.unsorted.BDKRB2_DHRS2.gtf"), "rb") as f:
        r = subprocess.run(base, stdin=f, capture_output=True)
    assert r.returncode != 0 and b"not sorted" in r.stderr
    with open(os.path.join(d, "chr14.sorted.DHRS2_BDKRB2.gtf"), "rb") as f:
        r = subprocess.run(base, stdin=f, capture_output=True)
    assert r.returncode == 0


def test_oracle_build_reference_matches_reference_fixture(built, tmp_path):
    # tests/lib.rs:132-143 (test_build_ref): translated FASTA is diffed; the peptide set is the one the filter fixtures hold
    import microphaser_amd as m
    out = tmp_path / "ref.bin"
    r = subprocess.run([ORACLE_CLI, "build_reference", "-r", os.path.join(GOLDEN, "test_build", "reference.fa"), "-l4", "-o", str(out)],
                       capture_output=True, check=True)
    assert r.stdout == open(os.path.join(GOLDEN, "test_build", "expected_output", "reference_peptides.fasta"), "rb").read()
    want = m.decode_bincode_set(open(os.path.join(GOLDEN, "test_filter", "reference.binary"), "rb").read())
    assert m.decode_bincode_set(out.read_bytes()) == want == {b"MRRR", b"PEXD", b"LWHL", b"STDQ"}




@pytest.mark.parametrize("name", sorted(NORMAL_FIXTURES))
def test_normal_oracle_matches_reference_expected_output(built, tmp_path, name):
    bam, vcf, gtf, fa, exp = NORMAL_FIXTURES[name]
    d = os.path.join(GOLDEN, name)
    with open(os.path.join(d, gtf), "rb") as g:
        r = subprocess.run([ORACLE_CLI, "normal", os.path.join(d, bam), "--variants", os.path.join(d, vcf), "--ref", os.path.join(d, fa),
                            "--tsv", str(tmp_path / "n.tsv")], stdin=g, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == open(os.path.join(d, "expected_output", exp), "rb").read()
    assert (tmp_path / "n.tsv").read_bytes().count(b"\n") == r.stdout.count(b">") + 1


FILTER_FIXTURES = {"test_filter": "filtered", "test_filter_long": "filtered_long", "test_filter_fs": "filtered_fs"}


@pytest.mark.parametrize("name", sorted(FILTER_FIXTURES))
def test_filter_oracle_matches_reference_expected_output(built, tmp_path, name):
    """tests/lib.rs:146-211: tumor FASTA, normal FASTA and filtered TSV (incl. the credible-interval strings) byte for byte."""
    d, stem = os.path.join(GOLDEN, name), FILTER_FIXTURES[name]
    r = subprocess.run([ORACLE_CLI, "filter", "--reference", os.path.join(d, "reference.binary"), "-l", "9", "--tsv", os.path.join(d, "info.tsv"),
                        "--tsv-output", str(tmp_path / "o.tsv"), "--normal-output", str(tmp_path / "o.normal.fa"),
                        "--similar-removed", str(tmp_path / "o.removed.tsv"), "--removed-peptides", str(tmp_path / "o.removed.fa")],
                       capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    exp = os.path.join(d, "expected_output")
    assert r.stdout == open(os.path.join(exp, "tumor.%s.fa" % stem), "rb").read()
    assert (tmp_path / "o.normal.fa").read_bytes() == open(os.path.join(exp, "normal.%s.fa" % stem), "rb").read()
    assert (tmp_path / "o.tsv").read_bytes() == open(os.path.join(exp, "info.%s.tsv" % stem), "rb").read()


@pytest.mark.parametrize("encoding", ["gzip", "bgzf", "bcf_idx", "bcf_implicit"])
@pytest.mark.parametrize("name", ["test_reverse", "splice_forward_test"])
def test_variants_file_may_be_gzip_bgzf_or_bcf(built, tmp_path, name, encoding):
    """`bcf::Reader::from_path` (src/main.rs:75) takes VCF text, compressed VCF and BCF alike. The fixture VCF (multi-allelic record,
    SOMATIC flags, ANN strings) re-encoded by tests/vcf_formats.py - an independent writer from the specification - must give the
    reference's expected output unchanged. The reference ships no compressed / BCF fixture, so the BCF reader is pinned by this only."""
    import vcf_formats as vf
    p = dict(fixture_paths(name))
    text = open(p["vcf"], "rb").read()
    data = {"gzip": lambda: vf.gzip_bytes(text), "bgzf": lambda: vf.bgzf_bytes(text),
            "bcf_idx": lambda: vf.bgzf_bytes(vf.vcf_to_bcf(text, True)), "bcf_implicit": lambda: vf.bgzf_bytes(vf.vcf_to_bcf(text, False))}[encoding]()
    p["vcf"] = str(tmp_path / ("v.bcf" if encoding.startswith("bcf") else "v.vcf.gz"))
    open(p["vcf"], "wb").write(data)
    got = run_oracle_files(p, str(tmp_path))
    exp = read_expected(p["expected"])
    for ext in ("fa", "normal.fa", "tsv"):
        assert got[ext] == exp[ext], "%s.%s differs with the variants as %s" % (name, ext, encoding)


def run_oracle_normal(bam, vcf, fasta, gtf, tmp):
    with open(gtf, "rb") as g:
        r = subprocess.run([ORACLE_CLI, "normal", bam, "--variants", vcf, "--ref", fasta, "--tsv", os.path.join(tmp, "n.tsv")], stdin=g, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    return r.stdout


def test_normal_oracle_reverse_strand_against_the_disabled_upstream_fixture(built, tmp_path):
    """tests/lib.rs:309-320 (test_reverse_germline, commented out upstream): `normal` on the '-' strand. See check_reverse_germline."""
    from conftest import REVERSE_GERMLINE as R, check_reverse_germline
    d = R["dir"]
    args = [os.path.join(d, R["bam"]), os.path.join(d, R["vcf"]), os.path.join(d, R["fasta"])]
    whole = run_oracle_normal(*args, os.path.join(d, R["gtf"]), str(tmp_path))
    tail = run_oracle_normal(*args, os.path.join(d, R["last_exon_gtf"]), str(tmp_path))
    check_reverse_germline(whole, tail)


def test_oracle_frameshift_fixture_rows_are_the_upstream_rows(built, tmp_path):
    """tests/lib.rs:396-407 (frameshift_test, commented out upstream): see check_frameshift_fixture."""
    from conftest import check_frameshift_fixture, disabled_paths
    p, w = disabled_paths("frameshift_test")
    check_frameshift_fixture(run_oracle_files(p, str(tmp_path), window_len=w))


def test_oracle_on_three_way_splice_reports_only(built, tmp_path, capsys):
    """tests/lib.rs:384-394 (three_way_splice, commented out upstream; a window that spans three exons, expectation in an older TSV
    layout): NON-GATING - the oracle must run; the number of stale expected rows it still reproduces is reported."""
    from conftest import disabled_paths, stale_report
    p, w = disabled_paths("three_way_splice")
    got = run_oracle_files(p, str(tmp_path), window_len=w)
    rep = stale_report("three_way_splice", got["tsv"])
    with capsys.disabled():
        print("\n[non-gating] oracle on three_way_splice (-w %d): %s" % (w, rep))
    assert rep["rows"] > 0
