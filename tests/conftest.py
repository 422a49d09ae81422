import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_CLI = os.path.join(ROOT, "oracle", "_build", "oracle_cli")

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
