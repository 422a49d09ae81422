"""C-ABI library on the host (no GPU): symbols, planning, loud failure without a device."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, SOMATIC_FIXTURES, fixture_paths


def test_library_exports_every_declared_symbol(built):
    import microphaser_amd as m
    hdr = open(os.path.join(ROOT, "include", "microphaser_hip.h")).read()
    declared = set(re.findall(r"\b(mp_[a-z_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = ctypes.CDLL(m.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), "libmicrophaser_hip.so does not export " + name
    assert declared == set(m.C_ABI_SYMBOLS)


@pytest.mark.parametrize("name", sorted(SOMATIC_FIXTURES))
def test_planner_runs_on_reference_fixtures_without_gpu(built, name):
    import microphaser_amd as m
    ctx = m.Context(-1)
    p = fixture_paths(name)
    ds = ctx.load(p["bam"], p["vcf"], p["fasta"], p["gtf"])
    assert ds.num_genes >= 1 and ds.num_reads > 0
    b = ds.batch()
    with pytest.raises(m.MicrophaserError, match="no CPU fallback"):
        b.run()


def test_unsorted_gtf_is_an_error(built):
    import microphaser_amd as m
    from conftest import GOLDEN
    ctx = m.Context(-1)
    d = os.path.join(GOLDEN, "test_unsorted_gtf")
    with pytest.raises(m.MicrophaserError, match="not sorted"):
        ctx.load(os.path.join(d, "forward_test.bam"), os.path.join(d, "empty.vcf"),
                 os.path.join(GOLDEN, "test_forward", "chr14.mini.fa"), os.path.join(d, "chr14.unsorted.BDKRB2_DHRS2.gtf"))


def test_synthetic_dataset_roundtrips_through_files(built, tmp_path):
    """The in-memory synthetic data set and its on-disk form (BAM/VCF/GTF/FASTA writers + readers) agree:
    the oracle gives identical output on both."""
    import json
    import subprocess
    import microphaser_amd as m
    from conftest import ORACLE_CLI, run_oracle_files
    ctx = m.Context(-1)
    ds = ctx.synth(11, 6)
    prefix = str(tmp_path / "syn")
    ds.write(prefix)
    files = dict(bam=prefix + ".bam", vcf=prefix + ".vcf", gtf=prefix + ".gtf", fasta=prefix + ".fa")
    from_files = run_oracle_files(files, str(tmp_path))
    r = subprocess.run([ORACLE_CLI, "synth", "--seed", "11", "--transcripts", "6", "--prefix", str(tmp_path / "mem")],
                       capture_output=True, check=True)
    stats = json.loads(r.stdout)
    assert stats["windows"] > 100
    for ext in ("fa", "normal.fa", "tsv"):
        assert open(str(tmp_path / "mem") + "." + ext, "rb").read() == from_files[ext]
    assert from_files["tsv"].count(b"\n") > 10
