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


def _arrays_equal(a, b):
    import numpy as np
    assert a.keys() == b.keys()
    for k in a:
        if isinstance(a[k], np.ndarray):
            assert np.array_equal(a[k], b[k]), k
        else:
            assert a[k] == b[k], k


def test_phase_gene_level_arrays_roundtrip_on_a_reference_fixture(built):
    """mp_dataset_to_arrays / mp_dataset_from_arrays (the phase_gene seam, src/microphasing.rs:882-893): the decoded records of
    a reference fixture, copied into host-owned arrays and handed back, give a data set with the same phase_gene-level inputs
    (gene model, refseq, reads before the mapq filter, variants) - and the planner accepts it. The GPU suite runs it through the
    kernels against the reference's expected output."""
    import microphaser_amd as m
    ctx = m.Context(-1)
    p = fixture_paths("test_reverse")
    ds = ctx.load(p["bam"], p["vcf"], p["fasta"], p["gtf"])
    a = ds.to_arrays()
    assert a["n_genes"] == 1 and 2000 < len(a["r_pos"]) <= ds.num_reads and len(a["v_pos"]) >= 9   # the records the gene's fetch yields
    assert a["tx_strand"].tolist().count(1) == len(a["tx_strand"])          # the gene is on the '-' strand
    assert set(a["v_kind"].tolist()) == {0, 1, 2}                          # SNVs, the multi-allelic insertion, the 6-nt deletion
    ds2 = ctx.from_arrays(a)
    assert ds2.num_genes == 1 and ds2.num_reads == len(a["r_pos"])
    b = ds2.to_arrays()
    a.pop("r_qname_hash"); b.pop("r_qname_hash")                           # names travel as hashes: re-hashed on the way back
    _arrays_equal(a, b)
    ds2.batch()                                                            # the planner takes it (no GPU: run() would fail loudly)


def test_synthetic_exome_subset_is_the_same_exome(built):
    """Sharded generation (multi-GPU runs): with per-gene random streams a rank that materialises only its genes holds exactly
    those genes of the whole exome."""
    import microphaser_amd as m
    ctx = m.Context(-1)
    whole = ctx.synth(77, 12, gene_streams=True)
    keep = [1, 4, 5, 11]
    part = ctx.synth(77, 12, keep=keep)
    assert whole.num_genes == 12 and part.num_genes == len(keep)
    A, B = whole.to_arrays(), part.to_arrays()
    import numpy as np
    for k, g in enumerate(keep):
        assert A["gene_id"][g] == B["gene_id"][k] and A["gene_start"][g] == B["gene_start"][k] and A["gene_end"][g] == B["gene_end"][k]
        for off, fields in (("ref_off", ["refseq"]), ("read_off", ["r_pos", "r_mapq"]), ("var_off", ["v_pos", "v_alt", "v_kind", "v_is_germline"])):
            a0, a1, b0, b1 = int(A[off][g]), int(A[off][g + 1]), int(B[off][k]), int(B[off][k + 1])
            for f in fields:
                assert np.array_equal(A[f][a0:a1], B[f][b0:b1]), (g, f)
        # read bases and qualities
        ra0, ra1, rb0, rb1 = int(A["read_off"][g]), int(A["read_off"][g + 1]), int(B["read_off"][k]), int(B["read_off"][k + 1])
        assert np.array_equal(A["seq"][int(A["r_seq_off"][ra0]):int(A["r_seq_off"][ra1])], B["seq"][int(B["r_seq_off"][rb0]):int(B["r_seq_off"][rb1])])
    costs = ctx.synth_gene_costs(77, 12)
    assert len(costs) == 12 and all(c > 0 for c in costs)
    # the estimate tracks the real work: correlation with the generated reads per gene
    reads = np.diff(A["read_off"]).astype(float)
    assert np.corrcoef(reads, np.array(costs, dtype=float))[0, 1] > 0.95


def test_lpt_partition_is_balanced_and_complete():
    from microphaser_amd.shard import lpt_partition
    import random
    rng = random.Random(5)
    costs = [rng.randint(100, 5000) for _ in range(2000)]
    for world in (1, 2, 3, 8):
        parts = lpt_partition(costs, world)
        assert sorted(g for p in parts for g in p) == list(range(len(costs)))
        assert all(p == sorted(p) for p in parts)
        loads = [sum(costs[g] for g in p) for p in parts]
        assert max(loads) - min(loads) <= max(costs)          # LPT bound
        assert max(loads) <= 1.01 * sum(costs) / world


def test_peptides_union_merges_sorted_key_arrays(built):
    """mp_peptides_union (host-side merge of the all-gathered key arrays): no GPU needed."""
    import numpy as np
    import microphaser_amd as m
    ctx = m.Context(-1)
    a = np.array([m_ for m_ in (3, 5, 9, 1 << 44)], dtype=np.uint64)
    b = np.array([5, 7, 9, 11, (1 << 44) + 1], dtype=np.uint64)
    u = ctx.peptides_union([a, b, np.zeros(0, dtype=np.uint64)], 9)
    assert u.keys == sorted({3, 5, 9, 1 << 44, 7, 11, (1 << 44) + 1})
    assert m.decode_bincode_set(u.binary) == {m.key_to_peptide(k, 9).encode() for k in u.keys}
    with pytest.raises(m.MicrophaserError, match="sorted and distinct"):
        ctx.peptides_union([np.array([5, 3], dtype=np.uint64)], 9)


def test_translate_needs_a_gpu(built):
    import microphaser_amd as m
    with pytest.raises(m.MicrophaserError, match="no CPU fallback"):
        m.Context(-1).translate(b"ATGGCC", [0], 2)


def test_gtf_on_stdin_like_the_reference_cli(built):
    """mp_dataset_load with gtf = NULL reads the annotation from stdin (`microphaser somatic ... < genes.gtf`)."""
    import subprocess
    import sys
    p = fixture_paths("test_forward")
    code = ("import sys; sys.path.insert(0, %r)\nimport microphaser_amd as m\n"
            "ds = m.Context(-1).load(%r, %r, %r, None)\nprint(ds.num_genes, ds.num_reads)\n" % (ROOT, p["bam"], p["vcf"], p["fasta"]))
    with open(p["gtf"], "rb") as g:
        r = subprocess.run([sys.executable, "-c", code], stdin=g, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    import microphaser_amd as m
    ds = m.Context(-1).load(p["bam"], p["vcf"], p["fasta"], p["gtf"])
    assert r.stdout.split() == [str(ds.num_genes), str(ds.num_reads)]


def test_peptides_union_in_key_slices_matches_numpy(built):
    """Arrays long enough for the sliced (multi-threaded) union: equal to numpy's union; an unsorted array is refused wherever the
    bad pair sits."""
    import numpy as np
    import microphaser_amd as m
    ctx = m.Context(-1)
    rng = np.random.default_rng(5)
    arrs = [np.unique(rng.integers(0, 1 << 44, size=n, dtype=np.uint64)) for n in (400000, 250000, 7, 130000)]
    arrs += [np.zeros(0, dtype=np.uint64), np.array([5], dtype=np.uint64), arrs[0][::3].copy()]
    u = ctx.peptides_union(arrs, 9)
    assert np.array_equal(u.keys_np, np.unique(np.concatenate(arrs)))
    bad = np.arange(1, 400000, dtype=np.uint64)
    bad[300000] = bad[299999]
    with pytest.raises(m.MicrophaserError, match="sorted and distinct"):
        ctx.peptides_union([bad, np.arange(0, 10 ** 6, 3, dtype=np.uint64)], 9)


def test_header_compiles_as_plain_c():
    """include/microphaser_hip.h is a C ABI: it must be valid C99 (what cgo / bindgen / ctypes-style binders read) and C++."""
    import subprocess
    hdr = os.path.join(ROOT, "include", "microphaser_hip.h")
    for cmd in (["gcc", "-std=c99", "-fsyntax-only", "-x", "c", hdr], ["g++", "-std=c++17", "-fsyntax-only", "-x", "c++", hdr]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
