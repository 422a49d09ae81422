"""Parity of the HIP path (through the C ABI) against the reference's golden outputs and the CPU oracle."""
import json
import os
import subprocess

import pytest

from conftest import GOLDEN, NORMAL_FIXTURES, ORACLE_CLI, SOMATIC_FIXTURES, fixture_paths, read_expected

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(built):
    import microphaser_amd as m
    return m.Context(0)


@pytest.mark.parametrize("name", sorted(SOMATIC_FIXTURES))
def test_gpu_matches_reference_expected_output(ctx, name):
    p = fixture_paths(name)
    res = ctx.load(p["bam"], p["vcf"], p["fasta"], p["gtf"]).phase()
    exp = read_expected(p["expected"])
    assert res.fasta == exp["fa"]
    assert res.normal_fasta == exp["normal.fa"]
    assert res.tsv == exp["tsv"]


def test_gpu_variants_as_bcf_and_bgzf(ctx, tmp_path):
    """The product loader shares the format detection with the oracle CLI (tests/test_oracle_fixtures.py): BCF2 and bgzip'ed VCF in,
    the reference's expected output out."""
    import vcf_formats as vf
    p = fixture_paths("test_reverse")
    text = open(p["vcf"], "rb").read()
    exp = read_expected(p["expected"])
    for fn, data in (("v.bcf", vf.bgzf_bytes(vf.vcf_to_bcf(text, True))), ("v.vcf.gz", vf.bgzf_bytes(text))):
        path = str(tmp_path / fn)
        open(path, "wb").write(data)
        res = ctx.load(p["bam"], path, p["fasta"], p["gtf"]).phase()
        assert (res.fasta, res.normal_fasta, res.tsv) == (exp["fa"], exp["normal.fa"], exp["tsv"])


def test_gpu_empty_vcf(ctx):
    p = fixture_paths("test_forward")
    res = ctx.load(p["bam"], os.path.join(GOLDEN, "test_empty", "empty_test.vcf"), p["fasta"], p["gtf"]).phase()
    assert (res.fasta, res.normal_fasta, res.tsv) == (b"", b"", b"")


def oracle_synth(tmp, seed, n, depth=30.0, spacing=5.4):
    prefix = os.path.join(tmp, "oracle_%d_%d" % (seed, n))
    r = subprocess.run([ORACLE_CLI, "synth", "--seed", str(seed), "--transcripts", str(n), "--depth", str(depth),
                        "--spacing", str(spacing), "--prefix", prefix], capture_output=True, check=True)
    st = json.loads(r.stdout)
    return {e: open(prefix + "." + e, "rb").read() for e in ("fa", "normal.fa", "tsv")}, st


@pytest.mark.parametrize("seed,n,depth,spacing", [(7, 40, 30.0, 5.4), (1001, 150, 30.0, 5.4), (5005, 6, 500.0, 1.35), (99, 30, 120.0, 2.5)])
def test_gpu_matches_oracle_on_synthetic_exome(ctx, tmp_path, seed, n, depth, spacing):
    exp, st = oracle_synth(str(tmp_path), seed, n, depth, spacing)
    ds = ctx.synth(seed, n, depth, spacing)
    res = ds.phase()
    assert res.windows == st["windows"]
    assert res.fasta == exp["fa"]
    assert res.normal_fasta == exp["normal.fa"]
    assert res.tsv == exp["tsv"]
    assert exp["tsv"].count(b"\n") > 100


def test_gpu_gene_sharding_is_order_preserving(ctx, tmp_path):
    """Shards of genes concatenated in order give the whole-job output (the multi-GPU contract)."""
    ds = ctx.synth(31, 24)
    whole = ds.phase()
    parts = []
    for lo, hi in ((0, 7), (7, 16), (16, 24)):
        b = ds.batch(gene_lo=lo, gene_hi=hi)
        b.run()
        parts.append(b.results())
    tsv = b"".join([parts[0].tsv] + [p.tsv.split(b"\n", 1)[1] if p.tsv else b"" for p in parts[1:]])
    assert b"".join(p.fasta for p in parts) == whole.fasta
    assert tsv == whole.tsv
    assert sum(p.windows for p in parts) == whole.windows


def oracle_synth_ex(tmp, seed, n, indel, multi, soft, depth=30.0, spacing=5.4):
    prefix = os.path.join(tmp, "oracle_ex_%d_%d" % (seed, n))
    r = subprocess.run([ORACLE_CLI, "synth", "--seed", str(seed), "--transcripts", str(n), "--depth", str(depth), "--spacing", str(spacing),
                        "--indel-rate", str(indel), "--multiallelic-rate", str(multi), "--softmask-rate", str(soft),
                        "--skip-panics", "--prefix", prefix], capture_output=True, check=True)
    st = json.loads(r.stdout)
    return {e: open(prefix + "." + e, "rb").read() for e in ("fa", "normal.fa", "tsv")}, st


@pytest.mark.parametrize("seed,n,indel,multi,soft", [(17, 60, 0.05, 0.0, 0.0), (23, 50, 0.0, 0.15, 0.0), (29, 50, 0.0, 0.0, 0.5),
                                                       (17, 60, 0.08, 0.05, 0.2), (57, 40, 0.2, 0.1, 0.3)])
def test_gpu_matches_oracle_with_indels_multiallelic_and_softmasked_reference(ctx, tmp_path, seed, n, indel, multi, soft):
    """Short indels (incl. frameshifts -> shifted ORFs, frame > 0 rows), multi-allelic sites (j-stuck cursor) and lower-case
    reference stretches. Genes on which the reference itself would panic (slice / subtraction overflow in the splice merge
    of indel haplotypes) are dropped by the oracle harness; the engine must fail on exactly those genes too."""
    import microphaser_amd as m
    from microphaser_amd.shard import merge_streams
    exp, st = oracle_synth_ex(str(tmp_path), seed, n, indel, multi, soft)
    ds = ctx.synth(seed, n, indel_rate=indel, multiallelic_rate=multi, softmask_rate=soft)
    skipped = st["skipped"]
    parts, windows, lo = [], 0, 0
    for g in skipped + [ds.num_genes]:
        if g > lo:
            b = ds.batch(gene_lo=lo, gene_hi=g)
            b.run()
            r = b.results()
            parts.append(dict(fasta=r.fasta, normal_fasta=r.normal_fasta, tsv=r.tsv))
            windows += r.windows
        if g < ds.num_genes:
            with pytest.raises(m.MicrophaserError):   # at plan time or - when the failing step lies in the speculative part of the
                b = ds.batch(gene_lo=g, gene_hi=g + 1)   # schedule - when the real walk reaches it
                b.run()
                b.results()
                b.run()
                b.results()
        lo = g + 1
    got = merge_streams(parts)
    assert windows == st["windows"]
    assert got["fasta"] == exp["fa"]
    assert got["normal_fasta"] == exp["normal.fa"]
    assert got["tsv"] == exp["tsv"]
    assert exp["tsv"].count(b"\n") > 100


def test_gpu_build_reference_matches_reference_fixture(ctx):
    import microphaser_amd as m
    pep = ctx.build_reference(os.path.join(GOLDEN, "test_build", "reference.fa"), 4)
    assert pep.fasta == open(os.path.join(GOLDEN, "test_build", "expected_output", "reference_peptides.fasta"), "rb").read()
    want = m.decode_bincode_set(open(os.path.join(GOLDEN, "test_filter", "reference.binary"), "rb").read())
    assert m.decode_bincode_set(pep.binary) == want
    assert [m.key_to_peptide(k, 4) for k in pep.keys] == sorted(p.decode() for p in want)


def test_gpu_build_reference_matches_oracle_on_somatic_output(ctx, tmp_path):
    """Config-E shaped input: the FASTA that `somatic` emits (ids ending in F and R, lower-case variant bases), 9-mers."""
    import microphaser_amd as m
    res = ctx.synth(77, 30).phase()
    fa = tmp_path / "tumor.fa"
    fa.write_bytes(res.fasta)
    assert res.fasta.count(b">") > 1000
    out = tmp_path / "o.bin"
    r = subprocess.run([ORACLE_CLI, "build_reference", "-r", str(fa), "-l", "9", "-o", str(out)], capture_output=True, check=True)
    pep = ctx.build_reference(str(fa), 9)
    assert pep.fasta == r.stdout
    assert m.decode_bincode_set(pep.binary) == m.decode_bincode_set(out.read_bytes())
    assert pep.keys == sorted(set(pep.keys)) and len(pep.keys) == len(m.decode_bincode_set(pep.binary))
    assert m.keys_to_bincode(pep.keys, 9) == pep.binary


def test_gpu_build_reference_rejects_non_acgt_codons(ctx, tmp_path):
    import microphaser_amd as m
    fa = tmp_path / "n.fa"
    fa.write_text(">x_F\nACGTNACGTACG\n")
    with pytest.raises(m.MicrophaserError, match="reference would panic"):
        ctx.build_reference(str(fa), 4)


# ------------------------------------------------------------------ `microphaser normal` (src/normal_microphasing.rs)
@pytest.mark.parametrize("name", sorted(NORMAL_FIXTURES))
def test_gpu_normal_mode_matches_reference_expected_output(ctx, tmp_path, name):
    """The reference's own germline expectations (tests/lib.rs:237-249, :273-285 diff the FASTA only); the TSV, which the
    reference does not check, is compared with the oracle's."""
    import microphaser_amd as m
    bam, vcf, gtf, fa, exp = NORMAL_FIXTURES[name]
    d = os.path.join(GOLDEN, name)
    res = ctx.load(os.path.join(d, bam), os.path.join(d, vcf), os.path.join(d, fa), os.path.join(d, gtf)).phase(mode=m.MODE_NORMAL)
    assert res.fasta == open(os.path.join(d, "expected_output", exp), "rb").read()
    assert res.normal_fasta == b""
    with open(os.path.join(d, gtf), "rb") as g:
        r = subprocess.run([ORACLE_CLI, "normal", os.path.join(d, bam), "--variants", os.path.join(d, vcf), "--ref", os.path.join(d, fa),
                            "--tsv", str(tmp_path / "n.tsv")], stdin=g, capture_output=True, check=True)
    assert res.tsv == (tmp_path / "n.tsv").read_bytes()


@pytest.mark.parametrize("seed,n,depth,spacing,indel,multi,soft", [(7, 20, 30.0, 5.4, 0.0, 0.0, 0.0), (41, 24, 12.0, 3.0, 0.0, 0.0, 0.0),
                                                                     (17, 20, 20.0, 5.4, 0.06, 0.1, 0.3)])
def test_gpu_normal_mode_matches_oracle_on_synthetic_exome(ctx, tmp_path, seed, n, depth, spacing, indel, multi, soft):
    """Every haplotype of every window is emitted in this mode; '-' strand transcripts re-push their reads at every step
    (no `contains`), so row counts far exceed the depth. Genes the reference would panic on are dropped on both sides."""
    import microphaser_amd as m
    from microphaser_amd.shard import merge_streams
    prefix = os.path.join(str(tmp_path), "on")
    r = subprocess.run([ORACLE_CLI, "synth", "--mode", "normal", "--seed", str(seed), "--transcripts", str(n), "--depth", str(depth),
                        "--spacing", str(spacing), "--indel-rate", str(indel), "--multiallelic-rate", str(multi), "--softmask-rate", str(soft),
                        "--skip-panics", "--prefix", prefix], capture_output=True, check=True)
    st = json.loads(r.stdout)
    exp = {e: open(prefix + "." + e, "rb").read() for e in ("fa", "tsv")}
    ds = ctx.synth(seed, n, depth, spacing, indel_rate=indel, multiallelic_rate=multi, softmask_rate=soft)
    parts, windows, lo = [], 0, 0
    for g in st["skipped"] + [ds.num_genes]:
        if g > lo:
            b = ds.batch(gene_lo=lo, gene_hi=g, mode=m.MODE_NORMAL)
            b.run()
            res = b.results()
            parts.append(dict(fasta=res.fasta, normal_fasta=res.normal_fasta, tsv=res.tsv))
            windows += res.windows
        if g < ds.num_genes:
            with pytest.raises(m.MicrophaserError):
                b = ds.batch(gene_lo=g, gene_hi=g + 1, mode=m.MODE_NORMAL)
                b.run()
                b.results()
        lo = g + 1
    got = merge_streams(parts)
    assert windows == st["windows"]
    assert got["fasta"] == exp["fa"]
    assert got["tsv"] == exp["tsv"]
    assert exp["tsv"].count(b"\n") > 1000


def test_gpu_normal_mode_columns_past_window_end_lengthen_the_sequence(ctx, tmp_path):
    """Found by tools/fuzz_vs_oracle.py (seed 70098): with a variant every 1.35 nt and 3 % indels a '-' strand window keeps
    stale columns past its end; an SNV at the window's last base starts a run of adjacent columns that the unbounded inner
    loop of the walk (src/normal_microphasing.rs:417-475) applies past window_end, so the sequence (52 nt here) outgrows
    window + inserted bases. The planner has to size the records for that run."""
    import microphaser_amd as m
    prefix = os.path.join(str(tmp_path), "on")
    subprocess.run([ORACLE_CLI, "synth", "--mode", "normal", "--seed", "70098", "--transcripts", "16", "--depth", "20", "--spacing", "1.35",
                    "--indel-rate", "0.03", "--genes", "9:10", "--skip-panics", "--prefix", prefix], capture_output=True, check=True)
    ds = ctx.synth(70098, 16, 20.0, 1.35, indel_rate=0.03)
    b = ds.batch(gene_lo=9, gene_hi=10, mode=m.MODE_NORMAL)
    b.run()
    res = b.results()
    assert res.fasta == open(prefix + ".fa", "rb").read()
    assert res.tsv == open(prefix + ".tsv", "rb").read()
    assert res.tsv.count(b"\n") > 100000


def test_gpu_merge_uses_the_window_left_by_a_stopped_shifted_orf(ctx, tmp_path):
    """Found by tools/fuzz_vs_oracle.py (seeds 97187, 97231): the planner's schedule is speculative, a shifted ORF that really stops
    (stop codon in its frame) prints no more, so at the next splice side hap_vec still holds an EARLIER window's haplotypes
    (src/microphasing.rs:1445-1454, 1505-1540). Those windows need full records too, or the merge sees empty ones."""
    prefix = os.path.join(str(tmp_path), "o")
    args = ["--seed", "97187", "--transcripts", "16", "--depth", "30", "--spacing", "5.4", "--indel-rate", "0.03", "--multiallelic-rate", "0.08",
            "--window-len", "33"]
    subprocess.run([ORACLE_CLI, "synth", *args, "--genes", "12:13", "--skip-panics", "--prefix", prefix], capture_output=True, check=True)
    ds = ctx.synth(97187, 16, 30.0, 5.4, indel_rate=0.03, multiallelic_rate=0.08)
    b = ds.batch(window_len=33, gene_lo=12, gene_hi=13)
    b.run()
    res = b.results()
    assert res.tsv == open(prefix + ".tsv", "rb").read()
    assert res.fasta == open(prefix + ".fa", "rb").read()
    assert res.normal_fasta == open(prefix + ".normal.fa", "rb").read()
    assert res.tsv.count(b"\n") > 300


def test_gpu_row_outliving_its_exon_is_not_admitted_twice(ctx, tmp_path):
    """Found by tools/fuzz_vs_oracle.py (seed 200350, 250-nt reads): on the '-' strand a read longer than an intron is still a row
    when the next exon's first window lists every read in range again; `contains` (src/microphasing.rs:281-294) must keep that
    copy out for good - also when the row has meanwhile gone low-quality - instead of admitting it a few windows later."""
    prefix = os.path.join(str(tmp_path), "o")
    args = ["--seed", "200350", "--transcripts", "16", "--depth", "6", "--spacing", "1.35", "--window-len", "15", "--read-len", "250"]
    subprocess.run([ORACLE_CLI, "synth", *args, "--genes", "1:2", "--skip-panics", "--prefix", prefix], capture_output=True, check=True)
    ds = ctx.synth(200350, 16, 6.0, 1.35, read_len=250)
    b = ds.batch(window_len=15, gene_lo=1, gene_hi=2)
    b.run()
    res = b.results()
    assert res.tsv == open(prefix + ".tsv", "rb").read()
    assert res.fasta == open(prefix + ".fa", "rb").read()
    assert res.tsv.count(b"\n") > 200


def test_gpu_segment_with_non_consecutive_initial_columns(ctx, tmp_path):
    """Found by tools/fuzz_vs_oracle.py (seed 720659, gene 53): a '-' exon starts with a stale frameshift column of the previous exon
    in the deque (src/microphasing.rs:1159) while the variants between it and the exon's first window were never appended, so the
    live columns are not consecutive in transcription order. A replay segment starting there cannot be given its initial deque as an
    index range; the rows pushed at that first window must still see the stale column (frame.1 != 0 keeps them out of the shifted
    ORF's counts, :395-399)."""
    prefix = os.path.join(str(tmp_path), "o")
    args = ["--seed", "720659", "--transcripts", "64", "--depth", "30", "--spacing", "2.0", "--indel-rate", "0.03", "--window-len", "33",
            "--isoform-rate", "0.5"]
    subprocess.run([ORACLE_CLI, "synth", *args, "--genes", "53:54", "--skip-panics", "--prefix", prefix], capture_output=True, check=True)
    ds = ctx.synth(720659, 64, 30.0, 2.0, indel_rate=0.03, isoform_rate=0.5)
    b = ds.batch(window_len=33, gene_lo=53, gene_hi=54)
    b.run()
    res = b.results()
    assert res.tsv == open(prefix + ".tsv", "rb").read()
    assert res.fasta == open(prefix + ".fa", "rb").read()
    assert res.tsv.count(b"\n") > 500


def test_gpu_merge_reaches_back_to_a_shifted_orf_window_with_window_length_28(ctx, tmp_path):
    """Found by tools/fuzz_vs_oracle.py (seed 970028, -w 28, 10 % indels): the real walk's hap_vec still held the window a shifted ORF
    printed thousands of bases earlier, which the speculative schedule had not marked for a merge. Every window of a stretch with a
    shifted ORF keeps its records (the device has them anyway), so the merge finds them."""
    prefix = os.path.join(str(tmp_path), "o")
    args = ["--seed", "970028", "--transcripts", "64", "--depth", "6", "--spacing", "9.0", "--indel-rate", "0.1", "--window-len", "28"]
    subprocess.run([ORACLE_CLI, "synth", *args, "--genes", "56:57", "--skip-panics", "--prefix", prefix], capture_output=True, check=True)
    ds = ctx.synth(970028, 64, 6.0, 9.0, indel_rate=0.1)
    b = ds.batch(window_len=28, gene_lo=56, gene_hi=57)
    b.run()
    res = b.results()
    assert res.tsv == open(prefix + ".tsv", "rb").read()
    assert res.fasta == open(prefix + ".fa", "rb").read()
    assert res.tsv.count(b"\n") > 100


# ------------------------------------------------------------------ the product CLI (src/cli.yaml surface): GTF on stdin, FASTA on stdout
PRODUCT_CLI = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "microphaser_amd", "_lib", "microphaser")


def test_gpu_cli_somatic_matches_reference_expected_output(built, tmp_path):
    p = fixture_paths("test_reverse")
    with open(p["gtf"], "rb") as g:
        r = subprocess.run([PRODUCT_CLI, "somatic", p["bam"], "--variants", p["vcf"], "--ref", p["fasta"], "--tsv", str(tmp_path / "o.tsv"),
                            "--normal-output", str(tmp_path / "o.normal.fa")], stdin=g, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    exp = read_expected(p["expected"])
    assert r.stdout == exp["fa"]
    assert (tmp_path / "o.normal.fa").read_bytes() == exp["normal.fa"]
    assert (tmp_path / "o.tsv").read_bytes() == exp["tsv"]
    # the CLI ends the process once its outputs are written; MP_CLEAN_EXIT=1 keeps the orderly teardown (same outputs, exit 0)
    with open(p["gtf"], "rb") as g:
        r2 = subprocess.run([PRODUCT_CLI, "somatic", p["bam"], "--variants", p["vcf"], "--ref", p["fasta"], "--tsv", str(tmp_path / "c.tsv"),
                             "--normal-output", str(tmp_path / "c.normal.fa")], stdin=g, capture_output=True, env=dict(os.environ, MP_CLEAN_EXIT="1"))
    assert r2.returncode == 0, r2.stderr.decode()
    assert r2.stdout == exp["fa"] and (tmp_path / "c.tsv").read_bytes() == exp["tsv"]


def test_gpu_cli_normal_matches_reference_expected_output(built, tmp_path):
    bam, vcf, gtf, fa, exp = NORMAL_FIXTURES["splice_forward_test"]
    d = os.path.join(GOLDEN, "splice_forward_test")
    with open(os.path.join(d, gtf), "rb") as g:
        r = subprocess.run([PRODUCT_CLI, "normal", os.path.join(d, bam), "-b", os.path.join(d, vcf), "-r", os.path.join(d, fa),
                            "-t", str(tmp_path / "n.tsv"), "-w", "27"], stdin=g, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == open(os.path.join(d, "expected_output", exp), "rb").read()
    assert (tmp_path / "n.tsv").read_bytes().count(b"\n") == r.stdout.count(b">") + 1


# ------------------------------------------------------------------ `microphaser filter` (src/peptides.rs:188-709)
FILTER_FIXTURES = {"test_filter": "filtered", "test_filter_long": "filtered_long", "test_filter_fs": "filtered_fs"}


@pytest.mark.parametrize("name", sorted(FILTER_FIXTURES))
def test_gpu_filter_matches_reference_expected_output(ctx, name):
    d, stem = os.path.join(GOLDEN, name), FILTER_FIXTURES[name]
    f = ctx.filter(os.path.join(d, "info.tsv"), os.path.join(d, "reference.binary"), 9)
    exp = os.path.join(d, "expected_output")
    assert f.fasta == open(os.path.join(exp, "tumor.%s.fa" % stem), "rb").read()
    assert f.normal_fasta == open(os.path.join(exp, "normal.%s.fa" % stem), "rb").read()
    assert f.tsv == open(os.path.join(exp, "info.%s.tsv" % stem), "rb").read()


def test_gpu_config_e_pipeline_matches_oracle(ctx, tmp_path):
    """Config E end to end on one synthetic exome: `normal` -> `build_reference -l 9` (normal peptidome) -> `somatic` ->
    `filter`; every stage on the GPU, every stage's output compared with the CPU oracle run on the same bytes."""
    import microphaser_amd as m
    ds = ctx.synth(303, 30, indel_rate=0.03)
    normal_fa = tmp_path / "normal_peptides.fa"
    parts = []
    for g in range(ds.num_genes):  # genes the reference would panic on are dropped from the peptidome input
        try:
            b = ds.batch(gene_lo=g, gene_hi=g + 1, mode=m.MODE_NORMAL)
            b.run()
            parts.append(b.results().fasta)
        except m.MicrophaserError:
            pass
    normal_fa.write_bytes(b"".join(parts))
    pep = ctx.build_reference(str(normal_fa), 9)
    ref_bin = tmp_path / "reference.binary"
    ref_bin.write_bytes(pep.binary)
    som_parts = []
    for g in range(ds.num_genes):
        try:
            b = ds.batch(gene_lo=g, gene_hi=g + 1)
            b.run()
            r = b.results()
            som_parts.append(dict(fasta=r.fasta, normal_fasta=r.normal_fasta, tsv=r.tsv))
        except m.MicrophaserError:
            pass
    from microphaser_amd.shard import merge_streams
    info = tmp_path / "info.tsv"
    info.write_bytes(merge_streams(som_parts)["tsv"])
    assert info.read_bytes().count(b"\n") > 500
    f = ctx.filter(str(info), str(ref_bin), 9)
    r = subprocess.run([ORACLE_CLI, "filter", "-r", str(ref_bin), "-l", "9", "-t", str(info), "-o", str(tmp_path / "o.tsv"),
                        "-n", str(tmp_path / "o.normal.fa"), "-s", str(tmp_path / "o.removed.tsv"), "-p", str(tmp_path / "o.removed.fa")],
                       capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    assert f.fasta == r.stdout
    assert f.normal_fasta == (tmp_path / "o.normal.fa").read_bytes()
    assert f.tsv == (tmp_path / "o.tsv").read_bytes()
    assert f.removed_tsv == (tmp_path / "o.removed.tsv").read_bytes()
    assert f.removed_fasta == (tmp_path / "o.removed.fa").read_bytes()
    assert f.kept > 100 and f.removed > 0 and f.groups > 20
    # the buffer entry point gives the same result
    f2 = ctx.filter(info.read_bytes(), ref_bin.read_bytes(), 9)
    assert (f2.fasta, f2.tsv, f2.removed_tsv) == (f.fasta, f.tsv, f.removed_tsv)


@pytest.mark.parametrize("gold_file", ["checksums_gene_streams.json", "checksums.json"])
def test_gpu_full_size_config_c_matches_the_oracle_verified_checksums(ctx, gold_file):
    """BASELINE config C at FULL size (20000 transcripts, 7.8 M windows, 8.8 M TSV rows, 3.5 GB of text): the oracle needs minutes for it
    on one thread, so the suite compares checksums - the md5 of the three streams recorded in a run in which the CPU oracle produced
    identical streams on the same exome - and two size-independent properties: a second pass over the resident batch gives the same
    bytes and window count. checksums_gene_streams.json is THE EXOME bench.py TIMES (per-gene random streams; recorded by
    tools/record_golden.py C, oracle on the host threads of the GPU box, window count 7755097 as in the bench line); checksums.json is
    the single-stream variant of the generator (tools/e2e_cli.py C --md5: product CLI and oracle_cli somatic on the same files)."""
    import hashlib
    gold = json.load(open(os.path.join(GOLDEN, "config_c", gold_file)))
    ds = ctx.synth(gold["seed"], gold["transcripts"], gold["depth"], gold["spacing"], gene_streams=bool(gold.get("gene_streams")))
    b = ds.batch()
    b.run()
    r = b.results()
    md5 = {k: hashlib.md5(getattr(r, k)).hexdigest() for k in ("fasta", "normal_fasta", "tsv")}
    assert md5 == gold["md5"]
    assert r.tsv.count(b"\n") - 1 == gold["tsv_rows"]
    if "windows" in gold:
        assert r.windows == gold["windows"]
    n_windows = r.windows
    r.close()
    b.run()                       # idempotence: the pass again over the same resident inputs
    r2 = b.results(m_stream_tsv())
    assert hashlib.md5(r2.tsv).hexdigest() == gold["md5"]["tsv"] and r2.windows == n_windows and r2.fasta == b""
    r2.close(); b.close()


def test_gpu_config_e_at_4000_transcripts_matches_the_oracle_verified_checksums(ctx, tmp_path):
    """Config E (normal -> build_reference -l 9 -> somatic -> filter) on a 4000-transcript exome of BASELINE's shape (seed 2020, 30x, SNV
    every 5.4 nt, per-gene random streams): 1.56 M `normal` windows, a 14 M-peptide normal peptidome, 1.76 M somatic TSV rows. The CPU
    oracle needs 71 s for its four stages on 32 threads, so the suite compares the md5 of EVERY stream with the values recorded in the
    run of tools/record_golden.py E 4000 in which the oracle's stages produced the same bytes (tests/golden/config_e/checksums_4000.json;
    the peptidome as the md5 of its sorted peptide list - HashSet order is arbitrary)."""
    import hashlib
    import numpy as np
    import microphaser_amd as m
    gold = json.load(open(os.path.join(GOLDEN, "config_e", "checksums_4000.json")))
    L = gold["peptide_len"]
    md5 = lambda b: hashlib.md5(b).hexdigest()
    ds = ctx.synth(gold["seed"], gold["transcripts"], gold["depth"], gold["spacing"], gene_streams=True)
    b = ds.batch(window_len=3 * L, mode=m.MODE_NORMAL)
    b.run()
    nres = b.results(m.STREAM_FASTA)
    normal_fa, normal_windows = nres.fasta, nres.windows
    nres.close(); b.close()
    got = {"normal_fasta_of_normal_mode": md5(normal_fa)}
    pep = ctx.peptidome(normal_fa, L, lazy=False)
    body = np.frombuffer(pep.binary, dtype=np.uint8)
    n_pep = int(body[:8].view("<u8")[0])
    peps = np.sort(np.ascontiguousarray(body[8:].reshape(n_pep, 8 + L)[:, 8:]).view("S%d" % L)[:, 0])
    got["peptidome_sorted"] = md5(peps.tobytes())
    sres = ds.phase()
    got.update({"somatic_fasta": md5(sres.fasta), "somatic_normal_fasta": md5(sres.normal_fasta), "somatic_tsv": md5(sres.tsv)})
    f = ctx.filter(sres.tsv, pep)    # the peptidome by handle (sorted keys straight to the GPU) ...
    got.update({"filter_fasta": md5(f.fasta), "filter_normal_fasta": md5(f.normal_fasta), "filter_tsv": md5(f.tsv),
                "filter_removed_tsv": md5(f.removed_tsv), "filter_removed_fasta": md5(f.removed_fasta)})
    assert got == gold["md5"]
    c = gold["counts"]
    assert (normal_windows, n_pep, sres.windows, f.kept, f.removed, f.groups) == (c["normal_windows"], c["peptidome"], c["somatic_windows"],
                                                                                  c["filter_kept"], c["filter_removed"], c["filter_groups"])


@pytest.mark.parametrize("name,seed,n,depth,spacing,windows", [("B", 1001, 1000, 30.0, 5.4, 388716), ("D", 5005, 500, 500.0, 1.35, 184824)])
def test_gpu_full_size_configs_b_and_d_match_oracle(ctx, tmp_path, name, seed, n, depth, spacing, windows):
    """BASELINE configs B (1000 transcripts, 30x) and D (500 transcripts at 500x, a variant every 1.35 nt: ~366 rows and 20 columns per
    window, two mask words, the sort-based counting kernel) at FULL size: all three streams byte-identical to the oracle, which shards
    the genes over the host threads here (D: ~20 s)."""
    prefix = str(tmp_path / name)
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    r = subprocess.run([ORACLE_CLI, "synth", "--seed", str(seed), "--transcripts", str(n), "--depth", str(depth), "--spacing", str(spacing),
                        "--gene-streams", "--threads", str(threads), "--prefix", prefix], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    st = json.loads(r.stdout)
    res = ctx.synth(seed, n, depth, spacing, gene_streams=True).phase()
    assert res.windows == st["windows"] == windows
    for k, ext in (("fasta", "fa"), ("normal_fasta", "normal.fa"), ("tsv", "tsv")):
        assert getattr(res, k) == open(prefix + "." + ext, "rb").read(), k


def m_stream_tsv():
    import microphaser_amd as m
    return m.STREAM_TSV


def test_gpu_results_stream_selection(ctx):
    """mp_batch_results_select: a stream that is not asked for is empty, the others are byte-identical to the full result."""
    import microphaser_amd as m
    ds = ctx.synth(99, 20, gene_streams=True)
    for mode in (m.MODE_SOMATIC, m.MODE_NORMAL):
        b = ds.batch(mode=mode)
        b.run()
        full = b.results()
        fa = b.results(m.STREAM_FASTA)
        assert fa.fasta == full.fasta and fa.tsv == b"" and fa.normal_fasta == b"" and fa.windows == full.windows and len(full.tsv) > 1000
        tsv = b.results(m.STREAM_TSV | m.STREAM_NORMAL_FASTA)
        assert tsv.tsv == full.tsv and tsv.normal_fasta == full.normal_fasta and tsv.fasta == b""
        assert tsv.gene_offsets(2) == full.gene_offsets(2) and fa.gene_offsets(0) == full.gene_offsets(0)


def test_gpu_filter_host_legs_threaded_quoted_and_by_handle(ctx, tmp_path, monkeypatch):
    """`filter` on a TSV large enough for the threaded host legs (the text is cut at line breaks and parsed / written by all host
    threads): byte-identical to the oracle's sequential filter and to the same call on one thread; the peptidome handed over as the
    build_reference handle (no bincode round trip) gives the same streams; a TSV with quoted fields (parsed in one piece) too."""
    import microphaser_amd as m
    L = 9
    ds = ctx.synth(515, 80, 30.0, 5.4, gene_streams=True)
    nres = ds.phase(window_len=3 * L, mode=m.MODE_NORMAL)
    pep = ctx.build_reference(nres.fasta, L)
    sres = ds.phase(window_len=3 * L)
    assert len(sres.tsv) > 2 * (4 << 20)                 # several parts of 4 MiB
    info, ref_bin = tmp_path / "info.tsv", tmp_path / "reference.binary"
    info.write_bytes(sres.tsv)
    ref_bin.write_bytes(pep.binary)
    r = subprocess.run([ORACLE_CLI, "filter", "-r", str(ref_bin), "-l", str(L), "-t", str(info), "-o", str(tmp_path / "o.tsv"),
                        "-n", str(tmp_path / "o.normal.fa"), "-s", str(tmp_path / "o.removed.tsv"), "-p", str(tmp_path / "o.removed.fa")],
                       capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    want = (r.stdout, (tmp_path / "o.normal.fa").read_bytes(), (tmp_path / "o.tsv").read_bytes(), (tmp_path / "o.removed.tsv").read_bytes(),
            (tmp_path / "o.removed.fa").read_bytes())
    streams = lambda f: (f.fasta, f.normal_fasta, f.tsv, f.removed_tsv, f.removed_fasta)
    f = ctx.filter(sres.tsv, pep.binary, L)
    assert streams(f) == want and f.kept > 1000 and f.removed > 100
    assert streams(ctx.filter(sres.tsv, pep)) == want                      # mp_filter_peptides
    monkeypatch.setenv("MP_THREADS", "1")
    assert streams(ctx.filter(sres.tsv, pep.binary, L)) == want
    monkeypatch.delenv("MP_THREADS")
    # quoted fields: gene names with a tab / a quote, written by the csv rules, read back and written again
    lines = sres.tsv.split(b"\n")
    for k in range(1, len(lines) - 1, 7):
        c = lines[k].split(b"\t")
        c[3] = b'"' + (b"na\tme" if k % 2 else b'na""me') + b'"'
        lines[k] = b"\t".join(c)
    quoted = b"\n".join(lines)
    info.write_bytes(quoted)
    r = subprocess.run([ORACLE_CLI, "filter", "-r", str(ref_bin), "-l", str(L), "-t", str(info), "-o", str(tmp_path / "q.tsv"),
                        "-n", str(tmp_path / "q.normal.fa"), "-s", str(tmp_path / "q.removed.tsv"), "-p", str(tmp_path / "q.removed.fa")],
                       capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    fq = ctx.filter(quoted, pep.binary, L)
    assert (fq.fasta, fq.tsv, fq.removed_tsv) == (r.stdout, (tmp_path / "q.tsv").read_bytes(), (tmp_path / "q.removed.tsv").read_bytes())
    assert b'"na\tme"' in fq.tsv + fq.removed_tsv and b'"na""me"' in fq.tsv + fq.removed_tsv
    # a malformed row in the middle of the file is reported the same way by every part
    bad = sres.tsv.replace(b"\tForward\t", b"\tForward\tx\t", 1)
    with pytest.raises(m.MicrophaserError, match="22 fields"):
        ctx.filter(bad, pep.binary, L)


def test_gpu_sequential_and_window_parallel_replay_agree(ctx, monkeypatch):
    """The two replay paths (K2 state machine per segment; K2a + K2w closed form per window) write identical results:
    the same exome planned with MP_SEQUENTIAL_REPLAY=1 (everything through K2) and by default (everything eligible
    through K2a + K2w), incl. low-quality rejections, start-codon variants and both strands."""
    ds = ctx.synth(4242, 60, 40.0, 4.0)
    b = ds.batch()
    st = b.run()
    fast = b.results()
    assert st.n_steps_w > 0 and st.n_steps_w >= 9 * st.n_steps_seq
    monkeypatch.setenv("MP_SEQUENTIAL_REPLAY", "1")
    b2 = ds.batch()
    st2 = b2.run()
    slow = b2.results()
    assert st2.n_steps_w == 0 and st2.n_steps_seq == st.n_steps_w + st.n_steps_seq
    assert (fast.fasta, fast.normal_fasta, fast.tsv, fast.windows) == (slow.fasta, slow.normal_fasta, slow.tsv, slow.windows)
    assert fast.tsv.count(b"\n") > 1000


def test_gpu_paths_agree_on_random_exomes(built):
    """Differential sweep (tools/stress_paths.py): window-parallel replay + byte-substitution sequences vs sequential replay +
    general sequence walk on random depths / variant spacings / indel, multi-allelic and soft-mask rates."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_paths.py"), "900", "6"], capture_output=True, text=True, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches: 0" in r.stdout


@pytest.mark.parametrize("window_len", [33, 15])
def test_gpu_matches_oracle_with_other_window_lengths(ctx, tmp_path, window_len):
    """-w 33: exons of 30..35 nt become "short" exons (one window spanning the exon, splice_pos 2, multi-exon merges) and
    windows exceed 32 nt (general sequence walk); -w 15: 5-mers. Same exome, oracle with the same --window-len."""
    import microphaser_amd as m
    from microphaser_amd.shard import merge_streams
    prefix = os.path.join(str(tmp_path), "ow")
    r = subprocess.run([ORACLE_CLI, "synth", "--seed", "61", "--transcripts", "40", "--window-len", str(window_len), "--skip-panics",
                        "--prefix", prefix], capture_output=True, check=True)
    st = json.loads(r.stdout)
    exp = {e: open(prefix + "." + e, "rb").read() for e in ("fa", "normal.fa", "tsv")}
    ds = ctx.synth(61, 40)
    parts, windows, lo = [], 0, 0
    for g in st["skipped"] + [ds.num_genes]:
        if g > lo:
            b = ds.batch(window_len=window_len, gene_lo=lo, gene_hi=g)
            b.run()
            res = b.results()
            parts.append(dict(fasta=res.fasta, normal_fasta=res.normal_fasta, tsv=res.tsv))
            windows += res.windows
        if g < ds.num_genes:
            with pytest.raises(m.MicrophaserError):
                b = ds.batch(window_len=window_len, gene_lo=g, gene_hi=g + 1)
                b.run()
                b.results()
        lo = g + 1
    got = merge_streams(parts)
    assert windows == st["windows"]
    assert got["fasta"] == exp["fa"]
    assert got["normal_fasta"] == exp["normal.fa"]
    assert got["tsv"] == exp["tsv"]
    assert exp["tsv"].count(b"\n") > 200


def test_gpu_cli_multi_device_sharding_is_byte_identical(built, tmp_path):
    """`--devices a,b`: one context + host thread per GPU over contiguous gene ranges (here both shards on device 0 - the box
    has one GPU - which exercises the same code path), outputs concatenated in gene order."""
    import microphaser_amd as m
    ctx = m.Context(-1)
    ds = ctx.synth(77, 24)
    prefix = str(tmp_path / "s")
    ds.write(prefix)
    outs = []
    for extra in ([], ["--devices", "0,0,0"]):
        tag = "m" if extra else "s"
        with open(prefix + ".gtf", "rb") as g:
            r = subprocess.run([PRODUCT_CLI, "somatic", prefix + ".bam", "-b", prefix + ".vcf", "-r", prefix + ".fa", "-t", str(tmp_path / (tag + ".tsv")),
                                "-n", str(tmp_path / (tag + ".nfa"))] + extra, stdin=g, capture_output=True)
        assert r.returncode == 0, r.stderr.decode()
        outs.append((r.stdout, (tmp_path / (tag + ".tsv")).read_bytes(), (tmp_path / (tag + ".nfa")).read_bytes()))
    assert outs[0] == outs[1]
    assert outs[0][1].count(b"\n") > 1000


def test_gpu_result_buffers_grow_on_overflow(ctx, monkeypatch):
    """The output allocators start from an estimate; when any of the 64 sub-ranges (groups, records, wanted-id lists) overflows,
    run() enlarges the buffers and repeats the pass. Forced here by starting from 256 slots per allocator."""
    ds = ctx.synth(515, 40)
    ref = ds.phase()
    monkeypatch.setenv("MP_TEST_SMALL_CAPS", "1")
    b = ds.batch()
    st = b.run()
    res = b.results()
    assert st.attempts > 1
    assert (res.fasta, res.normal_fasta, res.tsv, res.windows) == (ref.fasta, ref.normal_fasta, ref.tsv, ref.windows)
    # the normal-mode replay allocates larger chunks: same check
    monkeypatch.delenv("MP_TEST_SMALL_CAPS")
    import microphaser_amd as m
    nref = ds.phase(mode=m.MODE_NORMAL)
    monkeypatch.setenv("MP_TEST_SMALL_CAPS", "1")
    b = ds.batch(mode=m.MODE_NORMAL)
    st = b.run()
    res = b.results()
    assert st.attempts > 1
    assert (res.fasta, res.tsv) == (nref.fasta, nref.tsv)


def test_gpu_batches_of_one_context_do_not_alias(ctx):
    """A context holds one batch in HBM at a time. Running an older batch after a newer one was created re-uploads it;
    asking for results that were overwritten is an error, not stale data."""
    import microphaser_amd as m
    ds = ctx.synth(88, 12)
    b1 = ds.batch(gene_lo=0, gene_hi=6)
    b2 = ds.batch(gene_lo=6, gene_hi=12)     # replaces b1 in HBM
    b1.run()                                 # comes back
    r1 = b1.results()
    b2.run()
    r2 = b2.results()
    with pytest.raises(m.MicrophaserError):
        b1.results()                         # b2 ran since
    whole = ds.phase()
    assert r1.fasta + r2.fasta == whole.fasta
    assert r1.windows + r2.windows == whole.windows


# ---- reference fixtures whose upstream tests are disabled (tests/lib.rs:309-320, :384-408); see conftest.py for what each one pins
def test_gpu_normal_mode_reverse_strand_against_the_disabled_upstream_fixture(ctx):
    import microphaser_amd as m
    from conftest import REVERSE_GERMLINE as R, check_reverse_germline
    d = R["dir"]
    out = []
    for gtf in (R["gtf"], R["last_exon_gtf"]):
        res = ctx.load(os.path.join(d, R["bam"]), os.path.join(d, R["vcf"]), os.path.join(d, R["fasta"]), os.path.join(d, gtf)).phase(mode=m.MODE_NORMAL)
        out.append(res.fasta)
    check_reverse_germline(*out)


def test_gpu_frameshift_fixture_rows_are_the_upstream_rows(ctx):
    from conftest import check_frameshift_fixture, disabled_paths
    p, w = disabled_paths("frameshift_test")
    res = ctx.load(p["bam"], p["vcf"], p["fasta"], p["gtf"]).phase(window_len=w)
    check_frameshift_fixture({"fa": res.fasta, "normal.fa": res.normal_fasta, "tsv": res.tsv})


def test_gpu_three_way_splice_equals_the_oracle_and_reports(ctx, tmp_path, capsys):
    """NON-GATING against the stale upstream rows (reported); gating against the oracle."""
    from conftest import disabled_paths, run_oracle_files, stale_report
    p, w = disabled_paths("three_way_splice")
    res = ctx.load(p["bam"], p["vcf"], p["fasta"], p["gtf"]).phase(window_len=w)
    exp = run_oracle_files(p, str(tmp_path), window_len=w)
    assert (res.fasta, res.normal_fasta, res.tsv) == (exp["fa"], exp["normal.fa"], exp["tsv"])
    with capsys.disabled():
        print("\n[non-gating] GPU on three_way_splice (-w %d): %s" % (w, stale_report("three_way_splice", res.tsv)))


def test_gpu_lane_per_window_and_wave_per_window_replay_agree(ctx, monkeypatch):
    """K2l (lane per window: RowRecs + per-lane LDS counters) against K2w (wave per window) on the same exome: the default plan
    sends the windows with <= 8 columns to K2l, MP_NO_LANE_KERNEL=1 sends everything to K2w. Shallow and deep (rows spread over
    more than 64 reads: the multi-block wave kernel keeps the wide windows, the lane kernel takes the narrow ones)."""
    for seed, n, depth, spacing in ((4243, 50, 30.0, 5.4), (4244, 30, 45.0, 2.5), (4245, 8, 150.0, 9.0)):
        ds = ctx.synth(seed, n, depth, spacing)
        monkeypatch.delenv("MP_NO_LANE_KERNEL", raising=False)
        b = ds.batch()
        b.run()
        lanes = b.results()
        monkeypatch.setenv("MP_NO_LANE_KERNEL", "1")
        b2 = ds.batch()
        b2.run()
        waves = b2.results()
        assert (lanes.fasta, lanes.normal_fasta, lanes.tsv, lanes.windows) == (waves.fasta, waves.normal_fasta, waves.tsv, waves.windows)
        assert lanes.tsv.count(b"\n") > 200
    monkeypatch.delenv("MP_NO_LANE_KERNEL", raising=False)


def test_gpu_flat_and_per_exon_admission_agree(ctx, monkeypatch):
    """K2a in its flat form (a lane per (exon, read) entry across exon boundaries, the exon's fields packed on the device at upload)
    against the form with a wave per <= 64 reads of ONE exon (MP_K2A_CHUNKS=1), one and two entries per lane: same bytes - on an exome
    with both strands, indels, multi-allelic sites and soft-masked reference (exons of both kinds: window-parallel and sequential)
    and on a deep one (two mask words per read)."""
    for seed, n, depth, spacing, kw in ((5151, 60, 30.0, 5.4, {}), (5152, 40, 35.0, 6.0, dict(indel_rate=0.05, multiallelic_rate=0.05, softmask_rate=0.1)),
                                        (5153, 6, 400.0, 1.6, {})):
        import microphaser_amd as m
        ds = ctx.synth(seed, n, depth, spacing, **kw)
        got = []
        for env in ({}, {"MP_K2A_ITEMS": "2"}, {"MP_K2A_CHUNKS": "1"}):
            for k in ("MP_K2A_ITEMS", "MP_K2A_CHUNKS"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            per_gene = []   # the whole exome in one batch, then gene by gene: a gene on which the reference itself would panic must fail the same way in every form
            for g in [None] + list(range(ds.num_genes)):
                b = ds.batch() if g is None else ds.batch(gene_lo=g, gene_hi=g + 1)
                try:
                    b.run()
                    r = b.results()
                    per_gene.append((r.fasta, r.normal_fasta, r.tsv, r.windows))
                    r.close()
                except m.MicrophaserError as e:
                    per_gene.append(str(e))
                b.close()
            got.append(per_gene)
        assert got[0] == got[1] == got[2]
        assert sum(x[2].count(b"\n") for x in got[0] if not isinstance(x, str)) > 100
    for k in ("MP_K2A_ITEMS", "MP_K2A_CHUNKS"):
        monkeypatch.delenv(k, raising=False)


def test_gpu_chunked_overlapped_phasing_equals_the_single_batch(ctx):
    """pipeline.phase_chunked: gene chunks planned + uploaded on a second context of the GPU while the previous chunk is phased, downloaded
    and consumed - the concatenated streams are the single batch's bytes (genes are independent, src/microphasing.rs:895-942)."""
    from microphaser_amd.pipeline import phase_chunked
    from microphaser_amd.shard import merge_streams
    ds = ctx.synth(6262, 90, 30.0, 5.4)
    whole = ds.phase()
    for chunks in (2, 5):
        parts, windows = phase_chunked(ds, n_chunks=chunks)
        assert len(parts) == chunks and windows == whole.windows
        got = merge_streams([dict(fasta=r.fasta, normal_fasta=r.normal_fasta, tsv=r.tsv) for r in parts])
        assert (got["fasta"], got["normal_fasta"], got["tsv"]) == (whole.fasta, whole.normal_fasta, whole.tsv)
    assert whole.tsv.count(b"\n") > 500


# ---- the phase_gene-level boundary, cost-balanced shards and the multi-rank config E driver
@pytest.mark.parametrize("name", ["test_reverse", "splice_reverse_test", "splice_forward_test"])
def test_gpu_decoded_records_in_reference_output_out(ctx, name):
    """mp_dataset_from_arrays (include/microphaser_hip.h mp_gene_batch; reference seam src/microphasing.rs:882-893): a host that
    keeps its own readers hands over decoded records - here the fixture's, copied into host-owned numpy arrays - and gets the
    reference's expected bytes."""
    p = fixture_paths(name)
    arrays = ctx.load(p["bam"], p["vcf"], p["fasta"], p["gtf"]).to_arrays()
    res = ctx.from_arrays(arrays).phase()
    exp = read_expected(p["expected"])
    assert (res.fasta, res.normal_fasta, res.tsv) == (exp["fa"], exp["normal.fa"], exp["tsv"])


def test_gpu_decoded_records_normal_mode(ctx):
    import microphaser_amd as m
    bam, vcf, gtf, fa, exp = NORMAL_FIXTURES["splice_forward_test"]
    d = os.path.join(GOLDEN, "splice_forward_test")
    arrays = ctx.load(os.path.join(d, bam), os.path.join(d, vcf), os.path.join(d, fa), os.path.join(d, gtf)).to_arrays(mode=m.MODE_NORMAL)
    res = ctx.from_arrays(arrays).phase(mode=m.MODE_NORMAL)
    assert res.fasta == open(os.path.join(d, "expected_output", exp), "rb").read()


def test_gpu_cost_balanced_shards_merge_into_gtf_order(ctx):
    """One exome, genes dealt by cost (lpt_partition on mp_dataset_gene_costs) to three shards, each phased as its own batch
    (mp_batch_create_genes) - on one GPU here - and merged by the per-gene offsets (mp_results_gene_offsets): byte-identical to
    the single-batch run, including shards generated on their own (per-gene random streams)."""
    from microphaser_amd.shard import lpt_partition, merge_by_gene, shard_of
    seed, n = 61, 30
    ds = ctx.synth(seed, n, gene_streams=True)
    whole = ds.phase()
    parts = lpt_partition(ds.gene_costs(), 3)
    assert all(len(p) >= 5 for p in parts) and parts[0] != list(range(len(parts[0])))
    shards = []
    for p in parts:
        b = ds.batch_genes(p)
        b.run()
        shards.append(shard_of(b.results(), p))
    merged = merge_by_gene(shards)
    assert (merged["fasta"], merged["normal_fasta"], merged["tsv"], merged["windows"]) == (whole.fasta, whole.normal_fasta, whole.tsv, whole.windows)
    assert whole.tsv.count(b"\n") > 500
    # the same shards from ranks that generated only their genes
    parts2 = lpt_partition(ctx.synth_gene_costs(seed, n), 3)
    shards2 = []
    for p in parts2:
        sub = ctx.synth(seed, n, keep=p)
        b = sub.batch_genes(range(len(p)))
        b.run()
        shards2.append(shard_of(b.results(), p))
    merged2 = merge_by_gene(shards2)
    assert (merged2["fasta"], merged2["normal_fasta"], merged2["tsv"]) == (whole.fasta, whole.normal_fasta, whole.tsv)


def test_gpu_translate_entry_matches_build_reference(ctx, tmp_path):
    """mp_translate (to_protein, src/peptides.rs:128-146) on raw windows against build_reference's translation of the same FASTA."""
    fa = open(os.path.join(GOLDEN, "test_build", "reference.fa"), "rb").read()
    exp = open(os.path.join(GOLDEN, "test_build", "expected_output", "reference_peptides.fasta"), "rb").read()
    nt, rev = b"", []
    lines = fa.decode().split("\n")
    for i in range(0, len(lines) - 1, 2):
        rid, seq = lines[i][1:], lines[i + 1]
        for k in range(0, len(seq) - 12 + 1, 3):
            nt += seq[k:k + 12].encode()
            rev.append(0 if rid.endswith("F") else 1)
    aa, keys = ctx.translate(nt, rev, 4)
    want = b"".join(l.encode() for l in exp.decode().split("\n")[1::2])
    assert aa == want and len(keys) == len(rev)
    import microphaser_amd as m
    assert [m.key_to_peptide(k, 4).encode() for k in keys] == [want[i:i + 4] for i in range(0, len(want), 4)]


TWO_RANK_E = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import torch.distributed as dist
import microphaser_amd as m
from microphaser_amd.shard import lpt_partition
from microphaser_amd.pipeline import config_e_rank
dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
rank, world = dist.get_rank(), dist.get_world_size()
seed, n, L = 1212, 24, 9
ctx = m.Context(0)                                  # both ranks on the box's one GPU: a rehearsal of the 2-GPU run
parts = lpt_partition(ctx.synth_gene_costs(seed, n, 30.0, 5.4), world)
mine = parts[rank]
ds = ctx.synth(seed, n, 30.0, 5.4, keep=mine)
merged, peptidome, filtered = config_e_rank(ctx, ds, list(range(len(mine))), mine, L, dist)
open(os.path.join(%(tmp)r, "keys%%d.json" %% rank), "w").write(json.dumps(peptidome.keys))
if rank == 0:
    for name, data in (("fa", merged["fasta"]), ("normal.fa", merged["normal_fasta"]), ("tsv", merged["tsv"]), ("f.fa", filtered.fasta),
                       ("f.tsv", filtered.tsv), ("f.removed.tsv", filtered.removed_tsv)):
        open(os.path.join(%(tmp)r, "e." + name), "wb").write(data)
dist.barrier()
dist.destroy_process_group()
'''


def test_gpu_two_rank_config_e_equals_the_single_rank_pipeline(ctx, tmp_path):
    """BASELINE config E as two ranks (both on this box's GPU, gloo for the exchanges): normal shards -> per-rank build_reference ->
    all-gather of the key tensors -> mp_peptides_union -> somatic shards -> tensor gather -> merge by gene -> filter, against the
    same pipeline run by one process on the whole exome."""
    import sys
    import microphaser_amd as m
    from conftest import ROOT
    port = 33500 + (os.getpid() % 2000)
    script = tmp_path / "rank.py"
    script.write_text(TWO_RANK_E % dict(root=ROOT, port=port, tmp=str(tmp_path)))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err.decode()[-3000:]
    seed, n, L = 1212, 24, 9
    ds = ctx.synth(seed, n, 30.0, 5.4, gene_streams=True)
    nres = ds.phase(window_len=3 * L, mode=m.MODE_NORMAL)
    pep = ctx.build_reference(nres.fasta, L)
    sres = ds.phase(window_len=3 * L)
    f = ctx.filter(sres.tsv, pep.binary, L)
    for r in range(2):
        assert json.loads((tmp_path / ("keys%d.json" % r)).read_text()) == pep.keys
    got = {name: (tmp_path / ("e." + name)).read_bytes() for name in ("fa", "normal.fa", "tsv", "f.fa", "f.tsv", "f.removed.tsv")}
    assert (got["fa"], got["normal.fa"], got["tsv"]) == (sres.fasta, sres.normal_fasta, sres.tsv)
    assert (got["f.fa"], got["f.tsv"], got["f.removed.tsv"]) == (f.fasta, f.tsv, f.removed_tsv)
    assert len(pep.keys) > 1000 and f.kept > 20


def test_gpu_context_survives_a_failed_allocation(ctx, monkeypatch):
    """an allocation failure during upload / buffer growth must leave the context usable: the failed call reports an
    error, and the batch that was resident before runs again (re-uploaded) with identical results."""
    import microphaser_amd as m
    ds = ctx.synth(23, 10)
    b = ds.batch()
    b.run()
    want = b.results().tsv
    monkeypatch.setenv("MP_TEST_ALLOC_LIMIT", "4096")
    with pytest.raises(m.MicrophaserError, match="HIP error"):
        ds.batch(gene_lo=0, gene_hi=5)
    monkeypatch.delenv("MP_TEST_ALLOC_LIMIT")
    b.run()
    assert b.results().tsv == want
    monkeypatch.setenv("MP_TEST_SMALL_CAPS", "1")
    b2 = ds.batch()
    monkeypatch.setenv("MP_TEST_ALLOC_LIMIT", "60000")      # the first pass overflows the tiny result buffers; growing them fails
    with pytest.raises(m.MicrophaserError):
        b2.run()
    monkeypatch.delenv("MP_TEST_ALLOC_LIMIT")
    monkeypatch.delenv("MP_TEST_SMALL_CAPS")
    b.run()
    assert b.results().tsv == want


def test_gpu_cli_reports_a_failed_write(built, tmp_path):
    """a write error must end in exit status 1 (the reference propagates writer errors to main, src/main.rs:260-265)."""
    p = fixture_paths("test_forward")
    cli = os.path.join(os.path.dirname(os.path.abspath(__import__("microphaser_amd").LIB_PATH)), "microphaser")
    with open(p["gtf"], "rb") as gtf, open("/dev/full", "wb") as full:
        r = subprocess.run([cli, "somatic", p["bam"], "--ref", p["fasta"], "--variants", p["vcf"], "--tsv", str(tmp_path / "t.tsv"),
                            "--normal-output", str(tmp_path / "n.fa")], stdin=gtf, stdout=full, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"cannot write" in r.stderr
    with open(p["gtf"], "rb") as gtf:
        r = subprocess.run([cli, "somatic", p["bam"], "--ref", p["fasta"], "--variants", p["vcf"], "--tsv", "/dev/full",
                            "--normal-output", str(tmp_path / "n.fa")], stdin=gtf, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"cannot write /dev/full" in r.stderr


def test_gpu_normal_mode_dense_long_reads_rerun_with_the_large_tables(ctx, tmp_path):
    """VERDICT r1 weak 1e / fuzz finding 7: `normal` on 250-nt reads over a variant every ~1.6 nt holds more than 128 live column
    epochs per transcript - the first pass overflows the small per-wave tables, the batch runs again with the large ones (512
    epochs, 2048 haplotypes per window) and must then be the oracle's output."""
    import microphaser_amd as m
    seed, n, depth, spacing, rl = 70311, 5, 10.0, 1.6, 250
    prefix = os.path.join(str(tmp_path), "dense")
    r = subprocess.run([ORACLE_CLI, "synth", "--mode", "normal", "--seed", str(seed), "--transcripts", str(n), "--depth", str(depth),
                        "--spacing", str(spacing), "--read-len", str(rl), "--prefix", prefix], capture_output=True, check=True)
    st = json.loads(r.stdout)
    ds = ctx.synth(seed, n, depth, spacing, read_len=rl)
    b = ds.batch(mode=m.MODE_NORMAL)
    stats = b.run()
    res = b.results()
    assert stats.attempts >= 2, "this exome was meant to overflow the small tables"
    assert res.windows == st["windows"]
    assert res.fasta == open(prefix + ".fa", "rb").read()
    assert res.tsv == open(prefix + ".tsv", "rb").read()


def test_gpu_windows_deeper_than_1024_reads(ctx, tmp_path):
    """ADVICE r1 / VERDICT r1 weak 13: a locus with more than 1024 simultaneously live reads per window (1500x here) made the
    whole run fail; such exons now stream their rows through k2w_window_rows_deep (hash counting, no bound on the rows)."""
    exp, st = oracle_synth(str(tmp_path), 818, 3, 1500.0, 9.0)
    ds = ctx.synth(818, 3, 1500.0, 9.0)
    b = ds.batch()
    stats = b.run()
    res = b.results()
    assert stats.n_steps_seq == 0 and stats.n_windows_wave > 100     # window-parallel, and not the lane kernel's (rows > 255)
    assert max(int(l.split(b"\t")[8]) for l in exp["tsv"].split(b"\n")[1:] if l) > 1024   # the depth column
    assert res.windows == st["windows"]
    assert (res.fasta, res.normal_fasta, res.tsv) == (exp["fa"], exp["normal.fa"], exp["tsv"])


def _oracle_synth_args(tmp, tag, seed, n, depth, spacing, extra):
    prefix = os.path.join(tmp, "oracle_%s_%d_%d" % (tag, seed, n))
    r = subprocess.run([ORACLE_CLI, "synth", "--seed", str(seed), "--transcripts", str(n), "--depth", str(depth), "--spacing", str(spacing),
                        "--skip-panics", "--prefix", prefix] + [str(x) for x in extra], capture_output=True, check=True)
    return {e: open(prefix + "." + e, "rb").read() for e in ("fa", "normal.fa", "tsv")}, json.loads(r.stdout)


def _phase_without(ds, skipped):
    """All three streams of the genes the reference would not panic on (the oracle harness drops the others), window and planned-transcript counts."""
    from microphaser_amd.shard import merge_streams
    parts, windows, planned_tx, lo = [], 0, 0, 0
    for g in skipped + [ds.num_genes]:
        if g > lo:
            b = ds.batch(gene_lo=lo, gene_hi=g)
            planned_tx += b.run().n_transcripts
            r = b.results()
            parts.append(dict(fasta=r.fasta, normal_fasta=r.normal_fasta, tsv=r.tsv))
            windows += r.windows
        lo = g + 1
    return merge_streams(parts), windows, planned_tx


def test_gpu_deep_genes_are_phased_as_read_subsets_forced_on_a_shallow_exome(ctx, tmp_path, monkeypatch):
    """the sequential replay (indel / multi-allelic columns, exon chains) keeps its rows in the 1024 slots of one wave;
    a gene that can exceed them is planned as copies holding disjoint subsets of its reads, whose per-window counts the consumer adds up
    (rows are independent of each other, src/microphasing.rs:297-343). Here the slot count is forced down to 24 (MP_TEST_ROW_SLOTS), so
    that nearly every gene of an ordinary 30x exome with indels, multi-allelic sites, soft-masked stretches and same-name mates (`contains`
    on the '-' strand, :281-294: such reads must stay in one subset) is split - into 2..3 copies - and must still give the oracle's bytes."""
    extra = ["--indel-rate", 0.08, "--multiallelic-rate", 0.05, "--softmask-rate", 0.2, "--mate-rate", 0.3]
    exp, st = _oracle_synth_args(str(tmp_path), "split", 1717, 40, 30.0, 5.4, extra)
    ds = ctx.synth(1717, 40, indel_rate=0.08, multiallelic_rate=0.05, softmask_rate=0.2, mate_rate=0.3)
    plain, windows0, tx0 = _phase_without(ds, st["skipped"])
    monkeypatch.setenv("MP_TEST_ROW_SLOTS", "24")
    got, windows, tx1 = _phase_without(ds, st["skipped"])
    monkeypatch.delenv("MP_TEST_ROW_SLOTS")
    assert tx1 >= tx0 + 20                      # most genes were planned more than once
    assert windows == windows0 == st["windows"]
    assert got == plain
    assert (got["fasta"], got["normal_fasta"], got["tsv"]) == (exp["fa"], exp["normal.fa"], exp["tsv"])
    assert exp["tsv"].count(b"\n") > 100


def test_gpu_amplicon_deep_exons_with_indels_match_the_oracle(ctx, tmp_path):
    """A 1500x locus with a 5 % indel rate: its indel exons need the sequential replay with more live reads than one wave has row slots
    (1024). This used to fail the WHOLE batch (device.cpp: "more than 1024 simultaneously live reads"); now the deep genes are phased as
    read subsets and everything comes out byte-identical to the oracle."""
    exp, st = _oracle_synth_args(str(tmp_path), "deep", 919, 3, 1500.0, 9.0, ["--indel-rate", 0.05])
    ds = ctx.synth(919, 3, 1500.0, 9.0, indel_rate=0.05)
    got, windows, planned_tx = _phase_without(ds, st["skipped"])
    assert len(st["skipped"]) < 3
    assert planned_tx > 3 - len(st["skipped"])                                                # at least one gene went in as several copies
    assert max(int(l.split(b"\t")[8]) for l in exp["tsv"].split(b"\n")[1:] if l) > 1024      # the depth column
    assert windows == st["windows"]
    assert (got["fasta"], got["normal_fasta"], got["tsv"]) == (exp["fa"], exp["normal.fa"], exp["tsv"])
