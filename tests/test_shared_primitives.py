"""Primitives that the CPU oracle and the product both build on are pinned by INDEPENDENT statements of them, so that a shared mistake
cannot hide in synthetic-data parity (the oracle decodes files through the product's io.cpp):

* CIGAR -> read position (rust-htslib `CigarStringView::read_pos`, call sites src/microphasing.rs:106, src/normal_microphasing.rs:48):
  the product's host function (model.hpp cigar_read_pos; its device twin is covered by the GPU parity suite) against the oracle's OWN
  oracle_read_pos (oracle/oracle_util.hpp) on every read of the reference's fixture BAMs (M, S, I, D, H cigars) x every position, and on
  200 000 random CIGARs with every operation;
* VCF record -> Variant list (`Variant::new`, src/common.rs:16-175): the shared ingest's classification of every record of the fixture
  VCFs (SNV / insertion / deletion, multi-allelic, SOMATIC flag, ANN protein change) against a Python restatement written here from the
  reference source - plus a synthetic VCF with the allele shapes the fixtures lack (<DEL> with SVLEN, unsupported <INS>, MNV, deletion).
"""
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT, SOMATIC_FIXTURES, fixture_paths


@pytest.fixture(scope="module")
def check_exe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("prims") / "shared_prims_check")
    csrc = os.path.join(ROOT, "microphaser_amd", "csrc")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I", csrc, "-I", os.path.join(ROOT, "oracle"), "-o", exe,
                    os.path.join(ROOT, "tests", "shared_prims_check.cpp"), os.path.join(csrc, "io.cpp"), "-lz", "-lpthread"], check=True)
    return exe


def test_cigar_read_pos_product_and_oracle_statements_agree(check_exe):
    bams = [fixture_paths(n)["bam"] for n in SOMATIC_FIXTURES]
    r = subprocess.run([check_exe, "cigar"] + bams, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.splitlines()
    assert lines[0].startswith("fixture reads:") and lines[0].endswith(" 0 mismatches")
    assert int(lines[0].split()[2]) > 500000          # (8367 reads x ~110 positions)
    assert lines[1].startswith("random cigars:") and lines[1].endswith(" 0 mismatches")


def variant_new(fields):
    """Variant::new of the reference (src/common.rs:71-175) for one VCF data line, unsupported alleles as warnings (-u):
    list of (pos0, kind, alt, len, is_germline, seq, prot_change); kind 0 SNV, 1 insertion, 2 deletion."""
    chrom, pos, _id, ref, alt, _qual, _filt, info = fields[:8]
    info_kv = {}
    for item in info.split(";"):
        k, _, v = item.partition("=")
        info_kv[k] = v
    is_germline = "SOMATIC" not in info_kv                                   # :75
    ann = info_kv.get("ANN", "")
    first = ann.split(",")[0] if ann else ""                                 # Annotation::new :21-35: the first ANN value ...
    prot = ""
    for f in first.split("|"):                                               # ... its first |-field that contains "p."
        if "p." in f:
            prot = f
            break
    out = []
    pos0 = int(pos) - 1
    for a in alt.split(","):
        if len(a) == 1 and len(ref) > 1:                                     # :86-92
            out.append((pos0, 2, 0, len(ref) - 1, is_germline, "", prot))
        elif len(a) > 1 and len(ref) == 1:
            if a.startswith("<"):
                if a == "<DEL>":                                             # :95-142
                    sv = info_kv.get("SVLEN", "")
                    vals = [x for x in sv.split(",") if x != ""]
                    if len(vals) == 1 and vals[0] != ".":
                        out.append((pos0, 2, 0, abs(int(vals[0])), is_germline, "", prot))
                # every other case: warning only
            else:                                                            # :150-156
                out.append((pos0, 1, 0, len(a) - 1, is_germline, a, prot))
        elif len(a) == 1 and len(ref) == 1:                                  # :158-164
            out.append((pos0, 0, ord(a), 0, is_germline, "", prot))
        # else: MNV, dropped with a warning (:165-171)
    return chrom, out


def classify_with_python(vcf_path):
    rows = []
    for line in open(vcf_path):
        if line.startswith("#") or not line.strip():
            continue
        chrom, vs = variant_new(line.rstrip("\n").split("\t"))
        rows += [(chrom,) + v for v in vs]
    return rows


def classify_with_ingest(check_exe, vcf_path):
    r = subprocess.run([check_exe, "variants", vcf_path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = []
    for line in r.stdout.splitlines():
        c, pos, kind, alt, ln, germ, seq, prot = line.split("\t")
        rows.append((c, int(pos), int(kind), int(alt), int(ln), germ == "1", seq, prot))
    return rows


@pytest.mark.parametrize("name", sorted(SOMATIC_FIXTURES))
def test_variant_classification_of_the_fixture_vcfs(check_exe, name):
    vcf = fixture_paths(name)["vcf"]
    exp = classify_with_python(vcf)
    got = classify_with_ingest(check_exe, vcf)
    assert got == exp and len(exp) >= 1
    if name == "test_reverse":   # the multi-allelic record C -> A,CGGGACA and the 6-nt deletion
        kinds = sorted(set(k for _, _, k, *_ in exp))
        assert kinds == [0, 1, 2]
        assert sum(1 for r in exp if r[1] == 26282351) == 2


def test_variant_classification_of_the_other_allele_shapes(check_exe, tmp_path):
    vcf = tmp_path / "shapes.vcf"
    ann = "ANN=A|missense_variant|MODERATE|G1|G1|transcript|T1|protein_coding|1/2|c.10A>G|p.Lys4Glu|10/300|10/300|4/99||,A|second|x|p.Zzz9Yyy"
    vcf.write_text(
        "##fileformat=VCFv4.2\n##contig=<ID=chrS1>\n"
        "##INFO=<ID=SOMATIC,Number=0,Type=Flag,Description=\"s\">\n##INFO=<ID=SVLEN,Number=.,Type=Integer,Description=\"l\">\n"
        "##INFO=<ID=ANN,Number=.,Type=String,Description=\"a\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
        "chrS1\t100\t.\tA\tG\t.\tPASS\tSOMATIC;" + ann + "\n"
        "chrS1\t110\t.\tACGT\tA\t.\tPASS\t" + ann + "\n"
        "chrS1\t120\t.\tA\tACC\t.\tPASS\tSOMATIC\n"
        "chrS1\t130\t.\tA\t<DEL>\t.\tPASS\tSVLEN=-7\n"
        "chrS1\t140\t.\tA\t<DEL>\t.\tPASS\tSOMATIC\n"
        "chrS1\t150\t.\tA\t<INS>\t.\tPASS\tSVLEN=5\n"
        "chrS1\t160\t.\tAC\tGT\t.\tPASS\t.\n"
        "chrS1\t170\t.\tA\tC,AG,T\t.\tPASS\tSOMATIC\n"
        "chrS1\t180\t.\tAT\tA,ATG\t.\tPASS\t.\n")
    exp = classify_with_python(str(vcf))
    got = classify_with_ingest(check_exe, str(vcf))
    assert got == exp
    assert [(r[1], r[2], r[4]) for r in exp] == [(99, 0, 0), (109, 2, 3), (119, 1, 2), (129, 2, 7), (169, 0, 0), (169, 1, 1), (169, 0, 0), (179, 2, 1)]
    assert exp[0][7] == "p.Lys4Glu" and exp[0][5] is False and exp[1][5] is True
