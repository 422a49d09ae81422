#!/usr/bin/env python3
"""Build tests/golden/ from the reference's own test fixtures (run once, in the container that
has /root/reference; the results are committed).

What it does
  1. copies the DATA files of the reference's live tests (BAM/VCF/GTF inputs and the expected
     .fa/.normal.fa/.tsv outputs; tests/lib.rs:106-342) into tests/golden/<test>/ ;
  2. the reference's tests download hg38 chromosomes at test time (tests/lib.rs:79-104) - there is
     no network here, so for every fixture gene a *mini reference FASTA* is reconstructed from the
     BAM records' MD tags (+ the expected `normal` outputs where they tile the CDS), upper-case,
     'N' where nothing covers;  the case of the bases (hg38 soft-masking) is overlaid from the
     expected normal_sequence / mutant_sequence columns;
  3. writes a region-offset .fai (6th/7th column = region start / stored length; an extension our
     IndexedFasta understands) so that genome coordinates stay those of hg38.

No reference SOURCE is copied - only data files and derived data.
"""
import os
import re
import shutil
import struct
import sys
import zlib

REF = "/root/reference/tests/resources"
HERE = os.path.dirname(os.path.abspath(__file__))

FIXTURES = {
    "test_forward": dict(
        files=["forward_test.bam", "forward_test.bam.bai", "forward_test.vcf", "forward_test.gtf",
               "forward_test.germline.vcf", "empty_test.vcf"],
        expected=["forward_test.fa", "forward_test.normal.fa", "forward_test.tsv", "forward_test.germline.fa"],
        bam="forward_test.bam", chrom="chr14", fai="chr14.fa.fai", tsv=["forward_test.tsv"],
        germline_fa=("forward_test.germline.fa", "forward_test.gtf"),
    ),
    "test_reverse": dict(
        files=["reverse_test.bam", "reverse_test.bam.bai", "reverse_test.vcf", "reverse_test.gtf",
               "reverse_test.germline.vcf", "reverse_test_germline.gtf"],
        expected=["reverse_test.fa", "reverse_test.normal.fa", "reverse_test.tsv", "reverse_test.germline.fa"],
        bam="reverse_test.bam", chrom="chr1", fai="chr1.fa.fai", tsv=["reverse_test.tsv"],
        # the disabled test_reverse_germline (tests/lib.rs:309-320): its expectation is placed by id and overlays the hg38
        # soft-mask case of the last CDS exon; a derived single-exon GTF makes the current source reach that exon
        germline_overlay=("reverse_test.germline.fa", "reverse_test_germline.gtf", "reverse_test.germline.vcf"),
        last_exon_gtf=("reverse_test_germline.gtf", "reverse_test_germline.last_exon.gtf", 26282302, 26282423),
    ),
    "splice_forward_test": dict(
        files=["INSIG1.test.bam", "INSIG1.test.bam.bai", "INSIG1.test.vcf", "INSIG1.test.gtf", "INSIG1.test.germline.vcf"],
        expected=["splice_forward_test.fa", "splice_forward_test.normal.fa", "splice_forward_test.tsv",
                  "splice_forward_test.germline.fa"],
        bam="INSIG1.test.bam", chrom="chr7", fai="chr7.fa.fai", tsv=["splice_forward_test.tsv"],
        germline_fa=("splice_forward_test.germline.fa", "INSIG1.test.gtf"),
    ),
    "splice_reverse_test": dict(
        files=["MMS22L.test.bam", "MMS22L.test.bam.bai", "MMS22L.test.vcf", "MMS22L.test.gtf"],
        expected=["splice_reverse_test.fa", "splice_reverse_test.normal.fa", "splice_reverse_test.tsv"],
        bam="MMS22L.test.bam", chrom="chr6", fai="chr6.fa.fai", tsv=["splice_reverse_test.tsv"],
    ),
    # disabled upstream (tests/lib.rs:384-408), old 20-column TSV layout: NON-GATING report only. BAM / VCF use bare contig
    # names and have no .bai; three_way_splice's GTF says "chr19" -> a derived copy with the contig renamed.
    "frameshift_test": dict(
        files=["frameshift_test.bam", "frameshift_test.vcf", "frameshift_test.gtf"],
        expected=["frameshift_test.mt.fa", "frameshift_test.wt.fa", "frameshift_test.tsv"],
        bam="frameshift_test.bam", chrom="11", fai=None, tsv=[], compact=True,
    ),
    "three_way_splice": dict(
        files=["three_way_splice.bam", "three_way_splice.vcf", "three_way_splice.gtf"],
        expected=["three_way_splice.mt.fa", "three_way_splice.wt.fa", "three_way_splice.tsv"],
        bam="three_way_splice.bam", chrom="19", fai=None, tsv=[], compact=True,
        rename_gtf=("three_way_splice.gtf", "three_way_splice.contig19.gtf", "chr19", "19"),
    ),
    "test_empty": dict(
        files=["empty_test.vcf"],
        expected=["empty_test.fa", "empty_test.normal.fa", "empty_test.tsv"],
    ),
    "test_unsorted_gtf": dict(files=["chr14.sorted.DHRS2_BDKRB2.gtf", "chr14.unsorted.BDKRB2_DHRS2.gtf", "empty.vcf",
                                     "forward_test.bam", "forward_test.bam.bai"], expected=[]),
    "test_build": dict(files=["reference.fa"], expected=["reference_peptides.fasta"]),
    "test_filter": dict(files=["info.tsv", "reference.binary"],
                        expected=["tumor.filtered.fa", "normal.filtered.fa", "info.filtered.tsv"]),
    "test_filter_long": dict(files=["info.tsv", "reference.binary"],
                             expected=["tumor.filtered_long.fa", "normal.filtered_long.fa", "info.filtered_long.tsv"]),
    "test_filter_fs": dict(files=["info.tsv", "reference.binary"],
                           expected=["tumor.filtered_fs.fa", "normal.filtered_fs.fa", "info.filtered_fs.tsv"]),
}


def bgzf_blocks(data):
    off = 0
    while off < len(data):
        xlen = struct.unpack_from("<H", data, off + 10)[0]
        bsize = None
        x = 0
        while x < xlen:
            si1, si2, slen = struct.unpack_from("<BBH", data, off + 12 + x)
            if si1 == 66 and si2 == 67:
                bsize = struct.unpack_from("<H", data, off + 12 + x + 4)[0] + 1
            x += 4 + slen
        cdata = data[off + 12 + xlen: off + bsize - 8]
        yield zlib.decompress(cdata, -15)
        off += bsize


def read_bam(path):
    raw = b"".join(bgzf_blocks(open(path, "rb").read()))
    assert raw[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, p)[0]
    p += 4
    refs = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", raw, p)[0]
        name = raw[p + 4:p + 4 + l_name - 1].decode()
        l_ref = struct.unpack_from("<i", raw, p + 4 + l_name)[0]
        refs.append((name, l_ref))
        p += 8 + l_name
    recs = []
    while p < len(raw):
        bs = struct.unpack_from("<i", raw, p)[0]
        r = raw[p + 4:p + 4 + bs]
        p += 4 + bs
        ref_id, pos, l_rn, mapq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", r, 0)
        q = 32
        name = r[q:q + l_rn - 1].decode()
        q += l_rn
        cigar = [(c >> 4, "MIDNSHP=X"[c & 0xF]) for c in struct.unpack_from("<%dI" % n_cig, r, q)]
        q += 4 * n_cig
        seq4 = r[q:q + (l_seq + 1) // 2]
        seq = "".join("=ACMGRSVTWYHKDBN"[(seq4[i >> 1] >> (4 if i % 2 == 0 else 0)) & 0xF] for i in range(l_seq))
        q += (l_seq + 1) // 2
        q += l_seq
        tags = {}
        while q < len(r):
            tag = r[q:q + 2].decode()
            t = chr(r[q + 2])
            q += 3
            if t == "Z":
                e = r.index(b"\0", q)
                tags[tag] = r[q:e].decode()
                q = e + 1
            elif t in "cC":
                q += 1
            elif t in "sS":
                q += 2
            elif t in "iIf":
                q += 4
            elif t == "A":
                q += 1
            elif t == "H":
                e = r.index(b"\0", q)
                q = e + 1
            elif t == "B":
                sub = chr(r[q])
                n = struct.unpack_from("<i", r, q + 1)[0]
                q += 5 + n * {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[sub]
            else:
                raise ValueError("tag type " + t)
        recs.append(dict(ref=refs[ref_id][0] if ref_id >= 0 else None, pos=pos, flag=flag, cigar=cigar, seq=seq,
                         md=tags.get("MD"), name=name))
    return refs, recs


def ref_from_md(rec, out):
    """Write reference bases covered by the M/=/X/D ops of one record into dict out[pos]=base."""
    if rec["md"] is None or rec["flag"] & 4 or not rec["cigar"]:
        return
    # aligned read bases (M ops only) in order, and the reference positions of M and D ops
    qpos = 0
    rpos = rec["pos"]
    aligned = []  # (refpos, readbase) for M ops
    dels = []     # refpos for D ops
    for l, op in rec["cigar"]:
        if op in "M=X":
            for k in range(l):
                aligned.append((rpos + k, rec["seq"][qpos + k]))
            rpos += l
            qpos += l
        elif op in "IS":
            qpos += l
        elif op == "D":
            dels.extend(range(rpos, rpos + l))
            rpos += l
        elif op == "N":
            rpos += l
    # walk MD
    ai = 0
    for m in re.finditer(r"(\d+)|(\^[A-Za-z]+)|([A-Za-z])", rec["md"]):
        if m.group(1) is not None:
            n = int(m.group(1))
            for _ in range(n):
                p, b = aligned[ai]
                out.setdefault(p, b)
                ai += 1
        elif m.group(2) is not None:
            for b in m.group(2)[1:]:
                p = dels.pop(0)
                out.setdefault(p, b.upper())
        else:
            p, _b = aligned[ai]
            out.setdefault(p, m.group(3).upper())
            ai += 1


def parse_gtf_cds(path):
    """CDS intervals (0-based half-open) per transcript, with start_codon / three_prime_utr adjustments
    NOT applied (only used to place the `normal` germline.fa tiles)."""
    tx = {}
    for line in open(path):
        if line.startswith("#"):
            continue
        f = line.rstrip("\n").split("\t")
        if f[2] != "CDS":
            continue
        tid = re.search(r'transcript_id "([^"]+)"', f[8]).group(1)
        tx.setdefault(tid, dict(strand=f[6], cds=[]))["cds"].append((int(f[3]) - 1, int(f[4]), f[7]))
    return tx


def main():
    for name, fx in FIXTURES.items():
        d = os.path.join(HERE, name)
        os.makedirs(os.path.join(d, "expected_output"), exist_ok=True)
        for f in fx["files"]:
            src = os.path.join(REF, name, f)
            if not os.path.exists(src) and name == "test_empty":
                src = os.path.join(REF, "test_empty", f)
            shutil.copyfile(src, os.path.join(d, f))
            os.chmod(os.path.join(d, f), 0o644)
        for f in fx["expected"]:
            shutil.copyfile(os.path.join(REF, name, "expected_output", f), os.path.join(d, "expected_output", f))
            os.chmod(os.path.join(d, "expected_output", f), 0o644)
        if "bam" not in fx:
            continue
        refs, recs = read_bam(os.path.join(REF, name, fx["bam"]))
        chrom = fx["chrom"]
        bases = {}
        for r in recs:
            if r["ref"] == chrom:
                ref_from_md(r, bases)
        # gene span from the GTF copies
        gtf = [f for f in fx["files"] if f.endswith(".gtf")][0]
        lo, hi = 1 << 62, 0
        for line in open(os.path.join(REF, name, gtf)):
            f = line.split("\t")
            if len(f) > 4 and f[2] == "gene":
                lo = min(lo, int(f[3]) - 1)
                hi = max(hi, int(f[4]) + 100)
        lo = max(0, lo - 200)
        hi += 200
        seq = [bases.get(p, "N") for p in range(lo, hi)]
        # case overlay + gap filling from expected TSV rows (plain, non-merged windows only)
        n_case = 0
        for tsv in fx["tsv"]:
            rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(REF, name, "expected_output", tsv))]
            hdr = rows[0]
            for row in rows[1:]:
                rec = dict(zip(hdr, row))
                start = int(rec["offset"]) - 1
                for col in ("normal_sequence", "mutant_sequence"):
                    s = rec[col]
                    if not s:
                        continue
                    # only use rows that are linear in the genome (plain windows; splice-merged rows are not)
                    germ = {int(x) - 1 for x in rec["germline_positions"].split("|")} if rec["germline_positions"] else set()
                    som = {int(x) - 1 for x in rec["somatic_positions"].split("|")} if rec["somatic_positions"] else set()
                    # positions whose printed base is an ALT (case says nothing direct about the reference)
                    alt_pos = germ | (som if col == "mutant_sequence" else set())

                    def place(chars, positions):
                        nonlocal n_case
                        for ch, p in zip(chars, positions):
                            q = p - lo
                            if q < 0 or q >= len(seq):
                                continue
                            if p in alt_pos:
                                # switch_ascii_case: ALT printed lower-case iff the reference base is upper-case;
                                # an upper-case ALT therefore proves a lower-case (soft-masked) reference base
                                if ch.isupper() and seq[q] != "N" and seq[q].isupper():
                                    seq[q] = seq[q].lower()
                                    n_case += 1
                                continue
                            if seq[q] == "N":
                                seq[q] = ch
                            elif seq[q].upper() == ch.upper() and seq[q] != ch:
                                seq[q] = ch
                                n_case += 1

                    def matches(chars, positions):
                        bad = 0
                        for ch, p in zip(chars, positions):
                            q = p - lo
                            if q < 0 or q >= len(seq):
                                return False
                            if p not in alt_pos and seq[q] != "N" and seq[q].upper() != ch.upper():
                                bad += 1
                        return bad == 0

                    if len(s) == 27:
                        pos = list(range(start, start + 27))
                        if matches(s, pos):
                            place(s, pos)
                    elif len(s) < 27 and germ:
                        # a germline deletion inside the window: the part up to and including the anchor base is
                        # aligned at the window start, the rest at the window end (= start + 27)
                        for dp in sorted(germ):
                            k = dp - start + 1
                            if k <= 0 or k >= len(s):
                                continue
                            left, right = s[:k], s[k:]
                            lpos = list(range(start, start + k))
                            rpos = list(range(start + 27 - len(right), start + 27))
                            if matches(left, lpos) and matches(right, rpos):
                                place(left, lpos)
                                place(right, rpos)
                                break
        # gap filling from the `normal` outputs that tile the CDS every 3 nt (forward fixtures): every record
        # is placed EXACTLY by re-deriving its id = sha1("{:?}{}{}" % (seq bytes, transcript_id, offset))[:15]+'F'
        # (reference: src/normal_microphasing.rs:509-517) over all candidate offsets.
        if "germline_fa" in fx:
            import hashlib
            gfa, ggtf = fx["germline_fa"]
            lines = [l.strip() for l in open(os.path.join(REF, name, "expected_output", gfa))]
            tiles = [(lines[i][1:], lines[i + 1]) for i in range(0, len(lines) - 1, 2)]
            txs = parse_gtf_cds(os.path.join(REF, name, ggtf))
            placed = 0
            for tid, t in txs.items():
                cand = []
                for (a, b, _fr) in t["cds"]:
                    cand.extend(range(a - 3, b))
                for rid, tile in tiles:
                    dbg = "[" + ", ".join(str(ord(c)) for c in tile) + "]"
                    for a in cand:
                        h = hashlib.sha1((dbg + tid + str(a)).encode()).hexdigest()[:15] + "F"
                        if h == rid:
                            placed += 1
                            for k, ch in enumerate(tile):
                                if seq[a + k - lo] == "N" and ch.isupper():
                                    seq[a + k - lo] = ch
                            break
            print("  %s: %d/%d `normal` tiles placed by id" % (gfa, placed, len(tiles)))
            # the remaining tiles are splice-merged windows (last bases of one CDS exon + first bases of the next,
            # src/normal_microphasing.rs:1164-1233): use them to fill the few boundary bases no read covers
            filled = 0
            for tid, t in txs.items():
                cds = sorted((a, b) for (a, b, _fr) in t["cds"])
                for rid, tile in tiles:
                    if len(tile) != 27:
                        continue
                    for (a0, a1), (b0, b1) in zip(cds, cds[1:]):
                        for k in range(1, 27):
                            pos = list(range(a1 - (27 - k), a1)) + list(range(b0, b0 + k))
                            if pos[0] < a0 or pos[-1] >= b1:
                                continue
                            known = mism = 0
                            for ch, q in zip(tile, pos):
                                base = seq[q - lo]
                                if base == "N":
                                    continue
                                known += 1
                                if base.upper() != ch.upper():
                                    mism += 1
                            if known >= 20 and mism == 0 and known < 27:
                                for ch, q in zip(tile, pos):
                                    if seq[q - lo] == "N" and ch.isupper():
                                        seq[q - lo] = ch
                                        filled += 1
            print("  %s: %d boundary bases filled from splice-merged tiles" % (gfa, filled))
        if "germline_overlay" in fx:
            # `normal` expectation of the '-' strand fixture: every linear record is placed by re-deriving its id over the
            # CDS span; bases that are not a variant position give the reference base AND its case (soft-masking)
            import hashlib
            gfa, ggtf, gvcf = fx["germline_overlay"]
            lines = [l.strip() for l in open(os.path.join(REF, name, "expected_output", gfa))]
            tiles = [(lines[i][1:], lines[i + 1]) for i in range(0, len(lines) - 1, 2)]
            vpos = {int(l.split("\t")[1]) - 1 for l in open(os.path.join(REF, name, gvcf)) if not l.startswith("#")}
            txs = parse_gtf_cds(os.path.join(REF, name, ggtf))
            placed = cased = 0
            for tid, t in txs.items():
                cand = []
                for (a, b, _fr) in t["cds"]:
                    cand.extend(range(a - 30, b + 3))
                for rid, tile in tiles:
                    dbg = "[" + ", ".join(str(ord(c)) for c in tile) + "]"
                    for a in cand:
                        if hashlib.sha1((dbg + tid + str(a)).encode()).hexdigest()[:15] + rid[-1] == rid:
                            placed += 1
                            for k, ch in enumerate(tile):
                                q = a + k - lo
                                if a + k in vpos or q < 0 or q >= len(seq):
                                    continue
                                if seq[q] == "N":
                                    seq[q] = ch
                                elif seq[q].upper() == ch.upper() and seq[q] != ch:
                                    seq[q] = ch
                                    cased += 1
                            break
            print("  %s: %d/%d `normal` records placed by id, %d case overlays" % (gfa, placed, len(tiles), cased))
        if "last_exon_gtf" in fx:
            src, dst, cstart, cend = fx["last_exon_gtf"]
            keep, cds = [], None
            for l in open(os.path.join(REF, name, src)):
                f = l.rstrip("\n").split("\t")
                if len(f) < 9:
                    continue
                if f[2] in ("gene", "transcript"):
                    keep.append(l)
                if f[2] == "CDS":
                    cds = f
            cds[3], cds[4], cds[7] = str(cstart), str(cend), "0"
            keep.append("\t".join(cds) + "\n")
            open(os.path.join(d, dst), "w").writelines(keep)
        if "rename_gtf" in fx:
            src, dst, old, new = fx["rename_gtf"]
            with open(os.path.join(d, dst), "w") as f:
                for l in open(os.path.join(REF, name, src)):
                    f.write(new + l[len(old):] if l.startswith(old + "\t") else l)
        if fx.get("compact"):
            # a 0.5 Mb gene covered by reads only at its exons: keep the span from the first to the last known base
            first = next(i for i, c in enumerate(seq) if c != "N")
            last = len(seq) - next(i for i, c in enumerate(reversed(seq)) if c != "N")
            first = max(0, first - 200)
            seq = seq[first:min(len(seq), last + 200)]
            lo += first
        n_unknown = sum(1 for c in seq if c == "N")
        fa = os.path.join(d, chrom + ".mini.fa")
        with open(fa, "w") as f:
            f.write(">%s\n" % chrom)
            f.write("".join(seq) + "\n")
        full_len = None
        if fx["fai"] is None:
            full_len = dict(refs)[chrom]
        else:
            for line in open(os.path.join(REF, fx["fai"])):
                c = line.split("\t")
                if c[0] == chrom:
                    full_len = int(c[1])
        with open(fa + ".fai", "w") as f:
            f.write("%s\t%d\t%d\t%d\t%d\t%d\t%d\n" % (chrom, full_len, len(chrom) + 2, len(seq), len(seq) + 1, lo, len(seq)))
        print("%s: %s:%d-%d  %d bases, %d unknown (N), %d case overlays" % (name, chrom, lo, hi, len(seq), n_unknown, n_case))


if __name__ == "__main__":
    sys.exit(main())
