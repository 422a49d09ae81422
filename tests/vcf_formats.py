"""Test helpers: re-encode a VCF text fixture as gzip, BGZF and BCF2 - written from the VCF/BCF specification (section 6, "BCF
specification") independently of the C++ reader in io.cpp, with zlib/struct only. Not a general converter: INFO values are encoded by the
header's declared Type, FORMAT/sample columns are dropped (n_sample = 0), which is all the reader under test looks at."""
import gzip, re, struct, zlib


def bgzf_bytes(data, block=0xff00):
    out = bytearray()
    for off in list(range(0, len(data), block)) + [None]:
        chunk = b"" if off is None else data[off:off + block]   # the final empty block is the EOF marker
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(chunk) + c.flush()
        bsize = len(comp) + 25
        out += struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize)
        out += comp + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk))
    return bytes(out)


def gzip_bytes(data):
    return gzip.compress(data)


def _typed_int_scalar(v):
    if -120 <= v <= 127: return struct.pack("<Bb", 0x11, v)
    if -32000 <= v <= 32767: return struct.pack("<Bh", 0x12, v)
    return struct.pack("<Bi", 0x13, v)


def _descriptor(n, t):
    return bytes([(n << 4) | t]) if n < 15 else bytes([0xF0 | t]) + _typed_int_scalar(n)


def _typed_str(s):
    b = s.encode()
    return _descriptor(len(b), 7) + b


def _typed_ints(vals):   # None = missing
    real = [v for v in vals if v is not None]
    if all(-120 <= v <= 127 for v in real): t, f, miss = 1, "b", -128
    elif all(-32000 <= v <= 32767 for v in real): t, f, miss = 2, "h", -32768
    else: t, f, miss = 3, "i", -2147483648
    return _descriptor(len(vals), t) + b"".join(struct.pack("<" + f, miss if v is None else v) for v in vals)


def vcf_to_bcf(text, with_idx=True):
    """VCF text (bytes) -> uncompressed BCF2.2 bytes (wrap with bgzf_bytes). with_idx: write IDX= in the header like htslib does;
    without it the reader has to number the dictionaries implicitly (PASS = 0, then order of appearance)."""
    lines = text.decode().split("\n")
    header = [l for l in lines if l.startswith("#")]
    records = [l for l in lines if l and not l.startswith("#")]
    strings, contigs, types = {"PASS": 0}, {}, {}
    new_header = []
    for l in header:
        m = re.match(r"##(INFO|FILTER|FORMAT)=<ID=([^,>]+)", l)
        c = re.match(r"##contig=<ID=([^,>]+)", l)
        if m:
            kind, id_ = m.group(1), m.group(2)
            if id_ not in strings: strings[id_] = len(strings)
            if kind == "INFO": types[id_] = re.search(r"Type=([A-Za-z]+)", l).group(1)
            if with_idx: l = l[:-1] + ",IDX=%d>" % strings[id_]
        elif c:
            contigs[c.group(1)] = len(contigs)
            if with_idx: l = l[:-1] + ",IDX=%d>" % contigs[c.group(1)]
        new_header.append(l)
    for r in records:   # contigs that only appear in records (the text reader adds them too)
        ch = r.split("\t")[0]
        if ch not in contigs:
            contigs[ch] = len(contigs)
            new_header.insert(-1, "##contig=<ID=%s%s>" % (ch, ",IDX=%d" % contigs[ch] if with_idx else ""))
    htext = ("\n".join(new_header) + "\n").encode() + b"\0"
    out = bytearray(b"BCF\2\2" + struct.pack("<I", len(htext)) + htext)
    for r in records:
        f = r.split("\t")
        chrom, pos, id_, ref, alt, qual, flt, info = f[:8]
        alleles = [ref] + ([] if alt == "." else alt.split(","))
        infos = [] if info in (".", "") else info.split(";")
        shared = bytearray()
        rlen = len(ref)
        shared += struct.pack("<iii", contigs[chrom], int(pos) - 1, rlen)
        shared += struct.pack("<I", 0x7F800001) if qual == "." else struct.pack("<f", float(qual))
        shared += struct.pack("<II", (len(alleles) << 16) | len(infos), 0)
        shared += _typed_str("" if id_ == "." else id_)
        for a in alleles: shared += _typed_str(a)
        if flt == ".": shared += bytes([0x00])
        else: shared += _typed_ints([strings[x] for x in flt.split(";")])
        for kv in infos:
            k, _, v = kv.partition("=")
            shared += _typed_int_scalar(strings[k])
            t = types.get(k, "String")
            if t == "Flag": shared += bytes([0x00])
            elif t == "Integer": shared += _typed_ints([None if x == "." else int(x) for x in v.split(",")])
            elif t == "Float":
                vals = v.split(",")
                shared += _descriptor(len(vals), 5) + b"".join(struct.pack("<I", 0x7F800001) if x == "." else struct.pack("<f", float(x)) for x in vals)
            else: shared += _typed_str(v)
        out += struct.pack("<II", len(shared), 0) + shared
    return bytes(out)
