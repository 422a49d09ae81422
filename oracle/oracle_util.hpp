// TEST INFRASTRUCTURE - the CPU oracle's OWN primitives (the product does not include this file: its host legs use rowfmt.hpp /
// hostsha.hpp, its kernels their own device code; tests/test_host_formatting.py and tests/test_shared_primitives.py compare the two).
// Small host utilities whose exact output format is part of the parity contract: SHA-1 ids, shortest-round-trip f64 printing (Rust
// `ryu` pretty format, used by the `csv` crate's serde serializer), FASTA and TSV record writers, and the oracle's own restatement of
// rust-htslib's CigarStringView::read_pos.
#pragma once
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../microphaser_amd/csrc/model.hpp"

namespace mp {

// ------------------------------------------------------------------ CIGAR
// rust-htslib 0.36 `CigarStringView::read_pos(ref_pos, include_softclips = false, include_dels = false)` (the crate is not vendored in the
// reference; call sites src/microphasing.rs:106, src/normal_microphasing.rs:48; semantics pinned by the fixtures' depth / freq columns,
// SURVEY 3.5.3 and Appendix B). The oracle's own statement of it, written as the walk over (op, length) pairs the crate does - kept apart
// from the product's cigar_read_pos (model.hpp) and its device twin so that a mistake in one does not hide in the other
// (tests/test_shared_primitives.py runs both over every read of the fixture BAMs). Returns -1 for `None`.
inline int64_t oracle_read_pos(const uint32_t* cigar, uint32_t n_ops, int64_t read_start, int64_t ref_pos) {
    enum { M = 0, I = 1, D = 2, N = 3, S = 4, H = 5, P = 6, EQ = 7, X = 8 };
    // the crate first looks for the first operation that consumes read bases; leading hard clips / pads are stepped over, a leading
    // deletion / skip is an error (the callers treat an error like None), a hard clip that is neither first nor last as well
    uint32_t first = n_ops;
    for (uint32_t k = 0; k < n_ops; k++) {
        const uint32_t op = cigar[k] & 15u;
        if (op == M || op == EQ || op == X || op == I || op == S) { first = k; break; }
        if (op == D || op == N) return -1;
        if (op == H && k != 0 && k + 1 != n_ops) return -1;
    }
    if (first == n_ops) return -1;
    int64_t ref_cursor = read_start, read_cursor = 0;
    for (uint32_t k = first; k < n_ops; k++) {
        if (ref_cursor > ref_pos) return -1;           // walked past the position without an aligned base on it
        const uint32_t op = cigar[k] & 15u;
        const int64_t len = int64_t(cigar[k] >> 4);
        if (op == M || op == EQ || op == X) {
            if (ref_pos < ref_cursor + len) return read_cursor + (ref_pos - ref_cursor);
            ref_cursor += len;
            read_cursor += len;
        } else if (op == I || op == S) {
            read_cursor += len;                       // soft clips count as read bases but (include_softclips = false) never match a position
        } else if (op == D || op == N) {
            ref_cursor += len;                        // a position inside a deletion has no read base (include_dels = false): the next loop turn returns None
        } else if (op == H) {
            return -1;                                // the trailing hard clip: nothing behind it
        } else if (op != P) {
            return -1;
        }
    }
    return -1;
}

// ------------------------------------------------------------------ SHA-1
// (reference: `sha1 = "0.6"` crate, call sites src/microphasing.rs:667-675, src/common.rs:387-395)
struct Sha1 {
    uint32_t h[5] = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u, 0xC3D2E1F0u};
    uint8_t buf[64];
    uint64_t total = 0;
    size_t fill = 0;
    static uint32_t rol(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
    void block(const uint8_t* p) {
        uint32_t w[80];
        for (int i = 0; i < 16; i++)
            w[i] = (uint32_t(p[4 * i]) << 24) | (uint32_t(p[4 * i + 1]) << 16) | (uint32_t(p[4 * i + 2]) << 8) | p[4 * i + 3];
        for (int i = 16; i < 80; i++) w[i] = rol(w[i - 3] ^ w[i - 8] ^ w[i - 14] ^ w[i - 16], 1);
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4];
        for (int i = 0; i < 80; i++) {
            uint32_t f, k;
            if (i < 20) { f = (b & c) | (~b & d); k = 0x5A827999u; }
            else if (i < 40) { f = b ^ c ^ d; k = 0x6ED9EBA1u; }
            else if (i < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDCu; }
            else { f = b ^ c ^ d; k = 0xCA62C1D6u; }
            uint32_t t = rol(a, 5) + f + e + k + w[i];
            e = d; d = c; c = rol(b, 30); b = a; a = t;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e;
    }
    void update(const void* data, size_t n) {
        const uint8_t* p = static_cast<const uint8_t*>(data);
        total += n;
        while (n) {
            size_t take = 64 - fill < n ? 64 - fill : n;
            std::memcpy(buf + fill, p, take);
            fill += take; p += take; n -= take;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    std::string hexdigest() {
        uint64_t bits = total * 8;
        uint8_t pad = 0x80;
        update(&pad, 1);
        uint8_t z = 0;
        while (fill != 56) update(&z, 1);
        uint8_t lenb[8];
        for (int i = 0; i < 8; i++) lenb[i] = uint8_t(bits >> (56 - 8 * i));
        update(lenb, 8);
        char out[40];
        for (int i = 0; i < 40; i++) out[i] = "0123456789abcdef"[(h[i >> 3] >> (28 - 4 * (i & 7))) & 0xF];
        return std::string(out, 40);
    }
};

// `format!("{:?}{}{}", &seq, transcript_id, offset)` -> sha1 -> first 15 hex chars + 'F'|'R'
// (reference: src/microphasing.rs:667-675). `{:?}` of Vec<u8> is "[65, 67, ...]".
inline std::string haplotype_id(const uint8_t* seq, size_t n, const std::string& transcript_id, uint64_t offset,
                                char strand_initial) {
    std::string s;
    s.reserve(5 * n + transcript_id.size() + 24);
    s.push_back('[');
    for (size_t i = 0; i < n; i++) {
        if (i) { s.push_back(','); s.push_back(' '); }
        const unsigned v = seq[i];
        if (v >= 100) s.push_back(char('0' + v / 100));
        if (v >= 10) s.push_back(char('0' + (v / 10) % 10));
        s.push_back(char('0' + v % 10));
    }
    s.push_back(']');
    s += transcript_id;
    s += std::to_string(offset);
    Sha1 sh;
    sh.update(s.data(), s.size());
    std::string id = sh.hexdigest().substr(0, 15);
    id.push_back(strand_initial);
    return id;
}

// ------------------------------------------------------------------ f64 printing
// Rust ryu::Buffer::format_finite "pretty" layout (what csv+serde write for an f64 field).
inline std::string fmt_f64(double v) {
    if (v != v) return "NaN";
    if (v == 1.0 / 0.0) return "inf";
    if (v == -1.0 / 0.0) return "-inf";
    std::string out;
    if (std::signbit(v)) { out.push_back('-'); v = -v; }
    if (v == 0.0) { out += "0.0"; return out; }
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);  // shortest round-trip
    std::string s(buf, r.ptr);
    // parse d.ddddde[+-]xx
    size_t epos = s.find('e');
    std::string mant = s.substr(0, epos);
    int exp10 = std::atoi(s.c_str() + epos + 1);
    std::string digits;
    for (char c : mant)
        if (c != '.') digits.push_back(c);
    // strip trailing zeros (to_chars shortest never emits them, but be safe)
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    long length = long(digits.size());
    long kk = exp10 + 1;        // 10^(kk-1) <= v < 10^kk
    long k = kk - length;       // v = digits * 10^k
    if (0 <= k && kk <= 16) {
        out += digits;
        out.append(size_t(kk - length), '0');
        out += ".0";
    } else if (0 < kk && kk <= 16) {
        out.append(digits, 0, size_t(kk));
        out.push_back('.');
        out.append(digits, size_t(kk), std::string::npos);
    } else if (-5 < kk && kk <= 0) {
        out += "0.";
        out.append(size_t(-kk), '0');
        out += digits;
    } else if (length == 1) {
        out += digits;
        out.push_back('e');
        out += std::to_string(kk - 1);
    } else {
        out.push_back(digits[0]);
        out.push_back('.');
        out.append(digits, 1, std::string::npos);
        out.push_back('e');
        out += std::to_string(kk - 1);
    }
    return out;
}

// ------------------------------------------------------------------ writers
// bio::io::fasta::Writer::write(id, None, seq): ">id\nSEQ\n"
inline void write_fasta(std::string& out, const std::string& id, const uint8_t* seq, size_t n) {
    out.push_back('>');
    out += id;
    out.push_back('\n');
    out.append(reinterpret_cast<const char*>(seq), n);
    out.push_back('\n');
}

// csv crate, QuoteStyle::Necessary with delimiter '\t'
inline void tsv_field(std::string& out, const std::string& f) {
    bool need = false;
    for (char c : f)
        if (c == '\t' || c == '"' || c == '\n' || c == '\r') { need = true; break; }
    if (!need) { out += f; return; }
    out.push_back('"');
    for (char c : f) {
        if (c == '"') out.push_back('"');
        out.push_back(c);
    }
    out.push_back('"');
}

inline const char* idrecord_header() {
    return "id\ttranscript\tgene_id\tgene_name\tchrom\toffset\tframe\tfreq\tdepth\tnvar\tnsomatic\tnvariant_sites\t"
           "nsomvariant_sites\tstrand\tvariant_sites\tsomatic_positions\tsomatic_aa_change\tgermline_positions\t"
           "germline_aa_change\tnormal_sequence\tmutant_sequence\n";
}

// csv::Writer::serialize(IDRecord): header on first record only (reference: src/common.rs:350-373). The field-wise form lets the
// consumer write a row without building an IDRecord first (only windows that feed a splice-side merge need the record itself).
inline void write_tsv_fields(SomaticOutput& o, const std::string& id, const std::string& transcript, const std::string& gene_id,
                             const std::string& gene_name, const std::string& chrom, uint64_t offset, uint64_t frame, double freq, uint32_t depth,
                             uint32_t nvar, uint32_t nsomatic, uint32_t nvariant_sites, uint32_t nsomvariant_sites, const std::string& strand,
                             const std::string& variant_sites, const std::string& somatic_positions, const std::string& somatic_aa_change,
                             const std::string& germline_positions, const std::string& germline_aa_change, const std::string& normal_sequence,
                             const std::string& mutant_sequence) {
    if (!(o.streams & STREAM_TSV)) return;
    std::string& t = o.tsv;
    if (!o.tsv_header_written) { t += idrecord_header(); o.tsv_header_written = true; }
    auto S = [&](const std::string& f) { tsv_field(t, f); t.push_back('\t'); };
    auto U = [&](uint64_t v) { char b[24]; auto r = std::to_chars(b, b + sizeof b, v); t.append(b, r.ptr); t.push_back('\t'); };
    S(id); S(transcript); S(gene_id); S(gene_name); S(chrom);
    U(offset); U(frame);
    t += fmt_f64(freq); t.push_back('\t');
    U(depth); U(nvar); U(nsomatic); U(nvariant_sites); U(nsomvariant_sites);
    S(strand); S(variant_sites); S(somatic_positions); S(somatic_aa_change);
    S(germline_positions); S(germline_aa_change); S(normal_sequence);
    tsv_field(t, mutant_sequence);
    t.push_back('\n');
}
inline void write_tsv_record(SomaticOutput& o, const IDRecord& r) {
    write_tsv_fields(o, r.id, r.transcript, r.gene_id, r.gene_name, r.chrom, r.offset, r.frame, r.freq, r.depth, r.nvar, r.nsomatic, r.nvariant_sites,
                     r.nsomvariant_sites, r.strand, r.variant_sites, r.somatic_positions, r.somatic_aa_change, r.germline_positions,
                     r.germline_aa_change, r.normal_sequence, r.mutant_sequence);
}

// csv::Writer::serialize(normal_microphasing::IDRecord) (reference: src/normal_microphasing.rs:80-102)
inline void write_normal_tsv_fields(NormalOutput& o, const std::string& id, const std::string& transcript, const std::string& gene_id,
                                    const std::string& gene_name, const std::string& chrom, uint64_t offset, uint64_t frame, double freq, uint32_t depth,
                                    uint32_t nvar, uint32_t nsomatic, uint32_t nvariant_sites, uint32_t nsomvariant_sites, const std::string& strand,
                                    const std::string& variant_sites, const std::string& somatic_positions, const std::string& somatic_aa_change,
                                    const std::string& germline_positions, const std::string& germline_aa_change, const char* peptide, size_t peptide_len) {
    if (!(o.streams & STREAM_TSV)) return;
    std::string& t = o.tsv;
    if (!o.tsv_header_written) {
        t += "id\ttranscript\tgene_id\tgene_name\tchrom\toffset\tframe\tfreq\tdepth\tnvar\tnsomatic\tnvariant_sites\t"
             "nsomvariant_sites\tstrand\tvariant_sites\tsomatic_positions\tsomatic_aa_change\tgermline_positions\t"
             "germline_aa_change\tpeptide_sequence\n";
        o.tsv_header_written = true;
    }
    auto S = [&](const std::string& f) { tsv_field(t, f); t.push_back('\t'); };
    auto U = [&](uint64_t v) { char b[24]; auto r = std::to_chars(b, b + sizeof b, v); t.append(b, r.ptr); t.push_back('\t'); };
    S(id); S(transcript); S(gene_id); S(gene_name); S(chrom);
    U(offset); U(frame);
    t += fmt_f64(freq); t.push_back('\t');
    U(depth); U(nvar); U(nsomatic); U(nvariant_sites); U(nsomvariant_sites);
    S(strand); S(variant_sites); S(somatic_positions); S(somatic_aa_change);
    S(germline_positions); S(germline_aa_change);
    tsv_field(t, std::string(peptide, peptide_len));
    t.push_back('\n');
}
inline void write_normal_tsv_record(NormalOutput& o, const NormalRecord& r) {
    write_normal_tsv_fields(o, r.id, r.transcript, r.gene_id, r.gene_name, r.chrom, r.offset, r.frame, r.freq, r.depth, r.nvar, r.nsomatic, r.nvariant_sites,
                            r.nsomvariant_sites, r.strand, r.variant_sites, r.somatic_positions, r.somatic_aa_change, r.germline_positions,
                            r.germline_aa_change, r.peptide_sequence.data(), r.peptide_sequence.size());
}

}  // namespace mp
