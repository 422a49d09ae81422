// Row writers of the product's host legs (consumer, filter): the same bytes as the writers of util.hpp - which the CPU oracle keeps
// using, so that the two stay independent - without temporaries: a row is formatted in place at the end of its stream through a
// char cursor (numbers by to_chars, text fields from views).
#pragma once
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <string_view>
#include <vector>

#include "model.hpp"

namespace mp {

// An output stream under construction: a plain byte buffer (no zero-fill on growth, no terminator) that rows are formatted INTO - a
// writer asks for room(bound), fills it through a char cursor and advances; nothing is appended piece by piece.
struct TextBuf {
    char* p = nullptr;
    size_t n = 0, cap = 0;
    TextBuf() = default;
    TextBuf(const TextBuf&) = delete;
    TextBuf& operator=(const TextBuf&) = delete;
    TextBuf(TextBuf&& o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
    TextBuf& operator=(TextBuf&& o) noexcept {
        if (this != &o) { std::free(p); p = o.p; n = o.n; cap = o.cap; o.p = nullptr; o.n = o.cap = 0; }
        return *this;
    }
    ~TextBuf() { std::free(p); }
    const char* data() const { return p; }
    size_t size() const { return n; }
    size_t capacity() const { return cap; }
    bool empty() const { return n == 0; }
    void reserve(size_t c) {
        if (c <= cap) return;
        char* q = static_cast<char*>(std::realloc(p, c));
        if (!q) throw std::bad_alloc();
        p = q; cap = c;
    }
    char* room(size_t k) {   // at least k writable bytes at the end
        if (cap - n < k) reserve(std::max(n + k, cap + cap / 2 + 4096));
        return p + n;
    }
    void advance(size_t k) { n += k; }
    void append(const void* s, size_t k) { std::memcpy(room(k), s, k); n += k; }
    void append(std::string_view s) { append(s.data(), s.size()); }
};

// The output streams of one consumer thread (the product-side twins of SomaticOutput / NormalOutput, which the oracle writes)
struct SomaticText {
    TextBuf fasta, normal_fasta, tsv;
    uint32_t streams = STREAM_ALL;
    bool tsv_header_written = false;
    uint64_t n_windows = 0;
    std::vector<GeneEnds> gene_ends;
};
struct NormalText {
    TextBuf fasta, tsv;
    uint32_t streams = STREAM_ALL;
    bool tsv_header_written = false;
    uint64_t n_windows = 0;
    std::vector<GeneEnds> gene_ends;
};

// ---- cursor writers: each returns the cursor after what it wrote; the caller has made room for the bound it states
inline char* cur_u64(char* q, uint64_t v) { return std::to_chars(q, q + 20, v).ptr; }   // <= 20 bytes

// Rust ryu::Buffer::format_finite "pretty" layout (what csv + serde write for an f64 field; = fmt_f64 of util.hpp). <= 32 bytes
inline char* cur_f64(char* q, double v) {
    if (v != v) { std::memcpy(q, "NaN", 3); return q + 3; }
    if (v == 1.0 / 0.0) { std::memcpy(q, "inf", 3); return q + 3; }
    if (v == -1.0 / 0.0) { std::memcpy(q, "-inf", 4); return q + 4; }
    if (std::signbit(v)) { *q++ = '-'; v = -v; }
    if (v == 0.0) { std::memcpy(q, "0.0", 3); return q + 3; }
    char buf[48];
    const auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);  // shortest round trip: d[.ddd]e[+-]xx
    char digits[32];
    long nd = 0;
    const char* p = buf;
    for (; p < r.ptr && *p != 'e'; p++)
        if (*p != '.') digits[nd++] = *p;
    long exp10 = 0;
    if (p < r.ptr) {
        p++;
        bool neg = false;
        if (p < r.ptr && (*p == '-' || *p == '+')) { neg = *p == '-'; p++; }
        for (; p < r.ptr; p++) exp10 = exp10 * 10 + (*p - '0');
        if (neg) exp10 = -exp10;
    }
    while (nd > 1 && digits[nd - 1] == '0') nd--;
    const long kk = exp10 + 1;   // 10^(kk-1) <= v < 10^kk
    const long k = kk - nd;      // v = digits * 10^k
    if (0 <= k && kk <= 16) {
        std::memcpy(q, digits, size_t(nd)); q += nd;
        std::memset(q, '0', size_t(k)); q += k;
        *q++ = '.'; *q++ = '0';
    } else if (0 < kk && kk <= 16) {
        std::memcpy(q, digits, size_t(kk)); q += kk;
        *q++ = '.';
        std::memcpy(q, digits + kk, size_t(nd - kk)); q += nd - kk;
    } else if (-5 < kk && kk <= 0) {
        *q++ = '0'; *q++ = '.';
        std::memset(q, '0', size_t(-kk)); q += -kk;
        std::memcpy(q, digits, size_t(nd)); q += nd;
    } else {
        *q++ = digits[0];
        if (nd > 1) { *q++ = '.'; std::memcpy(q, digits + 1, size_t(nd - 1)); q += nd - 1; }
        *q++ = 'e';
        const long e = kk - 1;
        if (e < 0) { *q++ = '-'; q = cur_u64(q, uint64_t(-e)); } else q = cur_u64(q, uint64_t(e));
    }
    return q;
}

// csv crate, QuoteStyle::Necessary with delimiter '\t'. <= 2 * size + 2 bytes
inline char* cur_field(char* q, std::string_view f) {
    unsigned need = 0;
    for (size_t i = 0; i < f.size(); i++) {
        const char c = f[i];
        q[i] = c;
        need |= unsigned(c == '\t') | unsigned(c == '"') | unsigned(c == '\n') | unsigned(c == '\r');
    }
    if (!need) return q + f.size();
    *q++ = '"';
    for (char c : f) {
        if (c == '"') *q++ = '"';
        *q++ = c;
    }
    *q++ = '"';
    return q;
}
inline size_t field_bound(std::string_view f) { return 2 * f.size() + 3; }   // quoted worst case + the separator

// bio::io::fasta::Writer::write(id, None, seq): ">id\nSEQ\n"
inline void put_fasta(TextBuf& out, std::string_view id, const uint8_t* seq, size_t n) {
    char* const b = out.room(id.size() + n + 3);
    char* q = b;
    *q++ = '>';
    std::memcpy(q, id.data(), id.size()); q += id.size();
    *q++ = '\n';
    std::memcpy(q, seq, n); q += n;
    *q++ = '\n';
    out.advance(size_t(q - b));
}

// csv::Writer::serialize(IDRecord): header on first record only (reference: src/common.rs:350-373)
inline void put_tsv_row(SomaticText& o, std::string_view id, std::string_view transcript, std::string_view gene_id, std::string_view gene_name,
                        std::string_view chrom, uint64_t offset, uint64_t frame, double freq, uint32_t depth, uint32_t nvar, uint32_t nsomatic,
                        uint32_t nvariant_sites, uint32_t nsomvariant_sites, std::string_view strand, std::string_view variant_sites,
                        std::string_view somatic_positions, std::string_view somatic_aa_change, std::string_view germline_positions,
                        std::string_view germline_aa_change, std::string_view normal_sequence, std::string_view mutant_sequence) {
    if (!(o.streams & STREAM_TSV)) return;
    TextBuf& t = o.tsv;
    if (!o.tsv_header_written) {
        t.append("id\ttranscript\tgene_id\tgene_name\tchrom\toffset\tframe\tfreq\tdepth\tnvar\tnsomatic\tnvariant_sites\t"
                 "nsomvariant_sites\tstrand\tvariant_sites\tsomatic_positions\tsomatic_aa_change\tgermline_positions\t"
                 "germline_aa_change\tnormal_sequence\tmutant_sequence\n");
        o.tsv_header_written = true;
    }
    const size_t bound = field_bound(id) + field_bound(transcript) + field_bound(gene_id) + field_bound(gene_name) + field_bound(chrom) +
                         field_bound(strand) + field_bound(variant_sites) + field_bound(somatic_positions) + field_bound(somatic_aa_change) +
                         field_bound(germline_positions) + field_bound(germline_aa_change) + field_bound(normal_sequence) +
                         field_bound(mutant_sequence) + 7 * 21 + 33 + 1;
    char* const b = t.room(bound);
    char* q = b;
    auto S = [&](std::string_view f) { q = cur_field(q, f); *q++ = '\t'; };
    auto U = [&](uint64_t v) { q = cur_u64(q, v); *q++ = '\t'; };
    S(id); S(transcript); S(gene_id); S(gene_name); S(chrom);
    U(offset); U(frame);
    q = cur_f64(q, freq); *q++ = '\t';
    U(depth); U(nvar); U(nsomatic); U(nvariant_sites); U(nsomvariant_sites);
    S(strand); S(variant_sites); S(somatic_positions); S(somatic_aa_change);
    S(germline_positions); S(germline_aa_change); S(normal_sequence);
    q = cur_field(q, mutant_sequence);
    *q++ = '\n';
    t.advance(size_t(q - b));
}
inline void put_tsv_row(SomaticText& o, const IDRecord& r) {
    put_tsv_row(o, r.id, r.transcript, r.gene_id, r.gene_name, r.chrom, r.offset, r.frame, r.freq, r.depth, r.nvar, r.nsomatic, r.nvariant_sites,
                r.nsomvariant_sites, r.strand, r.variant_sites, r.somatic_positions, r.somatic_aa_change, r.germline_positions, r.germline_aa_change,
                r.normal_sequence, r.mutant_sequence);
}

// csv::Writer::serialize(normal_microphasing::IDRecord) (reference: src/normal_microphasing.rs:80-102)
inline void put_normal_tsv_row(NormalText& o, std::string_view id, std::string_view transcript, std::string_view gene_id, std::string_view gene_name,
                               std::string_view chrom, uint64_t offset, uint64_t frame, double freq, uint32_t depth, uint32_t nvar, uint32_t nsomatic,
                               uint32_t nvariant_sites, uint32_t nsomvariant_sites, std::string_view strand, std::string_view variant_sites,
                               std::string_view somatic_positions, std::string_view somatic_aa_change, std::string_view germline_positions,
                               std::string_view germline_aa_change, std::string_view peptide_sequence) {
    if (!(o.streams & STREAM_TSV)) return;
    TextBuf& t = o.tsv;
    if (!o.tsv_header_written) {
        t.append("id\ttranscript\tgene_id\tgene_name\tchrom\toffset\tframe\tfreq\tdepth\tnvar\tnsomatic\tnvariant_sites\t"
                 "nsomvariant_sites\tstrand\tvariant_sites\tsomatic_positions\tsomatic_aa_change\tgermline_positions\t"
                 "germline_aa_change\tpeptide_sequence\n");
        o.tsv_header_written = true;
    }
    const size_t bound = field_bound(id) + field_bound(transcript) + field_bound(gene_id) + field_bound(gene_name) + field_bound(chrom) +
                         field_bound(strand) + field_bound(variant_sites) + field_bound(somatic_positions) + field_bound(somatic_aa_change) +
                         field_bound(germline_positions) + field_bound(germline_aa_change) + field_bound(peptide_sequence) + 7 * 21 + 33 + 1;
    char* const b = t.room(bound);
    char* q = b;
    auto S = [&](std::string_view f) { q = cur_field(q, f); *q++ = '\t'; };
    auto U = [&](uint64_t v) { q = cur_u64(q, v); *q++ = '\t'; };
    S(id); S(transcript); S(gene_id); S(gene_name); S(chrom);
    U(offset); U(frame);
    q = cur_f64(q, freq); *q++ = '\t';
    U(depth); U(nvar); U(nsomatic); U(nvariant_sites); U(nsomvariant_sites);
    S(strand); S(variant_sites); S(somatic_positions); S(somatic_aa_change);
    S(germline_positions); S(germline_aa_change);
    q = cur_field(q, peptide_sequence);
    *q++ = '\n';
    t.advance(size_t(q - b));
}
inline void put_normal_tsv_row(NormalText& o, const NormalRecord& r) {
    put_normal_tsv_row(o, r.id, r.transcript, r.gene_id, r.gene_name, r.chrom, r.offset, r.frame, r.freq, r.depth, r.nvar, r.nsomatic, r.nvariant_sites,
                       r.nsomvariant_sites, r.strand, r.variant_sites, r.somatic_positions, r.somatic_aa_change, r.germline_positions,
                       r.germline_aa_change, r.peptide_sequence);
}

// list fields that are built up piece by piece (std::string scratch of the consumer)
inline void append_u64(std::string& out, uint64_t v) {
    char buf[24];
    out.append(buf, size_t(cur_u64(buf, v) - buf));
}
inline void append_f64(std::string& out, double v) {
    char buf[40];
    out.append(buf, size_t(cur_f64(buf, v) - buf));
}

}  // namespace mp
