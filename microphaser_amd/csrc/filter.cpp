#include "filter.hpp"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <charconv>
#include <deque>
#include <exception>
#include <limits>
#include <map>
#include <set>
#include <string_view>
#include <thread>
#include <tuple>
#include <vector>

#include "batch.hpp"
#include "kernels_filter.hpp"
#include "kernels_pep.hpp"
#include "pep.hpp"
#include "rowfmt.hpp"

namespace mp {

[[noreturn]] void throw_hip(hipError_t e, const char* file, int line);
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw_hip(e_, __FILE__, __LINE__); } while (0)

namespace {

// One row of info.tsv (IDRecord, src/common.rs:350-373): text fields as views into the TSV buffer (or, for a quoted field, into the
// arena that holds its unescaped form), numbers parsed.
// a string_view without a constructor, so that a Row can be created uninitialised (the parsing threads fill 8.8 M of them in place)
struct SV {
    const char* p;
    size_t n;
    SV& operator=(std::string_view v) { p = v.data(); n = v.size(); return *this; }
    operator std::string_view() const { return std::string_view(p, n); }
    bool empty() const { return n == 0; }
    size_t size() const { return n; }
    const char* data() const { return p; }
    char back() const { return p[n - 1]; }
    size_t find(char c) const { return std::string_view(p, n).find(c); }
};
inline bool operator==(const SV& a, const SV& b) { return std::string_view(a) == std::string_view(b); }
inline bool operator!=(const SV& a, const SV& b) { return !(a == b); }
inline bool operator==(const SV& a, const char* b) { return std::string_view(a) == std::string_view(b); }

struct Row {
    SV id, transcript, gene_id, gene_name, chrom, strand, variant_sites, somatic_positions, somatic_aa_change, germline_positions,
        germline_aa_change, normal_sequence, mutant_sequence;
    uint64_t offset, frame;
    double freq;
    uint32_t depth, nvar, nsomatic, nvariant_sites, nsomvariant_sites;
};
using RowVec = PodVec<Row>;   // rows are sized once and filled in place by the parsing threads (no zero fill: 250 bytes x 8.8 M rows)

uint64_t field_u64(std::string_view s, const char* name) {
    uint64_t v = 0;
    const auto r = std::from_chars(s.data(), s.data() + s.size(), v);
    if (s.empty() || r.ptr != s.data() + s.size() || r.ec == std::errc::invalid_argument)
        throw Error(std::string("CSV deserialize error: field ") + name + ": invalid digit found in string");
    if (r.ec == std::errc::result_out_of_range)
        throw Error(std::string("CSV deserialize error: field ") + name + ": number too large to fit in target type");
    return v;
}
uint32_t field_u32(std::string_view s, const char* name) {
    const uint64_t v = field_u64(s, name);
    if (v > 0xFFFFFFFFull) throw Error(std::string("CSV deserialize error: field ") + name + ": number too large to fit in target type");
    return uint32_t(v);
}
double field_f64(std::string_view s, const char* name) {
    char buf[64];
    std::string big;
    const char* z = buf;
    if (s.size() < sizeof buf) { std::memcpy(buf, s.data(), s.size()); buf[s.size()] = 0; }
    else { big.assign(s); z = big.c_str(); }
    char* e = nullptr;
    const double v = std::strtod(z, &e);
    if (s.empty() || *e) throw Error(std::string("CSV deserialize error: field ") + name + ": invalid float literal");
    return v;
}
void row_from_fields(const std::string_view* f, size_t n, Row& r) {   // serde positional deserialize of IDRecord
    if (n != 21) throw Error("CSV deserialize error: found record with " + std::to_string(n) + " fields, but expected 21");
    r.id = f[0]; r.transcript = f[1]; r.gene_id = f[2]; r.gene_name = f[3]; r.chrom = f[4];
    r.offset = field_u64(f[5], "offset");
    r.frame = field_u64(f[6], "frame");
    r.freq = field_f64(f[7], "freq");
    r.depth = field_u32(f[8], "depth");
    r.nvar = field_u32(f[9], "nvar");
    r.nsomatic = field_u32(f[10], "nsomatic");
    r.nvariant_sites = field_u32(f[11], "nvariant_sites");
    r.nsomvariant_sites = field_u32(f[12], "nsomvariant_sites");
    r.strand = f[13]; r.variant_sites = f[14]; r.somatic_positions = f[15]; r.somatic_aa_change = f[16];
    r.germline_positions = f[17]; r.germline_aa_change = f[18]; r.normal_sequence = f[19]; r.mutant_sequence = f[20];
}

// csv::ReaderBuilder::new().delimiter(b'\t') with default quoting over [p, end): every record after the first `skip` ones becomes a Row.
// Unquoted fields (all of them in what `somatic` writes, unless a value holds a tab, a quote or a line break) are byte ranges of the
// buffer; a field that starts with a quote is unescaped into `arena` ("" is a quote; the closing quote ends the quoting, further bytes
// up to the delimiter are appended).
void parse_rows(const char* p, const char* const end, size_t skip, RowVec& rows, std::deque<std::string>& arena) {
    std::vector<std::string_view> rec;
    rec.reserve(32);
    bool started = false;        // the current record holds at least one field or delimiter
    std::string_view field;
    auto end_record = [&]() {
        if (started || !field.empty()) {
            rec.push_back(field);
            if (skip) skip--;
            else { rows.emplace_back(); row_from_fields(rec.data(), rec.size(), rows.back()); }
            rec.clear();
        }
        field = std::string_view();
        started = false;
    };
    while (p < end) {
        if (*p == '"') {
            started = true;
            p++;
            std::string text;
            bool quoted = true;
            while (p < end) {
                const char c = *p;
                if (quoted) {
                    if (c != '"') { text.push_back(c); p++; }
                    else if (p + 1 < end && p[1] == '"') { text.push_back('"'); p += 2; }
                    else { quoted = false; p++; }
                } else if (c == '\t' || c == '\n' || c == '\r') break;
                else { text.push_back(c); p++; }
            }
            arena.push_back(std::move(text));
            field = arena.back();
        } else {
            const char* q = p;
            while (q < end && *q != '\t' && *q != '\n' && *q != '\r') q++;
            if (q > p) { field = std::string_view(p, size_t(q - p)); started = true; }
            p = q;
        }
        if (p >= end) break;
        if (*p == '\t') { rec.push_back(field); field = std::string_view(); started = true; p++; }
        else if (*p == '\n') { end_record(); p++; }
        else { if (p + 1 < end && p[1] == '\n') p++; end_record(); p++; }   // '\r' or "\r\n"
    }
    end_record();
}

// The same for a text without a quote and without a carriage return (what `somatic` writes): lines end at '\n', fields at '\t' - both
// found with memchr instead of a byte loop.
// A record of such a text = a non-empty line; counted first so that every part's rows get their place in the one row array.
size_t count_plain_records(const char* p, const char* const end) {
    size_t n = 0;
    while (p < end) {
        const char* nl = static_cast<const char*>(std::memchr(p, '\n', size_t(end - p)));
        const char* const le = nl ? nl : end;
        n += le > p;
        p = nl ? nl + 1 : end;
    }
    return n;
}
void parse_rows_plain(const char* p, const char* const end, size_t skip, Row* out, size_t n_out) {
    std::string_view rec[32];
    size_t k = 0;
    while (p < end) {
        const char* nl = static_cast<const char*>(std::memchr(p, '\n', size_t(end - p)));
        const char* const le = nl ? nl : end;
        if (le > p) {   // (an empty line is no record)
            size_t nf = 0;
            std::vector<std::string_view> many;   // a record with more fields than the fixed array holds (reported as malformed anyway)
            for (const char* q = p;;) {
                const char* tab = static_cast<const char*>(std::memchr(q, '\t', size_t(le - q)));
                const std::string_view f(q, size_t((tab ? tab : le) - q));
                if (nf < 32) rec[nf] = f; else { if (many.empty()) many.assign(rec, rec + 32); many.push_back(f); }
                nf++;
                if (!tab) break;
                q = tab + 1;
            }
            if (skip) skip--;
            else {
                if (k >= n_out) throw Error("internal error: TSV record count changed between the two passes");
                row_from_fields(many.empty() ? rec : many.data(), nf, out[k++]);
            }
        }
        p = nl ? nl + 1 : end;
    }
    if (k != n_out) throw Error("internal error: TSV record count changed between the two passes");
}

// Threads of the host legs, in order: the error of the earliest part is the one reported (what a sequential pass would have hit first).
template <class F> void run_parts(size_t n, F f) {
    std::vector<std::exception_ptr> errors(n);
    std::vector<std::thread> th;
    auto guarded = [&](size_t t) { try { f(t); } catch (...) { errors[t] = std::current_exception(); } };
    for (size_t t = 1; t < n; t++) th.emplace_back(guarded, t);
    guarded(0);
    for (auto& x : th) x.join();
    for (auto& e : errors) if (e) std::rethrow_exception(e);
}

// All rows of the TSV (first record = header). Without a quote anywhere in the text no field spans lines, so the text is cut at line
// breaks and the parts are parsed by all host threads; rows stay in file order.
void parse_tsv(std::string_view text, RowVec& rows, std::deque<std::string>& arena) {
    const char* const b = text.data();
    const char* const e = b + text.size();
    size_t parts = std::min<size_t>(host_threads(), text.size() / (4u << 20) + 1);
    // the threaded form needs records that are lines: no quote (a quoted field may hold a line break), no carriage return; and part 0
    // must hold the header (the first record that holds anything): a text that opens with a blank line is parsed in one piece
    if (parts > 1 && (*b == '\n' || std::memchr(b, '"', text.size()) != nullptr || std::memchr(b, '\r', text.size()) != nullptr)) parts = 1;
    if (parts <= 1) {
        size_t nl = 0;
        for (const char* q = b; (q = static_cast<const char*>(std::memchr(q, '\n', size_t(e - q)))) != nullptr; q++) nl++;
        rows.reserve(nl + 1);
        parse_rows(b, e, 1, rows, arena);
        return;
    }
    std::vector<const char*> cut(parts + 1, e);
    cut[0] = b;
    for (size_t t = 1; t < parts; t++) {
        const char* q = b + text.size() / parts * t;
        if (q < cut[t - 1]) q = cut[t - 1];
        const char* nl = static_cast<const char*>(std::memchr(q, '\n', size_t(e - q)));
        cut[t] = nl ? nl + 1 : e;
    }
    // pass 1: the parts' record counts -> every part's slice of the row array; pass 2: the rows, parsed in place
    std::vector<size_t> at(parts + 1, 0);
    run_parts(parts, [&](size_t t) { at[t + 1] = count_plain_records(cut[t], cut[t + 1]); });
    if (at[1] == 0) throw Error("internal error: TSV header not in the first part");
    at[1] -= 1;   // the header
    for (size_t t = 0; t < parts; t++) at[t + 1] += at[t];
    rows.resize(at[parts]);
    advise_huge(rows.data(), rows.size() * sizeof(Row));
    run_parts(parts, [&](size_t t) { parse_rows_plain(cut[t], cut[t + 1], t == 0 ? 1 : 0, rows.data() + at[t], at[t + 1] - at[t]); });
}

// FilteredRecord::FIELD_NAMES_AS_ARRAY (src/peptides.rs:21-47)
const char* const FILTERED_HEADER =
    "id\ttranscript\tgene_id\tgene_name\tchrom\toffset\tframe\tfreq\tcredible_interval\tdepth\tnvar\tnsomatic\tnvariant_sites\tnsomvariant_sites\t"
    "strand\tvariant_sites\tsomatic_positions\tsomatic_aa_change\tgermline_positions\tgermline_aa_change\tnormal_sequence\tmutant_sequence\t"
    "normal_peptide\ttumor_peptide\n";

// One filtered row (FilteredRecord, src/peptides.rs:21-47), formatted in place at the end of its stream (rowfmt.hpp cursor writers)
void write_filtered_record(TextBuf& t, const Row& r, double freq, std::string_view id, std::string_view ci, std::string_view normal_pep,
                           std::string_view tumor_pep) {
    const std::string_view text[] = {id, r.transcript, r.gene_id, r.gene_name, r.chrom, ci, r.strand, r.variant_sites, r.somatic_positions,
                                     r.somatic_aa_change, r.germline_positions, r.germline_aa_change, r.normal_sequence, r.mutant_sequence,
                                     normal_pep, tumor_pep};
    size_t bound = 7 * 21 + 33 + 1;
    for (std::string_view f : text) bound += field_bound(f);
    char* const b = t.room(bound);
    char* q = b;
    auto S = [&](std::string_view f) { q = cur_field(q, f); *q++ = '\t'; };
    auto U = [&](uint64_t v) { q = cur_u64(q, v); *q++ = '\t'; };
    S(id); S(r.transcript); S(r.gene_id); S(r.gene_name); S(r.chrom); U(r.offset); U(r.frame);
    q = cur_f64(q, freq); *q++ = '\t';
    S(ci); U(r.depth); U(r.nvar); U(r.nsomatic); U(r.nvariant_sites); U(r.nsomvariant_sites);
    S(r.strand); S(r.variant_sites); S(r.somatic_positions); S(r.somatic_aa_change); S(r.germline_positions); S(r.germline_aa_change);
    S(r.normal_sequence); S(r.mutant_sequence); S(normal_pep);
    q = cur_field(q, tumor_pep);
    *q++ = '\n';
    t.advance(size_t(q - b));
}

// The peptides already seen in the current (transcript, somatic_positions, germline_positions) run (seen_peptides, :262-400): a handful
// per run, so a flat list; a run that grows past it gets an ordered set.
struct SeenPeptides {
    std::vector<std::string_view> few;
    std::set<std::string_view> many;
    bool contains(std::string_view p) const {
        if (!many.empty()) return many.count(p) != 0;
        for (std::string_view q : few) if (q == p) return true;
        return false;
    }
    void insert(std::string_view p) {
        if (!many.empty()) { many.insert(p); return; }
        few.push_back(p);
        if (few.size() > 48) { many.insert(few.begin(), few.end()); few.clear(); }
    }
    void clear() { few.clear(); many.clear(); }
};

template <class V>
typename V::value_type* to_device(const V& v, hipStream_t stream, std::vector<void*>& owned, size_t pad = 0) {
    void* p = nullptr;
    using T = typename V::value_type;
    HIP_OK(hipMalloc(&p, v.size() * sizeof(T) + pad + 16));
    owned.push_back(p);
    if (!v.empty()) HIP_OK(hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, stream));
    return static_cast<T*>(p);
}

}  // namespace

void filter_device(int device, std::string_view reference_binary, const std::vector<uint64_t>* reference_keys, std::string_view tsv_text,
                   uint32_t L, FilterResult& out) {
    if (L == 0 || L > 12) throw Error("peptide length must be 1..12 for the device peptidome (5-bit residue keys in a u64)");
    out = FilterResult();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        throw Error("no HIP device available: `filter` translates and scores on the GPU, there is no CPU fallback");
    HIP_OK(hipSetDevice(device));
    const bool dbg = std::getenv("MP_DEBUG") != nullptr;   // wall time of the phases on stderr
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        const auto now = std::chrono::steady_clock::now();
        if (dbg) std::fprintf(stderr, "[mp]   filter: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    const size_t threads = host_threads();

    // ---- reference peptidome: bincode v1 HashSet<Vec<u8>> (deserialize_from(...).unwrap(), :242) -> keys of the length-L members;
    //      or the sorted distinct keys themselves (a peptidome that never left the library)
    std::vector<uint64_t> ref_keys;
    if (!reference_keys) {
        size_t p = 0;
        auto u64 = [&]() {
            if (p + 8 > reference_binary.size())
                throw Error("reference would panic: called `Result::unwrap()` on an `Err` value (bincode: unexpected end of file)");
            uint64_t v = 0;
            for (int i = 0; i < 8; i++) v |= uint64_t(uint8_t(reference_binary[p + i])) << (8 * i);
            p += 8;
            return v;
        };
        const uint64_t n = u64();
        for (uint64_t i = 0; i < n; i++) {
            const uint64_t l = u64();
            if (l > reference_binary.size() - p)
                throw Error("reference would panic: called `Result::unwrap()` on an `Err` value (bincode: unexpected end of file)");
            if (l == L) {  // only a member of the same length can equal a tumor peptide
                bool letters = true;
                for (uint64_t k = 0; k < l; k++) { const char c = reference_binary[p + k]; letters &= c >= 'A' && c <= 'Z'; }
                if (letters) {
                    uint64_t key = 0;
                    for (uint64_t k = 0; k < l; k++) key = (key << 5) | uint64_t((reference_binary[p + k] - 'A') & 31);   // = peptide_to_key
                    ref_keys.push_back(key);
                }
            }
            p += l;
        }
    }
    lap("reference set decoded");

    // ---- rows and their two nucleotide windows
    RowVec rows;
    std::deque<std::string> arena;
    parse_tsv(tsv_text, rows, arena);
    lap("tsv parsed");
    out.n_rows = rows.size();
    const size_t n_seq = rows.size() * 2;   // 2r = mutant, 2r + 1 = normal
    PodVec<uint8_t> nt, rev(n_seq);
    PodVec<uint64_t> nt_off(n_seq), aa_off(n_seq + 1);
    PodVec<uint32_t> nt_len(n_seq);
    aa_off[0] = 0;
    {
        uint64_t at = 0;
        for (size_t r = 0; r < rows.size(); r++) {
            const uint8_t rv = (!rows[r].id.empty() && rows[r].id.back() == 'F') ? 0 : 1;  // :291-294
            const std::string_view seqs[2] = {rows[r].mutant_sequence, rows[r].normal_sequence};
            for (int k = 0; k < 2; k++) {
                const size_t s = 2 * r + k;
                if (seqs[k].size() > 0xFFFFFFFFull) throw Error("sequence too long");
                nt_off[s] = at;
                nt_len[s] = uint32_t(seqs[k].size());
                rev[s] = rv;
                at += seqs[k].size();
                aa_off[s + 1] = aa_off[s] + (seqs[k].size() > 2 ? (seqs[k].size() - 2 + 2) / 3 : 0);
            }
        }
        nt.resize(at);
        advise_huge(nt.data(), nt.size());
        const size_t parts = std::min<size_t>(threads, rows.size() / 65536 + 1);
        run_parts(parts, [&](size_t t) {
            for (size_t r = rows.size() * t / parts, e = rows.size() * (t + 1) / parts; r < e; r++) {
                if (nt_len[2 * r]) std::memcpy(nt.data() + nt_off[2 * r], rows[r].mutant_sequence.data(), nt_len[2 * r]);
                if (nt_len[2 * r + 1]) std::memcpy(nt.data() + nt_off[2 * r + 1], rows[r].normal_sequence.data(), nt_len[2 * r + 1]);
            }
        });
    }
    const uint64_t n_aa = aa_off[n_seq];
    lap("windows laid out");

    hipStream_t stream;
    HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    hipEvent_t e0, e1, e2, e3;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1)); HIP_OK(hipEventCreate(&e2)); HIP_OK(hipEventCreate(&e3));
    std::vector<void*> owned;
    auto cleanup = [&] {
        for (void* p : owned) hipFree(p);
        owned.clear();
        hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(e2); hipEventDestroy(e3);
        hipStreamDestroy(stream);
    };
    try {
        // sorted distinct reference keys (device radix sort + unique, unless they came sorted)
        uint64_t n_ref = 0;
        uint64_t* d_ref = nullptr;
        if (reference_keys) {
            n_ref = reference_keys->size();
            if (n_ref) d_ref = to_device(*reference_keys, stream, owned);
        } else if (!ref_keys.empty()) {
            uint64_t* d_in = to_device(ref_keys, stream, owned);
            void *d_tmp = nullptr, *d_out = nullptr;
            HIP_OK(hipMalloc(&d_tmp, ref_keys.size() * 8)); owned.push_back(d_tmp);
            HIP_OK(hipMalloc(&d_out, ref_keys.size() * 8)); owned.push_back(d_out);
            n_ref = device_sort_unique(d_in, static_cast<uint64_t*>(d_tmp), static_cast<uint64_t*>(d_out), ref_keys.size(), 5 * L, stream);
            d_ref = static_cast<uint64_t*>(d_out);
        }
        // ---- K5
        uint8_t* d_nt = to_device(nt, stream, owned, 64);
        uint64_t* d_nt_off = to_device(nt_off, stream, owned);
        uint32_t* d_nt_len = to_device(nt_len, stream, owned);
        uint8_t* d_rev = to_device(rev, stream, owned);
        uint64_t* d_aa_off = to_device(aa_off, stream, owned);
        void *d_aa = nullptr, *d_flags = nullptr, *d_err = nullptr;
        HIP_OK(hipMalloc(&d_aa, n_aa + 16)); owned.push_back(d_aa);
        HIP_OK(hipMalloc(&d_flags, n_aa + 16)); owned.push_back(d_flags);
        HIP_OK(hipMalloc(&d_err, 4)); owned.push_back(d_err);
        HIP_OK(hipMemsetAsync(d_err, 0, 4, stream));
        HIP_OK(hipMemsetAsync(d_flags, 0, n_aa + 16, stream));
        HIP_OK(hipEventRecord(e0, stream));
        device_translate_records(d_nt, d_nt_off, d_nt_len, d_rev, d_aa_off, n_seq, L, d_ref, n_ref, static_cast<uint8_t*>(d_aa),
                                 static_cast<uint8_t*>(d_flags), static_cast<uint32_t*>(d_err), stream);
        HIP_OK(hipEventRecord(e1, stream));
        PodVec<uint8_t> aa(n_aa), flags(n_aa);
        uint32_t err = 0;
        if (n_aa) {
            HIP_OK(hipMemcpyAsync(aa.data(), d_aa, n_aa, hipMemcpyDeviceToHost, stream));
            HIP_OK(hipMemcpyAsync(flags.data(), d_flags, n_aa, hipMemcpyDeviceToHost, stream));
        }
        HIP_OK(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        HIP_OK(hipEventElapsedTime(&out.translate_ms, e0, e1));
        lap("reference sort + K5 + copies");

        // ---- the row stream (:262-563): which (row, i) peptides survive, and the groups they are scored in
        struct Entry { uint32_t row; uint32_t i1; uint64_t tumor_at; const uint8_t* normal_pep; uint32_t normal_len; uint64_t group; };
        std::vector<Entry> entries;                 // in output order
        std::vector<uint64_t> grp_off{0};
        std::vector<uint8_t> grp_final;
        std::vector<double> g_alt;
        std::vector<uint32_t> g_depth;
        entries.reserve(rows.size());
        // records / frequencies / depth of the current region, by (frame, somatic_positions, germline_positions): the reference's BTreeMaps.
        // A region holds a few keys; the slots keep their buffers from region to region and are put in key order when the region is flushed
        struct Key {
            uint64_t frame; std::string_view som, germ;
            bool operator==(const Key& o) const { return frame == o.frame && som == o.som && germ == o.germ; }
            bool operator<(const Key& o) const { return std::tie(frame, som, germ) < std::tie(o.frame, o.som, o.germ); }
        };
        struct Pending { Key key; std::vector<double> alt; std::vector<uint32_t> depth; std::vector<Entry> recs; };
        std::vector<Pending> slots;
        size_t n_used = 0;
        std::map<Key, size_t> slot_index;           // only for a region with more keys than a scan should look through
        constexpr size_t SCAN_LIMIT = 12;
        auto find_slot = [&](const Key& k) -> Pending* {
            if (n_used > SCAN_LIMIT) {
                auto it = slot_index.find(k);
                return it == slot_index.end() ? nullptr : &slots[it->second];
            }
            for (size_t i = 0; i < n_used; i++) if (slots[i].key == k) return &slots[i];
            return nullptr;
        };
        auto add_slot = [&](const Key& k) -> Pending& {
            if (n_used == slots.size()) slots.emplace_back();
            Pending& p = slots[n_used];
            p.key = k; p.alt.clear(); p.depth.clear(); p.recs.clear();
            n_used++;
            if (n_used == SCAN_LIMIT + 1) for (size_t i = 0; i + 1 < n_used; i++) slot_index[slots[i].key] = i;
            if (n_used > SCAN_LIMIT) slot_index[k] = n_used - 1;
            return p;
        };
        std::vector<size_t> order;
        auto flush = [&](bool final_pass) {
            order.resize(n_used);
            for (size_t i = 0; i < n_used; i++) order[i] = i;
            if (n_used > 1) std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return slots[a].key < slots[b].key; });
            for (size_t i : order) {
                Pending& p = slots[i];
                const uint64_t g = grp_final.size();
                grp_final.push_back(final_pass ? 1 : 0);
                g_alt.insert(g_alt.end(), p.alt.begin(), p.alt.end());
                g_depth.insert(g_depth.end(), p.depth.begin(), p.depth.end());
                grp_off.push_back(g_alt.size());
                for (Entry& e : p.recs) { e.group = g; entries.push_back(e); }
            }
            n_used = 0;
            slot_index.clear();
        };
        // `current` (transcript, somatic_positions, germline_positions) and `region_sites` (transcript, variant_sites) of the reference are
        // kept as the row that last set them; both start as tuples of empty strings
        static const Row kEmptyRow{};
        const Row* current = &kEmptyRow;
        const Row* region = &kEmptyRow;
        SeenPeptides seen_peptides;
        // stop_gained: (transcript, frame) -> offset of the stop-gain row. Rows come in runs of one (transcript, frame), so the entry
        // of the previous row is remembered.
        std::map<std::pair<std::string_view, uint64_t>, size_t> stop_gained;
        const Row* sg_row = nullptr;                // the row the remembered look-up was made for
        size_t* sg_at = nullptr;                    // its entry (nullptr: none)
        for (size_t r = 0; r < rows.size(); r++) {
            const Row& row = rows[r];
            if (row.mutant_sequence.size() < 2 || (!row.normal_sequence.empty() && row.normal_sequence.size() < 2))  // `r.len() - 2` (:139)
                throw Error("reference would panic: attempt to subtract with overflow (to_protein)");
            size_t som_pos = 0;
            if (!row.somatic_positions.empty() && row.somatic_positions.find('|') == std::string_view::npos) {
                uint64_t v = 0;
                const auto pr = std::from_chars(row.somatic_positions.data(), row.somatic_positions.data() + row.somatic_positions.size(), v);
                if (pr.ec != std::errc() || pr.ptr != row.somatic_positions.data() + row.somatic_positions.size())
                    throw Error("reference would panic: called `Result::unwrap()` on an `Err` value (somatic_positions)");
                som_pos = size_t(v);
            }
            const size_t offset = size_t(row.offset);
            const uint8_t* tp = aa.data() + aa_off[2 * r];
            const size_t tlen = size_t(aa_off[2 * r + 1] - aa_off[2 * r]);
            const uint8_t* np = aa.data() + aa_off[2 * r + 1];
            const size_t nlen = size_t(aa_off[2 * r + 2] - aa_off[2 * r + 1]);
            if (err & 1)   // some codon held a base other than A, C, G, T: find the first row it is in
                for (size_t k = 0; k < tlen + nlen; k++)
                    if ((k < tlen ? tp[k] : np[k - tlen]) == '?')
                        throw Error("reference would panic: called `Result::unwrap()` on an `Err` value (codon with a base other than A, C, G, T)");
            const bool fwd = row.strand == "Forward", rvs = row.strand == "Reverse";
            if (!(sg_row && sg_row->frame == row.frame && sg_row->transcript == row.transcript)) {
                auto sg = stop_gained.find({row.transcript, row.frame});
                sg_at = sg == stop_gained.end() ? nullptr : &sg->second;
            }
            sg_row = &row;
            if (sg_at) {  // :304-317
                const bool downstream = fwd ? offset > *sg_at : rvs ? offset < *sg_at : false;
                if (downstream) continue;
            }
            bool has_x = false;
            for (size_t k = 0; k < tlen; k++) has_x |= tp[k] == 'X';
            if (has_x && (std::fabs(row.freq - 1.0) < std::numeric_limits<double>::epsilon() || row.frame > 0)) {
                if (sg_at) *sg_at = offset;
                else sg_at = &(stop_gained[{row.transcript, row.frame}] = offset);
            }
            const uint8_t* tf = flags.data() + aa_off[2 * r];
            size_t i = 0;
            while (i + L <= tlen) {
                if (tf[i] & 1) break;  // the peptide contains a stop (:330-332)
                const bool normal_full = nlen >= i + L;
                const size_t nl = normal_full ? L : nlen;
                const uint8_t* npep = normal_full ? np + i : np;
                if (nl == 0 && som_pos > 0) {  // :343-360
                    if (fwd) {
                        if ((i + L) * 3 + offset <= som_pos) { i += 1; continue; }
                    } else if (rvs) {
                        if ((tlen - (i + L)) * 3 + offset > som_pos) { i += 1; continue; }
                    }
                }
                const size_t i0 = i;
                i += 1;
                if (nl == L && std::equal(tp + i0, tp + i0 + L, npep)) continue;  // self-similar (:363-365)
                const std::string_view tumor_pep(reinterpret_cast<const char*>(tp + i0), L);
                if (row.transcript == current->transcript && row.somatic_positions == current->somatic_positions &&
                    row.germline_positions == current->germline_positions) {
                    if (seen_peptides.contains(tumor_pep)) continue;
                } else {
                    current = &row;
                    seen_peptides.clear();
                }
                seen_peptides.insert(tumor_pep);
                const Entry e{uint32_t(r), uint32_t(i), aa_off[2 * r] + i0, npep, uint32_t(nl), 0};
                const Key key{row.frame, row.somatic_positions, row.germline_positions};
                const double alt = row.freq * double(row.depth);
                if (row.transcript != region->transcript || row.variant_sites != region->variant_sites) {  // :398-541
                    flush(false);
                    Pending& p = add_slot(key);
                    p.alt.push_back(alt); p.depth.push_back(row.depth); p.recs.push_back(e);
                    region = &row;
                } else {
                    // entry(key).or_insert_with(|| vec![x]).push(x): a key first seen here holds its first value twice (:548-560)
                    Pending* p = find_slot(key);
                    if (!p) { p = &add_slot(key); p->alt.push_back(alt); p->depth.push_back(row.depth); p->recs.push_back(e); }
                    p->alt.push_back(alt); p->depth.push_back(row.depth); p->recs.push_back(e);
                }
            }
        }
        flush(true);
        lap("row stream");
        out.n_peptides = entries.size();
        out.n_groups = grp_final.size();

        // ---- K6
        std::vector<CredibleInterval> ci(grp_final.size());
        if (!grp_final.empty()) {
            std::vector<double> ln_fact(171);
            {
                double f = 1.0;   // statrs FCACHE: f64 factorials by repeated multiplication, then .ln()
                ln_fact[0] = std::log(1.0);
                for (int k = 1; k < 171; k++) { f *= double(k); ln_fact[k] = std::log(f); }
            }
            uint64_t* d_goff = to_device(grp_off, stream, owned);
            uint8_t* d_gfin = to_device(grp_final, stream, owned);
            double* d_alt = to_device(g_alt, stream, owned);
            uint32_t* d_dep = to_device(g_depth, stream, owned);
            double* d_lf = to_device(ln_fact, stream, owned);
            void* d_ci = nullptr;
            HIP_OK(hipMalloc(&d_ci, ci.size() * sizeof(CredibleInterval))); owned.push_back(d_ci);
            HIP_OK(hipEventRecord(e2, stream));
            device_credible_intervals(d_goff, d_gfin, d_alt, d_dep, ci.size(), d_lf, static_cast<CredibleInterval*>(d_ci), stream);
            HIP_OK(hipEventRecord(e3, stream));
            HIP_OK(hipMemcpyAsync(ci.data(), d_ci, ci.size() * sizeof(CredibleInterval), hipMemcpyDeviceToHost, stream));
            HIP_OK(hipStreamSynchronize(stream));
            HIP_OK(hipEventElapsedTime(&out.stats_ms, e2, e3));
        }
        lap("K6 credible intervals");

        // ---- emission (:483-533, :662-706): the entries in order, written by all host threads (entry ranges) and joined
        for (const Entry& e : entries)
            if (ci[e.group].status) throw Error("reference would panic: called `Option::unwrap()` on a `None` value (partial_cmp of a NaN likelihood)");
        struct Streams { TextBuf tsv, removed_tsv, fasta, normal_fasta, removed_fasta; uint64_t kept = 0, removed = 0; };
        const size_t parts = std::max<size_t>(1, std::min<size_t>(threads, entries.size() / 2048 + 1));
        std::vector<Streams> part(parts);
        run_parts(parts, [&](size_t t) {
            Streams& o = part[t];
            const size_t lo = entries.size() * t / parts, hi = entries.size() * (t + 1) / parts;
            o.tsv.reserve((hi - lo) * 320);
            std::string id;
            for (size_t k = lo; k < hi; k++) {
                const Entry& e = entries[k];
                const Row& row = rows[e.row];
                const CredibleInterval& c = ci[e.group];
                char buf[64];
                std::snprintf(buf, sizeof buf, "%.2f-%.2f", c.a, c.b);
                id.clear();
                append_u64(id, e.i1);
                id.push_back('_');
                id.append(row.id.data(), row.id.size());
                const double freq = row.depth == 0 ? 0.0 : double(c.ml) * 0.01;
                const std::string_view tumor_pep(reinterpret_cast<const char*>(aa.data() + e.tumor_at), L);
                const std::string_view normal_pep(reinterpret_cast<const char*>(e.normal_pep), e.normal_len);
                if (flags[e.tumor_at] & 2) {
                    put_fasta(o.removed_fasta, id, aa.data() + e.tumor_at, L);
                    write_filtered_record(o.removed_tsv, row, freq, id, buf, normal_pep, tumor_pep);
                    o.removed++;
                } else {
                    put_fasta(o.fasta, id, aa.data() + e.tumor_at, L);
                    if (e.normal_len) put_fasta(o.normal_fasta, id, e.normal_pep, e.normal_len);
                    write_filtered_record(o.tsv, row, freq, id, buf, normal_pep, tumor_pep);
                    o.kept++;
                }
            }
        });
        for (const Streams& o : part) { out.n_kept += o.kept; out.n_removed += o.removed; }
        const std::string_view header(FILTERED_HEADER);
        auto join = [&](PodVec<char>& dst, TextBuf Streams::*m, bool with_header) {
            std::vector<size_t> at(parts + 1, with_header ? header.size() : 0);
            for (size_t t = 0; t < parts; t++) at[t + 1] = at[t] + (part[t].*m).size();
            dst.resize(at[parts]);
            advise_huge(dst.data(), dst.size());
            if (with_header) std::memcpy(dst.data(), header.data(), header.size());
            run_parts(parts, [&](size_t t) {
                TextBuf& src = part[t].*m;
                if (!src.empty()) std::memcpy(dst.data() + at[t], src.data(), src.size());
                src = TextBuf();
            });
        };
        join(out.tsv, &Streams::tsv, true);
        join(out.removed_tsv, &Streams::removed_tsv, out.n_removed > 0);   // its header is written with its first record
        join(out.fasta, &Streams::fasta, false);
        join(out.normal_fasta, &Streams::normal_fasta, false);
        join(out.removed_fasta, &Streams::removed_fasta, false);
        lap("emission");
    } catch (...) {
        cleanup();
        throw;
    }
    cleanup();
}

}  // namespace mp
