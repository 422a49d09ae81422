#include "filter.hpp"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <limits>
#include <map>
#include <set>
#include <tuple>
#include <vector>

#include "kernels_filter.hpp"
#include "kernels_pep.hpp"
#include "pep.hpp"
#include "util.hpp"

namespace mp {

[[noreturn]] void throw_hip(hipError_t e, const char* file, int line);
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw_hip(e_, __FILE__, __LINE__); } while (0)

namespace {

// csv::ReaderBuilder::new().delimiter(b'\t') with default quoting: records of fields
std::vector<std::vector<std::string>> parse_tsv(const std::string& text) {
    std::vector<std::vector<std::string>> rows;
    std::vector<std::string> row;
    std::string field;
    bool quoted = false, started = false;
    auto end_record = [&] {
        if (started || !field.empty()) { row.push_back(field); rows.push_back(row); }
        row.clear(); field.clear(); started = false;
    };
    for (size_t i = 0; i < text.size(); i++) {
        const char c = text[i];
        if (quoted) {
            if (c != '"') field.push_back(c);
            else if (i + 1 < text.size() && text[i + 1] == '"') { field.push_back('"'); i++; }
            else quoted = false;
        } else if (c == '"' && field.empty()) { quoted = true; started = true; }
        else if (c == '\t') { row.push_back(field); field.clear(); started = true; }
        else if (c == '\n') end_record();
        else if (c == '\r') { if (i + 1 < text.size() && text[i + 1] == '\n') i++; end_record(); }
        else { field.push_back(c); started = true; }
    }
    end_record();
    return rows;
}
uint64_t field_u64(const std::string& s, const char* name) {
    if (s.empty() || s.find_first_not_of("0123456789") != std::string::npos)
        throw Error(std::string("CSV deserialize error: field ") + name + ": invalid digit found in string");
    return std::strtoull(s.c_str(), nullptr, 10);
}
IDRecord row_from_fields(const std::vector<std::string>& f) {  // serde positional deserialize of IDRecord (src/common.rs:350-373)
    if (f.size() != 21) throw Error("CSV deserialize error: found record with " + std::to_string(f.size()) + " fields, but expected 21");
    IDRecord r;
    r.id = f[0]; r.transcript = f[1]; r.gene_id = f[2]; r.gene_name = f[3]; r.chrom = f[4];
    r.offset = field_u64(f[5], "offset");
    r.frame = field_u64(f[6], "frame");
    char* e = nullptr;
    r.freq = std::strtod(f[7].c_str(), &e);
    if (f[7].empty() || *e) throw Error("CSV deserialize error: field freq: invalid float literal");
    r.depth = uint32_t(field_u64(f[8], "depth"));
    r.nvar = uint32_t(field_u64(f[9], "nvar"));
    r.nsomatic = uint32_t(field_u64(f[10], "nsomatic"));
    r.nvariant_sites = uint32_t(field_u64(f[11], "nvariant_sites"));
    r.nsomvariant_sites = uint32_t(field_u64(f[12], "nsomvariant_sites"));
    r.strand = f[13]; r.variant_sites = f[14]; r.somatic_positions = f[15]; r.somatic_aa_change = f[16];
    r.germline_positions = f[17]; r.germline_aa_change = f[18]; r.normal_sequence = f[19]; r.mutant_sequence = f[20];
    return r;
}

// FilteredRecord::FIELD_NAMES_AS_ARRAY (src/peptides.rs:21-47)
const char* const FILTERED_HEADER =
    "id\ttranscript\tgene_id\tgene_name\tchrom\toffset\tframe\tfreq\tcredible_interval\tdepth\tnvar\tnsomatic\tnvariant_sites\tnsomvariant_sites\t"
    "strand\tvariant_sites\tsomatic_positions\tsomatic_aa_change\tgermline_positions\tgermline_aa_change\tnormal_sequence\tmutant_sequence\t"
    "normal_peptide\ttumor_peptide\n";

void write_filtered_record(std::string& t, const IDRecord& r, double freq, const std::string& id, const char* ci, const std::string& normal_pep,
                           const std::string& tumor_pep) {
    auto S = [&](const std::string& f) { tsv_field(t, f); t.push_back('\t'); };
    auto U = [&](uint64_t v) { t += std::to_string(v); t.push_back('\t'); };
    S(id); S(r.transcript); S(r.gene_id); S(r.gene_name); S(r.chrom); U(r.offset); U(r.frame);
    t += fmt_f64(freq); t.push_back('\t');
    S(ci); U(r.depth); U(r.nvar); U(r.nsomatic); U(r.nvariant_sites); U(r.nsomvariant_sites);
    S(r.strand); S(r.variant_sites); S(r.somatic_positions); S(r.somatic_aa_change); S(r.germline_positions); S(r.germline_aa_change);
    S(r.normal_sequence); S(r.mutant_sequence); S(normal_pep);
    tsv_field(t, tumor_pep);
    t.push_back('\n');
}

template <class T>
T* to_device(const std::vector<T>& v, hipStream_t stream, std::vector<void*>& owned, size_t pad = 0) {
    void* p = nullptr;
    HIP_OK(hipMalloc(&p, v.size() * sizeof(T) + pad + 16));
    owned.push_back(p);
    if (!v.empty()) HIP_OK(hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, stream));
    return static_cast<T*>(p);
}

}  // namespace

void filter_device(int device, const std::string& reference_binary, const std::string& tsv_text, uint32_t L, FilterResult& out) {
    if (L == 0 || L > 12) throw Error("peptide length must be 1..12 for the device peptidome (5-bit residue keys in a u64)");
    out = FilterResult();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        throw Error("no HIP device available: `filter` translates and scores on the GPU, there is no CPU fallback");
    HIP_OK(hipSetDevice(device));

    // ---- reference peptidome: bincode v1 HashSet<Vec<u8>> (deserialize_from(...).unwrap(), :242) -> keys of the length-L members
    std::vector<uint64_t> ref_keys;
    {
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
                if (letters) ref_keys.push_back(peptide_to_key(reference_binary.substr(p, l)));
            }
            p += l;
        }
    }

    // ---- rows and their two nucleotide windows
    const auto fields = parse_tsv(tsv_text);
    std::vector<IDRecord> rows;
    for (size_t i = 1; i < fields.size(); i++) rows.push_back(row_from_fields(fields[i]));  // first record = header
    out.n_rows = rows.size();
    const size_t n_seq = rows.size() * 2;   // 2r = mutant, 2r + 1 = normal
    std::vector<uint8_t> nt, rev(n_seq);
    std::vector<uint64_t> nt_off(n_seq), aa_off(n_seq + 1, 0);
    std::vector<uint32_t> nt_len(n_seq);
    for (size_t r = 0; r < rows.size(); r++) {
        const uint8_t rv = (!rows[r].id.empty() && rows[r].id.back() == 'F') ? 0 : 1;  // :291-294
        const std::string* seqs[2] = {&rows[r].mutant_sequence, &rows[r].normal_sequence};
        for (int k = 0; k < 2; k++) {
            const size_t s = 2 * r + k;
            if (seqs[k]->size() > 0xFFFFFFFFull) throw Error("sequence too long");
            nt_off[s] = nt.size();
            nt_len[s] = uint32_t(seqs[k]->size());
            rev[s] = rv;
            nt.insert(nt.end(), seqs[k]->begin(), seqs[k]->end());
            aa_off[s + 1] = aa_off[s] + (seqs[k]->size() > 2 ? (seqs[k]->size() - 2 + 2) / 3 : 0);
        }
    }
    const uint64_t n_aa = aa_off[n_seq];

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
        // sorted distinct reference keys (device radix sort + unique)
        uint64_t n_ref = 0;
        uint64_t* d_ref = nullptr;
        if (!ref_keys.empty()) {
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
        std::vector<uint8_t> aa(n_aa), flags(n_aa);
        uint32_t err = 0;
        if (n_aa) {
            HIP_OK(hipMemcpyAsync(aa.data(), d_aa, n_aa, hipMemcpyDeviceToHost, stream));
            HIP_OK(hipMemcpyAsync(flags.data(), d_flags, n_aa, hipMemcpyDeviceToHost, stream));
        }
        HIP_OK(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        HIP_OK(hipEventElapsedTime(&out.translate_ms, e0, e1));

        // ---- the row stream (:262-563): which (row, i) peptides survive, and the groups they are scored in
        struct Entry { uint32_t row; uint32_t i1; uint64_t tumor_at; std::string normal_pep; uint64_t group; };
        std::vector<Entry> entries;                 // in output order
        std::vector<uint64_t> grp_off{0};
        std::vector<uint8_t> grp_final;
        std::vector<double> g_alt;
        std::vector<uint32_t> g_depth;
        using Key = std::tuple<uint64_t, std::string, std::string>;
        struct Pending { std::vector<double> alt; std::vector<uint32_t> depth; std::vector<Entry> recs; };
        std::map<Key, Pending> pending;             // records / frequencies / depth of the current region
        std::tuple<std::string, std::string, std::string> current{"", "", ""};
        std::pair<std::string, std::string> region_sites{"", ""};
        std::set<std::string> seen_peptides;
        std::map<std::pair<std::string, uint64_t>, size_t> stop_gained;
        auto flush = [&](bool final_pass) {
            for (auto& kv : pending) {
                const uint64_t g = grp_final.size();
                grp_final.push_back(final_pass ? 1 : 0);
                g_alt.insert(g_alt.end(), kv.second.alt.begin(), kv.second.alt.end());
                g_depth.insert(g_depth.end(), kv.second.depth.begin(), kv.second.depth.end());
                grp_off.push_back(g_alt.size());
                for (Entry& e : kv.second.recs) { e.group = g; entries.push_back(std::move(e)); }
            }
            pending.clear();
        };
        for (size_t r = 0; r < rows.size(); r++) {
            const IDRecord& row = rows[r];
            if (row.mutant_sequence.size() < 2 || (!row.normal_sequence.empty() && row.normal_sequence.size() < 2))  // `r.len() - 2` (:139)
                throw Error("reference would panic: attempt to subtract with overflow (to_protein)");
            size_t som_pos = 0;
            if (!row.somatic_positions.empty() && row.somatic_positions.find('|') == std::string::npos) {
                if (row.somatic_positions.find_first_not_of("0123456789") != std::string::npos)
                    throw Error("reference would panic: called `Result::unwrap()` on an `Err` value (somatic_positions)");
                som_pos = size_t(std::strtoull(row.somatic_positions.c_str(), nullptr, 10));
            }
            const size_t offset = size_t(row.offset);
            const uint8_t* tp = aa.data() + aa_off[2 * r];
            const size_t tlen = size_t(aa_off[2 * r + 1] - aa_off[2 * r]);
            const uint8_t* np = aa.data() + aa_off[2 * r + 1];
            const size_t nlen = size_t(aa_off[2 * r + 2] - aa_off[2 * r + 1]);
            for (size_t k = 0; k < tlen + nlen; k++)
                if ((k < tlen ? tp[k] : np[k - tlen]) == '?')
                    throw Error("reference would panic: called `Result::unwrap()` on an `Err` value (codon with a base other than A, C, G, T)");
            const std::pair<std::string, uint64_t> check{row.transcript, row.frame};
            auto sg = stop_gained.find(check);
            if (sg != stop_gained.end()) {  // :304-317
                const bool downstream = row.strand == "Forward" ? offset > sg->second : row.strand == "Reverse" ? offset < sg->second : false;
                if (downstream) continue;
            }
            bool has_x = false;
            for (size_t k = 0; k < tlen; k++) has_x |= tp[k] == 'X';
            if (has_x && (std::fabs(row.freq - 1.0) < std::numeric_limits<double>::epsilon() || row.frame > 0)) stop_gained[check] = offset;
            const uint8_t* tf = flags.data() + aa_off[2 * r];
            size_t i = 0;
            while (i + L <= tlen) {
                if (tf[i] & 1) break;  // the peptide contains a stop (:330-332)
                const bool normal_full = nlen >= i + L;
                const size_t nl = normal_full ? L : nlen;
                const uint8_t* npep = normal_full ? np + i : np;
                if (nl == 0 && som_pos > 0) {  // :343-360
                    if (row.strand == "Forward") {
                        if ((i + L) * 3 + offset <= som_pos) { i += 1; continue; }
                    } else if (row.strand == "Reverse") {
                        if ((tlen - (i + L)) * 3 + offset > som_pos) { i += 1; continue; }
                    }
                }
                const size_t i0 = i;
                i += 1;
                if (nl == L && std::equal(tp + i0, tp + i0 + L, npep)) continue;  // self-similar (:363-365)
                const std::string tumor_pep(reinterpret_cast<const char*>(tp + i0), L);
                const std::tuple<std::string, std::string, std::string> cur{row.transcript, row.somatic_positions, row.germline_positions};
                if (cur == current) {
                    if (seen_peptides.count(tumor_pep)) continue;
                } else {
                    current = cur;
                    seen_peptides.clear();
                }
                seen_peptides.insert(tumor_pep);
                Entry e;
                e.row = uint32_t(r);
                e.i1 = uint32_t(i);
                e.tumor_at = aa_off[2 * r] + i0;
                e.normal_pep.assign(reinterpret_cast<const char*>(npep), nl);
                e.group = 0;
                const Key key{row.frame, row.somatic_positions, row.germline_positions};
                const double alt = row.freq * double(row.depth);
                const std::pair<std::string, std::string> current_sites{row.transcript, row.variant_sites};
                if (current_sites != region_sites) {  // :398-541
                    flush(false);
                    Pending& p = pending[key];
                    p.alt = {alt}; p.depth = {row.depth}; p.recs.push_back(std::move(e));
                    region_sites = current_sites;
                } else {
                    // entry(key).or_insert_with(|| vec![x]).push(x): a key first seen here holds its first value twice (:548-560)
                    const bool fresh = !pending.count(key);
                    Pending& p = pending[key];
                    if (fresh) { p.alt.push_back(alt); p.depth.push_back(row.depth); p.recs.push_back(e); }
                    p.alt.push_back(alt); p.depth.push_back(row.depth); p.recs.push_back(std::move(e));
                }
            }
        }
        flush(true);
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

        // ---- emission (:483-533, :662-706)
        out.tsv += FILTERED_HEADER;
        bool removed_header = false;
        for (const Entry& e : entries) {
            const IDRecord& row = rows[e.row];
            const CredibleInterval& c = ci[e.group];
            if (c.status) throw Error("reference would panic: called `Option::unwrap()` on a `None` value (partial_cmp of a NaN likelihood)");
            char buf[64];
            std::snprintf(buf, sizeof buf, "%.2f-%.2f", c.a, c.b);
            const std::string id = std::to_string(e.i1) + "_" + row.id;
            const double freq = row.depth == 0 ? 0.0 : double(c.ml) * 0.01;
            const std::string tumor_pep(reinterpret_cast<const char*>(aa.data() + e.tumor_at), L);
            if (flags[e.tumor_at] & 2) {
                write_fasta(out.removed_fasta, id, aa.data() + e.tumor_at, L);
                if (!removed_header) { out.removed_tsv += FILTERED_HEADER; removed_header = true; }
                write_filtered_record(out.removed_tsv, row, freq, id, buf, e.normal_pep, tumor_pep);
                out.n_removed++;
            } else {
                write_fasta(out.fasta, id, aa.data() + e.tumor_at, L);
                if (!e.normal_pep.empty())
                    write_fasta(out.normal_fasta, id, reinterpret_cast<const uint8_t*>(e.normal_pep.data()), e.normal_pep.size());
                write_filtered_record(out.tsv, row, freq, id, buf, e.normal_pep, tumor_pep);
                out.n_kept++;
            }
        }
    } catch (...) {
        cleanup();
        throw;
    }
    cleanup();
}

}  // namespace mp
