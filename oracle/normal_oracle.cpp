// TEST INFRASTRUCTURE - NOT PRODUCT CODE.
// Sequential CPU restatement of `microphaser normal` (reference: src/normal_microphasing.rs). Pinned by the
// reference's live fixtures test_forward_germline / splice_test_forward_germline (expected .germline.fa,
// tests/lib.rs:237-285); the TSV of this mode is not covered by any reference fixture.
// Quirks kept on purpose: push_read numbers the existing columns oldest = bit 0 while extend_right keeps
// newest = bit 0 (:260-262 vs :317-319); no `contains`, so on the '-' strand every in-range read is pushed again
// at every step (:942-967, :1010-1017); reverse cleanup without the +1 (:1001); the unconditional trailing base
// (:476); the multi-allelic skip (:429-431); deletions extend the window (:457); 0-based positions (:536).
#include "normal_oracle.hpp"

#include <algorithm>
#include <cmath>
#include <deque>
#include <limits>
#include <map>

#include "oracle_util.hpp"

using namespace mp;

namespace mp_oracle {
namespace {

using Bytes = std::vector<uint8_t>;
[[noreturn]] void ref_panic(const char* what) { throw Error(std::string("reference would panic: ") + what); }

inline uint8_t switch_ascii_case(uint8_t c, uint8_t r) {
    if (r >= 'A' && r <= 'Z') return (c >= 'A' && c <= 'Z') ? uint8_t(c + 32) : c;
    return c;
}
inline void switch_ascii_case_vec(const std::string& v, uint8_t r, Bytes& out) {
    bool upper_ref = r >= 'A' && r <= 'Z';
    for (char ch : v) {
        uint8_t c = uint8_t(ch);
        if (upper_ref) out.push_back((c >= 'A' && c <= 'Z') ? uint8_t(c + 32) : c);
        else out.push_back((c >= 'a' && c <= 'z') ? uint8_t(c - 32) : c);
    }
}

bool supports_variant(const ReadStore& rs, size_t read, const Variant& v) {  // :43-78 (no quality gate)
    switch (v.kind) {
        case VK_SNV: {
            int64_t p = oracle_read_pos(rs.cigar(read), rs.n_cigar[read], rs.pos[read], int64_t(uint32_t(v.pos)));
            if (p < 0) return false;
            if (uint64_t(p) >= rs.l_seq[read]) ref_panic("seq index out of range");
            return rs.base(read, uint32_t(p)) == v.alt;
        }
        case VK_INS:
        default: {
            uint32_t want = v.kind == VK_INS ? C_I : C_D;
            for (uint32_t k = 0; k < rs.n_cigar[read]; k++) {
                uint32_t c = rs.cigar(read)[k];
                if ((c & 0xF) == want && (c >> 4) == uint32_t(v.len)) return true;
            }
            return false;
        }
    }
}

struct Observation {  // :188-216
    size_t read;
    uint64_t haplotype = 0;
    void update_haplotype(const ReadStore& rs, size_t i, const Variant& variant) {
        if (uint64_t(rs.pos[read]) > variant.pos) ref_panic("bug: read starts right of variant");
        if (supports_variant(rs, read, variant)) haplotype |= uint64_t(1) << i;
    }
};

struct HaplotypeSeq {  // :182-186
    Bytes sequence;
    NormalRecord record;
};

// IDRecord::update (:105-146) and add_freq (:148-179) of the normal mode
NormalRecord record_update(const NormalRecord& self, const NormalRecord& rec, uint64_t offset, const Bytes& seq) {
    NormalRecord r = self;
    r.id = haplotype_id(seq.data(), seq.size(), self.transcript, offset, self.strand.empty() ? '?' : self.strand[0]);
    r.somatic_positions = self.somatic_positions + rec.somatic_positions;
    r.somatic_aa_change = self.somatic_aa_change + rec.somatic_aa_change;
    r.germline_positions = self.germline_positions + rec.germline_positions;
    r.germline_aa_change = self.germline_aa_change + rec.germline_aa_change;
    r.offset = offset + self.offset;
    r.freq = self.freq * rec.freq;
    r.nvar = self.nvar + rec.nvar;
    r.nsomatic = self.nsomatic + rec.nsomatic;
    r.nvariant_sites = self.nvariant_sites + rec.nvariant_sites;
    r.nsomvariant_sites = self.nsomvariant_sites + rec.nsomvariant_sites;
    r.variant_sites = self.variant_sites + rec.variant_sites;
    r.peptide_sequence.assign(reinterpret_cast<const char*>(seq.data()), seq.size());
    return r;
}
NormalRecord record_add_freq(const NormalRecord& self, double freq) {
    NormalRecord r = self;
    // `self.nvar - 1` on a u32 (:150): with nvar == 0 a debug build panics, the release build (Cargo.toml has no
    // overflow-checks profile; the published binary) wraps to 4294967295 and prints that. Release behaviour is restated:
    // it is what a user of `microphaser normal` observes, and about one synthetic gene in seven reaches this line.
    if (freq > 0.0) r.nvar = self.nvar - 1u;
    if (r.nvar < self.nsomatic) r.nsomatic = self.nsomatic - 1;
    r.freq = self.freq + freq;
    return r;
}

struct ObservationMatrix {  // :218-339
    std::map<uint64_t, std::vector<Observation>> observations;
    std::deque<const Variant*> variants;
    uint32_t ncols() const { return uint32_t(variants.size()); }
    size_t nrows() const { size_t n = 0; for (const auto& kv : observations) n += kv.second.size(); return n; }
    void shrink_left(size_t k) {
        if (k > variants.size()) ref_panic("drain range out of bounds");
        variants.erase(variants.begin(), variants.begin() + long(k));
        if (ncols() >= 64) ref_panic("2u64.pow(ncols) overflow");
        uint64_t mask = (uint64_t(1) << ncols()) - 1;
        for (auto& kv : observations) for (auto& obs : kv.second) obs.haplotype &= mask;
    }
    void extend_right(const ReadStore& rs, const std::vector<const Variant*>& nv) {
        size_t k = nv.size();
        if (k > 0) for (auto& kv : observations) for (auto& obs : kv.second) obs.haplotype <<= k;
        for (auto& kv : observations) for (auto& obs : kv.second)
            for (size_t i = 0; i < k; i++) obs.update_haplotype(rs, i, *nv[k - 1 - i]);
        for (const Variant* v : nv) variants.push_back(v);
    }
    void cleanup_reads(uint64_t interval_end, bool reverse) {
        auto it = observations.lower_bound(interval_end);
        if (!reverse) observations.erase(observations.begin(), it);
        else observations.erase(it, observations.end());
    }
    void push_read(const ReadStore& rs, size_t read, uint64_t interval_end, uint64_t interval_start, bool reverse) {  // :301-331
        uint64_t end_pos = uint64_t(rs.end_pos[read]), start_pos = uint64_t(rs.pos[read]);
        if (end_pos >= interval_end && start_pos <= interval_start) {
            Observation obs;
            obs.read = read;
            for (size_t i = 0; i < variants.size(); i++) obs.update_haplotype(rs, i, *variants[i]);  // oldest column = bit 0
            observations[reverse ? start_pos : end_pos].push_back(obs);
        }
    }
};

bool starts_with(const std::string& s, const char* p) { return s.size() >= 3 && s.compare(0, 3, p) == 0; }
bool ends_with(const std::string& s, const char* p) { return s.size() >= 3 && s.compare(s.size() - 3, 3, p) == 0; }

// print_haplotypes (:341-647)
std::vector<HaplotypeSeq> print_haplotypes(const ObservationMatrix& om, const Gene& gene, const Transcript& transcript, uint64_t offset,
                                           uint64_t splice_end, uint64_t splice_pos, uint64_t splice_gap, uint64_t window_len,
                                           const Bytes& refseq, NormalOutput& out, bool is_short_exon, uint64_t frame) {
    const bool is_fwd = transcript.strand == FORWARD;
    std::vector<const Variant*> variants(om.variants.begin(), om.variants.end());
    if (!is_fwd) std::reverse(variants.begin(), variants.end());
    std::map<uint64_t, size_t> haplotypes;  // VecMap<usize>: ascending key order
    for (const auto& kv : om.observations) for (const auto& obs : kv.second) haplotypes[obs.haplotype]++;
    const char* strand = is_fwd ? "Forward" : "Reverse";
    std::vector<HaplotypeSeq> haplotypes_vec;
    if (haplotypes.empty()) haplotypes[0] = 0;
    const size_t nrows = om.nrows();
    auto ref_at = [&](uint64_t i) -> uint8_t {
        uint64_t k = i - gene.start();
        if (k >= refseq.size()) ref_panic("refseq index out of range");
        return refseq[k];
    };
    Bytes seq;
    for (const auto& hk : haplotypes) {
        const uint64_t haplotype = hk.first;
        const size_t count = hk.second;
        seq.clear();
        bool insertion = false;
        uint32_t n_somatic = 0, n_variants = 0;
        const double freq = double(count) / double(nrows);  // 0/0 = NaN when no read covers the window
        uint64_t i = offset;
        size_t j = 0;
        uint64_t window_end = splice_end;
        std::vector<uint8_t> variant_profile;
        if (variants.empty()) {
            for (uint64_t p = offset; p < window_end; p++) seq.push_back(ref_at(p));
        } else {
            while (i < window_end) {
                while (j < variants.size() && i == variants[j]->pos) {
                    if (std::fabs(freq - 1.0) < std::numeric_limits<double>::epsilon() && !variants[j]->is_germline) {  // :422-426
                        j += 1;
                        variant_profile.push_back(0);
                        continue;
                    }
                    if (j >= 64) ref_panic("shift overflow (1 << k)");
                    if ((haplotype >> j) & 1) {
                        if (j + 1 < variants.size() && i == variants[j + 1]->pos) j += 1;  // :429-431
                        const Variant& v = *variants[j];
                        switch (v.kind) {
                            case VK_SNV: seq.push_back(switch_ascii_case(v.alt, ref_at(i))); i += 1; break;
                            case VK_INS: switch_ascii_case_vec(v.seq, ref_at(i), seq); insertion = true; i += 1; break;
                            case VK_DEL: seq.push_back(ref_at(i)); i += v.len + 1; window_end += v.len + 1; break;
                        }
                        if (!v.is_germline) { n_somatic++; variant_profile.push_back(2); }
                        else variant_profile.push_back(1);
                        n_variants++;
                    } else {
                        variant_profile.push_back(0);
                    }
                    j += 1;
                }
                seq.push_back(ref_at(i));  // :476 (unconditional)
                i += 1;
            }
        }
        const uint64_t this_window_len = seq.size() < window_len ? seq.size() : window_len;
        auto slice = [&](size_t a, size_t b) {
            if (a > b || b > seq.size()) ref_panic("slice index out of range");
            return std::string(reinterpret_cast<const char*>(seq.data()) + a, b - a);
        };
        std::string peptide = splice_pos == 1 ? slice(splice_gap, seq.size())
                              : splice_pos == 0 ? (insertion ? slice(0, seq.size()) : slice(0, this_window_len)) : slice(0, seq.size());
        const bool stop_gain = is_fwd ? (starts_with(peptide, "TGA") || starts_with(peptide, "TAG") || starts_with(peptide, "TAA"))
                                      : (ends_with(peptide, "TCA") || ends_with(peptide, "CTA") || ends_with(peptide, "TTA"));
        if (stop_gain && splice_pos != 2) continue;  // :503-507
        NormalRecord record;
        record.id = haplotype_id(seq.data(), seq.size(), transcript.id, offset, strand[0]);
        uint32_t n_sites = 0, n_som_sites = 0;
        std::string som_pos, som_pc, germ_pos, germ_pc, sites;
        bool f1 = true, f2 = true, f3 = true, f4 = true, f5 = true;
        auto add = [](std::string& s, const std::string& x, bool& first) { if (!first) s += "|"; s += x; first = false; };
        for (size_t c = 0; c < variants.size(); c++) {  // :531-557
            if (c < variant_profile.size()) {
                if (variant_profile[c] == 2) { add(som_pos, std::to_string(variants[c]->pos), f1); add(som_pc, variants[c]->prot_change, f2); }
                else if (variant_profile[c] == 1) { add(germ_pos, std::to_string(variants[c]->pos), f3); add(germ_pc, variants[c]->prot_change, f4); }
                if (c == 0 || variants[c]->pos != variants[c - 1]->pos) {
                    n_sites++;
                    add(sites, std::to_string(variants[c]->pos), f5);
                    if (!variants[c]->is_germline) n_som_sites++;
                }
            }
        }
        record.transcript = transcript.id; record.gene_id = gene.id; record.gene_name = gene.name; record.chrom = gene.chrom;
        record.offset = offset; record.frame = frame; record.freq = freq; record.depth = uint32_t(nrows);
        record.nvar = n_variants; record.nsomatic = n_somatic; record.nvariant_sites = n_sites; record.nsomvariant_sites = n_som_sites;
        record.strand = strand; record.variant_sites = sites; record.somatic_positions = som_pos; record.somatic_aa_change = som_pc;
        record.germline_positions = germ_pos; record.germline_aa_change = germ_pc; record.peptide_sequence = peptide;
        HaplotypeSeq hs;
        hs.sequence = seq;
        hs.record = record;
        hs.record.peptide_sequence.assign(reinterpret_cast<const char*>(seq.data()), seq.size());
        haplotypes_vec.push_back(std::move(hs));
        if (!is_short_exon) {  // :629-644
            if (splice_pos == 1) {
                if (splice_gap > seq.size()) ref_panic("slice index out of range");
                write_fasta(out.fasta, record.id, seq.data() + splice_gap, seq.size() - splice_gap);
            } else if (splice_pos == 0) {
                if (window_len > seq.size()) ref_panic("slice index out of range");
                write_fasta(out.fasta, record.id, seq.data(), size_t(window_len));
            }
            write_normal_tsv_record(out, record);
        }
    }
    return haplotypes_vec;
}

template <class M, class K>
size_t count_range(const M& tree, K lo, K hi) {
    if (lo > hi) ref_panic("range start is greater than range end in BTreeMap");
    size_t n = 0;
    for (auto it = tree.lower_bound(lo); it != tree.end() && it->first < hi; ++it) n += it->second.size();
    return n;
}

}  // namespace

// phase_gene (reference: src/normal_microphasing.rs:650-1279)
void normal_phase_gene(const GeneInput& gi, const ReadStore& rs, uint64_t window_len, NormalOutput& out) {
    const Gene& gene = gi.gene;
    const Bytes& refseq = gi.refseq;
    std::map<uint64_t, std::vector<const Variant*>> variant_tree;
    std::map<uint64_t, std::vector<size_t>> read_tree;
    uint64_t max_read_len = 0;
    for (size_t r : gi.reads) {  // :676-684 (no mapq filter)
        if (uint64_t(rs.l_seq[r]) > max_read_len) max_read_len = rs.l_seq[r];
        read_tree[uint64_t(rs.pos[r])].push_back(r);
    }
    for (const Variant& v : gi.variants) variant_tree[v.pos].push_back(&v);
    for (const Transcript& transcript : gene.transcripts) {
        if (!transcript.is_coding()) continue;
        const bool is_fwd = transcript.strand == FORWARD;
        size_t exon_number = transcript.exons.size();
        ObservationMatrix observations;
        std::map<uint64_t, uint64_t> frameshifts;
        if (is_fwd) frameshifts[0] = 0; else frameshifts[gene.end()] = 0;
        uint64_t exon_rest = 0;
        std::vector<HaplotypeSeq> prev_hap_vec, hap_vec;
        size_t last_window_vars = 0;
        for (size_t exon_count = 0; exon_count < transcript.exons.size(); exon_count++) {
            const Interval& exon = transcript.exons[exon_count];
            if (frameshifts.empty()) break;
            if (exon.start > exon.end) continue;
            bool is_last_exon = exon_count == exon_number - 1;
            bool is_first_exon = exon_count == 0;
            uint64_t exon_len = exon.end - exon.start;
            uint64_t ceo = exon_rest == 0 ? 0 : 3 - exon_rest;  // :739-742 (ignores exon.frame)
            bool is_short_exon = exon_len < 3 ? true : window_len >= exon_len - ceo - (3 - ceo) % 3;
            uint64_t ewl = !is_short_exon ? window_len : (exon_len - ceo) - ((exon_len - ceo) % 3);
            if (ewl == 0) ewl = exon_len;
            exon_rest = 0;
            uint64_t offset = !is_fwd ? exon.end - ewl - ceo : exon.start + ceo;
            bool reached_end = false;
            uint64_t old_offset = offset, old_end = old_offset + ewl;
            observations.shrink_left(last_window_vars);
            last_window_vars = 0;
            bool is_first_exon_window = true;
            for (;;) {
                if (frameshifts.empty()) break;
                bool valid = is_fwd ? offset + ewl <= exon.end : offset >= exon.start;
                if (!valid) break;
                if (max_read_len < ewl) break;
                uint64_t rest = is_fwd ? exon.end - (offset + ewl) : offset - exon.start;
                bool is_last_exon_window = rest < 3;
                uint64_t sso, splice_end, splice_gap, splice_pos;
                if (is_fwd) {
                    if (is_short_exon || (is_first_exon_window && is_last_exon_window)) { sso = offset - ceo; splice_end = offset + ewl + rest; splice_gap = ceo + rest; splice_pos = 2; }
                    else if (is_first_exon_window) { sso = offset - ceo; splice_end = offset + ewl; splice_gap = ceo; splice_pos = 1; }
                    else if (is_last_exon_window) { sso = offset; splice_end = offset + ewl + rest; splice_gap = rest; splice_pos = 0; }
                    else { sso = offset; splice_end = offset + ewl; splice_gap = 0; splice_pos = 0; }
                } else {
                    if (is_short_exon) { sso = offset - rest; splice_end = offset + ewl + ceo; splice_gap = ceo + rest; splice_pos = 2; }
                    else if (is_first_exon_window) { sso = offset; splice_end = offset + ewl + ceo; splice_gap = ceo; splice_pos = 0; }
                    else if (is_last_exon_window) { sso = offset - rest; splice_end = offset + ewl; splice_gap = rest; splice_pos = 1; }
                    else { sso = offset; splice_end = offset + ewl; splice_gap = 0; splice_pos = 0; }
                }
                size_t nvars = count_range(variant_tree, sso, splice_end);
                last_window_vars = nvars;
                size_t added_vars;
                if (is_first_exon_window) added_vars = nvars;
                else if (is_short_exon) added_vars = 0;
                else if (reached_end) added_vars = 0;
                else if (sso > old_offset) added_vars = count_range(variant_tree, old_end, splice_end);
                else added_vars = count_range(variant_tree, sso, old_offset);
                size_t deleted_vars;
                if (offset == old_offset || is_short_exon) deleted_vars = 0;
                else if (sso > old_offset) deleted_vars = count_range(variant_tree, old_offset, sso);
                else deleted_vars = count_range(variant_tree, splice_end, old_end);
                if (is_last_exon_window) reached_end = true;
                std::vector<size_t> reads;
                {
                    uint64_t lo, hi = sso + 1;
                    bool first_of_exon = is_fwd ? offset == exon.start + ceo : true;
                    if ((!is_fwd || first_of_exon) && sso < max_read_len - ewl) ref_panic("attempt to subtract with overflow (read range)");
                    lo = (!is_fwd || first_of_exon) ? sso - (max_read_len - ewl) : sso;
                    for (auto it = read_tree.lower_bound(lo); it != read_tree.end() && it->first < hi; ++it)
                        for (size_t r : it->second) reads.push_back(r);
                }
                bool reverse = !is_fwd;
                if (reverse) observations.cleanup_reads(sso, reverse);  // :1001
                else observations.cleanup_reads(splice_end, reverse);
                observations.shrink_left(deleted_vars);
                for (size_t r : reads) observations.push_read(rs, r, splice_end, sso, reverse);
                std::vector<const Variant*> variants;
                {
                    std::vector<const Variant*> all;
                    auto lo_it = variant_tree.lower_bound(sso), hi_it = variant_tree.lower_bound(splice_end);
                    if (is_fwd) { for (auto it = lo_it; it != hi_it; ++it) for (const Variant* v : it->second) all.push_back(v); }
                    else { for (auto it = hi_it; it != lo_it;) { --it; for (const Variant* v : it->second) all.push_back(v); } }
                    if (added_vars > nvars) ref_panic("attempt to subtract with overflow (nvars - added_vars)");
                    for (size_t k = nvars - added_vars; k < all.size(); k++) variants.push_back(all[k]);
                }
                for (const Variant* variant : variants) {  // :1039-1049
                    uint64_t s = variant->frameshift();
                    if (s > 0) {
                        std::vector<uint64_t> previous;
                        for (const auto& kv : frameshifts) previous.push_back(kv.second + s);
                        for (uint64_t s_ : previous) frameshifts[variant->end_pos()] = s_;
                    }
                }
                observations.extend_right(rs, variants);
                uint64_t stopped_frameshift = 3;
                std::vector<std::pair<uint64_t, uint64_t>> active;
                if (is_fwd) { for (auto it = frameshifts.begin(); it != frameshifts.end() && it->first < offset; ++it) active.push_back(*it); }
                else { for (auto it = frameshifts.lower_bound(offset + ewl); it != frameshifts.end(); ++it) active.push_back(*it); }
                size_t frameshift_count = 0;
                bool main_orf = false;
                for (const auto& kf : active) {
                    uint64_t key = kf.first, frameshift = kf.second;
                    if (frameshift == 0) main_orf = true;
                    frameshift_count++;
                    uint64_t coding_shift = is_fwd ? offset - exon.start : exon.end - offset;
                    bool has_frameshift = frameshift > 0;
                    if (coding_shift % 3 == (frameshift + ceo) % 3 || is_short_exon) {
                        if (!has_frameshift) {
                            exon_rest = is_fwd ? exon.end - (offset + ewl) : offset - exon.start;
                            if (ewl < 3) exon_rest = ewl;
                        }
                        if (frameshift == 0) out.n_windows++;
                        auto res = print_haplotypes(observations, gene, transcript, sso, splice_end, splice_pos, splice_gap, ewl, refseq, out,
                                                    is_short_exon, frameshift);
                        if (res.empty()) stopped_frameshift = key;
                        if (exon_rest < 3 && (!is_short_exon || is_first_exon)) prev_hap_vec = std::move(res);
                        else hap_vec = std::move(res);
                    }
                }
                if (frameshift_count == 0 || !main_orf) { frameshifts.clear(); break; }
                frameshifts.erase(stopped_frameshift);  // :1130
                if (frameshifts.empty()) break;
                bool at_splice_side = is_fwd ? offset - ceo == exon.start : offset + ewl + ceo == exon.end;
                is_first_exon_window = false;
                if (at_splice_side && !is_first_exon) {  // :1145-1250
                    const std::vector<HaplotypeSeq>& first_hap_vec = is_fwd ? hap_vec : prev_hap_vec;
                    const std::vector<HaplotypeSeq>& sec_hap_vec = is_fwd ? prev_hap_vec : hap_vec;
                    std::map<std::pair<uint64_t, Bytes>, std::pair<Bytes, NormalRecord>> output_map;
                    std::vector<HaplotypeSeq> new_hap_vec;
                    for (const HaplotypeSeq& hapseq : first_hap_vec) {
                        for (const HaplotypeSeq& prev_hapseq : sec_hap_vec) {
                            Bytes prev_sequence = prev_hapseq.sequence;
                            const NormalRecord& prev_record = prev_hapseq.record;
                            prev_sequence.insert(prev_sequence.end(), hapseq.sequence.begin(), hapseq.sequence.end());
                            if (is_short_exon) {
                                HaplotypeSeq nh;
                                nh.sequence = prev_sequence;
                                nh.record = record_update(prev_record, hapseq.record, 0, prev_sequence);
                                new_hap_vec.push_back(std::move(nh));
                            }
                            uint64_t splice_offset = 3;
                            if (!is_fwd && exon_rest < 3) splice_offset += exon_rest;
                            size_t end_offset = 3;
                            if (is_last_exon_window) end_offset = 0;
                            if (uint64_t(prev_sequence.size()) < 2 * window_len) {
                                if (is_fwd) splice_offset = 0; else end_offset = 0;
                            }
                            for (;;) {
                                if (end_offset > prev_sequence.size()) ref_panic("attempt to subtract with overflow (merge)");
                                if (!(splice_offset + window_len <= uint64_t(prev_sequence.size() - end_offset))) break;
                                Bytes out_seq(prev_sequence.begin() + long(splice_offset), prev_sequence.begin() + long(splice_offset + window_len));
                                NormalRecord out_record = record_update(prev_record, hapseq.record, splice_offset, out_seq);
                                auto key = std::make_pair(splice_offset, out_seq);
                                auto fit = output_map.find(key);
                                double old_freq = fit == output_map.end() ? 0.0 : fit->second.second.freq;
                                output_map[key] = std::make_pair(out_seq, record_add_freq(out_record, old_freq));
                                splice_offset += 3;
                            }
                        }
                    }
                    if (is_short_exon && !is_last_exon) {
                        prev_hap_vec = std::move(new_hap_vec);
                    } else {
                        for (const auto& kv : output_map) {
                            const Bytes& out_seq = kv.second.first;
                            const NormalRecord& out_record = kv.second.second;
                            if (out_seq.size() < window_len) ref_panic("slice index out of range");
                            write_fasta(out.fasta, out_record.id, out_seq.data(), size_t(window_len));
                            write_normal_tsv_record(out, out_record);
                        }
                    }
                }
                old_offset = sso;
                old_end = splice_end;
                if (is_fwd) offset += 1; else offset -= 1;
                if (frameshifts.empty()) break;
                if (is_short_exon) break;
            }
        }
    }
}

}  // namespace mp_oracle
