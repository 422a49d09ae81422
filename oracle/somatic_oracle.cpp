// TEST INFRASTRUCTURE - NOT PRODUCT CODE. See somatic_oracle.hpp.
//
// Sequential CPU restatement of the reference's somatic phasing engine. Every function cites
// the reference lines it follows. Quirks of the reference are reproduced on purpose (sticky
// bad_qual, forward `contains` no-op, reverse retry of rejected reads, the j-stuck variant
// cursor, shift_in_window / frame latches, stop-terminates-transcript, ...).
#include "somatic_oracle.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <deque>
#include <limits>
#include <map>
#include <tuple>

#include "oracle_util.hpp"

using namespace mp;

namespace mp_oracle {
namespace {

using Bytes = std::vector<uint8_t>;
using FsFreq = std::map<uint64_t, std::pair<double, bool>>;  // frameshift_frequencies

[[noreturn]] void ref_panic(const char* what) { throw Error(std::string("reference would panic: ") + what); }

inline bool bitvector_is_set(uint64_t b, size_t k) { return (b & (uint64_t(1) << k)) != 0; }  // microphasing.rs:22-24

inline uint8_t switch_ascii_case(uint8_t c, uint8_t r) {  // :26-32
    if (r >= 'A' && r <= 'Z') return (c >= 'A' && c <= 'Z') ? uint8_t(c + 32) : c;
    return c;
}
inline void switch_ascii_case_vec(const std::string& v, uint8_t r, Bytes& out) {  // :34-40
    bool upper_ref = r >= 'A' && r <= 'Z';
    for (char ch : v) {
        uint8_t c = uint8_t(ch);
        if (upper_ref) out.push_back((c >= 'A' && c <= 'Z') ? uint8_t(c + 32) : c);
        else out.push_back((c >= 'a' && c <= 'z') ? uint8_t(c - 32) : c);
    }
}

bool has_stop_codon(const std::string& peptide, bool forward) {  // :42-76
    if (peptide.size() < 3) return false;
    auto starts = [&](size_t c, const char* codon) { return peptide.compare(c, 3, codon) == 0 && c + 3 <= peptide.size(); };
    if (!forward) {
        size_t c = peptide.size() - 3;
        for (;;) {
            if (starts(c, "TCA") || starts(c, "CTA") || starts(c, "TTA")) return true;
            if (c < 3) return false;
            c -= 3;
        }
    }
    for (size_t c = 0; c < peptide.size(); c += 3)
        if (starts(c, "TGA") || starts(c, "TAG") || starts(c, "TAA")) return true;
    return false;
}

struct Ctx {
    const ReadStore& rs;
};

bool bad_quality(const Ctx& cx, size_t read, const Variant& v) {  // :78-93
    if (v.kind != VK_SNV) return false;
    uint64_t relative_pos = v.pos - uint64_t(cx.rs.pos[read]);
    if (relative_pos < cx.rs.l_seq[read]) {
        if (cx.rs.qual(read)[relative_pos] < 10) return true;
    }
    return false;
}

bool supports_variant(const Ctx& cx, size_t read, const Variant& v) {  // :95-139
    const ReadStore& rs = cx.rs;
    switch (v.kind) {
        case VK_SNV: {
            uint64_t relative_pos = v.pos - uint64_t(rs.pos[read]);
            if (relative_pos < rs.l_seq[read]) {
                if (rs.qual(read)[relative_pos] < 10) return false;
            }
            int64_t p = oracle_read_pos(rs.cigar(read), rs.n_cigar[read], rs.pos[read], int64_t(uint32_t(v.pos)));
            if (p < 0) return false;
            if (uint64_t(p) >= rs.l_seq[read]) ref_panic("seq index out of range");
            return rs.base(read, uint32_t(p)) == v.alt;
        }
        case VK_INS: {
            for (uint32_t k = 0; k < rs.n_cigar[read]; k++) {
                uint32_t c = rs.cigar(read)[k];
                if ((c & 0xF) == C_I && (c >> 4) == uint32_t(v.len)) return true;
            }
            return false;
        }
        default: {
            for (uint32_t k = 0; k < rs.n_cigar[read]; k++) {
                uint32_t c = rs.cigar(read)[k];
                if ((c & 0xF) == C_D && (c >> 4) == uint32_t(v.len)) return true;
            }
            return false;
        }
    }
}

struct HaplotypeSeq {  // :141-145
    IDRecord record;
};

struct Observation {  // :147-154
    size_t read;
    uint64_t haplotype = 0;
    uint64_t frame0 = 0, frame1 = 0;
    bool bad_qual = false, start_loss = false;

    void update_haplotype(const Ctx& cx, size_t i, const Variant& variant, bool has_start_loss) {  // :157-197
        if (uint64_t(cx.rs.pos[read]) > variant.pos) ref_panic("bug: read starts right of variant");
        if (variant.frameshift() > 0) frame1 += variant.pos;
        if (supports_variant(cx, read, variant)) {
            if (has_start_loss) start_loss = true;
            haplotype |= uint64_t(1) << i;
            frame0 += variant.frameshift();
        }
        if (bad_quality(cx, read, variant) || bad_qual || start_loss) {
            haplotype = 0;
            bad_qual = true;
        }
    }
};

inline bool contains_pos(const std::vector<uint64_t>& v, uint64_t p) { return std::find(v.begin(), v.end(), p) != v.end(); }

// IDRecord::update (reference: src/common.rs:376-526)
IDRecord record_update(const IDRecord& self, const IDRecord& rec, uint64_t offset, uint64_t frame, double freq,
                       const std::string& wt_seq, const std::string& mt_seq, uint64_t wlen) {
    std::string fasta_id = haplotype_id(reinterpret_cast<const uint8_t*>(mt_seq.data()), mt_seq.size(), self.transcript,
                                        offset, self.strand.empty() ? '?' : self.strand[0]);
    auto split = [](const std::string& s) {
        std::vector<std::string> out;
        size_t a = 0;
        for (;;) {
            size_t b = s.find('|', a);
            if (b == std::string::npos) { out.push_back(s.substr(a)); break; }
            out.push_back(s.substr(a, b - a));
            a = b + 1;
        }
        return out;
    };
    auto parse = [](const std::string& p) { return uint64_t(std::strtoull(p.c_str(), nullptr, 10)); };
    auto at = [](const std::vector<std::string>& v, size_t c) -> const std::string& {
        if (c >= v.size()) ref_panic("aa_change index out of range");
        return v[c];
    };
    auto somatic_positions = split(self.somatic_positions);
    auto somatic_aa_change = split(self.somatic_aa_change);
    auto other_somatic_aa_change = split(rec.somatic_aa_change);
    auto germline_positions = split(self.germline_positions);
    auto germline_aa_change = split(self.germline_aa_change);
    auto other_germline_aa_change = split(rec.germline_aa_change);
    std::vector<std::string> s_p, g_p, s_aa, g_aa;
    uint32_t nvariants = 0, nsomatic = 0;
    bool fwd = self.strand == "Forward";
    size_t c = 0;
    for (const auto& p : somatic_positions) {
        if (p.empty()) break;
        bool active = fwd ? (self.offset + offset <= parse(p)) : (self.offset + wlen - offset >= parse(p));
        if (active) { s_p.push_back(p); s_aa.push_back(at(somatic_aa_change, c)); nsomatic++; nvariants++; }
        c++;
    }
    c = 0;
    for (const auto& p : split(rec.somatic_positions)) {
        if (p.empty()) break;
        bool active = fwd ? (rec.offset + offset >= parse(p)) : (rec.offset + wlen - 3 - offset <= parse(p));
        if (active) { s_p.push_back(p); s_aa.push_back(at(other_somatic_aa_change, c)); nsomatic++; nvariants++; }
        c++;
    }
    c = 0;
    for (const auto& p : germline_positions) {
        if (p.empty()) break;
        if (self.offset + offset <= parse(p)) { g_p.push_back(p); g_aa.push_back(at(germline_aa_change, c)); nvariants++; }
        c++;
    }
    c = 0;
    for (const auto& p : split(rec.germline_positions)) {
        if (p.empty()) break;
        if (rec.offset >= parse(p) - offset) { g_p.push_back(p); g_aa.push_back(at(other_germline_aa_change, c)); nvariants++; }
        c++;
    }
    uint64_t new_offset = fwd ? self.offset + offset : rec.offset + wlen + 3 - offset;
    uint32_t new_depth = (rec.depth == 0 || self.depth == 0) ? 0 : (rec.depth + self.depth) / 2;
    std::string vr = self.variant_sites + "|" + rec.variant_sites;
    if (!vr.empty() && vr.front() == '|') vr = vr.substr(1);
    if (!vr.empty() && vr.back() == '|') vr.pop_back();
    auto join = [](const std::vector<std::string>& v) {
        std::string s;
        for (size_t i = 0; i < v.size(); i++) { if (i) s += "|"; s += v[i]; }
        return s;
    };
    IDRecord r;
    r.id = fasta_id;
    r.transcript = self.transcript; r.gene_id = self.gene_id; r.gene_name = self.gene_name; r.chrom = self.chrom;
    r.offset = new_offset; r.frame = frame; r.freq = freq; r.depth = new_depth;
    r.nvar = nvariants; r.nsomatic = nsomatic;
    r.nvariant_sites = self.nvariant_sites + rec.nvariant_sites;
    r.nsomvariant_sites = self.nsomvariant_sites + rec.nsomvariant_sites;
    r.strand = self.strand; r.variant_sites = vr;
    r.somatic_positions = join(s_p); r.somatic_aa_change = join(s_aa);
    r.germline_positions = join(g_p); r.germline_aa_change = join(g_aa);
    r.normal_sequence = wt_seq; r.mutant_sequence = mt_seq;
    return r;
}

// IDRecord::add_freq (reference: src/common.rs:528-568)
IDRecord record_add_freq(const IDRecord& self, double freq) {
    IDRecord r = self;
    uint32_t new_nvar = self.nvar == 0 ? self.nvar : (freq > 0.0 ? self.nvar - 1 : self.nvar);
    uint32_t new_somatic = new_nvar < self.nsomatic ? self.nsomatic - 1 : self.nsomatic;
    double new_freq = self.freq > 0.5 ? self.freq : self.freq + freq;
    r.nvar = new_nvar; r.nsomatic = new_somatic; r.freq = new_freq;
    return r;
}

struct ObservationMatrix {  // :200-351
    std::map<uint64_t, std::vector<Observation>> observations;
    std::deque<const Variant*> variants;

    uint32_t ncols() const { return uint32_t(variants.size()); }
    size_t nrows() const {
        size_t n = 0;
        for (const auto& kv : observations) n += kv.second.size();
        return n;
    }

    void shrink_left(size_t k) {  // :220-229
        if (k > variants.size()) ref_panic("drain range out of bounds");
        variants.erase(variants.begin(), variants.begin() + long(k));
        if (ncols() >= 64) ref_panic("2u64.pow(ncols) overflow");
        uint64_t mask = (uint64_t(1) << ncols()) - 1;
        for (auto& kv : observations)
            for (auto& obs : kv.second) obs.haplotype &= mask;
    }

    void extend_right(const Ctx& cx, const std::vector<const Variant*>& new_variants, const std::vector<uint64_t>& start_loss) {  // :232-256
        size_t k = new_variants.size();
        if (k > 0)
            for (auto& kv : observations)
                for (auto& obs : kv.second) obs.haplotype <<= k;
        for (auto& kv : observations)
            for (auto& obs : kv.second)
                for (size_t i = 0; i < k; i++) {
                    const Variant* v = new_variants[k - 1 - i];
                    obs.update_haplotype(cx, i, *v, contains_pos(start_loss, v->pos));
                }
        for (const Variant* v : new_variants) variants.push_back(v);
    }

    void cleanup_reads(uint64_t interval_end, bool reverse) {  // :259-278
        auto it = observations.lower_bound(interval_end);
        if (!reverse) observations.erase(observations.begin(), it);  // keep keys >= interval_end
        else observations.erase(it, observations.end());            // keep keys <  interval_end
    }

    bool contains(const Ctx& cx, size_t read) const {  // :281-294
        uint64_t pos = uint64_t(cx.rs.pos[read]);
        auto it = observations.find(pos);
        if (it == observations.end()) return false;
        const char* qn = cx.rs.qname(read);
        for (const auto& obs : it->second)
            if (std::strcmp(cx.rs.qname(obs.read), qn) == 0) return true;
        return false;
    }

    void push_read(const Ctx& cx, size_t read, uint64_t interval_end, uint64_t interval_start, bool reverse,
                   const std::vector<uint64_t>& start_loss) {  // :297-343
        uint64_t end_pos = uint64_t(cx.rs.end_pos[read]);
        uint64_t start_pos = uint64_t(cx.rs.pos[read]);
        if (end_pos >= interval_end && start_pos <= interval_start && !contains(cx, read)) {
            Observation obs;
            obs.read = read;
            size_t n = variants.size();
            for (size_t i = 0; i < n; i++) {
                const Variant* v = variants[n - 1 - i];
                obs.update_haplotype(cx, i, *v, contains_pos(start_loss, v->pos));
            }
            uint64_t pos = reverse ? start_pos : end_pos;
            if (obs.bad_qual) return;
            observations[pos].push_back(obs);
        }
    }
};

std::string slice_str(const Bytes& v, size_t a, size_t b) {
    if (a > b || b > v.size()) ref_panic("slice index out of range");
    return std::string(reinterpret_cast<const char*>(v.data()) + a, b - a);
}

// ObservationMatrix::print_haplotypes (reference: src/microphasing.rs:353-880)
std::pair<std::vector<HaplotypeSeq>, FsFreq> print_haplotypes(
    const ObservationMatrix& om, const Gene& gene, const Transcript& transcript, uint64_t offset, uint64_t splice_end,
    uint64_t splice_pos, uint64_t splice_gap, uint64_t /*exon_end*/, uint64_t /*exon_start*/, uint64_t window_len,
    const Bytes& refseq, SomaticOutput& out, bool is_short_exon, uint64_t frame_in, FsFreq frameshift_frequencies,
    bool is_first_exon_window) {
    bool is_fwd = transcript.strand == FORWARD;
    std::vector<const Variant*> variants(om.variants.begin(), om.variants.end());  // :373-379
    if (!is_fwd) std::reverse(variants.begin(), variants.end());
    uint64_t frame = frame_in;
    size_t frame_depth = 0;
    std::map<std::pair<uint64_t, uint64_t>, size_t> haplotypes;  // :383-411
    for (const auto& kv : om.observations)
        for (const auto& obs : kv.second) {
            if (obs.bad_qual) continue;
            if (frame > 0 && obs.frame0 != frame && obs.frame1 != 0) continue;
            frame_depth++;
            if (frame > 0) haplotypes[{obs.haplotype, frame}]++;
            else haplotypes[{obs.haplotype, obs.frame0}]++;
        }
    Bytes seq, germline_seq;
    const char* strand = is_fwd ? "Forward" : "Reverse";
    bool has_frameshift = frame > 0;
    std::vector<HaplotypeSeq> haplotypes_vec;
    if (haplotypes.empty()) haplotypes[{0, 0}] = 0;  // :429-431
    uint64_t shift_in_window = 0;
    auto ref_at = [&](uint64_t i) -> uint8_t {
        uint64_t k = i - gene.start();
        if (k >= refseq.size()) ref_panic("refseq index out of range");
        return refseq[k];
    };
    const uint32_t depth = uint32_t(om.nrows());

    if (const char* tr = std::getenv("MP_TRACE")) {
        FILE* tf = std::fopen(tr, "a");
        std::fprintf(tf, "P %s %llu f%llu depth=%u fd=%zu first=%d :", transcript.id.c_str(), (unsigned long long)offset, (unsigned long long)frame, depth, frame_depth, int(is_first_exon_window));
        for (const auto& hk : haplotypes) std::fprintf(tf, " (%llu,%llu)=%zu", (unsigned long long)hk.first.first, (unsigned long long)hk.first.second, hk.second);
        std::fprintf(tf, "\n");
        std::fclose(tf);
    }
    for (const auto& hk : haplotypes) {  // :434
        uint64_t haplotype = hk.first.first;
        uint64_t haplotype_frame = hk.first.second;
        size_t count = hk.second;
        bool indel = false, insertion = false, shift_is_set = false;
        seq.clear();
        germline_seq.clear();
        uint32_t n_somatic = 0, n_variants = 0;
        double freq = count == 0 ? 0.0 : double(count) / double(frame_depth);
        uint64_t i = offset;
        size_t j = 0;
        uint64_t window_end = splice_end;
        std::vector<uint8_t> variant_profile;
        if (variants.empty()) {  // :464-471
            for (uint64_t p = offset; p < window_end; p++) { uint8_t b = ref_at(p); germline_seq.push_back(b); seq.push_back(b); }
        } else {
            while (i < window_end) {  // :473
                while (j < variants.size() && i == variants[j]->pos) {  // :479
                    const Variant& v = *variants[j];
                    shift_in_window = shift_in_window > 0 ? shift_in_window : v.frameshift();
                    size_t bit_pos = is_fwd ? variants.size() - 1 - j : j;
                    if (bitvector_is_set(haplotype, bit_pos)) {
                        if (shift_in_window > 0) {  // :494-502
                            shift_is_set = true;
                            frameshift_frequencies[v.frameshift()] = {freq, !v.is_germline};
                            frameshift_frequencies[0] = {1.0 - freq, false};
                        }
                        bool broke = false;
                        switch (v.kind) {
                            case VK_SNV: {  // :505-521
                                if (v.is_germline) germline_seq.push_back(switch_ascii_case(v.alt, ref_at(i)));
                                else germline_seq.push_back(ref_at(i));
                                seq.push_back(switch_ascii_case(v.alt, ref_at(i)));
                                i += 1;
                                break;
                            }
                            case VK_INS: {  // :523-545
                                if (v.is_germline) switch_ascii_case_vec(v.seq, ref_at(i), germline_seq);
                                else indel = true;
                                switch_ascii_case_vec(v.seq, ref_at(i), seq);
                                insertion = true;
                                i += 1;
                                break;
                            }
                            case VK_DEL: {  // :547-577
                                if (!is_fwd && v.end_pos() >= window_end) { broke = true; break; }
                                if (v.is_germline || i == window_end - 1) {
                                    germline_seq.push_back(ref_at(i));
                                } else {
                                    for (uint64_t p = i; p < i + v.len + 1; p++) germline_seq.push_back(ref_at(p));
                                    indel = true;
                                }
                                seq.push_back(ref_at(i));
                                i += v.len + 1;
                                break;
                            }
                        }
                        if (broke) break;  // leaves the inner while (:551)
                        if (!v.is_germline) { n_somatic++; variant_profile.push_back(2); }
                        else variant_profile.push_back(1);
                        n_variants++;
                    } else {
                        variant_profile.push_back(0);
                    }
                    j++;
                }
                if (i < window_end) {  // :595-599
                    seq.push_back(ref_at(i));
                    germline_seq.push_back(ref_at(i));
                    i++;
                }
            }
        }
        double frame_frequency = freq;  // :605
        if (shift_is_set && frame == 0) frame = shift_in_window;
        frameshift_frequencies.emplace(frame, std::make_pair(0.0, false));  // entry().or_insert
        if (shift_in_window == 0) frame_frequency = freq * frameshift_frequencies.at(frame).first;
        if (shift_in_window == 0 && haplotype_frame > 0 && frame == 0) frame_frequency = 0.0;
        if ((indel && insertion) ||
            (shift_in_window == 0 && (frameshift_frequencies.at(frame).second || (has_frameshift && germline_seq != seq)))) {
            germline_seq.clear();
        }
        uint64_t this_window_len = seq.size() < window_len ? seq.size() : window_len;  // :651-654
        uint64_t normal_window_len = indel ? (germline_seq.size() < window_len ? germline_seq.size() : window_len) : this_window_len;
        std::string fasta_id = haplotype_id(seq.data(), seq.size(), transcript.id, offset, strand[0]);  // :667-675
        std::string normal_peptide;  // :677-684
        if (!germline_seq.empty()) {
            if (splice_pos == 1) normal_peptide = slice_str(germline_seq, splice_gap, germline_seq.size());
            else if (splice_pos == 0) normal_peptide = slice_str(germline_seq, 0, normal_window_len);
            else normal_peptide = slice_str(germline_seq, 0, germline_seq.size());
        }
        std::string neopeptide;  // :686-693
        if (splice_pos == 1) neopeptide = slice_str(seq, splice_gap, seq.size());
        else if (splice_pos == 0) neopeptide = insertion ? slice_str(seq, 0, seq.size()) : slice_str(seq, 0, this_window_len);
        else neopeptide = slice_str(seq, 0, seq.size());
        bool stop_gain = has_stop_codon(neopeptide, is_fwd);  // :694-697
        bool remove_peptide = false;
        if (stop_gain && splice_pos != 2 && (window_len == this_window_len || indel) && !is_first_exon_window &&
            ((normal_peptide != neopeptide) || !indel || std::fabs(freq - 1.0) < std::numeric_limits<double>::epsilon())) {  // :703-718
            remove_peptide = true;
            if (frame == 0) frameshift_frequencies[frame] = {0.0, false};
            else frameshift_frequencies.erase(frame);
        }
        // meta information (:720-764)
        uint32_t n_variantsites = 0, n_som_variantsites = 0;
        std::string som_pos, som_pc, germ_pos, germ_pc, sites;
        auto add = [](std::string& s, const std::string& x, bool& first) { if (!first) s += "|"; s += x; first = false; };
        bool f_sp = true, f_spc = true, f_gp = true, f_gpc = true, f_vs = true;
        for (size_t c = 0; c < variants.size(); c++) {
            if (c < variant_profile.size()) {
                if (variant_profile[c] == 2) { add(som_pos, std::to_string(variants[c]->pos + 1), f_sp); add(som_pc, variants[c]->prot_change, f_spc); }
                else if (variant_profile[c] == 1) { add(germ_pos, std::to_string(variants[c]->pos + 1), f_gp); add(germ_pc, variants[c]->prot_change, f_gpc); }
            }
            if (c == 0 || variants[c]->pos != variants[c - 1]->pos) {
                n_variantsites++;
                add(sites, std::to_string(variants[c]->pos + 1), f_vs);
                if (!variants[c]->is_germline) n_som_variantsites++;
            }
        }
        uint64_t inframe_offset = splice_pos == 0 ? offset + 1 : offset + 1 + splice_gap;  // :766-769
        IDRecord record;  // :772-794
        record.id = fasta_id; record.transcript = transcript.id; record.gene_id = gene.id; record.gene_name = gene.name;
        record.chrom = gene.chrom; record.offset = inframe_offset; record.frame = frame; record.freq = frame_frequency;
        record.depth = depth; record.nvar = n_variants; record.nsomatic = n_somatic;
        record.nvariant_sites = n_variantsites; record.nsomvariant_sites = n_som_variantsites; record.strand = strand;
        record.variant_sites = sites; record.somatic_positions = som_pos; record.somatic_aa_change = som_pc;
        record.germline_positions = germ_pos; record.germline_aa_change = germ_pc;
        record.normal_sequence = normal_peptide; record.mutant_sequence = neopeptide;
        HaplotypeSeq hap_seq;  // :807-832 (carries the UNSLICED sequences)
        hap_seq.record = record;
        hap_seq.record.normal_sequence.assign(reinterpret_cast<const char*>(germline_seq.data()), germline_seq.size());
        hap_seq.record.mutant_sequence.assign(reinterpret_cast<const char*>(seq.data()), seq.size());
        if (!remove_peptide || frame == 0) haplotypes_vec.push_back(std::move(hap_seq));  // :835-837
        if ((record.nsomatic > 0 || has_frameshift) && !is_short_exon && germline_seq != seq && record.freq > 0.0 &&
            (!stop_gain || has_frameshift)) {  // :839-875
            if (splice_pos == 1) {
                if (splice_gap > seq.size()) ref_panic("slice index out of range");
                write_fasta(out.fasta, record.id, seq.data() + splice_gap, seq.size() - splice_gap);
            } else if (splice_pos == 0) {
                write_fasta(out.fasta, record.id, seq.data(), this_window_len);
            }
            if (!germline_seq.empty()) {
                if (splice_pos == 1) {
                    if (splice_gap > germline_seq.size()) ref_panic("slice index out of range");
                    write_fasta(out.normal_fasta, record.id, germline_seq.data() + splice_gap, germline_seq.size() - splice_gap);
                } else if (splice_pos == 0) {
                    if (this_window_len > germline_seq.size()) ref_panic("slice index out of range");
                    write_fasta(out.normal_fasta, record.id, germline_seq.data(), this_window_len);
                }
            }
            write_tsv_record(out, record);
        }
    }
    return {std::move(haplotypes_vec), std::move(frameshift_frequencies)};
}

template <class M, class K>
size_t count_range(const M& tree, K lo, K hi) {  // flatten(tree.range(lo..hi)).count()
    if (lo > hi) ref_panic("range start is greater than range end in BTreeMap");
    size_t n = 0;
    for (auto it = tree.lower_bound(lo); it != tree.end() && it->first < hi; ++it) n += it->second.size();
    return n;
}

}  // namespace

// phase_gene (reference: src/microphasing.rs:882-1941)
void phase_gene(const GeneInput& gi, const ReadStore& rs, uint64_t window_len, SomaticOutput& out) {
    const Gene& gene = gi.gene;
    const Bytes& refseq = gi.refseq;
    Ctx cx{rs};
    std::map<uint64_t, std::vector<const Variant*>> variant_tree;
    std::map<uint64_t, std::vector<size_t>> read_tree;
    uint64_t max_read_len = 0;
    for (size_t r : gi.reads) {  // :909-920
        if (rs.mapq[r] < 5) continue;
        if (uint64_t(rs.l_seq[r]) > max_read_len) max_read_len = rs.l_seq[r];
        read_tree[uint64_t(rs.pos[r])].push_back(r);
    }
    for (const Variant& v : gi.variants) variant_tree[v.pos].push_back(&v);  // :932-942 (already de-duplicated per POS)

    for (const Transcript& transcript : gene.transcripts) {  // :944
        if (!transcript.is_coding()) continue;
        const bool is_fwd = transcript.strand == FORWARD;
        size_t exon_number = transcript.exons.size();
        ObservationMatrix observations;
        std::map<uint64_t, uint64_t> frameshifts;
        std::vector<uint64_t> deletions;
        if (is_fwd) frameshifts[0] = 0; else frameshifts[gene.end()] = 0;
        uint64_t exon_rest = 0;
        std::vector<HaplotypeSeq> prev_hap_vec, hap_vec;
        FsFreq frameshift_frequencies;
        frameshift_frequencies[0] = {1.0, false};
        std::vector<uint64_t> start_loss;
        size_t last_window_vars = 0;
        size_t exon_count = 0;
        for (const Interval& exon : transcript.exons) {  // :974
            if (frameshifts.empty()) break;
            if (exon.start > exon.end) continue;
            exon_count++;
            uint64_t exon_len = exon.end - exon.start;
            uint64_t current_exon_offset = exon_count == 1 ? exon.frame : (exon_rest == 0 ? 0 : 3 - exon_rest);
            bool is_last_exon = exon_count == exon_number;
            bool is_first_exon = exon_count == 1;
            bool is_short_exon = exon_len < 3 ? true : window_len >= exon_len - current_exon_offset - (3 - current_exon_offset) % 3;
            uint64_t exon_window_len = !is_short_exon ? window_len : (exon_len - current_exon_offset) - ((exon_len - current_exon_offset) % 3);
            if (exon_window_len == 0) exon_window_len = exon_len;
            exon_rest = 0;
            uint64_t offset = !is_fwd ? exon.end - exon_window_len - current_exon_offset : exon.start + current_exon_offset;
            bool reached_end = false;
            uint64_t old_offset = offset;
            uint64_t old_end = old_offset + exon_window_len;
            observations.shrink_left(last_window_vars);  // :1027
            last_window_vars = 0;
            bool is_first_exon_window = true;
            for (;;) {  // :1030
                if (frameshifts.empty()) break;
                bool valid = is_fwd ? offset + exon_window_len <= exon.end : offset >= exon.start;
                bool read_through = is_last_exon && !valid;
                if (!valid) break;
                if (max_read_len < exon_window_len) break;
                uint64_t rest = is_fwd ? exon.end - (offset + exon_window_len) : offset - exon.start;
                bool is_last_exon_window = rest < 3;
                uint64_t splice_side_offset, splice_end, splice_gap, splice_pos;  // :1058-1111
                if (is_fwd) {
                    if (is_short_exon || (is_first_exon_window && is_last_exon_window)) {
                        splice_side_offset = offset - current_exon_offset; splice_end = offset + exon_window_len + rest;
                        splice_gap = current_exon_offset + rest; splice_pos = 2;
                    } else if (is_first_exon_window) {
                        splice_side_offset = offset - current_exon_offset; splice_end = offset + exon_window_len;
                        splice_gap = current_exon_offset; splice_pos = 1;
                    } else if (is_last_exon_window) {
                        splice_side_offset = offset; splice_end = offset + exon_window_len + rest; splice_gap = rest; splice_pos = 0;
                    } else {
                        splice_side_offset = offset; splice_end = offset + exon_window_len; splice_gap = 0; splice_pos = 0;
                    }
                } else {
                    if (is_short_exon) {
                        splice_side_offset = offset - rest; splice_end = offset + exon_window_len + current_exon_offset;
                        splice_gap = current_exon_offset + rest; splice_pos = 2;
                    } else if (is_first_exon_window) {
                        splice_side_offset = offset; splice_end = offset + exon_window_len + current_exon_offset;
                        splice_gap = current_exon_offset; splice_pos = 0;
                    } else if (is_last_exon_window) {
                        splice_side_offset = offset - rest; splice_end = offset + exon_window_len; splice_gap = rest; splice_pos = 1;
                    } else {
                        splice_side_offset = offset; splice_end = offset + exon_window_len; splice_gap = 0; splice_pos = 0;
                    }
                }
                size_t nvars = count_range(variant_tree, splice_side_offset, splice_end);  // :1119-1124
                last_window_vars = nvars;
                size_t added_vars;  // :1129-1156
                if (is_first_exon_window) added_vars = nvars;
                else if (is_short_exon && !read_through) added_vars = 0;
                else if (reached_end && !read_through) added_vars = 0;
                else if (splice_side_offset > old_offset) added_vars = count_range(variant_tree, old_end, splice_end);
                else added_vars = count_range(variant_tree, splice_side_offset, old_offset);
                size_t deleted_vars;  // :1159-1178
                if (offset == old_offset || (is_short_exon && !read_through)) deleted_vars = 0;
                else if (splice_side_offset > old_offset) deleted_vars = count_range(variant_tree, old_offset, splice_side_offset);
                else deleted_vars = count_range(variant_tree, splice_end, old_end);
                if (is_last_exon_window && !read_through) reached_end = true;
                // candidate reads (:1191-1249)
                std::vector<size_t> reads;
                {
                    uint64_t lo, hi = splice_side_offset + 1;
                    bool first_of_exon = is_fwd ? offset == exon.start + current_exon_offset : true;
                    if (!is_fwd || first_of_exon) {
                        if (splice_side_offset < max_read_len - exon_window_len) ref_panic("attempt to subtract with overflow (#1)");
                        lo = splice_side_offset - (max_read_len - exon_window_len);
                    } else {
                        lo = splice_side_offset;
                    }
                    for (auto it = read_tree.lower_bound(lo); it != read_tree.end() && it->first < hi; ++it)
                        for (size_t r : it->second) reads.push_back(r);
                }
                {
                    bool reverse = !is_fwd;
                    if (reverse) observations.cleanup_reads(splice_side_offset + 1, reverse);  // :1259-1263
                    else observations.cleanup_reads(splice_end, reverse);
                    observations.shrink_left(deleted_vars);  // :1265
                    for (size_t r : reads) observations.push_read(cx, r, splice_end, splice_side_offset, reverse, start_loss);  // :1269-1277
                    // collect variants (:1280-1296)
                    std::vector<const Variant*> variants;
                    {
                        std::vector<const Variant*> all;
                        auto lo_it = variant_tree.lower_bound(splice_side_offset);
                        auto hi_it = variant_tree.lower_bound(splice_end);
                        if (is_fwd) {
                            for (auto it = lo_it; it != hi_it; ++it)
                                for (const Variant* v : it->second) all.push_back(v);
                        } else {
                            for (auto it = hi_it; it != lo_it;) {
                                --it;
                                for (const Variant* v : it->second) all.push_back(v);
                            }
                        }
                        if (added_vars > nvars) ref_panic("attempt to subtract with overflow (#2)");
                        for (size_t k = nvars - added_vars; k < all.size(); k++) variants.push_back(all[k]);
                    }
                    for (const Variant* variant : variants) {  // :1299-1342
                        bool is_start_loss = is_fwd ? (is_first_exon && variant->pos >= exon.start && variant->pos < exon.start + 3)
                                                    : (is_first_exon && variant->pos < exon.end && variant->pos >= exon.end - 3);
                        if (is_start_loss) start_loss.push_back(variant->pos);
                        if (variant->kind == VK_DEL) deletions.push_back(is_fwd ? variant->end_pos() : variant->pos);
                        uint64_t s = variant->frameshift();
                        if ((s % 3) > 0) {
                            std::vector<uint64_t> previous;
                            for (const auto& kv : frameshifts) previous.push_back(kv.second + s);
                            for (uint64_t s_ : previous) frameshifts[is_fwd ? variant->end_pos() : variant->pos] = s_ % 3;
                        }
                    }
                    observations.extend_right(cx, variants, start_loss);  // :1345
                    uint64_t stopped_frameshift = 3;
                    // active frameshifts (:1347-1350)
                    std::vector<std::pair<uint64_t, uint64_t>> active;
                    if (is_fwd) {
                        for (auto it = frameshifts.begin(); it != frameshifts.end() && it->first < offset; ++it) active.push_back(*it);
                    } else {
                        for (auto it = frameshifts.lower_bound(offset + exon_window_len); it != frameshifts.end(); ++it) active.push_back(*it);
                    }
                    bool closed_deletion = deletions.empty() ? false : (is_fwd ? deletions[0] < offset : deletions[0] >= offset + exon_window_len);
                    size_t frameshift_count = 0;
                    bool main_orf = false;
                    for (const auto& kf : active) {  // :1362
                        uint64_t key = kf.first, frameshift = kf.second;
                        frameshift_count++;
                        if (frameshift == 0) main_orf = true;
                        uint64_t coding_shift = is_fwd ? offset - exon.start : exon.end - offset;
                        bool has_frameshift = frameshift > 0;
                        if (coding_shift % 3 == (frameshift + current_exon_offset) % 3 || (is_short_exon && !read_through)) {
                            if (!has_frameshift && !read_through) {  // :1386-1400
                                exon_rest = is_fwd ? exon.end - (offset + exon_window_len) : offset - exon.start;
                                if (exon_window_len < 3) exon_rest = exon_window_len;
                            }
                            if (frameshift == 0) out.n_windows++;
                            auto res = print_haplotypes(observations, gene, transcript, splice_side_offset, splice_end, splice_pos,
                                                        splice_gap, exon.end, exon.start, exon_window_len, refseq, out, is_short_exon,
                                                        frameshift, std::move(frameshift_frequencies), is_first_exon_window);
                            frameshift_frequencies = std::move(res.second);
                            if (res.first.empty() || !frameshift_frequencies.count(frameshift)) stopped_frameshift = key;
                            if (closed_deletion) deletions.clear();
                            if (exon_rest < 3 && (!is_short_exon || is_first_exon) && !has_frameshift && !read_through) prev_hap_vec = std::move(res.first);
                            else hap_vec = std::move(res.first);
                            if (frameshift != 0 && frameshift_frequencies.count(frameshift) && frameshift_frequencies.at(frameshift).first == 0.0)
                                stopped_frameshift = key;
                        }
                    }
                    if (frameshift_count == 0 || !main_orf || !frameshift_frequencies.count(0)) {  // :1465-1473
                        frameshifts.clear();
                        break;
                    }
                    if (stopped_frameshift != 3) {  // :1477-1481
                        auto it = frameshifts.find(stopped_frameshift);
                        if (it == frameshifts.end()) ref_panic("unwrap on None (stopped_frameshift)");
                        if (it->second != 0) frameshifts.erase(it);
                    }
                    if (frameshifts.empty()) break;
                    if (frameshift_frequencies.at(0).first == 0.0 && frameshifts.size() == 1) {  // :1485-1488
                        frameshifts.clear();
                        break;
                    }
                    bool at_splice_side = is_fwd ? offset - current_exon_offset == exon.start
                                                 : offset + exon_window_len + current_exon_offset == exon.end;  // :1497-1502
                    is_first_exon_window = false;
                    if (at_splice_side && !is_first_exon) {  // :1505
                        const std::vector<HaplotypeSeq>& first_hap_vec = is_fwd ? hap_vec : prev_hap_vec;
                        const std::vector<HaplotypeSeq>& sec_hap_vec = is_fwd ? prev_hap_vec : hap_vec;
                        using Key = std::tuple<uint64_t, std::string, std::string>;
                        std::map<Key, std::tuple<std::string, IDRecord, std::string>> output_map;
                        std::vector<HaplotypeSeq> new_hap_vec;
                        if (const char* tr = std::getenv("MP_TRACE")) {
                            FILE* tf = std::fopen(tr, "a");
                            std::fprintf(tf, "M %s %llu\n", transcript.id.c_str(), (unsigned long long)offset);
                            for (const auto& h : first_hap_vec) std::fprintf(tf, "  F %.17g %s %s\n", h.record.freq, h.record.mutant_sequence.c_str(), h.record.normal_sequence.c_str());
                            for (const auto& h : sec_hap_vec) std::fprintf(tf, "  S %.17g %s %s\n", h.record.freq, h.record.mutant_sequence.c_str(), h.record.normal_sequence.c_str());
                            std::fclose(tf);
                        }
                        for (const HaplotypeSeq& hapseq : first_hap_vec) {  // :1527
                            const IDRecord& record = hapseq.record;
                            const std::string& wt_sequence = record.normal_sequence;
                            const std::string& mt_sequence = record.mutant_sequence;
                            for (const HaplotypeSeq& prev_hapseq : sec_hap_vec) {
                                const IDRecord& prev_record = prev_hapseq.record;
                                const std::string& prev_wt_sequence = prev_record.normal_sequence;
                                const std::string& prev_mt_sequence = prev_record.mutant_sequence;
                                std::string new_wt = prev_wt_sequence + wt_sequence;
                                std::vector<std::string> new_mt_sequences;  // :1545-1558
                                if (wt_sequence != mt_sequence) {
                                    new_mt_sequences.push_back(prev_wt_sequence + mt_sequence);
                                    if (prev_wt_sequence != prev_mt_sequence) {
                                        new_mt_sequences.push_back(prev_mt_sequence + wt_sequence);
                                        new_mt_sequences.push_back(prev_mt_sequence + mt_sequence);
                                    }
                                } else {
                                    new_mt_sequences.push_back(prev_mt_sequence + mt_sequence);
                                }
                                auto merge_freq = [&]() {
                                    return std::fabs(record.freq - prev_record.freq) < std::numeric_limits<double>::epsilon()
                                               ? record.freq : record.freq * prev_record.freq;
                                };
                                if (is_short_exon && !is_last_exon) {  // :1565-1587
                                    HaplotypeSeq nh;
                                    nh.record = record_update(prev_record, record, 0, record.frame, merge_freq(), new_wt, new_wt, window_len);
                                    new_hap_vec.push_back(std::move(nh));
                                }
                                for (const std::string& new_mt : new_mt_sequences) {  // :1590
                                    if (is_short_exon && !is_last_exon) {  // :1598-1623
                                        HaplotypeSeq nh;
                                        nh.record = record_update(prev_record, record, 0, record.frame, merge_freq(), new_wt, new_mt, window_len);
                                        new_hap_vec.push_back(std::move(nh));
                                        continue;
                                    }
                                    std::vector<std::pair<uint64_t, uint64_t>> active2;  // :1624-1629
                                    if (is_fwd) {
                                        for (auto it = frameshifts.begin(); it != frameshifts.end() && it->first < offset; ++it) active2.push_back(*it);
                                    } else {
                                        for (auto it = frameshifts.lower_bound(offset + exon_window_len); it != frameshifts.end(); ++it) active2.push_back(*it);
                                    }
                                    for (const auto& pf : active2) {  // :1632
                                        uint64_t pos = pf.first, frameshift = pf.second;
                                        frameshift_frequencies.emplace(frameshift, std::make_pair(0.0, false));
                                        bool shift_in_window = is_fwd ? pos >= prev_record.offset : pos < record.offset + exon_window_len;
                                        bool somatic_shift = frameshift_frequencies.at(frameshift).second;
                                        double frameshift_freq = frameshift_frequencies.at(frameshift).first;
                                        double f0 = frameshift_frequencies.at(0).first;
                                        double main_orf_freq = f0 == 0.0 ? frameshift_freq : f0;
                                        double shift_orf_freq = shift_in_window ? frameshift_freq : (f0 == 0.0 ? frameshift_freq : f0);
                                        double variant_freq_record = is_fwd ? record.freq / main_orf_freq : record.freq / shift_orf_freq;
                                        double variant_freq_prev_record = is_fwd ? prev_record.freq / shift_orf_freq : prev_record.freq / main_orf_freq;
                                        double freq_record = f0 == 0.0 ? frameshift_freq : variant_freq_record * frameshift_freq;
                                        double freq_prev_record = f0 == 0.0 ? frameshift_freq : variant_freq_prev_record * frameshift_freq;
                                        double out_freq = std::fabs(record.freq - prev_record.freq) < std::numeric_limits<double>::epsilon()
                                                              ? freq_record : freq_record * freq_prev_record;  // :1706-1711
                                        uint64_t out_shift = shift_in_window ? 0 : frameshift;
                                        uint64_t splice_offset = 3 - out_shift;  // :1719
                                        if (!is_fwd && exon_rest < 3) splice_offset += exon_rest;
                                        size_t end_offset = 3 + size_t(out_shift);
                                        if (is_last_exon_window) end_offset = 0;
                                        if (uint64_t(new_mt.size()) < 2 * window_len) {
                                            if (is_fwd) splice_offset = 0; else end_offset = 0;
                                        }
                                        for (;;) {  // :1743
                                            if (end_offset > new_mt.size()) ref_panic("attempt to subtract with overflow (#3)");
                                            if (!(splice_offset + window_len <= uint64_t(new_mt.size() - end_offset))) break;
                                            std::string out_wt_seq;
                                            if (splice_offset + window_len <= uint64_t(new_wt.size())) {
                                                if (is_fwd) out_wt_seq = new_wt.substr(size_t(splice_offset), size_t(window_len));
                                                else {
                                                    if (new_wt.size() < end_offset + window_len) ref_panic("attempt to subtract with overflow (#4)");
                                                    out_wt_seq = new_wt.substr(new_wt.size() - end_offset - size_t(window_len), size_t(window_len));
                                                }
                                            }
                                            std::string out_mt_seq;
                                            if (is_fwd) out_mt_seq = new_mt.substr(size_t(splice_offset), size_t(window_len));
                                            else out_mt_seq = new_mt.substr(new_mt.size() - end_offset - size_t(window_len), size_t(window_len));
                                            if (out_shift > 0 && out_wt_seq == out_mt_seq && somatic_shift) out_wt_seq.clear();  // :1794-1799
                                            if (out_wt_seq == out_mt_seq || (out_wt_seq.empty() && frameshift == 0)) {  // :1801-1810
                                                if (is_fwd) splice_offset += 3; else end_offset += 3;
                                                continue;
                                            }
                                            uint64_t out_offset = is_fwd ? splice_offset : uint64_t(end_offset);
                                            IDRecord out_record = is_fwd
                                                ? record_update(prev_record, record, out_offset, frameshift, out_freq, out_wt_seq, out_mt_seq, window_len)
                                                : record_update(record, prev_record, out_offset, frameshift, out_freq, out_wt_seq, out_mt_seq, window_len);
                                            Key id_tuple{out_offset, out_mt_seq, out_wt_seq};
                                            auto fit = output_map.find(id_tuple);
                                            double old_freq = fit == output_map.end() ? 0.0 : std::get<1>(fit->second).freq;
                                            output_map[id_tuple] = std::make_tuple(out_mt_seq, record_add_freq(out_record, old_freq), out_wt_seq);
                                            if (is_fwd) splice_offset += 3; else end_offset += 3;
                                        }
                                    }
                                }
                            }
                        }
                        if (is_short_exon && !is_last_exon) {  // :1874
                            prev_hap_vec = std::move(new_hap_vec);
                        } else {
                            for (const auto& kv : output_map) {
                                const std::string& out_mt_seq = std::get<0>(kv.second);
                                const IDRecord& out_record = std::get<1>(kv.second);
                                const std::string& out_wt_seq = std::get<2>(kv.second);
                                if (out_mt_seq != out_wt_seq) {
                                    if (out_mt_seq.size() < window_len) ref_panic("slice index out of range");
                                    write_fasta(out.fasta, out_record.id, reinterpret_cast<const uint8_t*>(out_mt_seq.data()), size_t(window_len));
                                    if (!out_wt_seq.empty()) {
                                        if (out_wt_seq.size() < window_len) ref_panic("slice index out of range");
                                        write_fasta(out.normal_fasta, out_record.id, reinterpret_cast<const uint8_t*>(out_wt_seq.data()), size_t(window_len));
                                    }
                                    write_tsv_record(out, out_record);
                                }
                            }
                            if (is_short_exon) prev_hap_vec = std::move(new_hap_vec);
                        }
                    }
                    old_offset = splice_side_offset;  // :1909-1914
                    old_end = splice_end;
                    if (is_fwd) offset += 1; else offset -= 1;
                    if (frameshifts.empty()) break;
                }
                if (frameshifts.empty()) break;
                if (is_short_exon) break;  // :1928-1931
            }
        }
    }
}

}  // namespace mp_oracle
