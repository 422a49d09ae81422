// Consumer: turns the kernels' per-window results into the reference's output streams.
//
// It re-runs the window scheduler (walk.hpp) and answers every print_haplotypes call
// (reference: src/microphasing.rs:353-880) from the device results:
//   * haplotype keys / counts / depth      <- K2 (WinDyn + Group)
//   * sequences, variant profile, stop, id  <- K3 (GroupSum + HapRec)
// and keeps, on the host, exactly the state the reference threads through the windows of a
// transcript: frameshift_frequencies, the `frame` / `shift_in_window` latches, termination,
// hap_vec / prev_hap_vec and the splice-side merge (:1505-1908, src/common.rs:376-568).
// No sequence is built and no read is touched on the host.
#include "consume.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <limits>
#include <thread>
#include <tuple>

#include "hostsha.hpp"
#include "rowfmt.hpp"

namespace mp {

#ifdef MP_PROFILE   // development build only (-DMP_PROFILE): cycle counts of the consumer's sections, printed at exit
struct ProfAcc { uint64_t cyc[8] = {0}; uint64_t n[8] = {0}; ~ProfAcc() { static const char* nm[8] = {"walk", "print", "merge", "print.rows", "merge.update", "merge.emit", "", ""}; for (int i = 0; i < 6; i++) std::fprintf(stderr, "[prof] %-14s %10.1f ms %10llu calls\n", nm[i], double(cyc[i]) / 2.0e6 /* ~GHz-agnostic: printed as cycles/2e6 */, (unsigned long long)n[i]); } };
static ProfAcc g_prof;
struct ProfScope { int k; uint64_t t0; explicit ProfScope(int k_) : k(k_), t0(__builtin_ia32_rdtsc()) {} ~ProfScope() { g_prof.cyc[k] += __builtin_ia32_rdtsc() - t0; g_prof.n[k]++; } };
#define PROF(k) ProfScope prof_scope_##k(k)
#else
#define PROF(k) do {} while (0)
#endif

namespace {

// One '|'-separated list, read in place: item(c) for c = 0, 1, 2, ... in order (what indexing the reference's split('|') vector gives)
struct BarList {
    std::string_view s;
    size_t at = 0;        // start of the current item
    bool valid = true;    // split("") has one (empty) item
    explicit BarList(std::string_view s_) : s(s_) {}
    std::string_view item() const {
        const size_t e = s.find('|', at);
        return s.substr(at, e == std::string_view::npos ? std::string_view::npos : e - at);
    }
    void next() {
        const size_t e = s.find('|', at);
        if (e == std::string_view::npos) valid = false; else at = e + 1;
    }
};
inline uint64_t leading_u64(std::string_view p) {   // strtoull of a decimal field
    uint64_t v = 0;
    std::from_chars(p.data(), p.data() + p.size(), v);
    return v;
}

// IDRecord::update (reference: src/common.rs:376-526), written into `r` (every field is set; r may hold an earlier record).
// with_id = false leaves the id (a SHA-1 over the formatted sequence) to the caller: the splice-side merge only needs it for the
// records it finally writes, a fraction of those it builds.
void record_update(IDRecord& r, const IDRecord& self, const IDRecord& rec, uint64_t offset, uint64_t frame, double freq, std::string_view wt_seq,
                   std::string_view mt_seq, uint64_t wlen, bool with_id = true) {
    if (with_id)
        haplotype_id_into(r.id, reinterpret_cast<const uint8_t*>(mt_seq.data()), mt_seq.size(), self.transcript, offset,
                          self.strand.empty() ? '?' : self.strand[0]);
    else
        r.id.clear();
    const bool fwd = self.strand == "Forward";
    uint32_t nvariants = 0, nsomatic = 0;
    // the positions of a list that are still inside the merged window, with their protein changes, joined by '|' (a list ends at its
    // first empty item)
    auto take = [&](std::string& out_p, std::string& out_aa, bool& first, const std::string& positions, const std::string& aa_changes, auto&& active,
                    bool somatic) {
        BarList pos(positions), aa(aa_changes);
        for (; pos.valid; pos.next(), aa.next()) {
            const std::string_view p = pos.item();
            if (p.empty()) break;
            if (!active(leading_u64(p))) continue;
            if (!aa.valid) throw Error("reference would panic: index out of bounds (aa_change)");
            const std::string_view a = aa.item();
            if (!first) { out_p.push_back('|'); out_aa.push_back('|'); }
            out_p.append(p.data(), p.size());
            out_aa.append(a.data(), a.size());
            first = false;
            if (somatic) nsomatic++;
            nvariants++;
        }
    };
    r.somatic_positions.clear(); r.somatic_aa_change.clear(); r.germline_positions.clear(); r.germline_aa_change.clear();
    bool first_s = true, first_g = true;
    take(r.somatic_positions, r.somatic_aa_change, first_s, self.somatic_positions, self.somatic_aa_change,
         [&](uint64_t n) { return fwd ? (self.offset + offset <= n) : (self.offset + wlen - offset >= n); }, true);
    take(r.somatic_positions, r.somatic_aa_change, first_s, rec.somatic_positions, rec.somatic_aa_change,
         [&](uint64_t n) { return fwd ? (rec.offset + offset >= n) : (rec.offset + wlen - 3 - offset <= n); }, true);
    take(r.germline_positions, r.germline_aa_change, first_g, self.germline_positions, self.germline_aa_change,
         [&](uint64_t n) { return self.offset + offset <= n; }, false);
    take(r.germline_positions, r.germline_aa_change, first_g, rec.germline_positions, rec.germline_aa_change,
         [&](uint64_t n) { return rec.offset >= n - offset; }, false);
    r.transcript = self.transcript; r.gene_id = self.gene_id; r.gene_name = self.gene_name; r.chrom = self.chrom;
    r.offset = fwd ? self.offset + offset : rec.offset + wlen + 3 - offset;
    r.frame = frame;
    r.freq = freq;
    r.depth = (rec.depth == 0 || self.depth == 0) ? 0 : (rec.depth + self.depth) / 2;
    r.nvar = nvariants;
    r.nsomatic = nsomatic;
    r.nvariant_sites = self.nvariant_sites + rec.nvariant_sites;
    r.nsomvariant_sites = self.nsomvariant_sites + rec.nsomvariant_sites;
    r.strand = self.strand;
    {   // self.variant_sites + "|" + rec.variant_sites, without a leading / trailing '|'
        std::string& vr = r.variant_sites;
        vr.clear();
        vr += self.variant_sites;
        vr.push_back('|');
        vr += rec.variant_sites;
        if (!vr.empty() && vr.front() == '|') vr.erase(vr.begin());
        if (!vr.empty() && vr.back() == '|') vr.pop_back();
    }
    r.normal_sequence.assign(wt_seq.data(), wt_seq.size());
    r.mutant_sequence.assign(mt_seq.data(), mt_seq.size());
}

// IDRecord::add_freq (reference: src/common.rs:528-568), in place
void record_add_freq(IDRecord& r, double freq) {
    const uint32_t nvar = r.nvar == 0 ? r.nvar : (freq > 0.0 ? r.nvar - 1 : r.nvar);
    r.nsomatic = nvar < r.nsomatic ? r.nsomatic - 1 : r.nsomatic;
    r.nvar = nvar;
    r.freq = r.freq > 0.5 ? r.freq : r.freq + freq;
}

// MP_TRACE=<file>: one line per print_haplotypes call / merge (debugging aid: the oracle writes the same trace)
static const char* trace_path() { static const char* const p = std::getenv("MP_TRACE"); return p; }

struct ConsumerHooks {
    static constexpr bool kNormal = false;
    const Batch& b;
    const HostResults& res;
    const GeneHost& gh;
    const Gene& gene;
    const Transcript& transcript;
    const TxDev& T;
    SomaticText& out;
    uint64_t window_len;
    size_t next_step = 0, cur_step = 0;
    bool is_fwd;
    // a deep gene is planned as 1 + n_extra copies with disjoint read subsets (batch.hpp GeneHost::n_extra): the same transcript of
    // copy k is T + tx_stride * k, its steps and windows mirror T's one to one
    uint32_t n_extra = 0, tx_stride = 0;

    void on_exon(const ExonGeom&) {}
    void begin_step() {}
    void routed(bool) {}

    void on_step(const ExonGeom&, const StepGeom& sg, const std::vector<size_t>&) {
        if (next_step >= T.n_steps) {
            const uint32_t ti = uint32_t(&T - b.tx.data());
            for (const auto& te : b.tx_errors)
                if (te.first == ti) throw Error(te.second);   // the schedule stopped here for this reason, and the real walk got here
            throw Error("internal error: consumer walked past the planned schedule");
        }
        cur_step = T.step_off + next_step++;
        const Step& st = b.steps[cur_step];
        if (st.sso != uint32_t(sg.sso) || uint64_t(st.wlen) != sg.splice_end - sg.sso)
            throw Error("internal error: consumer and planner schedules diverged");
    }

    std::pair<std::vector<HapSeq>, FsFreq> print(const ExonGeom& eg, const StepGeom& sg, uint64_t frame_in, FsFreq fsf,
                                                 bool is_first_exon_window, std::vector<HapSeq>&& recycled) {
        PROF(1);
        const Step& st = b.steps[cur_step];
        if (!(st.flags & SF_PRINT)) throw Error("internal error: print_haplotypes at a step the planner did not schedule");
        const WinStatic& ws = b.wins[st.win];
        WinDyn wd = res.win_dyn[st.win];
        if (!(wd.flags & WD_DONE)) throw Error("internal error: window was not computed on the device");
        // Deep gene: the groups of the window = the union of its copies' groups, counts and depth added up (rows are independent of each
        // other; every copy's list is ascending in (haplotype, frame), so is the merge). A merged entry keeps the slot of the first copy
        // that has the group: sequence, flags and record of a (window, haplotype) do not depend on the reads.
        struct MGroup { uint64_t hap; uint32_t aux, count; uint64_t slot; };
        static thread_local std::vector<MGroup> merged;
        if (n_extra) {
            merged.clear();
            for (uint32_t k = 0; k < wd.ngroups; k++) {
                const Group& G = res.grp(uint64_t(wd.group_off) + k);
                merged.push_back({G.hap, G.aux & ~GROUP_SETTLED, G.count, uint64_t(wd.group_off) + k});
            }
            for (uint32_t c = 1; c <= n_extra; c++) {
                const TxDev& Tc = (&T)[size_t(tx_stride) * c];
                const Step& sc = b.steps[Tc.step_off + (cur_step - T.step_off)];
                if (sc.win == 0xFFFFFFFFu) throw Error("internal error: a deep gene's copies diverged");
                const WinDyn& wc = res.win_dyn[sc.win];
                if (!(wc.flags & WD_DONE)) throw Error("internal error: window was not computed on the device");
                wd.nrows += wc.nrows;
                static thread_local std::vector<MGroup> next;
                next.clear();
                size_t i = 0;
                for (uint32_t k = 0; k < wc.ngroups; k++) {
                    const uint64_t slot = uint64_t(wc.group_off) + k;
                    const Group& G = res.grp(slot);
                    const uint32_t aux = G.aux & ~GROUP_SETTLED;
                    while (i < merged.size() && (merged[i].hap < G.hap || (merged[i].hap == G.hap && merged[i].aux < aux))) next.push_back(merged[i++]);
                    if (i < merged.size() && merged[i].hap == G.hap && merged[i].aux == aux) { MGroup m = merged[i++]; m.count += G.count; next.push_back(m); }
                    else next.push_back({G.hap, aux, G.count, slot});
                }
                while (i < merged.size()) next.push_back(merged[i++]);
                merged.swap(next);
            }
        }
        if (frame_in == 0) out.n_windows++;
        const std::vector<Variant>& gvars = gh.input->variants;
        const uint32_t ncols = ws.ncols;
        // window variants in ascending position (print_haplotypes order, :373-379); at most 63 columns (the planner enforces it)
        const Variant* variants[64];
        for (uint32_t j = 0; j < ncols && j < 64; j++) {
            uint32_t dq = is_fwd ? j : ncols - 1 - j;
            variants[j] = &gvars[b.win_cols[ws.col_off + dq].f];
        }
        // haplotype keys of this call (:383-411) from the device groups (ascending (hap, frame0, f1nz))
        struct Key { uint64_t hap, hframe; size_t count; uint64_t slot; };
        static thread_local std::vector<Key> keys;   // scratch: one allocation per consumer thread, not per window
        keys.clear();
        size_t frame_depth = 0;
        uint64_t frame = frame_in;
        uint64_t zero_slot = ~0ull;
        const uint32_t n_listed = n_extra ? uint32_t(merged.size()) : wd.ngroups;
        for (uint32_t k = 0; k < n_listed; k++) {
            uint64_t slot = n_extra ? merged[k].slot : uint64_t(wd.group_off) + k;
            const Group& G0 = res.grp(slot);
            const uint64_t g_hap = n_extra ? merged[k].hap : G0.hap;
            const uint32_t g_count = n_extra ? merged[k].count : G0.count, g_aux = n_extra ? merged[k].aux : (G0.aux & ~GROUP_SETTLED);
            if (g_hap == 0 && zero_slot == ~0ull) zero_slot = slot;
            if (g_count == 0) continue;
            uint64_t f0 = g_aux >> 1;
            bool f1nz = g_aux & 1;
            if (frame > 0 && f0 != frame && f1nz) continue;
            frame_depth += g_count;
            uint64_t kf = frame > 0 ? frame : f0;
            if (!keys.empty() && keys.back().hap == g_hap && keys.back().hframe == kf) keys.back().count += g_count;
            else keys.push_back({g_hap, kf, g_count, slot});
        }
        if (keys.empty()) {  // :429-431
            if (zero_slot == ~0ull) throw Error("internal error: reference haplotype missing from device results");
            keys.push_back({0, 0, 0, zero_slot});
        }
        if (const char* tr = trace_path()) {
            FILE* tf = std::fopen(tr, "a");
            std::fprintf(tf, "P %s %llu f%llu depth=%u fd=%zu first=%d :", transcript.id.c_str(), (unsigned long long)sg.sso, (unsigned long long)frame, wd.nrows, frame_depth, int(is_first_exon_window));
            for (const Key& k : keys) std::fprintf(tf, " (%llu,%llu)=%zu", (unsigned long long)k.hap, (unsigned long long)k.hframe, k.count);
            std::fprintf(tf, "\n");
            std::fclose(tf);
        }
        const char* strand = is_fwd ? "Forward" : "Reverse";
        static const std::string kForward("Forward"), kReverse("Reverse");
        const std::string& strand_s = is_fwd ? kForward : kReverse;
        const bool has_frameshift = frame > 0;
        const uint64_t offset = sg.sso, splice_pos = sg.splice_pos, splice_gap = sg.splice_gap;
        const uint64_t wl = eg.ewl;  // print_haplotypes' window_len parameter (:1421)
        const bool boundary = (ws.need_recs & WS_CARRY) != 0;  // haplotypes feed a splice-side merge
        std::vector<HapSeq> haplotypes_vec(std::move(recycled));   // (the buffer of the vector the last result replaced)
        haplotypes_vec.clear();
        haplotypes_vec.reserve(keys.size());
        uint64_t shift_in_window = 0;
        for (const Key& key : keys) {
            // a group the lane-per-window kernel settled itself (plan.hpp GROUP_SETTLED) has no GroupSum: it is {valid, no record}
            const GroupSum settled_gs{GS_VALID, 0};
            const GroupSum& gs = (res.grp(key.slot).aux & GROUP_SETTLED) ? settled_gs : res.gsm(key.slot);
            if (!(gs.flags & GS_VALID)) throw Error("internal error: haplotype was not processed by the window-sequence kernel");
            const HapRecHdr* rec = (gs.flags & GS_HAS_REC) ? res.rec(gs.rec) : nullptr;
            const uint8_t* rseq = rec ? res.rec_seq(gs.rec) : nullptr;
            const uint8_t* rgerm = rec ? res.rec_germ(gs.rec) : nullptr;
            const bool indel = gs.flags & GS_INDEL, insertion = gs.flags & GS_INSERTION, stop_gain = gs.flags & GS_STOP;
            const bool differs = gs.flags & GS_DIFFERS, broke = gs.flags & GS_BROKE;
            if ((ws.need_recs & WS_ALL_IDS) && !rec) throw Error("internal error: missing haplotype record for an indel window");
            const uint32_t n_somatic = rec ? rec->nsom : 0, n_variants = rec ? rec->nvar : 0;
            const double freq = key.count == 0 ? 0.0 : double(key.count) / double(frame_depth);
            bool shift_is_set = false;
            if (rec) {  // shift_in_window latch and frameshift events of the sequence walk (:479-502)
                uint32_t visited = rec->prof_len + (broke ? 1u : 0u);
                for (uint32_t j = 0; j < visited && j < ncols; j++) {
                    const Variant& v = *variants[j];
                    shift_in_window = shift_in_window > 0 ? shift_in_window : v.frameshift();
                    bool set = j < rec->prof_len ? ((rec->prof_set >> j) & 1) : true;
                    if (set && shift_in_window > 0) {
                        shift_is_set = true;
                        fsf[v.frameshift()] = {freq, !v.is_germline};
                        fsf[0] = {1.0 - freq, false};
                    }
                }
            }
            double frame_frequency = freq;  // :605-631
            if (shift_is_set && frame == 0) frame = shift_in_window;
            // (try_emplace: `emplace` builds its node before it looks the key up - a heap allocation per haplotype for an entry that nearly always exists)
            const std::pair<double, bool>& fs_entry = fsf.try_emplace(frame, 0.0, false).first->second;
            if (shift_in_window == 0) frame_frequency = freq * fs_entry.first;
            if (shift_in_window == 0 && key.hframe > 0 && frame == 0) frame_frequency = 0.0;
            const bool germ_cleared = (indel && insertion) || (shift_in_window == 0 && (fs_entry.second || (has_frameshift && differs)));
            const uint64_t seq_len = rec ? rec->seq_len : ws.wlen;
            const uint64_t germ_len = germ_cleared ? 0 : (rec ? rec->germ_len : ws.wlen);
            const bool germ_ne_seq = germ_cleared ? (seq_len != 0) : differs;
            const uint64_t this_window_len = seq_len < wl ? seq_len : wl;
            const uint64_t normal_window_len = indel ? (germ_len < wl ? germ_len : wl) : this_window_len;
            // peptides are only materialised when something can observe them
            const bool emit_pre = (n_somatic > 0 || has_frameshift) && !eg.is_short && germ_ne_seq && (!stop_gain || has_frameshift);
            std::string_view normal_peptide, neopeptide;   // slices of the device record
            const bool need_strings = rec && (indel || boundary || emit_pre);
            if ((indel || boundary || emit_pre) && !rec)
                throw Error("internal error: missing haplotype record (window sso " + std::to_string(ws.sso) + ", hap " + std::to_string(key.hap) +
                            ", need_recs " + std::to_string(ws.need_recs) + ", group flags " + std::to_string(gs.flags) + ", step flags " +
                            std::to_string(st.flags) + ", count " + std::to_string(key.count) + ")");
            if (need_strings) {
                auto sl = [](const uint8_t* p, uint64_t n, uint64_t a, uint64_t e) {
                    if (a > e || e > n) throw Error("reference would panic: slice index out of range");
                    return std::string_view(reinterpret_cast<const char*>(p) + a, size_t(e - a));
                };
                if (germ_len != 0) {
                    if (splice_pos == 1) normal_peptide = sl(rgerm, germ_len, splice_gap, germ_len);
                    else if (splice_pos == 0) normal_peptide = sl(rgerm, germ_len, 0, normal_window_len);
                    else normal_peptide = sl(rgerm, germ_len, 0, germ_len);
                }
                if (splice_pos == 1) neopeptide = sl(rseq, seq_len, splice_gap, seq_len);
                else if (splice_pos == 0) neopeptide = insertion ? sl(rseq, seq_len, 0, seq_len) : sl(rseq, seq_len, 0, this_window_len);
                else neopeptide = sl(rseq, seq_len, 0, seq_len);
            }
            bool remove_peptide = false;  // :702-718
            if (stop_gain && splice_pos != 2 && (wl == this_window_len || indel) && !is_first_exon_window &&
                (!indel || normal_peptide != neopeptide || std::fabs(freq - 1.0) < std::numeric_limits<double>::epsilon())) {
                remove_peptide = true;
                if (frame == 0) fsf[frame] = {0.0, false};
                else fsf.erase(frame);
            }
            const bool emit = emit_pre && frame_frequency > 0.0;
            HapSeq hs;
            hs.win = st.win; hs.frame = frame_in;
            // Which windows a splice-side merge will read is only known for certain while no shifted ORF exists: where one does (the planner
            // then asks for every haplotype's record anyway, WS_ALL_IDS) the real walk can keep an older window than the speculative schedule
            // marked - with window lengths that are not a multiple of 3 even beyond what the planner's candidate lists cover (fuzz seeds
            // 970028, 970791). So every window of such a stretch keeps its records.
            const bool keep_all = (ws.need_recs & WS_ALL_IDS) && rec;
            if (boundary || emit || keep_all) {
                PROF(3);
                // the row's list fields; the buffers live as long as the consumer thread (a row that is only written never allocates)
                static thread_local std::string sites, som_pos, som_pc, germ_pos, germ_pc, idstr;
                sites.clear(); som_pos.clear(); som_pc.clear(); germ_pos.clear(); germ_pc.clear();
                uint32_t n_sites = 0, n_som_sites = 0;
                auto add = [](std::string& s, uint64_t x) { if (!s.empty()) s.push_back('|'); append_u64(s, x); };
                bool f_sp = true, f_gp = true;
                for (uint32_t c = 0; c < ncols; c++) {  // :733-759
                    const Variant& v = *variants[c];
                    if (c < rec->prof_len && ((rec->prof_set >> c) & 1)) {
                        if (!v.is_germline) {
                            add(som_pos, v.pos + 1);
                            if (!f_sp) som_pc.push_back('|');
                            som_pc += v.prot_change; f_sp = false;
                        } else {
                            add(germ_pos, v.pos + 1);
                            if (!f_gp) germ_pc.push_back('|');
                            germ_pc += v.prot_change; f_gp = false;
                        }
                    }
                    if (c == 0 || v.pos != variants[c - 1]->pos) {
                        n_sites++;
                        add(sites, v.pos + 1);
                        if (!v.is_germline) n_som_sites++;
                    }
                }
                if (gs.flags & GS_ID_VALID) {
                    static const char HEX[] = "0123456789abcdef";
                    char idb[16];
                    for (int k = 0; k < 15; k++) idb[k] = HEX[(rec->id60 >> (4 * (14 - k))) & 0xF];
                    idb[15] = strand[0];
                    idstr.assign(idb, 16);
                } else {
                    haplotype_id_into(idstr, rseq, seq_len, transcript.id, offset, strand[0]);
                }
                const uint64_t roffset = splice_pos == 0 ? offset + 1 : offset + 1 + splice_gap;
                if (emit) {  // :839-875
                    if (splice_pos == 1) {
                        if (splice_gap > seq_len) throw Error("reference would panic: slice index out of range");
                        if (out.streams & STREAM_FASTA) put_fasta(out.fasta, idstr, rseq + splice_gap, seq_len - splice_gap);
                    } else if (splice_pos == 0) {
                        if (out.streams & STREAM_FASTA) put_fasta(out.fasta, idstr, rseq, this_window_len);
                    }
                    if (germ_len != 0) {
                        if (splice_pos == 1) {
                            if (splice_gap > germ_len) throw Error("reference would panic: slice index out of range");
                            if (out.streams & STREAM_NORMAL_FASTA) put_fasta(out.normal_fasta, idstr, rgerm + splice_gap, germ_len - splice_gap);
                        } else if (splice_pos == 0) {
                            if (this_window_len > germ_len) throw Error("reference would panic: slice index out of range");
                            if (out.streams & STREAM_NORMAL_FASTA) put_fasta(out.normal_fasta, idstr, rgerm, this_window_len);
                        }
                    }
                    // the row is written field by field (no copy through the record)
                    put_tsv_row(out, idstr, transcript.id, gene.id, gene.name, gene.chrom, roffset, frame, frame_frequency, wd.nrows, n_variants,
                                n_somatic, n_sites, n_som_sites, strand_s, sites, som_pos, som_pc, germ_pos, germ_pc, normal_peptide, neopeptide);
                }
                // The record is kept for every window that is marked as feeding a merge; an emitted window keeps it too where a merge can
                // reach a window the planner's marks miss: window lengths that are not a multiple of 3, or an indel / frameshift context
                // (keep_all; fuzz seeds 970028, 970791). An SNV-only window of a 3n-nt run that is emitted but not marked is never read
                // again - and were it, the merge fails loudly on the missing record (splice_merge) instead of using a wrong one.
                if (boundary || keep_all || window_len % 3 != 0) {
                    hs.filled = true;
                    IDRecord& r = hs.make().record;
                    r.id = idstr;
                    r.transcript = transcript.id; r.gene_id = gene.id; r.gene_name = gene.name; r.chrom = gene.chrom;
                    r.offset = roffset;
                    r.frame = frame;
                    r.freq = frame_frequency;
                    r.depth = wd.nrows;
                    r.nvar = n_variants; r.nsomatic = n_somatic;
                    r.nvariant_sites = n_sites; r.nsomvariant_sites = n_som_sites;
                    r.strand = strand_s;
                    r.variant_sites = sites; r.somatic_positions = som_pos; r.somatic_aa_change = som_pc;
                    r.germline_positions = germ_pos; r.germline_aa_change = germ_pc;
                    // the carried-over record holds the UNSLICED sequences (:807-832)
                    r.normal_sequence.assign(reinterpret_cast<const char*>(rgerm), germ_len);
                    r.mutant_sequence.assign(reinterpret_cast<const char*>(rseq), seq_len);
                }
            }
            if (!remove_peptide || frame == 0) haplotypes_vec.push_back(std::move(hs));  // :835-837
        }
        return {std::move(haplotypes_vec), std::move(fsf)};
    }

    // splice-side merge (reference: src/microphasing.rs:1505-1908)
    void splice_merge(const ExonGeom& eg, const StepGeom& sg, uint64_t exon_rest, std::map<uint64_t, uint64_t>& frameshifts,
                      FsFreq& fsf, std::vector<HapSeq>& hap_vec, std::vector<HapSeq>& prev_hap_vec) {
        PROF(2);
        const uint64_t offset = sg.offset;
        const std::vector<HapSeq>& first_hap_vec = is_fwd ? hap_vec : prev_hap_vec;
        const std::vector<HapSeq>& sec_hap_vec = is_fwd ? prev_hap_vec : hap_vec;
        for (const std::vector<HapSeq>* v : {&first_hap_vec, &sec_hap_vec})
            for (const HapSeq& h : *v)
                if (!h.filled) {
                    std::string where;
                    if (h.win != 0xFFFFFFFFu) {
                        const WinStatic& w = b.wins[h.win];
                        where = " (window at sso " + std::to_string(w.sso) + ", printed for frame " + std::to_string(h.frame) + ", need_recs " + std::to_string(w.need_recs) +
                                ", merge at sso " + std::to_string(sg.sso) + " of transcript " + transcript.id + (v == &hap_vec ? ", hap_vec" : ", prev_hap_vec") + ")";
                    }
                    throw Error("internal error: splice-side merge over a window whose records were not requested from the device" + where);
                }
        using MKey = std::tuple<uint64_t, std::string, std::string>;   // (offset, mutant window, wild-type window)
        std::map<MKey, IDRecord> output_map;
        std::vector<HapSeq> new_hap_vec;
        const double eps = std::numeric_limits<double>::epsilon();
        if (const char* tr = trace_path()) {
            FILE* tf = std::fopen(tr, "a");
            std::fprintf(tf, "M %s %llu\n", transcript.id.c_str(), (unsigned long long)offset);
            for (const auto& h : first_hap_vec) std::fprintf(tf, "  F %.17g %s %s\n", h.get().record.freq, h.get().record.mutant_sequence.c_str(), h.get().record.normal_sequence.c_str());
            for (const auto& h : sec_hap_vec) std::fprintf(tf, "  S %.17g %s %s\n", h.get().record.freq, h.get().record.mutant_sequence.c_str(), h.get().record.normal_sequence.c_str());
            std::fclose(tf);
        }
        // the frameshifts that reach this splice side (the same for every pair of haplotypes)
        static thread_local std::vector<std::pair<uint64_t, uint64_t>> active;
        active.clear();
        if (is_fwd) {
            for (auto it = frameshifts.begin(); it != frameshifts.end() && it->first < offset; ++it) active.push_back(*it);
        } else {
            for (auto it = frameshifts.lower_bound(offset + eg.ewl); it != frameshifts.end(); ++it) active.push_back(*it);
        }
        static thread_local std::string new_wt, new_mts[3];
        for (const HapSeq& hapseq : first_hap_vec) {
            const IDRecord& record = hapseq.get().record;
            const std::string& wt = record.normal_sequence;
            const std::string& mt = record.mutant_sequence;
            for (const HapSeq& prev_hapseq : sec_hap_vec) {
                const IDRecord& prev = prev_hapseq.get().record;
                const std::string& pwt = prev.normal_sequence;
                const std::string& pmt = prev.mutant_sequence;
                auto cat = [](std::string& dst, const std::string& a, const std::string& b) { dst.clear(); dst += a; dst += b; };
                cat(new_wt, pwt, wt);
                size_t n_mts = 0;
                if (wt != mt) {
                    cat(new_mts[n_mts++], pwt, mt);
                    if (pwt != pmt) { cat(new_mts[n_mts++], pmt, wt); cat(new_mts[n_mts++], pmt, mt); }
                } else {
                    cat(new_mts[n_mts++], pmt, mt);
                }
                const double merged = std::fabs(record.freq - prev.freq) < eps ? record.freq : record.freq * prev.freq;
                if (eg.is_short && !eg.is_last) {
                    HapSeq nh;
                    nh.filled = true;
                    record_update(nh.make().record, prev, record, 0, record.frame, merged, new_wt, new_wt, window_len, false);   // carried, never written
                    new_hap_vec.push_back(std::move(nh));
                }
                for (size_t m = 0; m < n_mts; m++) {
                    const std::string& new_mt = new_mts[m];
                    if (eg.is_short && !eg.is_last) {
                        HapSeq nh;
                        nh.filled = true;
                        record_update(nh.make().record, prev, record, 0, record.frame, merged, new_wt, new_mt, window_len, false);
                        new_hap_vec.push_back(std::move(nh));
                        continue;
                    }
                    for (const auto& pf : active) {
                        const uint64_t pos = pf.first, frameshift = pf.second;
                        fsf.emplace(frameshift, std::make_pair(0.0, false));
                        const bool shift_in_window = is_fwd ? pos >= prev.offset : pos < record.offset + eg.ewl;
                        const bool somatic_shift = fsf.at(frameshift).second;
                        const double fs_freq = fsf.at(frameshift).first;
                        const double f0 = fsf.at(0).first;
                        const double main_orf_freq = f0 == 0.0 ? fs_freq : f0;
                        const double shift_orf_freq = shift_in_window ? fs_freq : (f0 == 0.0 ? fs_freq : f0);
                        const double vf_rec = is_fwd ? record.freq / main_orf_freq : record.freq / shift_orf_freq;
                        const double vf_prev = is_fwd ? prev.freq / shift_orf_freq : prev.freq / main_orf_freq;
                        const double freq_rec = f0 == 0.0 ? fs_freq : vf_rec * fs_freq;
                        const double freq_prev = f0 == 0.0 ? fs_freq : vf_prev * fs_freq;
                        const double out_freq = std::fabs(record.freq - prev.freq) < eps ? freq_rec : freq_rec * freq_prev;
                        const uint64_t out_shift = shift_in_window ? 0 : frameshift;
                        uint64_t splice_offset = 3 - out_shift;
                        if (!is_fwd && exon_rest < 3) splice_offset += exon_rest;
                        size_t end_offset = 3 + size_t(out_shift);
                        if (sg.is_last_exon_window) end_offset = 0;
                        if (uint64_t(new_mt.size()) < 2 * window_len) {
                            if (is_fwd) splice_offset = 0; else end_offset = 0;
                        }
                        for (;;) {
                            if (end_offset > new_mt.size()) throw Error("reference would panic: attempt to subtract with overflow");
                            if (!(splice_offset + window_len <= uint64_t(new_mt.size() - end_offset))) break;
                            // the two windows are compared in place; strings are only built for a window that is kept (most are not:
                            // a merged window without a variant equals its wild type)
                            const size_t wl_ = size_t(window_len);
                            const char* wt_p = nullptr;
                            if (splice_offset + window_len <= uint64_t(new_wt.size())) {
                                if (is_fwd) wt_p = new_wt.data() + size_t(splice_offset);
                                else {
                                    if (new_wt.size() < end_offset + window_len) throw Error("reference would panic: attempt to subtract with overflow");
                                    wt_p = new_wt.data() + (new_wt.size() - end_offset - wl_);
                                }
                            }
                            const char* mt_p = is_fwd ? new_mt.data() + size_t(splice_offset) : new_mt.data() + (new_mt.size() - end_offset - wl_);
                            bool same = wt_p != nullptr && std::memcmp(wt_p, mt_p, wl_) == 0;
                            if (out_shift > 0 && same && somatic_shift) { wt_p = nullptr; same = false; }   // out_wt.clear()
                            if (same || (wt_p == nullptr && frameshift == 0)) {
                                if (is_fwd) splice_offset += 3; else end_offset += 3;
                                continue;
                            }
                            PROF(4);
                            const uint64_t out_offset = is_fwd ? splice_offset : uint64_t(end_offset);
                            // output_map[(offset, mt, wt)] = update(...).add_freq(frequency of the record it replaces)
                            const auto ins = output_map.try_emplace(MKey{out_offset, std::string(mt_p, wl_), wt_p ? std::string(wt_p, wl_) : std::string()});
                            IDRecord& out_record = ins.first->second;
                            const double old_freq = ins.second ? 0.0 : out_record.freq;
                            const std::string& out_mt = std::get<1>(ins.first->first);
                            const std::string& out_wt = std::get<2>(ins.first->first);
                            if (is_fwd) record_update(out_record, prev, record, out_offset, frameshift, out_freq, out_wt, out_mt, window_len, false);
                            else record_update(out_record, record, prev, out_offset, frameshift, out_freq, out_wt, out_mt, window_len, false);
                            record_add_freq(out_record, old_freq);
                            if (is_fwd) splice_offset += 3; else end_offset += 3;
                        }
                    }
                }
            }
        }
        if (eg.is_short && !eg.is_last) {
            prev_hap_vec = std::move(new_hap_vec);
        } else {
            PROF(5);
            for (auto& kv : output_map) {
                const std::string& out_mt = std::get<1>(kv.first);
                IDRecord& out_record = kv.second;
                const std::string& out_wt = std::get<2>(kv.first);
                if (out_mt != out_wt) {
                    haplotype_id_into(out_record.id, reinterpret_cast<const uint8_t*>(out_mt.data()), out_mt.size(), out_record.transcript,
                                      std::get<0>(kv.first), out_record.strand.empty() ? '?' : out_record.strand[0]);
                    if (out_mt.size() < window_len) throw Error("reference would panic: slice index out of range");
                    if (out.streams & STREAM_FASTA) put_fasta(out.fasta, out_record.id, reinterpret_cast<const uint8_t*>(out_mt.data()), size_t(window_len));
                    if (!out_wt.empty()) {
                        if (out_wt.size() < window_len) throw Error("reference would panic: slice index out of range");
                        if (out.streams & STREAM_NORMAL_FASTA) put_fasta(out.normal_fasta, out_record.id, reinterpret_cast<const uint8_t*>(out_wt.data()), size_t(window_len));
                    }
                    put_tsv_row(out, out_record);
                }
            }
            if (eg.is_short) prev_hap_vec = std::move(new_hap_vec);
        }
    }
};

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// `microphaser normal` (reference: src/normal_microphasing.rs). Same division of labour: the device supplies
// haplotype keys, counts, depth (K2) and sequence / variant profile / first-or-last-codon stop / id (K3 normal + K3b);
// the host keeps frameshifts, hap_vec / prev_hap_vec and the (sequence-concatenating) splice-side merge.
namespace {

// IDRecord::update / add_freq of the normal mode (reference: src/normal_microphasing.rs:105-146, :148-179), written into `r`
// (every field is set; r may hold an earlier record). with_id = false: the caller sets the id on the records it writes.
void nrecord_update(NormalRecord& r, const NormalRecord& self, const NormalRecord& rec, uint64_t offset, const uint8_t* seq, size_t seq_len,
                    bool with_id = true) {
    if (with_id) haplotype_id_into(r.id, seq, seq_len, self.transcript, offset, self.strand.empty() ? '?' : self.strand[0]);
    else r.id.clear();
    auto cat = [](std::string& dst, const std::string& x, const std::string& y) { dst.clear(); dst += x; dst += y; };
    r.transcript = self.transcript; r.gene_id = self.gene_id; r.gene_name = self.gene_name; r.chrom = self.chrom;
    r.frame = self.frame; r.depth = self.depth; r.strand = self.strand;
    cat(r.somatic_positions, self.somatic_positions, rec.somatic_positions);
    cat(r.somatic_aa_change, self.somatic_aa_change, rec.somatic_aa_change);
    cat(r.germline_positions, self.germline_positions, rec.germline_positions);
    cat(r.germline_aa_change, self.germline_aa_change, rec.germline_aa_change);
    r.offset = offset + self.offset;
    r.freq = self.freq * rec.freq;
    r.nvar = self.nvar + rec.nvar;
    r.nsomatic = self.nsomatic + rec.nsomatic;
    r.nvariant_sites = self.nvariant_sites + rec.nvariant_sites;
    r.nsomvariant_sites = self.nsomvariant_sites + rec.nsomvariant_sites;
    cat(r.variant_sites, self.variant_sites, rec.variant_sites);
    r.peptide_sequence.assign(reinterpret_cast<const char*>(seq), seq_len);
}
void nrecord_add_freq(NormalRecord& r, double freq) {
    const uint32_t nsomatic = r.nsomatic;
    if (freq > 0.0) r.nvar = r.nvar - 1u;  // u32 wrap-around at nvar == 0, as the reference's release build does (:150)
    if (r.nvar < nsomatic) r.nsomatic = nsomatic - 1;
    r.freq = r.freq + freq;
}

struct NormalConsumerHooks {
    static constexpr bool kNormal = true;
    const Batch& b;
    const HostResults& res;
    const GeneHost& gh;
    const Gene& gene;
    const Transcript& transcript;
    const TxDev& T;
    NormalText& out;
    uint64_t window_len;
    size_t next_step = 0, cur_step = 0;
    bool is_fwd;

    void on_exon(const ExonGeom&) {}
    void begin_step() {}
    void routed(bool) {}

    void on_step(const ExonGeom&, const StepGeom& sg, const std::vector<size_t>&) {
        if (next_step >= T.n_steps) {
            const uint32_t ti = uint32_t(&T - b.tx.data());
            for (const auto& te : b.tx_errors)
                if (te.first == ti) throw Error(te.second);   // the schedule stopped here for this reason, and the real walk got here
            throw Error("internal error: consumer walked past the planned schedule");
        }
        cur_step = T.step_off + next_step++;
        const Step& st = b.steps[cur_step];
        if (st.sso != uint32_t(sg.sso) || uint64_t(st.wlen) != sg.splice_end - sg.sso)
            throw Error("internal error: consumer and planner schedules diverged");
    }

    struct RowLists {
        std::string somatic_positions, somatic_aa_change, germline_positions, germline_aa_change, variant_sites;
        uint32_t n_sites = 0, n_som_sites = 0;
    };
    // the '|'-joined list fields of one haplotype (:531-557: 0-based positions; the profile index is the visit order)
    static void build_lists(const Variant* const* variants, uint32_t ncols, const HapRecHdr* rec, uint64_t prof_som, RowLists& L) {
        L.somatic_positions.clear(); L.somatic_aa_change.clear(); L.germline_positions.clear(); L.germline_aa_change.clear(); L.variant_sites.clear();
        L.n_sites = L.n_som_sites = 0;
        auto add = [](std::string& s, const std::string& x, bool& first) { if (!first) s.push_back('|'); s += x; first = false; };
        auto addn = [](std::string& s, uint64_t x, bool& first) { if (!first) s.push_back('|'); append_u64(s, x); first = false; };
        bool f1 = true, f2 = true, f3 = true, f4 = true, f5 = true;
        for (uint32_t c = 0; c < ncols; c++) {
            if (c >= rec->prof_len) break;
            const Variant& v = *variants[c];
            if ((rec->prof_set >> c) & 1) {
                if ((prof_som >> c) & 1) { addn(L.somatic_positions, v.pos, f1); add(L.somatic_aa_change, v.prot_change, f2); }
                else { addn(L.germline_positions, v.pos, f3); add(L.germline_aa_change, v.prot_change, f4); }
            }
            if (c == 0 || v.pos != variants[c - 1]->pos) {
                L.n_sites++;
                addn(L.variant_sites, v.pos, f5);
                if (!v.is_germline) L.n_som_sites++;
            }
        }
    }
    // the window's variants in print_haplotypes order (:373-379)
    const Variant** window_variants(const WinStatic& ws, const Variant** few, std::vector<const Variant*>& many) const {
        const std::vector<Variant>& gvars = gh.input->variants;
        const uint32_t ncols = ws.ncols;
        const Variant** variants = few;
        if (ncols > 64) { many.resize(ncols); variants = many.data(); }
        for (uint32_t j = 0; j < ncols; j++) variants[j] = &gvars[b.win_cols[ws.col_off + (is_fwd ? j : ncols - 1 - j)].f];
        return variants;
    }
    // The record print_haplotypes builds for a haplotype (:559-625), for the windows a splice-side merge reads: from the device record
    // and the values print() noted in the HapSeq.
    void materialize(HapSeq& h) const {
        if (h.np || h.lazy_rec == 0xFFFFFFFFu) return;
        const WinStatic& ws = b.wins[h.win];
        const Variant* few[64];
        std::vector<const Variant*> many;
        const Variant** variants = window_variants(ws, few, many);
        const HapRecHdr* rec = res.rec(h.lazy_rec);
        const uint8_t* rseq = res.rec_seq(h.lazy_rec);
        uint64_t prof_som;
        std::memcpy(&prof_som, res.rec_germ(h.lazy_rec), 8);
        static thread_local RowLists L;
        build_lists(variants, ws.ncols, rec, prof_som, L);
        HapSeq::NormalPayload& P = h.nmake();
        NormalRecord& r = P.nrecord;
        {
            static const char HEX[] = "0123456789abcdef";
            char idb[16];
            for (int k = 0; k < 15; k++) idb[k] = HEX[(rec->id60 >> (4 * (14 - k))) & 0xF];
            idb[15] = is_fwd ? 'F' : 'R';
            r.id.assign(idb, 16);
        }
        r.somatic_positions = L.somatic_positions; r.somatic_aa_change = L.somatic_aa_change;
        r.germline_positions = L.germline_positions; r.germline_aa_change = L.germline_aa_change;
        r.variant_sites = L.variant_sites;
        r.transcript = transcript.id; r.gene_id = gene.id; r.gene_name = gene.name; r.chrom = gene.chrom;
        r.offset = ws.sso; r.frame = h.frame; r.freq = h.lazy_freq; r.depth = h.lazy_nrows;
        r.nvar = rec->nvar; r.nsomatic = rec->nsom; r.nvariant_sites = L.n_sites; r.nsomvariant_sites = L.n_som_sites;
        r.strand = is_fwd ? "Forward" : "Reverse";
        P.sequence.assign(rseq, rseq + rec->seq_len);
        r.peptide_sequence.assign(reinterpret_cast<const char*>(rseq), rec->seq_len);  // carried record: unsliced (:618-625)
    }

    // print_haplotypes (reference: src/normal_microphasing.rs:341-647)
    std::pair<std::vector<HapSeq>, FsFreq> print(const ExonGeom& eg, const StepGeom& sg, uint64_t frame, FsFreq fsf, bool, std::vector<HapSeq>&& recycled) {
        const Step& st = b.steps[cur_step];
        if (!(st.flags & SF_PRINT)) throw Error("internal error: print_haplotypes at a step the planner did not schedule");
        const WinStatic& ws = b.wins[st.win];
        const WinDyn& wd = res.win_dyn[st.win];
        if (!(wd.flags & WD_DONE)) throw Error("internal error: window was not computed on the device");
        if (frame == 0) out.n_windows++;
        const uint32_t ncols = ws.ncols;
        const Variant* variants_few[64];
        std::vector<const Variant*> variants_many;
        const Variant** variants = window_variants(ws, variants_few, variants_many);
        const char* strand = is_fwd ? "Forward" : "Reverse";
        static const std::string kForward("Forward"), kReverse("Reverse");
        const std::string& strand_s = is_fwd ? kForward : kReverse;
        const uint64_t offset = sg.sso, splice_pos = sg.splice_pos, splice_gap = sg.splice_gap;
        const uint64_t wl = eg.ewl;
        const bool boundary = (ws.need_recs & WS_CARRY) != 0;
        const uint32_t nrows = wd.nrows;
        std::vector<HapSeq> haplotypes_vec(std::move(recycled));
        haplotypes_vec.clear();
        bool any_counted = false;
        for (uint32_t k = 0; k < wd.ngroups; k++) any_counted |= res.grp(uint64_t(wd.group_off) + k).count != 0;
        for (uint32_t k = 0; k < wd.ngroups; k++) {
            const uint64_t slot = uint64_t(wd.group_off) + k;
            const Group& G = res.grp(slot);
            // the zero-count reference group only stands in when no read covers the window (:388-390)
            if (G.count == 0 && (any_counted || G.hap != 0)) continue;
            const GroupSum& gs = res.gsm(slot);
            if (!(gs.flags & GS_VALID) || !(gs.flags & GS_HAS_REC))
                throw Error("internal error: haplotype was not processed by the window-sequence kernel");
            if (gs.flags & GS_BROKE) throw Error("internal error: haplotype sequence exceeds the record capacity");
            const bool stop_gain = gs.flags & GS_STOP, insertion = gs.flags & GS_INSERTION;
            if (stop_gain && splice_pos != 2) continue;  // :503-507
            const HapRecHdr* rec = res.rec(gs.rec);
            const uint8_t* rseq = res.rec_seq(gs.rec);
            uint64_t prof_som;
            std::memcpy(&prof_som, res.rec_germ(gs.rec), 8);
            const uint64_t seq_len = rec->seq_len;
            const double freq = double(G.count) / double(nrows);  // NaN when no read covers the window
            HapSeq hs;
            if (boundary || !eg.is_short) {
                const uint64_t this_window_len = seq_len < wl ? seq_len : wl;
                auto check = [&](uint64_t a, uint64_t e) { if (a > e || e > seq_len) throw Error("reference would panic: slice index out of range"); };
                uint64_t pep_lo, pep_hi;   // the peptide_sequence column: a slice of the haplotype sequence
                if (splice_pos == 1) { pep_lo = splice_gap; pep_hi = seq_len; }
                else if (splice_pos == 0) { pep_lo = 0; pep_hi = insertion ? seq_len : this_window_len; }
                else { pep_lo = 0; pep_hi = seq_len; }
                check(pep_lo, pep_hi);
                if (!(gs.flags & GS_ID_VALID)) throw Error("internal error: haplotype id was not computed on the device");
                // the row's id and list fields; the buffers live as long as the consumer thread. The lists are only built for a row that
                // is written to the TSV; the record a splice-side merge may ask for later is built then (materialize)
                static thread_local std::string idstr;
                static thread_local RowLists L;
                {
                    static const char HEX[] = "0123456789abcdef";
                    char idb[16];
                    for (int k = 0; k < 15; k++) idb[k] = HEX[(rec->id60 >> (4 * (14 - k))) & 0xF];
                    idb[15] = strand[0];
                    idstr.assign(idb, 16);
                }
                if (!eg.is_short) {  // :629-644
                    if (splice_pos == 1) {
                        if (splice_gap > seq_len) throw Error("reference would panic: slice index out of range");
                        if (out.streams & STREAM_FASTA) put_fasta(out.fasta, idstr, rseq + splice_gap, seq_len - splice_gap);
                    } else if (splice_pos == 0) {
                        if (wl > seq_len) throw Error("reference would panic: slice index out of range");
                        if (out.streams & STREAM_FASTA) put_fasta(out.fasta, idstr, rseq, size_t(wl));
                    }
                    if (out.streams & STREAM_TSV) {
                        build_lists(variants, ncols, rec, prof_som, L);
                        put_normal_tsv_row(out, idstr, transcript.id, gene.id, gene.name, gene.chrom, offset, frame, freq, nrows, rec->nvar, rec->nsom, L.n_sites,
                                           L.n_som_sites, strand_s, L.variant_sites, L.somatic_positions, L.somatic_aa_change, L.germline_positions,
                                           L.germline_aa_change, std::string_view(reinterpret_cast<const char*>(rseq) + pep_lo, size_t(pep_hi - pep_lo)));
                    }
                }
                // the record is kept for every window of a regular exon: `normal` merges reach further back than the planner's marks
                hs.filled = true;
                hs.win = st.win;
                hs.frame = frame;
                hs.lazy_rec = gs.rec; hs.lazy_nrows = nrows; hs.lazy_freq = freq;
            }
            haplotypes_vec.push_back(std::move(hs));
        }
        return {std::move(haplotypes_vec), std::move(fsf)};
    }

    // splice-side merge (reference: src/normal_microphasing.rs:1145-1250)
    void splice_merge(const ExonGeom& eg, const StepGeom& sg, uint64_t exon_rest, std::map<uint64_t, uint64_t>&, FsFreq&,
                      std::vector<HapSeq>& hap_vec, std::vector<HapSeq>& prev_hap_vec) {
        const std::vector<HapSeq>& first_hap_vec = is_fwd ? hap_vec : prev_hap_vec;
        const std::vector<HapSeq>& sec_hap_vec = is_fwd ? prev_hap_vec : hap_vec;
        for (const std::vector<HapSeq>* v : {&first_hap_vec, &sec_hap_vec})
            for (const HapSeq& h : *v)
                if (!h.filled) throw Error("internal error: splice-side merge over a window whose records were not requested from the device");
        for (HapSeq& h : hap_vec) materialize(h);        // the records of these two windows are needed now
        for (HapSeq& h : prev_hap_vec) materialize(h);
        using Bytes = std::vector<uint8_t>;
        std::map<std::pair<uint64_t, Bytes>, NormalRecord> output_map;
        std::vector<HapSeq> new_hap_vec;
        static thread_local Bytes joined;   // prev sequence + this sequence
        for (const HapSeq& hapseq : first_hap_vec) {
            const NormalRecord& record = hapseq.nget().nrecord;
            for (const HapSeq& prev_hapseq : sec_hap_vec) {
                const NormalRecord& prev_record = prev_hapseq.nget().nrecord;
                joined.assign(prev_hapseq.nget().sequence.begin(), prev_hapseq.nget().sequence.end());
                joined.insert(joined.end(), hapseq.nget().sequence.begin(), hapseq.nget().sequence.end());
                if (eg.is_short) {
                    HapSeq nh;
                    nh.nmake().sequence = joined;
                    nh.filled = true;
                    nrecord_update(nh.nmake().nrecord, prev_record, record, 0, joined.data(), joined.size(), false);   // carried, never written
                    new_hap_vec.push_back(std::move(nh));
                }
                uint64_t splice_offset = 3;
                if (!is_fwd && exon_rest < 3) splice_offset += exon_rest;
                size_t end_offset = 3;
                if (sg.is_last_exon_window) end_offset = 0;
                if (uint64_t(joined.size()) < 2 * window_len) {
                    if (is_fwd) splice_offset = 0; else end_offset = 0;
                }
                for (;;) {
                    if (end_offset > joined.size()) throw Error("reference would panic: attempt to subtract with overflow (merge)");
                    if (!(splice_offset + window_len <= uint64_t(joined.size() - end_offset))) break;
                    // output_map[(offset, window)] = update(...).add_freq(frequency of the record it replaces)
                    const auto ins = output_map.try_emplace(std::make_pair(splice_offset, Bytes(joined.begin() + long(splice_offset),
                                                                                                 joined.begin() + long(splice_offset + window_len))));
                    NormalRecord& out_record = ins.first->second;
                    const double old_freq = ins.second ? 0.0 : out_record.freq;
                    const Bytes& out_seq = ins.first->first.second;
                    nrecord_update(out_record, prev_record, record, splice_offset, out_seq.data(), out_seq.size(), false);
                    nrecord_add_freq(out_record, old_freq);
                    splice_offset += 3;
                }
            }
        }
        if (eg.is_short && !eg.is_last) {
            prev_hap_vec = std::move(new_hap_vec);
        } else {
            for (auto& kv : output_map) {
                const Bytes& out_seq = kv.first.second;
                NormalRecord& rec = kv.second;
                haplotype_id_into(rec.id, out_seq.data(), out_seq.size(), rec.transcript, kv.first.first, rec.strand.empty() ? '?' : rec.strand[0]);
                if (out_seq.size() < window_len) throw Error("reference would panic: slice index out of range");
                if (out.streams & STREAM_FASTA) put_fasta(out.fasta, kv.second.id, out_seq.data(), size_t(window_len));
                put_normal_tsv_row(out, kv.second);
            }
        }
    }
};

void reserve_streams(SomaticText& p, size_t recs) {
    if (p.streams & STREAM_TSV) p.tsv.reserve(recs * 340);
    if (p.streams & STREAM_FASTA) p.fasta.reserve(recs * 52);
    if (p.streams & STREAM_NORMAL_FASTA) p.normal_fasta.reserve(recs * 52);
    advise_huge(p.tsv.data(), p.tsv.capacity());
}
void reserve_streams(NormalText& p, size_t recs) {
    if (p.streams & STREAM_TSV) p.tsv.reserve(recs * 320);
    if (p.streams & STREAM_FASTA) p.fasta.reserve(recs * 56);
    advise_huge(p.tsv.data(), p.tsv.capacity()); advise_huge(p.fasta.data(), p.fasta.capacity());
}
const TextBuf* normal_stream(const SomaticText& p) { return &p.normal_fasta; }
const TextBuf* normal_stream(const NormalText&) { return nullptr; }

inline void set_copies(ConsumerHooks& h, const GeneHost& gh) { h.n_extra = gh.n_extra; h.tx_stride = gh.n_tx; }   // (copies are planned back to back)
inline void set_copies(NormalConsumerHooks&, const GeneHost&) {}                                                   // (`normal` genes are never split)

template <class Hooks, class Out>
void consume_range(const Batch& b, const HostResults& res, size_t g0, size_t g1, Out& out) {
    for (size_t g = g0; g < g1; g++) {
        const GeneHost& gh = b.genes[g];
        if (gh.is_extra) continue;   // a read-subset copy of the deep gene before it: consumed with that one
        const GeneInput& gi = *gh.input;
        VarIndex vi{&gi.variants};
        for (uint32_t k = 0; k < gh.n_tx; k++) {
            const TxDev& T = b.tx[gh.tx_off + k];
            const Transcript& t = gi.gene.transcripts[gh.tx_src[k]];
            PROF(0);
            Hooks hooks{b, res, gh, gi.gene, t, T, out, b.window_len, 0, 0, t.strand == FORWARD};
            set_copies(hooks, gh);
            walk_transcript(gi.gene, t, vi, gh.max_read_len, b.window_len, hooks);
        }
        const TextBuf* nf = normal_stream(out);
        out.gene_ends.push_back(GeneEnds{out.fasta.size(), nf ? nf->size() : 0, out.tsv.size()});
    }
}


struct CopyTask { char* dst; const char* src; size_t n; };

// Concatenate the per-range streams in gene order; the TSV header is kept from the first range that wrote a record. The copies
// (and the release of the pieces) run on all host threads.
template <class Out>
void assemble(std::vector<Out>& parts, PhasedStreams& out, size_t nthreads) {
    std::vector<CopyTask> tasks;
    constexpr size_t CHUNK = size_t(8) << 20;
    auto plan = [&](Bytes& dst, const std::vector<std::pair<const char*, size_t>>& pieces) {
        size_t total = 0;
        for (const auto& pc : pieces) total += pc.second;
        dst.n = total;
        dst.p.reset(total ? new char[total] : nullptr);
        if (total) advise_huge(dst.p.get(), total);
        size_t at = 0;
        for (const auto& pc : pieces) {
            for (size_t o = 0; o < pc.second; o += CHUNK) tasks.push_back({dst.p.get() + at + o, pc.first + o, std::min(CHUNK, pc.second - o)});
            at += pc.second;
        }
    };
    std::vector<std::pair<const char*, size_t>> fa, nfa, tsv;
    bool header = false;
    uint64_t fa_at = 0, nfa_at = 0, tsv_at = 0, header_len = 0;
    for (int k = 0; k < 3; k++) out.gene_off[k].assign(1, 0);
    for (Out& p : parts) {
        out.n_windows += p.n_windows;
        fa.push_back({p.fasta.data(), p.fasta.size()});
        const TextBuf* n = normal_stream(p);
        if (n) nfa.push_back({n->data(), n->size()});
        size_t skip = 0;
        if (!p.tsv.empty()) {
            const size_t hl = size_t(static_cast<const char*>(std::memchr(p.tsv.data(), '\n', p.tsv.size())) - p.tsv.data()) + 1;
            skip = header ? hl : 0;
            if (!header) header_len = hl;
            tsv.push_back({p.tsv.data() + skip, p.tsv.size() - skip});
            header = true;
        }
        for (const GeneEnds& ge : p.gene_ends) {   // this range's genes, in the merged streams
            out.gene_off[0].push_back(fa_at + ge.fasta);
            out.gene_off[1].push_back(nfa_at + ge.normal_fasta);
            out.gene_off[2].push_back(tsv_at + (ge.tsv > skip ? ge.tsv - skip : 0));
        }
        fa_at += p.fasta.size();
        if (n) nfa_at += n->size();
        tsv_at += p.tsv.size() - skip;
    }
    for (uint64_t& o : out.gene_off[2]) o = std::max(o, header_len);   // the header line belongs to no gene
    plan(out.fasta, fa);
    plan(out.normal_fasta, nfa);
    plan(out.tsv, tsv);
    std::atomic<size_t> next{0}, next_free{0};
    auto work = [&] {
        for (size_t i; (i = next.fetch_add(1)) < tasks.size();) std::memcpy(tasks[i].dst, tasks[i].src, tasks[i].n);
    };
    const auto t_copy = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    const size_t nt = std::max<size_t>(1, std::min(nthreads, tasks.size()));
    for (size_t k = 1; k < nt; k++) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
    th.clear();
    const auto t_release = std::chrono::steady_clock::now();
    (void)next_free;
    release_later(std::move(parts));
    parts.clear();
    if (std::getenv("MP_DEBUG"))
        std::fprintf(stderr, "[mp]   assemble: %zu copy tasks %.1f ms, release of the pieces %.1f ms\n", tasks.size(),
                     std::chrono::duration<double, std::milli>(t_release - t_copy).count(),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_release).count());
}

template <class Hooks, class Out>
void consume_sharded(const Batch& b, const HostResults& res, PhasedStreams& out, uint32_t streams) {
    size_t nthreads = host_threads();
    const size_t ng = b.genes.size();
    if (nthreads > ng) nthreads = ng ? ng : 1;
    if (nthreads <= 1) {
        std::vector<Out> one(1);
        one[0].streams = streams;
        consume_range<Hooks>(b, res, 0, ng, one[0]);
        assemble(one, out, 1);
        return;
    }
    // balance by planned steps
    std::vector<uint64_t> cost(ng + 1, 0);
    for (size_t g = 0; g < ng; g++) {
        uint64_t c = 0;
        for (uint32_t k = 0; k < b.genes[g].n_tx; k++) c += b.tx[b.genes[g].tx_off + k].n_steps;
        cost[g + 1] = cost[g] + c + 1;
    }
    std::vector<size_t> cut(nthreads + 1, ng);
    cut[0] = 0;
    for (size_t t = 1; t < nthreads; t++) {
        uint64_t target = cost[ng] * t / nthreads;
        cut[t] = size_t(std::lower_bound(cost.begin(), cost.end(), target) - cost.begin());
        if (cut[t] > ng) cut[t] = ng;
        if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<Out> parts(nthreads);
    for (Out& p : parts) p.streams = streams;
    std::vector<std::string> errors(nthreads);
    std::vector<std::thread> th;
    // Reserve each range's streams up front (an estimate from the records the device produced, shared out by planned steps; pages
    // that are never written are never touched): a growing std::string re-maps its buffer again and again, and every re-map takes
    // the process-wide mm lock that all other threads' page faults wait on.
    const double per_step = cost[ng] ? double(res.n_recs) / double(cost[ng]) : 0.0;
    for (size_t t = 0; t < nthreads; t++)
        th.emplace_back([&, t] {
            try {
                const double recs = per_step * double(cost[cut[t + 1]] - cost[cut[t]]) * 1.25 + 1024;
                reserve_streams(parts[t], size_t(recs));
                consume_range<Hooks>(b, res, cut[t], cut[t + 1], parts[t]);
            }
            catch (const std::exception& e) { errors[t] = e.what(); if (errors[t].empty()) errors[t] = "error"; }
        });
    for (auto& x : th) x.join();
    for (size_t t = 0; t < nthreads; t++)
        if (!errors[t].empty()) throw Error(errors[t]);  // the first failing gene range in gene order, like a sequential run
    const auto t1 = std::chrono::steady_clock::now();
    assemble(parts, out, nthreads);
    if (std::getenv("MP_DEBUG"))
        std::fprintf(stderr, "[mp]   consume on %zu threads %.1f ms, concatenate %.1f ms\n", nthreads, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
}

}  // namespace

void consume_batch(const Batch& b, const HostResults& res, PhasedStreams& out, uint32_t streams) {
    if (b.normal) throw Error("internal error: somatic consumer on a normal-mode batch");
    consume_sharded<ConsumerHooks, SomaticText>(b, res, out, streams);
}

void consume_batch_normal(const Batch& b, const HostResults& res, PhasedStreams& out, uint32_t streams) {
    if (!b.normal) throw Error("internal error: normal consumer on a somatic-mode batch");
    consume_sharded<NormalConsumerHooks, NormalText>(b, res, out, streams);
}

}  // namespace mp
