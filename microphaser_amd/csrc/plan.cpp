// Planner: packs loaded genes into the device batch and records the static window schedule.
// reference: the data-independent part of phase_gene (src/microphasing.rs:905-942 loading,
// :944-1342 scheduler) - see walk.hpp for the shared control flow.
#include <algorithm>
#include <array>
#include <atomic>
#include <functional>
#include <chrono>
#include <cstdio>
#include <deque>
#include <map>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>
#include <type_traits>

#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include "batch.hpp"

namespace mp {

namespace {

template <bool NORMAL>
struct PlannerHooksT {
    static constexpr bool kNormal = NORMAL;
    Batch& b;
    const GeneHost& gh;
    const std::vector<Variant>& vars;
    const std::vector<uint32_t>& fwd2rev;  // gene-relative
    bool is_fwd;
    uint32_t tx_idx;
    uint32_t cur_exon = 0;
    std::deque<uint32_t> cols;  // forward variant indices (gene-relative) of the live columns, oldest first
    bool cols_dirty = true;
    uint32_t cols_off = 0;
    uint32_t col_hi = 0;
    bool fs_seen = false;
    uint64_t prev_cand_lo = 0;
    bool have_prev_cand = false;
    size_t cur_step = 0;
    uint64_t max_live = 0;
    uint64_t max_seq_len = 0;
    // windows whose haplotypes currently sit in prev_hap_vec / hap_vec (they feed the next splice-side merge)
    // Windows whose haplotypes may be in prev_hap_vec / hap_vec when a splice-side merge runs. The schedule is speculative: a shifted
    // ORF that really stops (data dependent) no longer prints, so the vector then still holds an EARLIER window's haplotypes. A main-ORF
    // print always happens while the walk is alive and replaces the candidates; a shifted-ORF print only adds one.
    uint32_t last_print_win = 0xFFFFFFFFu;
    uint64_t last_print_frame = 0;
    std::vector<uint32_t> held_prev, held_hap;
    // segments: where the current one starts, the extreme candidate keys seen so far, prefix maximum of the read ends
    const std::vector<uint32_t>* pmax_end = nullptr;
    uint32_t seg_start = 0xFFFFFFFFu;
    uint64_t seg_last_sso = 0, seg_low_key = ~0ull;
    bool seg_any = false;
    Batch::SegInfo seg_info;       // eligibility inputs of the current segment (window-parallel replay)
    uint32_t seg_prev_f = 0, seg_init_cols = 0;

    // window-parallel replay wants SNV columns at consecutive forward indices (a linear tr -> f map over the segment)
    void note_column(uint32_t f) {
        if (!seg_info.have_col) { seg_info.have_col = true; seg_info.tr0 = tr_index(f); seg_info.f0 = f; }
        else if (f != (is_fwd ? seg_prev_f + 1 : seg_prev_f - 1)) seg_info.cols_ok = false;
        seg_prev_f = f;
        if (vars[f].kind != VK_SNV) seg_info.cols_ok = false;
    }
    // normal mode: list-once bookkeeping and the geometry of the previous step (epoch breaks)
    bool have_listed = false;
    uint32_t listed_lo = 0, listed_hi = 0;
    uint64_t n_steps_tx = 0, last_sso = 0, last_end = 0, last_wlen = 0, last_range = 0, last_added = 0;

    uint32_t tr_index(size_t fwd_idx) const { return is_fwd ? uint32_t(fwd_idx) : fwd2rev[fwd_idx]; }

    // The schedule is speculative: it keeps walking where the real run may already have stopped (a stop codon ends a transcript, a
    // shifted ORF dies). A condition under which the reference would panic - or a limit of this build - met out there must not fail
    // the batch: the plan of the transcript is cut back to its last complete step and the message kept; the consumer raises it only
    // if the real walk gets that far (consume.cpp on_step).
    struct Mark { size_t steps, aux, ncols, rlo, rn, wins, win_cols; uint64_t n_main; } mark{};
    void take_mark() { mark = Mark{b.steps.size(), b.step_aux.size(), b.step_ncols.size(), b.step_rlo.size(), b.step_rn.size(), b.wins.size(), b.win_cols.size(), b.n_main_windows}; }
    void begin_step() { take_mark(); }
    void rollback() {
        b.steps.resize(mark.steps); b.step_aux.resize(mark.aux); b.step_ncols.resize(mark.ncols); b.step_rlo.resize(mark.rlo); b.step_rn.resize(mark.rn);
        b.wins.resize(mark.wins); b.win_cols.resize(mark.win_cols); b.n_main_windows = mark.n_main;
        if (seg_start != 0xFFFFFFFFu && seg_start > b.steps.size()) seg_start = uint32_t(b.steps.size());
    }

    void on_exon(const ExonGeom& eg) {
        take_mark();
        ExonPlan ep;
        ep.geom = eg;
        ep.tx = tx_idx;
        cur_exon = uint32_t(b.exons.size());
        b.exons.push_back(ep);
        have_prev_cand = false;
        seg_info.n_exons++;
    }

    // gene-relative index of the first kept read with pos >= key
    HintedLower read_hint;   // (the queries of consecutive steps are one nt apart)
    uint32_t read_lower(uint64_t key) const {
        const uint32_t* p = b.r_pos.data() + gh.read_off;
        return uint32_t(read_hint.lower(key, gh.n_reads, [p](size_t i) { return uint64_t(p[i]); }));
    }

    void ensure_window(const ExonGeom& eg, const StepGeom& sg) {
        Step& st = b.steps[cur_step];
        if (st.flags & SF_PRINT) return;
        st.flags |= SF_PRINT;
        st.win = uint32_t(b.wins.size());
        WinStatic w{};
        w.tx = tx_idx;
        w.sso = st.sso;
        w.ncols = uint16_t(cols.size());
        if (cols_dirty) {  // consecutive windows with unchanged columns share one list
            cols_off = uint32_t(b.win_cols.size());
            for (uint32_t c : cols) b.win_cols.push_back(WinCol{c, b.v_pos[gh.var_off + c], b.v_info[gh.var_off + c]});
            cols_dirty = false;
        }
        w.col_off = cols_off;
        w.ref_off = uint32_t(gh.ref_off + (st.sso - uint32_t(gh.input->gene.start())));
        w.vbase = gh.var_off;
        w.step = uint32_t(cur_step);
        w.wlen = st.wlen;
        w.ewl = uint8_t(eg.ewl);
        w.splice_pos = uint8_t(sg.splice_pos);
        w.splice_gap = uint8_t(sg.splice_gap);
        w.flags = uint8_t((st.flags & 0x7C) | (is_fwd ? 0 : WSF_REVERSE));
        uint32_t walk_prefix = 0;
        {   // WSF_SIMPLE: see plan.hpp. Walk order = ascending position = deque order on '+', reversed deque on '-'.
            static const bool general_walk_only = std::getenv("MP_GENERAL_WALK") != nullptr;   // testing: K3's general walk everywhere
            bool simple = !NORMAL && st.wlen <= 32 && !general_walk_only;
            auto col = [&](size_t k) -> const Variant& { return vars[is_fwd ? cols[k] : cols[cols.size() - 1 - k]]; };
            uint64_t prev = 0;
            size_t e = 0;
            bool tail = false;   // a run of columns right behind the window that a set SNV at the window's last base pulls in (below)
            for (; e < cols.size(); e++) {
                const Variant& v = col(e);
                if (v.kind != VK_SNV || v.pos < st.sso || v.pos >= uint64_t(st.sso) + st.wlen || (e > 0 && v.pos <= prev)) break;
                prev = v.pos;
            }
            if (simple && e < cols.size()) {   // the first column outside the prefix must be unreachable for good
                const Variant& v = col(e);
                const bool beyond = v.pos >= uint64_t(st.sso) + st.wlen;
                const bool behind = e == 0 ? v.pos < st.sso : v.pos < prev;   // == prev (second ALT of a site) depends on the haplotype
                if (!beyond && !behind) simple = false;
                // the inner loop of the walk (`while j < ncols && i == pos_j`, :479) has no window bound: an applied SNV at the
                // window's last base moves the cursor to window_end, where a column sitting exactly there is applied too
                // - still byte substitution when the columns from there on are SNVs at consecutive positions (one more base per applied column,
                // the run ends at its first column the haplotype does not set) and what follows the run lies strictly beyond it; the window
                // may then hold more than wlen bases, so it is never WSF_NOSTOP (K3's list C: substitution + codon scan). At a variant every
                // 1.35 nt (config D) most windows are of this kind; they used to take the general per-base walk.
                if (beyond && e > 0 && v.pos == uint64_t(st.sso) + st.wlen && prev + 1 == v.pos) {
                    size_t t = e;
                    uint64_t expect = uint64_t(st.sso) + st.wlen;
                    while (t < cols.size() && col(t).kind == VK_SNV && col(t).pos == expect) { t++; expect++; }
                    if (t < cols.size() && col(t).pos <= expect) simple = false;   // an indel at the run's end, or a second ALT of one of its sites: the general walk
                    else tail = true;
                }
            }
            walk_prefix = uint32_t(e);
            if (!simple && !NORMAL && std::getenv("MP_DEBUG_SIMPLE")) {
                std::string m;
                for (size_t k = 0; k < cols.size(); k++) { const Variant& v = vars[is_fwd ? cols[k] : cols[cols.size() - 1 - k]]; m += " " + std::to_string(v.pos) + (v.kind != VK_SNV ? "*" : ""); }
                std::fprintf(stderr, "nonsimple %s sso %u wlen %u cols:%s\n", is_fwd ? "+" : "-", st.sso, unsigned(st.wlen), m.c_str());
            }
            if (simple) w.flags |= WSF_SIMPLE;
            if (simple && !tail) {   // WSF_NOSTOP: has_stop_codon (:42-76) on the reference slice, which no SNV haplotype can extend
                const uint8_t* rw = gh.input->refseq.data() + (st.sso - gh.input->gene.start());
                bool upper = true;
                for (uint32_t k = 0; k < st.wlen; k++) upper &= rw[k] >= 'A' && rw[k] <= 'Z';
                const uint32_t seq_len = st.wlen, this_len = std::min<uint32_t>(seq_len, uint32_t(eg.ewl));
                uint32_t nlo = 0, nhi = seq_len;
                if (sg.splice_pos == 1) nlo = std::min<uint32_t>(uint32_t(sg.splice_gap), seq_len);
                else if (sg.splice_pos == 0) nhi = this_len;
                bool ref_stop = false;
                auto is_stop = [&](uint32_t c) {
                    const uint8_t a = rw[nlo + c], b2 = rw[nlo + c + 1], e = rw[nlo + c + 2];
                    if (is_fwd) return a == 'T' && ((b2 == 'G' && e == 'A') || (b2 == 'A' && (e == 'G' || e == 'A')));
                    return (a == 'T' && b2 == 'C' && e == 'A') || (a == 'C' && b2 == 'T' && e == 'A') || (a == 'T' && b2 == 'T' && e == 'A');
                };
                const uint32_t nlen = nhi - nlo;
                if (nlen >= 3) {
                    if (is_fwd) { for (uint32_t c = 0; c + 3 <= nlen; c += 3) ref_stop |= is_stop(c); }
                    else { for (int c = int(nlen) - 3; c >= 0; c -= 3) ref_stop |= is_stop(uint32_t(c)); }
                }
                if (upper && !ref_stop) w.flags |= WSF_NOSTOP;
            }
        }
        b.wins.push_back(w);
        // upper bound of the sequence lengths print_haplotypes can build for this window: the window itself, plus the
        // inserted bases, plus the reference bases a somatic deletion restores in the germline sequence (:547-577)
        uint64_t max_len = st.wlen;
        bool non_snv = false;
        for (uint32_t c : cols) {
            const Variant& v = vars[c];
            if (v.kind == VK_INS) max_len += v.len;
            if (v.kind == VK_DEL) max_len += v.len;
            if (v.kind != VK_SNV) non_snv = true;
        }
        {   // the inner loop of the walk has no window bound: a run of ADJACENT columns starting at the window's last base is applied past
            // window_end, one more base each (stale columns of '-' exons and `normal` epochs sit there) - so only the columns at
            // window_end, window_end + 1, ... without a gap can add a base (K3 flags a record that outgrows its capacity all the same)
            const uint64_t wend = uint64_t(st.sso) + st.wlen;
            bool at_last_base = false;
            for (uint32_t c : cols) at_last_base |= vars[c].pos + 1 == wend;
            if (at_last_base) {   // a set SNV at the last base moves the cursor to window_end: the run there, then one reference base
                max_len += 1;
                for (uint64_t p = wend;; p++) {
                    uint32_t here = 0;
                    for (uint32_t c : cols) here += vars[c].pos == p;
                    if (!here) break;
                    max_len += here;
                }
            }
        }
        b.wins.back().need_recs = uint8_t(((NORMAL || non_snv || fs_seen) ? WS_ALL_IDS : 0) | (walk_prefix << WS_PREFIX_SHIFT));  // `normal` emits every haplotype
        if (NORMAL) max_len += 1;  // the unconditional trailing base (src/normal_microphasing.rs:476)
        if (b.wins.back().need_recs & WS_MASK) b.steps[cur_step].flags |= SF_NEED_RECS;
        if (max_len > SEQ_CAP_MAX)
            throw Error("window at " + std::to_string(st.sso) + " can build a sequence of " + std::to_string(max_len) +
                        " nt; this build supports at most " + std::to_string(SEQ_CAP_MAX) + " (very long indel in a window)");
        max_seq_len = std::max<uint64_t>(max_seq_len, max_len);
    }

    void on_step(const ExonGeom& eg, const StepGeom& sg, const std::vector<size_t>& new_cols) {
        Step st{};
        if (sg.sso > 0xFFFFFFFFull) throw Error("coordinate exceeds 32 bits");
        st.sso = uint32_t(sg.sso);
        uint64_t wlen = sg.splice_end - sg.sso;
        if (wlen > 255) throw Error("window longer than 255 nt is not supported");
        st.wlen = uint8_t(wlen);
        if (sg.deleted > cols.size())
            throw Error("reference would panic: drain range out of bounds (shrink_left) at sso " + std::to_string(sg.sso) + " end " +
                        std::to_string(sg.splice_end) + " ncols " + std::to_string(cols.size()) + " deleted " + std::to_string(sg.deleted) +
                        " added " + std::to_string(sg.added) + " nvars " + std::to_string(sg.nvars) + " first " + std::to_string(sg.is_first_exon_window) +
                        " last " + std::to_string(sg.is_last_exon_window) + " short " + std::to_string(eg.is_short) + " exon " +
                        std::to_string(eg.start) + "-" + std::to_string(eg.end) + " ceo " + std::to_string(eg.ceo) + " ewl " + std::to_string(eg.ewl) + (is_fwd ? " fwd" : " rev"));
        if (sg.deleted > 255 || new_cols.size() > 255) throw Error("more than 255 column changes in one step");
        st.n_del = uint8_t(sg.deleted);
        st.n_add = uint8_t(new_cols.size());
        // ---- segment break: nothing of the matrix may survive into this step (rows, pending candidates, columns)
        {
            const uint32_t here = uint32_t(b.steps.size());
            if (seg_start == 0xFFFFFFFFu) seg_start = here;
            else if (sg.is_first_exon_window && seg_any) {
                bool rows_may_survive;
                if (is_fwd) {   // a read listed so far (start <= last sso) that still encloses: end >= splice_end
                    const uint32_t n = read_lower(seg_last_sso + 1);
                    rows_may_survive = n > 0 && uint64_t((*pmax_end)[gh.read_off + n - 1]) >= sg.splice_end;
                } else {        // a read listed so far (start >= lowest key) that is not cleaned up: start <= sso
                    rows_may_survive = seg_low_key <= sg.sso && read_lower(seg_low_key) < read_lower(sg.sso + 1);
                }
                if (std::getenv("MP_DEBUG_SEG")) std::fprintf(stderr, "seg? tx %u %s sso %llu end %llu cols %zu del %zu last_sso %llu low_key %llu survive %d\n", tx_idx, is_fwd ? "+" : "-", (unsigned long long)sg.sso, (unsigned long long)sg.splice_end, cols.size(), sg.deleted, (unsigned long long)seg_last_sso, (unsigned long long)seg_low_key, int(rows_may_survive));
                // A segment that starts here is given its initial deque as "the init_cols columns right before this step's first new
                // one" (SegDev::init_cols). The deque is a run of the append sequence, but that sequence can skip variants (no window
                // of the previous exon reached them), so the run is not always consecutive in transcription order: then the wave
                // has to carry its real columns across this exon start - no break.
                bool init_contiguous = true;
                {
                    const uint32_t x = new_cols.empty() ? col_hi : tr_index(new_cols[0]);
                    size_t k = 0;
                    for (uint32_t c : cols) { if (uint64_t(tr_index(c)) + cols.size() != uint64_t(x) + k) init_contiguous = false; k++; }
                }
                if (!rows_may_survive && !init_contiguous && std::getenv("MP_DEBUG_SEG")) std::fprintf(stderr, "seg: tx %u sso %llu keeps its segment: live columns are not consecutive\n", tx_idx, (unsigned long long)sg.sso);
                if (!rows_may_survive && init_contiguous) {
                    seg_info.n_exons--;   // this exon's on_exon() already counted itself on the segment being closed
                    b.segs.push_back(SegDev{tx_idx, seg_start, here - seg_start, seg_init_cols});
                    b.seg_info.push_back(seg_info);
                    seg_info = Batch::SegInfo();
                    seg_info.n_exons = 1;
                    seg_start = here;
                    seg_low_key = ~0ull;
                    seg_init_cols = uint32_t(cols.size());   // columns alive before this step's shrink_left
                    for (uint32_t c : cols) note_column(c);  // they belong to the new segment's column range
                }
            }
            seg_any = true;
            seg_last_sso = is_fwd ? std::max(seg_last_sso, sg.sso) : sg.sso;
            seg_low_key = std::min(seg_low_key, sg.cand_lo);
        }
        for (size_t k = 0; k < sg.deleted; k++) cols.pop_front();
        if (sg.deleted || !new_cols.empty()) cols_dirty = true;
        for (size_t k : new_cols) note_column(uint32_t(k));
        for (size_t k : new_cols) {
            // NOTE: the live columns are not always the window's own variants - the reference can leave a
            // stale column behind (deleted_vars forced to 0 when offset == old_offset, :1159) - so every
            // printing window carries an explicit column list (win_cols).
            cols.push_back(uint32_t(k));
            col_hi = tr_index(k) + 1;
            if (vars[k].frameshift() > 0) fs_seen = true;
        }
        if (cols.size() > 63) throw Error("more than 63 variant columns in one window (reference overflows its u64 haplotype word)");
        st.col_hi = col_hi;
        st.win = 0xFFFFFFFFu;
        st.splice_pos = uint8_t(sg.splice_pos);
        st.splice_gap = uint8_t(sg.splice_gap);
        st.exon = cur_exon;
        uint8_t fl = 0;
        if (sg.is_first_exon_window) fl |= SF_FIRST_EXON_WIN;
        if (sg.is_last_exon_window) fl |= SF_LAST_EXON_WIN;
        if (eg.is_short) fl |= SF_SHORT_EXON;
        if (eg.is_first) fl |= SF_FIRST_EXON;
        if (eg.is_last) fl |= SF_LAST_EXON;
        // candidate reads: forward = every key is tried exactly once (:1229-1248); reverse = the whole
        // range is re-scanned every step (:1198-1226) - only the keys that newly entered the range are
        // listed, the kernel keeps the not-yet-admitted ones pending.
        uint64_t lo = sg.cand_lo, hi = sg.cand_hi;
        if (sg.is_first_exon_window) {
            fl |= SF_FULL_RANGE;
        } else if (!NORMAL && !is_fwd && have_prev_cand) {
            hi = std::min(hi, prev_cand_lo);
            if (hi < lo) hi = lo;
        }
        prev_cand_lo = sg.cand_lo;
        have_prev_cand = true;
        uint32_t c0 = read_lower(lo), c1 = read_lower(hi);
        if (NORMAL) {
            // `normal` has no `contains`: a read is pushed again at EVERY step whose key range and window it satisfies
            // (src/normal_microphasing.rs:942-967, 1010-1017). The kernel keeps ONE slot per read and derives the number
            // of copies per column epoch in closed form (k2n_window_replay), so a read is listed exactly once per
            // transcript: when its key first enters a candidate range. That needs sso / splice_end to move one way only.
            if (n_steps_tx) {
                bool ok = is_fwd ? (sg.sso >= last_sso && sg.splice_end >= last_end) : (sg.sso <= last_sso);
                if (!ok) throw Error("normal mode: exons of a transcript overlap or are not in transcription order (not supported by this build)");
            }
            if (!have_listed) { listed_lo = c1; listed_hi = c0; have_listed = true; }  // nothing listed yet on either side
            if (is_fwd) { c0 = std::max(c0, listed_hi); c1 = std::max(c1, c0); listed_hi = std::max(listed_hi, c1); }
            else { c1 = std::min(c1, listed_lo); c0 = std::min(c0, c1); listed_lo = std::min(listed_lo, c0); }
            const uint64_t range = sg.sso - sg.cand_lo;   // key range extent R: candidates have start in [sso - R, sso]
            if (range > 0x7FFF) throw Error("normal mode: candidate key range wider than 32767 nt");
            const uint64_t wl_now = sg.splice_end - sg.sso;
            const bool new_epoch = n_steps_tx == 0 || sg.deleted > 0 || last_added > 0 || wl_now != last_wlen || range != last_range ||
                                   (is_fwd ? sg.sso != last_sso + 1 : sg.sso + 1 != last_sso);
            b.step_aux.push_back(uint16_t(range | (new_epoch ? 0x8000u : 0u)));
            last_sso = sg.sso; last_end = sg.splice_end; last_wlen = wl_now; last_range = range; last_added = new_cols.size();
            n_steps_tx++;
        }
        if (c1 - c0 > 65535) throw Error("more than 65535 candidate reads in one step");
        st.cand_lo = c0;
        st.cand_n = uint16_t(c1 - c0);
        st.flags = fl;
        cur_step = b.steps.size();
        b.steps.push_back(st);
        // live-row bound: reads whose start lies in the full candidate key range of this step
        // rows enclose the window (start <= sso, end >= splice_end, so start >= splice_end - max reference span);
        // pending reverse-strand candidates have start >= sso - (max_read_len - ewl)
        uint64_t lo_rows = sg.splice_end > gh.max_ref_span ? sg.splice_end - gh.max_ref_span : 0;
        uint64_t lo_cand = sg.sso > gh.max_read_len - eg.ewl ? sg.sso - (gh.max_read_len - eg.ewl) : 0;
        const uint32_t r_hi = read_lower(sg.cand_hi), r_lo_rows = read_lower(lo_rows);
        uint64_t span = uint64_t(r_hi) - std::min(r_lo_rows, read_lower(lo_cand));
        max_live = std::max(max_live, span);
        if (!NORMAL) {   // side arrays of the window-parallel replay
            const uint32_t rn = r_hi > r_lo_rows ? r_hi - r_lo_rows : 0;
            b.step_ncols.push_back(uint8_t(cols.size()));
            b.step_rlo.push_back(r_lo_rows);
            b.step_rn.push_back(uint16_t(std::min<uint32_t>(rn, 0xFFFF)));
            seg_info.max_rn = std::max(seg_info.max_rn, rn);
            seg_info.read_lo = std::min(seg_info.read_lo, r_lo_rows);
            seg_info.read_hi = std::max(seg_info.read_hi, r_hi);
            if (b.steps.size() - 1 == seg_start) {   // first step of the segment
                seg_info.first_key_lo = uint32_t(sg.cand_lo);
                seg_info.range = uint32_t(sg.sso - sg.cand_lo);
            }
        }
        if (fs_seen) ensure_window(eg, sg);
    }

    std::pair<std::vector<HapSeq>, FsFreq> print(const ExonGeom& eg, const StepGeom& sg, uint64_t frame, FsFreq fsf, bool, std::vector<HapSeq>&& recycled) {
        ensure_window(eg, sg);
        last_print_win = b.steps[cur_step].win;
        last_print_frame = frame;
        if (frame == 0) b.n_main_windows++;
        fsf.try_emplace(frame, 1.0, false);   // (no node is built when the entry exists)
        std::vector<HapSeq> v(std::move(recycled));
        v.clear();
        v.resize(1);
        return {std::move(v), std::move(fsf)};
    }

    void finish() {  // close the transcript's last segment
        const uint32_t here = uint32_t(b.steps.size());
        if (seg_start != 0xFFFFFFFFu && here > seg_start) { b.segs.push_back(SegDev{tx_idx, seg_start, here - seg_start, seg_init_cols}); b.seg_info.push_back(seg_info); }
    }

    void routed(bool to_prev) {
        std::vector<uint32_t>& held = to_prev ? held_prev : held_hap;
        if (last_print_frame == 0) held.clear();
        if (held.empty() || held.back() != last_print_win) held.push_back(last_print_win);
    }

    // the merge reads the full records of both carried-over windows (:1527-1540)
    void splice_merge(const ExonGeom&, const StepGeom&, uint64_t, std::map<uint64_t, uint64_t>&, FsFreq&, std::vector<HapSeq>&,
                      std::vector<HapSeq>&) {
        for (const std::vector<uint32_t>* held : {&held_prev, &held_hap})
            for (uint32_t wi : *held)
                if (wi != 0xFFFFFFFFu) {
                    b.wins[wi].need_recs |= WS_CARRY;
                    b.steps[b.wins[wi].step].flags |= SF_NEED_RECS;
                }
    }
};

}  // namespace

// One wave per exon shortens the critical path but costs throughput (per-wave start-up, one partly used output chunk per
// wave). Exon segments of one transcript are therefore merged back, in order, up to a target length that still leaves
// every SIMD several waves: target = total steps / (2 x 8192 wave slots), at least 256 steps.
static void merge_short_segments(Batch& b) {
    if (b.segs.empty()) return;
    const uint64_t target = std::max<uint64_t>(256, b.steps.size() / 16384);
    PodVec<SegDev> out;
    out.reserve(b.segs.size());
    for (const SegDev& g : b.segs) {
        if (!out.empty() && out.back().tx == g.tx && out.back().step_off + out.back().n_steps == g.step_off &&
            uint64_t(out.back().n_steps) + g.n_steps <= target)
            out.back().n_steps += g.n_steps;   // the wave simply carries its columns across this exon start
        else
            out.push_back(g);
    }
    b.segs.swap(out);
}

// The replay kernels trust the plan: check on the host that every segment's column arithmetic stays in range.
static void validate_segments(const Batch& b) {
    // (the per-segment walks read every step once: shared out over the host threads)
    const size_t nparts = std::max<size_t>(1, std::min<size_t>(host_threads(), (b.segs.size() + b.exons_w.size()) / 256 + 1));
    std::vector<uint64_t> covered_part(nparts, 0);
    std::vector<std::string> errors(nparts);
    auto check = [&](size_t t) {
        try {
            uint64_t covered = 0;
            for (size_t i = b.segs.size() * t / nparts, e = b.segs.size() * (t + 1) / nparts; i < e; i++) {
                const SegDev& g = b.segs[i];
                if (g.tx >= b.tx.size() || uint64_t(g.step_off) + g.n_steps > b.steps.size()) throw Error("internal error: segment outside the plan");
                uint32_t ncols = g.init_cols;
                if (g.n_steps && g.init_cols + b.steps[g.step_off].n_add > b.steps[g.step_off].col_hi) throw Error("internal error: initial columns underflow");
                for (uint32_t k = 0; k < g.n_steps; k++) {
                    const Step& st = b.steps[g.step_off + k];
                    if (st.n_del > ncols) throw Error("internal error: segment drops more columns than it holds");
                    ncols = ncols - st.n_del + st.n_add;
                    if (ncols > 63) throw Error("internal error: more than 63 live columns in a segment");
                    if (st.n_add > st.col_hi) throw Error("internal error: column index underflow in the plan");
                }
                covered += g.n_steps;
            }
            for (size_t i = b.exons_w.size() * t / nparts, e = b.exons_w.size() * (t + 1) / nparts; i < e; i++) {
                const ExonW& x = b.exons_w[i];
                if (uint64_t(x.step_off) + x.n_steps > b.steps.size() || x.tx >= b.tx.size()) throw Error("internal error: window-parallel exon outside the plan");
                covered += x.n_steps;
            }
            for (const PodVec<WChunk>* list : {&b.wchunks, &b.wchunks_m, &b.wchunks_d})
                for (size_t i = list->size() * t / nparts, e = list->size() * (t + 1) / nparts; i < e; i++) {
                    const WChunk& c = (*list)[i];
                    if (c.exon >= b.exons_w.size() || c.step_first < b.exons_w[c.exon].step_off ||
                        uint64_t(c.step_first) + c.n_steps > uint64_t(b.exons_w[c.exon].step_off) + b.exons_w[c.exon].n_steps)
                        throw Error("internal error: work item outside its exon");
                }
            covered_part[t] = covered;
        } catch (const std::exception& e) { errors[t] = e.what(); if (errors[t].empty()) errors[t] = "error"; }
    };
    if (nparts == 1) check(0);
    else {
        std::vector<std::thread> th;
        for (size_t t = 0; t < nparts; t++) th.emplace_back(check, t);
        for (auto& x : th) x.join();
    }
    for (const std::string& e : errors) if (!e.empty()) throw Error(e);
    uint64_t covered = 0;
    for (uint64_t c : covered_part) covered += c;
    if (!b.normal && (b.step_ncols.size() != b.steps.size() || b.step_rlo.size() != b.steps.size() || b.step_rn.size() != b.steps.size()))
        throw Error("internal error: step side arrays out of step with the plan");
    if (covered != b.steps.size()) throw Error("internal error: segments do not cover the plan");
    if (b.normal && b.step_aux.size() != b.steps.size()) throw Error("internal error: step_aux out of step with the plan");
}

// Lane-per-window replay (plan.hpp WinW): flatten every eligible printing window of the window-parallel exons, small windows
// (<= K2L_SMALL_COLS columns) first; the wave-per-window kernels keep only the work items that still hold one of their windows.
static bool lane_window(const Batch& b, uint32_t si) {
    return b.lane_on && (b.steps[si].flags & SF_PRINT) && k2l_takes(b.step_ncols[si], b.step_rn[si], b.lane_hash);
}
static void route_lane_windows(Batch& b) {
    b.winw.clear();
    b.lane_win.clear();
    b.n_lane_small = 0;
    b.n_lane_mid = 0;
    b.win_trivial.assign(b.wins.size() / 32 + 2, 0u);
    b.win_simple.assign(b.wins.size() / 32 + 2, 0u);
    b.win_walk.assign(b.wins.size() / 32 + 2, 0u);
    auto on_threads = [](size_t n, const std::function<void(size_t)>& f) {
        if (n == 1) { f(0); return; }
        std::vector<std::thread> th;
        for (size_t t = 0; t < n; t++) th.emplace_back(f, t);
        for (auto& x : th) x.join();
    };
    {   // the two per-window bitmaps, a range of whole words per thread
        const size_t words = (b.wins.size() + 31) / 32, nt = std::max<size_t>(1, std::min<size_t>(host_threads(), words / 4096 + 1));
        on_threads(nt, [&](size_t t) {
            for (size_t w = words * t / nt * 32, we = std::min(b.wins.size(), words * (t + 1) / nt * 32); w < we; w++) {
                const WinStatic& ws = b.wins[w];
                if ((ws.flags & WSF_SIMPLE) && (ws.flags & WSF_NOSTOP)) b.win_simple[w >> 5] |= 1u << (w & 31);
                if (!(ws.flags & WSF_SIMPLE)) b.win_walk[w >> 5] |= 1u << (w & 31);
                if (!b.normal && (ws.flags & WSF_SIMPLE) && (ws.flags & WSF_NOSTOP) && !(ws.need_recs & WS_MASK) && !(b.steps[ws.step].flags & SF_NEED_RECS))
                    b.win_trivial[w >> 5] |= 1u << (w & 31);
            }
        });
    }
    if (!b.lane_on) return;
    const size_t nthreads = std::max<size_t>(1, std::min<size_t>(host_threads(), b.exons_w.size() / 64 + 1));
    struct Part { PodVec<WinW> w[3]; PodVec<uint32_t> id[3]; };
    std::vector<Part> parts(nthreads);
    auto work = [&](size_t t) {
        Part& P = parts[t];
        const size_t e0 = b.exons_w.size() * t / nthreads, e1 = b.exons_w.size() * (t + 1) / nthreads;
        for (size_t ei = e0; ei < e1; ei++) {
            const ExonW& e = b.exons_w[ei];
            const bool rev = e.strand != 0;
            for (uint32_t k = 0; k < e.n_steps; k++) {
                const uint32_t si = e.step_off + k;
                if (!lane_window(b, si)) continue;
                const Step& st = b.steps[si];
                const uint32_t nc = b.step_ncols[si], rn = b.step_rn[si];
                WinW w{};
                w.rr_lo = rn ? e.adm_off + (b.step_rlo[si] - e.read_lo) : 0;
                const bool trivial = (b.win_trivial[st.win >> 5] >> (st.win & 31)) & 1u;
                w.pack = rn | (nc << 10) | (rev ? 0u : WW_FWD) | ((st.flags & SF_NEED_RECS) ? WW_NEED_ALL : 0u) | (trivial ? WW_TRIVIAL : 0u) |
                         ((b.wins[st.win].need_recs & WS_ALL_IDS) ? WW_ALL_IDS : 0u) | (((b.wins[st.win].flags & WSF_SIMPLE) && (b.wins[st.win].flags & WSF_NOSTOP)) ? WW_SIMPLE : 0u) |
                         ((b.wins[st.win].flags & WSF_SIMPLE) ? 0u : WW_WALK);
                w.wkey = rev ? ~st.sso : st.sso + uint32_t(st.wlen);
                w.step = si;
                w.col_hi = st.col_hi;
                uint64_t som = 0;
                if (nc) {   // forward indices of the live columns: transcription order [col_hi - nc, col_hi)
                    w.flo = rev ? e.f0 - (st.col_hi - 1 - e.tr0) : st.col_hi - nc;
                    const uint64_t gv = uint64_t(e.vbase) + w.flo;
                    const uint64_t x0 = b.v_sombits[gv >> 6], x1 = b.v_sombits[(gv >> 6) + 1];
                    const uint32_t sh = uint32_t(gv & 63);
                    som = (sh ? ((x0 >> sh) | (x1 << (64 - sh))) : x0) & (~0ull >> (64 - nc));
                    if (!rev) {   // haplotype bit order on '+': newest column (highest position) = bit 0
                        uint64_t r = 0;
                        for (uint32_t c = 0; c < nc; c++) if ((som >> c) & 1) r |= 1ull << (nc - 1 - c);
                        som = r;
                    }
                }
                w.som_lo = uint32_t(som); w.som_hi = uint32_t(som >> 32);
                const int cls = nc <= K2L_SMALL_COLS ? 0 : nc <= K2L_MAX_COLS ? 1 : 2;
                P.w[cls].push_back(w);
                P.id[cls].push_back(st.win);
            }
        }
    };
    on_threads(nthreads, work);
    size_t n[3] = {0, 0, 0};
    for (const Part& P : parts) { n[0] += P.w[0].size(); n[1] += P.w[1].size(); n[2] += P.w[2].size(); }
    b.winw.resize(n[0] + n[1] + n[2]);
    b.lane_win.resize(n[0] + n[1] + n[2]);
    b.n_lane_small = uint32_t(n[0]);
    b.n_lane_mid = uint32_t(n[0] + n[1]);
    // every part copies its three runs to their places (class by class, the parts in exon order)
    std::vector<std::array<size_t, 3>> at(nthreads);
    size_t run[3] = {0, n[0], n[0] + n[1]};
    for (size_t t = 0; t < nthreads; t++)
        for (int c = 0; c < 3; c++) { at[t][c] = run[c]; run[c] += parts[t].w[c].size(); }
    on_threads(nthreads, [&](size_t t) {
        Part& P = parts[t];
        for (int c = 0; c < 3; c++) {
            if (P.w[c].empty()) continue;
            std::memcpy(static_cast<void*>(b.winw.data() + at[t][c]), P.w[c].data(), P.w[c].size() * sizeof(WinW));
            std::memcpy(b.lane_win.data() + at[t][c], P.id[c].data(), P.id[c].size() * 4);
            P.w[c] = PodVec<WinW>(); P.id[c] = PodVec<uint32_t>();
        }
    });
}

// Decide which single-exon segments go to the window-parallel replay (plan.hpp ExonW) and cut them into work items.
static void route_window_parallel(Batch& b) {
    const auto t_route0 = std::chrono::steady_clock::now();
    // packed somatic flags of all variants (K2w derives a window's somatic-column mask from it)
    b.v_sombits.assign(b.v_pos.size() / 64 + 2, 0);
    for (size_t v = 0; v < b.v_info.size(); v++)
        if (!(b.v_info[v] & VI_GERMLINE)) b.v_sombits[v >> 6] |= 1ull << (v & 63);
    b.exons_w.clear();
    b.wchunks.clear();
    b.n_adm = 0;
    if (b.seg_info.size() != b.segs.size()) throw Error("internal error: segment info out of step");
    const bool enabled = !b.normal && b.mask_words <= 2 && !std::getenv("MP_SEQUENTIAL_REPLAY");
    b.lane_on = enabled && b.mask_words == 1 && !std::getenv("MP_NO_LANE_KERNEL");
    b.lane_hash = b.lane_on && !std::getenv("MP_NO_LANE_HASH");   // (measurements: the 9..16-column windows back to the wave kernels)
    b.wchunks_m.clear();
    b.wchunks_d.clear();
    b.achunks.clear();
    uint32_t max_rn_multi = 0;
    PodVec<SegDev> keep;
    constexpr uint32_t CHUNK_STEPS = 96;
    // The segments are independent: ranges of them are routed by all host threads (the per-exon scans over the steps - unit stride,
    // which work items still hold a window for the wave kernels - touch all 26 M steps of a whole exome), each into its own lists with
    // range-local exon indices and admission offsets; the lists are then joined in segment order and rebased.
    struct Part {
        PodVec<SegDev> keep;
        PodVec<ExonW> exons;
        PodVec<WChunk> achunks, wchunks, wchunks_m, wchunks_d;
        uint64_t n_adm = 0;
        uint32_t max_rn_multi = 0;
        std::string error;
    };
    const size_t nparts = std::max<size_t>(1, std::min<size_t>(host_threads(), b.segs.size() / 256 + 1));
    std::vector<Part> parts(nparts);
    auto route_range = [&](size_t t) {
      Part& P = parts[t];
      try {
        for (size_t i = b.segs.size() * t / nparts, i_end = b.segs.size() * (t + 1) / nparts; i < i_end; i++) {
        const SegDev& g = b.segs[i];
        const Batch::SegInfo& si = b.seg_info[i];
        const TxDev& T = b.tx[g.tx];
        const GeneHost& gh = b.genes[T.gene];
        bool ok = enabled && si.n_exons == 1 && si.cols_ok && g.n_steps > 0;   // (any depth: k2w_window_rows_deep streams the rows)
        const uint32_t read_lo = si.read_lo == 0xFFFFFFFFu ? 0 : si.read_lo;
        const uint32_t read_hi = std::max(si.read_hi, read_lo);
        if (ok && T.strand)   // `contains` (:281-294) can only hit when two reads of the range share a name
            for (uint32_t r = read_lo; r < read_hi && ok; r++) ok = !(b.r_dup[gh.read_off + r] >> 31);
        if (!ok && std::getenv("MP_DEBUG_SEG")) std::fprintf(stderr, "seq: tx %u steps %u n_exons %u cols_ok %d max_rn %u\n", g.tx, g.n_steps, si.n_exons, int(si.cols_ok), si.max_rn);
        if (!ok) { P.keep.push_back(g); continue; }
        ExonW e{};
        e.tx = g.tx; e.step_off = g.step_off; e.n_steps = g.n_steps;
        e.read_lo = read_lo; e.n_reads = read_hi - read_lo;
        if (P.n_adm + e.n_reads > 0xFFFFFFF0ull) throw Error("batch too large for 32-bit admission-table offsets: split the batch by genes");
        e.adm_off = uint32_t(P.n_adm);   // (range-local: rebased when the ranges are joined)
        P.n_adm += e.n_reads;
        e.first_key_lo = si.first_key_lo; e.range = si.range; e.tr0 = si.tr0; e.f0 = si.f0;
        const uint32_t* vp = b.v_pos.data() + gh.var_off;
        e.sl_f_lo = e.sl_f_hi = 0;
        if (T.sl_lo < T.sl_hi) {
            e.sl_f_lo = uint32_t(std::lower_bound(vp, vp + gh.n_vars, T.sl_lo) - vp);
            e.sl_f_hi = uint32_t(std::lower_bound(vp, vp + gh.n_vars, T.sl_hi) - vp);
        }
        e.strand = T.strand; e.rbase = gh.read_off; e.vbase = gh.var_off;
        e.sso0 = b.steps[g.step_off].sso;
        e.sso1 = g.n_steps > 1 ? b.steps[g.step_off + 1].sso : e.sso0;
        {
            uint32_t u = 1;
            if (T.strand) { while (u < g.n_steps && b.steps[g.step_off + u].sso + u == e.sso0) u++; }
            else { while (u < g.n_steps && b.steps[g.step_off + u].sso == e.sso1 + (u - 1)) u++; }
            e.unit_steps = u;
            uint32_t wmin = 255;
            for (uint32_t k = 0; k < g.n_steps; k++) wmin = std::min<uint32_t>(wmin, b.steps[g.step_off + k].wlen);
            e.wlen_min = wmin;
        }
        const uint32_t ei = uint32_t(P.exons.size());   // (range-local)
        P.exons.push_back(e);
        for (uint32_t k0 = 0; k0 < e.n_reads; k0 += 64) P.achunks.push_back(WChunk{ei, k0, std::min(64u, e.n_reads - k0), 0});
        const bool multi = si.max_rn > 63 || b.mask_words > 1;   // needs several reads per lane / two mask words (63: rows + the reference haplotype fit 64 lanes)
        const bool deep = si.max_rn > 512;                        // beyond the resident block of the multi kernel: rows are streamed
        if (multi && !deep) P.max_rn_multi = std::max(P.max_rn_multi, si.max_rn);
        // deep windows cost ~RPL x more each and there are few of them: smaller work items keep the chip full
        const uint32_t chunk = deep ? 6 : multi ? CHUNK_STEPS / 4 : CHUNK_STEPS;
        uint32_t consumers = 0;
        for (uint32_t s0 = 0; s0 < g.n_steps; s0 += chunk) {
            const uint32_t n = std::min(chunk, g.n_steps - s0);
            bool mine = !b.lane_on;   // a printing window the lane kernel does not take
            for (uint32_t k = 0; k < n; k++) {
                if (!(b.steps[g.step_off + s0 + k].flags & SF_PRINT)) continue;
                if (lane_window(b, g.step_off + s0 + k)) consumers |= EW_LANE; else mine = true;
            }
            if (mine) { (deep ? P.wchunks_d : multi ? P.wchunks_m : P.wchunks).push_back(WChunk{ei, g.step_off + s0, n, 0}); consumers |= EW_WAVE; }
        }
        P.exons[ei].consumers = consumers;
        }
      } catch (const std::exception& e) { P.error = e.what(); if (P.error.empty()) P.error = "error"; }
    };
    if (nparts == 1) route_range(0);
    else {
        std::vector<std::thread> th;
        for (size_t t = 0; t < nparts; t++) th.emplace_back(route_range, t);
        for (auto& x : th) x.join();
    }
    for (const Part& P : parts) if (!P.error.empty()) throw Error(P.error);
    for (Part& P : parts) {   // join in segment order: exon indices and admission offsets move by what came before
        const uint32_t e0 = uint32_t(b.exons_w.size());
        if (b.n_adm + P.n_adm > 0xFFFFFFF0ull) throw Error("batch too large for 32-bit admission-table offsets: split the batch by genes");
        const uint32_t a0 = uint32_t(b.n_adm);
        for (ExonW& e : P.exons) e.adm_off += a0;
        b.exons_w.insert(b.exons_w.end(), P.exons.begin(), P.exons.end());
        auto take = [&](PodVec<WChunk>& dst, PodVec<WChunk>& src) {
            for (WChunk& c : src) c.exon += e0;
            dst.insert(dst.end(), src.begin(), src.end());
        };
        take(b.achunks, P.achunks); take(b.wchunks, P.wchunks); take(b.wchunks_m, P.wchunks_m); take(b.wchunks_d, P.wchunks_d);
        keep.insert(keep.end(), P.keep.begin(), P.keep.end());
        b.n_adm += P.n_adm;
        max_rn_multi = std::max(max_rn_multi, P.max_rn_multi);
    }
    b.rows_per_lane_w = 1;
    while (64u * b.rows_per_lane_w < max_rn_multi) b.rows_per_lane_w *= 2;
    b.segs.swap(keep);
    b.seg_info.clear();
    const auto t_lanes = std::chrono::steady_clock::now();
    route_lane_windows(b);
    if (std::getenv("MP_DEBUG"))
        std::fprintf(stderr, "[mp]   routing: exons and work items %.1f ms, lane windows %.1f ms\n",
                     std::chrono::duration<double, std::milli>(t_lanes - t_route0).count(),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_lanes).count());
}

static void finalize_segments(Batch& b) {
    const bool dbg = std::getenv("MP_DEBUG") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
    const auto t0 = now();
    route_window_parallel(b);
    const auto t1 = now();
    merge_short_segments(b);
    b.seg_order.resize(b.segs.size());
    std::iota(b.seg_order.begin(), b.seg_order.end(), 0u);
    std::stable_sort(b.seg_order.begin(), b.seg_order.end(), [&](uint32_t a, uint32_t c) { return b.segs[a].n_steps > b.segs[c].n_steps; });
    const auto t2 = now();
    validate_segments(b);
    if (dbg) std::fprintf(stderr, "[mp]   finalize: routing %.1f ms, segment order %.1f ms, validation %.1f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, now()));
}

// bit k of the appended dwords = quals[k] < 10 (k < n; the last dword is zero-padded)
static void pack_low_quality(PodVec<uint8_t>& pool, const uint8_t* quals, uint32_t n) {
    const size_t at = pool.size();
    const uint32_t words = (n + 31) / 32;
    pool.resize(at + size_t(words) * 4);
    uint32_t k = 0;
    for (uint32_t w = 0; w < words; w++) {
        uint32_t bits = 0;
#if defined(__SSE2__)
        for (uint32_t h = 0; h < 2 && k + 16 <= n; h++, k += 16) {
            const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(quals + k));
            const __m128i le9 = _mm_cmpeq_epi8(_mm_min_epu8(v, _mm_set1_epi8(9)), v);   // unsigned v <= 9
            bits |= uint32_t(_mm_movemask_epi8(le9)) << (16 * h);
        }
#endif
        for (; k < std::min(n, (w + 1) * 32); k++) bits |= uint32_t(quals[k] < 10) << (k & 31);
        std::memcpy(pool.data() + at + size_t(w) * 4, &bits, 4);
    }
}

static void build_batch_range(const GeneInput* const* genes, size_t n_genes, const ReadStore& rs, uint64_t window_len, bool normal, Batch& b, bool finalize) {
    const uint8_t mapq_min = normal ? 0 : 5;  // src/microphasing.rs:910 vs src/normal_microphasing.rs:676-684
    b = Batch();
    b.window_len = window_len;
    b.normal = normal;
    b.genes.resize(n_genes);
    uint32_t max_span_vars = 0;
    std::vector<uint32_t> pmax_end;
    {   // reserve the big pools once
        size_t nr = 0, nref = 0, nv = 0;
        for (size_t g = 0; g < n_genes; g++) { nr += genes[g]->reads.size(); nref += genes[g]->refseq.size(); nv += genes[g]->variants.size(); }
        for (auto* v : {&b.r_pos, &b.r_end, &b.r_lseq, &b.r_ncig, &b.r_dup}) v->reserve(nr);
        for (auto* v : {&b.r_cigoff, &b.r_seqoff}) v->reserve(nr);
        b.r_src.reserve(nr);
        b.cigar_pool.reserve(nr * 2);
        b.seq_pool.reserve(nr * 72);
        b.ref_pool.reserve(nref);
        for (auto* v : {&b.v_pos, &b.v_info, &b.v_len, &b.v_insoff, &b.v_rev2fwd}) v->reserve(nv);
        b.steps.reserve(nref);
        b.wins.reserve(nref / 3);
    }
    for (size_t gi_ = 0; gi_ < n_genes; gi_++) {
        const GeneInput& gi = *genes[gi_];
        GeneHost& gh = b.genes[gi_];
        gh.input = &gi;
        if (gi.gene.end() + 100 > 0xFFFFFFF0ull) throw Error("coordinate exceeds 32 bits");
        // ---- reads: read_tree (:909-920)
        std::vector<size_t> kept;
        for (size_t r : gi.reads) {
            if (rs.mapq[r] < mapq_min) continue;
            gh.max_read_len = std::max<uint64_t>(gh.max_read_len, rs.l_seq[r]);   // (a read-subset copy of a deep gene: overridden below)
            gh.max_ref_span = std::max<uint64_t>(gh.max_ref_span, uint64_t(rs.end_pos[r] - rs.pos[r]));
            kept.push_back(r);
        }
        if (gi.max_read_len_override) gh.max_read_len = gi.max_read_len_override;
        std::stable_sort(kept.begin(), kept.end(), [&](size_t a, size_t c) { return rs.pos[a] < rs.pos[c]; });
        gh.read_off = uint32_t(b.r_pos.size());
        gh.n_reads = uint32_t(kept.size());
        std::map<std::pair<int64_t, std::string>, uint32_t> first_of;
        for (size_t k = 0; k < kept.size(); k++) {
            size_t r = kept[k];
            if (rs.pos[r] < 0) throw Error("negative read position");
            b.r_pos.push_back(uint32_t(rs.pos[r]));
            b.r_end.push_back(uint32_t(rs.end_pos[r]));
            b.r_lseq.push_back(rs.l_seq[r]);
            b.r_ncig.push_back(rs.n_cigar[r]);
            b.r_cigoff.push_back(b.cigar_pool.size());
            b.cigar_pool.insert(b.cigar_pool.end(), rs.cigar(r), rs.cigar(r) + rs.n_cigar[r]);
            // per read: the low-quality bitmap (bit k = base quality of read offset k below 10, the only thing bad_quality asks of the
            // qualities, src/microphasing.rs:78-93; whole dwords), then the 4-bit packed bases, padded to a dword: K1's two gathers per
            // variant land in one or two cache lines of the read instead of four, and 100 quality bytes per read never travel
            b.r_seqoff.push_back(b.seq_pool.size());
            pack_low_quality(b.seq_pool, rs.qual(r), rs.l_seq[r]);
            const uint8_t* s4 = rs.seq_pool.data() + rs.seq_off[r];
            b.seq_pool.insert(b.seq_pool.end(), s4, s4 + (rs.l_seq[r] + 1) / 2);
            b.seq_pool.resize((b.seq_pool.size() + 3) & ~size_t(3));
            auto key = std::make_pair(rs.pos[r], std::string(rs.qname(r)));
            auto it = first_of.find(key);
            uint32_t dup;
            if (it == first_of.end()) { first_of.emplace(key, uint32_t(k)); dup = uint32_t(k); }
            else dup = it->second | 0x80000000u;  // bit31: an earlier read has the same (pos, qname)
            b.r_dup.push_back(dup);
            b.r_src.push_back(r);
        }
        // mark the FIRST read of each duplicated (pos, qname) key too
        for (size_t k = 0; k < kept.size(); k++) {
            uint32_t d = b.r_dup[gh.read_off + k];
            if (d & 0x80000000u) b.r_dup[gh.read_off + (d & 0x7FFFFFFFu)] |= 0x80000000u;
        }
        // ---- variants
        gh.var_off = uint32_t(b.v_pos.size());
        gh.n_vars = uint32_t(gi.variants.size());
        for (const Variant& v : gi.variants) {
            b.v_pos.push_back(uint32_t(v.pos));
            uint32_t info = uint32_t(v.kind) | (v.is_germline ? VI_GERMLINE : 0) | (uint32_t(v.frameshift()) << VI_FS_SHIFT) |
                            (uint32_t(v.alt) << VI_ALT_SHIFT);
            b.v_info.push_back(info);
            b.v_len.push_back(uint32_t(v.len));
            b.v_insoff.push_back(uint32_t(b.ins_pool.size()));
            if (v.kind == VK_INS) b.ins_pool.insert(b.ins_pool.end(), v.seq.begin(), v.seq.end());
        }
        std::vector<uint32_t> fwd2rev(gi.variants.size());
        {
            size_t hi = gi.variants.size();
            uint32_t rv = 0;
            b.v_rev2fwd.resize(gh.var_off + gi.variants.size());
            while (hi > 0) {
                size_t lo = hi - 1;
                while (lo > 0 && gi.variants[lo - 1].pos == gi.variants[hi - 1].pos) lo--;
                for (size_t k = lo; k < hi; k++) {
                    b.v_rev2fwd[gh.var_off + rv] = uint32_t(k);
                    fwd2rev[k] = rv++;
                }
                hi = lo;
            }
        }
        // mask width: variants spanned by one read
        {
            size_t vlo = 0, vhi = 0;
            const auto& vs = gi.variants;
            for (size_t k = 0; k < kept.size(); k++) {
                uint64_t s = b.r_pos[gh.read_off + k];
                uint64_t e = std::max<uint64_t>(b.r_end[gh.read_off + k], s + b.r_lseq[gh.read_off + k]);
                while (vlo < vs.size() && vs[vlo].pos < s) vlo++;
                b.r_varlo.push_back(uint32_t(vlo));
                if (vhi < vlo) vhi = vlo;
                while (vhi < vs.size() && vs[vhi].pos < e) vhi++;
                size_t hi2 = vhi;
                // reads are start-sorted but ends are not monotone: recount the tail exactly
                while (hi2 > vlo && vs[hi2 - 1].pos >= e) hi2--;
                max_span_vars = std::max<uint32_t>(max_span_vars, uint32_t(hi2 - vlo));
            }
        }
        // prefix maximum of the read ends (reads are start-sorted): "can a read starting at or before X still reach Y"
        pmax_end.resize(b.r_end.size());
        for (uint32_t k = 0, m = 0; k < gh.n_reads; k++) { m = std::max(m, b.r_end[gh.read_off + k]); pmax_end[gh.read_off + k] = m; }
        // ---- refseq
        gh.ref_off = b.ref_pool.size();
        b.ref_pool.insert(b.ref_pool.end(), gi.refseq.begin(), gi.refseq.end());
        b.g_read_off.push_back(gh.read_off);
        b.g_var_off.push_back(gh.var_off);
        b.g_start.push_back(uint32_t(gi.gene.start()));
        b.g_ref_off.push_back(gh.ref_off);
        // ---- transcripts
        gh.tx_off = uint32_t(b.tx.size());
        VarIndex vi{&gi.variants};
        for (size_t ti = 0; ti < gi.gene.transcripts.size(); ti++) {
            const Transcript& t = gi.gene.transcripts[ti];
            if (!t.is_coding()) continue;
            TxDev td{};
            td.gene = uint32_t(gi_);
            td.step_off = uint32_t(b.steps.size());
            td.strand = t.strand == FORWARD ? 0 : 1;
            td.id_off = uint32_t(b.str_pool.size());
            td.id_len = uint32_t(t.id.size());
            b.str_pool.insert(b.str_pool.end(), t.id.begin(), t.id.end());
            // start-loss interval (:1305-1319): the first scheduled exon's first codon
            td.sl_lo = 1; td.sl_hi = 0;
            for (const Interval& ex : t.exons) {
                if (normal) break;  // `normal` has no start-loss bookkeeping
                if (ex.start > ex.end) continue;
                if (t.strand == FORWARD) { td.sl_lo = uint32_t(ex.start); td.sl_hi = uint32_t(ex.start + 3); }
                else { td.sl_lo = uint32_t(ex.end >= 3 ? ex.end - 3 : 0); td.sl_hi = uint32_t(ex.end); }
                break;
            }
            uint64_t tx_live = 0;
            auto run = [&](auto& hooks) {
                hooks.pmax_end = &pmax_end;
                hooks.take_mark();
                try { walk_transcript(gi.gene, t, vi, gh.max_read_len, window_len, hooks); }
                catch (const Error& e) {
                    hooks.rollback();
                    b.tx_errors.emplace_back(uint32_t(b.tx.size()), std::string(e.what()));
                }
                hooks.finish();
                td.n_steps = uint32_t(b.steps.size()) - td.step_off;
                b.max_rows_bound = std::max<uint32_t>(b.max_rows_bound, uint32_t(hooks.max_live));
                tx_live = hooks.max_live;
                b.seq_cap = std::max(b.seq_cap, seq_cap_for(hooks.max_seq_len));
            };
            if (normal) {
                PlannerHooksT<true> hooks{b, gh, gi.variants, fwd2rev, t.strand == FORWARD, uint32_t(b.tx.size())};
                run(hooks);
            } else {
                PlannerHooksT<false> hooks{b, gh, gi.variants, fwd2rev, t.strand == FORWARD, uint32_t(b.tx.size())};
                run(hooks);
            }
            b.tx.push_back(td);
            b.tx_max_live.push_back(uint32_t(std::min<uint64_t>(tx_live, 0xFFFFFFFFull)));
            gh.tx_src.push_back(uint32_t(ti));
        }
        gh.n_tx = uint32_t(b.tx.size()) - gh.tx_off;
    }
    b.g_read_off.push_back(uint32_t(b.r_pos.size()));
    b.g_var_off.push_back(uint32_t(b.v_pos.size()));
    b.mask_words = max_span_vars <= 64 ? 1 : max_span_vars <= 128 ? 2 : 4;
    if (max_span_vars > 256) throw Error("a read spans more than 256 variants; mask width not supported");
    if (finalize) finalize_segments(b);
}

size_t host_threads() {
    if (const char* e = std::getenv("MP_THREADS")) {
        long v = std::atol(e);
        if (v >= 1) return size_t(v);
    }
    unsigned hc = std::thread::hardware_concurrency();
    return hc == 0 ? 1 : std::min<unsigned>(hc, 32);
}

namespace {
template <class T> void append(std::vector<T>& a, const std::vector<T>& b) { a.insert(a.end(), b.begin(), b.end()); }

// Concatenate the sub-batches (gene ranges planned by the worker threads) in gene order, rebasing every absolute index.
// One task per array, run on the same number of threads: the copies are first-touch bound, so they are spread over the cores.
struct PartOff { uint64_t g, r, v, ins, t, s, w, wc, e, str, ref, cig, seq, qual; };

// One array of the merged batch: size it (PodVec: no page is touched), then one copy task per part.
struct MergeTasks {
    std::vector<std::function<void()>> run;
    template <class V, class Fix> void add(V& dst, std::vector<Batch>& parts, V Batch::*m, Fix fix, size_t drop_last = 0) {
        std::vector<size_t> at(parts.size() + 1, 0);
        for (size_t t = 0; t < parts.size(); t++) at[t + 1] = at[t] + (parts[t].*m).size() - std::min(drop_last, (parts[t].*m).size());
        dst.clear();
        dst.reserve(at.back() + 1);   // +1: the offset arrays get their end sentinel appended afterwards
        dst.resize(at.back());
        advise_huge(dst.data(), dst.size() * sizeof(typename V::value_type));
        for (size_t t = 0; t < parts.size(); t++) {
            const size_t n = at[t + 1] - at[t], o = at[t];
            run.push_back([&dst, &parts, m, fix, t, n, o] {
                V& src = parts[t].*m;   // (the sources are given back later, off this path: release_later)
                for (size_t i = 0; i < n; i++) { dst[o + i] = std::move(src[i]); fix(dst[o + i], t); }
            });
        }
    }
    template <class V> void add(V& dst, std::vector<Batch>& parts, V Batch::*m) {
        using T = typename V::value_type;
        static_assert(std::is_trivially_copyable<T>::value, "plain copy");
        std::vector<size_t> at(parts.size() + 1, 0);
        for (size_t t = 0; t < parts.size(); t++) at[t + 1] = at[t] + (parts[t].*m).size();
        dst.clear();
        dst.resize(at.back());
        advise_huge(dst.data(), dst.size() * sizeof(T));
        for (size_t t = 0; t < parts.size(); t++) {
            const size_t o = at[t];
            run.push_back([&dst, &parts, m, t, o] {
                V& src = parts[t].*m;
                if (!src.empty()) std::memcpy(static_cast<void*>(dst.data() + o), src.data(), src.size() * sizeof(T));
            });
        }
    }
};

void merge_parts(Batch& b, std::vector<Batch>& parts, size_t nthreads) {
    const auto t_merge0 = std::chrono::steady_clock::now();
    std::vector<PartOff> o(parts.size() + 1, PartOff{});
    for (size_t t = 0; t < parts.size(); t++) {
        const Batch& s = parts[t];
        PartOff& n = o[t + 1];
        const PartOff& c = o[t];
        n.g = c.g + s.genes.size(); n.r = c.r + s.r_pos.size(); n.v = c.v + s.v_pos.size(); n.ins = c.ins + s.ins_pool.size();
        n.t = c.t + s.tx.size(); n.s = c.s + s.steps.size(); n.w = c.w + s.wins.size(); n.wc = c.wc + s.win_cols.size();
        n.e = c.e + s.exons.size(); n.str = c.str + s.str_pool.size(); n.ref = c.ref + s.ref_pool.size();
        n.cig = c.cig + s.cigar_pool.size(); n.seq = c.seq + s.seq_pool.size();
        b.mask_words = std::max(b.mask_words, s.mask_words);
        b.max_rows_bound = std::max(b.max_rows_bound, s.max_rows_bound);
        b.seq_cap = std::max(b.seq_cap, s.seq_cap);
        b.n_main_windows += s.n_main_windows;
    }
    const PartOff& tot = o.back();
    if (tot.ref > 0xFFFFFFF0ull) throw Error("reference bytes of one batch exceed 4 GiB: split the batch by genes");
    if (tot.s > 0xFFFFFFF0ull || tot.r > 0xFFFFFFF0ull || tot.w > 0xFFFFFFF0ull || tot.wc > 0xFFFFFFF0ull)
        throw Error("batch too large for 32-bit indices: split the batch by genes");
    using B = Batch;
    MergeTasks mt;   // largest arrays first
    mt.add(b.steps, parts, &B::steps, [&o](Step& st, size_t t) { if (st.win != 0xFFFFFFFFu) st.win += uint32_t(o[t].w); st.exon += uint32_t(o[t].e); });
    mt.add(b.seq_pool, parts, &B::seq_pool);
    mt.add(b.wins, parts, &B::wins, [&o](WinStatic& w, size_t t) {
        w.tx += uint32_t(o[t].t); w.col_off += uint32_t(o[t].wc); w.ref_off += uint32_t(o[t].ref); w.vbase += uint32_t(o[t].v); w.step += uint32_t(o[t].s); });
    mt.add(b.win_cols, parts, &B::win_cols);
    mt.add(b.ref_pool, parts, &B::ref_pool);
    mt.add(b.r_src, parts, &B::r_src);
    mt.add(b.r_cigoff, parts, &B::r_cigoff, [&o](uint64_t& x, size_t t) { x += o[t].cig; });
    mt.add(b.r_seqoff, parts, &B::r_seqoff, [&o](uint64_t& x, size_t t) { x += o[t].seq; });
    mt.add(b.step_rlo, parts, &B::step_rlo);
    mt.add(b.cigar_pool, parts, &B::cigar_pool);
    mt.add(b.r_pos, parts, &B::r_pos); mt.add(b.r_end, parts, &B::r_end); mt.add(b.r_lseq, parts, &B::r_lseq);
    mt.add(b.r_ncig, parts, &B::r_ncig); mt.add(b.r_dup, parts, &B::r_dup); mt.add(b.r_varlo, parts, &B::r_varlo);
    mt.add(b.step_rn, parts, &B::step_rn); mt.add(b.step_aux, parts, &B::step_aux); mt.add(b.step_ncols, parts, &B::step_ncols);
    mt.add(b.v_pos, parts, &B::v_pos); mt.add(b.v_info, parts, &B::v_info); mt.add(b.v_len, parts, &B::v_len); mt.add(b.v_rev2fwd, parts, &B::v_rev2fwd);
    mt.add(b.v_insoff, parts, &B::v_insoff, [&o](uint32_t& x, size_t t) { x += uint32_t(o[t].ins); });
    mt.add(b.ins_pool, parts, &B::ins_pool);
    mt.add(b.genes, parts, &B::genes, [&o](GeneHost& g, size_t t) { g.read_off += uint32_t(o[t].r); g.var_off += uint32_t(o[t].v); g.ref_off += o[t].ref; g.tx_off += uint32_t(o[t].t); });
    mt.add(b.g_read_off, parts, &B::g_read_off, [&o](uint32_t& x, size_t t) { x += uint32_t(o[t].r); }, 1);
    mt.add(b.g_var_off, parts, &B::g_var_off, [&o](uint32_t& x, size_t t) { x += uint32_t(o[t].v); }, 1);
    mt.add(b.g_start, parts, &B::g_start);
    mt.add(b.g_ref_off, parts, &B::g_ref_off, [&o](uint64_t& x, size_t t) { x += o[t].ref; });
    mt.add(b.tx, parts, &B::tx, [&o](TxDev& x, size_t t) { x.gene += uint32_t(o[t].g); x.step_off += uint32_t(o[t].s); x.id_off += uint32_t(o[t].str); });
    mt.add(b.segs, parts, &B::segs, [&o](SegDev& x, size_t t) { x.tx += uint32_t(o[t].t); x.step_off += uint32_t(o[t].s); });
    mt.add(b.seg_info, parts, &B::seg_info, [](Batch::SegInfo&, size_t) {});
    mt.add(b.exons, parts, &B::exons, [&o](ExonPlan& e, size_t t) { e.tx += uint32_t(o[t].t); });
    mt.add(b.str_pool, parts, &B::str_pool);
    mt.add(b.tx_max_live, parts, &B::tx_max_live);
    mt.add(b.tx_errors, parts, &B::tx_errors, [&o](std::pair<uint32_t, std::string>& x, size_t t) { x.first += uint32_t(o[t].t); });
    std::vector<std::function<void()>>& tasks = mt.run;
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    const auto t_run = std::chrono::steady_clock::now();
    std::vector<double> busy(nthreads, 0.0);
    for (size_t k = 0; k < std::min(nthreads, tasks.size()); k++)
        th.emplace_back([&, k] {
            const auto a = std::chrono::steady_clock::now();
            for (size_t i; (i = next.fetch_add(1)) < tasks.size();) tasks[i]();
            busy[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
        });
    for (auto& x : th) x.join();
    if (std::getenv("MP_DEBUG")) {
        double lo = 1e30, hi = 0;
        for (size_t k = 0; k < std::min(nthreads, tasks.size()); k++) { lo = std::min(lo, busy[k]); hi = std::max(hi, busy[k]); }
        std::fprintf(stderr, "[mp]   merge: set-up %.1f ms, %zu copy tasks %.1f ms (threads busy %.1f .. %.1f ms)\n",
                     std::chrono::duration<double, std::milli>(t_run - t_merge0).count(), tasks.size(),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_run).count(), lo, hi);
    }
    // (the caller gives the emptied-out sub-batches back - later, off its path: release_later)
}
}  // namespace

// Genes are independent: plan gene ranges on worker threads, then concatenate the sub-batches in gene order.
static void build_batch_plain(const GeneInput* const* genes, size_t n_genes, const ReadStore& rs, uint64_t window_len, bool normal, Batch& b) {
    size_t nthreads = std::min(host_threads(), std::max<size_t>(1, n_genes / 8));
    if (nthreads <= 1) { build_batch_range(genes, n_genes, rs, window_len, normal, b, true); return; }
    std::vector<uint64_t> cost(n_genes + 1, 0);
    for (size_t g = 0; g < n_genes; g++) cost[g + 1] = cost[g] + genes[g]->reads.size() + genes[g]->refseq.size() / 8 + 1;
    std::vector<size_t> cut(nthreads + 1, n_genes);
    cut[0] = 0;
    for (size_t t = 1; t < nthreads; t++) {
        uint64_t target = cost[n_genes] * t / nthreads;
        cut[t] = std::min<size_t>(n_genes, size_t(std::lower_bound(cost.begin(), cost.end(), target) - cost.begin()));
        if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    }
    const bool dbg = std::getenv("MP_DEBUG") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    std::vector<Batch> parts(nthreads);
    std::vector<std::string> errors(nthreads);
    std::vector<std::thread> th;
    for (size_t t = 0; t < nthreads; t++)
        th.emplace_back([&, t] {
            try { build_batch_range(genes + cut[t], cut[t + 1] - cut[t], rs, window_len, normal, parts[t], false); }
            catch (const std::exception& e) { errors[t] = e.what(); if (errors[t].empty()) errors[t] = "error"; }
        });
    for (auto& x : th) x.join();
    for (size_t t = 0; t < nthreads; t++)
        if (!errors[t].empty()) throw Error(errors[t]);
    b = Batch();
    b.window_len = window_len;
    b.normal = normal;
    const auto t1 = now();
    merge_parts(b, parts, nthreads);
    b.g_read_off.push_back(uint32_t(b.r_pos.size()));
    b.g_var_off.push_back(uint32_t(b.v_pos.size()));
    const auto t2 = now();
    finalize_segments(b);
    release_later(std::move(parts));   // after the routing pass: unmapping 5 GB of sub-batches holds the mm lock its page faults wait on
    if (dbg) std::fprintf(stderr, "[mp]   plan on %zu threads %.1f ms, merge %.1f ms, finalize %.1f ms\n", nthreads, ms(t0, t1), ms(t1, t2), ms(t2, now()));
}

// Deep genes. The sequential replay (k2_window_replay: indel / multi-allelic columns, exon chains linked by spanning reads) keeps its
// rows in the lanes of ONE wave - at most 64 x 16 = 1024 simultaneously live reads; the closed-form kernels stream rows and have no
// such limit. A gene that needs the sequential replay somewhere AND can exceed the slots there is planned as K copies, each with
// every K-th read (reads that share (start, name) stay together: `contains`, :281-294, compares exactly those): push_read,
// extend_right, shrink_left and cleanup_reads act on each row by itself (:220-343), so every copy's matrix is the sub-matrix of
// its reads, and a window's haplotype counts, frame depths and nrows are the sums over the copies - formed by the consumer
// (consume.cpp). One amplicon-deep exon with an indel costs its own gene K passes' worth of rows and nothing else. The copies walk
// the whole gene's schedule (max_read_len_override) - checked step by step below.
static constexpr uint32_t K2_ROW_SLOTS = 1024;   // kernels.hip: 64 lanes x RPL <= 16
static uint32_t deep_split_factor(const Batch& b, size_t g) {
    const GeneHost& gh = b.genes[g];
    uint32_t live = 0;
    for (const SegDev& sg : b.segs)   // (segments left to the sequential replay after routing; none in an SNV-only exome)
        if (sg.tx >= gh.tx_off && sg.tx < gh.tx_off + gh.n_tx) live = std::max(live, b.tx_max_live[sg.tx]);
    const char* const e = std::getenv("MP_TEST_ROW_SLOTS");   // tests: split shallow genes
    const uint32_t slots = e && std::atoi(e) >= 8 ? uint32_t(std::atoi(e)) : K2_ROW_SLOTS;
    if (live <= slots) return 1;
    return (live + slots - slots / 8 - 1) / (slots - slots / 8);   // copies of at most 7/8 of the slots each (the bound is per transcript, not per window)
}

void build_batch(const GeneInput* const* genes, size_t n_genes, const ReadStore& rs, uint64_t window_len, bool normal, Batch& b) {
    build_batch_plain(genes, n_genes, rs, window_len, normal, b);
    if (normal || b.segs.empty()) return;   // (`normal` keeps one slot per read and transcript: its own, loud limits)
    std::vector<uint32_t> k_of(n_genes, 1);
    bool any = false;
    for (size_t g = 0; g < n_genes; g++) { k_of[g] = deep_split_factor(b, g); any = any || k_of[g] > 1; }
    if (!any) return;
    std::deque<GeneInput> copies;
    std::vector<const GeneInput*> list;
    std::vector<uint32_t> extra_of;   // per planned gene: copies that follow it (first copy), or 0xFFFFFFFF (a following copy)
    for (size_t g = 0; g < n_genes; g++) {
        const GeneInput& gi = *genes[g];
        if (k_of[g] == 1) { list.push_back(&gi); extra_of.push_back(0); continue; }
        const uint32_t K = k_of[g];
        // the whole gene's max_read_len over the reads the matrix would keep (mapq filter of :910), and the subsets: reads in start order,
        // every first occurrence of a (start, name) pair dealt round robin, later occurrences to the subset of the first
        uint64_t mrl = 0;
        std::vector<size_t> kept;
        for (size_t r : gi.reads) if (rs.mapq[r] >= 5) { mrl = std::max<uint64_t>(mrl, rs.l_seq[r]); kept.push_back(r); }
        std::stable_sort(kept.begin(), kept.end(), [&](size_t a, size_t c) { return rs.pos[a] < rs.pos[c]; });
        std::map<std::pair<int64_t, std::string>, uint32_t> subset_of;
        std::vector<std::vector<size_t>> sub(K);
        uint32_t next = 0;
        for (size_t r : kept) {
            auto ins = subset_of.emplace(std::make_pair(rs.pos[r], std::string(rs.qname(r))), next);
            if (ins.second) next = (next + 1) % K;
            sub[ins.first->second].push_back(r);
        }
        for (uint32_t k = 0; k < K; k++) {
            copies.emplace_back();
            GeneInput& c = copies.back();
            c.gene = gi.gene; c.refseq = gi.refseq; c.variants = gi.variants;
            c.reads = std::move(sub[k]);
            std::sort(c.reads.begin(), c.reads.end());   // BAM file order, as phase_gene lists them
            c.max_read_len_override = mrl ? mrl : 1;
            list.push_back(&c);
            extra_of.push_back(k == 0 ? K - 1 : 0xFFFFFFFFu);
        }
    }
    build_batch_plain(list.data(), list.size(), rs, window_len, normal, b);
    b.split_inputs = std::move(copies);   // (a deque: the addresses the GeneHosts hold stay valid)
    for (size_t p = 0; p < list.size(); p++) {
        if (extra_of[p] == 0xFFFFFFFFu) { b.genes[p].is_extra = true; continue; }
        b.genes[p].n_extra = extra_of[p];
        // every copy walks the first one's schedule: same transcripts, same steps, same printing windows
        for (uint32_t k = 1; k <= extra_of[p]; k++) {
            const GeneHost &a = b.genes[p], &c = b.genes[p + k];
            bool same = a.n_tx == c.n_tx;
            for (uint32_t t = 0; same && t < a.n_tx; t++) {
                const TxDev &ta = b.tx[a.tx_off + t], &tc = b.tx[c.tx_off + t];
                same = ta.n_steps == tc.n_steps;
                for (uint32_t i = 0; same && i < ta.n_steps; i++) {
                    const Step &x = b.steps[ta.step_off + i], &y = b.steps[tc.step_off + i];
                    same = x.sso == y.sso && x.wlen == y.wlen && x.col_hi == y.col_hi && (x.win == 0xFFFFFFFFu) == (y.win == 0xFFFFFFFFu);
                }
            }
            if (!same) throw Error("internal error: the read-subset copies of a deep gene do not share one schedule");
        }
    }
    if (std::getenv("MP_DEBUG")) {
        size_t n = 0, k = 0;
        for (size_t g = 0; g < n_genes; g++) if (k_of[g] > 1) { n++; k += k_of[g]; }
        std::fprintf(stderr, "[mp]   %zu deep gene(s) planned as %zu read-subset copies (sequential replay: %u row slots per wave)\n", n, k, K2_ROW_SLOTS);
    }
}

uint64_t Batch::bytes_k1_in() const {
    return r_pos.size() * (4 * 5 + 8 * 2) + cigar_pool.size() * 4 + seq_pool.size() + v_pos.size() * 12;
}
uint64_t Batch::bytes_k1_out() const { return r_pos.size() * (4 + 16ull * mask_words); }

}  // namespace mp
