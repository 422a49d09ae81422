// The window scheduler of phase_gene, host side of the product.
//
// reference: src/microphasing.rs:944-1941 (per-transcript exon loop, window tuples, column
// deltas, candidate read ranges, frameshift bookkeeping, termination, splice-side merge trigger).
//
// The reference interleaves this control flow with the per-window work (matrix update and
// print_haplotypes). Here the control flow is ONE template that is instantiated twice:
//   * the planner (plan.cpp) runs it with hooks that never terminate and records the static
//     schedule (Step / WinStatic) that the HIP kernels replay;
//   * the consumer (consume.cpp) runs it again with hooks that answer print_haplotypes from the
//     kernels' results, so termination / frameshift pruning / splice merges happen exactly where
//     the reference does them.
// Everything a hook does not decide is a pure function of the gene model, the variant
// positions and max_read_len, so both runs see the same step sequence until termination.
#pragma once
#include <algorithm>
#include <map>
#include <memory>
#include <vector>

#include "model.hpp"

namespace mp {

using FsFreq = std::map<uint64_t, std::pair<double, bool>>;  // frameshift_frequencies

struct HapSeq {  // reference: HaplotypeSeq (microphasing.rs:141-145); record carries the unsliced sequences
    // The record types hold two dozen strings; most haplotypes of most windows never get one (nothing can observe them), and the
    // consumer creates a HapSeq per haplotype per window - so the payload is allocated only when a record is actually built.
    struct Payload { IDRecord record; };
    struct NormalPayload {   // `normal` mode (normal_microphasing.rs:182-186): the sequence bytes + its own record type
        std::vector<uint8_t> sequence;
        NormalRecord nrecord;
    };
    std::unique_ptr<Payload> p;
    std::unique_ptr<NormalPayload> np;
    // `normal` consumer: what it takes to build the NormalPayload when a splice-side merge asks for it (only the windows at exon ends
    // are ever merged; every window of every exon has haplotypes): the device record and the three per-call values
    uint32_t lazy_rec = 0xFFFFFFFFu, lazy_nrows = 0;
    double lazy_freq = 0;
    bool filled = false;   // consumer: the record was built (the planner marked the window as carried or it is emitted)
    uint32_t win = 0xFFFFFFFFu;   // consumer: window the haplotype came from (diagnostics)
    uint64_t frame = 0;
    Payload& make() { if (!p) p.reset(new Payload()); return *p; }
    const Payload& get() const { static const Payload empty; return p ? *p : empty; }
    NormalPayload& nmake() { if (!np) np.reset(new NormalPayload()); return *np; }
    const NormalPayload& nget() const { static const NormalPayload empty; return np ? *np : empty; }
};

struct ExonGeom {
    size_t exon_idx = 0;      // index into transcript.exons
    uint64_t start = 0, end = 0;
    uint64_t ceo = 0;         // current_exon_offset
    uint64_t ewl = 0;         // exon_window_len
    bool is_short = false, is_first = false, is_last = false;
};

struct StepGeom {
    uint64_t offset = 0, sso = 0, splice_end = 0, splice_gap = 0, splice_pos = 0, rest = 0;
    bool is_first_exon_window = false, is_last_exon_window = false;
    bool first_of_exon_candidates = false;  // candidate reads come from the full key range
    uint64_t cand_lo = 0, cand_hi = 0;      // read start keys in [cand_lo, cand_hi)
    size_t nvars = 0, added = 0, deleted = 0;  // deleted includes the exon-start shrink of last_window_vars
    size_t vlo = 0, vhi = 0;                // forward (position-sorted) variant index range of [sso, splice_end)
};

// lower_bound on a sorted position array for query streams that move by a few positions per call (the window scheduler asks for
// sso, splice_end, old_offset, old_end ... of consecutive nt offsets): a few remembered (position, index) pairs, the nearest one is
// walked linearly to the answer; a far query falls back to the binary search. Pure function of (array, pos): same results.
struct HintedLower {
    static constexpr int SLOTS = 4;
    mutable uint64_t hpos[SLOTS] = {0, 0, 0, 0};
    mutable size_t hidx[SLOTS] = {0, 0, 0, 0};
    mutable int filled = 0, next = 0;
    template <class At>   // at(i) = position of element i, ascending; n elements
    size_t lower(uint64_t pos, size_t n, const At& at) const {
        int best = -1;
        uint64_t best_d = 64;   // only hints within 64 positions are worth walking from
        for (int k = 0; k < filled; k++) {
            const uint64_t dlt = hpos[k] > pos ? hpos[k] - pos : pos - hpos[k];
            if (dlt < best_d) { best_d = dlt; best = k; }
        }
        size_t i;
        if (best >= 0) {
            i = hidx[best];
            while (i < n && at(i) < pos) i++;
            while (i > 0 && at(i - 1) >= pos) i--;
        } else {
            size_t lo = 0, hi = n;
            while (lo < hi) {
                size_t mid = (lo + hi) / 2;
                if (at(mid) < pos) lo = mid + 1; else hi = mid;
            }
            i = lo;
            best = next;
            next = (next + 1) % SLOTS;
            if (filled < SLOTS) filled++;
        }
        hpos[best] = pos;
        hidx[best] = i;
        return i;
    }
};

// Variant positions of one gene in forward order (ascending pos, ALT order within a pos) with range counting.
struct VarIndex {
    const std::vector<Variant>* vars = nullptr;
    HintedLower hint;
    size_t lower(uint64_t pos) const {
        const std::vector<Variant>& v = *vars;
        return hint.lower(pos, v.size(), [&v](size_t i) { return v[i].pos; });
    }
    size_t count(uint64_t a, uint64_t b) const {
        if (a > b) throw Error("reference would panic: range start is greater than range end in BTreeMap");
        return lower(b) - lower(a);
    }
};

// Hooks concept:
//   void on_exon(const ExonGeom&);
//   void on_step(const ExonGeom&, const StepGeom&, const std::vector<size_t>& new_cols_fwd_idx);
//       called once per step after the column delta is known (columns are appended in the given order)
//   std::pair<std::vector<HapSeq>, FsFreq> print(const ExonGeom&, const StepGeom&, uint64_t frame, FsFreq, bool is_first_exon_window,
//                                                std::vector<HapSeq>&& recycled);   // recycled: a vector whose buffer the result may take over
//   void begin_step();                   // top of every window-loop iteration (the previous step is complete)
//   void routed(bool to_prev_hap_vec);   // which carry-over vector the last print's haplotypes went to
//   void splice_merge(const ExonGeom&, const StepGeom&, uint64_t exon_rest, std::map<uint64_t,uint64_t>& frameshifts,
//                     FsFreq&, std::vector<HapSeq>& hap_vec, std::vector<HapSeq>& prev_hap_vec);
//   static constexpr bool kNormal;    // true: the control flow of `microphaser normal` (src/normal_microphasing.rs:700-1277)
template <class Hooks>
void walk_transcript(const Gene& gene, const Transcript& transcript, const VarIndex& vi, uint64_t max_read_len,
                     uint64_t window_len, Hooks& hooks) {
    const std::vector<Variant>& vars = *vi.vars;
    const bool is_fwd = transcript.strand == FORWARD;
    size_t exon_number = transcript.exons.size();
    std::map<uint64_t, uint64_t> frameshifts;
    if (is_fwd) frameshifts[0] = 0; else frameshifts[gene.end()] = 0;
    uint64_t exon_rest = 0;
    std::vector<HapSeq> prev_hap_vec, hap_vec;
    std::vector<HapSeq> spare;   // the vector the last print's result replaced: its buffer goes to the next print (a heap allocation per window otherwise)
    FsFreq frameshift_frequencies;
    frameshift_frequencies[0] = {1.0, false};
    size_t last_window_vars = 0;
    size_t carry_shrink = 0;  // exon-start shrink not yet folded into a step (exon without any window)
    size_t exon_count = 0;
    std::vector<size_t> new_cols, all;                       // per-step scratch, allocated once per transcript
    std::vector<std::pair<uint64_t, uint64_t>> active;
    for (size_t ei = 0; ei < transcript.exons.size(); ei++) {
        const Interval& exon = transcript.exons[ei];
        if (frameshifts.empty()) break;
        if (exon.start > exon.end) continue;
        exon_count++;
        ExonGeom eg;
        eg.exon_idx = ei;
        eg.start = exon.start;
        eg.end = exon.end;
        uint64_t exon_len = exon.end - exon.start;
        if constexpr (Hooks::kNormal) {  // src/normal_microphasing.rs:733-742 (enumerate index, exon.frame ignored)
            eg.ceo = exon_rest == 0 ? 0 : 3 - exon_rest;
            eg.is_last = ei == exon_number - 1;
            eg.is_first = ei == 0;
        } else {
            eg.ceo = exon_count == 1 ? exon.frame : (exon_rest == 0 ? 0 : 3 - exon_rest);
            eg.is_last = exon_count == exon_number;
            eg.is_first = exon_count == 1;
        }
        eg.is_short = exon_len < 3 ? true : window_len >= exon_len - eg.ceo - (3 - eg.ceo) % 3;
        eg.ewl = !eg.is_short ? window_len : (exon_len - eg.ceo) - ((exon_len - eg.ceo) % 3);
        if (eg.ewl == 0) eg.ewl = exon_len;
        exon_rest = 0;
        uint64_t offset = !is_fwd ? exon.end - eg.ewl - eg.ceo : exon.start + eg.ceo;
        bool reached_end = false;
        uint64_t old_offset = offset;
        uint64_t old_end = old_offset + eg.ewl;
        size_t pending_shrink = last_window_vars + carry_shrink;  // observations.shrink_left(last_window_vars) (:1027)
        last_window_vars = 0;
        carry_shrink = 0;
        bool is_first_exon_window = true;
        hooks.on_exon(eg);
        for (;;) {
            hooks.begin_step();   // everything of the previous step (prints, merge) is done
            if (frameshifts.empty()) break;
            bool valid = is_fwd ? offset + eg.ewl <= exon.end : offset >= exon.start;
            if (!valid) break;
            if (max_read_len < eg.ewl) break;
            StepGeom sg;
            sg.offset = offset;
            sg.rest = is_fwd ? exon.end - (offset + eg.ewl) : offset - exon.start;
            sg.is_first_exon_window = is_first_exon_window;
            sg.is_last_exon_window = sg.rest < 3;
            if (is_fwd) {  // :1058-1089
                if (eg.is_short || (is_first_exon_window && sg.is_last_exon_window)) {
                    sg.sso = offset - eg.ceo; sg.splice_end = offset + eg.ewl + sg.rest; sg.splice_gap = eg.ceo + sg.rest; sg.splice_pos = 2;
                } else if (is_first_exon_window) {
                    sg.sso = offset - eg.ceo; sg.splice_end = offset + eg.ewl; sg.splice_gap = eg.ceo; sg.splice_pos = 1;
                } else if (sg.is_last_exon_window) {
                    sg.sso = offset; sg.splice_end = offset + eg.ewl + sg.rest; sg.splice_gap = sg.rest; sg.splice_pos = 0;
                } else {
                    sg.sso = offset; sg.splice_end = offset + eg.ewl; sg.splice_gap = 0; sg.splice_pos = 0;
                }
            } else {  // :1090-1110
                if (eg.is_short) {
                    sg.sso = offset - sg.rest; sg.splice_end = offset + eg.ewl + eg.ceo; sg.splice_gap = eg.ceo + sg.rest; sg.splice_pos = 2;
                } else if (is_first_exon_window) {
                    sg.sso = offset; sg.splice_end = offset + eg.ewl + eg.ceo; sg.splice_gap = eg.ceo; sg.splice_pos = 0;
                } else if (sg.is_last_exon_window) {
                    sg.sso = offset - sg.rest; sg.splice_end = offset + eg.ewl; sg.splice_gap = sg.rest; sg.splice_pos = 1;
                } else {
                    sg.sso = offset; sg.splice_end = offset + eg.ewl; sg.splice_gap = 0; sg.splice_pos = 0;
                }
            }
            sg.vlo = vi.lower(sg.sso);
            sg.vhi = vi.lower(sg.splice_end);
            if (sg.sso > sg.splice_end) throw Error("reference would panic: range start is greater than range end in BTreeMap");
            sg.nvars = sg.vhi - sg.vlo;
            last_window_vars = sg.nvars;
            if (is_first_exon_window) sg.added = sg.nvars;  // :1129-1156 (read_through is always false here)
            else if (eg.is_short) sg.added = 0;
            else if (reached_end) sg.added = 0;
            else if (sg.sso > old_offset) sg.added = vi.count(old_end, sg.splice_end);
            else sg.added = vi.count(sg.sso, old_offset);
            if (offset == old_offset || eg.is_short) sg.deleted = 0;  // :1159-1178
            else if (sg.sso > old_offset) sg.deleted = vi.count(old_offset, sg.sso);
            else sg.deleted = vi.count(sg.splice_end, old_end);
            sg.deleted += pending_shrink;
            pending_shrink = 0;
            if (sg.is_last_exon_window) reached_end = true;
            // candidate reads (:1191-1249)
            sg.first_of_exon_candidates = is_fwd ? offset == exon.start + eg.ceo : true;
            sg.cand_hi = sg.sso + 1;
            if (sg.first_of_exon_candidates) {
                if (sg.sso < max_read_len - eg.ewl) throw Error("reference would panic: attempt to subtract with overflow (read range)");
                sg.cand_lo = sg.sso - (max_read_len - eg.ewl);
            } else {
                sg.cand_lo = sg.sso;
            }
            // new columns in transcription order (:1280-1296): the last `added` of the window's variants
            if (sg.added > sg.nvars) throw Error("reference would panic: attempt to subtract with overflow (nvars - added_vars)");
            new_cols.clear();
            if (is_fwd) {
                for (size_t k = sg.vlo + (sg.nvars - sg.added); k < sg.vhi; k++) new_cols.push_back(k);
            } else {
                // positions descending, ALT order kept within one position
                all.clear();
                size_t hi = sg.vhi;
                while (hi > sg.vlo) {
                    size_t lo = hi - 1;
                    while (lo > sg.vlo && vars[lo - 1].pos == vars[hi - 1].pos) lo--;
                    for (size_t k = lo; k < hi; k++) all.push_back(k);
                    hi = lo;
                }
                for (size_t k = sg.nvars - sg.added; k < all.size(); k++) new_cols.push_back(all[k]);
            }
            for (size_t k : new_cols) {  // frameshift bookkeeping (:1299-1342); start_loss handled by position interval
                const Variant& variant = vars[k];
                uint64_t s = variant.frameshift();
                if constexpr (Hooks::kNormal) {  // src/normal_microphasing.rs:1042-1048: no % 3, keyed by end_pos on both strands
                    if (s > 0) {
                        std::vector<uint64_t> previous;
                        for (const auto& kv : frameshifts) previous.push_back(kv.second + s);
                        for (uint64_t s_ : previous) frameshifts[variant.end_pos()] = s_;
                    }
                } else if ((s % 3) > 0) {
                    std::vector<uint64_t> previous;
                    for (const auto& kv : frameshifts) previous.push_back(kv.second + s);
                    for (uint64_t s_ : previous) frameshifts[is_fwd ? variant.end_pos() : variant.pos] = s_ % 3;
                }
            }
            hooks.on_step(eg, sg, new_cols);
            uint64_t stopped_frameshift = 3;
            active.clear();  // :1347-1350
            if (is_fwd) {
                for (auto it = frameshifts.begin(); it != frameshifts.end() && it->first < offset; ++it) active.push_back(*it);
            } else {
                for (auto it = frameshifts.lower_bound(offset + eg.ewl); it != frameshifts.end(); ++it) active.push_back(*it);
            }
            size_t frameshift_count = 0;
            bool main_orf = false;
            for (const auto& kf : active) {  // :1362-1463
                uint64_t key = kf.first, frameshift = kf.second;
                frameshift_count++;
                if (frameshift == 0) main_orf = true;
                uint64_t coding_shift = is_fwd ? offset - exon.start : exon.end - offset;
                bool has_frameshift = frameshift > 0;
                if (coding_shift % 3 == (frameshift + eg.ceo) % 3 || eg.is_short) {
                    if (!has_frameshift) {
                        exon_rest = sg.rest;
                        if (eg.ewl < 3) exon_rest = eg.ewl;
                    }
                    auto res = hooks.print(eg, sg, frameshift, std::move(frameshift_frequencies), is_first_exon_window, std::move(spare));
                    frameshift_frequencies = std::move(res.second);
                    bool to_prev;
                    if constexpr (Hooks::kNormal) {  // src/normal_microphasing.rs:1113-1122
                        if (res.first.empty()) stopped_frameshift = key;
                        to_prev = exon_rest < 3 && (!eg.is_short || eg.is_first);
                    } else {
                        if (res.first.empty() || !frameshift_frequencies.count(frameshift)) stopped_frameshift = key;
                        to_prev = exon_rest < 3 && (!eg.is_short || eg.is_first) && !has_frameshift;  // :1445-1454
                    }
                    if (to_prev) std::swap(prev_hap_vec, res.first);
                    else std::swap(hap_vec, res.first);
                    spare = std::move(res.first);
                    hooks.routed(to_prev);
                    if constexpr (!Hooks::kNormal) {
                        if (frameshift != 0 && frameshift_frequencies.count(frameshift) && frameshift_frequencies.at(frameshift).first == 0.0)
                            stopped_frameshift = key;
                    }
                }
            }
            if constexpr (Hooks::kNormal) {  // src/normal_microphasing.rs:1125-1135
                if (frameshift_count == 0 || !main_orf) { frameshifts.clear(); break; }
                frameshifts.erase(stopped_frameshift);
                if (frameshifts.empty()) break;
            } else {
                if (frameshift_count == 0 || !main_orf || !frameshift_frequencies.count(0)) {  // :1465-1473
                    frameshifts.clear();
                    break;
                }
                if (stopped_frameshift != 3) {  // :1477-1481
                    auto it = frameshifts.find(stopped_frameshift);
                    if (it == frameshifts.end()) throw Error("reference would panic: unwrap on None (stopped_frameshift)");
                    if (it->second != 0) frameshifts.erase(it);
                }
                if (frameshifts.empty()) break;
                if (frameshift_frequencies.at(0).first == 0.0 && frameshifts.size() == 1) {  // :1485-1488
                    frameshifts.clear();
                    break;
                }
            }
            bool at_splice_side = is_fwd ? offset - eg.ceo == exon.start : offset + eg.ewl + eg.ceo == exon.end;  // :1497-1502
            is_first_exon_window = false;
            if (at_splice_side && !eg.is_first)  // :1505-1908
                hooks.splice_merge(eg, sg, exon_rest, frameshifts, frameshift_frequencies, hap_vec, prev_hap_vec);
            old_offset = sg.sso;  // :1909-1914
            old_end = sg.splice_end;
            if (is_fwd) offset += 1; else offset -= 1;
            if (frameshifts.empty()) break;
            if (eg.is_short) break;  // :1928-1931
        }
        carry_shrink = pending_shrink;
    }
}

}  // namespace mp
