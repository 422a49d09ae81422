// Host-side packed batch: what mp_phase_genes uploads to HBM. Built by the planner (plan.cpp)
// from the per-gene inputs phase_gene would load (reference: src/microphasing.rs:895-942).
#pragma once
#include <deque>
#include <string>
#include <utility>
#include <vector>

#include "model.hpp"
#include "plan.hpp"
#include "walk.hpp"

namespace mp {

// Host worker threads for planning / consuming (MP_THREADS, default = hardware concurrency capped at 32).
size_t host_threads();

struct ExonPlan {           // host-only: geometry of one scheduled exon pass
    ExonGeom geom;
    uint32_t tx = 0;
};

struct GeneHost {           // host-only per-gene bookkeeping
    const GeneInput* input = nullptr;
    uint64_t max_read_len = 0;
    uint64_t max_ref_span = 0;            // max(end_pos - pos) over the kept reads
    uint32_t read_off = 0, n_reads = 0;   // into the batch read arrays
    uint32_t var_off = 0, n_vars = 0;
    uint64_t ref_off = 0;
    uint32_t tx_off = 0, n_tx = 0;        // coding transcripts only
    std::vector<uint32_t> tx_src;         // index into gene.transcripts for each planned transcript
    // Deep genes (plan.cpp split_deep_genes): the planned genes [this, this + n_extra] are copies of ONE gene that hold disjoint subsets
    // of its reads and walk the same schedule; rows are independent of each other (src/microphasing.rs:297-343: push_read looks at the
    // read and the columns only), so the windows' haplotype counts and depths are the sums over the copies - the consumer adds them up.
    uint32_t n_extra = 0;                 // copies that follow this gene in Batch::genes
    bool is_extra = false;                // this entry is such a copy: no output of its own
};

struct Batch {
    uint64_t window_len = 27;
    bool normal = false;                  // `microphaser normal` semantics (src/normal_microphasing.rs) instead of `somatic`
    // ---- genes
    std::vector<GeneHost> genes;
    PodVec<uint32_t> g_read_off, g_var_off, g_start;  // per gene (+1 for the offsets)
    PodVec<uint64_t> g_ref_off;
    // ---- reads (gene-major, start-sorted, mapq-filtered)
    PodVec<uint32_t> r_pos, r_end, r_lseq, r_ncig, r_dup, r_varlo;
    PodVec<uint64_t> r_cigoff, r_seqoff;   // r_seqoff: dword-aligned start of the read in seq_pool (low-quality bitmap, then 4-bit bases)
    PodVec<uint32_t> cigar_pool;
    PodVec<uint8_t> seq_pool;
    PodVec<size_t> r_src;            // host-only: ReadStore index of each batch read
    // ---- variants
    PodVec<uint32_t> v_pos, v_info, v_len, v_insoff, v_rev2fwd;
    PodVec<uint8_t> ins_pool;
    // ---- refseq
    PodVec<uint8_t> ref_pool;
    // ---- plan
    PodVec<TxDev> tx;
    PodVec<Step> steps;
    PodVec<uint16_t> step_aux;       // normal mode only, one per Step: candidate key range R | 0x8000 = a new column epoch starts
    PodVec<WinStatic> wins;
    PodVec<WinCol> win_cols;         // column lists of the printing windows
    std::vector<ExonPlan> exons;
    PodVec<uint8_t> str_pool;        // transcript ids
    PodVec<SegDev> segs;             // independent replay units (whole exons of one transcript), see plan.hpp
    PodVec<uint32_t> seg_order;      // launch order (longest first)
    // window-parallel replay (somatic): per-step side arrays, eligible exons, their work items
    PodVec<uint8_t> step_ncols;      // live columns after the step's appends
    PodVec<uint32_t> step_rlo;       // gene-relative index of the first read that can still enclose the step's window
    PodVec<uint16_t> step_rn;        // number of reads from there up to start <= sso (saturating)
    struct SegInfo {                      // planner scratch, parallel to segs until finalize
        uint32_t n_exons = 0; bool cols_ok = true; uint32_t max_rn = 0; uint32_t read_lo = 0xFFFFFFFFu, read_hi = 0;
        uint32_t first_key_lo = 0, range = 0, tr0 = 0, f0 = 0; bool have_col = false; bool dup = false;
    };
    std::vector<SegInfo> seg_info;
    PodVec<ExonW> exons_w;
    PodVec<WChunk> wchunks, wchunks_m, wchunks_d;   // single-block / multi-block / streamed (any depth) window-parallel work items (kernels.hpp)
    uint32_t rows_per_lane_w = 1;
    PodVec<WinW> winw;                   // lane-per-window replay (plan.hpp WinW): one per entry of lane_small / lane_wide
    uint32_t n_lane_small = 0;           // winw[0 .. n_lane_small): windows with <= K2L_SMALL_COLS columns, [n_lane_small, n_lane_mid): up to K2L_MAX_COLS,
    uint32_t n_lane_mid = 0;             // the rest: 9..K2L_HASH_COLS columns (the lane kernel's hash-table form)
    PodVec<uint32_t> lane_win;           // window index of each winw entry
    PodVec<uint32_t> win_trivial;        // bit per window, see kernels.hpp
    PodVec<uint32_t> win_walk;           // bit per window: !WSF_SIMPLE (the general sequence walk: K3's list D)
    PodVec<uint32_t> win_simple;         // bit per window: WSF_SIMPLE && WSF_NOSTOP (the wave-per-window kernels route a window's groups to K3's list A / B or C by it)
    bool lane_on = false;                // K2a writes RowRecs and the lane kernel takes the eligible windows
    bool lane_hash = false;              // ... including those of 9..16 columns (its hash-table form)
    PodVec<WChunk> achunks;              // admission work items: (exon, first read, count <= 64)
    uint64_t n_adm = 0;                   // AdmEntry count (sum of ExonW::n_reads)
    PodVec<uint64_t> v_sombits;      // bit (variant index in the batch) set <=> somatic
    // transcripts whose speculative schedule ran into a failure: (index into tx, message); the plan stops before that step and the
    // consumer raises the message only if the real walk reaches it
    std::vector<std::pair<uint32_t, std::string>> tx_errors;
    PodVec<uint32_t> tx_max_live;         // host-only, per transcript: upper bound on its simultaneously live rows (+ pending candidates)
    std::deque<GeneInput> split_inputs;   // the read-subset copies of deep genes (owned here: GeneHost::input points into it)
    // ---- sizing
    uint32_t seq_cap = SEQ_CAPS[0];       // HapRec sequence capacity of this batch (SEQ_CAPS)
    uint32_t mask_words = 1;              // W: u64 words of the per-read support / low-qual masks
    uint32_t max_rows_bound = 0;          // upper bound on simultaneously live rows (+pending) of any transcript
    uint64_t n_main_windows = 0;          // main-ORF printing steps in the plan (speculative upper bound)

    // algorithmic byte counts for the roofline line
    uint64_t bytes_k1_in() const;
    uint64_t bytes_k1_out() const;
};

// Build the batch + plan for a list of loaded genes (any subset, in output order), for `somatic` (normal = false) or `normal` mode.
void build_batch(const GeneInput* const* genes, size_t n_genes, const ReadStore& reads, uint64_t window_len, bool normal, Batch& out);
inline void build_batch(const GeneInput* genes, size_t n_genes, const ReadStore& reads, uint64_t window_len, bool normal, Batch& out) {
    std::vector<const GeneInput*> ptrs(n_genes);
    for (size_t g = 0; g < n_genes; g++) ptrs[g] = genes + g;
    build_batch(ptrs.data(), n_genes, reads, window_len, normal, out);
}
inline void build_batch(const std::vector<GeneInput>& genes, const ReadStore& reads, uint64_t window_len, bool normal, Batch& out) {
    build_batch(genes.data(), genes.size(), reads, window_len, normal, out);
}

}  // namespace mp
