// Launch interface of the gfx950 kernels (kernels.hip). Host code never sees kernel symbols.
#pragma once
#include <hip/hip_runtime.h>

#include "plan.hpp"

namespace mp {

struct DeviceBatch {  // device pointers (all hipMalloc'ed by DeviceContext)
    // genes
    const uint32_t *g_read_off, *g_var_off, *g_start;
    const uint64_t* g_ref_off;
    // reads
    const uint32_t *r_gene, *r_pos, *r_end, *r_lseq, *r_ncig, *r_dup;
    const uint64_t *r_cigoff, *r_seqoff, *r_qualoff;
    const uint32_t* cigar_pool;
    const uint8_t *seq_pool, *qual_pool;
    // variants
    const uint32_t *v_pos, *v_info, *v_len, *v_insoff, *v_rev2fwd;
    const uint8_t* ins_pool;
    const uint8_t* ref_pool;
    // plan
    const TxDev* tx;
    const Step* steps;
    const uint16_t* step_aux;     // normal mode: per Step, candidate key range | 0x8000 = new column epoch
    const WinStatic* wins;
    const WinCol* win_cols;
    const uint8_t* str_pool;
    const SegDev* segs;           // replay units (K2 / K2n launch one wave per segment)
    const uint32_t* seg_order;
    uint32_t n_segs;
    uint32_t n_reads, n_tx, n_wins, mask_words;
    uint32_t normal;              // 1: `microphaser normal` semantics (src/normal_microphasing.rs)
    const uint32_t* r_varlo;      // planner: gene-relative index of the first variant with pos >= r_pos
    // K1 output
    uint32_t* r_ncov;             // number of variants (from r_varlo on) whose bits K1 evaluated
    uint64_t *r_sup, *r_lq;  // [read * mask_words + w]
    // K2 output
    WinDyn* win_dyn;
    Group* groups;
    uint32_t* g_win;              // window of each group slot (0xFFFFFFFF = unused slot)
    uint32_t* g_rec;              // HapRec slot reserved for the group by K2 (0xFFFFFFFF = none)
    uint32_t* live_groups;        // dense list of the used group slots (built after K2; K3 runs over it)
    uint8_t* rec_want;            // K3: 1 = this HapRec needs a SHA-1 id
    uint32_t* want_recs;          // dense list of those records (built after K3; K3b runs over it)
    unsigned long long* cursors;  // [0] group-slot cursor, [1] record-slot cursor, [2] number of groups
    uint64_t group_cap, rec_cap;
    uint32_t* err;                // sticky error word (WD_* bits)
    // K3 output
    GroupSum* gsum;
    uint8_t* recs;                // HapRecHdr + seq[seq_cap] + germ[seq_cap], rec_stride bytes apart
    uint32_t seq_cap, rec_stride;
    uint32_t* tx_first_stop;      // per transcript: smallest window index with a main-ORF stop (0xFFFFFFFF none)
};

// rows_per_lane in {1,2,4,8,16}. All launches are asynchronous on `stream`.
void launch_k1_pileup_bits(const DeviceBatch& d, hipStream_t stream);
void launch_k2_window_replay(const DeviceBatch& d, int rows_per_lane, hipStream_t stream);
// dense index lists for K3 / K3b (the chunk allocators of K2 leave unused slots behind); counts land in *d_count (u64)
size_t compaction_temp_bytes(uint64_t n_max);
void launch_compact_live_groups(const DeviceBatch& d, uint64_t n_slots, void* temp, size_t temp_bytes, uint64_t* d_count, hipStream_t stream);
void launch_compact_wanted_recs(const DeviceBatch& d, uint64_t n_recs, void* temp, size_t temp_bytes, uint64_t* d_count, hipStream_t stream);
void launch_k3_window_seq(const DeviceBatch& d, uint64_t n_live_groups, hipStream_t stream);
void launch_k3b_haplotype_ids(const DeviceBatch& d, uint64_t n_recs, hipStream_t stream);

}  // namespace mp
