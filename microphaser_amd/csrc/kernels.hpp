// Launch interface of the gfx950 kernels (kernels.hip). Host code never sees kernel symbols.
#pragma once
#include <hip/hip_runtime.h>

#include "plan.hpp"

namespace mp {

constexpr uint32_t NPART = 64;   // output allocators (power of two)

struct DeviceBatch {  // device pointers (all hipMalloc'ed by DeviceContext)
    // genes
    const uint32_t *g_read_off, *g_var_off, *g_start;
    const uint64_t* g_ref_off;
    // reads
    uint32_t n_genes, n_genes_pad_;
    uint2* r_var;                 // per read: {absolute index of its first variant (g_var_off + r_varlo), variants of the gene from there on}
    const uint32_t *r_pos, *r_end, *r_lseq, *r_ncig, *r_dup;
    const uint64_t *r_cigoff, *r_seqoff;
    const uint32_t* cigar_pool;
    const uint8_t* seq_pool;      // per read: low-quality bitmap (ceil(l_seq / 32) dwords), then the 4-bit packed bases (plan.cpp)
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
    ExonA* exons_a;               // per window-parallel exon / per admission-table entry, packed on the device at upload (launch_k0_pack_admission)
    AdmMap* adm_map;
    uint32_t k2a_flat, k2a_pad_;  // the flat form of K2a (a lane per table entry across exon boundaries); MP_K2A_CHUNKS=1: a wave per <= 64 reads of ONE exon
    WinBlob* win_blobs;           // one per window, packed on the device at upload (launch_k0_pack_windows); somatic mode only
    const uint8_t* str_pool;
    const SegDev* segs;           // replay units (K2 / K2n launch one wave per segment)
    const uint32_t* seg_order;
    uint32_t n_segs;
    // window-parallel replay (K2a + K2w), see plan.hpp ExonW
    const ExonW* exons_w;
    const WChunk* wchunks;          // work items of k2w_window_rows (<= 64 candidate reads per window, one mask word)
    const WChunk* wchunks_m;        // work items of k2w_window_rows_multi (deeper exons, or two mask words)
    const WChunk* wchunks_d;        // work items of k2w_window_rows_deep (more than 512 candidate reads per window)
    uint32_t n_wchunks_d, lane_hash;   // lane_hash: the lane kernel also takes the windows of 9..16 columns (plan.hpp k2l_takes)
    const uint8_t* step_ncols;
    const uint32_t* step_rlo;
    const uint16_t* step_rn;
    const uint64_t* v_sombits;      // bit (var_off + f) set <=> that variant is somatic
    AdmEntry* adm;                  // K2a output
    // lane-per-window replay (plan.hpp WinW): K2a also writes a RowRec per (exon, read); k2l_window_lanes takes the listed windows
    RowRecA* rr_a;
    uint64_t* rr_sup;
    const WinW* winw;
    const uint32_t* lane_win;
    const uint32_t* win_trivial;    // bit per window: WSF_SIMPLE && WSF_NOSTOP and no record demand of its own (plan.hpp WW_TRIVIAL)
    const uint32_t* win_walk;       // bit per window: !WSF_SIMPLE (needs the general sequence walk: list D)
    const uint32_t* win_simple;     // bit per window: WSF_SIMPLE && WSF_NOSTOP
    uint32_t n_lane_small, n_lane_all, lane_on, n_lane_mid;   // winw[0, small): <= 6 columns, [small, mid): 7-8, [mid, all): 9-16 (hash form)
    const WChunk* achunks;          // work items of k2a_admission: (exon, first read of the exon's range, count <= 64)
    const ExonW* achunk_exons;      // the exon record of every admission work item, beside it (one load level less in a latency-bound kernel)
    uint32_t n_exons_w, n_wchunks, n_wchunks_m, n_achunks;
    uint32_t rows_per_lane_w;       // RPL of k2w_window_rows_multi: 64 * RPL >= candidate reads of any of its windows
    uint64_t n_adm;
    uint32_t n_reads, n_tx, n_wins, mask_words;
    uint32_t normal;              // 1: `microphaser normal` semantics (src/normal_microphasing.rs)
    uint32_t normal_large;        // k2n_window_replay with the large per-wave tables (set after the small ones overflowed)
    uint32_t timing_skip_ids;     // MP_TIMING_SKIP_IDS=1 (measurements only, WRONG ids): K3 skips the SHA-1 arithmetic - what is left is its gather / walk / store time
    uint32_t timing_pad_;
    const uint32_t* r_varlo;      // planner: gene-relative index of the first variant with pos >= r_pos
    // K1 output
    uint32_t* r_ncov;             // number of variants (from r_varlo on) whose bits K1 evaluated
    uint64_t *r_sup, *r_lq;  // [read * mask_words + w]
    // K2 output
    WinDyn* win_dyn;
    Group* groups;
    uint4* k3_items;              // K2 -> K3: per allocator three dense lists of 16-byte items {group slot, window, record slot K2 reserved or 0xFFFFFFFF, 0}: list A
                                  // (ids hashed) upwards from the sub-range's start, list B downwards from its end, list C (windows whose sequences need the
                                  // general walk) upwards in a second array of the same size behind the first (kernels.hip k3_enqueue)
    uint32_t* want_recs;          // normal mode, K3n: NPART dense lists of the records that need a SHA-1 id (K3b runs over them)
    // Output slots are handed out by NPART independent allocators (a wave uses allocator blockIdx & (NPART - 1)), each with
    // its own cursors in their own 128-byte lines and its own power-of-two sub-range of the output arrays: slot =
    // (partition << log2 size) + offset. One shared cursor serialises in L2 at ~60 atomics/us - with one wave per run of
    // windows that alone would bound the replay.
    unsigned long long* cursors;  // [p * 32] group-slot cursor of partition p, [p * 32 + 8] / [p * 32 + 12] / [p * 32 + 20] length of its K3 lists A / B / C, [p * 32 + 16] record-slot
                                  // cursor, [p * 32 + 24] somatic: ids hashed by K3's workgroups & 63 == p (statistics); normal: length of K3n's wanted list p
    uint32_t group_part_log2, rec_part_log2;
    uint64_t group_cap, rec_cap;  // NPART << log2
    uint32_t* err;                // sticky error word (WD_* bits)
    // K3 output
    GroupSum* gsum;
    uint8_t* recs;                // HapRecHdr + seq[seq_cap] + germ[seq_cap], rec_stride bytes apart
    uint32_t seq_cap, rec_stride;
    uint32_t* tx_first_stop;      // per transcript: smallest window index with a main-ORF stop (0xFFFFFFFF none)
};

// rows_per_lane in {1,2,4,8,16}. All launches are asynchronous on `stream`.
void launch_k0_pack_admission(const DeviceBatch& d, hipStream_t stream); // once per batch, after the upload: fills exons_a and adm_map
void launch_k0_read_variants(const DeviceBatch& d, hipStream_t stream);  // once per batch, after the upload: fills r_var
void launch_k0_pack_windows(const DeviceBatch& d, hipStream_t stream);   // once per batch, after the upload: fills win_blobs
void launch_k1_pileup_bits(const DeviceBatch& d, hipStream_t stream);
void launch_k2_window_replay(const DeviceBatch& d, int rows_per_lane, hipStream_t stream);
void launch_k2_admission(const DeviceBatch& d, hipStream_t stream);     // K2a over the ExonW part of the plan
void launch_k2_window_rows(const DeviceBatch& d, hipStream_t stream);   // K2w, after K2a
// K2l, after K2a: the windows of winw; the <= 6-column, the 7-8-column and the 9-16-column (hash table) launch are independent and may go to different streams
void launch_k2_window_lanes(const DeviceBatch& d, hipStream_t stream_small, hipStream_t stream_wide, hipStream_t stream_hash);
// K3 / K3b: one grid row per output allocator; a row reads its list's length from the allocator's cursor on the device, the host only
// passes an upper bound of the total that sizes the rows.
// somatic: three launches - list A (simple windows: sequences, records AND their SHA-1 ids), list B (simple windows: flags, carried records) and list C
// (windows that need the general sequence walk: everything) - independent, may go to three streams;
// normal: k3_window_seq_normal over list A on stream_a, ids by launch_k3b_haplotype_ids afterwards
void launch_k3_window_seq(const DeviceBatch& d, uint64_t max_list_a, uint64_t max_list_b, uint64_t max_list_c, uint64_t max_list_d,
                          hipStream_t stream_a, hipStream_t stream_b, hipStream_t stream_c, hipStream_t stream_d);
void launch_k3b_haplotype_ids(const DeviceBatch& d, uint64_t max_recs, hipStream_t stream);   // `microphaser normal` only

}  // namespace mp
