// Packed, device-ready form of a batch of genes ("the batch") and the static window schedule
// ("the plan") the HIP kernels replay. Plain-old-data structs in this header are shared
// verbatim between host code and kernels.hip.
//
// Layout in HBM (all struct-of-arrays, gene-major):
//   reads     : r_pos r_end r_lseq r_ncig r_cigoff r_seqoff r_dup   + cigar pool, seq pool (per read: low-quality bitmap + 4-bit bases)
//   variants  : v_pos v_info v_len v_insoff (forward = ascending pos, ALT order within a pos)
//               v_rev2fwd (transcription order of '-' strand genes -> forward index)   + ins pool
//   refseq    : bytes of [gene.start, gene.end+100) per gene, case preserved
//   plan      : TxDev per transcript, Step per nt-offset step, WinStatic per printing step,
//               win_cols = the variant columns of each printing window (not always the window's own
//               variants: the reference can leave stale columns behind, microphasing.rs:1159)
//   K1 output : r_varlo, r_sup[W], r_lq[W]            (read x variant predicate bits)
//   K2 output : WinDyn per printing step, Group per distinct (haplotype, frame) key
//   K3 output : GroupSum per group, HapRec (header + seq + germ) per group that can be emitted / merged
#pragma once
#include <cstdint>

namespace mp {

// v_info bit layout
enum : uint32_t {
    VI_KIND_MASK = 0x3,        // VarKind
    VI_GERMLINE = 1u << 2,
    VI_FS_SHIFT = 3,           // frameshift() in {0,1,2}
    VI_FS_MASK = 0x3u << 3,
    VI_ALT_SHIFT = 8,          // SNV alt byte
};

struct TxDev {           // one per coding transcript
    uint32_t gene;       // batch gene index
    uint32_t step_off;   // first Step
    uint32_t n_steps;
    uint32_t strand;     // 0 forward, 1 reverse
    uint32_t sl_lo, sl_hi;   // start-loss position interval [lo, hi) (empty if lo >= hi)
    uint32_t id_off, id_len; // transcript id bytes in the string pool (hashed into haplotype ids)
};

// A transcript is replayed as one or more independent segments (whole exons): a new segment starts at the first window
// of an exon when the planner can prove that no row or pending candidate of the matrix survives into it; the columns that
// do survive (the reference can leave stale columns behind, :1159) are handed to the segment as its initial deque. The
// replay kernels give one wave to each segment instead of one to each transcript (shorter critical path).
struct SegDev {
    uint32_t tx;         // TxDev index
    uint32_t step_off;   // first Step of the segment
    uint32_t n_steps;
    uint32_t init_cols;  // live columns before the first step: transcription-order indices [x - init_cols, x), x = col_hi - n_add of that step
};

// Window-parallel replay (somatic mode, K2a + K2w): an independent single-exon segment whose columns are all SNVs at
// consecutive variant indices, with at most 64 candidate reads per window and W = 1, needs no sequential state machine:
// a read's admission step follows from its K1 masks and the plan (K2a), and the rows of a window are then a closed form
// of (admission step, read span, masks, live column range) - one wave per run of windows (K2w).
struct ExonW {
    uint32_t tx;
    uint32_t step_off, n_steps;
    uint32_t read_lo, n_reads;      // gene-relative range of the reads that can be candidates in this exon
    uint32_t adm_off;               // first AdmEntry of the exon (one per read of the range)
    uint32_t first_key_lo;          // '+': lowest start key of the first window's candidate range (:1229-1248)
    uint32_t range;                 // '-': candidate key range extent R - candidates have start in [sso - R, sso] (:1198-1226)
    uint32_t tr0, f0;               // transcription-order index -> forward index: f = strand ? f0 - (tr - tr0) : tr
    uint32_t sl_f_lo, sl_f_hi;      // forward index range of the variants inside the start-loss interval
    // flattened copies (one hop instead of three for every wave that starts on this exon)
    uint32_t strand;                // TxDev::strand
    uint32_t rbase, vbase;          // batch index of the gene's first read / first variant
    uint32_t sso0, sso1;            // splice_side_offset of the exon's first and second step
    uint32_t unit_steps;            // steps [0, unit_steps) move by exactly one nt each: '+' sso(t) = sso1 + (t - 1) for t >= 1, '-' sso(t) = sso0 - t -
                                    // K2a then finds a read's first candidate step by arithmetic alone, without looking at the steps
    uint32_t consumers;             // EW_*: which kernels read K2a's outputs for this exon (it writes only what somebody reads)
    uint32_t wlen_min;              // the shortest window (Step::wlen) of the exon's steps: on '-' K2a skips, by arithmetic, the steps at which even such a
                                    // window would stick out beyond the read's end (a read becomes a candidate R steps before a window can lie inside it)
};
enum : uint32_t { EW_WAVE = 1,      // a wave-per-window kernel has a window here: AdmEntry
                  EW_LANE = 2 };    // the lane-per-window kernel has one: RowRec
// What K2a needs of an ExonW, in four 16-byte loads (packed on the device at upload, k0_pack_admission); the flat form of K2a gives every
// lane one (exon, read) entry of the admission table, whatever exon it belongs to, so the exon's fields are per-lane values there.
struct ExonA {
    uint32_t step_off, n_steps, unit_steps, sso0;
    uint32_t sso1, first_key_lo, range, tr0;
    uint32_t f0, sl_f_lo, sl_f_hi, flags;      // flags: strand | consumers << 8
    uint32_t adm_off, read0, wlen_min, pad1;   // read0: batch index of the first read of the exon's range
};
static_assert(sizeof(ExonA) == 64, "ExonA layout");
struct AdmMap { uint32_t exon, read; };        // per admission-table entry: its exon and the batch index of its read

struct WChunk {                     // K2w work item: a run of steps of one ExonW
    uint32_t exon, step_first, n_steps, pad;
};
struct AdmEntry {                   // K2a output per (ExonW, read)
    uint32_t ord;                   // step (index into steps) at which push_read inserts the read; 0xFFFFFFFF = never
    uint32_t seen_lo;               // transcription-order index of the oldest column the row has seen (deque at that push)
};

// Lane-per-window replay (K2l): a printing window of a window-parallel exon with at most K2L_MAX_COLS columns and at most
// K2L_MAX_ROWS candidate reads (one mask word per read) is one LANE's work: the lane walks the window's candidate reads
// (RowRec, written by K2a per (exon, read)), derives row / bad / haplotype with three compares and two shifts and counts the
// haplotypes in its own column of an LDS table of 8-bit counters indexed by the haplotype word - no ballots, no readlanes, no
// barriers. Everything a lane needs is flattened into its WinW by the planner.
constexpr uint32_t K2L_MAX_COLS = 8;
constexpr uint32_t K2L_SMALL_COLS = 6;    // windows with <= 6 columns: 64 counters per lane (4 KB of LDS per wave); 7-8: 256 (16 KB)
constexpr uint32_t K2L_MAX_ROWS = 255;    // 8-bit counters
constexpr uint32_t K2L_HASH_COLS = 16;    // windows with 9..16 columns: a 64-slot hash table per lane (entry = 16-bit haplotype word << 8 | count) ...
constexpr uint32_t K2L_HASH_ROWS = 63;    // ... which must keep a free slot: at most 63 candidate reads (so at most 63 distinct words)
// does the lane-per-window kernel take a printing window with this many columns / candidate reads? (planner and wave kernels agree on it)
#if defined(__HIPCC__) || defined(__CUDACC__)
__host__ __device__
#endif
inline bool k2l_takes(uint32_t ncols, uint32_t rn, bool hash_form) {
    return (ncols <= K2L_MAX_COLS && rn <= K2L_MAX_ROWS) || (hash_form && ncols <= K2L_HASH_COLS && rn <= K2L_HASH_ROWS);
}
struct WinW {
    uint32_t rr_lo;      // RowRec index of the window's first candidate read
    uint32_t pack;       // r_n (bits 0-9) | ncols (10-15) | WW_FWD | WW_NEED_ALL
    uint32_t wkey;       // a read is a row iff RowRec.key >= wkey: '+' splice_end vs end, '-' ~sso vs ~start (:259-278, :312-315)
    uint32_t step;       // ... and it was inserted at or before this step (RowRec.ord <= step)
    uint32_t col_hi;     // the row is not sticky-bad iff col_hi <= RowRec.bad_from
    uint32_t flo;        // gene-relative forward index of the window's lowest-position column
    uint32_t som_lo, som_hi;   // somatic columns of the window in haplotype bit order
};
static_assert(sizeof(WinW) == 32, "WinW layout");
enum : uint32_t { WW_FWD = 1u << 16, WW_NEED_ALL = 1u << 17,
                  WW_TRIVIAL = 1u << 18,     // WSF_SIMPLE && WSF_NOSTOP and no record demand of its own: a group without a somatic column is settled by K2l
                  WW_SIMPLE = 1u << 20,      // WSF_SIMPLE && WSF_NOSTOP: K3 builds the window's sequences by byte substitution and needs no stop scan; the groups of
                                             // other windows go to K3's list C (one such lane would make its whole K3 wave run the per-base walk / the codon loop)
                  WW_WALK = 1u << 21,        // !WSF_SIMPLE: the window's sequences need the general walk - its groups go to K3's list D (the other windows that are
                                             // not WW_SIMPLE - simple, but a stop codon is possible - to list C)
                  WW_ALL_IDS = 1u << 19 };   // WS_ALL_IDS: every haplotype of the window gets an id (indel / frameshift context); without it only a
                                             // haplotype that sets a somatic column is hashed - the others go to K3's list B even when they need a record
struct RowRecA {         // K2a output per (ExonW, read), first half (the second is the 64-bit support mask)
    uint32_t key;        // '+': end_pos, '-': ~start
    uint32_t ord;        // AdmEntry::ord
    uint32_t bad_from;   // transcription-order index of the first column, among those the row sees after its insertion, on which it has a
                         // low-quality / start-loss bit (0xFFFFFFFF: none): the row is bad from the step that appends it (:157-197)
    uint32_t rvl;        // r_varlo: forward index of the variant bit 0 of the masks belongs to
};
static_assert(sizeof(RowRecA) == 16, "RowRecA layout");

// Step.flags
enum : uint8_t {
    SF_PRINT = 1,        // print_haplotypes may be called at this step (superset)
    SF_FULL_RANGE = 2,   // candidates are a full key range (first window of an exon): reverse strand re-scans
    SF_FIRST_EXON_WIN = 4,
    SF_LAST_EXON_WIN = 8,
    SF_SHORT_EXON = 16,
    SF_FIRST_EXON = 32,
    SF_LAST_EXON = 64,
    SF_NEED_RECS = 128,  // K2 reserves a HapRec slot for every haplotype of this window (WinStatic.need_recs != 0)
};

struct Step {            // one per nt-offset step of the window scheduler (reference: microphasing.rs:1030-1914)
    uint32_t sso;        // splice_side_offset (absolute, 0-based)
    uint32_t cand_lo;    // gene-relative index of the first NEW candidate read
    uint32_t col_hi;     // transcription-order variant index one past the newest column after this step
    uint32_t win;        // window index if SF_PRINT else 0xFFFFFFFF
    uint16_t cand_n;     // number of new candidate reads
    uint8_t wlen;        // splice_end - sso
    uint8_t n_del;       // columns dropped at this step (incl. the exon-start shrink of last_window_vars)
    uint8_t n_add;       // columns appended at this step
    uint8_t flags;       // SF_*
    uint8_t splice_pos;  // 0,1,2
    uint8_t splice_gap;
    uint32_t exon;       // index into the host-side ExonPlan array
};
static_assert(sizeof(Step) == 28, "Step layout");

struct WinStatic {       // one per printing step; everything K3 needs that does not depend on reads (flattened: one hop)
    uint32_t tx;
    uint32_t sso;
    uint32_t col_off;    // first entry of this window's column list in win_cols (oldest column first)
    uint32_t ref_off;    // index into ref_pool of the reference base at sso
    uint32_t vbase;      // gene's first variant in the v_* arrays
    uint16_t ncols;
    uint8_t wlen;
    uint8_t ewl;         // exon_window_len passed to print_haplotypes as window_len
    uint8_t splice_pos;
    uint8_t splice_gap;
    uint8_t flags;       // SF_* of the step | WSF_REVERSE
    uint8_t need_recs;   // bits 0-1 WS_* : K3 writes a HapRec for EVERY haplotype of this window; bits 2-7: simple-walk prefix length
    uint32_t step;       // the Step this window belongs to
};
enum : uint8_t {
    WS_ALL_IDS = 1,      // indel / frameshift context: every haplotype can be emitted -> records + SHA-1 ids
    WS_CARRY = 2,        // the window's haplotypes are carried into a splice-side merge -> records
};
constexpr uint8_t WS_MASK = 3;
constexpr uint8_t WS_PREFIX_SHIFT = 2;  // need_recs >> 2 = number of leading walk-order columns the sequence walk can ever visit (WSF_SIMPLE)
constexpr uint8_t WSF_REVERSE = 128;  // transcript on the '-' strand (SF_* use bits 0..6)
constexpr uint8_t WSF_NOSTOP = 2;     // (replaces SF_FULL_RANGE) simple window over an all-upper-case reference without a stop codon in the
                                      // peptide frame: a variant base is lower-case there and can never complete a stop (:42-76) -> no scan
constexpr uint8_t WSF_SIMPLE = 1;     // (replaces SF_PRINT, always set for a window) wlen <= 32 and the walk of :473-601 can only ever visit a
                                      // prefix of the columns (walk order), all SNVs at strictly increasing positions inside the window - the
                                      // next column, if any, lies behind the cursor or beyond the window (a stale column, :1159) and blocks
                                      // everything after it: K3 builds the sequences by byte substitution instead of walking. Also simple:
                                      // a prefix that ends at the window's last base followed by SNVs at window_end, window_end + 1, ... without
                                      // a gap (the walk's inner loop has no window bound, :479: a set SNV at the last base pulls that run in, one
                                      // more base per applied column) - such a window is never WSF_NOSTOP (plan.cpp)
// What K3 reads per listed haplotype, in ONE 128-byte record per window (built on the device once per batch by k0_pack_windows from wins, the
// reference bytes, win_cols and the transcripts: a layout transformation, not part of the pass). K3 was bound by the number of scattered
// loads and their dependent levels - entry -> window record -> reference bytes / columns / transcript -> id text; with the record the
// second and third level are eight 16-byte loads of two adjacent cache lines.
struct WinBlob {
    WinStatic ws;
    uint32_t ref[10];    // the 40 bytes from the dword at or before the window's first reference base (ws.ref_off & ~3)
    uint32_t cpi[12];    // (pos, info) of the window's first six columns, deque order (the others: win_cols)
    uint32_t id_off, id_len;   // the transcript's id in str_pool
};
static_assert(sizeof(WinBlob) == 128, "WinBlob layout");
constexpr uint32_t WINBLOB_COLS = 6;

struct WinCol {          // one live variant column of a printing window
    uint32_t f;          // gene-relative forward variant index
    uint32_t pos;        // v_pos
    uint32_t info;       // v_info
};
static_assert(sizeof(WinStatic) == 32, "WinStatic layout");

struct WinDyn {          // K2 output per printing step
    uint32_t group_off;  // first Group
    uint32_t ngroups;
    uint32_t nrows;      // ObservationMatrix::nrows() (depth column)
    uint32_t flags;      // WD_*
};
enum : uint32_t { WD_DONE = 1, WD_ROW_OVERFLOW = 2, WD_GROUP_OVERFLOW = 4, WD_REC_OVERFLOW = 8,
                  WD_EPOCH_OVERFLOW = 16, WD_HAP_OVERFLOW = 32,    // normal mode: column-epoch ring / haplotype table full
                  WD_INTERNAL = 64 };                              // a kernel met something the planner promised it would not

struct Group {           // K2 output: one distinct (haplotype, frame.0, frame.1 != 0) key of one window, ascending
    uint64_t hap;
    uint32_t count;
    uint32_t aux;        // frame0 << 1 | (frame1 != 0); GROUP_SETTLED: see below
};
// K2l settles a group of a simple window that cannot hold a stop and that carries no somatic column itself: what K3 would find for it is
// GroupSum {GS_VALID, no record}. It says so in the group (no GroupSum is written, K3 never sees the group).
constexpr uint32_t GROUP_SETTLED = 1u << 31;

// GroupSum.flags
enum : uint32_t {
    GS_VALID = 1,
    GS_STOP = 2,         // has_stop_codon(neopeptide)
    GS_DIFFERS = 4,      // germline_seq != seq (before any clearing)
    GS_INDEL = 8,
    GS_INSERTION = 16,
    GS_BROKE = 32,       // reverse-strand incomplete deletion hit (variant index = prof_len)
    GS_HAS_REC = 64,     // a HapRec was written (rec index in GroupSum.rec)
    GS_ID_VALID = 128,   // HapRec.id holds the SHA-1 prefix
};
struct GroupSum {        // K3 output per group
    uint32_t flags;
    uint32_t rec;        // HapRec index if GS_HAS_REC
};

// K3 output for groups whose sequence the host needs: a 32-byte header followed by seq[cap] and germ[cap], where
// cap (the batch's sequence capacity) is 48, 112 or 240 bytes - the smallest that holds the longest sequence any
// window of the batch can build (window + inserted bases + bases restored by somatic deletions).
struct HapRecHdr {
    uint64_t prof_set;   // bit c set <=> variant_profile[c] != 0 (c < prof_len)
    uint64_t id60;       // first 60 bits of the SHA-1 (15 hex digits), big-endian in the low 60 bits
    uint8_t seq_len, germ_len, prof_len, nvar, nsom, first_fs, first_fs_j, pad;
    uint32_t win;        // window index (K3b hashes transcript id + offset of that window)
    uint32_t want_id;    // 1: K3b computes id60
};
static_assert(sizeof(HapRecHdr) == 32, "HapRecHdr layout");
constexpr uint32_t SEQ_CAPS[4] = {32, 48, 112, 240};   // 32: a 27-nt window without indels -> 96-byte records
constexpr uint32_t SEQ_CAP_MAX = SEQ_CAPS[3];
inline uint32_t seq_cap_for(uint64_t max_len) { for (uint32_t c : SEQ_CAPS) if (max_len <= c) return c; return SEQ_CAP_MAX; }
inline uint32_t hap_rec_stride(uint32_t cap) { return 32u + 2u * cap; }

}  // namespace mp
