// gfx950 (CDNA4, wave64) kernels of the phasing hot path. Integer / bitset work, HBM- and
// latency-bound: no MFMA. One wavefront = 64 lanes everywhere (hard-coded).
//
//   K1 k1_pileup_bits     read x variant predicates  (reference: supports_variant / bad_quality,
//                         src/microphasing.rs:78-139) -> per-read support / low-quality bit masks
//   K2 k2_window_replay   ObservationMatrix sliding-window state machine, one wave per transcript,
//                         rows (reads) live in lanes, columns (variants) in an LDS ring
//                         (reference: src/microphasing.rs:220-343 + count phase :383-411)
//   K3 k3_window_seq      per (window, haplotype): mutant / germline window sequence, variant
//                         profile, stop scan, SHA-1 id (reference: src/microphasing.rs:434-603,
//                         :42-76, :667-675)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "kernels.hpp"

namespace mp {

// This file is compiled in parts (Makefile: -DMP_KPART=0..6, one object each, side by side): every part holds all the templates and
// the launchers - hence the kernel instantiations - of its share. Without MP_KPART it is one translation unit.
#ifndef MP_KPART
#define MP_KPART -1
#endif
#define MP_IN_PART(n) (MP_KPART == -1 || MP_KPART == (n))
#define HIP_CHECK_LAUNCH() \
    do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { throw_hip(e_, __FILE__, __LINE__); } } while (0)

[[noreturn]] void throw_hip(hipError_t e, const char* file, int line);

// ====================================================================== K0 (layout, once per batch)
#if MP_IN_PART(0)
// per read, for K1: the absolute index of the first variant at / after the read's start and the number of the gene's variants from
// there on - one load instead of the hops read -> gene -> variant range (was a serial host loop over all reads at upload: ~100 ms)
__global__ __launch_bounds__(256) void k0_read_variants(DeviceBatch d) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= d.n_reads) return;
    uint32_t lo = 0, hi = d.n_genes;   // g_read_off[lo] <= i < g_read_off[hi] (offsets ascend; genes without reads share their successor's)
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (d.g_read_off[mid] <= i) lo = mid; else hi = mid;
    }
    const uint32_t vb = d.g_var_off[lo], nv = d.g_var_off[lo + 1] - vb, vl = d.r_varlo[i];
    d.r_var[i] = make_uint2(vb + vl, nv - vl);
}
void launch_k0_read_variants(const DeviceBatch& d, hipStream_t stream) {
    if (!d.n_reads || !d.n_genes) return;
    hipLaunchKernelGGL(k0_read_variants, dim3((d.n_reads + 255) / 256), dim3(256), 0, stream, d);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) throw_hip(e_, __FILE__, __LINE__);
}
__global__ __launch_bounds__(256) void k0_pack_windows(DeviceBatch d) {
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    if (w >= d.n_wins) return;
    WinBlob B;
    B.ws = d.wins[w];
    const uint32_t* src = reinterpret_cast<const uint32_t*>(d.ref_pool + (B.ws.ref_off & ~3u));   // (the pool is padded and 256-byte aligned)
#pragma unroll
    for (int k = 0; k < 10; k++) B.ref[k] = src[k];
    const uint32_t ncols = B.ws.ncols;
#pragma unroll
    for (uint32_t k = 0; k < WINBLOB_COLS; k++) {
        const WinCol c = d.win_cols[B.ws.col_off + (k < ncols ? k : 0u)];
        B.cpi[2 * k] = k < ncols ? c.pos : 0xFFFFFFFFu;
        B.cpi[2 * k + 1] = k < ncols ? c.info : 0u;
    }
    const TxDev T = d.tx[B.ws.tx];
    B.id_off = T.id_off; B.id_len = T.id_len;
    d.win_blobs[w] = B;
}
void launch_k0_pack_windows(const DeviceBatch& d, hipStream_t stream) {
    if (!d.n_wins || !d.win_blobs) return;
    hipLaunchKernelGGL(k0_pack_windows, dim3((d.n_wins + 255) / 256), dim3(256), 0, stream, d);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) throw_hip(e_, __FILE__, __LINE__);
}
#endif

// ====================================================================== K1
// rust-htslib CigarStringView::read_pos(ref_pos, false, false), see model.hpp cigar_read_pos.
__device__ __attribute__((noinline)) int cigar_read_pos_dev(const uint32_t* cig, uint32_t ncig, uint32_t read_start, uint32_t ref_pos) {
    int64_t rpos = read_start;
    int64_t qpos = 0;
    uint32_t j = 0;
    bool found = false;
    for (uint32_t i = 0; i < ncig; i++) {
        uint32_t op = cig[i] & 0xF;
        if (op == 0 || op == 7 || op == 8 || op == 1 || op == 4) { j = i; found = true; break; }
        if (op == 2 || op == 3) return -1;
        if (op == 5 && i > 0 && i + 1 < ncig) return -1;
        if (i + 1 == ncig) return -1;
    }
    if (!found) return -1;
    while (rpos <= int64_t(ref_pos) && j < ncig) {
        uint32_t op = cig[j] & 0xF;
        int64_t l = cig[j] >> 4;
        if (op == 0 || op == 7 || op == 8) {
            if (rpos + l > int64_t(ref_pos)) return int(qpos + (int64_t(ref_pos) - rpos));
            rpos += l; qpos += l; j++;
        } else if (op == 4 || op == 1) { qpos += l; j++; }
        else if (op == 2 || op == 3) { rpos += l; j++; }
        else if (op == 6) { j++; }
        else return -1;
    }
    return -1;
}

// K1_LANES lanes per read. The first variant at or after the read start (r_varlo) comes from the planner; the variants the
// read can see are then a short contiguous run of the gene's variant array. Lane k of the group takes variants k, k + K1_LANES,
// ... of that run, so more of a read's ~20 (variant position -> base, quality) gathers are in flight at once than in one
// thread's chain of dependent loads; a K1_LANES-bit slice of a wave ballot IS the next K1_LANES bits of the read's mask.
__device__ __forceinline__ uint8_t decode_base4(uint32_t code) {  // BAM 4-bit code -> "=ACMGRSVTWYHKDBN"
    const uint64_t lo = 0x565352474d43413dull, hi = 0x4e42444b48595754ull;
    return uint8_t(((code & 8) ? hi : lo) >> (8 * (code & 7)));
}

constexpr uint32_t K1_LANES = 4;    // lanes per read (measured at config C: 1 lane 1.40 ms, 2: 1.07, 4: 1.10, 8: 1.31, 16: 1.71)
static_assert(K1_LANES == 4, "the group's lanes combine their bits with quad permutes");
constexpr uint32_t K1_PAY_DW = 32;      // dwords of a read's payload (low-quality bitmap + packed bases) staged in LDS: reads of up to ~200 nt whole
constexpr uint32_t K1_PAY_STRIDE = 33;  // odd dword stride: the groups' slots fall into different banks
struct __attribute__((packed, aligned(4))) K1Quad { uint32_t x, y, z, w; };   // four consecutive dwords at a dword-aligned address (one 16-byte load)

// K1 is bound by the NUMBER of scattered memory instructions, not by bytes or arithmetic (a slimmer instruction stream did not move
// its time): the round-based form issued 7 + 12 + 12 narrow gathers per wave in three dependent levels (read fields -> variant positions
// -> base / quality). Now a group of four lanes (one read) fetches
//   - the read's payload - low-quality bitmap and packed bases, 68 bytes for a 101-nt read - with two 16-byte loads per lane into an LDS
//     slot (every base / quality look-up is then an LDS read), and
//   - its next 32 variants - positions and info words - with four 16-byte loads per lane (lane k of the group takes variants 4 k .. 4 k + 3
//     and 16 + 4 k .. 16 + 4 k + 3),
// both straight after the read's own fields: two dependent levels, 13 wide loads per wave. A batch of 32 variants is evaluated branch-free
// for the common case (SNV under a single-M CIGAR); each lane collects coverage / support / low-quality bits of its eight variants in
// three words at the variants' own bit positions, the rare other cases (indels, SNVs under a CIGAR with clips / indels) are marked and
// settled in a loop of their own, and the group's four lanes OR their words with two quad permutes each: the batch's 32 mask bits.
template <int W>
__global__ __launch_bounds__(256) void k1_pileup_bits(DeviceBatch d) {
    __shared__ uint32_t pay[(256 / K1_LANES) * K1_PAY_STRIDE];
    const uint64_t t = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    const uint32_t sub = threadIdx.x & (K1_LANES - 1);             // lane within the read's group
    const uint64_t i_raw = t / K1_LANES;
    const bool valid = i_raw < d.n_reads;
    const uint32_t i = valid ? uint32_t(i_raw) : d.n_reads - 1;              // surplus groups shadow the last read (no stores)
    const uint2 rv = d.r_var[i];          // first variant at / after the read's start (absolute index), variants left in the gene
    const uint32_t vfirst = rv.x;
    const uint32_t rpos = d.r_pos[i], rend = d.r_end[i], lseq = d.r_lseq[i], ncig = d.r_ncig[i];
    const uint32_t* cig = d.cigar_pool + d.r_cigoff[i];
    const uint32_t* lowq = reinterpret_cast<const uint32_t*>(d.seq_pool + d.r_seqoff[i]);   // bit k: base quality at read offset k below 10
    const uint32_t nlq = (lseq + 31) >> 5;                                                  // dwords of the bitmap; the packed bases follow
    const uint8_t* seq4 = reinterpret_cast<const uint8_t*>(lowq + nlq);
    // a variant can be a (stale) column of a window the read encloses without lying inside the read's aligned span;
    // bad_quality still indexes the qualities by reference offset (:82-88): cover max(end, start + l_seq)
    const uint32_t cover_end = max(rend, rpos + lseq);
    const uint32_t maxn = min(rv.y, 64u * W);
    // ---- second level, all issued together: the payload (-> LDS), the first CIGAR operation, the first batch of variants
    uint32_t* const slot = pay + (threadIdx.x / K1_LANES) * K1_PAY_STRIDE;
    const K1Quad pa = reinterpret_cast<const K1Quad*>(lowq)[sub], pb = reinterpret_cast<const K1Quad*>(lowq)[4 + sub];   // (the pools are padded)
    const uint32_t c0 = ncig > 0 ? cig[0] : 0;
    auto load_variants = [&](uint32_t b0, K1Quad (&vp)[2], K1Quad (&vi)[2]) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t at = vfirst + b0 + 16u * uint32_t(h) + 4u * sub;
            vp[h] = *reinterpret_cast<const K1Quad*>(d.v_pos + at);
            vi[h] = *reinterpret_cast<const K1Quad*>(d.v_info + at);
        }
    };
    K1Quad vp[2], vi[2];
    load_variants(0, vp, vi);
    slot[4 * sub] = pa.x; slot[4 * sub + 1] = pa.y; slot[4 * sub + 2] = pa.z; slot[4 * sub + 3] = pa.w;
    slot[16 + 4 * sub] = pb.x; slot[16 + 4 * sub + 1] = pb.y; slot[16 + 4 * sub + 2] = pb.z; slot[16 + 4 * sub + 3] = pb.w;
    __syncthreads();   // (a group's four lanes sit in one wave; the barrier only orders the LDS writes before the reads)
    const bool simple = ncig == 1 && (c0 & 0xF) == 0;  // a single M op: read_pos(p) = p - start
    const uint32_t len0 = c0 >> 4;
    // (a wave that holds a read whose payload does not fit the slot - longer than ~200 nt - reads the payloads from memory instead)
    const bool any_long = __ballot(nlq + ((lseq + 7) >> 3) > K1_PAY_DW) != 0;
    uint64_t sup[W], lq[W];
#pragma unroll
    for (int w = 0; w < W; w++) { sup[w] = 0; lq[w] = 0; }
    uint32_t ncov = 0;
    bool more = true;   // the group's run may continue (every variant of the previous batch was covered)
    const uint32_t last_q = lseq ? lseq - 1 : 0;
    auto run_batches = [&](auto from_memory) {   // from_memory: std::true_type for a wave with a read too long for its LDS slot
    for (uint32_t b0 = 0; b0 < 64u * W; b0 += 32) {
        if (__ballot(more) == 0) break;   // wave-uniform
        if (b0) load_variants(b0, vp, vi);
        const uint32_t vpos[8] = {vp[0].x, vp[0].y, vp[0].z, vp[0].w, vp[1].x, vp[1].y, vp[1].z, vp[1].w};
        const uint32_t info[8] = {vi[0].x, vi[0].y, vi[0].z, vi[0].w, vi[1].x, vi[1].y, vi[1].z, vi[1].w};
        uint32_t acc_c = 0, acc_s = 0, acc_q = 0, slow = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t tb = (k < 4 ? 0u : 16u) + 4u * sub + (k & 3u);   // the variant's place in the batch = its bit in the three words
            const bool in = more && b0 + tb < maxn;
            const bool cov = in && vpos[k] < cover_end;
            const bool snv = cov && (info[k] & VI_KIND_MASK) == 0;
            const uint32_t rel = vpos[k] - rpos;
            const uint32_t relc = min(rel, last_q);
            uint32_t qw, sw;
            if constexpr (!decltype(from_memory)::value) { qw = slot[relc >> 5]; sw = slot[nlq + (relc >> 3)]; }
            else { qw = lowq[relc >> 5]; sw = lowq[nlq + (relc >> 3)]; }
            // the reference indexes the qualities by reference offset (:82-88); `normal` has no quality gate (src/normal_microphasing.rs:43-52)
            const bool q = snv && !d.normal && rel < lseq && ((qw >> (rel & 31u)) & 1u);
            const bool fast = snv && simple && rel < len0 && rel < lseq;   // a single M op: read_pos(p) = p - start
            const uint32_t byte = (sw >> (8u * ((relc >> 1) & 3u))) & 0xFFu;
            const uint32_t code = (rel & 1) ? (byte & 0xFu) : (byte >> 4);
            const bool sp = fast && !q && decode_base4(code) == uint8_t(info[k] >> VI_ALT_SHIFT);   // SNV (:97-112, :80-92)
            const uint32_t bit = 1u << tb;
            acc_c |= cov ? bit : 0u;
            acc_s |= sp ? bit : 0u;
            acc_q |= q ? bit : 0u;
            slow |= (cov && ((snv && !simple && !q) || !snv)) ? (1u << k) : 0u;
        }
        if (__ballot(slow != 0)) {   // insertions / deletions, SNVs under a CIGAR with more than one M: rare, settled apart
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                if (!((slow >> k) & 1u)) continue;
                const uint32_t tb = (k < 4 ? 0u : 16u) + 4u * sub + (k & 3u);
                bool sp = false;
                if ((info[k] & VI_KIND_MASK) == 0) {
                    const int p = cigar_read_pos_dev(cig, ncig, rpos, vpos[k]);
                    if (p >= 0 && uint32_t(p) < lseq) {
                        const uint8_t byte = seq4[p >> 1];
                        const uint32_t code = (p & 1) ? (byte & 0xFu) : (uint32_t(byte) >> 4);
                        sp = decode_base4(code) == uint8_t(info[k] >> VI_ALT_SHIFT);
                    }
                } else {  // insertion / deletion: any I / D op of exactly that length (:113-137)
                    const uint32_t want = (info[k] & VI_KIND_MASK) == 1 ? 1u : 2u;
                    const uint32_t vlen = d.v_len[vfirst + b0 + tb];
                    for (uint32_t c = 0; c < ncig; c++)
                        if ((cig[c] & 0xF) == want && (cig[c] >> 4) == vlen) { sp = true; break; }
                }
                acc_s |= sp ? (1u << tb) : 0u;
            }
        }
        // the four lanes of the group OR their words (quad permutes: lane ^ 1, then lane ^ 2)
        auto quad_or = [](uint32_t v) {
            v |= uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1, 0, 3, 2]
            v |= uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2, 3, 0, 1]
            return v;
        };
        acc_c = quad_or(acc_c); acc_s = quad_or(acc_s); acc_q = quad_or(acc_q);
        ncov += __popc(acc_c);
#pragma unroll
        for (int w = 0; w < W; w++)
            if ((b0 >> 6) == uint32_t(w)) { sup[w] |= uint64_t(acc_s) << (b0 & 63u); lq[w] |= uint64_t(acc_q) << (b0 & 63u); }
        more = more && acc_c == 0xFFFFFFFFu;
    }
    };
    if (!any_long) run_batches(std::false_type{});
    else run_batches(std::true_type{});
    if (valid && sub == 0) {
        d.r_ncov[i] = ncov;
#pragma unroll
        for (int w = 0; w < W; w++) {
            d.r_sup[uint64_t(i) * W + w] = sup[w];
            d.r_lq[uint64_t(i) * W + w] = lq[w];
        }
    }
}

// ====================================================================== K2
enum : uint32_t { ST_EMPTY = 0, ST_PENDING = 1, ST_ROW = 2, ST_MASK = 3, RF_BAD = 4, RF_SL = 8, RF_F1 = 16 };

__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ uint32_t lanes_below(uint64_t mask, uint32_t lane) { return __popcll(mask & ((1ull << lane) - 1ull)); }

constexpr uint32_t REC_CHUNK = 128;     // HapRec slots per allocation (>= 64: one emit call)

// K2 -> K3: the group slots K3 has to look at are appended to two dense lists per output allocator, both inside the allocator's
// sub-range of k3_items (one 16-byte item {group slot, window, record slot, 0} per listed group):
//   list A - groups whose haplotype id K3 hashes: a somatic column is set, or every haplotype of the window gets an id (indel / frameshift
//            context) - the lane kernel knows which; the wave kernels send every group with a record slot here. K3 builds their sequences,
//            hashes their ids and writes their records; grows upwards from the sub-range's first entry, length in cursors[p * 32 + 8]
//   list B - the other listed groups: stop / differs / indel flags, and the record of a group with a slot that needs no id (a window
//            carried into a splice merge); grows downwards from the sub-range's last entry, length in cursors[p * 32 + 12]
// so the lanes of a K3 wave all do the same kind of work (the SHA-1 of the ids is two thirds of K3's instructions; in one mixed list
// 40 % of its lanes would idle through it). The lists cannot meet while the groups fit their sub-range (one entry per group slot at
// most); one atomic instruction per call (two lanes, one per list). The lane-per-window kernel settles most groups itself (no somatic
// column set, simple window without a possible stop: GroupSum = {GS_VALID}) and lists only the rest.
__device__ __forceinline__ void k3_enqueue(const DeviceBatch& d, uint32_t part, bool on, uint64_t slot, uint32_t win, uint32_t rec) {
    // list C: the groups of windows whose sequences need the general walk (indel / multi-allelic columns, long windows) or a stop scan
    // (a stop codon is possible). One such lane makes its whole K3 wave run the per-base walk / the codon loop - at config C 0.8 % of the
    // windows did the former to 40 % of the waves, 6 % the latter to nearly all - so they get a list (and a launch) of their own, in the
    // second half of k3_items. (`normal` mode has one K3 kernel for everything: list A.)
    // The two kinds are kept apart as well: list C (upwards) = simple windows that may hold a stop, list D (downwards from the second
    // half's end) = windows that need the general walk - 11 % of what was one list made nearly all of its waves run the per-base walk.
    const uint32_t wbit = on ? d.win_simple[win >> 5] : 0u, kbit = on ? d.win_walk[win >> 5] : 0u;
    const bool simple = d.normal || ((wbit >> (win & 31u)) & 1u);
    const bool walk = !d.normal && ((kbit >> (win & 31u)) & 1u);
    const bool to_d = on && !simple && walk;
    const bool to_c = on && !simple && !walk;
    const bool to_a = on && simple && rec != 0xFFFFFFFFu;
    const bool to_b = on && simple && rec == 0xFFFFFFFFu;
    const uint64_t ma = __ballot(to_a), mb = __ballot(to_b), mc = __ballot(to_c), md = __ballot(to_d);
    if (!(ma | mb | mc | md)) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t la = ma ? uint32_t(__builtin_ctzll(ma)) : 64u, lb = mb ? uint32_t(__builtin_ctzll(mb)) : 64u, lc = mc ? uint32_t(__builtin_ctzll(mc)) : 64u,
                   ld = md ? uint32_t(__builtin_ctzll(md)) : 64u;
    unsigned long long base = 0;
    if (lane == la || lane == lb || lane == lc || lane == ld)   // one atomic instruction, a lane per list
        base = atomicAdd(d.cursors + part * 32 + (lane == la ? 8 : lane == lb ? 12 : lane == lc ? 20 : 28),
                         (unsigned long long)__popcll(lane == la ? ma : lane == lb ? mb : lane == lc ? mc : md));
    const uint32_t src = (to_a ? la : to_b ? lb : to_c ? lc : ld) & 63u;
    const uint64_t b0 = (uint64_t(uint32_t(__shfl(int(uint32_t(base >> 32)), int(src)))) << 32) | uint32_t(__shfl(int(uint32_t(base)), int(src)));
    // (the cursors also count the entries of waves whose groups did NOT fit their sub-range: never store outside the sub-range, flag
    //  the pass instead; it is run again with larger arenas and K3 never walks the lists of a flagged pass)
    if (on) {
        const uint64_t size = 1ull << d.group_part_log2;
        const uint64_t at = b0 + lanes_below(to_a ? ma : to_b ? mb : to_c ? mc : md, lane);
        const uint64_t sub = (uint64_t(part) << d.group_part_log2) + ((to_c || to_d) ? d.group_cap : 0ull);
        if (at < size) d.k3_items[sub + ((to_b || to_d) ? size - 1 - at : at)] = make_uint4(uint32_t(slot), win, rec, 0u);
        else atomicOr(d.err, WD_GROUP_OVERFLOW);
    }
}
// A pass whose group / record arenas overflowed is discarded and run again with larger ones (DeviceContext::run); its K3 lists then hold
// entries that were reserved but never written. K3 / K3b leave at once in that case (the error word is complete: the K2 kernels have ended).
__device__ __forceinline__ bool pass_overflowed(const DeviceBatch& d) {
    return (__builtin_nontemporal_load(d.err) & (WD_GROUP_OVERFLOW | WD_REC_OVERFLOW)) != 0;
}

template <int RPL>
__global__ __launch_bounds__(64) void k2_window_replay(DeviceBatch d) {
    constexpr uint32_t GROUP_CHUNK = RPL <= 4 ? 256u : 64u * RPL;  // group slots per allocation (>= rows of one window)
    const uint32_t lane = threadIdx.x;
    // this wave's output allocator (kernels.hpp NPART)
    const uint32_t part = blockIdx.x & (NPART - 1);
    unsigned long long* const gcur = d.cursors + part * 32;
    unsigned long long* const rcur = gcur + 16;
    const uint64_t gpart_lo = uint64_t(part) << d.group_part_log2, gpart_hi = uint64_t(part + 1) << d.group_part_log2;
    const uint64_t rpart_lo = uint64_t(part) << d.rec_part_log2, rpart_hi = uint64_t(part + 1) << d.rec_part_log2;
    const SegDev S = d.segs[d.seg_order[blockIdx.x]];
    const TxDev T = d.tx[S.tx];
    const uint32_t rbase = d.g_read_off[T.gene];
    const uint32_t vbase = d.g_var_off[T.gene];
    const bool is_rev = T.strand != 0;
    const uint32_t W = d.mask_words;

    __shared__ uint32_t colf[64];     // forward variant index (gene-relative) of each live column
    __shared__ uint32_t colinfo[64];  // v_info | start-loss bit 31

    uint32_t fl[RPL], rs[RPL], re[RPL], rvl[RPL], rcov[RPL], rdup[RPL], ridx[RPL], fr0[RPL], pver[RPL];
    uint64_t hap[RPL], msup[RPL], mlq[RPL];
#pragma unroll
    for (int r = 0; r < RPL; r++) { fl[r] = ST_EMPTY; rs[r] = re[r] = rvl[r] = rcov[r] = rdup[r] = ridx[r] = fr0[r] = pver[r] = 0; hap[r] = msup[r] = mlq[r] = 0; }

    uint32_t ncols = 0, head = 0;
    uint32_t colver = 1;  // bumped whenever the column set changes
    // Fast admission: while the live columns are a contiguous, monotone run of forward variant indices and none of them is
    // a frameshift / start-loss variant, a new row's haplotype is one shift+mask (+ bit reversal on '+') of its K1 mask.
    uint32_t n_special = 0;     // live columns with frameshift() > 0 or start-loss
    bool contig = true;
    uint32_t f_oldest = 0, f_newest = 0;
    uint64_t special_mask = 0;  // bit c (hap bit order) set <=> column c is special
    uint64_t chunk_pos = 0, chunk_end = 0;   // group slots (per-wave chunk allocator: one atomic per GROUP_CHUNK groups)
    uint64_t rec_pos = 0, rec_end = 0;       // haplotype-record slots, same scheme (K3 then needs no atomics at all)
    uint64_t som_mask = 0;                   // bit c set <=> live column c is a somatic variant (same bit order as hap)
    uint64_t n_groups_tx = 0;
    uint32_t sticky_err = 0;

    // support / low-quality bit of row slot r for forward variant index f
    auto bits_of = [&](int r, uint32_t f, uint32_t info, bool& s, bool& q) {
        uint32_t b = f - rvl[r];
        s = false; q = false;
        if (f >= rvl[r] && b < rcov[r]) {
            if (W == 1) { s = (msup[r] >> b) & 1; q = (mlq[r] >> b) & 1; }
            else {
                uint64_t a = d.r_sup[uint64_t(ridx[r]) * W + (b >> 6)], c = d.r_lq[uint64_t(ridx[r]) * W + (b >> 6)];
                s = (a >> (b & 63)) & 1; q = (c >> (b & 63)) & 1;
            }
            return;
        }
        // Outside what K1 evaluated (a stale column beyond the read, :1159): an SNV there is neither supported nor
        // low-quality, but indel support is position-independent - any I / D op of that length (:113-137).
        const uint32_t kind = info & VI_KIND_MASK;
        if (kind != 0) {
            const uint32_t want = kind == 1 ? 1u : 2u, vlen = d.v_len[vbase + f];
            const uint32_t* cig = d.cigar_pool + d.r_cigoff[ridx[r]];
            const uint32_t ncig = d.r_ncig[ridx[r]];
            for (uint32_t c = 0; c < ncig; c++)
                if ((cig[c] & 0xF) == want && (cig[c] >> 4) == vlen) { s = true; break; }
        }
    };
    // Observation::update_haplotype for one variant (reference: microphasing.rs:157-197); hap already shifted
    auto update_row = [&](int r, bool s, bool q, uint32_t info) {
        uint32_t fs = (info & VI_FS_MASK) >> VI_FS_SHIFT;
        if (fs) fl[r] |= RF_F1;
        if (s) {
            if (info & 0x80000000u) fl[r] |= RF_SL;
            hap[r] |= 1ull;
            fr0[r] += fs;
        }
        if (q || (fl[r] & (RF_BAD | RF_SL))) { hap[r] = 0; fl[r] |= RF_BAD; }
    };

    // extend_right for one column (transcription-order index tr): bookkeeping + the new bit of every row (:232-256)
    auto append_column = [&](uint32_t tr) {
        uint32_t f = is_rev ? d.v_rev2fwd[vbase + tr] : tr;
        uint32_t info = d.v_info[vbase + f];
        uint32_t pos = d.v_pos[vbase + f];
        if (pos >= T.sl_lo && pos < T.sl_hi) info |= 0x80000000u; else info &= 0x7FFFFFFFu;
        __syncthreads();
        if (lane == 0) { colf[(head + ncols) & 63] = f; colinfo[(head + ncols) & 63] = info; }
        if (ncols == 0) { contig = true; f_oldest = f; }
        else contig = contig && (f == (is_rev ? f_newest - 1 : f_newest + 1));
        f_newest = f;
        ncols++;
        som_mask = (som_mask << 1) | ((info & VI_GERMLINE) ? 0ull : 1ull);
        special_mask = (special_mask << 1) | (((info & VI_FS_MASK) || (info & 0x80000000u)) ? 1ull : 0ull);
        n_special = uint32_t(__popcll(special_mask));
#pragma unroll
        for (int r = 0; r < RPL; r++)
            if ((fl[r] & ST_MASK) == ST_ROW) {
                bool s, q;
                bits_of(r, f, info, s, q);
                hap[r] <<= 1;
                update_row(r, s, q, info);
            }
    };
    // the columns that were alive when the previous exon ended (no row is): the segment's initial deque
    if (S.init_cols) {
        const Step st0 = d.steps[S.step_off];
        const uint32_t x = st0.col_hi - st0.n_add;
        for (uint32_t a = 0; a < S.init_cols; a++) append_column(x - S.init_cols + a);
    }

    for (uint32_t s0 = 0; s0 < S.n_steps; s0 += 64) {
        const uint32_t nb = min(64u, S.n_steps - s0);
        uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0, w5 = 0;
        if (lane < nb) {
            const uint32_t* sp = reinterpret_cast<const uint32_t*>(d.steps + (S.step_off + s0 + lane));
            w0 = sp[0]; w1 = sp[1]; w2 = sp[2]; w3 = sp[3]; w4 = sp[4]; w5 = sp[5];
        }
        for (uint32_t i = 0; i < nb; i++) {
            const uint32_t sso = rdlane(w0, i), cand_lo = rdlane(w1, i), col_hi = rdlane(w2, i), win = rdlane(w3, i);
            const uint32_t p4 = rdlane(w4, i), p5 = rdlane(w5, i);
            const uint32_t cand_n = p4 & 0xFFFF, wlen = (p4 >> 16) & 0xFF, n_del = p4 >> 24;
            const uint32_t n_add = p5 & 0xFF, sflags = (p5 >> 8) & 0xFF;
            const uint32_t splice_end = sso + wlen;

            // ---- cleanup_reads (:259-278): forward keeps end >= splice_end, reverse keeps start <= sso
#pragma unroll
            for (int r = 0; r < RPL; r++) {
                uint32_t st = fl[r] & ST_MASK;
                if (st != ST_EMPTY) {
                    bool keep = is_rev ? (rs[r] <= sso) : (re[r] >= splice_end);
                    if (is_rev && (sflags & SF_FULL_RANGE) && st == ST_PENDING) keep = false;  // re-listed below if still in range
                    if (!keep) fl[r] = ST_EMPTY;
                }
            }
            // ---- shrink_left (:220-229)
            if (n_del | n_add) colver++;
            if (n_del) {
                ncols -= n_del;
                head = (head + n_del) & 63;
                uint64_t mask = (1ull << ncols) - 1ull;
                som_mask &= mask;
                special_mask &= mask;
                n_special = uint32_t(__popcll(special_mask));
                if (ncols == 0) contig = true;
                else f_oldest = is_rev ? f_oldest - n_del : f_oldest + n_del;
#pragma unroll
                for (int r = 0; r < RPL; r++) hap[r] &= mask;
            }
            // ---- new candidate reads into empty slots
            bool any_rows_before = false;
            if (cand_n) {
                if (is_rev && (sflags & SF_FULL_RANGE)) {
#pragma unroll
                    for (int r = 0; r < RPL; r++) any_rows_before |= (__ballot((fl[r] & ST_MASK) == ST_ROW) != 0);
                }
                uint32_t left = cand_n, c = cand_lo;
#pragma unroll
                for (int r = 0; r < RPL; r++) {
                    if (left) {
                        bool empty = (fl[r] & ST_MASK) == ST_EMPTY;
                        uint64_t freem = __ballot(empty);
                        uint32_t rank = lanes_below(freem, lane);
                        if (empty && rank < left) {
                            uint32_t gi = rbase + c + rank;
                            ridx[r] = gi;
                            rs[r] = d.r_pos[gi];
                            re[r] = d.r_end[gi];
                            rvl[r] = d.r_varlo[gi];
                            rcov[r] = d.r_ncov[gi];
                            rdup[r] = d.r_dup[gi];
                            if (W == 1) { msup[r] = d.r_sup[gi]; mlq[r] = d.r_lq[gi]; }
                            fl[r] = ST_PENDING;
                            pver[r] = 0;
                        }
                        uint32_t took = min(uint32_t(__popcll(freem)), left);
                        c += took;
                        left -= took;
                    }
                }
                if (left) sticky_err |= WD_ROW_OVERFLOW;
                // A row that outlived its exon (a read longer than the intron) is listed again by the next exon's full-range scan.
                // `contains` (:281-294, tested before anything else in push_read) rejects that copy for as long as the row lives,
                // and the row leaves by the very test (start > sso) that ends the candidate: the copy is dropped here, whatever
                // its quality state, instead of being kept pending for a retry.
                if (any_rows_before) {
#pragma unroll
                    for (int rc = 0; rc < RPL; rc++) {
                        uint64_t m = __ballot((fl[rc] & ST_MASK) == ST_PENDING);
                        while (m) {
                            const uint32_t l = __builtin_ctzll(m);
                            m &= m - 1;
                            const uint32_t myidx = rdlane(ridx[rc], l);
                            bool is_row = false;
#pragma unroll
                            for (int r = 0; r < RPL; r++) is_row |= (__ballot((fl[r] & ST_MASK) == ST_ROW && ridx[r] == myidx) != 0);
                            if (is_row && lane == l) fl[rc] = ST_EMPTY;
                        }
                    }
                }
            }
            // ---- push_read (:297-343) for every pending candidate
            bool att[RPL];
            bool any_att = false;
#pragma unroll
            for (int r = 0; r < RPL; r++) {
                // a read rejected for bad quality stays rejected until the column set changes (:192-195, :333-335)
                att[r] = (fl[r] & ST_MASK) == ST_PENDING && re[r] >= splice_end && rs[r] <= sso && pver[r] != colver;
                any_att |= att[r];
            }
            if (__ballot(any_att)) {
#pragma unroll
                for (int r = 0; r < RPL; r++)
                    if (att[r]) { hap[r] = 0; fr0[r] = 0; fl[r] &= ST_MASK; }
                const uint32_t lo_f = is_rev ? f_newest : f_oldest;  // smallest forward index among the live columns
                bool fast = W == 1 && contig && n_special == 0;
                if (fast) {
                    bool cov_bad = false;
#pragma unroll
                    for (int r = 0; r < RPL; r++) cov_bad |= att[r] && ncols && !(lo_f >= rvl[r] && lo_f + ncols - rvl[r] <= rcov[r]);
                    fast = __ballot(cov_bad) == 0;
                }
                if (fast) {
                    const uint64_t cmask = ncols ? (~0ull >> (64 - ncols)) : 0ull;
#pragma unroll
                    for (int r = 0; r < RPL; r++)
                        if (att[r] && ncols) {
                            const uint32_t sh = lo_f - rvl[r];
                            const uint64_t sb = (msup[r] >> sh) & cmask, qb = (mlq[r] >> sh) & cmask;
                            // '+': column j (oldest = 0) is forward index lo_f + j and haplotype bit ncols-1-j -> reverse the run
                            hap[r] = is_rev ? sb : (__brevll(sb) >> (64 - ncols));
                            if (qb) { hap[r] = 0; fl[r] |= RF_BAD; }  // :192-195
                        }
                } else {
                    __syncthreads();
                    for (uint32_t j = 0; j < ncols; j++) {
                        uint32_t f = colf[(head + j) & 63], info = colinfo[(head + j) & 63];
#pragma unroll
                        for (int r = 0; r < RPL; r++)
                            if (att[r]) {
                                bool s, q;
                                bits_of(r, f, info, s, q);
                                hap[r] <<= 1;
                                update_row(r, s, q, info);
                            }
                    }
                }
                // `contains` (:281-294) only ever matches on the reverse strand (rows keyed by start)
                bool need_dup = false;
                if (is_rev) {
#pragma unroll
                    for (int r = 0; r < RPL; r++) need_dup |= att[r] && !(fl[r] & RF_BAD) && ((rdup[r] >> 31) || any_rows_before);
                }
                if (__ballot(need_dup)) {
#pragma unroll
                    for (int rc = 0; rc < RPL; rc++) {
                        bool chk = is_rev && att[rc] && !(fl[rc] & RF_BAD) && ((rdup[rc] >> 31) || any_rows_before);
                        uint64_t m = __ballot(chk);
                        while (m) {
                            uint32_t l = __builtin_ctzll(m);
                            m &= m - 1;
                            uint32_t key = rdlane(rdup[rc], l) & 0x7FFFFFFFu;
                            uint32_t myidx = rdlane(ridx[rc], l);
                            bool hit = false;
#pragma unroll
                            for (int r = 0; r < RPL; r++) {
                                bool same = (rdup[r] & 0x7FFFFFFFu) == key;
                                bool row_hit = (fl[r] & ST_MASK) == ST_ROW && same;
                                bool cand_hit = att[r] && !(fl[r] & RF_BAD) && same && ridx[r] < myidx;
                                hit |= (__ballot(row_hit || cand_hit) != 0);
                            }
                            if (hit && lane == l) att[rc] = false;  // contained: stays pending
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < RPL; r++)
                    if (att[r]) {
                        if (fl[r] & RF_BAD) { fl[r] = is_rev ? uint32_t(ST_PENDING) : uint32_t(ST_EMPTY); pver[r] = colver; }  // :333-335 not inserted
                        else fl[r] = (fl[r] & ~ST_MASK) | ST_ROW;
                    }
            }
            if (!is_rev) {  // forward: every key is offered exactly once
#pragma unroll
                for (int r = 0; r < RPL; r++)
                    if ((fl[r] & ST_MASK) == ST_PENDING) fl[r] = ST_EMPTY;
            }
            // ---- extend_right (:232-256)
            for (uint32_t a = 0; a < n_add; a++) append_column(col_hi - n_add + a);
            // ---- count phase of print_haplotypes (:383-411)
            if (sflags & SF_PRINT) {
                uint32_t nrows = 0, nvalid = 0;
                bool act[RPL];
#pragma unroll
                for (int r = 0; r < RPL; r++) {
                    bool row = (fl[r] & ST_MASK) == ST_ROW;
                    act[r] = row && !(fl[r] & RF_BAD);
                    nrows += __popcll(__ballot(row));
                    nvalid += __popcll(__ballot(act[r]));
                }
                // the reference haplotype (0, frame 0) is always listed (count 0 if no row carries it): the
                // consumer needs its sequence for the empty-window case (:429-431)
                bool zero_here = false;
#pragma unroll
                for (int r = 0; r < RPL; r++) zero_here |= act[r] && hap[r] == 0 && fr0[r] == 0 && !(fl[r] & RF_F1);
                const bool has_zero = __ballot(zero_here) != 0;
                nvalid += has_zero ? 0u : 1u;
                uint32_t werr = sticky_err;
                if (chunk_pos + nvalid > chunk_end) {
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(gcur, (unsigned long long)GROUP_CHUNK);
                    uint32_t blo = rdlane(uint32_t(base), 0), bhi = rdlane(uint32_t(base >> 32), 0);
                    chunk_pos = gpart_lo + ((uint64_t(bhi) << 32) | blo);
                    chunk_end = chunk_pos + GROUP_CHUNK;
                }
                const bool can_write = chunk_end <= gpart_hi && nvalid <= GROUP_CHUNK;
                if (!can_write) werr |= WD_GROUP_OVERFLOW;
                const uint64_t gbase = chunk_pos;
                uint32_t ng = 0;
                const bool need_all = (sflags & SF_NEED_RECS) != 0;
                // write the staged groups of the `on` lanes; groups whose sequence the host may need (a somatic column
                // is set, or the planner asked for every haplotype of this window) get a HapRec slot here
                auto emit = [&](bool on, uint64_t gi, uint32_t kh, uint32_t kl, uint32_t ka, uint32_t cnt) {
                    const uint64_t key = (uint64_t(kh) << 32) | kl;
                    const bool need = on && can_write && (need_all || (key & som_mask) != 0);
                    const uint64_t nm = __ballot(need);
                    const uint32_t nneed = __popcll(nm);
                    if (nneed && rec_pos + nneed > rec_end) {
                        unsigned long long base = 0;
                        if (lane == 0) base = atomicAdd(rcur, (unsigned long long)REC_CHUNK);
                        uint32_t blo = rdlane(uint32_t(base), 0), bhi = rdlane(uint32_t(base >> 32), 0);
                        rec_pos = rpart_lo + ((uint64_t(bhi) << 32) | blo);
                        rec_end = rec_pos + REC_CHUNK;
                    }
                    uint32_t rec = 0xFFFFFFFFu;
                    if (need) {
                        const uint64_t r = rec_pos + lanes_below(nm, lane);
                        if (r < rpart_hi) rec = uint32_t(r);
                    }
                    if (nneed && rec_pos + nneed > rpart_hi) sticky_err |= WD_REC_OVERFLOW;
                    rec_pos += nneed;
                    if (on && can_write) {
                        Group G; G.hap = key; G.count = cnt; G.aux = ka;
                        d.groups[gi] = G;
                    }
                    k3_enqueue(d, part, on && can_write, gi, win, rec);
                };
                if constexpr (RPL == 1) {
                    // <= 64 rows: (1) leader loop - pick any remaining row, ballot the rows with the same key (its count),
                    // stage the key in lane `ng`; (2) rank every staged key among the others (all-pairs via readlane) and
                    // store it at its rank, so memory holds the keys in the reference's ascending BTreeMap order (:383).
                    uint32_t khi_s = 0, klo_s = 0, ka_s = 0, cnt_s = 0;   // lane g stages group g
                    if (!has_zero) ng = 1;                                // lane 0 = zero-count reference haplotype (0, frame 0)
                    const uint32_t my_hi = uint32_t(hap[0] >> 32), my_lo = uint32_t(hap[0]);
                    const uint32_t my_a = (fr0[0] << 1) | ((fl[0] & RF_F1) ? 1u : 0u);
                    uint64_t rem = __ballot(act[0]);
                    while (rem) {
                        const uint32_t l = __builtin_ctzll(rem);
                        const uint32_t kh = rdlane(my_hi, l), kl = rdlane(my_lo, l), ka = rdlane(my_a, l);
                        const uint64_t m = __ballot(act[0] && my_hi == kh && my_lo == kl && my_a == ka);
                        rem &= ~m;
                        if (lane == ng) { khi_s = kh; klo_s = kl; ka_s = ka; cnt_s = uint32_t(__popcll(m)); }
                        ng++;
                    }
                    uint32_t rank = 0;
                    for (uint32_t j = 0; j < ng; j++) {
                        const uint32_t oh = rdlane(khi_s, j), ol = rdlane(klo_s, j), oa = rdlane(ka_s, j);
                        rank += (oh < khi_s || (oh == khi_s && (ol < klo_s || (ol == klo_s && oa < ka_s)))) ? 1u : 0u;
                    }
                    emit(lane < ng, gbase + rank, khi_s, klo_s, ka_s, cnt_s);
                } else {
                    ng = has_zero ? 0u : 1u;  // lane 0 stages the zero-count reference group
                    uint32_t sg_hi = 0, sg_lo = 0, sg_aux = 0, sg_cnt = 0;
                    // Distinct keys in ascending (haplotype, frame) order = the reference's BTreeMap order (:383).
                    // Repeated minimum extraction by bitwise descent: starting from all remaining rows, keep the
                    // rows whose key has a 0 at each bit (MSB first) whenever any such row exists. The survivors
                    // all carry the minimum key; their number is that haplotype's count. Wave-wide ballots only.
                    bool aux_here = false;
    #pragma unroll
                    for (int r = 0; r < RPL; r++) aux_here |= act[r] && (fr0[r] != 0 || (fl[r] & RF_F1));
                    const bool any_aux = __ballot(aux_here) != 0;
                    for (;;) {
                        bool c[RPL];
                        bool any_c = false;
    #pragma unroll
                        for (int r = 0; r < RPL; r++) { c[r] = act[r]; any_c |= c[r]; }
                        if (!__ballot(any_c)) break;
                        for (int bit = int(ncols) - 1; bit >= 0; bit--) {
                            const uint64_t mk = 1ull << bit;
                            bool z = false;
    #pragma unroll
                            for (int r = 0; r < RPL; r++) z |= c[r] && !(hap[r] & mk);
                            if (__ballot(z)) {
    #pragma unroll
                                for (int r = 0; r < RPL; r++) c[r] = c[r] && !(hap[r] & mk);
                            }
                        }
                        if (any_aux) {
                            for (int bit = 31; bit >= 0; bit--) {
                                const uint32_t mk = 1u << bit;
                                bool z = false;
    #pragma unroll
                                for (int r = 0; r < RPL; r++) z |= c[r] && !(((fr0[r] << 1) | ((fl[r] & RF_F1) ? 1u : 0u)) & mk);
                                if (__ballot(z)) {
    #pragma unroll
                                    for (int r = 0; r < RPL; r++) c[r] = c[r] && !(((fr0[r] << 1) | ((fl[r] & RF_F1) ? 1u : 0u)) & mk);
                                }
                            }
                        }
                        uint32_t cnt = 0, khi = 0, klo = 0, ka = 0;
                        bool have_key = false;
    #pragma unroll
                        for (int r = 0; r < RPL; r++) {
                            uint64_t m = __ballot(c[r]);
                            cnt += __popcll(m);
                            if (m && !have_key) {
                                uint32_t l = __builtin_ctzll(m);
                                khi = rdlane(uint32_t(hap[r] >> 32), l);
                                klo = rdlane(uint32_t(hap[r]), l);
                                ka = rdlane((fr0[r] << 1) | ((fl[r] & RF_F1) ? 1u : 0u), l);
                                have_key = true;
                            }
                            if (c[r]) act[r] = false;
                        }
                        if (lane == (ng & 63)) { sg_hi = khi; sg_lo = klo; sg_aux = ka; sg_cnt = cnt; }
                        ng++;
                        if ((ng & 63) == 0) emit(true, gbase + ng - 64 + lane, sg_hi, sg_lo, sg_aux, sg_cnt);
                    }
                    if (ng & 63) emit(lane < (ng & 63), gbase + (ng & ~63u) + lane, sg_hi, sg_lo, sg_aux, sg_cnt);
                }
                if (lane == 0) {
                    WinDyn wd;
                    wd.group_off = uint32_t(gbase);
                    wd.ngroups = ng;
                    wd.nrows = nrows;
                    wd.flags = WD_DONE | werr;
                    d.win_dyn[win] = wd;
                }
                if (can_write) { chunk_pos += ng; n_groups_tx += ng; }
            }
        }
    }
    if (sticky_err && lane == 0) atomicOr(d.err, sticky_err);
}

// ====================================================================== K2a + K2w (window-parallel replay)
// For an ExonW (plan.hpp) the observation matrix needs no sequential replay. push_read inserts a read at the first
// step at which it is a candidate, encloses the window and carries no low-quality / start-loss support bit on a live
// column (:297-343; '+': offered exactly once, '-': offered again until it is inserted or cleaned up); from then on
// its row is sticky-bad iff such a bit shows up on any column appended later (:157-197), and its haplotype is its
// support mask over the live columns. So:
//   K2a  one thread per (exon, read): the insertion step and the oldest column the row has seen  -> AdmEntry
//   K2w  one wave per run of steps: at every printing step the lanes take the <= 64 reads that can enclose the window,
//        derive row / bad / haplotype from AdmEntry + K1 masks + the step's column range, then count exactly like K2.
__device__ __forceinline__ uint64_t bit_range(uint32_t flo, uint32_t fhi, uint32_t base) {  // bits (f - base) for f in [flo, fhi), f - base in [0, 64)
    const uint32_t lo = flo > base ? flo - base : 0u;
    const uint32_t hi = fhi > base ? min(fhi - base, 64u) : 0u;
    if (hi <= lo) return 0ull;
    const uint64_t upto_hi = hi >= 64 ? ~0ull : ((1ull << hi) - 1ull);
    return upto_hi & ~((1ull << lo) - 1ull);
}
// forward-index range [flo, fhi) of the columns with transcription-order index in [tlo, thi)
__device__ __forceinline__ void tr_to_f(const ExonW& e, bool is_rev, uint32_t tlo, uint32_t thi, uint32_t& flo, uint32_t& fhi) {
    if (thi <= tlo) { flo = fhi = 0; return; }
    if (!is_rev) { flo = tlo; fhi = thi; }
    else { flo = e.f0 - (thi - 1 - e.tr0); fhi = e.f0 - (tlo - e.tr0) + 1; }
}

// mask helpers for W = 1 or 2 words per read (bit b of the mask = variant r_varlo + b)
template <int W>
__device__ __forceinline__ bool mask_hits(const uint64_t (&m)[W], uint32_t flo, uint32_t fhi, uint32_t base) {
    bool hit = false;
#pragma unroll
    for (int w = 0; w < W; w++) hit |= (m[w] & bit_range(flo, fhi, base + 64u * w)) != 0;
    return hit;
}
// bits (flo + b - base), b in [0, 64), of the mask as one word (only the low ncols <= 63 bits are used by the caller)
template <int W>
__device__ __forceinline__ uint64_t mask_extract(const uint64_t (&m)[W], uint32_t flo, uint32_t base) {
    if (flo >= base) {
        const uint32_t rel = flo - base, wi = rel >> 6, sh = rel & 63u;
        uint64_t lo = 0, hi = 0;
#pragma unroll
        for (int w = 0; w < W; w++) { if (uint32_t(w) == wi) lo = m[w]; if (uint32_t(w) == wi + 1) hi = m[w]; }
        return sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
    }
    const uint32_t nsh = base - flo;
    return nsh < 64 ? m[0] << nsh : 0ull;
}

template <int W>
__global__ __launch_bounds__(64) void k2a_admission(DeviceBatch d) {
    const WChunk A = d.achunks[blockIdx.x];   // one wave per 64 candidate reads of an exon: step_first = first read, n_steps = count
    const ExonW e = d.achunk_exons[blockIdx.x];   // (= exons_w[A.exon], stored beside the work item: both loads go out together)
    const bool is_rev = e.strand != 0;
    const uint32_t rbase = e.rbase + e.read_lo;
    const uint32_t sso0 = e.sso0, sso1 = e.sso1;
    for (uint32_t k = A.step_first + threadIdx.x; k < A.step_first + A.n_steps; k += 64) {
        const uint32_t gi = rbase + k;
        const uint32_t start = d.r_pos[gi], end = d.r_end[gi], rvl = d.r_varlo[gi];
        uint64_t dirty[W], sup[W];   // dirty: low quality, or support of a start-loss variant
#pragma unroll
        for (int w = 0; w < W; w++) {
            sup[w] = d.r_sup[uint64_t(gi) * W + w];
            dirty[w] = d.r_lq[uint64_t(gi) * W + w] | (sup[w] & bit_range(e.sl_f_lo, e.sl_f_hi, rvl + 64u * w));
        }
        AdmEntry out;
        out.ord = 0xFFFFFFFFu;
        out.seen_lo = 0;
        auto try_step = [&](uint32_t si) -> bool {   // push_read at step si (after shrink_left, before extend_right)
            // everything the decision needs is fetched in ONE round of loads and the decision itself is branch-free: with an early
            // return after the window test the compiler fetches the column fields only afterwards - a second dependent round trip
            const uint32_t* sp = reinterpret_cast<const uint32_t*>(d.steps + si);
            const uint32_t s_sso = sp[0], s_col_hi = sp[2], s_w4 = sp[4], s_w5 = sp[5];   // [4]: cand_n | wlen << 16 | n_del << 24; [5]: n_add | ...
            const uint32_t nc = d.step_ncols[si];
            const uint32_t s_wlen = (s_w4 >> 16) & 0xFFu, s_nadd = s_w5 & 0xFFu;
            const uint32_t tlo = s_col_hi - nc, thi = s_col_hi - s_nadd;
            uint32_t flo, fhi;
            tr_to_f(e, is_rev, tlo, thi, flo, fhi);
            const bool encloses = !(end < s_sso + s_wlen) & !(start > s_sso);
            const bool clean = !mask_hits<W>(dirty, flo, fhi, rvl);
            const bool ok = encloses & clean;
            out.ord = ok ? si : out.ord;
            out.seen_lo = ok ? tlo : out.seen_lo;
            return ok;
        };
        if (!is_rev) {
            if (start >= e.first_key_lo && start <= sso0) {
                try_step(e.step_off);                   // the first window's candidate range (:1229-1240)
            } else if (start > sso0 && e.n_steps > 1 && start >= sso1) {
                // afterwards only the reads that start exactly at sso (:1241-1248); steps 1.. advance by one nt each
                uint32_t t = 1 + (start - sso1);
                if (t < e.unit_steps) {
                    try_step(e.step_off + t);            // sso(t) == start by the exon's arithmetic (plan.hpp ExonW::unit_steps)
                } else if (t < e.n_steps) {
                    uint32_t s_t = d.steps[e.step_off + t].sso;
                    while (s_t > start && t > 1) s_t = d.steps[e.step_off + --t].sso;            // (never taken for unit steps)
                    while (s_t < start && t + 1 < e.n_steps) s_t = d.steps[e.step_off + ++t].sso;
                    if (s_t == start) try_step(e.step_off + t);
                }
            }
        } else {
            // candidate while sso - R <= start <= sso; sso never increases along the exon (one nt per step, repeats at the end)
            const uint32_t top = start + e.range;
            uint32_t t = sso0 > top ? sso0 - top : 0;   // first step with sso <= start + R if every step moved by one
            if (t >= e.unit_steps) {                    // beyond the arithmetic stretch: look at the steps
                if (t >= e.n_steps) t = e.n_steps - 1;
                while (t > 0 && d.steps[e.step_off + t - 1].sso <= top) t--;
                while (t < e.n_steps && d.steps[e.step_off + t].sso > top) t++;
            }
            for (; t < e.n_steps; t++) {
                const uint32_t s_t = t < e.unit_steps ? sso0 - t : d.steps[e.step_off + t].sso;
                if (s_t < start) break;
                if (try_step(e.step_off + t)) break;
            }
        }
        if (e.consumers & EW_WAVE) d.adm[uint64_t(e.adm_off) + k] = out;   // (only where a wave-per-window kernel will read it: 145 windows at config C)
        if constexpr (W == 1) {
            if (d.lane_on && (e.consumers & EW_LANE)) {   // the same facts flattened for the lane-per-window kernel (plan.hpp RowRecA)
                uint32_t bad_from = 0xFFFFFFFFu;
                if (out.ord != 0xFFFFFFFFu) {
                    const uint64_t dm = dirty[0];
                    if (!is_rev) {   // transcription order = forward index: the first dirty column at or after the oldest one seen
                        const uint32_t from = max(out.seen_lo, rvl), sh = from - rvl;
                        const uint64_t m = sh < 64 ? dm >> sh : 0ull;
                        if (m) bad_from = from + uint32_t(__builtin_ctzll(m));
                    } else {         // transcription order descends in forward index: the highest dirty column at or below the oldest one seen
                        const uint32_t f_hi = e.f0 - (out.seen_lo - e.tr0);
                        if (f_hi >= rvl) {
                            const uint32_t rel = f_hi - rvl;
                            const uint64_t m = rel >= 63 ? dm : (dm & ((2ull << rel) - 1ull));
                            if (m) bad_from = e.tr0 + (e.f0 - (rvl + 63u - uint32_t(__builtin_clzll(m))));
                        }
                    }
                }
                RowRecA a;
                a.key = is_rev ? ~start : end;
                a.ord = out.ord;
                a.bad_from = bad_from;
                a.rvl = rvl;
                d.rr_a[uint64_t(e.adm_off) + k] = a;
                d.rr_sup[uint64_t(e.adm_off) + k] = sup[0];
            }
        }
    }
}

// The flat form: a lane per admission-table entry (exon, read), whatever exon it belongs to - the chunk form above leaves a third of its
// lanes empty (60 candidate reads per exon on average) and every wave has ONE chain of dependent loads in flight; here a lane holds
// ITEMS entries and the loads of each level - entry -> exon fields + read fields -> the first step to try - go out for all of them
// before anything is used. Same decisions, same outputs (the table index IS the entry).
#if MP_IN_PART(0)
__global__ __launch_bounds__(256) void k0_pack_admission(DeviceBatch d) {
    const uint64_t i = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    if (i < d.n_exons_w) {
        const ExonW e = d.exons_w[i];
        ExonA a;
        a.step_off = e.step_off; a.n_steps = e.n_steps; a.unit_steps = e.unit_steps; a.sso0 = e.sso0;
        a.sso1 = e.sso1; a.first_key_lo = e.first_key_lo; a.range = e.range; a.tr0 = e.tr0;
        a.f0 = e.f0; a.sl_f_lo = e.sl_f_lo; a.sl_f_hi = e.sl_f_hi; a.flags = (e.strand ? 1u : 0u) | (e.consumers << 8);
        a.adm_off = e.adm_off; a.read0 = e.rbase + e.read_lo; a.wlen_min = e.wlen_min; a.pad1 = 0;
        d.exons_a[i] = a;
    }
    if (i < d.n_adm) {   // the last exon whose first entry is at or before this one (offsets ascend; exons without reads share their successor's)
        uint32_t lo = 0, hi = d.n_exons_w;   // exons_w[lo].adm_off <= i < exons_w[hi].adm_off (hi == n: beyond the table)
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (d.exons_w[mid].adm_off <= i) lo = mid; else hi = mid;
        }
        const ExonW* e = d.exons_w + lo;
        AdmMap m;
        m.exon = lo;
        m.read = e->rbase + e->read_lo + uint32_t(i - e->adm_off);
        d.adm_map[i] = m;
    }
}
void launch_k0_pack_admission(const DeviceBatch& d, hipStream_t stream) {
    if (!d.k2a_flat) return;
    const uint64_t n = d.n_adm > d.n_exons_w ? d.n_adm : d.n_exons_w;
    hipLaunchKernelGGL(k0_pack_admission, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, stream, d);
    HIP_CHECK_LAUNCH();
}
#endif

template <int W, int ITEMS>
__global__ __launch_bounds__(64) void k2a_admission_flat(DeviceBatch d) {
    struct In {
        bool on, rev, go;
        uint32_t entry, t;
        AdmMap xr;
        ExonA e;
        uint32_t start, end, rvl;
        uint64_t sup[W], dirty[W];
        uint32_t s_sso, s_col_hi, s_w4, s_w5, nc;
        AdmEntry out;
    };
    const uint32_t n_adm = uint32_t(d.n_adm);
    auto load_entry = [&](In& I, uint32_t tile) __attribute__((always_inline)) {
        const uint64_t entry = uint64_t(tile) * 64u + threadIdx.x;
        I.on = entry < n_adm;
        I.entry = I.on ? uint32_t(entry) : n_adm - 1u;   // (idle lanes repeat the last entry and store nothing)
        I.xr = d.adm_map[I.entry];
    };
    auto load_fields = [&](In& I) __attribute__((always_inline)) {
        const uint4* ep = reinterpret_cast<const uint4*>(d.exons_a + I.xr.exon);
        const uint4 q0 = ep[0], q1 = ep[1], q2 = ep[2], q3 = ep[3];
        const uint32_t gi = I.xr.read;
        I.start = d.r_pos[gi]; I.end = d.r_end[gi]; I.rvl = d.r_varlo[gi];
#pragma unroll
        for (int w = 0; w < W; w++) { I.sup[w] = d.r_sup[uint64_t(gi) * W + w]; I.dirty[w] = d.r_lq[uint64_t(gi) * W + w]; }
        I.e.step_off = q0.x; I.e.n_steps = q0.y; I.e.unit_steps = q0.z; I.e.sso0 = q0.w;
        I.e.sso1 = q1.x; I.e.first_key_lo = q1.y; I.e.range = q1.z; I.e.tr0 = q1.w;
        I.e.f0 = q2.x; I.e.sl_f_lo = q2.y; I.e.sl_f_hi = q2.z; I.e.flags = q2.w;
        I.e.adm_off = q3.x; I.e.read0 = q3.y; I.e.wlen_min = q3.z;
    };
    auto load_step = [&](In& I, uint32_t si) __attribute__((always_inline)) {   // the fields push_read's decision needs, in one round of loads
        const uint32_t* sp = reinterpret_cast<const uint32_t*>(d.steps + si);
        I.s_sso = sp[0]; I.s_col_hi = sp[2]; I.s_w4 = sp[4]; I.s_w5 = sp[5];   // [4]: cand_n | wlen << 16 | n_del << 24; [5]: n_add | ...
        I.nc = d.step_ncols[si];
    };
    auto eval_step = [&](In& I, uint32_t si) __attribute__((always_inline)) -> bool {   // push_read at step si (after shrink_left, before extend_right), branch-free
        const uint32_t s_wlen = (I.s_w4 >> 16) & 0xFFu, s_nadd = I.s_w5 & 0xFFu;
        const uint32_t tlo = I.s_col_hi - I.nc, thi = I.s_col_hi - s_nadd;
        uint32_t flo = tlo, fhi = thi;
        if (I.rev) { flo = I.e.f0 - (thi - 1 - I.e.tr0); fhi = I.e.f0 - (tlo - I.e.tr0) + 1; }
        if (thi <= tlo) flo = fhi = 0;
        const bool encloses = !(I.end < I.s_sso + s_wlen) & !(I.start > I.s_sso);
        const bool clean = !mask_hits<W>(I.dirty, flo, fhi, I.rvl);
        const bool ok = encloses & clean;
        I.out.ord = ok ? si : I.out.ord;
        I.out.seen_lo = ok ? tlo : I.out.seen_lo;
        return ok;
    };
    auto first_step = [&](In& I) __attribute__((always_inline)) {   // the first step at which the read is offered (:1191-1249), by arithmetic where the steps move by one nt
        const ExonA& e = I.e;
        I.rev = (e.flags & 1u) != 0;
#pragma unroll
        for (int w = 0; w < W; w++) I.dirty[w] |= I.sup[w] & bit_range(e.sl_f_lo, e.sl_f_hi, I.rvl + 64u * w);   // low quality, or support of a start-loss variant
        I.out.ord = 0xFFFFFFFFu; I.out.seen_lo = 0;
        I.go = false; I.t = 0;
        const uint32_t start = I.start;
        if (!I.rev) {
            if (start >= e.first_key_lo && start <= e.sso0) {
                I.go = true;                                  // the first window's candidate range (:1229-1240)
            } else if (start > e.sso0 && e.n_steps > 1 && start >= e.sso1) {
                uint32_t t = 1 + (start - e.sso1);            // afterwards only the reads that start exactly at sso (:1241-1248)
                if (t < e.unit_steps) { I.go = true; I.t = t; }
                else if (t < e.n_steps) {
                    uint32_t s_t = d.steps[e.step_off + t].sso;
                    while (s_t > start && t > 1) s_t = d.steps[e.step_off + --t].sso;            // (never taken for unit steps)
                    while (s_t < start && t + 1 < e.n_steps) s_t = d.steps[e.step_off + ++t].sso;
                    if (s_t == start) { I.go = true; I.t = t; }
                }
            }
        } else {
            // candidate while sso - R <= start <= sso; sso never increases along the exon (one nt per step, repeats at the end)
            const uint32_t top = start + e.range;
            uint32_t t = e.sso0 > top ? e.sso0 - top : 0;   // first step with sso <= start + R if every step moved by one
            if (t >= e.unit_steps) {                        // beyond the arithmetic stretch: look at the steps
                if (t >= e.n_steps) t = e.n_steps - 1;
                while (t > 0 && d.steps[e.step_off + t - 1].sso <= top) t--;
                while (t < e.n_steps && d.steps[e.step_off + t].sso > top) t++;
            }
            if (t < e.unit_steps) {
                // a read is offered from the step at which sso has come down to start + R - R steps before a window can lie inside the
                // read -, so the first offers would all fail `encloses`: the steps at which even the exon's shortest window sticks out
                // beyond the read's end are skipped by arithmetic (one dependent round of step loads less for nearly every '-' read)
                const uint32_t need = e.sso0 + e.wlen_min > I.end ? e.sso0 + e.wlen_min - I.end : 0u;   // first t with sso0 - t + wlen_min <= end
                t = max(t, min(need, e.unit_steps));
            }
            I.go = t < e.n_steps; I.t = I.go ? t : 0;
        }
    };
    auto finish = [&](In& I) __attribute__((always_inline)) {
        const ExonA& e = I.e;
        if (I.go) {
            if (!I.rev) eval_step(I, e.step_off + I.t);
            else {
                // offered again at every step until it is inserted or sso has passed its start (the first step's fields are already here)
                uint32_t t = I.t;
                for (;;) {
                    if (I.s_sso < I.start) break;
                    if (eval_step(I, e.step_off + t)) break;
                    if (++t >= e.n_steps) break;
                    load_step(I, e.step_off + t);
                }
            }
        }
        if (!I.on) return;
        const uint32_t consumers = e.flags >> 8;
        if (consumers & EW_WAVE) d.adm[I.entry] = I.out;
        if constexpr (W == 1) {
            if (d.lane_on && (consumers & EW_LANE)) {   // the same facts flattened for the lane-per-window kernel (plan.hpp RowRecA)
                uint32_t bad_from = 0xFFFFFFFFu;
                if (I.out.ord != 0xFFFFFFFFu) {
                    const uint64_t dm = I.dirty[0];
                    if (!I.rev) {
                        const uint32_t from = max(I.out.seen_lo, I.rvl), sh = from - I.rvl;
                        const uint64_t m = sh < 64 ? dm >> sh : 0ull;
                        if (m) bad_from = from + uint32_t(__builtin_ctzll(m));
                    } else {
                        const uint32_t f_hi = e.f0 - (I.out.seen_lo - e.tr0);
                        if (f_hi >= I.rvl) {
                            const uint32_t rel = f_hi - I.rvl;
                            const uint64_t m = rel >= 63 ? dm : (dm & ((2ull << rel) - 1ull));
                            if (m) bad_from = e.tr0 + (e.f0 - (I.rvl + 63u - uint32_t(__builtin_clzll(m))));
                        }
                    }
                }
                RowRecA a;
                a.key = I.rev ? ~I.start : I.end;
                a.ord = I.out.ord;
                a.bad_from = bad_from;
                a.rvl = I.rvl;
                d.rr_a[I.entry] = a;
                d.rr_sup[I.entry] = I.sup[0];
            }
        }
    };
    In in0, in1;
    load_entry(in0, blockIdx.x * ITEMS);
    if constexpr (ITEMS == 2) load_entry(in1, blockIdx.x * ITEMS + 1);
    load_fields(in0);
    if constexpr (ITEMS == 2) load_fields(in1);
    first_step(in0);
    if constexpr (ITEMS == 2) first_step(in1);
    load_step(in0, in0.e.step_off + in0.t);   // (t = 0 where no step is tried: a valid address, the fields unused)
    if constexpr (ITEMS == 2) load_step(in1, in1.e.step_off + in1.t);
    finish(in0);
    if constexpr (ITEMS == 2) finish(in1);
}

// ====================================================================== K2l (lane-per-window replay)
// One lane = one printing window (plan.hpp WinW). The lane walks the window's candidate reads - consecutive RowRecs, shared
// with the neighbouring lanes' windows, so the gathers hit L1 / L2 - and counts the haplotype words of the rows that are not
// sticky-bad in ITS OWN column of an LDS table of 8-bit counters (table word t of lane l at hist[t * 64 + l]: every access of
// a wave instruction falls into a different bank). Keys come out in ascending order by walking the touched table words;
// the wave then takes one run of group slots / record slots for all its windows (prefix sums, two atomics per 64 windows),
// so consecutive windows' groups are consecutive in memory and no slot is left unused.
//   reference: ObservationMatrix rows and the count phase of print_haplotypes, src/microphasing.rs:220-343, :383-411
// HB = 6 / 8: windows of <= 6 / 8 columns, a direct table of 2^HB byte counters per lane. HB = 16 (K2L_HASH_COLS): windows of 9..16 columns
// with at most K2L_HASH_ROWS candidate reads - the direct table would need 64 K counters, so the lane keeps a 64-slot open-addressing
// table in its LDS column instead (entry = haplotype word << 8 | count; linear probing; at most 63 distinct words, so a probe always ends
// at a free slot): one dependent LDS round trip per row instead of the wave-per-window kernel's ballots and readlanes (470 wave
// instructions per WINDOW there, ~40 here); the keys come out ascending by repeated minimum over the lane's occupied slots.
template <int HB, int STAGE_, int GATHER_ = 8>   // STAGE_: RowRecs staged in LDS per pass (24 bytes each); 0 = straight from memory, GATHER_ rows at a time
__global__ __launch_bounds__(64) void k2l_window_lanes(DeviceBatch d, uint32_t first, uint32_t count) {
    constexpr bool HASH = HB > 8;
    constexpr uint32_t NW = HASH ? 64u : (1u << HB) / 4;   // table words per lane (direct: four 8-bit counters each; hash: one entry each); NW <= 64
    constexpr uint32_t EMPTY = 0xFFFFFFFFu;
    constexpr uint32_t STAGE = STAGE_ ? STAGE_ : 1;
    static_assert(STAGE_ == 0 || STAGE_ > int(K2L_MAX_ROWS), "a window's candidate reads must fit one stage pass, or the staging loop never ends");
    __shared__ uint32_t hist[NW * 64];
    __shared__ uint4 st_a[STAGE];
    __shared__ uint64_t st_s[STAGE];
    const uint32_t lane = threadIdx.x;
#pragma unroll 4
    for (uint32_t t = 0; t < NW; t++) hist[t * 64 + lane] = HASH ? EMPTY : 0u;
    const uint32_t part = blockIdx.x & (NPART - 1);
    unsigned long long* const gcur = d.cursors + part * 32;
    unsigned long long* const rcur = gcur + 16;
    const uint64_t gpart_lo = uint64_t(part) << d.group_part_log2, gpart_size = 1ull << d.group_part_log2;
    const uint64_t rpart_lo = uint64_t(part) << d.rec_part_log2, rpart_size = 1ull << d.rec_part_log2;
    uint32_t sticky_err = 0;
    const uint32_t n_tiles = (count + 63) / 64;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t li = tile * 64 + lane;
        const bool valid = li < count;
        uint32_t win = 0, rr_lo = 0, pack = 0, wkey = 0, step = 0, col_hi = 0, flo = 0;
        uint64_t som_mask = 0;
        {   // (unconditional loads at a clamped index, selected afterwards: no memory-counter drain at a branch join)
            const uint32_t at = first + min(li, count - 1);
            const uint4* wp = reinterpret_cast<const uint4*>(d.winw + at);
            const uint4 w0 = wp[0], w1 = wp[1];
            const uint32_t wn = d.lane_win[at];
            if (valid) {
                rr_lo = w0.x; pack = w0.y; wkey = w0.z; step = w0.w;
                col_hi = w1.x; flo = w1.y; som_mask = (uint64_t(w1.w) << 32) | w1.z;
                win = wn;
            }
        }
        const uint32_t r_n = pack & 0x3FF, ncols = (pack >> 10) & 0x3F;
        const bool fwd = (pack & WW_FWD) != 0, need_all = (pack & WW_NEED_ALL) != 0;
        const bool trivial = (pack & WW_TRIVIAL) != 0;   // simple window that cannot hold a stop: a group without a somatic column needs no K3
        const bool all_ids = (pack & WW_ALL_IDS) != 0;   // every haplotype of the window is hashed (else only those that set a somatic column)
        const bool simple = (pack & WW_SIMPLE) != 0;     // K3 builds the sequences by byte substitution; every group of another window goes to K3's list C ...
        const bool walk = (pack & WW_WALK) != 0;         // ... or, when its sequences need the general walk, to list D (walk implies !simple)
        // ---- rows of the window, haplotypes counted as they come (branch-free body: every lane adds 0 or 1 to one counter)
        const uint32_t cmask32 = ncols ? (0xFFFFFFFFu >> (32 - ncols)) : 0u;   // ncols <= 8: the low dword of the shifted mask is enough
        const uint32_t rev_sh = (32u - ncols) & 31u;
        uint32_t nrows = 0;
        uint64_t touched = 0;   // table words this lane incremented
        auto count_row = [&](const uint4 a, const uint64_t sup, const bool in) {
            const bool row = in && a.y <= step && a.x >= wkey;   // inserted, and not cleaned up since (:259-278)
            const bool act = row && col_hi <= a.z;               // not sticky-bad (:192-195)
            const int dsh = int(flo - a.w);                      // window's lowest column relative to the mask's bit 0
            const uint32_t right = uint32_t(sup >> (uint32_t(dsh) & 63u)), left = uint32_t(sup) << (uint32_t(-dsh) & 31u);
            const uint32_t bits = dsh >= 0 ? (dsh < 64 ? right : 0u) : (dsh > -32 ? left : 0u);
            const uint32_t h = (fwd ? (__brev(bits) >> rev_sh) : bits) & cmask32;
            nrows += row ? 1u : 0u;
            if constexpr (!HASH) {
                atomicAdd(&hist[(h >> 2) * 64 + lane], act ? (1u << (8 * (h & 3))) : 0u);   // the lane's own counter: a plain ds_add
                touched |= act ? (1ull << (h >> 2)) : 0ull;
            } else {
                // insert / count in the lane's own table column (every lane probes in step; `touched` = the occupied slots)
                uint32_t slot = (h * 0x9E3779B1u) >> 26;
                bool todo = act;
                while (__ballot(todo)) {
                    const uint32_t e = hist[slot * 64 + lane];
                    const bool fresh = e == EMPTY, hit = (e >> 8) == h;
                    if (todo && (fresh || hit)) {
                        hist[slot * 64 + lane] = fresh ? ((h << 8) | 1u) : e + 1u;
                        touched |= fresh ? (1ull << slot) : 0ull;
                        todo = false;
                    }
                    slot = (slot + 1) & 63u;
                }
            }
        };
        // The candidate ranges of a tile's windows overlap almost completely (consecutive windows of one exon): the wave stages
        // their union in LDS, STAGE records at a time starting at the lowest range not done yet - one pass for most tiles, two
        // or three where a tile crosses into another exon (r_n <= K2L_MAX_ROWS < STAGE: a window always fits the pass it opens).
        auto wave_min = [](uint32_t v) {
#pragma unroll
            for (uint32_t off = 1; off < 64; off <<= 1) v = min(v, uint32_t(__shfl_xor(v, off)));
            return rdlane(v, 0);
        };
        auto wave_max = [](uint32_t v) {
#pragma unroll
            for (uint32_t off = 1; off < 64; off <<= 1) v = max(v, uint32_t(__shfl_xor(v, off)));
            return rdlane(v, 0);
        };
        if constexpr (STAGE_ == 0) {
            // (rows straight from memory: the loads of GATHER_BLK rows go out together, at clamped indices and unconditionally - a lane
            //  walks ~30 rows, and with one or two rows in flight every few rows cost a full memory round trip)
            constexpr uint32_t GATHER_BLK = GATHER_;
            const uint32_t rn_max = wave_max(r_n);
            const uint32_t r_last = r_n ? r_n - 1 : 0u;
            const uint4* const ra = reinterpret_cast<const uint4*>(d.rr_a) + rr_lo;
            const uint64_t* const rs = d.rr_sup + rr_lo;
            for (uint32_t k0 = 0; k0 < rn_max; k0 += GATHER_BLK) {
                uint4 va[GATHER_BLK];
                uint64_t vs[GATHER_BLK];
#pragma unroll
                for (uint32_t u = 0; u < GATHER_BLK; u++) { const uint32_t i = min(k0 + u, r_last); va[u] = ra[i]; vs[u] = rs[i]; }
#pragma unroll
                for (uint32_t u = 0; u < GATHER_BLK; u++) count_row(va[u], vs[u], k0 + u < r_n);
            }
        } else {
        bool pending = r_n != 0;
        while (__ballot(pending)) {
            const uint32_t base = wave_min(pending ? rr_lo : 0xFFFFFFFFu);
            const bool fit = pending && rr_lo - base + r_n <= STAGE;
            const uint32_t n_stage = wave_max(fit ? rr_lo - base + r_n : 0u), rn_max = wave_max(fit ? r_n : 0u);
            __syncthreads();   // the previous pass's reads of the stage are done
            {   // n_stage <= STAGE records: all of a lane's loads first (clamped index), then the LDS writes
                constexpr uint32_t PER_LANE = (STAGE + 63) / 64;
                uint4 va[PER_LANE];
                uint64_t vs[PER_LANE];
                const uint32_t last = n_stage ? n_stage - 1 : 0;
#pragma unroll
                for (uint32_t u = 0; u < PER_LANE; u++) {
                    const uint32_t i = min(lane + 64u * u, last);
                    va[u] = *reinterpret_cast<const uint4*>(d.rr_a + base + i);
                    vs[u] = d.rr_sup[base + i];
                }
#pragma unroll
                for (uint32_t u = 0; u < PER_LANE; u++) {
                    const uint32_t i = lane + 64u * u;
                    if (i < n_stage) { st_a[i] = va[u]; st_s[i] = vs[u]; }
                }
            }
            __syncthreads();
            const uint32_t s_at = fit ? rr_lo - base : 0u, s_last = fit ? r_n - 1 : 0u;
#pragma unroll 4
            for (uint32_t k = 0; k < rn_max; k++) {
                const uint32_t at = s_at + min(k, s_last);
                count_row(st_a[at], st_s[at], fit && k < r_n);
            }
            pending = pending && !fit;
        }
        }
        // ---- groups in ascending key order: key 0 (the reference haplotype) is always listed (:429-431)
        if constexpr (!HASH) touched |= 1ull;
        uint32_t ng = 0, nneed = 0, nhash = 0;   // groups; those with a record slot; those of them whose id K3 will hash
        bool has_zero = false;   // (hash form) the reference haplotype has an entry of its own
        if (valid) {
            uint64_t tm = touched;
            while (tm) {
                const uint32_t t = uint32_t(__builtin_ctzll(tm));
                tm &= tm - 1;
                const uint32_t x = hist[t * 64 + lane];
                if constexpr (HASH) {
                    const uint32_t key = x >> 8;
                    const bool som = (uint64_t(key) & som_mask) != 0;
                    ng++; nneed += (need_all || som) ? 1u : 0u; nhash += (som || (need_all && all_ids)) ? 1u : 0u;
                    has_zero = has_zero || key == 0;
                } else {
#pragma unroll
                    for (uint32_t b = 0; b < 4; b++) {
                        const uint32_t key = 4 * t + b;
                        if (((x >> (8 * b)) & 0xFF) || key == 0) {
                            const bool som = (uint64_t(key) & som_mask) != 0;
                            ng++; nneed += (need_all || som) ? 1u : 0u; nhash += (som || (need_all && all_ids)) ? 1u : 0u;
                        }
                    }
                }
            }
            if (HASH && !has_zero) { ng++; nneed += need_all ? 1u : 0u; nhash += (need_all && all_ids) ? 1u : 0u; }   // key 0 is listed with count 0 (:429-431)
        }
        // wave-inclusive prefix sums of (ng, nneed), packed: ng <= 256 per lane -> sum <= 16384; nneed likewise
        uint32_t scan = ng | (nneed << 16);
#pragma unroll
        for (uint32_t off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(scan, off);
            if (lane >= off) scan += up;
        }
        const uint32_t total = rdlane(scan, 63);
        const uint32_t tot_g = total & 0xFFFF, tot_r = total >> 16;
        // K3's two lists (k3_enqueue): A = the groups whose id K3 will hash (a subset of those with a record slot), B = every other
        // group that is not settled here - with a record slot (a window whose haplotypes are carried into a splice merge) or without
        const uint32_t na = simple ? nhash : 0u;
        const uint32_t nb = simple ? (nneed - nhash) + (trivial ? 0u : ng - nneed) : 0u;
        const uint32_t nc = (simple || walk) ? 0u : ng;   // (a window that is not simple is never trivial: all its groups are listed)
        const uint32_t nd = walk ? ng : 0u;
        uint32_t scanb = na | (nb << 16);   // (both sums stay below 2^16)
#pragma unroll
        for (uint32_t off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(scanb, off);
            if (lane >= off) scanb += up;
        }
        const uint32_t tot_a = rdlane(scanb, 63) & 0xFFFFu, tot_b = rdlane(scanb, 63) >> 16;
        uint32_t scanc = nc | (nd << 16), tot_c = 0, tot_d = 0;
        if (__ballot((nc | nd) != 0)) {   // (wave-uniform: most tiles hold simple windows only)
#pragma unroll
            for (uint32_t off = 1; off < 64; off <<= 1) {
                const uint32_t up = __shfl_up(scanc, off);
                if (lane >= off) scanc += up;
            }
            tot_c = rdlane(scanc, 63) & 0xFFFFu; tot_d = rdlane(scanc, 63) >> 16;
        }
        // the tile's four allocations - group slots, record slots, entries of K3's two lists - in ONE atomic instruction: lanes 0..3 each
        // add to their own cursor (one after the other, each under its own condition, they were dependent round trips to L2)
        unsigned long long got = 0;
        if (lane < 6) {
            unsigned long long* const cur = lane == 0 ? gcur : lane == 1 ? rcur : lane == 2 ? gcur + 8 : lane == 3 ? gcur + 12 : lane == 4 ? gcur + 20 : gcur + 28;
            got = atomicAdd(cur, (unsigned long long)(lane == 0 ? tot_g : lane == 1 ? tot_r : lane == 2 ? tot_a : lane == 3 ? tot_b : lane == 4 ? tot_c : tot_d));
        }
        const uint64_t gbase = (uint64_t(rdlane(uint32_t(got >> 32), 0)) << 32) | rdlane(uint32_t(got), 0);
        const uint64_t rbase = (uint64_t(rdlane(uint32_t(got >> 32), 1)) << 32) | rdlane(uint32_t(got), 1);
        const uint64_t la_base = (uint64_t(rdlane(uint32_t(got >> 32), 2)) << 32) | rdlane(uint32_t(got), 2);
        const uint64_t lb_base = (uint64_t(rdlane(uint32_t(got >> 32), 3)) << 32) | rdlane(uint32_t(got), 3);
        const uint64_t lc_base = (uint64_t(rdlane(uint32_t(got >> 32), 4)) << 32) | rdlane(uint32_t(got), 4);
        const uint64_t ld_base = (uint64_t(rdlane(uint32_t(got >> 32), 5)) << 32) | rdlane(uint32_t(got), 5);
        // list A: upwards from the sub-range's first entry; list C: upwards in the second array (a lane needs one of the two)
        uint64_t up_slot = simple ? gpart_lo + la_base + ((scanb & 0xFFFFu) - na) : d.group_cap + gpart_lo + lc_base + ((scanc & 0xFFFFu) - nc);
        // list B: downwards from the sub-range's last entry; list D: downwards in the second array (again one of the two per lane)
        uint64_t lb_slot = walk ? d.group_cap + gpart_lo + gpart_size - 1 - (ld_base + ((scanc >> 16) - nd))
                                : gpart_lo + gpart_size - 1 - (lb_base + ((scanb >> 16) - nb));
        // (the list cursors also count the entries of tiles that could not write - they leave holes -, so a tile whose groups fit can
        //  still find a list run past the sub-range: such a tile writes nothing either; K3 never walks a list with holes, the error
        //  word makes it leave and the pass is run again with larger arenas. The two lists cannot meet unless the groups overflow.)
        const bool can_write = gbase + tot_g <= gpart_size && la_base + tot_a <= gpart_size && lb_base + tot_b <= gpart_size && lc_base + tot_c <= gpart_size &&
                               ld_base + tot_d <= gpart_size;
        const bool rec_ok = rbase + tot_r <= rpart_size;
        uint32_t werr = 0;
        if (!can_write) werr |= WD_GROUP_OVERFLOW;
        if (!rec_ok) { werr |= WD_REC_OVERFLOW; }
        sticky_err |= werr;
        uint64_t gslot = gpart_lo + gbase + ((scan & 0xFFFF) - ng);
        uint64_t rslot = rpart_lo + rbase + ((scan >> 16) - nneed);
        if (valid) {
            const uint32_t goff = uint32_t(gslot);
            auto emit = [&](const uint32_t key, const uint32_t cnt) __attribute__((always_inline)) {   // (inlined: the slot cursors it advances stay in registers)
                const bool need = need_all || (uint64_t(key) & som_mask) != 0;
                if (can_write) {
                    const bool settled = trivial && !need;   // what K3 would find: valid, no stop, mutant == germline, no record
                    Group G; G.hap = key; G.count = cnt; G.aux = settled ? GROUP_SETTLED : 0u;
                    d.groups[gslot] = G;
                    // the rest is for K3 only: one item (k3_enqueue's layout) in list A or B
                    // (one store at a selected address and plain cursor arithmetic: with a store per branch the compiler kept the two cursors
                    //  in a dynamically indexed scratch array - 0.3 GB of private-memory traffic per pass)
                    const bool hashes = (uint64_t(key) & som_mask) != 0 || (need_all && all_ids);   // its id will be hashed
                    // (a lane's window is simple or it is not: the lane walks list A or list C - one upward cursor - and list B)
                    const bool to_up = simple ? hashes : !walk, to_b = simple ? (!hashes && !settled) : walk;
                    const uint64_t at = to_up ? up_slot : lb_slot;
                    if (to_up || to_b) d.k3_items[at] = make_uint4(uint32_t(gslot), win, (need && rec_ok) ? uint32_t(rslot) : 0xFFFFFFFFu, 0u);
                    up_slot += to_up ? 1u : 0u;
                    lb_slot -= to_b ? 1u : 0u;
                }
                gslot++;
                rslot += need ? 1u : 0u;
            };
            if constexpr (HASH) {
                // ascending keys (the reference's BTreeMap order, :383): the smallest remaining entry of the lane's occupied slots, again
                // and again (entries are key << 8 | count with distinct keys: comparing entries compares keys)
                if (!has_zero) emit(0u, 0u);
                uint64_t left = touched;
                while (left) {
                    uint32_t best = EMPTY, bs = 0;
                    uint64_t tm = left;
                    while (tm) {
                        const uint32_t t = uint32_t(__builtin_ctzll(tm));
                        tm &= tm - 1;
                        const uint32_t e = hist[t * 64 + lane];
                        if (e < best) { best = e; bs = t; }
                    }
                    left &= ~(1ull << bs);
                    hist[bs * 64 + lane] = EMPTY;   // ready for the next tile
                    emit(best >> 8, best & 0xFFu);
                }
            } else {
            uint64_t tm = touched;
            while (tm) {
                const uint32_t t = uint32_t(__builtin_ctzll(tm));
                tm &= tm - 1;
                const uint32_t x = hist[t * 64 + lane];
                hist[t * 64 + lane] = 0;   // ready for the next tile
#pragma unroll
                for (uint32_t b = 0; b < 4; b++) {
                    const uint32_t key = 4 * t + b, cnt = (x >> (8 * b)) & 0xFF;
                    if (cnt || key == 0) emit(key, cnt);
                }
            }
            }
            WinDyn wd;
            wd.group_off = goff;
            wd.ngroups = ng;
            wd.nrows = nrows;
            wd.flags = WD_DONE | werr;
            d.win_dyn[win] = wd;
        }
    }
    if (sticky_err && lane == 0) atomicOr(d.err, sticky_err);
}

constexpr uint32_t K2W_ITEMS = 1;       // work items per wave: with the lane kernel taking most windows, the remaining items hold few windows each -
                                        // one item per wave keeps four times as many waves in flight (window phase 1.14 -> 1.10 ms; four were
                                        // right while K3 had to skip the unused tail of every wave's last slot chunk)
constexpr uint32_t K2W_HIST_BITS = 6;   // windows with at most this many columns count their haplotypes in a 64-entry LDS table
#if MP_IN_PART(2)
__global__ __launch_bounds__(64) void k2w_window_rows(DeviceBatch d) {
    constexpr uint32_t GROUP_CHUNK = 64, REC_CHUNK_W = 64;   // small chunks: 64 allocators share the atomics; the unused tail of a
                                                             // wave's last chunk is capacity only (K3 walks dense lists)
    const uint32_t lane = threadIdx.x;
    // this wave's output allocator (kernels.hpp NPART)
    const uint32_t part = blockIdx.x & (NPART - 1);
    unsigned long long* const gcur = d.cursors + part * 32;
    unsigned long long* const rcur = gcur + 16;
    const uint64_t gpart_lo = uint64_t(part) << d.group_part_log2, gpart_hi = uint64_t(part + 1) << d.group_part_log2;
    const uint64_t rpart_lo = uint64_t(part) << d.rec_part_log2, rpart_hi = uint64_t(part + 1) << d.rec_part_log2;
    uint64_t chunk_pos = 0, chunk_end = 0, rec_pos = 0, rec_end = 0;
    uint32_t sticky_err = 0;
    __shared__ __attribute__((aligned(16))) uint32_t hist[256];
    __shared__ uint2 glist[64];
    // a wave takes K2W_ITEMS consecutive work items
    for (uint32_t item = blockIdx.x * K2W_ITEMS; item < min(d.n_wchunks, (blockIdx.x + 1) * K2W_ITEMS); item++) {
    const WChunk C = d.wchunks[item];
    const ExonW e = d.exons_w[C.exon];
    const bool is_rev = e.strand != 0;
    const uint32_t rbase = e.rbase;
    const uint32_t vbase = e.vbase;
    // the lanes hold a sliding block of 64 consecutive reads [L, L + 64) of the exon's candidate range; a window's rows are
    // the lanes inside its own range [r_lo, r_lo + r_n). The block moves only when a window's range leaves it.
    uint32_t L = 0;
    bool have_block = false;
    uint32_t q_start = 0, q_end = 0, q_rvl = 0, q_ord = 0xFFFFFFFFu, q_seen = 0;
    uint64_t q_sup = 0, q_dirty = 0;
    for (uint32_t s0 = 0; s0 < C.n_steps; s0 += 64) {
        const uint32_t nb = min(64u, C.n_steps - s0);
        uint32_t w0 = 0, w2 = 0, w3 = 0, w4 = 0, w5 = 0, w6 = 0, w7 = 0, w8 = 0, w9 = 0;
        if (lane < nb) {
            const uint32_t si = C.step_first + s0 + lane;
            const uint32_t* sp = reinterpret_cast<const uint32_t*>(d.steps + si);
            w0 = sp[0]; w2 = sp[2]; w3 = sp[3]; w4 = sp[4]; w5 = sp[5];
            w6 = d.step_rlo[si];
            const uint32_t nc = d.step_ncols[si];
            w7 = uint32_t(d.step_rn[si]) | (nc << 16);
            if (((w5 >> 8) & SF_PRINT) && nc) {   // somatic columns of the step's window, already in haplotype bit order
                uint32_t flo, fhi;
                tr_to_f(e, is_rev, w2 - nc, w2, flo, fhi);
                const uint64_t gv = uint64_t(vbase) + flo;
                const uint64_t x0 = d.v_sombits[gv >> 6], x1 = d.v_sombits[(gv >> 6) + 1];
                const uint32_t sh = uint32_t(gv & 63);
                uint64_t bits = (sh ? ((x0 >> sh) | (x1 << (64 - sh))) : x0) & (~0ull >> (64 - nc));
                if (!is_rev) bits = __brevll(bits) >> (64 - nc);
                w8 = uint32_t(bits); w9 = uint32_t(bits >> 32);
            }
        }
        // (the windows the lane-per-window kernel takes are not this kernel's: plan.cpp lane_window)
        uint64_t printing = __ballot(lane < nb && ((w5 >> 8) & SF_PRINT) &&
                                     !(d.lane_on && k2l_takes(w7 >> 16, w7 & 0xFFFF, d.lane_hash != 0)));
        while (printing) {
            const uint32_t i = uint32_t(__builtin_ctzll(printing));
            printing &= printing - 1;
            const uint32_t si = C.step_first + s0 + i;
            const uint32_t sso = rdlane(w0, i), col_hi = rdlane(w2, i), win = rdlane(w3, i);
            const uint32_t p4 = rdlane(w4, i), p5 = rdlane(w5, i), r_lo = rdlane(w6, i), p7 = rdlane(w7, i);
            const uint64_t som_mask = (uint64_t(rdlane(w9, i)) << 32) | rdlane(w8, i);
            const uint32_t wlen = (p4 >> 16) & 0xFF, sflags = (p5 >> 8) & 0xFF;
            const uint32_t r_n = p7 & 0xFFFF, ncols = p7 >> 16;
            const uint32_t splice_end = sso + wlen;
            uint32_t flo, fhi;   // live columns: transcription order [col_hi - ncols, col_hi) -> forward indices [flo, fhi)
            tr_to_f(e, is_rev, col_hi - ncols, col_hi, flo, fhi);
            const uint64_t cmask = ncols ? (~0ull >> (64 - ncols)) : 0ull;
            // ---- make sure the lanes hold this window's reads
            if (r_n && (!have_block || r_lo < L || r_lo + r_n > L + 64)) {
                const uint32_t r_hi = r_lo + r_n;
                if (!is_rev) L = r_lo;                                   // '+' windows move to later reads
                else L = r_hi > e.read_lo + 64 ? r_hi - 64 : e.read_lo;   // '-' windows move to earlier reads
                have_block = true;
                const uint32_t ri = L + lane;
                q_ord = 0xFFFFFFFFu;
                if (ri >= e.read_lo && ri < e.read_lo + e.n_reads) {
                    const uint32_t gi = rbase + ri;
                    q_start = d.r_pos[gi]; q_end = d.r_end[gi]; q_rvl = d.r_varlo[gi];
                    q_sup = d.r_sup[gi];
                    q_dirty = d.r_lq[gi] | (q_sup & bit_range(e.sl_f_lo, e.sl_f_hi, q_rvl));
                    const AdmEntry a = d.adm[uint64_t(e.adm_off) + (ri - e.read_lo)];
                    q_ord = a.ord; q_seen = a.seen_lo;
                }
                // consume the loads inside this (rare) branch: otherwise the wait for them lands after the join, where it would
                // also drain the previous window's stores on every iteration
                asm volatile("" : "+v"(q_start), "+v"(q_end), "+v"(q_rvl), "+v"(q_ord), "+v"(q_seen), "+v"(q_sup), "+v"(q_dirty));
            }
            // ---- the rows of this window
            const uint32_t ri = L + lane;
            bool row = r_n && ri >= r_lo && ri < r_lo + r_n && q_ord <= si &&
                       (is_rev ? q_start <= sso : q_end >= splice_end);   // inserted, and not cleaned up since (:259-278)
            bool act = false;
            uint64_t hap = 0;
            if (row) {
                uint32_t slo, shi;
                tr_to_f(e, is_rev, q_seen, col_hi, slo, shi);             // every column the row has seen
                act = (q_dirty & bit_range(slo, shi, q_rvl)) == 0;
                if (act && ncols) {
                    const uint64_t sb = (flo >= q_rvl ? (flo - q_rvl < 64 ? q_sup >> (flo - q_rvl) : 0ull)
                                                      : (q_rvl - flo < 64 ? q_sup << (q_rvl - flo) : 0ull)) & cmask;
                    hap = is_rev ? sb : (__brevll(sb) >> (64 - ncols));
                }
            }
            // ---- count phase of print_haplotypes (:383-411), as in k2_window_replay<1>
            const uint32_t nrows = __popcll(__ballot(row));
            const bool has_zero = __ballot(act && hap == 0) != 0;
            uint32_t ng;
            const bool need_all = (sflags & SF_NEED_RECS) != 0;
            uint32_t khi_s = 0, klo_s = 0, cnt_s = 0, rank = 0;
            bool on;
            if (ncols <= K2W_HIST_BITS) {
                // few columns (the common case): the key space fits the wave - one LDS counter per key, lane k then owns key k,
                // so the groups come out counted AND in ascending key order without any search. Key 0 (the reference haplotype)
                // is always listed, with count 0 if no row carries it (:429-431).
                hist[lane] = 0;
                __syncthreads();
                if (act) atomicAdd(&hist[uint32_t(hap)], 1u);
                __syncthreads();
                cnt_s = hist[lane];
                on = cnt_s != 0 || lane == 0;
                const uint64_t pm = __ballot(on);
                ng = uint32_t(__popcll(pm));
                rank = lanes_below(pm, lane);
                klo_s = lane;
            } else if (ncols <= K2W_HIST_BITS + 2) {
                // up to 256 keys: four counters per lane (keys 4 * lane .. 4 * lane + 3, still ascending), then the present ones are
                // compacted through LDS so that lane i holds the i-th group (at most 64: a window has at most 63 rows here)
                uint4* const hist4 = reinterpret_cast<uint4*>(hist);
                hist4[lane] = make_uint4(0, 0, 0, 0);
                __syncthreads();
                if (act) atomicAdd(&hist[uint32_t(hap)], 1u);
                __syncthreads();
                const uint4 c = hist4[lane];
                const bool p0 = c.x != 0 || lane == 0, p1 = c.y != 0, p2 = c.z != 0, p3 = c.w != 0;
                const uint64_t m0 = __ballot(p0), m1 = __ballot(p1), m2 = __ballot(p2), m3 = __ballot(p3);
                ng = uint32_t(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));
                uint32_t at = lanes_below(m0, lane) + lanes_below(m1, lane) + lanes_below(m2, lane) + lanes_below(m3, lane);
                if (p0) { glist[at] = make_uint2(4 * lane, c.x); at++; }
                if (p1) { glist[at] = make_uint2(4 * lane + 1, c.y); at++; }
                if (p2) { glist[at] = make_uint2(4 * lane + 2, c.z); at++; }
                if (p3) { glist[at] = make_uint2(4 * lane + 3, c.w); }
                __syncthreads();
                on = lane < ng;
                if (on) { const uint2 g2 = glist[lane]; klo_s = g2.x; cnt_s = g2.y; }
                rank = lane;
            } else {
                ng = has_zero ? 0u : 1u;   // lane 0 = zero-count reference haplotype
                const uint32_t my_hi = uint32_t(hap >> 32), my_lo = uint32_t(hap);
                uint64_t rem = __ballot(act);
                while (rem) {
                    const uint32_t l = __builtin_ctzll(rem);
                    const uint32_t kh = rdlane(my_hi, l), kl = rdlane(my_lo, l);
                    const uint64_t m = __ballot(act && my_hi == kh && my_lo == kl);
                    rem &= ~m;
                    if (lane == ng) { khi_s = kh; klo_s = kl; cnt_s = uint32_t(__popcll(m)); }
                    ng++;
                }
                for (uint32_t j = 0; j < ng; j++) {
                    const uint32_t oh = rdlane(khi_s, j), ol = rdlane(klo_s, j);
                    rank += (oh < khi_s || (oh == khi_s && ol < klo_s)) ? 1u : 0u;
                }
                on = lane < ng;
            }
            // group slots: exactly ng, from this wave's current chunk
            uint32_t werr = sticky_err;
            if (chunk_pos + ng > chunk_end) {
                unsigned long long base = 0;
                const uint32_t want = max(GROUP_CHUNK, ng);
                if (lane == 0) base = atomicAdd(gcur, (unsigned long long)want);
                const uint32_t blo = rdlane(uint32_t(base), 0), bhi = rdlane(uint32_t(base >> 32), 0);
                chunk_pos = gpart_lo + ((uint64_t(bhi) << 32) | blo);
                chunk_end = chunk_pos + want;
            }
            const bool can_write = chunk_end <= gpart_hi;
            if (!can_write) werr |= WD_GROUP_OVERFLOW;
            const uint64_t gbase = chunk_pos;
            {
                const uint64_t key = (uint64_t(khi_s) << 32) | klo_s;
                const bool need = on && can_write && (need_all || (key & som_mask) != 0);
                const uint64_t nm = __ballot(need);
                const uint32_t nneed = __popcll(nm);
                if (nneed && rec_pos + nneed > rec_end) {
                    unsigned long long base = 0;
                    const uint32_t want = max(REC_CHUNK_W, nneed);
                    if (lane == 0) base = atomicAdd(rcur, (unsigned long long)want);
                    const uint32_t blo = rdlane(uint32_t(base), 0), bhi = rdlane(uint32_t(base >> 32), 0);
                    rec_pos = rpart_lo + ((uint64_t(bhi) << 32) | blo);
                    rec_end = rec_pos + want;
                }
                uint32_t rec = 0xFFFFFFFFu;
                if (need) {
                    const uint64_t r = rec_pos + lanes_below(nm, lane);
                    if (r < rpart_hi) rec = uint32_t(r);
                }
                if (nneed && rec_pos + nneed > rpart_hi) sticky_err |= WD_REC_OVERFLOW;
                rec_pos += nneed;
                if (on && can_write) {
                    Group G; G.hap = key; G.count = cnt_s; G.aux = 0;
                    d.groups[gbase + rank] = G;
                }
                // (settling the trivial groups here as K2l does - one more dependent load per window for the window's flag - cost this
                //  kernel more than it saved K3: measured 0.53 -> 0.59 ms against no change in K3)
                k3_enqueue(d, part, on && can_write, gbase + rank, win, rec);
            }
            if (lane == 0) {
                WinDyn wd;
                wd.group_off = uint32_t(gbase);
                wd.ngroups = ng;
                wd.nrows = nrows;
                wd.flags = WD_DONE | werr;
                d.win_dyn[win] = wd;
            }
            if (can_write) chunk_pos += ng;
        }
    }
    }   // work items of this wave
    if (sticky_err && lane == 0) atomicOr(d.err, sticky_err);
}
#endif

// K2w for deeper data: RPL reads per lane (a block of 64 * RPL consecutive reads) and W mask words per read. Same row
// derivation as k2w_window_rows; the haplotypes are counted by repeated minimum extraction (bitwise descent with ballots),
// which yields them in ascending order, 64 at a time.
// Bitonic sort of n (a power of two, >= 64) keys in LDS by ONE wave, ascending.
template <class K>
__device__ __forceinline__ void bitonic_sort_wave(K* keys, uint32_t n, uint32_t lane) {
    for (uint32_t k = 2; k <= n; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = lane; t < n / 2; t += 64) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i + j;
                const K a = keys[i], b = keys[p];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { keys[i] = b; keys[p] = a; }
            }
            __syncthreads();
        }
}

template <int RPL, int W>
__global__ __launch_bounds__(64) void k2w_window_rows_multi(DeviceBatch d) {
    constexpr uint32_t CAP = 64u * RPL;
    __shared__ uint64_t sk[CAP];              // the window's haplotype words, sorted
    __shared__ uint16_t starts[CAP + 2];      // first entry of every run of equal words
    // windows of < 32 columns: the words are first de-duplicated in an LDS hash table (key / count; open addressing, the all-ones word
    // is free), so that only the DISTINCT words - ~80 of ~370 at 500x - are sorted, as (word << 32 | count) pairs
    constexpr uint32_t HT = 2 * CAP;          // slots: at most CAP distinct words, so the table is at most half full
    static_assert((HT & (HT - 1)) == 0 && HT <= 65536, "hash table size");
    __shared__ uint32_t tkey[HT], tcnt[HT];
    __shared__ uint16_t dlist[CAP];           // the slots in use
    __shared__ uint32_t n_dist;
    for (uint32_t i = threadIdx.x; i < HT; i += 64) { tkey[i] = 0xFFFFFFFFu; tcnt[i] = 0; }
    if (threadIdx.x == 0) n_dist = 0;
    __syncthreads();
    constexpr uint32_t GROUP_CHUNK = 64u * RPL + 64u, REC_CHUNK_W = 64;
    const uint32_t lane = threadIdx.x;
    const uint32_t part = blockIdx.x & (NPART - 1);
    unsigned long long* const gcur = d.cursors + part * 32;
    unsigned long long* const rcur = gcur + 16;
    const uint64_t gpart_lo = uint64_t(part) << d.group_part_log2, gpart_hi = uint64_t(part + 1) << d.group_part_log2;
    const uint64_t rpart_lo = uint64_t(part) << d.rec_part_log2, rpart_hi = uint64_t(part + 1) << d.rec_part_log2;
    uint64_t chunk_pos = 0, chunk_end = 0, rec_pos = 0, rec_end = 0;
    uint32_t sticky_err = 0;
    const WChunk C = d.wchunks_m[blockIdx.x];
    const ExonW e = d.exons_w[C.exon];
    const bool is_rev = e.strand != 0;
    const uint32_t rbase = e.rbase;
    const uint32_t vbase = e.vbase;
    uint32_t L = 0;
    bool have_block = false;
    uint32_t q_start[RPL], q_end[RPL], q_rvl[RPL], q_ord[RPL], q_seen[RPL];
    uint64_t q_sup[RPL][W], q_dirty[RPL][W];
#pragma unroll
    for (int k = 0; k < RPL; k++) {
        q_start[k] = q_end[k] = q_rvl[k] = q_seen[k] = 0; q_ord[k] = 0xFFFFFFFFu;
#pragma unroll
        for (int w = 0; w < W; w++) { q_sup[k][w] = 0; q_dirty[k][w] = 0; }
    }
    for (uint32_t s0 = 0; s0 < C.n_steps; s0 += 64) {
        const uint32_t nb = min(64u, C.n_steps - s0);
        uint32_t w0 = 0, w2 = 0, w3 = 0, w4 = 0, w5 = 0, w6 = 0, w7 = 0, w8 = 0, w9 = 0;
        if (lane < nb) {
            const uint32_t si = C.step_first + s0 + lane;
            const uint32_t* sp = reinterpret_cast<const uint32_t*>(d.steps + si);
            w0 = sp[0]; w2 = sp[2]; w3 = sp[3]; w4 = sp[4]; w5 = sp[5];
            w6 = d.step_rlo[si];
            const uint32_t nc = d.step_ncols[si];
            w7 = uint32_t(d.step_rn[si]) | (nc << 16);
            if (((w5 >> 8) & SF_PRINT) && nc) {
                uint32_t flo, fhi;
                tr_to_f(e, is_rev, w2 - nc, w2, flo, fhi);
                const uint64_t gv = uint64_t(vbase) + flo;
                const uint64_t x0 = d.v_sombits[gv >> 6], x1 = d.v_sombits[(gv >> 6) + 1];
                const uint32_t sh = uint32_t(gv & 63);
                uint64_t bits = (sh ? ((x0 >> sh) | (x1 << (64 - sh))) : x0) & (~0ull >> (64 - nc));
                if (!is_rev) bits = __brevll(bits) >> (64 - nc);
                w8 = uint32_t(bits); w9 = uint32_t(bits >> 32);
            }
        }
        // (the windows the lane-per-window kernel takes are not this kernel's: plan.cpp lane_window)
        uint64_t printing = __ballot(lane < nb && ((w5 >> 8) & SF_PRINT) &&
                                     !(d.lane_on && k2l_takes(w7 >> 16, w7 & 0xFFFF, d.lane_hash != 0)));
        while (printing) {
            const uint32_t i = uint32_t(__builtin_ctzll(printing));
            printing &= printing - 1;
            const uint32_t si = C.step_first + s0 + i;
            const uint32_t sso = rdlane(w0, i), col_hi = rdlane(w2, i), win = rdlane(w3, i);
            const uint32_t p4 = rdlane(w4, i), p5 = rdlane(w5, i), r_lo = rdlane(w6, i), p7 = rdlane(w7, i);
            const uint64_t som_mask = (uint64_t(rdlane(w9, i)) << 32) | rdlane(w8, i);
            const uint32_t wlen = (p4 >> 16) & 0xFF, sflags = (p5 >> 8) & 0xFF;
            const uint32_t r_n = p7 & 0xFFFF, ncols = p7 >> 16;
            const uint32_t splice_end = sso + wlen;
            uint32_t flo, fhi;
            tr_to_f(e, is_rev, col_hi - ncols, col_hi, flo, fhi);
            const uint64_t cmask = ncols ? (~0ull >> (64 - ncols)) : 0ull;
            if (r_n && (!have_block || r_lo < L || r_lo + r_n > L + CAP)) {
                const uint32_t r_hi = r_lo + r_n;
                if (!is_rev) L = r_lo;
                else L = r_hi > e.read_lo + CAP ? r_hi - CAP : e.read_lo;
                have_block = true;
#pragma unroll
                for (int k = 0; k < RPL; k++) {
                    const uint32_t ri = L + 64u * k + lane;
                    q_ord[k] = 0xFFFFFFFFu;
                    if (ri >= e.read_lo && ri < e.read_lo + e.n_reads) {
                        const uint32_t gi = rbase + ri;
                        q_start[k] = d.r_pos[gi]; q_end[k] = d.r_end[gi]; q_rvl[k] = d.r_varlo[gi];
#pragma unroll
                        for (int w = 0; w < W; w++) {
                            q_sup[k][w] = d.r_sup[uint64_t(gi) * W + w];
                            q_dirty[k][w] = d.r_lq[uint64_t(gi) * W + w] | (q_sup[k][w] & bit_range(e.sl_f_lo, e.sl_f_hi, q_rvl[k] + 64u * w));
                        }
                        const AdmEntry a = d.adm[uint64_t(e.adm_off) + (ri - e.read_lo)];
                        q_ord[k] = a.ord; q_seen[k] = a.seen_lo;
                    }
                }
            }
            // ---- the rows of this window
            bool act[RPL];
            uint64_t hap[RPL];
            uint32_t nrows = 0, nvalid = 0;
            bool zero_here = false;
#pragma unroll
            for (int k = 0; k < RPL; k++) {
                const uint32_t ri = L + 64u * k + lane;
                const bool row = r_n && ri >= r_lo && ri < r_lo + r_n && q_ord[k] <= si &&
                                 (is_rev ? q_start[k] <= sso : q_end[k] >= splice_end);
                act[k] = false;
                hap[k] = 0;
                if (row) {
                    uint32_t slo, shi;
                    tr_to_f(e, is_rev, q_seen[k], col_hi, slo, shi);
                    act[k] = !mask_hits<W>(q_dirty[k], slo, shi, q_rvl[k]);
                    if (act[k] && ncols) {
                        const uint64_t sb = mask_extract<W>(q_sup[k], flo, q_rvl[k]) & cmask;
                        hap[k] = is_rev ? sb : (__brevll(sb) >> (64 - ncols));
                    }
                }
                nrows += __popcll(__ballot(row));
                nvalid += __popcll(__ballot(act[k]));
                zero_here |= act[k] && hap[k] == 0;
            }
            const bool has_zero = __ballot(zero_here) != 0;
            nvalid += has_zero ? 0u : 1u;
            uint32_t werr = sticky_err;
            if (chunk_pos + nvalid > chunk_end) {
                unsigned long long base = 0;
                const uint32_t want = max(GROUP_CHUNK, nvalid);
                if (lane == 0) base = atomicAdd(gcur, (unsigned long long)want);
                const uint32_t blo = rdlane(uint32_t(base), 0), bhi = rdlane(uint32_t(base >> 32), 0);
                chunk_pos = gpart_lo + ((uint64_t(bhi) << 32) | blo);
                chunk_end = chunk_pos + want;
            }
            const bool can_write = chunk_end <= gpart_hi;
            if (!can_write) werr |= WD_GROUP_OVERFLOW;
            const uint64_t gbase = chunk_pos;
            const bool need_all = (sflags & SF_NEED_RECS) != 0;
            auto emit = [&](bool on, uint64_t gi, uint32_t kh, uint32_t kl, uint32_t cnt) {
                const uint64_t key = (uint64_t(kh) << 32) | kl;
                const bool need = on && can_write && (need_all || (key & som_mask) != 0);
                const uint64_t nm = __ballot(need);
                const uint32_t nneed = __popcll(nm);
                if (nneed && rec_pos + nneed > rec_end) {
                    unsigned long long base = 0;
                    const uint32_t want = max(REC_CHUNK_W, nneed);
                    if (lane == 0) base = atomicAdd(rcur, (unsigned long long)want);
                    const uint32_t blo = rdlane(uint32_t(base), 0), bhi = rdlane(uint32_t(base >> 32), 0);
                    rec_pos = rpart_lo + ((uint64_t(bhi) << 32) | blo);
                    rec_end = rec_pos + want;
                }
                uint32_t rec = 0xFFFFFFFFu;
                if (need) {
                    const uint64_t r = rec_pos + lanes_below(nm, lane);
                    if (r < rpart_hi) rec = uint32_t(r);
                }
                if (nneed && rec_pos + nneed > rpart_hi) sticky_err |= WD_REC_OVERFLOW;
                rec_pos += nneed;
                if (on && can_write) {
                    Group G; G.hap = key; G.count = cnt; G.aux = 0;
                    d.groups[gi] = G;
                }
                k3_enqueue(d, part, on && can_write, gi, win, rec);
            };
            // ---- count phase of print_haplotypes (:383-411): the haplotype words of the rows that are not bad are compacted into
            // LDS, sorted (bitonic, one wave, 32-bit keys when the window has at most 32 columns) and run-length counted - ascending
            // key order falls out. (The reference's BTreeMap does the same in O(R log H); repeated minimum extraction over the
            // lanes cost O(H x ncols x RPL) ballots and was 98 % of this kernel at 500x.)
            const uint32_t lead = has_zero ? 0u : 1u;   // the zero-count reference group goes first (:429-431)
            uint32_t ng = 0;
            auto count_sorted = [&](auto* keys) {
                using K = typename std::remove_pointer<decltype(keys)>::type;
                uint32_t n_act = 0;
#pragma unroll
                for (int k = 0; k < RPL; k++) {
                    const uint64_t m = __ballot(act[k]);
                    if (act[k]) keys[n_act + lanes_below(m, lane)] = K(hap[k]);
                    n_act += __popcll(m);
                }
                uint32_t N = 64;
                while (N < n_act) N <<= 1;
                for (uint32_t i = n_act + lane; i < N; i += 64) keys[i] = K(~K(0));   // padding sorts to the end (never a key: see below)
                __syncthreads();
                bitonic_sort_wave<K>(keys, N, lane);
                // group starts -> starts[0 .. n_groups], in order
                uint32_t n_groups = 0;
                for (uint32_t i0 = 0; i0 < n_act; i0 += 64) {
                    const uint32_t i = i0 + lane;
                    const bool is_start = i < n_act && (i == 0 || keys[i] != keys[i - 1]);
                    const uint64_t m = __ballot(is_start);
                    if (is_start) starts[n_groups + lanes_below(m, lane)] = uint16_t(i);
                    n_groups += __popcll(m);
                }
                if (lane == 0) starts[n_groups] = uint16_t(n_act);
                __syncthreads();
                ng = n_groups + lead;
                for (uint32_t g0 = 0; g0 < ng; g0 += 64) {
                    const uint32_t g = g0 + lane;
                    const bool on = g < ng;
                    uint32_t kh = 0, kl = 0, cnt = 0;
                    if (on && g >= lead) {
                        const uint32_t a0 = starts[g - lead], a1 = starts[g - lead + 1];
                        const uint64_t key = keys[a0];
                        kh = uint32_t(key >> 32); kl = uint32_t(key); cnt = a1 - a0;
                    }
                    emit(on, gbase + g, kh, kl, cnt);
                }
            };
            auto count_hashed = [&]() {
                constexpr uint32_t HBITS = 31u - uint32_t(__builtin_clz(HT));   // log2(HT)
#pragma unroll
                for (int k = 0; k < RPL; k++) {
                    if (act[k]) {
                        const uint32_t key = uint32_t(hap[k]);
                        uint32_t h = (key * 0x9E3779B1u) >> (32u - HBITS);
                        for (;;) {
                            const uint32_t old = atomicCAS(&tkey[h], 0xFFFFFFFFu, key);
                            if (old == 0xFFFFFFFFu) dlist[atomicAdd(&n_dist, 1u)] = uint16_t(h);   // this lane claimed the slot: list it once
                            if (old == 0xFFFFFFFFu || old == key) { atomicAdd(&tcnt[h], 1u); break; }
                            h = (h + 1u) & (HT - 1u);
                        }
                    }
                }
                __syncthreads();
                const uint32_t n = n_dist;
                uint32_t N = 64;
                while (N < n) N <<= 1;
                for (uint32_t i = lane; i < N; i += 64) {   // the distinct words with their counts -> sk; their slots are free again
                    uint64_t v = ~0ull;
                    if (i < n) {
                        const uint32_t h = dlist[i];
                        v = (uint64_t(tkey[h]) << 32) | tcnt[h];
                        tkey[h] = 0xFFFFFFFFu;
                        tcnt[h] = 0;
                    }
                    sk[i] = v;
                }
                __syncthreads();
                if (lane == 0) n_dist = 0;
                bitonic_sort_wave<uint64_t>(sk, N, lane);   // ascending in the word (the high half); ends with a barrier
                ng = n + lead;
                for (uint32_t g0 = 0; g0 < ng; g0 += 64) {
                    const uint32_t g = g0 + lane;
                    const bool on = g < ng;
                    uint32_t kl = 0, cnt = 0;
                    if (on && g >= lead) { const uint64_t v = sk[g - lead]; kl = uint32_t(v >> 32); cnt = uint32_t(v); }
                    emit(on, gbase + g, 0u, kl, cnt);
                }
            };
            // < 32 columns: the all-ones word cannot be a haplotype word, so it marks free table slots / sort padding; wider windows sort
            // all their 64-bit words (a word has at most 63 bits)
            if (ncols < 32) count_hashed();
            else count_sorted(sk);
            __syncthreads();   // sk / starts are reused by the next window
            if (lane == 0) {
                WinDyn wd;
                wd.group_off = uint32_t(gbase);
                wd.ngroups = ng;
                wd.nrows = nrows;
                wd.flags = WD_DONE | werr;
                d.win_dyn[win] = wd;
            }
            if (can_write) chunk_pos += ng;
        }
    }
    if (sticky_err && lane == 0) atomicOr(d.err, sticky_err);
}

// K2w for windows of ANY depth (exons with more than 512 candidate reads per window: amplicons, duplicate pile-ups, highly expressed
// genes): the window's candidate reads are streamed through the lanes 64 at a time - same row derivation as above - and the haplotype
// words of the rows that are not bad are counted in an LDS hash table (64-bit CAS, K2D_TABLE slots), so the number of ROWS is not
// bounded by anything; the distinct words are then sorted (bitonic) and written in ascending order with their counts looked up again.
// Limit (loud): K2D_TABLE distinct haplotypes in one window.
constexpr uint32_t K2D_TABLE = 4096;
template <int W>
__global__ __launch_bounds__(64) void k2w_window_rows_deep(DeviceBatch d) {
    __shared__ unsigned long long tkey[K2D_TABLE];   // haplotype + 1 (0 = free)
    __shared__ uint32_t tcnt[K2D_TABLE];
    __shared__ uint64_t skey[K2D_TABLE];             // the distinct haplotypes, sorted
    __shared__ uint32_t ng_lds;
    const uint32_t lane = threadIdx.x;
    const uint32_t part = blockIdx.x & (NPART - 1);
    unsigned long long* const gcur = d.cursors + part * 32;
    unsigned long long* const rcur = gcur + 16;
    const uint64_t gpart_lo = uint64_t(part) << d.group_part_log2, gpart_size = 1ull << d.group_part_log2;
    const uint64_t rpart_lo = uint64_t(part) << d.rec_part_log2, rpart_hi = uint64_t(part + 1) << d.rec_part_log2;
    uint64_t rec_pos = 0, rec_end = 0;
    uint32_t sticky_err = 0;
    for (uint32_t k = lane; k < K2D_TABLE; k += 64) { tkey[k] = 0; tcnt[k] = 0; }
    if (lane == 0) ng_lds = 0;
    __syncthreads();
    const WChunk C = d.wchunks_d[blockIdx.x];
    const ExonW e = d.exons_w[C.exon];
    const bool is_rev = e.strand != 0;
    const uint32_t rbase = e.rbase, vbase = e.vbase;
    auto slot_of_key = [](uint64_t hap) { return uint32_t((hap * 0x9E3779B97F4A7C15ull) >> 52) & (K2D_TABLE - 1); };
    for (uint32_t s = 0; s < C.n_steps; s++) {   // work items are a few steps long: every printing window costs thousands of reads
        const uint32_t si = C.step_first + s;
        const Step st = d.steps[si];
        const uint32_t ncols = d.step_ncols[si], rn16 = d.step_rn[si];
        if (!(st.flags & SF_PRINT) || (d.lane_on && ncols <= K2L_MAX_COLS && rn16 <= K2L_MAX_ROWS)) continue;   // (wave-uniform)
        const uint32_t sso = st.sso, col_hi = st.col_hi, win = st.win, splice_end = st.sso + st.wlen, r_lo = d.step_rlo[si];
        uint64_t som_mask = 0;
        uint32_t flo = 0, fhi = 0;
        tr_to_f(e, is_rev, col_hi - ncols, col_hi, flo, fhi);
        if (ncols) {
            const uint64_t gv = uint64_t(vbase) + flo;
            const uint64_t x0 = d.v_sombits[gv >> 6], x1 = d.v_sombits[(gv >> 6) + 1];
            const uint32_t sh = uint32_t(gv & 63);
            som_mask = (sh ? ((x0 >> sh) | (x1 << (64 - sh))) : x0) & (~0ull >> (64 - ncols));
            if (!is_rev) som_mask = __brevll(som_mask) >> (64 - ncols);
        }
        const uint64_t cmask = ncols ? (~0ull >> (64 - ncols)) : 0ull;
        uint32_t nrows = 0;
        bool zero_here = false;
        for (uint32_t ri0 = r_lo; ri0 < e.read_lo + e.n_reads; ri0 += 64) {
            const uint32_t ri = ri0 + lane;
            const bool in = ri >= e.read_lo && ri < e.read_lo + e.n_reads;
            const uint32_t gi = rbase + (in ? ri : e.read_lo);
            const uint32_t start = d.r_pos[gi];
            if (rdlane(start, 0) > sso) break;   // reads are start-sorted and a row starts at or before sso (:312-315): nothing further on
            bool row = false, act = false;
            uint64_t hap = 0;
            if (in && start <= sso) {
                const AdmEntry a = d.adm[uint64_t(e.adm_off) + (ri - e.read_lo)];
                const uint32_t end = d.r_end[gi];
                row = a.ord <= si && (is_rev || end >= splice_end);   // inserted, and not cleaned up since (:259-278)
                if (row) {
                    const uint32_t rvl = d.r_varlo[gi];
                    uint64_t sup[W], dirty[W];
#pragma unroll
                    for (int w = 0; w < W; w++) {
                        sup[w] = d.r_sup[uint64_t(gi) * W + w];
                        dirty[w] = d.r_lq[uint64_t(gi) * W + w] | (sup[w] & bit_range(e.sl_f_lo, e.sl_f_hi, rvl + 64u * w));
                    }
                    uint32_t slo, shi;
                    tr_to_f(e, is_rev, a.seen_lo, col_hi, slo, shi);
                    act = !mask_hits<W>(dirty, slo, shi, rvl);
                    if (act && ncols) {
                        const uint64_t sb = mask_extract<W>(sup, flo, rvl) & cmask;
                        hap = is_rev ? sb : (__brevll(sb) >> (64 - ncols));
                    }
                }
            }
            nrows += __popcll(__ballot(row));
            zero_here |= act && hap == 0;
            if (act) {
                const unsigned long long key1 = hap + 1ull;
                uint32_t slot = slot_of_key(hap);
                bool done = false;
                for (uint32_t probe = 0; probe < K2D_TABLE && !done; probe++) {
                    const unsigned long long old = atomicCAS(&tkey[slot], 0ull, key1);
                    if (old == 0ull) atomicAdd(&ng_lds, 1u);
                    if (old == 0ull || old == key1) { atomicAdd(&tcnt[slot], 1u); done = true; }
                    slot = (slot + 1) & (K2D_TABLE - 1);
                }
                if (!done) sticky_err |= WD_HAP_OVERFLOW;
            }
        }
        __syncthreads();
        const bool has_zero = __ballot(zero_here) != 0;
        const uint32_t n_distinct = ng_lds;
        // the distinct words, compacted and sorted
        uint32_t n = 0;
        for (uint32_t k0 = 0; k0 < K2D_TABLE; k0 += 64) {
            const unsigned long long kk = tkey[k0 + lane];
            const uint64_t m = __ballot(kk != 0ull);
            if (kk != 0ull) skey[n + lanes_below(m, lane)] = kk - 1ull;
            n += __popcll(m);
        }
        uint32_t N = 64;
        while (N < n) N <<= 1;
        for (uint32_t i = n + lane; i < N; i += 64) skey[i] = ~0ull;
        __syncthreads();
        bitonic_sort_wave<uint64_t>(skey, N, lane);
        const uint32_t lead = has_zero ? 0u : 1u;   // the zero-count reference group goes first (:429-431)
        const uint32_t ng = n_distinct + lead;
        unsigned long long gb = 0;
        if (lane == 0) gb = atomicAdd(gcur, (unsigned long long)ng);
        const uint64_t gbase_rel = (uint64_t(rdlane(uint32_t(gb >> 32), 0)) << 32) | rdlane(uint32_t(gb), 0);
        const bool can_write = gbase_rel + ng <= gpart_size;
        uint32_t werr = sticky_err;
        if (!can_write) werr |= WD_GROUP_OVERFLOW;
        const uint64_t gbase = gpart_lo + gbase_rel;
        const bool need_all = (st.flags & SF_NEED_RECS) != 0;
        for (uint32_t g0 = 0; g0 < ng; g0 += 64) {
            const uint32_t g = g0 + lane;
            const bool on = g < ng;
            uint64_t key = 0;
            uint32_t cnt = 0;
            if (on && g >= lead) {
                key = skey[g - lead];
                uint32_t slot = slot_of_key(key);
                while (tkey[slot] != key + 1ull) slot = (slot + 1) & (K2D_TABLE - 1);   // it is there
                cnt = tcnt[slot];
            }
            const bool need = on && can_write && (need_all || (key & som_mask) != 0);
            const uint64_t nm = __ballot(need);
            const uint32_t nneed = __popcll(nm);
            if (nneed && rec_pos + nneed > rec_end) {
                unsigned long long base = 0;
                const uint32_t want = max(64u, nneed);
                if (lane == 0) base = atomicAdd(rcur, (unsigned long long)want);
                const uint32_t blo = rdlane(uint32_t(base), 0), bhi = rdlane(uint32_t(base >> 32), 0);
                rec_pos = rpart_lo + ((uint64_t(bhi) << 32) | blo);
                rec_end = rec_pos + want;
            }
            uint32_t rec = 0xFFFFFFFFu;
            if (need) {
                const uint64_t r = rec_pos + lanes_below(nm, lane);
                if (r < rpart_hi) rec = uint32_t(r);
            }
            if (nneed && rec_pos + nneed > rpart_hi) sticky_err |= WD_REC_OVERFLOW;
            rec_pos += nneed;
            if (on && can_write) {
                Group G; G.hap = key; G.count = cnt; G.aux = 0;
                d.groups[gbase + g] = G;
            }
            k3_enqueue(d, part, on && can_write, gbase + g, win, rec);
        }
        if (lane == 0) {
            WinDyn wd;
            wd.group_off = uint32_t(gbase);
            wd.ngroups = ng;
            wd.nrows = nrows;
            wd.flags = WD_DONE | werr;
            d.win_dyn[win] = wd;
            ng_lds = 0;
        }
        __syncthreads();
        for (uint32_t k = lane; k < K2D_TABLE; k += 64) { tkey[k] = 0; tcnt[k] = 0; }   // clean for the next window
        __syncthreads();
    }
    if (sticky_err && lane == 0) atomicOr(d.err, sticky_err);
}

// ====================================================================== K2 (normal)
// `microphaser normal` replays the same schedule on a different matrix (reference: src/normal_microphasing.rs:218-339):
// no qualities, frames or `contains`, and push_read numbers the columns it finds OLDEST = bit 0 while extend_right
// shifts NEWEST into bit 0. Without `contains` a read is pushed again at every step whose candidate key range and window
// it satisfies, so on the '-' strand (whole range re-scanned every step) the matrix holds ~(read length - window) copies
// of each read. Copies pushed while the column set did not change ("a column epoch") are identical and stay identical;
// copies from different epochs differ by a bit permutation. The kernel therefore keeps
//   * ONE lane slot per live read (listed once per transcript by the planner),
//   * per live epoch its "recipe" (which variant each haplotype bit holds - a ring that shares head / length with the
//     column deque: extend_right appends, shrink_left drops the high bits) and its run of windows (sso range, window
//     length, candidate key range),
// and at a printing step derives, per (read, epoch), the number of live copies in closed form and the haplotype word
// from the K1 support mask; haplotypes are counted in an LDS hash table and written in ascending key order.
// Two sizes of the per-wave tables: the small one (128 live column epochs of a transcript, 512 haplotype hash slots per window: 30 KB of
// LDS, 5 waves per CU) serves every ordinary exome; a batch that overflows it is run again with the large one (512 epochs, 2048
// slots: 120 KB, one wave per CU) - slower, but dense long-read data no longer fails.
struct EpochMeta { int32_t xmin, xmax; int32_t w, range; };

template <int RPL, int K2N_EPOCHS_, int K2N_TABLE_>
__global__ __launch_bounds__(64) void k2n_window_replay(DeviceBatch d) {
    constexpr uint32_t K2N_EPOCHS = K2N_EPOCHS_;       // live column epochs of one transcript (ring)
    constexpr uint32_t K2N_TABLE = K2N_TABLE_;         // haplotype hash slots per window
    constexpr uint32_t K2N_GROUP_CHUNK = K2N_TABLE_;   // group slots per allocation (>= K2N_TABLE)
    constexpr uint32_t K2N_HASH_SHIFT = K2N_TABLE_ == 512 ? 55 : 53;   // top 9 / 11 bits of the multiplicative hash
    const uint32_t lane = threadIdx.x;
    // this wave's output allocator (kernels.hpp NPART)
    const uint32_t part = blockIdx.x & (NPART - 1);
    unsigned long long* const gcur = d.cursors + part * 32;
    unsigned long long* const rcur = gcur + 16;
    const uint64_t gpart_lo = uint64_t(part) << d.group_part_log2, gpart_hi = uint64_t(part + 1) << d.group_part_log2;
    const uint64_t rpart_lo = uint64_t(part) << d.rec_part_log2, rpart_hi = uint64_t(part + 1) << d.rec_part_log2;
    const SegDev S = d.segs[d.seg_order[blockIdx.x]];
    const TxDev T = d.tx[S.tx];
    const uint32_t rbase = d.g_read_off[T.gene];
    const uint32_t vbase = d.g_var_off[T.gene];
    const bool is_rev = T.strand != 0;
    const uint32_t W = d.mask_words;

    __shared__ uint16_t colf[64];                     // column deque: gene-relative forward variant index
    __shared__ uint16_t recipe[K2N_EPOCHS][64];
    __shared__ EpochMeta emeta[K2N_EPOCHS];
    __shared__ unsigned long long tkey[K2N_TABLE];    // haplotype + 1 (0 = free)
    __shared__ uint32_t tcnt[K2N_TABLE];
    __shared__ unsigned long long skey[K2N_TABLE];    // compacted (haplotype, count) of the current window
    __shared__ uint32_t scnt[K2N_TABLE];
    __shared__ uint32_t ng_lds;
    for (uint32_t k = lane; k < K2N_TABLE; k += 64) { tkey[k] = 0; tcnt[k] = 0; }
    if (lane == 0) ng_lds = 0;
    __syncthreads();

    bool occ[RPL];
    int32_t rs[RPL], re[RPL];
    uint32_t rvl[RPL], rcov[RPL], ridx[RPL];
    uint64_t msup[RPL];
#pragma unroll
    for (int r = 0; r < RPL; r++) { occ[r] = false; rs[r] = re[r] = 0; rvl[r] = rcov[r] = ridx[r] = 0; msup[r] = 0; }

    uint32_t ncols = 0, head = 0, ep_head = 0, n_ep = 0;
    uint64_t chunk_pos = 0, chunk_end = 0, rec_pos = 0, rec_end = 0, n_groups_tx = 0;
    uint32_t sticky_err = 0;

    // supports_variant(read of slot r, variant f) from the K1 mask (reference: normal_microphasing.rs:43-78)
    auto support = [&](int r, uint32_t f) -> bool {
        const uint32_t b = f - rvl[r];
        if (f >= rvl[r] && b < rcov[r]) {
            if (W == 1) return (msup[r] >> b) & 1;
            return (d.r_sup[uint64_t(ridx[r]) * W + (b >> 6)] >> (b & 63)) & 1;
        }
        const uint32_t kind = d.v_info[vbase + f] & VI_KIND_MASK;   // outside the read: only an indel can match (any op of that length)
        if (kind != 0) {
            const uint32_t want = kind == 1 ? 1u : 2u, vlen = d.v_len[vbase + f];
            const uint32_t* cig = d.cigar_pool + d.r_cigoff[ridx[r]];
            const uint32_t ncig = d.r_ncig[ridx[r]];
            for (uint32_t c = 0; c < ncig; c++)
                if ((cig[c] & 0xF) == want && (cig[c] >> 4) == vlen) return true;
        }
        return false;
    };
    // live copies of the read in slot r that were pushed during epoch m, at the step with splice_side_offset `sso`:
    // pushed at window x <=> start in [x - range, x] and end >= x + w (push_read :301-331 + the candidate range :942-967);
    // '-': cleanup_reads(sso) drops every copy of a read with start >= sso before the pushes of the step (:1001)
    auto copies = [&](int r, const EpochMeta& m, bool is_current, int32_t sso) -> uint32_t {
        int32_t lo = max(m.xmin, rs[r]);
        int32_t hi = min(m.xmax, min(rs[r] + m.range, re[r] - m.w));
        if (is_rev && rs[r] >= sso) {
            if (!is_current) return 0;
            return (lo <= sso && sso <= hi) ? 1u : 0u;
        }
        return hi >= lo ? uint32_t(hi - lo + 1) : 0u;
    };

    if (S.init_cols) {   // the columns that were alive when the previous exon ended: the segment's initial deque
        const Step st0 = d.steps[S.step_off];
        const uint32_t x = st0.col_hi - st0.n_add;
        for (uint32_t a = lane; a < S.init_cols; a += 64) {
            const uint32_t tr = x - S.init_cols + a;
            colf[a & 63] = uint16_t(is_rev ? d.v_rev2fwd[vbase + tr] : tr);
        }
        ncols = S.init_cols;
        __syncthreads();
    }

    for (uint32_t s0 = 0; s0 < S.n_steps; s0 += 64) {
        const uint32_t nb = min(64u, S.n_steps - s0);
        uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0, w5 = 0, w6 = 0;
        if (lane < nb) {
            const uint32_t* sp = reinterpret_cast<const uint32_t*>(d.steps + (S.step_off + s0 + lane));
            w0 = sp[0]; w1 = sp[1]; w2 = sp[2]; w3 = sp[3]; w4 = sp[4]; w5 = sp[5];
            w6 = d.step_aux[S.step_off + s0 + lane];
        }
        for (uint32_t i = 0; i < nb; i++) {
            const int32_t sso = int32_t(rdlane(w0, i));
            const uint32_t cand_lo = rdlane(w1, i), col_hi = rdlane(w2, i), win = rdlane(w3, i);
            const uint32_t p4 = rdlane(w4, i), p5 = rdlane(w5, i), ax = rdlane(w6, i);
            const uint32_t cand_n = p4 & 0xFFFF, wlen = (p4 >> 16) & 0xFF, n_del = p4 >> 24;
            const uint32_t n_add = p5 & 0xFF, sflags = (p5 >> 8) & 0xFF;
            const int32_t splice_end = sso + int32_t(wlen);
            const int32_t range = int32_t(ax & 0x7FFF);
            const bool new_epoch = (ax & 0x8000u) != 0;

            // ---- cleanup_reads (:286-299): a slot is released once no copy of its read can exist or be pushed any more
#pragma unroll
            for (int r = 0; r < RPL; r++)
                if (occ[r] && (is_rev ? rs[r] > sso : re[r] < splice_end)) occ[r] = false;
            // ---- shrink_left (:239-252): every haplotype keeps its low ncols bits
            if (n_del) { ncols -= n_del; head = (head + n_del) & 63; }
            // ---- newly listed reads
            if (cand_n) {
                uint32_t left = cand_n, c = cand_lo;
#pragma unroll
                for (int r = 0; r < RPL; r++) {
                    if (left) {
                        const bool empty = !occ[r];
                        const uint64_t freem = __ballot(empty);
                        const uint32_t rank = lanes_below(freem, lane);
                        if (empty && rank < left) {
                            const uint32_t gi = rbase + c + rank;
                            ridx[r] = gi;
                            rs[r] = int32_t(d.r_pos[gi]);
                            re[r] = int32_t(d.r_end[gi]);
                            rvl[r] = d.r_varlo[gi];
                            rcov[r] = d.r_ncov[gi];
                            if (W == 1) msup[r] = d.r_sup[gi];
                            occ[r] = true;
                        }
                        const uint32_t took = min(uint32_t(__popcll(freem)), left);
                        c += took;
                        left -= took;
                    }
                }
                if (left) sticky_err |= WD_ROW_OVERFLOW;
            }
            // ---- the epoch this step's pushes belong to
            __syncthreads();
            if (new_epoch || n_ep == 0) {
                if (n_ep == K2N_EPOCHS) { sticky_err |= WD_EPOCH_OVERFLOW; ep_head = (ep_head + 1) & (K2N_EPOCHS - 1); n_ep--; }
                const uint32_t e = (ep_head + n_ep) & (K2N_EPOCHS - 1);
                if (lane < ncols) recipe[e][(head + lane) & 63] = colf[(head + ncols - 1 - lane) & 63];  // oldest column = bit 0 (:317-319)
                if (lane == 0) { EpochMeta m; m.xmin = sso; m.xmax = sso; m.w = int32_t(wlen); m.range = range; emeta[e] = m; }
                n_ep++;
            } else if (lane == 0) {
                const uint32_t e = (ep_head + n_ep - 1) & (K2N_EPOCHS - 1);
                if (is_rev) emeta[e].xmin = sso; else emeta[e].xmax = sso;
            }
            // ---- extend_right (:254-284): every copy shifts left, the new columns enter at bit 0
            for (uint32_t a = 0; a < n_add; a++) {
                const uint32_t tr = col_hi - n_add + a;
                const uint32_t f = is_rev ? d.v_rev2fwd[vbase + tr] : tr;
                const uint32_t slot = (head + ncols) & 63;
                if (lane == 0) colf[slot] = uint16_t(f);
                for (uint32_t k = lane; k < n_ep; k += 64) recipe[(ep_head + k) & (K2N_EPOCHS - 1)][slot] = uint16_t(f);
                ncols++;
            }
            __syncthreads();
            // ---- retire the oldest epochs nothing alive was pushed in
            while (n_ep > 1) {
                const EpochMeta m = emeta[ep_head];
                bool any = false;
#pragma unroll
                for (int r = 0; r < RPL; r++) any |= occ[r] && copies(r, m, false, sso) != 0;
                if (__ballot(any)) break;
                ep_head = (ep_head + 1) & (K2N_EPOCHS - 1);
                n_ep--;
            }
            if (!(sflags & SF_PRINT)) continue;

            // ---- count phase of print_haplotypes (:352-390)
            uint32_t nrows_l = 0;
            for (uint32_t k = 0; k < n_ep; k++) {
                const uint32_t e = (ep_head + k) & (K2N_EPOCHS - 1);
                const EpochMeta m = emeta[e];
                uint32_t n[RPL];
                bool any = false;
#pragma unroll
                for (int r = 0; r < RPL; r++) { n[r] = occ[r] ? copies(r, m, k + 1 == n_ep, sso) : 0u; any |= n[r] != 0; }
                if (!__ballot(any)) continue;
                uint64_t hv[RPL];
#pragma unroll
                for (int r = 0; r < RPL; r++) hv[r] = 0;
                for (uint32_t p = 0; p < ncols; p++) {
                    const uint32_t f = recipe[e][(head + p) & 63];
                    const uint32_t bit = ncols - 1 - p;
#pragma unroll
                    for (int r = 0; r < RPL; r++)
                        if (n[r] && support(r, f)) hv[r] |= 1ull << bit;
                }
#pragma unroll
                for (int r = 0; r < RPL; r++)
                    if (n[r]) {
                        nrows_l += n[r];
                        const unsigned long long key1 = hv[r] + 1ull;
                        uint32_t slot = uint32_t((hv[r] * 0x9E3779B97F4A7C15ull) >> K2N_HASH_SHIFT) & (K2N_TABLE - 1);
                        bool done = false;
                        for (uint32_t probe = 0; probe < K2N_TABLE && !done; probe++) {
                            const unsigned long long old = atomicCAS(&tkey[slot], 0ull, key1);
                            if (old == 0ull) atomicAdd(&ng_lds, 1u);
                            if (old == 0ull || old == key1) { atomicAdd(&tcnt[slot], n[r]); done = true; }
                            slot = (slot + 1) & (K2N_TABLE - 1);
                        }
                        if (!done) sticky_err |= WD_HAP_OVERFLOW;
                    }
            }
            __syncthreads();
            uint32_t nrows = nrows_l;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) nrows += __shfl_xor(nrows, off, 64);
            // compact the table (and leave it clean for the next window)
            uint32_t ng = 0;
            for (uint32_t s = 0; s < K2N_TABLE / 64; s++) {
                const uint32_t idx = s * 64 + lane;
                const unsigned long long kk = tkey[idx];
                const bool o = kk != 0ull;
                const uint64_t m = __ballot(o);
                if (o) {
                    const uint32_t pos = ng + lanes_below(m, lane);
                    skey[pos] = kk - 1ull;
                    scnt[pos] = tcnt[idx];
                    tkey[idx] = 0ull;
                    tcnt[idx] = 0;
                }
                ng += uint32_t(__popcll(m));
            }
            if (ng == 0) {  // no read covers the window: the reference haplotype stands in with count 0 (:388-390)
                if (lane == 0) { skey[0] = 0ull; scnt[0] = 0; }
                ng = 1;
            }
            if (lane == 0) ng_lds = 0;
            __syncthreads();
            uint32_t werr = sticky_err;
            // group slots + one HapRec slot per haplotype (every haplotype of every window can be emitted in this mode)
            if (chunk_pos + ng > chunk_end) {
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(gcur, (unsigned long long)K2N_GROUP_CHUNK);
                const uint32_t blo = rdlane(uint32_t(base), 0), bhi = rdlane(uint32_t(base >> 32), 0);
                chunk_pos = gpart_lo + ((uint64_t(bhi) << 32) | blo);
                chunk_end = chunk_pos + K2N_GROUP_CHUNK;
            }
            const bool can_write = chunk_end <= gpart_hi;
            if (!can_write) werr |= WD_GROUP_OVERFLOW;
            if (can_write && rec_pos + ng > rec_end) {
                const uint32_t want = max(uint32_t(REC_CHUNK), ng);
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(rcur, (unsigned long long)want);
                const uint32_t blo = rdlane(uint32_t(base), 0), bhi = rdlane(uint32_t(base >> 32), 0);
                rec_pos = rpart_lo + ((uint64_t(bhi) << 32) | blo);
                rec_end = rec_pos + want;
            }
            if (can_write && rec_pos + ng > rpart_hi) sticky_err |= WD_REC_OVERFLOW;
            const uint64_t gbase = chunk_pos;
            if (can_write) {
                for (uint32_t g0 = 0; g0 < ng; g0 += 64) {
                    const uint32_t g = g0 + lane;
                    if (g < ng) {
                        const unsigned long long key = skey[g];
                        uint32_t rank = 0;   // ascending key order = the reference's VecMap iteration order (:394)
                        for (uint32_t j = 0; j < ng; j++) rank += skey[j] < key ? 1u : 0u;
                        Group G; G.hap = key; G.count = scnt[g]; G.aux = 0;
                        d.groups[gbase + rank] = G;
                    }
                    // (the slots [gbase, gbase + ng) in any order; slot gbase + x owns record rec_pos + x)
                    k3_enqueue(d, part, g < ng, gbase + g, win, rec_pos + g < rpart_hi ? uint32_t(rec_pos + g) : 0xFFFFFFFFu);
                }
                rec_pos += ng;
                chunk_pos += ng;
                n_groups_tx += ng;
            }
            if (lane == 0) {
                WinDyn wd;
                wd.group_off = uint32_t(gbase);
                wd.ngroups = ng;
                wd.nrows = nrows;
                wd.flags = WD_DONE | werr;
                d.win_dyn[win] = wd;
            }
            __syncthreads();
        }
    }
    if (sticky_err) atomicOr(d.err, sticky_err);
}

// ====================================================================== K3
// One thread per (window, haplotype) group. Per-thread byte strings live in LDS slots with an odd
// dword stride (bank-conflict free for "same offset in every lane" byte traffic); nothing is
// spilled to scratch: the reference window is staged into LDS with aligned dword loads, the SHA-1
// message is streamed through a 64-byte LDS block buffer and compressed with a fully unrolled,
// register-resident schedule.
__device__ __forceinline__ bool is_upper(uint8_t c) { return c >= 'A' && c <= 'Z'; }
__device__ __forceinline__ uint8_t to_lower(uint8_t c) { return is_upper(c) ? uint8_t(c + 32) : c; }
__device__ __forceinline__ uint8_t to_upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? uint8_t(c - 32) : c; }
__device__ __forceinline__ uint32_t rol32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

// K3 -> K3b: the records that need a SHA-1 id are appended to a dense list, one wave-aggregated atomic per wave on the
// wave's own allocator (same NPART scheme as the output slots; list p lives at want_recs[p << rec_part_log2 ...]).
__device__ __forceinline__ void append_wanted(const DeviceBatch& d, bool want, uint32_t rec_slot) {
    const uint64_t m = __ballot(want);
    if (!m) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t leader = uint32_t(__builtin_ctzll(m));
    const uint32_t part = (blockIdx.x + blockIdx.y) & (NPART - 1);
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(d.cursors + part * 32 + 24, (unsigned long long)__popcll(m));
    const uint64_t b0 = (uint64_t(rdlane(uint32_t(base >> 32), leader)) << 32) | rdlane(uint32_t(base), leader);
    if (want) {
        const uint64_t off = b0 + lanes_below(m, lane);
        if (off < (1ull << d.rec_part_log2)) d.want_recs[(uint64_t(part) << d.rec_part_log2) + off] = rec_slot;
        else atomicOr(d.err, WD_REC_OVERFLOW);
    }
}

constexpr int K3_THREADS = 64;                             // one wave per workgroup: LDS granularity 10.5 KB -> 15 waves/CU
constexpr int K3_REFCAP = 36;                              // staged reference bytes (31-nt window + 3 alignment + 2)
template <int CAP> struct K3Cfg {
    static constexpr int SLOT_BYTES = K3_REFCAP + 2 * CAP;  // ref | seq | germ  (132 bytes at CAP = 48)
    static constexpr int SLOT_DW = (SLOT_BYTES / 4) | 1;    // odd dword stride (33 at CAP = 48)
};

// BUF = false: the message streams through a 16-word block buffer and is compressed as it goes. BUF = true: the whole padded
// message is first laid out in a larger per-thread buffer and compressed afterwards in a wave-uniform loop - with variable
// length decimal text the lanes reach their block boundaries at different feeds, and the streaming form then executes the
// (fully unrolled, ~800 instruction) compression once per distinct boundary instead of once per block.
template <bool BUF>
struct ShaStreamT {
    uint32_t h0, h1, h2, h3, h4;
    uint64_t q;       // byte queue (big-endian, low `nq` bytes valid)
    uint32_t nq, widx, total;
    uint32_t* blk;    // this thread's 16-word block buffer in LDS
    __device__ __forceinline__ void init(uint32_t* b) {
        h0 = 0x67452301u; h1 = 0xEFCDAB89u; h2 = 0x98BADCFEu; h3 = 0x10325476u; h4 = 0xC3D2E1F0u;
        q = 0; nq = 0; widx = 0; total = 0; blk = b;
    }
    __device__ void compress(uint32_t off = 0) {
        uint32_t w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) w[i] = blk[off + i];
        uint32_t a = h0, b = h1, c = h2, d = h3, e = h4;
#pragma unroll
        for (int i = 0; i < 80; i++) {
            uint32_t wi;
            if (i < 16) wi = w[i];
            else {
                wi = rol32(w[(i + 13) & 15] ^ w[(i + 8) & 15] ^ w[(i + 2) & 15] ^ w[i & 15], 1);
                w[i & 15] = wi;
            }
            uint32_t f, k;
            if (i < 20) { f = (b & c) | (~b & d); k = 0x5A827999u; }
            else if (i < 40) { f = b ^ c ^ d; k = 0x6ED9EBA1u; }
            else if (i < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDCu; }
            else { f = b ^ c ^ d; k = 0xCA62C1D6u; }
            uint32_t t = rol32(a, 5) + f + e + k + wi;
            e = d; d = c; c = rol32(b, 30); b = a; a = t;
        }
        h0 += a; h1 += b; h2 += c; h3 += d; h4 += e;
    }
    // append n <= 5 bytes given as a big-endian integer (at most 3 bytes are queued, so the queue holds <= 8)
    __device__ __forceinline__ void feed(uint64_t v, uint32_t n) {
        q = (q << (8 * n)) | v;
        nq += n;
        total += n;
        if (nq >= 4) {
            nq -= 4;
            blk[widx++] = uint32_t(q >> (8 * nq));
            if constexpr (!BUF) { if (widx == 16) { compress(); widx = 0; } }
            if (nq >= 4) {
                nq -= 4;
                blk[widx++] = uint32_t(q >> (8 * nq));
                if constexpr (!BUF) { if (widx == 16) { compress(); widx = 0; } }
            }
        }
    }
    __device__ void feed_dec(uint32_t v) {  // decimal digits of v, most significant first, fed in chunks of <= 5 characters
        const uint32_t hi = v / 100000u, lo = v - hi * 100000u;
        auto chunk = [&](uint32_t x, bool pad) {   // x < 100000; pad: always five digits (the low half below a non-zero high half)
            uint64_t txt = 0;
            uint32_t n = 0;
            const uint32_t dg[5] = {x / 10000u, (x / 1000u) % 10u, (x / 100u) % 10u, (x / 10u) % 10u, x % 10u};
#pragma unroll
            for (int k = 0; k < 5; k++)
                if (pad || n || dg[k] || k == 4) { txt = (txt << 8) | ('0' + dg[k]); n++; }
            feed(txt, n);
        };
        if (hi) { chunk(hi, false); chunk(lo, true); }
        else chunk(lo, false);
    }
    __device__ void finish() {
        const uint32_t bits = total * 8;
        feed(0x80, 1);
        if (nq) feed(0, 4 - nq);                 // complete the current word
        while ((widx & 15) != 14) feed(0, 4);    // zero words up to the 64-bit length field
        feed(0, 4);
        feed(bits, 4);
    }
};

__device__ __forceinline__ bool stop_codon_at(const uint8_t* s, uint32_t c, bool fwd) {
    uint8_t a = s[c], b = s[c + 1], e = s[c + 2];
    if (fwd) return a == 'T' && ((b == 'G' && e == 'A') || (b == 'A' && (e == 'G' || e == 'A')));
    return (a == 'T' && b == 'C' && e == 'A') || (a == 'C' && b == 'T' && e == 'A') || (a == 'T' && b == 'T' && e == 'A');
}

// ---- haplotype ids: id = sha1(format!("{:?}{}{}", seq, transcript.id, offset))[..15]   (reference: src/microphasing.rs:667-675)
constexpr uint32_t K3B_BUF_WORDS = 48;   // three SHA-1 blocks per thread (a 27..31-nt window + an id of <= 17 characters always fits)
// decimal text of a byte value followed by ", ", packed big-endian: text << 8 | length (256 entries, filled by the workgroup)
__device__ __forceinline__ void fill_byte_text(uint64_t* byte_text, uint32_t n_threads) {
    for (uint32_t v = threadIdx.x; v < 256; v += n_threads) {
        uint64_t txt;
        uint32_t n;
        if (v >= 100) { txt = (uint64_t('0' + v / 100) << 16) | (uint64_t('0' + (v / 10) % 10) << 8) | ('0' + v % 10); n = 3; }
        else if (v >= 10) { txt = (uint64_t('0' + v / 10) << 8) | ('0' + v % 10); n = 2; }
        else { txt = '0' + v; n = 1; }
        txt = (txt << 16) | (uint64_t(',') << 8) | ' ';
        byte_text[v] = (txt << 8) | (n + 2);
    }
}
// The first 60 bits of the id of one haplotype per lane (wave-level: every lane of the wave calls it, `active` lanes get an id). The
// sequence comes in registers (sq: SEQ_CAP / 4 dwords), the transcript id as its first 20+ characters in six aligned dwords (idw; longer
// ids: the rest byte by byte from the pool). The padded message (<= 3 blocks for every 27..31-nt window) is first laid out in the lane's LDS buffer
// `blk` (K3B_BUF_WORDS + 1 dwords) - decimal text from the 256-entry table, id and offset fed four / five characters at a time - and then
// compressed in a wave-uniform loop with a fully unrolled, register-resident schedule. Streaming the compression executes the
// ~800-instruction round function once per distinct block boundary in the wave, because lanes with different numbers of three-digit
// bytes reach their boundaries at different feeds (kept for messages that do not fit the buffer).
template <int SEQ_CAP>
__device__ __forceinline__ uint64_t haplotype_id60(const DeviceBatch& d, bool active, const uint32_t* sq, uint32_t seq_len, uint32_t id_off, uint32_t id_len,
                                                   const uint32_t* idw, uint32_t id_mis, uint32_t win_sso, uint32_t* blk, const uint64_t* byte_text) {
    auto feed_message = [&](auto& sh) {
        sh.feed('[', 1);
#pragma unroll
        for (uint32_t k0 = 0; k0 < uint32_t(SEQ_CAP); k0 += 4) {
            if (k0 >= seq_len) break;
            const uint32_t dw = sq[k0 >> 2];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const uint32_t k = k0 + b;
                if (k < seq_len) {
                    // "{:?}" of a Vec<u8>: decimal value, then ", " unless it is the last element - one feed of <= 5 bytes
                    const uint64_t e = byte_text[(dw >> (8 * b)) & 0xFF];   // text << 8 | length, incl. the separator
                    uint64_t txt = e >> 8;
                    uint32_t n = uint32_t(e & 0xFF);
                    if (k + 1 >= seq_len) { txt >>= 16; n -= 2; }
                    sh.feed(txt, n);
                }
            }
        }
        sh.feed(']', 1);
#pragma unroll
        for (uint32_t k = 0; k < 20; k += 4) {   // transcript id, four characters per feed: from the prefetched dwords
            if (k < id_len) {
                const uint32_t n = min(4u, id_len - k);
                const uint32_t le = __builtin_amdgcn_alignbyte(idw[k / 4 + 1], idw[k / 4], id_mis);   // characters k .. k + 3, first one in the low byte
                const uint32_t be = __builtin_bswap32(le) >> (8 * (4 - n));                           // big-endian, the first n of them
                sh.feed(be, n);
            }
        }
        for (uint32_t k = 20; k < id_len; k += 4) {
            uint64_t txt = 0;
            const uint32_t n = min(4u, id_len - k);
            for (uint32_t c = 0; c < n; c++) txt = (txt << 8) | d.str_pool[id_off + k + c];
            sh.feed(txt, n);
        }
        sh.feed_dec(win_sso);
        sh.finish();
    };
    uint32_t o0 = 0, o1 = 0;
    // <= 192 bytes incl. padding (every 27..31-nt window with a transcript id of up to ~30 characters): buffered form
    const bool fits = 5 * seq_len + id_len + 20 <= K3B_BUF_WORDS * 4;
    if (__ballot(active && !fits) == 0) {
        ShaStreamT<true> sh;
        sh.init(blk);
        if (active) feed_message(sh);
        const uint32_t nblk = active ? sh.widx >> 4 : 0u;
        // wave-uniform trip count from ballots (nblk <= K3B_BUF_WORDS / 16 = 3)
        const uint32_t maxblk = __ballot(nblk >= 3) ? 3u : __ballot(nblk >= 2) ? 2u : __ballot(nblk >= 1) ? 1u : 0u;
        for (uint32_t bk = 0; bk < maxblk; bk++)
            if (bk < nblk) sh.compress(bk * 16);
        o0 = sh.h0; o1 = sh.h1;
    } else if (active) {
        ShaStreamT<false> sh;
        sh.init(blk);
        feed_message(sh);
        o0 = sh.h0; o1 = sh.h1;
    }
    return (uint64_t(o0) << 28) | (uint64_t(o1) >> 4);
}

// LIST_A: the groups of list A (k3_enqueue: those whose id will be hashed) - sequences, flags, the record AND its id in one go: the sequence is still in the lane's LDS
// slot when the id is hashed, so the record is written once, complete, and never read again on the device (a separate id kernel re-read
// 0.5 GB of records per config C pass and ran at 9 % of the HBM roofline; here its ALU work overlaps the other waves' gathers).
// !LIST_A: list B - flags, and the record where the window kernel reserved a slot (haplotypes carried into a splice merge); no id.
template <int SEQ_CAP, int LIST, int THREADS, int K3_ITEMS>   // LIST: 0 = A, 1 = B, 2 = C, 3 = D; THREADS: workgroup size; K3_ITEMS: list entries per lane, loaded together
__global__ __launch_bounds__(THREADS, LIST != 1 ? 2 : 4) void k3_window_seq(DeviceBatch d) {   // (the message buffers of A and C allow < 3 waves per SIMD anyway: registers are free there)
    constexpr int K3_THREADS = THREADS;
    constexpr bool LIST_A = LIST != 1;        // this launch hashes ids (lists A, C and D)
    constexpr bool STOPSCAN = LIST >= 2;      // ... scans for stop codons (lists C and D: A and B hold windows that cannot have one)
    constexpr bool WALK = LIST == 3;          // ... and carries the general sequence walk (list D only: the others hold simple windows)
    // the lane's LDS slot: ref | seq | germ while the sequences are built; list A re-uses it as the SHA-1 message buffer afterwards (the
    // sequences are in registers by then), so it is at least K3B_BUF_WORDS + 1 dwords there (odd stride: bank-conflict free)
    constexpr int K3_SLOT_DW = LIST_A ? ((K3Cfg<SEQ_CAP>::SLOT_DW > int(K3B_BUF_WORDS + 1) ? K3Cfg<SEQ_CAP>::SLOT_DW : int(K3B_BUF_WORDS + 1)) | 1) : K3Cfg<SEQ_CAP>::SLOT_DW;
    __shared__ uint32_t lds_slots[K3_THREADS * K3_SLOT_DW];
    __shared__ uint64_t byte_text[LIST_A ? 256 : 1];
    if constexpr (LIST_A) { fill_byte_text(byte_text, K3_THREADS); __syncthreads(); }
    const uint32_t tid = threadIdx.x;
    // blockIdx.y = the output allocator whose list this workgroup walks; the list's length is only known on the device - one scalar
    // load of its cursor (no prefix table, no search) - and the grid's x extent covers the host's upper bound of it
    const uint32_t lpart = blockIdx.y;
    const uint64_t part_size = 1ull << d.group_part_log2;
    // (the list's length and the pass's error word are needed only to VALIDATE entries: the first entries are fetched beside them, at
    //  addresses that depend on the workgroup's indices alone - one dependent load level less for every wave)
    const uint64_t n_slots = min((unsigned long long)d.cursors[lpart * 32 + (LIST == 0 ? 8 : LIST == 1 ? 12 : LIST == 2 ? 20 : 28)], (unsigned long long)part_size);
    // Every wave takes K3_ITEMS tiles at once - K3_ITEMS list entries per lane - and issues the loads of ALL of them level by level (entry,
    // then window record + haplotype word, then reference bytes + columns + transcript, then the id's characters) before it works through
    // them one after the other: the kernel's time without the SHA-1 arithmetic was 0.91 of 1.42 ms, all of it dependent-load latency at
    // the 11 waves per CU the message buffers allow, and a wave with twice the loads in flight per level needs half the waves.
    struct K3In {
        bool in_list;
        uint64_t li;
        uint64_t g;
        uint32_t w, rec_pre;
        uint64_t hap;
        WinStatic ws;
        const uint8_t* wref;
        uint32_t mis;
        uint32_t refw[K3_REFCAP / 4];
        uint32_t cp[8], ci[8];
        uint32_t id_off, id_len, id_mis;
        uint32_t idw[6];
    };
    auto load_entry = [&](K3In& I, uint64_t tile) __attribute__((always_inline)) {
        const uint64_t li = tile * K3_THREADS + tid;   // index into this allocator's list of items (k3_enqueue: A upwards, B downwards)
        I.li = li;
        const uint64_t lidx = li < part_size ? li : part_size - 1;   // (inside the allocator's sub-range whatever the list's length: validated in load_window)
        const uint64_t lpos = (uint64_t(lpart) << d.group_part_log2) + ((LIST & 1) ? part_size - 1 - lidx : lidx) + (LIST >= 2 ? d.group_cap : 0ull);
        // the item K2 listed: group slot, window, reserved record slot (one 16-byte load)
        const uint4 item = d.k3_items[lpos];
        I.g = item.x; I.w = item.y; I.rec_pre = item.z;
    };
    auto load_window = [&](K3In& I) __attribute__((always_inline)) {   // the window's record (plan.hpp WinBlob: eight 16-byte loads) and the haplotype word (both addresses come from the item)
        I.in_list = I.li < n_slots;
        if (!I.in_list) { I.g = 0; I.w = 0xFFFFFFFFu; I.rec_pre = 0xFFFFFFFFu; }
        if (I.w != 0xFFFFFFFFu && I.w >= d.n_wins) { I.w = 0xFFFFFFFFu; I.g = 0; }   // (never true for a written entry)
        if (I.g >= d.group_cap) I.g = 0;
        const uint4* bp = reinterpret_cast<const uint4*>(d.win_blobs + (I.w != 0xFFFFFFFFu ? I.w : 0u));
        uint4 q[8];
#pragma unroll
        for (int k = 0; k < 8; k++) q[k] = bp[k];
        I.hap = d.groups[I.g].hap;
        static_assert(sizeof(WinStatic) == 32 && K3_REFCAP == 36, "WinBlob unpacking");
        uint32_t wsw[8] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w};
        __builtin_memcpy(&I.ws, wsw, 32);
        I.refw[0] = q[2].x; I.refw[1] = q[2].y; I.refw[2] = q[2].z; I.refw[3] = q[2].w; I.refw[4] = q[3].x; I.refw[5] = q[3].y; I.refw[6] = q[3].z; I.refw[7] = q[3].w;
        I.refw[8] = q[4].x;   // (q[4].y = the tenth reference dword: K3_REFCAP is 36 bytes)
        const uint32_t cw[12] = {q[4].z, q[4].w, q[5].x, q[5].y, q[5].z, q[5].w, q[6].x, q[6].y, q[6].z, q[6].w, q[7].x, q[7].y};
#pragma unroll
        for (int k = 0; k < int(WINBLOB_COLS); k++) { I.cp[k] = cw[2 * k]; I.ci[k] = cw[2 * k + 1]; }
        I.id_off = q[7].z; I.id_len = q[7].w;
    };
    auto load_payload = [&](K3In& I) __attribute__((always_inline)) {
        // what the record does not hold: the window's seventh and eighth column (a wave-uniform branch: most windows have at most six)
        const WinStatic& ws = I.ws;
        I.wref = d.ref_pool + ws.ref_off;
        I.mis = ws.ref_off & 3u;   // (= the address's misalignment: the pool is 256-byte aligned)
        I.cp[6] = I.cp[7] = 0xFFFFFFFFu; I.ci[6] = I.ci[7] = 0;
        if (__ballot(I.w != 0xFFFFFFFFu && ws.ncols > WINBLOB_COLS)) {
            const uint32_t ncols = ws.ncols, last_c = ncols ? ncols - 1 : 0;
#pragma unroll
            for (uint32_t k = WINBLOB_COLS; k < 8; k++) { const WinCol* wc = d.win_cols + ws.col_off + min(k, last_c); I.cp[k] = wc->pos; I.ci[k] = wc->info; }
        }
    };
    auto load_id_text = [&](K3In& I) __attribute__((always_inline)) {   // the id's first 20+ characters as six aligned dwords, fetched together (the pool is padded); used after the walk
        I.id_mis = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) I.idw[k] = 0;
        if constexpr (LIST_A) {
            const uint8_t* idp = d.str_pool + I.id_off;
            I.id_mis = uint32_t(reinterpret_cast<uintptr_t>(idp) & 3u);
#pragma unroll
            for (int k = 0; k < 6; k++) I.idw[k] = reinterpret_cast<const uint32_t*>(idp - I.id_mis)[k];
        }
    };
    uint32_t* const slot = lds_slots + tid * K3_SLOT_DW;
    uint8_t* const refb = reinterpret_cast<uint8_t*>(slot);
    uint8_t* const seq = refb + K3_REFCAP;
    uint8_t* const germ = seq + SEQ_CAP;
    auto process = [&](K3In& I) __attribute__((always_inline)) {
    const uint64_t g = I.g;
    const uint32_t w = I.w, rec_pre = I.rec_pre;
    const bool live = w != 0xFFFFFFFFu;
    const uint32_t id_off = I.id_off, id_len = I.id_len, id_mis = I.id_mis;
    const uint32_t* const idw = I.idw;
    uint32_t sumflags = 0;
    bool need_rec = false;
    uint64_t prof_set = 0;
    bool want_id = false;
    uint32_t seq_len = 0, germ_len = 0, prof_len = 0, nvar = 0, nsom = 0, first_fs = 0, first_fs_j = 0;
    uint32_t rec_sso = 0;
    if (live) {
        const WinStatic& ws = I.ws;
        rec_sso = ws.sso;
        const uint32_t vbase = ws.vbase;
        const uint64_t hap = I.hap;
        const bool is_rev = (ws.flags & WSF_REVERSE) != 0;
        const uint32_t ncols = ws.ncols;
        const uint32_t window_end = ws.sso + ws.wlen;
        // the reference window [sso, sso + wlen), staged with aligned dword loads (load_payload), goes to the lane's slot
        const uint8_t* const wref = I.wref;
        const uint32_t mis = I.mis;
        const uint32_t* const cp = I.cp;
        const uint32_t* const ci = I.ci;
#pragma unroll
        for (int k = 0; k < K3_REFCAP / 4; k++) slot[k] = I.refw[k];
        const uint32_t staged = min(uint32_t(ws.wlen), uint32_t(K3_REFCAP) - mis);
        auto ref_at = [&](uint32_t pos) -> uint8_t {  // reference base at absolute position pos (>= sso)
            uint32_t k = pos - ws.sso;
            return k < staged ? refb[mis + k] : wref[k];
        };
        uint32_t i = ws.sso, j = 0, ns = 0, ngm = 0;
        bool indel = false, insertion = false, broke_flag = false;
        if (ws.flags & WSF_SIMPLE) {
            // SNV-only window whose columns lie at strictly increasing positions inside the window (planner-checked): the
            // walk of :473-601 visits every column exactly once, so the two sequences are the reference window with one
            // byte replaced per set column - no per-base loop. 32 bytes are copied as dwords (wlen <= 32).
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t v = __builtin_amdgcn_alignbyte(slot[k + 1], slot[k], mis);
                slot[K3_REFCAP / 4 + k] = v;
                slot[(K3_REFCAP + SEQ_CAP) / 4 + k] = v;
            }
            auto substitute = [&](uint32_t dq, uint32_t pos, uint32_t info) {   // deque column dq is set on this haplotype
                const uint32_t off = pos - ws.sso;
                const uint8_t r = refb[mis + off];
                const uint8_t alt = uint8_t(info >> VI_ALT_SHIFT);
                const uint8_t sw = is_upper(r) ? to_lower(alt) : alt;
                seq[off] = sw;
                if (info & VI_GERMLINE) germ[off] = sw; else nsom++;
                nvar++;
                prof_set |= 1ull << (is_rev ? (ncols - 1 - dq) : dq);
            };
            // the first eight columns were fetched together above (cp / ci, deque order), the rare rest one by one
            // only the first `vis` columns in walk order can ever be visited: deque columns [0, vis) on '+', [ncols - vis, ncols) on '-'
            const uint32_t vis = ws.need_recs >> WS_PREFIX_SHIFT;
            const uint32_t dlo = is_rev ? ncols - vis : 0u, dhi = is_rev ? ncols : vis;
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (uint32_t(k) >= dlo && uint32_t(k) < dhi && ((hap >> (ncols - 1 - k)) & 1)) substitute(uint32_t(k), cp[k], ci[k]);
            for (uint32_t k = max(8u, dlo); k < dhi; k++)
                if ((hap >> (ncols - 1 - k)) & 1) { const WinCol wc = d.win_cols[ws.col_off + k]; substitute(k, wc.pos, wc.info); }
            ns = ngm = ws.wlen;
            j = vis;
            // The walk's inner loop has no window bound (:479): a set SNV at the window's LAST base moves the cursor to window_end, where a
            // column sitting exactly there is applied too (one more base), and so on along a gap-free run of columns; the run ends at its
            // first column the haplotype does not set (visited, not applied). The planner leaves such windows "simple" only when the run is
            // SNVs at consecutive positions (plan.cpp); no other simple window has a column at window_end behind one at the last base.
            if (STOPSCAN && vis > 0 && vis < ncols) {   // (such a window is never WSF_NOSTOP: lists C / D only)
                const uint32_t dq_last = is_rev ? ncols - vis : vis - 1;   // the last column of the prefix, in deque order
                const WinCol lc = d.win_cols[ws.col_off + dq_last];
                if (lc.pos + 1 == window_end && ((hap >> (ncols - 1 - dq_last)) & 1)) {
                    uint32_t cnt = 0;
                    for (uint32_t wi = vis; wi < ncols; wi++) {
                        const uint32_t dq = is_rev ? ncols - 1 - wi : wi;
                        const WinCol wc = d.win_cols[ws.col_off + dq];
                        if (wc.pos != window_end + cnt) break;
                        j = wi + 1;
                        if (!((hap >> (ncols - 1 - dq)) & 1)) break;
                        const uint32_t off = uint32_t(ws.wlen) + cnt;
                        const uint8_t r = ref_at(wc.pos);
                        const uint8_t alt = uint8_t(wc.info >> VI_ALT_SHIFT);
                        const uint8_t sw = is_upper(r) ? to_lower(alt) : alt;
                        if (off < uint32_t(SEQ_CAP)) { seq[off] = sw; germ[off] = (wc.info & VI_GERMLINE) ? sw : r; }
                        ns++; ngm++; cnt++;
                        if (!(wc.info & VI_GERMLINE)) nsom++;
                        nvar++;
                        prof_set |= 1ull << wi;
                    }
                }
            }
        } else if constexpr (!WALK) {
            atomicOr(d.err, WD_INTERNAL);   // a window that needs the general walk in list A / B / C (the window kernels send those to list D)
        } else {
            uint32_t pos_j = 0xFFFFFFFFu, info_j = 0, f_j = 0;
            // the first 8 columns (in walk order) are fetched up front so their loads overlap instead of forming a dependent chain
            uint32_t cpos[8], cinfo[8];
    #pragma unroll
            for (int k = 0; k < 8; k++) {
                cpos[k] = 0xFFFFFFFFu; cinfo[k] = 0;
                if (uint32_t(k) < ncols) {
                    const uint32_t dq = is_rev ? (ncols - 1 - k) : uint32_t(k);
                    if (is_rev || dq >= 8) { const WinCol* wc = d.win_cols + ws.col_off + dq; cpos[k] = wc->pos; cinfo[k] = wc->info; }
                    else { cpos[k] = cp[k]; cinfo[k] = ci[k]; }
                }
            }
            auto load_j = [&]() {
                if (j < 8) {
                    pos_j = cpos[0]; info_j = cinfo[0];
    #pragma unroll
                    for (int k = 1; k < 8; k++) if (j == uint32_t(k)) { pos_j = cpos[k]; info_j = cinfo[k]; }
                    if (j >= ncols) pos_j = 0xFFFFFFFFu;
                } else if (j < ncols) {
                    uint32_t dq = is_rev ? (ncols - 1 - j) : j;
                    const WinCol wc = d.win_cols[ws.col_off + dq];
                    pos_j = wc.pos;
                    info_j = wc.info;
                } else {
                    pos_j = 0xFFFFFFFFu;
                }
            };
            auto col_f = [&]() { return d.win_cols[ws.col_off + (is_rev ? (ncols - 1 - j) : j)].f; };  // only indels need it
            auto push_s = [&](uint8_t c) { if (ns < SEQ_CAP) seq[ns] = c; ns++; };
            auto push_g = [&](uint8_t c) { if (ngm < SEQ_CAP) germ[ngm] = c; ngm++; };
            load_j();
            // sequence walk of print_haplotypes (:473-601); runs of plain reference bases are copied in bulk
            while (i < window_end) {
                while (j < ncols && i == pos_j) {
                    uint32_t fs = (info_j & VI_FS_MASK) >> VI_FS_SHIFT;
                    if (first_fs == 0 && fs) { first_fs = fs; first_fs_j = j; }
                    uint32_t dq = is_rev ? (ncols - 1 - j) : j;
                    uint32_t bit = ncols - 1 - dq;
                    if ((hap >> bit) & 1) {
                        uint32_t kind = info_j & VI_KIND_MASK;
                        bool germline = info_j & VI_GERMLINE;
                        uint8_t r = ref_at(i);
                        bool brk = false;
                        if (kind == 0) {
                            uint8_t alt = uint8_t(info_j >> VI_ALT_SHIFT);
                            uint8_t sw = is_upper(r) ? to_lower(alt) : alt;
                            push_g(germline ? sw : r);
                            push_s(sw);
                            i += 1;
                        } else if (kind == 1) {
                            f_j = col_f();
                            uint32_t il = d.v_len[vbase + f_j] + 1;
                            const uint8_t* ins = d.ins_pool + d.v_insoff[vbase + f_j];
                            bool up = is_upper(r);
                            for (uint32_t k = 0; k < il; k++) {
                                uint8_t c = up ? to_lower(ins[k]) : to_upper(ins[k]);
                                if (germline) push_g(c);
                                push_s(c);
                            }
                            if (!germline) indel = true;
                            insertion = true;
                            i += 1;
                        } else {
                            f_j = col_f();
                            uint32_t dl = d.v_len[vbase + f_j];
                            if (is_rev && pos_j + dl - 1 >= window_end) { brk = true; }
                            else {
                                if (germline || i == window_end - 1) push_g(r);
                                else {
                                    for (uint32_t k = 0; k < dl + 1; k++) push_g(ref_at(i + k));
                                    indel = true;
                                }
                                push_s(r);
                                i += dl + 1;
                            }
                        }
                        if (brk) { broke_flag = true; break; }
                        if (!germline) nsom++;
                        nvar++;
                        prof_set |= 1ull << j;
                    }
                    j++;
                    load_j();
                }
                if (i < window_end) {
                    // the next position at which anything other than a reference copy can happen
                    // (a stuck cursor - pos_j <= i, e.g. after the incomplete-deletion break - never matches again)
                    uint32_t stop_at = (j < ncols && pos_j > i) ? min(pos_j, window_end) : window_end;
                    for (; i < stop_at; i++) { uint8_t r = ref_at(i); push_s(r); push_g(r); }
                }
            }
        }
        seq_len = min(ns, uint32_t(SEQ_CAP));
        germ_len = min(ngm, uint32_t(SEQ_CAP));
        prof_len = j;
        // neopeptide slice and stop scan (:686-697, :42-76)
        uint32_t this_len = seq_len < ws.ewl ? seq_len : ws.ewl;
        uint32_t nlo = 0, nhi = seq_len;
        if (ws.splice_pos == 1) nlo = min(uint32_t(ws.splice_gap), seq_len);
        else if (ws.splice_pos == 0 && !insertion) nhi = this_len;
        bool stop = false;
        uint32_t nlen = nhi - nlo;
        if constexpr (!STOPSCAN) {   // lists A / B hold windows for which the planner proved that no haplotype can hold a stop (WSF_NOSTOP): no codon loop in these kernels
            if (!(ws.flags & WSF_NOSTOP)) atomicOr(d.err, WD_INTERNAL);
        } else if (nlen >= 3 && !(ws.flags & WSF_NOSTOP)) {   // WSF_NOSTOP: planner proved that no haplotype of this window can hold a stop
            if (!is_rev) {
                for (uint32_t c = 0; c + 3 <= nlen; c += 3)
                    if (stop_codon_at(seq + nlo, c, true)) { stop = true; break; }
            } else {
                for (int c = int(nlen) - 3; c >= 0; c -= 3)
                    if (stop_codon_at(seq + nlo, uint32_t(c), false)) { stop = true; break; }
            }
        }
        bool differs;
        if (!WALK || (ws.flags & WSF_SIMPLE)) differs = nsom > 0;   // a set somatic SNV always changes its byte (base or case)
        else {
            differs = seq_len != germ_len;
            if (!differs)
                for (uint32_t k = 0; k < seq_len; k++)
                    if (seq[k] != germ[k]) { differs = true; break; }
        }
        sumflags = GS_VALID | (stop ? GS_STOP : 0) | (differs ? GS_DIFFERS : 0) | (indel ? GS_INDEL : 0) |
                   (insertion ? GS_INSERTION : 0) | (broke_flag ? GS_BROKE : 0);
        need_rec = nsom > 0 || (ws.need_recs & WS_MASK) != 0;
        want_id = need_rec && (nsom > 0 || (ws.need_recs & WS_ALL_IDS));  // hashed by k3b_haplotype_ids
        if (stop && ws.splice_pos != 2 && !(ws.flags & SF_FIRST_EXON_WIN)) atomicMin(&d.tx_first_stop[ws.tx], w);
    }
    // the record slot (if any) was assigned by K2; a slot K3 turns out not to need is marked "no id"
    uint32_t recidx = 0;
    {
        // sequences -> registers (list A: the slot becomes the message buffer), id, then the whole record in one run of 16-byte stores
        // (records are 16-byte aligned: rec_stride and the header are multiples of 16)
        uint32_t sqr[2 * SEQ_CAP / 4];
#pragma unroll
        for (int k = 0; k < 2 * SEQ_CAP / 4; k++) sqr[k] = slot[K3_REFCAP / 4 + k];
        const bool has_slot = live && rec_pre != 0xFFFFFFFFu;
        const bool hash = has_slot && need_rec && want_id;
        uint64_t id60 = 0;
        if constexpr (LIST_A) {
            id60 = haplotype_id60<SEQ_CAP>(d, hash && !d.timing_skip_ids, sqr, seq_len, id_off, id_len, idw, id_mis, rec_sso, slot, byte_text);
            // the number of ids of the pass (statistics only: one wave-aggregated add without a return value)
            const uint64_t hm = __ballot(hash);
            if (hm && (tid & 63u) == uint32_t(__builtin_ctzll(hm))) atomicAdd(d.cursors + ((blockIdx.x + blockIdx.y) & (NPART - 1)) * 32 + 24, (unsigned long long)__popcll(hm));
        } else if (hash) {
            atomicOr(d.err, WD_REC_OVERFLOW);   // a haplotype that wants an id was listed without one (must not happen: the window kernels send every such group to list A)
        }
        if (has_slot) {
            uint32_t* out = reinterpret_cast<uint32_t*>(d.recs + uint64_t(rec_pre) * d.rec_stride);
            if (need_rec) {
                uint4* const out4 = reinterpret_cast<uint4*>(out);
                out4[0] = make_uint4(uint32_t(prof_set), uint32_t(prof_set >> 32), uint32_t(id60), uint32_t(id60 >> 32));
                out4[1] = make_uint4(seq_len | (germ_len << 8) | (prof_len << 16) | (nvar << 24), nsom | (first_fs << 8) | (first_fs_j << 16), w, want_id ? 1u : 0u);
                // only the 16-byte pieces that hold sequence bytes are stored (the host reads seq_len / germ_len bytes): a 27..32-nt window
                // in a batch whose capacity is 48 writes 96 of its record's 128 bytes
#pragma unroll
                for (int k = 0; k < SEQ_CAP / 16; k++) {
                    if (uint32_t(16 * k) < seq_len) out4[2 + k] = make_uint4(sqr[4 * k], sqr[4 * k + 1], sqr[4 * k + 2], sqr[4 * k + 3]);
                    if (uint32_t(16 * k) < germ_len) {
                        constexpr int G = SEQ_CAP / 16;
                        out4[2 + G + k] = make_uint4(sqr[4 * (G + k)], sqr[4 * (G + k) + 1], sqr[4 * (G + k) + 2], sqr[4 * (G + k) + 3]);
                    }
                }
                sumflags |= GS_HAS_REC | ((LIST_A && want_id) ? uint32_t(GS_ID_VALID) : 0u);
                recidx = rec_pre;
            } else {
                out[7] = 0;
            }
        } else if (live && need_rec) {
            atomicOr(d.err, WD_REC_OVERFLOW);  // the record buffer is full (K2 flagged it too), or K2's superset rule missed a haplotype (must not happen)
        }
    }
    if (live) {
        GroupSum gs;
        gs.flags = sumflags;
        gs.rec = recidx;
        d.gsum[g] = gs;
    }
    };   // process
    for (uint64_t tile0 = uint64_t(blockIdx.x) * K3_ITEMS;; tile0 += uint64_t(gridDim.x) * K3_ITEMS) {
        static_assert(K3_ITEMS == 1 || K3_ITEMS == 2, "entries per lane");
        K3In in0, in1;   // (named objects, every use spelled out: a loop over an array of them kept the array in scratch)
        load_entry(in0, tile0);
        if constexpr (K3_ITEMS == 2) load_entry(in1, tile0 + 1);
        if (tile0 * K3_THREADS >= n_slots) break;   // (checked once the entry loads are on their way)
        if (pass_overflowed(d)) return;             // (a discarded pass: its lists have holes)
        load_window(in0);
        if constexpr (K3_ITEMS == 2) load_window(in1);
        load_payload(in0);
        if constexpr (K3_ITEMS == 2) load_payload(in1);
        load_id_text(in0);
        if constexpr (K3_ITEMS == 2) load_id_text(in1);
        process(in0);
        if constexpr (K3_ITEMS == 2) process(in1);
    }
}

// K3 for `microphaser normal` (reference: src/normal_microphasing.rs:341-647): same slots and record layout, but the
// sequence walk of the normal-sample tool - haplotype bit j is tested directly, a somatic variant is skipped on a
// haplotype every row carries (:422-426), the second ALT at a position replaces the first (:429-431), a deletion moves
// window_end (:451-456), one reference base is appended after every variant run (:476), and "stop" means the FIRST
// ('+') / LAST ('-') codon of the peptide slice. germ_len is 0; the germ area's first 8 bytes hold the somatic subset of
// the variant profile.
template <int SEQ_CAP>
__device__ __forceinline__ uint32_t k3n_one(const DeviceBatch& d, uint32_t* lds_slots, uint64_t g, uint32_t item_win, uint32_t item_rec, bool valid,
                                            uint32_t& recidx) {
    constexpr int K3_SLOT_DW = K3Cfg<SEQ_CAP>::SLOT_DW;
    const uint32_t tid = threadIdx.x;
    recidx = 0;
    uint32_t* slot = lds_slots + tid * K3_SLOT_DW;
    uint8_t* refb = reinterpret_cast<uint8_t*>(slot);
    uint8_t* seq = refb + K3_REFCAP;
    uint32_t* germ_dw = slot + (K3_REFCAP + SEQ_CAP) / 4;
    const uint32_t w = valid ? item_win : 0xFFFFFFFFu;
    if (w == 0xFFFFFFFFu) return 0;
    const WinStatic ws = d.wins[w];
    const uint32_t vbase = ws.vbase;
    const Group G = d.groups[g];
    const uint64_t hap = G.hap;
    const uint32_t nrows = d.win_dyn[w].nrows;
    const bool freq_one = G.count != 0 && G.count == nrows;  // |count / nrows - 1| < EPSILON
    const bool is_rev = (ws.flags & WSF_REVERSE) != 0;
    const uint32_t ncols = ws.ncols;
    uint32_t window_end = ws.sso + ws.wlen;
    const uint8_t* wref = d.ref_pool + ws.ref_off;
    const uint32_t mis = uint32_t(reinterpret_cast<uintptr_t>(wref) & 3u);
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(wref - mis);
        const uint32_t ndw = (mis + ws.wlen + 3) >> 2;
        for (uint32_t k = 0; k < ndw && k < K3_REFCAP / 4; k++) slot[k] = src[k];
    }
    const uint32_t staged = min(uint32_t(ws.wlen), uint32_t(K3_REFCAP) - mis);
    auto ref_at = [&](uint32_t pos) -> uint8_t {
        uint32_t k = pos - ws.sso;
        return k < staged ? refb[mis + k] : wref[k];
    };
    auto col = [&](uint32_t j) { return d.win_cols[ws.col_off + (is_rev ? (ncols - 1 - j) : j)]; };
    uint32_t i = ws.sso, j = 0, ns = 0, prof_len = 0, nvar = 0, nsom = 0;
    uint64_t prof_set = 0, prof_som = 0;
    bool insertion = false;
    auto push_s = [&](uint8_t c) { if (ns < SEQ_CAP) seq[ns] = c; ns++; };
    if (ncols == 0) {
        for (; i < window_end; i++) push_s(ref_at(i));
    } else {
        WinCol cj = col(0);
        while (i < window_end) {
            while (j < ncols && i == cj.pos) {
                if (freq_one && !(cj.info & VI_GERMLINE)) {
                    j++; prof_len++;
                    if (j < ncols) cj = col(j);
                    continue;
                }
                if ((hap >> j) & 1) {
                    if (j + 1 < ncols) {
                        const WinCol nx = col(j + 1);
                        if (nx.pos == i) { j++; cj = nx; }
                    }
                    const uint32_t kind = cj.info & VI_KIND_MASK;
                    const uint8_t r = ref_at(i);
                    if (kind == 0) {
                        const uint8_t alt = uint8_t(cj.info >> VI_ALT_SHIFT);
                        push_s(is_upper(r) ? to_lower(alt) : alt);
                        i += 1;
                    } else if (kind == 1) {
                        const uint32_t il = d.v_len[vbase + cj.f] + 1;
                        const uint8_t* ins = d.ins_pool + d.v_insoff[vbase + cj.f];
                        const bool up = is_upper(r);
                        for (uint32_t k = 0; k < il; k++) push_s(up ? to_lower(ins[k]) : to_upper(ins[k]));
                        insertion = true;
                        i += 1;
                    } else {
                        const uint32_t dl = d.v_len[vbase + cj.f];
                        push_s(r);
                        i += dl + 1;
                        window_end += dl + 1;
                    }
                    if (prof_len < 64) {
                        prof_set |= 1ull << prof_len;
                        if (!(cj.info & VI_GERMLINE)) prof_som |= 1ull << prof_len;
                    }
                    if (!(cj.info & VI_GERMLINE)) nsom++;
                    nvar++;
                }
                prof_len++;
                j++;
                if (j < ncols) cj = col(j);
            }
            push_s(ref_at(i));
            i += 1;
        }
    }
    const uint32_t seq_len = min(ns, uint32_t(SEQ_CAP));
    uint32_t nlo = 0, nhi = seq_len;
    if (ws.splice_pos == 1) nlo = min(uint32_t(ws.splice_gap), seq_len);
    else if (ws.splice_pos == 0 && !insertion) nhi = min(seq_len, uint32_t(ws.ewl));
    bool stop = false;
    if (nhi - nlo >= 3) stop = is_rev ? stop_codon_at(seq, nhi - 3, false) : stop_codon_at(seq, nlo, true);
    const bool skipped = stop && ws.splice_pos != 2;   // :503-507: such a haplotype produces nothing
    uint32_t sumflags = GS_VALID | (stop ? GS_STOP : 0) | (insertion ? GS_INSERTION : 0) | (ns > uint32_t(SEQ_CAP) ? uint32_t(GS_BROKE) : 0u);
    const uint32_t slot_idx = item_rec;
    if (slot_idx != 0xFFFFFFFFu) {
        uint32_t* out = reinterpret_cast<uint32_t*>(d.recs + uint64_t(slot_idx) * d.rec_stride);
        germ_dw[0] = uint32_t(prof_som); germ_dw[1] = uint32_t(prof_som >> 32);
        out[0] = uint32_t(prof_set); out[1] = uint32_t(prof_set >> 32);
        out[2] = ws.sso; out[3] = ws.tx;   // what K3b hashes besides the sequence (it overwrites them with the id)
        out[4] = seq_len | (0u << 8) | (min(prof_len, 255u) << 16) | (min(nvar, 255u) << 24);
        out[5] = min(nsom, 255u);
        out[6] = w;
        out[7] = skipped ? 0u : 1u;
        const uint32_t* sq = slot + K3_REFCAP / 4;
#pragma unroll
        for (int k = 0; k < 2 * SEQ_CAP / 4; k++) out[8 + k] = sq[k];
        sumflags |= GS_HAS_REC | (skipped ? 0u : uint32_t(GS_ID_VALID));
        recidx = slot_idx;
    } else {
        atomicOr(d.err, WD_REC_OVERFLOW);
    }
    GroupSum gs;
    gs.flags = sumflags;
    gs.rec = recidx;
    d.gsum[g] = gs;
    return sumflags;
}
template <int SEQ_CAP>
__global__ __launch_bounds__(K3_THREADS) void k3_window_seq_normal(DeviceBatch d) {
    __shared__ uint32_t lds_slots[K3_THREADS * K3Cfg<SEQ_CAP>::SLOT_DW];
    const uint32_t lpart = blockIdx.y;   // (as in k3_window_seq: one allocator's list per grid row)
    if (pass_overflowed(d)) return;
    const uint64_t n_slots = min((unsigned long long)d.cursors[lpart * 32 + 8], 1ull << d.group_part_log2);
    for (uint64_t tile = blockIdx.x; tile * K3_THREADS < n_slots; tile += gridDim.x) {
        const uint64_t li = tile * K3_THREADS + threadIdx.x;
        const uint64_t lpos = (uint64_t(lpart) << d.group_part_log2) + li;
        const uint4 item = d.k3_items[li < n_slots ? lpos : (uint64_t(lpart) << d.group_part_log2)];   // the K2 kernels' lists: group slot, window, record slot
        uint32_t recidx;
        const uint32_t sumflags = k3n_one<SEQ_CAP>(d, lds_slots, item.x, item.y, item.z, li < n_slots, recidx);
        append_wanted(d, (sumflags & GS_ID_VALID) != 0, recidx);   // wave-level: every lane takes part
    }
}

// K3b: SHA-1 ids of the haplotype records of `microphaser normal` (dense: one thread per record of K3n's wanted lists; the somatic path
// hashes its ids inside k3_window_seq<CAP, true>).
constexpr uint32_t K3B_THREADS = 192;    // three waves share the byte_text table: 39.7 KB of LDS per workgroup -> 4 workgroups = 12 waves per CU
template <int SEQ_CAP>
__global__ __launch_bounds__(K3B_THREADS) void k3b_haplotype_ids(DeviceBatch d) {
    __shared__ uint32_t lds_blk[K3B_THREADS * (K3B_BUF_WORDS + 1)];   // odd stride: bank-conflict free
    __shared__ uint64_t byte_text[256];
    fill_byte_text(byte_text, K3B_THREADS);
    __syncthreads();
    const uint32_t wp = blockIdx.y;      // the list of wanted records this workgroup walks; its length: one scalar load (known on the device only)
    if (pass_overflowed(d)) return;
    const uint64_t n_recs = min((unsigned long long)d.cursors[wp * 32 + 24], 1ull << d.rec_part_log2);
    for (uint64_t tile = blockIdx.x; tile * K3B_THREADS < n_recs; tile += gridDim.x) {
    const uint64_t li = tile * K3B_THREADS + threadIdx.x;   // index into this list
    const bool active = li < n_recs;
    const uint64_t r = d.want_recs[(uint64_t(wp) << d.rec_part_log2) + (active ? li : 0)];
    uint32_t* rec = reinterpret_cast<uint32_t*>(d.recs + r * d.rec_stride);
    // everything the id needs sits in the record K3n wrote (header: sequence length, window offset, transcript; then the sequence):
    // one contiguous read instead of record -> window -> transcript hops, issued at once
    const uint4* rec4 = reinterpret_cast<const uint4*>(rec);
    const uint4 h0 = rec4[0], h1 = rec4[1];
    uint32_t sq[SEQ_CAP / 4];
#pragma unroll
    for (int k = 0; k < SEQ_CAP / 16; k++) { const uint4 v = rec4[2 + k]; sq[4 * k] = v.x; sq[4 * k + 1] = v.y; sq[4 * k + 2] = v.z; sq[4 * k + 3] = v.w; }
    const uint32_t seq_len = h1.x & 0xFF, win_sso = h0.z;
    const TxDev T = d.tx[active ? h0.w : 0];
    // the transcript id: its first 20+ characters as six aligned dwords, fetched together (longer ids: the rest byte by byte)
    const uint8_t* idp = d.str_pool + T.id_off;
    const uint32_t id_mis = uint32_t(reinterpret_cast<uintptr_t>(idp) & 3u);
    uint32_t idw[6];
#pragma unroll
    for (int k = 0; k < 6; k++) idw[k] = reinterpret_cast<const uint32_t*>(idp - id_mis)[k];   // (the pool is padded)
    const uint64_t id60 = haplotype_id60<SEQ_CAP>(d, active, sq, seq_len, T.id_off, T.id_len, idw, id_mis, win_sso,
                                                  lds_blk + threadIdx.x * (K3B_BUF_WORDS + 1), byte_text);
    if (active) {
        rec[2] = uint32_t(id60);
        rec[3] = uint32_t(id60 >> 32);
    }
    }   // tiles of this wave
}

// ====================================================================== launchers
#if MP_IN_PART(0)
void launch_k1_pileup_bits(const DeviceBatch& d, hipStream_t stream) {
    if (d.n_reads == 0) return;
    dim3 grid(uint32_t((uint64_t(d.n_reads) * K1_LANES + 255) / 256)), block(256);
    switch (d.mask_words) {
        case 1: hipLaunchKernelGGL(k1_pileup_bits<1>, grid, block, 0, stream, d); break;
        case 2: hipLaunchKernelGGL(k1_pileup_bits<2>, grid, block, 0, stream, d); break;
        case 3: case 4: {
            DeviceBatch d4 = d;
            if (d.mask_words == 3) throw_hip(hipErrorInvalidValue, __FILE__, __LINE__);
            hipLaunchKernelGGL(k1_pileup_bits<4>, grid, block, 0, stream, d4);
            break;
        }
        default: throw_hip(hipErrorInvalidValue, __FILE__, __LINE__);
    }
    HIP_CHECK_LAUNCH();
}

#endif
#if MP_IN_PART(1)
void launch_k2_window_replay(const DeviceBatch& d, int rows_per_lane, hipStream_t stream) {
    if (d.n_segs == 0) return;
    dim3 grid(d.n_segs), block(64);
    if (d.normal) {
#define K2N_CASE(R) case R: if (d.normal_large) hipLaunchKernelGGL((k2n_window_replay<R, 512, 2048>), grid, block, 0, stream, d); \
                            else hipLaunchKernelGGL((k2n_window_replay<R, 128, 512>), grid, block, 0, stream, d); break;
        switch (rows_per_lane) {
            K2N_CASE(1) K2N_CASE(2) K2N_CASE(4) K2N_CASE(8) K2N_CASE(16)
            default: throw_hip(hipErrorInvalidValue, __FILE__, __LINE__);
        }
#undef K2N_CASE
        HIP_CHECK_LAUNCH();
        return;
    }
    switch (rows_per_lane) {
        case 1: hipLaunchKernelGGL(k2_window_replay<1>, grid, block, 0, stream, d); break;
        case 2: hipLaunchKernelGGL(k2_window_replay<2>, grid, block, 0, stream, d); break;
        case 4: hipLaunchKernelGGL(k2_window_replay<4>, grid, block, 0, stream, d); break;
        case 8: hipLaunchKernelGGL(k2_window_replay<8>, grid, block, 0, stream, d); break;
        case 16: hipLaunchKernelGGL(k2_window_replay<16>, grid, block, 0, stream, d); break;
        default: throw_hip(hipErrorInvalidValue, __FILE__, __LINE__);
    }
    HIP_CHECK_LAUNCH();
}

#endif
#if MP_IN_PART(0)
void launch_k2_admission(const DeviceBatch& d, hipStream_t stream) {
    if (!d.n_exons_w) return;
    if (!d.n_achunks) return;
    if (d.k2a_flat && d.mask_words <= 2) {
        // (entries per lane: 1 measured 0.389 ms, 2 0.405 ms, the chunk form 0.418 ms on one box - the kernel is not short of loads in flight)
        const char* env_items = std::getenv("MP_K2A_ITEMS");
        const int items = env_items && std::atoi(env_items) == 2 ? 2 : 1;
        const uint32_t waves = uint32_t((d.n_adm + 63) / 64), grid = (waves + items - 1) / items;
        if (d.mask_words == 1) { if (items == 2) hipLaunchKernelGGL((k2a_admission_flat<1, 2>), dim3(grid), dim3(64), 0, stream, d); else hipLaunchKernelGGL((k2a_admission_flat<1, 1>), dim3(grid), dim3(64), 0, stream, d); }
        else { if (items == 2) hipLaunchKernelGGL((k2a_admission_flat<2, 2>), dim3(grid), dim3(64), 0, stream, d); else hipLaunchKernelGGL((k2a_admission_flat<2, 1>), dim3(grid), dim3(64), 0, stream, d); }
        HIP_CHECK_LAUNCH();
        return;
    }
    if (d.mask_words == 1) hipLaunchKernelGGL(k2a_admission<1>, dim3(d.n_achunks), dim3(64), 0, stream, d);
    else if (d.mask_words == 2) hipLaunchKernelGGL(k2a_admission<2>, dim3(d.n_achunks), dim3(64), 0, stream, d);
    else throw_hip(hipErrorInvalidValue, __FILE__, __LINE__);
    HIP_CHECK_LAUNCH();
}
#endif
#if MP_IN_PART(2)
template <int W>
static void launch_k2w_multi(const DeviceBatch& d, hipStream_t stream) {
    dim3 grid(d.n_wchunks_m), block(64);
    switch (d.rows_per_lane_w) {
        case 1: hipLaunchKernelGGL((k2w_window_rows_multi<1, W>), grid, block, 0, stream, d); break;
        case 2: hipLaunchKernelGGL((k2w_window_rows_multi<2, W>), grid, block, 0, stream, d); break;
        case 4: hipLaunchKernelGGL((k2w_window_rows_multi<4, W>), grid, block, 0, stream, d); break;
        case 8: hipLaunchKernelGGL((k2w_window_rows_multi<8, W>), grid, block, 0, stream, d); break;
        default: throw_hip(hipErrorInvalidValue, __FILE__, __LINE__);
    }
}
void launch_k2_window_rows(const DeviceBatch& d, hipStream_t stream) {
    if (d.n_wchunks) {
        hipLaunchKernelGGL(k2w_window_rows, dim3((d.n_wchunks + K2W_ITEMS - 1) / K2W_ITEMS), dim3(64), 0, stream, d);
        HIP_CHECK_LAUNCH();
    }
    if (d.n_wchunks_m) {   // exons with more than 64 candidate reads per window, or two mask words per read
        if (d.mask_words == 1) launch_k2w_multi<1>(d, stream);
        else if (d.mask_words == 2) launch_k2w_multi<2>(d, stream);
        else throw_hip(hipErrorInvalidValue, __FILE__, __LINE__);
        HIP_CHECK_LAUNCH();
    }
    if (d.n_wchunks_d) {   // exons with more than 512 candidate reads per window: rows streamed, any depth
        if (d.mask_words == 1) hipLaunchKernelGGL(k2w_window_rows_deep<1>, dim3(d.n_wchunks_d), dim3(64), 0, stream, d);
        else if (d.mask_words == 2) hipLaunchKernelGGL(k2w_window_rows_deep<2>, dim3(d.n_wchunks_d), dim3(64), 0, stream, d);
        else throw_hip(hipErrorInvalidValue, __FILE__, __LINE__);
        HIP_CHECK_LAUNCH();
    }
}

#endif
#if MP_IN_PART(0)
template <int STAGE>
static void launch_k2l(const DeviceBatch& d, hipStream_t stream_small, hipStream_t stream_wide, hipStream_t stream_hash) {
    // one wave per tile of 64 windows: a wave that walked several tiles would wait for its own result stores to drain before the
    // next tile's loads return (loads and stores share the in-order vmcnt counter)
    const uint32_t n_small = d.n_lane_small, n_wide = d.n_lane_mid - d.n_lane_small, n_hash = d.n_lane_all - d.n_lane_mid;
    const uint32_t lds_small = 4096 + 24 * STAGE, lds_wide = 16384 + 24 * STAGE;
    static const bool persistent = std::getenv("MP_K2L_PERSISTENT") != nullptr;   // experiments: a fixed grid that walks the tiles
    static const bool gather16 = [] { const char* e = std::getenv("MP_K2L_GATHER"); return e && std::atoi(e) == 16; }();   // rows fetched together by the gather forms (default 8)
    if (n_small) {
        const uint32_t tiles = (n_small + 63) / 64, waves = min(32u, 163840u / lds_small);
        hipLaunchKernelGGL((k2l_window_lanes<K2L_SMALL_COLS, STAGE>), dim3(persistent ? min(tiles, 256u * waves) : tiles), dim3(64), 0, stream_small, d, 0u, n_small);
        HIP_CHECK_LAUNCH();
    }
    if (n_wide) {
        // The windows of 7-8 columns are a sparse subset too (a fifth of the lane kernel's windows): their tiles span several exons and
        // need several staging passes; reading the records straight from memory is faster for them (window phase 1.15 -> 1.05 ms at
        // config C, same box); MP_K2L_WIDE_STAGED=1 brings the staged form back for comparisons.
        static const bool wide_gather = std::getenv("MP_K2L_WIDE_STAGED") == nullptr;
        const uint32_t tiles = (n_wide + 63) / 64, waves = min(32u, 163840u / lds_wide);
        if (wide_gather && gather16) hipLaunchKernelGGL((k2l_window_lanes<K2L_MAX_COLS, 0, 16>), dim3(tiles), dim3(64), 0, stream_wide, d, n_small, n_wide);
        else if (wide_gather) hipLaunchKernelGGL((k2l_window_lanes<K2L_MAX_COLS, 0>), dim3(tiles), dim3(64), 0, stream_wide, d, n_small, n_wide);
        else hipLaunchKernelGGL((k2l_window_lanes<K2L_MAX_COLS, STAGE>), dim3(persistent ? min(tiles, 256u * waves) : tiles), dim3(64), 0, stream_wide, d, n_small, n_wide);
        HIP_CHECK_LAUNCH();
    }
    if (n_hash) {   // 9..16 columns: the per-lane hash table (16 KB of LDS per wave, as the 7-8 column form)
        // These windows are a sparse subset (7.5 % at config C): the 64 windows of a tile come from many exons, their candidate ranges do
        // not overlap, and staging them through LDS took one pass - barrier, reductions, a round of loads - per WINDOW (1.36 ms for
        // 0.66 M windows). Every lane reads its own window's records straight from memory instead (two loads in flight ahead).
        const uint32_t tiles = (n_hash + 63) / 64, waves = min(32u, 163840u / 16384u);
        if (gather16) hipLaunchKernelGGL((k2l_window_lanes<K2L_HASH_COLS, 0, 16>), dim3(tiles), dim3(64), 0, stream_hash, d, d.n_lane_mid, n_hash);
        else hipLaunchKernelGGL((k2l_window_lanes<K2L_HASH_COLS, 0>), dim3(persistent ? min(tiles, 256u * waves) : tiles), dim3(64), 0, stream_hash, d, d.n_lane_mid, n_hash);
        HIP_CHECK_LAUNCH();
    }
}
void launch_k2_window_lanes(const DeviceBatch& d, hipStream_t stream_small, hipStream_t stream_wide, hipStream_t stream_hash) {
    if (!d.lane_on) return;
    static const int stage = [] { const char* e = std::getenv("MP_K2L_STAGE"); return e ? std::atoi(e) : 256; }();   // experiments
    switch (stage) {
        case 0: launch_k2l<0>(d, stream_small, stream_wide, stream_hash); break;
        case 384: launch_k2l<384>(d, stream_small, stream_wide, stream_hash); break;
        default: launch_k2l<256>(d, stream_small, stream_wide, stream_hash); break;
    }
}

#endif
template <int LIST>
void launch_k3_list(const DeviceBatch& d, uint64_t max_items, hipStream_t stream) {
    if (max_items == 0) return;
    // the grid covers the host's upper bound of the list lengths (surplus waves find nothing and leave; were the bound too low, the
    // waves walk the rest in turn - slower, still complete)
    // one grid row per allocator list; the lists fill evenly (allocator = workgroup index & 63 in the K2 kernels), a quarter more for the spread
    // one wave per workgroup for all lists. (List A in 256-thread workgroups - four waves sharing one 2 KB decimal-text table, 12
    // instead of 11 waves per CU - measured 1.55 against 1.46 ms: the four waves of a workgroup start together and stay in step, gather
    // phase on gather phase; the instantiations were dropped again.)
    static const int items = [] { const char* e = std::getenv("MP_K3_ITEMS"); return e && std::atoi(e) == 1 ? 1 : 2; }();   // list entries per lane (1: the round-2 form)
    const int T = K3_THREADS;
    // (two entries per lane pay where the list fills the chip several times over - 7 M entries of list A at config C: 1.45 -> 1.37 ms; a
    //  short list - list D, every list of a small batch - is a single round of waves, whose length two entries per lane would double:
    //  config B 0.41 -> 0.46 ms when lists C and D arrived with two)
    const int U = d.seq_cap <= 48 && max_items >= 640 * 1024 ? items : 1;
    const uint64_t per_wave = uint64_t(T) * uint64_t(U);
    const uint64_t per_list = max_items / NPART + max_items / (4 * NPART) + per_wave;
    dim3 grid(uint32_t(std::min<uint64_t>((per_list + per_wave - 1) / per_wave, 0x7FFFFFFFull)), NPART), block(T);
#define K3_LAUNCH(CAP, UU) hipLaunchKernelGGL((k3_window_seq<CAP, LIST, K3_THREADS, UU>), grid, block, 0, stream, d)
    switch (d.seq_cap) {
        case 32: if (U == 2) K3_LAUNCH(32, 2); else K3_LAUNCH(32, 1); break;
        case 48: if (U == 2) K3_LAUNCH(48, 2); else K3_LAUNCH(48, 1); break;
        case 112: K3_LAUNCH(112, 1); break;
        case 240: K3_LAUNCH(240, 1); break;
        default: throw_hip(hipErrorInvalidValue, __FILE__, __LINE__);
    }
#undef K3_LAUNCH
    HIP_CHECK_LAUNCH();
}
#if MP_KPART != -1   // the three lists' kernels are compiled in a part each
#if MP_KPART == 3
template void launch_k3_list<0>(const DeviceBatch&, uint64_t, hipStream_t);
#elif MP_KPART == 4
template void launch_k3_list<1>(const DeviceBatch&, uint64_t, hipStream_t);
extern template void launch_k3_list<0>(const DeviceBatch&, uint64_t, hipStream_t);
extern template void launch_k3_list<2>(const DeviceBatch&, uint64_t, hipStream_t);
extern template void launch_k3_list<3>(const DeviceBatch&, uint64_t, hipStream_t);
#elif MP_KPART == 5
template void launch_k3_list<2>(const DeviceBatch&, uint64_t, hipStream_t);
#elif MP_KPART == 6
template void launch_k3_list<3>(const DeviceBatch&, uint64_t, hipStream_t);
#endif
#endif
#if MP_IN_PART(4)
void launch_k3_window_seq(const DeviceBatch& d, uint64_t max_list_a, uint64_t max_list_b, uint64_t max_list_c, uint64_t max_list_d,
                          hipStream_t stream_a, hipStream_t stream_b, hipStream_t stream_c, hipStream_t stream_d) {
    if (d.normal) {   // `microphaser normal`: every group has a record (all of them are in list A); ids by k3b_haplotype_ids afterwards
        if (max_list_a == 0) return;
        const uint64_t per_list = max_list_a / NPART + max_list_a / (4 * NPART) + K3_THREADS;
        dim3 grid(uint32_t(std::min<uint64_t>((per_list + K3_THREADS - 1) / K3_THREADS, 0x7FFFFFFFull)), NPART), block(K3_THREADS);
        switch (d.seq_cap) {
            case 32: hipLaunchKernelGGL(k3_window_seq_normal<32>, grid, block, 0, stream_a, d); break;
            case 48: hipLaunchKernelGGL(k3_window_seq_normal<48>, grid, block, 0, stream_a, d); break;
            case 112: hipLaunchKernelGGL(k3_window_seq_normal<112>, grid, block, 0, stream_a, d); break;
            case 240: hipLaunchKernelGGL(k3_window_seq_normal<240>, grid, block, 0, stream_a, d); break;
            default: throw_hip(hipErrorInvalidValue, __FILE__, __LINE__);
        }
        HIP_CHECK_LAUNCH();
        return;
    }
    launch_k3_list<0>(d, max_list_a, stream_a);   // simple windows: sequences + records + ids
    launch_k3_list<1>(d, max_list_b, stream_b);   // simple windows: flags, carried records; independent of list A's groups, may run beside it
    launch_k3_list<2>(d, max_list_c, stream_c);   // simple windows that may hold a stop codon: the same + the codon scan
    launch_k3_list<3>(d, max_list_d, stream_d);   // windows that need the general walk: everything
}

void launch_k3b_haplotype_ids(const DeviceBatch& d, uint64_t max_recs, hipStream_t stream) {
    if (max_recs == 0) return;
    const uint64_t per_list = max_recs / NPART + max_recs / (4 * NPART) + K3B_THREADS;
    dim3 grid(uint32_t(std::min<uint64_t>((per_list + K3B_THREADS - 1) / K3B_THREADS, 0x7FFFFFFFull)), NPART), block(K3B_THREADS);
    switch (d.seq_cap) {
        case 32: hipLaunchKernelGGL(k3b_haplotype_ids<32>, grid, block, 0, stream, d); break;
        case 48: hipLaunchKernelGGL(k3b_haplotype_ids<48>, grid, block, 0, stream, d); break;
        case 112: hipLaunchKernelGGL(k3b_haplotype_ids<112>, grid, block, 0, stream, d); break;
        case 240: hipLaunchKernelGGL(k3b_haplotype_ids<240>, grid, block, 0, stream, d); break;
        default: throw_hip(hipErrorInvalidValue, __FILE__, __LINE__);
    }
    HIP_CHECK_LAUNCH();
}
#endif

}  // namespace mp
