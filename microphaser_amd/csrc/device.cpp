#include "device.hpp"
#include <chrono>
#include <cstdio>
#include <algorithm>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace mp {

[[noreturn]] void throw_hip(hipError_t e, const char* file, int line) {
    throw Error(std::string("HIP error: ") + hipGetErrorString(e) + " at " + file + ":" + std::to_string(line));
}
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw_hip(e_, __FILE__, __LINE__); } while (0)

DeviceContext::DeviceContext(int device) : device_(device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) throw Error("no HIP device available: the phasing hot path needs an MI355X (gfx950); there is no CPU fallback");
    if (device < 0 || device >= n) throw Error("invalid HIP device index " + std::to_string(device));
    HIP_OK(hipSetDevice(device));
    HIP_OK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    for (auto& ev : ev_) HIP_OK(hipEventCreate(&ev));
    for (auto& st : side_) HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (auto& ev : fork_) HIP_OK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&cleared_, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&k3_fork_, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&k3_join_, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&k3_join2_, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&k3_join3_, hipEventDisableTiming));
    for (auto& ev : join_) HIP_OK(hipEventCreate(&ev));
}

DeviceContext::~DeviceContext() {
    hipSetDevice(device_);
    free_batch();
    pool_trim(true);
    for (auto& p : pin_) if (p) (void)hipHostFree(p);
    for (auto& ev : pin_ev_) if (ev) (void)hipEventDestroy(ev);
    if (xfer_stream_) (void)hipStreamDestroy(xfer_stream_);
    for (auto& ev : ev_) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : fork_) if (ev) (void)hipEventDestroy(ev);
    if (cleared_) (void)hipEventDestroy(cleared_);
    if (k3_fork_) (void)hipEventDestroy(k3_fork_);
    if (k3_join_) (void)hipEventDestroy(k3_join_);
    if (k3_join2_) (void)hipEventDestroy(k3_join2_);
    if (k3_join3_) (void)hipEventDestroy(k3_join3_);
    for (auto& ev : join_) if (ev) (void)hipEventDestroy(ev);
    for (auto& st : side_) if (st) (void)hipStreamDestroy(st);
    if (stream_) hipStreamDestroy(stream_);
}

void* DeviceContext::dalloc(size_t bytes) {
    void* p = nullptr;
    if (const char* lim = std::getenv("MP_TEST_ALLOC_LIMIT"))   // tests: allocations above this many bytes fail like an exhausted HBM
        if (bytes > std::strtoull(lim, nullptr, 10)) throw_hip(hipErrorOutOfMemory, __FILE__, __LINE__);
    static const bool dbg = std::getenv("MP_DEBUG") != nullptr;
    static const bool pooled = std::getenv("MP_NO_POOL") == nullptr;
    const size_t want = std::max<size_t>(bytes, 256);
    if (pooled) {   // the smallest free block that holds the request (and is not absurdly larger), cleared like a fresh allocation
        PoolBlock* best = nullptr;
        for (PoolBlock& k : pool_)
            if (!k.in_use && k.cap >= want && k.cap <= 2 * want + (size_t(16) << 20) && (!best || k.cap < best->cap)) best = &k;
        if (best) {
            best->in_use = best->used_now = true;
            HIP_OK(hipMemsetAsync(best->p, 0, want, stream_));
            hbm_bytes_ += bytes;
            alloc_reused_++;
            return best->p;
        }
    }
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t me = hipMalloc(&p, want);
    if (me == hipErrorOutOfMemory && pooled) { (void)hipGetLastError(); pool_trim(true); me = hipMalloc(&p, want); }   // (give the idle blocks back first)
    HIP_OK(me);
    if (pooled) pool_.push_back(PoolBlock{p, want, true, true});
    if (dbg) {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        alloc_ms_ += ms; alloc_calls_++;
        if (ms > 5.0) std::fprintf(stderr, "[mp]     hipMalloc of %.1f MB took %.1f ms\n", double(bytes) / 1e6, ms);
    }
    hbm_bytes_ += bytes;
    return p;
}

template <class V>
typename V::value_type* DeviceContext::up(const V& v) {
    using T = typename V::value_type;
    T* p = static_cast<T*>(dalloc(v.size() * sizeof(T) + 256));  // +256: kernels stage pools with 16-byte loads at clamped-late addresses (K1: up to 128 bytes of payload / 32 variants past a start)
    allocs_.push_back(p);
    // (copied by upload_impl once every array has its device memory: one pipelined run through the pinned ring)
    if (!v.empty()) pending_up_.push_back(XferSeg{const_cast<char*>(reinterpret_cast<const char*>(v.data())), reinterpret_cast<char*>(p), v.size() * sizeof(T)});
    return p;
}

// Parallel memcpy of one staging slot's worth of bytes (the destination's pages may be touched here for the first time).
static void copy_parallel(char* dst, const char* src, size_t n, size_t nthreads) {
    const size_t piece = size_t(4) << 20;
    const size_t parts = std::min<size_t>(nthreads, (n + piece - 1) / piece);
    if (parts <= 1) { std::memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    for (size_t t = 1; t < parts; t++) {
        const size_t lo = n * t / parts, hi = n * (t + 1) / parts;
        th.emplace_back([=] { std::memcpy(dst + lo, src + lo, hi - lo); });
    }
    std::memcpy(dst, src, n / parts);
    for (auto& x : th) x.join();
}

void DeviceContext::xfer(const std::vector<XferSeg>& segs, bool to_device) {
    size_t total = 0;
    for (const XferSeg& s : segs) total += s.bytes;
    if (!total) return;
    // Uploads go straight from the pageable arrays unless MP_PINNED_H2D is set: measured at config C (8.1 GB up, 32 host threads), the
    // runtime's own pageable path already moves 32 GB/s and filling the ring from the same pageable pages with 8 threads is no faster
    // (284 vs 250 ms); downloads through the ring take 90 instead of 174 ms (2.3 GB; profiles/r05g_e2e_pinned_vs_pageable.log) - their
    // destination pages are fresh and are then first touched by eight threads instead of the runtime's one.
    const bool plain = std::getenv("MP_NO_PINNED") || total < (size_t(1) << 20) || (to_device && !std::getenv("MP_PINNED_H2D"));
    if (plain) {   // small transfers, uploads, and the A/B switch of the measurements: plain copies
        for (const XferSeg& s : segs)
            if (s.bytes) HIP_OK(hipMemcpyAsync(to_device ? (void*)s.dev : (void*)s.host, to_device ? (const void*)s.host : (const void*)s.dev, s.bytes,
                                               to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, stream_));
        HIP_OK(hipStreamSynchronize(stream_));
        return;
    }
    if (!xfer_stream_) HIP_OK(hipStreamCreateWithFlags(&xfer_stream_, hipStreamNonBlocking));
    for (size_t k = 0; k < XFER_SLOTS; k++) {
        if (!pin_[k]) HIP_OK(hipHostMalloc(&pin_[k], XFER_SLOT_BYTES, hipHostMallocDefault));
        if (!pin_ev_[k]) HIP_OK(hipEventCreateWithFlags(&pin_ev_[k], hipEventDisableTiming));
    }
    HIP_OK(hipStreamSynchronize(stream_));   // the kernels that produced / will read these buffers run on stream_
    // chunks: every segment cut into slot-sized pieces (a piece never spans two segments: both sides stay contiguous)
    struct Chunk { size_t seg, off, len; };
    std::vector<Chunk> chunks;
    for (size_t i = 0; i < segs.size(); i++)
        for (size_t off = 0; off < segs[i].bytes; off += XFER_SLOT_BYTES) chunks.push_back(Chunk{i, off, std::min(XFER_SLOT_BYTES, segs[i].bytes - off)});
    const size_t nthreads = std::max<size_t>(1, std::min<size_t>(8, host_threads()));
    const size_t n = chunks.size();
    if (to_device) {
        // fill slot (host threads) -> DMA; a slot is refilled only after its previous DMA has completed
        for (size_t i = 0; i < n; i++) {
            const size_t k = i % XFER_SLOTS;
            if (i >= XFER_SLOTS) HIP_OK(hipEventSynchronize(pin_ev_[k]));
            const Chunk& c = chunks[i];
            copy_parallel(static_cast<char*>(pin_[k]), segs[c.seg].host + c.off, c.len, nthreads);
            HIP_OK(hipMemcpyAsync(segs[c.seg].dev + c.off, pin_[k], c.len, hipMemcpyHostToDevice, xfer_stream_));
            HIP_OK(hipEventRecord(pin_ev_[k], xfer_stream_));
        }
        HIP_OK(hipStreamSynchronize(xfer_stream_));
    } else {
        // DMA into slots up to XFER_SLOTS ahead; drain a slot (host threads) as soon as its DMA has completed. A slot takes a RUN of
        // pieces whose host destinations follow one another (the used prefixes of the 64 allocators' sub-ranges of one array: separate
        // ranges on the device, one array on the host): one DMA per piece, ONE drain of the whole slot by all the threads - a slot per
        // 12 MB segment had left the drain to three threads at a time.
        struct Run { size_t first, count, len; char* host; };   // pieces chunks[first, first + count)
        std::vector<Run> runs;
        for (size_t i = 0; i < n; i++) {
            const Chunk& c = chunks[i];
            char* const h = segs[c.seg].host + c.off;
            if (!runs.empty() && runs.back().host + runs.back().len == h && runs.back().len + c.len <= XFER_SLOT_BYTES) { runs.back().count++; runs.back().len += c.len; }
            else runs.push_back(Run{i, 1, c.len, h});
        }
        const size_t nr = runs.size();
        size_t issued = 0;
        auto issue = [&] {
            const Run& r = runs[issued];
            const size_t k = issued % XFER_SLOTS;
            size_t at = 0;
            for (size_t j = r.first; j < r.first + r.count; j++) {
                const Chunk& c = chunks[j];
                HIP_OK(hipMemcpyAsync(static_cast<char*>(pin_[k]) + at, segs[c.seg].dev + c.off, c.len, hipMemcpyDeviceToHost, xfer_stream_));
                at += c.len;
            }
            HIP_OK(hipEventRecord(pin_ev_[k], xfer_stream_));
            issued++;
        };
        while (issued < nr && issued < XFER_SLOTS) issue();
        for (size_t i = 0; i < nr; i++) {
            const size_t k = i % XFER_SLOTS;
            HIP_OK(hipEventSynchronize(pin_ev_[k]));
            copy_parallel(runs[i].host, static_cast<const char*>(pin_[k]), runs[i].len, nthreads);
            if (issued < nr) issue();   // (slot k is free again: run i + XFER_SLOTS goes there)
        }
    }
}

void DeviceContext::dfree(void* p) {
    for (PoolBlock& k : pool_)
        if (k.p == p) { k.in_use = false; return; }
    (void)hipFree(p);   // (not pooled: MP_NO_POOL)
}
void DeviceContext::pool_trim(bool all) {   // frees the idle blocks (all of them, or those the current batch did not take)
    std::vector<PoolBlock> keep;
    for (PoolBlock& k : pool_) {
        if (!k.in_use && (all || !k.used_now)) (void)hipFree(k.p);
        else keep.push_back(k);
    }
    pool_.swap(keep);
}
void DeviceContext::free_outputs() {
    for (void* p : out_allocs_) dfree(p);
    out_allocs_.clear();
    hbm_bytes_ -= out_bytes_;
    out_bytes_ = 0;
    d_.groups = nullptr; d_.k3_items = nullptr; d_.gsum = nullptr; d_.recs = nullptr; d_.want_recs = nullptr;
}

void DeviceContext::free_batch() {
    pending_up_.clear();
    free_outputs();
    for (void* p : allocs_) dfree(p);
    allocs_.clear();
    hbm_bytes_ = 0;
    out_bytes_ = 0;
    std::memset(&d_, 0, sizeof d_);
}

void DeviceContext::upload(const Batch& b) {
    try { upload_impl(b); }
    catch (...) { free_batch(); throw; }   // never leave a half-uploaded batch behind (dangling or null pointers in d_)
}

void DeviceContext::upload_impl(const Batch& b) {
    HIP_OK(hipSetDevice(device_));
    const bool dbg = std::getenv("MP_DEBUG") != nullptr;
    const auto t_up0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    free_batch();
    for (PoolBlock& k : pool_) k.used_now = false;
    alloc_ms_ = 0; alloc_calls_ = alloc_reused_ = 0;
    const double ms_free = ms_since(t_up0);
    d_.g_read_off = up(b.g_read_off);
    d_.g_var_off = up(b.g_var_off);
    d_.g_start = up(b.g_start);
    d_.g_ref_off = up(b.g_ref_off);
    d_.n_genes = b.g_read_off.size() > 1 ? uint32_t(b.g_read_off.size() - 1) : 0u;
    d_.r_var = static_cast<uint2*>(dalloc((b.r_pos.size() + 1) * sizeof(uint2))); allocs_.push_back(d_.r_var);   // filled by k0_read_variants below
    d_.r_pos = up(b.r_pos);
    d_.r_end = up(b.r_end);
    d_.r_lseq = up(b.r_lseq);
    d_.r_ncig = up(b.r_ncig);
    d_.r_dup = up(b.r_dup);
    d_.r_varlo = up(b.r_varlo);
    d_.r_cigoff = up(b.r_cigoff);
    d_.r_seqoff = up(b.r_seqoff);
    d_.cigar_pool = up(b.cigar_pool);
    d_.seq_pool = up(b.seq_pool);
    d_.v_pos = up(b.v_pos);
    d_.v_info = up(b.v_info);
    d_.v_len = up(b.v_len);
    d_.v_insoff = up(b.v_insoff);
    d_.v_rev2fwd = up(b.v_rev2fwd);
    d_.ins_pool = up(b.ins_pool);
    d_.ref_pool = up(b.ref_pool);
    d_.tx = up(b.tx);
    d_.steps = up(b.steps);
    d_.step_aux = up(b.step_aux);
    d_.wins = up(b.wins);
    d_.win_cols = up(b.win_cols);
    d_.str_pool = up(b.str_pool);
    d_.exons_w = up(b.exons_w);
    d_.wchunks = up(b.wchunks);
    d_.wchunks_m = up(b.wchunks_m);
    d_.n_wchunks_m = uint32_t(b.wchunks_m.size());
    d_.wchunks_d = up(b.wchunks_d);
    d_.n_wchunks_d = uint32_t(b.wchunks_d.size());
    d_.rows_per_lane_w = b.rows_per_lane_w;
    d_.achunks = up(b.achunks);
    d_.n_achunks = uint32_t(b.achunks.size());
    d_.k2a_flat = !std::getenv("MP_K2A_CHUNKS") && !b.exons_w.empty() && b.n_adm && b.n_adm < 0xFFFFFF00ull ? 1u : 0u;
    d_.achunk_exons = nullptr;
    if (!d_.k2a_flat) {   // the form with a wave per work item: every work item gets its exon's record beside it (one level of scalar loads
        // instead of work item -> exon)
        achunk_exons_.resize(b.achunks.size());
        for (size_t k = 0; k < b.achunks.size(); k++) achunk_exons_[k] = b.exons_w[b.achunks[k].exon];
        d_.achunk_exons = up(achunk_exons_);
    }
    d_.step_ncols = up(b.step_ncols);
    d_.step_rlo = up(b.step_rlo);
    d_.step_rn = up(b.step_rn);
    d_.v_sombits = up(b.v_sombits);
    d_.n_exons_w = uint32_t(b.exons_w.size());
    d_.n_wchunks = uint32_t(b.wchunks.size());
    d_.n_adm = b.n_adm;
    d_.adm = static_cast<AdmEntry*>(dalloc(size_t(b.n_adm + 1) * sizeof(AdmEntry))); allocs_.push_back(d_.adm);
    d_.lane_on = b.lane_on ? 1u : 0u;
    d_.lane_hash = b.lane_hash ? 1u : 0u;
    d_.winw = up(b.winw);
    d_.lane_win = up(b.lane_win);
    d_.win_trivial = up(b.win_trivial);
    d_.win_simple = up(b.win_simple);
    d_.win_walk = up(b.win_walk);
    d_.n_lane_small = b.n_lane_small;
    d_.n_lane_mid = b.n_lane_mid;
    d_.n_lane_all = uint32_t(b.winw.size());
    if (b.lane_on) {
        d_.rr_a = static_cast<RowRecA*>(dalloc(size_t(b.n_adm + 1) * sizeof(RowRecA))); allocs_.push_back(d_.rr_a);
        d_.rr_sup = static_cast<uint64_t*>(dalloc(size_t(b.n_adm + 1) * 8)); allocs_.push_back(d_.rr_sup);
    }
    d_.segs = up(b.segs);
    d_.seg_order = up(b.seg_order);
    d_.n_segs = uint32_t(b.segs.size());
    d_.n_reads = uint32_t(b.r_pos.size());
    d_.n_tx = uint32_t(b.tx.size());
    d_.n_wins = uint32_t(b.wins.size());
    d_.mask_words = b.mask_words;
    d_.normal = b.normal ? 1u : 0u;
    d_.normal_large = 0;
    d_.timing_skip_ids = std::getenv("MP_TIMING_SKIP_IDS") ? 1u : 0u;
    d_.seq_cap = b.seq_cap;
    d_.rec_stride = hap_rec_stride(b.seq_cap);
    // K1 outputs
    d_.r_ncov = static_cast<uint32_t*>(dalloc(size_t(d_.n_reads) * 4)); allocs_.push_back(d_.r_ncov);
    d_.r_sup = static_cast<uint64_t*>(dalloc(size_t(d_.n_reads) * 8 * b.mask_words)); allocs_.push_back(d_.r_sup);
    d_.r_lq = static_cast<uint64_t*>(dalloc(size_t(d_.n_reads) * 8 * b.mask_words)); allocs_.push_back(d_.r_lq);
    d_.win_dyn = static_cast<WinDyn*>(dalloc(size_t(d_.n_wins) * sizeof(WinDyn))); allocs_.push_back(d_.win_dyn);
    d_.cursors = static_cast<unsigned long long*>(dalloc((NPART * 32 + 16) * 8)); allocs_.push_back(d_.cursors);
    d_.err = reinterpret_cast<uint32_t*>(d_.cursors + NPART * 32);   // the error word sits behind the cursors: one memset, one copy back per pass
    d_.tx_first_stop = static_cast<uint32_t*>(dalloc(size_t(d_.n_tx) * 4)); allocs_.push_back(d_.tx_first_stop);
    max_rows_bound_ = b.max_rows_bound;
    last_slots_ = last_recs_ = last_want_ = last_k3a_ = last_k3b_ = last_k3c_ = last_k3d_ = 0;
    rpl_ = 1;
    while (rpl_ < 16 && uint32_t(64 * rpl_) < max_rows_bound_) rpl_ *= 2;  // overflow -> run() retries with more
    // first guess: 6 distinct haplotypes per window + one partly used chunk per wave, split over the NPART allocators
    const uint64_t waves = b.wchunks.size() + b.segs.size() + b.wchunks_m.size() * (1 + b.rows_per_lane_w) + b.wchunks_d.size() * 16;
    uint64_t g_need = uint64_t(d_.n_wins) * 6 + waves * 256 + 4096;
    uint64_t r_need = g_need / 3 + waves * 128 + 4096;
    if (b.normal) { g_need += waves * 1024; r_need = g_need; }   // every haplotype of every window has a record in this mode
    glog_ = 12; rlog_ = 12;
    while ((uint64_t(NPART) << glog_) < g_need + g_need / 4) glog_++;
    while ((uint64_t(NPART) << rlog_) < r_need + r_need / 4) rlog_++;
    if (std::getenv("MP_TEST_SMALL_CAPS")) glog_ = rlog_ = 8;   // tests: start far too small, so that run() has to grow the buffers
    d_.win_blobs = nullptr;
    if (!b.normal && d_.n_wins) { d_.win_blobs = static_cast<WinBlob*>(dalloc(size_t(d_.n_wins) * sizeof(WinBlob))); allocs_.push_back(d_.win_blobs); }
    d_.exons_a = nullptr; d_.adm_map = nullptr;
    if (d_.k2a_flat) {
        d_.exons_a = static_cast<ExonA*>(dalloc(size_t(d_.n_exons_w) * sizeof(ExonA))); allocs_.push_back(d_.exons_a);
        d_.adm_map = static_cast<AdmMap*>(dalloc(size_t(b.n_adm) * sizeof(AdmMap))); allocs_.push_back(d_.adm_map);
    }
    const double ms_inputs = ms_since(t_up0);
    alloc_outputs();
    const double ms_alloc = ms_since(t_up0);
    uint64_t up_bytes = 0;
    for (const XferSeg& sgm : pending_up_) up_bytes += sgm.bytes;
    if (dbg) {   // the largest uploads
        std::vector<uint64_t> sz;
        for (const XferSeg& sgm : pending_up_) sz.push_back(sgm.bytes);
        std::sort(sz.rbegin(), sz.rend());
        std::fprintf(stderr, "[mp]   upload: %zu arrays, %.2f GB; the largest (MB):", pending_up_.size(), double(up_bytes) / 1e9);
        for (size_t k = 0; k < sz.size() && k < 14; k++) std::fprintf(stderr, " %.0f", double(sz[k]) / 1e6);
        std::fprintf(stderr, "\n[mp]   upload sizes (MB): reads fields %.0f, seq pool %.0f, cigar %.0f, ref %.0f, steps %.0f (+ side arrays %.0f), wins %.0f, win_cols %.0f, winw %.0f\n",
                     double(b.r_pos.size()) * (4 * 7 + 8 * 3) / 1e6, double(b.seq_pool.size()) / 1e6, double(b.cigar_pool.size()) * 4 / 1e6, double(b.ref_pool.size()) / 1e6,
                     double(b.steps.size()) * sizeof(Step) / 1e6, double(b.steps.size()) * 13 / 1e6 + double(b.step_aux.size()) * sizeof(b.step_aux[0]) / 1e6,
                     double(b.wins.size()) * sizeof(WinStatic) / 1e6, double(b.win_cols.size()) * sizeof(WinCol) / 1e6, double(b.winw.size()) * (sizeof(WinW) + 4) / 1e6);
    }
    xfer(pending_up_, true);
    pending_up_.clear();
    // (One allocation for everything upload() places was tried: the ~60 hipMalloc calls then cost 24 instead of 170 ms - and the result
    //  arenas and the copies behind them 60 + 75 ms more: what costs is making ~14 GB of fresh device memory usable, not the calls.)
    if (dbg) std::fprintf(stderr, "[mp]   upload: %u hipMalloc calls took %.1f ms in all; %u blocks came from the context's pool\n", alloc_calls_, alloc_ms_, alloc_reused_);
    if (dbg)
        std::fprintf(stderr, "[mp]   upload: release of the previous batch %.1f ms, per-read table + input allocations %.1f ms, result arenas %.1f ms, copies %.1f ms\n",
                     ms_free, ms_inputs - ms_free, ms_alloc - ms_inputs, ms_since(t_up0) - ms_alloc);
    achunk_exons_ = PodVec<ExonW>();
    launch_k0_read_variants(d_, stream_);
    if (d_.k2a_flat) launch_k0_pack_admission(d_, stream_);
    if (d_.win_blobs) launch_k0_pack_windows(d_, stream_);   // (once per batch: K3's per-window records, plan.hpp WinBlob)
    HIP_OK(hipStreamSynchronize(stream_));
    pool_trim(false);   // idle blocks this batch had no use for go back to the device
}

void DeviceContext::alloc_outputs() {
    if ((uint64_t(NPART) << glog_) > 0x7FFFFFFFull || (uint64_t(NPART) << rlog_) > 0x7FFFFFFFull)
        throw Error("result buffers exceed 2^31 slots: split the batch by genes");   // (checked before anything is released)
    free_outputs();
    group_cap_ = uint64_t(NPART) << glog_;
    rec_cap_ = uint64_t(NPART) << rlog_;
    auto oalloc = [&](size_t bytes) { void* p = dalloc(bytes); out_allocs_.push_back(p); out_bytes_ += bytes; return p; };
    d_.groups = static_cast<Group*>(oalloc(group_cap_ * sizeof(Group)));
    d_.k3_items = static_cast<uint4*>(oalloc(2 * group_cap_ * sizeof(uint4)));   // lists A / B in the first half, list C in the second
    d_.gsum = static_cast<GroupSum*>(oalloc(group_cap_ * sizeof(GroupSum)));
    d_.recs = static_cast<uint8_t*>(oalloc(rec_cap_ * d_.rec_stride));
    d_.want_recs = static_cast<uint32_t*>(oalloc(rec_cap_ * 4));
    d_.group_part_log2 = glog_;
    d_.rec_part_log2 = rlog_;
    d_.group_cap = group_cap_;
    d_.rec_cap = rec_cap_;
}

void DeviceContext::run(RunTiming& t) {
    HIP_OK(hipSetDevice(device_));
    t = RunTiming();
    // One pass = one stream of launches with a single host synchronisation at its end: the counts the later kernels need (used group
    // slots, records that want an id) stay on the device (the allocators' cursors, read by K3 / K3b themselves). Only then are the cursors and the error
    // word read; an overflow grows the buffers (or the rows per lane) and runs the pass again.
    for (int attempt = 0; attempt < 8; attempt++) {
        t.attempts = uint32_t(attempt + 1);
        t.rows_per_lane = rpl_;
        HIP_OK(hipMemsetAsync(d_.cursors, 0, (NPART * 32 + 16) * 8, stream_));   // (incl. the error word)
        // the per-window / per-transcript outputs are cleared beside K1, which does not touch them (141 MB at config C)
        HIP_OK(hipMemsetAsync(d_.win_dyn, 0, size_t(d_.n_wins) * sizeof(WinDyn), side_[0]));
        HIP_OK(hipMemsetAsync(d_.tx_first_stop, 0xFF, size_t(d_.n_tx) * 4, side_[0]));
        HIP_OK(hipEventRecord(cleared_, side_[0]));
        HIP_OK(hipEventRecord(ev_[0], stream_));
        launch_k1_pileup_bits(d_, stream_);
        HIP_OK(hipEventRecord(ev_[1], stream_));
        HIP_OK(hipStreamWaitEvent(stream_, cleared_, 0));   // everything after K1 is ordered behind the clears (the side streams fork from here)
        // The launches of the window phase work on disjoint windows and are each bound by latency at modest occupancy, so they run side
        // by side: the sequential replay (needs K1 only) beside K2a; after K2a the two lane-per-window launches and the wave-per-window
        // kernels on three streams. Everything joins before K3. (The output allocators are shared: atomics.)
        HIP_OK(hipEventRecord(fork_[0], stream_));
        HIP_OK(hipStreamWaitEvent(side_[2], fork_[0], 0));
        launch_k2_window_replay(d_, rpl_, side_[2]);    // sequential replay of the segments that need it
        HIP_OK(hipEventRecord(join_[2], side_[2]));
        HIP_OK(hipEventRecord(ev_[5], stream_));
        launch_k2_admission(d_, stream_);                // K2a + K2l + K2w: everything else, window-parallel
        HIP_OK(hipEventRecord(ev_[6], stream_));
        HIP_OK(hipEventRecord(fork_[1], stream_));
        HIP_OK(hipStreamWaitEvent(side_[0], fork_[1], 0));
        HIP_OK(hipStreamWaitEvent(side_[1], fork_[1], 0));
        HIP_OK(hipStreamWaitEvent(side_[3], fork_[1], 0));
        launch_k2_window_lanes(d_, stream_, side_[0], side_[3]);   // lane per window: <= 6 columns here, 7-8 and 9-16 columns beside it
        HIP_OK(hipEventRecord(ev_[7], stream_));
        HIP_OK(hipEventRecord(join_[0], side_[0]));
        HIP_OK(hipEventRecord(join_[3], side_[3]));
        launch_k2_window_rows(d_, side_[1]);             // wave per window: the rest, beside them
        HIP_OK(hipEventRecord(join_[1], side_[1]));
        for (auto& ev : join_) HIP_OK(hipStreamWaitEvent(stream_, ev, 0));
        HIP_OK(hipEventRecord(ev_[2], stream_));
        // K3 walks the lists the K2 kernels made: list A (sequences, records and - somatic - their ids in one kernel) and, beside it on a
        // side stream, list B (flags only). `normal`: K3n over list A, then K3b over the records K3n wants ids for.
        // (the grids cover an upper bound of the list lengths: the previous pass's counts of this batch plus a margin, else an estimate)
        const uint64_t a_bound = last_slots_ ? last_k3a_ + last_k3a_ / 16 + 4096 : std::min<uint64_t>(group_cap_, uint64_t(d_.n_wins) * (d_.normal ? 8 : 2) + 65536);
        const uint64_t b_bound = last_slots_ ? last_k3b_ + last_k3b_ / 16 + 4096 : std::min<uint64_t>(group_cap_, uint64_t(d_.n_wins) * 2 + 65536);
        const uint64_t c_bound = last_slots_ ? (last_k3c_ ? last_k3c_ + last_k3c_ / 16 + 4096 : 0) : std::min<uint64_t>(group_cap_, uint64_t(d_.n_wins) / 4 + 65536);
        const uint64_t d_bound = last_slots_ ? (last_k3d_ ? last_k3d_ + last_k3d_ / 16 + 4096 : 0) : std::min<uint64_t>(group_cap_, uint64_t(d_.n_wins) / 8 + 65536);
        const uint64_t want_bound = last_slots_ ? last_want_ + last_want_ / 16 + 4096 : std::min<uint64_t>(rec_cap_, uint64_t(d_.n_wins) * 2 + 65536);
        if (!d_.normal) {
            HIP_OK(hipEventRecord(k3_fork_, stream_));
            HIP_OK(hipStreamWaitEvent(side_[0], k3_fork_, 0));
            HIP_OK(hipStreamWaitEvent(side_[1], k3_fork_, 0));
            HIP_OK(hipStreamWaitEvent(side_[3], k3_fork_, 0));
        }
        launch_k3_window_seq(d_, a_bound, b_bound, c_bound, d_bound, stream_, side_[0], side_[1], side_[3]);
        if (!d_.normal) {
            HIP_OK(hipEventRecord(k3_join3_, side_[3]));
            HIP_OK(hipStreamWaitEvent(stream_, k3_join3_, 0));
            HIP_OK(hipEventRecord(k3_join_, side_[0]));
            HIP_OK(hipStreamWaitEvent(stream_, k3_join_, 0));
            HIP_OK(hipEventRecord(k3_join2_, side_[1]));
            HIP_OK(hipStreamWaitEvent(stream_, k3_join2_, 0));
        }
        HIP_OK(hipEventRecord(ev_[3], stream_));
        if (d_.normal) launch_k3b_haplotype_ids(d_, want_bound, stream_);
        HIP_OK(hipEventRecord(ev_[4], stream_));
        std::vector<unsigned long long> cur(NPART * 32 + 1);
        HIP_OK(hipMemcpyAsync(cur.data(), d_.cursors, cur.size() * 8, hipMemcpyDeviceToHost, stream_));
        HIP_OK(hipStreamSynchronize(stream_));
        const uint32_t err = uint32_t(cur[NPART * 32]);
        if (err & WD_ROW_OVERFLOW) {
            if (rpl_ >= 16) throw Error("more than 1024 simultaneously live reads in one window (depth too high for this build)");
            rpl_ *= 2;
            continue;
        }
        if ((err & (WD_EPOCH_OVERFLOW | WD_HAP_OVERFLOW)) && !d_.normal_large) { d_.normal_large = 1; continue; }   // the large tables, again
        if (err & WD_EPOCH_OVERFLOW) throw Error("normal mode: more than 512 live column epochs in one transcript (variant density too high for this build)");
        if (err & WD_HAP_OVERFLOW) throw Error("normal mode: more than 2048 distinct haplotypes in one window");
        uint64_t max_g = 0, max_r = 0, max_w = 0;
        for (uint32_t p = 0; p < NPART; p++) {
            max_g = std::max<uint64_t>(max_g, cur[p * 32]); max_r = std::max<uint64_t>(max_r, cur[p * 32 + 16]);
            if (d_.normal) max_w = std::max<uint64_t>(max_w, cur[p * 32 + 24]);   // (`normal`: the length of K3n's wanted list; somatic: a count of ids, not a list)
        }
        if ((err & (WD_GROUP_OVERFLOW | WD_REC_OVERFLOW)) || max_g > (1ull << glog_) || max_r > (1ull << rlog_) || max_w > (1ull << rlog_)) {
            const uint32_t g0 = glog_, r0 = rlog_;
            while ((1ull << glog_) < max_g + max_g / 8) glog_++;
            while ((1ull << rlog_) < std::max(max_r, max_w) + std::max(max_r, max_w) / 8) rlog_++;
            if ((err & WD_GROUP_OVERFLOW) && glog_ == g0) glog_++;   // flagged without a cursor past the end: grow anyway
            if ((err & WD_REC_OVERFLOW) && rlog_ == r0) rlog_++;     // (also: a wanted list outgrew its sub-range, or K3 found a record slot missing)
            if (glog_ == g0 && rlog_ == r0) rlog_++;
            alloc_outputs();
            continue;
        }
        if (err) throw Error("device kernels reported an internal inconsistency (error word " + std::to_string(err) + ")");
        uint64_t slots = 0, rec_slots = 0, n_want = 0, n_k3a = 0, n_k3b = 0, n_k3c = 0, n_k3d = 0;
        for (uint32_t p = 0; p < NPART; p++) {
            used_g_[p] = cur[p * 32];
            used_r_[p] = cur[p * 32 + 16];
            slots += used_g_[p];
            rec_slots += used_r_[p];
            n_want += cur[p * 32 + 24];
            n_k3a += cur[p * 32 + 8];
            n_k3b += cur[p * 32 + 12];
            n_k3c += cur[p * 32 + 20];
            n_k3d += cur[p * 32 + 28];
        }
        HIP_OK(hipEventElapsedTime(&t.k3b_ms, ev_[3], ev_[4]));
        HIP_OK(hipEventElapsedTime(&t.k1_ms, ev_[0], ev_[1]));
        HIP_OK(hipEventElapsedTime(&t.k2_ms, ev_[1], ev_[2]));
        // (overlapping intervals: k2seq beside k2a; k2l = the longer of its two launches, k2w beside them, both from the end of K2a)
        HIP_OK(hipEventElapsedTime(&t.k2seq_ms, ev_[1], join_[2]));
        HIP_OK(hipEventElapsedTime(&t.k2a_ms, ev_[1], ev_[6]));
        float l6 = 0, l8 = 0, l16 = 0;
        HIP_OK(hipEventElapsedTime(&l6, ev_[6], ev_[7]));
        HIP_OK(hipEventElapsedTime(&l8, ev_[6], join_[0]));
        HIP_OK(hipEventElapsedTime(&l16, ev_[6], join_[3]));
        t.k2l_ms = std::max(std::max(l6, l8), l16);
        HIP_OK(hipEventElapsedTime(&t.k2w_ms, ev_[6], join_[1]));
        HIP_OK(hipEventElapsedTime(&t.k2win_ms, ev_[6], ev_[2]));
        HIP_OK(hipEventElapsedTime(&t.k3_ms, ev_[2], ev_[3]));
        HIP_OK(hipEventElapsedTime(&t.total_ms, ev_[0], ev_[4]));
        last_slots_ = slots;
        last_recs_ = rec_slots;
        last_want_ = n_want;
        last_k3a_ = n_k3a;
        last_k3b_ = n_k3b;
        last_k3c_ = n_k3c;
        last_k3d_ = n_k3d;
        t.n_group_slots = slots;    // group slots K3 walked (incl. the unused tail of each wave's last chunk) / records K3b hashed
        t.n_recs = n_want;
        t.n_groups = slots;
        t.n_k3 = n_k3a + n_k3b + n_k3c + n_k3d;
        t.n_k3c = n_k3c;
        t.n_k3d = n_k3d;
        t.n_k3a = n_k3a;
        t.n_rec_slots = rec_slots;
        return;
    }
    throw Error("device result buffers kept overflowing");
}

void DeviceContext::download(HostResults& r) {
    HIP_OK(hipSetDevice(device_));
    r.n_group_slots = last_slots_;
    r.n_recs = last_recs_;
    r.win_dyn.resize(d_.n_wins);
    r.groups.resize(last_slots_);
    r.gsum.resize(last_slots_);
    r.seq_cap = d_.seq_cap;
    r.rec_stride = d_.rec_stride;
    r.recs.resize(last_recs_ * d_.rec_stride);
    advise_huge(r.groups.data(), r.groups.size() * sizeof(Group));
    advise_huge(r.gsum.data(), r.gsum.size() * sizeof(GroupSum));
    advise_huge(r.recs.data(), r.recs.size());
    advise_huge(r.win_dyn.data(), r.win_dyn.size() * sizeof(WinDyn));
    r.group_part_log2 = glog_;
    r.rec_part_log2 = rlog_;
    std::vector<XferSeg> segs;
    auto seg = [&](void* host, const void* dev, size_t bytes) { if (bytes) segs.push_back(XferSeg{static_cast<char*>(host), const_cast<char*>(static_cast<const char*>(dev)), bytes}); };
    seg(r.win_dyn.data(), d_.win_dyn, size_t(d_.n_wins) * sizeof(WinDyn));
    uint64_t go = 0, ro = 0;
    for (uint32_t p = 0; p < NPART; p++) {   // the used prefix of every allocator's sub-range, back to back
        r.group_prefix[p] = go;
        r.rec_prefix[p] = ro;
        go += used_g_[p];
        ro += used_r_[p];
    }
    r.group_prefix[NPART] = go;
    r.rec_prefix[NPART] = ro;
    // array by array: consecutive segments then have consecutive host destinations, and xfer() packs them into full staging slots
    for (uint32_t p = 0; p < NPART; p++) seg(r.groups.data() + r.group_prefix[p], d_.groups + (uint64_t(p) << glog_), used_g_[p] * sizeof(Group));
    for (uint32_t p = 0; p < NPART; p++) seg(r.gsum.data() + r.group_prefix[p], d_.gsum + (uint64_t(p) << glog_), used_g_[p] * sizeof(GroupSum));
    for (uint32_t p = 0; p < NPART; p++) seg(r.recs.data() + r.rec_prefix[p] * d_.rec_stride, d_.recs + (uint64_t(p) << rlog_) * d_.rec_stride, used_r_[p] * d_.rec_stride);
    xfer(segs, false);   // pipelined through the pinned ring: DMA of one slot beside the host threads draining the others
}

}  // namespace mp
