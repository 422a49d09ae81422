#include "io.hpp"
#include <functional>
#include <deque>
#include <condition_variable>
#include <cerrno>
#include <unistd.h>
#include <sys/stat.h>
#include <fcntl.h>

#include <zlib.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

#include <sys/mman.h>

namespace mp {

void advise_huge(const void* p, size_t bytes) {
#ifdef MADV_HUGEPAGE
    if (bytes < (size_t(4) << 20)) return;
    const uintptr_t two_mb = uintptr_t(1) << 21;
    const uintptr_t a = (reinterpret_cast<uintptr_t>(p) + two_mb - 1) & ~(two_mb - 1), e = (reinterpret_cast<uintptr_t>(p) + bytes) & ~(two_mb - 1);
    if (e > a) (void)madvise(reinterpret_cast<void*>(a), e - a, MADV_HUGEPAGE);
#else
    (void)p; (void)bytes;
#endif
}

// The reaper (model.hpp release_later): one owned worker thread, started by the first job, joined when the last context dies or the
// library is torn down.
namespace {
struct Reaper {
    std::mutex m;
    std::condition_variable work_cv, idle_cv;
    std::deque<std::function<void()>> q;
    std::thread th;
    bool running = false, stop = false, busy = false;
    int users = 0;
    void loop() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            work_cv.wait(lk, [&] { return stop || !q.empty(); });
            if (q.empty()) return;   // (stop is only honoured once the queue is empty)
            std::function<void()> job = std::move(q.front());
            q.pop_front();
            busy = true;
            lk.unlock();
            job();
            job = nullptr;           // the held object dies here, outside the lock
            lk.lock();
            busy = false;
            if (q.empty()) idle_cv.notify_all();
        }
    }
    void post(std::function<void()> job) {
        std::lock_guard<std::mutex> lk(m);
        q.push_back(std::move(job));
        if (!running) { stop = false; th = std::thread([this] { loop(); }); running = true; }
        work_cv.notify_one();
    }
    void drain() {
        std::unique_lock<std::mutex> lk(m);
        idle_cv.wait(lk, [&] { return q.empty() && !busy; });
    }
    void shutdown() {
        std::thread t;
        {
            std::lock_guard<std::mutex> lk(m);
            if (!running) return;
            stop = true;
            running = false;
            t = std::move(th);
            work_cv.notify_all();
        }
        t.join();   // the loop leaves only with an empty queue
    }
    ~Reaper() { shutdown(); }
};
Reaper& reaper() { static Reaper r; return r; }
}  // namespace
void reaper_post(std::function<void()> job) { reaper().post(std::move(job)); }
void reaper_retain() { Reaper& r = reaper(); std::lock_guard<std::mutex> lk(r.m); r.users++; }
void reaper_release() {
    Reaper& r = reaper();
    bool last;
    { std::lock_guard<std::mutex> lk(r.m); last = --r.users <= 0; if (last) r.users = 0; }
    if (last) r.shutdown();
}
void reaper_drain() { reaper().drain(); }
bool reaper_running() { Reaper& r = reaper(); std::lock_guard<std::mutex> lk(r.m); return r.running; }


// =================================================================== BGZF / BAM
namespace {

struct FileBytes {
    PodVec<uint8_t> data;   // not zero-filled before the read
    // Whole file into memory: a few threads pread their pieces (copying out of the page cache and first-touching the buffer is CPU
    // work: one thread moves ~2.5 GB/s, and these are 0.4-0.8 GB files on the CLI's critical path)
    explicit FileBytes(const std::string& path) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw Error("cannot open " + path);
        struct stat st;
        if (::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {   // not a regular file (a pipe, /dev/stdin): read it sequentially
            uint8_t buf[1 << 16];
            for (;;) {
                const ssize_t n = ::read(fd, buf, sizeof buf);
                if (n < 0) { if (errno == EINTR) continue; ::close(fd); throw Error("read error on " + path); }
                if (n == 0) break;
                data.insert(data.end(), buf, buf + n);
            }
            ::close(fd);
            return;
        }
        const size_t n = size_t(st.st_size);
        data.resize(n);
        constexpr size_t PIECE = size_t(32) << 20;
        const size_t pieces = (n + PIECE - 1) / PIECE;
        const size_t nthreads = std::max<size_t>(1, std::min<size_t>(pieces, 6));
        std::atomic<size_t> next{0};
        std::atomic<bool> failed{false};
        auto work = [&] {
            for (size_t i; (i = next.fetch_add(1)) < pieces;) {
                size_t off = i * PIECE;
                const size_t end = std::min(n, off + PIECE);
                while (off < end) {
                    const ssize_t got = ::pread(fd, data.data() + off, end - off, off_t(off));
                    if (got < 0 && errno == EINTR) continue;
                    if (got <= 0) { failed = true; return; }
                    off += size_t(got);
                }
            }
        };
        std::vector<std::thread> th;
        for (size_t t = 1; t < nthreads; t++) th.emplace_back(work);
        work();
        for (auto& x : th) x.join();
        ::close(fd);
        if (failed) throw Error("short read on " + path);
    }
};

// Streaming BGZF inflater: yields the concatenated uncompressed stream block by block.
class BgzfReader {
  public:
    explicit BgzfReader(const PodVec<uint8_t>& file) : f_(file) {}
    // Append the next block's payload to `out`; false at EOF.
    bool next_block(std::vector<uint8_t>& out) {
        if (off_ >= f_.size()) return false;
        if (off_ + 18 > f_.size()) throw Error("truncated BGZF header");
        const uint8_t* p = f_.data() + off_;
        if (p[0] != 31 || p[1] != 139 || p[2] != 8 || !(p[3] & 4)) throw Error("not a BGZF block");
        uint32_t xlen = p[10] | (p[11] << 8);
        uint32_t bsize = 0;
        bool found = false;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            const uint8_t* e = p + 12 + x;
            uint32_t slen = e[2] | (e[3] << 8);
            if (e[0] == 'B' && e[1] == 'C' && slen == 2) { bsize = (e[4] | (e[5] << 8)) + 1u; found = true; }
            x += 4 + slen;
        }
        if (!found || off_ + bsize > f_.size()) throw Error("bad BGZF block size");
        const uint8_t* cdata = p + 12 + xlen;
        uint32_t clen = bsize - xlen - 12 - 8;
        uint32_t isize = p[bsize - 4] | (p[bsize - 3] << 8) | (p[bsize - 2] << 16) | (uint32_t(p[bsize - 1]) << 24);
        size_t old = out.size();
        out.resize(old + isize);
        if (isize) {
            z_stream zs;
            std::memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) throw Error("inflateInit2 failed");
            zs.next_in = const_cast<Bytef*>(cdata);
            zs.avail_in = clen;
            zs.next_out = out.data() + old;
            zs.avail_out = isize;
            int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END) throw Error("BGZF inflate failed");
        }
        off_ += bsize;
        return true;
    }

  private:
    const PodVec<uint8_t>& f_;
    size_t off_ = 0;
};

// Whole-file BGZF inflate on all host threads: BGZF blocks are independent deflate streams whose compressed and
// uncompressed sizes are in the block itself, so one scan lays out the output and the blocks inflate in parallel
// (the end-to-end path is ingest-bound once the kernels exist).
PodVec<uint8_t> bgzf_inflate_all(const PodVec<uint8_t>& f) {
    struct Blk { size_t in_off; uint32_t clen, isize; size_t out_off; };
    std::vector<Blk> blks;
    size_t off = 0, total = 0;
    while (off < f.size()) {
        if (off + 18 > f.size()) throw Error("truncated BGZF header");
        const uint8_t* p = f.data() + off;
        if (p[0] != 31 || p[1] != 139 || p[2] != 8 || !(p[3] & 4)) throw Error("not a BGZF block");
        const uint32_t xlen = p[10] | (p[11] << 8);
        uint32_t bsize = 0;
        bool found = false;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            const uint8_t* e = p + 12 + x;
            const uint32_t slen = e[2] | (e[3] << 8);
            if (e[0] == 'B' && e[1] == 'C' && slen == 2) { bsize = (e[4] | (e[5] << 8)) + 1u; found = true; }
            x += 4 + slen;
        }
        if (!found || bsize < xlen + 20 || off + bsize > f.size()) throw Error("bad BGZF block size");
        const uint32_t isize = p[bsize - 4] | (p[bsize - 3] << 8) | (p[bsize - 2] << 16) | (uint32_t(p[bsize - 1]) << 24);
        blks.push_back(Blk{off + 12 + xlen, bsize - xlen - 12 - 8, isize, total});
        total += isize;
        off += bsize;
    }
    auto tA = std::chrono::steady_clock::now();
    PodVec<uint8_t> out;
    out.resize(total);   // not touched here: the inflating threads first-touch their own blocks
    auto tB = std::chrono::steady_clock::now();
    size_t nthreads = std::max<size_t>(1, std::min<size_t>(std::thread::hardware_concurrency(), 32));
    if (const char* e = std::getenv("MP_THREADS")) nthreads = std::max<size_t>(1, size_t(std::atoi(e)));
    nthreads = std::min(nthreads, std::max<size_t>(1, blks.size() / 16));
    std::vector<std::string> errors(nthreads);
    auto work = [&](size_t t) {
        try {
            for (size_t b = t; b < blks.size(); b += nthreads) {
                const Blk& k = blks[b];
                if (!k.isize) continue;
                z_stream zs;
                std::memset(&zs, 0, sizeof zs);
                if (inflateInit2(&zs, -15) != Z_OK) throw Error("inflateInit2 failed");
                zs.next_in = const_cast<Bytef*>(f.data() + k.in_off);
                zs.avail_in = k.clen;
                zs.next_out = out.data() + k.out_off;
                zs.avail_out = k.isize;
                const int rc = inflate(&zs, Z_FINISH);
                inflateEnd(&zs);
                if (rc != Z_STREAM_END) throw Error("BGZF inflate failed");
            }
        } catch (const std::exception& e) { errors[t] = e.what(); }
    };
    if (nthreads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (size_t t = 0; t < nthreads; t++) th.emplace_back(work, t);
        for (auto& x : th) x.join();
    }
    for (const std::string& e : errors) if (!e.empty()) throw Error(e);
    if (std::getenv("MP_DEBUG")) std::fprintf(stderr, "[mp] bgzf: %zu blocks, %zu threads, alloc %.0f ms, inflate %.0f ms\n", blks.size(), nthreads, std::chrono::duration<double, std::milli>(tB - tA).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tB).count());
    return out;
}

inline uint32_t rd32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | (uint32_t(p[3]) << 24); }
inline uint16_t rd16(const uint8_t* p) { return uint16_t(p[0] | (p[1] << 8)); }

}  // namespace

namespace {
size_t io_threads(size_t cap) {
    size_t n = std::max<size_t>(1, std::min<size_t>(std::thread::hardware_concurrency(), cap));
    if (const char* e = std::getenv("MP_THREADS")) n = std::max<size_t>(1, size_t(std::atoi(e)));
    return n;
}
template <class F> void run_threads(size_t n, F f) {
    std::vector<std::string> errors(n);
    std::vector<std::thread> th;
    auto guarded = [&](size_t t) { try { f(t); } catch (const std::exception& e) { errors[t] = e.what(); if (errors[t].empty()) errors[t] = "error"; } };
    for (size_t t = 1; t < n; t++) th.emplace_back(guarded, t);
    guarded(0);
    for (auto& x : th) x.join();
    for (const std::string& e : errors) if (!e.empty()) throw Error(e);
}
}  // namespace

// BAM records -> ReadStore. One sequential hop over the record sizes finds the placed records; their fields and the cigar / base /
// quality / name pools are then sized once and filled by all host threads (record ranges), in file order.
void load_bam(const std::string& path, BamData& out) {
    const bool dbg = std::getenv("MP_DEBUG") != nullptr;
    auto clk = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = clk();
    PodVec<uint8_t> buf;
    {
        FileBytes fb(path);
        buf = bgzf_inflate_all(fb.data);
    }
    const auto t1 = clk();
    size_t cur = 0;
    auto need = [&](size_t n) -> bool { return buf.size() - cur >= n; };
    if (!need(12)) throw Error("empty BAM " + path);
    if (std::memcmp(buf.data() + cur, "BAM\1", 4) != 0) throw Error("bad BAM magic in " + path);
    uint32_t l_text = rd32(buf.data() + cur + 4);
    cur += 8;
    if (!need(l_text + 4)) throw Error("truncated BAM header");
    cur += l_text;
    uint32_t n_ref = rd32(buf.data() + cur);
    cur += 4;
    out.ref_names.clear();
    out.ref_lens.clear();
    for (uint32_t i = 0; i < n_ref; i++) {
        if (!need(4)) throw Error("truncated BAM refs");
        uint32_t l_name = rd32(buf.data() + cur);
        cur += 4;
        if (!need(l_name + 4)) throw Error("truncated BAM refs");
        out.ref_names.emplace_back(reinterpret_cast<const char*>(buf.data() + cur), l_name ? l_name - 1 : 0);
        cur += l_name;
        out.ref_lens.push_back(int64_t(rd32(buf.data() + cur)));
        cur += 4;
    }
    out.reads = ReadStore();
    ReadStore& rs = out.reads;
    std::vector<uint64_t> rec_at;   // buffer offset of every placed record (ref_id >= 0), file order
    rec_at.reserve(buf.size() / 160);
    while (need(4)) {
        const uint32_t bs = rd32(buf.data() + cur);
        if (!need(4 + size_t(bs))) throw Error("truncated BAM record");
        if (bs < 32) throw Error("malformed BAM record");
        if (int32_t(rd32(buf.data() + cur + 4)) >= 0) rec_at.push_back(cur + 4);
        // the hop is a dependent-load chain (next offset = this record's size); neighbouring records have similar sizes, so the
        // header some records ahead can be prefetched by extrapolation, which turns the chain from latency- into bandwidth-bound
        {
            const size_t ahead = cur + 24 * (4 + size_t(bs));
            if (ahead + 64 < buf.size()) { __builtin_prefetch(buf.data() + ahead); __builtin_prefetch(buf.data() + ahead + 64); }
        }
        cur += 4 + size_t(bs);
    }
    const auto t2 = clk();
    const size_t n = rec_at.size();
    const size_t nt = std::max<size_t>(1, std::min(io_threads(32), n / 4096));
    struct Sum { uint64_t cig = 0, seq = 0, qual = 0, name = 0; };
    std::vector<Sum> base(nt + 1);
    auto range = [&](size_t t) { return std::make_pair(n * t / nt, n * (t + 1) / nt); };
    auto fields = [&](size_t i, uint32_t& l_name, uint32_t& n_cig, uint32_t& l_seq, size_t& name_len) {
        const uint8_t* r = buf.data() + rec_at[i];
        const uint32_t bs = rd32(r - 4);
        l_name = r[8]; n_cig = rd16(r + 12); l_seq = rd32(r + 16);
        if (32 + uint64_t(l_name) + 4ull * n_cig + (uint64_t(l_seq) + 1) / 2 + l_seq > bs) throw Error("malformed BAM record");
        name_len = ::strnlen(reinterpret_cast<const char*>(r + 32), l_name);   // ReadStore keeps the name up to its first NUL
    };
    run_threads(nt, [&](size_t t) {
        Sum s;
        for (size_t i = range(t).first; i < range(t).second; i++) {
            uint32_t l_name, n_cig, l_seq; size_t nl;
            fields(i, l_name, n_cig, l_seq, nl);
            s.cig += n_cig; s.seq += (l_seq + 1) / 2; s.qual += l_seq; s.name += nl + 1;
        }
        base[t + 1] = s;
    });
    for (size_t t = 0; t < nt; t++) { base[t + 1].cig += base[t].cig; base[t + 1].seq += base[t].seq; base[t + 1].qual += base[t].qual; base[t + 1].name += base[t].name; }
    rs.tid.resize(n); rs.pos.resize(n); rs.end_pos.resize(n); rs.mapq.resize(n); rs.flag.resize(n); rs.l_seq.resize(n); rs.n_cigar.resize(n);
    rs.cigar_off.resize(n); rs.seq_off.resize(n); rs.qual_off.resize(n); rs.qname_off.resize(n);
    rs.cigar_pool.resize(base[nt].cig); rs.seq_pool.resize(base[nt].seq); rs.qual_pool.resize(base[nt].qual); rs.qname_pool.resize(base[nt].name);
    run_threads(nt, [&](size_t t) {
        Sum at = base[t];
        for (size_t i = range(t).first; i < range(t).second; i++) {
            uint32_t l_name, n_cig, l_seq; size_t nl;
            fields(i, l_name, n_cig, l_seq, nl);
            const uint8_t* r = buf.data() + rec_at[i];
            const uint8_t* p = r + 32;
            rs.tid[i] = int32_t(rd32(r));
            const int64_t pos = int32_t(rd32(r + 4));
            rs.pos[i] = pos;
            rs.mapq[i] = r[9];
            rs.flag[i] = rd16(r + 14);
            rs.l_seq[i] = l_seq;
            rs.n_cigar[i] = n_cig;
            rs.qname_off[i] = at.name;
            std::memcpy(rs.qname_pool.data() + at.name, p, nl);
            rs.qname_pool[at.name + nl] = 0;
            at.name += nl + 1;
            p += l_name;
            rs.cigar_off[i] = at.cig;
            int64_t e = pos;
            for (uint32_t k = 0; k < n_cig; k++) {
                const uint32_t c = rd32(p + 4 * k), op = c & 0xF, l = c >> 4;
                rs.cigar_pool[at.cig + k] = c;
                if (op == C_M || op == C_EQ || op == C_X || op == C_D || op == C_N) e += l;
            }
            rs.end_pos[i] = e;
            at.cig += n_cig;
            p += 4 * size_t(n_cig);
            rs.seq_off[i] = at.seq;
            std::memcpy(rs.seq_pool.data() + at.seq, p, (l_seq + 1) / 2);
            at.seq += (l_seq + 1) / 2;
            p += (l_seq + 1) / 2;
            rs.qual_off[i] = at.qual;
            std::memcpy(rs.qual_pool.data() + at.qual, p, l_seq);
            at.qual += l_seq;
        }
    });
    if (dbg) std::fprintf(stderr, "[mp] bam: read + inflate %.0f ms, record scan %.0f ms, parse on %zu threads %.0f ms\n", ms(t0, t1), ms(t1, t2), nt, ms(t2, clk()));
    // tid_begin (file is coordinate sorted: tids ascending)
    out.tid_begin.assign(n_ref + 1, out.reads.size());
    {
        size_t n = out.reads.size();
        size_t i = 0;
        for (uint32_t t = 0; t < n_ref; t++) {
            while (i < n && out.reads.tid[i] < int32_t(t)) i++;
            out.tid_begin[t] = i;
        }
        out.tid_begin[n_ref] = n;
    }
}

namespace {
void bgzf_write_block(FILE* f, const uint8_t* data, size_t n) {
    std::vector<uint8_t> comp(compressBound(uLong(n)) + 64);
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw Error("deflateInit2 failed");
    zs.next_in = const_cast<Bytef*>(data);
    zs.avail_in = uInt(n);
    zs.next_out = comp.data();
    zs.avail_out = uInt(comp.size());
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { deflateEnd(&zs); throw Error("deflate failed"); }
    size_t clen = zs.total_out;
    deflateEnd(&zs);
    uint32_t bsize = uint32_t(clen + 12 + 6 + 8);
    uint8_t hdr[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, 0, 0};
    hdr[16] = uint8_t((bsize - 1) & 0xFF);
    hdr[17] = uint8_t((bsize - 1) >> 8);
    std::fwrite(hdr, 1, 18, f);
    std::fwrite(comp.data(), 1, clen, f);
    uint32_t crc = uint32_t(crc32(crc32(0L, Z_NULL, 0), data, uInt(n)));
    uint8_t tail[8];
    for (int i = 0; i < 4; i++) tail[i] = uint8_t(crc >> (8 * i));
    for (int i = 0; i < 4; i++) tail[4 + i] = uint8_t(uint32_t(n) >> (8 * i));
    std::fwrite(tail, 1, 8, f);
}
void put32(std::vector<uint8_t>& b, uint32_t v) {
    for (int i = 0; i < 4; i++) b.push_back(uint8_t(v >> (8 * i)));
}
}  // namespace

void write_bam(const std::string& path, const std::vector<std::string>& ref_names, const std::vector<int64_t>& ref_lens,
               const ReadStore& reads) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw Error("cannot write " + path);
    std::vector<uint8_t> b;
    auto flush = [&](bool force) {
        size_t off = 0;
        while (b.size() - off >= 0xff00 || (force && off < b.size())) {
            size_t n = std::min<size_t>(0xff00, b.size() - off);
            bgzf_write_block(f, b.data() + off, n);
            off += n;
        }
        b.erase(b.begin(), b.begin() + long(off));
    };
    std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
    for (size_t i = 0; i < ref_names.size(); i++)
        text += "@SQ\tSN:" + ref_names[i] + "\tLN:" + std::to_string(ref_lens[i]) + "\n";
    b.insert(b.end(), {'B', 'A', 'M', 1});
    put32(b, uint32_t(text.size()));
    b.insert(b.end(), text.begin(), text.end());
    put32(b, uint32_t(ref_names.size()));
    for (size_t i = 0; i < ref_names.size(); i++) {
        put32(b, uint32_t(ref_names[i].size() + 1));
        b.insert(b.end(), ref_names[i].begin(), ref_names[i].end());
        b.push_back(0);
        put32(b, uint32_t(ref_lens[i]));
    }
    for (size_t i = 0; i < reads.size(); i++) {
        const char* name = reads.qname(i);
        size_t nl = std::strlen(name) + 1;
        uint32_t ncig = reads.n_cigar[i], lseq = reads.l_seq[i];
        uint32_t bs = uint32_t(32 + nl + 4 * ncig + (lseq + 1) / 2 + lseq);
        put32(b, bs);
        put32(b, uint32_t(reads.tid[i]));
        put32(b, uint32_t(reads.pos[i]));
        b.push_back(uint8_t(nl));
        b.push_back(reads.mapq[i]);
        b.push_back(0x48); b.push_back(0x12);  // bin (unused by our reader)
        b.push_back(uint8_t(ncig & 0xFF)); b.push_back(uint8_t(ncig >> 8));
        b.push_back(uint8_t(reads.flag[i] & 0xFF)); b.push_back(uint8_t(reads.flag[i] >> 8));
        put32(b, lseq);
        put32(b, 0xFFFFFFFFu);  // next refID
        put32(b, 0xFFFFFFFFu);  // next pos
        put32(b, 0);            // tlen
        b.insert(b.end(), name, name + nl);
        for (uint32_t k = 0; k < ncig; k++) put32(b, reads.cigar(i)[k]);
        const uint8_t* s = reads.seq_pool.data() + reads.seq_off[i];
        b.insert(b.end(), s, s + (lseq + 1) / 2);
        const uint8_t* q = reads.qual(i);
        b.insert(b.end(), q, q + lseq);
        if (b.size() >= (1u << 20)) flush(false);
    }
    flush(true);
    bgzf_write_block(f, nullptr, 0);  // EOF marker
    std::fclose(f);
}

// ------------------------------------------------------------------- ReadBuffer
// Underlying iterator = htslib region iterator created by `reader.fetch(tid, beg, target_len)`:
// yields, in file order, the records of `tid` whose end position is > beg.
bool ReadBuffer::next_record(size_t& idx) {
    if (!iter_valid_) return false;
    const ReadStore& rs = bam_.reads;
    while (cursor_ < cursor_end_) {
        size_t i = cursor_++;
        int64_t e = rs.end_pos[i];
        if (e <= rs.pos[i]) e = rs.pos[i] + 1;  // bam_endpos of a record without reference length
        if (e > iter_beg_) { idx = i; return true; }
    }
    return false;
}

void ReadBuffer::fetch(const std::string& chrom, uint64_t start, uint64_t end) {
    const ReadStore& rs = bam_.reads;
    if (has_overflow_) {
        inner_.push_back(overflow_);
        has_overflow_ = false;
    }
    int tid = bam_.tid_of(chrom);
    if (tid < 0) throw Error("sequence " + chrom + " not found in BAM header");
    bool reseek = inner_.empty();
    if (!reseek) {
        size_t last = inner_.back();
        if (uint64_t(rs.pos[last]) < start || rs.tid[last] != tid) reseek = true;
    }
    if (reseek) {
        iter_tid_ = tid;
        iter_beg_ = int64_t(start);
        cursor_ = bam_.tid_begin[size_t(tid)];
        cursor_end_ = bam_.tid_begin[size_t(tid) + 1];
        {   // records that can overlap `start` begin after start - max_ref_span: skip the rest by binary search
            int64_t span = bam_.max_ref_span;
            if (span <= 0) {
                for (size_t i = 0; i < rs.size(); i++) span = std::max<int64_t>(span, std::max<int64_t>(rs.end_pos[i] - rs.pos[i], 1));
                const_cast<BamData&>(bam_).max_ref_span = span;
            }
            int64_t from = int64_t(start) - span;
            const int64_t* p0 = rs.pos.data();
            cursor_ = size_t(std::lower_bound(p0 + cursor_, p0 + cursor_end_, from) - p0);
        }
        iter_valid_ = true;
        inner_.clear();
    } else {
        while (!inner_.empty() && rs.pos[inner_.front()] < int64_t(start)) inner_.pop_front();
    }
    size_t idx;
    while (next_record(idx)) {
        if (rs.flag[idx] & 0x4) continue;  // is_unmapped
        if (rs.pos[idx] >= int64_t(end)) {
            overflow_ = idx;
            has_overflow_ = true;
            break;
        }
        inner_.push_back(idx);
    }
}

// =================================================================== VCF
namespace {
std::vector<std::string> split(const std::string& s, char d) {
    std::vector<std::string> out;
    size_t a = 0;
    for (;;) {
        size_t b = s.find(d, a);
        if (b == std::string::npos) { out.push_back(s.substr(a)); break; }
        out.push_back(s.substr(a, b - a));
        a = b + 1;
    }
    return out;
}
}  // namespace

// The text is cut at line ends into one piece per host thread; the pieces are parsed concurrently and joined in file order.
namespace {
// Plain gzip (one or more members, not BGZF): sequential zlib inflate.
PodVec<uint8_t> gunzip_all(const PodVec<uint8_t>& f) {
    PodVec<uint8_t> out;
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 32) != Z_OK) throw Error("inflateInit2 failed");
    if (f.size() > 0x7FFFFFFFull) { inflateEnd(&zs); throw Error("gzip input larger than 2 GiB is not supported: compress with bgzip"); }
    zs.next_in = const_cast<Bytef*>(f.data());
    zs.avail_in = uInt(f.size());
    std::vector<uint8_t> chunk(size_t(4) << 20);
    for (;;) {
        zs.next_out = chunk.data();
        zs.avail_out = uInt(chunk.size());
        const int rc = inflate(&zs, Z_NO_FLUSH);
        out.insert(out.end(), chunk.data(), chunk.data() + (chunk.size() - zs.avail_out));
        if (rc == Z_STREAM_END) {
            if (zs.avail_in == 0) break;
            if (inflateReset(&zs) != Z_OK) { inflateEnd(&zs); throw Error("inflateReset failed"); }   // next member
            continue;
        }
        if (rc != Z_OK) { inflateEnd(&zs); throw Error("gzip inflate failed"); }
        if (zs.avail_in == 0 && zs.avail_out != 0) { inflateEnd(&zs); throw Error("truncated gzip stream"); }
    }
    inflateEnd(&zs);
    return out;
}

// BCF2 (what `bcf::Reader` reads besides VCF text; the reference's CLI help names both): typed values per the VCF/BCF
// specification section 6. Only the fields Variant::new looks at are decoded (src/common.rs:38-147): CHROM, POS, alleles,
// INFO/SOMATIC (flag), INFO/ANN (string), INFO/SVLEN (integers).
struct BcfCursor {
    const uint8_t* p; const uint8_t* end;
    void need(size_t n) const { if (size_t(end - p) < n) throw Error("truncated BCF record"); }
    uint8_t u8() { need(1); return *p++; }
    uint32_t u32() { need(4); uint32_t v = rd32(p); p += 4; return v; }
    int64_t int_of(int type) {   // one integer of the given width; missing / end-of-vector -> INT64_MIN
        switch (type) {
            case 1: { int8_t v = int8_t(u8()); return (v == INT8_MIN || v == INT8_MIN + 1) ? INT64_MIN : v; }
            case 2: { need(2); int16_t v = int16_t(rd16(p)); p += 2; return (v == INT16_MIN || v == INT16_MIN + 1) ? INT64_MIN : v; }
            case 3: { int32_t v = int32_t(u32()); return (v == INT32_MIN || v == INT32_MIN + 1) ? INT64_MIN : v; }
            default: throw Error("malformed BCF: integer expected");
        }
    }
    void descriptor(int& type, size_t& len) {   // typed-value header: low nibble type, high nibble length (15 = typed integer follows)
        const uint8_t d = u8();
        type = d & 0xF;
        len = d >> 4;
        if (len == 15) {
            int t2; size_t l2;
            descriptor(t2, l2);
            const int64_t v = int_of(t2);
            if (l2 != 1 || v < 0) throw Error("malformed BCF: bad vector length");
            len = size_t(v);
        }
    }
    static size_t width(int type) {
        switch (type) { case 0: return 0; case 1: return 1; case 2: return 2; case 3: return 4; case 5: return 4; case 7: return 1; default: throw Error("malformed BCF: unknown value type"); }
    }
    std::string str() {
        int t; size_t n;
        descriptor(t, n);
        if (n == 0) return std::string();
        if (t != 7) throw Error("malformed BCF: string expected");
        need(n);
        std::string v(reinterpret_cast<const char*>(p), n);
        p += n;
        while (!v.empty() && v.back() == 0) v.pop_back();
        return v;
    }
    void skip_value() { int t; size_t n; descriptor(t, n); need(n * width(t)); p += n * width(t); }
};

void load_bcf(const PodVec<uint8_t>& buf, VcfData& out) {
    if (buf.size() < 9 || buf[3] != 2) throw Error("unsupported BCF version (BCF2 expected)");
    const uint32_t l_text = rd32(buf.data() + 5);
    if (buf.size() < 9 + size_t(l_text)) throw Error("truncated BCF header");
    std::string text(reinterpret_cast<const char*>(buf.data() + 9), l_text);
    while (!text.empty() && text.back() == 0) text.pop_back();
    // dictionaries: strings = FILTER / INFO / FORMAT ids (PASS first unless IDX says otherwise), contigs = ##contig lines
    std::vector<std::string> strings, contigs;
    auto put = [](std::vector<std::string>& d, size_t idx, const std::string& v) { if (d.size() <= idx) d.resize(idx + 1); d[idx] = v; };
    bool explicit_pass = false;
    size_t next_str = 1, next_ctg = 0;   // implicit numbering when the header carries no IDX (0 = PASS)
    put(strings, 0, "PASS");
    std::map<std::string, size_t> seen;
    size_t a = 0;
    while (a < text.size()) {
        size_t e = text.find('\n', a);
        if (e == std::string::npos) e = text.size();
        const std::string line = text.substr(a, e - a);
        a = e + 1;
        const bool is_ctg = line.rfind("##contig=<", 0) == 0;
        const bool is_str = line.rfind("##INFO=<", 0) == 0 || line.rfind("##FILTER=<", 0) == 0 || line.rfind("##FORMAT=<", 0) == 0;
        if (!is_ctg && !is_str) continue;
        auto field = [&](const char* key) -> std::string {
            const std::string k = std::string(key) + "=";
            size_t q = line.find("<" + k);
            if (q == std::string::npos) q = line.find("," + k);
            if (q == std::string::npos) return std::string();
            q += 1 + k.size();
            const size_t r = line.find_first_of(",>", q);
            return line.substr(q, r == std::string::npos ? std::string::npos : r - q);
        };
        const std::string id = field("ID"), idx = field("IDX");
        if (id.empty()) continue;
        if (is_ctg) {
            put(contigs, idx.empty() ? next_ctg : size_t(std::strtoull(idx.c_str(), nullptr, 10)), id);
            next_ctg = contigs.size();
        } else {
            if (id == "PASS") { explicit_pass = true; if (!idx.empty()) put(strings, size_t(std::strtoull(idx.c_str(), nullptr, 10)), id); continue; }
            auto it = seen.find(id);   // one id may be declared as INFO and FORMAT: same dictionary entry
            if (it != seen.end()) continue;
            const size_t at = idx.empty() ? next_str : size_t(std::strtoull(idx.c_str(), nullptr, 10));
            put(strings, at, id);
            seen[id] = at;
            next_str = std::max(next_str, at) + 1;
        }
    }
    (void)explicit_pass;
    out.contigs = contigs;
    const uint8_t* p = buf.data() + 9 + l_text;
    const uint8_t* const end = buf.data() + buf.size();
    while (p < end) {
        if (size_t(end - p) < 8) throw Error("truncated BCF record");
        const uint32_t l_shared = rd32(p), l_indiv = rd32(p + 4);
        p += 8;
        if (size_t(end - p) < size_t(l_shared) + l_indiv) throw Error("truncated BCF record");
        BcfCursor c{p, p + l_shared};
        p += size_t(l_shared) + l_indiv;
        const int32_t chrom = int32_t(c.u32());
        const int32_t pos = int32_t(c.u32());
        c.u32();   // rlen
        c.u32();   // QUAL
        const uint32_t nai = c.u32();
        const uint32_t n_info = nai & 0xFFFF, n_allele = nai >> 16;
        c.u32();   // n_fmt << 24 | n_sample
        c.str();   // ID
        if (chrom < 0 || size_t(chrom) >= contigs.size() || contigs[size_t(chrom)].empty()) throw Error("BCF record refers to a contig that is not in the header");
        VcfRecord r;
        r.chrom = contigs[size_t(chrom)];
        r.pos = uint64_t(pos);
        for (uint32_t k = 0; k < n_allele; k++) {
            std::string al = c.str();
            if (k == 0) r.ref = std::move(al); else r.alts.push_back(std::move(al));
        }
        c.skip_value();   // FILTER
        for (uint32_t k = 0; k < n_info; k++) {
            int kt; size_t kn;
            c.descriptor(kt, kn);
            if (kn != 1) throw Error("malformed BCF: INFO key");
            const int64_t key = c.int_of(kt);
            const std::string* name = (key >= 0 && size_t(key) < strings.size()) ? &strings[size_t(key)] : nullptr;
            if (name && *name == "SOMATIC") { r.somatic = true; c.skip_value(); }
            else if (name && *name == "ANN") {
                const std::string v = c.str();
                const size_t cm = v.find(',');
                r.ann_first = cm == std::string::npos ? v : v.substr(0, cm);
            } else if (name && *name == "SVLEN") {
                int t; size_t n;
                c.descriptor(t, n);
                r.has_svlen = true;
                for (size_t q = 0; q < n; q++) r.svlen.push_back(c.int_of(t));
            } else c.skip_value();
        }
        out.records.push_back(std::move(r));
    }
}
}  // namespace

void load_vcf(const std::string& path, VcfData& out) {
    FileBytes fb(path);
    out.contigs.clear();
    out.records.clear();
    // bcf::Reader::from_path accepts VCF text, bgzip / gzip compressed VCF and BCF (htslib detects the format from the bytes)
    if (fb.data.size() >= 2 && fb.data[0] == 31 && fb.data[1] == 139) {
        const bool bgzf = fb.data.size() >= 18 && (fb.data[3] & 4) && fb.data[12] == 'B' && fb.data[13] == 'C';
        PodVec<uint8_t> plain = bgzf ? bgzf_inflate_all(fb.data) : gunzip_all(fb.data);
        fb.data.swap(plain);
    }
    if (fb.data.size() >= 5 && std::memcmp(fb.data.data(), "BCF", 3) == 0) { load_bcf(fb.data, out); return; }
    const PodVec<uint8_t>& buf = fb.data;
    const size_t nt = std::max<size_t>(1, std::min(io_threads(32), buf.size() >> 20));
    std::vector<size_t> cut(nt + 1, buf.size());
    cut[0] = 0;
    for (size_t t = 1; t < nt; t++) {
        size_t at = std::max(cut[t - 1], buf.size() * t / nt);
        while (at < buf.size() && buf[at] != '\n') at++;
        cut[t] = std::min(buf.size(), at + 1);
    }
    struct Piece {
        std::vector<VcfRecord> records;
        std::vector<std::pair<std::string, bool>> contigs;   // in order: every ##contig ID (true) and the first record of each contig (false)
    };
    std::vector<Piece> pieces(nt);
    run_threads(nt, [&](size_t t) {
        Piece& pc = pieces[t];
        pc.records.reserve((cut[t + 1] - cut[t]) / 120 + 16);   // (a record line with its ANN string is ~150 bytes)
        std::string line;
        for (size_t at = cut[t]; at < cut[t + 1];) {
            const uint8_t* nl = static_cast<const uint8_t*>(std::memchr(buf.data() + at, '\n', cut[t + 1] - at));
            const size_t end = nl ? size_t(nl - buf.data()) : cut[t + 1];
            line.assign(reinterpret_cast<const char*>(buf.data() + at), end - at);
            at = end + 1;
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty()) continue;
            if (line[0] == '#') {
                if (line.rfind("##contig=<", 0) == 0) {
                    size_t a = line.find("ID=");
                    if (a != std::string::npos) {
                        size_t b = line.find_first_of(",>", a);
                        pc.contigs.emplace_back(line.substr(a + 3, b - a - 3), true);
                    }
                }
                continue;
            }
            // fields by position in the line (no per-field copies: the INFO column with its ANN string is most of the line)
            const char* const lb = line.data();
            const char* const le = lb + line.size();
            const char* fb[8]; const char* fe[8];
            {
                const char* q = lb;
                int nf = 0;
                for (; nf < 8; nf++) {
                    const char* t = static_cast<const char*>(std::memchr(q, '\t', size_t(le - q)));
                    fb[nf] = q;
                    fe[nf] = t ? t : le;
                    if (!t) { nf++; break; }
                    q = t + 1;
                }
                if (nf < 8) throw Error("malformed VCF line: " + line);
            }
            VcfRecord r;
            r.chrom.assign(fb[0], fe[0]);
            {   // POS: leading decimal digits, like strtoull
                uint64_t v = 0;
                for (const char* q = fb[1]; q < fe[1] && *q >= '0' && *q <= '9'; q++) v = v * 10 + uint64_t(*q - '0');
                r.pos = v - 1;
            }
            r.ref.assign(fb[3], fe[3]);
            for (const char* a = fb[4];;) {   // ALT, comma separated (an empty field is one empty allele, like split())
                const char* c = static_cast<const char*>(std::memchr(a, ',', size_t(fe[4] - a)));
                r.alts.emplace_back(a, c ? c : fe[4]);
                if (!c) break;
                a = c + 1;
            }
            for (const char* a = fb[7];;) {   // INFO, ';' separated
                const char* c = static_cast<const char*>(std::memchr(a, ';', size_t(fe[7] - a)));
                const char* e = c ? c : fe[7];
                const size_t n = size_t(e - a);
                if (n == 7 && std::memcmp(a, "SOMATIC", 7) == 0) r.somatic = true;
                else if (n >= 4 && std::memcmp(a, "ANN=", 4) == 0) {
                    const char* cm = static_cast<const char*>(std::memchr(a + 4, ',', n - 4));
                    r.ann_first.assign(a + 4, cm ? cm : e);
                } else if (n >= 6 && std::memcmp(a, "SVLEN=", 6) == 0) {
                    r.has_svlen = true;
                    for (const auto& x : split(std::string(a + 6, e), ',')) r.svlen.push_back(x == "." ? INT64_MIN : std::atoll(x.c_str()));
                }
                if (!c) break;
                a = c + 1;
            }
            if (pc.records.empty() || pc.records.back().chrom != r.chrom) {   // (records of a contig come in a row: look only when it changes)
                bool seen = false;
                for (const auto& c : pc.contigs) seen |= !c.second && c.first == r.chrom;
                if (!seen) pc.contigs.emplace_back(r.chrom, false);
            }
            pc.records.push_back(std::move(r));
        }
    });
    size_t total = 0;
    for (const Piece& pc : pieces) total += pc.records.size();
    out.records.reserve(total);
    for (Piece& pc : pieces) {
        for (auto& c : pc.contigs)   // a contig named only by records is listed when first seen
            if (c.second || std::find(out.contigs.begin(), out.contigs.end(), c.first) == out.contigs.end()) out.contigs.push_back(c.first);
        for (VcfRecord& r : pc.records) out.records.push_back(std::move(r));
        std::vector<VcfRecord>().swap(pc.records);
    }
}

void VcfData::build_index() const {
    if (indexed) return;
    by_chrom.clear();
    ContigIndex* ci = nullptr;
    const std::string* last = nullptr;
    for (size_t i = 0; i < records.size(); i++) {
        if (!last || records[i].chrom != *last) {   // (one map look-up per run of a contig's records, not per record)
            ci = &by_chrom[records[i].chrom];
            last = &records[i].chrom;
        }
        if (!ci->recs.empty() && records[ci->recs.back()].pos > records[i].pos) ci->sorted = false;
        ci->recs.push_back(i);
    }
    indexed = true;
}

static thread_local std::string* g_warning_sink = nullptr;   // set while genes are loaded on worker threads: warnings are printed in gene order afterwards
static void warn_or_error(const std::string& msg, bool warning_only) {  // common.rs:62-69
    if (!warning_only) throw Error(msg);
    if (g_warning_sink) { *g_warning_sink += msg; g_warning_sink->push_back('\n'); }
    else std::fprintf(stderr, "%s\n", msg.c_str());
}

void variants_from_record(const VcfRecord& rec, bool warning_only, std::vector<Variant>& out) {
    bool is_germline = !rec.somatic;  // common.rs:75
    // Annotation::new (common.rs:21-35): first '|' field of the first ANN entry that contains "p."
    std::string prot_change;
    if (!rec.ann_first.empty()) {
        for (const auto& e : split(rec.ann_first, '|'))
            if (e.find("p.") != std::string::npos) { prot_change = e; break; }
    }
    const std::string& ref = rec.ref;
    for (const auto& a : rec.alts) {
        Variant v;
        v.pos = rec.pos;
        v.is_germline = is_germline;
        v.prot_change = prot_change;
        if (a.size() == 1 && ref.size() > 1) {  // common.rs:86-92
            v.kind = VK_DEL;
            v.len = ref.size() - 1;
            out.push_back(v);
        } else if (a.size() > 1 && ref.size() == 1) {
            if (a[0] == '<') {
                if (a == "<DEL>") {  // common.rs:95-142
                    if (!rec.has_svlen || rec.svlen.empty()) {
                        warn_or_error("Found no 'SVLEN' info tag for <DEL> alternative allele at chr " + rec.chrom + " pos " + std::to_string(rec.pos), warning_only);
                    } else if (rec.svlen.size() > 1) {
                        warn_or_error("microphaser does not handle multiallelic records. Please normalize, e.g. with `bcftools norm -m-`.", warning_only);
                    } else if (rec.svlen[0] == INT64_MIN) {
                        warn_or_error("Found no 'SVLEN' info tag for <DEL> alternative allele on contig " + rec.chrom + " at pos " + std::to_string(rec.pos), warning_only);
                    } else {
                        v.kind = VK_DEL;
                        v.len = uint64_t(rec.svlen[0] < 0 ? -rec.svlen[0] : rec.svlen[0]);
                        out.push_back(v);
                    }
                } else {
                    warn_or_error("Alternative allele type '" + a + "' not yet supported, but found on contig " + rec.chrom + " at position " + std::to_string(rec.pos) + ".", warning_only);
                }
            } else {  // common.rs:150-156
                v.kind = VK_INS;
                v.seq = a;
                v.len = a.size() - 1;
                out.push_back(v);
            }
        } else if (a.size() == 1 && ref.size() == 1) {  // common.rs:158-164
            v.kind = VK_SNV;
            v.alt = uint8_t(a[0]);
            out.push_back(v);
        } else {  // common.rs:165-171
            std::fprintf(stderr, "Unsupported variant %s -> %s\n", ref.c_str(), a.c_str());
        }
    }
}

void gene_variants(const VcfData& vcf, const std::string& chrom, uint64_t start, uint64_t end, bool warning_only,
                   std::vector<Variant>& out) {
    // bcf::buffer::RecordBuffer::fetch resolves the contig through the header (name2rid) and errors if unknown
    if (std::find(vcf.contigs.begin(), vcf.contigs.end(), chrom) == vcf.contigs.end())
        throw Error("contig " + chrom + " not found in VCF header");
    std::map<uint64_t, std::vector<Variant>> tree;  // variant_tree.insert(pos, ...) : later record replaces
    vcf.build_index();
    auto ci = vcf.by_chrom.find(chrom);
    if (ci != vcf.by_chrom.end()) {
        const auto& recs = ci->second.recs;
        size_t a = 0, e = recs.size();
        if (ci->second.sorted) {
            a = size_t(std::lower_bound(recs.begin(), recs.end(), start, [&](size_t r, uint64_t p) { return vcf.records[r].pos < p; }) - recs.begin());
            e = size_t(std::upper_bound(recs.begin(), recs.end(), end, [&](uint64_t p, size_t r) { return p < vcf.records[r].pos; }) - recs.begin());
        }
        for (size_t k = a; k < e; k++) {
            const VcfRecord& r = vcf.records[recs[k]];
            if (r.pos < start || r.pos > end) continue;
            std::vector<Variant> vs;
            variants_from_record(r, warning_only, vs);
            tree[r.pos] = std::move(vs);
        }
    }
    out.clear();
    for (auto& kv : tree)
        for (auto& v : kv.second) out.push_back(std::move(v));
}

// =================================================================== FASTA
IndexedFasta::IndexedFasta(const std::string& path) : path_(path) {
    std::ifstream fai(path + ".fai");
    if (!fai) throw Error("cannot open " + path + ".fai");
    std::string line;
    while (std::getline(fai, line)) {
        if (line.empty()) continue;
        auto f = split(line, '\t');
        if (f.size() < 5) throw Error("malformed .fai line");
        Entry e{std::strtoull(f[1].c_str(), nullptr, 10), std::strtoull(f[2].c_str(), nullptr, 10),
                std::strtoull(f[3].c_str(), nullptr, 10), std::strtoull(f[4].c_str(), nullptr, 10), 0, 0};
        e.region_len = e.len;
        if (f.size() >= 7) {  // region-offset extension (tests/golden mini references): only [start, start+n) is stored
            e.region_start = std::strtoull(f[5].c_str(), nullptr, 10);
            e.region_len = std::strtoull(f[6].c_str(), nullptr, 10);
        }
        idx_[f[0]] = e;
    }
}

void MemFasta::fetch(const std::string& chrom, uint64_t start, uint64_t stop, std::vector<uint8_t>& out) const {
    auto it = contigs.find(chrom);
    if (it == contigs.end()) throw Error("Unknown sequence name: " + chrom);
    const std::string& s = *it->second;
    if (stop > s.size()) throw Error("FASTA read interval was out of bounds");
    if (start > stop) throw Error("Invalid query interval");
    out.assign(s.begin() + long(start), s.begin() + long(stop));
}

void IndexedFasta::ensure_loaded() const {
    std::call_once(once_, [&] {
        FileBytes fb(path_);
        file_.swap(fb.data);
    });
}

void IndexedFasta::fetch(const std::string& chrom, uint64_t start, uint64_t stop, std::vector<uint8_t>& out) const {
    auto it = idx_.find(chrom);
    if (it == idx_.end()) throw Error("Unknown sequence name: " + chrom);
    const Entry& e = it->second;
    if (stop > e.len) throw Error("FASTA read interval was out of bounds");
    if (start > stop) throw Error("Invalid query interval");
    ensure_loaded();
    out.clear();
    out.reserve(stop - start);
    for (uint64_t p = start; p < stop; p++) {
        if (p < e.region_start || p >= e.region_start + e.region_len) { out.push_back('N'); continue; }
        uint64_t q = p - e.region_start;
        uint64_t off = e.offset + (q / e.line_bases) * e.line_bytes + (q % e.line_bases);
        if (off >= file_.size()) throw Error("FASTA file shorter than its index");
        out.push_back(file_[off]);
    }
}

// =================================================================== GTF
namespace {
struct GtfRecord {
    std::string seqname, feature, strand, frame;
    uint64_t start = 0, end = 0;
    std::vector<std::pair<std::string, std::string>> attrs;
    const std::string* get(const char* k) const {
        for (const auto& a : attrs)
            if (a.first == k) return &a.second;
        return nullptr;
    }
};

bool parse_gtf_line(const std::string& line, GtfRecord& r) {
    if (line.empty() || line[0] == '#') return false;
    auto f = split(line, '\t');
    if (f.size() < 9) throw Error("malformed GTF line: " + line);
    r.seqname = f[0];
    r.feature = f[2];
    r.start = std::strtoull(f[3].c_str(), nullptr, 10);
    r.end = std::strtoull(f[4].c_str(), nullptr, 10);
    r.strand = f[6];
    r.frame = f[7];
    r.attrs.clear();
    // rust-bio GTF2 attribute grammar: \s*key<space>value;  with quotes trimmed from value
    const std::string& a = f[8];
    size_t i = 0, n = a.size();
    while (i < n) {
        while (i < n && (a[i] == ' ' || a[i] == ';')) i++;
        size_t ks = i;
        while (i < n && a[i] != ' ' && a[i] != ';') i++;
        if (i >= n || a[i] != ' ') break;
        std::string key = a.substr(ks, i - ks);
        i++;
        size_t vs = i;
        while (i < n && a[i] != ';') i++;
        std::string val = a.substr(vs, i - vs);
        while (!val.empty() && (val.front() == '\'' )) val.erase(val.begin());
        while (!val.empty() && (val.back() == '\'')) val.pop_back();
        while (!val.empty() && (val.front() == '"')) val.erase(val.begin());
        while (!val.empty() && (val.back() == '"')) val.pop_back();
        r.attrs.emplace_back(std::move(key), std::move(val));
    }
    return true;
}

Interval make_interval(uint64_t start, uint64_t end, const std::string& frame) {  // common.rs:329-338
    Interval iv;
    iv.start = start;
    iv.end = end;
    iv.frame = frame == "." ? 0 : std::strtoull(frame.c_str(), nullptr, 10);
    return iv;
}
}  // namespace

void stream_gtf(std::istream& in, const std::function<void(const Gene&)>& on_gene, bool use_three_prime_utr) {
    bool have_gene = false;
    Gene gene;
    bool start_codon_found = false, three_prime_found = false;
    std::string last_chrom = "not_yet_set";
    uint64_t last_start = 0;
    std::string line;
    GtfRecord rec;
    auto need = [&](const char* k, const char* msg) -> const std::string& {
        const std::string* v = rec.get(k);
        if (!v) throw Error(msg);
        return *v;
    };
    auto last_tx = [&](const char* msg) -> Transcript& {
        if (!have_gene) throw Error("no gene record before feature in GTF");
        if (gene.transcripts.empty()) throw Error(msg);
        return gene.transcripts.back();
    };
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (!parse_gtf_line(line, rec)) continue;
        if (rec.feature == "gene") {  // microphasing.rs:1986-2018
            if (have_gene) {
                on_gene(gene);
                last_chrom = gene.chrom;
                last_start = gene.start();
            }
            const std::string& gene_name = need("gene_name", "missing gene_name in GTF");
            if (last_chrom == rec.seqname && !(last_start <= rec.start)) {
                throw Error("Your GTF file is not sorted correctly. Gene " + gene_name + " starts at " +
                            std::to_string(rec.start) + ", while previous gene record started at " + std::to_string(last_start) + ".");
            }
            gene = Gene();
            gene.id = need("gene_id", "missing gene_id in GTF");
            gene.name = gene_name;
            gene.chrom = rec.seqname;
            gene.interval = make_interval(rec.start - 1, rec.end, rec.frame);
            gene.biotype = need("gene_biotype", "missing gene_biotype in GTF");
            have_gene = true;
        } else if (rec.feature == "transcript") {  // :2019-2040
            start_codon_found = false;
            three_prime_found = false;
            if (!have_gene) throw Error("no gene record before transcript in GTF");
            Transcript t;
            t.id = need("transcript_id", "missing transcript_id attribute in GTF");
            t.biotype = need("transcript_biotype", "missing transcript_biotype in GTF");
            if (rec.strand == "+") t.strand = FORWARD;
            else if (rec.strand == "-") t.strand = REVERSE;
            else throw Error("Unsupported Strand orientation! Only Forward (+) and Reverse(-) allowed");
            gene.transcripts.push_back(std::move(t));
        } else if (rec.feature == "CDS") {  // :2041-2055
            last_tx("no transcript record before exon in GTF").exons.push_back(make_interval(rec.start - 1, rec.end, rec.frame));
        } else if (rec.feature == "start_codon") {  // :2056-2082
            if (start_codon_found) continue;
            start_codon_found = true;
            Transcript& t = last_tx("no transcript record before start codon in GTF");
            if (t.exons.empty()) throw Error("no exon record before start codon in GTF");
            if (rec.strand == "+") t.exons.back().start = rec.start - 1;
            else t.exons.back().end = rec.end;
        } else if (use_three_prime_utr && rec.feature == "three_prime_utr") {  // :2083-2122
            Transcript& t = last_tx("no transcript record before exon in GTF");
            if (three_prime_found) {
                t.exons.push_back(make_interval(rec.start - 1, rec.end, rec.frame));
            } else {
                three_prime_found = true;
                if (t.exons.empty()) throw Error("no exon record before start codon in GTF");
                if (rec.strand == "+") t.exons.back().end = rec.end;
                else t.exons.back().start = rec.start - 1;
            }
        }
    }
    if (have_gene) on_gene(gene);
}

// Pass 1 (sequential, cheap): the GTF stream and the stateful read buffer (its contents depend on the previous gene's fetch).
// Pass 2 (all host threads): reference bases and variants of every gene, which are independent. Errors surface as in a gene-by-gene
// run: the first failing gene in GTF order, refseq before reads before variants (:895-942); warnings are printed in gene order.
void load_gene_inputs(std::istream& gtf, const BamData& bam, const VcfData& vcf, const RefSource& fasta,
                      bool warning_only, const std::function<void(GeneInput&)>& on_gene, bool use_three_prime_utr) {
    const bool dbg = std::getenv("MP_DEBUG") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    ReadBuffer rb(bam);
    std::vector<GeneInput> genes;
    struct GeneErr { std::string fasta, reads, variants, warnings; };
    std::vector<GeneErr> errs;
    std::string gtf_err;
    try {
        stream_gtf(gtf, [&](const Gene& g) {
            if (g.biotype != "protein_coding") return;  // microphasing.rs:1964
            genes.emplace_back();
            errs.emplace_back();
            GeneInput& gi = genes.back();
            gi.gene = g;
            try {
                rb.fetch(g.chrom, g.start(), g.end());                      // :905
                gi.reads.assign(rb.records().begin(), rb.records().end());
            } catch (const std::exception& e) { errs.back().reads = e.what(); }
        }, use_three_prime_utr);
    } catch (const std::exception& e) { gtf_err = e.what(); if (gtf_err.empty()) gtf_err = "error"; }
    const auto t1 = std::chrono::steady_clock::now();
    vcf.build_index();
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (size_t g; (g = next.fetch_add(1)) < genes.size();) {
            GeneInput& gi = genes[g];
            try { fasta.fetch(gi.gene.chrom, gi.gene.start(), gi.gene.end() + 100, gi.refseq); }  // :895-901
            catch (const std::exception& e) { errs[g].fasta = e.what(); continue; }
            if (!errs[g].reads.empty()) continue;
            g_warning_sink = &errs[g].warnings;
            try { gene_variants(vcf, gi.gene.chrom, gi.gene.start(), gi.gene.end(), warning_only, gi.variants); }  // :932-942
            catch (const std::exception& e) { errs[g].variants = e.what(); }
            g_warning_sink = nullptr;
        }
    };
    size_t hw = std::max<size_t>(1, std::min<size_t>(std::thread::hardware_concurrency(), 32));
    if (const char* e = std::getenv("MP_THREADS")) hw = std::max<size_t>(1, size_t(std::atoi(e)));
    const size_t nt = std::max<size_t>(1, std::min(hw, genes.size() / 16));
    std::vector<std::thread> th;
    for (size_t k = 1; k < nt; k++) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
    if (dbg) std::fprintf(stderr, "[mp]   gene inputs: GTF + read buffer %.0f ms, refseq + variants on %zu threads %.0f ms\n",
                          std::chrono::duration<double, std::milli>(t1 - t0).count(), nt,
                          std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    for (size_t g = 0; g < genes.size(); g++) {
        const GeneErr& e = errs[g];
        if (!e.fasta.empty()) throw Error(e.fasta);
        if (!e.reads.empty()) throw Error(e.reads);
        if (!e.warnings.empty()) std::fputs(e.warnings.c_str(), stderr);
        if (!e.variants.empty()) throw Error(e.variants);
        on_gene(genes[g]);
    }
    if (!gtf_err.empty()) throw Error(gtf_err);
}

}  // namespace mp
