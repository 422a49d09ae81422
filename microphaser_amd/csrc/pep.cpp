#include "pep.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <sstream>
#include <thread>

#include "batch.hpp"
#include "kernels_pep.hpp"

namespace mp {

[[noreturn]] void throw_hip(hipError_t e, const char* file, int line);
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw_hip(e_, __FILE__, __LINE__); } while (0)

std::string peptide_from_key(uint64_t key, uint32_t L) {
    std::string s(L, 'A');
    for (uint32_t j = 0; j < L; j++) s[L - 1 - j] = char('A' + ((key >> (5 * j)) & 31));
    return s;
}
uint64_t peptide_to_key(const std::string& pep) {
    uint64_t k = 0;
    for (char c : pep) k = (k << 5) | uint64_t((c - 'A') & 31);
    return k;
}

std::string PeptideResult::binary() const {
    std::string b;
    auto u64 = [&](uint64_t v) { for (int i = 0; i < 8; i++) b.push_back(char(v >> (8 * i))); };
    u64(keys.size());
    for (uint64_t k : keys) { u64(peptide_len); b += peptide_from_key(k, peptide_len); }
    return b;
}

void build_reference_device(int device, std::string_view fasta_text, uint32_t L, PeptideResult& out, bool want_fasta) {
    if (L == 0 || L > 12) throw Error("peptide length must be 1..12 for the device peptidome (5-bit residue keys in a u64)");
    out = PeptideResult();
    out.peptide_len = L;
    // parse records (bio::io::fasta::Reader), lay out the windows: i = 0, 3, ... while i + 3L <= len  (:165-174)
    // The text is cut at record starts ("\n>") and the pieces are parsed by all host threads (a whole-exome `normal` FASTA is 14 M
    // records per 2500 transcripts); the pieces' bases, window offsets and strand flags are then joined in file order.
    PodVec<uint8_t> nt;
    PodVec<uint64_t> off;
    PodVec<uint8_t> rev;
    std::vector<std::pair<std::string, uint64_t>> recs;  // (id, number of windows); only kept when the translated FASTA is wanted
    {
        struct Piece {
            PodVec<uint8_t> nt, rev;
            PodVec<uint64_t> off;      // relative to the piece's own bases
            std::vector<std::pair<std::string, uint64_t>> recs;
        };
        const char* const text = fasta_text.data();
        const size_t size = fasta_text.size();
        const size_t nparts = std::max<size_t>(1, std::min<size_t>(host_threads(), size / (size_t(4) << 20) + 1));
        std::vector<size_t> cut(nparts + 1, size);
        cut[0] = 0;
        for (size_t t = 1; t < nparts; t++) {   // the next record start at or after the even split (a '>' at the beginning of a line)
            size_t at = std::max(cut[t - 1], size * t / nparts);
            for (;;) {
                const char* gt = at < size ? static_cast<const char*>(std::memchr(text + at, '>', size - at)) : nullptr;
                if (!gt) { at = size; break; }
                at = size_t(gt - text);
                if (at == 0 || text[at - 1] == '\n') break;
                at++;
            }
            cut[t] = at;
        }
        std::vector<Piece> pieces(nparts);
        auto parse = [&](size_t t) {
            Piece& P = pieces[t];
            const char* p = text + cut[t];
            const char* const end = text + cut[t + 1];
            P.nt.reserve(size_t(end - p));
            P.off.reserve(size_t(end - p) / (3 * size_t(L) + 20) + 16);
            P.rev.reserve(P.off.capacity());
            std::string id;
            bool have = false;
            uint64_t base = 0;
            auto flush = [&]() {
                if (!have) return;
                const uint8_t r = (!id.empty() && id.back() == 'F') ? 0 : 1;  // :161-164
                const uint64_t len = P.nt.size() - base;
                uint64_t nwin = 0;
                for (uint64_t i = 0; i + 3ull * L <= len; i += 3) { P.off.push_back(base + i); P.rev.push_back(r); nwin++; }
                if (want_fasta) P.recs.emplace_back(id, nwin);
            };
            while (p < end) {
                const char* nl = static_cast<const char*>(std::memchr(p, '\n', size_t(end - p)));
                const char* le = nl ? nl : end;
                const char* lend = (le > p && le[-1] == '\r') ? le - 1 : le;
                if (lend > p && *p == '>') {
                    flush();
                    const char* q = p + 1;
                    while (q < lend && *q != ' ' && *q != '\t') q++;
                    id.assign(p + 1, q);
                    base = P.nt.size();
                    have = true;
                } else if (have) {
                    P.nt.insert(P.nt.end(), reinterpret_cast<const uint8_t*>(p), reinterpret_cast<const uint8_t*>(lend));
                }
                p = nl ? nl + 1 : end;
            }
            flush();
        };
        {
            std::vector<std::thread> th;
            for (size_t t = 1; t < nparts; t++) th.emplace_back(parse, t);
            parse(0);
            for (auto& x : th) x.join();
        }
        std::vector<size_t> nt_at(nparts + 1, 0), w_at(nparts + 1, 0);
        for (size_t t = 0; t < nparts; t++) { nt_at[t + 1] = nt_at[t] + pieces[t].nt.size(); w_at[t + 1] = w_at[t] + pieces[t].off.size(); }
        nt.resize(nt_at[nparts]);
        off.resize(w_at[nparts]);
        rev.resize(w_at[nparts]);
        auto join = [&](size_t t) {
            const Piece& P = pieces[t];
            if (!P.nt.empty()) std::memcpy(nt.data() + nt_at[t], P.nt.data(), P.nt.size());
            if (!P.rev.empty()) std::memcpy(rev.data() + w_at[t], P.rev.data(), P.rev.size());
            for (size_t i = 0; i < P.off.size(); i++) off[w_at[t] + i] = P.off[i] + nt_at[t];
        };
        {
            std::vector<std::thread> th;
            for (size_t t = 1; t < nparts; t++) th.emplace_back(join, t);
            join(0);
            for (auto& x : th) x.join();
        }
        if (want_fasta) for (Piece& P : pieces) for (auto& r : P.recs) recs.push_back(std::move(r));
    }
    const uint64_t n = off.size();
    out.n_peptides = n;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        throw Error("no HIP device available: peptide translation runs on the GPU, there is no CPU fallback");
    HIP_OK(hipSetDevice(device));
    hipStream_t stream;
    HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    hipEvent_t e0, e1, e2;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1)); HIP_OK(hipEventCreate(&e2));
    uint8_t *d_nt = nullptr, *d_rev = nullptr, *d_aa = nullptr;
    uint64_t *d_off = nullptr, *d_keys = nullptr, *d_tmp = nullptr, *d_out = nullptr;
    uint32_t* d_err = nullptr;
    std::vector<uint8_t> aa(want_fasta ? n * L : 0);
    if (n) {
        HIP_OK(hipMalloc(&d_nt, nt.size() + 64)); HIP_OK(hipMalloc(&d_rev, n)); HIP_OK(hipMalloc(&d_aa, n * L));
        HIP_OK(hipMalloc(&d_off, n * 8)); HIP_OK(hipMalloc(&d_keys, n * 8)); HIP_OK(hipMalloc(&d_tmp, n * 8)); HIP_OK(hipMalloc(&d_out, n * 8));
        HIP_OK(hipMalloc(&d_err, 4));
        HIP_OK(hipMemcpyAsync(d_nt, nt.data(), nt.size(), hipMemcpyHostToDevice, stream));
        HIP_OK(hipMemcpyAsync(d_off, off.data(), n * 8, hipMemcpyHostToDevice, stream));
        HIP_OK(hipMemcpyAsync(d_rev, rev.data(), n, hipMemcpyHostToDevice, stream));
        HIP_OK(hipMemsetAsync(d_err, 0, 4, stream));
        HIP_OK(hipEventRecord(e0, stream));
        device_translate(d_nt, d_off, d_rev, n, L, d_aa, d_keys, d_err, stream);
        HIP_OK(hipEventRecord(e1, stream));
        uint64_t nu = device_sort_unique(d_keys, d_tmp, d_out, n, 5 * L, stream);
        HIP_OK(hipEventRecord(e2, stream));
        uint32_t err = 0;
        out.keys.resize(nu);
        HIP_OK(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, stream));
        if (want_fasta) HIP_OK(hipMemcpyAsync(aa.data(), d_aa, n * L, hipMemcpyDeviceToHost, stream));
        if (nu) HIP_OK(hipMemcpyAsync(out.keys.data(), d_out, nu * 8, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        HIP_OK(hipEventElapsedTime(&out.translate_ms, e0, e1));
        HIP_OK(hipEventElapsedTime(&out.dedup_ms, e1, e2));
        for (void* p : {(void*)d_nt, (void*)d_rev, (void*)d_aa, (void*)d_off, (void*)d_keys, (void*)d_tmp, (void*)d_out, (void*)d_err}) hipFree(p);
        if (err) throw Error("reference would panic: called `Result::unwrap()` on an `Err` value (codon with a base other than A, C, G, T)");
    }
    hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(e2);
    hipStreamDestroy(stream);
    // FASTA in record order (fasta_writer.write(id, None, pepseq), :171)
    if (want_fasta) {
        size_t total = 0;
        for (const auto& r : recs) total += size_t(r.second) * (r.first.size() + L + 3);
        out.fasta.reserve(total);
        uint64_t k = 0;
        for (const auto& r : recs)
            for (uint64_t w = 0; w < r.second; w++, k++) {
                out.fasta.push_back('>');
                out.fasta += r.first;
                out.fasta.push_back('\n');
                out.fasta.append(reinterpret_cast<const char*>(aa.data() + k * L), L);
                out.fasta.push_back('\n');
            }
    }
}

}  // namespace mp
