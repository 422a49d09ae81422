// C ABI, second part: the phase_gene-level boundary (decoded records in, include/microphaser_hip.h mp_gene_batch), gene lists,
// per-gene stream offsets, translation and the peptidome union - what a host that keeps its own readers and its own multi-GPU
// work queue binds (reference seam: src/microphasing.rs:882-893, :1963-1979; src/peptides.rs:128-186).
#include <atomic>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>

#include <hip/hip_runtime.h>

#include <thread>
#include "capi_types.hpp"
#include "kernels_pep.hpp"

namespace mp {
[[noreturn]] void throw_hip(hipError_t e, const char* file, int line);
}
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) mp::throw_hip(e_, __FILE__, __LINE__); } while (0)

namespace {

uint8_t code_of_base(uint8_t c) {   // inverse of "=ACMGRSVTWYHKDBN" (bam::record::Seq::as_bytes)
    static const char* T = "=ACMGRSVTWYHKDBN";
    if (c >= 'a' && c <= 'z') c = uint8_t(c - 32);
    for (int k = 0; k < 16; k++)
        if (uint8_t(T[k]) == c) return uint8_t(k);
    return 15;
}

struct ArraysHolder {          // mp_dataset_to_arrays: the struct handed out + the storage its pointers refer to
    mp_gene_batch b;           // first member: mp_gene_batch_free casts back
    std::vector<std::string> s_gene_id, s_gene_name, s_chrom, s_tx_id, s_prot;
    std::vector<const char*> p_gene_id, p_gene_name, p_chrom, p_tx_id, p_prot;
    std::vector<uint64_t> gene_start, gene_end, ref_off, exon_start, exon_end, exon_frame, read_off, r_cigar_off, r_seq_off, r_qname_hash, var_off,
        v_pos, v_len, v_seq_off;
    std::vector<uint32_t> tx_off, exon_off, cigar;
    std::vector<uint8_t> refseq, tx_strand, r_mapq, seq, qual, v_kind, v_alt, v_is_germline;
    std::vector<int64_t> r_pos;
    std::vector<uint16_t> r_flag;
    std::string v_seq;
};

uint64_t fnv1a(const char* s) {
    uint64_t h = 1469598103934665603ull;
    for (; *s; s++) { h ^= uint8_t(*s); h *= 1099511628211ull; }
    return h;
}

}  // namespace

extern "C" {

int mp_dataset_from_arrays(mp_ctx* ctx, const mp_gene_batch* g, mp_dataset** out) {
    return guarded(ctx, [&] {
        if (!g || !out) throw Error("mp_dataset_from_arrays: null argument");
        std::unique_ptr<mp_dataset> d(new mp_dataset());
        Dataset& ds = d->ds;
        ReadStore& rs = ds.bam.reads;
        std::map<std::string, int32_t> tid_of;
        const uint64_t n_reads = g->n_genes ? g->read_off[g->n_genes] : 0;
        std::vector<uint8_t> seq4;
        char name[24];
        for (uint64_t r = 0; r < n_reads; r++) {   // the records, once, in the order given
            const uint64_t s0 = g->r_seq_off[r], s1 = g->r_seq_off[r + 1], c0 = g->r_cigar_off[r], c1 = g->r_cigar_off[r + 1];
            if (s1 < s0 || c1 < c0 || s1 - s0 > 0xFFFFFFFFull || c1 - c0 > 0xFFFFull) throw Error("mp_dataset_from_arrays: read offsets are not ascending");
            const uint32_t l = uint32_t(s1 - s0);
            seq4.assign((l + 1) / 2, 0);
            for (uint32_t k = 0; k < l; k++) seq4[k >> 1] |= uint8_t(code_of_base(g->seq[s0 + k]) << ((k & 1) ? 0 : 4));
            std::snprintf(name, sizeof name, "%016llx", (unsigned long long)g->r_qname_hash[r]);
            rs.add(0, g->r_pos[r], g->r_mapq[r], g->r_flag[r], g->cigar + c0, uint32_t(c1 - c0), seq4.data(), l, g->qual + s0, name);
        }
        ds.genes.resize(g->n_genes);
        uint64_t rd = 0;
        for (uint32_t k = 0; k < g->n_genes; k++) {
            GeneInput& gi = ds.genes[k];
            gi.gene.id = g->gene_id[k];
            gi.gene.name = g->gene_name[k];
            gi.gene.chrom = g->chrom[k];
            gi.gene.biotype = "protein_coding";   // phase_gene is only called for these (:1964)
            gi.gene.interval.start = g->gene_start[k];
            gi.gene.interval.end = g->gene_end[k];
            if (gi.gene.interval.end < gi.gene.interval.start) throw Error("mp_dataset_from_arrays: gene " + gi.gene.id + " ends before it starts");
            auto it = tid_of.find(gi.gene.chrom);
            if (it == tid_of.end()) {
                it = tid_of.emplace(gi.gene.chrom, int32_t(ds.contig_names.size())).first;
                ds.contig_names.push_back(gi.gene.chrom);
            }
            const uint64_t r0 = g->ref_off[k], r1 = g->ref_off[k + 1];
            if (r1 < r0) throw Error("mp_dataset_from_arrays: refseq offsets are not ascending");
            gi.refseq.assign(g->refseq + r0, g->refseq + r1);
            for (uint32_t t = g->tx_off[k]; t < g->tx_off[k + 1]; t++) {
                Transcript tr;
                tr.id = g->tx_id[t];
                tr.biotype = "protein_coding";
                tr.strand = g->tx_strand[t] ? REVERSE : FORWARD;
                for (uint32_t e = g->exon_off[t]; e < g->exon_off[t + 1]; e++) {
                    Interval iv;
                    iv.start = g->exon_start[e]; iv.end = g->exon_end[e]; iv.frame = g->exon_frame[e];
                    tr.exons.push_back(iv);
                }
                gi.gene.transcripts.push_back(std::move(tr));
            }
            if (g->read_off[k] != rd || g->read_off[k + 1] < rd) throw Error("mp_dataset_from_arrays: read ranges of the genes must be consecutive");
            for (; rd < g->read_off[k + 1]; rd++) { gi.reads.push_back(size_t(rd)); rs.tid[rd] = it->second; }
            for (uint64_t v = g->var_off[k]; v < g->var_off[k + 1]; v++) {
                Variant x;
                if (g->v_kind[v] > 2) throw Error("mp_dataset_from_arrays: unknown variant kind");
                x.kind = VarKind(g->v_kind[v]);
                x.pos = g->v_pos[v];
                x.alt = g->v_alt[v];
                x.len = g->v_len[v];
                x.is_germline = g->v_is_germline[v] != 0;
                x.seq.assign(g->v_seq + g->v_seq_off[v], g->v_seq + g->v_seq_off[v + 1]);
                if (g->v_prot_change && g->v_prot_change[v]) x.prot_change = g->v_prot_change[v];
                gi.variants.push_back(std::move(x));
            }
        }
        ds.bam.ref_names = ds.contig_names;
        // the caller has built the Gene model for the sub-command it runs: both modes see these genes
        ds.genes_normal = ds.genes;
        ds.genes_normal_ready = true;
        *out = d.release();
    });
}

int mp_dataset_to_arrays(mp_ctx* ctx, const mp_dataset* dsp, int mode, mp_gene_batch** out) {
    return guarded(ctx, [&] {
        if (mode != MP_MODE_SOMATIC && mode != MP_MODE_NORMAL) throw Error("unknown mode");
        const std::vector<GeneInput>& genes = dataset_genes(const_cast<Dataset&>(dsp->ds), mode == MP_MODE_NORMAL);
        const ReadStore& rs = dsp->ds.bam.reads;
        std::unique_ptr<ArraysHolder> h(new ArraysHolder());
        h->ref_off.push_back(0); h->tx_off.push_back(0); h->exon_off.push_back(0); h->read_off.push_back(0);
        h->r_cigar_off.push_back(0); h->r_seq_off.push_back(0); h->var_off.push_back(0); h->v_seq_off.push_back(0);
        for (const GeneInput& gi : genes) {
            h->s_gene_id.push_back(gi.gene.id); h->s_gene_name.push_back(gi.gene.name); h->s_chrom.push_back(gi.gene.chrom);
            h->gene_start.push_back(gi.gene.start()); h->gene_end.push_back(gi.gene.end());
            h->refseq.insert(h->refseq.end(), gi.refseq.begin(), gi.refseq.end());
            h->ref_off.push_back(h->refseq.size());
            for (const Transcript& t : gi.gene.transcripts) {
                h->s_tx_id.push_back(t.id);
                h->tx_strand.push_back(t.strand == REVERSE ? 1 : 0);
                for (const Interval& e : t.exons) { h->exon_start.push_back(e.start); h->exon_end.push_back(e.end); h->exon_frame.push_back(e.frame); }
                h->exon_off.push_back(uint32_t(h->exon_start.size()));
            }
            h->tx_off.push_back(uint32_t(h->s_tx_id.size()));
            for (size_t r : gi.reads) {
                h->r_pos.push_back(rs.pos[r]); h->r_mapq.push_back(rs.mapq[r]); h->r_flag.push_back(rs.flag[r]);
                h->cigar.insert(h->cigar.end(), rs.cigar(r), rs.cigar(r) + rs.n_cigar[r]);
                h->r_cigar_off.push_back(h->cigar.size());
                for (uint32_t k = 0; k < rs.l_seq[r]; k++) h->seq.push_back(rs.base(r, k));
                h->qual.insert(h->qual.end(), rs.qual(r), rs.qual(r) + rs.l_seq[r]);
                h->r_seq_off.push_back(h->seq.size());
                h->r_qname_hash.push_back(fnv1a(rs.qname(r)));
            }
            h->read_off.push_back(h->r_pos.size());
            for (const Variant& v : gi.variants) {
                h->v_pos.push_back(v.pos); h->v_kind.push_back(uint8_t(v.kind)); h->v_alt.push_back(v.alt); h->v_len.push_back(v.len);
                h->v_is_germline.push_back(v.is_germline ? 1 : 0);
                h->v_seq += v.seq;
                h->v_seq_off.push_back(h->v_seq.size());
                h->s_prot.push_back(v.prot_change);
            }
            h->var_off.push_back(h->v_pos.size());
        }
        auto ptrs = [](const std::vector<std::string>& s, std::vector<const char*>& p) { p.clear(); for (const std::string& x : s) p.push_back(x.c_str()); };
        ptrs(h->s_gene_id, h->p_gene_id); ptrs(h->s_gene_name, h->p_gene_name); ptrs(h->s_chrom, h->p_chrom); ptrs(h->s_tx_id, h->p_tx_id); ptrs(h->s_prot, h->p_prot);
        mp_gene_batch& b = h->b;
        std::memset(&b, 0, sizeof b);
        b.n_genes = uint32_t(genes.size());
        b.gene_id = h->p_gene_id.data(); b.gene_name = h->p_gene_name.data(); b.chrom = h->p_chrom.data();
        b.gene_start = h->gene_start.data(); b.gene_end = h->gene_end.data(); b.ref_off = h->ref_off.data(); b.refseq = h->refseq.data();
        b.tx_off = h->tx_off.data(); b.tx_id = h->p_tx_id.data(); b.tx_strand = h->tx_strand.data(); b.exon_off = h->exon_off.data();
        b.exon_start = h->exon_start.data(); b.exon_end = h->exon_end.data(); b.exon_frame = h->exon_frame.data();
        b.read_off = h->read_off.data(); b.r_pos = h->r_pos.data(); b.r_mapq = h->r_mapq.data(); b.r_flag = h->r_flag.data();
        b.r_cigar_off = h->r_cigar_off.data(); b.cigar = h->cigar.data(); b.r_seq_off = h->r_seq_off.data(); b.seq = h->seq.data(); b.qual = h->qual.data();
        b.r_qname_hash = h->r_qname_hash.data();
        b.var_off = h->var_off.data(); b.v_pos = h->v_pos.data(); b.v_kind = h->v_kind.data(); b.v_alt = h->v_alt.data(); b.v_len = h->v_len.data();
        b.v_is_germline = h->v_is_germline.data(); b.v_seq_off = h->v_seq_off.data(); b.v_seq = h->v_seq.data(); b.v_prot_change = h->p_prot.data();
        *out = &h.release()->b;
    });
}

void mp_gene_batch_free(mp_gene_batch* b) { delete reinterpret_cast<ArraysHolder*>(b); }

int mp_dataset_gene_costs(mp_ctx* ctx, const mp_dataset* dsp, uint64_t* costs) {
    return guarded(ctx, [&] {
        const ReadStore& rs = dsp->ds.bam.reads;
        size_t k = 0;
        for (const GeneInput& gi : dsp->ds.genes) {
            uint64_t cds = 0, bases = 0;
            for (const Transcript& t : gi.gene.transcripts)
                for (const Interval& e : t.exons) cds += e.end > e.start ? e.end - e.start : 0;
            for (size_t r : gi.reads) bases += rs.l_seq[r];
            const uint64_t span = std::max<uint64_t>(1, gi.gene.end() - gi.gene.start());
            costs[k++] = cds * std::max<uint64_t>(1, bases / span) + 1;   // CDS_nt x depth
        }
    });
}

int mp_batch_create_genes(mp_ctx* ctx, const mp_dataset* ds, int mode, uint64_t window_len, const uint32_t* list, uint32_t n, mp_batch** out) {
    return guarded(ctx, [&] {
        if (mode != MP_MODE_SOMATIC && mode != MP_MODE_NORMAL) throw Error("unknown mode");
        const std::vector<GeneInput>& genes = dataset_genes(const_cast<Dataset&>(ds->ds), mode == MP_MODE_NORMAL);
        std::vector<const GeneInput*> ptrs(n);
        for (uint32_t k = 0; k < n; k++) {
            if (list[k] >= genes.size() || (k && list[k] <= list[k - 1])) throw Error("mp_batch_create_genes: the gene list must be strictly ascending and inside the data set");
            ptrs[k] = &genes[list[k]];
        }
        std::unique_ptr<mp_batch> b(new mp_batch());
        b->reads = &ds->ds.bam.reads;
        build_batch(ptrs.data(), n, *b->reads, window_len, mode == MP_MODE_NORMAL, b->batch);
        if (ctx->dev) {
            ctx->resident = nullptr; ctx->last_run = nullptr;   // (an upload that throws leaves nothing resident)
            ctx->dev->upload(b->batch);
            b->uploaded = true;
            ctx->resident = b.get();
        }
        *out = b.release();
    });
}

const uint64_t* mp_results_gene_offsets(const mp_results* r, int which, size_t* n_plus_1) {
    if (which < 0 || which > 2) { if (n_plus_1) *n_plus_1 = 0; return nullptr; }
    if (n_plus_1) *n_plus_1 = r->out.gene_off[which].size();
    return r->out.gene_off[which].data();
}

int mp_translate(mp_ctx* ctx, const uint8_t* nt, const uint8_t* reverse, uint64_t n, uint32_t L, uint8_t* aa, uint64_t* keys) {
    return guarded(ctx, [&] {
        DeviceContext& dev = need_device(ctx);
        if (L == 0 || L > 12) throw Error("peptide length must be 1..12 for the device peptidome (5-bit residue keys in a u64)");
        if (!n) return;
        HIP_OK(hipSetDevice(dev.device()));
        hipStream_t stream;
        HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        uint8_t *d_nt = nullptr, *d_rev = nullptr, *d_aa = nullptr;
        uint64_t *d_off = nullptr, *d_keys = nullptr;
        uint32_t* d_err = nullptr;
        std::vector<uint64_t> off(n);
        for (uint64_t i = 0; i < n; i++) off[i] = i * 3ull * L;
        std::vector<void*> owned;
        auto release = [&] { for (void* p : owned) (void)hipFree(p); (void)hipStreamDestroy(stream); };
        try {
            HIP_OK(hipMalloc(&d_nt, n * 3ull * L + 64)); owned.push_back(d_nt);
            HIP_OK(hipMalloc(&d_rev, n)); owned.push_back(d_rev);
            HIP_OK(hipMalloc(&d_aa, n * L)); owned.push_back(d_aa);
            HIP_OK(hipMalloc(&d_off, n * 8)); owned.push_back(d_off);
            HIP_OK(hipMalloc(&d_keys, n * 8)); owned.push_back(d_keys);
            HIP_OK(hipMalloc(&d_err, 4)); owned.push_back(d_err);
            HIP_OK(hipMemcpyAsync(d_nt, nt, n * 3ull * L, hipMemcpyHostToDevice, stream));
            HIP_OK(hipMemcpyAsync(d_off, off.data(), n * 8, hipMemcpyHostToDevice, stream));
            HIP_OK(hipMemcpyAsync(d_rev, reverse, n, hipMemcpyHostToDevice, stream));
            HIP_OK(hipMemsetAsync(d_err, 0, 4, stream));
            device_translate(d_nt, d_off, d_rev, n, L, d_aa, d_keys, d_err, stream);
            uint32_t err = 0;
            HIP_OK(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, stream));
            HIP_OK(hipMemcpyAsync(aa, d_aa, n * L, hipMemcpyDeviceToHost, stream));
            if (keys) HIP_OK(hipMemcpyAsync(keys, d_keys, n * 8, hipMemcpyDeviceToHost, stream));
            HIP_OK(hipStreamSynchronize(stream));
            if (err) throw Error("reference would panic: called `Result::unwrap()` on an `Err` value (codon with a base other than A, C, G, T)");
        } catch (...) { release(); throw; }
        release();
    });
}

int mp_peptides_union(mp_ctx* ctx, const uint64_t* const* keys, const uint64_t* counts, uint32_t n_arrays, uint32_t L, mp_peptides** out) {
    return guarded(ctx, [&] {
        if (L == 0 || L > 12) throw Error("peptide length must be 1..12 for the device peptidome (5-bit residue keys in a u64)");
        std::unique_ptr<mp_peptides> p(new mp_peptides());
        p->res.peptide_len = L;
        // the key range is cut into one slice per host thread at pivots taken from the longest array; every thread checks its share of
        // the inputs, merges the arrays' parts of its slice (sorted distinct runs: repeated two-way unions) and the slices - disjoint in
        // key - are joined in order. 70 M keys from eight arrays: 2.6 s sequentially
        uint32_t longest = 0;
        for (uint32_t a = 0; a < n_arrays; a++) if (counts[a] > counts[longest]) longest = a;
        const uint64_t n_long = n_arrays ? counts[longest] : 0;
        const size_t parts = std::max<size_t>(1, std::min<size_t>(host_threads(), size_t(n_long / 65536 + 1)));
        {   // every array is checked WHOLE before anything is derived from it (pivots from an unsorted array are not ascending, and a
            // lower_bound pair on unsorted data can come out reversed): one linear pass, the arrays' chunks dealt to the host threads
            std::atomic<bool> bad{false};
            auto check = [&](size_t t) {
                for (uint32_t a = 0; a < n_arrays; a++) {
                    const uint64_t* k = keys[a];
                    const uint64_t n = counts[a], lo = n * t / parts, hi = n * (t + 1) / parts;
                    for (uint64_t i = std::max<uint64_t>(lo, 1); i < hi; i++)   // (incl. the pair that straddles the chunk's lower end)
                        if (k[i] <= k[i - 1]) { bad = true; return; }
                }
            };
            std::vector<std::thread> cth;
            for (size_t t = 1; t < parts; t++) cth.emplace_back(check, t);
            check(0);
            for (auto& x : cth) x.join();
            if (bad) throw Error("mp_peptides_union: key arrays must be sorted and distinct");
        }
        std::vector<uint64_t> pivot(parts + 1, 0);           // slice t takes the keys in [pivot[t], pivot[t + 1]); the last one is open
        for (size_t t = 1; t < parts; t++) pivot[t] = keys[longest][n_long * t / parts];
        std::vector<std::vector<uint64_t>> slice(parts);
        std::vector<std::string> errors(parts);
        std::vector<std::thread> th;
        auto work = [&](size_t t) {
            try {
                std::vector<uint64_t> acc, tmp;
                for (uint32_t a = 0; a < n_arrays; a++) {
                    const uint64_t* k = keys[a];
                    const uint64_t n = counts[a];
                    const uint64_t* lo = t == 0 ? k : std::lower_bound(k, k + n, pivot[t]);
                    const uint64_t* hi = t + 1 == parts ? k + n : std::lower_bound(k, k + n, pivot[t + 1]);
                    if (lo >= hi) continue;
                    if (acc.empty()) { acc.assign(lo, hi); continue; }
                    tmp.resize(acc.size() + size_t(hi - lo));
                    tmp.resize(size_t(std::set_union(acc.begin(), acc.end(), lo, hi, tmp.begin()) - tmp.begin()));
                    acc.swap(tmp);
                }
                slice[t].swap(acc);
            } catch (const std::exception& e) { errors[t] = e.what(); if (errors[t].empty()) errors[t] = "error"; }
        };
        for (size_t t = 1; t < parts; t++) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
        for (const std::string& e : errors) if (!e.empty()) throw Error(e);
        std::vector<size_t> at(parts + 1, 0);
        for (size_t t = 0; t < parts; t++) at[t + 1] = at[t] + slice[t].size();
        p->res.keys.resize(at[parts]);
        th.clear();
        auto copy = [&](size_t t) { if (!slice[t].empty()) std::memcpy(p->res.keys.data() + at[t], slice[t].data(), slice[t].size() * 8); };
        for (size_t t = 1; t < parts; t++) th.emplace_back(copy, t);
        copy(0);
        for (auto& x : th) x.join();
        p->res.n_peptides = p->res.keys.size();
        *out = p.release();   // (the bincode image is built on demand: mp_peptides_binary)
    });
}

}  // extern "C"
