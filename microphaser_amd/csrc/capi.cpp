// C ABI (include/microphaser_hip.h) over the planner / device context / consumer.
#include "../../include/microphaser_hip.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <unistd.h>
#include <cerrno>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>

#include "batch.hpp"
#include "consume.hpp"
#include "device.hpp"
#include "filter.hpp"
#include "pep.hpp"
#include "synth.hpp"

using namespace mp;

#include "capi_types.hpp"


namespace {
// MP_DEBUG=1: wall time of every C-ABI call that does real work (where an end-to-end run spends its time)
struct PhaseTimer {
    const char* what;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit PhaseTimer(const char* w) : what(w) {}
    ~PhaseTimer() {
        if (std::getenv("MP_DEBUG"))
            std::fprintf(stderr, "[mp] %-18s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};
}  // namespace

extern "C" {

int mp_create(int device, mp_ctx** out) {
    PhaseTimer phase_timer("create");
    if (!out) return 1;
    *out = nullptr;
    std::unique_ptr<mp_ctx> c(new mp_ctx());
    reaper_retain();   // (released by mp_destroy: the last context to go joins the library's worker thread)
    int rc = guarded(c.get(), [&] {
        if (device >= 0) c->dev.reset(new DeviceContext(device));
    });
    // hand the context back even on failure so the caller can read the message
    *out = c.release();
    return rc;
}

void mp_destroy(mp_ctx* ctx) {
    if (!ctx) return;
    delete ctx;
    reaper_release();   // scratch handed to release_later is gone, and no thread of the library is left, once the last context is
}

const char* mp_last_error(const mp_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int mp_dataset_load(mp_ctx* ctx, const char* bam, const char* vcf, const char* fasta, const char* gtf, int warn_only, mp_dataset** out) {
    PhaseTimer phase_timer("dataset_load");
    return guarded(ctx, [&] {
        std::unique_ptr<mp_dataset> d(new mp_dataset());
        if (gtf) {
            std::ifstream in(gtf);
            if (!in) throw Error(std::string("cannot open ") + gtf);
            dataset_load_files(bam, vcf, fasta, in, warn_only != 0, d->ds);
        } else {
            // the GTF on stdin, as the reference takes it: read with plain read(2) calls (std::cin synchronised with stdio hands it
            // over a character at a time - 1.5 s for a 47 MB annotation)
            std::string text;
            char buf[1 << 16];
            for (;;) {
                const ssize_t n = ::read(0, buf, sizeof buf);
                if (n < 0) { if (errno == EINTR) continue; throw Error("cannot read the GTF from stdin"); }
                if (n == 0) break;
                text.append(buf, size_t(n));
            }
            std::istringstream in(std::move(text));
            dataset_load_files(bam, vcf, fasta, in, warn_only != 0, d->ds);
        }
        *out = d.release();
    });
}

int mp_dataset_synth(mp_ctx* ctx, uint64_t seed, uint32_t n_transcripts, double depth, double var_spacing, mp_dataset** out) {
    return guarded(ctx, [&] {
        std::unique_ptr<mp_dataset> d(new mp_dataset());
        SynthConfig cfg;
        cfg.seed = seed;
        cfg.n_transcripts = n_transcripts;
        cfg.depth = depth;
        cfg.var_spacing = var_spacing;
        synth_generate(cfg, d->ds);
        *out = d.release();
    });
}

static SynthConfig synth_config_of(const mp_synth_config* c) {
    SynthConfig cfg;
    cfg.seed = c->seed;
    cfg.n_transcripts = c->n_transcripts;
    if (c->read_len) cfg.read_len = c->read_len;
    cfg.depth = c->depth;
    cfg.var_spacing = c->var_spacing;
    cfg.indel_rate = c->indel_rate;
    cfg.multiallelic_rate = c->multiallelic_rate;
    cfg.softmask_rate = c->softmask_rate;
    cfg.mate_rate = c->mate_rate;
    cfg.isoform_rate = c->isoform_rate;
    cfg.gene_streams = c->gene_streams != 0;
    if (c->gene_keep) cfg.keep.assign(c->gene_keep, c->gene_keep + c->n_transcripts);
    return cfg;
}

int mp_dataset_synth_ex(mp_ctx* ctx, const mp_synth_config* c, mp_dataset** out) {
    return guarded(ctx, [&] {
        std::unique_ptr<mp_dataset> d(new mp_dataset());
        synth_generate(synth_config_of(c), d->ds);
        *out = d.release();
    });
}

int mp_synth_gene_costs(mp_ctx* ctx, const mp_synth_config* c, uint64_t* costs) {
    return guarded(ctx, [&] {
        SynthConfig cfg = synth_config_of(c);
        cfg.keep.clear();
        const std::vector<uint64_t> v = synth_gene_costs(cfg);
        std::copy(v.begin(), v.end(), costs);
    });
}

int mp_dataset_write(mp_ctx* ctx, const mp_dataset* ds, const char* prefix) {
    return guarded(ctx, [&] { dataset_write_files(ds->ds, prefix); });
}

uint32_t mp_dataset_num_genes(const mp_dataset* ds) { return ds ? uint32_t(ds->ds.genes.size()) : 0; }
uint64_t mp_dataset_num_reads(const mp_dataset* ds) { return ds ? uint64_t(ds->ds.bam.reads.size()) : 0; }
void mp_dataset_free(mp_dataset* ds) { delete ds; }

int mp_batch_create(mp_ctx* ctx, const mp_dataset* ds, int mode, uint64_t window_len, uint32_t gene_lo, uint32_t gene_hi, mp_batch** out) {
    PhaseTimer phase_timer("batch_create");
    return guarded(ctx, [&] {
        if (mode != MP_MODE_SOMATIC && mode != MP_MODE_NORMAL) throw Error("unknown mode");
        const auto t_genes = std::chrono::steady_clock::now();
        const std::vector<GeneInput>& genes = dataset_genes(const_cast<Dataset&>(ds->ds), mode == MP_MODE_NORMAL);
        if (std::getenv("MP_DEBUG"))
            std::fprintf(stderr, "[mp]   gene inputs %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_genes).count());
        if (gene_hi > genes.size()) gene_hi = uint32_t(genes.size());
        if (gene_lo > gene_hi) gene_lo = gene_hi;
        std::unique_ptr<mp_batch> b(new mp_batch());
        b->reads = &ds->ds.bam.reads;
        build_batch(genes.data() + gene_lo, size_t(gene_hi - gene_lo), *b->reads, window_len, mode == MP_MODE_NORMAL, b->batch);
        if (std::getenv("MP_DEBUG")) {
            uint32_t mx = 0;
            for (const SegDev& g : b->batch.segs) mx = std::max(mx, g.n_steps);
            uint64_t wsteps = 0, simple = 0, nostop = 0;
            for (const ExonW& e : b->batch.exons_w) wsteps += e.n_steps;
            for (const WinStatic& w : b->batch.wins) { simple += (w.flags & WSF_SIMPLE) ? 1 : 0; nostop += (w.flags & WSF_NOSTOP) ? 1 : 0; }
            std::fprintf(stderr, "[mp] windows: %zu, simple %llu, no-stop %llu; record capacity %u nt (records of %u bytes)\n", b->batch.wins.size(),
                         (unsigned long long)simple, (unsigned long long)nostop, b->batch.seq_cap, hap_rec_stride(b->batch.seq_cap));
            std::fprintf(stderr, "[mp] plan: %zu transcripts, %zu steps, %zu windows; sequential replay: %zu segments (longest %u steps); "
                         "window-parallel: %zu exons, %llu steps, %zu work items, %llu admission entries\n", b->batch.tx.size(),
                         b->batch.steps.size(), b->batch.wins.size(), b->batch.segs.size(), mx, b->batch.exons_w.size(),
                         (unsigned long long)wsteps, b->batch.wchunks.size(), (unsigned long long)b->batch.n_adm);
        }
        if (ctx->dev) {
            ctx->resident = nullptr;   // an upload that throws leaves nothing resident (the device context frees what it had allocated)
            ctx->last_run = nullptr;
            ctx->dev->upload(b->batch);
            b->uploaded = true;
            ctx->resident = b.get();
        }
        *out = b.release();
    });
}

int mp_batch_run(mp_ctx* ctx, mp_batch* batch, mp_run_stats* st) {
    PhaseTimer phase_timer("batch_run");
    return guarded(ctx, [&] {
        DeviceContext& dev = need_device(ctx);
        if (!batch->uploaded || ctx->resident != batch) {   // another batch was made resident in between: bring this one back
            ctx->resident = nullptr;
            ctx->last_run = nullptr;
            dev.upload(batch->batch);
            batch->uploaded = true;
            ctx->resident = batch;
        }
        try { dev.run(batch->timing); }
        catch (...) {   // a failed pass may have re-sized or released buffers: nothing counts as resident, the next run uploads again
            ctx->resident = nullptr;
            ctx->last_run = nullptr;
            batch->uploaded = false;
            dev.free_batch();
            throw;
        }
        batch->ran = true;
        ctx->last_run = batch;
        if (st) {
            const Batch& b = batch->batch;
            const RunTiming& t = batch->timing;
            *st = mp_run_stats();
            st->k1_ms = t.k1_ms; st->k2_ms = t.k2_ms; st->k3_ms = t.k3_ms; st->k3b_ms = t.k3b_ms; st->total_ms = t.total_ms;
            st->n_windows_planned = b.n_main_windows;
            st->n_steps = b.steps.size(); st->n_transcripts = b.tx.size();
            st->n_reads = b.r_pos.size(); st->n_variants = b.v_pos.size();
            st->n_groups = t.n_groups; st->n_records = t.n_recs;
            // algorithmic (compulsory) HBM bytes per launch, every byte counted once
            if (batch->sum_wlen == 0 && !b.wins.empty())
                for (const WinStatic& w : b.wins) { batch->sum_wlen += w.wlen; batch->sum_cols += w.ncols; }
            const uint64_t sum_wlen = batch->sum_wlen, sum_cols = batch->sum_cols;
            st->bytes_k1 = b.bytes_k1_in() + b.bytes_k1_out();
            {   // the three parts of K2 (see include/microphaser_hip.h): each byte of its inputs / outputs counted once per launch
                if (!batch->w_wins_known) {   // once per batch: these walk the whole plan
                    for (const ExonW& e : b.exons_w) {
                        batch->w_steps += e.n_steps;
                        for (uint32_t k = 0; k < e.n_steps; k++) batch->w_wins += (b.steps[e.step_off + k].flags & SF_PRINT) ? 1 : 0;
                    }
                    batch->w_wins_known = true;
                }
                const uint64_t w_steps = batch->w_steps;
                const uint64_t seq_steps = b.steps.size() - w_steps, seq_wins = b.wins.size() - batch->w_wins;
                const double wfrac = b.wins.empty() ? 0.0 : double(batch->w_wins) / double(b.wins.size());
                const uint64_t groups_w = uint64_t(double(t.n_groups) * wfrac), groups_seq = t.n_groups - groups_w;
                const uint64_t read_bytes = 20 + 16ull * b.mask_words;   // start, end, first variant, coverage, dup + the two masks
                st->k2seq_ms = t.k2seq_ms; st->k2a_ms = t.k2a_ms; st->k2w_ms = t.k2w_ms; st->k2win_ms = t.k2win_ms;
                st->n_steps_seq = seq_steps; st->n_steps_w = w_steps; st->n_adm = b.n_adm;
                // K2a writes an AdmEntry per (exon, read) and, for the lane-per-window kernel, a RowRec (16 + 8 bytes)
                const uint64_t rowrec = b.lane_on ? sizeof(RowRecA) + 8 : 0;
                if (!batch->adm_known) {   // (K2a writes an exon's AdmEntries / RowRecs only if a wave / lane kernel reads them)
                    for (const ExonW& e : b.exons_w) { if (e.consumers & EW_WAVE) batch->adm_wave += e.n_reads; if (e.consumers & EW_LANE) batch->adm_lane += e.n_reads; }
                    batch->adm_known = true;
                }
                st->bytes_k2a = b.n_adm * read_bytes + batch->adm_wave * sizeof(AdmEntry) + batch->adm_lane * rowrec + b.exons_w.size() * sizeof(ExonW);
                // window rows: the lane kernel reads a WinW + window index per window and every RowRec once; the wave-per-window kernels
                // read the steps of their work items and their share of the read fields / admission entries
                uint64_t wave_steps = 0;
                for (const PodVec<WChunk>* list : {&b.wchunks, &b.wchunks_m, &b.wchunks_d}) for (const WChunk& c : *list) wave_steps += c.n_steps;
                const double wave_share = w_steps ? double(wave_steps) / double(w_steps) : 0.0;
                // groups are shared out by window counts (the lane kernel's windows are the narrow ones: fewer groups each, so this
                // overstates the lane kernel's bytes a little and understates the wave kernels')
                const uint64_t lane_wins = b.winw.size(), wave_wins = batch->w_wins - std::min<uint64_t>(batch->w_wins, lane_wins);
                const uint64_t groups_l = batch->w_wins ? uint64_t(double(groups_w) * double(lane_wins) / double(batch->w_wins)) : 0;
                st->k2l_ms = t.k2l_ms; st->n_windows_lane = lane_wins; st->n_windows_wave = wave_wins;
                // the lane kernel writes a Group per group; window / record index / K3-list entry only for the groups it hands on to K3
                const uint64_t listed_l = t.n_k3 > (t.n_groups - groups_l) ? t.n_k3 - (t.n_groups - groups_l) : 0;
                st->bytes_k2l = lane_wins * (sizeof(WinW) + 4 + sizeof(WinDyn)) + batch->adm_lane * rowrec + groups_l * sizeof(Group) + listed_l * 16;
                st->bytes_k2w = wave_steps * (sizeof(Step) + 7) + uint64_t(double(b.n_adm) * wave_share) * (read_bytes + sizeof(AdmEntry)) +
                                wave_wins * sizeof(WinDyn) + (groups_w - groups_l) * (sizeof(Group) + 16);
                st->bytes_k2seq = seq_steps * sizeof(Step) + (b.steps.empty() ? 0 : uint64_t(double(b.r_pos.size()) * double(seq_steps) / double(b.steps.size()))) * read_bytes +
                                  seq_wins * sizeof(WinDyn) + groups_seq * (sizeof(Group) + 16);
                st->bytes_k2 = st->bytes_k2a + st->bytes_k2l + st->bytes_k2w + st->bytes_k2seq;
            }
            st->n_groups_k3 = t.n_k3;
            st->n_groups_k3a = t.n_k3a;
            st->n_groups_k3c = t.n_k3c;
            st->n_groups_k3d = t.n_k3d;
            st->n_windows_device = b.wins.size();
            st->n_ids = t.n_recs;
            // K3 looks at the listed groups only; their windows' static records / reference bytes / columns are shared out by the listed share
            const double k3_share = t.n_groups ? double(t.n_k3) / double(t.n_groups) : 0.0;
            // item, haplotype word, summary per listed group; a record (written once, complete with its id in somatic mode) per record slot
            // the window kernels reserved; the transcript's id text per hashed id
            st->bytes_k3 = t.n_k3 * (16 + 8 + sizeof(GroupSum)) +
                           uint64_t(k3_share * double(b.wins.size() * sizeof(WinStatic) + sum_wlen + sizeof(WinCol) * sum_cols)) +
                           t.n_rec_slots * hap_rec_stride(b.seq_cap) + (b.normal ? 0 : t.n_recs * 24);
            st->bytes_k3b = b.normal ? t.n_recs * (32 + b.seq_cap + 8) : 0;   // (somatic ids are hashed inside K3: no second pass over the records)
            st->hbm_bytes = dev.hbm_bytes();
            st->rows_per_lane = uint32_t(t.rows_per_lane); st->mask_words = b.mask_words; st->attempts = t.attempts;
        }
    });
}

int mp_batch_results(mp_ctx* ctx, mp_batch* batch, mp_results** out) { return mp_batch_results_select(ctx, batch, MP_STREAM_ALL, out); }
int mp_batch_results_select(mp_ctx* ctx, mp_batch* batch, uint32_t streams, mp_results** out) {
    PhaseTimer phase_timer("batch_results");
    return guarded(ctx, [&] {
        DeviceContext& dev = need_device(ctx);
        if (!batch->ran) throw Error("mp_batch_results before mp_batch_run");
        if (ctx->last_run != batch) throw Error("mp_batch_results: another batch has been created or run on this context since this one ran - run it again");
        HostResults hr;
        const auto t_dl = std::chrono::steady_clock::now();
        dev.download(hr);
        if (std::getenv("MP_DEBUG"))
            std::fprintf(stderr, "[mp]   download %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_dl).count());
        std::unique_ptr<mp_results> r(new mp_results());
        if (batch->batch.normal) consume_batch_normal(batch->batch, hr, r->out, streams);
        else consume_batch(batch->batch, hr, r->out, streams);
        release_later(std::move(hr));
        *out = r.release();
    });
}

void mp_batch_free(mp_batch* batch) { delete batch; }   // (a context never dereferences its resident / last_run pointers)

namespace {
struct DumpHeader {
    char magic[8];   // "MPRES01\0"
    uint64_t n_wins, n_group_slots, n_recs;
    uint32_t seq_cap, rec_stride, group_part_log2, rec_part_log2;
    uint64_t group_prefix[NPART + 1], rec_prefix[NPART + 1];
};
}  // namespace

int mp_batch_results_dump(mp_ctx* ctx, mp_batch* batch, const char* path) {
    return guarded(ctx, [&] {
        DeviceContext& dev = need_device(ctx);
        if (!batch->ran) throw Error("mp_batch_results_dump before mp_batch_run");
        if (ctx->last_run != batch) throw Error("mp_batch_results_dump: another batch has been created or run on this context since this one ran - run it again");
        HostResults hr;
        dev.download(hr);
        DumpHeader h{};
        std::memcpy(h.magic, "MPRES01", 8);
        h.n_wins = hr.win_dyn.size(); h.n_group_slots = hr.n_group_slots; h.n_recs = hr.n_recs;
        h.seq_cap = hr.seq_cap; h.rec_stride = hr.rec_stride; h.group_part_log2 = hr.group_part_log2; h.rec_part_log2 = hr.rec_part_log2;
        std::memcpy(h.group_prefix, hr.group_prefix, sizeof h.group_prefix);
        std::memcpy(h.rec_prefix, hr.rec_prefix, sizeof h.rec_prefix);
        // (K3 stores only the 16-byte pieces of a record that hold sequence bytes; what lies behind them is whatever the arena held:
        //  zeroed here so that a dump is a function of the inputs alone)
        for (uint64_t i = 0; i < hr.n_recs; i++) {
            uint8_t* r = hr.recs.data() + i * hr.rec_stride;
            HapRecHdr hd;
            std::memcpy(&hd, r, sizeof hd);
            // (`normal` records keep the somatic subset of the variant profile in the first 8 bytes of the germline area)
            const uint32_t sl = (uint32_t(hd.seq_len) + 15u) & ~15u, gl = batch->batch.normal ? 16u : (uint32_t(hd.germ_len) + 15u) & ~15u;
            if (sl < hr.seq_cap) std::memset(r + 32 + sl, 0, hr.seq_cap - sl);
            if (gl < hr.seq_cap) std::memset(r + 32 + hr.seq_cap + gl, 0, hr.seq_cap - gl);
        }
        std::ofstream out(path, std::ios::binary);
        if (!out) throw Error(std::string("cannot write ") + path);
        out.write(reinterpret_cast<const char*>(&h), sizeof h);
        out.write(reinterpret_cast<const char*>(hr.win_dyn.data()), std::streamsize(hr.win_dyn.size() * sizeof(WinDyn)));
        out.write(reinterpret_cast<const char*>(hr.groups.data()), std::streamsize(hr.groups.size() * sizeof(Group)));
        out.write(reinterpret_cast<const char*>(hr.gsum.data()), std::streamsize(hr.gsum.size() * sizeof(GroupSum)));
        out.write(reinterpret_cast<const char*>(hr.recs.data()), std::streamsize(hr.recs.size()));
        if (!out) throw Error(std::string("cannot write ") + path);
    });
}

int mp_batch_results_from_dump(mp_ctx* ctx, mp_batch* batch, const char* path, uint32_t streams, mp_results** out) {
    return guarded(ctx, [&] {
        std::ifstream in(path, std::ios::binary);
        if (!in) throw Error(std::string("cannot open ") + path);
        DumpHeader h{};
        in.read(reinterpret_cast<char*>(&h), sizeof h);
        if (!in || std::memcmp(h.magic, "MPRES01", 8) != 0) throw Error(std::string(path) + " is not a results dump");
        if (h.n_wins != batch->batch.wins.size()) throw Error("results dump does not belong to this batch (window counts differ)");
        if (h.seq_cap != batch->batch.seq_cap || h.rec_stride != hap_rec_stride(h.seq_cap)) throw Error("results dump does not belong to this batch (record layout differs)");
        if (h.n_group_slots > (1ull << 34) || h.n_recs > (1ull << 34) || h.group_part_log2 > 40 || h.rec_part_log2 > 40) throw Error("results dump is damaged");
        HostResults hr;
        hr.n_group_slots = h.n_group_slots; hr.n_recs = h.n_recs; hr.seq_cap = h.seq_cap; hr.rec_stride = h.rec_stride;
        hr.group_part_log2 = h.group_part_log2; hr.rec_part_log2 = h.rec_part_log2;
        std::memcpy(hr.group_prefix, h.group_prefix, sizeof h.group_prefix);
        std::memcpy(hr.rec_prefix, h.rec_prefix, sizeof h.rec_prefix);
        if (hr.group_prefix[NPART] != h.n_group_slots || hr.rec_prefix[NPART] != h.n_recs) throw Error("results dump is damaged");
        hr.win_dyn.resize(h.n_wins); hr.groups.resize(h.n_group_slots); hr.gsum.resize(h.n_group_slots); hr.recs.resize(h.n_recs * h.rec_stride);
        in.read(reinterpret_cast<char*>(hr.win_dyn.data()), std::streamsize(hr.win_dyn.size() * sizeof(WinDyn)));
        in.read(reinterpret_cast<char*>(hr.groups.data()), std::streamsize(hr.groups.size() * sizeof(Group)));
        in.read(reinterpret_cast<char*>(hr.gsum.data()), std::streamsize(hr.gsum.size() * sizeof(GroupSum)));
        in.read(reinterpret_cast<char*>(hr.recs.data()), std::streamsize(hr.recs.size()));
        if (!in) throw Error("results dump is truncated");
        std::unique_ptr<mp_results> r(new mp_results());
        if (batch->batch.normal) consume_batch_normal(batch->batch, hr, r->out, streams);
        else consume_batch(batch->batch, hr, r->out, streams);
        *out = r.release();
    });
}

int mp_phase_dataset(mp_ctx* ctx, const mp_dataset* ds, int mode, uint64_t window_len, mp_results** out) {
    mp_batch* b = nullptr;
    int rc = mp_batch_create(ctx, ds, mode, window_len, 0, mp_dataset_num_genes(ds), &b);
    if (rc == 0) rc = mp_batch_run(ctx, b, nullptr);
    if (rc == 0) rc = mp_batch_results(ctx, b, out);
    mp_batch_free(b);
    return rc;
}

const char* mp_results_fasta(const mp_results* r, size_t* len) { if (len) *len = r->out.fasta.size(); return r->out.fasta.data(); }
const char* mp_results_normal_fasta(const mp_results* r, size_t* len) { if (len) *len = r->out.normal_fasta.size(); return r->out.normal_fasta.data(); }
const char* mp_results_tsv(const mp_results* r, size_t* len) { if (len) *len = r->out.tsv.size(); return r->out.tsv.data(); }
uint64_t mp_results_windows(const mp_results* r) { return r->out.n_windows; }
void mp_results_free(mp_results* r) { delete r; }

int mp_build_reference(mp_ctx* ctx, const char* fasta_path, uint32_t peptide_len, mp_peptides** out) {
    return guarded(ctx, [&] {
        DeviceContext& dev = need_device(ctx);
        std::ifstream in(fasta_path);
        if (!in) throw Error(std::string("cannot open ") + fasta_path);
        std::stringstream ss;
        ss << in.rdbuf();
        std::unique_ptr<mp_peptides> p(new mp_peptides());
        build_reference_device(dev.device(), ss.str(), peptide_len, p->res);
        p->bin = p->res.binary();
        *out = p.release();
    });
}
int mp_build_reference_buffer(mp_ctx* ctx, const char* fasta_text, size_t len, uint32_t peptide_len, mp_peptides** out) {
    return guarded(ctx, [&] {
        DeviceContext& dev = need_device(ctx);
        std::unique_ptr<mp_peptides> p(new mp_peptides());
        build_reference_device(dev.device(), std::string_view(fasta_text, len), peptide_len, p->res);
        p->bin = p->res.binary();
        *out = p.release();
    });
}
int mp_peptidome_from_buffer(mp_ctx* ctx, const char* fasta_text, size_t len, uint32_t peptide_len, mp_peptides** out) {
    return guarded(ctx, [&] {
        DeviceContext& dev = need_device(ctx);
        std::unique_ptr<mp_peptides> p(new mp_peptides());
        build_reference_device(dev.device(), std::string_view(fasta_text, len), peptide_len, p->res, false);
        *out = p.release();   // (the bincode image is built on demand: mp_peptides_binary)
    });
}
const char* mp_peptides_fasta(const mp_peptides* p, size_t* len) { if (len) *len = p->res.fasta.size(); return p->res.fasta.data(); }
const char* mp_peptides_binary(const mp_peptides* p, size_t* len) {
    mp_peptides* q = const_cast<mp_peptides*>(p);
    std::call_once(q->bin_once, [q] { if (q->bin.empty()) q->bin = q->res.binary(); });   // (no peptides: u64 0, like serialize_into of an empty HashSet)
    if (len) *len = q->bin.size();
    return q->bin.data();
}
const uint64_t* mp_peptides_keys(const mp_peptides* p, size_t* n) { if (n) *n = p->res.keys.size(); return p->res.keys.data(); }
uint64_t mp_peptides_count(const mp_peptides* p) { return p->res.n_peptides; }
void mp_peptides_free(mp_peptides* p) { delete p; }

int mp_filter_buffers(mp_ctx* ctx, const char* tsv, size_t tsv_len, const char* reference_binary, size_t reference_len, uint32_t peptide_len,
                      mp_filtered** out) {
    return guarded(ctx, [&] {
        DeviceContext& dev = need_device(ctx);
        std::unique_ptr<mp_filtered> f(new mp_filtered());
        filter_device(dev.device(), std::string_view(reference_binary, reference_len), nullptr, std::string_view(tsv, tsv_len), peptide_len, f->res);
        *out = f.release();
    });
}
int mp_filter_peptides(mp_ctx* ctx, const char* tsv, size_t tsv_len, const mp_peptides* reference, mp_filtered** out) {
    return guarded(ctx, [&] {
        DeviceContext& dev = need_device(ctx);
        if (!reference) throw Error("mp_filter_peptides: no peptidome");
        std::unique_ptr<mp_filtered> f(new mp_filtered());
        filter_device(dev.device(), std::string_view(), &reference->res.keys, std::string_view(tsv, tsv_len), reference->res.peptide_len, f->res);
        *out = f.release();
    });
}
int mp_filter(mp_ctx* ctx, const char* tsv_path, const char* reference_binary_path, uint32_t peptide_len, mp_filtered** out) {
    std::string tsv, ref;
    int rc = guarded(ctx, [&] {
        auto slurp = [](const char* path) {
            std::ifstream in(path, std::ios::binary);
            if (!in) throw Error(std::string("cannot open ") + path);
            std::stringstream ss;
            ss << in.rdbuf();
            return ss.str();
        };
        ref = slurp(reference_binary_path);
        tsv = slurp(tsv_path);
    });
    if (rc != 0) return rc;
    return mp_filter_buffers(ctx, tsv.data(), tsv.size(), ref.data(), ref.size(), peptide_len, out);
}
const char* mp_filtered_fasta(const mp_filtered* f, size_t* len) { if (len) *len = f->res.fasta.size(); return f->res.fasta.data(); }
const char* mp_filtered_normal_fasta(const mp_filtered* f, size_t* len) { if (len) *len = f->res.normal_fasta.size(); return f->res.normal_fasta.data(); }
const char* mp_filtered_tsv(const mp_filtered* f, size_t* len) { if (len) *len = f->res.tsv.size(); return f->res.tsv.data(); }
const char* mp_filtered_removed_tsv(const mp_filtered* f, size_t* len) { if (len) *len = f->res.removed_tsv.size(); return f->res.removed_tsv.data(); }
const char* mp_filtered_removed_fasta(const mp_filtered* f, size_t* len) { if (len) *len = f->res.removed_fasta.size(); return f->res.removed_fasta.data(); }
uint64_t mp_filtered_count(const mp_filtered* f, int which) {
    switch (which) {
        case 0: return f->res.n_rows; case 1: return f->res.n_peptides; case 2: return f->res.n_groups; case 3: return f->res.n_kept;
        case 4: return f->res.n_removed; default: return 0;
    }
}
void mp_filtered_free(mp_filtered* f) { delete f; }

}  // extern "C"
