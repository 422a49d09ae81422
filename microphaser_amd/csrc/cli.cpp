// `microphaser` command line over the C ABI - same surface as the reference binary for the
// accelerated path (reference: src/cli.yaml, src/somatic_cli.yaml:9-50, src/main.rs:34-102):
//   microphaser somatic <tumor.bam> -r/--ref F -b/--variants V [-t/--tsv info.tsv]
//              [-n/--normal-output normal.fasta] [-w/--window-len 27] [-u] [-v]  < GTF  > FASTA
// exit status 1 on error, message on stderr (src/main.rs:260-265).
// Extra (not in the reference): --device N, or --devices a,b,... = genes sharded over several GPUs from this one process
// (one context + host thread per GPU; the genes are dealt to the GPUs by estimated cost - longest processing time first on
// coding nt x depth - and the shards' outputs are merged back into GTF order: byte-identical to one GPU).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <string>
#include <thread>
#include <vector>

#include <chrono>
#include <unistd.h>
#include "../../include/microphaser_hip.h"

static int fail(mp_ctx* ctx, const char* what) {
    std::fprintf(stderr, "%s: %s\n", what, mp_last_error(ctx));
    return 1;
}

// Every write is checked: a full disk or a closed pipe must end in exit status 1, not in a truncated file
// (the reference propagates writer errors to main, src/main.rs:260-265).
static bool write_file(const std::string& path, const char* data, size_t n) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(data, 1, n, f) == n;
    return (std::fclose(f) == 0) && ok;
}
static bool write_stdout(const char* data, size_t n) {
    return std::fwrite(data, 1, n, stdout) == n && std::fflush(stdout) == 0;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: microphaser somatic <tumor.bam> --ref <fasta> --variants <vcf> [--tsv info.tsv] [--normal-output normal.fasta] [--window-len 27] < gtf > fasta\n");
        return 1;
    }
    std::string sub = argv[1];
    if (sub == "build_reference") {
        // microphaser build_reference --reference peptides.fasta -l 9 --output peptides.bin > translated.fasta
        // (reference: src/build_ref_cli.yaml:10-31, src/main.rs:146-169)
        std::string ref, outp;
        unsigned peptide_len = 9;
        int device = 0;
        for (int i = 2; i < argc; i++) {
            std::string a = argv[i];
            auto val = [&]() -> const char* {
                if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(1); }
                return argv[++i];
            };
            if (a == "--reference" || a == "-r") ref = val();
            else if (a == "--output" || a == "-o") outp = val();
            else if (a == "--peptide-length" || a == "-l") peptide_len = unsigned(std::atoi(val()));
            else if (a.rfind("-l", 0) == 0 && a.size() > 2) peptide_len = unsigned(std::atoi(a.c_str() + 2));
            else if (a == "--device") device = std::atoi(val());
            else if (a == "-v" || a == "--verbose") {}
            else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 1; }
        }
        if (ref.empty() || outp.empty()) { std::fprintf(stderr, "--reference and --output are required\n"); return 1; }
        mp_ctx* ctx = nullptr;
        if (mp_create(device, &ctx) != 0) { int rc = fail(ctx, "mp_create"); mp_destroy(ctx); return rc; }
        mp_peptides* pep = nullptr;
        if (mp_build_reference(ctx, ref.c_str(), peptide_len, &pep) != 0) { int rc = fail(ctx, "microphaser"); mp_destroy(ctx); return rc; }
        size_t n = 0;
        const char* p = mp_peptides_fasta(pep, &n);
        if (!write_stdout(p, n)) { std::fprintf(stderr, "cannot write the translated FASTA to stdout\n"); return 1; }
        p = mp_peptides_binary(pep, &n);
        if (!write_file(outp, p, n)) { std::fprintf(stderr, "cannot write %s\n", outp.c_str()); return 1; }
        mp_peptides_free(pep);
        mp_destroy(ctx);
        return 0;
    }
    if (sub == "filter") {
        // microphaser filter -t info.tsv -r reference.binary [-o info.filtered.tsv] [-s info.removed.tsv] [-p peptides.removed.fasta]
        //                    [-n normal.filtered.fa] [-l 9] > tumor.filtered.fa          (src/filter_cli.yaml, src/main.rs:170-214)
        std::string tsv, ref, tsvo = "info.filtered.tsv", simo = "info.removed.tsv", remp = "peptides.removed.fasta", normo = "normal.filtered.fa";
        unsigned peptide_len = 9;
        int device = 0;
        for (int i = 2; i < argc; i++) {
            std::string a = argv[i];
            auto val = [&]() -> const char* {
                if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(1); }
                return argv[++i];
            };
            if (a == "--tsv" || a == "-t") tsv = val();
            else if (a == "--reference" || a == "-r") ref = val();
            else if (a == "--tsv-output" || a == "-o") tsvo = val();
            else if (a == "--similar-removed" || a == "-s") simo = val();
            else if (a == "--removed-peptides" || a == "-p") remp = val();
            else if (a == "--normal-output" || a == "-n") normo = val();
            else if (a == "--peptide-length" || a == "-l") peptide_len = unsigned(std::strtoul(val(), nullptr, 10));
            else if (a == "--device") device = std::atoi(val());
            else if (a == "-v" || a == "--verbose") {}
            else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 1; }
        }
        if (tsv.empty() || ref.empty()) { std::fprintf(stderr, "--tsv and --reference are required\n"); return 1; }
        mp_ctx* ctx = nullptr;
        if (mp_create(device, &ctx) != 0) { int rc = fail(ctx, "mp_create"); mp_destroy(ctx); return rc; }
        mp_filtered* f = nullptr;
        if (mp_filter(ctx, tsv.c_str(), ref.c_str(), peptide_len, &f) != 0) { int rc = fail(ctx, "microphaser"); mp_destroy(ctx); return rc; }
        size_t n = 0;
        const char* p = mp_filtered_fasta(f, &n);
        if (!write_stdout(p, n)) { std::fprintf(stderr, "cannot write the filtered FASTA to stdout\n"); return 1; }
        p = mp_filtered_normal_fasta(f, &n);
        if (!write_file(normo, p, n)) { std::fprintf(stderr, "cannot write %s\n", normo.c_str()); return 1; }
        p = mp_filtered_tsv(f, &n);
        if (!write_file(tsvo, p, n)) { std::fprintf(stderr, "cannot write %s\n", tsvo.c_str()); return 1; }
        p = mp_filtered_removed_tsv(f, &n);
        if (!write_file(simo, p, n)) { std::fprintf(stderr, "cannot write %s\n", simo.c_str()); return 1; }
        p = mp_filtered_removed_fasta(f, &n);
        if (!write_file(remp, p, n)) { std::fprintf(stderr, "cannot write %s\n", remp.c_str()); return 1; }
        mp_filtered_free(f);
        mp_destroy(ctx);
        return 0;
    }
    // microphaser normal <normal.bam> --ref F --variants V [--tsv info.tsv] [-w 27] < gtf > fasta   (src/germline_cli.yaml, src/main.rs:104-143)
    const bool normal_mode = sub == "normal";
    if (sub != "somatic" && !normal_mode) {
        std::fprintf(stderr, "microphaser (MI355X build): sub-command `%s` is not accelerated in this build; only `somatic`, `normal`, `build_reference` and `filter` are available\n", sub.c_str());
        return 1;
    }
    std::string bam, vcf, ref, tsv = "info.tsv", normal = "normal.fasta";
    unsigned long long window_len = 27;
    int warn_only = 0, device = 0;
    std::vector<int> devices;
    for (int i = 2; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(1); }
            return argv[++i];
        };
        if (a == "--variants" || a == "-b") vcf = val();
        else if (a == "--ref" || a == "-r") ref = val();
        else if (a == "--tsv" || a == "-t") tsv = val();
        else if (!normal_mode && (a == "--normal-output" || a == "-n")) normal = val();
        else if (a == "--window-len" || a == "-w") window_len = std::strtoull(val(), nullptr, 10);
        else if (a == "--unsupported-allele-warning-only" || a == "-u") warn_only = 1;
        else if (a == "--device") device = std::atoi(val());
        else if (a == "--devices") {
            devices.clear();
            for (const char* q = val(); *q;) {
                char* end = nullptr;
                devices.push_back(int(std::strtol(q, &end, 10)));
                if (end == q) { std::fprintf(stderr, "bad --devices list\n"); return 1; }
                q = *end == ',' ? end + 1 : end;
            }
        }
        else if (a == "-v" || a == "--verbose") {}
        else if (!a.empty() && a[0] != '-') bam = a;
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 1; }
    }
    if (bam.empty() || vcf.empty() || ref.empty()) { std::fprintf(stderr, "the sample BAM, --variants and --ref are required\n"); return 1; }
    if (devices.size() == 1) device = devices[0];
    mp_ctx* ctx = nullptr;
    if (mp_create(devices.size() > 1 ? -1 : device, &ctx) != 0) { int rc = fail(ctx, "mp_create"); mp_destroy(ctx); return rc; }   // -1: host-only loader
    mp_dataset* ds = nullptr;
    if (mp_dataset_load(ctx, bam.c_str(), vcf.c_str(), ref.c_str(), nullptr, warn_only, &ds) != 0) { int rc = fail(ctx, "microphaser"); mp_destroy(ctx); return rc; }
    const int mode = normal_mode ? MP_MODE_NORMAL : MP_MODE_SOMATIC;
    if (devices.size() > 1) {
        // genes are independent units (src/microphasing.rs:895-942): a cost-weighted deal of the genes (greedy longest
        // processing time first on coding nt x depth), one context + thread per GPU, outputs merged back into GTF order
        const uint32_t ng = mp_dataset_num_genes(ds);
        const size_t nd = devices.size();
        {   // the data set builds its `normal` gene view lazily: do it once here, the shards then only read
            mp_batch* warm = nullptr;
            if (mp_batch_create(ctx, ds, mode, window_len, 0, 0, &warm) != 0) { int rc = fail(ctx, "microphaser"); mp_dataset_free(ds); mp_destroy(ctx); return rc; }
            mp_batch_free(warm);
        }
        std::vector<uint64_t> cost(ng ? ng : 1);
        if (mp_dataset_gene_costs(ctx, ds, cost.data()) != 0) { int rc = fail(ctx, "microphaser"); mp_dataset_free(ds); mp_destroy(ctx); return rc; }
        std::vector<std::vector<uint32_t>> deal(nd);
        {
            std::vector<uint32_t> order(ng);
            for (uint32_t g = 0; g < ng; g++) order[g] = g;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
            typedef std::pair<uint64_t, size_t> Load;   // (load, device slot): the least loaded first, ties by slot
            std::priority_queue<Load, std::vector<Load>, std::greater<Load>> pq;
            for (size_t k = 0; k < nd; k++) pq.push(Load(0, k));
            for (uint32_t g : order) { Load l = pq.top(); pq.pop(); deal[l.second].push_back(g); pq.push(Load(l.first + cost[g], l.second)); }
            for (auto& v : deal) std::sort(v.begin(), v.end());
        }
        struct Shard { mp_ctx* ctx = nullptr; mp_results* res = nullptr; std::string err; };
        std::vector<Shard> shards(nd);
        std::vector<std::thread> th;
        for (size_t k = 0; k < nd; k++)
            th.emplace_back([&, k] {
                Shard& sh = shards[k];
                mp_batch* b = nullptr;
                if (mp_create(devices[k], &sh.ctx) != 0 || mp_batch_create_genes(sh.ctx, ds, mode, window_len, deal[k].data(), uint32_t(deal[k].size()), &b) != 0 ||
                    mp_batch_run(sh.ctx, b, nullptr) != 0 || mp_batch_results(sh.ctx, b, &sh.res) != 0)
                    sh.err = sh.ctx ? mp_last_error(sh.ctx) : "mp_create failed";
                if (b) mp_batch_free(b);
            });
        for (auto& t : th) t.join();
        auto release = [&] {
            for (Shard& sh : shards) { if (sh.res) mp_results_free(sh.res); if (sh.ctx) mp_destroy(sh.ctx); sh.res = nullptr; sh.ctx = nullptr; }
            mp_dataset_free(ds);
            mp_destroy(ctx);
        };
        {   // the error of the first gene (in GTF order) whose shard failed, like a sequential run
            size_t bad = nd;
            for (size_t k = 0; k < nd; k++)
                if (!shards[k].err.empty() && (bad == nd || (!deal[k].empty() && (deal[bad].empty() || deal[k][0] < deal[bad][0])))) bad = k;
            if (bad != nd) { std::fprintf(stderr, "microphaser: %s\n", shards[bad].err.c_str()); release(); return 1; }
        }
        // merge by gene ordinal with the per-gene offsets of every shard; the TSV header line is kept once
        std::vector<std::pair<uint32_t, uint32_t>> where(ng);   // gene -> (shard, index in the shard)
        for (size_t k = 0; k < nd; k++) for (size_t i = 0; i < deal[k].size(); i++) where[deal[k][i]] = {uint32_t(k), uint32_t(i)};
        std::string streams[3];
        for (int which = 0; which < 3; which++) {
            std::vector<const char*> base(nd);
            std::vector<const uint64_t*> off(nd);
            for (size_t k = 0; k < nd; k++) {
                size_t n = 0, no = 0;
                base[k] = which == 0 ? mp_results_fasta(shards[k].res, &n) : which == 1 ? mp_results_normal_fasta(shards[k].res, &n) : mp_results_tsv(shards[k].res, &n);
                off[k] = mp_results_gene_offsets(shards[k].res, which, &no);
                if (which == 2 && n && streams[2].empty() && no) streams[2].assign(base[k], size_t(off[k][0]));   // the header
                if (no != deal[k].size() + 1) off[k] = nullptr;   // (an empty stream)
            }
            for (uint32_t g = 0; g < ng; g++) {
                const auto w = where[g];
                if (off[w.first]) streams[which].append(base[w.first] + off[w.first][w.second], size_t(off[w.first][w.second + 1] - off[w.first][w.second]));
            }
        }
        {   // a TSV that is nothing but its header cannot happen (the header comes with the first record); keep it empty if no shard wrote
            bool any = false;
            for (size_t k = 0; k < nd; k++) { size_t n = 0; mp_results_tsv(shards[k].res, &n); any = any || n; }
            if (!any) streams[2].clear();
        }
        int rc = 0;
        if (!write_stdout(streams[0].data(), streams[0].size())) { std::fprintf(stderr, "cannot write the FASTA records to stdout\n"); rc = 1; }
        if (!rc && !normal_mode && !write_file(normal, streams[1].data(), streams[1].size())) { std::fprintf(stderr, "cannot write %s\n", normal.c_str()); rc = 1; }
        if (!rc && !write_file(tsv, streams[2].data(), streams[2].size())) { std::fprintf(stderr, "cannot write %s\n", tsv.c_str()); rc = 1; }
        release();
        return rc;
    }
    mp_results* res = nullptr;
    mp_batch* batch = nullptr;
    if (mp_batch_create(ctx, ds, mode, window_len, 0, mp_dataset_num_genes(ds), &batch) != 0 || mp_batch_run(ctx, batch, nullptr) != 0 ||
        mp_batch_results(ctx, batch, &res) != 0) {
        int rc = fail(ctx, "microphaser");
        mp_batch_free(batch); mp_dataset_free(ds); mp_destroy(ctx);
        return rc;
    }
    // the TSV is by far the largest stream: it is written by its own thread while this one writes the two FASTA streams
    const auto t_write = std::chrono::steady_clock::now();
    bool tsv_ok = true;
    std::thread tsv_writer([&] {
        size_t tn = 0;
        const char* tp = mp_results_tsv(res, &tn);
        tsv_ok = write_file(tsv, tp, tn);
    });
    size_t n = 0;
    const char* p = mp_results_fasta(res, &n);
    const bool stdout_ok = write_stdout(p, n);
    p = mp_results_normal_fasta(res, &n);
    const bool normal_ok = normal_mode || write_file(normal, p, n);
    tsv_writer.join();
    if (!stdout_ok) { std::fprintf(stderr, "cannot write the FASTA records to stdout\n"); return 1; }
    if (!normal_ok) { std::fprintf(stderr, "cannot write %s\n", normal.c_str()); return 1; }
    if (!tsv_ok) { std::fprintf(stderr, "cannot write %s\n", tsv.c_str()); return 1; }
    const auto t_free = std::chrono::steady_clock::now();
    if (!std::getenv("MP_CLEAN_EXIT")) {
        // everything is written: the process ends here, without unmapping tens of gigabytes page by page first (over a second at
        // whole-exome size); MP_CLEAN_EXIT=1 keeps the orderly teardown for leak checkers
        if (std::getenv("MP_DEBUG")) std::fprintf(stderr, "[mp] write outputs %.1f ms\n", std::chrono::duration<double, std::milli>(t_free - t_write).count());
        std::fflush(nullptr);
        _exit(0);
    }
    mp_batch_free(batch);
    mp_results_free(res);
    mp_dataset_free(ds);
    mp_destroy(ctx);
    if (std::getenv("MP_DEBUG"))
        std::fprintf(stderr, "[mp] write outputs %.1f ms, release %.1f ms\n", std::chrono::duration<double, std::milli>(t_free - t_write).count(),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_free).count());
    return 0;
}
