// `microphaser` command line over the C ABI - same surface as the reference binary for the
// accelerated path (reference: src/cli.yaml, src/somatic_cli.yaml:9-50, src/main.rs:34-102):
//   microphaser somatic <tumor.bam> -r/--ref F -b/--variants V [-t/--tsv info.tsv]
//              [-n/--normal-output normal.fasta] [-w/--window-len 27] [-u] [-v]  < GTF  > FASTA
// exit status 1 on error, message on stderr (src/main.rs:260-265).
// Extra (not in the reference): --device N, or --devices a,b,... = genes sharded over several GPUs from this one process
// (one context + host thread per GPU, contiguous gene ranges, outputs concatenated in gene order: byte-identical to one GPU).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <chrono>
#include "../../include/microphaser_hip.h"

static int fail(mp_ctx* ctx, const char* what) {
    std::fprintf(stderr, "%s: %s\n", what, mp_last_error(ctx));
    return 1;
}

static bool write_file(const std::string& path, const char* data, size_t n) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fwrite(data, 1, n, f);
    std::fclose(f);
    return true;
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
        std::fwrite(p, 1, n, stdout);
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
        std::fwrite(p, 1, n, stdout);
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
        // genes are independent units (src/microphasing.rs:895-942): contiguous gene ranges, one context + thread per GPU
        const uint32_t ng = mp_dataset_num_genes(ds);
        const size_t nd = devices.size();
        {   // the data set builds its `normal` gene view lazily: do it once here, the shards then only read
            mp_batch* warm = nullptr;
            if (mp_batch_create(ctx, ds, mode, window_len, 0, 0, &warm) != 0) { int rc = fail(ctx, "microphaser"); mp_dataset_free(ds); mp_destroy(ctx); return rc; }
            mp_batch_free(warm);
        }
        struct Shard { mp_ctx* ctx = nullptr; mp_results* res = nullptr; std::string err; };
        std::vector<Shard> shards(nd);
        std::vector<std::thread> th;
        for (size_t k = 0; k < nd; k++)
            th.emplace_back([&, k] {
                Shard& sh = shards[k];
                const uint32_t lo = uint32_t(uint64_t(ng) * k / nd), hi = uint32_t(uint64_t(ng) * (k + 1) / nd);
                mp_batch* b = nullptr;
                if (mp_create(devices[k], &sh.ctx) != 0 || mp_batch_create(sh.ctx, ds, mode, window_len, lo, hi, &b) != 0 ||
                    mp_batch_run(sh.ctx, b, nullptr) != 0 || mp_batch_results(sh.ctx, b, &sh.res) != 0)
                    sh.err = sh.ctx ? mp_last_error(sh.ctx) : "mp_create failed";
                if (b) mp_batch_free(b);
            });
        for (auto& t : th) t.join();
        for (size_t k = 0; k < nd; k++)   // the first failing gene range in gene order, like a sequential run
            if (!shards[k].err.empty()) { std::fprintf(stderr, "microphaser: %s\n", shards[k].err.c_str()); return 1; }
        std::string fa, nfa, tsvs;
        for (size_t k = 0; k < nd; k++) {
            size_t n = 0;
            const char* p = mp_results_fasta(shards[k].res, &n); fa.append(p, n);
            p = mp_results_normal_fasta(shards[k].res, &n); nfa.append(p, n);
            p = mp_results_tsv(shards[k].res, &n);
            if (n) {   // the header is the first line of every non-empty shard: keep the first one only
                if (tsvs.empty()) tsvs.append(p, n);
                else { const char* nl = static_cast<const char*>(std::memchr(p, '\n', n)); if (nl) tsvs.append(nl + 1, size_t(p + n - (nl + 1))); }
            }
            mp_results_free(shards[k].res);
            mp_destroy(shards[k].ctx);
        }
        std::fwrite(fa.data(), 1, fa.size(), stdout);
        if (!normal_mode && !write_file(normal, nfa.data(), nfa.size())) { std::fprintf(stderr, "cannot write %s\n", normal.c_str()); return 1; }
        if (!write_file(tsv, tsvs.data(), tsvs.size())) { std::fprintf(stderr, "cannot write %s\n", tsv.c_str()); return 1; }
        mp_dataset_free(ds);
        mp_destroy(ctx);
        return 0;
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
    std::fwrite(p, 1, n, stdout);
    std::fflush(stdout);
    p = mp_results_normal_fasta(res, &n);
    const bool normal_ok = normal_mode || write_file(normal, p, n);
    tsv_writer.join();
    if (!normal_ok) { std::fprintf(stderr, "cannot write %s\n", normal.c_str()); return 1; }
    if (!tsv_ok) { std::fprintf(stderr, "cannot write %s\n", tsv.c_str()); return 1; }
    const auto t_free = std::chrono::steady_clock::now();
    mp_batch_free(batch);
    mp_results_free(res);
    mp_dataset_free(ds);
    mp_destroy(ctx);
    if (std::getenv("MP_DEBUG"))
        std::fprintf(stderr, "[mp] write outputs %.1f ms, release %.1f ms\n", std::chrono::duration<double, std::milli>(t_free - t_write).count(),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_free).count());
    return 0;
}
