// TEST INFRASTRUCTURE - NOT PRODUCT CODE.
// Command-line front end of the CPU oracle, same surface as `microphaser somatic`
// (reference: src/somatic_cli.yaml, src/main.rs:60-102): GTF on stdin, FASTA on stdout.
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iostream>

#include "../microphaser_amd/csrc/io.hpp"
#include "somatic_oracle.hpp"

using namespace mp;

static void write_file(const std::string& path, const std::string& data) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw Error("cannot write " + path);
    std::fwrite(data.data(), 1, data.size(), f);
    std::fclose(f);
}

int main(int argc, char** argv) {
    try {
        if (argc < 3) {
            std::fprintf(stderr, "usage: oracle_cli somatic <tumor.bam> --variants V --ref R --tsv T --normal-output N [-w 27] [-u] < gtf > fasta\n");
            return 2;
        }
        std::string sub = argv[1];
        if (sub != "somatic") throw Error("oracle_cli: only `somatic` is restated so far");
        std::string bam_path, vcf_path, ref_path, tsv_path = "info.tsv", normal_path = "normal.fasta", stats_path;
        uint64_t window_len = 27;
        bool warn_only = false;
        for (int i = 2; i < argc; i++) {
            std::string a = argv[i];
            auto val = [&]() -> std::string {
                if (i + 1 >= argc) throw Error("missing value for " + a);
                return argv[++i];
            };
            if (a == "--variants" || a == "-b") vcf_path = val();
            else if (a == "--ref" || a == "-r") ref_path = val();
            else if (a == "--tsv" || a == "-t") tsv_path = val();
            else if (a == "--normal-output" || a == "-n") normal_path = val();
            else if (a == "--window-len" || a == "-w") window_len = std::stoull(val());
            else if (a == "--unsupported-allele-warning-only" || a == "-u") warn_only = true;
            else if (a == "--stats") stats_path = val();
            else if (a == "-v" || a == "--verbose") {}
            else if (!a.empty() && a[0] != '-') bam_path = a;
            else throw Error("unknown argument " + a);
        }
        BamData bam;
        load_bam(bam_path, bam);
        VcfData vcf;
        load_vcf(vcf_path, vcf);
        IndexedFasta fasta(ref_path);
        SomaticOutput out;
        double phase_s = 0;
        load_gene_inputs(std::cin, bam, vcf, fasta, warn_only, [&](GeneInput& gi) {
            auto t0 = std::chrono::steady_clock::now();
            mp_oracle::phase_gene(gi, bam.reads, window_len, out);
            phase_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        });
        std::fwrite(out.fasta.data(), 1, out.fasta.size(), stdout);
        write_file(normal_path, out.normal_fasta);
        write_file(tsv_path, out.tsv);
        if (!stats_path.empty()) {
            char buf[256];
            std::snprintf(buf, sizeof buf, "{\"windows\": %llu, \"phase_seconds\": %.6f}\n", (unsigned long long)out.n_windows, phase_s);
            write_file(stats_path, buf);
        }
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
}
