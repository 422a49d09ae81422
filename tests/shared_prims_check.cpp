// Test infrastructure (built and run by tests/test_shared_primitives.py). Two jobs:
//   cigar <bam>...   the product's host cigar_read_pos (model.hpp) against the oracle's own oracle_read_pos (oracle/oracle_util.hpp) on
//                    every read of the given BAMs x every reference position from 3 before its start to 3 behind its end, and on
//                    random CIGARs with every operation (M I D N S H P = X, leading / trailing clips); prints case and mismatch counts
//   variants <vcf>   what the shared ingest (io.cpp variants_from_record, unsupported alleles as warnings) makes of every record of a
//                    VCF, one line per Variant: the test compares it with its own Python restatement of Variant::new
//                    (reference src/common.rs:16-175)
#include <cstdio>
#include <cstring>
#include <random>
#include <string>

#include "io.hpp"
#include "oracle_util.hpp"

using namespace mp;

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    try {
        if (!std::strcmp(argv[1], "cigar")) {
            unsigned long long n = 0, bad = 0, hits = 0;
            for (int a = 2; a < argc; a++) {
                BamData bam;
                load_bam(argv[a], bam);
                const ReadStore& rs = bam.reads;
                for (size_t i = 0; i < rs.size(); i++)
                    for (int64_t p = rs.pos[i] - 3; p <= rs.end_pos[i] + 3; p++) {
                        const int64_t x = cigar_read_pos(rs.cigar(i), rs.n_cigar[i], rs.pos[i], p), y = oracle_read_pos(rs.cigar(i), rs.n_cigar[i], rs.pos[i], p);
                        n++; bad += x != y; hits += x >= 0;
                    }
            }
            std::printf("fixture reads: %llu cases, %llu with a read position, %llu mismatches\n", n, hits, bad);
            std::mt19937_64 rng(11);
            unsigned long long rn = 0, rbad = 0;
            for (int t = 0; t < 200000; t++) {
                uint32_t cig[8];
                const uint32_t nops = 1 + uint32_t(rng() % 7);
                for (uint32_t k = 0; k < nops; k++) {
                    uint32_t op = uint32_t(rng() % 9);
                    if (rng() % 3 == 0) op = 0;                              // mostly matches
                    if ((k == 0 || k + 1 == nops) && rng() % 3 == 0) op = (rng() & 1) ? 4u : 5u;   // clips at the ends
                    cig[k] = (uint32_t(1 + rng() % 9) << 4) | op;
                }
                const int64_t start = 100 + int64_t(rng() % 50);
                for (int64_t p = start - 2; p < start + 70; p++) {
                    const int64_t x = cigar_read_pos(cig, nops, start, p), y = oracle_read_pos(cig, nops, start, p);
                    rn++; rbad += x != y;
                }
            }
            std::printf("random cigars: %llu cases, %llu mismatches\n", rn, rbad);
            return bad || rbad ? 1 : 0;
        }
        if (!std::strcmp(argv[1], "variants")) {
            VcfData vcf;
            load_vcf(argv[2], vcf);
            for (const VcfRecord& r : vcf.records) {
                std::vector<Variant> vs;
                variants_from_record(r, true, vs);
                for (const Variant& v : vs)
                    std::printf("%s\t%llu\t%d\t%d\t%llu\t%d\t%s\t%s\n", r.chrom.c_str(), (unsigned long long)v.pos, int(v.kind), v.kind == VK_SNV ? int(v.alt) : 0,
                                (unsigned long long)v.len, v.is_germline ? 1 : 0, v.seq.c_str(), v.prot_change.c_str());
            }
            return 0;
        }
    } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return 3; }
    return 2;
}
