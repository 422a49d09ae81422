// Test infrastructure (built and run by tests/test_host_formatting.py): the product's in-place row writers (rowfmt.hpp) and host SHA-1 ids
// (hostsha.hpp) against the oracle's own plain versions (oracle/oracle_util.hpp).
#include <chrono>
#include <cstdio>
#include <random>

#include "hostsha.hpp"
#include "rowfmt.hpp"
#include "oracle_util.hpp"

int main() {
    std::mt19937_64 rng(7);
    size_t n = 0, bad = 0;
    auto check = [&](double v) {
        std::string a = mp::fmt_f64(v), b;
        mp::append_f64(b, v);
        n++;
        if (a != b) { if (bad < 10) std::printf("f64 MISMATCH %.17g: %s vs %s\n", v, a.c_str(), b.c_str()); bad++; }
    };
    for (int d = 1; d <= 600; d++) for (int c = 0; c <= d; c++) check(double(c) / double(d));
    for (int i = 0; i < 1000000; i++) { uint64_t x = rng(); double v; std::memcpy(&v, &x, 8); check(v); }
    for (int i = 0; i < 500000; i++) { double a = double(rng() % 1000) / double(rng() % 1000 + 1), b = double(rng() % 1000 + 1) / 1000.0; check(a * b); check(1.0 - a * b); }
    for (double v : {0.0, 1.0, 1e15, 1e16, 1e17, 123456789012345678.0, 1e-4, 1e-5, 1e-6, 0.00001234, 5e-324, 1.7976931348623157e308, 1e21, 1e22, 100.0, 0.1, 0.5}) check(v), check(-v);
    std::printf("f64: %zu values, %zu mismatches\n", n, bad);

    // csv quoting + a whole row
    {
        mp::SomaticOutput so;
        mp::SomaticText st;
        const char* texts[] = {"", "plain", "with\ttab", "quo\"te", "line\nbreak", "cr\rx", "\"", "a|b|c"};
        for (int i = 0; i < 2000; i++) {
            mp::IDRecord r;
            std::string* f[] = {&r.id, &r.transcript, &r.gene_id, &r.gene_name, &r.chrom, &r.strand, &r.variant_sites, &r.somatic_positions, &r.somatic_aa_change,
                                &r.germline_positions, &r.germline_aa_change, &r.normal_sequence, &r.mutant_sequence};
            for (std::string* s : f) *s = texts[rng() % 8];
            r.offset = rng(); r.frame = rng() % 3; r.freq = double(rng() % 97) / 96.0; r.depth = uint32_t(rng()); r.nvar = uint32_t(rng() % 9);
            mp::write_tsv_record(so, r);
            mp::put_tsv_row(st, r);
            mp::write_fasta(so.fasta, r.id, reinterpret_cast<const uint8_t*>(r.mutant_sequence.data()), r.mutant_sequence.size());
            mp::put_fasta(st.fasta, r.id, reinterpret_cast<const uint8_t*>(r.mutant_sequence.data()), r.mutant_sequence.size());
        }
        const bool same = so.tsv == std::string(st.tsv.data(), st.tsv.size()) && so.fasta == std::string(st.fasta.data(), st.fasta.size());
        std::printf("rows: %s (%zu bytes)\n", same ? "identical" : "DIFFERENT", so.tsv.size());
        if (!same) bad++;
    }
    {
        mp::NormalOutput no;
        mp::NormalText nt;
        for (int i = 0; i < 500; i++) {
            mp::NormalRecord r;
            r.id = "0123456789abcdeF"; r.transcript = "T\"1"; r.peptide_sequence = i % 2 ? "ACGT" : "AC\tGT"; r.freq = 1.0 / double(i + 1); r.offset = uint64_t(i);
            mp::write_normal_tsv_record(no, r);
            mp::put_normal_tsv_row(nt, r);
        }
        if (no.tsv != std::string(nt.tsv.data(), nt.tsv.size())) { std::printf("normal rows DIFFERENT\n"); bad++; }
    }

    // ids: SHA extensions (where the CPU has them) and the portable block function against oracle_util.hpp
    std::printf("cpu has sha: %d\n", int(mp::hostsha::cpu_has_sha()));
    size_t ids = 0;
    for (int it = 0; it < 60000; it++) {
        size_t len = it < 1000 ? size_t(it % 300) : size_t(rng() % 64);
        std::vector<uint8_t> seq(len);
        for (auto& c : seq) c = it % 3 ? uint8_t("ACGTacgtN"[rng() % 9]) : uint8_t(rng());
        std::string tid = "ENST" + std::to_string(rng() % 100000000);
        if (it % 7 == 0) tid.assign(rng() % 200, 'x');
        const uint64_t off = rng() % 300000000ull;
        std::string a = mp::haplotype_id(seq.data(), len, tid, off, it & 1 ? 'F' : 'R'), b;
        mp::haplotype_id_into(b, seq.data(), len, tid, off, it & 1 ? 'F' : 'R');
        ids++;
        if (a != b) { if (bad < 15) std::printf("id MISMATCH %s %s\n", a.c_str(), b.c_str()); bad++; }
        uint8_t blk[128];
        for (auto& c : blk) c = uint8_t(rng());
        uint32_t h0[5] = {1, 2, 3, 4, 5}, h1[5] = {1, 2, 3, 4, 5};
        mp::Sha1 ref;
        std::memcpy(ref.h, h0, 20);
        ref.block(blk); ref.block(blk + 64);
        mp::hostsha::blocks_portable(h1, blk, 2);
        if (std::memcmp(ref.h, h1, 20)) bad++;
        if (mp::hostsha::cpu_has_sha()) {
            uint32_t h2[5] = {1, 2, 3, 4, 5};
            mp::hostsha::blocks_shani(h2, blk, 2);
            if (std::memcmp(ref.h, h2, 20)) bad++;
        }
    }
    std::printf("ids: %zu, total mismatches %zu\n", ids, bad);
    return bad != 0;
}
