// Deterministic synthetic exome generator: the benchmark workload of
// BASELINE.json configs[1..3] ("synthetic 1k / 20k-transcript exome, 30x, ~5 variant sites per
// window" and the 500x / 20-sites stress case). Everything is generated in memory in the very
// same containers the file readers fill (BamData, VcfData, GTF text, contigs), so the in-memory
// path and the on-disk path (dataset_write_files + dataset_load_files) see identical inputs.
#include "synth.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <thread>

namespace mp {

namespace {

struct Rng {  // splitmix64 seeded xoshiro256**
    uint64_t s[4];
    explicit Rng(uint64_t seed) {
        uint64_t z = seed;
        for (auto& x : s) {
            z += 0x9E3779B97F4A7C15ull;
            uint64_t y = z;
            y = (y ^ (y >> 30)) * 0xBF58476D1CE4E5B9ull;
            y = (y ^ (y >> 27)) * 0x94D049BB133111EBull;
            x = y ^ (y >> 31);
        }
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uni() { return double(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint64_t below(uint64_t n) { return n ? next() % n : 0; }
    double normal() {
        double u1 = uni(), u2 = uni();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
    uint32_t poisson(double lambda) {
        double l = std::exp(-lambda), p = 1.0;
        uint32_t k = 0;
        do { k++; p *= uni(); } while (p > l);
        return k - 1;
    }
};

const char BASES[4] = {'A', 'C', 'G', 'T'};
inline uint8_t code4(char b) { return b == 'A' ? 1 : b == 'C' ? 2 : b == 'G' ? 4 : b == 'T' ? 8 : 15; }
inline char comp(char b) { return b == 'A' ? 'T' : b == 'C' ? 'G' : b == 'G' ? 'C' : b == 'T' ? 'A' : 'N'; }
inline bool is_stop(char a, char b, char c) { return a == 'T' && ((b == 'A' && (c == 'A' || c == 'G')) || (b == 'G' && c == 'A')); }
const char* AA3[20] = {"Ala", "Arg", "Asn", "Asp", "Cys", "Gln", "Glu", "Gly", "His", "Ile",
                       "Leu", "Lys", "Met", "Phe", "Pro", "Ser", "Thr", "Trp", "Tyr", "Val"};

struct SynVar {
    uint64_t pos; char alt; bool somatic; int hap; /* 0 = A, 1 = B, 2 = both */
    int kind = 0;          // 0 SNV, 1 insertion, 2 deletion
    uint32_t len = 0;      // indel length
    std::string ins;       // inserted bases (without the anchor)
    char alt2 = 0;         // second ALT of a multi-allelic SNV site (never carried by a read)
};

}  // namespace

namespace {
uint64_t stream_seed(uint64_t seed, uint64_t stream) {   // splitmix64 of (seed, stream): independent per-gene streams
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (stream + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
}  // namespace

static void synth_generate_impl(const SynthConfig& cfg, Dataset& ds, std::vector<uint64_t>* costs_only);

void synth_generate(const SynthConfig& cfg, Dataset& ds) { synth_generate_impl(cfg, ds, nullptr); }

std::vector<uint64_t> synth_gene_costs(const SynthConfig& cfg) {
    if (!cfg.gene_streams) throw Error("synth_gene_costs needs gene_streams");
    std::vector<uint64_t> costs;
    Dataset scratch;
    synth_generate_impl(cfg, scratch, &costs);
    return costs;
}

static void synth_generate_impl(const SynthConfig& cfg, Dataset& ds, std::vector<uint64_t>* costs_only) {
    ds = Dataset();
    Rng rng(cfg.seed);
    if (!cfg.keep.empty() && (!cfg.gene_streams || cfg.keep.size() != cfg.n_transcripts))
        throw Error("synthetic data set: a gene selection needs gene_streams and one flag per transcript");
    const uint32_t n_contigs = 4;
    const uint32_t L = cfg.read_len;
    ds.contig_names.resize(n_contigs);
    ds.contig_seq.resize(n_contigs);
    for (uint32_t c = 0; c < n_contigs; c++) ds.contig_names[c] = "chrS" + std::to_string(c + 1);
    std::ostringstream gtf;
    ReadStore& rs = ds.bam.reads;
    ds.bam.ref_names = ds.contig_names;
    uint64_t read_serial = 0;
    struct PendingRead { uint64_t pos; uint64_t off; uint8_t mapq; uint64_t serial; std::vector<uint32_t> cigar; };  // off: index into the staging pools
    std::vector<uint8_t> seq4((L + 1) / 2), qual(L);
    std::vector<uint8_t> stage_seq, stage_qual;   // per-gene staging (genes are laid out in ascending coordinates,
    std::vector<PendingRead> creads;              //  so sorting the reads of one gene keeps the BAM coordinate-sorted)
    {
        // rough totals, to avoid re-growing multi-GB pools
        uint64_t n_kept = cfg.n_transcripts;
        if (!cfg.keep.empty()) { n_kept = 0; for (uint8_t k : cfg.keep) n_kept += k ? 1 : 0; }
        double reads_est = costs_only ? 0.0 : double(n_kept) * cfg.depth * 2750.0 / double(L) * 1.15;
        rs.pos.reserve(size_t(reads_est)); rs.end_pos.reserve(size_t(reads_est)); rs.tid.reserve(size_t(reads_est));
        rs.mapq.reserve(size_t(reads_est)); rs.flag.reserve(size_t(reads_est)); rs.l_seq.reserve(size_t(reads_est));
        rs.n_cigar.reserve(size_t(reads_est)); rs.cigar_off.reserve(size_t(reads_est)); rs.seq_off.reserve(size_t(reads_est));
        rs.qual_off.reserve(size_t(reads_est)); rs.qname_off.reserve(size_t(reads_est)); rs.cigar_pool.reserve(size_t(reads_est));
        rs.seq_pool.reserve(size_t(reads_est * ((L + 1) / 2))); rs.qual_pool.reserve(size_t(reads_est * L));
        rs.qname_pool.reserve(size_t(reads_est * 10));
    }
    const uint32_t per_contig = (cfg.n_transcripts + n_contigs - 1) / n_contigs;
    uint32_t tx_serial = 0;
    for (uint32_t c = 0; c < n_contigs; c++) {
        std::string& contig = ds.contig_seq[c];
        contig.assign(2000, 'N');
        if (cfg.gene_streams) rng = Rng(stream_seed(cfg.seed, 0xC0000000ull + c));
        for (char& ch : contig) ch = BASES[rng.below(4)];
        ds.bam.tid_begin.push_back(rs.size());
        for (uint32_t gi = 0; gi < per_contig && tx_serial < cfg.n_transcripts; gi++, tx_serial++) {
            const bool reverse = (tx_serial & 1) != 0;
            if (cfg.gene_streams) { rng = Rng(stream_seed(cfg.seed, tx_serial)); read_serial = uint64_t(tx_serial) << 24; }
            // ---- exon structure
            uint32_t n_exons = std::min<uint32_t>(30, std::max<uint32_t>(1, 1 + rng.poisson(8.0)));
            std::vector<uint64_t> elen(n_exons);
            uint64_t cds = 0;
            for (auto& e : elen) {
                double v = 130.0 * std::exp(0.6 * rng.normal());
                e = uint64_t(std::min(1500.0, std::max(30.0, v)));
                cds += e;
            }
            elen.back() += (3 - cds % 3) % 3;
            cds += (3 - cds % 3) % 3;
            // A 31-nt exon entered with a 2-nt codon remainder on the '-' strand makes the reference itself
            // panic (first window == last window, the variants below it are counted but never added, then
            // shrink_left drains past the end; src/microphasing.rs:1098-1104, 1136-1138, 223): keep the
            // workload inside what the reference can process.
            for (auto& e : elen) if (e == 31) { e = 34; cds += 3; }
            const uint64_t utr = 150, margin = 200;
            // genomic layout (ascending): margin | [3'UTR+stop if reverse] exons/introns [stop+3'UTR if forward] | margin
            const uint64_t gene_start = contig.size();
            uint64_t cur = gene_start + margin;
            std::vector<std::pair<uint64_t, uint64_t>> gexons(n_exons);  // genomic order
            uint64_t utr_lo = 0, utr_hi = 0, stop_lo = 0;
            if (reverse) { utr_lo = cur; utr_hi = cur + utr; stop_lo = utr_hi; cur = stop_lo + 3; }
            for (uint32_t k = 0; k < n_exons; k++) {
                uint64_t len = reverse ? elen[n_exons - 1 - k] : elen[k];
                gexons[k] = {cur, cur + len};
                cur += len;
                if (k + 1 < n_exons) cur += 200 + rng.below(4801);
            }
            if (!reverse) { stop_lo = cur; utr_lo = cur + 3; utr_hi = utr_lo + utr; cur = utr_hi; }
            const uint64_t gene_end = cur + margin;
            if (costs_only) {
                uint64_t cost = 0;
                for (const auto& ex : gexons) cost += (ex.second - ex.first) + L;
                costs_only->push_back(cost);
            }
            if (costs_only || (!cfg.keep.empty() && !cfg.keep[tx_serial])) {   // only the coordinates of this gene are needed
                if (costs_only) contig.resize(gene_end + 2000);                // (sizes only; never read)
                else contig.resize(gene_end + 2000, 'N');
                continue;
            }
            // ---- reference bases
            contig.resize(gene_end + 2000);
            for (uint64_t p = gene_start; p < gene_end + 2000; p++) contig[p] = BASES[rng.below(4)];
            // spliced CDS in transcript orientation: remove in-frame stops, put ATG first, real stop after
            std::vector<uint64_t> cds_pos;  // genomic position of each CDS base in transcript order
            if (!reverse) {
                for (auto& ex : gexons) for (uint64_t p = ex.first; p < ex.second; p++) cds_pos.push_back(p);
            } else {
                for (size_t k = gexons.size(); k-- > 0;) for (uint64_t p = gexons[k].second; p-- > gexons[k].first;) cds_pos.push_back(p);
            }
            auto tbase = [&](size_t i) { return reverse ? comp(contig[cds_pos[i]]) : contig[cds_pos[i]]; };
            auto set_tbase = [&](size_t i, char b) { contig[cds_pos[i]] = reverse ? comp(b) : b; };
            set_tbase(0, 'A'); set_tbase(1, 'T'); set_tbase(2, 'G');
            for (size_t i = 3; i + 2 < cds_pos.size(); i += 3)
                while (is_stop(tbase(i), tbase(i + 1), tbase(i + 2))) set_tbase(i + 1, BASES[rng.below(4)]);
            if (!reverse) { contig[stop_lo] = 'T'; contig[stop_lo + 1] = 'A'; contig[stop_lo + 2] = 'A'; }
            else { contig[stop_lo] = 'T'; contig[stop_lo + 1] = 'T'; contig[stop_lo + 2] = 'A'; }  // revcomp(TAA)
            // ---- GTF (transcription order, Ensembl style)
            char gid[32], tid[32], gname[32];
            std::snprintf(gid, sizeof gid, "SYNG%08u", tx_serial);
            std::snprintf(tid, sizeof tid, "SYNT%08u", tx_serial);
            std::snprintf(gname, sizeof gname, "SG%u", tx_serial);
            const char* chrom = ds.contig_names[c].c_str();
            const char strand = reverse ? '-' : '+';
            std::string attr_g = std::string("gene_id \"") + gid + "\"; gene_name \"" + gname + "\"; gene_biotype \"protein_coding\";";
            std::string attr_t = std::string("gene_id \"") + gid + "\"; transcript_id \"" + tid + "\"; gene_name \"" + gname +
                                 "\"; gene_biotype \"protein_coding\"; transcript_biotype \"protein_coding\";";
            auto line = [&](const char* feat, uint64_t s0, uint64_t e0, const std::string& frame, const std::string& attr) {
                gtf << chrom << "\tsynth\t" << feat << "\t" << (s0 + 1) << "\t" << e0 << "\t.\t" << strand << "\t" << frame << "\t" << attr << "\n";
            };
            line("gene", gene_start, gene_end, ".", attr_g);
            line("transcript", gene_start, gene_end, ".", attr_t);
            uint64_t consumed = 0;
            for (uint32_t k = 0; k < n_exons; k++) {
                const auto& ex = reverse ? gexons[n_exons - 1 - k] : gexons[k];
                line("CDS", ex.first, ex.second, std::to_string((3 - consumed % 3) % 3), attr_t);
                if (k == 0) {
                    if (!reverse) line("start_codon", ex.first, ex.first + 3, "0", attr_t);
                    else line("start_codon", ex.second - 3, ex.second, "0", attr_t);
                }
                consumed += ex.second - ex.first;
            }
            line("three_prime_utr", utr_lo, utr_hi, ".", attr_t);
            // ---- test-only: a second coding transcript of the gene, its first m exons (shares the gene's reads and variants)
            if (cfg.isoform_rate > 0 && n_exons >= 2 && rng.uni() < cfg.isoform_rate) {
                const uint32_t m = 1 + uint32_t(rng.below(n_exons - 1));
                char tid2[32];
                std::snprintf(tid2, sizeof tid2, "SYNU%08u", tx_serial);
                const std::string attr_t2 = std::string("gene_id \"") + gid + "\"; transcript_id \"" + tid2 + "\"; gene_name \"" + gname +
                                            "\"; gene_biotype \"protein_coding\"; transcript_biotype \"protein_coding\";";
                line("transcript", gene_start, gene_end, ".", attr_t2);
                uint64_t used = 0;
                for (uint32_t k = 0; k < m; k++) {
                    const auto& ex = reverse ? gexons[n_exons - 1 - k] : gexons[k];
                    line("CDS", ex.first, ex.second, std::to_string((3 - used % 3) % 3), attr_t2);
                    if (k == 0) {
                        if (!reverse) line("start_codon", ex.first, ex.first + 3, "0", attr_t2);
                        else line("start_codon", ex.second - 3, ex.second, "0", attr_t2);
                    }
                    used += ex.second - ex.first;
                }
            }
            // ---- soft-masked stretch (test-only): case changes only
            if (cfg.softmask_rate > 0 && rng.uni() < cfg.softmask_rate) {
                const auto& ex = gexons[rng.below(gexons.size())];
                uint64_t a = ex.first + rng.below(ex.second - ex.first), e = std::min<uint64_t>(gene_end, a + 40 + rng.below(260));
                for (uint64_t q = a; q < e; q++) contig[q] = char(contig[q] | 0x20);
            }
            // ---- variants: SNV sites inside the CDS (+ optional short indels / second ALT alleles)
            std::vector<SynVar> vars;
            const double p_site = 1.0 / cfg.var_spacing;
            for (auto& ex : gexons) {
                uint64_t blocked_until = 0;  // positions covered by a deletion carry no further variant
                for (uint64_t p = ex.first; p < ex.second; p++) {
                    if (rng.uni() >= p_site) continue;
                    if (p < blocked_until) continue;
                    SynVar v;
                    v.pos = p;
                    const char refb = char(contig[p] & ~0x20);
                    do { v.alt = BASES[rng.below(4)]; } while (v.alt == refb);
                    v.somatic = rng.uni() < 0.2;
                    if (v.somatic) v.hap = int(rng.below(2));
                    else v.hap = rng.uni() < 0.1 ? 2 : int(rng.below(2));
                    if (cfg.indel_rate > 0 && rng.uni() < cfg.indel_rate && p + 8 < ex.second) {
                        static const uint32_t lens[6] = {1, 2, 3, 3, 4, 6};
                        v.len = lens[rng.below(6)];
                        if (rng.uni() < 0.5) {
                            v.kind = 1;
                            for (uint32_t k = 0; k < v.len; k++) v.ins.push_back(BASES[rng.below(4)]);
                        } else {
                            v.kind = 2;
                            blocked_until = p + v.len + 1;
                        }
                    } else if (cfg.multiallelic_rate > 0 && rng.uni() < cfg.multiallelic_rate) {
                        do { v.alt2 = BASES[rng.below(4)]; } while (v.alt2 == refb || v.alt2 == v.alt);
                    }
                    vars.push_back(v);
                }
            }
            for (const SynVar& v : vars) {
                VcfRecord r;
                r.chrom = chrom;
                r.pos = v.pos;
                if (v.kind == 0) {
                    r.ref = std::string(1, contig[v.pos]);
                    r.alts = {std::string(1, v.alt)};
                    if (v.alt2) r.alts.push_back(std::string(1, v.alt2));
                } else if (v.kind == 1) {
                    r.ref = std::string(1, contig[v.pos]);
                    r.alts = {std::string(1, contig[v.pos]) + v.ins};
                } else {
                    r.ref = contig.substr(v.pos, v.len + 1);
                    r.alts = {std::string(1, contig[v.pos])};
                }
                r.somatic = v.somatic;
                uint32_t aa = uint32_t((v.pos * 2654435761ull) >> 7);
                char ann[160];
                std::snprintf(ann, sizeof ann, "%c|missense_variant|MODERATE|%s|%s|transcript|%s|protein_coding|1/1|c.%lluN>%c|p.%s%llu%s|||||",
                              v.alt, gname, gid, tid, (unsigned long long)(v.pos % 100000), v.alt, AA3[aa % 20],
                              (unsigned long long)(v.pos % 1000 + 1), AA3[(aa / 20) % 20]);
                r.ann_first = ann;
                ds.vcf.records.push_back(std::move(r));
            }
            // ---- reads: exome-capture shaped (exon +- flank), 101M, haplotype-resolved
            for (auto& ex : gexons) {
                uint64_t lo = ex.first - 100, hi = ex.second;  // start range
                uint64_t span = hi - lo;
                uint64_t n = uint64_t(std::llround(cfg.depth * double(span) / double(L)));
                for (uint64_t k = 0; k < n; k++) {
                    PendingRead pr;
                    pr.pos = lo + rng.below(span);
                    const int hap = int(rng.below(2));
                    const bool tumor = rng.uni() < 0.6;
                    pr.mapq = rng.uni() < 0.01 ? 0 : 60;
                    pr.serial = read_serial++;
                    // test-only: a mate that starts at the same position (same name, its own errors and qualities) - what `contains`
                    // exists for (src/microphasing.rs:281-294)
                    const bool with_mate = cfg.mate_rate > 0 && rng.uni() < cfg.mate_rate;
                    for (int copy = 0; copy < (with_mate ? 2 : 1); copy++) {
                    std::fill(seq4.begin(), seq4.end(), 0);
                    // walk the reference from pr.pos, applying the variants this read's haplotype carries
                    auto it = std::lower_bound(vars.begin(), vars.end(), pr.pos, [](const SynVar& v, uint64_t p) { return v.pos < p; });
                    pr.cigar.clear();
                    auto add_op = [&](uint32_t op, uint32_t l) {
                        if (!l) return;
                        if (!pr.cigar.empty() && (pr.cigar.back() & 0xF) == op) pr.cigar.back() += l << 4;
                        else pr.cigar.push_back((l << 4) | op);
                    };
                    uint32_t q = 0;
                    uint64_t p = pr.pos;
                    auto put = [&](char bch) {
                        if (rng.uni() < 0.001) { char e; do { e = BASES[rng.below(4)]; } while (e == bch); bch = e; }
                        seq4[q >> 1] |= uint8_t(code4(bch) << ((q & 1) ? 0 : 4));
                        qual[q] = rng.uni() < 0.02 ? 5 : 35;
                        q++;
                    };
                    while (q < L) {
                        char bch = char(contig[p] & ~0x20);
                        while (it != vars.end() && it->pos < p) ++it;
                        const SynVar* v = (it != vars.end() && it->pos == p) ? &*it : nullptr;
                        bool carries = v && (v->somatic ? (v->hap == hap && tumor) : (v->hap == 2 || v->hap == hap));
                        if (carries && v->kind == 0) bch = v->alt;
                        put(bch);
                        add_op(C_M, 1);
                        p++;
                        if (carries && v->kind == 1 && q < L) {
                            uint32_t n = std::min<uint32_t>(v->len, L - q);
                            for (uint32_t k = 0; k < n; k++) put(v->ins[k]);
                            add_op(C_I, n);
                        } else if (carries && v->kind == 2 && q < L) {
                            add_op(C_D, v->len);
                            p += v->len;
                        }
                    }
                    pr.off = creads.size();
                    stage_seq.insert(stage_seq.end(), seq4.begin(), seq4.end());
                    stage_qual.insert(stage_qual.end(), qual.begin(), qual.end());
                    creads.push_back(pr);
                    }
                }
            }
            std::stable_sort(creads.begin(), creads.end(), [](const PendingRead& a, const PendingRead& b2) { return a.pos < b2.pos; });
            for (const PendingRead& pr : creads) {
                char name[32];
                std::snprintf(name, sizeof name, "r%llu", (unsigned long long)pr.serial);
                rs.add(int32_t(c), int64_t(pr.pos), pr.mapq, 0, pr.cigar.data(), uint32_t(pr.cigar.size()), stage_seq.data() + pr.off * seq4.size(), L,
                       stage_qual.data() + pr.off * L, name);
            }
            creads.clear();
            stage_seq.clear();
            stage_qual.clear();
        }
        ds.bam.ref_lens.push_back(int64_t(contig.size()));
    }
    if (costs_only) return;
    ds.bam.tid_begin.push_back(rs.size());
    ds.vcf.contigs = ds.contig_names;
    ds.gtf = gtf.str();
    dataset_load_genes(ds, false);
}

static void load_genes_into(Dataset& ds, bool warn_only, bool use_utr, std::vector<GeneInput>& out) {
    MemFasta mf;
    for (size_t c = 0; c < ds.contig_names.size(); c++) mf.contigs[ds.contig_names[c]] = &ds.contig_seq[c];
    std::istringstream in(ds.gtf);
    out.clear();
    load_gene_inputs(in, ds.bam, ds.vcf, ds.fasta ? static_cast<const RefSource&>(*ds.fasta) : static_cast<const RefSource&>(mf),
                     warn_only, [&](GeneInput& gi) { out.push_back(std::move(gi)); }, use_utr);
}

void dataset_load_genes(Dataset& ds, bool warn_only) {
    ds.warn_only = warn_only;
    load_genes_into(ds, warn_only, true, ds.genes);
    ds.genes_normal.clear();
    ds.genes_normal_ready = false;
}

const std::vector<GeneInput>& dataset_genes(Dataset& ds, bool normal) {
    if (!normal) return ds.genes;
    if (!ds.genes_normal_ready) {
        load_genes_into(ds, ds.warn_only, false, ds.genes_normal);
        ds.genes_normal_ready = true;
    }
    return ds.genes_normal;
}

void dataset_load_files(const std::string& bam, const std::string& vcf, const std::string& fasta, std::istream& gtf,
                        bool warn_only, Dataset& ds) {
    ds = Dataset();
    const bool dbg = std::getenv("MP_DEBUG") != nullptr;
    auto clk = [] { return std::chrono::steady_clock::now(); };
    auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(clk() - t).count(); };
    const auto t_start = clk();
    double ms_bam = 0, ms_vcf = 0;
    // the three inputs are independent: read them concurrently (the BAM inflate and the VCF text parse dominate)
    std::string err_bam, err_vcf;
    std::thread t_bam([&] { try { const auto t0 = clk(); load_bam(bam, ds.bam); ms_bam = since(t0); } catch (const std::exception& e) { err_bam = e.what(); if (err_bam.empty()) err_bam = "error"; } });
    std::thread t_vcf([&] { try { const auto t0 = clk(); load_vcf(vcf, ds.vcf); ds.vcf.build_index(); ms_vcf = since(t0); } catch (const std::exception& e) { err_vcf = e.what(); if (err_vcf.empty()) err_vcf = "error"; } });
    std::string err_fa;
    try {
        auto fa = std::make_shared<IndexedFasta>(fasta);
        fa->preload();   // the per-gene fetches that follow read from memory: bring the file in while the BAM / VCF threads work
        ds.fasta = fa;
        std::ostringstream ss;
        ss << gtf.rdbuf();
        ds.gtf = ss.str();
    } catch (const std::exception& e) { err_fa = e.what(); if (err_fa.empty()) err_fa = "error"; }
    t_bam.join();
    t_vcf.join();
    // report in the order the reference opens its readers (src/main.rs:73-77): BAM, BCF, FASTA
    if (!err_bam.empty()) throw Error(err_bam);
    if (!err_vcf.empty()) throw Error(err_vcf);
    if (!err_fa.empty()) throw Error(err_fa);
    const double ms_files = since(t_start);
    const auto t_genes = clk();
    dataset_load_genes(ds, warn_only);
    if (dbg) std::fprintf(stderr, "[mp] load: bam %.0f ms | vcf %.0f ms (concurrent, %.0f ms wall), per-gene fetch %.0f ms\n", ms_bam, ms_vcf, ms_files, since(t_genes));
}

void dataset_write_files(const Dataset& ds, const std::string& prefix) {
    write_bam(prefix + ".bam", ds.bam.ref_names, ds.bam.ref_lens, ds.bam.reads);
    {
        std::ofstream v(prefix + ".vcf");
        v << "##fileformat=VCFv4.2\n";
        for (size_t c = 0; c < ds.contig_names.size(); c++) v << "##contig=<ID=" << ds.contig_names[c] << ",length=" << ds.contig_seq[c].size() << ">\n";
        v << "##INFO=<ID=SOMATIC,Number=0,Type=Flag,Description=\"somatic\">\n##INFO=<ID=ANN,Number=.,Type=String,Description=\"ann\">\n";
        v << "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n";
        for (const VcfRecord& r : ds.vcf.records) {
            v << r.chrom << "\t" << (r.pos + 1) << "\t.\t" << r.ref << "\t";
            for (size_t a = 0; a < r.alts.size(); a++) v << (a ? "," : "") << r.alts[a];
            v << "\t.\t.\t";
            if (r.somatic) v << "SOMATIC;";
            v << "ANN=" << r.ann_first << "\n";
        }
    }
    { std::ofstream g(prefix + ".gtf"); g << ds.gtf; }
    {
        std::ofstream f(prefix + ".fa"), fai(prefix + ".fa.fai");
        uint64_t off = 0;
        for (size_t c = 0; c < ds.contig_names.size(); c++) {
            std::string hdr = ">" + ds.contig_names[c] + "\n";
            f << hdr;
            off += hdr.size();
            const std::string& s = ds.contig_seq[c];
            fai << ds.contig_names[c] << "\t" << s.size() << "\t" << off << "\t60\t61\n";
            for (size_t p = 0; p < s.size(); p += 60) {
                size_t n = std::min<size_t>(60, s.size() - p);
                f.write(s.data() + p, long(n));
                f << "\n";
                off += n + 1;
            }
        }
    }
}

}  // namespace mp
