// In-memory data set (what `microphaser somatic` reads from BAM / VCF / FASTA / GTF) and the
// deterministic synthetic exome generator used by the benchmark.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "io.hpp"

namespace mp {

struct SynthConfig {
    uint64_t seed = 1001;
    uint32_t n_transcripts = 1000;
    double depth = 30.0;
    double var_spacing = 5.4;  // mean nt between SNV sites inside the CDS
    uint32_t read_len = 101;
    // test-only extensions (all 0 in the benchmark workloads): fraction of variant sites that are short indels,
    // fraction of SNV sites with a second ALT allele, fraction of genes with a lower-case (soft-masked) stretch
    double indel_rate = 0.0, multiallelic_rate = 0.0, softmask_rate = 0.0;
    double mate_rate = 0.0;    // fraction of reads followed by a second record with the same name and start
    double isoform_rate = 0.0; // fraction of genes with a second coding transcript (a prefix of the exons)
    // Sharded generation (multi-GPU runs: one exome, every rank materialises only its own genes). With gene_streams every gene
    // draws from its own random stream seeded by (seed, gene ordinal), so a gene's reads / variants / reference bases do not depend
    // on which other genes are generated; `keep` (empty = all, else one flag per transcript) selects the genes to materialise -
    // the others only reserve their coordinates (their stretch of the contig is left 'N', no GTF / VCF / BAM records).
    bool gene_streams = false;
    std::vector<uint8_t> keep;
};

struct Dataset {
    std::vector<std::string> contig_names;
    std::vector<std::string> contig_seq;       // synthetic data sets only
    std::shared_ptr<IndexedFasta> fasta;       // file-backed data sets
    BamData bam;
    VcfData vcf;
    std::string gtf;
    std::vector<GeneInput> genes;              // protein-coding genes in GTF order, loaded as phase_gene would
    std::vector<GeneInput> genes_normal;       // the same for `normal` (three_prime_utr records ignored); built on demand
    bool genes_normal_ready = false;
    bool warn_only = false;
};

void synth_generate(const SynthConfig& cfg, Dataset& ds);
// Work estimate per gene of the data set `cfg` describes, without generating it: sum over exons of (exon length + one read length)
// - proportional to the gene's reads and windows at the configured depth (cost ~ CDS_nt x depth). gene_streams only.
std::vector<uint64_t> synth_gene_costs(const SynthConfig& cfg);
void dataset_load_genes(Dataset& ds, bool unsupported_allele_warning_only);
const std::vector<GeneInput>& dataset_genes(Dataset& ds, bool normal);  // genes as the given mode's phase() builds them
void dataset_load_files(const std::string& bam, const std::string& vcf, const std::string& fasta, std::istream& gtf,
                        bool unsupported_allele_warning_only, Dataset& ds);
void dataset_write_files(const Dataset& ds, const std::string& prefix);  // prefix.{bam,vcf,gtf,fa,fa.fai}

}  // namespace mp
