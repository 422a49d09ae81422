/* microphaser_hip.h - C ABI of the MI355X phasing engine (libmicrophaser_hip.so).
 *
 * The reference (koesterlab/microphaser, Rust) has no plugin / FFI interface; the seam is cut
 * at `phase_gene` (reference: src/microphasing.rs:882-893), called once per protein-coding gene
 * by `phase` (src/microphasing.rs:1943-2131) from `run_somatic` (src/main.rs:60-102).
 * Every entry point below names the reference interface it replaces. Conventions:
 *   - plain C types only; every function returns 0 on success, non-zero on error
 *     (message via mp_last_error); nothing throws or aborts across the boundary
 *     (the reference propagates Box<dyn Error> to main -> exit(1), src/main.rs:260-265);
 *   - the caller owns inputs for the duration of a call, the library owns what it returns
 *     until the matching *_free;
 *   - one mp_ctx per GPU; a ctx is not shared between host threads; no global state;
 *   - there is NO CPU fallback: without a gfx950 device mp_batch_run fails.
 */
#ifndef MICROPHASER_HIP_H
#define MICROPHASER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mp_ctx mp_ctx;
typedef struct mp_dataset mp_dataset;
typedef struct mp_batch mp_batch;
typedef struct mp_results mp_results;

enum {
    MP_MODE_SOMATIC = 0, /* microphasing::phase   (src/microphasing.rs)        : stdout FASTA, normal FASTA, TSV */
    MP_MODE_NORMAL = 1   /* normal_microphasing::phase (src/normal_microphasing.rs): stdout FASTA, TSV           */
};

/* Context. device >= 0: HIP device ordinal. device == -1: host-only context (data sets and
 * planning work, anything that needs the kernels fails loudly). */
int mp_create(int device, mp_ctx** out);
void mp_destroy(mp_ctx* ctx);
const char* mp_last_error(const mp_ctx* ctx);

/* Inputs of `microphaser somatic` (reference: run_somatic opens them, src/main.rs:71-87;
 * phase_gene loads per gene: refseq :895-901, reads :905-920, variants :932-942).
 * gtf_path == NULL reads the GTF from stdin like the reference. */
int mp_dataset_load(mp_ctx* ctx, const char* bam_path, const char* vcf_path, const char* fasta_path, const char* gtf_path,
                    int unsupported_allele_warning_only, mp_dataset** out);
/* Deterministic synthetic exome: the benchmark workload. */
int mp_dataset_synth(mp_ctx* ctx, uint64_t seed, uint32_t n_transcripts, double depth, double var_spacing, mp_dataset** out);
/* Same generator with the test-only knobs (short indels incl. frameshifts, multi-allelic sites, soft-masked
 * reference stretches) that exercise the less common branches of print_haplotypes / phase_gene. */
typedef struct mp_synth_config {
    uint64_t seed;
    uint32_t n_transcripts;
    uint32_t read_len;           /* 0 = 101 */
    double depth, var_spacing;
    double indel_rate, multiallelic_rate, softmask_rate;
    double mate_rate;            /* fraction of reads followed by a same-name record starting at the same position */
    double isoform_rate;         /* fraction of genes with a second coding transcript (a prefix of the exons) */
} mp_synth_config;
int mp_dataset_synth_ex(mp_ctx* ctx, const mp_synth_config* cfg, mp_dataset** out);
/* Write prefix.{bam,vcf,gtf,fa,fa.fai} so that a CLI run sees the same inputs. */
int mp_dataset_write(mp_ctx* ctx, const mp_dataset* ds, const char* prefix);
uint32_t mp_dataset_num_genes(const mp_dataset* ds);   /* protein-coding genes, GTF order */
uint64_t mp_dataset_num_reads(const mp_dataset* ds);
void mp_dataset_free(mp_dataset* ds);

/* Statistics of one pass of the hot path (filled by mp_batch_run). */
typedef struct mp_run_stats {
    double k1_ms, k2_ms, k3_ms, k3b_ms, total_ms;  /* HIP-event times on the launch stream */
    uint64_t n_windows_planned;            /* main-ORF windows in the speculative schedule */
    uint64_t n_steps, n_transcripts, n_reads, n_variants;
    uint64_t n_groups, n_records;
    uint64_t bytes_k1, bytes_k2, bytes_k3, bytes_k3b; /* algorithmic HBM bytes per launch */
    uint64_t hbm_bytes;                    /* device memory held by the batch */
    uint32_t rows_per_lane, mask_words, attempts;
    uint32_t pad_;
    /* k2_ms is three launches: the sequential replay of the segments that need it (k2_window_replay / k2n_window_replay), and
     * the window-parallel replay of everything else (k2a_admission, k2w_window_rows); their times and algorithmic bytes */
    double k2seq_ms, k2a_ms, k2w_ms;
    uint64_t bytes_k2seq, bytes_k2a, bytes_k2w;
    uint64_t n_steps_seq, n_steps_w, n_adm;    /* steps replayed sequentially / window-parallel, (exon, read) admission entries */
    /* the window-parallel part is two kinds of launch: k2l_window_lanes (one LANE per window: windows with <= 8 variant columns,
     * most of them) and k2w_window_rows (one WAVE per window: the rest). k2w_ms / bytes_k2w above cover the wave kernels only. */
    double k2l_ms;
    uint64_t bytes_k2l, n_windows_lane, n_windows_wave;
} mp_run_stats;

/* Plan + pack genes [gene_lo, gene_hi) of a data set and make them resident in HBM
 * (replaces the per-gene loading + the data-independent part of the window scheduler,
 * src/microphasing.rs:905-1342). */
int mp_batch_create(mp_ctx* ctx, const mp_dataset* ds, int mode, uint64_t window_len, uint32_t gene_lo, uint32_t gene_hi,
                    mp_batch** out);
/* One pass of the hot path over the resident batch: read pileup -> haplotype bitsets,
 * sliding-window haplotype counting, window sequences (replaces ObservationMatrix::{cleanup_reads,
 * shrink_left, push_read, extend_right} and the count + sequence phases of print_haplotypes,
 * src/microphasing.rs:220-343, 373-603). Inputs and results stay in HBM. */
int mp_batch_run(mp_ctx* ctx, mp_batch* batch, mp_run_stats* stats);
/* A context keeps ONE batch resident in HBM: creating another batch on the same context evicts the previous one;
 * mp_batch_run brings its batch back if needed, mp_batch_results fails if another batch ran since. */
/* Copy the results back and produce the reference's three output streams (replaces the rest of
 * print_haplotypes + phase_gene: :604-880, :1345-1941, src/common.rs:376-568). */
int mp_batch_results(mp_ctx* ctx, mp_batch* batch, mp_results** out);
void mp_batch_free(mp_batch* batch);

/* Convenience: create + run + results for all genes (what `microphaser somatic` does). */
int mp_phase_dataset(mp_ctx* ctx, const mp_dataset* ds, int mode, uint64_t window_len, mp_results** out);

const char* mp_results_fasta(const mp_results* r, size_t* len);         /* stdout FASTA        */
const char* mp_results_normal_fasta(const mp_results* r, size_t* len);  /* --normal-output     */
const char* mp_results_tsv(const mp_results* r, size_t* len);           /* --tsv               */
uint64_t mp_results_windows(const mp_results* r);  /* main-ORF print_haplotypes calls actually made */
void mp_results_free(mp_results* r);

/* `microphaser build_reference` (reference: peptides::build, src/peptides.rs:148-186 <- run_build, src/main.rs:146-169):
 * translate every 3-nt-step window of every record of a nucleotide FASTA (reverse-complemented when the id does not
 * end in 'F', stop codons -> 'X') and build the set of distinct peptides. Translation and de-duplication run on the
 * GPU. peptide_len <= 12. */
typedef struct mp_peptides mp_peptides;
int mp_build_reference(mp_ctx* ctx, const char* fasta_path, uint32_t peptide_len, mp_peptides** out);
const char* mp_peptides_fasta(const mp_peptides* p, size_t* len);     /* stdout of build_reference            */
const char* mp_peptides_binary(const mp_peptides* p, size_t* len);    /* --output: bincode HashSet<Vec<u8>>   */
const uint64_t* mp_peptides_keys(const mp_peptides* p, size_t* n);    /* sorted distinct keys (5 bits/residue): the unit
                                                                         exchanged in the multi-GPU peptidome union */
uint64_t mp_peptides_count(const mp_peptides* p);                     /* translated windows                   */
void mp_peptides_free(mp_peptides* p);

/* `microphaser filter` (reference: peptides::filter, src/peptides.rs:221-709 <- run_filtering, src/main.rs:170-214,
 * src/filter_cli.yaml): translate the mutant / normal windows of a `somatic` info.tsv, drop self-similar, repeated and
 * post-stop peptides, remove those present in the reference peptidome (bincode HashSet<Vec<u8>> from build_reference)
 * and annotate the rest with the maximum-likelihood frequency and the 95 % credible interval of their variant region.
 * Translation, peptidome membership and the per-region statistics run on the GPU. peptide_len <= 12.
 * mp_filter reads the two files; mp_filter_buffers takes their bytes. */
typedef struct mp_filtered mp_filtered;
int mp_filter(mp_ctx* ctx, const char* tsv_path, const char* reference_binary_path, uint32_t peptide_len, mp_filtered** out);
int mp_filter_buffers(mp_ctx* ctx, const char* tsv, size_t tsv_len, const char* reference_binary, size_t reference_len,
                      uint32_t peptide_len, mp_filtered** out);
const char* mp_filtered_fasta(const mp_filtered* f, size_t* len);          /* stdout: kept tumor peptides           */
const char* mp_filtered_normal_fasta(const mp_filtered* f, size_t* len);   /* --normal-output                       */
const char* mp_filtered_tsv(const mp_filtered* f, size_t* len);            /* --tsv-output (header always present)  */
const char* mp_filtered_removed_tsv(const mp_filtered* f, size_t* len);    /* --similar-removed                     */
const char* mp_filtered_removed_fasta(const mp_filtered* f, size_t* len);  /* --removed-peptides                    */
uint64_t mp_filtered_count(const mp_filtered* f, int which);               /* 0 rows, 1 peptides scored, 2 groups, 3 kept, 4 removed */
void mp_filtered_free(mp_filtered* f);

#ifdef __cplusplus
}
#endif
#endif
