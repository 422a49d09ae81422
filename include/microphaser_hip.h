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
    /* Sharded generation (one exome over several GPUs, every rank materialises only its genes): with gene_streams != 0 every
     * gene draws from its own random stream (seed, gene ordinal); gene_keep (NULL = all) holds one flag per transcript. */
    uint32_t gene_streams, pad_;
    const uint8_t* gene_keep;
} mp_synth_config;
int mp_dataset_synth_ex(mp_ctx* ctx, const mp_synth_config* cfg, mp_dataset** out);
/* Work estimate of every gene of the data set cfg describes (gene_streams != 0), without generating it: sum over its exons of
 * (exon length + read length), i.e. proportional to CDS_nt x depth. costs holds cfg->n_transcripts entries. */
int mp_synth_gene_costs(mp_ctx* ctx, const mp_synth_config* cfg, uint64_t* costs);
/* Write prefix.{bam,vcf,gtf,fa,fa.fai} so that a CLI run sees the same inputs. */
int mp_dataset_write(mp_ctx* ctx, const mp_dataset* ds, const char* prefix);
/* ---- The phase_gene seam itself (reference: src/microphasing.rs:882-893, called per protein-coding gene from :1963-1979;
 * normal twin src/normal_microphasing.rs:650-660): the host keeps its own readers (rust-htslib RecordBuffers, bio FASTA / GFF)
 * and hands over, for a batch of genes, exactly what phase_gene works on - the Gene model as `phase` has built it, the bytes of
 * refseq, the records read_buffer.iter() yields after fetch (BEFORE the mapq filter, :909-920) and the Variants of variant_tree
 * (after Variant::new and the same-position overwrite, :932-942). Struct-of-arrays with CSR offsets; the caller owns everything
 * for the duration of the call (the data set copies what it keeps). Positions are 0-based, intervals half-open. */
typedef struct mp_gene_batch {
    uint32_t n_genes;
    uint32_t pad_;
    /* genes (src/common.rs:224-253) */
    const char* const* gene_id;      /* [n_genes] */
    const char* const* gene_name;
    const char* const* chrom;
    const uint64_t* gene_start;      /* gene.start() / gene.end() */
    const uint64_t* gene_end;
    const uint64_t* ref_off;         /* [n_genes + 1] into refseq: gene g owns bytes of [gene.start, gene.end + 100), case preserved (:895-901) */
    const uint8_t* refseq;
    /* transcripts of gene g: [tx_off[g], tx_off[g + 1])  (src/common.rs:271-291; all of them, coding or not) */
    const uint32_t* tx_off;          /* [n_genes + 1] */
    const char* const* tx_id;        /* [n_tx] */
    const uint8_t* tx_strand;        /* 0 forward, 1 reverse */
    const uint32_t* exon_off;        /* [n_tx + 1]; exons in the order Transcript.exons holds them */
    const uint64_t* exon_start;      /* Interval { start, end, frame }  (src/common.rs:293-338) */
    const uint64_t* exon_end;
    const uint64_t* exon_frame;
    /* reads of gene g: [read_off[g], read_off[g + 1]) of the read arrays, in the order the buffer yields them */
    const uint64_t* read_off;        /* [n_genes + 1] */
    const int64_t* r_pos;            /* [n_reads] bam::Record::pos() */
    const uint8_t* r_mapq;
    const uint16_t* r_flag;
    const uint64_t* r_cigar_off;     /* [n_reads + 1] into cigar: BAM-encoded ops (len << 4 | op, op in MIDNSHP=X) */
    const uint32_t* cigar;
    const uint64_t* r_seq_off;       /* [n_reads + 1] in BASES: read r has l_seq = r_seq_off[r + 1] - r_seq_off[r] */
    const uint8_t* seq;              /* one byte per base: what bam::Record::seq().as_bytes() returns (upper-case "=ACMGRSVTWYHKDBN") */
    const uint8_t* qual;             /* one byte per base, same offsets (bam::Record::qual()) */
    const uint64_t* r_qname_hash;    /* any hash of qname(): `contains` (:281-294) only compares names for equality */
    /* variants of gene g: [var_off[g], var_off[g + 1]), ascending position, ALT order within a position (src/common.rs:38-59) */
    const uint64_t* var_off;         /* [n_genes + 1] */
    const uint64_t* v_pos;
    const uint8_t* v_kind;           /* 0 SNV, 1 Insertion, 2 Deletion */
    const uint8_t* v_alt;            /* SNV: the alt base as written in the VCF */
    const uint64_t* v_len;           /* indel length (Variant::Insertion/Deletion len) */
    const uint8_t* v_is_germline;
    const uint64_t* v_seq_off;       /* [n_vars + 1] into v_seq: insertion = whole ALT incl. the anchor base; empty otherwise */
    const char* v_seq;
    const char* const* v_prot_change;/* [n_vars] (may be "") */
} mp_gene_batch;
int mp_dataset_from_arrays(mp_ctx* ctx, const mp_gene_batch* genes, mp_dataset** out);
/* The same view of a loaded / synthetic data set (what a host would have handed over); owned by the library until freed.
 * mode selects the gene model (`normal` ignores three_prime_utr records, src/normal_microphasing.rs:1319-1433). */
int mp_dataset_to_arrays(mp_ctx* ctx, const mp_dataset* ds, int mode, mp_gene_batch** out);
void mp_gene_batch_free(mp_gene_batch* b);

uint32_t mp_dataset_num_genes(const mp_dataset* ds);   /* protein-coding genes, GTF order */
/* Work estimate per gene for sharding genes over GPUs (cost ~ sum of CDS_nt x depth): coding nucleotides x mean read
 * depth of the gene's records. costs holds mp_dataset_num_genes entries. */
int mp_dataset_gene_costs(mp_ctx* ctx, const mp_dataset* ds, uint64_t* costs);
uint64_t mp_dataset_num_reads(const mp_dataset* ds);
void mp_dataset_free(mp_dataset* ds);

/* Statistics of one pass of the hot path (filled by mp_batch_run). */
typedef struct mp_run_stats {
    double k1_ms, k2_ms, k3_ms, k3b_ms, total_ms;  /* HIP-event times on the launch stream (k3b_ms: `normal` mode only - somatic ids are hashed inside K3) */
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
    uint64_t n_groups_k3;                  /* groups k3_window_seq looked at (k2l settles a group of a simple window without a somatic column itself) */
    /* The launches of the window phase run side by side on several streams (sequential replay beside k2a; then k2l <= 6 columns, k2l 7-8
     * columns and the wave-per-window kernels together): k2seq_ms, k2l_ms and k2w_ms are overlapping intervals; k2win_ms is the wall
     * time from the end of k2a until all of them have finished. */
    double k2win_ms;
    uint64_t n_windows_device;             /* printing windows the device computed (main ORF and shifted frames, speculative past a stop) */
    uint64_t n_groups_k3a;                 /* of n_groups_k3: groups in k3_window_seq's list A (their ids are hashed) */
    uint64_t n_ids;                        /* haplotype ids hashed (somatic: inside k3_window_seq; normal: k3b_haplotype_ids) */
    uint64_t n_groups_k3c, n_groups_k3d;   /* of n_groups_k3: list C (simple windows that may hold a stop codon), list D (general sequence walk) */
} mp_run_stats;

/* Plan + pack genes [gene_lo, gene_hi) of a data set and make them resident in HBM
 * (replaces the per-gene loading + the data-independent part of the window scheduler,
 * src/microphasing.rs:905-1342). */
int mp_batch_create(mp_ctx* ctx, const mp_dataset* ds, int mode, uint64_t window_len, uint32_t gene_lo, uint32_t gene_hi,
                    mp_batch** out);
/* The same for an arbitrary, strictly ascending list of genes: a shard of a cost-balanced partition (genes are independent
 * units, src/microphasing.rs:895-942). The results list these genes in the given order. */
int mp_batch_create_genes(mp_ctx* ctx, const mp_dataset* ds, int mode, uint64_t window_len, const uint32_t* genes, uint32_t n_genes,
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
/* The same with a choice of streams (MP_STREAM_* mask): the text of a stream that is not asked for is never produced and reads as
 * empty. `normal` feeding build_reference needs the FASTA only - its TSV is 85 % of the bytes the consumer writes. */
enum { MP_STREAM_FASTA = 1, MP_STREAM_NORMAL_FASTA = 2, MP_STREAM_TSV = 4, MP_STREAM_ALL = 7 };
int mp_batch_results_select(mp_ctx* ctx, mp_batch* batch, uint32_t streams, mp_results** out);
void mp_batch_free(mp_batch* batch);
/* The seam between the device pass and the host consumer, on disk (diagnostics and CPU-only testing of the consumer - the part of
 * print_haplotypes / phase_gene that stays on the host, :604-880, :1345-1941): mp_batch_results_dump writes the device results of the
 * last mp_batch_run (per window: depth and haplotype groups; per group: flags; haplotype records) to a file; mp_batch_results_from_dump
 * consumes such a file with a batch planned from the same inputs and parameters - no GPU needed, a host-only context does - and yields
 * the same streams mp_batch_results_select gave on the GPU box. */
int mp_batch_results_dump(mp_ctx* ctx, mp_batch* batch, const char* path);
int mp_batch_results_from_dump(mp_ctx* ctx, mp_batch* batch, const char* path, uint32_t streams, mp_results** out);

/* Convenience: create + run + results for all genes (what `microphaser somatic` does). */
int mp_phase_dataset(mp_ctx* ctx, const mp_dataset* ds, int mode, uint64_t window_len, mp_results** out);

const char* mp_results_fasta(const mp_results* r, size_t* len);         /* stdout FASTA        */
const char* mp_results_normal_fasta(const mp_results* r, size_t* len);  /* --normal-output     */
const char* mp_results_tsv(const mp_results* r, size_t* len);           /* --tsv               */
uint64_t mp_results_windows(const mp_results* r);  /* main-ORF print_haplotypes calls actually made */
/* Byte offsets of the batch's genes in a stream (which: 0 stdout FASTA, 1 normal FASTA, 2 TSV): n_genes + 1 values, gene k owns
 * [off[k], off[k + 1]); off[0] is the length of the TSV header line (0 for the FASTA streams, and for an empty TSV). With them a
 * host merges the shards of several GPUs back into GTF order - the reference's output order. */
const uint64_t* mp_results_gene_offsets(const mp_results* r, int which, size_t* n_plus_1);
void mp_results_free(mp_results* r);

/* `microphaser build_reference` (reference: peptides::build, src/peptides.rs:148-186 <- run_build, src/main.rs:146-169):
 * translate every 3-nt-step window of every record of a nucleotide FASTA (reverse-complemented when the id does not
 * end in 'F', stop codons -> 'X') and build the set of distinct peptides. Translation and de-duplication run on the
 * GPU. peptide_len <= 12. */
typedef struct mp_peptides mp_peptides;
int mp_build_reference(mp_ctx* ctx, const char* fasta_path, uint32_t peptide_len, mp_peptides** out);
int mp_build_reference_buffer(mp_ctx* ctx, const char* fasta_text, size_t len, uint32_t peptide_len, mp_peptides** out);   /* the FASTA's bytes */
/* The same without the translated FASTA (mp_peptides_fasta is empty): the peptidome only - keys and binary -, what a pipeline that
 * goes on to `filter` needs (a whole-exome `normal` FASTA translates into ~10^8 peptide records of text nobody reads). */
int mp_peptidome_from_buffer(mp_ctx* ctx, const char* fasta_text, size_t len, uint32_t peptide_len, mp_peptides** out);
const char* mp_peptides_fasta(const mp_peptides* p, size_t* len);     /* stdout of build_reference            */
const char* mp_peptides_binary(const mp_peptides* p, size_t* len);    /* --output: bincode HashSet<Vec<u8>>   */
const uint64_t* mp_peptides_keys(const mp_peptides* p, size_t* n);    /* sorted distinct keys (5 bits/residue): the unit
                                                                         exchanged in the multi-GPU peptidome union */
uint64_t mp_peptides_count(const mp_peptides* p);                     /* translated windows                   */
void mp_peptides_free(mp_peptides* p);
/* to_protein (reference: src/peptides.rs:128-146) for n windows of 3 * peptide_len nucleotides laid out back to back in nt;
 * reverse[i] != 0: reverse complement first (ids that do not end in 'F', :161-164). aa gets n * peptide_len residues (stop = 'X'),
 * keys (may be NULL) the 5-bit-per-residue keys. Runs on the GPU. */
int mp_translate(mp_ctx* ctx, const uint8_t* nt, const uint8_t* reverse, uint64_t n, uint32_t peptide_len, uint8_t* aa, uint64_t* keys);
/* The exchange step of a multi-GPU build_reference: the union of sorted distinct key arrays (one per rank, as
 * all-gathered) -> one peptidome with the same accessors (its FASTA stream is empty). Host-side merge of sorted runs; no GPU needed. */
int mp_peptides_union(mp_ctx* ctx, const uint64_t* const* keys, const uint64_t* counts, uint32_t n_arrays, uint32_t peptide_len, mp_peptides** out);

/* `microphaser filter` (reference: peptides::filter, src/peptides.rs:221-709 <- run_filtering, src/main.rs:170-214,
 * src/filter_cli.yaml): translate the mutant / normal windows of a `somatic` info.tsv, drop self-similar, repeated and
 * post-stop peptides, remove those present in the reference peptidome (bincode HashSet<Vec<u8>> from build_reference)
 * and annotate the rest with the maximum-likelihood frequency and the 95 % credible interval of their variant region.
 * Translation, peptidome membership and the per-region statistics run on the GPU. peptide_len <= 12.
 * mp_filter reads the two files; mp_filter_buffers takes their bytes (read in place); mp_filter_peptides takes the peptidome as the
 * handle build_reference / mp_peptides_union returned (its sorted keys go to the GPU as they are: no bincode round trip) and
 * filters at that peptidome's peptide length. */
typedef struct mp_filtered mp_filtered;
int mp_filter(mp_ctx* ctx, const char* tsv_path, const char* reference_binary_path, uint32_t peptide_len, mp_filtered** out);
int mp_filter_buffers(mp_ctx* ctx, const char* tsv, size_t tsv_len, const char* reference_binary, size_t reference_len,
                      uint32_t peptide_len, mp_filtered** out);
int mp_filter_peptides(mp_ctx* ctx, const char* tsv, size_t tsv_len, const mp_peptides* reference, mp_filtered** out);
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
