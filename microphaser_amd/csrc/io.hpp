// Minimal native ingest for the on-disk formats either side of the hot path:
// BGZF/BAM, plain-text VCF, GTF2, FASTA + .fai. The reference gets these from
// rust-htslib 0.36 / rust-bio 0.34 (not vendored); their observable behaviour at
// the call sites in src/microphasing.rs:895-942, 1943-2131 is restated here.
#pragma once
#include <deque>
#include <functional>
#include <mutex>
#include <istream>
#include <map>
#include <string>
#include <vector>

#include "model.hpp"

namespace mp {

// ---------------------------------------------------------------- BAM
struct BamData {
    std::vector<std::string> ref_names;
    std::vector<int64_t> ref_lens;
    ReadStore reads;                  // mapped + placed records in file (coordinate) order
    std::vector<size_t> tid_begin;    // first read index of each tid (size n_ref+1)
    int64_t max_ref_span = 0;         // max(end_pos - pos, 1) over all reads (bounds the region-iterator seek)
    int tid_of(const std::string& chrom) const {
        for (size_t i = 0; i < ref_names.size(); i++)
            if (ref_names[i] == chrom) return int(i);
        return -1;
    }
};

// Load a whole coordinate-sorted BAM into memory. Records with refID < 0 are dropped.
void load_bam(const std::string& path, BamData& out);
// Write a coordinate-sorted BAM (BGZF, no index) from a ReadStore - used by the synthetic generator.
void write_bam(const std::string& path, const std::vector<std::string>& ref_names, const std::vector<int64_t>& ref_lens,
               const ReadStore& reads);

// Emulation of rust_htslib::bam::RecordBuffer::fetch over an in-memory BAM
// (call site: src/microphasing.rs:905; constructed with cache_cigar=false at :1954).
class ReadBuffer {
  public:
    explicit ReadBuffer(const BamData& bam) : bam_(bam) {}
    // After the call, `records()` holds what `read_buffer.iter()` would yield.
    void fetch(const std::string& chrom, uint64_t start, uint64_t end);
    const std::deque<size_t>& records() const { return inner_; }

  private:
    bool next_record(size_t& idx);
    const BamData& bam_;
    std::deque<size_t> inner_;
    bool has_overflow_ = false;
    size_t overflow_ = 0;
    // iterator state of the underlying IndexedReader
    int iter_tid_ = -1;
    int64_t iter_beg_ = 0;
    size_t cursor_ = 0, cursor_end_ = 0;
    bool iter_valid_ = false;
};

// ---------------------------------------------------------------- VCF
struct VcfRecord {
    std::string chrom;
    uint64_t pos = 0;  // 0-based
    std::string ref;
    std::vector<std::string> alts;
    bool somatic = false;       // INFO/SOMATIC flag
    std::string ann_first;      // first comma-separated entry of INFO/ANN ("" if absent)
    bool has_svlen = false;
    std::vector<int64_t> svlen;
};

struct VcfData {
    std::vector<std::string> contigs;  // from ##contig header lines (+ contigs seen in records)
    std::vector<VcfRecord> records;    // file order
    // lazily built per-contig index (record numbers in file order + whether their positions are sorted)
    struct ContigIndex { std::vector<size_t> recs; bool sorted = true; };
    mutable std::map<std::string, ContigIndex> by_chrom;
    mutable bool indexed = false;
    void build_index() const;
};

void load_vcf(const std::string& path, VcfData& out);

// Variant::new (reference: src/common.rs:71-175). Appends to `out`; throws mp::Error where the
// reference panics (unsupported allele without -u).
void variants_from_record(const VcfRecord& rec, bool unsupported_allele_warning_only, std::vector<Variant>& out);

// variant_tree of one gene (reference: src/microphasing.rs:932-942): records with
// gene.start <= pos <= gene.end on gene.chrom; a later record at the same POS replaces the earlier.
void gene_variants(const VcfData& vcf, const std::string& chrom, uint64_t start, uint64_t end,
                   bool unsupported_allele_warning_only, std::vector<Variant>& out);

// ---------------------------------------------------------------- FASTA
// Anything that can serve reference bases for [start, stop) of a contig.
struct RefSource {
    virtual ~RefSource() = default;
    virtual void fetch(const std::string& chrom, uint64_t start, uint64_t stop, std::vector<uint8_t>& out) const = 0;
};

// In-memory contigs (synthetic data sets).
struct MemFasta : RefSource {
    std::map<std::string, const std::string*> contigs;
    void fetch(const std::string& chrom, uint64_t start, uint64_t stop, std::vector<uint8_t>& out) const override;
};

class IndexedFasta : public RefSource {
  public:
    explicit IndexedFasta(const std::string& path);  // needs path + ".fai"
    // bio::io::fasta::IndexedReader::fetch + read: [start, stop), case preserved.
    void fetch(const std::string& chrom, uint64_t start, uint64_t stop, std::vector<uint8_t>& out) const override;
    bool has(const std::string& chrom) const { return idx_.count(chrom) != 0; }
    void preload() const { ensure_loaded(); }   // read the file now (fetch does it on first use otherwise)

  private:
    struct Entry { uint64_t len, offset, line_bases, line_bytes, region_start, region_len; };
    std::map<std::string, Entry> idx_;
    std::string path_;
    mutable PodVec<uint8_t> file_;  // whole file (lazily loaded)
    mutable std::once_flag once_;
    void ensure_loaded() const;
};

// ---------------------------------------------------------------- GTF
// Streams a GTF the way microphasing::phase does (reference: src/microphasing.rs:1982-2128) and calls
// `on_gene` for every completed gene (all biotypes; the caller applies the protein_coding filter
// exactly where the reference does, :1964). Throws mp::Error for an unsorted GTF (:1998-2000) and
// for missing attributes. `normal` mode ignores three_prime_utr records (src/normal_microphasing.rs:1319-1433).
void stream_gtf(std::istream& in, const std::function<void(const Gene&)>& on_gene, bool use_three_prime_utr = true);

// Convenience: run the per-gene loading of phase_gene (refseq, reads, variants) for every
// protein-coding gene of a GTF stream.
void load_gene_inputs(std::istream& gtf, const BamData& bam, const VcfData& vcf, const RefSource& fasta,
                      bool unsupported_allele_warning_only, const std::function<void(GeneInput&)>& on_gene,
                      bool use_three_prime_utr = true);

}  // namespace mp
