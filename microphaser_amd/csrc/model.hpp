// Shared host-side data model: the types that cross the phase_gene seam.
//
// Mirrors the reference's model types (reference: src/common.rs:38-222 Variant,
// :224-348 Gene/Transcript/Interval/PhasingStrand, :350-373 IDRecord) and the
// in-memory form of the BAM/VCF/FASTA inputs the reference gets from rust-htslib
// and rust-bio. Reads are kept in pooled struct-of-arrays form (ReadStore) so the
// very same pools can be uploaded to HBM unchanged.
#pragma once
#include <functional>
#include <memory>
#include <memory>
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace mp {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// ---------------------------------------------------------------- variants
enum VarKind : uint8_t { VK_SNV = 0, VK_INS = 1, VK_DEL = 2 };

// reference: src/common.rs:38-59 (enum Variant) + accessors :177-221
struct Variant {
    VarKind kind = VK_SNV;
    uint64_t pos = 0;        // 0-based
    uint8_t alt = 0;         // SNV alt base (as written in the VCF)
    std::string seq;         // insertion: whole ALT incl. anchor base
    uint64_t len = 0;        // indel length (ALT-1 for ins, REF-1 / |SVLEN| for del)
    bool is_germline = true;
    std::string prot_change;

    uint64_t end_pos() const { return kind == VK_DEL ? pos + len - 1 : pos; }  // common.rs:185-191
    uint64_t frameshift() const {                                              // common.rs:215-221
        switch (kind) {
            case VK_SNV: return 0;
            case VK_DEL: return len % 3;
            default: return (3 - ((uint64_t(seq.size()) - 1) % 3)) % 3;
        }
    }
};

// ---------------------------------------------------------------- gene model
enum Strand : uint8_t { FORWARD = 0, REVERSE = 1 };

struct Interval {  // common.rs:293-338 (0-based half-open)
    uint64_t start = 0, end = 0, frame = 0;
};

struct Transcript {  // common.rs:271-291
    std::string id, biotype;
    Strand strand = FORWARD;
    std::vector<Interval> exons;
    bool is_coding() const { return !exons.empty(); }
};

struct Gene {  // common.rs:224-253
    std::string id, name, chrom, biotype;
    Interval interval;
    std::vector<Transcript> transcripts;
    uint64_t start() const { return interval.start; }
    uint64_t end() const { return interval.end; }
};

// ---------------------------------------------------------------- reads
// BAM CIGAR op codes (SAM spec): MIDNSHP=X
enum CigarOp : uint32_t { C_M = 0, C_I = 1, C_D = 2, C_N = 3, C_S = 4, C_H = 5, C_P = 6, C_EQ = 7, C_X = 8 };

// Pooled struct-of-arrays read storage. cigar words are BAM-encoded (len<<4|op),
// seq is BAM 4-bit packed (two bases per byte, high nibble first), qual is raw phred.
// std::vector whose resize() leaves trivially-constructible elements uninitialised: the big host arrays are sized once and then
// filled (and their pages first touched) by all host threads, instead of being zero-filled by one.
void advise_huge(const void* p, size_t bytes);
template <class T> struct DefaultInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = DefaultInitAlloc<U>; };
    // a large array asks for transparent huge pages before it is touched: its page faults - and, when it is given back, the page-table
    // teardown under the process-wide mm lock - shrink by a factor of 512
    T* allocate(size_t n) {
        T* p = std::allocator<T>::allocate(n);
        advise_huge(p, n * sizeof(T));
        return p;
    }
    template <class U, class... A> void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new (static_cast<void*>(p)) U;
        else ::new (static_cast<void*>(p)) U(std::forward<A>(a)...);
    }
};
template <class T> using PodVec = std::vector<T, DefaultInitAlloc<T>>;

// Large, freshly allocated host buffers (the merged batch arrays, the device results, the output streams - gigabytes at whole-exome
// size) are first touched by many threads at once: ask for transparent huge pages so that this costs thousands of page faults, not
// millions. A hint only; no effect where the kernel does not offer it.
void advise_huge(const void* p, size_t bytes);

// Giving gigabytes of scratch back to the OS (the planner's sub-batches, the consumer's pieces, the downloaded results) is page-table
// work under the process-wide mm lock: it is taken off the caller's path - the object is moved to the library's reaper thread, which
// destroys it. The reaper is OWNED: it starts with the first job, mp_destroy of the last live context waits for the queue to drain and
// joins it (reaper_release), and so does the library's static teardown - no detached thread outlives the library or races an unload.
void reaper_post(std::function<void()> job);
void reaper_retain();    // a context came to life
void reaper_release();   // a context died; the last one drains the queue and joins the thread
void reaper_drain();     // blocks until every posted job has run (tests, orderly teardown)
bool reaper_running();   // a reaper thread exists (tests)
template <class T> void release_later(T&& obj) {
    auto held = std::make_shared<std::decay_t<T>>(std::move(obj));   // (std::function needs a copyable callable)
    reaper_post([held]() mutable { held.reset(); });
}

struct ReadStore {
    PodVec<int32_t> tid;
    PodVec<int64_t> pos;
    PodVec<int64_t> end_pos;  // pos + sum(M,=,X,D,N)   (rust-htslib CigarStringView::end_pos)
    PodVec<uint8_t> mapq;
    PodVec<uint16_t> flag;
    PodVec<uint32_t> l_seq;
    PodVec<uint32_t> n_cigar;
    PodVec<uint64_t> cigar_off;  // index into cigar_pool
    PodVec<uint64_t> seq_off;    // byte index into seq_pool
    PodVec<uint64_t> qual_off;   // byte index into qual_pool
    PodVec<uint64_t> qname_off;  // byte index into qname_pool (NUL terminated)
    PodVec<uint32_t> cigar_pool;
    PodVec<uint8_t> seq_pool;
    PodVec<uint8_t> qual_pool;
    PodVec<char> qname_pool;

    size_t size() const { return pos.size(); }
    const char* qname(size_t i) const { return &qname_pool[qname_off[i]]; }
    const uint32_t* cigar(size_t i) const { return cigar_pool.data() + cigar_off[i]; }
    const uint8_t* qual(size_t i) const { return qual_pool.data() + qual_off[i]; }
    // decoded upper-case base (rust-htslib Seq::index -> "=ACMGRSVTWYHKDBN")
    uint8_t base(size_t i, uint32_t k) const {
        uint8_t b = seq_pool[seq_off[i] + (k >> 1)];
        return "=ACMGRSVTWYHKDBN"[(k & 1) ? (b & 0xF) : (b >> 4)];
    }
    // append one read; returns its index
    size_t add(int32_t tid_, int64_t pos_, uint8_t mapq_, uint16_t flag_, const uint32_t* cig, uint32_t ncig,
               const uint8_t* seq4, uint32_t lseq, const uint8_t* q, const char* name) {
        size_t i = pos.size();
        tid.push_back(tid_);
        pos.push_back(pos_);
        int64_t e = pos_;
        for (uint32_t k = 0; k < ncig; k++) {
            uint32_t op = cig[k] & 0xF, l = cig[k] >> 4;
            if (op == C_M || op == C_EQ || op == C_X || op == C_D || op == C_N) e += l;
        }
        end_pos.push_back(e);
        mapq.push_back(mapq_);
        flag.push_back(flag_);
        l_seq.push_back(lseq);
        n_cigar.push_back(ncig);
        cigar_off.push_back(cigar_pool.size());
        cigar_pool.insert(cigar_pool.end(), cig, cig + ncig);
        seq_off.push_back(seq_pool.size());
        seq_pool.insert(seq_pool.end(), seq4, seq4 + (lseq + 1) / 2);
        qual_off.push_back(qual_pool.size());
        qual_pool.insert(qual_pool.end(), q, q + lseq);
        qname_off.push_back(qname_pool.size());
        size_t nl = std::strlen(name);
        qname_pool.insert(qname_pool.end(), name, name + nl + 1);
        return i;
    }
};

// rust-htslib CigarStringView::read_pos(ref_pos, include_softclips=false, include_dels=false)
// restated (the crate is not vendored in the reference; call site src/microphasing.rs:106).
// Returns -1 for None.
inline int64_t cigar_read_pos(const uint32_t* cig, uint32_t ncig, int64_t read_start, int64_t ref_pos) {
    int64_t rpos = read_start;  // reference position
    int64_t qpos = 0;           // position within read
    uint32_t j = 0;
    // find the first op that refers to read sequence (M,=,X,I,S); leading H/P are skipped,
    // a leading D/N is an error in the crate (mapped to "no position" here, as the caller does).
    for (uint32_t i = 0; i < ncig; i++) {
        uint32_t op = cig[i] & 0xF;
        if (op == C_M || op == C_EQ || op == C_X || op == C_I || op == C_S) { j = i; break; }
        if (op == C_D || op == C_N) return -1;
        if (op == C_H && i > 0 && i + 1 < ncig) return -1;
        if (i + 1 == ncig) return -1;  // only pads / hard clips
    }
    while (rpos <= ref_pos && j < ncig) {
        uint32_t op = cig[j] & 0xF;
        int64_t l = cig[j] >> 4;
        switch (op) {
            case C_M: case C_EQ: case C_X:
                if (rpos <= ref_pos && rpos + l > ref_pos) return qpos + (ref_pos - rpos);
                rpos += l; qpos += l; j++; break;
            case C_S: case C_I: qpos += l; j++; break;
            case C_D: case C_N: rpos += l; j++; break;
            case C_P: j++; break;
            case C_H: return -1;  // trailing hard clip (or misplaced one: error -> no position)
            default: return -1;
        }
    }
    return -1;
}

// A contig's reference bases, case preserved (bio::io::fasta::IndexedReader::read).
struct RefContig {
    std::string name;
    uint64_t len = 0;
};

// Everything phase_gene loads for one gene (reference: src/microphasing.rs:895-942).
struct GeneInput {
    Gene gene;
    std::vector<uint8_t> refseq;      // [gene.start, gene.end+100), case preserved
    std::vector<size_t> reads;        // indices into the ReadStore, BAM file order (mapq NOT yet filtered)
    std::vector<Variant> variants;    // variant_tree flattened: ascending pos, ALT order within pos
    // A gene too deep for the row slots of the sequential replay is planned as several copies that each hold a SUBSET of its reads
    // (plan.cpp split_deep_genes); the copies must walk the schedule of the whole gene, which depends on the reads only through
    // max_read_len (candidate key range, src/microphasing.rs:913-915, :1198-1248): the whole gene's value, 0 = derive from the reads
    uint64_t max_read_len_override = 0;
};

// One output record (reference: src/common.rs:350-373, field order = TSV column order).
struct IDRecord {
    std::string id, transcript, gene_id, gene_name, chrom;
    uint64_t offset = 0, frame = 0;
    double freq = 0;
    uint32_t depth = 0, nvar = 0, nsomatic = 0, nvariant_sites = 0, nsomvariant_sites = 0;
    std::string strand, variant_sites, somatic_positions, somatic_aa_change, germline_positions,
        germline_aa_change, normal_sequence, mutant_sequence;
};

// The three output streams of `microphaser somatic` (stdout FASTA, --normal-output FASTA, --tsv).
struct GeneEnds { uint64_t fasta, normal_fasta, tsv; };
// Which streams a caller wants (mp_batch_results_select): the text of a stream whose bit is clear is never produced.
enum : uint32_t { STREAM_FASTA = 1, STREAM_NORMAL_FASTA = 2, STREAM_TSV = 4, STREAM_ALL = 7 };   // sizes of the three streams after a gene has been written
struct SomaticOutput {
    std::string fasta, normal_fasta, tsv;
    uint32_t streams = STREAM_ALL;
    bool tsv_header_written = false;
    uint64_t n_windows = 0;  // main-ORF print_haplotypes invocations (the benchmark unit)
    std::vector<GeneEnds> gene_ends;   // filled by the device consumer only (one entry per gene of its range)
};

// `microphaser normal`: its own record type with a single peptide_sequence column
// (reference: src/normal_microphasing.rs:80-102) and two output streams (stdout FASTA, --tsv).
struct NormalRecord {
    std::string id, transcript, gene_id, gene_name, chrom;
    uint64_t offset = 0, frame = 0;
    double freq = 0;
    uint32_t depth = 0, nvar = 0, nsomatic = 0, nvariant_sites = 0, nsomvariant_sites = 0;
    std::string strand, variant_sites, somatic_positions, somatic_aa_change, germline_positions, germline_aa_change, peptide_sequence;
};
struct NormalOutput {
    std::string fasta, tsv;
    uint32_t streams = STREAM_ALL;
    bool tsv_header_written = false;
    uint64_t n_windows = 0;
    std::vector<GeneEnds> gene_ends;
};

// A finished output stream: one uninitialised allocation, so that the pieces written by the consumer threads are copied into it
// (and its pages first touched) by all host threads at once - at config C the three streams are 3.5 GB of text.
struct Bytes {
    std::unique_ptr<char[]> p;
    size_t n = 0;
    const char* data() const { return p ? p.get() : ""; }
    size_t size() const { return n; }
};
struct PhasedStreams {   // what mp_results holds: somatic -> three streams, normal -> fasta + tsv
    Bytes fasta, normal_fasta, tsv;
    uint64_t n_windows = 0;
    // byte offsets of the batch's genes in each stream (n_genes + 1 entries; [0] = length of the TSV header line for the TSV):
    // what a host needs to merge the shards of several GPUs back into GTF order
    std::vector<uint64_t> gene_off[3];
};

}  // namespace mp
