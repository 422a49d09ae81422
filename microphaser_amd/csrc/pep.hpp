// `microphaser build_reference` on the device (reference: src/peptides.rs:148-186, src/main.rs:146-169).
#pragma once
#include <string>
#include <string_view>
#include <vector>

#include "model.hpp"

namespace mp {

struct PeptideResult {
    std::string fasta;                // translated FASTA (stdout of `build_reference`)
    std::vector<uint64_t> keys;       // sorted distinct peptide keys (5 bits per residue, first residue most significant)
    uint32_t peptide_len = 9;
    uint64_t n_peptides = 0;
    float translate_ms = 0, dedup_ms = 0;
    std::string binary() const;       // bincode v1 HashSet<Vec<u8>> of the distinct peptides (order = key order)
};

std::string peptide_from_key(uint64_t key, uint32_t L);
uint64_t peptide_to_key(const std::string& pep);

// Translate every 3-nt-step window of every record of a nucleotide FASTA and de-duplicate, on HIP device `device`.
// want_fasta = false: the peptidome (keys, binary) only - what a pipeline that feeds `filter` needs; the translated FASTA stays empty.
void build_reference_device(int device, std::string_view fasta_text, uint32_t peptide_len, PeptideResult& out, bool want_fasta = true);   // (the text is read in place)

}  // namespace mp
