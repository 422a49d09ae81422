// `microphaser filter` (reference: src/peptides.rs:188-709, src/main.rs:170-214): removes neopeptides that also occur in
// the normal peptidome and annotates the rest with a maximum-likelihood frequency and a 95 % credible interval.
// Translation + peptidome membership (K5) and the per-group statistics (K6) run on the device; the row-stream
// bookkeeping of the reference (stop-gain suppression, per-variant-region grouping, de-duplication) stays on the host.
#pragma once
#include <string>

#include "model.hpp"

namespace mp {

struct FilterResult {
    std::string fasta;          // stdout: kept tumor peptides
    std::string normal_fasta;   // --normal-output
    std::string tsv;            // --tsv-output (header always present)
    std::string removed_tsv;    // --similar-removed
    std::string removed_fasta;  // --removed-peptides
    uint64_t n_rows = 0, n_peptides = 0, n_groups = 0, n_kept = 0, n_removed = 0;
    float translate_ms = 0, stats_ms = 0;
};

// reference_binary: bytes of the bincode HashSet<Vec<u8>> written by build_reference; tsv_text: info.tsv of `somatic`.
void filter_device(int device, const std::string& reference_binary, const std::string& tsv_text, uint32_t peptide_len, FilterResult& out);

}  // namespace mp
