// `microphaser filter` (reference: src/peptides.rs:188-709, src/main.rs:170-214): removes neopeptides that also occur in
// the normal peptidome and annotates the rest with a maximum-likelihood frequency and a 95 % credible interval.
// Translation + peptidome membership (K5) and the per-group statistics (K6) run on the device; the row-stream
// bookkeeping of the reference (stop-gain suppression, per-variant-region grouping, de-duplication) stays on the host.
#pragma once
#include <string>
#include <string_view>
#include <vector>

#include "model.hpp"

namespace mp {

struct FilterResult {
    PodVec<char> fasta;          // stdout: kept tumor peptides
    PodVec<char> normal_fasta;   // --normal-output
    PodVec<char> tsv;            // --tsv-output (header always present)
    PodVec<char> removed_tsv;    // --similar-removed
    PodVec<char> removed_fasta;  // --removed-peptides
    uint64_t n_rows = 0, n_peptides = 0, n_groups = 0, n_kept = 0, n_removed = 0;
    float translate_ms = 0, stats_ms = 0;
};

// reference_binary: bytes of the bincode HashSet<Vec<u8>> written by build_reference - or, with reference_keys != nullptr, the
// peptidome as sorted distinct keys of peptide_len residues (PeptideResult::keys; reference_binary is then not read);
// tsv_text: info.tsv of `somatic`. Both buffers are read in place and must stay valid for the call.
void filter_device(int device, std::string_view reference_binary, const std::vector<uint64_t>* reference_keys, std::string_view tsv_text,
                   uint32_t peptide_len, FilterResult& out);

}  // namespace mp
