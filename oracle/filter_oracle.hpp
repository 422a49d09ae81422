// TEST INFRASTRUCTURE - NOT PRODUCT CODE.
// CPU restatement of `microphaser filter` (reference: src/peptides.rs:188-709, src/main.rs:170-214, src/filter_cli.yaml).
// Third-party arithmetic restated from the published algorithms of the crates the reference pins in Cargo.toml (no
// Cargo.lock in the tree): statrs 0.15 `Binomial::pmf` / `factorial::ln_binomial` / `gamma::ln_gamma`, bio 0.34
// `LogProb::{ln_simpsons_integrate_exp, ln_sum_exp}`, itertools-num `linspace`. Pinned by the reference's three filter
// fixtures (tests/lib.rs:146-211: tumor FASTA, normal FASTA and TSV incl. seven distinct credible-interval strings);
// the depth > 170 branch of ln_factorial (Lanczos ln_gamma) is not reached by any fixture: parity unpinned there.
#pragma once
#include <string>

namespace mp_oracle {

struct FilterOutput {
    std::string fasta;          // stdout: kept tumor peptides
    std::string normal_fasta;   // --normal-output
    std::string tsv;            // --tsv-output (header always written)
    std::string removed_tsv;    // --similar-removed (header only when a record is written)
    std::string removed_fasta;  // --removed-peptides
};

// peptides::filter. `reference_binary` = bytes of the bincode HashSet<Vec<u8>> file, `tsv_text` = info.tsv of `somatic`.
void filter(const std::string& reference_binary, const std::string& tsv_text, size_t peptide_length, FilterOutput& out);

}  // namespace mp_oracle
