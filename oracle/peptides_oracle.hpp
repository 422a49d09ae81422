// TEST INFRASTRUCTURE - NOT PRODUCT CODE.
// CPU restatement of the translation / reference-peptidome path (reference: src/peptides.rs:85-186).
// Pinned by the reference's own fixture tests/resources/test_build (FASTA byte-equal) and the
// test_filter*/reference.binary files (same four peptides, bincode layout; set-equal).
#pragma once
#include <set>
#include <string>
#include <vector>

namespace mp_oracle {

// to_protein (src/peptides.rs:128-146): upper-case, reverse-complement if frame < 0, codon -> amino acid
// (stop = 'X'). Throws mp::Error where the reference unwraps an Err (codon not in the table).
std::string to_protein(const std::string& nt, int frame);

// peptides::build (src/peptides.rs:148-186): returns the translated FASTA (stdout of `build_reference`) and fills
// `set` with the distinct peptides (the HashSet that is bincode-serialized to --output).
std::string build_reference(const std::string& fasta_text, size_t peptide_length, std::set<std::string>& set);

// bincode v1 layout of HashSet<Vec<u8>>: u64 LE count, then u64 LE length + bytes per element (order arbitrary).
std::string bincode_set(const std::set<std::string>& set);

}  // namespace mp_oracle
