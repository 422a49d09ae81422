// Launch interface of the `microphaser filter` kernels (kernels_filter.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mp {

// K5: one thread per sequence (mutant / normal nucleotide window of a TSV row): to_protein (src/peptides.rs:128-146) into
// aa[aa_off[s] ..], then for every peptide_len-mer of it a 5-bit key, a "contains X" flag and membership in the sorted
// reference peptidome keys (binary search).  flags bit0 = has X, bit1 = member of the reference set.
void device_translate_records(const uint8_t* d_nt, const uint64_t* d_nt_off, const uint32_t* d_nt_len, const uint8_t* d_rev,
                              const uint64_t* d_aa_off, uint64_t n_seq, uint32_t L, const uint64_t* d_ref_keys, uint64_t n_ref,
                              uint8_t* d_aa, uint8_t* d_flags, uint32_t* d_err, hipStream_t stream);

struct CredibleInterval { uint32_t ml; uint32_t status; double a, b; };  // status != 0: the reference would have panicked (NaN density)

// K6: one thread per record group: maximum-likelihood grid, Simpson normalisation and the credible-interval search
// (src/peptides.rs:398-481 when the region changes, :569-660 after the last row).
void device_credible_intervals(const uint64_t* d_grp_off, const uint8_t* d_grp_final, const double* d_alt, const uint32_t* d_depth,
                               uint64_t n_groups, const double* d_ln_fact /* 171 entries */, CredibleInterval* d_out, hipStream_t stream);

}  // namespace mp
