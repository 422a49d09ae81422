// TEST INFRASTRUCTURE - NOT PRODUCT CODE. CPU restatement of `microphaser normal`, see normal_oracle.cpp.
#pragma once
#include "../microphaser_amd/csrc/model.hpp"

namespace mp_oracle {
// normal_microphasing::phase_gene (reference: src/normal_microphasing.rs:650-1279) for one loaded gene
// (reads NOT mapq-filtered, GTF three_prime_utr records ignored by the caller).
void normal_phase_gene(const mp::GeneInput& gi, const mp::ReadStore& reads, uint64_t window_len, mp::NormalOutput& out);
}  // namespace mp_oracle
