// TEST INFRASTRUCTURE - NOT PRODUCT CODE.
// CPU oracle: a sequential restatement of microphaser's somatic phasing path
// (reference: src/microphasing.rs, src/common.rs). Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Parity status: PINNED by the reference's own fixtures (tests/golden/, see
// tests/test_oracle_fixtures.py): forward_test, reverse_test, splice_forward_test,
// splice_reverse_test, test_empty expected .fa/.normal.fa/.tsv files. The reference itself
// (Rust, needs cargo + crates.io + htslib) cannot be built in this environment.
#pragma once
#include "../microphaser_amd/csrc/model.hpp"

namespace mp_oracle {

// microphasing::phase_gene (reference: src/microphasing.rs:882-1941) for one loaded gene.
// `reads` is the BAM the indices in `gi.reads` refer to. Appends to `out`.
void phase_gene(const mp::GeneInput& gi, const mp::ReadStore& reads, uint64_t window_len, mp::SomaticOutput& out);

}  // namespace mp_oracle
