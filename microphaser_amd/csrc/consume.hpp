// Consumer entry point, see consume.cpp.
#pragma once
#include "batch.hpp"
#include "device.hpp"

namespace mp {

// Walk every planned transcript of the batch and emit FASTA / normal FASTA / TSV exactly as
// microphasing::phase_gene would (reference: src/microphasing.rs:882-1941), answering every
// print_haplotypes call from the device results.
void consume_batch(const Batch& b, const HostResults& res, SomaticOutput& out);

}  // namespace mp
