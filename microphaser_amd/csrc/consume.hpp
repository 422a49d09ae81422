// Consumer entry point, see consume.cpp.
#pragma once
#include "batch.hpp"
#include "device.hpp"

namespace mp {

// Walk every planned transcript of the batch and emit FASTA / normal FASTA / TSV exactly as
// microphasing::phase_gene would (reference: src/microphasing.rs:882-1941), answering every
// print_haplotypes call from the device results. streams: STREAM_* mask (model.hpp) - the text of a stream that is not asked for is
// not produced (it stays empty, and its per-gene offsets stay 0).
void consume_batch(const Batch& b, const HostResults& res, PhasedStreams& out, uint32_t streams = STREAM_ALL);

// The same for `microphaser normal` (reference: src/normal_microphasing.rs:650-1279); the batch must have been planned
// with normal = true.
void consume_batch_normal(const Batch& b, const HostResults& res, PhasedStreams& out, uint32_t streams = STREAM_ALL);

}  // namespace mp
