// Launch interface of the translation / peptidome kernels (kernels_pep.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mp {
void device_translate(const uint8_t* d_nt, const uint64_t* d_off, const uint8_t* d_rev, uint64_t n, uint32_t L, uint8_t* d_aa,
                      uint64_t* d_keys, uint32_t* d_err, hipStream_t stream);
uint64_t device_sort_unique(uint64_t* d_keys, uint64_t* d_tmp, uint64_t* d_out, uint64_t n, uint32_t key_bits, hipStream_t stream);
}  // namespace mp
