// K4: codon -> peptide translation and reference-peptidome keys (reference: src/peptides.rs:85-146 to_protein /
// to_aminoacid / make_pairs, :148-186 build). One thread per peptide window; integer/byte work, HBM-bound:
// reads 3L nucleotide bytes (neighbouring windows overlap by 3L-3, so the stream is read once through L2),
// writes L amino-acid bytes + one u64 key. De-duplication = radix sort + unique on the keys (rocPRIM device primitives, called directly).
#include <hip/hip_runtime.h>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/functional.hpp>

#include "kernels_pep.hpp"

namespace mp {

[[noreturn]] void throw_hip(hipError_t e, const char* file, int line);
#define HIP_OK_(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw_hip(e_, __FILE__, __LINE__); } while (0)

// index = 16*b0 + 4*b1 + b2 with A=0 C=1 G=2 T=3; stop codons -> 'X' (src/peptides.rs:85-117)
__constant__ char CODON_AA[65] = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVVXYXYSSSSXCWCLFLF";

__device__ __forceinline__ int base2(uint8_t c, bool complement) {
    if (c >= 'a' && c <= 'z') c -= 32;  // to_ascii_uppercase (:129)
    int b;
    switch (c) { case 'A': b = 0; break; case 'C': b = 1; break; case 'G': b = 2; break; case 'T': b = 3; break; default: return -1; }
    return complement ? 3 - b : b;
}

__global__ __launch_bounds__(256) void k4_translate(const uint8_t* __restrict__ nt, const uint64_t* __restrict__ win_off,
                                                    const uint8_t* __restrict__ win_rev, uint64_t n, uint32_t L,
                                                    uint8_t* __restrict__ aa, uint64_t* __restrict__ keys, uint32_t* __restrict__ err) {
    uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const uint8_t* s = nt + win_off[i];
    const bool rev = win_rev[i] != 0;
    const uint32_t n3 = 3 * L;
    uint64_t key = 0;
    bool bad = false;
    for (uint32_t j = 0; j < L; j++) {
        int b0, b1, b2;
        if (!rev) { b0 = base2(s[3 * j], false); b1 = base2(s[3 * j + 1], false); b2 = base2(s[3 * j + 2], false); }
        else { b0 = base2(s[n3 - 1 - 3 * j], true); b1 = base2(s[n3 - 2 - 3 * j], true); b2 = base2(s[n3 - 3 - 3 * j], true); }  // dna::revcomp (:133)
        char a = '?';
        if ((b0 | b1 | b2) < 0) bad = true;  // codon not in the table: the reference unwraps an Err (:136-139)
        else a = CODON_AA[16 * b0 + 4 * b1 + b2];
        aa[i * L + j] = uint8_t(a);
        key = (key << 5) | uint64_t((a - 'A') & 31);
    }
    keys[i] = key;
    if (bad) atomicOr(err, 1u);
}

void device_translate(const uint8_t* d_nt, const uint64_t* d_off, const uint8_t* d_rev, uint64_t n, uint32_t L, uint8_t* d_aa,
                      uint64_t* d_keys, uint32_t* d_err, hipStream_t stream) {
    if (!n) return;
    dim3 grid(uint32_t((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(k4_translate, grid, block, 0, stream, d_nt, d_off, d_rev, n, L, d_aa, d_keys, d_err);
    HIP_OK_(hipGetLastError());
}

// sort + unique of the u64 peptide keys; returns the number of distinct keys (in d_out[0..n_unique))
uint64_t device_sort_unique(uint64_t* d_keys, uint64_t* d_tmp, uint64_t* d_out, uint64_t n, uint32_t key_bits, hipStream_t stream) {
    if (!n) return 0;
    const size_t count = size_t(n);   // (64-bit sizes throughout: a whole-exome normal peptidome has > 2^31 / 8 windows within reach)
    size_t bytes1 = 0, bytes2 = 0;
    HIP_OK_(rocprim::radix_sort_keys(nullptr, bytes1, d_keys, d_tmp, count, 0u, key_bits, stream));
    HIP_OK_(rocprim::unique(nullptr, bytes2, d_tmp, d_out, static_cast<uint64_t*>(nullptr), count, rocprim::equal_to<uint64_t>(), stream));
    const size_t ws_bytes = (std::max(bytes1, bytes2) + 255) & ~size_t(255);
    char* d_ws = nullptr;   // one allocation: workspace + the count word behind it
    HIP_OK_(hipMalloc(&d_ws, ws_bytes + 256));
    uint64_t* d_count = reinterpret_cast<uint64_t*>(d_ws + ws_bytes);
    uint64_t cnt = 0;
    hipError_t e = rocprim::radix_sort_keys(d_ws, bytes1, d_keys, d_tmp, count, 0u, key_bits, stream);
    if (e == hipSuccess) e = rocprim::unique(d_ws, bytes2, d_tmp, d_out, d_count, count, rocprim::equal_to<uint64_t>(), stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&cnt, d_count, 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_ws);
    HIP_OK_(e);
    return cnt;
}

}  // namespace mp
