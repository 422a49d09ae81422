// Micro-benchmark: sustained SALU vs VALU issue rate per CU on gfx950 with all wave slots occupied
// (used to decide which instruction class bounds the replay kernel). Build: hipcc --offload-arch=gfx950 -O3 issue_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned* out, int iters) {
    unsigned a = threadIdx.x, b = blockIdx.x, c = 3, d = 5;
    unsigned sa = blockIdx.x, sb = 7, sc = 11, sd = 13;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0 || MODE == 2) {
            asm volatile("s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %0\n"
                         "s_xor_b32 %0, %0, %2\n s_xor_b32 %1, %1, %3\n s_lshl_b32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                         : "+s"(sa), "+s"(sb), "+s"(sc), "+s"(sd) : : "scc");
        }
        if (MODE == 3) {   // the SHA-1 round's instruction mix: rotate (v_alignbit), choose (v_bfi), three-operand add, xor
            asm volatile("v_alignbit_b32 %0, %0, %0, 27\n v_bfi_b32 %1, %0, %2, %3\n v_add3_u32 %2, %2, %1, %0\n v_xor_b32 %3, %3, %0\n"
                         "v_alignbit_b32 %1, %1, %1, 2\n v_bfi_b32 %0, %1, %3, %2\n v_add3_u32 %3, %3, %0, %1\n v_xor_b32 %2, %2, %1\n"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        }
        if (MODE == 4) {   // f32 FMA, for comparison with the guide's 2 cycles per wave64 on a SIMD-32
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1\n"
                         "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %3, %0\n v_fma_f32 %2, %2, %0, %1\n v_fma_f32 %3, %3, %1, %2\n"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        }
        if (MODE == 1 || MODE == 2) {
            asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n"
                         "v_xor_b32 %0, %0, %2\n v_xor_b32 %1, %1, %3\n v_lshlrev_b32 %2, 1, %2\n v_add_u32 %3, 1, %3\n"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + sa + sb + sc + sd;
}
int main() {
    unsigned* out; hipMalloc(&out, 8192 * 64 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, blocks = 8192 * 2;
    for (int mode = 0; mode < 5; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, out, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, out, iters);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double instr = double(blocks) * iters * 8 * (mode == 2 ? 2 : 1);
            if (rep) printf("mode %d (%s): %.3f ms, %.1f G wave-instr/s, per CU per clock (2.4 GHz, 256 CUs): %.2f\n", mode,
                            mode == 0 ? "SALU" : mode == 1 ? "VALU int add/xor/shift" : mode == 2 ? "SALU+VALU" : mode == 3 ? "VALU alignbit/bfi/add3/xor (SHA-1 mix)" : "VALU v_fma_f32", ms, instr / ms / 1e6, instr / (ms * 1e-3) / 2.4e9 / 256);
        }
    }
    return 0;
}
