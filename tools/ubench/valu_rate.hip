// Micro-benchmark: sustained wave64 issue rate of the integer VALU ops the matcher uses (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters)
{
    uint32_t a = threadIdx.x * 2654435761u, b = blockIdx.x + 12345u, c = a ^ b, d = a + b, e = 1, f = 2, g = 3, h = 4;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {            // xor only (8 independent)
                asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                             "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(i));
            } else if (MODE == 1) {     // bcnt accumulate (8 independent chains)
                asm volatile("v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %8, %1\n v_bcnt_u32_b32 %2, %8, %2\n v_bcnt_u32_b32 %3, %8, %3\n"
                             "v_bcnt_u32_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %8, %5\n v_bcnt_u32_b32 %6, %8, %6\n v_bcnt_u32_b32 %7, %8, %7"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(i));
            } else if (MODE == 2) {     // med3 + min
                asm volatile("v_med3_u32 %0, %1, %0, %8\n v_min_u32 %1, %1, %8\n v_med3_u32 %2, %3, %2, %8\n v_min_u32 %3, %3, %8\n"
                             "v_med3_u32 %4, %5, %4, %8\n v_min_u32 %5, %5, %8\n v_med3_u32 %6, %7, %6, %8\n v_min_u32 %7, %7, %8"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(i));
            } else if (MODE == 3) {     // the matcher's per-pair mix: xor,bcnt x4 (dependent chain) twice
                asm volatile("v_xor_b32 %0, %4, %8\n v_bcnt_u32_b32 %1, %0, %1\n v_xor_b32 %2, %5, %8\n v_bcnt_u32_b32 %3, %2, %3\n"
                             "v_xor_b32 %0, %6, %8\n v_bcnt_u32_b32 %1, %0, %1\n v_xor_b32 %2, %7, %8\n v_bcnt_u32_b32 %3, %2, %3"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(i));
            } else {                    // lshl_or
                asm volatile("v_lshl_or_b32 %0, %0, 1, %8\n v_lshl_or_b32 %1, %1, 1, %8\n v_lshl_or_b32 %2, %2, 1, %8\n v_lshl_or_b32 %3, %3, 1, %8\n"
                             "v_lshl_or_b32 %4, %4, 1, %8\n v_lshl_or_b32 %5, %5, 1, %8\n v_lshl_or_b32 %6, %6, 1, %8\n v_lshl_or_b32 %7, %7, 1, %8"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(i));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}
template <int MODE> void run(const char* name, uint32_t* d, int blocksPerCU)
{
    const int iters = 2000, blocks = 256 * blocksPerCU;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * 16 * 8;
    printf("%-28s blocks/CU %d: %.3f ms, %.2f T lane-ops/s, %.2f cycles/wave-instr/SIMD @2.4GHz\n", name, blocksPerCU, ms,
           winstr * 64 / (ms * 1e-3) / 1e12, (ms * 1e-3 * 2.4e9) / (winstr / 1024));
}
int main()
{
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int b : {1, 2, 4, 8}) {
        if (b == 1) printf("--- 1 wave/SIMD\n"); 
        run<0>("v_xor_b32", d, b); run<1>("v_bcnt_u32_b32 acc", d, b); run<2>("v_med3_u32+v_min_u32", d, b); run<3>("xor->bcnt chains (matcher)", d, b); run<4>("v_lshl_or_b32", d, b);
    }
    return 0;
}
