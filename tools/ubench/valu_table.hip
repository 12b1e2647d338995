// Micro-benchmark: sustained wave64 issue rate (cycles per wave-instruction per SIMD at 8 waves/SIMD) of the VALU ops the hot
// kernels are made of (gfx950).  Each op runs as 8 independent chains.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_table valu_table.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define R8(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define OPS(X) \
    X(0,  "v_xor_b32 %0, %0, %8")            X(1,  "v_and_b32 %0, %0, %8")           X(2,  "v_or_b32 %0, %0, %8") \
    X(3,  "v_add_u32 %0, %0, %8")            X(4,  "v_sub_u32 %0, %0, %8")           X(5,  "v_min_u32 %0, %0, %8") \
    X(6,  "v_max_i32 %0, %0, %8")            X(7,  "v_lshlrev_b32 %0, 1, %0")        X(8,  "v_lshrrev_b32 %0, 1, %0") \
    X(9,  "v_mul_u32_u24 %0, %0, %8")        X(10, "v_mul_lo_u32 %0, %0, %8")        X(11, "v_mad_u32_u24 %0, %0, %8, %0") \
    X(12, "v_min3_i32 %0, %0, %8, %0")       X(13, "v_perm_b32 %0, %0, %8, %0")      X(14, "v_alignbyte_b32 %0, %0, %8, 1") \
    X(15, "v_add3_u32 %0, %0, %8, %0")       X(16, "v_lshl_add_u32 %0, %0, 2, %8")   X(17, "v_bfe_u32 %0, %0, 3, 8") \
    X(18, "v_pk_min_i16 %0, %0, %8")         X(19, "v_pk_sub_i16 %0, %0, %8")        X(20, "v_pk_max_u16 %0, %0, %8") \
    X(21, "v_dot4_u32_u8 %0, %0, %8, %0")    X(22, "v_dot2_u32_u16 %0, %0, %8, %0")  X(23, "v_mov_b32 %0, %8") \
    X(24, "v_cvt_f32_ubyte0 %0, %0")         X(25, "v_rndne_f32 %0, %0")             X(26, "v_cvt_i32_f32 %0, %0") \
    X(27, "v_add_f32 %0, %0, %8")            X(28, "v_mul_f32 %0, %0, %8")           X(29, "v_fma_f32 %0, %0, %8, %0") \
    X(30, "v_fmac_f32 %0, %8, %8")           X(31, "v_min_f32 %0, %0, %8")           X(32, "v_min3_f32 %0, %0, %8, %0") \
    X(33, "v_sad_u8 %0, %0, %8, %0")         X(34, "v_bcnt_u32_b32 %0, %8, %0")      X(35, "v_cndmask_b32 %0, %0, %8, vcc") \
    X(36, "v_min_i16 %0, %0, %8")            X(37, "v_lerp_u8 %0, %0, %8, %0")       X(38, "v_and_or_b32 %0, %0, %8, %0") \
    X(39, "v_med3_u32 %0, %0, %8, %0")       X(40, "v_sub_u16 %0, %0, %8")           X(41, "v_mul_i32_i24 %0, %0, %8") \
    X(42, "v_xor_b32 %0, 0x55aa55aa, %0")    X(45, "v_cndmask_b32_e64 %0, %0, %8, s[20:21]") X(46, "v_cndmask_b32 %0, %0, %8, vcc\n v_add_u32 %0, %0, %8") \
    X(47, "v_cmp_lt_i32 vcc, %0, %8\n v_add_u32 %0, %0, %8")   X(48, "v_cmp_lt_i32_e64 s[20:21], %0, %8\n v_add_u32 %0, %0, %8")  X(49, "v_lshlrev_b32 %0, %8, %0") \
    X(50, "v_lshl_or_b32 %0, %0, 1, %8")     X(51, "v_max_u16 %0, %0, %8")           X(52, "v_add_u16 %0, %0, %8") \
    X(53, "v_min_i32 %0, %0, %8")            X(54, "v_ashrrev_i32 %0, 1, %0")        X(55, "v_subrev_u32 %0, %0, %8") \
    X(56, "v_and_b32 %0, 0x7f7f7f7f, %0")    X(57, "v_lshrrev_b32 %0, %8, %0")       X(58, "v_mul_lo_u16 %0, %0, %8") \
    X(59, "v_fmac_f32 %0, %8, %0")            X(60, "v_max_f32 %0, %0, %8")           X(61, "v_sub_f32 %0, %0, %8") \
    X(62, "v_cvt_f32_i32 %0, %0")            X(63, "v_mad_i32_i24 %0, %0, %8, %0")   X(64, "v_xad_u32 %0, %0, %8, %0") \
    X(65, "v_add_u32 %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD") X(66, "v_max_i16 %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2") \
    X(67, "v_mul_hi_u32 %0, %0, %8")         X(68, "v_bitop3_b32 %0, %0, %8, %0 bitop3:0xc8") X(69, "v_mul_hi_i32_i24 %0, %0, %8") \
    X(43, "v_pk_add_f32 %0, %0, %9")        X(44, "v_pk_mul_f32 %0, %0, %9")

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters)
{
    uint32_t a = threadIdx.x * 2654435761u, b = blockIdx.x + 12345u, c = a ^ b, d = a + b, e = 1, f = 2, g = 3, h = 4;
    uint64_t w = ((uint64_t)a << 32) | b, a2 = w, b2 = w + 1, c2 = w + 2, d2 = w + 3, e2 = w + 4, f2 = w + 5, g2 = w + 6, h2 = w + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#define X(N, S) if (MODE == N && (N < 43 || N > 44)) { \
            asm volatile(S : "+v"(a) : "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(i), "v"(w)); asm volatile(S : "+v"(b) : "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(i), "v"(w)); \
            asm volatile(S : "+v"(c) : "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(i), "v"(w)); asm volatile(S : "+v"(d) : "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(i), "v"(w)); \
            asm volatile(S : "+v"(e) : "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(i), "v"(w)); asm volatile(S : "+v"(f) : "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(i), "v"(w)); \
            asm volatile(S : "+v"(g) : "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(i), "v"(w)); asm volatile(S : "+v"(h) : "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(i), "v"(w)); } \
          if (MODE == N && (N == 43 || N == 44)) { \
            asm volatile(S : "+v"(a2) : "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(i), "v"(w)); asm volatile(S : "+v"(b2) : "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(i), "v"(w)); \
            asm volatile(S : "+v"(c2) : "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(i), "v"(w)); asm volatile(S : "+v"(d2) : "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(i), "v"(w)); \
            asm volatile(S : "+v"(e2) : "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(i), "v"(w)); asm volatile(S : "+v"(f2) : "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(i), "v"(w)); \
            asm volatile(S : "+v"(g2) : "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(i), "v"(w)); asm volatile(S : "+v"(h2) : "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(b), "v"(i), "v"(w)); }
            OPS(X)
#undef X
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ (uint32_t)(a2 ^ b2 ^ c2 ^ d2 ^ e2 ^ f2 ^ g2 ^ h2);
}
template <int MODE> void run(const char* name, uint32_t* d)
{
    const int iters = 1000, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * 16 * 8;
    printf("%-42s %.2f cycles per wave-instruction per SIMD @2.4 GHz\n", name, (ms * 1e-3 * 2.4e9) / (winstr / 1024));
}
int main()
{
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
#define X(N, S) run<N>(S, d);
    OPS(X)
#undef X
    return 0;
}
