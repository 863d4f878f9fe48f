// VALU issue-rate probe (gfx950): how many cycles does one wave64 instruction of each kind hold a SIMD when many waves are
// resident?  The FAST kernels are issue-bound on packed-f16 min/max; this decides which instruction forms are worth using.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_rate_probe.hip -o tools/_build/valu_rate_probe && tools/_build/valu_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef short short2v __attribute__((ext_vector_type(2)));
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
#define CHAINS 8
#define ITERS 4096

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
    uint32_t x[CHAINS];
    for (int c = 0; c < CHAINS; c++) x[c] = seed * (threadIdx.x + 17u * c + 1u);
    uint32_t a = seed ^ 0x00030005u, b2 = seed ^ 0x00110007u;
    unsigned long long pr[CHAINS];
    for (int c = 0; c < CHAINS; c++) pr[c] = seed * 77ull + c;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (OP == 0) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 1) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 2) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 3) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 4) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 5) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 6) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 7) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 8) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 9) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 10) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 11) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 12) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 13) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 14) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 15) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 16) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 17) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 18) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 19) asm volatile("v_mov_b32 %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 20) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 21) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 22) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 23) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 24) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 25) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 26) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 27) asm volatile("v_mbcnt_lo_u32_b32 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 28) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 29) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 30) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 31) asm volatile("v_cmp_lt_i32 vcc, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 32) asm volatile("v_cmp_lt_i16 vcc, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 33) asm volatile("v_cmp_gt_i16_sdwa vcc, %0, %1 src0_sel:WORD_1 src1_sel:DWORD" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 34) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 35) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 36) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 37) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 38) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 39) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 40) asm volatile("v_lshrrev_b64 %1, %2, %1" : "+v"(x[c]), "+v"(pr[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 41) asm volatile("v_readfirstlane_b32 s20, %0" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 42) asm volatile("v_max_u16 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 43) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 44) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc", "s20");
            if (OP == 45) asm volatile("v_pk_add_f32 %1, %1, %1" : "+v"(x[c]), "+v"(pr[c]) : "v"(a), "v"(b2) : "vcc", "s20");
        }
    }
    uint32_t s = 0;
    for (int c = 0; c < CHAINS; c++) s ^= x[c] ^ (uint32_t)pr[c];
    if (s == 0x12345678u) out[0] = s;
}
static const char *names[] = {"v_pk_minimum3_f16", "v_pk_min_f16", "v_pk_max_i16", "v_pk_add_f16", "v_pk_add_u16", "v_pk_mul_lo_u16", "v_min_u32", "v_min3_u32", "v_min_f32", "v_max3_f32", "v_fma_f32", "v_add_f32", "v_add_u32", "v_sub_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_mov_b32", "v_cndmask_b32", "v_add3_u32", "v_or3_b32", "v_lshl_add_u32", "v_and_or_b32", "v_bfe_u32", "v_bcnt_u32_b32", "v_mbcnt_lo_u32_b32", "v_perm_b32", "v_alignbyte_b32", "v_mov_b32_dpp", "v_cmp_lt_i32", "v_cmp_lt_i16", "v_cmp_gt_i16_sdwa", "v_mul_lo_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_dot4_u32_u8", "v_dot2_u32_u16", "v_sad_u8", "v_lshrrev_b64 (pair)", "v_readfirstlane_b32", "v_med3_f16? v_max_u16", "v_pk_fma_f16", "v_cvt_f32_ubyte0", "v_pk_add_f32 (pair)"};
template <int OP> static void run(uint32_t *d, int wavesPerSimd) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * wavesPerSimd;   // 256 threads = 4 waves = 1 per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double instrPerSimd = (double)wavesPerSimd * ITERS * CHAINS;
    printf("%-26s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", names[OP], wavesPerSimd, ms,
           ms * 1e6 / instrPerSimd, ms * 1e6 / instrPerSimd * 2.4);
}
int main() {
    uint32_t *d;
    hipMalloc(&d, 64);
    for (int w : {1, 4, 8}) {
        run<0>(d, w); run<1>(d, w); run<2>(d, w); run<3>(d, w); run<4>(d, w); run<5>(d, w); run<6>(d, w); run<7>(d, w); run<8>(d, w); run<9>(d, w); run<10>(d, w); run<11>(d, w); run<12>(d, w); run<13>(d, w); run<14>(d, w); run<15>(d, w); run<16>(d, w); run<17>(d, w); run<18>(d, w); run<19>(d, w); run<20>(d, w); run<21>(d, w); run<22>(d, w); run<23>(d, w); run<24>(d, w); run<25>(d, w); run<26>(d, w); run<27>(d, w); run<28>(d, w); run<29>(d, w); run<30>(d, w); run<31>(d, w); run<32>(d, w); run<33>(d, w); run<34>(d, w); run<35>(d, w); run<36>(d, w); run<37>(d, w); run<38>(d, w); run<39>(d, w); run<40>(d, w); run<41>(d, w); run<42>(d, w); run<43>(d, w); run<44>(d, w); run<45>(d, w); 
        printf("\n");
    }
    return 0;
}
