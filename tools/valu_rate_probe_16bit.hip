// Second VALU issue-rate probe (gfx950, round 4): v_cndmask forms and the NON-packed 16-bit min / max / add forms against the packed ones.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_rate_probe_16bit.hip -o tools/_build/valu_rate_probe_16bit && tools/_build/valu_rate_probe_16bit
// Result (profiles/r04_valu_issue_rates_16bit.txt): v_max_u16 / v_min_u16 / v_max_f16 / v_add_u16 issue in 2.4 cycles per wave64 (packed and 32-bit
// min / max: 4.2-4.4; v_max3_u16: 8.3) - one pixel per lane, so a packed 3-input minimum (two pixels, two comparisons each: 1.09 cycles per
// pixel comparison) still beats them (2.4) by 2.2x: the FAST score stays on v_pk_minimum3_f16 / v_pk_maximum3_f16.  v_cndmask costs 4.3 like
// any VOP3 (the 23 cycles of profiles/r02_valu_issue_rates.txt were an artefact of a VCC that nothing had written).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHAINS 8
#define ITERS 4096
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
    uint32_t x[CHAINS];
    for (int c = 0; c < CHAINS; c++) x[c] = seed * (threadIdx.x + 17u * c + 1u);
    uint32_t a = seed ^ 0x00030005u, b2 = seed ^ 0x00110007u;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (OP == 0) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x[c]) : "v"(a), "v"(b2));
            if (OP == 1) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc");
            if (OP == 2) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2) : "vcc");
            if (OP == 3) asm volatile("v_cmp_lt_u32 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]" : "+v"(x[c]) : "v"(a), "v"(b2) : "s20", "s21");
            if (OP == 4) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2));
            if (OP == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(a), "v"(b2));
            if (OP == 6) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2));
            if (OP == 7) asm volatile("v_max_u16 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2));
            if (OP == 8) asm volatile("v_max_i16 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2));
            if (OP == 9) asm volatile("v_min_u16 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2));
            if (OP == 10) asm volatile("v_max_f16 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2));
            if (OP == 11) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2));
            if (OP == 12) asm volatile("v_max_u32 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2));
            if (OP == 13) asm volatile("v_max3_u16 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b2));
            if (OP == 14) asm volatile("v_max_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(x[c]) : "v"(a), "v"(b2));
            if (OP == 15) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2));
            if (OP == 16) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2));
            if (OP == 17) asm volatile("v_sub_u16 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2));
            if (OP == 18) asm volatile("v_add_u16 %0, %0, %1" : "+v"(x[c]) : "v"(a), "v"(b2));
        }
    }
    uint32_t s = 0;
    for (int c = 0; c < CHAINS; c++) s ^= x[c];
    if (s == 0x12345678u) out[0] = s;
}
static const char *names[] = {"cndmask sgpr-pair (no write)", "cmp vcc + cndmask vcc", "cmp vcc", "cmp sgpr + cndmask sgpr", "v_and_b32", "cndmask vcc (no clobber)", "v_min_u32", "v_max_u16", "v_max_i16", "v_min_u16", "v_max_f16", "v_max_f32", "v_max_u32", "v_max3_u16", "v_max_u16_sdwa hi", "v_pk_max_u16", "v_mul_hi_u32_u24", "v_sub_u16", "v_add_u16"};
template <int OP> static void run(uint32_t *d, int w) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * w;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double n = (double)w * ITERS * CHAINS;
    printf("%-30s waves/SIMD %d: %.2f cycles per asm statement\n", names[OP], w, ms * 1e6 / n * 2.4);
}
int main() {
    uint32_t *d; hipMalloc(&d, 64);
    for (int w : {4, 8}) { run<0>(d, w); run<1>(d, w); run<2>(d, w); run<3>(d, w); run<4>(d, w); run<5>(d, w); run<6>(d, w); run<7>(d, w); run<8>(d, w); run<9>(d, w); run<10>(d, w); run<11>(d, w); run<12>(d, w); run<13>(d, w); run<14>(d, w); run<15>(d, w); run<16>(d, w); run<17>(d, w); run<18>(d, w); printf("\n"); }
    return 0;
}
