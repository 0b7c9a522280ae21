// VALU issue-rate microbenchmark, second form: one instruction kind per kernel (inline asm, 8 independent chains per lane),
// W waves per SIMD obtained with W blocks of 256 threads per CU.  Prints cycles per wave-instruction per SIMD
// (= wave-cycles / instructions / waves-per-SIMD) -- the price list behind DESIGN.md's VALU cost model of the traversal loop.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate2.hip -o tools/ubench/valu_rate2 && tools/ubench/valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CH 8
#define REP 16

#define OP1(str) asm volatile(str : "+v"(a[i]) : "v"(m))
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
    float a[CH];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) { a[i] = threadIdx.x * 0.001f + i; p[i] = (f2){ a[i], a[i] + 1.f }; }
    float m = out[0] + 1.0f;
    f2 pm = { m, m };
    unsigned sh = (unsigned)(threadIdx.x & 7);
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int i = 0; i < CH; i++) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) % CH]));
                else if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
                else if (KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
                else if (KIND == 6) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 7) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(sh));
                else if (KIND == 8) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[i]));
                else if (KIND == 9) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 10) asm volatile("v_ffbh_u32 %0, %0" : "+v"(a[i]));
                else if (KIND == 11) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[i]) : "v"(m));
                else if (KIND == 12) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
                else if (KIND == 13) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
                else if (KIND == 14) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");
                else if (KIND == 15) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x80" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) % CH]));
                else if (KIND == 16) asm volatile("v_lshl_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(sh), "v"(m));
                else if (KIND == 17) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) % CH]));
                else if (KIND == 18) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 19) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 20) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) % CH]));
                else if (KIND == 21) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 22) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(sh));
                else if (KIND == 23) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 24) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 25) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[i]) : "v"(pm));
                else if (KIND == 26) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");
                else if (KIND == 27) asm volatile("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a[i]) : "v"(m));
            }
        }
    }
    unsigned long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < CH; i++) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name) {
    printf("%-26s", name);
    for (int w : { 1, 2, 4, 6, 8 }) {
        const int blocks = 256 * w, threads = 256;
        float* out; unsigned long long* cyc;
        hipMalloc(&out, sizeof(float) * blocks * threads); hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
        hipMemset(out, 0, sizeof(float) * blocks * threads);
        const int iters = 1000;
        hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instrPerWave = (double)iters * REP * CH;
        // wall-clock view: SIMD-cycles at 2.4 GHz per wave-instruction (what a saturated SIMD pays per instruction)
        const double perSimdInstr = instrPerWave * w;
        printf("  w%d %5.2f", w, ms * 1e-3 * 2.4e9 / perSimdInstr);
        hipFree(out); hipFree(cyc);
    }
    printf("   (cycles @2.4 GHz per wave-instruction per SIMD, by wall time)\n");
}

int main() {
    run<0>("v_add_f32"); run<1>("v_mul_f32"); run<18>("v_sub_f32"); run<19>("v_fma_f32"); run<2>("v_max_f32"); run<3>("v_max3_f32"); run<20>("v_min3_f32");
    run<4>("v_pk_mul_f32"); run<5>("v_pk_add_f32"); run<25>("v_pk_add_f32 neg");
    run<6>("v_and_b32"); run<27>("v_and_b32_sdwa"); run<7>("v_lshlrev_b32"); run<8>("v_bfe_u32"); run<9>("v_bcnt_u32_b32"); run<10>("v_ffbh_u32");
    run<11>("v_alignbit_b32"); run<12>("v_cvt_f32_i32"); run<13>("v_cndmask_b32 vcc"); run<14>("v_cmp_lt_f32 vcc"); run<26>("v_cmp_eq_u32 vcc");
    run<15>("v_bitop3_b32"); run<16>("v_lshl_or_b32"); run<17>("v_and_or_b32"); run<21>("v_mov_b32"); run<22>("v_xad_u32"); run<23>("v_add_u32"); run<24>("v_lshl_add_u32");
    return 0;
}
