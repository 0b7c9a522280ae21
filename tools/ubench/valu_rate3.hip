// v_cndmask / v_perm / v_or / v_xor issue rates (same harness as valu_rate2.hip; masks in an SGPR pair)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CH 8
#define REP 16
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) a[i] = threadIdx.x * 0.001f + i;
    float m = out[0] + 1.0f;
    unsigned long long mask = __builtin_amdgcn_ballot_w64((threadIdx.x & 1) != 0);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int i = 0; i < CH; i++) {
                if (KIND == 0) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(mask));
                else if (KIND == 1) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) % CH]));
                else if (KIND == 2) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 3) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 4) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 5) asm volatile("v_lshrrev_b32 %0, 8, %0" : "+v"(a[i]));
                else if (KIND == 6) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 7) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 8) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 9) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) % CH]));
                else if (KIND == 10) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a[i]));
                else if (KIND == 11) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 12) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 13) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) % CH]));
                else if (KIND == 14) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 15) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(a[i]));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < CH; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> void run(const char* name) {
    printf("%-26s", name);
    for (int w : { 1, 2, 6 }) {
        const int blocks = 256 * w, threads = 256;
        float* out; (void)hipMalloc(&out, sizeof(float) * blocks * threads); (void)hipMemset(out, 0, sizeof(float) * blocks * threads);
        const int iters = 1000;
        hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(threads), 0, 0, out, iters); (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(threads), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  w%d %5.2f", w, ms * 1e-3 * 2.4e9 / ((double)iters * REP * CH * w));
        (void)hipFree(out);
    }
    printf("\n");
}
int main() {
    run<0>("v_cndmask_b32 (sgpr mask)"); run<1>("v_perm_b32"); run<2>("v_or_b32"); run<3>("v_xor_b32"); run<4>("v_min_f32"); run<5>("v_lshrrev_b32 imm");
    run<6>("v_sub_u32"); run<7>("v_add3_u32"); run<8>("v_or3_b32"); run<9>("v_bfi_b32"); run<10>("v_cvt_f32_ubyte0"); run<11>("v_mul_u32_u24"); run<12>("v_mad_u32_u24");
    run<13>("v_med3_f32"); run<14>("v_max_i32"); run<15>("v_ashrrev_i32");
    return 0;
}
