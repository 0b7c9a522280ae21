// round 5: issue rates of the opcodes the earlier lists left out (same harness as valu_rate3.hip)
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
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[CH]; f2 pm = { m, m };
    for (int i = 0; i < CH; i++) p[i] = (f2){ a[i], a[i] };
    float sm = __builtin_amdgcn_readfirstlane(m);
    unsigned long long mask = __builtin_amdgcn_ballot_w64((threadIdx.x & 1) != 0);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int i = 0; i < CH; i++) {
                if (KIND == 0) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
                else if (KIND == 1) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
                else if (KIND == 2) asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
                else if (KIND == 3) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pm));
                else if (KIND == 6) asm volatile("v_subrev_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sm));
                else if (KIND == 7) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sm));
                else if (KIND == 8) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a[i]));
                else if (KIND == 9) asm volatile("v_ffbh_u32 %0, %0" : "+v"(a[i]));
                else if (KIND == 10) asm volatile("v_bfm_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 11) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "s"(sm));
                else if (KIND == 12) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 13) asm volatile("v_add_u32 %0, -1, %0" : "+v"(a[i]));
                else if (KIND == 14) asm volatile("v_xor_b32 %0, 31, %0" : "+v"(a[i]));
                else if (KIND == 15) asm volatile("v_and_b32 %0, 0x7fffff, %0" : "+v"(a[i]));
                else if (KIND == 16) asm volatile("v_add_f32 %0, 0xcb000000, %0" : "+v"(a[i]));
                else if (KIND == 17) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xd8" : "+v"(a[i]) : "v"(m), "s"(sm));
                else if (KIND == 18) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 19) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
                else if (KIND == 20) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 21) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
                else if (KIND == 22) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                else if (KIND == 23) asm volatile("v_sad_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < CH; i++) s += a[i] + p[i].x + p[i].y;
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
    run<0>("v_lshlrev_b32 imm");
    run<1>("v_lshrrev_b32 var");
    run<2>("v_ashrrev_i32 var");
    run<3>("v_lshl_add_u32 imm");
    run<4>("v_fma_f32");
    run<5>("v_pk_fma_f32");
    run<6>("v_subrev_f32 sgpr");
    run<7>("v_mul_f32 sgpr");
    run<8>("v_ffbl_b32");
    run<9>("v_ffbh_u32");
    run<10>("v_bfm_b32");
    run<11>("v_mov_b32 sgpr");
    run<12>("v_sub_f32");
    run<13>("v_add_u32 imm");
    run<14>("v_xor_b32 imm");
    run<15>("v_and_b32 lit");
    run<16>("v_add_f32 lit");
    run<17>("v_bitop3 sgpr");
    run<18>("v_mul_lo_u32");
    run<19>("v_cvt_f32_u32");
    run<20>("v_max3_f32");
    run<21>("v_pk_add_f32");
    run<22>("v_min_u32");
    run<23>("v_sad_u32");
    return 0;
}
