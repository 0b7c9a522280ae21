// VALU issue-rate microbenchmark: N dependent-free chains of v_fma_f32 / v_and_b32 / v_cndmask per wave,
// W waves per SIMD.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int CHAINS>
__global__ void k(float* out, int iters, unsigned long long* cyc) {
    float a[CHAINS];
    unsigned u[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; i++) { a[i] = threadIdx.x * 0.001f + i; u[i] = threadIdx.x + i; }
    float m = out[0];
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int i = 0; i < CHAINS; i++) {
                if (KIND == 0) a[i] = __builtin_fmaf(a[i], m, 1.0f);
                else if (KIND == 1) u[i] = (u[i] & 0x7fffffffu) + 3u;          // v_and + v_add (2 ops)
                else if (KIND == 2) a[i] = __builtin_fmaxf(a[i], m) + 1.0f;    // v_max + v_add
                else if (KIND == 3) u[i] = __builtin_popcount(u[i]) + u[i];    // v_bcnt (with add)
                else if (KIND == 4) a[i] = (a[i] < m) ? a[i] + 1.0f : a[i] * 0.5f;  // cmp + add + mul + cndmask
                else if (KIND == 5) {                                               // v_pk_mul_f32 + v_pk_add_f32 (2 ops, 2 chains' worth)
                    if (i + 1 < CHAINS && (i & 1) == 0) {
                        f32x2 v = { a[i], a[i + 1] };
                        v = v * (f32x2){ m, m };
                        asm volatile("" : "+v"(v));
                        v = v + (f32x2){ 1.0f, 1.0f };
                        a[i] = v.x; a[i + 1] = v.y;
                    }
                }
                else if (KIND == 6) {                                               // v_max3_f32 + v_min3_f32
                    float t;
                    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(a[i]), "v"(m), "v"(a[(i + 1) % CHAINS]));
                    asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(t), "v"(m), "v"(a[(i + 2) % CHAINS]));
                }
                else if (KIND == 7) u[i] |= (a[i] <= m) ? (1u << (r & 7)) : 0u;     // v_cmp + v_cndmask + v_or (the RTO_CHILD shape)
                else if (KIND == 8) a[i] = (float)(int)u[i] + a[i];                 // v_cvt_f32_i32 + v_add
            }
        }
    }
    unsigned long long t1 = clock64();
    float s = 0; unsigned su = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; i++) { s += a[i]; su += u[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + su;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int CHAINS>
void run(const char* name, int opsPerStep, int wavesPerSimd) {
    const int blocks = 256, threads = 256 * wavesPerSimd;   // 4 SIMDs x wavesPerSimd waves
    if (threads > 1024) return;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads); hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    hipMemset(out, 0, sizeof(float) * blocks * threads);
    const int iters = 2000;
    hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += (double)v; avg /= blocks;
    const double instrPerWave = (double)iters * 16 * CHAINS * opsPerStep;
    const double perSimd = instrPerWave * wavesPerSimd;
    printf("%-22s chains %d waves/SIMD %d: %.2f clk64-ticks per wave-instr per SIMD (wave view %.2f); wall %.3f ms -> %.2f ns/instr/SIMD\n",
           name, CHAINS, wavesPerSimd, avg / perSimd, avg / instrPerWave, ms, ms * 1e6 / perSimd);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 4}) {
        run<0, 1>("fma dep-chain", 1, w);
        run<0, 8>("fma 8 chains", 1, w);
        run<1, 8>("and+add 8 chains", 2, w);
        run<2, 8>("max+add 8 chains", 2, w);
        run<3, 8>("bcnt 8 chains", 1, w);
        run<4, 8>("cmp/add/mul/cndmask", 4, w);
        run<5, 8>("pk_mul+pk_add (per pk op)", 1, w);     // 8 chains = 4 pairs x 2 pk ops = 8 pk instructions per step
        run<6, 8>("max3+min3", 2, w);
        run<7, 8>("cmp+cndmask+or", 3, w);
        run<8, 8>("cvt_f32_i32+add", 2, w);
    }
    return 0;
}
