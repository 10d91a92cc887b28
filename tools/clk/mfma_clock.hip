// how fast does the chip actually clock during short launches?  a pure-MFMA kernel (no memory) of a known instruction count, timed with HIP
// events for several launch lengths, cold and after a sustained burn; s_memtime / s_memrealtime deltas give the counters' rates.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void burn(int iters, float* out, unsigned long long* clk) {
    f32x16 acc = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (acc[0] == 123.456f) out[0] = acc[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
static double run(int iters, int reps, unsigned long long* dclk, float* dout, unsigned long long* hclk) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(burn, dim3(256 * 2), dim3(256), 0, 0, iters, dout, dclk);   // 2 workgroups (8 waves) per CU
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hclk, dclk, 16, hipMemcpyDeviceToHost);
    return ms / reps;
}
int main() {
    float* dout; unsigned long long* dclk; unsigned long long h[2];
    hipMalloc(&dout, 64); hipMalloc(&dclk, 16);
    // per launch: 512 workgroups x 4 waves x iters x 16 MFMAs x 4096 flop
    auto tf = [](int iters, double ms) { return 512.0 * 4 * iters * 16 * 4096 / (ms * 1e-3) / 1e12; };
    run(10, 1, dclk, dout, h);                                      // module load
    usleep(500000);
    const int its[] = {50, 500, 5000, 50000};
    for (int pass = 0; pass < 2; ++pass) {
        for (int k = 0; k < 4; ++k) {
            usleep(pass == 0 ? 300000 : 0);                         // pass 0: every launch after 0.3 s of idle; pass 1: back to back
            const double ms = run(its[k], 1, dclk, dout, h);
            printf("%s iters %6d: %9.3f ms  %6.1f TF (peak 157.3)   s_memtime ticks %llu -> %.0f MHz   wall_clock64 ticks %llu -> %.1f MHz\n",
                   pass == 0 ? "after idle  " : "back to back", its[k], ms, tf(its[k], ms), h[0], h[0] / (ms * 1e3), h[1], h[1] / (ms * 1e3));
        }
    }
    // 2 ms bursts separated by idle gaps, like a bench step train: does the clock hold?
    for (int gap_us : {0, 100, 1000, 10000}) {
        double worst = 0, best = 1e9;
        for (int r = 0; r < 20; ++r) {
            const double ms = run(1500, 1, dclk, dout, h);
            worst = ms > worst ? ms : worst; best = ms < best ? ms : best;
            usleep(gap_us);
        }
        printf("20 x (1500-iter launch + %5d us idle): best %.3f ms (%.1f TF)  worst %.3f ms (%.1f TF)\n", gap_us, best, tf(1500, best), worst, tf(1500, worst));
    }
    return 0;
}
