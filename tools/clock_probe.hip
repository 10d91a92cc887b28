// clock_probe: effective shader clock under partial-chip load (how fast do 8 busy CUs clock?)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void busy(int iters, long long* out, float seed) {
    float a = seed + threadIdx.x, b = 1.0001f, c = 0.5f;
    long long r0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 64; ++u) a = __builtin_fmaf(a, b, c);
    }
    long long c1 = clock64(), r1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 3 + 0] = c1 - c0; out[blockIdx.x * 3 + 1] = r1 - r0; }
    if (a == 12345.f) out[0] = 0;
}
int main() {
    long long* d; hipMalloc(&d, 4096 * 3 * 8);
    long long h[3];
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("wall clock rate %d kHz, shader clock attr %d kHz\n", rate, clk);
    for (int rep = 0; rep < 3; ++rep)
    for (int g : {8, 64, 256, 1024}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        busy<<<g, 512>>>(20000, d, 1.0f);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        double mhz = (double)h[0] / ((double)h[1] / (rate * 1e3)) / 1e6;
        // per wave: 20000*64 dependent FMAs; 2 waves per SIMD
        printf("grid %4d: %.3f ms, clock64 delta %lld, wall delta %lld -> %.0f MHz (if clock64 counts shader cycles); %.2f cyc/FMA-instr/wave\n",
               g, ms, h[0], h[1], mhz, (double)h[0] / (20000.0 * 64));
    }
    return 0;
}
