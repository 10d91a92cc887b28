// can fp32 MFMA work hide under an HBM stream on this chip?  one kernel, three modes: stream only, MFMAs only, both (the loads of iteration
// i+1 are in flight during the MFMAs of iteration i).  If both ~ max(stream, mfma) the hardware overlaps them and a kernel that shows
// stream + mfma has a structural problem; if both ~ sum, it is the chip (issue, power, clocks).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <bool LOADS, int M>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ src, long n4_per_wg, float* out, unsigned long long* clk) {
    f32x16 acc = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    const float4* p = src + (size_t)blockIdx.x * n4_per_wg + threadIdx.x;
    const long iters = n4_per_wg / (256 * 4);
    float4 r[4], q[4];
    float s = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (LOADS) for (int j = 0; j < 4; ++j) r[j] = p[j * 256];
    for (long i = 0; i < iters; ++i) {
        if (LOADS) {
            const float4* pn = p + (i + 1 < iters ? (i + 1) * 1024 : 0);
            for (int j = 0; j < 4; ++j) q[j] = pn[j * 256];
        }
#pragma unroll
        for (int u = 0; u < M; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        if (LOADS) {
            for (int j = 0; j < 4; ++j) { s += r[j].x + r[j].y + r[j].z + r[j].w; r[j] = q[j]; }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc[0] + s == 123.456f) out[0] = acc[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}
template <bool LOADS, int M> static void run(const char* name, const float4* src, long n4, int wgs, float* dout, unsigned long long* dclk) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const long per = n4 / wgs / 1024 * 1024;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<LOADS, M>), dim3(wgs), dim3(256), 0, 0, src, per, dout, dclk);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<LOADS, M>), dim3(wgs), dim3(256), 0, 0, src, per, dout, dclk);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    unsigned long long h; (void)hipMemcpy(&h, dclk, 8, hipMemcpyDeviceToHost);
    const double bytes = LOADS ? (double)per * wgs * 16 : 0, flops = (double)(per / 1024) * wgs * 4 * M * 4096;
    printf("%-28s %8.3f ms   %6.2f TB/s   %6.1f TF   clock %.0f MHz\n", name, ms, bytes / ms / 1e9, flops / ms / 1e9, h / (ms * 1e3));
}
int main() {
    const long n4 = (1L << 30) / 16 * 2;        // 2 GiB
    float4* src; float* dout; unsigned long long* dclk;
    (void)hipMalloc(&src, n4 * 16); (void)hipMemset(src, 0, n4 * 16); (void)hipMalloc(&dout, 64); (void)hipMalloc(&dclk, 16);
    for (int wgs : {512, 768, 1024, 2048}) {
        printf("-- %d workgroups of 256 threads (%d per CU)\n", wgs, wgs / 256);
        // per iteration a workgroup moves 16 KB; M MFMAs per wave = M * 64 cycles on its SIMD
        run<true, 0>("stream only", src, n4, wgs, dout, dclk);
        run<false, 4>("4 MFMA / iteration only", src, n4, wgs, dout, dclk);
        run<true, 4>("stream + 4 MFMA / iteration", src, n4, wgs, dout, dclk);
        run<false, 8>("8 MFMA / iteration only", src, n4, wgs, dout, dclk);
        run<true, 8>("stream + 8 MFMA / iteration", src, n4, wgs, dout, dclk);
        run<false, 16>("16 MFMA / iteration only", src, n4, wgs, dout, dclk);
        run<true, 16>("stream + 16 MFMA / iteration", src, n4, wgs, dout, dclk);
    }
    return 0;
}
