// mfma_probe: how do fp32 32x32x2 MFMAs overlap with VALU / LDS work of the same and of other waves on one SIMD?
// 1 block of W*4 waves (W per SIMD); each iteration = NM independent-accumulator MFMAs + NV VALU ops (+ NL ds_reads).
// Reports cycles per iteration per SIMD (first start .. last end over all waves), and the MFMA-only bound NM*64*W.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NM, int NV, int NL, bool DEP>
__global__ __launch_bounds__(1024) void probe(int iters, long long* out, float seed) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = seed + i;
    __syncthreads();
    f32x16 acc[NM];
    for (int m = 0; m < NM; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
    float a = seed + threadIdx.x, b = seed * 0.5f;
    const float* lp = lds + (threadIdx.x & 1023);
    long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        float x = a, y = b;
#pragma unroll
        for (int l = 0; l < NL; ++l) { float t; asm volatile("ds_read_b32 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"((int)(size_t)lp * 0 + (int)((threadIdx.x & 1023) * 4)), "n"(0)); x += t; }
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(b), "v"(a));
        if (DEP) { x = v[0]; y = v[1]; }                 // MFMA operands produced by the VALU work of this iteration
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[m], 0, 0, 0);
    }
    long long c1 = clock64();
    float s = 0; for (int m = 0; m < NM; ++m) s += acc[m][0]; for (int i = 0; i < 8; ++i) s += v[i];
    if ((threadIdx.x & 63) == 0) { atomicMin((unsigned long long*)&out[0], (unsigned long long)c0); atomicMax((unsigned long long*)&out[1], (unsigned long long)c1); }
    if (s == 1.2345f) out[2] = 0;
}
template <int NM, int NV, int NL, bool DEP>
void run(long long* d) {
    for (int W = 1; W <= 3; ++W) {
        long long init[2] = {0x7fffffffffffffffLL, 0};
        hipMemcpy(d, init, 16, hipMemcpyHostToDevice);
        const int iters = 2000;
        probe<NM, NV, NL, DEP><<<1, 256 * W>>>(iters, d, 1.5f);
        hipDeviceSynchronize();
        long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        double cyc = (double)(h[1] - h[0]) / iters;
        printf("NM=%d NV=%2d NL=%d dep=%d  W=%d waves/SIMD: %7.1f cyc/iter  (MFMA-only bound %d, pipe util %.2f)\n", NM, NV, NL, (int)DEP, W, cyc, NM * 64 * W, NM * 64.0 * W / cyc);
    }
}
int main() {
    long long* d; hipMalloc(&d, 64);
    run<2, 0, 0, false>(d); run<2, 8, 0, false>(d); run<2, 16, 0, false>(d); run<2, 16, 0, true>(d); run<2, 16, 3, true>(d);
    run<2, 32, 0, true>(d); run<4, 32, 0, true>(d); run<8, 40, 0, true>(d); run<1, 8, 0, true>(d);
    return 0;
}
