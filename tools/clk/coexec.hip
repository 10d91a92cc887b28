// do fp32 MFMAs and fp32 VALU instructions of DIFFERENT waves on one SIMD execute concurrently on gfx950?  workgroups of 8 waves: waves 0-3
// (one per SIMD) run a chain of MFMAs, waves 4-7 (the second wave of each SIMD) run independent v_fma chains; time each role alone and both.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int ROLES>   // bit 0: MFMA waves work, bit 1: VALU waves work
__global__ __launch_bounds__(512) void k(int iters, float* out) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (!(ROLES & 1)) return;
        f32x16 acc = {0};
        float a = threadIdx.x * 1e-3f, b = 1.0f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (acc[0] == 123.456f) out[0] = acc[1];
    } else {
        if (!(ROLES & 2)) return;
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = threadIdx.x * 1e-3f + u;
        const float m = 1.0001f, c = 1e-3f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r)                       // 16 x 16 = 256 v_fma per iteration = 1024 issue cycles, like 16 MFMAs
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = __builtin_fmaf(v[u], m, c);
        }
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
        if (s == 123.456f) out[1] = s;
    }
}
template <int ROLES> static float run(int iters, float* dout) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<ROLES>), dim3(256), dim3(512), 0, 0, iters, dout);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<ROLES>), dim3(256), dim3(512), 0, 0, iters, dout);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}
int main() {
    float* dout; (void)hipMalloc(&dout, 64);
    const int iters = 4000;
    const float a = run<1>(iters, dout), b = run<2>(iters, dout), c = run<3>(iters, dout);
    printf("MFMA waves alone %.3f ms (%.1f TF)   VALU waves alone %.3f ms (%.1f TFLOP/s of v_fma)   both %.3f ms  -> %s\n", a,
           256.0 * 4 * iters * 16 * 4096 / a / 1e9, b, 256.0 * 4 * iters * 256 * 128 / b / 1e9, c,
           c < 0.75 * (a + b) ? "they overlap" : "they add");
    return 0;
}
