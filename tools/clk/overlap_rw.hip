// overlap.hip with a write stream: per iteration a workgroup reads RB KB and writes WB KB (float4 per lane) around M MFMAs per wave.
// Does matrix work hide under a read + write stream as it does under a read stream?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NR, int NW, int M, bool DATA = false, int LDSB = 0>      // LDSB: 1 = the B operand of every MFMA is read from LDS (ds_read_b32), 2 = + 4 v_fma per MFMA; float4 loads / stores per lane per iteration; DATA: MFMA operands come from the loaded (random) data
__global__ __launch_bounds__(256) void k(const float4* __restrict__ src, float4* __restrict__ dst, long iters, float* out) {
    __shared__ float sB[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sB[i] = i * 1e-4f;
    __syncthreads();
    f32x16 acc = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    float vv[4] = {a, b, a + b, a - b};
    const float4* p = src + (size_t)blockIdx.x * iters * 256 * (NR > 0 ? NR : 1) + threadIdx.x;
    float4* d = dst + (size_t)blockIdx.x * iters * 256 * (NW > 0 ? NW : 1) + threadIdx.x;
    float4 r[NR > 0 ? NR : 1], q[NR > 0 ? NR : 1];
    float s = 0.f;
    for (int j = 0; j < NR; ++j) r[j] = p[j * 256];
    for (long i = 0; i < iters; ++i) {
        if (NR) {
            const float4* pn = p + (i + 1 < iters ? (i + 1) * 256 * NR : 0);
            for (int j = 0; j < NR; ++j) q[j] = pn[j * 256];
        }
#pragma unroll
        for (int u = 0; u < M; ++u) {
            if (DATA && NR) { const float* rf = reinterpret_cast<const float*>(r); a = rf[u % (4 * NR)]; b = rf[(u + 1) % (4 * NR)]; }
            if (LDSB) b = sB[(threadIdx.x & 31) + 64 * u + ((int)i & 1)];
            if (LDSB == 2) { for (int q = 0; q < 4; ++q) vv[q] = __builtin_fmaf(vv[q], 1.0001f, 1e-3f); }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        for (int j = 0; j < NR; ++j) { s += r[j].x + r[j].y + r[j].z + r[j].w; r[j] = q[j]; }
        for (int j = 0; j < NW; ++j) d[(size_t)i * 256 * NW + j * 256] = make_float4(acc[j & 15], s, a, b);
    }
    if (acc[0] + s + vv[0] + vv[1] + vv[2] + vv[3] == 123.456f) out[0] = acc[1];
}
template <int NR, int NW, int M, bool DATA = false, int LDSB = 0> static void run(const char* name, const float4* src, float4* dst, int wgs, long iters, float* dout) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NR, NW, M, DATA, LDSB>), dim3(wgs), dim3(256), 0, 0, src, dst, iters, dout);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<NR, NW, M, DATA, LDSB>), dim3(wgs), dim3(256), 0, 0, src, dst, iters, dout);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double rb = (double)wgs * iters * 256 * NR * 16, wb = (double)wgs * iters * 256 * NW * 16, fl = (double)wgs * iters * 4 * M * 4096;
    printf("%-34s %8.3f ms   read %5.2f + write %5.2f = %5.2f TB/s   %6.1f TF\n", name, ms, rb / ms / 1e9, wb / ms / 1e9, (rb + wb) / ms / 1e9, fl / ms / 1e9);
}
int main() {
    const long bytes = 1L << 30;
    float4 *src, *dst; float* dout;
    (void)hipMalloc(&src, bytes);
    {   // random finite floats (constant operands would understate the matrix units' switching power)
        unsigned* h = (unsigned*)malloc(bytes);
        unsigned x = 12345u;
        for (long i = 0; i < bytes / 4; ++i) { x = x * 1664525u + 1013904223u; h[i] = 0x3f000000u | (x >> 9); if (x & 256) h[i] |= 0x80000000u; }
        (void)hipMemcpy(src, h, bytes, hipMemcpyHostToDevice); free(h);
    } (void)hipMalloc(&dst, bytes); (void)hipMalloc(&dout, 64);
    const int wgs = 1024;
    const long iters = bytes / 16 / 256 / 4 / wgs;            // 4 float4 per lane per iteration at most
    printf("-- %d workgroups, %ld iterations\n", wgs, iters);
    run<4, 0, 0>("read 16 KB", src, dst, wgs, iters, dout);
    run<0, 4, 0>("write 16 KB", src, dst, wgs, iters, dout);
    run<2, 2, 0>("read 8 + write 8 KB", src, dst, wgs, iters, dout);
    run<0, 0, 16>("16 MFMA", src, dst, wgs, iters, dout);
    run<4, 0, 16>("read 16 KB + 16 MFMA", src, dst, wgs, iters, dout);
    run<0, 4, 16>("write 16 KB + 16 MFMA", src, dst, wgs, iters, dout);
    run<2, 2, 16>("read 8 + write 8 KB + 16 MFMA", src, dst, wgs, iters, dout);
    run<2, 2, 8>("read 8 + write 8 KB + 8 MFMA", src, dst, wgs, iters, dout);
    run<2, 4, 8>("read 8 + write 16 KB + 8 MFMA", src, dst, wgs, iters, dout);
    run<4, 0, 16, true>("read 16 KB + 16 MFMA on the data", src, dst, wgs, iters, dout);
    run<2, 2, 16, true>("read 8 + write 8 KB + 16 MFMA on the data", src, dst, wgs, iters, dout);
    run<2, 2, 8, true>("read 8 + write 8 KB + 8 MFMA on the data", src, dst, wgs, iters, dout);
    run<0, 0, 16, false, 1>("16 MFMA, B from LDS", src, dst, wgs, iters, dout);
    run<2, 2, 16, true, 1>("r8 + w8 KB + 16 MFMA, B from LDS", src, dst, wgs, iters, dout);
    run<0, 0, 16, false, 2>("16 MFMA, B from LDS, 4 v_fma each", src, dst, wgs, iters, dout);
    run<2, 2, 16, true, 2>("r8 + w8 + 16 MFMA, LDS, 4 v_fma each", src, dst, wgs, iters, dout);
    return 0;
}
