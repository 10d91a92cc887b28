// overlap_rw.hip with the access pattern of a row-tile GEMM kernel: per iteration a WAVE reads a 32-row x 256-byte tile and writes one, with
// 64 MFMAs in between.  LOADP 0: coalesced float4 (lane i reads 16 bytes at 16 i); 1: row-per-lane (lane (m, h) reads the 128 bytes at
// m*256 + h*128 as 8 float4s: every instruction touches 64 cache lines).  STOREP 0: coalesced float4; 1: accumulator layout (32 dword
// stores per tile: lanes 0-31 one 128-byte row segment, lanes 32-63 another).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int LOADP, int STOREP, int M>
__global__ __launch_bounds__(256) void k(const char* __restrict__ src, char* __restrict__ dst, int tiles_per_wave, float* out) {
    f32x16 acc0 = {0}, acc1 = {0};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t gw = (size_t)blockIdx.x * 4 + wave, nw = (size_t)gridDim.x * 4;
    const unsigned lo = LOADP ? (unsigned)((lane & 31) * 256 + (lane >> 5) * 128) : (unsigned)(lane * 16);
    const unsigned ls = LOADP ? 16u : 1024u;
    float4 r[8], q[8];
    auto fetch = [&](float4 (&x)[8], size_t tile) {
        const char* b = src + tile * 8192 + lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = *reinterpret_cast<const float4*>(b + j * ls);
    };
    fetch(r, gw);
    for (int i = 0; i < tiles_per_wave; ++i) {
        const size_t tile = gw + (size_t)i * nw;
        if (i + 1 < tiles_per_wave) fetch(q, tile + nw);
        const float* rf = reinterpret_cast<const float*>(r);
#pragma unroll
        for (int u = 0; u < M / 2; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(rf[u & 31], rf[(u + 7) & 31], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(rf[u & 31], rf[(u + 13) & 31], acc1, 0, 0, 0);
        }
        char* d = dst + tile * 8192;
        if (STOREP) {
            const unsigned so = (unsigned)(4 * (lane >> 5) * 256 + (lane & 31) * 4);
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                *reinterpret_cast<float*>(d + so + ((rr & 3) + 8 * (rr >> 2)) * 256) = acc0[rr];
                *reinterpret_cast<float*>(d + so + ((rr & 3) + 8 * (rr >> 2)) * 256 + 128) = acc1[rr];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                *reinterpret_cast<float4*>(d + lane * 16 + j * 1024) = make_float4(acc0[2 * j], acc0[2 * j + 1], acc1[2 * j], acc1[2 * j + 1]);
        }
        acc0 = (f32x16){0}; acc1 = (f32x16){0};
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = q[j];
    }
    if (acc0[0] == 123.456f) out[0] = acc1[1];
}
template <int LOADP, int STOREP, int M> static void run(const char* name, const char* src, char* dst, int wgs, int tpw, float* dout) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<LOADP, STOREP, M>), dim3(wgs), dim3(256), 0, 0, src, dst, tpw, dout);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<LOADP, STOREP, M>), dim3(wgs), dim3(256), 0, 0, src, dst, tpw, dout);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double by = (double)wgs * 4 * tpw * 8192 * 2, fl = (double)wgs * 4 * tpw * M * 4096;
    printf("%-58s %8.3f ms   %5.2f TB/s   %6.1f TF\n", name, ms, by / ms / 1e9, fl / ms / 1e9);
}
int main() {
    const long bytes = 1L << 30;
    char *src, *dst; float* dout;
    (void)hipMalloc(&src, bytes); (void)hipMalloc(&dst, bytes); (void)hipMalloc(&dout, 64);
    {
        unsigned* h = (unsigned*)malloc(bytes); unsigned x = 12345u;
        for (long i = 0; i < bytes / 4; ++i) { x = x * 1664525u + 1013904223u; h[i] = 0x3f000000u | (x >> 9); }
        (void)hipMemcpy(src, h, bytes, hipMemcpyHostToDevice); free(h);
    }
    for (int wgs : {896, 1024}) {
        const int tpw = (int)(bytes / 8192 / (wgs * 4));
        printf("-- %d workgroups, %d tiles of 32 rows x 256 B per wave (%.0f MB each way)\n", wgs, tpw, wgs * 4.0 * tpw * 8192 / 1e6);
        run<0, 0, 0>("coalesced loads, coalesced stores, no MFMA", src, dst, wgs, tpw, dout);
        run<0, 0, 64>("coalesced loads, coalesced stores, 64 MFMA", src, dst, wgs, tpw, dout);
        run<1, 0, 0>("row-per-lane loads, coalesced stores, no MFMA", src, dst, wgs, tpw, dout);
        run<1, 0, 64>("row-per-lane loads, coalesced stores, 64 MFMA", src, dst, wgs, tpw, dout);
        run<0, 1, 0>("coalesced loads, accumulator-layout stores, no MFMA", src, dst, wgs, tpw, dout);
        run<0, 1, 64>("coalesced loads, accumulator-layout stores, 64 MFMA", src, dst, wgs, tpw, dout);
        run<1, 1, 0>("row-per-lane loads, accumulator-layout stores, no MFMA", src, dst, wgs, tpw, dout);
        run<1, 1, 64>("row-per-lane loads, accumulator-layout stores, 64 MFMA", src, dst, wgs, tpw, dout);
    }
    return 0;
}
