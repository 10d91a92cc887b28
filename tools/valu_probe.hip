// valu_probe: issue rate of the VALU ops the FPS loop is made of (gfx950), 1/2/4 waves per SIMD,
// 8 independent dependency chains per wave.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ void k(int iters, long long* out, float seed) {
    float a[8]; v2f p[8]; int ii[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = v2f{a[i], a[i] + 1}; ii[i] = (int)a[i]; }
    float b = seed * 1.0001f, c = seed * 0.5f; v2f pb = {b, b}, pc = {c, c};
    long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#define ONE(i) \
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); \
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc)); \
            if (OP == 2) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
            if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb)); \
            if (OP == 4) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
            if (OP == 5) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(ii[i]) : "v"(ii[(i + 1) & 7]), "v"(ii[(i + 2) & 7])); \
            if (OP == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb)); \
            if (OP == 7) asm volatile("v_subrev_f32 %0, %1, %0" : "+v"(a[i]) : "s"(seed)); \
            if (OP == 8) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[i])); \
            if (OP == 9) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(b));
            REP8(ONE)
        }
    }
    long long c1 = clock64();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + ii[i];
    if ((threadIdx.x & 63) == 0) { atomicMin((unsigned long long*)&out[0], (unsigned long long)c0); atomicMax((unsigned long long*)&out[1], (unsigned long long)c1); }
    if (s == 1.2345f) out[2] = 0;
}
template <int OP>
void run(const char* name, long long* d) {
    for (int threads : {256, 512, 768, 1024}) {
        long long init[2] = {0x7fffffffffffffffLL, 0};
        hipMemcpy(d, init, 16, hipMemcpyHostToDevice);
        k<OP><<<1, threads>>>(2000, d, 1.5f);
        hipDeviceSynchronize();
        long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        int wps = threads / 256;
        printf("%-14s %d wave/SIMD: %.2f cyc per wave-instr per SIMD (all waves, first start to last end)\n", name, wps, (double)(h[1] - h[0]) / (2000.0 * 64 * wps));
    }
}
int main() {
    long long* d; hipMalloc(&d, 64 * 8);
    run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_sub_f32", d); run<3>("v_pk_add_f32", d);
    run<4>("v_min_f32", d); run<5>("v_max3_i32", d); run<6>("v_pk_mul_f32", d); run<7>("v_subrev sgpr", d);
    run<8>("v_mul_f32", d); run<9>("v_fmac_f32", d);
    return 0;
}
