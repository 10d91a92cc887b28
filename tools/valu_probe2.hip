// valu_probe2: dependent-issue latency of packed/scalar fp32 VALU ops on gfx950.
// NCH independent dependency chains per wave, 4 waves per SIMD (1024-thread blocks).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int OP, int NCH>
__global__ __launch_bounds__(1024) void k(int iters, long long* out, float seed) {
    float a[8]; v2f p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = v2f{a[i], a[i] + 1}; }
    float b = seed * 1.0001f, c = seed * 0.5f; v2f pb = {b, b}, pc = {c, c};
    long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 64 / NCH; ++u) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
                if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
                if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[i]) : "s"(pb));
                if (OP == 4) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i]));
            }
        }
    }
    long long c1 = clock64();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
    if (s == 1.2345f) out[0] = 0;
}
template <int OP, int NCH>
void run(const char* name, long long* d) {
    k<OP, NCH><<<8, 1024>>>(2000, d, 1.5f);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-22s chains/wave=%d, 4 waves/SIMD: %.2f cyc per instr per WAVE (%.2f per SIMD)\n", name, NCH, (double)h / (2000.0 * 64), (double)h / (2000.0 * 64 * 4));
}
int main() {
    long long* d; hipMalloc(&d, 64 * 8);
    run<0,1>("v_fma_f32", d); run<0,2>("v_fma_f32", d); run<0,4>("v_fma_f32", d);
    run<1,1>("v_pk_fma_f32", d); run<1,2>("v_pk_fma_f32", d); run<1,4>("v_pk_fma_f32", d); run<1,8>("v_pk_fma_f32", d);
    run<2,1>("v_pk_add_f32", d); run<2,2>("v_pk_add_f32", d); run<2,4>("v_pk_add_f32", d);
    run<3,1>("v_pk_add_f32 sgpr", d); run<3,4>("v_pk_add_f32 sgpr", d); run<3,8>("v_pk_add_f32 sgpr", d);
    run<4,1>("v_min_f32", d); run<4,4>("v_min_f32", d);
    run<5,1>("v_pk_mul_f32", d); run<5,4>("v_pk_mul_f32", d);
    return 0;
}
