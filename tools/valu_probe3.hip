// valu_probe3: which candidate ops for the FPS min/max step run in the fast VALU class on gfx950?
// 1 block, 4 waves/SIMD, 8 independent chains; cycles per wave-instr per SIMD over all waves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define K(NAME, ASM) \
__global__ __launch_bounds__(1024) void NAME(int iters, long long* out, float seed) { \
    float a[8]; for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x; \
    float b = seed * 1.0001f, c = seed * 0.5f; \
    long long c0 = clock64(); \
    for (int it = 0; it < iters; ++it) { \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) { \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c) : "vcc"); } } \
    long long c1 = clock64(); float s = 0; for (int i = 0; i < 8; ++i) s += a[i]; \
    if ((threadIdx.x & 63) == 0) { atomicMin((unsigned long long*)&out[0], (unsigned long long)c0); atomicMax((unsigned long long*)&out[1], (unsigned long long)c1); } \
    if (s == 1.2345f) out[2] = 0; }
K(k_min32, "v_min_f32_e32 %0, %1, %0")
K(k_min64, "v_min_f32_e64 %0, %0, %1")
K(k_max32, "v_max_f32_e32 %0, %1, %0")
K(k_min3, "v_min3_f32 %0, %0, %1, %2")
K(k_max3f, "v_max3_f32 %0, %0, %1, %2")
K(k_med3, "v_med3_f32 %0, %0, %1, %2")
K(k_maxi, "v_max_i32_e32 %0, %1, %0")
K(k_maxu, "v_max_u32_e32 %0, %1, %0")
K(k_and, "v_and_b32_e32 %0, %1, %0")
K(k_addu, "v_add_u32_e32 %0, %1, %0")
K(k_mov, "v_mov_b32_e32 %0, %1")
K(k_cmp, "v_cmp_lt_f32_e32 vcc, %1, %0")
K(k_cnd, "v_cndmask_b32_e32 %0, %1, %0, vcc")
K(k_add, "v_add_f32_e32 %0, %1, %0")
K(k_sub3, "v_sub_f32_e64 %0, %0, %1")
K(k_fma, "v_fma_f32 %0, %0, %1, %2")
K(k_fmac, "v_fmac_f32_e32 %0, %1, %2")
K(k_fmasame, "v_fma_f32 %0, %1, %1, %0")
K(k_mulsame, "v_mul_f32_e32 %0, %0, %0")
K(k_pkmin16, "v_pk_min_f16 %0, %0, %1")
typedef void (*KF)(int, long long*, float);
void run(const char* name, KF f, long long* d) {
    long long init[2] = {0x7fffffffffffffffLL, 0};
    hipMemcpy(d, init, 16, hipMemcpyHostToDevice);
    f<<<1, 1024>>>(2000, d, 1.5f);
    hipDeviceSynchronize();
    long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-28s %.2f cyc per wave-instr per SIMD\n", name, (double)(h[1] - h[0]) / (2000.0 * 64 * 4));
}
int main() {
    long long* d; hipMalloc(&d, 64);
    run("v_min_f32_e32", k_min32, d); run("v_min_f32_e64", k_min64, d); run("v_max_f32_e32", k_max32, d);
    run("v_min3_f32", k_min3, d); run("v_max3_f32", k_max3f, d); run("v_med3_f32", k_med3, d);
    run("v_max_i32", k_maxi, d); run("v_max_u32", k_maxu, d); run("v_and_b32", k_and, d); run("v_add_u32", k_addu, d);
    run("v_mov_b32", k_mov, d); run("v_cmp_lt_f32 vcc", k_cmp, d); run("v_cndmask_b32", k_cnd, d);
    run("v_add_f32_e32", k_add, d); run("v_sub_f32_e64", k_sub3, d); run("v_fma_f32 (a,b,c)", k_fma, d);
    run("v_fmac_f32", k_fmac, d); run("v_fma_f32 d=b*b+d", k_fmasame, d); run("v_mul_f32 a*a", k_mulsame, d);
    run("v_pk_min_f16", k_pkmin16, d);
    return 0;
}
