// fps_common.h -- pieces shared by the farthest-point-sampling kernels (sampling.hip: one CU per scene;
// sampling_multi.hip: several CUs per scene).
#pragma once
#include "common.h"

#define FPS_T 1024
#define FPS_W (FPS_T / 64)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define NEG_ONE_BITS ((int)0xBF800000)

// dist2_cuda (common.h) on two points at once: the same contraction policy, spelled with element-wise fused multiply-adds
// (v_pk_fma_f32 / v_pk_mul_f32).  EVERY squared distance of the FPS kernels goes through this or through dist2_cuda, so
// -DGSPN_DIST_POLICY=0/1/2 is the whole change (tests/test_gpu_policy.py builds all three and compares with the oracle built alike).
__device__ __forceinline__ v2f dist2_cuda_v2(v2f a, v2f b, v2f c) {
#if GSPN_DIST_POLICY == 2
    v2f d = b * b;
    d = __builtin_elementwise_fma(a, a, d);
    return __builtin_elementwise_fma(c, c, d);
#elif GSPN_DIST_POLICY == 1
    v2f d = a * a;
    d = __builtin_elementwise_fma(b, b, d);
    return __builtin_elementwise_fma(c, c, d);
#elif GSPN_DIST_POLICY == 3
    v2f d = b * b;
    d = __builtin_elementwise_fma(a, a, d);
    return d + c * c;
#else
    return (a * a + b * b) + c * c;
#endif
}

template <int P>
struct FpsGroup {
    static constexpr int G = (P >= 8) ? 8 : P;   // points per resolve group
    static constexpr int NG = P / G;
};

__device__ __forceinline__ int vmax3_i32(int a, int b, int c) {
    int r;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// reference tie rank of original index k: the block arg-max of tf_sampling_g.cu:151-165 prefers (k mod 512) asc, then k asc
__host__ __device__ __forceinline__ unsigned fps_tie_rank(int k) { return ((unsigned)(k & 511) << 22) | (unsigned)(k >> 9); }
__host__ __device__ __forceinline__ int fps_tie_rank_inv(unsigned r) { return (int)(((r & 0x3FFFFFu) << 9) | (r >> 22)); }

// Spatial pre-pass shared by both cell kernels (defined in sampling.hip): every scene is cut into `ncell` cells of csz points
// (consecutive runs of the 12-bit Morton voxel order), reference tie rank ascending inside a cell.
//   perm (b,n) i32: sorted position -> original index (the buffer doubles as the key scratch of the counting sort)
//   sxyz (b,n,3) f32: coordinates in sorted order
int gspn_fps_prepass_cells(int b, int n, int ncell, int csz, const float* inp, int* perm, float* sxyz, hipStream_t st, int* vorder = nullptr);
