// common.h -- shared device helpers for the gfx950 kernels of libgspn_hip.so.
// CDNA4 only: wave = 64 lanes, DPP row operations of the GFX9 family, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gspn_hip.h"

#ifndef GSPN_DIST_POLICY
#define GSPN_DIST_POLICY 2
#endif

#define GSPN_WAVE 64
#ifndef GSPN_GEOM_CLAIM_LDS_DEFAULT
#define GSPN_GEOM_CLAIM_LDS_DEFAULT 4
#endif

// Squared distance exactly as the nvcc-built reference kernels evaluate
//   (x2-x1)*(x2-x1)+(y2-y1)*(y2-y1)+(z2-z1)*(z2-z1)          (tf_sampling_g.cu:142, tf_grouping_g.cu:27)
// under the default --fmad=true.  The translation unit is compiled with -ffp-contract=off so the
// only fused operations are the ones written here.  Same switch as oracle/gspn_oracle.c.
__device__ __forceinline__ float dist2_cuda(float a, float b, float c) {
#if GSPN_DIST_POLICY == 2
    return __builtin_fmaf(c, c, __builtin_fmaf(a, a, b * b));
#elif GSPN_DIST_POLICY == 1
    return __builtin_fmaf(c, c, __builtin_fmaf(b, b, a * a));
#elif GSPN_DIST_POLICY == 3
    // what hipcc (default and -ffp-contract=fast) makes of the reference's expression on gfx950: the left multiply fused, then a plain
    // add -- the form under which this library is BIT-EQUAL to the reference's own sources as compiled here (oracle/_ref, tests/test_gpu_policy3.py)
    return __builtin_fmaf(a, a, b * b) + c * c;
#else
    return (a * a + b * b) + c * c;
#endif
}
// Host-compiled reference loops (g++ -O2, no FMA): tf_interpolate.cpp:74
__device__ __forceinline__ float dist2_host(float a, float b, float c) { return (a * a + b * b) + c * c; }

// v_min_f32 without the canonicalising v_max the compiler puts in front of fminf() when it cannot
// prove an operand is not a signalling NaN (one extra VALU per point in the FPS loop).  Inputs
// here are always finite results of arithmetic.  (CUDA min(float,float), tf_sampling_g.cu:143.)
__device__ __forceinline__ float vmin_f32(float a, float b) {
    float r;
    asm("v_min_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---- DPP helpers (row = 16 lanes) ---------------------------------------------------------
#define DPP_QUAD_XOR1 0xB1     // quad_perm:[1,0,3,2]
#define DPP_QUAD_XOR2 0x4E     // quad_perm:[2,3,0,1]
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
// max over each aligned group of 4/8/16 lanes, result in every lane of the group
__device__ __forceinline__ int row_max_i32(int v) {
    v = max(v, dpp_i32<DPP_QUAD_XOR1>(v));
    v = max(v, dpp_i32<DPP_QUAD_XOR2>(v));
    v = max(v, dpp_i32<DPP_ROW_HALF_MIRROR>(v));
    v = max(v, dpp_i32<DPP_ROW_MIRROR>(v));
    return v;
}
// bitwise OR over the 16 lanes of a DPP row (every lane of the row gets the result)
__device__ __forceinline__ int row_or_i32(int v) {
    v |= dpp_i32<DPP_QUAD_XOR1>(v);
    v |= dpp_i32<DPP_QUAD_XOR2>(v);
    v |= dpp_i32<DPP_ROW_HALF_MIRROR>(v);
    v |= dpp_i32<DPP_ROW_MIRROR>(v);
    return v;
}
__device__ __forceinline__ int oct_max_i32(int v) {
    v = max(v, dpp_i32<DPP_QUAD_XOR1>(v));
    v = max(v, dpp_i32<DPP_QUAD_XOR2>(v));
    v = max(v, dpp_i32<DPP_ROW_HALF_MIRROR>(v));
    return v;
}
// wave-uniform max of a per-lane int (result is an SGPR value)
__device__ __forceinline__ int wave_max_i32(int v) {
    v = row_max_i32(v);
    int a = __builtin_amdgcn_readlane(v, 0);
    int b = __builtin_amdgcn_readlane(v, 16);
    int c = __builtin_amdgcn_readlane(v, 32);
    int d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}
__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

static inline int gspn_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// Straggler experiment: a geometry kernel that leaves part of its CU's 160 KiB of LDS free lets a workgroup of the
// layers' persistent grids land on that CU, where it shares the SIMDs with the geometry waves and finishes last.  GSPN_GEOM_CLAIM_LDS is a
// bit mask (1 = fps_cell_kernel, 2 = fps_small_kernel, 4 = csr_build_lds_kernel, 8 = fps_resident_kernel) of kernels that then ask for
// the whole CU's LDS.  Read once.  Returns the dynamic LDS size to launch with (>= dyn), or dyn if the bit is not set / on any error.
#include <stdlib.h>
static inline int gspn_claim_lds_mask() {
    static const int v = getenv("GSPN_GEOM_CLAIM_LDS") ? atoi(getenv("GSPN_GEOM_CLAIM_LDS")) : GSPN_GEOM_CLAIM_LDS_DEFAULT;
    return v;
}
static inline size_t gspn_claim_lds(int bit, const void* fn, size_t dyn) {
    if (!(gspn_claim_lds_mask() & bit)) return dyn;
    hipFuncAttributes a;
    if (hipFuncGetAttributes(&a, fn) != hipSuccess) return dyn;
    const size_t full = (size_t)160 * 1024;
    if (a.sharedSizeBytes + dyn >= full) return dyn;
    const size_t want = full - a.sharedSizeBytes;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want) != hipSuccess) { (void)hipGetLastError(); return dyn; }
    return want;
}

// CUs the persistent / statically partitioned kernels plan for.  MI355X has 256 (8 XCDs x 32; workgroups go round-robin to the XCDs);
// the schedule this library is built around keeps up to 16 of them busy with farthest point sampling of the NEXT batches on side
// streams (one CU per scene, two streams: 2 per XCD), plus the short-lived 8-workgroup FPS launches of the smaller levels.  A grid sized
// for exactly the CUs that happen to be free runs its last workgroups as a second wave, so the plan leaves 4 CUs per XCD out.
// Measured on the full step (bench.py, ms per step): 256 -> 3.74, 240 -> 3.68, 224 -> 3.56, 216 -> 3.58, 208 -> 3.60, 192 -> 3.64.
// Re-swept on round 4's kernels (tools/plan_cus_sweep.sh): 216 -> 1.752, 224 -> 1.740, 232 -> 1.763, 240 -> 1.773 (the layers alone: 1.532 / 1.527 / 1.517 / 1.509).
#ifndef GSPN_PLAN_CUS
#define GSPN_PLAN_CUS 224
#endif

// grid for a grid-stride kernel over `total` items: enough blocks to fill 256 CUs several times
// over, capped so very large problems loop instead of launching millions of tiny blocks
static inline unsigned grid_for(long total, int block) {
    long g = (total + block - 1) / block;
#ifdef GSPN_GRID_ROUND      // (r06 experiment) whole multiples of the planned CU count: no workgroup of an elementwise kernel is a 'fifth on a CU that was given four'
    const long cap = (long)GSPN_PLAN_CUS * 16;
    if (g > cap) g = cap;
    if (g > GSPN_PLAN_CUS) g = g / GSPN_PLAN_CUS * GSPN_PLAN_CUS;
    return (unsigned)(g < 1 ? 1 : g);
#else
    const long cap = 256L * 16;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
#endif
}
