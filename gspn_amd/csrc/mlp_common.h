// mlp_common.h -- the few definitions shared by the translation units of the shared-MLP kernels (mlp.hip, mlp_short.hip).
#pragma once
#include "common.h"
#include <stdlib.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// MFMA 32x32x2 f32 fragment layout (wave64):  A: lane l holds A[i=l&31][k=l>>5];  B: lane l holds B[k=l>>5][j=l&31];
// C/D: 16 registers, reg r -> row (r&3)+8*(r>>2)+4*(l>>5), col l&31.
__device__ __forceinline__ int c_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// THE ReLU mask of a batch-normalised element, the forward's own expression: relu(y*scale + shift) is open iff the two-rounding value
// round(round(y*scale) + shift) is positive (-ffp-contract=off: no fused form).  Every backward kernel forms its mask with this, so the
// BN reductions (r0, r1), the coefficients and dY are built from the same set of live elements as the forward's activations.
// (round(t + shift) > 0  <=>  t > -shift exactly: a multiply and a compare, the cost of the fused form.)
__device__ __forceinline__ bool relu_open(float y, float sc, float sh) { return y * sc > -sh; }
__device__ __forceinline__ float act1(float v, bool act, float sc, float sh) {
    if (act) { v = v * sc + sh; v = v > 0.f ? v : 0.f; }     // relu(x*scale+shift): two roundings, like tf.nn.batch_normalization
    return v;
}

// max-pool over groups of 32 rows folded into the forward epilogue: per (group, channel) the largest raw output and the row it is first reached in
struct PoolOut { float* vmax; int* amax; };

static inline int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static inline bool vec_ok(const void* p, int ld) { return (ld % 4 == 0) && (((uintptr_t)p) % 16 == 0); }

// rows at or below which the FORWARD of a layer takes the split-K kernel of mlp_short.hip.  Measured (tools/r05_short_ab.sh, graph-timed,
// us, round-4 kernel -> split-K): 4096 x 256 -> 128 9.9 -> 7.1, 4096 x 128 -> 128 6.0 -> 4.4, 4096 x 384 -> 256 13.4 -> 13.3; at 16384 rows
// 12.4 -> 13.9 / 6.9 -> 7.4 and at 32768 rows 17.5 -> 19.3 / 32.4 -> 38.4: with K = 128 a 32-row tile is 16 MFMAs per wave behind a full
// global-load round trip, and the LDS footprint (partials + staged A) keeps 4 workgroups per CU -- not enough to cover it.  (A trivial
// dependent node of the captured chain costs 1.6 us, tools/r05_launch_floor.py; 16384 x 64 -> 64 reads 4.9 us.)
#ifndef GSPN_SHORT_ROWS
#define GSPN_SHORT_ROWS 8192
#endif
// partial-statistics rows of a short layer's forward launch (one per 32-row tile, at most 512)
static inline long short_fwd_parts(long rows) { const long t = rows / 32; return t < 512 ? t : 512; }

// (r05 also measured a register-streaming pass A for these layers -- a wash against the LDS-DMA streaming kernel: profiles/r05_experiments.txt item 2;
// the kernel lives in tools/patches/r06_pruned_alternates.patch)
// mlp_short.hip: forward of a short layer (returns false when the shape is not one it takes: the caller goes on to the general kernels)
bool gspn_fwd_short_go(long rows, int cin, int cout, const float* X, int ldx, const float* in_scale, const float* in_shift, const float* W, const float* bias,
                       float* Y, int ldy, float* stats, unsigned nparts, PoolOut po, hipStream_t st);
