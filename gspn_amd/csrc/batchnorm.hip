// batchnorm.hip -- stand-alone batch normalisation over the rows of a (rows, c) matrix: the device side of
// tf_util.batch_norm_for_conv2d / batch_norm_for_conv1d / batch_norm_for_fc (utils/tf_util.py:515-580, all of them
// batch_norm_template over the leading axes).  The set-abstraction path itself never calls these -- its batch norm rides inside the
// shared-MLP kernels (mlp.hip) -- but they are part of the module the path is a drop-in for, so they run on the same pieces:
//   forward   gspn_bn_colsum (column sums of x and x^2, per-workgroup partial rows)  ->  gspn_bn_finalize_parts (mlp.hip: double
//             sum of the partials, batch / moving statistics, scale and shift)      ->  gspn_bn_apply (y = x*scale + shift)
//   backward  gspn_bn_colsum with dZ (column sums of dz and dz*xhat)                ->  gspn_mlp_bwd_coef (mlp.hip: cA, cB, cC,
//             dgamma, dbeta)                                                        ->  gspn_bn_backward_apply (dx = cA*dz + cB*x + cC)
// All three kernels are plain HBM streams: one read (two in backward) and at most one write per element, lanes along the row so
// that every wave touches whole rows of c contiguous floats.
#include "common.h"

#define BNC_T 256
#define BNC_MAX_PARTS 1024

// lanes: cw = the power of two >= min(c, 256) columns side by side, 256 / cw row lanes; a workgroup owns a contiguous run of rows
__global__ __launch_bounds__(BNC_T) void bn_colsum_kernel(long rows, int c, int cw, const float* __restrict__ X, int ldx,
                                                           const float* __restrict__ dZ, int ldz, const float* __restrict__ mean,
                                                           const float* __restrict__ var, float eps, float* __restrict__ part, long rows_per_block) {
    extern __shared__ float sred[];                   // [2][rl][cw]
    const int t = threadIdx.x;
    const int cl = t % cw, rlane = t / cw, rl = BNC_T / cw;
    const long r0 = blockIdx.x * rows_per_block;
    const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    for (int col0 = 0; col0 < c; col0 += cw) {
        const int col = col0 + cl;
        float a0 = 0.f, a1 = 0.f;
        if (col < c) {
            if (dZ) {
                const float rs = (float)(1.0 / sqrt((double)var[col] + (double)eps)), mr = -mean[col] * rs;
                for (long r = r0 + rlane; r < r1; r += rl) {
                    const float dz = dZ[r * ldz + col];
                    a0 += dz;
                    a1 = __builtin_fmaf(dz, __builtin_fmaf(X[r * ldx + col], rs, mr), a1);      // dz * xhat (as pass B's epilogue forms it)
                }
            } else {
                // mean != NULL here: a PIVOT row -- the sums are of (x - pivot) and its square, so that var = E[d^2] - E[d]^2 does not
                // cancel when |mean| >> std (x = 100 + 1e-3 * noise lost every digit of the variance before)
                const float pv = mean ? mean[col] : 0.f;
                for (long r = r0 + rlane; r < r1; r += rl) {
                    const float x = X[r * ldx + col] - pv;
                    a0 += x;
                    a1 = __builtin_fmaf(x, x, a1);
                }
            }
        }
        sred[rlane * cw + cl] = a0;
        sred[(rl + rlane) * cw + cl] = a1;
        __syncthreads();
        if (rlane == 0 && col < c) {
            float s0 = 0.f, s1 = 0.f;
            for (int q = 0; q < rl; ++q) { s0 += sred[q * cw + cl]; s1 += sred[(rl + q) * cw + cl]; }     // fixed order: deterministic
            part[(size_t)blockIdx.x * 2 * c + col] = s0;
            part[(size_t)blockIdx.x * 2 * c + c + col] = s1;
        }
        __syncthreads();
    }
}
static inline long bnc_parts(long rows) {
    long n = (rows + 63) / 64;                        // at least 64 rows per workgroup
    if (n > BNC_MAX_PARTS) n = BNC_MAX_PARTS;
    return n < 1 ? 1 : n;
}
extern "C" long gspn_bn_colsum_part_floats(long rows, int c) {
    if (rows < 0 || c <= 0) return GSPN_ERR_ARG;
    return bnc_parts(rows) * 2 * c;
}
// part [nparts][2][c] <- per-workgroup column sums over the rows of X (rows, ldx >= c):
//   dZ == NULL : (sum x, sum x^2), or with mean != NULL (a pivot row, e.g. X's own first row) (sum (x - pivot), sum (x - pivot)^2)
//                                                      -- the input of gspn_bn_finalize_parts[_pivot]
//   dZ != NULL : (sum dz, sum dz * xhat), xhat = (x - mean) * rsqrt(var + eps)   -- the input of gspn_mlp_bwd_coef
extern "C" int gspn_bn_colsum(long rows, int c, const float* X, int ldx, const float* dZ, int ldz, const float* mean, const float* var, float eps,
                              float* part, int* nparts_out, void* stream) {
    if (rows <= 0 || c <= 0 || !X || ldx < c || !part || !nparts_out) return GSPN_ERR_ARG;
    if (dZ && (ldz < c || !mean || !var)) return GSPN_ERR_ARG;
    int cw = 1;
    while (cw < c && cw < BNC_T) cw <<= 1;
    long nblk = bnc_parts(rows);
    const long rpb = (rows + nblk - 1) / nblk;
    nblk = (rows + rpb - 1) / rpb;
    hipLaunchKernelGGL(bn_colsum_kernel, dim3((unsigned)nblk), dim3(BNC_T), sizeof(float) * 2 * BNC_T, (hipStream_t)stream, rows, c, cw, X, ldx, dZ, ldz,
                       mean, var, eps, part, rpb);
    *nparts_out = (int)nblk;
    return gspn_launch_status();
}

// out = x*scale + shift (two roundings, like every batch-norm application of this library), optionally through a ReLU
__global__ void bn_apply_kernel(long total, int c, const float* __restrict__ X, int ldx, const float* __restrict__ scale,
                                const float* __restrict__ shift, int relu, float* __restrict__ out, int ldo) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / c;
        const int col = (int)(i - row * c);
        float z = X[row * ldx + col] * scale[col] + shift[col];
        if (relu) z = z > 0.f ? z : 0.f;
        out[row * ldo + col] = z;
    }
}
extern "C" int gspn_bn_apply(long rows, int c, const float* X, int ldx, const float* scale, const float* shift, int relu, float* out, int ldo, void* stream) {
    if (rows < 0 || c <= 0 || ldx < c || ldo < c || !scale || !shift) return GSPN_ERR_ARG;
    const long total = rows * c;
    if (total == 0) return 0;
    if (!X || !out) return GSPN_ERR_ARG;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, total, c, X, ldx, scale, shift, relu, out, ldo);
    return gspn_launch_status();
}

// dx = cA*dz + cB*x + cC  (training-mode batch-norm backward with the coefficients of gspn_mlp_bwd_coef; inference: cA = scale, cB = cC = 0)
__global__ void bn_backward_apply_kernel(long total, int c, const float* __restrict__ dZ, int ldz, const float* __restrict__ X, int ldx,
                                         const float* __restrict__ cA, const float* __restrict__ cB, const float* __restrict__ cC,
                                         float* __restrict__ dX, int lddx) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / c;
        const int col = (int)(i - row * c);
        dX[row * lddx + col] = __builtin_fmaf(cA[col], dZ[row * ldz + col], __builtin_fmaf(cB[col], X[row * ldx + col], cC[col]));
    }
}
extern "C" int gspn_bn_backward_apply(long rows, int c, const float* dZ, int ldz, const float* X, int ldx, const float* cA, const float* cB,
                                      const float* cC, float* dX, int lddx, void* stream) {
    if (rows < 0 || c <= 0 || ldz < c || ldx < c || lddx < c || !cA || !cB || !cC) return GSPN_ERR_ARG;
    const long total = rows * c;
    if (total == 0) return 0;
    if (!dZ || !X || !dX) return GSPN_ERR_ARG;
    hipLaunchKernelGGL(bn_backward_apply_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, total, c, dZ, ldz, X, ldx, cA, cB, cC, dX, lddx);
    return gspn_launch_status();
}
