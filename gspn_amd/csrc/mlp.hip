// mlp.hip -- the shared MLP of utils/tf_util.py:conv2d (1x1 conv + bias + batch-norm + ReLU) and the
// max-pool of utils/pointnet_util.py:123-124, forward and backward, on gfx950 matrix cores.
//
// The reference hands this arithmetic to TensorFlow/cuDNN as separate conv, bias_add, batch_norm,
// relu and reduce_max kernels (tf_util.py:170-183), each a full pass over a (b*m*nsample, C)
// activation tensor.  Here one layer is ONE fp32-MFMA GEMM (v_mfma_f32_32x32x2_f32: exact fp32
// products and fp32 accumulation, the 157 TFLOP/s rate of MI355X for fp32 inputs; there is no
// TF32/xf32 on gfx950 and the 1e-5 parity target rules out bf16):
//     * the previous layer's BN+ReLU is applied while the A operand is staged into LDS,
//     * bias and the per-channel sum / sum-of-squares for training-mode BN come out of the epilogue,
//     * backward is two passes per layer and never materialises dY:
//         pass A  reads (X, Y, dz) once and produces the BN reductions r0, r1 AND the raw weight-gradient
//                 products G1 = A^T.dyh, Gx = A^T.xhat, g3 = A^T.1   (dW is linear in them:
//                 dW = cA (.) (G1 - r0/R g3 1^T - r1/R (.) Gx), so no coefficient is needed up front);
//         pass B  dX = dY.W^T with dY rebuilt on the fly from the coefficients.
// Rows are the long dimension (up to 524288), channels are 6..384: every GEMM is tall and skinny and
// HBM-bound, so the design goal is one read + one write per activation: float4 global loads, register
// prefetch of the next K chunk under the MFMAs of the current one, persistent row tiles.
//
// MFMA 32x32x2 f32 fragment layout (wave64):  A: lane l holds A[i=l&31][k=l>>5];  B: lane l holds
// B[k=l>>5][j=l&31];  C/D: 16 registers, reg r -> row (r&3)+8*(r>>2)+4*(l>>5), col l&31.
#include "mlp_common.h"
#include <type_traits>
#include <utility>
#include <stdlib.h>
#include <stdio.h>
#include <algorithm>

#define TM 128          // rows of C per workgroup (4 waves x 32)
#define TK 32           // K chunk staged per iteration
#define LDT 129         // pitch of a K-major LDS tile [TK][128]: == 1 (mod 32) -> transposing writes and column reads conflict-free

// (c_row, relu_open, act1, PoolOut, env_int, vec_ok: mlp_common.h)

#define MAXCH GSPN_MLP_MAX_CHANNELS      // (1024) per-channel constants are staged in LDS for layers up to this many channels (the 4-level networks of
                        // model_rpointnet.py:109,181 feed fa_layer1 with 256 + 512 = 768 input channels)
// The staged constants live in DYNAMIC shared memory sized for the layer at hand (cpad = chan_pad(channels) floats per array): fixed
// MAXCH-sized arrays cost pass B 20 KB of LDS and a workgroup per CU of occupancy when the limit went from 512 to 1024 channels.
static inline int chan_pad(int c) { return (c + 3) / 4 * 4 + 4; }
// fill dst[0..cpad) with src[0..n) (or `dflt` when src is NULL / beyond n); callers __syncthreads() afterwards
__device__ __forceinline__ void stage_chan(float* dst, const float* __restrict__ src, int n, float dflt, int cpad) {
    for (int i = threadIdx.x; i < cpad; i += blockDim.x) dst[i] = (src && i < n) ? src[i] : dflt;
}
__device__ __forceinline__ float4 lds4(const float* p, int k, int cpad) {      // 4 consecutive constants, index clamped
    return make_float4(p[min(k, cpad - 1)], p[min(k + 1, cpad - 1)], p[min(k + 2, cpad - 1)], p[min(k + 3, cpad - 1)]);
}

// ---- 4-wide row-segment loads --------------------------------------------------------------------------------
// load4_raw issues UNCONDITIONAL loads (indices clamped into the array) so that all of a thread's loads of one stage are
// in flight together; validity is applied afterwards by mask4 (a select), never by a branch around the load.
template <bool VEC>
__device__ __forceinline__ float4 load4_raw(const float* __restrict__ src, long row, int ld, int k, int kvalid) {
    if (VEC) {
        const int kc = k < kvalid ? k : 0;                  // ld % 4 == 0, k % 4 == 0: an aligned quad inside the row
        return *reinterpret_cast<const float4*>(src + row * ld + kc);
    }
    const float* p = src + row * ld;
    const int last = kvalid - 1;
    float4 v;
    v.x = p[min(k + 0, last)];
    v.y = p[min(k + 1, last)];
    v.z = p[min(k + 2, last)];
    v.w = p[min(k + 3, last)];
    return v;
}
__device__ __forceinline__ float4 mask4(float4 v, int k, int kvalid, bool live) {
    v.x = (live && k + 0 < kvalid) ? v.x : 0.f;
    v.y = (live && k + 1 < kvalid) ? v.y : 0.f;
    v.z = (live && k + 2 < kvalid) ? v.z : 0.f;
    v.w = (live && k + 3 < kvalid) ? v.w : 0.f;
    return v;
}

struct Chan4 { float4 sc, sh; };
__device__ __forceinline__ Chan4 load_chan4(const float* scale, const float* shift, int k, int kvalid) {
    Chan4 c;
    c.sc = make_float4(1.f, 1.f, 1.f, 1.f);
    c.sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (scale) {                                             // wave-uniform branch; clamped, unconditional loads inside
        const int last = kvalid - 1;
        c.sc.x = scale[min(k + 0, last)]; c.sh.x = shift[min(k + 0, last)];
        c.sc.y = scale[min(k + 1, last)]; c.sh.y = shift[min(k + 1, last)];
        c.sc.z = scale[min(k + 2, last)]; c.sh.z = shift[min(k + 2, last)];
        c.sc.w = scale[min(k + 3, last)]; c.sh.w = shift[min(k + 3, last)];
    }
    return c;
}
// relu(x*scale+shift) on a raw quad, then validity mask
__device__ __forceinline__ float4 act4(float4 v, bool act, const Chan4& c, int k, int kvalid, bool live) {
    if (act) {
        v.x = act1(v.x, true, c.sc.x, c.sh.x);
        v.y = act1(v.y, true, c.sc.y, c.sh.y);
        v.z = act1(v.z, true, c.sc.z, c.sh.z);
        v.w = act1(v.w, true, c.sc.w, c.sh.w);
    }
    return mask4(v, k, kvalid, live);
}

// raw dz for (row, 4 channels): dense load, or (POOLED) the arg-max offsets + pooled gradient; resolved by dz4_resolve.
// Dense vs pooled is a template parameter: a runtime switch would leave the pooled path's integer division in every kernel
// and split the load burst into basic blocks with waits in between.
struct DzRaw { float4 v; int4 arg; };
template <bool VEC, bool POOLED>
__device__ __forceinline__ DzRaw dz4_raw(const gspn_dy_args& a, long row, int col, int c) {
    DzRaw r;
    r.arg = make_int4(0, 0, 0, 0);
    if constexpr (!POOLED) {
        r.v = load4_raw<VEC>(a.dZ, row, a.ldz, col, c);
    } else {
        // rows < 2^31 (checked by the launcher).  Pool groups of 2^k rows (the usual nsample = 16/32/64) take a shift: a 32-bit integer
        // division is ~25 VALU instructions, and this runs per row quad of every chunk
        const int g = ((a.ns & (a.ns - 1)) == 0) ? ((int)row >> __builtin_ctz(a.ns)) : ((int)row / a.ns);
        const int last = c - 1;
        const int* ar = a.pool_arg + (size_t)g * c;
        const float* dp = a.dPool + (size_t)g * c;
        if (VEC && (c & 3) == 0) {                           // (groups, c) rows are 16-byte aligned: one quad load each
            const int kc = col < c ? col : 0;
            r.arg = *reinterpret_cast<const int4*>(ar + kc);
            r.v = *reinterpret_cast<const float4*>(dp + kc);
        } else {
            r.arg = make_int4(ar[min(col + 0, last)], ar[min(col + 1, last)], ar[min(col + 2, last)], ar[min(col + 3, last)]);
            r.v = make_float4(dp[min(col + 0, last)], dp[min(col + 1, last)], dp[min(col + 2, last)], dp[min(col + 3, last)]);
        }
    }
    return r;
}
template <bool POOLED>
__device__ __forceinline__ float4 dz4_resolve(const gspn_dy_args& a, const DzRaw& r, long row) {
    if constexpr (!POOLED) return r.v;
    const int off = ((a.ns & (a.ns - 1)) == 0) ? ((int)row & (a.ns - 1)) : ((int)row % a.ns);
    float4 v;
    v.x = r.arg.x == off ? r.v.x : 0.f;
    v.y = r.arg.y == off ? r.v.y : 0.f;
    v.z = r.arg.z == off ? r.v.z : 0.f;
    v.w = r.arg.w == off ? r.v.w : 0.f;
    return v;
}

// ---- max-pool over groups of 32 rows folded into the forward epilogue (pointnet_util.py:123-124 with nsample = 32) -------------
// A pool group of 32 consecutive rows is exactly one wave's 32-row MFMA tile, so the group extrema of the raw output y are taken
// from the accumulators, before y ever leaves the registers.  BN+ReLU is monotone per channel (increasing for scale >= 0, decreasing
// for scale < 0), hence max_k relu(scale*y_k + shift) = relu(scale*ymax + shift) resp. relu(scale*ymin + shift): once the batch
// statistics are known a (groups x c) kernel (pool_select_kernel) finishes the pool, and the (rows x c) tensor is not read again
// (134 MB for SA level 1 of the benchmark).  The first maximum wins ties (lowest row), like the stand-alone kernel.  Only the maximum
// is kept: channels with a negative scale are finished from Y itself (pool_select_kernel).
// acc: this lane's 16 accumulator values of one 32x32 tile (bias `bv` still to be added); lane l and l^32 hold the same column.
// Writes the column's maximum over the tile's 32 rows and the row it is first reached in (lanes < 32).  VALU work is not hidden under
// the MFMAs on this chip, so: one max3 tree for the value, then the first register that equals it (c_row ascends with the register
// index inside a lane), then one exchange with the other half-wave.
__device__ __forceinline__ void pool32_tile(const f32x16& acc, float bv, int lane, const PoolOut& po, size_t at) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r] + bv;
    float mx = __builtin_fmaxf(__builtin_fmaxf(v[0], v[1]), v[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, v[r]), v[r + 1]);
    mx = __builtin_fmaxf(mx, v[15]);
    int ir = 15;
#pragma unroll
    for (int r = 14; r >= 0; --r) ir = v[r] == mx ? r : ir;
    int imx = c_row(ir, lane);
    const float omx = __shfl_xor(mx, 32, 64);
    const int oimx = __shfl_xor(imx, 32, 64);
    if (omx > mx || (omx == mx && oimx < imx)) { mx = omx; imx = oimx; }
    if (lane < 32) { po.vmax[at] = mx; po.amax[at] = imx; }
}

// out = relu(scale * max_k y_k + shift) for scale >= 0 (BN+ReLU increasing in y).  A channel with a NEGATIVE scale (gamma < 0: legal,
// rare) needs the group's minimum instead, which the forward epilogue does not keep: those (group, channel) pairs re-read their 32
// values of Y here.
__global__ void pool_select_kernel(long total, int c, PoolOut po, const float* __restrict__ Y, int ldy, const float* __restrict__ scale,
                                   const float* __restrict__ shift, float* __restrict__ out, int* __restrict__ arg) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long g = i / c;
        const int col = (int)(i - g * c);
        const float sc = scale[col], sh = shift[col];
        float y = po.vmax[i];
        int a = po.amax[i];
        if (sc < 0.f) {
            const float* p = Y + g * 32 * ldy + col;
            y = p[0]; a = 0;
            for (int k = 1; k < 32; ++k) { const float t = p[(size_t)k * ldy]; if (t < y) { y = t; a = k; } }
            po.vmax[i] = y;                                          // vmax leaves as "y at the arg row" for every channel (gspn_pool_rsum)
        }
        float z = y * sc + sh;                                       // two roundings, as every other BN application here
        z = z > 0.f ? z : 0.f;
        out[i] = z;
        if (arg) arg[i] = a;
    }
}
// finishes a pool the forward launch started (gspn_mlp_fwd_pool32): out (groups, c), arg (groups, c) row offset of the maximum
extern "C" int gspn_pool32_select(long groups, int c, float* vmax, const int* amax, const float* Y, int ldy,
                                  const float* scale, const float* shift, float* out, int* arg, void* stream) {
    if (groups < 0 || c <= 0 || !scale || !shift || !out || !Y || ldy < c) return GSPN_ERR_ARG;
    const long total = groups * c;
    if (total == 0) return 0;
    PoolOut po{vmax, const_cast<int*>(amax)};
    hipLaunchKernelGGL(pool_select_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, total, c, po, Y, ldy, scale, shift, out, arg);
    return gspn_launch_status();
}

// The same for a max-pool over ns = 32 * sub rows (the proposal head's 256 / 512-row groups, model_rpointnet.py:68): the forward launch
// left the maximum and its row offset of every 32-row tile; a group's maximum is the first largest of its `sub` tile maxima (ascending
// tiles, strict '>': the first row that reaches the maximum, as the tile epilogue picks it inside a tile).  Replaces a pass over the
// (rows, c) tensor (bnrelu_maxpool4_kernel: 1 GB at the 1 M-row branch of the configs[3] shard) by one over (rows / 32, c).  Channels with
// a negative scale take the group MINIMUM from Y.  yarg (groups, c) receives y at the arg row (what gspn_pool_rsum reads in backward).
__global__ void pool_select_groups_kernel(long total4, int sub, int c4, PoolOut po, const float* __restrict__ Y, int ldy, const float* __restrict__ scale,
                                          const float* __restrict__ shift, float* __restrict__ out, int* __restrict__ arg, float* __restrict__ yarg) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long g = i / c4;
        const int col = (int)(i - g * c4) * 4;
        const long c = 4L * c4;
        const float4 sc = *reinterpret_cast<const float4*>(scale + col), sh = *reinterpret_cast<const float4*>(shift + col);
        const float* pv = po.vmax + (g * sub) * c + col;
        const int* pa = po.amax + (g * sub) * c + col;
        float4 best = *reinterpret_cast<const float4*>(pv);
        int4 bi = *reinterpret_cast<const int4*>(pa);
        for (int k0 = 1; k0 < sub; k0 += 4) {
            float4 v[4];
            int4 a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = min(k0 + u, sub - 1);
                v[u] = *reinterpret_cast<const float4*>(pv + (size_t)k * c);
                a[u] = *reinterpret_cast<const int4*>(pa + (size_t)k * c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u;
                if (k < sub) {
                    if (v[u].x > best.x) { best.x = v[u].x; bi.x = 32 * k + a[u].x; }
                    if (v[u].y > best.y) { best.y = v[u].y; bi.y = 32 * k + a[u].y; }
                    if (v[u].z > best.z) { best.z = v[u].z; bi.z = 32 * k + a[u].z; }
                    if (v[u].w > best.w) { best.w = v[u].w; bi.w = 32 * k + a[u].w; }
                }
            }
        }
        float y[4] = {best.x, best.y, best.z, best.w};
        int ai[4] = {bi.x, bi.y, bi.z, bi.w};
        const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
        float z[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (scv[j] < 0.f) {                                  // (rare) the group minimum, first row that reaches it
                const float* p = Y + g * (32L * sub) * ldy + col + j;
                y[j] = p[0]; ai[j] = 0;
                for (int k = 1; k < 32 * sub; ++k) { const float t = p[(size_t)k * ldy]; if (t < y[j]) { y[j] = t; ai[j] = k; } }
            }
            z[j] = y[j] * scv[j] + shv[j];                       // two roundings, as every other BN application here
            z[j] = z[j] > 0.f ? z[j] : 0.f;
        }
        *reinterpret_cast<float4*>(out + g * c + col) = make_float4(z[0], z[1], z[2], z[3]);
        *reinterpret_cast<float4*>(yarg + g * c + col) = make_float4(y[0], y[1], y[2], y[3]);
        if (arg) *reinterpret_cast<int4*>(arg + g * c + col) = make_int4(ai[0], ai[1], ai[2], ai[3]);
    }
}
extern "C" int gspn_pool32_select_groups(long groups, int sub, int c, const float* vmax, const int* amax, const float* Y, int ldy,
                                         const float* scale, const float* shift, float* out, int* arg, float* yarg, void* stream) {
    if (groups < 0 || sub <= 0 || c <= 0 || !scale || !shift || !out || !yarg || !Y || !vmax || !amax || ldy < c) return GSPN_ERR_ARG;
    if (c % 4 || ((uintptr_t)vmax | (uintptr_t)amax | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)out | (uintptr_t)yarg | (uintptr_t)arg) % 16) return GSPN_ERR_UNSUPPORTED;
    const long total4 = groups * (c / 4);
    if (total4 == 0) return 0;
    PoolOut po{const_cast<float*>(vmax), const_cast<int*>(amax)};
    hipLaunchKernelGGL(pool_select_groups_kernel, dim3(grid_for(total4, 256)), dim3(256), 0, (hipStream_t)stream, total4, sub, c / 4, po, Y, ldy, scale, shift, out, arg, yarg);
    return gspn_launch_status();
}

// ============================================================================================
// Forward:  Y = act(X).W + bias  (+ column sum / sumsq)
// grid (persistent row tiles, cout tiles of BN); block 256 = 4 waves, wave w owns rows w*32..+31.
// sA is K-major [TK][129] (written transposed from float4 row segments, 2 x ds_write2_b32),
// sB is [TK][BN+4] (straight float4 copy of W rows).
// ============================================================================================
// (waves per SIMD: left alone, the 128-column instance takes 277 registers = ONE workgroup per CU; capped at 256 it spills nothing that
//  matters and two reside: 45 -> 33 us on the 32768 x 128 -> 256 layer.  Three waves for the 64-column instance measured no gain.)
template <int BN, bool VEC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BN <= 32 ? 3 : 2)))
void mlp_fwd_kernel(long rows, int cin, int cout, const float* __restrict__ X, int ldx,
                                                      const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                      const float* __restrict__ W, const float* __restrict__ bias,
                                                      float* __restrict__ Y, int ldy, float* __restrict__ stats, PoolOut po) {
    constexpr int NT = BN / 32;
    constexpr int LDB = BN + 4;
    constexpr int BQ = BN / 4;                   // float4 per B row
    constexpr int NB = (TK * BQ) / 256;          // B float4 per thread per chunk
    __shared__ __attribute__((aligned(16))) float sA[TK * LDT];
    __shared__ __attribute__((aligned(16))) float sB[TK * LDB];
    __shared__ float sRed[2 * 4 * BN];
    extern __shared__ __attribute__((aligned(16))) float s_chan[];         // [2][cpad]
    const int cpad = (cin + 3) / 4 * 4 + 4;
    float* sSc = s_chan;
    float* sSh = s_chan + cpad;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n0 = blockIdx.y * BN;
    const bool act = in_scale != nullptr;
    stage_chan(sSc, in_scale, cin, 1.f, cpad);
    stage_chan(sSh, in_shift, cin, 0.f, cpad);
    __syncthreads();
    const long ntiles = (rows + TM - 1) / TM;
    const int nchunks = (cin + TK - 1) / TK;
    const int kq = (t & 7) * 4;                  // this thread's k offset inside a chunk (fixed: 256 % 8 == 0)
    const int arow = t >> 3;                     // + 32*i

    float csum[NT], csq[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) { csum[i] = 0.f; csq[i] = 0.f; }

    float4 ra[4], rb[NB];
    long f_tile = 0;
    int f_c = 0;
    auto fetch = [&](long tile, int c) {              // raw loads only: everything stays in flight until commit()
        f_tile = tile; f_c = c;
        const long m0 = tile * TM;
        const int k = c * TK + kq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = m0 + arow + 32 * i;
            ra[i] = load4_raw<VEC>(X, row < rows ? row : rows - 1, ldx, k, cin);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = t + 256 * i;
            const int kk = f / BQ, nq = (f - kk * BQ) * 4;
            const int kg = c * TK + kk;
            rb[i] = load4_raw<VEC>(W, kg < cin ? kg : cin - 1, cout, n0 + nq, cout);
        }
    };
    auto commit = [&]() {
        const long m0 = f_tile * TM;
        const int k = f_c * TK + kq;
        Chan4 ch;
        ch.sc = lds4(sSc, k, cpad);
        ch.sh = lds4(sSh, k, cpad);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 v = act4(ra[i], act, ch, k, cin, (m0 + arow + 32 * i) < rows);
            float* d = sA + kq * LDT + arow + 32 * i;
            d[0 * LDT] = v.x; d[1 * LDT] = v.y; d[2 * LDT] = v.z; d[3 * LDT] = v.w;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = t + 256 * i;
            const int kk = f / BQ, nq = (f - kk * BQ) * 4;
            *reinterpret_cast<float4*>(sB + kk * LDB + nq) = mask4(rb[i], n0 + nq, cout, (f_c * TK + kk) < cin);
        }
    };

    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = n0 + nt * 32 + (lane & 31);
        bv[nt] = (bias && col < cout) ? bias[col] : 0.f;
    }
    f32x16 acc[NT];
    // software pipeline over the flat sequence of (tile, chunk) steps of this block: each iteration issues the loads of
    // step s, runs the MFMAs of step s-1 out of LDS while they fly, then commits step s to LDS.  No load result is live
    // across a loop back-edge (hipcc otherwise shuffles loop-carried quads with v_mov right after the loads and waits).
    const long my_tiles = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const long nsteps = my_tiles * nchunks;
    for (long sidx = 0; sidx <= nsteps; ++sidx) {
        if (sidx < nsteps) fetch(blockIdx.x + (sidx / nchunks) * gridDim.x, (int)(sidx % nchunks));
        if (sidx > 0) {
            const long ps = sidx - 1;
            const long tile = blockIdx.x + (ps / nchunks) * gridDim.x;
            const int c = (int)(ps % nchunks);
            if (c == 0) {
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            }
            const int kmax = min(TK, (cin - c * TK + 1) & ~1);
            for (int kk = 0; kk < kmax; kk += 2) {
                const float a = sA[(kk + (lane >> 5)) * LDT + wave * 32 + (lane & 31)];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float b = sB[(kk + (lane >> 5)) * LDB + nt * 32 + (lane & 31)];
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[nt], 0, 0, 0);
                }
            }
            if (c == nchunks - 1) {
                // ---- epilogue: + bias, store, column statistics ----
                const long m0 = tile * TM;
                const bool full = m0 + TM <= rows;          // wave-uniform: full tiles store without per-row guards
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = n0 + nt * 32 + (lane & 31);
                    if (col < cout) {
                        if (po.vmax && m0 + wave * 32 < rows) pool32_tile(acc[nt], bv[nt], lane, po, (size_t)((m0 >> 5) + wave) * cout + col);   // rows % 32 == 0 (launcher)
                        float* yp = Y + (m0 + wave * 32 + 4 * (lane >> 5)) * ldy + col;
                        if (full) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float y = acc[nt][r] + bv[nt];
                                yp[(long)((r & 3) + 8 * (r >> 2)) * ldy] = y;
                                csum[nt] += y;
                                csq[nt] += y * y;
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const long row = m0 + wave * 32 + c_row(r, lane);
                                if (row < rows) {
                                    const float y = acc[nt][r] + bv[nt];
                                    Y[row * ldy + col] = y;
                                    csum[nt] += y;
                                    csq[nt] += y * y;
                                }
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (sidx < nsteps) commit();
        __syncthreads();
    }
    if (stats) {
        // lanes l and l+32 hold the same columns; then 4 waves -> LDS -> one double atomic per column
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            csum[nt] += __shfl_xor(csum[nt], 32, 64);
            csq[nt] += __shfl_xor(csq[nt], 32, 64);
            if (lane < 32) {
                sRed[(wave * 2 + 0) * BN + nt * 32 + lane] = csum[nt];
                sRed[(wave * 2 + 1) * BN + nt * 32 + lane] = csq[nt];
            }
        }
        __syncthreads();
        // per-block partial sums (no hot-spot atomics): stats[blockIdx.x][2][cout]; gspn_bn_finalize adds them in double
        float* ws = stats + (size_t)blockIdx.x * 2 * cout;
        for (int j = t; j < BN; j += 256) {
            const int col = n0 + j;
            if (col < cout) {
                float s = 0.f, q = 0.f;
                for (int w = 0; w < 4; ++w) { s += sRed[(w * 2 + 0) * BN + j]; q += sRed[(w * 2 + 1) * BN + j]; }
                ws[col] = s;
                ws[cout + col] = q;
            }
        }
    }
}

__device__ __forceinline__ void glds16(const float* g, float* l) {        // 64 lanes x 16 B -> 1 KiB of LDS at l (wave-uniform)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// ---- forward, streaming variant (16-byte aligned rows, cin small enough for W + two row tiles to sit in LDS) ----
// Raw X row tiles go HBM -> LDS directly (global_load_lds_dwordx4), double buffered, one barrier per row tile; W is resident in LDS for
// the whole (persistent) workgroup.  The tile is row-major as in memory -- no transposition pass: lane (i, kh) of an MFMA reads its A
// operand as ONE ds_read_b64 per two k-steps (k is assigned 4j+2kh+e, e = step parity; B is pre-arranged in LDS to match), BN+ReLU of
// the previous layer is applied on that read with (scale, shift) pairs from LDS.  Quads of a row are XOR-swizzled by the row index on
// the SOURCE side of the LDS-DMA (the LDS image itself must stay lane-linear), which spreads the 32 rows of an operand read over banks.
// Waves: TRG row groups of 32 rows x (4/TRG) column groups.
__device__ __forceinline__ int fwd_swz_key(int row, int qx) {            // qx a power of two >= 2, else no swizzle
    if (qx & (qx - 1)) return 0;
    return qx >= 16 ? (row & (qx - 1)) : ((row * qx) >> 4) & (qx - 1);
}
// GATHER (the first layer of a set-abstraction stack, pointnet_util.py:36-52 + :109-113 without the grouped tensor): input row r is
// VIRTUAL -- [ feat[gidx[r]][0 .. 4*cq) | rel[r][0..4) ], the grouped point's features followed by its centred coordinates -- and every
// LDS-DMA lane simply takes its quad from where it lives: the gather costs nothing on top of the load.  The kernel's k order is that
// internal one; W's rows are picked accordingly while W is staged (gspn_gather_args::xyz_first says where the 3 xyz rows of W sit).
struct GatherSrc { const float* feat; const int* gidx; const float* rel; int cq, c_real, xyz_first; };
__device__ __forceinline__ int gather_w_row(const GatherSrc& g, int k) {           // row of the caller's W for internal k, or -1 (zero)
    if (k < 4 * g.cq) return k < g.c_real ? (g.xyz_first ? 3 + k : k) : -1;
    const int a = k - 4 * g.cq;
    return a < 3 ? (g.xyz_first ? a : g.c_real + a) : -1;
}
// BWDP: the same streaming structure as pass B of a POOLED TOP layer (pool groups of 32 rows = one wave's row group).  With
// dY = cA*dyh + cB*y + cC and y = xhat.W + b,
//     dX = dY.W^T = xhat.(W diag(cB) W^T) + S.W^T + 1.((b*cB + cC).W^T),      S = cA*dyh: ONE non-zero per (pool group, channel),
// so the layer's own (rows, ctop) output is not needed: X is the previous layer's pre-BN output (xhat = relu(bn(X)), exactly the forward's
// operand), `W` the (cin, cin) matrix W diag(cB) W^T, `bias` the constant row, `Y` dX -- and after the dense k loop a second one runs over
// the ctop channels with the A operand built from the group's (arg, V) pairs (V = cA * dPool * [pooled output > 0]) and
// the B operand W^T from LDS.  The epilogue takes the previous layer's BN reductions sum(dyh), sum(dyh*xhat_n) where the forward takes the
// column statistics (same partial layout), reading the raw X tile that is still in LDS.  134 MB of Y (SA level 1) are not read.
struct BwdPool {
    const int* arg;       // (groups, ctop) row offset of each group's arg-max
    const float* dPool;   // (groups, ctop) upstream gradient of the pooled output
    const float* pooled;  // (groups, ctop) the pooled output itself: where it is 0 the ReLU passes nothing
    const float* cA;      // (ctop)        V = cA * dPool * [pooled > 0] is formed as the pairs are fetched
    const float* Wtop;    // (cin, ctop) the layer's weights
    int ctop;
    const float* mean;    // batch statistics of the previous layer (whose pre-BN output X is)
    const float* var;
    float eps;
};
template <int BN, int TRG, bool GATHER, bool BWDP = false>
__global__ __launch_bounds__(256) void mlp_fwd_stream_kernel(int rows, int cin, int cout, const float* __restrict__ X, int ldx,
                                                             const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                             const float* __restrict__ W, const float* __restrict__ bias,
                                                             float* __restrict__ Y, int ldy, float* __restrict__ stats, int nparts, PoolOut po,
                                                             GatherSrc gs, BwdPool bp) {
    constexpr int NT = BN / 32, CG = 4 / TRG, NTW = NT / CG, TR = 32 * TRG;
    static_assert(NT % CG == 0 && NTW >= 1, "column groups tile the block");
    constexpr int JPMAX = 8;                              // 1-KiB pieces per wave per row tile (TR * QX / 64 / 4 <= 8 by the launcher's LDS check)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int QX = (cin + 3) >> 2, KP = QX * 4;
    float* sSS = smem;                                    // [KP][2]  (scale, shift) of the input channels
    float* sW = sSS + KP * 2;                             // [QX][2][BN][2]  W[4j+2kh+e][n]
    float* sXb = sW + KP * BN;                            // 2 x [TR][QX quads], quads swizzled
    // BWDP: W^T of the layer in the layout of sW (k = channel of the top layer), then 2 x TRG groups of (arg, V) pairs
    const int CT4 = BWDP ? (bp.ctop + 3) / 4 * 4 : 0;
    float* sWt = sXb + 2 * TR * KP;
    int2* sPV = reinterpret_cast<int2*>(sWt + CT4 * BN);  // [2][TRG][CT4]
    __shared__ float sRed[2 * TRG * BN];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int rg = wave % TRG, cg = wave / TRG;
    const int n0 = blockIdx.y * BN;
    const float lo = in_scale ? 0.f : -__builtin_inff();
    for (int k = t; k < KP; k += 256) {
        sSS[2 * k] = (in_scale && k < cin) ? in_scale[k] : 1.f;
        sSS[2 * k + 1] = (in_scale && k < cin) ? in_shift[k] : 0.f;
    }
    for (int f = t; f < KP * BN; f += 256) {
        const int k = f / BN, n = f - k * BN;
        const int kw = GATHER ? gather_w_row(gs, k) : (k < cin ? k : -1);
        const float w = (kw >= 0 && n0 + n < cout) ? W[(size_t)kw * cout + n0 + n] : 0.f;
        const int j = k >> 2, h = (k >> 1) & 1, e = k & 1;
        sW[(((j * 2 + h) * BN) + n) * 2 + e] = w;
    }
    if constexpr (BWDP) {
        for (int f = t; f < CT4 * BN; f += 256) {
            const int k = f / BN, n = f - k * BN;
            const float w = (k < bp.ctop && n0 + n < cout) ? bp.Wtop[(size_t)(n0 + n) * bp.ctop + k] : 0.f;
            const int j = k >> 2, h = (k >> 1) & 1, e = k & 1;
            sWt[(((j * 2 + h) * BN) + n) * 2 + e] = w;
        }
    }
    const int ntiles = (rows + TR - 1) / TR;
    const int npiece = TR * QX / 64;                       // whole pieces: TR is a multiple of 64 or QX is even (launcher)
    int p_row[JPMAX], p_col[JPMAX];
#pragma unroll
    for (int j = 0; j < JPMAX; ++j) {
        const int f = (wave + 4 * j) * 64 + lane;
        const int row = f / QX, slot = f - row * QX;
        p_row[j] = row;
        p_col[j] = GATHER ? (slot ^ fwd_swz_key(row, QX)) : min((slot ^ fwd_swz_key(row, QX)) * 4, ldx - 4);      // GATHER: the source QUAD
    }
    // GATHER: source rows of the lane's pieces, fetched one tile ahead of the tile whose DMA they address (g_nxt is consumed right
    // after the vmcnt(0) that opens an iteration, so the look-up never adds a wait of its own)
    int g_nxt[GATHER ? JPMAX : 1];
    auto gfetch = [&](int tile) {
        if constexpr (GATHER) {
            const int r0 = tile * TR;
#pragma unroll
            for (int j = 0; j < JPMAX; ++j) g_nxt[j] = gs.gidx[min(r0 + p_row[j], rows - 1)];
        }
    };
    auto issue = [&](int tile, int buf) {
        float* dst = sXb + buf * (TR * KP);
        const int r0 = tile * TR;
#pragma unroll
        for (int j = 0; j < JPMAX; ++j)
            if (wave + 4 * j < npiece) {
                if constexpr (GATHER) {
                    const float* src = p_col[j] < gs.cq ? gs.feat + ((size_t)g_nxt[j] * gs.cq + p_col[j]) * 4
                                                        : gs.rel + (size_t)min(r0 + p_row[j], rows - 1) * 4;
                    glds16(src, dst + (wave + 4 * j) * 256);
                } else {
                    glds16(X + ((size_t)min(r0 + p_row[j], rows - 1) * ldx + p_col[j]), dst + (wave + 4 * j) * 256);
                }
            }
    };
    // BWDP: the (arg, V) pairs of a tile's TRG pool groups, fetched a tile ahead into registers (<= 2 per thread: TRG * ctop <= 512, launcher)
    // and handed to LDS right after the vmcnt(0) that opens the tile's iteration -- like g_nxt, never a wait of their own
    int pv_a[BWDP ? 2 : 1];
    float pv_v[BWDP ? 2 : 1];
    auto pfetch = [&](int tile) {
        if constexpr (BWDP) {
            const long ngroups = rows >> 5;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = t + 256 * i;
                const int gl = e / CT4, c = e - gl * CT4;
                pv_a[i] = -1; pv_v[i] = 0.f;
                if (gl < TRG && c < bp.ctop) {
                    const long g = min((long)tile * TRG + gl, ngroups - 1);
                    pv_a[i] = bp.arg[g * bp.ctop + c];
                    pv_v[i] = bp.pooled[g * bp.ctop + c] > 0.f ? bp.cA[c] * bp.dPool[g * bp.ctop + c] : 0.f;
                }
            }
        }
    };
    float bv[NTW], csum[NTW], csq[NTW], p_rs[NTW], p_mr[NTW];
#pragma unroll
    for (int y = 0; y < NTW; ++y) {
        const int col = n0 + (cg * NTW + y) * 32 + l31;
        bv[y] = (bias && col < cout) ? bias[col] : 0.f;
        csum[y] = csq[y] = 0.f;
        p_rs[y] = p_mr[y] = 0.f;
        if constexpr (BWDP) {
            const int cc = min(col, cout - 1);
            p_rs[y] = (float)(1.0 / sqrt((double)bp.var[cc] + (double)bp.eps));
            p_mr[y] = -bp.mean[cc] * p_rs[y];
        }
    }
    const int arow = rg * 32 + l31;                         // this lane's row inside the tile (A operand)
    const int akey = fwd_swz_key(arow, QX);
    // epilogue of a finished tile: + bias, store, column statistics
    auto epilogue = [&](int tile, const f32x16 (&acc)[NTW]) {
        const int m0 = tile * TR + rg * 32;
        const bool full = tile * TR + TR <= rows;           // uniform: full tiles store without per-row guards
#pragma unroll
        for (int y = 0; y < NTW; ++y) {
            const int col = n0 + (cg * NTW + y) * 32 + l31;
            if (col < cout) {
                if (po.vmax && m0 < rows) pool32_tile(acc[y], bv[y], lane, po, (size_t)(m0 >> 5) * cout + col);      // rows % 32 == 0 (launcher)
                float* yp = Y + (size_t)(m0 + 4 * kh) * ldy + col;
                if (full) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[y][r] + bv[y];
                        yp[(size_t)((r & 3) + 8 * (r >> 2)) * ldy] = v;
                        csum[y] += v;
                        csq[y] += v * v;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + c_row(r, lane);
                        if (row < rows) {
                            const float v = acc[y][r] + bv[y];
                            Y[(size_t)row * ldy + col] = v;
                            csum[y] += v;
                            csq[y] += v * v;
                        }
                    }
                }
            }
        }
    };
    // BWDP: dX = acc + constant row, stored; the previous layer's reductions from the finished tile and the raw X tile (still in LDS: called
    // right after the tile's MFMAs, not a tile later like the forward's)
    auto epilogue_b = [&](int tile, const f32x16 (&acc)[NTW], int buf) {
        const int m0 = tile * TR + rg * 32;
        const bool full = tile * TR + TR <= rows;
        const float* xt = sXb + buf * (TR * KP);
#pragma unroll
        for (int y = 0; y < NTW; ++y) {
            const int col = n0 + (cg * NTW + y) * 32 + l31;
            if (col < cout) {
                const float sc = sSS[2 * col], sh = sSS[2 * col + 1];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = rg * 32 + c_row(r, lane);
                    const int row = tile * TR + rl;
                    const float v = acc[y][r] + bv[y];
                    const float xr = xt[rl * KP + ((((col >> 2) ^ fwd_swz_key(rl, QX)) << 2) | (col & 3))];
                    const bool live = full || row < rows;
                    if (live) Y[(size_t)row * ldy + col] = v;
                    const float dyh = (live && xr * sc + sh > 0.f) ? v : 0.f;
                    csum[y] += dyh;
                    csq[y] = __builtin_fmaf(dyh, __builtin_fmaf(xr, p_rs[y], p_mr[y]), csq[y]);
                }
            }
        }
        (void)m0;
    };
    f32x16 acc[NTW], pacc[NTW];
    int it = 0, ptile = -1;
    // tiles of this workgroup: strided over the grid, or (GATHER) a contiguous run placed so that the workgroups of one XCD (block % 8)
    // cover one contiguous eighth of the rows -- the feature rows they gather then belong to one or two scenes and stay in that XCD's L2
    int tile0 = blockIdx.x, tstep = gridDim.x, tend = ntiles;
    if (GATHER && (gridDim.x & 7) == 0) {
        const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
        const int cid = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
        tile0 = cid * per;
        tstep = 1;
        tend = min(ntiles, tile0 + per);
    }
    if (tile0 < tend) {
        gfetch(tile0);
        issue(tile0, 0);
        if (tile0 + tstep < tend) gfetch(tile0 + tstep);
        pfetch(tile0);
    }
    for (int tile = tile0; tile < tend; tile += tstep, ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (BWDP) {                               // this tile's (arg, V) pairs: registers -> LDS half `it & 1` (last read two tiles ago)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = t + 256 * i;
                if (e < TRG * CT4) sPV[(it & 1) * (TRG * CT4) + e] = make_int2(pv_a[i], __float_as_int(pv_v[i]));
            }
        }
        __syncthreads();                                    // tile `it` has landed (and W / constants are visible the first time)
        if (tile + tstep < tend) {
            issue(tile + tstep, (it + 1) & 1);
            if (tile + 2 * tstep < tend) gfetch(tile + 2 * tstep);
            pfetch(tile + tstep);
        }
        // the previous tile's stores go out here, a whole compute phase before the next vmcnt(0): their latency is never waited on
        if (ptile >= 0) epilogue(ptile, pacc);
        const float* sx = sXb + (it & 1) * (TR * KP) + arow * KP;
#pragma unroll
        for (int y = 0; y < NTW; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[y][r] = 0.f;
        auto kstep = [&](int j) {                           // 4 k's: operands of two MFMA steps per tile
            const float2 a2 = *reinterpret_cast<const float2*>(sx + ((j ^ akey) << 2) + kh * 2);
            const float4 ss = *reinterpret_cast<const float4*>(sSS + (4 * j + 2 * kh) * 2);
            float2 b2[NTW];
#pragma unroll
            for (int y = 0; y < NTW; ++y)
                b2[y] = *reinterpret_cast<const float2*>(sW + (((j * 2 + kh) * BN) + (cg * NTW + y) * 32 + l31) * 2);
            const float a0 = __builtin_fmaxf(a2.x * ss.x + ss.y, lo);
            const float a1 = __builtin_fmaxf(a2.y * ss.z + ss.w, lo);
#pragma unroll
            for (int y = 0; y < NTW; ++y) acc[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b2[y].x, acc[y], 0, 0, 0);
#pragma unroll
            for (int y = 0; y < NTW; ++y) acc[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b2[y].y, acc[y], 0, 0, 0);
        };
        int j = 0;
        for (; j + 3 < QX; j += 4) { kstep(j); kstep(j + 1); kstep(j + 2); kstep(j + 3); }   // (hipcc will not partially unroll a loop of MFMAs itself)
        for (; j < QX; ++j) kstep(j);
        if constexpr (BWDP) {
            // the sparse term: k runs over the top layer's channels; lane (row i, k half) takes S[i][k] = (arg[k] == i) ? V[k] : 0
            const int2* pv = sPV + (it & 1) * (TRG * CT4) + rg * CT4;
            auto sstep = [&](int q) {
                const int4 e2 = *reinterpret_cast<const int4*>(pv + 4 * q + 2 * kh);
                float2 b2[NTW];
#pragma unroll
                for (int y = 0; y < NTW; ++y)
                    b2[y] = *reinterpret_cast<const float2*>(sWt + (((q * 2 + kh) * BN) + (cg * NTW + y) * 32 + l31) * 2);
                const float a0 = e2.x == l31 ? __int_as_float(e2.y) : 0.f;
                const float a1 = e2.z == l31 ? __int_as_float(e2.w) : 0.f;
#pragma unroll
                for (int y = 0; y < NTW; ++y) acc[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b2[y].x, acc[y], 0, 0, 0);
#pragma unroll
                for (int y = 0; y < NTW; ++y) acc[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b2[y].y, acc[y], 0, 0, 0);
            };
            const int QS = CT4 >> 2;
            int q = 0;
            for (; q + 3 < QS; q += 4) { sstep(q); sstep(q + 1); sstep(q + 2); sstep(q + 3); }
            for (; q < QS; ++q) sstep(q);
            epilogue_b(tile, acc, it & 1);
        } else {
#pragma unroll
            for (int y = 0; y < NTW; ++y) pacc[y] = acc[y];
            ptile = tile;
        }
    }
    if (ptile >= 0) epilogue(ptile, pacc);
    if (stats) {
        __syncthreads();
#pragma unroll
        for (int y = 0; y < NTW; ++y) {
            csum[y] += __shfl_xor(csum[y], 32, 64);
            csq[y] += __shfl_xor(csq[y], 32, 64);
            if (lane < 32) {
                sRed[(rg * 2 + 0) * BN + (cg * NTW + y) * 32 + lane] = csum[y];
                sRed[(rg * 2 + 1) * BN + (cg * NTW + y) * 32 + lane] = csq[y];
            }
        }
        __syncthreads();
        // per-block partial sums stats[part][2][cout]; parts this launch does not produce are zero-filled
        float* ws = stats + (size_t)blockIdx.x * 2 * cout;
        for (int j = t; j < BN; j += 256) {
            const int col = n0 + j;
            if (col < cout) {
                float sm = 0.f, q = 0.f;
                for (int w = 0; w < TRG; ++w) { sm += sRed[(w * 2 + 0) * BN + j]; q += sRed[(w * 2 + 1) * BN + j]; }
                ws[col] = sm;
                ws[cout + col] = q;
            }
        }
        // (spread over all workgroups: a single one would serialise ~500 stores per thread at the tail of the launch)
        for (long f = (long)gridDim.x * 2 * cout + (long)blockIdx.x * 256 + t; f < (long)nparts * 2 * cout; f += (long)gridDim.x * 256) {
            const int col = (int)(f % cout);
            if (col >= n0 && col < n0 + BN) stats[f] = 0.f;
        }
    }
}

static inline unsigned row_grid(long rows, int ytiles, int per_cu) {
    const long ntiles = (rows + TM - 1) / TM;
    long cap = (long)GSPN_PLAN_CUS * per_cu / (ytiles > 0 ? ytiles : 1);
    if (cap < 64) cap = 64;
    return (unsigned)(ntiles < cap ? ntiles : cap);
}
// Column-tile width of the register-staged GEMM kernels (forward fallback, pass B) for `cols` output columns: the widest of 128/64/32
// that still gives the chip one workgroup per CU, unless a narrower one wastes markedly fewer padded columns (131 columns: 3 x 64
// instead of 2 x 128).  Short layers (<= 16384 rows) thus run 32-wide tiles in 4-8 column blocks instead of a quarter of the CUs
// (tools/fwd_sweep.py on MI355X: 4096 x 384 -> 256 forward 44 -> 27 us, pass B 33 -> 26 us).  GSPN_FWD_FORCE_BN / GSPN_BWD_FORCE_BN
// override it (tuning hooks).
static inline int pick_bn(long rows, int cols, const char* env) {
    {
        const char* e = getenv(env);
        const int f = e ? atoi(e) : 0;
        if (f == 32 || f == 64 || f == 128) return f;
    }
    const long ntiles = (rows + TM - 1) / TM;
    auto nblk = [&](int bn) { return ntiles * ((cols + bn - 1) / bn); };
    auto pad = [&](int bn) { return (cols + bn - 1) / bn * bn; };
    int bn = cols <= 32 ? 32 : (cols <= 64 ? 64 : 128);
    static int fill = 0;
    if (!fill) { const char* e = getenv("GSPN_BN_FILL"); fill = e ? atoi(e) : 2 * GSPN_PLAN_CUS; if (fill <= 0) fill = 2 * GSPN_PLAN_CUS; }
    while (bn > 32 && (nblk(bn) < fill || 4 * pad(bn) > 5 * pad(bn / 2))) bn /= 2;
    return bn;
}
// number of row-blocks (= partial-statistics rows) the forward launch of a (rows, cout) layer uses
// (workgroups per CU = what the kernel's LDS footprint lets reside at once: a persistent grid larger than that runs a second wave)
static inline int bwd_bpc_narrow() { static const int v = env_int("GSPN_BWD_BPC", 4); return v < 1 ? 1 : v; }
static inline bool fwd_short_rows(long rows) { static const int on = env_int("GSPN_FWD_SHORT", 1); return on && rows >= 64 && rows <= GSPN_SHORT_ROWS && !(rows & 63); }
// (the count depends on (rows, cout) only -- it sizes a caller-allocated buffer through gspn_mlp_fwd_stats_bytes, which has no cin / pointer
//  arguments -- so a layer of <= 8192 rows that the split-K kernel then DECLINES (cin not one of its instances, an unaligned operand, a gathered source)
//  still gets one partial row per 32 rows: the 128-row-tile kernels below are launched with up to 4x more workgroups than tiles; the extra ones write their
//  zero statistics row and leave.  Correct, and bounded at 256 workgroups of a few hundred cycles on layers of <= 8192 rows.)
static inline unsigned fwd_blocks(long rows, int cout) {
    if (fwd_short_rows(rows)) return (unsigned)short_fwd_parts(rows);      // short layers (mlp_short.hip): one partial row per 32-row tile, at most 512
    return row_grid(rows, cout <= 64 ? 1 : (cout + 127) / 128, cout <= 64 ? 4 : env_int("GSPN_FWD_WIDE_BPC", 3));
}
extern "C" long gspn_mlp_fwd_stats_bytes(long rows, int cout) {
    if (rows < 0 || cout <= 0) return GSPN_ERR_ARG;
    return (long)sizeof(float) * 2 * cout * (long)fwd_blocks(rows > 0 ? rows : 1, cout);
}

// ---- MFMA operand streams from LDS with explicit immediates and explicit waits ------------------------------------------------------
// hipcc pairs the unrolled operand reads of a k loop into ds_read2_b32 whose 8-bit offsets do not reach across k steps of a transposed
// tile: it materialises one base register per pair (30 registers in the 64 x 64 fused kernel, an occupancy step) and waits lgkmcnt(0)
// in front of every pair of MFMAs.  lds_product issues ds_read_b32 with 16-bit byte offsets from ONE base per operand, four k-pairs per
// batch, the next batch in flight during the current batch's MFMAs.  The waits carry the operand registers as "+v" so that no use can
// move above them; LDS returns in order, so `lgkmcnt(8)` with the next batch's eight reads behind it covers the current one whatever
// else the compiler has in flight.
// The compiler does not track LDS operations issued from inline asm: correctness rests on the read's destination staying in ITS register
// until the tied wait ("=&v": early clobber, never shared with an input; the "+v" ties pin the value across the wait).  GSPN_FUSED_ASM=0
// (a build flag: GSPN_EXTRA_HIPCC_FLAGS=-DGSPN_FUSED_ASM=0) replaces the asm by plain LDS loads the compiler schedules and waits for
// itself -- slower (ds_read2 pairs, one base register per pair), the reference point when a ROCm bump needs checking.
#ifndef GSPN_FUSED_ASM
#define GSPN_FUSED_ASM 1
#endif
template <int OFF> __device__ __forceinline__ void lds_rd(float& v, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read_b32 offset field");
#if GSPN_FUSED_ASM
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(v) : "v"(addr), "n"(OFF) : "memory");
#else
    v = *reinterpret_cast<const __attribute__((address_space(3))) float*>((uintptr_t)(addr + OFF));
#endif
}
__device__ __forceinline__ unsigned lds_addr(const float* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p; }
#if GSPN_FUSED_ASM
#define LDS_WAIT(N_, A_, B_)                                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(A_[0]), "+v"(A_[1]), "+v"(A_[2]), "+v"(A_[3]), "+v"(B_[0]), "+v"(B_[1]), "+v"(B_[2]), "+v"(B_[3]) :: "memory")
#else
#define LDS_WAIT(N_, A_, B_) ((void)0)
#endif
// wait until at most CNT LDS operations are outstanding; the N registers of `v` are tied to the wait
template <int CNT, int N> __device__ __forceinline__ void lds_wait_n(float (&v)[N]) {
    static_assert(N == 4 || N == 8 || N == 16, "operand batch");
    static_assert(CNT >= 0 && CNT <= 15, "lgkmcnt is a 4-bit counter");
    if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "n"(CNT) : "memory");
    else if constexpr (N == 8)
        asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "n"(CNT) : "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(%16)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                     "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : "n"(CNT) : "memory");
}
template <class F, int... Is> __device__ __forceinline__ void sfor_(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F> __device__ __forceinline__ void sfor(F&& f) { sfor_(f, std::make_integer_sequence<int, N>{}); }
// acc += sum over NS k-pairs s of A_s x B_s, A_s at a_addr + s*SA bytes, B_s at b_addr + s*SB bytes (this lane's element of each operand);
// XF: the A operand goes through relu(a*xs + xh), two roundings (the forward's operand form)
template <int NS, int SA, int SB, bool XF, int ABL>
__device__ __forceinline__ void lds_product(f32x16& acc, unsigned a_addr, unsigned b_addr, float xs, float xh) {
    static_assert(NS % 4 == 0, "four k-pairs per batch");
    constexpr int NB = NS / 4;
    float a[2][4], b[2][4];
    sfor<4>([&](auto u_) { constexpr int u = decltype(u_)::value; lds_rd<u * SA>(a[0][u], a_addr); lds_rd<u * SB>(b[0][u], b_addr); });
    sfor<NB>([&](auto bi_) {
        constexpr int bi = decltype(bi_)::value, cur = bi & 1, nxt = cur ^ 1;
        if constexpr (bi + 1 < NB) {
            sfor<4>([&](auto u_) {
                constexpr int u = decltype(u_)::value;
                lds_rd<((bi + 1) * 4 + u) * SA>(a[nxt][u], a_addr);
                lds_rd<((bi + 1) * 4 + u) * SB>(b[nxt][u], b_addr);
            });
            LDS_WAIT(8, a[cur], b[cur]);
        } else {
            LDS_WAIT(0, a[cur], b[cur]);
        }
        sfor<4>([&](auto u_) {
            constexpr int u = decltype(u_)::value;
            float x = a[cur][u];
            if constexpr (XF) { x = x * xs + xh; x = x > 0.f ? x : 0.f; }
            if constexpr (ABL) acc[u] += x * b[cur][u];
            else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, b[cur][u], acc, 0, 0, 0);
        });
    });
}
// ============================================================================================
// Forward, lean form of the register-staged kernel (r03; see bwd_lean_kernel for the reasoning: these kernels are bound by the vector
// instructions they issue around the MFMAs, and most of those were 64-bit index arithmetic, clamps and masks).  Shapes: 16-byte aligned
// pitches, rows a multiple of 128, cin and cout multiples of 32, every offset below 2^32 bytes.  Same arithmetic as mlp_fwd_kernel in the
// same order -- relu(x*scale + shift) with two roundings on the operand, K chunks of 32 accumulated in ascending k, bias added last --
// so Y is bit-identical to that kernel's; the column sums of squares use one fused multiply-add per element.
// ============================================================================================
template <int NT, bool ACT, bool POOL>
__global__ __launch_bounds__(256) void fwd_lean_kernel(int rows, int cin, int cout, const float* __restrict__ X, int ldx, const float* __restrict__ in_scale,
                                                       const float* __restrict__ in_shift, const float* __restrict__ W, const float* __restrict__ bias,
                                                       float* __restrict__ Y, int ldy, float* __restrict__ stats, PoolOut po) {
    constexpr int BN = 32 * NT;
    constexpr int LDA = 129, LDB = BN + 4;
    __shared__ __attribute__((aligned(16))) float sA[32 * LDA];
    __shared__ __attribute__((aligned(16))) float sB[32 * LDB];
    __shared__ float sRed[2 * 4 * BN];
    extern __shared__ __attribute__((aligned(16))) float s_chan[];         // [2][cpad]: scale, shift of the input channels
    const int cpad = (cin + 3) / 4 * 4 + 4;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int n0 = blockIdx.y * BN;
    const int ntiles = rows >> 7, nchunks = cin >> 5;
    if constexpr (ACT) {
        for (int i = t; i < cpad; i += 256) {
            s_chan[i] = i < cin ? in_scale[i] : 1.f;
            s_chan[cpad + i] = i < cin ? in_shift[i] : 0.f;
        }
    }
    const int kq = (t & 7) * 4, arow = t >> 3;                  // A: rows arow + 32 i, k = kq .. kq+3 of the chunk
    const unsigned oa = (unsigned)(arow * ldx + kq) * 4u;
    constexpr int BQ = BN / 4;                                  // B: W rows (k) of the chunk, float4 along n
    const int bk = t / BQ, bq = (t % BQ) * 4;                   // + (256 / BQ) k-rows per i
    const unsigned ob = (unsigned)(bk * cout + bq) * 4u;
    float4 ra[4], rb[NT];
    auto fetch = [&](int tile, int c) {
        const char* xb = reinterpret_cast<const char*>(X + (size_t)(tile << 7) * ldx + c * 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const float4*>(xb + (size_t)(32 * i) * ldx * 4 + oa);
        const char* wb = reinterpret_cast<const char*>(W + (size_t)(c * 32) * cout + n0);
#pragma unroll
        for (int i = 0; i < NT; ++i) rb[i] = *reinterpret_cast<const float4*>(wb + (size_t)(i * (256 / BQ)) * cout * 4 + ob);
    };
    auto commit = [&](int c) {
        float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (ACT) {
            const float4 q_sc = *reinterpret_cast<const float4*>(s_chan + c * 32 + kq), q_sh = *reinterpret_cast<const float4*>(s_chan + cpad + c * 32 + kq);
            sc[0] = q_sc.x; sc[1] = q_sc.y; sc[2] = q_sc.z; sc[3] = q_sc.w;
            sh[0] = q_sh.x; sh[1] = q_sh.y; sh[2] = q_sh.z; sh[3] = q_sh.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float xv[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
            float* d = sA + kq * LDA + arow + 32 * i;
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j * LDA] = act1(xv[j], ACT, sc[j], sh[j]);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) *reinterpret_cast<float4*>(sB + (bk + i * (256 / BQ)) * LDB + bq) = rb[i];
    };
    float bv[NT], csum[NT], csq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        bv[nt] = bias ? bias[n0 + nt * 32 + l31] : 0.f;
        csum[nt] = csq[nt] = 0.f;
    }
    f32x16 acc[NT];
    const float* pa = sA + kh * LDA + wave * 32 + l31;
    const float* pb = sB + kh * LDB + l31;
    const unsigned lo_y = (unsigned)(4 * kh * ldy + n0 + l31) * 4u;
    if ((int)blockIdx.x < ntiles) fetch((int)blockIdx.x, 0);
    __syncthreads();
    for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        for (int c = 0; c < nchunks; ++c) {
            commit(c);
            __syncthreads();
            {
                int nc = c + 1, ntl = tile;
                if (nc == nchunks) { nc = 0; ntl = tile + (int)gridDim.x; }
                if (ntl < ntiles) fetch(ntl, nc);
            }
#pragma unroll
            for (int k0 = 0; k0 < 32; k0 += 8) {
                float av[4], bw[4][NT];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    av[u] = pa[(k0 + 2 * u) * LDA];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bw[u][nt] = pb[(k0 + 2 * u) * LDB + nt * 32];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bw[u][nt], acc[nt], 0, 0, 0);
            }
            __syncthreads();
        }
        const int m0 = (tile << 7) + wave * 32;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if constexpr (POOL) pool32_tile(acc[nt], bv[nt], lane, po, (size_t)(m0 >> 5) * cout + n0 + nt * 32 + l31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[nt][r] + bv[nt];
                *reinterpret_cast<float*>(reinterpret_cast<char*>(Y + (size_t)(m0 + (r & 3) + 8 * (r >> 2)) * ldy) + lo_y + nt * 128) = v;
                csum[nt] += v;
                csq[nt] = __builtin_fmaf(v, v, csq[nt]);
            }
        }
    }
    if (stats) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            csum[nt] += __shfl_xor(csum[nt], 32, 64);
            csq[nt] += __shfl_xor(csq[nt], 32, 64);
            if (lane < 32) {
                sRed[(wave * 2 + 0) * BN + nt * 32 + lane] = csum[nt];
                sRed[(wave * 2 + 1) * BN + nt * 32 + lane] = csq[nt];
            }
        }
        __syncthreads();
        float* ws = stats + (size_t)blockIdx.x * 2 * cout;
        for (int j = t; j < BN; j += 256) {
            float sm = 0.f, q = 0.f;
            for (int w = 0; w < 4; ++w) { sm += sRed[(w * 2 + 0) * BN + j]; q += sRed[(w * 2 + 1) * BN + j]; }
            ws[n0 + j] = sm;
            ws[cout + n0 + j] = q;
        }
    }
}
// ============================================================================================
// Forward with both operands split into three bf16 pieces (r04; OPT-IN: GSPN_MFMA_SPLIT=1.  The default build computes exact fp32
// products).  VERDICT r03 item 8 asked whether the SIMD issue budget of the fp32 MFMAs (16 passes per 32x32x2) can be bought back:
// an fp32 value is hi + mid + lo with three bf16 numbers exactly (8 + 8 + 8 significand bits by truncation; every residual is an
// exact fp32 subtraction), a product of two pieces is exact in fp32, and six v_mfma_f32_32x32x16_bf16 (8 passes per 16 k) --
// hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid, smallest first, fp32 accumulation -- replace eight fp32 MFMAs per 16 k; the
// three dropped products are below 2^-24 of |x.w| each.  Measured against fp64 the result is as good as the fp32 MFMA's
// (tools/clk/bf16_split.hip: max error 2.5e-7 of max |y| both, rms 1.5e-7 vs 1.3e-7), 1.3-1.45x faster on the long layers.
//   * W (cin, cout) is split by the workgroup while it is staged: three (cout, cin) bf16 planes in LDS, k contiguous, pitch 2 cin + 16
//     bytes (conflict-free ds_read_b128), once per persistent workgroup;
//   * the A operand does not pass through LDS: lane (row l & 31, half h = l >> 5) loads cin/2 consecutive floats of ITS row (the next
//     tile's in flight during this one), applies relu(x*scale + shift) with two roundings and splits in registers (11 vector
//     instructions per two elements).  Which k a (half, slot) pair of the MFMA carries is free as long as A and B agree: half h
//     takes k in [h cin/2, (h+1) cin/2);
//   * a wave owns 32 rows x cout columns; no barrier and no LDS write in the row loop.
// Contract as fwd_lean_kernel (Y, per-workgroup partial column sums, optional 32-row pool epilogue); results are NOT bit-identical to
// the fp32 kernels' (another summation order and the dropped terms), which is why this is a switch and not the default.
// ============================================================================================
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ unsigned bf_pack_hi(float x1, float x0) {       // (top half of x1) : (top half of x0)
    return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
}
__device__ __forceinline__ float bf_drop_hi(float x) { return x - __uint_as_float(__float_as_uint(x) & 0xFFFF0000u); }      // exact

template <int K, int N, bool ACT, bool POOL>
__global__ __launch_bounds__(256, 2) void fwd_split_kernel(int rows, const float* __restrict__ X, int ldx, const float* __restrict__ in_scale,
                                                           const float* __restrict__ in_shift, const float* __restrict__ W, const float* __restrict__ bias,
                                                           float* __restrict__ Y, int ldy, float* __restrict__ stats, PoolOut po) {
    constexpr int NT = N / 32, KS = K / 16, KH = K / 2, PITCH = K * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_split[];
    unsigned char* sW = s_split;                                             // [3][N][PITCH]
    float* sC = reinterpret_cast<float*>(s_split + 3 * N * PITCH);           // [2][K]
    float* sRed = sC + 2 * K;                                                // [4][2][N]
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, kh = lane >> 5;
    for (int i = t; i < (K / 2) * (N / 4); i += 256) {                       // item: rows k, k+1 of W x four columns
        const int k = (i / (N / 4)) * 2, n4 = (i % (N / 4)) * 4;
        const float4 w0 = *reinterpret_cast<const float4*>(W + (size_t)k * N + n4), w1 = *reinterpret_cast<const float4*>(W + (size_t)(k + 1) * N + n4);
        const float a0[4] = {w0.x, w0.y, w0.z, w0.w}, a1[4] = {w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float r0 = bf_drop_hi(a0[j]), r1 = bf_drop_hi(a1[j]);
            unsigned char* d = sW + (size_t)(n4 + j) * PITCH + k * 2;
            *reinterpret_cast<unsigned*>(d) = bf_pack_hi(a1[j], a0[j]);
            *reinterpret_cast<unsigned*>(d + (size_t)N * PITCH) = bf_pack_hi(r1, r0);
            *reinterpret_cast<unsigned*>(d + (size_t)2 * N * PITCH) = bf_pack_hi(bf_drop_hi(r1), bf_drop_hi(r0));
        }
    }
    if constexpr (ACT)
        for (int i = t; i < K; i += 256) { sC[i] = in_scale[i]; sC[K + i] = in_shift[i]; }
    float bv[NT], csum[NT], csq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { bv[nt] = bias ? bias[nt * 32 + l31] : 0.f; csum[nt] = csq[nt] = 0.f; }
    const int ntiles = rows >> 7;
    float4 xr[KH / 4], xn[KH / 4];
    auto fetch = [&](int tile, float4* d) {
        const float* p = X + (size_t)((tile << 7) + wave * 32 + l31) * ldx + KH * kh;
#pragma unroll
        for (int i = 0; i < KH / 4; ++i) d[i] = *reinterpret_cast<const float4*>(p + 4 * i);
    };
    if ((int)blockIdx.x < ntiles) fetch((int)blockIdx.x, xn);
    __syncthreads();
    const unsigned char* pb = sW + (size_t)l31 * PITCH + KH * kh * 2;
    for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
#pragma unroll
        for (int i = 0; i < KH / 4; ++i) xr[i] = xn[i];
        if (tile + (int)gridDim.x < ntiles) fetch(tile + (int)gridDim.x, xn);
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 a0 = xr[2 * s], a1 = xr[2 * s + 1];
            float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            if constexpr (ACT) {
                const float4 s0 = *reinterpret_cast<const float4*>(sC + KH * kh + 8 * s), s1 = *reinterpret_cast<const float4*>(sC + KH * kh + 8 * s + 4);
                const float4 h0 = *reinterpret_cast<const float4*>(sC + K + KH * kh + 8 * s), h1 = *reinterpret_cast<const float4*>(sC + K + KH * kh + 8 * s + 4);
                const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float y = __fadd_rn(__fmul_rn(v[j], sc[j]), sh[j]);       // two roundings: act1's operand
                    v[j] = y > 0.f ? y : 0.f;
                }
            }
            u32x4 ap[3];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x0 = v[2 * j], x1 = v[2 * j + 1];
                const float r0 = bf_drop_hi(x0), r1 = bf_drop_hi(x1);
                ap[0][j] = bf_pack_hi(x1, x0);
                ap[1][j] = bf_pack_hi(r1, r0);
                ap[2][j] = bf_pack_hi(bf_drop_hi(r1), bf_drop_hi(r0));
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                u32x4 bp[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) bp[p] = *reinterpret_cast<const u32x4*>(pb + (size_t)(p * N + nt * 32) * PITCH + 16 * s);
#define MF(ia, ib) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ap[ia]), __builtin_bit_cast(bf16x8, bp[ib]), acc[nt], 0, 0, 0)
                MF(2, 0); MF(0, 2); MF(1, 1); MF(1, 0); MF(0, 1); MF(0, 0);
#undef MF
                __builtin_amdgcn_sched_barrier(0);               // (or hipcc hoists every step's B fragments to the top: 308 registers, one workgroup per CU)
            }
        }
        const int m0 = (tile << 7) + wave * 32;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if constexpr (POOL) pool32_tile(acc[nt], bv[nt], lane, po, (size_t)(m0 >> 5) * N + nt * 32 + l31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float y = acc[nt][r] + bv[nt];
                Y[(size_t)(m0 + 4 * kh + (r & 3) + 8 * (r >> 2)) * ldy + nt * 32 + l31] = y;
                csum[nt] += y;
                csq[nt] = __builtin_fmaf(y, y, csq[nt]);
            }
        }
    }
    if (stats) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            csum[nt] += __shfl_xor(csum[nt], 32, 64);
            csq[nt] += __shfl_xor(csq[nt], 32, 64);
            if (lane < 32) { sRed[(wave * 2 + 0) * N + nt * 32 + lane] = csum[nt]; sRed[(wave * 2 + 1) * N + nt * 32 + lane] = csq[nt]; }
        }
        __syncthreads();
        float* ws = stats + (size_t)blockIdx.x * 2 * N;
        for (int j = t; j < N; j += 256) {
            float sm = 0.f, q = 0.f;
            for (int w = 0; w < 4; ++w) { sm += sRed[(w * 2 + 0) * N + j]; q += sRed[(w * 2 + 1) * N + j]; }
            ws[j] = sm;
            ws[N + j] = q;
        }
    }
}
// the shapes the split kernel takes: cin 32 / 64, cout 32 / 64 / 128, rows a multiple of 128 and at least 65536 (below that the layer is launch- and
// latency-bound, not issue-bound), dense W, 16-byte rows
static bool fwd_split_go(long rows, int cin, int cout, const float* X, int ldx, const float* in_scale, const float* in_shift, const float* W, const float* bias,
                         float* Y, int ldy, float* stats, PoolOut po, hipStream_t st) {
    static const int on = env_int("GSPN_MFMA_SPLIT", 0);
    if (!on || rows < 65536 || (rows & 127) || rows >= (1L << 31) || !vec_ok(X, ldx) || !vec_ok(W, cout)) return false;
    if (!((cin == 32 || cin == 64) && (cout == 32 || cout == 64 || cout == 128))) return false;
    const dim3 g(fwd_blocks(rows, cout));
#define FS_GO(K_, N_, A_, P_) do { const size_t dyn = 3 * N_ * (K_ * 2 + 16) + 2 * K_ * 4 + 8 * N_ * 4;                                              \
        hipLaunchKernelGGL((fwd_split_kernel<K_, N_, A_, P_>), g, dim3(256), dyn, st, (int)rows, X, ldx, in_scale, in_shift, W, bias, Y, ldy, stats, po); } while (0)
#define FS_P(K_, N_, A_) do { if (po.vmax) FS_GO(K_, N_, A_, true); else FS_GO(K_, N_, A_, false); } while (0)
#define FS_A(K_, N_) do { if (in_scale) FS_P(K_, N_, true); else FS_P(K_, N_, false); } while (0)
#define FS_N(K_) do { if (cout == 32) FS_A(K_, 32); else if (cout == 64) FS_A(K_, 64); else FS_A(K_, 128); } while (0)
    if (cin == 32) FS_N(32); else FS_N(64);
#undef FS_N
#undef FS_A
#undef FS_P
#undef FS_GO
    return true;
}

static int mlp_fwd_impl(long rows, int cin, int cout, const float* X, int ldx, const float* in_scale, const float* in_shift,
                        const float* W, const float* bias, float* Y, int ldy, float* stats, PoolOut po, void* stream,
                        const GatherSrc* gsrc = nullptr) {
    // gsrc: the input rows are virtual (GatherSrc); cin is then the INTERNAL width 4*cq + 4, X is unused, only the streaming kernel applies
    if (rows < 0 || cin <= 0 || cout <= 0 || (!gsrc && ldx < cin) || ldy < cout) return GSPN_ERR_ARG;
    if ((in_scale == nullptr) != (in_shift == nullptr)) return GSPN_ERR_ARG;
    if (cin > MAXCH || cout > MAXCH) return GSPN_ERR_UNSUPPORTED;
    if (rows == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    {
        if (!gsrc && (!po.vmax || !(rows & 31)) && fwd_split_go(rows, cin, cout, X, ldx, in_scale, in_shift, W, bias, Y, ldy, stats, po, st)) return gspn_launch_status();
        // short layers (<= GSPN_SHORT_ROWS = 8192 rows): split-K over the waves of a workgroup, operands straight from global memory (mlp_short.hip, r05)
        if (!gsrc && fwd_short_rows(rows) && gspn_fwd_short_go(rows, cin, cout, X, ldx, in_scale, in_shift, W, bias, Y, ldy, stats, fwd_blocks(rows, cout), po, st))
            return gspn_launch_status();
        static const int lean_on = env_int("GSPN_FWD_LEAN", 1);             // (A/B hook)
        static const int lean_bn = env_int("GSPN_FWD_LEAN_BN", 0);          // (tuning hook)
        // (beyond ~0.75 M rows the eight 32-column blocks of a 256-column layer re-read X from beyond the L2s: 1 M x 128 -> 256 946 us against
        //  874 for the 128-column register-staged kernel, 1 M x 64 -> 128 274 against 259 for the streaming kernel; 0.5 M rows: 431 / 437, 137 / 151)
        // Beyond that the layer runs as `nsplit` launches over row slabs of at most 0.75 M rows, each with its share of the partial-sum rows (r04: the
        // configs[3] shard's 1 M x 128 -> 256 layer, 929 us on the register-staged kernel -> 2 x 431)
        int nsplit = 1;
        if (rows > 786432 && cout >= 256) {
            nsplit = (int)((rows + 786431) / 786432);
            const unsigned nb = fwd_blocks(rows, cout);
            if (nb % nsplit || rows % ((long)nsplit * 128)) nsplit = 0;
        } else if (rows > 786432) nsplit = 0;
        if (lean_on && !gsrc && nsplit >= 1 && vec_ok(X, ldx) && vec_ok(W, cout) && !(rows & 127) && !(cin & 31) && !(cout & 31) && rows * (long)std::max(ldx, ldy) < (1L << 30) && (long)cin * cout < (1L << 30) &&
            (!po.vmax || !(rows & 31))) {
            // 32-column blocks: measured best on every layer shape of the benchmark and of the configs[3] shard (tools/fwd_ablate.py with
            // GSPN_FWD_LEAN_BN = 32 / 64): 131072 x 64 -> 128 35 us against 39 (and 44 for the LDS-DMA streaming kernel), 32768 x 128 -> 256 30 against 40
            int bn = 32;
            if (lean_bn && cout % lean_bn == 0 && lean_bn <= 64) bn = lean_bn;
            const unsigned nbs = fwd_blocks(rows, cout) / nsplit;
            const long srows = rows / nsplit;
            const dim3 g(nbs, cout / bn);
            const size_t dyn = sizeof(float) * 2 * chan_pad(cin);
            for (int sl = 0; sl < nsplit; ++sl) {
                const float* Xs = X + (size_t)sl * srows * ldx;
                float* Ys = Y + (size_t)sl * srows * ldy;
                float* sts = stats ? stats + (size_t)sl * nbs * 2 * cout : nullptr;
                PoolOut pos = po;
                if (po.vmax) { pos.vmax = po.vmax + (size_t)sl * (srows / 32) * cout; pos.amax = po.amax + (size_t)sl * (srows / 32) * cout; }
#define FL_GO(NT_, A_, P_) hipLaunchKernelGGL((fwd_lean_kernel<NT_, A_, P_>), g, dim3(256), dyn, st, (int)srows, cin, cout, Xs, ldx, in_scale, in_shift, W, bias, Ys, ldy, sts, pos)
#define FL_P(NT_, A_) do { if (po.vmax) FL_GO(NT_, A_, true); else FL_GO(NT_, A_, false); } while (0)
#define FL_A(NT_) do { if (in_scale) FL_P(NT_, true); else FL_P(NT_, false); } while (0)
                if (bn == 64) FL_A(2); else FL_A(1);
#undef FL_A
#undef FL_P
#undef FL_GO
            }
            return gspn_launch_status();
        }
    }
    {
        // streaming kernel: needs 16-byte rows, 32-bit offsets and W + two row tiles within 80 KB of LDS (>= 2 workgroups per CU)
        const int BNs = cout <= 32 ? 32 : (cout <= 64 ? 64 : 128);
        const int yt = (cout + BNs - 1) / BNs;
        const int QX = (cin + 3) / 4;
        const long lds4 = 16L * QX * (BNs + 64 * 4), lds2 = 16L * QX * (BNs + 64 * 2);      // bytes with TRG = 4 / 2 (+ 8*KP for the constants)
        int trg = 0;
        if (lds4 + 32L * QX <= 53 * 1024) trg = 4;
        else if (BNs >= 64 && lds2 + 32L * QX <= 80 * 1024) trg = 2;
        if (trg && (32 * trg * QX) % 64 != 0) trg = 0;
        if (trg && 32 * trg * QX / 64 > 32) trg = 0;                                      // <= 8 pieces per wave
        const bool src_ok = gsrc ? true : (vec_ok(X, ldx) && ldx >= 4 && rows * (long)ldx < (1L << 31));
        if (gsrc && !trg) return GSPN_ERR_UNSUPPORTED;
        if (trg && src_ok && rows < (1L << 31) && rows * (long)ldy < (1L << 31) && (gsrc || getenv("GSPN_FWD_NO_STREAM") == nullptr)) {
            const size_t dyn = (size_t)(trg == 4 ? lds4 : lds2) + 32u * QX;
            const long ntiles = (rows + 32 * trg - 1) / (32 * trg);
            long bpc = (160L * 1024) / (long)(dyn + 2 * trg * BNs * 4 + 512);
            static int bpc_cap = 0;
            if (!bpc_cap) { const char* e = getenv("GSPN_FWD_BPC"); bpc_cap = e ? atoi(e) : 4; if (bpc_cap < 1) bpc_cap = 4; }
            if (bpc > bpc_cap) bpc = bpc_cap;
            if (bpc < 1) bpc = 1;
            const unsigned nparts = fwd_blocks(rows, cout);
            long gx = (long)GSPN_PLAN_CUS * bpc / yt;
            if (gx > ntiles) gx = ntiles;
            if (gx > (long)nparts) gx = nparts;
            if (gx < 1) gx = 1;
#define FWDS_GO1(BN_, TRG_, G_, GS_)                                                                                                  \
            do {                                                                                                                       \
                static bool attr_done = false;                                                                                         \
                if (!attr_done) {                                                                                                      \
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fwd_stream_kernel<BN_, TRG_, G_>),           \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);                         \
                    if (e != hipSuccess) return (int)e;                                                                                \
                    attr_done = true;                                                                                                  \
                }                                                                                                                      \
                hipLaunchKernelGGL((mlp_fwd_stream_kernel<BN_, TRG_, G_>), dim3((unsigned)gx, yt), dim3(256), dyn, st, (int)rows, cin, cout, X, ldx, \
                                   in_scale, in_shift, W, bias, Y, ldy, stats, (int)nparts, po, GS_, BwdPool{});                                  \
                return gspn_launch_status();                                                                                           \
            } while (0)
#define FWDS_GO(BN_, TRG_)                                                                                                             \
            do {                                                                                                                       \
                if (gsrc) FWDS_GO1(BN_, TRG_, true, *gsrc);                                                                            \
                else FWDS_GO1(BN_, TRG_, false, GatherSrc{});                                                                          \
            } while (0)
            if (BNs == 32 && trg == 4) FWDS_GO(32, 4);
            if (BNs == 64 && trg == 4) FWDS_GO(64, 4);
            if (BNs == 64 && trg == 2) FWDS_GO(64, 2);
            if (BNs == 128 && trg == 4) FWDS_GO(128, 4);
            if (BNs == 128 && trg == 2) FWDS_GO(128, 2);
#undef FWDS_GO
#undef FWDS_GO1
        }
    }
    if (gsrc) return GSPN_ERR_UNSUPPORTED;
    const bool v = vec_ok(X, ldx) && vec_ok(W, cout);
#define FWD_LAUNCH(BN_, V_, YT_)                                                                                                   \
    hipLaunchKernelGGL((mlp_fwd_kernel<BN_, V_>), dim3(fwd_blocks(rows, cout), YT_), dim3(256), sizeof(float) * 2 * chan_pad(cin), st, rows, cin, cout, X, ldx, \
                       in_scale, in_shift, W, bias, Y, ldy, stats, po)
    const int bn = pick_bn(rows, cout, "GSPN_FWD_FORCE_BN");
    const int yt = (cout + bn - 1) / bn;
    if (bn == 32) { if (v) FWD_LAUNCH(32, true, yt); else FWD_LAUNCH(32, false, yt); }
    else if (bn == 64) { if (v) FWD_LAUNCH(64, true, yt); else FWD_LAUNCH(64, false, yt); }
    else { if (v) FWD_LAUNCH(128, true, yt); else FWD_LAUNCH(128, false, yt); }
#undef FWD_LAUNCH
    return gspn_launch_status();
}
extern "C" int gspn_mlp_fwd(long rows, int cin, int cout, const float* X, int ldx, const float* in_scale, const float* in_shift,
                            const float* W, const float* bias, float* Y, int ldy, float* stats, void* stream) {
    return mlp_fwd_impl(rows, cin, cout, X, ldx, in_scale, in_shift, W, bias, Y, ldy, stats, PoolOut{nullptr, nullptr}, stream);
}
// gspn_mlp_fwd that also leaves, per group of 32 consecutive rows and channel, the largest raw output and its row offset (each
// (rows/32, cout)): the first half of the max-pool of pointnet_util.py:123-124 for nsample = 32; gspn_pool32_select is the second, once
// the batch statistics of this layer are known.  rows must be a multiple of 32.
extern "C" int gspn_mlp_fwd_pool32(long rows, int cin, int cout, const float* X, int ldx, const float* in_scale, const float* in_shift,
                                   const float* W, const float* bias, float* Y, int ldy, float* stats, float* vmax, int* amax, void* stream) {
    if (!vmax || !amax || (rows & 31)) return GSPN_ERR_ARG;
    return mlp_fwd_impl(rows, cin, cout, X, ldx, in_scale, in_shift, W, bias, Y, ldy, stats, PoolOut{vmax, amax}, stream);
}

// ============================================================================================
// BN finalize / element-wise tails
// ============================================================================================
// wave-wide sum of a double (all lanes get the total)
// The butterfly v += v[lane ^ s], s = 32, 16, 8, 4, 2, 1.  The four steps inside a 16-lane row go through DPP moves (a few cycles each) instead
// of ds_bpermute (an LDS-crossbar round trip each, twelve of them in a dependent chain for one double): row_ror:8 / row_ror:4 stand in for
// lane ^ 8 / lane ^ 4 -- after the previous step lanes i and i ^ 8 (i ^ 4) hold the same bits, so the rotated partner holds exactly what the
// xor partner holds -- and quad_perm for lane ^ 2, lane ^ 1.  Same operands in the same order at every step: bit-identical sums.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += __shfl_xor(v, 32, 64);
    v += __shfl_xor(v, 16, 64);
    v += dpp_f64<0x128>(v);          // row_ror:8
    v += dpp_f64<0x124>(v);          // row_ror:4
    v += dpp_f64<0x4E>(v);           // quad_perm [2,3,0,1]
    v += dpp_f64<0xB1>(v);           // quad_perm [1,0,3,2]
    return v;
}
// sum over the partial rows p = t, t + 256, ... of src[p * ld + off] and src[p * ld + off2] in double, ascending p.  The loads of FOUR rows (eight
// values) are issued before the first is consumed: the rows were written by other XCDs in the kernel before, so every dependent batch is a
// round trip to the Infinity Cache -- a loop that adds as it loads pays one per iteration (r05: bn_finalize 4.8 us in the step, 1.6 of it the node).
__device__ __forceinline__ void part_rows_sum2(const float* __restrict__ src, size_t ld, int off, int off2, int nparts, int t, double& a0, double& a1) {
    for (int base = 0; base < nparts; base += 1024) {
        float u0[4], u1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = base + t + 256 * u;
            const int pc = p < nparts ? p : nparts - 1;              // clamped, unconditional: all eight loads in flight together
            u0[u] = src[(size_t)pc * ld + off];
            u1[u] = src[(size_t)pc * ld + off2];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (base + t + 256 * u < nparts) { a0 += (double)u0[u]; a1 += (double)u1[u]; }
    }
}
// one WORKGROUP per channel: 256 threads stride over the forward's per-block partials (double accumulation), then finalise
__global__ __launch_bounds__(256) void bn_finalize_kernel(long rows, int c, const float* __restrict__ stats, int nparts, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float decay, int is_training,
                                                          float* __restrict__ moving_mean, float* __restrict__ moving_var,
                                                          float* __restrict__ mean, float* __restrict__ var, float* __restrict__ scale, float* __restrict__ shift,
                                                          const float* __restrict__ pivot) {
    // pivot (optional): the partial sums are of (x - pivot[j]) and its square (gspn_bn_colsum with a pivot row: no cancellation in
    // E[x^2] - mean^2 when |mean| >> std); the mean is shifted back here, the variance needs no correction
    __shared__ double sh2[2][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = blockIdx.x;
    // (the channel's constants are asked for BEFORE the partial rows: behind the reduction they were one more dependent memory round trip)
    float mm0 = 0.f, mv0 = 0.f, g = 1.f, be = 0.f, pv = 0.f;
    if (threadIdx.x == 0) {
        if (moving_mean) mm0 = moving_mean[j];
        if (moving_var) mv0 = moving_var[j];
        if (gamma) g = gamma[j];
        if (beta) be = beta[j];
        if (pivot) pv = pivot[j];
    }
    double mu, v;
    if (is_training) {
        double a0 = 0.0, a1 = 0.0;
        part_rows_sum2(stats, (size_t)2 * c, j, c + j, nparts, (int)threadIdx.x, a0, a1);
        a0 = wave_sum_f64(a0);
        a1 = wave_sum_f64(a1);
        if (lane == 0) { sh2[0][wv] = a0; sh2[1][wv] = a1; }
        __syncthreads();
        a0 = (sh2[0][0] + sh2[0][1]) + (sh2[0][2] + sh2[0][3]);
        a1 = (sh2[1][0] + sh2[1][1]) + (sh2[1][2] + sh2[1][3]);
        mu = a0 / (double)rows;
        v = a1 / (double)rows - mu * mu;                   // biased variance (tf.nn.moments)
        if (v < 0.0) v = 0.0;
        if (pivot) mu += (double)pv;
    } else {
        mu = mm0;
        v = mv0;
    }
    if (threadIdx.x != 0) return;
    if (is_training) {
        if (moving_mean) moving_mean[j] = (float)((double)mm0 * decay + mu * (1.0 - (double)decay));
        if (moving_var) moving_var[j] = (float)((double)mv0 * decay + v * (1.0 - (double)decay));
    }
    const float inv = (float)(1.0 / sqrt(v + (double)eps)) * g;       // inv = rsqrt(var+eps)*gamma
    mean[j] = (float)mu;
    var[j] = (float)v;
    scale[j] = inv;
    shift[j] = be - (float)mu * inv;                                     // beta - mean*inv
}
// the same for column sums that did not come from gspn_mlp_fwd: nparts partial rows [nparts][2][c]
extern "C" int gspn_bn_finalize_parts_pivot(long rows, int c, const float* stats, int nparts, const float* gamma, const float* beta, float eps, float decay,
                                            int is_training, float* moving_mean, float* moving_var, float* mean, float* var, float* scale, float* shift,
                                            const float* pivot, void* stream) {
    if (rows <= 0 || c <= 0 || !mean || !var || !scale || !shift || nparts < 0) return GSPN_ERR_ARG;
    if (is_training && !stats) return GSPN_ERR_ARG;
    if (!is_training && (!moving_mean || !moving_var)) return GSPN_ERR_ARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(c), dim3(256), 0, (hipStream_t)stream, rows, c, stats, nparts, gamma, beta,
                       eps, decay, is_training, moving_mean, moving_var, mean, var, scale, shift, pivot);
    return gspn_launch_status();
}
extern "C" int gspn_bn_finalize_parts(long rows, int c, const float* stats, int nparts, const float* gamma, const float* beta, float eps, float decay,
                                      int is_training, float* moving_mean, float* moving_var, float* mean, float* var, float* scale, float* shift,
                                      void* stream) {
    return gspn_bn_finalize_parts_pivot(rows, c, stats, nparts, gamma, beta, eps, decay, is_training, moving_mean, moving_var, mean, var, scale, shift, nullptr, stream);
}
extern "C" int gspn_bn_finalize(long rows, int c, const float* stats, const float* gamma, const float* beta, float eps, float decay,
                                int is_training, float* moving_mean, float* moving_var, float* mean, float* var,
                                float* scale, float* shift, void* stream) {
    if (rows <= 0 || c <= 0 || !mean || !var || !scale || !shift) return GSPN_ERR_ARG;
    if (is_training && !stats) return GSPN_ERR_ARG;
    if (!is_training && (!moving_mean || !moving_var)) return GSPN_ERR_ARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(c), dim3(256), 0, (hipStream_t)stream, rows, c, stats, (int)fwd_blocks(rows, c), gamma, beta,
                       eps, decay, is_training, moving_mean, moving_var, mean, var, scale, shift, (const float*)nullptr);
    return gspn_launch_status();
}

__global__ void bnrelu_maxpool_kernel(long total, int ns, int c, const float* __restrict__ Y, int ldy, const float* __restrict__ scale,
                                      const float* __restrict__ shift, float* __restrict__ out, int* __restrict__ arg) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long g = i / c;
        const int col = (int)(i - g * c);
        const float sc = scale ? scale[col] : 1.f, sh = scale ? shift[col] : 0.f;
        const float* p = Y + g * ns * ldy + col;
        float best = 0.f;
        int bi = 0;
        for (int k = 0; k < ns; ++k) {
            float z = p[(size_t)k * ldy];
            if (scale) { z = z * sc + sh; z = z > 0.f ? z : 0.f; }
            if (k == 0 || z > best) { best = z; bi = k; }
        }
        out[i] = best;
        if (arg) arg[i] = bi;
    }
}
// same, four channels per thread (c, ldy multiples of 4, 16-byte rows) and eight row loads in flight: the scalar kernel above is
// latency-bound (1.6 TB/s on the 134 MB of SA level 1)
__global__ void bnrelu_maxpool4_kernel(long total4, int ns, int c4, const float* __restrict__ Y, int ldy, const float* __restrict__ scale,
                                       const float* __restrict__ shift, float* __restrict__ out, int* __restrict__ arg) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long g = i / c4;
        const int col = (int)(i - g * c4) * 4;
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (scale) { sc = *reinterpret_cast<const float4*>(scale + col); sh = *reinterpret_cast<const float4*>(shift + col); }
        const float* p = Y + g * ns * ldy + col;
        float4 best = make_float4(0.f, 0.f, 0.f, 0.f);
        int4 bi = make_int4(0, 0, 0, 0);
        for (int k0 = 0; k0 < ns; k0 += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)min(k0 + u, ns - 1) * ldy);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u;
                if (k < ns) {
                    float4 z = v[u];
                    if (scale) {
                        z.x = z.x * sc.x + sh.x; z.x = z.x > 0.f ? z.x : 0.f;
                        z.y = z.y * sc.y + sh.y; z.y = z.y > 0.f ? z.y : 0.f;
                        z.z = z.z * sc.z + sh.z; z.z = z.z > 0.f ? z.z : 0.f;
                        z.w = z.w * sc.w + sh.w; z.w = z.w > 0.f ? z.w : 0.f;
                    }
                    if (k == 0 || z.x > best.x) { best.x = z.x; bi.x = k; }
                    if (k == 0 || z.y > best.y) { best.y = z.y; bi.y = k; }
                    if (k == 0 || z.z > best.z) { best.z = z.z; bi.z = k; }
                    if (k == 0 || z.w > best.w) { best.w = z.w; bi.w = k; }
                }
            }
        }
        *reinterpret_cast<float4*>(out + g * (4L * c4) + col) = best;
        if (arg) *reinterpret_cast<int4*>(arg + g * (4L * c4) + col) = bi;
    }
}
extern "C" int gspn_bnrelu_maxpool(long groups, int ns, int c, const float* Y, int ldy, const float* scale, const float* shift,
                                   float* out, int* arg, void* stream) {
    if (groups < 0 || ns <= 0 || c <= 0 || ldy < c) return GSPN_ERR_ARG;
    if ((scale == nullptr) != (shift == nullptr)) return GSPN_ERR_ARG;
    const long total = groups * c;
    if (total == 0) return 0;
    const bool v4 = (c % 4 == 0) && (ldy % 4 == 0) && ((uintptr_t)Y % 16 == 0) && ((uintptr_t)out % 16 == 0) && (!arg || (uintptr_t)arg % 16 == 0) &&
                    (!scale || ((uintptr_t)scale % 16 == 0 && (uintptr_t)shift % 16 == 0));
    if (v4) hipLaunchKernelGGL(bnrelu_maxpool4_kernel, dim3(grid_for(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, total / 4, ns, c / 4, Y, ldy, scale, shift, out, arg);
    else hipLaunchKernelGGL(bnrelu_maxpool_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, total, ns, c, Y, ldy, scale, shift, out, arg);
    return gspn_launch_status();
}
__global__ void bnrelu_apply_kernel(long total, int c, const float* __restrict__ Y, int ldy, const float* __restrict__ scale,
                                    const float* __restrict__ shift, float* __restrict__ out, int ldo) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / c;
        const int col = (int)(i - row * c);
        float z = Y[row * ldy + col];
        if (scale) { z = z * scale[col] + shift[col]; z = z > 0.f ? z : 0.f; }
        out[row * ldo + col] = z;
    }
}
// the same, four channels per thread (16-byte rows): the scalar form above pays a 64-bit division and 4-byte accesses per element (4.1 TB/s on the
// 134 MB of the extractor's output; this one streams)
__global__ void bnrelu_apply4_kernel(long total4, int c4, const float* __restrict__ Y, int ldy, const float* __restrict__ scale,
                                     const float* __restrict__ shift, float* __restrict__ out, int ldo) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long row = i / c4;
        const int col = (int)(i - row * c4) * 4;
        float4 z = *reinterpret_cast<const float4*>(Y + row * ldy + col);
        if (scale) {
            const float4 sc = *reinterpret_cast<const float4*>(scale + col), sh = *reinterpret_cast<const float4*>(shift + col);
            z.x = z.x * sc.x + sh.x; z.x = z.x > 0.f ? z.x : 0.f;
            z.y = z.y * sc.y + sh.y; z.y = z.y > 0.f ? z.y : 0.f;
            z.z = z.z * sc.z + sh.z; z.z = z.z > 0.f ? z.z : 0.f;
            z.w = z.w * sc.w + sh.w; z.w = z.w > 0.f ? z.w : 0.f;
        }
        *reinterpret_cast<float4*>(out + row * ldo + col) = z;
    }
}
extern "C" int gspn_bnrelu_apply(long rows, int c, const float* Y, int ldy, const float* scale, const float* shift, float* out, int ldo, void* stream) {
    if (rows < 0 || c <= 0 || ldy < c || ldo < c) return GSPN_ERR_ARG;
    if ((scale == nullptr) != (shift == nullptr)) return GSPN_ERR_ARG;
    const long total = rows * c;
    if (total == 0) return 0;
    if (c % 4 == 0 && ldy % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)Y | (uintptr_t)out | (uintptr_t)scale | (uintptr_t)shift) % 16 == 0) {
        hipLaunchKernelGGL(bnrelu_apply4_kernel, dim3(grid_for(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, total / 4, c / 4, Y, ldy, scale, shift, out, ldo);
        return gspn_launch_status();
    }
    hipLaunchKernelGGL(bnrelu_apply_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, total, c, Y, ldy, scale, shift, out, ldo);
    return gspn_launch_status();
}

// ============================================================================================
// Backward pass A:  one read of (X, Y, dz) per layer ->
//    red[0:c]   += sum_rows dyh                 red[c:2c] += sum_rows dyh*xhat        (double atomics)
//    G1(cin,cout) += A^T.dyh     Gx(cin,cout) += A^T.xhat     g3(cin) += A^T.1         (fp32 atomics)
// with dyh = dz*[scale*y+shift > 0], xhat = (y-mean)*rstd, A = act(X).
// M = cin tile (32*WM), N = cout tile (BN), K = rows (split over grid.x chunks).  Both operands are
// K-major in memory (one row of X / of Y is one k), so staging is a straight float4 copy.
// WANT_GX = training-mode BN (otherwise dW = cA (.) G1 and Gx is skipped).
// ============================================================================================
// Tile shape: MT x NTT tiles of 32x32 (BM = 32*MT rows of dW, BN = 32*NTT columns), right-sized to the layer so
// no MFMA is spent on padding.  T = MT*NTT tiles are dealt round-robin to the 4 waves; when T < 4 the spare waves
// split K instead (each takes a slice of the staged rows) so all four SIMDs work.  TKW rows are staged per
// iteration (more for narrow layers, to keep ~25-50 KB in flight per workgroup).
template <int MT, int NTT, int TKW, bool VEC, bool WANT_GX, bool POOLED>
__global__ __launch_bounds__(256) void mlp_bwd_wgrad_kernel(long rows, int cin, int cout, gspn_dy_args a, const float* __restrict__ X, int ldx,
                                                            const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                            const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                            float* __restrict__ RP, float* __restrict__ GP, float* __restrict__ PP,
                                                            long rows_per_chunk, int nslots) {
    // per-chunk partial outputs (no hot-spot atomics; summed by the finalize kernels):
    //   RP[chunk][2][cout] = (sum dyh, sum dyh*xhat)   GP[chunk][cin] = column sums of A   PP[chunk][2][cin][cout] = (A^T.dyh, A^T.xhat)
    constexpr int BM = 32 * MT, BN = 32 * NTT;
    constexpr int T = MT * NTT;
    constexpr int WK = T >= 4 ? 1 : (T == 2 ? 2 : (T == 1 ? 4 : 1));   // K split among waves when tiles are scarce
    constexpr int TPW = (T * WK + 3) / 4;                                // tiles per wave
    constexpr int LDAW = BM + 4;
    constexpr int LDB = BN + 4;
    constexpr int AQ = BM / 4, BQ = BN / 4;
    constexpr int NA = (TKW * AQ + 255) / 256;    // A float4 per thread per chunk
    constexpr int NBV = (TKW * BQ + 255) / 256;   // B float4 per thread per chunk
    constexpr int ASTEP = 256 / AQ, BSTEP = 256 / BQ;   // row stride between a thread's consecutive quads (256 % AQ == 0 needs AQ | 256)
    static_assert(256 % AQ == 0 || MT == 3, "A quad mapping");
    static_assert((TKW * AQ) % 256 == 0 && (TKW * BQ) % 256 == 0, "every thread stages whole quads: no bounds check on the staged row");
    __shared__ __attribute__((aligned(16))) float sA[TKW * LDAW];     // [k=row][m=cin]
    __shared__ __attribute__((aligned(16))) float sB[TKW * LDB];      // [k=row][n=cout]  dyh
    __shared__ __attribute__((aligned(16))) float sX[WANT_GX ? TKW * LDB : 4];   // [k=row][n=cout]  xhat
    extern __shared__ __attribute__((aligned(16))) float s_chan[];         // [2][cpad]
    const int cpad = (cin + 3) / 4 * 4 + 4;
    float* sSc = s_chan;
    float* sSh = s_chan + cpad;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.z * BN;
    const bool act = in_scale != nullptr;
    stage_chan(sSc, in_scale, cin, 1.f, cpad);
    stage_chan(sSh, in_shift, cin, 0.f, cpad);
    __syncthreads();
    const long r_begin = blockIdx.x * rows_per_chunk;
    const long r_end = r_begin + rows_per_chunk < rows ? r_begin + rows_per_chunk : rows;
    // quad mapping: item f = t + 256*i -> row f / Q, quad f % Q.  When Q divides 256 the quad is fixed per thread.
    constexpr bool AFIX = (256 % AQ) == 0, BFIX = (256 % BQ) == 0;
    static_assert(BFIX, "BN must be 32, 64 or 128");
    const int b_nq = (t % BQ) * 4, b_kk0 = t / BQ;
    float4 bsc = make_float4(1, 1, 1, 1), bsh = make_float4(0, 0, 0, 0), bmu = make_float4(0, 0, 0, 0), brs = make_float4(1, 1, 1, 1);
    {
        float* psc = &bsc.x; float* psh = &bsh.x; float* pmu = &bmu.x; float* prs = &brs.x;
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + b_nq + j;
            if (n < cout) {
                psc[j] = a.scale[n]; psh[j] = a.shift[n];
                if (mean) pmu[j] = mean[n];
                if (var) prs[j] = (float)(1.0 / sqrt((double)var[n] + (double)eps));
            }
        }
    }
    float4 s_r0 = make_float4(0, 0, 0, 0), s_r1 = make_float4(0, 0, 0, 0);

    f32x16 acc1[TPW], accx[WANT_GX ? TPW : 1];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc1[i][r] = 0.f; if (WANT_GX) accx[i][r] = 0.f; }

    float4 ra[NA], ry[NBV];
    DzRaw rz[NBV];
    const long r_last = r_end - 1;
    auto fetch = [&](long k0) {                       // raw, unconditional loads (clamped rows): all in flight until commit()
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = t + 256 * i;
            const int kk = f / AQ, mq = (f - kk * AQ) * 4;
            const long row = k0 + kk;
            ra[i] = load4_raw<VEC>(X, row < r_end ? row : r_last, ldx, m0 + mq, cin);
        }
#pragma unroll
        for (int i = 0; i < NBV; ++i) {
            const long row = k0 + b_kk0 + BSTEP * i;
            const long rc = row < r_end ? row : r_last;
            ry[i] = load4_raw<VEC>(a.Y, rc, a.ldy, n0 + b_nq, cout);
            rz[i] = dz4_raw<VEC, POOLED>(a, rc, n0 + b_nq, cout);
        }
    };
    auto commit = [&](long k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int f = t + 256 * i;
            const int kk = f / AQ, mq = (f - kk * AQ) * 4;
            Chan4 cha;
            cha.sc = lds4(sSc, m0 + mq, cpad);
            cha.sh = lds4(sSh, m0 + mq, cpad);
            *reinterpret_cast<float4*>(sA + kk * LDAW + mq) = act4(ra[i], act, cha, m0 + mq, cin, (k0 + kk) < r_end);
        }
#pragma unroll
        for (int i = 0; i < NBV; ++i) {
            const int kk = b_kk0 + BSTEP * i;
            {
                const long row = k0 + kk;
                const bool live = row < r_end;
                const float4 y = mask4(ry[i], n0 + b_nq, cout, live);
                const float4 dz = mask4(dz4_resolve<POOLED>(a, rz[i], row < r_end ? row : r_last), n0 + b_nq, cout, live);
                float4 dyh, xh;
                dyh.x = (y.x * bsc.x + bsh.x) > 0.f ? dz.x : 0.f;
                dyh.y = (y.y * bsc.y + bsh.y) > 0.f ? dz.y : 0.f;
                dyh.z = (y.z * bsc.z + bsh.z) > 0.f ? dz.z : 0.f;
                dyh.w = (y.w * bsc.w + bsh.w) > 0.f ? dz.w : 0.f;
                xh = mask4(make_float4((y.x - bmu.x) * brs.x, (y.y - bmu.y) * brs.y, (y.z - bmu.z) * brs.z, (y.w - bmu.w) * brs.w), n0 + b_nq, cout, live);
                *reinterpret_cast<float4*>(sB + kk * LDB + b_nq) = dyh;
                if (WANT_GX) *reinterpret_cast<float4*>(sX + kk * LDB + b_nq) = xh;
                if (blockIdx.y == 0) {
                    s_r0.x += dyh.x; s_r0.y += dyh.y; s_r0.z += dyh.z; s_r0.w += dyh.w;
                    s_r1.x += dyh.x * xh.x; s_r1.y += dyh.y * xh.y; s_r1.z += dyh.z * xh.z; s_r1.w += dyh.w * xh.w;
                }
            }
        }
    };
    (void)AFIX; (void)ASTEP;

    // g3 = column sums of A: accumulated from LDS by the first BM threads (cheap: TKW adds per iteration)
    float g3acc = 0.f;

    const long nit = r_begin < r_end ? (r_end - r_begin + TKW - 1) / TKW : 0;
    for (long it = 0; it <= nit; ++it) {
        if (it < nit) fetch(r_begin + it * TKW);
        if (it > 0) {
            if (blockIdx.z == 0 && t < BM) {
#pragma unroll 8
                for (int kk = 0; kk < TKW; ++kk) g3acc += sA[kk * LDAW + t];
            }
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int slot = wave + 4 * i;              // (tile, k-part) slot of this wave
                if (slot < T * WK) {
                    const int tile = slot % T, kp = slot / T;
                    const int tm = tile % MT, tn = tile / MT;
                    constexpr int KPER = TKW / WK;
#pragma unroll 4
                    for (int kk = kp * KPER; kk < (kp + 1) * KPER; kk += 2) {
                        const float av = sA[(kk + (lane >> 5)) * LDAW + tm * 32 + (lane & 31)];
                        const int off = (kk + (lane >> 5)) * LDB + tn * 32 + (lane & 31);
                        acc1[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, sB[off], acc1[i], 0, 0, 0);
                        if (WANT_GX) accx[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, sX[off], accx[i], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
        if (it < nit) commit(r_begin + it * TKW);
        __syncthreads();
    }
    // ---- epilogue: this chunk's partial tiles / sums ----
    float* P1 = PP + (size_t)(blockIdx.x % nslots) * 2 * cin * cout;      // shared by nch/nslots chunks: zeroed by the launcher
    float* Px = P1 + (size_t)cin * cout;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int slot = wave + 4 * i;
        if (slot < T * WK) {
            const int tile = slot % T;
            const int tm = tile % MT, tn = tile / MT;
            const int col = n0 + tn * 32 + (lane & 31);
            if (col < cout) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + tm * 32 + c_row(r, lane);
                    if (m < cin) {
                        atomicAdd(P1 + (size_t)m * cout + col, acc1[i][r]);
                        if (WANT_GX) atomicAdd(Px + (size_t)m * cout + col, accx[i][r]);
                    }
                }
            }
        }
    }
    if (blockIdx.z == 0 && t < BM && m0 + t < cin) GP[(size_t)blockIdx.x * cin + m0 + t] = g3acc;
    __syncthreads();
    // column sums: threads with the same quad differ in (t / BQ); reduce through LDS (reuse sB)
    if (blockIdx.y == 0) {
        float* s0 = sB;                 // [BSTEP][BN] x2
        float* s1 = sB + BSTEP * BN;
        *reinterpret_cast<float4*>(s0 + b_kk0 * BN + b_nq) = s_r0;
        *reinterpret_cast<float4*>(s1 + b_kk0 * BN + b_nq) = s_r1;
    }
    __syncthreads();
    if (blockIdx.y == 0) {
        for (int j = t; j < BN; j += 256) {
            const int n = n0 + j;
            if (n < cout) {
                float v0 = 0.f, v1 = 0.f;
                for (int i = 0; i < BSTEP; ++i) { v0 += sB[i * BN + j]; v1 += sB[BSTEP * BN + i * BN + j]; }
                RP[(size_t)blockIdx.x * 2 * cout + n] = v0;
                RP[(size_t)blockIdx.x * 2 * cout + cout + n] = v1;
            }
        }
    }
}


// ---- pass A, streaming variant (the one used whenever rows are 16-byte aligned) -------------------------------------
// The staged operands go HBM -> LDS directly (global_load_lds_dwordx4: no VGPR round trip, no per-element VALU in the
// staging path) as RAW x / y / dz rows, double buffered: stage s+1 is in flight while stage s is consumed, one barrier per
// stage.  BN+ReLU of the input, the ReLU mask dyh = dz*[scale*y+shift > 0] and xhat = (y-mean)*rstd are applied when a wave
// reads its MFMA operands (~10 VALU per pair of 32x32x2 MFMAs, hidden under the 128 matrix-pipe cycles of the pair), where
// the column sums r0 = sum dyh, r1 = sum dyh*xhat (tiles with tm == 0) and g3 = sum A (tiles with tn == 0) are also taken.
// LDS image of one stage: [A: TKW x BM][Y: TKW x BN][dZ: TKW x BN (dense only)], linear (a wave-load writes 1 KiB contiguous).
// A max-pooled upstream gradient (POOLED) is never expanded: per stage a lane fetches the arg-max offset and the pooled
// gradient of its column for the <= NGMAX pool groups the stage covers, and dz = (arg == row % ns) ? dPool : 0.
// Three stage buffers (two stages in flight) instead of two: measured on MI355X at the bench shapes it does NOT pay -- the third buffer
// costs a workgroup per CU (72 KB of LDS) and pass A is bound by MFMA + VALU issue, not by bytes in flight (64->64 x 262144 rows:
// 72.7 us with two buffers, 76.4 us with three).  Kept as a build option.
#ifndef WGRAD_NBUF3
#define WGRAD_NBUF3 0
#endif
// Work split inside a workgroup: the T = MT*NTT tiles and WK k-parts are dealt to the 4 waves as a (GM x GN) grid of wave groups
// times WK k-parts (GM*GN*WK == 4).  Wave (gm, gn, kp) owns the AM x BNW block of tiles {tm = gm + GM*a} x {tn = gn + GN*b} over
// the rows of k-part kp: one k-loop with all its accumulators live, A operands shared along b and B operands along a.
template <int MT, int NTT> struct WgradSplit {
    static constexpr int T = MT * NTT;
    static constexpr int WK = (T % 4 == 0) ? 1 : ((T % 2 == 0) ? 2 : 4);
    static constexpr int G = 4 / WK;
    static constexpr int GM = G == 4 ? ((MT % 2 == 0 && NTT % 2 == 0) ? 2 : (MT % 4 == 0 ? 4 : 1)) : (G == 2 ? (NTT % 2 == 0 ? 1 : 2) : 1);
    static constexpr int GN = G / GM;
    static constexpr int AM = MT / GM, BNW = NTT / GN;
    static_assert(MT % GM == 0 && NTT % GN == 0, "wave groups tile the block");
};
// GATHER: the A operand's rows are virtual (GatherSrc, see mlp_fwd_stream_kernel); cin is the internal width 4*cq + 4 and the partial
// tiles / column sums come out in that internal row order (the dW reduction maps them back: DwJob::gq).
// KNOWN (early coefficients): the BN reductions r0, r1 of this layer were taken BEFORE this pass (by the epilogue of the next layer's
// pass B, or from the pool arg-max: gspn_mlp_bwd_coef), so a.cA/cB/cC are final and the B operand is dY = cA*dyh + cB*y + cC itself:
// ONE GEMM dW = A^T.dY instead of G1 and Gx, no x-hat, no column sums.  Implies !WANT_GX; PP holds dW's partial tiles directly.
template <int MT, int NTT, int TKW, bool WANT_GX, bool POOLED, bool GATHER = false, bool KNOWN = false>
__global__ __launch_bounds__(256, ((WgradSplit<MT, NTT>::AM * WgradSplit<MT, NTT>::BNW * (WANT_GX ? 2 : 1) >= 4 || (POOLED && TKW >= 32)) ? 2 : 3)) void wgrad_stream_kernel(int rows, int cin, int cout, gspn_dy_args a, const float* __restrict__ X, int ldx,
                                                           const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                           const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                           float* __restrict__ RP, float* __restrict__ GP, float* __restrict__ PP,
                                                           int rows_per_chunk, int nslots, int shared, int nch, int nrow, int ncol, GatherSrc gsrc) {
    static_assert(!(GATHER && POOLED), "the gathered layer is the first of a stack: its upstream gradient is dense");
    static_assert(!(KNOWN && WANT_GX), "known coefficients need no second product");
    using SP = WgradSplit<MT, NTT>;
    constexpr int BM = 32 * MT, BN = 32 * NTT;
    constexpr int WK = SP::WK, GM = SP::GM, GN = SP::GN, AM = SP::AM, BNW = SP::BNW;
    constexpr int KPER = TKW / WK;
    static_assert(KPER % 2 == 0 && KPER >= 2, "k-part must hold whole MFMA k-pairs");
    constexpr int AQ = BM / 4, BQ = BN / 4;
    constexpr int PA = TKW * AQ / 64, PB = TKW * BQ / 64;             // 1-KiB pieces per array
    static_assert((TKW * AQ) % 64 == 0 && (TKW * BQ) % 64 == 0, "whole pieces");
    constexpr int SF = TKW * (BM + BN + (POOLED ? 0 : BN));           // floats per stage
    constexpr int NGMAX = POOLED ? (TKW >= 16 ? TKW / 16 : 1) : 1;    // pool groups per stage (ns >= 16)
    constexpr int JA = (PA + 3) / 4, JB = (PB + 3) / 4;
    // Stage buffers.  With three, TWO stages are in flight while one is consumed (more bytes in flight per CU: Little's law at 8 TB/s);
    // the wait for "stage s has landed" is then a COUNTED vmcnt that leaves the newer stage's pieces outstanding.  That is only sound
    // when the loop's only vector-memory operations are the in-order LDS-DMA loads and every wave issues the same number per stage:
    // dense upstream gradient (the pooled variant interleaves ordinary loads) and whole multiples of 4 pieces per array.
    constexpr int PPW = JA + (POOLED ? JB : 2 * JB);                  // pieces per wave per stage
    constexpr int NBUF = (WGRAD_NBUF3 && !POOLED && PA % 4 == 0 && PB % 4 == 0 && 3 * SF * 4 <= 80 * 1024 && PPW <= 15) ? 3 : 2;
    __shared__ __attribute__((aligned(16))) float sbuf[NBUF * SF];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int kp = wave % WK, grp = wave / WK;
    const int gm = grp % GM, gn = grp / GM;

    // block -> (chunk, tile): the tiles of one chunk are adjacent in launch order AND on the same XCD (id % 8), so the rows they
    // share are fetched from HBM once per XCD L2
    const int ntile = nrow * ncol;
    const int bid = blockIdx.x;
    const int rest = bid >> 3;
    const int tile_id = rest % ntile;
    const int chunk = (rest / ntile) * 8 + (bid & 7);
    if (chunk >= nch) return;
    const int ty = tile_id % nrow, tz = tile_id / nrow;
    const int m0 = ty * BM, n0 = tz * BN;
    const int r_begin = chunk * rows_per_chunk;
    const int r_end = min(r_begin + rows_per_chunk, rows);
    const int nloc = r_end - r_begin;                                  // rows of this chunk (> 0)
    const int nit = (nloc + TKW - 1) / TKW;
    const int lrmax = nloc - 1;

    // ---- per-lane piece descriptors (loop invariant): row-in-stage and clamped column of the lane's quad ----
    int a_kk[JA], a_col[JA], b_kk[JB], b_col[JB];
#pragma unroll
    for (int j = 0; j < JA; ++j) {
        const int f = (wave + 4 * j) * 64 + lane;
        a_kk[j] = f / AQ;
        a_col[j] = min(m0 + (f - a_kk[j] * AQ) * 4, (GATHER ? cin : ldx) - 4);
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const int f = (wave + 4 * j) * 64 + lane;
        b_kk[j] = f / BQ;
        b_col[j] = min(n0 + (f - b_kk[j] * BQ) * 4, cout - 4);
    }
    const float* Xc = GATHER ? nullptr : X + (size_t)r_begin * ldx;
    const float* Yc = a.Y + (size_t)r_begin * a.ldy;
    const float* Zc = POOLED ? nullptr : a.dZ + (size_t)r_begin * a.ldz;
    // GATHER: source rows of the lane's A pieces for the NEXT stage to issue (fetched a stage ahead, consumed after the opening vmcnt(0))
    int ga_nxt[GATHER ? JA : 1];
    auto gfetch = [&](int s) {
        if constexpr (GATHER) {
            const int k0 = s * TKW;
#pragma unroll
            for (int j = 0; j < JA; ++j) ga_nxt[j] = gsrc.gidx[r_begin + min(k0 + a_kk[j], lrmax)];
        }
    };
    auto issue = [&](int s) {
        float* dst = sbuf + (s % NBUF) * SF;
        const int k0 = s * TKW;
#pragma unroll
        for (int j = 0; j < JA; ++j) {
            if (PA % 4 == 0 || wave + 4 * j < PA) {
                if constexpr (GATHER) {
                    const float* src = a_col[j] < 4 * gsrc.cq ? gsrc.feat + (size_t)ga_nxt[j] * (4 * gsrc.cq) + a_col[j]
                                                            : gsrc.rel + (size_t)(r_begin + min(k0 + a_kk[j], lrmax)) * 4;
                    glds16(src, dst + (wave + 4 * j) * 256);
                } else {
                    glds16(Xc + (size_t)(min(k0 + a_kk[j], lrmax) * ldx + a_col[j]), dst + (wave + 4 * j) * 256);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            if (PB % 4 == 0 || wave + 4 * j < PB) {
                const int lr = min(k0 + b_kk[j], lrmax);
                glds16(Yc + (size_t)(lr * a.ldy + b_col[j]), dst + TKW * BM + (wave + 4 * j) * 256);
                if constexpr (!POOLED) glds16(Zc + (size_t)(lr * a.ldz + b_col[j]), dst + TKW * (BM + BN) + (wave + 4 * j) * 256);
            }
        }
    };

    // ---- per-operand constants: AM input-channel columns, BNW output-channel columns per lane ----
    float isc[AM], ish[AM], bsc[BNW], bsh[BNW], bmr[BNW], brs[BNW];
    int ao[AM], bo[BNW];
    const float lo = in_scale ? 0.f : -__builtin_inff();               // relu floor (none for a raw first-layer input)
#pragma unroll
    for (int x = 0; x < AM; ++x) {
        ao[x] = (gm + GM * x) * 32 + l31;
        const int m = min(m0 + ao[x], cin - 1);
        isc[x] = in_scale ? in_scale[m] : 1.f;
        ish[x] = in_scale ? in_shift[m] : 0.f;
    }
#pragma unroll
    for (int y = 0; y < BNW; ++y) {
        bo[y] = (gn + GN * y) * 32 + l31;
        const int n = min(n0 + bo[y], cout - 1);
        bsc[y] = a.scale[n];
        bsh[y] = a.shift[n];
        brs[y] = var ? (float)(1.0 / sqrt((double)var[n] + (double)eps)) : 1.f;
        bmr[y] = -(mean ? mean[n] : 0.f) * brs[y];
        if constexpr (KNOWN) { brs[y] = a.cB[n]; bmr[y] = a.cC[n]; }      // reused: dY = cA*dyh + (cB*y + cC)
    }
    float kca[KNOWN ? BNW : 1];
    if constexpr (KNOWN) {
#pragma unroll
        for (int y = 0; y < BNW; ++y) kca[y] = a.cA[min(n0 + bo[y], cout - 1)];
    }
    // ---- pooled gradient: per (column tile, group-in-stage) arg-max offset and pooled gradient of the lane's column ----
    const int ns = POOLED ? a.ns : TKW;
    const int gs = min(ns, TKW);                      // rows of one pool group inside a stage
    const int ng = TKW / gs;                          // groups per stage (<= NGMAX)
    const int ngroups = POOLED ? rows / ns : 0;
    int g_next = POOLED ? r_begin / ns : 0;           // first group / row offset within it of the NEXT stage to prefetch
    int off_next = POOLED ? r_begin - g_next * ns : 0;
    int p_arg[BNW][NGMAX], n_arg[BNW][NGMAX];
    float p_dp[BNW][NGMAX], n_dp[BNW][NGMAX];
    bool n_ok[BNW][NGMAX];
    int off_cur = 0, off_nxt = 0;
    auto pool_fetch = [&]() {                          // loads for the stage (g_next, off_next) into n_*; advances the cursor
        if constexpr (POOLED) {
            off_nxt = off_next;
#pragma unroll
            for (int y = 0; y < BNW; ++y) {
                const int n = min(n0 + bo[y], cout - 1);
#pragma unroll
                for (int gl = 0; gl < NGMAX; ++gl) {
                    const int g = g_next + gl;
                    const size_t at = (size_t)min(g, ngroups - 1) * cout + n;
                    n_arg[y][gl] = a.pool_arg[at];
                    n_dp[y][gl] = a.dPool[at];
                    n_ok[y][gl] = g < ngroups && gl < ng;                   // rows past the end / unused slots never match
                    // (the select is applied when the values are consumed, one stage later: selecting here would make hipcc wait
                    //  for these loads -- and with them for the stage's LDS-DMA pieces -- right after issuing them)
                }
            }
            off_next += TKW;
            if (off_next >= ns) { off_next = 0; g_next += ng; }
        }
    };

    f32x16 acc1[AM][BNW], accx[WANT_GX ? AM : 1][WANT_GX ? BNW : 1];
    float r0a[BNW], r1a[BNW], g3a[AM];
#pragma unroll
    for (int x = 0; x < AM; ++x) {
        g3a[x] = 0.f;
#pragma unroll
        for (int y = 0; y < BNW; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc1[x][y][r] = 0.f; if (WANT_GX) accx[x][y][r] = 0.f; }
    }
#pragma unroll
    for (int y = 0; y < BNW; ++y) r0a[y] = r1a[y] = 0.f;

    static_assert(!(GATHER && NBUF == 3), "the gather look-ahead is written for two stage buffers");
    pool_fetch();
    gfetch(0);
    issue(0);
    if (nit > 1) gfetch(1);
    if (NBUF == 3 && nit > 1) issue(1);
    for (int s = 0; s < nit; ++s) {
        // this wave's pieces of stage s have landed (with three buffers the PPW pieces of stage s+1 may still be in flight) ...
        if (NBUF == 3 && s + 1 < nit) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                        // ... and so have everyone else's; stage s-1 is fully consumed
        const float* buf = sbuf + (s % NBUF) * SF;
        if constexpr (POOLED) {
#pragma unroll
            for (int y = 0; y < BNW; ++y)
#pragma unroll
                for (int gl = 0; gl < NGMAX; ++gl) { p_arg[y][gl] = n_ok[y][gl] ? n_arg[y][gl] : -1; p_dp[y][gl] = n_dp[y][gl]; }
            off_cur = off_nxt;
        }
        if (NBUF == 3) { if (s + 2 < nit) issue(s + 2); }
        else if (s + 1 < nit) { pool_fetch(); issue(s + 1); if (s + 2 < nit) gfetch(s + 2); }
        const float* sA = buf;
        const float* sY = buf + TKW * BM;
        const float* sZ = buf + TKW * (BM + BN);
        const int live = nloc - s * TKW;                        // < TKW only in a ragged last stage (clamped duplicate rows follow)
        auto compute = [&](auto tail_c) {
            constexpr bool TAIL = decltype(tail_c)::value;
            // one MFMA k-pair: this lane's row r of the stage; gl = pool group of that row (compile-time when it matters)
            // VALU instructions are not hidden under the MFMAs on this chip (~3.4 SIMD cycles each, tools/mfma_probe.hip), so the operand
            // arithmetic is spelled with single fused operations (the mask / x-hat differ from the forward's two-rounding forms by one
            // ulp at most: irrelevant for a gradient) and the column sums are taken only by the wave groups that report them.
            auto step = [&](int r, int gl_rt) {
                float av[AM], dyh[BNW], xh[BNW];
#pragma unroll
                for (int x = 0; x < AM; ++x) {
                    // (no inline asm here: hipcc's hazard recogniser does not see an asm VALU write feeding an MFMA operand)
                    av[x] = __builtin_fmaxf(__builtin_fmaf(sA[r * BM + ao[x]], isc[x], ish[x]), lo);
                    if (TAIL) av[x] = r < live ? av[x] : 0.f;
                }
#pragma unroll
                for (int y = 0; y < BNW; ++y) {
                    const float yv = sY[r * BN + bo[y]];
                    float dz;
                    if constexpr (!POOLED) dz = sZ[r * BN + bo[y]];
                    else if constexpr (NGMAX == 1) dz = p_arg[y][0] == off_cur + r ? p_dp[y][0] : 0.f;
                    else {
                        dz = 0.f;
#pragma unroll
                        for (int gl = 0; gl < NGMAX; ++gl)
                            if (gl == gl_rt) dz = p_arg[y][gl] == off_cur + r - gl * gs ? p_dp[y][gl] : 0.f;
                    }
                    if (TAIL) dz = r < live ? dz : 0.f;
                    dyh[y] = relu_open(yv, bsc[y], bsh[y]) ? dz : 0.f;
                    xh[y] = __builtin_fmaf(yv, brs[y], bmr[y]);              // (y - mean) * rstd   [KNOWN: cB*y + cC]
                    if constexpr (KNOWN) {
                        dyh[y] = __builtin_fmaf(kca[y], dyh[y], xh[y]);      // dY (rows past the end are killed through av = 0)
                    }
                }
                if (!KNOWN && gm == 0) {
#pragma unroll
                    for (int y = 0; y < BNW; ++y) {
                        r0a[y] += dyh[y];
                        r1a[y] = __builtin_fmaf(dyh[y], xh[y], r1a[y]);
                    }
                }
                if (!KNOWN && gn == 0) {
#pragma unroll
                    for (int x = 0; x < AM; ++x) g3a[x] += av[x];
                }
#pragma unroll
                for (int x = 0; x < AM; ++x) {
#pragma unroll
                    for (int y = 0; y < BNW; ++y) {
                        acc1[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x], dyh[y], acc1[x][y], 0, 0, 0);
                        if (WANT_GX) accx[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x], xh[y], accx[x][y], 0, 0, 0);
                    }
                }
            };
            const int k_lo = kp * KPER;
            if constexpr ((!POOLED || NGMAX == 1) && AM * BNW * (WANT_GX ? 2 : 1) >= 4) {
                // 64+ accumulator registers: no unrolling (the MFMAs of one step cover the LDS latency of the next wave's step)
#pragma unroll 1
                for (int kk = 0; kk < KPER; kk += 2) step(k_lo + kk + half, 0);
            } else if constexpr (!POOLED || NGMAX == 1) {
#pragma unroll 2
                for (int kk = 0; kk < KPER; kk += 2) step(k_lo + kk + half, 0);
            } else {
                // rows of pool group gl inside the stage: [gl*gs, (gl+1)*gs); ns is even, so a k-pair never straddles two groups
                for (int kk = 0; kk < KPER; kk += 2) {
                    const int r = k_lo + kk + half;
                    step(r, (k_lo + kk) / gs);
                }
            }
        };
        if (live >= TKW) compute(std::false_type{});
        else compute(std::true_type{});
    }
    // ---- epilogue: this chunk's partial tiles / sums ----
    // k-parts > 0 hand their accumulators to part 0 through LDS (fixed order), so ONE partial tile per chunk leaves the CU.
    if constexpr (WK > 1) {
        constexpr int TILEF = 16 * 64;                                    // floats of one 32x32 accumulator tile, [reg][lane]
        static_assert((WK - 1) * SP::G * BNW * TILEF <= NBUF * SF, "k-part exchange fits the stage buffers");
#pragma unroll
        for (int which = 0; which < (WANT_GX ? 2 : 1); ++which) {
#pragma unroll
            for (int x = 0; x < AM; ++x) {                                // one row of tiles per round (bounds the scratch)
                __syncthreads();
                if (kp > 0) {
#pragma unroll
                    for (int y = 0; y < BNW; ++y) {
                        float* dst = sbuf + (size_t)(((kp - 1) * SP::G + grp) * BNW + y) * TILEF;
#pragma unroll
                        for (int r = 0; r < 16; ++r) dst[r * 64 + lane] = which ? accx[WANT_GX ? x : 0][WANT_GX ? y : 0][r] : acc1[x][y][r];
                    }
                }
                __syncthreads();
                if (kp == 0) {
#pragma unroll
                    for (int q = 1; q < WK; ++q)
#pragma unroll
                        for (int y = 0; y < BNW; ++y) {
                            const float* src = sbuf + (size_t)(((q - 1) * SP::G + grp) * BNW + y) * TILEF;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                if (which) accx[WANT_GX ? x : 0][WANT_GX ? y : 0][r] += src[r * 64 + lane];
                                else acc1[x][y][r] += src[r * 64 + lane];
                            }
                        }
                }
            }
        }
    }
    // shared == 0: every chunk owns a partial-tile slot -> plain stores; else chunks share zero-filled slots (fp32 atomics)
    if (kp == 0) {
        const int slot = shared ? chunk % nslots : chunk;
        float* P1 = PP + (size_t)slot * 2 * cin * cout;
        float* Px = P1 + (size_t)cin * cout;
#pragma unroll
        for (int x = 0; x < AM; ++x)
#pragma unroll
            for (int y = 0; y < BNW; ++y) {
                const int col = n0 + bo[y];
                if (col < cout) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + (gm + GM * x) * 32 + c_row(r, lane);
                        if (m < cin) {
                            if (shared) {
                                atomicAdd(P1 + (size_t)m * cout + col, acc1[x][y][r]);
                                if (WANT_GX) atomicAdd(Px + (size_t)m * cout + col, accx[x][y][r]);
                            } else {
                                P1[(size_t)m * cout + col] = acc1[x][y][r];
                                if (WANT_GX) Px[(size_t)m * cout + col] = accx[x][y][r];
                            }
                        }
                    }
                }
            }
    }
    // column sums: one LDS cell per (k-part, lane half, column), summed in a fixed order.  Every column tile is seen by GM wave
    // groups (only gm == 0 reports it), every row tile by GN (only gn == 0 reports it).
    if constexpr (KNOWN) return;
    __syncthreads();
    float* sR = sbuf;                                   // [WK*2][2][BN]
    float* sG = sbuf + WK * 2 * 2 * BN;                 // [WK*2][BM]
    static_assert(WK * 2 * (2 * BN + BM) <= NBUF * SF, "reduction scratch fits the stage buffers");
    {
        const int part = kp * 2 + half;
        if (gm == 0) {
#pragma unroll
            for (int y = 0; y < BNW; ++y) {
                sR[(part * 2 + 0) * BN + bo[y]] = r0a[y];
                sR[(part * 2 + 1) * BN + bo[y]] = r1a[y];
            }
        }
        if (gn == 0) {
#pragma unroll
            for (int x = 0; x < AM; ++x) sG[part * BM + ao[x]] = g3a[x];
        }
    }
    __syncthreads();
    if (ty == 0) {
        for (int j = t; j < BN; j += 256) {
            const int n = n0 + j;
            if (n < cout) {
                float v0 = 0.f, v1 = 0.f;
#pragma unroll
                for (int p = 0; p < WK * 2; ++p) { v0 += sR[(p * 2 + 0) * BN + j]; v1 += sR[(p * 2 + 1) * BN + j]; }
                RP[(size_t)chunk * 2 * cout + n] = v0;
                RP[(size_t)chunk * 2 * cout + cout + n] = v1;
            }
        }
    }
    if (tz == 0) {
        for (int j = t; j < BM; j += 256) {
            const int m = m0 + j;
            if (m < cin) {
                float v = 0.f;
#pragma unroll
                for (int p = 0; p < WK * 2; ++p) v += sG[p * BM + j];
                GP[(size_t)chunk * cin + m] = v;
            }
        }
    }
}

// ---- second stage: sum the per-chunk partials (double), then coefficients / parameter gradients / dW ----
// Plan: tile shape per workgroup, rows per chunk, number of partial-tile slots.
//   * chunks: ~3 workgroups per CU when the layer is long enough, but a chunk always reads at least ~2x the bytes of the partial
//     tile it writes (rpc_min), as long as that still leaves one workgroup per CU;
//   * slots: when nch partial tiles fit in 64 MB every chunk owns a slot and the main kernel uses plain stores
//     (shared == 0: no zero fill, no atomics); otherwise chunks share `nslots` zero-filled slots through fp32 atomics.
struct WgradPlan { int MTs, NTs, nrow, ncol, TKW, WK, shared; long rpc, nch, nslots; };
static WgradPlan wgrad_plan(long rows, int cin, int cout, bool generic = false, bool pooled = false) {
    WgradPlan p;
    const int mt = (cin + 31) / 32, nt = (cout + 31) / 32;
    if (generic) {
        p.nrow = (mt + 3) / 4; p.ncol = (nt + 3) / 4;
        p.MTs = (mt + p.nrow - 1) / p.nrow;
        p.NTs = (nt + p.ncol - 1) / p.ncol;
        if (p.NTs == 3) p.NTs = 4;
    } else {
        // Small tiles: one 32-row strip of dW per workgroup, 1-4 column tiles.  More tiles per layer means fewer row chunks for the same
        // number of workgroups, i.e. fewer partial tiles to write and sum (that traffic rivals the inputs of the small-row layers), and
        // a re-read input tile is an L2 hit.  Swept on MI355X over all 16 layers of the benchmark stack (tools/wgrad_sweep.py): MT = 1
        // wins everywhere (by 10-25 % below 131072 rows); NT = 4 for the pooled kernel, 2 for the long layers, 1 for the short ones.
        p.MTs = 1; p.nrow = mt;
        int want = pooled ? 4 : (rows >= 131072 ? 2 : 1);
        while (want > 1 && want / 2 >= nt) want /= 2;
        p.NTs = want; p.ncol = (nt + want - 1) / want;
    }
    // rows staged per iteration (streaming kernel): one stage is 14-24 KB of LDS, two stages per workgroup, 3 workgroups per CU
    static const int tkw[4][3] = {{64, 32, 16}, {32, 32, 16}, {32, 16, 16}, {32, 16, 16}};      // [MT-1][NT: 1,2,4]
    p.TKW = tkw[p.MTs - 1][p.NTs == 1 ? 0 : (p.NTs == 2 ? 1 : 2)];
    // tuning hook (tools/wgrad_sweep.py): GSPN_WGRAD_FORCE="MTs,NTs,chunks" overrides the tile shape / the number of row chunks (0 = keep)
    long forced_chunks = 0;
    if (!generic) {
        const char* e = getenv("GSPN_WGRAD_FORCE");
        int fm = 0, fn = 0;
        long fc = 0;
        if (e && sscanf(e, "%d,%d,%ld", &fm, &fn, &fc) >= 2) {
            if (fm >= 1 && fm <= 4) { p.MTs = fm; p.nrow = (mt + fm - 1) / fm; }
            if (fn == 1 || fn == 2 || fn == 4) { p.NTs = fn; p.ncol = (nt + fn - 1) / fn; }
            p.TKW = tkw[p.MTs - 1][p.NTs == 1 ? 0 : (p.NTs == 2 ? 1 : 2)];
            forced_chunks = fc;
        }
    }
    if (generic) { p.nrow = (cin + 127) / 128; p.ncol = (cout + 127) / 128; p.MTs = 4; p.NTs = 4; p.TKW = 32; }
    const int T = p.MTs * p.NTs;
    p.WK = generic ? 1 : ((T % 4 == 0) ? 1 : ((T % 2 == 0) ? 2 : 4));
    const long ntile = (long)p.ncol * p.nrow;
    const long BM = 32L * p.MTs, BN = 32L * p.NTs;
    long rpc_min = 4 * BM * BN / (BM + 2 * BN);            // input bytes of a chunk >= 2 x its partial-tile bytes
    if (rpc_min < 4L * p.TKW) rpc_min = 4L * p.TKW;
    // workgroups per CU: 3 with two stage buffers, 2 when the kernel takes three (same condition as NBUF in wgrad_stream_kernel)
    // (the configurations with >= 64 accumulator registers per wave, or 3 row tiles, are compiled for 2 waves per SIMD = 2 workgroups
    //  per CU -- see the launch bounds of wgrad_stream_kernel; a grid sized for 3 would run a partial second wave)
    int bpc = 3;
    if (!generic) {
        const int T = p.MTs * p.NTs;
        const int WKq = (T % 4 == 0) ? 1 : ((T % 2 == 0) ? 2 : 4);
        const int tiles_per_wave = T * WKq / 4;
        if (tiles_per_wave >= 2 || (pooled && p.TKW >= 32)) bpc = 2;
    }
    if (!generic && !pooled && WGRAD_NBUF3) {
        const long pa = p.TKW * BM / 4 / 64, pb = p.TKW * BN / 4 / 64, sf = (long)p.TKW * (BM + 2 * BN);
        if (pa % 4 == 0 && pb % 4 == 0 && 3 * sf * 4 <= 80 * 1024 && pa / 4 + 2 * (pb / 4) <= 15) bpc = 2;
    }
    long chunks = ((long)GSPN_PLAN_CUS * bpc) / ntile;        // upper target: every workgroup resident at once
    if (chunks < 1) chunks = 1;
    const long by_size = rows / rpc_min;                    // chunks allowed by the size rule
    long floor_ch = GSPN_PLAN_CUS / ntile;                            // but never fewer than one workgroup per CU (if the layer has the rows)
    if (floor_ch < 1) floor_ch = 1;
    if (chunks > by_size) chunks = by_size > floor_ch ? by_size : floor_ch;
    if (forced_chunks > 0) chunks = forced_chunks;
    long rpc = (rows + chunks - 1) / chunks;
    if (rpc < 4L * p.TKW) rpc = 4L * p.TKW;
    const long rq = p.TKW > 32 ? p.TKW : 32;              // (r03) whole stages of the lean kernel (32 rows) as well as of the streaming one (TKW)
    p.rpc = (rpc + rq - 1) / rq * rq;
    p.nch = (rows + p.rpc - 1) / p.rpc;
    if (p.nch < 1) p.nch = 1;
    const long tile_bytes = 8L * cin * cout;
    if (!generic && p.nch * tile_bytes <= (64L << 20)) { p.shared = 0; p.nslots = p.nch; }
    else {
        long cap = (12L << 20) / tile_bytes;
        if (cap < 8) cap = 8;
        p.shared = 1;
        p.nslots = p.nch < cap ? p.nch : cap;
    }
    return p;
}
// workspace: [red: 2*cout doubles][g3: cin floats, padded to 4][RP: nch*2*cout][GP: nch*cin][PP: nslots*2*cin*cout]
static size_t ws_off_g3(int cout) { return sizeof(double) * 2 * (size_t)cout; }
static size_t ws_off_rp(int cin, int cout) { return ws_off_g3(cout) + sizeof(float) * (size_t)((cin + 3) / 4 * 4); }
static size_t ws_off_gp(long nch, int cin, int cout) { return ws_off_rp(cin, cout) + sizeof(float) * (size_t)nch * 2 * cout; }
static size_t ws_off_pp(long nch, int cin, int cout) { return (ws_off_gp(nch, cin, cout) + sizeof(float) * (size_t)nch * cin + 15) / 16 * 16; }
static size_t ws_total(long nch, long nslots, int cin, int cout) { return ws_off_pp(nch, cin, cout) + sizeof(float) * (size_t)nslots * 2 * cin * cout; }
extern "C" long gspn_mlp_bwd_fused_work_bytes(long rows, int cin, int cout);
extern "C" long gspn_mlp_bwd_work_bytes(long rows, int cin, int cout) {
    if (rows <= 0 || cin <= 0 || cout <= 0) return GSPN_ERR_ARG;
    const WgradPlan p = wgrad_plan(rows, cin, cout), q = wgrad_plan(rows, cin, cout, true), r = wgrad_plan(rows, cin, cout, false, true);
    size_t a = ws_total(p.nch, p.nslots, cin, cout);
    const size_t b = ws_total(q.nch, q.nslots, cin, cout), c = ws_total(r.nch, r.nslots, cin, cout);
    if (b > a) a = b;
    if (c > a) a = c;
    const size_t f = (size_t)gspn_mlp_bwd_fused_work_bytes(rows, cin, cout);      // (one partial tile set per workgroup of the fused kernel)
    if (f > a) a = f;
    return (long)a;
}

// one WORKGROUP per channel index n in [0, max(cin,cout)): r0, r1 (and g3[n]) summed over chunks in double -> coefficients etc.
__global__ __launch_bounds__(256) void wgrad_small_reduce_kernel(long rows, int cin, int cout, int nch, const float* __restrict__ RP, const float* __restrict__ GP,
                                                                 double* __restrict__ red, float* __restrict__ g3, const float* __restrict__ mean,
                                                                 const float* __restrict__ var, const float* __restrict__ gamma, float eps, int use_bn, int is_training,
                                                                 float* __restrict__ cA, float* __restrict__ cB, float* __restrict__ cC,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias) {
    __shared__ double sh[3][4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = blockIdx.x;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int p = t; p < nch; p += 256) {
        if (n < cout) { a0 += (double)RP[(size_t)p * 2 * cout + n]; a1 += (double)RP[(size_t)p * 2 * cout + cout + n]; }
        if (n < cin) a2 += (double)GP[(size_t)p * cin + n];
    }
    a0 = wave_sum_f64(a0); a1 = wave_sum_f64(a1); a2 = wave_sum_f64(a2);
    if (lane == 0) { sh[0][wave] = a0; sh[1][wave] = a1; sh[2][wave] = a2; }
    __syncthreads();
    if (t != 0) return;
    const double r0 = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
    const double r1 = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    const double gg = (sh[2][0] + sh[2][1]) + (sh[2][2] + sh[2][3]);
    if (n < cin) g3[n] = (float)gg;
    if (n >= cout) return;
    red[n] = r0;
    red[cout + n] = r1;
    const double R = (double)rows;
    double A = 1.0, B = 0.0, C = 0.0;
    const bool tr = use_bn && is_training;
    if (use_bn) {
        const double g = gamma ? (double)gamma[n] : 1.0;
        const double rstd = 1.0 / sqrt((double)var[n] + (double)eps);
        const double mu = (double)mean[n];
        A = g * rstd;
        if (tr) {
            B = -g * rstd * rstd * (r1 / R);
            C = -g * rstd * (r0 / R - mu * rstd * (r1 / R));
        }
        if (dgamma) dgamma[n] = (float)r1;
        if (dbeta) dbeta[n] = (float)r0;
    }
    if (cA) cA[n] = (float)A;
    if (cB) cB[n] = (float)B;
    if (cC) cC[n] = (float)C;
    if (dbias) dbias[n] = (float)(tr ? 0.0 : A * r0);      // sum(dY): exactly 0 under batch statistics
}
// dW[m][n] = cA[n] * (sum_slots G1 - r0/R * g3[m] - r1/R * sum_slots Gx)
// block 1024 = 16 consecutive outputs x 64 interleaved slot slices (a wave = 16 outputs x 4 slices: 64-byte segments): cin*cout/16
// workgroups of 16 waves keep the whole chip streaming the partial tiles, up to 16 (+16) loads in flight per thread; the 64 slices are
// summed in double, in a fixed order: 4 by lane shuffles, 16 through one LDS hop.
#define DW_OX 16
#define DW_SL 64
#define DW_UN 8
// everything the dW reduction needs besides its block index (also carried by mlp_bwd_data_kernel, which can run it in spare workgroups)
struct DwJob {
    long rows;
    int cin, cout, nslots, nblk;          // nblk = dw_blocks(cin*cout, nslots, 256): workgroups of 256 threads the reduction takes
    const float* PP;
    const double* red;
    const float* g3;
    const float* var;
    const float* gamma;
    float eps;
    int use_bn, is_training;
    float* dW;
    int plain;                            // the partial tiles ARE dW's (pass A ran with known coefficients): dW = their sum
    int gq, gc_real, gxyz_first;          // gq > 0: rows of the partial tiles are in the gathered layer's internal order (GatherSrc)
    int rowgrid;                          // (set by the pass-B launcher that carries the job: its GEMM workgroups per column block)
    // an optional second, plain job riding along (the side-column dW of a pre-aggregated layer): dW2 (cin2, cout) = sum of nslots2 tiles
    const float* PP2;
    float* dW2;
    int cin2, nslots2, nblk2;
};
// One workgroup of NTH threads = DW_OX consecutive outputs x NTH/DW_OX interleaved slot slices; sh = 2 * (NTH/64) * DW_OX doubles.
// the last step of the reduction, shared by both thread mappings: BN correction, scale, (gathered layers) row mapping, store
__device__ __forceinline__ void wgrad_dw_store(const DwJob& j, long i, double w1, double wx, bool tr) {
    const int n = (int)(i % j.cout), m = (int)(i / j.cout);
    double A = 1.0;
    if (j.use_bn && !j.plain) A = (j.gamma ? (double)j.gamma[n] : 1.0) / sqrt((double)j.var[n] + (double)j.eps);
    if (tr) w1 -= (j.red[n] / (double)j.rows) * (double)j.g3[m] + (j.red[j.cout + n] / (double)j.rows) * wx;
    if (j.gq > 0) {                                               // internal row m -> row of the caller's dW (padding rows have none)
        GatherSrc g{nullptr, nullptr, nullptr, j.gq, j.gc_real, j.gxyz_first};
        const int mo = gather_w_row(g, m);
        if (mo >= 0) j.dW[(size_t)mo * j.cout + n] = (float)(A * w1);
        return;
    }
    j.dW[i] = (float)(A * w1);
}
// Few slots (the short layers: 4096-32768 rows give 4-16 row chunks, but up to 384 x 256 outputs): one thread per output, the slots summed
// in order -- NTH outputs per workgroup instead of DW_OX (the sliced mapping below would spend 6144 workgroups, each with fifteen of
// its sixteen slot slices idle, on the 384 -> 256 layer of the FP stack).
#define DW_FLAT_MAX 16
template <int NTH>
__device__ __forceinline__ void wgrad_dw_block_flat(const DwJob& j, unsigned blk) {
    const long total = (long)j.cin * j.cout;
    const long i = blk * (long)NTH + threadIdx.x;
    if (i >= total) return;
    const bool tr = j.use_bn && j.is_training && !j.plain;
    const float* p1 = j.PP + i;
    const size_t st = 2 * (size_t)total;
    double w1 = 0.0, wx = 0.0;
    int p = 0;
    for (; p + 3 < j.nslots; p += 4) {
        const float v0 = p1[(size_t)p * st], v1 = p1[(size_t)(p + 1) * st], v2 = p1[(size_t)(p + 2) * st], v3 = p1[(size_t)(p + 3) * st];
        float u0 = 0.f, u1 = 0.f, u2 = 0.f, u3 = 0.f;
        if (tr) { u0 = p1[(size_t)p * st + total]; u1 = p1[(size_t)(p + 1) * st + total]; u2 = p1[(size_t)(p + 2) * st + total]; u3 = p1[(size_t)(p + 3) * st + total]; }
        w1 += (double)v0; w1 += (double)v1; w1 += (double)v2; w1 += (double)v3;
        wx += (double)u0; wx += (double)u1; wx += (double)u2; wx += (double)u3;
    }
    for (; p < j.nslots; ++p) {
        w1 += (double)p1[(size_t)p * st];
        if (tr) wx += (double)p1[(size_t)p * st + total];
    }
    wgrad_dw_store(j, i, w1, wx, tr);
}
template <int NTH>
__device__ __forceinline__ void wgrad_dw_block(const DwJob& j, unsigned blk, double* sh) {
    if (j.nslots <= DW_FLAT_MAX) { wgrad_dw_block_flat<NTH>(j, blk); return; }
    constexpr int SL = NTH / DW_OX, NW = NTH / 64;
    double* s1 = sh;
    double* sx = sh + NW * DW_OX;
    const long total = (long)j.cin * j.cout;
    const bool tr = j.use_bn && j.is_training && !j.plain;
    const int ox = threadIdx.x % DW_OX, sl = threadIdx.x / DW_OX, wave = threadIdx.x >> 6;
    const long i = blk * (long)DW_OX + ox;
    const long ic = i < total ? i : total - 1;                   // clamped: every lane takes part in the shuffles
    double w1 = 0.0, wx = 0.0;
    {
        const float* p1 = j.PP + ic;
        const size_t st = 2 * (size_t)total;
        const int last = j.nslots - 1;
        for (int p = sl; p < j.nslots; p += DW_UN * SL) {
            float v[DW_UN], u[DW_UN];
#pragma unroll
            for (int q = 0; q < DW_UN; ++q) {                    // clamped slot index: unconditional loads, masked below
                const int pq = min(p + q * SL, last);
                v[q] = p1[(size_t)pq * st];
                u[q] = tr ? p1[(size_t)pq * st + total] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < DW_UN; ++q) {
                const bool ok = p + q * SL < j.nslots;
                w1 += ok ? (double)v[q] : 0.0;
                wx += ok ? (double)u[q] : 0.0;
            }
        }
    }
    // lanes l, l^16, l^32, l^48 hold the 4 slices of one output
    w1 += __shfl_xor(w1, 16, 64); wx += __shfl_xor(wx, 16, 64);
    w1 += __shfl_xor(w1, 32, 64); wx += __shfl_xor(wx, 32, 64);
    if ((threadIdx.x & 63) < DW_OX) { s1[wave * DW_OX + ox] = w1; sx[wave * DW_OX + ox] = wx; }
    __syncthreads();
    if (threadIdx.x >= DW_OX || i >= total) return;
    w1 = 0.0; wx = 0.0;
#pragma unroll
    for (int q = 0; q < NW; ++q) { w1 += s1[q * DW_OX + ox]; wx += sx[q * DW_OX + ox]; }
    wgrad_dw_store(j, i, w1, wx, tr);
}
// workgroups of NTH threads the reduction of job j takes
static long dw_blocks(long total, long nslots, int nth) { return nslots <= DW_FLAT_MAX ? (total + nth - 1) / nth : (total + DW_OX - 1) / DW_OX; }
__global__ __launch_bounds__(1024) void wgrad_dw_kernel(DwJob j) {
    __shared__ double sh[2 * 16 * DW_OX];
    wgrad_dw_block<1024>(j, blockIdx.x, sh);
}
static DwJob dw_job(long rows, int cin, int cout, long nslots, const float* PP, const double* red, const float* g3, const float* var, const float* gamma,
                    float eps, int use_bn, int is_training, float* dW) {
    DwJob j;
    j.rows = rows; j.cin = cin; j.cout = cout; j.nslots = (int)nslots; j.nblk = (int)dw_blocks((long)cin * cout, nslots, 256);      // as a ride in pass B (256 threads)
    j.PP = PP; j.red = red; j.g3 = g3; j.var = var; j.gamma = gamma; j.eps = eps; j.use_bn = use_bn; j.is_training = is_training; j.dW = dW;
    j.gq = 0; j.gc_real = 0; j.gxyz_first = 0; j.plain = 0; j.rowgrid = 0;
    j.PP2 = nullptr; j.dW2 = nullptr; j.cin2 = 0; j.nslots2 = 0; j.nblk2 = 0;
    return j;
}

// which kernel / plan a (layer, operand alignment) gets: shared by gspn_mlp_bwd_wgrad and gspn_mlp_bwd_dw so both find the same workspace layout
// ============================================================================================
// Pass A with known coefficients, lean form (r03): dW = act(X)^T . dY over one row chunk per workgroup, into the chunk's partial-tile slot
// (same workspace layout and chunking as wgrad_stream_kernel, so the reductions that follow do not care which kernel ran).
// The streaming kernel builds both MFMA operands while it READS them from the raw LDS-DMA tiles -- relu(bn(x)) and
// dY = cA*[mask]*dz + cB*y + cC are recomputed by every wave for every MFMA that consumes the element: 11-23 VALU instructions per MFMA
// (profiles/r03_sq_insts_by_kernel.txt), and fp32 MFMAs do not hide them.  Here a stage of 32 rows is loaded into registers, each element
// is transformed ONCE on its way into LDS (row-major, 16-byte stores), and the MFMA loop reads finished operands: a workgroup covers a
// block of MB x NB 32x32 tiles of dW, 3/NB + 5/MB vector instructions per MFMA.  Full stages only (rows and chunk length multiples of
// 32), 16-byte aligned pitches, per-thread constant offsets on uniform bases as in bwd_lean_kernel.
// ============================================================================================
template <int MB, int NB, int PK, bool ACT>
__global__ __launch_bounds__(256) void wgrad_lean_kernel(int rows, int cin, int cout, gspn_dy_args a, const float* __restrict__ X, int ldx,
                                                         const float* __restrict__ in_scale, const float* __restrict__ in_shift, float* __restrict__ PP,
                                                         int rpc, int nch, int nbm, int nbn, int pool_sh) {
    constexpr int CM = 32 * MB, CN = 32 * NB;
    constexpr int WGM = MB >= 4 ? 4 : MB, WGN = 4 / WGM;        // wave grid over the block's tiles
    constexpr int AM = MB / WGM, BNW = NB / WGN;
    static_assert(WGM * WGN == 4 && AM >= 1 && BNW >= 1 && AM * WGM == MB && BNW * WGN == NB, "4 waves tile the block");
    constexpr int LDXS = CM + 4, LDYS = CN + 4;                 // (+4 floats: the two k halves of an operand read start 4 banks apart)
    __shared__ __attribute__((aligned(16))) float sX[32 * LDXS];
    __shared__ __attribute__((aligned(16))) float sD[32 * LDYS];
    extern __shared__ __attribute__((aligned(16))) float s_chan[];         // [2][cpin] input scale / shift, [5][cpout] scale, -shift, cA, cB, cC
    const int cpin = (cin + 3) / 4 * 4 + 4, cpout = (cout + 3) / 4 * 4 + 4;
    float* s_in = s_chan;
    float* s_out = s_chan + 2 * cpin;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    // block -> (chunk, output block): the blocks of one chunk are adjacent in launch order AND on the same XCD (id % 8)
    const int nblk = nbm * nbn;
    const int bid = blockIdx.x, rest = bid >> 3;
    const int blk = rest % nblk;
    const int chunk = (rest / nblk) * 8 + (bid & 7);
    if (chunk >= nch) return;
    const int m0 = (blk % nbm) * CM, n0 = (blk / nbm) * CN;
    const int r_begin = chunk * rpc;
    const int nst = (min(r_begin + rpc, rows) - r_begin) >> 5;         // stages of 32 rows
    if constexpr (ACT) {
        for (int i = t; i < cpin; i += 256) {
            s_in[i] = i < cin ? in_scale[i] : 1.f;
            s_in[cpin + i] = i < cin ? in_shift[i] : 0.f;
        }
    }
    for (int i = t; i < cpout; i += 256) {
        const bool in = i < cout;
        s_out[i] = in ? a.scale[i] : 1.f;
        s_out[cpout + i] = in ? -a.shift[i] : 0.f;
        s_out[2 * cpout + i] = in ? a.cA[i] : 0.f;
        s_out[3 * cpout + i] = in ? a.cB[i] : 0.f;
        s_out[4 * cpout + i] = in ? a.cC[i] : 0.f;
    }
    // X block: 32 rows x CM columns = 8 MB quads per row; thread t takes quad (t % (8 MB)) of rows t / (8 MB) + i * (32 / MB), i < MB
    constexpr int XQ = 8 * MB, XR = 32 / MB;
    const int xq = (t % XQ) * 4, xr = t / XQ;
    const unsigned ox = (unsigned)(xr * ldx + m0 + xq) * 4u;
    constexpr int YQ = 8 * NB, YR = 32 / NB;
    const int yq = (t % YQ) * 4, yr = t / YQ;
    const unsigned oy = (unsigned)(yr * a.ldy + n0 + yq) * 4u;
    const unsigned oz = PK ? (unsigned)(n0 + yq) * 4u : (unsigned)(yr * a.ldz + n0 + yq) * 4u;
    float4 rx[MB], ry[NB], rz[PK ? 1 : NB];
    int4 rarg[PK ? 1 : 1];
    auto fetch = [&](int s) {
        const int r0 = r_begin + (s << 5);
        const char* xb = reinterpret_cast<const char*>(X + (size_t)r0 * ldx);
#pragma unroll
        for (int i = 0; i < MB; ++i) rx[i] = *reinterpret_cast<const float4*>(xb + (size_t)(i * XR) * ldx * 4 + ox);
        const char* yb = reinterpret_cast<const char*>(a.Y + (size_t)r0 * a.ldy);
#pragma unroll
        for (int i = 0; i < NB; ++i) ry[i] = *reinterpret_cast<const float4*>(yb + (size_t)(i * YR) * a.ldy * 4 + oy);
        if constexpr (PK != 0) {                                 // the 32 rows of a stage lie in ONE pool group (ns = 32, or a power of two >= 64)
            const size_t g = (size_t)(r0 >> pool_sh) * cout;
            rarg[0] = *reinterpret_cast<const int4*>(reinterpret_cast<const char*>(a.pool_arg + g) + oz);
            rz[0] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.dPool + g) + oz);
        } else {
            const char* zb = reinterpret_cast<const char*>(a.dZ + (size_t)r0 * a.ldz);
#pragma unroll
            for (int i = 0; i < NB; ++i) rz[i] = *reinterpret_cast<const float4*>(zb + (size_t)(i * YR) * a.ldz * 4 + oz);
        }
    };
    auto commit = [&](int s) {
        {
            float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (ACT) {
                const float4 q_sc = *reinterpret_cast<const float4*>(s_in + m0 + xq), q_sh = *reinterpret_cast<const float4*>(s_in + cpin + m0 + xq);
                sc[0] = q_sc.x; sc[1] = q_sc.y; sc[2] = q_sc.z; sc[3] = q_sc.w;
                sh[0] = q_sh.x; sh[1] = q_sh.y; sh[2] = q_sh.z; sh[3] = q_sh.w;
            }
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                float4 v;
                v.x = act1(rx[i].x, ACT, sc[0], sh[0]); v.y = act1(rx[i].y, ACT, sc[1], sh[1]);
                v.z = act1(rx[i].z, ACT, sc[2], sh[2]); v.w = act1(rx[i].w, ACT, sc[3], sh[3]);
                *reinterpret_cast<float4*>(sX + (xr + i * XR) * LDXS + xq) = v;
            }
        }
        const int k = n0 + yq;
        const float4 q_sc = *reinterpret_cast<const float4*>(s_out + k), q_ns = *reinterpret_cast<const float4*>(s_out + cpout + k);
        const float4 q_a = *reinterpret_cast<const float4*>(s_out + 2 * cpout + k), q_b = *reinterpret_cast<const float4*>(s_out + 3 * cpout + k);
        const float4 q_c = *reinterpret_cast<const float4*>(s_out + 4 * cpout + k);
        const float sc[4] = {q_sc.x, q_sc.y, q_sc.z, q_sc.w}, ns[4] = {q_ns.x, q_ns.y, q_ns.z, q_ns.w};
        const float cA[4] = {q_a.x, q_a.y, q_a.z, q_a.w}, cB[4] = {q_b.x, q_b.y, q_b.z, q_b.w}, cC[4] = {q_c.x, q_c.y, q_c.z, q_c.w};
        const int off0 = PK ? (((r_begin + (s << 5)) & ((1 << pool_sh) - 1)) + yr) : 0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const float yv[4] = {ry[i].x, ry[i].y, ry[i].z, ry[i].w};
            float zv[4];
            if constexpr (PK != 0) {
                const int off = off0 + i * YR;
                const int av[4] = {rarg[0].x, rarg[0].y, rarg[0].z, rarg[0].w};
                const float dv[4] = {rz[0].x, rz[0].y, rz[0].z, rz[0].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) zv[j] = av[j] == off ? dv[j] : 0.f;
            } else {
                zv[0] = rz[i].x; zv[1] = rz[i].y; zv[2] = rz[i].z; zv[3] = rz[i].w;
            }
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dyh = yv[j] * sc[j] > ns[j] ? zv[j] : 0.f;
                o[j] = __builtin_fmaf(cA[j], dyh, __builtin_fmaf(cB[j], yv[j], cC[j]));       // the streaming kernel's own form of dY
            }
            *reinterpret_cast<float4*>(sD + (yr + i * YR) * LDYS + yq) = make_float4(o[0], o[1], o[2], o[3]);
        }
    };
    f32x16 acc[AM][BNW];
#pragma unroll
    for (int x = 0; x < AM; ++x)
#pragma unroll
        for (int y = 0; y < BNW; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    const int gm = wave % WGM, gn = wave / WGM;                 // this wave's tiles: rows (gm + WGM x) of the block, columns (gn + WGN y)
    const float* pa = sX + kh * LDXS + gm * 32 + l31;
    const float* pb = sD + kh * LDYS + gn * 32 + l31;
    if (nst > 0) fetch(0);
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
        commit(s);
        __syncthreads();
        if (s + 1 < nst) fetch(s + 1);
#pragma unroll
        for (int k0 = 0; k0 < 32; k0 += 4) {
            float av[2][AM], bw[2][BNW];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int x = 0; x < AM; ++x) av[u][x] = pa[(k0 + 2 * u) * LDXS + x * WGM * 32];
#pragma unroll
                for (int y = 0; y < BNW; ++y) bw[u][y] = pb[(k0 + 2 * u) * LDYS + y * WGN * 32];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int x = 0; x < AM; ++x)
#pragma unroll
                    for (int y = 0; y < BNW; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][x], bw[u][y], acc[x][y], 0, 0, 0);
        }
        __syncthreads();
    }
    float* P1 = PP + (size_t)chunk * 2 * cin * cout;
#pragma unroll
    for (int x = 0; x < AM; ++x)
#pragma unroll
        for (int y = 0; y < BNW; ++y) {
            const int mrow = m0 + (gm + WGM * x) * 32 + 4 * kh;
            const unsigned lo = (unsigned)(n0 + (gn + WGN * y) * 32 + l31) * 4u;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                *reinterpret_cast<float*>(reinterpret_cast<char*>(P1 + (size_t)(mrow + (r & 3) + 8 * (r >> 2)) * cout) + lo) = acc[x][y][r];
        }
}
// the lean kernel takes a known-coefficient pass A when its shape assumptions hold (GSPN_WGRAD_LEAN=0: never); false otherwise
static bool wgrad_lean_try(long rows, int cin, int cout, const gspn_dy_args* a, const float* X, int ldx, const float* in_scale, const float* in_shift,
                           float* PP, const WgradPlan& p, hipStream_t st) {
    static const int on = env_int("GSPN_WGRAD_LEAN", 1);
    if (!on || p.shared || (rows & 31) || (p.rpc & 31) || (cin & 31) || (cout & 63)) return false;
    const bool pooled = a->dZ == nullptr;
    // Measured against the LDS-DMA streaming kernel on MI355X (tools/wgrad_ablate.py, us lean / streaming): it wins where the streaming
    // kernel's on-read operand construction is most expensive -- the pooled top layers with >= 128 output channels (131072 x 64^T x 128:
    // 42 / 52, 32768 x 128^T x 256: 56 / 59, 1 M x 128^T x 256: 1021 / 1126) -- and loses on the dense layers (262144 x 64^T x 64: 54 / 48,
    // 32768 x 128^T x 128: 56 / 33, 4096 x 384^T x 256: 39 / 24): two barriers per 32-row stage and the chunk count planned for the streaming
    // kernel's finer tiles leave it short of workgroups there.  GSPN_WGRAD_LEAN=2 takes every eligible shape (A/B hook).
    if (on != 2 && !(pooled && cout >= 128)) return false;
    int pool_sh = 0;
    if (pooled) {
        if (a->ns < 32 || (a->ns & (a->ns - 1)) || rows % a->ns) return false;
        pool_sh = __builtin_ctz(a->ns);
    }
    const long ldmax = std::max(std::max((long)ldx, (long)a->ldy), (long)(pooled ? 0 : a->ldz));
    if (p.rpc * ldmax >= (1L << 30) || rows * ldmax >= (1L << 40)) return false;
    const int mt = cin / 32, nt = cout / 32;
    const int MBs = (mt % 4 == 0) ? 4 : ((mt % 2 == 0) ? 2 : 0);
    if (!MBs) return false;
    const int NBs = 2;                                         // (cout % 64 == 0)
    const int nbm = mt / MBs, nbn = nt / NBs;
    const dim3 g((unsigned)((p.nch + 7) / 8 * 8 * nbm * nbn));
    const size_t dyn = sizeof(float) * (2 * chan_pad(cin) + 5 * chan_pad(cout));
    const bool act = in_scale != nullptr;
#define WL_GO(MB_, PK_, A_) hipLaunchKernelGGL((wgrad_lean_kernel<MB_, 2, PK_, A_>), g, dim3(256), dyn, st, (int)rows, cin, cout, *a, X, ldx, in_scale, in_shift, PP, \
                                               (int)p.rpc, (int)p.nch, nbm, nbn, pool_sh)
#define WL_A(MB_, PK_) do { if (act) WL_GO(MB_, PK_, true); else WL_GO(MB_, PK_, false); } while (0)
#define WL_P(MB_) do { if (pooled) WL_A(MB_, 1); else WL_A(MB_, 0); } while (0)
    if (MBs == 4) WL_P(4); else WL_P(2);
#undef WL_P
#undef WL_A
#undef WL_GO
    return true;
}

static WgradPlan wgrad_choose(long rows, int cin, int cout, const gspn_dy_args* a, const float* X, int ldx, bool* use_stream_out) {
    const bool pooled = a->dZ == nullptr;
    // streaming kernel: 16-byte aligned rows, 32-bit in-chunk offsets, pool groups that tile the stage
    bool use_stream = vec_ok(X, ldx) && vec_ok(a->Y, a->ldy) && (pooled || vec_ok(a->dZ, a->ldz)) && ldx >= 4 && cout >= 4;
    WgradPlan p = wgrad_plan(rows, cin, cout, false, pooled);
    if (use_stream) {
        const long ldmax = ldx > a->ldy ? (ldx > a->ldz ? ldx : a->ldz) : (a->ldy > a->ldz ? a->ldy : a->ldz);
        if (p.rpc * ldmax >= (1L << 31)) use_stream = false;
        if (pooled && !(a->ns >= 16 && a->ns % 2 == 0 && (a->ns % p.TKW == 0 || p.TKW % a->ns == 0) && rows % a->ns == 0)) use_stream = false;
    }
    if (!use_stream) p = wgrad_plan(rows, cin, cout, true);
    *use_stream_out = use_stream;
    return p;
}

static int wgrad_impl(long rows, int cin, int cout, const gspn_dy_args* a, const float* X, int ldx,
                      const float* in_scale, const float* in_shift, const float* mean, const float* var, const float* gamma,
                      float eps, int use_bn, int is_training, float* work, float* cA, float* cB, float* cC,
                      float* dgamma, float* dbeta, float* dbias, float* dW, void* stream, const GatherSrc* gsrc, bool known = false) {
    // gsrc: virtual input rows (GatherSrc); cin is then the internal width 4*cq + 4 and X / ldx are unused
    // known: a->cA/cB/cC are already final (gspn_mlp_bwd_coef): one GEMM with dY as the operand, no reductions, no coefficient kernel
    if (rows <= 0 || cin <= 0 || cout <= 0 || (!gsrc && ldx < cin) || !a || !a->Y || !a->scale || !a->shift || !work) return GSPN_ERR_ARG;
    if (!a->dZ && !(a->dPool && a->pool_arg && a->ns > 0)) return GSPN_ERR_ARG;
    if ((in_scale == nullptr) != (in_shift == nullptr)) return GSPN_ERR_ARG;
    if (use_bn && (!mean || !var)) return GSPN_ERR_ARG;
    if (cin > MAXCH || cout > MAXCH) return GSPN_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (rows >= (1L << 31)) return GSPN_ERR_UNSUPPORTED;
    const bool pooled = a->dZ == nullptr;
    const bool tr = use_bn && is_training;
    bool use_stream;
    WgradPlan p;
    if (gsrc) {
        if (pooled || !vec_ok(a->Y, a->ldy) || !vec_ok(a->dZ, a->ldz) || cout < 4) return GSPN_ERR_UNSUPPORTED;
        p = wgrad_plan(rows, cin, cout, false, false);
        const long ldmax = a->ldy > a->ldz ? a->ldy : a->ldz;
        if (p.rpc * ldmax >= (1L << 31) || p.MTs != 1 || !(p.NTs == 1 || p.NTs == 2)) return GSPN_ERR_UNSUPPORTED;
        use_stream = true;
    } else {
        p = wgrad_choose(rows, cin, cout, a, X, ldx, &use_stream);
    }
    if (known && (!use_stream || p.MTs != 1 || !a->cA || !a->cB || !a->cC)) return GSPN_ERR_UNSUPPORTED;
    char* wb = reinterpret_cast<char*>(work);
    if (reinterpret_cast<uintptr_t>(wb) % 16) return GSPN_ERR_ARG;
    double* red = reinterpret_cast<double*>(wb);
    float* g3 = reinterpret_cast<float*>(wb + ws_off_g3(cout));
    float* RP = reinterpret_cast<float*>(wb + ws_off_rp(cin, cout));
    float* GP = reinterpret_cast<float*>(wb + ws_off_gp(p.nch, cin, cout));
    float* PP = reinterpret_cast<float*>(wb + ws_off_pp(p.nch, cin, cout));
    if (p.shared) {
        hipError_t e = hipMemsetAsync(PP, 0, sizeof(float) * (size_t)p.nslots * 2 * cin * cout, st);
        if (e != hipSuccess) return (int)e;
    }
    const float* mu = use_bn ? mean : nullptr;
    const float* vr = use_bn ? var : nullptr;
    if (use_stream) {
        int launched = 0;
        if (!launched && known && !gsrc && wgrad_lean_try(rows, cin, cout, a, X, ldx, in_scale, in_shift, PP, p, st)) launched = 1;
        const dim3 grid((unsigned)((p.nch + 7) / 8 * 8 * p.nrow * p.ncol));
#define WS_ARGS (int)rows, cin, cout, *a, X, ldx, in_scale, in_shift, mu, vr, eps, RP, GP, PP, (int)p.rpc, (int)p.nslots, p.shared, (int)p.nch, p.nrow, p.ncol
#define WS_GO(MT_, NT_, TKW_, G_, P_) hipLaunchKernelGGL((wgrad_stream_kernel<MT_, NT_, TKW_, G_, P_>), grid, dim3(256), 0, st, WS_ARGS, GatherSrc{})
#define WS_GATHER(MT_, NT_, TKW_)                                                                \
        if (!launched && gsrc && p.MTs == MT_ && p.NTs == NT_ && p.TKW == TKW_) {               \
            if (tr) hipLaunchKernelGGL((wgrad_stream_kernel<MT_, NT_, TKW_, true, false, true>), grid, dim3(256), 0, st, WS_ARGS, *gsrc);    \
            else hipLaunchKernelGGL((wgrad_stream_kernel<MT_, NT_, TKW_, false, false, true>), grid, dim3(256), 0, st, WS_ARGS, *gsrc);      \
            launched = 1;                                                                       \
        }
#define WS_KNOWN(MT_, NT_, TKW_)                                                                 \
        if (!launched && known && p.MTs == MT_ && p.NTs == NT_ && p.TKW == TKW_) {              \
            if (gsrc) hipLaunchKernelGGL((wgrad_stream_kernel<MT_, NT_, TKW_, false, false, true, true>), grid, dim3(256), 0, st, WS_ARGS, *gsrc);          \
            else if (pooled) hipLaunchKernelGGL((wgrad_stream_kernel<MT_, NT_, TKW_, false, true, false, true>), grid, dim3(256), 0, st, WS_ARGS, GatherSrc{}); \
            else hipLaunchKernelGGL((wgrad_stream_kernel<MT_, NT_, TKW_, false, false, false, true>), grid, dim3(256), 0, st, WS_ARGS, GatherSrc{});        \
            launched = 1;                                                                       \
        }
        if (!launched) { WS_KNOWN(1, 1, 64) WS_KNOWN(1, 2, 32) }
        if (!launched && known && !gsrc && p.MTs == 1 && p.NTs == 4 && p.TKW == 16) {
            if (pooled) hipLaunchKernelGGL((wgrad_stream_kernel<1, 4, 16, false, true, false, true>), grid, dim3(256), 0, st, WS_ARGS, GatherSrc{});
            else hipLaunchKernelGGL((wgrad_stream_kernel<1, 4, 16, false, false, false, true>), grid, dim3(256), 0, st, WS_ARGS, GatherSrc{});
            launched = 1;
        }
        if (known && !launched) return GSPN_ERR_UNSUPPORTED;
#undef WS_KNOWN
        WS_GATHER(1, 1, 64) WS_GATHER(1, 2, 32)
        if (gsrc && !launched) return GSPN_ERR_UNSUPPORTED;
#undef WS_GATHER
#define WS_TRY(MT_, NT_, TKW_)                                                                   \
        if (!launched && p.MTs == MT_ && p.NTs == NT_ && p.TKW == TKW_) {                       \
            if (tr) { if (pooled) WS_GO(MT_, NT_, TKW_, true, true); else WS_GO(MT_, NT_, TKW_, true, false); }     \
            else    { if (pooled) WS_GO(MT_, NT_, TKW_, false, true); else WS_GO(MT_, NT_, TKW_, false, false); }   \
            launched = 1;                                                                       \
        }
        WS_TRY(1, 1, 64) WS_TRY(1, 2, 32) WS_TRY(1, 4, 16)
        WS_TRY(2, 1, 32) WS_TRY(2, 2, 32) WS_TRY(2, 4, 16)
        WS_TRY(3, 1, 32) WS_TRY(3, 2, 16)
        WS_TRY(4, 1, 32) WS_TRY(4, 2, 16)
#undef WS_TRY
#undef WS_GO
#undef WS_ARGS
        if (!launched) return GSPN_ERR_UNSUPPORTED;
    } else {
        // any alignment / pool size: register-staged kernel with scalar loads, 128x128 tiles
        const dim3 grid((unsigned)p.nch, p.nrow, p.ncol);
#define WG_ARGS rows, cin, cout, *a, X, ldx, in_scale, in_shift, mu, vr, eps, RP, GP, PP, p.rpc, (int)p.nslots
#define WG_GO(G_, P_) hipLaunchKernelGGL((mlp_bwd_wgrad_kernel<4, 4, 32, false, G_, P_>), grid, dim3(256), sizeof(float) * 2 * chan_pad(cin), st, WG_ARGS)
        if (tr) { if (pooled) WG_GO(true, true); else WG_GO(true, false); }
        else    { if (pooled) WG_GO(false, true); else WG_GO(false, false); }
#undef WG_GO
#undef WG_ARGS
    }
    const int cmax = cin > cout ? cin : cout;
    if (!known)
        hipLaunchKernelGGL(wgrad_small_reduce_kernel, dim3(cmax), dim3(256), 0, st, rows, cin, cout, (int)p.nch, RP, GP, red, g3, mean, var, gamma, eps,
                           use_bn, is_training, cA, cB, cC, dgamma, dbeta, dbias);
    if (dW) {
        DwJob j = dw_job(rows, cin, cout, p.nslots, PP, red, g3, var, gamma, eps, use_bn, is_training, dW);
        j.plain = known ? 1 : 0;
        if (gsrc) { j.gq = gsrc->cq; j.gc_real = gsrc->c_real; j.gxyz_first = gsrc->xyz_first; }
        hipLaunchKernelGGL(wgrad_dw_kernel, dim3((unsigned)dw_blocks((long)j.cin * j.cout, j.nslots, 1024)), dim3(1024), 0, st, j);
    }
    return gspn_launch_status();
}
extern "C" int gspn_mlp_bwd_wgrad(long rows, int cin, int cout, const gspn_dy_args* a, const float* X, int ldx,
                                  const float* in_scale, const float* in_shift, const float* mean, const float* var, const float* gamma,
                                  float eps, int use_bn, int is_training, float* work, float* cA, float* cB, float* cC,
                                  float* dgamma, float* dbeta, float* dbias, float* dW, void* stream) {
    return wgrad_impl(rows, cin, cout, a, X, ldx, in_scale, in_shift, mean, var, gamma, eps, use_bn, is_training, work, cA, cB, cC,
                      dgamma, dbeta, dbias, dW, stream, nullptr);
}

// ============================================================================================
// Early coefficients.  Training-mode BN's backward needs r0 = sum(dyh), r1 = sum(dyh*xhat) over ALL rows before any dY can be formed;
// pass A above avoids waiting for them by carrying a second product (Gx) through the GEMM -- twice the MFMA work.  When the two sums
// are available BEFORE pass A, the coefficients are final and pass A is one GEMM on dY (wgrad_stream_kernel<..., KNOWN>):
//   * top layer of a pooled stack: the upstream gradient is (groups, c) with one live row per group -> pool_rsum_kernel;
//   * every other layer l: its dz is the dX that pass B of layer l+1 writes -> that kernel's epilogue takes the sums (RsumArgs);
//   * gspn_mlp_bwd_coef turns the per-workgroup partials into cA/cB/cC, dgamma, dbeta, dbias (what wgrad_small_reduce_kernel does
//     for the two-product form).
// ============================================================================================
// every workgroup takes a slab of groups; thread (sub, col) walks the groups sub, sub + nsub, ... of the slab for its channel(s) in a
// fixed order and the nsub partial sums of a channel meet in LDS, again in a fixed order: deterministic, one partial row per workgroup
#ifndef RSUM_POOL_BLOCKS
#define RSUM_POOL_BLOCKS 1024
#endif
__global__ __launch_bounds__(256) void pool_rsum_kernel(long groups, int ns, int c, const float* __restrict__ dPool, const int* __restrict__ arg,
                                                        const float* __restrict__ Y, int ldy, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ var,
                                                        float eps, float* __restrict__ part, long gpb) {
    extern __shared__ float sred[];                       // [nsub][2][c]
    const int cw = (c <= 256 && 256 % c == 0) ? c : 256;  // channels covered by one row of threads
    const int nsub = 256 / cw;
    const int sub = threadIdx.x / cw, lc = threadIdx.x % cw;
    const long g0 = blockIdx.x * gpb, g1 = min(groups, g0 + gpb);
    for (int col = lc; col < c; col += cw) {
        const float sc = scale[col], sh = shift[col];
        const float rs = (float)(1.0 / sqrt((double)var[col] + (double)eps)), mr = -mean[col] * rs;
        float r0 = 0.f, r1 = 0.f;
        for (long g = g0 + sub; g < g1; g += nsub) {
            const float dp = dPool[g * c + col];
            const float yv = ldy ? Y[(g * ns + arg[g * c + col]) * ldy + col] : Y[g * c + col];     // ldy == 0: Y is (groups, c), y at the arg row
            const float dyh = relu_open(yv, sc, sh) ? dp : 0.f;
            r0 += dyh;
            r1 = __builtin_fmaf(dyh, __builtin_fmaf(yv, rs, mr), r1);
        }
        sred[(sub * 2 + 0) * c + col] = r0;
        sred[(sub * 2 + 1) * c + col] = r1;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * c; i += 256) {
        float v = 0.f;
        for (int q = 0; q < nsub; ++q) v += sred[(size_t)q * 2 * c + i];
        part[(size_t)blockIdx.x * 2 * c + i] = v;
    }
}
// floats of a partial-sum buffer [nparts][2][c] large enough for gspn_pool_rsum and for pass B's epilogue (one row per workgroup)
extern "C" long gspn_rsum_part_floats(long rows, int c) {
    if (c <= 0) return GSPN_ERR_ARG;
    // pass B writes one partial row per workgroup of row_grid(rows, yt, bpc), bpc = bwd_bpc_narrow() (or 2 for 128-column tiles): size the
    // buffer from the same value, so the GSPN_BWD_BPC tuning hook cannot push the epilogue past it
    const int bpc = bwd_bpc_narrow() > 4 ? bwd_bpc_narrow() : 4;
    long n = row_grid(rows > 0 ? rows : 1, 1, bpc);
    if (n < RSUM_POOL_BLOCKS) n = RSUM_POOL_BLOCKS;
    return n * 2 * c;
}
extern "C" int gspn_pool_rsum(long groups, int ns, int c, const float* dPool, const int* arg, const float* Y, int ldy, const float* scale,
                              const float* shift, const float* mean, const float* var, float eps, float* part, int* nparts_out, void* stream) {
    if (groups <= 0 || ns <= 0 || c <= 0 || !dPool || !arg || !Y || !scale || !shift || !mean || !var || !part || !nparts_out) return GSPN_ERR_ARG;
    long nblk = groups < RSUM_POOL_BLOCKS ? groups : RSUM_POOL_BLOCKS;
    const long gpb = (groups + nblk - 1) / nblk;
    nblk = (groups + gpb - 1) / gpb;
    const int cw = (c <= 256 && 256 % c == 0) ? c : 256;
    hipLaunchKernelGGL(pool_rsum_kernel, dim3((unsigned)nblk), dim3(256), sizeof(float) * 2 * c * (256 / cw), (hipStream_t)stream, groups, ns, c, dPool, arg, Y, ldy,
                       scale, shift, mean, var, eps, part, gpb);
    *nparts_out = (int)nblk;
    return gspn_launch_status();
}
// top layer of a DENSE stack (the feature-propagation stacks: the upstream gradient is a full (rows, c) tensor): the two sums in one
// streaming pass over (dZ, Y) -- 2 x rows x c x 4 bytes, a quad of channels per thread, a slab of rows per workgroup, fixed order --
// so that this layer, too, runs pass A as one GEMM (or both passes as one launch) instead of the two-product form.
__global__ __launch_bounds__(256) void dense_rsum_kernel(long rows, int c, const float* __restrict__ dZ, int ldz, const float* __restrict__ Y, int ldy,
                                                         const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
                                                         const float* __restrict__ var, float eps, float* __restrict__ part, long rpb) {
    extern __shared__ float sred[];                       // [nsub][2][c]
    const int cq = c >> 2, nsub = 256 / cq;
    const int sub = threadIdx.x / cq, col = (threadIdx.x % cq) * 4;
    const long g0 = blockIdx.x * rpb, g1 = min(rows, g0 + rpb);
    float sc[4], ns[4], rs[4], mr[4], r0[4] = {0.f, 0.f, 0.f, 0.f}, r1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sc[j] = scale[col + j];
        ns[j] = -shift[col + j];
        rs[j] = (float)(1.0 / sqrt((double)var[col + j] + (double)eps));
        mr[j] = -mean[col + j] * rs[j];
    }
    if (sub < nsub) {
        long g = g0 + sub;
        for (; g + 3L * nsub < g1; g += 4L * nsub) {      // four rows in flight per thread
            float4 z[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                z[u] = *reinterpret_cast<const float4*>(dZ + (g + (long)u * nsub) * ldz + col);
                y[u] = *reinterpret_cast<const float4*>(Y + (g + (long)u * nsub) * ldy + col);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float zv[4] = {z[u].x, z[u].y, z[u].z, z[u].w}, yv[4] = {y[u].x, y[u].y, y[u].z, y[u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float dyh = yv[j] * sc[j] > ns[j] ? zv[j] : 0.f;       // relu_open
                    r0[j] += dyh;
                    r1[j] = __builtin_fmaf(dyh, __builtin_fmaf(yv[j], rs[j], mr[j]), r1[j]);
                }
            }
        }
        for (; g < g1; g += nsub) {
            const float4 z = *reinterpret_cast<const float4*>(dZ + g * ldz + col), y = *reinterpret_cast<const float4*>(Y + g * ldy + col);
            const float zv[4] = {z.x, z.y, z.z, z.w}, yv[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dyh = yv[j] * sc[j] > ns[j] ? zv[j] : 0.f;
                r0[j] += dyh;
                r1[j] = __builtin_fmaf(dyh, __builtin_fmaf(yv[j], rs[j], mr[j]), r1[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sred[(sub * 2 + 0) * c + col + j] = r0[j];
            sred[(sub * 2 + 1) * c + col + j] = r1[j];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * c; i += 256) {
        float v = 0.f;
        for (int q = 0; q < nsub; ++q) v += sred[(size_t)q * 2 * c + i];
        part[(size_t)blockIdx.x * 2 * c + i] = v;
    }
}
// part [*nparts_out][2][c] <- (sum dyh, sum dyh * xhat) of a layer with a dense upstream gradient dZ (rows, ldz) and output Y (rows, ldy):
// the input of gspn_mlp_bwd_coef.  c a multiple of 4 with c / 4 <= 256, 16-byte aligned pitches; GSPN_ERR_UNSUPPORTED otherwise.
extern "C" int gspn_dense_rsum(long rows, int c, const float* dZ, int ldz, const float* Y, int ldy, const float* scale, const float* shift,
                               const float* mean, const float* var, float eps, float* part, int* nparts_out, void* stream) {
    if (rows <= 0 || c <= 0 || !dZ || !Y || ldz < c || ldy < c || !scale || !shift || !mean || !var || !part || !nparts_out) return GSPN_ERR_ARG;
    if ((c & 3) || (c >> 2) > 256 || !vec_ok(dZ, ldz) || !vec_ok(Y, ldy)) return GSPN_ERR_UNSUPPORTED;
    const int nsub = 256 / (c >> 2);
    long nblk = (rows + 4L * nsub - 1) / (4L * nsub);                      // at least one four-row round per thread
    if (nblk > RSUM_POOL_BLOCKS) nblk = RSUM_POOL_BLOCKS;
    if (nblk < 1) nblk = 1;
    const long rpb = (rows + nblk - 1) / nblk;
    nblk = (rows + rpb - 1) / rpb;
    hipLaunchKernelGGL(dense_rsum_kernel, dim3((unsigned)nblk), dim3(256), sizeof(float) * 2 * c * nsub, (hipStream_t)stream, rows, c, dZ, ldz, Y, ldy, scale, shift,
                       mean, var, eps, part, rpb);
    *nparts_out = (int)nblk;
    return gspn_launch_status();
}
struct CoefJob {
    long rows; int c, nparts; const float* part; const float* mean; const float* var; const float* gamma; float eps;
    float *cA, *cB, *cC, *dgamma, *dbeta, *dbias;
};
__device__ __forceinline__ void bwd_coef_block(const CoefJob& q, int n, double (*sh)[4]);
__global__ __launch_bounds__(256) void bwd_coef_kernel(long rows, int c, int nparts, const float* __restrict__ part, const float* __restrict__ mean,
                                                       const float* __restrict__ var, const float* __restrict__ gamma, float eps,
                                                       float* __restrict__ cA, float* __restrict__ cB, float* __restrict__ cC,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias) {
    __shared__ double sh[2][4];
    const CoefJob q{rows, c, nparts, part, mean, var, gamma, eps, cA, cB, cC, dgamma, dbeta, dbias};
    bwd_coef_block(q, blockIdx.x, sh);
}
// r04: the coefficient kernel of the PREVIOUS layer and the dW reduction of THIS layer both depend on the fused backward launch only and
// on nothing else: one launch of c + nblk workgroups instead of two dependent ~3.5 us kernels (gspn_mlp_bwd_fused_coef)
__global__ __launch_bounds__(256) void bwd_coef_dw_kernel(CoefJob q, DwJob j) {
    __shared__ double sh[2 * 4 * DW_OX];
    if ((int)blockIdx.x < q.c) bwd_coef_block(q, blockIdx.x, reinterpret_cast<double (*)[4]>(sh));
    else wgrad_dw_block<256>(j, blockIdx.x - (unsigned)q.c, sh);
}
__device__ __forceinline__ void bwd_coef_block(const CoefJob& q, int n, double (*sh)[4]) {
    const long rows = q.rows; const int c = q.c, nparts = q.nparts;
    const float* __restrict__ part = q.part; const float* __restrict__ mean = q.mean; const float* __restrict__ var = q.var; const float* __restrict__ gamma = q.gamma;
    const float eps = q.eps;
    float* __restrict__ cA = q.cA; float* __restrict__ cB = q.cB; float* __restrict__ cC = q.cC;
    float* __restrict__ dgamma = q.dgamma; float* __restrict__ dbeta = q.dbeta; float* __restrict__ dbias = q.dbias;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float g0 = 1.f, v0 = 1.f, m0 = 0.f;                                // (asked for before the partial rows, see bn_finalize_kernel)
    if (t == 0) { if (gamma) g0 = gamma[n]; v0 = var[n]; m0 = mean[n]; }
    double a0 = 0.0, a1 = 0.0;
    part_rows_sum2(part, (size_t)2 * c, n, c + n, nparts, t, a0, a1);
    a0 = wave_sum_f64(a0); a1 = wave_sum_f64(a1);
    if (lane == 0) { sh[0][wave] = a0; sh[1][wave] = a1; }
    __syncthreads();
    if (t != 0) return;
    const double r0 = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
    const double r1 = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    const double R = (double)rows;
    const double g = (double)g0;
    const double rstd = 1.0 / sqrt((double)v0 + (double)eps);
    const double mu = (double)m0;
    cA[n] = (float)(g * rstd);                                         // same formulas as wgrad_small_reduce_kernel (training-mode BN)
    cB[n] = (float)(-g * rstd * rstd * (r1 / R));
    cC[n] = (float)(-g * rstd * (r0 / R - mu * rstd * (r1 / R)));
    if (dgamma) dgamma[n] = (float)r1;
    if (dbeta) dbeta[n] = (float)r0;
    if (dbias) dbias[n] = 0.f;                                        // sum(dY) is exactly 0 under batch statistics
}
// coefficients of a training-mode BN layer from partial sums [nparts][2][c] (gspn_pool_rsum, or pass B's epilogue: gspn_mlp_bwd_data_rsum)
extern "C" int gspn_mlp_bwd_coef(long rows, int c, int nparts, const float* part, const float* mean, const float* var, const float* gamma, float eps,
                                 float* cA, float* cB, float* cC, float* dgamma, float* dbeta, float* dbias, void* stream) {
    if (rows <= 0 || c <= 0 || nparts <= 0 || !part || !mean || !var || !cA || !cB || !cC) return GSPN_ERR_ARG;
    hipLaunchKernelGGL(bwd_coef_kernel, dim3(c), dim3(256), 0, (hipStream_t)stream, rows, c, nparts, part, mean, var, gamma, eps, cA, cB, cC, dgamma, dbeta, dbias);
    return gspn_launch_status();
}
static int gather_src(const gspn_gather_args* g, GatherSrc* out);
// pass A with final coefficients in a->cA/cB/cC (one GEMM).  g: gather descriptor of a fused-front-end first layer, or NULL (then X/ldx/
// in_scale/in_shift as in gspn_mlp_bwd_wgrad).  dW may be NULL (sum of the partial tiles left to gspn_mlp_bwd_data_dw with use_bn = 0).
extern "C" int gspn_mlp_bwd_wgrad_known(long rows, int cin, int cout, const gspn_dy_args* a, const float* X, int ldx, const float* in_scale,
                                        const float* in_shift, const gspn_gather_args* g, float* work, float* dW, void* stream) {
    if (g) {
        GatherSrc gs;
        const int rc = gather_src(g, &gs);
        if (rc) return rc;
        if (!dW) return GSPN_ERR_ARG;
        return wgrad_impl(rows, 4 * gs.cq + 4, cout, a, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0, 0, work, nullptr, nullptr, nullptr,
                          nullptr, nullptr, nullptr, dW, stream, &gs, true);
    }
    return wgrad_impl(rows, cin, cout, a, X, ldx, in_scale, in_shift, nullptr, nullptr, nullptr, 0.f, 0, 0, work, nullptr, nullptr, nullptr,
                      nullptr, nullptr, nullptr, dW, stream, nullptr, true);
}

// ============================================================================================
// Fused set-abstraction front end (pointnet_util.py:36-52 + the first conv2d of :109-113): the grouped tensor
// (b, npoint, nsample, 3+c) is never written.  gspn_sa_rel leaves, per grouped row, its source row and its centred coordinates
// (20 bytes instead of 4*(3+c)); the first layer's forward GEMM and its weight-gradient pass then take every input row straight
// from the (b, n, c) feature tensor through the LDS-DMA's per-lane addresses (GatherSrc).
// ============================================================================================
__global__ void sa_rel_kernel(long total, int n, int m, int ns, const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                              const float* __restrict__ shift, const int* __restrict__ idx, float* __restrict__ rel, int* __restrict__ gidx) {
    for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < total; r += (long)gridDim.x * blockDim.x) {
        const long q = r / ns;                       // (scene, query)
        const int scene = (int)(q / m);
        const int src = scene * n + idx[r];
        const float* p = xyz + (size_t)src * 3;
        const float* c = new_xyz + (size_t)q * 3;
        float dx = p[0] - c[0], dy = p[1] - c[1], dz = p[2] - c[2];                                          // :41-42
        if (shift) {                                 // model_rpointnet.py:56-57: grouped_xyz -= tile(shift_pred) -- a second fp32 subtraction
            const float* s = shift + (size_t)q * 3;
            dx -= s[0]; dy -= s[1]; dz -= s[2];
        }
        *reinterpret_cast<float4*>(rel + r * 4) = make_float4(dx, dy, dz, 0.f);
        gidx[r] = src;
    }
}
extern "C" int gspn_sa_rel_shift(int b, int n, int m, int ns, const float* xyz, const float* new_xyz, const float* shift, const int* idx,
                                 float* rel, int* gidx, void* stream) {
    if (b < 0 || n <= 0 || m < 0 || ns <= 0) return GSPN_ERR_ARG;
    const long total = (long)b * m * ns;
    if (total == 0) return 0;
    if (!xyz || !new_xyz || !idx || !rel || !gidx || ((uintptr_t)rel % 16)) return GSPN_ERR_ARG;
    if ((long)b * n >= (1L << 31)) return GSPN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(sa_rel_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, total, n, m, ns, xyz, new_xyz, shift, idx, rel, gidx);
    return gspn_launch_status();
}
extern "C" int gspn_sa_rel(int b, int n, int m, int ns, const float* xyz, const float* new_xyz, const int* idx, float* rel, int* gidx, void* stream) {
    return gspn_sa_rel_shift(b, n, m, ns, xyz, new_xyz, nullptr, idx, rel, gidx, stream);
}
static int gather_src(const gspn_gather_args* g, GatherSrc* out) {
    if (!g || !g->feat || !g->gidx || !g->rel || g->c <= 0 || g->ldf < g->c || (g->ldf & 3)) return GSPN_ERR_ARG;
    if (((uintptr_t)g->feat % 16) || ((uintptr_t)g->rel % 16)) return GSPN_ERR_ARG;
    out->feat = g->feat; out->gidx = g->gidx; out->rel = g->rel;
    out->cq = g->ldf / 4; out->c_real = g->c; out->xyz_first = g->xyz_first ? 1 : 0;
    return 0;
}
extern "C" int gspn_mlp_gather_cin(const gspn_gather_args* g) { return (g && g->ldf > 0) ? g->ldf + 4 : GSPN_ERR_ARG; }
extern "C" int gspn_mlp_fwd_gather(long rows, const gspn_gather_args* g, int cout, const float* W, const float* bias, float* Y, int ldy,
                                   float* stats, void* stream) {
    GatherSrc gs;
    const int rc = gather_src(g, &gs);
    if (rc) return rc;
    return mlp_fwd_impl(rows, 4 * gs.cq + 4, cout, nullptr, 0, nullptr, nullptr, W, bias, Y, ldy, stats, PoolOut{nullptr, nullptr}, stream, &gs);
}
extern "C" int gspn_mlp_bwd_wgrad_gather(long rows, const gspn_gather_args* g, int cout, const gspn_dy_args* a, const float* mean, const float* var,
                                         const float* gamma, float eps, int use_bn, int is_training, float* work, float* cA, float* cB, float* cC,
                                         float* dgamma, float* dbeta, float* dbias, float* dW, void* stream) {
    GatherSrc gs;
    const int rc = gather_src(g, &gs);
    if (rc) return rc;
    if (!dW) return GSPN_ERR_ARG;                    // the row mapping back to the caller's dW happens in this call's reduction
    return wgrad_impl(rows, 4 * gs.cq + 4, cout, a, nullptr, 0, nullptr, nullptr, mean, var, gamma, eps, use_bn, is_training, work, cA, cB, cC,
                      dgamma, dbeta, dbias, dW, stream, &gs);
}

// The deferred last kernel of pass A: dW from the partial tiles / sums that gspn_mlp_bwd_wgrad(..., dW = NULL, ...) left in `work`.
// Nothing downstream of the layer needs dW before the optimiser, so a caller may run this on another stream (after the wgrad call,
// with the same rows/cin/cout/a/X/ldx so that the workspace layout is found again) and let it overlap the next layer's kernels.
// ============================================================================================
// Pre-aggregated first layer.  The first 1x1 layer of an SA module multiplies GROUPED rows [feat[gidx[r]] | rel[r]] -- every point's
// feature row 8x over at nsample 32 / 4x decimation -- and that of an FP module multiplies interpolated rows sum_t w_t * feat[idx_t[r]].
// The layer is linear, so the feature part can be multiplied BEFORE the grouping / interpolation, on the source points:
//     F = feat . W_feat                                   (source rows only: 8x fewer than grouped rows; a small GEMM)
//     Y[r] = sum_t w_t[r] * F[idx_t[r]] + side[r] . W_side + bias      (T = 1, w = 1: grouping;  T = 3: 3-NN interpolation)
// with `side` the <= 4 columns that exist per output row only (centred coordinates / the skip-link colours).  The second line is an
// element-wise kernel bound by the write of Y; it also takes the column statistics of Y for the batch norm (same partial layout as
// the GEMM kernels, so gspn_bn_finalize follows unchanged).  Backward: dY is rebuilt once from (dz, Y, coefficients) and written out;
// its transpose-gather onto the source points (the inverse lists' kernels) gives G = d(loss)/dF, and dW_feat = feat^T . G,
// d(feat) = G . W_feat^T are small GEMMs again; dW_side = side^T . dY is accumulated by the kernel that writes dY.
// (Same mathematics, a different order of fp32 additions than the grouped GEMM: results agree to rounding, not bit for bit.)
// ============================================================================================
struct PreaggSrc {
    const float* F;        // (source rows, cout)
    const int* idx;        // (rows, T) source rows; scene-local when per_scene_rows > 0
    const float* w;        // (rows, T) weights or NULL (= 1)
    int per_scene_rows;    // rows per scene of the OUTPUT (0: idx are global source rows)
    int per_scene_src;     // source rows per scene
    const float* side;     // (rows, side_ld), side_n <= 4 columns used
    int side_ld, side_n;
};
template <int T, int U>
__global__ __launch_bounds__(256) void preagg_fwd_kernel(long rows, int cout, PreaggSrc ps, const float* __restrict__ Ws, const float* __restrict__ bias,
                                                         float* __restrict__ Y, float* __restrict__ stats) {
    extern __shared__ float pa_sh[];                 // [2][rpi][cout]
    // U rows per thread in flight (the chain index -> gathered row -> store is all latency)
    const int cq = cout >> 2, rpi = 256 / cq;
    const int q = threadIdx.x % cq, rr = threadIdx.x / cq;
    float4 wsd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) wsd[k] = k < ps.side_n ? *reinterpret_cast<const float4*>(Ws + (size_t)k * cout + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 bq = bias ? *reinterpret_cast<const float4*>(bias + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    // workgroup b runs on XCD b % 8: give each XCD one contiguous eighth of the rows (with 8 scenes, one scene), so that the source rows
    // it gathers -- one scene's slice of F -- stay in that XCD's L2 instead of all of F in every L2
    const int nx = (gridDim.x % 8 == 0 && rows >= 8L * 256) ? 8 : 1;
    const long part = (rows + nx - 1) / nx;
    const long p0 = (long)(blockIdx.x % nx) * part, p1 = min(rows, p0 + part);
    const long stride = (long)(gridDim.x / nx) * rpi;
    for (long r0 = p0 + (long)(blockIdx.x / nx) * rpi + rr; r0 < p1; r0 += U * stride) {
        float4 f[U][T];
        float wt[U][T], sd[U][4];
        long rowq[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long r = r0 + u * stride;
            long rc = r < p1 ? r : p1 - 1;                       // clamped: unconditional loads, masked at the store
            rowq[u] = rc;
            const long base = ps.per_scene_rows > 0 ? (rc / ps.per_scene_rows) * (long)ps.per_scene_src : 0;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const long src = base + ps.idx[rc * T + t];
                f[u][t] = *reinterpret_cast<const float4*>(ps.F + (size_t)src * cout + 4 * q);
                wt[u][t] = ps.w ? ps.w[rc * T + t] : 1.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) sd[u][k] = k < ps.side_n ? ps.side[(size_t)rc * ps.side_ld + k] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r0 + u * stride >= p1) break;
            const long r = rowq[u];
            float4 y;
            if (T == 1 && !ps.w) y = f[u][0];
            else {
                y = make_float4(f[u][0].x * wt[u][0], f[u][0].y * wt[u][0], f[u][0].z * wt[u][0], f[u][0].w * wt[u][0]);
#pragma unroll
                for (int t = 1; t < T; ++t) { y.x += f[u][t].x * wt[u][t]; y.y += f[u][t].y * wt[u][t]; y.z += f[u][t].z * wt[u][t]; y.w += f[u][t].w * wt[u][t]; }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { y.x += sd[u][k] * wsd[k].x; y.y += sd[u][k] * wsd[k].y; y.z += sd[u][k] * wsd[k].z; y.w += sd[u][k] * wsd[k].w; }
            y.x += bq.x; y.y += bq.y; y.z += bq.z; y.w += bq.w;
            *reinterpret_cast<float4*>(Y + (size_t)r * cout + 4 * q) = y;
            s1.x += y.x; s1.y += y.y; s1.z += y.z; s1.w += y.w;
            s2.x += y.x * y.x; s2.y += y.y * y.y; s2.z += y.z * y.z; s2.w += y.w * y.w;
        }
    }
    if (!stats) return;
    *reinterpret_cast<float4*>(pa_sh + ((size_t)0 * rpi + rr) * cout + 4 * q) = s1;
    *reinterpret_cast<float4*>(pa_sh + ((size_t)1 * rpi + rr) * cout + 4 * q) = s2;
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * cout; j += 256) {
        const int h = j / cout, c = j - h * cout;
        float v = 0.f;
        for (int k = 0; k < rpi; ++k) v += pa_sh[((size_t)h * rpi + k) * cout + c];
        stats[(size_t)blockIdx.x * 2 * cout + j] = v;                // [block][2][cout]: sum, sum of squares
    }
}
// r04: the same computation on 16-lane DPP rows.  An output row is 64 * HPR channels = HPR DPP rows of 16 lanes x float4; a group of HPR
// DPP rows walks blocks of 16 consecutive output rows: lane q FETCHES everything row r0 + q needs (its T source rows' byte offsets, their
// weights, its side columns -- coalesced loads, issued one block ahead) and the 16 lanes hand those values round with row_newbcast DPP
// moves, so that the gathered float4 loads of U rows x T sources are all in flight before the first is used.  Round 3's kernel had every
// lane of a row fetch that row's index itself and two rows in flight: 18 % of its wave cycles issued an instruction (r03_sq_pmc_by_kernel).
// Y is bit-identical (same expression per element); the column sums are accumulated in a different (fixed) order.
template <int K>
__device__ __forceinline__ int pa_bcast(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + K, 0xF, 0xF, false); }
template <int I, int N, class F>
__device__ __forceinline__ void pa_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        pa_static_for<I + 1, N>(f);
    }
}
template <int T, int HPR>
__global__ __launch_bounds__(256) void preagg_fwd16_kernel(long rows, PreaggSrc ps, const float* __restrict__ Ws, const float* __restrict__ bias,
                                                           float* __restrict__ Y, float* __restrict__ stats) {
    constexpr int COUT = 64 * HPR;
    constexpr int NG = 16 / HPR;                     // row groups per workgroup
    constexpr int U = T == 1 ? 8 : 4;                // output rows whose gathered loads are issued together (16 / 8: no faster, measured)
    extern __shared__ float pa_sh[];                 // [2][NG][COUT]
    const int lane16 = threadIdx.x & 15, drow = threadIdx.x >> 4;
    const int grp = drow / HPR, half = drow % HPR;
    const int q = half * 16 + lane16;                // this lane's float4 of a row
    float4 wsd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) wsd[k] = k < ps.side_n ? *reinterpret_cast<const float4*>(Ws + (size_t)k * COUT + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 bq = bias ? *reinterpret_cast<const float4*>(bias + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    // XCD x = blockIdx % 8 takes the x-th eighth of the rows (one scene's slice of F stays in that XCD's L2), in blocks of 16 * NG rows
    const int nx = (gridDim.x % 8 == 0 && rows >= 8L * 256) ? 8 : 1;
    const long part = ((rows + nx - 1) / nx + 15) / 16 * 16;
    const long p0 = (long)(blockIdx.x % nx) * part, p1 = min(rows, p0 + part);
    const long stride = (long)(gridDim.x / nx) * (16 * NG);
    const char* Fb = reinterpret_cast<const char*>(ps.F) + 16 * q;
    int ob[T], obn[T];
    float wt[T], wtn[T], sd[4], sdn[4];
    auto fetch = [&](long r0, int (&o)[T], float (&w)[T], float (&sv)[4]) {      // what row r0 + lane16 needs (clamped: unconditional loads)
        long rc = r0 + lane16;
        rc = rc < p1 ? rc : p1 - 1;
        const long base = ps.per_scene_rows > 0 ? (rc / ps.per_scene_rows) * (long)ps.per_scene_src : 0;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            o[t] = (int)(base + ps.idx[rc * T + t]) * (COUT * 4);
            w[t] = ps.w ? ps.w[rc * T + t] : 1.f;
        }
        if (ps.side_n > 0 && ps.side_ld == 4) {
            const float4 v = *reinterpret_cast<const float4*>(ps.side + (size_t)rc * 4);
            sv[0] = v.x; sv[1] = v.y; sv[2] = v.z; sv[3] = v.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) sv[k] = k < ps.side_n ? ps.side[(size_t)rc * ps.side_ld + k] : 0.f;
        }
    };
    long r0 = p0 + (long)(blockIdx.x / nx) * (16 * NG) + 16 * grp;
    if (r0 < p1) fetch(r0, ob, wt, sd);
    for (; r0 < p1; r0 += stride) {
        const long rn = r0 + stride;
        if (rn < p1) fetch(rn, obn, wtn, sdn);                              // the next block's indices: in flight under this block's gathers
        const int cnt = (int)min(16L, p1 - r0);
        pa_static_for<0, 16 / U>([&](auto gi) {
            constexpr int K0 = decltype(gi)::value * U;
            if (K0 < cnt) {
                float4 f[U][T];
                float wv[U][T], sv[U][4];
                pa_static_for<0, U>([&](auto ui) {
                    constexpr int u = decltype(ui)::value, K = K0 + u;
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        const int o = pa_bcast<K>(ob[t]);                   // (rows past the end re-read the clamped last row; masked at the store)
                        f[u][t] = *reinterpret_cast<const float4*>(Fb + (size_t)(unsigned)o);
                        wv[u][t] = __int_as_float(pa_bcast<K>(__float_as_int(wt[t])));
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) sv[u][k] = __int_as_float(pa_bcast<K>(__float_as_int(sd[k])));
                });
                pa_static_for<0, U>([&](auto ui) {
                    constexpr int u = decltype(ui)::value, K = K0 + u;
                    if (K < cnt) {
                        float4 y;
                        if (T == 1 && !ps.w) y = f[u][0];
                        else {
                            y = make_float4(f[u][0].x * wv[u][0], f[u][0].y * wv[u][0], f[u][0].z * wv[u][0], f[u][0].w * wv[u][0]);
#pragma unroll
                            for (int t = 1; t < T; ++t) { y.x += f[u][t].x * wv[u][t]; y.y += f[u][t].y * wv[u][t]; y.z += f[u][t].z * wv[u][t]; y.w += f[u][t].w * wv[u][t]; }
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) { y.x += sv[u][k] * wsd[k].x; y.y += sv[u][k] * wsd[k].y; y.z += sv[u][k] * wsd[k].z; y.w += sv[u][k] * wsd[k].w; }
                        y.x += bq.x; y.y += bq.y; y.z += bq.z; y.w += bq.w;
                        *reinterpret_cast<float4*>(Y + (size_t)(r0 + K) * COUT + 4 * q) = y;
                        s1.x += y.x; s1.y += y.y; s1.z += y.z; s1.w += y.w;
                        s2.x += y.x * y.x; s2.y += y.y * y.y; s2.z += y.z * y.z; s2.w += y.w * y.w;
                    }
                });
            }
        });
#pragma unroll
        for (int t = 0; t < T; ++t) { ob[t] = obn[t]; wt[t] = wtn[t]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) sd[k] = sdn[k];
    }
    if (!stats) return;
    *reinterpret_cast<float4*>(pa_sh + ((size_t)0 * NG + grp) * COUT + 4 * q) = s1;
    *reinterpret_cast<float4*>(pa_sh + ((size_t)1 * NG + grp) * COUT + 4 * q) = s2;
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * COUT; j += 256) {
        const int h = j / COUT, c = j - h * COUT;
        float v = 0.f;
        for (int k = 0; k < NG; ++k) v += pa_sh[((size_t)h * NG + k) * COUT + c];
        stats[(size_t)blockIdx.x * 2 * COUT + j] = v;                // [block][2][cout]: sum, sum of squares
    }
}
static bool preagg_shape_ok(int cout) { return cout >= 4 && cout <= 1024 && (cout & 3) == 0 && ((cout >> 2) & ((cout >> 2) - 1)) == 0; }
extern "C" int gspn_preagg_ok(int cout) { return preagg_shape_ok(cout) ? 1 : 0; }
// workgroups (= partial statistics rows) of gspn_preagg_fwd: every thread gets about four rows, 2048 workgroups at most
#ifndef PREAGG_FWD_BLOCKS
#define PREAGG_FWD_BLOCKS 2048
#endif
static unsigned preagg_fwd_blocks(long rows, int cout) {
    const long rpi = 256 / (cout >> 2);
    long nb = (rows + 4 * rpi - 1) / (4 * rpi);
    static const int cap = env_int("GSPN_PREAGG_BLOCKS", PREAGG_FWD_BLOCKS);
    if (nb > cap) nb = cap;
    if (nb >= 64) nb = nb / 8 * 8;                       // a multiple of the 8 XCDs: the kernel then gives each XCD its own eighth of the rows
    return (unsigned)(nb < 1 ? 1 : nb);
}
extern "C" long gspn_preagg_fwd_parts(long rows, int cout) { return (rows > 0 && preagg_shape_ok(cout)) ? (long)preagg_fwd_blocks(rows, cout) : GSPN_ERR_ARG; }
extern "C" int gspn_preagg_fwd(long rows, int cout, int T, const float* F, const int* idx, const float* w, int per_scene_rows, int per_scene_src,
                               const float* side, int side_ld, int side_n, const float* Wside, const float* bias, float* Y, float* stats, void* stream) {
    if (rows <= 0 || !F || !idx || !Y || (T != 1 && T != 3) || side_n < 0 || side_n > 4 || (side_n > 0 && (!side || !Wside || side_ld < side_n))) return GSPN_ERR_ARG;
    if (!preagg_shape_ok(cout) || rows >= (1L << 31)) return GSPN_ERR_UNSUPPORTED;
    if (((uintptr_t)F % 16) || ((uintptr_t)Y % 16) || (Wside && ((uintptr_t)Wside % 16)) || (bias && ((uintptr_t)bias % 16))) return GSPN_ERR_ARG;
    const PreaggSrc ps{F, idx, w, per_scene_rows, per_scene_src, side, side_ld, side_n};
    const unsigned nb = preagg_fwd_blocks(rows, cout);   // = the number of partial rows: gspn_bn_finalize_parts(..., gspn_preagg_fwd_parts(rows, cout), ...)
    const size_t sh = sizeof(float) * 2 * 256 * 4;       // 2 * rpi * cout floats, rpi * cout = 1024
    static const int rows16 = env_int("GSPN_PREAGG_FWD16", 1);          // (A/B hook: 0 = round 3's kernel)
    const long fsrc_bytes = (long)(per_scene_rows > 0 ? ((rows + per_scene_rows - 1) / per_scene_rows) * (long)per_scene_src : 0) * cout * 4;
    if (rows16 && (cout == 64 || cout == 128 || cout == 256) && fsrc_bytes < (1L << 31) && (side_n == 0 || side_ld != 4 || ((uintptr_t)side % 16) == 0)
        && (per_scene_rows > 0 || true)) {
        // (global idx, per_scene_rows == 0: the kernel forms 32-bit byte offsets into F.  per_scene_src then carries the TOTAL number of source rows
        //  (r05, ADVICE r04: with m*nsample < n the source table is larger than `rows`); 0 = not given, the caller's F is taken to have at
        //  most `rows` source rows, as round 4 assumed.  Beyond 2^31 bytes the general kernel below runs.)
        const long src_rows = per_scene_src > 0 ? (long)per_scene_src : rows;
        if (per_scene_rows > 0 || (src_rows * (long)cout * 4 < (1L << 31) && rows * (long)cout * 4 < (1L << 31))) {
#define PA16_GO(T_, H_) hipLaunchKernelGGL((preagg_fwd16_kernel<T_, H_>), dim3(nb), dim3(256), sh, (hipStream_t)stream, rows, ps, Wside, bias, Y, stats)
            if (T == 1) { if (cout == 64) PA16_GO(1, 1); else if (cout == 128) PA16_GO(1, 2); else PA16_GO(1, 4); }
            else { if (cout == 64) PA16_GO(3, 1); else if (cout == 128) PA16_GO(3, 2); else PA16_GO(3, 4); }
#undef PA16_GO
            return gspn_launch_status();
        }
    }
    static const int uu = env_int("GSPN_PREAGG_U", 2);
#define PA_GO(T_, U_) hipLaunchKernelGGL((preagg_fwd_kernel<T_, U_>), dim3(nb), dim3(256), sh, (hipStream_t)stream, rows, cout, ps, Wside, bias, Y, stats)
    if (T == 1) { if (uu == 1) PA_GO(1, 1); else if (uu == 2) PA_GO(1, 2); else PA_GO(1, 4); }
    else { if (uu == 1) PA_GO(3, 1); else if (uu == 2) PA_GO(3, 2); else PA_GO(3, 4); }
#undef PA_GO
    return gspn_launch_status();
}
// dY = cA * relu'(y*scale+shift) * dz + cB * y + cC, written out (rows, cout); per workgroup the partial side^T . dY (side_n x cout) into
// part[workgroup][2][side_n*cout] (first half; the layout of the dW reduction's slots)
#ifndef PREAGG_BWD_BLOCKS
#define PREAGG_BWD_BLOCKS 1024
#endif
__global__ __launch_bounds__(256) void preagg_bwd_dy_kernel(long rows, int cout, gspn_dy_args a, const float* __restrict__ side, int side_ld, int side_n,
                                                            float* __restrict__ dY, float* __restrict__ part) {
    extern __shared__ float pa_sh[];                 // [rpi][side_n][cout]
    const int cq = cout >> 2, rpi = 256 / cq;
    const int q = threadIdx.x % cq, rr = threadIdx.x / cq;
    const float4 sc = *reinterpret_cast<const float4*>(a.scale + 4 * q), sh = *reinterpret_cast<const float4*>(a.shift + 4 * q);
    const float4 cA = *reinterpret_cast<const float4*>(a.cA + 4 * q), cB = *reinterpret_cast<const float4*>(a.cB + 4 * q);
    const float4 cC = *reinterpret_cast<const float4*>(a.cC + 4 * q);
    float4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long r = (long)blockIdx.x * rpi + rr; r < rows; r += (long)gridDim.x * rpi) {
        const float4 y = *reinterpret_cast<const float4*>(a.Y + (size_t)r * a.ldy + 4 * q);
        const float4 z = *reinterpret_cast<const float4*>(a.dZ + (size_t)r * a.ldz + 4 * q);
        float4 d;
        d.x = cA.x * (y.x * sc.x + sh.x > 0.f ? z.x : 0.f) + cB.x * y.x + cC.x;
        d.y = cA.y * (y.y * sc.y + sh.y > 0.f ? z.y : 0.f) + cB.y * y.y + cC.y;
        d.z = cA.z * (y.z * sc.z + sh.z > 0.f ? z.z : 0.f) + cB.z * y.z + cC.z;
        d.w = cA.w * (y.w * sc.w + sh.w > 0.f ? z.w : 0.f) + cB.w * y.w + cC.w;
        *reinterpret_cast<float4*>(dY + (size_t)r * cout + 4 * q) = d;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < side_n) {
                const float sv = side[(size_t)r * side_ld + k];
                acc[k].x += sv * d.x; acc[k].y += sv * d.y; acc[k].z += sv * d.z; acc[k].w += sv * d.w;
            }
    }
    if (side_n == 0 || !part) return;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < side_n) *reinterpret_cast<float4*>(pa_sh + ((size_t)rr * side_n + k) * cout + 4 * q) = acc[k];
    __syncthreads();
    const int tot = side_n * cout;
    for (int j = threadIdx.x; j < tot; j += 256) {
        float v = 0.f;
        for (int k = 0; k < rpi; ++k) v += pa_sh[(size_t)k * tot + j];
        part[(size_t)blockIdx.x * 2 * tot + j] = v;
    }
}
extern "C" long gspn_preagg_part_floats(int cout, int side_n) { return (long)PREAGG_BWD_BLOCKS * 2 * side_n * cout + 4; }
// dY (rows, cout) from (a.dZ, a.Y, a.scale/shift, a.cA/cB/cC); dWside (side_n, cout) = side^T . dY (deterministic two-level sum).
// dWside == NULL leaves the second level to the caller: *nslots_out partial tiles at part (gspn_mlp_bwd_data_dw2 can carry it).
extern "C" int gspn_preagg_bwd_dy(long rows, int cout, const gspn_dy_args* a, const float* side, int side_ld, int side_n, float* dY, float* part,
                                  float* dWside, int* nslots_out, void* stream) {
    if (rows <= 0 || !a || !a->Y || !a->dZ || !a->scale || !a->shift || !a->cA || !a->cB || !a->cC || !dY || side_n < 0 || side_n > 4) return GSPN_ERR_ARG;
    if (side_n > 0 && (!side || !part || (!dWside && !nslots_out) || side_ld < side_n)) return GSPN_ERR_ARG;
    if (!preagg_shape_ok(cout) || rows >= (1L << 31) || (a->ldy & 3) || (a->ldz & 3)) return GSPN_ERR_UNSUPPORTED;
    if (((uintptr_t)a->Y % 16) || ((uintptr_t)a->dZ % 16) || ((uintptr_t)dY % 16)) return GSPN_ERR_ARG;
    const int rpi = 256 / (cout >> 2);
    long nb = (rows + rpi - 1) / rpi;
    if (nb > PREAGG_BWD_BLOCKS) nb = PREAGG_BWD_BLOCKS;
    const size_t sh = sizeof(float) * 4 * 1024;
    hipLaunchKernelGGL(preagg_bwd_dy_kernel, dim3((unsigned)nb), dim3(256), sh, (hipStream_t)stream, rows, cout, *a, side, side_ld, side_n, dY, part);
    if (nslots_out) *nslots_out = (int)nb;
    if (side_n > 0 && dWside) {
        DwJob j = dw_job(rows, side_n, cout, nb, part, nullptr, nullptr, nullptr, nullptr, 0.f, 0, 0, dWside);
        j.plain = 1;
        hipLaunchKernelGGL(wgrad_dw_kernel, dim3((unsigned)dw_blocks((long)side_n * cout, nb, 1024)), dim3(1024), 0, (hipStream_t)stream, j);
    }
    return gspn_launch_status();
}

extern "C" int gspn_mlp_bwd_dw(long rows, int cin, int cout, const gspn_dy_args* a, const float* X, int ldx, const float* var, const float* gamma,
                               float eps, int use_bn, int is_training, const float* work, float* dW, void* stream) {
    if (rows <= 0 || cin <= 0 || cout <= 0 || ldx < cin || !a || !a->Y || !work || !dW) return GSPN_ERR_ARG;
    if (use_bn && !var) return GSPN_ERR_ARG;
    if (cin > MAXCH || cout > MAXCH || rows >= (1L << 31)) return GSPN_ERR_UNSUPPORTED;
    bool use_stream;
    const WgradPlan p = wgrad_choose(rows, cin, cout, a, X, ldx, &use_stream);
    const char* wb = reinterpret_cast<const char*>(work);
    const double* red = reinterpret_cast<const double*>(wb);
    const float* g3 = reinterpret_cast<const float*>(wb + ws_off_g3(cout));
    const float* PP = reinterpret_cast<const float*>(wb + ws_off_pp(p.nch, cin, cout));
    const DwJob j = dw_job(rows, cin, cout, p.nslots, PP, red, g3, var, gamma, eps, use_bn, is_training, dW);
    hipLaunchKernelGGL(wgrad_dw_kernel, dim3((unsigned)dw_blocks((long)j.cin * j.cout, j.nslots, 1024)), dim3(1024), 0, (hipStream_t)stream, j);
    return gspn_launch_status();
}

// ============================================================================================
// Backward pass B:  dX(rows, cin) = dY(rows, cout) . W^T      (M = rows, K = cout, N = cin)
// A = dY rebuilt on the fly (dY = cA*dyh + cB*y + cC) while staging; B[k][n] = W[n][k] staged transposed.
// ============================================================================================
// DW: the first dwj.nblk workgroups (of row blockIdx.y == 0) are not part of the GEMM: they reduce the layer's partial dW tiles (the
// last kernel of pass A, which nothing in pass B depends on) while the rest of the grid computes dX -- one launch, no tail, instead of
// a 15 us kernel of its own per layer.
// RsumArgs (Yp != NULL): this launch's dX is the dz of the PREVIOUS layer (pre-BN output Yp, batch statistics mean/var, forward
// scale/shift); the epilogue also takes that layer's BN reductions sum(dyh), sum(dyh*xhat) from the finished tiles -- per-workgroup
// partials part[workgroup][2][cin], summed by gspn_mlp_bwd_coef -- so its pass A can run with final coefficients (one GEMM).
struct RsumArgs { const float* Yp; int ldyp; const float* scale; const float* shift; const float* mean; const float* var; float eps; float* part; };
#ifndef GSPN_BWD_TK
#define GSPN_BWD_TK 32
#endif
#ifndef GSPN_BWD_WPE32
#define GSPN_BWD_WPE32 4         // measured on the bench step (pass B of SA1's two 32-column layers): 3 -> 4 waves per SIMD: 147 -> 130 us
#endif
#ifndef GSPN_BWD_WPE64
#define GSPN_BWD_WPE64 3
#endif
template <int BN, bool VEC, bool POOLED, bool DW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BN <= 32 ? GSPN_BWD_WPE32 : (BN <= 64 ? GSPN_BWD_WPE64 : 2))))
void mlp_bwd_data_kernel(long rows, int cin, int cout, gspn_dy_args a, const float* __restrict__ W,
                                                           float* __restrict__ dX, int ldx, int col0, DwJob dwj, RsumArgs rs) {
    // `cin` is the END of the column range [col0, cin) of dX this launch produces (gspn_mlp_bwd_data_cols)
    constexpr int NT = BN / 32;
    constexpr int LDBT = BN + 1;                 // B written transposed: odd pitch
    constexpr int TKB = GSPN_BWD_TK;             // K chunk of THIS kernel: the chunk sequence (fetch, barrier, transform, barrier, MFMA) bounds it
    constexpr int TPR = TKB / 4;                 // threads per row of a chunk (one float4 each)
    constexpr int RPP = 256 / TPR;               // rows per pass
    constexpr int NPASS = TM / RPP;
    constexpr int NB = (TKB / 4 * BN) / 256;     // float4 (along k) per thread per chunk
    __shared__ __attribute__((aligned(16))) float sA[TKB * LDT];
    __shared__ __attribute__((aligned(16))) float sB[TKB * LDBT];
    extern __shared__ __attribute__((aligned(16))) float s_chan[];         // [5][cpad]: forward scale / shift, cA, cB, cC of the cout channels
    const int cpad = (cout + 3) / 4 * 4 + 4;
    float* sSc = s_chan;
    float* sSh = s_chan + cpad;
    float* sCA = s_chan + 2 * cpad;
    float* sCB = s_chan + 3 * cpad;
    float* sCC = s_chan + 4 * cpad;
    unsigned bx = blockIdx.x, gx = gridDim.x, by = blockIdx.y;
    if constexpr (DW) {
        // one-dimensional grid: [dwj.nblk reduction workgroups][column block 0: dwj.rowgrid GEMM workgroups][column block 1: ...]
        // (a second grid dimension would launch the reduction workgroups once per column block, all but the first to exit at once:
        //  67 584 empty workgroups for the 384-column layer of the FP stack, 30 us of dispatch)
        if (bx < (unsigned)(dwj.nblk + dwj.nblk2)) {
            if (bx < (unsigned)dwj.nblk) wgrad_dw_block<256>(dwj, bx, reinterpret_cast<double*>(sA));
            else {
                DwJob j2 = dwj;
                j2.PP = dwj.PP2; j2.dW = dwj.dW2; j2.cin = dwj.cin2; j2.nslots = dwj.nslots2; j2.plain = 1; j2.use_bn = 0; j2.gq = 0;
                wgrad_dw_block<256>(j2, bx - (unsigned)dwj.nblk, reinterpret_cast<double*>(sA));
            }
            return;
        }
        bx -= (unsigned)(dwj.nblk + dwj.nblk2);
        gx = (unsigned)dwj.rowgrid;
        by = bx / gx;
        bx -= by * gx;
    }
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n0 = col0 + by * BN;
    const long ntiles = (rows + TM - 1) / TM;
    const int nchunks = (cout + TKB - 1) / TKB;
    stage_chan(sSc, a.scale, cout, 1.f, cpad);
    stage_chan(sSh, a.shift, cout, 0.f, cpad);
    stage_chan(sCA, a.cA, cout, 0.f, cpad);         // channels >= cout contribute dY = 0
    stage_chan(sCB, a.cB, cout, 0.f, cpad);
    stage_chan(sCC, a.cC, cout, 0.f, cpad);
    __syncthreads();
    const int kq = (t % TPR) * 4;
    const int arow = t / TPR;
    float4 ry[NPASS], rb[NB];
    DzRaw rz[NPASS];
    long f_tile = 0;
    int f_c = 0;
    auto fetch = [&](long tile, int c) {              // raw, unconditional loads
        f_tile = tile; f_c = c;
        const long m0 = tile * TM;
        const int k = c * TKB + kq;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const long row = m0 + arow + RPP * i;
            const long rc = row < rows ? row : rows - 1;
            ry[i] = load4_raw<VEC>(a.Y, rc, a.ldy, k, cout);
            rz[i] = dz4_raw<VEC, POOLED>(a, rc, k, cout);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = t + 256 * i;
            const int j = f / TPR, kk4 = (f % TPR) * 4;    // W row n0+j, columns c*TKB + kk4 ..
            const int n = n0 + j;
            rb[i] = load4_raw<VEC>(W, n < cin ? n : cin - 1, cout, c * TKB + kk4, cout);
        }
    };
    auto commit = [&]() {
        const long m0 = f_tile * TM;
        const int c = f_c;
        const int k = c * TKB + kq;
        const float4 q_sc = lds4(sSc, k, cpad), q_sh = lds4(sSh, k, cpad), q_a = lds4(sCA, k, cpad), q_b = lds4(sCB, k, cpad), q_c = lds4(sCC, k, cpad);
        const float sc[4] = {q_sc.x, q_sc.y, q_sc.z, q_sc.w}, sh[4] = {q_sh.x, q_sh.y, q_sh.z, q_sh.w};
        const float cA[4] = {q_a.x, q_a.y, q_a.z, q_a.w}, cB[4] = {q_b.x, q_b.y, q_b.z, q_b.w}, cC[4] = {q_c.x, q_c.y, q_c.z, q_c.w};
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const long row = m0 + arow + RPP * i;
            const float4 dzv = dz4_resolve<POOLED>(a, rz[i], row < rows ? row : rows - 1);
            const float yv[4] = {ry[i].x, ry[i].y, ry[i].z, ry[i].w};
            const float zv[4] = {dzv.x, dzv.y, dzv.z, dzv.w};
            float* d = sA + kq * LDT + arow + RPP * i;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dyh = (yv[j] * sc[j] + sh[j]) > 0.f ? zv[j] : 0.f;
                d[j * LDT] = cA[j] * dyh + cB[j] * yv[j] + cC[j];       // columns >= cout have cA=cB=cC=0
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = t + 256 * i;
            const int j = f / TPR, kk4 = (f % TPR) * 4;
            const float4 w = mask4(rb[i], c * TKB + kk4, cout, (n0 + j) < cin);
            float* d = sB + kk4 * LDBT + j;
            d[0 * LDBT] = w.x; d[1 * LDBT] = w.y; d[2 * LDBT] = w.z; d[3 * LDBT] = w.w;
        }
    };
    f32x16 acc[NT];
    // previous layer's reductions (RsumArgs): per-lane constants of this lane's columns and running sums over the workgroup's tiles
    const bool rsum = rs.Yp != nullptr;
    float p_sc[NT], p_sh[NT], p_rs[NT], p_mr[NT], r0s[NT], r1s[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        r0s[nt] = r1s[nt] = 0.f;
        p_sc[nt] = p_sh[nt] = p_rs[nt] = p_mr[nt] = 0.f;
        if (rsum) {
            const int col = min(n0 + nt * 32 + (lane & 31), cin - 1);
            p_sc[nt] = rs.scale[col];
            p_sh[nt] = rs.shift[col];
            p_rs[nt] = (float)(1.0 / sqrt((double)rs.var[col] + (double)rs.eps));
            p_mr[nt] = -rs.mean[col] * p_rs[nt];
        }
    }
    const long my_tiles = bx < ntiles ? (ntiles - bx + gx - 1) / gx : 0;
    const long nsteps = my_tiles * nchunks;
    for (long sidx = 0; sidx <= nsteps; ++sidx) {
        if (sidx < nsteps) fetch(bx + (sidx / nchunks) * gx, (int)(sidx % nchunks));
        if (sidx > 0) {
            const long ps = sidx - 1;
            const long tile = bx + (ps / nchunks) * gx;
            const int c = (int)(ps % nchunks);
            if (c == 0) {
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            }
            const int kmax = min(TKB, (cout - c * TKB + 1) & ~1);
            for (int kk = 0; kk < kmax; kk += 2) {
                const float av = sA[(kk + (lane >> 5)) * LDT + wave * 32 + (lane & 31)];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float bvv = sB[(kk + (lane >> 5)) * LDBT + nt * 32 + (lane & 31)];
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bvv, acc[nt], 0, 0, 0);
                }
            }
            if (c == nchunks - 1) {
                const long m0 = tile * TM;
                const bool full = m0 + TM <= rows;
                if (rsum) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int col = n0 + nt * 32 + (lane & 31);
                        if (col < cin) {
                            const float* yp = rs.Yp + (m0 + wave * 32 + 4 * (lane >> 5)) * rs.ldyp + col;
                            float yv[16];
#pragma unroll
                            for (int r = 0; r < 16; ++r) {         // clamped, unconditional loads: all sixteen in flight
                                const long rr = (r & 3) + 8 * (r >> 2);
                                const long rowc = min(m0 + wave * 32 + 4 * (lane >> 5) + rr, rows - 1) - (m0 + wave * 32 + 4 * (lane >> 5));
                                yv[r] = yp[rowc * rs.ldyp];
                            }
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const bool live = full || (m0 + wave * 32 + c_row(r, lane)) < rows;
                                const float dyh = (live && relu_open(yv[r], p_sc[nt], p_sh[nt])) ? acc[nt][r] : 0.f;
                                r0s[nt] += dyh;
                                r1s[nt] = __builtin_fmaf(dyh, __builtin_fmaf(yv[r], p_rs[nt], p_mr[nt]), r1s[nt]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = n0 + nt * 32 + (lane & 31);
                    if (col < cin) {
                        float* xp = dX + (m0 + wave * 32 + 4 * (lane >> 5)) * ldx + col;
                        if (full) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) xp[(long)((r & 3) + 8 * (r >> 2)) * ldx] = acc[nt][r];
                        } else {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const long row = m0 + wave * 32 + c_row(r, lane);
                                if (row < rows) dX[row * ldx + col] = acc[nt][r];
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (sidx < nsteps) commit();
        __syncthreads();
    }
    if (rsum) {
        // lanes l and l+32 hold the same column; then the four waves through LDS (sA is free now); one partial row per workgroup
        float* sR = sA;                                  // [4 waves][2][BN]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            r0s[nt] += __shfl_xor(r0s[nt], 32, 64);
            r1s[nt] += __shfl_xor(r1s[nt], 32, 64);
            if (lane < 32) {
                sR[(wave * 2 + 0) * BN + nt * 32 + lane] = r0s[nt];
                sR[(wave * 2 + 1) * BN + nt * 32 + lane] = r1s[nt];
            }
        }
        __syncthreads();
        float* pr = rs.part + (size_t)bx * 2 * cin;
        for (int j = t; j < BN; j += 256) {
            const int col = n0 + j;
            if (col < cin) {
                float v0 = 0.f, v1 = 0.f;
                for (int w = 0; w < 4; ++w) { v0 += sR[(w * 2 + 0) * BN + j]; v1 += sR[(w * 2 + 1) * BN + j]; }
                pr[col] = v0;
                pr[cin + col] = v1;
            }
        }
    }
}
// ============================================================================================
// Pass B, lean form (r03).  Same decomposition as mlp_bwd_data_kernel -- 128-row tiles, K chunks of 32, dY rebuilt while staging,
// W^T staged transposed, optional dW-reduction ride and BN-reduction epilogue -- for the shapes the network actually runs: 16-byte
// aligned pitches, rows a multiple of 128, cout a multiple of 32, a column range that is a whole number of BN-wide blocks, pool groups
// of 32 rows or of a power of two >= 128, every offset below 2^32 bytes.  What it drops is vector arithmetic: the counters say these
// GEMM kernels issue 12-32 VALU instructions per MFMA (profiles/r03_sq_insts_by_kernel.txt) and fp32 MFMAs do not hide them, and in
// the general kernel most of them are 64-bit index arithmetic, clamps and masks.  Here
//   * a thread's quads sit at CONSTANT byte offsets inside a tile and a chunk (three registers in all); tile and chunk position go into
//     uniform bases (scalar arithmetic): the loads of a step are `global_load v, v_off, s[base]` without any vector address work;
//   * no row / column guards anywhere (full tiles only); the negated shift is staged once, the ReLU mask is one multiply + one compare;
//   * the MFMA loop is unrolled: every LDS address is a lane base plus an immediate;
//   * the epilogue's 16 loads of the previous layer's output and 16 stores per 32x32 tile take their row from a scalar base as well.
// dY = fma(cA, dyh, fma(cB, y, cC)) -- the form pass A's one-GEMM kernel uses (the general kernel rounds the three terms separately).
// ============================================================================================
template <int NT, int PK, bool RSUM, bool DW>      // PK: 0 dense dz, 1 pool groups of 32 rows, 2 pool groups of 2^k >= 128 rows (one group per tile)
__global__ __launch_bounds__(256) void bwd_lean_kernel(int rows, int cout, gspn_dy_args a, const float* __restrict__ W, float* __restrict__ dX, int ldx,
                                                       int col0, DwJob dwj, RsumArgs rs, int rowgrid, int pool_sh, int rs_cin) {
    constexpr int BN = 32 * NT;
    constexpr int LDA = 129, LDB = BN + 1;
    __shared__ __attribute__((aligned(16))) float sA[32 * LDA];
    __shared__ __attribute__((aligned(16))) float sB[32 * LDB];
    extern __shared__ __attribute__((aligned(16))) float s_chan[];         // [5][cpad]: forward scale, MINUS shift, cA, cB, cC of the cout channels
    const int cpad = (cout + 3) / 4 * 4 + 4;
    unsigned bx = blockIdx.x;
    if constexpr (DW) {
        if (bx < (unsigned)(dwj.nblk + dwj.nblk2)) {
            if (bx < (unsigned)dwj.nblk) wgrad_dw_block<256>(dwj, bx, reinterpret_cast<double*>(sA));
            else {
                DwJob j2 = dwj;
                j2.PP = dwj.PP2; j2.dW = dwj.dW2; j2.cin = dwj.cin2; j2.nslots = dwj.nslots2; j2.plain = 1; j2.use_bn = 0; j2.gq = 0;
                wgrad_dw_block<256>(j2, bx - (unsigned)dwj.nblk, reinterpret_cast<double*>(sA));
            }
            return;
        }
        bx -= (unsigned)(dwj.nblk + dwj.nblk2);
    }
    const int by = (int)(bx / (unsigned)rowgrid);
    bx -= (unsigned)by * (unsigned)rowgrid;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int n0 = col0 + by * BN;
    const int ntiles = rows >> 7;
    const int nchunks = cout >> 5;
    for (int i = t; i < cpad; i += 256) {
        const bool in = i < cout;
        s_chan[i] = in ? a.scale[i] : 1.f;
        s_chan[cpad + i] = in ? -a.shift[i] : 0.f;
        s_chan[2 * cpad + i] = in ? a.cA[i] : 0.f;
        s_chan[3 * cpad + i] = in ? a.cB[i] : 0.f;
        s_chan[4 * cpad + i] = in ? a.cC[i] : 0.f;
    }
    const int kq = (t & 7) * 4, arow = t >> 3;                  // this thread's quad: rows arow + 32 i, k = kq .. kq+3 of the chunk
    const unsigned oy = (unsigned)(arow * a.ldy + kq) * 4u;
    constexpr bool POOLED = PK != 0;
    constexpr int NZ = PK == 2 ? 1 : 4;                         // (PK 2: the tile's rows share ONE (arg, dPool) quad per k)
    const unsigned oz = POOLED ? (unsigned)kq * 4u : (unsigned)(arow * a.ldz + kq) * 4u;
    const unsigned ow = (unsigned)(arow * cout + kq) * 4u;      // W row n0 + arow + 32 i
    float4 ry[4], rz[NZ], rb[NT];
    int4 rarg[PK == 1 ? 4 : 1];
    auto fetch = [&](int tile, int c) {
        const int m0 = tile << 7;
        const char* yb = reinterpret_cast<const char*>(a.Y + (size_t)m0 * a.ldy + c * 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) ry[i] = *reinterpret_cast<const float4*>(yb + (size_t)(32 * i) * a.ldy * 4 + oy);
        if constexpr (PK == 1) {                                 // groups of 32 rows: rows arow + 32 i of the tile belong to group (m0 >> 5) + i
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const size_t g = (size_t)((m0 >> 5) + i) * cout + c * 32;
                rarg[i] = *reinterpret_cast<const int4*>(reinterpret_cast<const char*>(a.pool_arg + g) + oz);
                rz[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.dPool + g) + oz);
            }
        } else if constexpr (PK == 2) {                          // groups of >= 128 rows: the whole tile lies in one group
            const size_t g = (size_t)(m0 >> pool_sh) * cout + c * 32;
            rarg[0] = *reinterpret_cast<const int4*>(reinterpret_cast<const char*>(a.pool_arg + g) + oz);
            rz[0] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.dPool + g) + oz);
        } else {
            const char* zb = reinterpret_cast<const char*>(a.dZ + (size_t)m0 * a.ldz + c * 32);
#pragma unroll
            for (int i = 0; i < 4; ++i) rz[i] = *reinterpret_cast<const float4*>(zb + (size_t)(32 * i) * a.ldz * 4 + oz);
        }
        const char* wb = reinterpret_cast<const char*>(W + (size_t)n0 * cout + c * 32);
#pragma unroll
        for (int i = 0; i < NT; ++i) rb[i] = *reinterpret_cast<const float4*>(wb + (size_t)(32 * i) * cout * 4 + ow);
    };
    auto commit = [&](int tile, int c) {
        const int k = c * 32 + kq;
        const float4 q_sc = *reinterpret_cast<const float4*>(s_chan + k), q_ns = *reinterpret_cast<const float4*>(s_chan + cpad + k);
        const float4 q_a = *reinterpret_cast<const float4*>(s_chan + 2 * cpad + k), q_b = *reinterpret_cast<const float4*>(s_chan + 3 * cpad + k);
        const float4 q_c = *reinterpret_cast<const float4*>(s_chan + 4 * cpad + k);
        const float sc[4] = {q_sc.x, q_sc.y, q_sc.z, q_sc.w}, ns[4] = {q_ns.x, q_ns.y, q_ns.z, q_ns.w};
        const float cA[4] = {q_a.x, q_a.y, q_a.z, q_a.w}, cB[4] = {q_b.x, q_b.y, q_b.z, q_b.w}, cC[4] = {q_c.x, q_c.y, q_c.z, q_c.w};
        const int off0 = PK == 1 ? arow : (PK == 2 ? (((tile << 7) & ((1 << pool_sh) - 1)) + arow) : 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float yv[4] = {ry[i].x, ry[i].y, ry[i].z, ry[i].w};
            float zv[4];
            if constexpr (POOLED) {
                const int gi = PK == 1 ? i : 0;
                const int off = PK == 1 ? off0 : off0 + 32 * i;
                const int av[4] = {rarg[gi].x, rarg[gi].y, rarg[gi].z, rarg[gi].w};
                const float dv[4] = {rz[gi].x, rz[gi].y, rz[gi].z, rz[gi].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) zv[j] = av[j] == off ? dv[j] : 0.f;
            } else {
                zv[0] = rz[i].x; zv[1] = rz[i].y; zv[2] = rz[i].z; zv[3] = rz[i].w;
            }
            float* d = sA + kq * LDA + arow + 32 * i;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dyh = yv[j] * sc[j] > ns[j] ? zv[j] : 0.f;                                   // relu_open: the forward's own mask
                d[j * LDA] = __builtin_fmaf(cA[j], dyh, __builtin_fmaf(cB[j], yv[j], cC[j]));
            }
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            float* d = sB + kq * LDB + arow + 32 * i;
            d[0 * LDB] = rb[i].x; d[1 * LDB] = rb[i].y; d[2 * LDB] = rb[i].z; d[3 * LDB] = rb[i].w;
        }
    };
    f32x16 acc[NT];
    float p_sc[NT], p_ns[NT], p_rs[NT], p_mr[NT], r0s[NT], r1s[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        r0s[nt] = r1s[nt] = 0.f;
        p_sc[nt] = p_ns[nt] = p_rs[nt] = p_mr[nt] = 0.f;
        if constexpr (RSUM) {
            const int col = n0 + nt * 32 + l31;
            p_sc[nt] = rs.scale[col];
            p_ns[nt] = -rs.shift[col];
            p_rs[nt] = (float)(1.0 / sqrt((double)rs.var[col] + (double)rs.eps));
            p_mr[nt] = -rs.mean[col] * p_rs[nt];
        }
    }
    const float* pa = sA + kh * LDA + wave * 32 + l31;
    const float* pb = sB + kh * LDB + l31;
    const unsigned lo_x = (unsigned)(4 * kh * ldx + n0 + l31) * 4u;
    const unsigned lo_p = RSUM ? (unsigned)(4 * kh * rs.ldyp + n0 + l31) * 4u : 0u;
    // tile loop around a chunk loop (not one flat loop over (tile, chunk) steps: with `if (chunk == 0) acc = 0` inside a flat loop hipcc
    // keeps the accumulators in VGPRs and moves all of them to the AGPRs and back around every chunk's MFMAs -- 96 moves per chunk)
    if ((int)bx < ntiles) fetch((int)bx, 0);
    __syncthreads();                                          // the channel constants
    for (int tile = (int)bx; tile < ntiles; tile += rowgrid) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        for (int c = 0; c < nchunks; ++c) {
            commit(tile, c);
            __syncthreads();
            // the next chunk's loads fly during this chunk's MFMAs; the next TILE's first chunk is fetched after this tile's epilogue (its ten
            // quads would otherwise be live across the epilogue's own 16 loads + 16 stores per column tile: 185 registers, two waves per SIMD)
            if (c + 1 < nchunks) fetch(tile, c + 1);
            constexpr int U = NT >= 2 ? 2 : 4;                   // k-pairs whose operands are fetched ahead of their MFMAs (register budget)
#pragma unroll
            for (int k0 = 0; k0 < 32; k0 += 2 * U) {
                float av[U], bv[U][NT];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    av[u] = pa[(k0 + 2 * u) * LDA];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bv[u][nt] = pb[(k0 + 2 * u) * LDB + nt * 32];
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u][nt], acc[nt], 0, 0, 0);
            }
            __syncthreads();
        }
        const int m0 = (tile << 7) + wave * 32;
        // (opaque per tile: otherwise hipcc hoists the 32 (lane offset + row * pitch) sums out of the tile loop as 64-bit pairs -- 50 registers,
        //  a wave per SIMD -- instead of adding a scalar row base per access)
        unsigned lpx = lo_p, lxx = lo_x;
        asm volatile("" : "+v"(lpx), "+v"(lxx));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if constexpr (RSUM) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {                  // eight loads in flight at a time (register budget: three waves per SIMD)
                    float yv[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        yv[r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(rs.Yp + (size_t)(m0 + (r & 3) + 8 * ((8 * h + r) >> 2)) * rs.ldyp) + lpx + nt * 128);
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const float dyh = yv[r] * p_sc[nt] > p_ns[nt] ? acc[nt][8 * h + r] : 0.f;
                        r0s[nt] += dyh;
                        r1s[nt] = __builtin_fmaf(dyh, __builtin_fmaf(yv[r], p_rs[nt], p_mr[nt]), r1s[nt]);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                *reinterpret_cast<float*>(reinterpret_cast<char*>(dX + (size_t)(m0 + (r & 3) + 8 * (r >> 2)) * ldx) + lxx + nt * 128) = acc[nt][r];
        }
        if (tile + rowgrid < ntiles) fetch(tile + rowgrid, 0);
    }
    if constexpr (RSUM) {
        float* sR = sA;                                  // [4 waves][2][BN]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            r0s[nt] += __shfl_xor(r0s[nt], 32, 64);
            r1s[nt] += __shfl_xor(r1s[nt], 32, 64);
            if (lane < 32) {
                sR[(wave * 2 + 0) * BN + nt * 32 + lane] = r0s[nt];
                sR[(wave * 2 + 1) * BN + nt * 32 + lane] = r1s[nt];
            }
        }
        __syncthreads();
        float* pr = rs.part + (size_t)bx * 2 * rs_cin;
        for (int j = t; j < BN; j += 256) {
            float v0 = 0.f, v1 = 0.f;
            for (int w = 0; w < 4; ++w) { v0 += sR[(w * 2 + 0) * BN + j]; v1 += sR[(w * 2 + 1) * BN + j]; }
            pr[n0 + j] = v0;
            pr[rs_cin + n0 + j] = v1;
        }
    }
}
// the lean kernel takes the launch if every one of its shape assumptions holds (GSPN_BWD_LEAN=0: never -- A/B hook); returns false otherwise
static bool bwd_lean_try(long rows, int cin, int cout, const gspn_dy_args* a, const float* W, int col0, int ncols, float* dX, int ldx, const DwJob* dwj,
                         hipStream_t st, const RsumArgs& rs, int* nparts_out, const DwJob& none) {
    static const int on = env_int("GSPN_BWD_LEAN", 1);
    if (!on) return false;
    const bool pooled = a->dZ == nullptr;
    if ((rows & 127) || rows <= 0 || (cout & 31) || (ncols & 31) || (col0 & 3)) return false;
    if (!vec_ok(a->Y, a->ldy) || !vec_ok(W, cout) || !vec_ok(dX, ldx)) return false;
    const long maxld = std::max(std::max((long)a->ldy, (long)ldx), std::max((long)(pooled ? 0 : a->ldz), (long)(rs.Yp ? rs.ldyp : 0)));
    if (rows * maxld >= (1L << 30) || (long)cin * cout >= (1L << 30)) return false;
    int pool_sh = 0;
    if (pooled) {
        if (a->ns < 32 || (a->ns & (a->ns - 1)) || (a->ns > 32 && a->ns < 128)) return false;
        if (((uintptr_t)a->dPool % 16) || ((uintptr_t)a->pool_arg % 16)) return false;
        pool_sh = __builtin_ctz(a->ns);
    } else if (!vec_ok(a->dZ, a->ldz)) return false;
    if (rs.Yp && (rs.ldyp & 3)) return false;
    static const int force_bn = env_int("GSPN_BWD_LEAN_BN", 0);          // (tuning hook)
    // measured on MI355X (tools/bwd_ablate.py, GSPN_BWD_LEAN_BN sweep): 64-column blocks everywhere they divide the range -- also for 128
    // columns, where a 128-wide block would read dY once instead of twice but leaves one wave per SIMD (1 M x 128 <- 256 pooled: 934 us
    // against 1165) -- and 32-column blocks for the short layers (<= 16384 rows), which need the workgroups
    int bn = (ncols % 64 == 0 && rows > 16384) ? 64 : 32;
    if (force_bn && ncols % force_bn == 0 && force_bn <= 64) bn = force_bn;
    const int yt = ncols / bn;
    const unsigned extra = dwj ? (unsigned)(dwj->nblk + dwj->nblk2) : 0u;
    const unsigned rg = row_grid(rows, yt, bwd_bpc_narrow());
    if (nparts_out) *nparts_out = (int)rg;
    DwJob dj = dwj ? *dwj : none;
    dj.rowgrid = (int)rg;
    const dim3 g(rg * (unsigned)yt + extra);
    const size_t dyn = sizeof(float) * 5 * chan_pad(cout);
#define BL_GO(NT_, P_, R_, D_) hipLaunchKernelGGL((bwd_lean_kernel<NT_, P_, R_, D_>), g, dim3(256), dyn, st, (int)rows, cout, *a, W, dX, ldx, col0, dj, rs, (int)rg, pool_sh, col0 + ncols)
#define BL_D(NT_, P_, R_) do { if (dwj) BL_GO(NT_, P_, R_, true); else BL_GO(NT_, P_, R_, false); } while (0)
#define BL_R(NT_, P_) do { if (rs.Yp) BL_D(NT_, P_, true); else BL_D(NT_, P_, false); } while (0)
#define BL_P(NT_) do { if (!pooled) BL_R(NT_, 0); else if (pool_sh == 5) BL_R(NT_, 1); else BL_R(NT_, 2); } while (0)
    if (bn == 64) BL_P(2); else BL_P(1);
#undef BL_P
#undef BL_R
#undef BL_D
#undef BL_GO
    return true;
}
// ============================================================================================
// Pass A and pass B of one layer in ONE launch (r03): dW = x^T . dY and dX = dY . W^T from the same staged dY tile.
// The two-launch form reads (x, Y, dZ) for dW and (Y, dZ, previous Y for the BN reductions) again for dX, and rebuilds dY from
// (Y, dZ, coefficients) in both -- seven row streams and twice the vector work for four streams' worth of information.  Here a
// workgroup walks row tiles; per tile it stages dY (built once) and the previous layer's raw output y_p (= x before BN + ReLU)
// TRANSPOSED in LDS, and every wave runs two 32x32 products from them:
//     dX tile (its 32 rows x 32 input channels)   += dY[rows, :] . W^T         A = dY column k, B = W^T (resident in LDS for the whole kernel)
//     dW tile (32 input x 32 output channels)     += x^T[:, rows] . dY[rows, :]  A = relu(y_p*scale + shift) (two roundings, the forward's
//                                                                               operand), B = dY; accumulated over ALL tiles of the workgroup
// and the epilogue of dX takes the previous layer's BN reductions (sum dyh, sum dyh*xhat) with y_p read back from LDS instead of HBM.
// Each workgroup leaves one partial dW tile set (slot = workgroup) for the usual fixed-order reduction (wgrad_dw_kernel, plain) and one
// partial row of BN reductions: results do not depend on scheduling.  Shapes (fused_grid() is the authority): known coefficients; a dense
// dZ or the gradient of a max-pool over groups of 32 rows (POOL32); cin in {32, 64}, cout in {32, 64, 128} -- 128 as two chunks of 64
// columns --; rows >= 65536 and a multiple of the tile (128 / (cin/32)); 16-byte aligned pitches.
// ============================================================================================
#ifndef GSPN_FUSED_ABL
#define GSPN_FUSED_ABL 0
#endif
template <int CI, int CO, bool RSUM, bool POOL32>      // POOL32: the upstream gradient is (rows / 32, cout) + arg-max rows (a max-pool over groups of 32 rows)
__global__ __launch_bounds__(256) void bwd_fused_kernel(int rows, gspn_dy_args a, const float* __restrict__ W, const float* __restrict__ Xp, int ldxp,
                                                        const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                        float* __restrict__ dX, int ldx, float* __restrict__ PP, RsumArgs rs) {
    constexpr int CIN = 32 * CI, COUT = 32 * CO;
    // cout wider than 64: the layer is walked in NCH chunks of CW = 2 column tiles -- dY is built and staged one chunk at a time (the dX tile
    // accumulates over the chunks, every chunk has its own persistent dW tiles), W^T stays resident for all of them
    constexpr int NCH = CO > 2 ? CO / 2 : 1, CW = CO / NCH, COUTC = 32 * CW;
    static_assert(CW * NCH == CO && NCH <= 2, "cout in {32, 64, 128}");
    constexpr int TR = 128 / CI;                                  // rows per tile: (TR / 32) * CI = 4 dX tiles, one per wave
    constexpr int LD = TR + 1, LDW = CIN + 1;                     // odd pitches: lanes along either axis of a transposed tile hit distinct banks
    constexpr int S = (CI * CW >= 4) ? 1 : 4 / (CI * CW);         // waves sharing one dW tile (each takes TR / S rows of every row tile)
    constexpr int QY = 8 * CW, RY = 256 / QY, PY = TR / RY;       // Y / dZ of a chunk: quads per row, rows per pass, passes
    constexpr int QX = 8 * CI, RX = 256 / QX, PX = TR / RX;       // y_p
    __shared__ __attribute__((aligned(16))) float s_tile[COUTC * LD + CIN * LD];
    float* const sdY = s_tile;                                     // [o][r]   (the chunk's columns)
    float* const sX = s_tile + COUTC * LD;                         // [i][r]   raw y_p
    __shared__ __attribute__((aligned(16))) float sW[COUT * LDW];  // [o][i]   W^T
    __shared__ __attribute__((aligned(16))) float s_chan[5 * COUT];     // forward scale, MINUS shift, cA, cB, cC
    static_assert(4 * 32 * 33 <= COUTC * LD + CIN * LD, "the end-of-kernel reductions reuse the tile buffers");
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int bx = blockIdx.x, grid = gridDim.x;
    const int ntiles = rows / TR;
    for (int i = t; i < COUT; i += 256) {
        s_chan[i] = a.scale[i];
        s_chan[COUT + i] = -a.shift[i];
        s_chan[2 * COUT + i] = a.cA[i];
        s_chan[3 * COUT + i] = a.cB[i];
        s_chan[4 * COUT + i] = a.cC[i];
    }
    for (int i = t; i < CIN * COUT; i += 256) {                   // W (cin, cout) row-major -> sW[o][i]
        const int ci = i / COUT, o = i - ci * COUT;
        sW[o * LDW + ci] = W[i];
    }
    const int kqy = (t % QY) * 4, ary = t / QY;
    const int kqx = (t % QX) * 4, arx = t / QX;
    const unsigned oy = (unsigned)(ary * a.ldy + kqy) * 4u, oz = POOL32 ? 0u : (unsigned)(ary * a.ldz + kqy) * 4u, ox = (unsigned)(arx * ldxp + kqx) * 4u;
    constexpr int NG = TR / 32;                                   // pool groups per tile
    constexpr int NZ = POOL32 ? NG : PY;
    static_assert(!POOL32 || (RY <= 32 && 32 % RY == 0), "a pass of the pooled form stays inside one group");
    float4 ry[PY], rz[NZ], rx[PX];
    int4 rarg[POOL32 ? NG : 1];
    auto fetch = [&](int tile, int ch) {                           // chunk ch of the tile (y_p comes with chunk 0)
        const size_t m0 = (size_t)tile * TR;
        const char* yb = reinterpret_cast<const char*>(a.Y + m0 * a.ldy + ch * COUTC);
        const char* xb = reinterpret_cast<const char*>(Xp + m0 * ldxp);
#pragma unroll
        for (int i = 0; i < PY; ++i) ry[i] = *reinterpret_cast<const float4*>(yb + (size_t)(RY * i) * a.ldy * 4 + oy);
        if constexpr (POOL32) {                                   // this thread's four channels of the tile's NG groups
            const size_t g0 = (size_t)tile * NG * COUT + ch * COUTC + kqy;
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                rz[gi] = *reinterpret_cast<const float4*>(a.dPool + g0 + (size_t)gi * COUT);
                rarg[gi] = *reinterpret_cast<const int4*>(a.pool_arg + g0 + (size_t)gi * COUT);
            }
        } else {
            const char* zb = reinterpret_cast<const char*>(a.dZ + m0 * a.ldz + ch * COUTC);
#pragma unroll
            for (int i = 0; i < PY; ++i) rz[i] = *reinterpret_cast<const float4*>(zb + (size_t)(RY * i) * a.ldz * 4 + oz);
        }
        if (NCH == 1 || ch == 0) {
#pragma unroll
            for (int i = 0; i < PX; ++i) rx[i] = *reinterpret_cast<const float4*>(xb + (size_t)(RX * i) * ldxp * 4 + ox);
        }
    };
    // prepare(): dY of the fetched chunk, in registers.  It runs BEFORE the epilogue's stores are issued: hipcc waits with vmcnt(0) for a load
    // whenever loads and stores are both in flight (on gfx9 they share the counter and count as out of order), so a wait for the prefetched
    // tile placed after the stores -- at the top of the next tile, where the values are needed -- also waits for stores issued a moment ago:
    // their whole latency, every tile.  Consumed before the stores, the prefetch waits for nothing but itself.
    float vdy[PY][4];
    auto prepare = [&](int ch) {
        const int kc = ch * COUTC + kqy;
        const float4 q_sc = *reinterpret_cast<const float4*>(s_chan + kc), q_ns = *reinterpret_cast<const float4*>(s_chan + COUT + kc);
        const float4 q_a = *reinterpret_cast<const float4*>(s_chan + 2 * COUT + kc), q_b = *reinterpret_cast<const float4*>(s_chan + 3 * COUT + kc);
        const float4 q_c = *reinterpret_cast<const float4*>(s_chan + 4 * COUT + kc);
        const float sc[4] = {q_sc.x, q_sc.y, q_sc.z, q_sc.w}, ns[4] = {q_ns.x, q_ns.y, q_ns.z, q_ns.w};
        const float cA[4] = {q_a.x, q_a.y, q_a.z, q_a.w}, cB[4] = {q_b.x, q_b.y, q_b.z, q_b.w}, cC[4] = {q_c.x, q_c.y, q_c.z, q_c.w};
#pragma unroll
        for (int i = 0; i < PY; ++i) {
            const float yv[4] = {ry[i].x, ry[i].y, ry[i].z, ry[i].w};
            float zv[4];
            if constexpr (POOL32) {                               // row ary + RY*i of the tile: group (ary + RY*i) / 32, row in group the remainder
                constexpr int PPG = 32 / RY;                      // passes per group
                const int gi = i / PPG, rin = ary + RY * (i % PPG);
                const int av[4] = {rarg[gi].x, rarg[gi].y, rarg[gi].z, rarg[gi].w};
                const float dv[4] = {rz[gi].x, rz[gi].y, rz[gi].z, rz[gi].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) zv[j] = av[j] == rin ? dv[j] : 0.f;
            } else {
                zv[0] = rz[i].x; zv[1] = rz[i].y; zv[2] = rz[i].z; zv[3] = rz[i].w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#if GSPN_FUSED_ABL & 4
                vdy[i][j] = yv[j] + zv[j];
#else
                const float dyh = yv[j] * sc[j] > ns[j] ? zv[j] : 0.f;                                   // relu_open: the forward's own mask
                vdy[i][j] = __builtin_fmaf(cA[j], dyh, __builtin_fmaf(cB[j], yv[j], cC[j]));
#endif
            }
        }
        if (NCH == 1 || ch == 0) {
#pragma unroll
            for (int i = 0; i < PX; ++i) asm volatile("" : "+v"(rx[i].x), "+v"(rx[i].y), "+v"(rx[i].z), "+v"(rx[i].w));      // (y_p is consumed here too: its wait belongs in front of the stores)
        }
    };
    auto commit = [&](int ch) {
#pragma unroll
        for (int i = 0; i < PY; ++i) {
            float* d = sdY + kqy * LD + ary + RY * i;
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j * LD] = vdy[i][j];
        }
        if (NCH == 1 || ch == 0) {
#pragma unroll
            for (int i = 0; i < PX; ++i) {
                float* d = sX + kqx * LD + arx + RX * i;
                d[0 * LD] = rx[i].x; d[1 * LD] = rx[i].y; d[2 * LD] = rx[i].z; d[3 * LD] = rx[i].w;
            }
        }
    };
    // this wave's tiles
    const int rt = wave / CI, ct = wave % CI;                    // dX: rows rt*32.., input channels ct*32..
    const int wt = wave / S, part = wave % S;                    // dW: tile wt = (mi, ni), rows part*(TR/S).. of every row tile
    const int mi = wt / CW, ni = wt % CW;                        // (ni: column tile inside a chunk)
    constexpr int KW = TR / S;                                   // rows (k) of a row tile this wave feeds into its dW tile
    const float* pa_x = sdY + kh * LD + rt * 32 + l31;           // dX  A: dY[row][k]   at sdY[k][row]
    const float* pb_x = sW + kh * LDW + ct * 32 + l31;           //     B: W^T[k][i]    at sW[k][i]
    const float* pa_w = sX + (mi * 32 + l31) * LD + part * KW + kh;     // dW  A: x[row k][i]  at sX[i][row]
    const float* pb_w = sdY + (ni * 32 + l31) * LD + part * KW + kh;    //     B: dY[row k][o] at sdY[o][row]
    const float w_sc = in_scale[mi * 32 + l31], w_sh = in_shift[mi * 32 + l31];
    float p_sc = 0.f, p_ns = 0.f, p_rs = 0.f, p_mr = 0.f, r0s = 0.f, r1s = 0.f;
    if constexpr (RSUM) {
        const int col = ct * 32 + l31;
        p_sc = rs.scale[col];
        p_ns = -rs.shift[col];
        p_rs = (float)(1.0 / sqrt((double)rs.var[col] + (double)rs.eps));
        p_mr = -rs.mean[col] * p_rs;
    }
    const unsigned ax_x = lds_addr(pa_x), bx_x = lds_addr(pb_x), ax_w = lds_addr(pa_w), bx_w = lds_addr(pb_w);
    const float* pe = sX + (ct * 32 + l31) * LD + rt * 32 + 4 * kh;     // the epilogue's y_p: rows 8*(r>>2) + (r&3) from here
    const unsigned lo_x = (unsigned)((rt * 32 + 4 * kh) * ldx + ct * 32 + l31) * 4u;
    f32x16 accw[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) accw[c][r] = 0.f;
    fetch(bx, 0);
    __syncthreads();                                             // the constants and W^T
    prepare(0);
    for (int tile = bx; tile < ntiles; tile += grid) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        sfor<NCH>([&](auto ch_) {
            constexpr int ch = decltype(ch_)::value;
            commit(ch);
            __syncthreads();
            // the next (tile, chunk)'s quads fly during this chunk's MFMAs
            if constexpr (ch + 1 < NCH) fetch(tile, ch + 1);
            else if (!(GSPN_FUSED_ABL & 32) && tile + grid < ntiles) fetch(tile + grid, 0);
            lds_product<COUTC / 2, 2 * LD * 4, 2 * LDW * 4, false, (GSPN_FUSED_ABL & 1)>(acc, ax_x, bx_x + ch * COUTC * LDW * 4, 0.f, 0.f);
            lds_product<KW / 2, 8, 8, true, (GSPN_FUSED_ABL & 2)>(accw[ch], ax_w, bx_w, w_sc, w_sh);
            if constexpr (ch + 1 < NCH) {
                prepare(ch + 1);
                __syncthreads();                                 // (the chunk's dY is consumed: the next chunk may overwrite it)
            }
        });
        if constexpr (RSUM && !(GSPN_FUSED_ABL & 16)) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float yv[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) yv[r] = pe[8 * ((8 * h + r) >> 2) + (r & 3)];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float dyh = yv[r] * p_sc > p_ns ? acc[8 * h + r] : 0.f;
                    r0s += dyh;
                    r1s = __builtin_fmaf(dyh, __builtin_fmaf(yv[r], p_rs, p_mr), r1s);
                }
            }
        }
        if (tile + grid < ntiles) prepare(0);                   // the next tile's values, before this tile's stores (see prepare)
        if (!(GSPN_FUSED_ABL & 8) || acc[0] == 1.2345f) {
            // (opaque per tile: otherwise hipcc hoists the 16 (lane offset + row * pitch) sums out of the tile loop as 64-bit pairs)
            unsigned lxx = lo_x;
            asm volatile("" : "+v"(lxx));
            const char* xrow = reinterpret_cast<const char*>(dX + (size_t)tile * TR * ldx);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                *reinterpret_cast<float*>(const_cast<char*>(xrow) + (size_t)(8 * (r >> 2) + (r & 3)) * ldx * 4 + lxx) = acc[r];
        }
        __syncthreads();
    }
    // ---- the workgroup's partial dW tile set -> slot bx; its BN-reduction row -> part[bx] ----
    float* sred = s_tile;                                        // [4 waves][32][33] (the tile buffers are free now)
    float* slot = PP + (size_t)bx * 2 * CIN * COUT;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        if constexpr (S == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) slot[(size_t)(mi * 32 + c_row(r, lane)) * COUT + ch * COUTC + ni * 32 + l31] = accw[ch][r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) sred[(wave * 32 + c_row(r, lane)) * 33 + l31] = accw[ch][r];
            __syncthreads();
            for (int i = t; i < CIN * COUTC; i += 256) {
                const int m = i / COUTC, n = i - m * COUTC;
                const int w0 = ((m >> 5) * CW + (n >> 5)) * S;   // the first of the S waves of tile (m / 32, n / 32) of the chunk
                float v = 0.f;
#pragma unroll
                for (int q = 0; q < S; ++q) v += sred[((w0 + q) * 32 + (m & 31)) * 33 + (n & 31)];
                slot[(size_t)m * COUT + ch * COUTC + n] = v;
            }
            __syncthreads();
        }
    }
    if constexpr (RSUM) {
        float* sR = sX;                                          // [4 waves][2][32]
        r0s += __shfl_xor(r0s, 32, 64);
        r1s += __shfl_xor(r1s, 32, 64);
        if (lane < 32) { sR[(wave * 2 + 0) * 32 + lane] = r0s; sR[(wave * 2 + 1) * 32 + lane] = r1s; }
        __syncthreads();
        float* pr = rs.part + (size_t)bx * 2 * CIN;
        for (int j = t; j < CIN; j += 256) {
            const int c = j >> 5;                                // waves with wave % CI == c hold column block c
            float v0 = 0.f, v1 = 0.f;
            for (int w = c; w < 4; w += CI) { v0 += sR[(w * 2 + 0) * 32 + (j & 31)]; v1 += sR[(w * 2 + 1) * 32 + (j & 31)]; }
            pr[j] = v0;
            pr[CIN + j] = v1;
        }
    }
}
// workgroups of the fused kernel for (rows, cin, cout); 0 = the shape is not one of its own.  The grid is what the chip holds AT ONCE
// (the kernel's measured occupancy x the planned CUs): 672 workgroups of the two-per-CU 64 x 64 instance ran as one and a half rounds
// -- 90 us against 73 for 448 (tools/fused_ablate.py, GSPN_BWD_FUSED_BPC sweep).
template <int CI, int CO> static int fused_occupancy() {
    static int occ = 0;
    if (!occ) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bwd_fused_kernel<CI, CO, true, false>, 256, 0) != hipSuccess || n < 1) n = 2;
        occ = n > 4 ? 4 : n;
    }
    return occ;
}
static unsigned fused_grid(long rows, int cin, int cout) {
    static const int on = env_int("GSPN_BWD_FUSED", 1);
    if (!on) return 0;
    if (!((cin == 32 || cin == 64) && (cout == 32 || cout == 64 || cout == 128))) return 0;
    const int tr = 128 / (cin / 32);
    if (rows < 65536 || rows % tr || rows % 128) return 0;        // (short layers: their launches are latency chains, two kernels overlap better)
    static const int bpc_env = env_int("GSPN_BWD_FUSED_BPC", 0);
    int bpc = cin == 32 ? (cout == 32 ? fused_occupancy<1, 1>() : (cout == 64 ? fused_occupancy<1, 2>() : fused_occupancy<1, 4>()))
                        : (cout == 32 ? fused_occupancy<2, 1>() : (cout == 64 ? fused_occupancy<2, 2>() : fused_occupancy<2, 4>()));
    if (bpc_env > 0) bpc = bpc_env > 4 ? 4 : bpc_env;
    // never more workgroups than 128-row tiles: the BN-reduction buffer (gspn_rsum_part_floats) holds one row per such tile at most
    const long ntiles = rows / 128, cap = (long)GSPN_PLAN_CUS * bpc;
    return (unsigned)(ntiles < cap ? ntiles : cap);
}
extern "C" long gspn_mlp_bwd_fused_work_bytes(long rows, int cin, int cout) {
    const unsigned g = fused_grid(rows, cin, cout);
    return g ? (long)(sizeof(float) * (size_t)g * 2 * cin * cout + 64) : 0;
}
// One launch for both backward products of a layer with KNOWN coefficients (a->cA/cB/cC final) and a dense upstream gradient:
//   dW (cin, cout) = relu(Xp*in_scale + in_shift)^T . dY        (work: gspn_mlp_bwd_work_bytes(rows, cin, cout) bytes, as for pass A)
//   dX (rows, ldx >= cin) = dY . W^T
//   part (optional, with Xp's layer statistics): the BN reductions of dX against Xp, as gspn_mlp_bwd_data_ex leaves them; *nparts_out rows
// GSPN_ERR_UNSUPPORTED for every shape outside the kernel's own (the caller then runs pass A and pass B).
static int bwd_fused_impl(long rows, int cin, int cout, const gspn_dy_args* a, const float* W, const float* Xp, int ldxp, const float* in_scale,
                          const float* in_shift, float* dX, int ldx, float* work, float* dW, const float* mean_p, const float* var_p,
                          float eps_p, float* part, int* nparts_out, void* stream, CoefJob* cj) {
    if (rows <= 0 || cin <= 0 || cout <= 0 || !a || !a->Y || !a->scale || !a->shift || !W || !Xp || !dX || !work || !dW || ldxp < cin || ldx < cin)
        return GSPN_ERR_ARG;
    if (part && (!mean_p || !var_p || !nparts_out)) return GSPN_ERR_ARG;
    if (a->ldy < cout || (a->dZ && a->ldz < cout)) return GSPN_ERR_ARG;
    const unsigned g = fused_grid(rows, cin, cout);
    const bool pooled = a->dZ == nullptr;
    if (!g || !a->cA || !a->cB || !a->cC || !in_scale || !in_shift) return GSPN_ERR_UNSUPPORTED;
    if (pooled) {                                                 // a max-pool over groups of 32 rows (nsample = 32): (rows / 32, cout) gradient + arg rows
        if (!a->dPool || !a->pool_arg) return GSPN_ERR_ARG;
        if (a->ns != 32 || ((uintptr_t)a->dPool % 16) || ((uintptr_t)a->pool_arg % 16)) return GSPN_ERR_UNSUPPORTED;
    } else if (!vec_ok(a->dZ, a->ldz)) return GSPN_ERR_UNSUPPORTED;
    if (!vec_ok(a->Y, a->ldy) || !vec_ok(Xp, ldxp) || (ldx & 3) || ((uintptr_t)dX % 16) || ((uintptr_t)work % 16)) return GSPN_ERR_UNSUPPORTED;
    const long maxld = std::max(std::max((long)a->ldy, pooled ? 0L : (long)a->ldz), std::max((long)ldx, (long)ldxp));
    if (rows * maxld >= (1L << 30)) return GSPN_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    // in_scale / in_shift ARE the previous layer's forward scale / shift: the epilogue's mask uses the same pair
    RsumArgs rs{Xp, ldxp, in_scale, in_shift, mean_p, var_p, eps_p, part};
    float* PP = work;
#define BF_GO(CI_, CO_)                                                                                                                   \
    do {                                                                                                                                   \
        if (pooled) {                                                                                                                      \
            if (part) hipLaunchKernelGGL((bwd_fused_kernel<CI_, CO_, true, true>), dim3(g), dim3(256), 0, st, (int)rows, *a, W, Xp, ldxp, in_scale, in_shift, dX, ldx, PP, rs);  \
            else hipLaunchKernelGGL((bwd_fused_kernel<CI_, CO_, false, true>), dim3(g), dim3(256), 0, st, (int)rows, *a, W, Xp, ldxp, in_scale, in_shift, dX, ldx, PP, rs); \
        } else if (part) hipLaunchKernelGGL((bwd_fused_kernel<CI_, CO_, true, false>), dim3(g), dim3(256), 0, st, (int)rows, *a, W, Xp, ldxp, in_scale, in_shift, dX, ldx, PP, rs);  \
        else hipLaunchKernelGGL((bwd_fused_kernel<CI_, CO_, false, false>), dim3(g), dim3(256), 0, st, (int)rows, *a, W, Xp, ldxp, in_scale, in_shift, dX, ldx, PP, rs); \
    } while (0)
    if (cin == 32 && cout == 32) BF_GO(1, 1);
    else if (cin == 32 && cout == 64) BF_GO(1, 2);
    else if (cin == 32) BF_GO(1, 4);
    else if (cout == 32) BF_GO(2, 1);
    else if (cout == 64) BF_GO(2, 2);
    else BF_GO(2, 4);
#undef BF_GO
    if (nparts_out) *nparts_out = (int)g;
    DwJob j = dw_job(rows, cin, cout, (long)g, PP, nullptr, nullptr, nullptr, nullptr, 0.f, 0, 0, dW);
    j.plain = 1;
    if (cj) {                                                      // the previous layer's coefficients from `part` + this layer's dW reduction: one launch
        cj->nparts = (int)g;
        hipLaunchKernelGGL(bwd_coef_dw_kernel, dim3((unsigned)(cj->c + j.nblk)), dim3(256), 0, st, *cj, j);
        return gspn_launch_status();
    }
    hipLaunchKernelGGL(wgrad_dw_kernel, dim3((unsigned)dw_blocks((long)cin * cout, g, 1024)), dim3(1024), 0, st, j);
    return gspn_launch_status();
}
extern "C" int gspn_mlp_bwd_fused(long rows, int cin, int cout, const gspn_dy_args* a, const float* W, const float* Xp, int ldxp, const float* in_scale,
                                  const float* in_shift, float* dX, int ldx, float* work, float* dW, const float* mean_p, const float* var_p,
                                  float eps_p, float* part, int* nparts_out, void* stream) {
    return bwd_fused_impl(rows, cin, cout, a, W, Xp, ldxp, in_scale, in_shift, dX, ldx, work, dW, mean_p, var_p, eps_p, part, nparts_out, stream, nullptr);
}
// gspn_mlp_bwd_fused followed by gspn_mlp_bwd_coef of the PREVIOUS layer (its cin channels: batch statistics mean_p / var_p, gamma_p; outputs
// cA_p .. dbias_p as gspn_mlp_bwd_coef writes them) with this layer's dW reduction in the same second launch.  part / nparts_out required.
extern "C" int gspn_mlp_bwd_fused_coef(long rows, int cin, int cout, const gspn_dy_args* a, const float* W, const float* Xp, int ldxp, const float* in_scale,
                                       const float* in_shift, float* dX, int ldx, float* work, float* dW, const float* mean_p, const float* var_p,
                                       float eps_p, float* part, int* nparts_out, const float* gamma_p, float* cA_p, float* cB_p, float* cC_p,
                                       float* dgamma_p, float* dbeta_p, float* dbias_p, void* stream) {
    if (!part || !nparts_out || !mean_p || !var_p || !cA_p || !cB_p || !cC_p) return GSPN_ERR_ARG;
    CoefJob cj{rows, cin, 0, part, mean_p, var_p, gamma_p, eps_p, cA_p, cB_p, cC_p, dgamma_p, dbeta_p, dbias_p};
    return bwd_fused_impl(rows, cin, cout, a, W, Xp, ldxp, in_scale, in_shift, dX, ldx, work, dW, mean_p, var_p, eps_p, part, nparts_out, stream, &cj);
}
static int bwd_data_launch(long rows, int cin, int cout, const gspn_dy_args* a, const float* W, int col0, int ncols, float* dX, int ldx, const DwJob* dwj,
                           hipStream_t st, const RsumArgs* rsp = nullptr, int* nparts_out = nullptr) {
    const RsumArgs rs = rsp ? *rsp : RsumArgs{nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr};
    const bool pooled = a->dZ == nullptr;
    const bool v = vec_ok(a->Y, a->ldy) && vec_ok(W, cout) &&
                   (pooled ? (((uintptr_t)a->dPool) % 16 == 0 && ((uintptr_t)a->pool_arg) % 16 == 0) : vec_ok(a->dZ, a->ldz));
    const int cend = col0 + ncols;
    const DwJob none = dw_job(0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0, 0, nullptr);
    const unsigned extra = dwj ? (unsigned)(dwj->nblk + dwj->nblk2) : 0u;
    // workgroups per CU the grid is sized for.  3 reside (waves per SIMD: the amdgpu_waves_per_eu attribute of the kernel keeps the 32- and
    // 64-column tiles under 168 VGPRs; 2 for 128), yet 4 measures best (pass B of the bench step: 648 / 596 / 581 us at 2 / 3 / 4):
    // the fourth quarter of the workgroups fills the slots the first finishers free.  GSPN_BWD_BPC overrides (tuning hook).
    const int bpc_narrow = bwd_bpc_narrow();
    if (bwd_lean_try(rows, cin, cout, a, W, col0, ncols, dX, ldx, dwj, st, rs, nparts_out, none)) return gspn_launch_status();
#define BD_GO(BN_, V_, P_, YT_)                                                                                                       \
    do {                                                                                                                               \
        const unsigned rg = row_grid(rows, YT_, BN_ >= 128 ? 2 : bpc_narrow);                                                          \
        const dim3 g(dwj ? rg * (unsigned)(YT_) + extra : rg, dwj ? 1 : YT_);                                                          \
        if (nparts_out) *nparts_out = (int)rg;                                                                                         \
        DwJob dj = dwj ? *dwj : none;                                                                                                  \
        dj.rowgrid = (int)rg;                                                                                                          \
        if (dwj) hipLaunchKernelGGL((mlp_bwd_data_kernel<BN_, V_, P_, true>), g, dim3(256), sizeof(float) * 5 * chan_pad(cout), st, rows, cend, cout, *a, W, dX, ldx, col0, dj, rs);   \
        else     hipLaunchKernelGGL((mlp_bwd_data_kernel<BN_, V_, P_, false>), g, dim3(256), sizeof(float) * 5 * chan_pad(cout), st, rows, cend, cout, *a, W, dX, ldx, col0, none, rs);  \
    } while (0)
#define BD_LAUNCH(BN_, V_, YT_) do { if (pooled) BD_GO(BN_, V_, true, YT_); else BD_GO(BN_, V_, false, YT_); } while (0)
    const int bn = pick_bn(rows, ncols, "GSPN_BWD_FORCE_BN");
    const int yt = (ncols + bn - 1) / bn;
    if (bn == 32) { if (v) BD_LAUNCH(32, true, yt); else BD_LAUNCH(32, false, yt); }
    else if (bn == 64) { if (v) BD_LAUNCH(64, true, yt); else BD_LAUNCH(64, false, yt); }
    else { if (v) BD_LAUNCH(128, true, yt); else BD_LAUNCH(128, false, yt); }
#undef BD_LAUNCH
#undef BD_GO
    return gspn_launch_status();
}
static int bwd_data_check(long rows, int cin, int cout, const gspn_dy_args* a, int col0, int ncols, int ldx) {
    if (rows < 0 || cin <= 0 || cout <= 0 || ldx < cin || !a || !a->Y || !a->scale || !a->shift || !a->cA || !a->cB || !a->cC) return GSPN_ERR_ARG;
    if (!a->dZ && !(a->dPool && a->pool_arg && a->ns > 0)) return GSPN_ERR_ARG;
    if (col0 < 0 || ncols <= 0 || col0 + ncols > cin) return GSPN_ERR_ARG;
    if (cin > MAXCH || cout > MAXCH) return GSPN_ERR_UNSUPPORTED;
    if (rows >= (1L << 31)) return GSPN_ERR_UNSUPPORTED;
    return 0;
}
// columns [col0, col0 + ncols) of dX only: the rest of the row is left untouched (the caller does not need it -- e.g. the xyz columns
// of a set-abstraction input, whose gradient pointnet_util.py never uses -- which can halve the N dimension of the GEMM)
extern "C" int gspn_mlp_bwd_data_cols(long rows, int cin, int cout, const gspn_dy_args* a, const float* W, int col0, int ncols, float* dX, int ldx,
                                      void* stream) {
    const int rc = bwd_data_check(rows, cin, cout, a, col0, ncols, ldx);
    if (rc) return rc;
    if (rows == 0) return 0;
    return bwd_data_launch(rows, cin, cout, a, W, col0, ncols, dX, ldx, nullptr, (hipStream_t)stream);
}
// gspn_mlp_bwd_data_cols and gspn_mlp_bwd_dw in ONE launch: pass B of a layer plus the dW reduction that gspn_mlp_bwd_wgrad(..., dW = NULL)
// left undone (same rows/cin/cout/a/X/ldx_in as that call, so that the workspace layout is found again).  dX and dW are what the two
// separate calls produce (dW: same sums, 16 instead of 64 slot slices per output).
extern "C" int gspn_mlp_bwd_data_dw(long rows, int cin, int cout, const gspn_dy_args* a, const float* W, int col0, int ncols, float* dX, int ldx,
                                    const float* X, int ldx_in, const float* var, const float* gamma, float eps, int use_bn, int is_training,
                                    const float* work, float* dW, void* stream) {
    const int rc = bwd_data_check(rows, cin, cout, a, col0, ncols, ldx);
    if (rc) return rc;
    if (rows <= 0 || !work || !dW || ldx_in < cin || (use_bn && !var)) return GSPN_ERR_ARG;
    bool use_stream;
    const WgradPlan p = wgrad_choose(rows, cin, cout, a, X, ldx_in, &use_stream);
    const char* wb = reinterpret_cast<const char*>(work);
    const DwJob j = dw_job(rows, cin, cout, p.nslots, reinterpret_cast<const float*>(wb + ws_off_pp(p.nch, cin, cout)), reinterpret_cast<const double*>(wb),
                           reinterpret_cast<const float*>(wb + ws_off_g3(cout)), var, gamma, eps, use_bn, is_training, dW);
    return bwd_data_launch(rows, cin, cout, a, W, col0, ncols, dX, ldx, &j, (hipStream_t)stream);
}
// gspn_mlp_bwd_data_dw with a second, plain reduction riding along: dW2 (cin2, cout) = the sum of nslots2 partial tiles at part2
// (layout [slot][2][cin2*cout], first half used -- what gspn_preagg_bwd_dy leaves when it is not asked to reduce them itself)
extern "C" int gspn_mlp_bwd_data_dw2(long rows, int cin, int cout, const gspn_dy_args* a, const float* W, int col0, int ncols, float* dX, int ldx,
                                     const float* X, int ldx_in, const float* var, const float* gamma, float eps, int use_bn, int is_training,
                                     const float* work, float* dW, const float* part2, int cin2, int nslots2, float* dW2, void* stream) {
    const int rc = bwd_data_check(rows, cin, cout, a, col0, ncols, ldx);
    if (rc) return rc;
    if (rows <= 0 || !work || !dW || ldx_in < cin || (use_bn && !var)) return GSPN_ERR_ARG;
    if (cin2 < 0 || nslots2 < 0 || (cin2 > 0 && (!part2 || !dW2 || nslots2 <= 0))) return GSPN_ERR_ARG;
    bool use_stream;
    const WgradPlan p = wgrad_choose(rows, cin, cout, a, X, ldx_in, &use_stream);
    const char* wb = reinterpret_cast<const char*>(work);
    DwJob j = dw_job(rows, cin, cout, p.nslots, reinterpret_cast<const float*>(wb + ws_off_pp(p.nch, cin, cout)), reinterpret_cast<const double*>(wb),
                     reinterpret_cast<const float*>(wb + ws_off_g3(cout)), var, gamma, eps, use_bn, is_training, dW);
    if (cin2 > 0) {
        j.PP2 = part2; j.dW2 = dW2; j.cin2 = cin2; j.nslots2 = nslots2;
        j.nblk2 = (int)dw_blocks((long)cin2 * cout, nslots2, 256);
    }
    return bwd_data_launch(rows, cin, cout, a, W, col0, ncols, dX, ldx, &j, (hipStream_t)stream);
}
// ---- pass B of a pooled top layer as a streaming GEMM on the layer's input (mlp_fwd_stream_kernel<.., BWDP>) ----
// the small operands: workgroups [0, cin): row n of M = W diag(cB) W^T and cvec[n] = sum_c (b[c]*cB[c] + cC[c]) * W[n][c]; the rest: the
// dW reduction of the layer (dwj.nblk workgroups, its usual ride)
__global__ __launch_bounds__(256) void pooltop_prep_kernel(int cin, int ctop, const float* __restrict__ W, const float* __restrict__ bias,
                                                           const float* __restrict__ cB, const float* __restrict__ cC, float* __restrict__ M,
                                                           float* __restrict__ cvec, DwJob dwj) {
    __shared__ __attribute__((aligned(16))) double prep_sh[2 * 16 * DW_OX + MAXCH / 2 + 8];
    const int t = threadIdx.x;
    int b = blockIdx.x;
    if (b < cin) {
        float* wrow = reinterpret_cast<float*>(prep_sh + 8);       // W[n][:] * cB[:]
        const float* wn = W + (size_t)b * ctop;
        double cv = 0.0;
        for (int c = t; c < ctop; c += 256) {
            wrow[c] = wn[c] * cB[c];
            cv += ((double)(bias ? bias[c] : 0.f) * (double)cB[c] + (double)cC[c]) * (double)wn[c];
        }
        cv = wave_sum_f64(cv);
        if ((t & 63) == 0) prep_sh[t >> 6] = cv;
        __syncthreads();
        if (t == 0) cvec[b] = (float)((prep_sh[0] + prep_sh[1]) + (prep_sh[2] + prep_sh[3]));
        for (int m = t; m < cin; m += 256) {
            const float* wm = W + (size_t)m * ctop;
            double acc = 0.0;
            for (int c = 0; c < ctop; ++c) acc += (double)wrow[c] * (double)wm[c];
            M[(size_t)b * cin + m] = (float)acc;
        }
        return;
    }
    b -= cin;
    if (b < dwj.nblk) wgrad_dw_block<256>(dwj, (unsigned)b, prep_sh);
}
extern "C" long gspn_pooltop_scratch_floats(long rows, int cin, int ctop) {
    if (rows <= 0 || cin <= 0 || ctop <= 0) return GSPN_ERR_ARG;
    return (long)cin * cin + cin + 16;
}
// Pass B of a pooled top layer (pool groups of 32 rows) without reading the layer's (rows, ctop) output -- see BwdPool.  a: the layer's
// gspn_dy_args (dPool, pool_arg, ns = 32, cA/cB/cC; Y only for the dW job's plan); pooled: the (rows/32, ctop) pooled output of the
// forward pass; scratch: gspn_pooltop_scratch_floats floats.  The dW reduction of the layer rides in the small-operand launch; the previous
// layer's BN reductions come out of the GEMM's epilogue (part / nparts_out as for gspn_mlp_bwd_data_ex, both required).
// GSPN_ERR_UNSUPPORTED outside cin <= 64 (multiple of 4), ctop <= 128 (multiple of 4), ns = 32, 16-byte aligned operands.
extern "C" int gspn_mlp_bwd_data_pooltop(long rows, int cin, int ctop, const gspn_dy_args* a, const float* W, const float* bias, const float* pooled,
                                         float* scratch, float* dX, int ldx,
                                         const float* X, int ldx_in, const float* var, const float* gamma, float eps, int use_bn, int is_training,
                                         const float* work, float* dW,
                                         const float* Yp, int ldyp, const float* scale_p, const float* shift_p, const float* mean_p, const float* var_p,
                                         float eps_p, float* part, int* nparts_out, void* stream) {
    if (rows <= 0 || cin <= 0 || ctop <= 0 || !a || !a->dPool || !a->pool_arg || !a->cA || !a->cB || !a->cC || !W || !pooled || !scratch || !dX ||
        ldx < cin || !work || !dW || !part || !Yp || ldyp < cin || !scale_p || !shift_p || !mean_p || !var_p || !nparts_out)
        return GSPN_ERR_ARG;
    if ((X && ldx_in < cin) || (use_bn && !var)) return GSPN_ERR_ARG;
    if (a->ns != 32 || (rows & 31) || (cin & 3) || cin > 64 || (ctop & 3) || ctop > 128 || rows >= (1L << 31) || rows * (long)ldx >= (1L << 31) ||
        rows * (long)ldyp >= (1L << 31) || !vec_ok(Yp, ldyp) || ((uintptr_t)scratch % 16))
        return GSPN_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int BNs = cin <= 32 ? 32 : 64;
    const int QX = cin / 4;
    const long extra4 = 4L * ctop * BNs + 2L * 4 * ctop * 8, extra2 = 4L * ctop * BNs + 2L * 2 * ctop * 8;
    const long lds4 = 16L * QX * (BNs + 64 * 4) + 32L * QX + extra4, lds2 = 16L * QX * (BNs + 64 * 2) + 32L * QX + extra2;
    int trg = 0;
    if (lds4 <= 53 * 1024 && 4 * ctop <= 512) trg = 4;
    else if (BNs >= 64 && lds2 <= 80 * 1024 && 2 * ctop <= 512) trg = 2;
    if (trg && (32 * trg * QX) % 64 != 0) trg = 0;
    if (trg && 32 * trg * QX / 64 > 32) trg = 0;
    if (!trg) return GSPN_ERR_UNSUPPORTED;
    float* M = scratch;
    float* cvec = M + (size_t)cin * cin;
    bool use_stream;
    const WgradPlan p = wgrad_choose(rows, cin, ctop, a, X, ldx_in, &use_stream);
    const char* wb = reinterpret_cast<const char*>(work);
    const DwJob j = dw_job(rows, cin, ctop, p.nslots, reinterpret_cast<const float*>(wb + ws_off_pp(p.nch, cin, ctop)), reinterpret_cast<const double*>(wb),
                           reinterpret_cast<const float*>(wb + ws_off_g3(ctop)), var, gamma, eps, use_bn, is_training, dW);
    hipLaunchKernelGGL(pooltop_prep_kernel, dim3((unsigned)(cin + j.nblk)), dim3(256), 0, st, cin, ctop, W, bias, a->cB, a->cC, M, cvec, j);
    const size_t dyn = (size_t)(trg == 4 ? lds4 : lds2);
    const long ntiles = (rows + 32 * trg - 1) / (32 * trg);
    long bpc = (160L * 1024) / (long)(dyn + 2 * trg * BNs * 4 + 512);
    if (bpc > 4) bpc = 4;
    if (bpc < 1) bpc = 1;
    long gx = (long)GSPN_PLAN_CUS * bpc;
    if (gx > ntiles) gx = ntiles;
    const BwdPool bp{a->pool_arg, a->dPool, pooled, a->cA, W, ctop, mean_p, var_p, eps_p};
#define BWDP_GO(BN_, TRG_)                                                                                                             \
    do {                                                                                                                               \
        static bool attr_done = false;                                                                                                 \
        if (!attr_done) {                                                                                                              \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fwd_stream_kernel<BN_, TRG_, false, true>),          \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);                                 \
            if (e != hipSuccess) return (int)e;                                                                                        \
            attr_done = true;                                                                                                          \
        }                                                                                                                              \
        hipLaunchKernelGGL((mlp_fwd_stream_kernel<BN_, TRG_, false, true>), dim3((unsigned)gx, 1), dim3(256), dyn, st, (int)rows, cin, cin, Yp, ldyp, \
                           scale_p, shift_p, M, cvec, dX, ldx, part, (int)gx, PoolOut{nullptr, nullptr}, GatherSrc{}, bp);              \
    } while (0)
    if (BNs == 32 && trg == 4) BWDP_GO(32, 4);
    else if (BNs == 64 && trg == 4) BWDP_GO(64, 4);
    else BWDP_GO(64, 2);
#undef BWDP_GO
    *nparts_out = (int)gx;
    return gspn_launch_status();
}
// Pass B with both options: the fused dW reduction of gspn_mlp_bwd_data_dw (work != NULL) and the previous layer's BN reductions in the
// epilogue (part != NULL: Yp (rows, ldyp) = that layer's pre-BN output = this layer's input before activation; scale_p/shift_p its
// forward scale/shift; mean_p/var_p its batch statistics; part = gspn_rsum_part_floats(rows, cin) floats; *nparts_out = rows of partials
// written, for gspn_mlp_bwd_coef).  The reductions need the whole row: col0 = 0, ncols = cin.
extern "C" int gspn_mlp_bwd_data_ex(long rows, int cin, int cout, const gspn_dy_args* a, const float* W, int col0, int ncols, float* dX, int ldx,
                                    const float* X, int ldx_in, const float* var, const float* gamma, float eps, int use_bn, int is_training,
                                    const float* work, float* dW,
                                    const float* Yp, int ldyp, const float* scale_p, const float* shift_p, const float* mean_p, const float* var_p,
                                    float eps_p, float* part, int* nparts_out, void* stream) {
    const int rc = bwd_data_check(rows, cin, cout, a, col0, ncols, ldx);
    if (rc) return rc;
    if (rows <= 0) return GSPN_ERR_ARG;
    DwJob j;
    const DwJob* jp = nullptr;
    if (work) {
        if (!dW || (X && ldx_in < cin) || (use_bn && !var)) return GSPN_ERR_ARG;
        bool use_stream;
        const WgradPlan p = wgrad_choose(rows, cin, cout, a, X, ldx_in, &use_stream);
        const char* wb = reinterpret_cast<const char*>(work);
        j = dw_job(rows, cin, cout, p.nslots, reinterpret_cast<const float*>(wb + ws_off_pp(p.nch, cin, cout)), reinterpret_cast<const double*>(wb),
                   reinterpret_cast<const float*>(wb + ws_off_g3(cout)), var, gamma, eps, use_bn, is_training, dW);
        jp = &j;
    }
    RsumArgs rs{nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr};
    if (part) {
        if (!Yp || ldyp < cin || !scale_p || !shift_p || !mean_p || !var_p || !nparts_out || col0 != 0 || ncols != cin) return GSPN_ERR_ARG;
        rs = RsumArgs{Yp, ldyp, scale_p, shift_p, mean_p, var_p, eps_p, part};
    }
    return bwd_data_launch(rows, cin, cout, a, W, col0, ncols, dX, ldx, jp, (hipStream_t)stream, part ? &rs : nullptr, nparts_out);
}
extern "C" int gspn_mlp_bwd_data(long rows, int cin, int cout, const gspn_dy_args* a, const float* W, float* dX, int ldx, void* stream) {
    return gspn_mlp_bwd_data_cols(rows, cin, cout, a, W, 0, cin, dX, ldx, stream);
}
