// mlp_short.hip -- the shared-MLP GEMMs of the SHORT layers (4096-32768 rows x 64-384 channels: SA3, FP1, FP2 of the benchmark stack;
// utils/pointnet_util.py:109-113,165-169, models/model_rpointnet.py:226-230).
//
// What was wrong with running them on the long layers' kernels (r04, profiles/r04_mlp_layer_table.txt: 559 us for 15 % of the flops):
// a 128-row x 32-column tile per workgroup makes 256-1024 workgroups of ONE dependent chain each -- fetch a 32-wide K chunk, barrier,
// transpose into LDS, barrier, 16 MFMAs -- 12 times over for K = 384 with a one-chunk prefetch: every chunk pays a full L2 / Infinity
// Cache round trip that 0.45 us of matrix work cannot cover, and at one workgroup per CU nothing else is resident to cover it either.
//
// Here (r05) the contraction is split over the FOUR WAVES of a workgroup instead (intra-workgroup split-K):
//   * a workgroup owns a (32 MT) x (32 NT) output tile; wave w contracts k in [w K/4, (w+1) K/4) for the whole tile -- no operand is
//     shared between waves, so nothing is staged through LDS and the main loop has no barrier;
//   * v_mfma_f32_32x32x2_f32 takes A[row = l & 31][slot = l >> 5] and B[slot][col = l & 31]; which k a (slot, step) pair carries is free as
//     long as A and B agree, so half h of a wave takes the contiguous range k0 + h KW/2 .. ;
//   * B (the weights) goes from global memory straight into the lanes that multiply it: one NT-wide vector load per k step, 128 NT bytes
//     contiguous per half-wave (lane l owns columns NT l .. NT l + NT - 1 of the column block, a permutation the epilogue's vector
//     stores undo for free); issued first, in flight while A is staged;
//   * A (the activations) is staged ONCE per tile for its whole K: coalesced float4 loads (8 lanes per 128-byte row segment), the
//     previous layer's BN + ReLU applied once per element, written row-major into LDS with pitch K + 4 floats (== 4 mod 32: the
//     ds_read_b128 of 8 consecutive rows cover the 32 banks), one barrier; every wave then reads ITS k range as float4s.
//     (First cut, measured and dropped: A straight from global memory too, lane = row, KW/2 consecutive floats each -- every lane of a
//     load instruction touches its own cache line, 16 bytes of 128, and the vector L1's tag rate then costs more than the MFMAs:
//     4096 x 384 -> 256 18.1 us against 13.5 for the round-4 kernel; tools/r05_short_ab.sh.)
//   * the four partial tiles meet in LDS once: every wave writes its accumulators, and wave q sums (((p0 + p1) + p2) + p3, a fixed
//     order) the four accumulator registers 4q..4q+3 of every MFMA tile -- rows 8q + {0..3} + 4 (l >> 5) -- adds the bias, stores,
//     and keeps the column sums / the 32-row pool extrema the long kernels' epilogues keep.
// 4-8x the workgroups of the old decomposition (4 waves per SIMD resident: latency is covered by occupancy as well), K/4-long chains,
// ~3 vector instructions per MFMA instead of 9.  Same contract as the long kernels (Y, per-workgroup partial column sums, optional
// pool epilogue); the summation order over k differs from theirs (four k-parts), results agree to rounding and are deterministic.
#include "mlp_common.h"
#include <type_traits>
#include <algorithm>

namespace gspn_k {

template <int N> struct VecF;
template <> struct VecF<1> { typedef float type; };
template <> struct VecF<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct VecF<4> { typedef float type __attribute__((ext_vector_type(4))); };

template <int NT> __device__ __forceinline__ float vget(const typename VecF<NT>::type& v, int i) {
    if constexpr (NT == 1) return v; else return v[i];
}

// the four partial accumulator sets of a workgroup meet here: R = MT*NT*16 registers per wave
template <int R>
__device__ __forceinline__ void park_partials(float* sAcc, int wave, int lane, const f32x16* acc, int ntile) {
#pragma unroll
    for (int i = 0; i < R / 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sAcc[((wave * R) + i * 16 + r) * 64 + lane] = acc[i][r];
}
// register r of tile i, summed over the four waves' partials in the fixed order ((p0 + p1) + p2) + p3
template <int R>
__device__ __forceinline__ float sum_partials(const float* sAcc, int i, int r, int lane) {
    const float* p = sAcc + (i * 16 + r) * 64 + lane;
    return ((p[0] + p[R * 64]) + p[2 * R * 64]) + p[3 * R * 64];
}

// ============================================================================================
// Forward:  Y = act(X) . W + bias,  act = relu(x*scale + shift) of the previous layer's BN (two roundings, as act1) or identity.
// grid (nparts, cout / (32 NT)); workgroup bx takes row tiles bx, bx + nparts, ... (one, for the shapes this is launched on) and writes
// partial-statistics row bx (zeros if it has no tile).
// ============================================================================================
template <int KW, int MT, int NT, bool ACT, bool POOL>
__global__ __launch_bounds__(256) void fwd_short_kernel(int rows, int cin, int cout, const float* __restrict__ X, int ldx, const float* __restrict__ in_scale,
                                                        const float* __restrict__ in_shift, const float* __restrict__ W, const float* __restrict__ bias,
                                                        float* __restrict__ Y, int ldy, float* __restrict__ stats, PoolOut po) {
    constexpr int KH = KW / 2, R = MT * NT * 16, BN = 32 * NT, K = 4 * KW;
    typedef typename VecF<NT>::type vecn;
    constexpr int SA = K + 4;                              // pitch of the staged A tile
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    float* sAcc = s_dyn;                                   // [4][R][64]
    float* sA = s_dyn + 4 * R * 64;                        // [32 MT][SA]
    float* sC = sA + 32 * MT * SA;                         // [2][K]: scale, shift of the input channels
    __shared__ float sRed[4][2][BN];
    __shared__ float sPv[POOL ? 4 * MT * BN : 1];
    __shared__ int sPi[POOL ? 4 * MT * BN : 1];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, kh = lane >> 5;
    const int n0 = blockIdx.y * BN;
    const int kb = wave * KW + kh * KH;                    // this lane's first k
    const int ntiles = rows / (32 * MT);
    // B operand: W rows kb .. kb + KH - 1, columns n0 + NT l31 .. (+NT): once per workgroup
    vecn b[KH];
    {
        const float* wp = W + (size_t)kb * cout + n0 + NT * l31;
#pragma unroll
        for (int s = 0; s < KH; ++s) b[s] = *reinterpret_cast<const vecn*>(wp + (size_t)s * cout);
    }
    const int ar = t >> 3, aq = (t & 7) * 4;               // staging: row ar (+ 32 per mt), floats aq + 32 i
    float4 xr[MT][K / 32];
    auto fetch = [&](int tile) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float* xp = X + (size_t)(tile * 32 * MT + mt * 32 + ar) * ldx + aq;
#pragma unroll
            for (int i = 0; i < K / 32; ++i) xr[mt][i] = *reinterpret_cast<const float4*>(xp + 32 * i);
        }
    };
    if ((int)blockIdx.x < ntiles) fetch((int)blockIdx.x);
    if constexpr (ACT) {
        for (int i = t; i < K; i += 256) { sC[i] = in_scale[i]; sC[K + i] = in_shift[i]; }
    }
    float bv[NT], csum[NT], csq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { bv[nt] = bias ? bias[n0 + NT * l31 + nt] : 0.f; csum[nt] = csq[nt] = 0.f; }
    __syncthreads();
    for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
        const int m0 = tile * 32 * MT;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < K / 32; ++i) {
                float4 v = xr[mt][i];
                if constexpr (ACT) {
                    const float4 q = *reinterpret_cast<const float4*>(sC + aq + 32 * i), h = *reinterpret_cast<const float4*>(sC + K + aq + 32 * i);
                    v.x = act1(v.x, true, q.x, h.x); v.y = act1(v.y, true, q.y, h.y); v.z = act1(v.z, true, q.z, h.z); v.w = act1(v.w, true, q.w, h.w);
                }
                *reinterpret_cast<float4*>(sA + (mt * 32 + ar) * SA + aq + 32 * i) = v;
            }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + (int)gridDim.x);       // the next tile's rows, in flight under this tile's MFMAs
        f32x16 acc[MT * NT];
#pragma unroll
        for (int i = 0; i < MT * NT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int i = 0; i < KH / 4; ++i) {
            float4 a4[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a4[mt] = *reinterpret_cast<const float4*>(sA + (mt * 32 + l31) * SA + kb + 4 * i);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float av = j == 0 ? a4[mt].x : (j == 1 ? a4[mt].y : (j == 2 ? a4[mt].z : a4[mt].w));
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt * NT + nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, vget<NT>(b[4 * i + j], nt), acc[mt * NT + nt], 0, 0, 0);
                }
            }
        }
        park_partials<R>(sAcc, wave, lane, acc, MT * NT);
        __syncthreads();
        // wave q finishes registers 4q..4q+3 of every tile: rows m0 + 32 mt + 8q + j + 4 kh, this lane's NT columns
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float v[4][NT];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    v[j][nt] = sum_partials<R>(sAcc, mt * NT + nt, 4 * wave + j, lane) + bv[nt];
                    csum[nt] += v[j][nt];
                    csq[nt] = __builtin_fmaf(v[j][nt], v[j][nt], csq[nt]);
                }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                vecn o;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { if constexpr (NT == 1) o = v[j][0]; else o[nt] = v[j][nt]; }
                *reinterpret_cast<vecn*>(Y + (size_t)(m0 + mt * 32 + 8 * wave + j + 4 * kh) * ldy + n0 + NT * l31) = o;
            }
            if constexpr (POOL) {
                // the group's maximum and the row that reaches it first, among this wave's 8 rows (ascending: j, then the half-wave)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float mx = v[0][nt];
                    int ir = 0;
#pragma unroll
                    for (int j = 1; j < 4; ++j) if (v[j][nt] > mx) { mx = v[j][nt]; ir = j; }
                    int row = 8 * wave + ir + 4 * kh;
                    const float omx = __shfl_xor(mx, 32, 64);
                    const int orow = __shfl_xor(row, 32, 64);
                    if (omx > mx || (omx == mx && orow < row)) { mx = omx; row = orow; }
                    if (lane < 32) { sPv[(wave * MT + mt) * BN + NT * l31 + nt] = mx; sPi[(wave * MT + mt) * BN + NT * l31 + nt] = row; }
                }
            }
        }
        __syncthreads();                                   // sAcc is free again; the pool candidates are in place
        if constexpr (POOL) {
            for (int j = t; j < MT * BN; j += 256) {
                const int mt = j / BN, col = j % BN;
                float mx = sPv[(0 * MT + mt) * BN + col];
                int row = sPi[(0 * MT + mt) * BN + col];
#pragma unroll
                for (int q = 1; q < 4; ++q) {
                    const float o = sPv[(q * MT + mt) * BN + col];
                    if (o > mx) { mx = o; row = sPi[(q * MT + mt) * BN + col]; }      // strict: the lower rows (lower q) keep ties
                }
                const size_t at = (size_t)((m0 >> 5) + mt) * cout + n0 + col;
                po.vmax[at] = mx;
                po.amax[at] = row;
            }
            if (tile + (int)gridDim.x < ntiles) __syncthreads();
        }
    }
    if (stats) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            csum[nt] += __shfl_xor(csum[nt], 32, 64);
            csq[nt] += __shfl_xor(csq[nt], 32, 64);
            if (lane < 32) { sRed[wave][0][NT * l31 + nt] = csum[nt]; sRed[wave][1][NT * l31 + nt] = csq[nt]; }
        }
        __syncthreads();
        float* ws = stats + (size_t)blockIdx.x * 2 * cout;
        for (int j = t; j < BN; j += 256) {
            ws[n0 + j] = ((sRed[0][0][j] + sRed[1][0][j]) + sRed[2][0][j]) + sRed[3][0][j];
            ws[cout + n0 + j] = ((sRed[0][1][j] + sRed[1][1][j]) + sRed[2][1][j]) + sRed[3][1][j];
        }
    }
}

// false: the instance could not be configured on this device (the attribute call failed) -- nothing was launched, the caller goes on to the general kernels
template <int KW, int MT, int NT, bool ACT, bool POOL>
static bool fwd_short_launch(dim3 g, hipStream_t st, int rows, int cin, int cout, const float* X, int ldx, const float* in_scale, const float* in_shift,
                             const float* W, const float* bias, float* Y, int ldy, float* stats, PoolOut po) {
    constexpr size_t dyn = sizeof(float) * (4 * MT * NT * 16 * 64 + 32 * MT * (4 * KW + 4) + 2 * 4 * KW);
    if constexpr (dyn > 48 * 1024) {
        // the attribute belongs to (function, DEVICE): one flag per device, not per process; a failure is reported, not swallowed
        static bool done[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = -1;
        if (dev < 0 || dev >= 64 || !done[dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_short_kernel<KW, MT, NT, ACT, POOL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) {
                (void)hipGetLastError();
                return false;
            }
            if (dev >= 0 && dev < 64) done[dev] = true;
        }
    }
    hipLaunchKernelGGL((fwd_short_kernel<KW, MT, NT, ACT, POOL>), g, dim3(256), dyn, st, rows, cin, cout, X, ldx, in_scale, in_shift, W, bias, Y, ldy, stats, po);
    return true;
}

}  // namespace gspn_k
using namespace gspn_k;

// Tile shape per layer shape: enough workgroups to give every SIMD several waves (>= ~4 x 224 CUs), MFMA tiles per k step as large as that
// allows (an A value feeds NT MFMAs, a B vector MT).  GSPN_FWD_SHORT_MT / _NT force a shape (tools/short_sweep.py), GSPN_FWD_SHORT=0 turns
// the path off (the long layers' kernels then run these layers as in round 4).
bool gspn_fwd_short_go(long rows, int cin, int cout, const float* X, int ldx, const float* in_scale, const float* in_shift, const float* W, const float* bias,
                       float* Y, int ldy, float* stats, unsigned nparts, PoolOut po, hipStream_t st) {
    static const int on = env_int("GSPN_FWD_SHORT", 1);
    static const int f_mt = env_int("GSPN_FWD_SHORT_MT", 0), f_nt = env_int("GSPN_FWD_SHORT_NT", 0);
    if (!on || rows > GSPN_SHORT_ROWS || rows < 64 || (rows & 63) || !vec_ok(X, ldx) || !vec_ok(W, cout) || (ldy & 3) || (((uintptr_t)Y) & 15)) return false;
    if (!(cin == 64 || cin == 128 || cin == 192 || cin == 256 || cin == 384) || (cout & 31)) return false;
    if (rows * (long)std::max(ldx, ldy) >= (1L << 30) || (long)cin * cout >= (1L << 28)) return false;
    if (nparts != (unsigned)short_fwd_parts(rows)) return false;
    // workgroups wanted: ~4 per CU; 32 x 32 tiles unless that makes more than ~2048 of them
    const int kw = cin / 4;
    int mt = 1, nt = 1;
    const long w11 = (rows / 32) * (cout / 32);
    if (w11 > 2048 && !(cout & 63)) nt = 2;
    if (w11 / nt > 2048) mt = 2;
    if (w11 / (nt * mt) > 4096 && kw <= 32 && !(cout & 127)) { nt = 4; mt = 1; }
    if (f_mt == 1 || f_mt == 2) mt = f_mt;
    if ((f_nt == 1 || f_nt == 2 || f_nt == 4) && cout % (32 * f_nt) == 0) nt = f_nt;
    // register budget (KW/2 x (MT + NT) operand registers + 16 MT NT accumulators): the shapes instantiated
    if (nt == 4 && (kw > 32 || mt == 2)) nt = 2;
    if (nt == 4 && mt == 1) { /* (1,4) */ }
    if (mt == 2 && kw > 48) mt = 1;                                        // the staged A tile: 32 MT x (K + 4) floats of LDS
    const dim3 g(nparts, cout / (32 * nt));
    const int irows = (int)rows;
    bool ok = true;
#define FS_GO(KW_, MT_, NT_, A_, P_) ok = fwd_short_launch<KW_, MT_, NT_, A_, P_>(g, st, irows, cin, cout, X, ldx, in_scale, in_shift, W, bias, Y, ldy, stats, po)
#define FS_P(KW_, MT_, NT_, A_) do { if (po.vmax) FS_GO(KW_, MT_, NT_, A_, true); else FS_GO(KW_, MT_, NT_, A_, false); } while (0)
#define FS_A(KW_, MT_, NT_) do { if (in_scale) FS_P(KW_, MT_, NT_, true); else FS_P(KW_, MT_, NT_, false); } while (0)
#define FS_T2(KW_) do { if (nt == 1) FS_A(KW_, 1, 1); else FS_A(KW_, 1, 2); } while (0)
#define FS_T4(KW_) do { if (mt == 2 && nt == 2) FS_A(KW_, 2, 2); else if (mt == 2) FS_A(KW_, 2, 1); else FS_T2(KW_); } while (0)
#define FS_T5(KW_) do { if (nt == 4) FS_A(KW_, 1, 4); else FS_T4(KW_); } while (0)
    switch (kw) {
        case 16: FS_T5(16); break;
        case 32: FS_T5(32); break;
        case 48: FS_T4(48); break;
        case 64: FS_T2(64); break;
        default: FS_T2(96); break;
    }
#undef FS_T5
#undef FS_T4
#undef FS_T2
#undef FS_A
#undef FS_P
#undef FS_GO
    return ok;
}
