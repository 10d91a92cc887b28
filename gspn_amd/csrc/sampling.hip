// sampling.hip -- tf_ops/sampling on gfx950: farthest point sampling, gather_point (+grad),
// prob_sample.  Reference semantics: tf_ops/sampling/tf_sampling_g.cu (cited per kernel).
#include "common.h"

// ============================================================================================
// Farthest point sampling (reference: tf_sampling_g.cu:105-170)
//
// The reference is m-1 strictly serial rounds of {update the min-dist of all n points, arg-max}.
// It re-reads xyz + the min-dist scratch from L2/global every round (20 B/point/round) and
// spends 10 block barriers per round.  On MI355X a whole 32768-point scene fits ON CHIP in one
// CU, so the resident kernel keeps it there for all rounds (zero HBM/L2 traffic in the loop):
//
//   * one 1024-thread workgroup (16 waves, 4 per SIMD) per scene.  Measured on gfx950
//     (tools/valu_probe.hip): one wave issues at most one VALU op per ~5 cycles, a SIMD reaches
//     ~1.27 cycles/op only with 4 waves, and v_pk_*_f32 costs the same issue slot as a scalar
//     op -- hence 16 waves and packed-fp32 math (two points per instruction);
//   * thread t = 2*rho + half owns the points k = (half*P + p)*512 + rho, p = 0..P-1: all of
//     one residue class k mod 512 sits in two adjacent lanes in ascending k, so the reference
//     tie order (d desc, k mod 512 asc, k asc) is simply (d desc, t asc, p asc);
//   * x, y and the running min-dist live in VGPRs (3*P registers); z lives in VGPRs too for
//     P <= 16 and in LDS (128 KiB, read-only, ds_read_b128) for P = 32, which is what lets
//     32768 points fit: 384 KiB of registers + 128 KiB of LDS;
//   * arg-max = integer max on the float bit patterns (all candidates are >= +0, padding is
//     -1.0f, so signed-int order == float order): DPP row reductions inside a wave, one LDS hop
//     across the 16 waves, ONE workgroup barrier per round (double-buffered candidates);
//   * the slot of the maximum is NOT tracked in the hot loop (2 more VALU per point); each wave
//     resolves it afterwards for its best lane only, through per-8-point group maxima that the
//     max3 tree yields for free, wave-uniform switches and v_readlane.
// ============================================================================================

#define FPS_T 1024
#define FPS_W (FPS_T / 64)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define NEG_ONE_BITS ((int)0xBF800000)
#ifdef FPS_PROFILE
__device__ long long g_fps_prof[8];
__device__ long long g_fps_tl[16 * 8];
#define FPS_TICK(i) do { const long long _n = clock64(); prof[i] += _n - tprev; tprev = _n; if (j == 100 && blockIdx.x == 0 && lane == 0) g_fps_tl[wave * 8 + i] = _n; } while (0)
#else
#define FPS_TICK(i) do {} while (0)
#endif

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

// P = points per thread (even), ZLDS = z plane in LDS instead of VGPRs
template <int P, bool ZLDS>
__global__ __launch_bounds__(FPS_T) void fps_resident_kernel(int n, int m, const float* __restrict__ inp, int* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [0,512)    : candidates, 2 buffers x 16 waves x int4 {max bits, x, y, z}
    // [512,640)  : candidate indices, 2 buffers x 16 waves x int
    // [1024,..)  : z plane, float4 [P/4][1024]   (ZLDS only)
    int4* s_cand = reinterpret_cast<int4*>(smem);
    int* s_k = reinterpret_cast<int*>(smem + 512);
    v4f* s_z = reinterpret_cast<v4f*>(smem + 1024);

    constexpr int G = FpsGroup<P>::G;
    constexpr int NG = FpsGroup<P>::NG;
    static_assert(P % 2 == 0 && (!ZLDS || P % 4 == 0), "P must be even (multiple of 4 with ZLDS)");

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int rho = t >> 1;                       // residue class k mod 512 of this thread
    const int kbase = (t & 1) * P * 512 + rho;    // slot p holds point kbase + p*512
    const float* xyz = inp + (size_t)blockIdx.x * n * 3;
    int* o = out + (size_t)blockIdx.x * m;

    v2f x[P / 2], y[P / 2], z[ZLDS ? 1 : P / 2], td[P / 2];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int k = kbase + p * 512;
        float px = 0.f, py = 0.f, pz = 0.f;
        float d0 = -1.0f;                         // padding never wins (real candidates are >= 0)
        if (k < n) {
            px = xyz[k * 3 + 0];
            py = xyz[k * 3 + 1];
            pz = xyz[k * 3 + 2];
            d0 = 1e38f;                           // :117-119
        }
        // detach x/y/z from the dwordx3 load tuple so the allocator may place them independently
        asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));
        x[p >> 1][p & 1] = px;
        y[p >> 1][p & 1] = py;
        td[p >> 1][p & 1] = d0;
        if (ZLDS) reinterpret_cast<float*>(s_z)[((p >> 2) * FPS_T + t) * 4 + (p & 3)] = pz;
        else z[p >> 1][p & 1] = pz;
    }
    if (t == 0) o[0] = 0;                          // :114-116
    float cx = xyz[0], cy = xyz[1], cz = xyz[2];   // centre of round 1 = point 0
    __syncthreads();

#ifdef FPS_PROFILE
    long long prof[5] = {0, 0, 0, 0, 0};
    long long tprev = clock64();
#endif
    for (int j = 1; j < m; ++j) {
        // ---- update min-dist + per-group maxima (value only) ----
        int g[NG];
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            int gm = NEG_ONE_BITS;
            v2f zhold = {0.f, 0.f};
#pragma unroll
            for (int h = 0; h < G / 2; ++h) {
                const int pp = (q * G) / 2 + h;            // pair index
                v2f zz;
                if constexpr (ZLDS) {
                    // one ds_read_b128 serves 4 consecutive slots; the empty asm keeps the compiler
                    // from narrowing it back into scalar LDS reads
                    if ((h & 1) == 0) {
                        v4f z4 = s_z[(pp >> 1) * FPS_T + t];
                        asm("" : "+v"(z4));
                        zz = z4.xy;
                        zhold = z4.zw;
                    } else {
                        zz = zhold;
                    }
                } else {
                    zz = z[pp];
                }
                const v2f dx = x[pp] - cx, dy = y[pp] - cy, dz = zz - cz;
                v2f d = dy * dy;                                   // dist2_cuda, GSPN_DIST_POLICY 2, :142
                d = __builtin_elementwise_fma(dx, dx, d);
                d = __builtin_elementwise_fma(dz, dz, d);
                td[pp][0] = vmin_f32(d[0], td[pp][0]);             // :143
                td[pp][1] = vmin_f32(d[1], td[pp][1]);
                gm = vmax3_i32(gm, __float_as_int(td[pp][0]), __float_as_int(td[pp][1]));
            }
            g[q] = gm;
            __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from interleaving groups (VGPR pressure)
        }
        int best = g[0];
#pragma unroll
        for (int q = 1; q < NG; ++q) best = max(best, g[q]);
        FPS_TICK(0);

        // ---- wave arg-max: value -> lowest lane holding it -> lowest slot inside that lane ----
        int qsel = NG - 1;                         // per lane: first group that holds the lane's best
#pragma unroll
        for (int q = NG - 2; q >= 0; --q) qsel = (g[q] == best) ? q : qsel;
        const int wmax = wave_max_i32(best);
        const int lw = __builtin_ctzll(__ballot(best == wmax));
        const int qw = __builtin_amdgcn_readlane(qsel, lw);
        int isel = G - 1;                          // per lane: first slot of group qw equal to the lane's best
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            if (qw == q) {                         // wave-uniform: static register indices inside
#pragma unroll
                for (int i = G - 2; i >= 0; --i) {
                    const int p = q * G + i;
                    isel = (__float_as_int(td[p >> 1][p & 1]) == best) ? i : isel;
                }
            }
        }
        const int iw = __builtin_amdgcn_readlane(isel, lw);
        float fx = 0.f, fy = 0.f, fz = 0.f;
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            if (qw == q) {
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    if (iw == i) {
                        const int p = q * G + i;
                        fx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[p >> 1][p & 1]), lw));
                        fy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y[p >> 1][p & 1]), lw));
                        if (!ZLDS) fz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z[ZLDS ? 0 : (p >> 1)][p & 1]), lw));
                    }
                }
            }
        }
        const int fp = qw * G + iw;
        const int tw = wave * 64 + lw;
        if (ZLDS) fz = reinterpret_cast<const float*>(s_z)[((fp >> 2) * FPS_T + tw) * 4 + (fp & 3)];
        FPS_TICK(1);
        const int buf = (j & 1) * FPS_W;
        if (lane == 0) {
            s_cand[buf + wave] = make_int4(wmax, __float_as_int(fx), __float_as_int(fy), __float_as_int(fz));
            s_k[buf + wave] = ((tw & 1) * P + fp) * 512 + (tw >> 1);
        }
        __syncthreads();
        FPS_TICK(2);

        // ---- workgroup arg-max: every wave reduces the 16 candidates redundantly (lowest wave wins ties)
        const int4 cand = s_cand[buf + (lane & (FPS_W - 1))];
        const int ck = s_k[buf + (lane & (FPS_W - 1))];
        const int M = __builtin_amdgcn_readfirstlane(row_max_i32(cand.x));
        const int wbest = __builtin_ctz((unsigned)(__ballot(cand.x == M) & 0xFFFFull));
        cx = __int_as_float(__builtin_amdgcn_readlane(cand.y, wbest));
        cy = __int_as_float(__builtin_amdgcn_readlane(cand.z, wbest));
        cz = __int_as_float(__builtin_amdgcn_readlane(cand.w, wbest));
        // cross-lane reads stay OUTSIDE the divergent store (a readlane sunk under `if (t == 0)` would
        // let the compiler load s_k for lane 0 only)
        const int kbest = __builtin_amdgcn_readlane(ck, wbest);
        if (t == 0) o[j] = kbest;                                  // :166-168
        FPS_TICK(3);
    }
#ifdef FPS_PROFILE
    if (blockIdx.x == 0 && (t == 0 || t == FPS_T - 64))
        for (int i = 0; i < 4; ++i) g_fps_prof[i + (t ? 4 : 0)] = prof[i];
#endif
}

// Fallback for scenes that do not fit one CU (n > 32768): one 1024-thread workgroup per scene
// slot, min-dist in the caller's scratch (L2 resident), 64-bit (dist, tie-rank) keys.
// Thread t visits k = t, t+1024, ... so k mod 512 == t mod 512 for all of its points.
__global__ __launch_bounds__(1024) void fps_streaming_kernel(int b, int n, int m, const float* __restrict__ inp,
                                                             float* __restrict__ temp, int* __restrict__ out) {
    __shared__ unsigned long long s_key[16];
    __shared__ int s_old;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = blockIdx.x; i < b; i += gridDim.x) {
        const float* xyz = inp + (size_t)i * n * 3;
        float* td = temp + (size_t)blockIdx.x * n;
        int* o = out + (size_t)i * m;
        for (int k = t; k < n; k += 1024) td[k] = 1e38f;
        if (t == 0) o[0] = 0;
        int old = 0;
        __syncthreads();
        for (int j = 1; j < m; ++j) {
            const float cx = xyz[old * 3 + 0], cy = xyz[old * 3 + 1], cz = xyz[old * 3 + 2];
            unsigned long long key = 0;   // (dist bits << 32) | ~rank ; larger is better
            for (int k = t; k < n; k += 1024) {
                const float d = dist2_cuda(xyz[k * 3 + 0] - cx, xyz[k * 3 + 1] - cy, xyz[k * 3 + 2] - cz);
                const float o0 = td[k];
                const float nt = __builtin_fminf(d, o0);
                if (nt != o0) td[k] = nt;
                const unsigned rank = ((unsigned)(k & 511) << 22) | (unsigned)(k >> 9);
                const unsigned long long kk = ((unsigned long long)(unsigned)__float_as_int(nt) << 32) | (unsigned)(~rank);
                key = kk > key ? kk : key;
            }
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) {
                const unsigned long long other = __shfl_xor(key, s, 64);
                key = other > key ? other : key;
            }
            if (lane == 0) s_key[wave] = key;
            __syncthreads();
            if (wave == 0) {
                unsigned long long v = s_key[lane & 15];
#pragma unroll
                for (int s = 8; s >= 1; s >>= 1) {
                    const unsigned long long other = __shfl_xor(v, s, 64);
                    v = other > v ? other : v;
                }
                if (lane == 0) {
                    const unsigned rank = ~(unsigned)(v & 0xFFFFFFFFull);
                    const int k = (int)(((rank & 0x3FFFFFu) << 9) | (rank >> 22));
                    s_old = k;
                    o[j] = k;
                }
            }
            __syncthreads();
            old = s_old;
        }
        __syncthreads();
    }
}

template <int P, bool ZLDS>
static int launch_fps_resident(int b, int n, int m, const float* inp, int* out, hipStream_t st) {
    const size_t lds = 1024 + (ZLDS ? (size_t)P * FPS_T * sizeof(float) : 0);
    if (ZLDS) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_resident_kernel<P, ZLDS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((fps_resident_kernel<P, ZLDS>), dim3(b), dim3(FPS_T), lds, st, n, m, inp, out);
    return gspn_launch_status();
}

extern "C" int gspn_farthestpointsampling(int b, int n, int m, const float* inp, float* temp, int* out, void* stream) {
    if (b < 0 || n <= 0 || m <= 0) return GSPN_ERR_ARG;          // tf_sampling.cpp:99,105
    if (b == 0) return 0;
    if (!inp || !out) return GSPN_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (n <= 2048) return launch_fps_resident<2, false>(b, n, m, inp, out, st);
    if (n <= 4096) return launch_fps_resident<4, false>(b, n, m, inp, out, st);
    if (n <= 8192) return launch_fps_resident<8, false>(b, n, m, inp, out, st);
    if (n <= 16384) return launch_fps_resident<16, false>(b, n, m, inp, out, st);
    if (n <= GSPN_FPS_RESIDENT_MAX) return launch_fps_resident<32, true>(b, n, m, inp, out, st);
    if (!temp) return GSPN_ERR_ARG;
    if ((long long)n >= (1ll << 31) / 3) return GSPN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(fps_streaming_kernel, dim3(b < 32 ? b : 32), dim3(1024), 0, st, b, n, m, inp, temp, out);
    return gspn_launch_status();
}

// ============================================================================================
// gather_point / gather_point_grad   (tf_sampling_g.cu:172-192)
// One thread per output point; the three floats of a point are contiguous.
// ============================================================================================
__global__ void gatherpoint_kernel(int n, int m, const float* __restrict__ inp, const int* __restrict__ idx, float* __restrict__ out) {
    const int i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int a = idx[(size_t)i * m + j];
    const float* s = inp + ((size_t)i * n + a) * 3;
    float* d = out + ((size_t)i * m + j) * 3;
    d[0] = s[0];
    d[1] = s[1];
    d[2] = s[2];
}
__global__ void scatteraddpoint_kernel(int n, int m, const float* __restrict__ out_g, const int* __restrict__ idx, float* __restrict__ inp_g) {
    const int i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int a = idx[(size_t)i * m + j];
    const float* s = out_g + ((size_t)i * m + j) * 3;
    float* d = inp_g + ((size_t)i * n + a) * 3;
    atomicAdd(d + 0, s[0]);
    atomicAdd(d + 1, s[1]);
    atomicAdd(d + 2, s[2]);
}
extern "C" int gspn_gatherpoint(int b, int n, int m, const float* inp, const int* idx, float* out, void* stream) {
    if (b < 0 || n <= 0 || m < 0) return GSPN_ERR_ARG;
    if (b == 0 || m == 0) return 0;
    if (b > 65535) return GSPN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gatherpoint_kernel, dim3((m + 255) / 256, b), dim3(256), 0, (hipStream_t)stream, n, m, inp, idx, out);
    return gspn_launch_status();
}
extern "C" int gspn_scatteraddpoint(int b, int n, int m, const float* out_g, const int* idx, float* inp_g, void* stream) {
    if (b < 0 || n <= 0 || m < 0) return GSPN_ERR_ARG;
    if (b == 0) return 0;
    if (b > 65535) return GSPN_ERR_UNSUPPORTED;
    hipError_t e = hipMemsetAsync(inp_g, 0, sizeof(float) * (size_t)b * n * 3, (hipStream_t)stream);   // tf_sampling.cpp:174
    if (e != hipSuccess) return (int)e;
    if (m == 0) return 0;
    hipLaunchKernelGGL(scatteraddpoint_kernel, dim3((m + 255) / 256, b), dim3(256), 0, (hipStream_t)stream, n, m, out_g, idx, inp_g);
    return gspn_launch_status();
}

// ============================================================================================
// prob_sample (tf_sampling_g.cu:7-104): per-row inclusive prefix sum + inverse-CDF lookup.
// Not reached from the model; kept for API completeness.  One workgroup per row; the tile
// arithmetic follows the reference's 8192-element tiles (groups of 4, tree over group totals,
// compensated carry between tiles) so the cumulative sums round identically.
// ============================================================================================
#define CS_BLOCK 2048
#define CS_PAD 5
__global__ __launch_bounds__(512) void cumsum_kernel(int b, int n, const float* __restrict__ inp, float* __restrict__ out) {
    __shared__ float buffer4[CS_BLOCK * 4];
    __shared__ float buffer[CS_BLOCK + (CS_BLOCK >> CS_PAD)];
    for (int i = blockIdx.x; i < b; i += gridDim.x) {
        float runningsum = 0, runningsum2 = 0;
        for (int j = 0; j < n; j += CS_BLOCK * 4) {
            const int n24_i = min(n - j, CS_BLOCK * 4);
            const int n24 = (n24_i + 3) & ~3;
            const int n2 = n24 >> 2;
            for (int k = threadIdx.x * 4; k < n24_i; k += blockDim.x * 4) {
                if (k + 3 < n24_i) {
                    float v1 = inp[(size_t)i * n + j + k];
                    float v2 = inp[(size_t)i * n + j + k + 1];
                    v2 += v1;
                    float v3 = inp[(size_t)i * n + j + k + 2];
                    float v4 = inp[(size_t)i * n + j + k + 3];
                    v4 += v3;
                    v3 += v2;
                    v4 += v2;
                    buffer4[k] = v1; buffer4[k + 1] = v2; buffer4[k + 2] = v3; buffer4[k + 3] = v4;
                    buffer[(k >> 2) + (k >> (2 + CS_PAD))] = v4;
                } else {
                    float v = 0;
                    for (int k2 = k; k2 < n24_i; k2++) { v += inp[(size_t)i * n + j + k2]; buffer4[k2] = v; }
                    for (int k2 = n24_i; k2 < n24; k2++) buffer4[k2] = v;
                    buffer[(k >> 2) + (k >> (2 + CS_PAD))] = v;
                }
            }
            int u = 0;
            for (; (2 << u) <= n2; u++) {
                __syncthreads();
                for (int k = threadIdx.x; k < (n2 >> (u + 1)); k += blockDim.x) {
                    int i1 = (((k << 1) + 2) << u) - 1, i2 = (((k << 1) + 1) << u) - 1;
                    i1 += i1 >> CS_PAD; i2 += i2 >> CS_PAD;
                    buffer[i1] += buffer[i2];
                }
            }
            u--;
            for (; u >= 0; u--) {
                __syncthreads();
                for (int k = threadIdx.x; k < ((n2 - (1 << u)) >> (u + 1)); k += blockDim.x) {
                    int i1 = (((k << 1) + 3) << u) - 1, i2 = (((k << 1) + 2) << u) - 1;
                    i1 += i1 >> CS_PAD; i2 += i2 >> CS_PAD;
                    buffer[i1] += buffer[i2];
                }
            }
            __syncthreads();
            for (int k = threadIdx.x * 4; k < n24; k += blockDim.x * 4) {
                if (k != 0) {
                    const int k2 = ((k >> 2) - 1) + (((k >> 2) - 1) >> CS_PAD);
                    buffer4[k] += buffer[k2]; buffer4[k + 1] += buffer[k2]; buffer4[k + 2] += buffer[k2]; buffer4[k + 3] += buffer[k2];
                }
            }
            __syncthreads();
            for (int k = threadIdx.x; k < n24_i; k += blockDim.x) out[(size_t)i * n + j + k] = buffer4[k] + runningsum;
            const float tt = buffer[(n2 - 1) + ((n2 - 1) >> CS_PAD)] + runningsum2;
            const float r2 = runningsum + tt;
            runningsum2 = tt - (r2 - runningsum);
            runningsum = r2;
            __syncthreads();
        }
    }
}
__global__ void binarysearch_kernel(int b, int n, int m, const float* __restrict__ dataset, const float* __restrict__ query, int* __restrict__ result) {
    int base = 1;
    while (base < n) base <<= 1;
    const int i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const float* ds = dataset + (size_t)i * n;
    const float q = query[(size_t)i * m + j] * ds[n - 1];
    int r = n - 1;
    for (int k = base; k >= 1; k >>= 1)
        if (r >= k && ds[r - k] >= q) r -= k;
    result[(size_t)i * m + j] = r;
}
extern "C" int gspn_probsample(int b, int n, int m, const float* inp_p, const float* inp_r, float* temp, int* out, void* stream) {
    if (b < 0 || n <= 0 || m < 0) return GSPN_ERR_ARG;
    if (b == 0 || m == 0) return 0;
    if (b > 65535) return GSPN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(cumsum_kernel, dim3(b < 1024 ? b : 1024), dim3(512), 0, (hipStream_t)stream, b, n, inp_p, temp);
    hipLaunchKernelGGL(binarysearch_kernel, dim3((m + 255) / 256, b), dim3(256), 0, (hipStream_t)stream, b, n, m, temp, inp_r, out);
    return gspn_launch_status();
}

extern "C" int gspn_dist_policy(void) { return GSPN_DIST_POLICY; }
extern "C" int gspn_abi_version(void) { return 1; }
extern "C" int gspn_fill_zero(void* ptr, long bytes, void* stream) {
    if (bytes < 0) return GSPN_ERR_ARG;
    if (bytes == 0) return 0;
    hipError_t e = hipMemsetAsync(ptr, 0, (size_t)bytes, (hipStream_t)stream);
    return e == hipSuccess ? 0 : (int)e;
}
