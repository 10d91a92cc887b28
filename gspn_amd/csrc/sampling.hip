// sampling.hip -- tf_ops/sampling on gfx950: farthest point sampling, gather_point (+grad),
// prob_sample.  Reference semantics: tf_ops/sampling/tf_sampling_g.cu (cited per kernel).
#include "common.h"

// ============================================================================================
// Farthest point sampling (reference: tf_sampling_g.cu:105-170)
//
// The reference is m-1 strictly serial rounds of {update min-dist of all n points, arg-max}.
// It re-reads xyz + the min-dist scratch from L2/global every round (20 B/point/round) and
// spends 10 block barriers per round.  On MI355X a whole 32768-point scene fits ON CHIP in one
// CU, so the resident kernel below keeps it there for all rounds:
//
//   * one 512-thread workgroup (8 waves, 2 per SIMD) per scene; thread t owns the points
//     k = p*512 + t, p = 0..P-1  -- exactly the reference's per-thread stride (:130), so the
//     reference tie order (d desc, k mod 512 asc, k asc) becomes (d desc, t asc, p asc);
//   * x, y and the running min-dist live in VGPRs (3*P registers); z lives in VGPRs too for
//     P <= 32 and in LDS (128 KiB, read-only, ds_read_b128) for P = 64, which is what lets
//     32768 points fit: 384 KiB of registers + 128 KiB of LDS, no HBM/L2 traffic in the loop;
//   * arg-max = integer max on the float bit patterns (all candidates are >= +0, padding is
//     -1.0f, so signed-int order == float order): DPP row reductions inside a wave, one LDS
//     hop across the 8 waves, 2 workgroup barriers per round;
//   * the index of the maximum is NOT tracked in the hot loop (that would cost 2 more VALU per
//     point); only the winning wave resolves it afterwards, helped by per-8-point group maxima
//     that the max3 tree produces for free.
// ============================================================================================

#define FPS_T 512
typedef float v4f __attribute__((ext_vector_type(4)));
#define NEG_ONE_BITS ((int)0xBF800000)

template <int P>
struct FpsGroup {
    static constexpr int G = (P >= 8) ? 8 : P;   // points per resolve group
    static constexpr int NG = P / G;
};

template <int P, bool ZLDS>
__global__ __launch_bounds__(FPS_T) void fps_resident_kernel(int n, int m, const float* __restrict__ inp, int* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [0,64)  : per-wave candidates, int2 {max bits, lane}
    // [64,80) : winner record int4 {k, x, y, z}
    // [128,..): z plane, float4 [P/4][512]   (ZLDS only)
    int2* s_wave = reinterpret_cast<int2*>(smem);
    int4* s_ctr = reinterpret_cast<int4*>(smem + 64);
    v4f* s_z = reinterpret_cast<v4f*>(smem + 128);

    constexpr int G = FpsGroup<P>::G;
    constexpr int NG = FpsGroup<P>::NG;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const float* xyz = inp + (size_t)blockIdx.x * n * 3;
    int* o = out + (size_t)blockIdx.x * m;

    float x[P], y[P], z[ZLDS ? 1 : P];
    float td[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int k = p * FPS_T + t;
        float px = 0.f, py = 0.f, pz = 0.f;
        float d0 = -1.0f;                         // padding never wins (real candidates are >= 0)
        if (k < n) {
            px = xyz[k * 3 + 0];
            py = xyz[k * 3 + 1];
            pz = xyz[k * 3 + 2];
            d0 = 1e38f;                           // :117-119
        }
        // detach x/y from the dwordx2/x3 load tuple so the allocator may place them independently
        asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));
        x[p] = px;
        y[p] = py;
        td[p] = d0;
        if (ZLDS) reinterpret_cast<float*>(s_z)[((p >> 2) * FPS_T + t) * 4 + (p & 3)] = pz;
        else z[p] = pz;
    }
    if (t == 0) o[0] = 0;                          // :114-116
    float cx = xyz[0], cy = xyz[1], cz = xyz[2];   // centre of round 1 = point 0
    __syncthreads();

    for (int j = 1; j < m; ++j) {
        // ---- update + per-group maxima (value only) ----
        int g[NG];
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            int gm = NEG_ONE_BITS;
            if constexpr (ZLDS) {
#pragma unroll
                for (int h = 0; h < G / 4; ++h) {
                    // one ds_read_b128 serves 4 consecutive p; the empty asm keeps the compiler from
                    // narrowing it back into four scalar LDS reads
                    v4f zz = s_z[((q * G) / 4 + h) * FPS_T + t];
                    asm("" : "+v"(zz));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int p = q * G + h * 4 + e;
                        const float d = dist2_cuda(x[p] - cx, y[p] - cy, zz[e] - cz);   // :142
                        td[p] = vmin_f32(d, td[p]);                                     // :143
                        gm = max(gm, __float_as_int(td[p]));
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    const int p = q * G + i;
                    const float d = dist2_cuda(x[p] - cx, y[p] - cy, z[p] - cz);        // :142
                    td[p] = vmin_f32(d, td[p]);                                         // :143
                    gm = max(gm, __float_as_int(td[p]));
                }
            }
            g[q] = gm;
            __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from interleaving groups (VGPR pressure)
        }
        int best = g[0];
#pragma unroll
        for (int q = 1; q < NG; ++q) best = max(best, g[q]);

        // ---- wave arg-max: value, then lowest lane holding it ----
        const int wmax = wave_max_i32(best);
        const unsigned long long wm = __ballot(best == wmax);
        const int wlane = __builtin_ctzll(wm);
        if (lane == 0) s_wave[wave] = make_int2(wmax, wlane);
        __syncthreads();

        // ---- workgroup arg-max: every wave reduces the 8 candidates redundantly ----
        const int2 cand = s_wave[lane & 7];
        const int M = __builtin_amdgcn_readfirstlane(oct_max_i32(cand.x));
        const unsigned wmask = (unsigned)(__ballot(cand.x == M) & 0xFFull);
        const int wbest = __builtin_ctz(wmask);                  // lowest wave wins ties
        if (wave == wbest) {
            const int lw = __builtin_amdgcn_readlane(cand.y, wbest);
            // resolve the register slot inside lane lw: lowest p with td[p] == M
            int fp = -1;
            float fx = 0.f, fy = 0.f, fz = 0.f;
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                const unsigned long long gq = __ballot(g[q] == M);
                if (fp < 0 && ((gq >> lw) & 1ull)) {
#pragma unroll
                    for (int i = 0; i < G; ++i) {
                        const int p = q * G + i;
                        const unsigned long long pq = __ballot(__float_as_int(td[p]) == M);
                        if (fp < 0 && ((pq >> lw) & 1ull)) {
                            fp = p;
                            fx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[p]), lw));
                            fy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y[p]), lw));
                            if (!ZLDS) fz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z[p]), lw));
                        }
                    }
                }
            }
            const int tw = wbest * 64 + lw;
            if (ZLDS) fz = reinterpret_cast<const float*>(s_z)[((fp >> 2) * FPS_T + tw) * 4 + (fp & 3)];
            const int k = fp * FPS_T + tw;
            if (lane == 0) {
                *s_ctr = make_int4(k, __float_as_int(fx), __float_as_int(fy), __float_as_int(fz));
                o[j] = k;                                      // :166-168
            }
        }
        __syncthreads();
        const int4 c = *s_ctr;
        cx = __int_as_float(__builtin_amdgcn_readfirstlane(c.y));
        cy = __int_as_float(__builtin_amdgcn_readfirstlane(c.z));
        cz = __int_as_float(__builtin_amdgcn_readfirstlane(c.w));
    }
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
    const size_t lds = 128 + (ZLDS ? (size_t)P * FPS_T * sizeof(float) : 0);
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
    if (n <= 512) return launch_fps_resident<1, false>(b, n, m, inp, out, st);
    if (n <= 1024) return launch_fps_resident<2, false>(b, n, m, inp, out, st);
    if (n <= 2048) return launch_fps_resident<4, false>(b, n, m, inp, out, st);
    if (n <= 4096) return launch_fps_resident<8, false>(b, n, m, inp, out, st);
    if (n <= 8192) return launch_fps_resident<16, false>(b, n, m, inp, out, st);
    if (n <= 16384) return launch_fps_resident<32, false>(b, n, m, inp, out, st);
    if (n <= GSPN_FPS_RESIDENT_MAX) return launch_fps_resident<64, true>(b, n, m, inp, out, st);
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
