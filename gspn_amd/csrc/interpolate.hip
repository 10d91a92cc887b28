// interpolate.hip -- tf_ops/3d_interpolation on gfx950: three_nn, three_interpolate (+grad).
// The reference has CPU kernels only (tf_interpolate.cpp:60-153; ops registered DEVICE_CPU
// :187,222,262), so every FP module round-trips through the host.  These kernels keep the data
// on the device and reproduce the host arithmetic: unfused fp32, left to right.
#include <math.h>
#include <stdlib.h>
#include <stdint.h>

#include "common.h"
#include "csr_gather.h"

// ============================================================================================
// three_nn (tf_interpolate.cpp:60-103): for each dense point j the 3 smallest squared distances
// to the m sparse points, strict '<' insertion cascade (ties keep ascending k), squared
// distances returned.  One thread per dense point; the sparse cloud is staged through LDS in
// tiles of float4 {x,y,z,-} so one ds_read_b128 (broadcast, conflict free) feeds a whole wave.
// ============================================================================================
#define NN_TILE 1024
#define NN_BLOCK 256

// NN_Q queries per lane (1: measured on MI355X, 2 per lane halves the LDS reads but runs 1.8x slower -- half the waves, and a batch now
// takes the slow path when either query passes).  The scan is latency-bound per wave, not VALU-bound: what pays is testing NN_B
// candidates per branch (independent rejection values, one compare) instead of one data-dependent branch per candidate.
#define NN_Q 1
#define NN_B 8             // candidates tested per branch: the rejection values of a batch are independent (ILP), one compare decides
__global__ __launch_bounds__(NN_BLOCK) void three_nn_kernel(int b, int n, int m, const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                            float* __restrict__ dist, int* __restrict__ idx, const int* __restrict__ order, unsigned nblocks) {
    __shared__ float4 tile[NN_TILE + NN_B];
    __shared__ float tile_pm[NN_BLOCK / 64];
    for (unsigned blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {      // (a capped grid walks the query tiles: GSPN_NN_GRID)
    const int scene = blk % b;                      // scene <-> XCD affinity for the sparse cloud
    const int j0 = (blk / b) * (NN_BLOCK * NN_Q) + threadIdx.x;      // queries j0 + u*NN_BLOCK
    float x1[NN_Q], y1[NN_Q], z1[NN_Q];
#pragma unroll
    for (int u = 0; u < NN_Q; ++u) {
        int j = j0 + u * NN_BLOCK;
        x1[u] = y1[u] = z1[u] = 0.f;
        if (j < n) {
            if (order) j = order[(size_t)scene * n + j];        // thread t takes dense point order[t]: neighbours in space share a wave
            const float* q = xyz1 + ((size_t)scene * n + j) * 3;
            x1[u] = q[0]; y1[u] = q[1]; z1[u] = q[2];
        }
    }
    // double best=1e40 in the reference (:66): any finite float is smaller, +inf/NaN are not ->
    // identical to a float +inf initialiser; (float)1e40 == +inf on output (:91-96)
    float b1[NN_Q], b2[NN_Q], b3[NN_Q];
    int i1[NN_Q], i2[NN_Q], i3[NN_Q];
    // Exact search, cheap rejection: a candidate can only enter the top three if d < b3.  d = |p|^2 - 2 p.q + |q|^2, so the test
    // "|p|^2 - 2 p.q < b3 - |q|^2 + margin" (3 FMAs against a per-lane threshold, |p|^2 precomputed in the tile) never rejects such a
    // candidate as long as `margin` covers the rounding of both evaluations; survivors are re-evaluated with the reference's own
    // expression and compared exactly as before.
    float qx2[NN_Q], qy2[NN_Q], qz2[NN_Q], qq[NN_Q], qn[NN_Q], thr[NN_Q], margin[NN_Q];
#pragma unroll
    for (int u = 0; u < NN_Q; ++u) {
        b1[u] = b2[u] = b3[u] = INFINITY;
        i1[u] = i2[u] = i3[u] = 0;
        qx2[u] = -2.f * x1[u]; qy2[u] = -2.f * y1[u]; qz2[u] = -2.f * z1[u];
        qq[u] = __builtin_fmaf(x1[u], x1[u], __builtin_fmaf(y1[u], y1[u], z1[u] * z1[u]));
        qn[u] = sqrtf(qq[u]);
        thr[u] = INFINITY;
        margin[u] = 0.f;
    }
    const float* sp = xyz2 + (size_t)scene * m * 3;
    for (int k0 = 0; k0 < m; k0 += NN_TILE) {
        const int cnt = min(NN_TILE, m - k0);
        __syncthreads();
        float pm = 0.f;
        for (int t = threadIdx.x; t < cnt; t += NN_BLOCK) {
            const float px = sp[(size_t)(k0 + t) * 3 + 0], py = sp[(size_t)(k0 + t) * 3 + 1], pz = sp[(size_t)(k0 + t) * 3 + 2];
            const float pw = __builtin_fmaf(px, px, __builtin_fmaf(py, py, pz * pz));
            tile[t] = make_float4(px, py, pz, pw);
            pm = fmaxf(pm, pw);
        }
        // largest |p|^2 of the tile (bounds the rounding error of the rejection test for every candidate in it)
        for (int s_ = 32; s_ >= 1; s_ >>= 1) pm = fmaxf(pm, __shfl_xor(pm, s_, 64));
        if ((threadIdx.x & 63) == 0) tile_pm[threadIdx.x >> 6] = pm;
        __syncthreads();
        pm = fmaxf(fmaxf(tile_pm[0], tile_pm[1]), fmaxf(tile_pm[2], tile_pm[3]));
#pragma unroll
        for (int u = 0; u < NN_Q; ++u) {
            const float r = sqrtf(pm) + qn[u];
            margin[u] = 4e-6f * r * r;                   // >> 8 ulp of (|p| + |q|)^2: both evaluations err by a few ulp of that scale
            thr[u] = (b3[u] - qq[u]) + margin[u];        // +inf while fewer than three candidates were seen
        }
        // pad the tile to a whole batch with candidates that can never pass (|p|^2 = +inf)
        for (int t = cnt + threadIdx.x; t < ((cnt + NN_B - 1) / NN_B) * NN_B; t += NN_BLOCK) tile[t] = make_float4(0.f, 0.f, 0.f, INFINITY);
        __syncthreads();
        for (int k = 0; k < cnt; k += NN_B) {
            float4 p[NN_B];
#pragma unroll
            for (int v = 0; v < NN_B; ++v) p[v] = tile[k + v];
#pragma unroll
            for (int u = 0; u < NN_Q; ++u) {
                float sd[NN_B];
#pragma unroll
                for (int v = 0; v < NN_B; ++v)
                    sd[v] = __builtin_fmaf(p[v].x, qx2[u], __builtin_fmaf(p[v].y, qy2[u], __builtin_fmaf(p[v].z, qz2[u], p[v].w)));
                float smin = sd[0];
#pragma unroll
                for (int v = 1; v < NN_B; ++v) smin = fminf(smin, sd[v]);
                if (smin < thr[u]) {                                                // rare after warm-up: walk the batch in index order
#pragma unroll
                    for (int v = 0; v < NN_B; ++v) {
                        if (sd[v] < thr[u]) {
                            const float d = dist2_host(p[v].x - x1[u], p[v].y - y1[u], p[v].z - z1[u]);   // :74, the reference's expression
                            if (d < b3[u]) {
                                const int kk = k0 + k + v;
                                if (d < b1[u]) { b3[u] = b2[u]; i3[u] = i2[u]; b2[u] = b1[u]; i2[u] = i1[u]; b1[u] = d; i1[u] = kk; }
                                else if (d < b2[u]) { b3[u] = b2[u]; i3[u] = i2[u]; b2[u] = d; i2[u] = kk; }
                                else { b3[u] = d; i3[u] = kk; }
                                thr[u] = (b3[u] - qq[u]) + margin[u];
                            }
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < NN_Q; ++u) {
        int j = j0 + u * NN_BLOCK;
        if (j < n) {
            if (order) j = order[(size_t)scene * n + j];
            float* od = dist + ((size_t)scene * n + j) * 3;
            int* oi = idx + ((size_t)scene * n + j) * 3;
            od[0] = b1[u]; od[1] = b2[u]; od[2] = b3[u];
            oi[0] = i1[u]; oi[1] = i2[u]; oi[2] = i3[u];
        }
    }
    }
}
// ---- small known clouds (m <= NN_WAVE_MAX_M): one WAVE per query ------------------------------------------------------------
// The thread-per-query kernel above walks the known cloud serially and is fast only once a query's third-best distance has dropped far
// enough for the cheap rejection test to discard whole batches; on the small levels of the feature-propagation stack (2048 <- 512,
// 512 <- 128) the known points come in farthest-point order -- coarse to fine -- so nearly every batch improves some query of the wave
// and the exact path runs all the time: 76 / 27 us for 64 / 16 workgroups' worth of work (r03).  Here the known cloud of the scene
// sits in REGISTERS, spread over the lanes of a wave (candidate k = lane + 64 i), every lane evaluates the reference's expression for
// its <= NN_WAVE_P candidates of the wave's current query and keeps its own three best by the reference's rule (strict '<' in ascending
// k), and three wave-wide (distance, index) minima pick the result: the three smallest pairs in (d, k) order are exactly what the
// sequential cascade of tf_interpolate.cpp:69-89 leaves (ties keep the lower k).
#define NN_WAVE_P 16
#define NN_WAVE_MAX_M (64 * NN_WAVE_P)
#define NN_WAVE_QPW 16            // queries per wave (sequential): amortises loading the known cloud into registers
__device__ __forceinline__ int wave_min_i32(int v) {
    v = min(v, dpp_i32<DPP_QUAD_XOR1>(v));
    v = min(v, dpp_i32<DPP_QUAD_XOR2>(v));
    v = min(v, dpp_i32<DPP_ROW_HALF_MIRROR>(v));
    v = min(v, dpp_i32<DPP_ROW_MIRROR>(v));
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return min(min(a, b), min(c, d));
}
template <int P>
__global__ __launch_bounds__(256) void three_nn_wave_kernel(int b, int n, int m, const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                            float* __restrict__ dist, int* __restrict__ idx, int wpscene) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);           // global wave: (scene, run of NN_WAVE_QPW queries)
    const int scene = gw / wpscene;
    if (scene >= b) return;
    const int q0 = (gw - scene * wpscene) * NN_WAVE_QPW;
    const float* sp = xyz2 + (size_t)scene * m * 3;
    float px[P], py[P], pz[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int k = lane + 64 * i;
        const int kc = k < m ? k : 0;                              // clamped, unconditional loads
        px[i] = sp[kc * 3 + 0]; py[i] = sp[kc * 3 + 1]; pz[i] = sp[kc * 3 + 2];
    }
    const int INF_BITS = 0x7F800000;
    for (int qi = 0; qi < NN_WAVE_QPW; ++qi) {
        const int j = q0 + qi;
        if (j >= n) break;                                         // wave-uniform
        const float* q = xyz1 + ((size_t)scene * n + j) * 3;
        const float qx = q[0], qy = q[1], qz = q[2];               // uniform address: scalar loads
        // (float)1e40 == +inf: any finite distance is smaller, +inf / NaN never enter (tf_interpolate.cpp:66, :75-89)
        float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
        int i1 = 0, i2 = 0, i3 = 0;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int k = lane + 64 * i;
            const float d = dist2_host(px[i] - qx, py[i] - qy, pz[i] - qz);                       // :71-74, unfused, left to right
            if (k < m && d < b3) {
                if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
                else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
                else { b3 = d; i3 = k; }
            }
        }
        float od[3];
        int oi[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            // distances are >= 0 (or +inf): their bit patterns order like the values
            const int dmin = wave_min_i32(__float_as_int(b1));
            const int kmin = wave_min_i32(__float_as_int(b1) == dmin ? i1 : 0x7FFFFFFF);
            od[r] = __int_as_float(dmin);
            oi[r] = dmin == INF_BITS ? 0 : kmin;                   // fewer than three known points: the slot keeps (inf, 0) (:91-96)
            if (__float_as_int(b1) == dmin && i1 == kmin && dmin != INF_BITS) { b1 = b2; i1 = i2; b2 = b3; i2 = i3; b3 = INFINITY; i3 = 0; }
        }
        if (lane < 3) {
            const float dv = lane == 0 ? od[0] : (lane == 1 ? od[1] : od[2]);
            const int iv = lane == 0 ? oi[0] : (lane == 1 ? oi[1] : oi[2]);
            dist[((size_t)scene * n + j) * 3 + lane] = dv;
            idx[((size_t)scene * n + j) * 3 + lane] = iv;
        }
    }
}
static bool nn_wave_ok(int n, int m) {
    static const int on = [] { const char* e = getenv("GSPN_NN_WAVE"); return e ? atoi(e) : 1; }();     // (tuning / A-B hook)
    return on && m >= 1 && m <= NN_WAVE_MAX_M && n >= 1;
}
static int launch_three_nn_wave(int b, int n, int m, const float* xyz1, const float* xyz2, float* dist, int* idx, hipStream_t st) {
    const int wpscene = (n + NN_WAVE_QPW - 1) / NN_WAVE_QPW;
    const long long waves = (long long)b * wpscene;
    const long long blocks = (waves + 3) / 4;
    if (blocks > 0x7FFFFFFFll) return GSPN_ERR_UNSUPPORTED;
    if (m <= 64 * 2) hipLaunchKernelGGL(three_nn_wave_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, b, n, m, xyz1, xyz2, dist, idx, wpscene);
    else if (m <= 64 * 4) hipLaunchKernelGGL(three_nn_wave_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, b, n, m, xyz1, xyz2, dist, idx, wpscene);
    else if (m <= 64 * 8) hipLaunchKernelGGL(three_nn_wave_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, b, n, m, xyz1, xyz2, dist, idx, wpscene);
    else hipLaunchKernelGGL(three_nn_wave_kernel<NN_WAVE_P>, dim3((unsigned)blocks), dim3(256), 0, st, b, n, m, xyz1, xyz2, dist, idx, wpscene);
    return gspn_launch_status();
}

// ---- large known clouds: a uniform grid over the known points, built in LDS by every workgroup (r04) ---------------------------------
// three_nn_kernel tests every known point against every query: 537 M pairs at 8 x 32768 <- 2048, ~12 instructions each even with the
// cheap rejection test (VALU-bound at 0.86 of peak) -- 0.11 ms of full-chip work per step on the geometry stream.  The three
// nearest neighbours of a query lie in its own cell's neighbourhood: here a workgroup sorts the scene's known points into G^3 cells
// (counting sort in LDS: bounding box, histogram, scan, scatter of (x, y, z, k)), a thread per query scans the 3 x 3 x 3 block of cells
// around its own (nine contiguous runs of the sorted array) and keeps the three smallest (distance, index) pairs -- distance by the
// reference's own expression (tf_interpolate.cpp:71-74), ties to the lower index, which is what its strict-'<' cascade over ascending
// k leaves.  EXACT, not approximate: the block is accepted only if the third distance is smaller than the (squared, slightly shrunk)
// distance from the query to every face of the block that has cells behind it; otherwise the search restarts over all cells that the
// ball of that radius touches (with fewer than three points found: all cells).  No workspace, same signature.
#define NNG_T 256
#define NNG_MAX_G 16
#define NNG_MAX_M 8192                      // 128 KB of sorted points + the cell table
struct NNBest { float b1, b2, b3; int i1, i2, i3; };
__device__ __forceinline__ void nn_insert(NNBest& s, float d, int k) {
    if (d < s.b3 || (d == s.b3 && k < s.i3)) {
        if (d < s.b1 || (d == s.b1 && k < s.i1)) { s.b3 = s.b2; s.i3 = s.i2; s.b2 = s.b1; s.i2 = s.i1; s.b1 = d; s.i1 = k; }
        else if (d < s.b2 || (d == s.b2 && k < s.i2)) { s.b3 = s.b2; s.i3 = s.i2; s.b2 = d; s.i2 = k; }
        else { s.b3 = d; s.i3 = k; }
    }
}
template <int PPT>                        // known points per thread, held in registers through the three passes of the sort (m <= PPT * NNG_T)
__global__ __launch_bounds__(NNG_T) void three_nn_grid_kernel(int b, int n, int m, int G, int qpw, const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                              float* __restrict__ dist, int* __restrict__ idx, const int* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) float4 s_pts[];          // [m] known points sorted by cell: (x, y, z, bits of k)
    const int G3 = G * G * G;
    int* s_start = reinterpret_cast<int*>(s_pts + m);                       // [G3 + 1]
    int* s_cur = s_start + G3 + 1;                                          // [G3]
    __shared__ float s_red[6][NNG_T / 64];
    __shared__ int s_wsum[NNG_T / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int scene = blockIdx.x % b;                                       // scene <-> XCD affinity
    const int qb = blockIdx.x / b;
    const float* sp = xyz2 + (size_t)scene * m * 3;
    // ---- the thread's known points (k = t + 256 i): loaded once, all loads in flight ----
    float px[PPT], py[PPT], pz[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = t + NNG_T * i;
        const int kc = k < m ? k : 0;                                        // clamped, unconditional
        px[i] = sp[(size_t)kc * 3]; py[i] = sp[(size_t)kc * 3 + 1]; pz[i] = sp[(size_t)kc * 3 + 2];
    }
    // ---- bounding box of the known points ----
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int i = 0; i < PPT; ++i)
        if (t + NNG_T * i < m) {
            lo[0] = fminf(lo[0], px[i]); hi[0] = fmaxf(hi[0], px[i]);
            lo[1] = fminf(lo[1], py[i]); hi[1] = fmaxf(hi[1], py[i]);
            lo[2] = fminf(lo[2], pz[i]); hi[2] = fmaxf(hi[2], pz[i]);
        }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int sft = 32; sft >= 1; sft >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], sft, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], sft, 64)); }
        if (lane == 0) { s_red[a][wave] = lo[a]; s_red[3 + a][wave] = hi[a]; }
    }
    for (int c = t; c < G3; c += NNG_T) s_cur[c] = 0;
    __syncthreads();
    float inv[3], h[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = fminf(fminf(s_red[a][0], s_red[a][1]), fminf(s_red[a][2], s_red[a][3]));
        hi[a] = fmaxf(fmaxf(s_red[3 + a][0], s_red[3 + a][1]), fmaxf(s_red[3 + a][2], s_red[3 + a][3]));
        const float ext = hi[a] - lo[a];
        h[a] = ext / (float)G;
        inv[a] = (ext > 0.f && isfinite(ext)) ? (float)G / ext : 0.f;        // a flat (or non-finite) axis: every point in cell 0 of that axis
    }
    auto cell_of = [&](float x, float y, float z, int& cx, int& cy, int& cz) {
        cx = min(max((int)floorf((x - lo[0]) * inv[0]), 0), G - 1);
        cy = min(max((int)floorf((y - lo[1]) * inv[1]), 0), G - 1);
        cz = min(max((int)floorf((z - lo[2]) * inv[2]), 0), G - 1);
    };
    // ---- counting sort of the known points by cell ----
    int pcell[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        int cx, cy, cz;
        cell_of(px[i], py[i], pz[i], cx, cy, cz);
        pcell[i] = (cz * G + cy) * G + cx;
        if (t + NNG_T * i < m) atomicAdd(&s_cur[pcell[i]], 1);
    }
    __syncthreads();
    {
        const int per = (G3 + NNG_T - 1) / NNG_T;                            // consecutive cells per thread
        int local = 0;
        for (int c = t * per; c < min((t + 1) * per, G3); ++c) local += s_cur[c];
        int incl = local;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) { const int u = __shfl_up(incl, sft, 64); if (lane >= sft) incl += u; }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        int run = incl - local;
        for (int w = 0; w < wave; ++w) run += s_wsum[w];
        for (int c = t * per; c < min((t + 1) * per, G3); ++c) { const int v = s_cur[c]; s_start[c] = run; s_cur[c] = run; run += v; }
        if (t == NNG_T - 1) s_start[G3] = run;                               // == m
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PPT; ++i)
        if (t + NNG_T * i < m) s_pts[atomicAdd(&s_cur[pcell[i]], 1)] = make_float4(px[i], py[i], pz[i], __int_as_float(t + NNG_T * i));
    __syncthreads();
    // ---- the queries of this workgroup ----
    for (int u = 0; u < qpw; ++u) {
        int j = (qb * qpw + u) * NNG_T + t;
        if (j >= n) break;
        if (order) j = order[(size_t)scene * n + j];                         // spatially coherent lanes visit the same cells (broadcast reads)
        const float* q = xyz1 + ((size_t)scene * n + j) * 3;
        const float qx = q[0], qy = q[1], qz = q[2];
        // (float)1e40 == +inf (tf_interpolate.cpp:66): any finite distance is smaller; fewer than three known points leave (inf, 0)
        NNBest s{INFINITY, INFINITY, INFINITY, 0, 0, 0};
        int c[3];
        cell_of(qx, qy, qz, c[0], c[1], c[2]);
        int c0[3], c1[3];
        float bound = INFINITY;                                              // distance to the nearest face of the block with cells behind it
        const float qv[3] = {qx, qy, qz};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            c0[a] = max(c[a] - 1, 0);
            c1[a] = min(c[a] + 1, G - 1);
            // box-relative: q - lo is the subtraction cell_of itself performs; lo + c*h would round at the scale of the ABSOLUTE
            // coordinate (metres around 5e5 with a small extent: percent of h, far beyond the 0.999 shrink below).  A flat axis
            // (inv == 0: every known point in cell 0, nothing behind any face) contributes no face.
            const float rel = qv[a] - lo[a];
            if (inv[a] > 0.f) {
                if (c0[a] > 0) bound = fminf(bound, rel - (float)c0[a] * h[a]);
                if (c1[a] < G - 1) bound = fminf(bound, (float)(c1[a] + 1) * h[a] - rel);
            }
        }
        auto scan_block = [&]() {
            for (int cz = c0[2]; cz <= c1[2]; ++cz)
                for (int cy = c0[1]; cy <= c1[1]; ++cy) {
                    const int row = (cz * G + cy) * G;
                    const int e1 = s_start[row + c1[0] + 1];
                    int e = s_start[row + c0[0]];
                    for (; e + 3 < e1; e += 4) {                             // four candidates in flight, one test for "none of them enters"
                        const float4 p0 = s_pts[e], p1 = s_pts[e + 1], p2 = s_pts[e + 2], p3 = s_pts[e + 3];
                        const float d0 = dist2_host(p0.x - qx, p0.y - qy, p0.z - qz), d1 = dist2_host(p1.x - qx, p1.y - qy, p1.z - qz);     // :71-74, the reference's expression
                        const float d2 = dist2_host(p2.x - qx, p2.y - qy, p2.z - qz), d3 = dist2_host(p3.x - qx, p3.y - qy, p3.z - qz);
                        if (fminf(fminf(d0, d1), fminf(d2, d3)) <= s.b3) {
                            nn_insert(s, d0, __float_as_int(p0.w)); nn_insert(s, d1, __float_as_int(p1.w));
                            nn_insert(s, d2, __float_as_int(p2.w)); nn_insert(s, d3, __float_as_int(p3.w));
                        }
                    }
                    for (; e < e1; ++e) {
                        const float4 p = s_pts[e];
                        nn_insert(s, dist2_host(p.x - qx, p.y - qy, p.z - qz), __float_as_int(p.w));
                    }
                }
        };
        scan_block();
        // accepted iff no point outside the block can be as close as the third one found: its distance along the axis of the face it
        // lies behind is at least `bound` (cells are assigned with the same lo / inv, one rounding: the 1e-3 shrink covers it many times)
        const float safe = bound * 0.999f;
        if (!(s.b3 < safe * safe)) {
            // restart over every cell the ball of radius sqrt(b3) touches (b3 = +inf: all cells); the true three lie inside it
            const float r = s.b3 < INFINITY ? sqrtf(s.b3) * 1.001f : INFINITY;
            s = NNBest{INFINITY, INFINITY, INFINITY, 0, 0, 0};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (r < INFINITY && inv[a] > 0.f) {
                    const float rel = qv[a] - lo[a];                                                   // (box-relative, as above)
                    c0[a] = min(max((int)floorf((rel - r) * inv[a]) - 1, 0), G - 1);                  // (one cell of slack on both sides)
                    c1[a] = min(max((int)floorf((rel + r) * inv[a]) + 1, 0), G - 1);
                } else { c0[a] = 0; c1[a] = G - 1; }
            }
            scan_block();
        }
        float* od = dist + ((size_t)scene * n + j) * 3;
        int* oi = idx + ((size_t)scene * n + j) * 3;
        od[0] = s.b1; od[1] = s.b2; od[2] = s.b3;
        oi[0] = s.i1; oi[1] = s.i2; oi[2] = s.i3;
    }
}
static bool nn_grid_ok(int n, int m) {
    static const int on = [] { const char* e = getenv("GSPN_NN_CELLS"); return e ? atoi(e) : 1; }();     // (A/B hook: 0 = the all-pairs kernel)
    return on && m > NN_WAVE_MAX_M && m <= NNG_MAX_M && n >= 1;
}
static int launch_three_nn_grid(int b, int n, int m, const float* xyz1, const float* xyz2, float* dist, int* idx, const int* order, hipStream_t st) {
    static const int ppc = [] { const char* e = getenv("GSPN_NN_CELL_POINTS"); const int v = e ? atoi(e) : 4; return v > 0 ? v : 4; }();       // target points per cell
    static const int qpw = [] { const char* e = getenv("GSPN_NN_CELL_QPW"); const int v = e ? atoi(e) : 1; return v > 0 ? v : 1; }();          // queries per thread
    int G = (int)floor(cbrt((double)m / ppc) + 0.5);
    G = G < 2 ? 2 : (G > NNG_MAX_G ? NNG_MAX_G : G);
    const size_t dyn = sizeof(float4) * (size_t)m + sizeof(int) * (2 * (size_t)G * G * G + 1);
    const long long blocks = (long long)b * ((n + NNG_T * qpw - 1) / (NNG_T * qpw));
    if (blocks > 0x7FFFFFFFll) return GSPN_ERR_UNSUPPORTED;
#define NNG_GO(PPT_)                                                                                                                              \
    do {                                                                                                                                          \
        static size_t attr = 0;                                                                                                                   \
        if (dyn > attr) {                                                                                                                         \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&three_nn_grid_kernel<PPT_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); \
            if (e != hipSuccess) return (int)e;                                                                                                   \
            attr = dyn;                                                                                                                           \
        }                                                                                                                                         \
        hipLaunchKernelGGL(three_nn_grid_kernel<PPT_>, dim3((unsigned)blocks), dim3(NNG_T), dyn, st, b, n, m, G, qpw, xyz1, xyz2, dist, idx, order); \
    } while (0)
    if (m <= 8 * NNG_T) NNG_GO(8); else if (m <= 16 * NNG_T) NNG_GO(16); else NNG_GO(32);
#undef NNG_GO
    return gspn_launch_status();
}

static unsigned nn_grid(long long blocks) {
    static const long long cap = [] { const char* e = getenv("GSPN_NN_GRID"); const long long v = e ? atoll(e) : 0; return v > 0 ? v : (1ll << 31); }();
    return (unsigned)(blocks < cap ? blocks : cap);
}
extern "C" int gspn_threenn(int b, int n, int m, const float* xyz1, const float* xyz2, float* dist, int* idx, void* stream) {
    if (b < 0 || n < 0 || m < 0) return GSPN_ERR_ARG;
    if (b == 0 || n == 0) return 0;
    if (nn_wave_ok(n, m)) return launch_three_nn_wave(b, n, m, xyz1, xyz2, dist, idx, (hipStream_t)stream);
    if (nn_grid_ok(n, m)) return launch_three_nn_grid(b, n, m, xyz1, xyz2, dist, idx, nullptr, (hipStream_t)stream);
    const long long blocks = (long long)b * ((n + NN_BLOCK * NN_Q - 1) / (NN_BLOCK * NN_Q));
    if (blocks > 0x7FFFFFFFll) return GSPN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(three_nn_kernel, dim3(nn_grid(blocks)), dim3(NN_BLOCK), 0, (hipStream_t)stream, b, n, m, xyz1, xyz2, dist, idx, (const int*)nullptr, (unsigned)blocks);
    return gspn_launch_status();
}
extern "C" int gspn_threenn_ordered(int b, int n, int m, const float* xyz1, const float* xyz2, const int* order, float* dist, int* idx, void* stream) {
    if (b < 0 || n < 0 || m < 0) return GSPN_ERR_ARG;
    if (b == 0 || n == 0) return 0;
    if (nn_wave_ok(n, m)) return launch_three_nn_wave(b, n, m, xyz1, xyz2, dist, idx, (hipStream_t)stream);       // (the result never depends on `order`)
    if (nn_grid_ok(n, m)) return launch_three_nn_grid(b, n, m, xyz1, xyz2, dist, idx, order, (hipStream_t)stream);
    const long long blocks = (long long)b * ((n + NN_BLOCK * NN_Q - 1) / (NN_BLOCK * NN_Q));
    if (blocks > 0x7FFFFFFFll) return GSPN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(three_nn_kernel, dim3(nn_grid(blocks)), dim3(NN_BLOCK), 0, (hipStream_t)stream, b, n, m, xyz1, xyz2, dist, idx, order, (unsigned)blocks);
    return gspn_launch_status();
}

// ============================================================================================
// three_interpolate (tf_interpolate.cpp:107-127):  out[j,l] = p[i1,l]*w1 + p[i2,l]*w2 + p[i3,l]*w3
// (each product rounded, summed left to right).  One thread per output element.
// ============================================================================================
__global__ void three_interpolate_kernel(long total, int m, int c, int n, const float* __restrict__ points, const int* __restrict__ idx,
                                         const float* __restrict__ weight, float* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / c;                 // b*n + j
        const int l = (int)(i - row * c);
        const long bi = row / n;
        const int* ii = idx + row * 3;
        const float* w = weight + row * 3;
        const float* P = points + (size_t)bi * m * c + l;
        const float a = P[(size_t)ii[0] * c] * w[0];
        const float bb = P[(size_t)ii[1] * c] * w[1];
        const float cc = P[(size_t)ii[2] * c] * w[2];
        out[i] = (a + bb) + cc;
    }
}
// three_interpolate_grad (tf_interpolate.cpp:131-153): grad_points[i_t,l] += grad_out[j,l]*w_t.
// The reference is a sequential CPU loop (deterministic order); this is a hardware-atomic scatter,
// so sums agree to rounding, not bitwise.
__global__ void three_interpolate_grad_kernel(long total, int n, int c, int m, const float* __restrict__ grad_out, const int* __restrict__ idx,
                                              const float* __restrict__ weight, float* __restrict__ grad_points) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / c;
        const int l = (int)(i - row * c);
        const long bi = row / n;
        const int* ii = idx + row * 3;
        const float* w = weight + row * 3;
        const float g = grad_out[i];
        float* G = grad_points + (size_t)bi * m * c + l;
        atomicAdd(G + (size_t)ii[0] * c, g * w[0]);
        atomicAdd(G + (size_t)ii[1] * c, g * w[1]);
        atomicAdd(G + (size_t)ii[2] * c, g * w[2]);
    }
}
extern "C" int gspn_threeinterpolate(int b, int m, int c, int n, const float* points, const int* idx, const float* weight, float* out, void* stream) {
    if (b < 0 || m <= 0 || c <= 0 || n < 0) return GSPN_ERR_ARG;
    const long total = (long)b * n * c;
    if (total == 0) return 0;
    hipLaunchKernelGGL(three_interpolate_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, total, m, c, n, points, idx, weight, out);
    return gspn_launch_status();
}
extern "C" int gspn_threeinterpolate_grad(int b, int n, int c, int m, const float* grad_out, const int* idx, const float* weight, float* grad_points, void* stream) {
    if (b < 0 || m <= 0 || c <= 0 || n < 0) return GSPN_ERR_ARG;
    if (b == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * m * c, st);      // tf_interpolate.cpp:258
    if (e != hipSuccess) return (int)e;
    const long total = (long)b * n * c;
    if (total == 0) return 0;
    hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, total, n, c, m, grad_out, idx, weight, grad_points);
    return gspn_launch_status();
}

// ============================================================================================
// fp_concat: the input matrix of a feature-propagation MLP in one pass (pointnet_util.py:161-166):
//   out[row, 0:c2]      = three_interpolate(points2, idx, weight)[row]        ((p1*w1 + p2*w2) + p3*w3, as above)
//   out[row, c2:c2+c1]  = points1[row]
//   out[row, c2+c1:ld]  = 0                                                   (row pitch padded to 16 bytes for the MLP)
// replacing interpolate + tf.concat + pad (three full passes over the (b*n1, c) matrix).  One thread per output element.
// ============================================================================================
__global__ void fp_concat_kernel(long total, int n, int m, int c2, int c1, int ld, const float* __restrict__ points2, const int* __restrict__ idx,
                                 const float* __restrict__ weight, const float* __restrict__ points1, float* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / ld;                // b*n + j
        const int l = (int)(i - row * ld);
        float v = 0.f;
        if (l < c2) {
            const long bi = row / n;
            const int* ii = idx + row * 3;
            const float* w = weight + row * 3;
            const float* P = points2 + (size_t)bi * m * c2 + l;
            const float a = P[(size_t)ii[0] * c2] * w[0];
            const float bb = P[(size_t)ii[1] * c2] * w[1];
            const float cc = P[(size_t)ii[2] * c2] * w[2];
            v = (a + bb) + cc;
        } else if (l < c2 + c1) {
            v = points1[row * c1 + (l - c2)];
        }
        out[i] = v;
    }
}
// The same, four columns per thread (ld and c2 multiples of 4, 16-byte aligned rows -- what the MLP's input matrices are): one
// dwordx4 gather per neighbour and one dwordx4 store instead of four of each, and the row's three indices / weights fetched once per
// four outputs.  Same products, same left-to-right sum per element.
__global__ __launch_bounds__(256) void fp_concat4_kernel(long total4, int n, int m, int c2, int c1, int ld, const float* __restrict__ points2,
                                                         const int* __restrict__ idx, const float* __restrict__ weight,
                                                         const float* __restrict__ points1, float* __restrict__ out) {
    const int ld4 = ld >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long row = i / ld4;
        const int l = (int)(i - row * ld4) << 2;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (l < c2) {
            const long bi = row / n;
            const int* ii = idx + row * 3;
            const float* w = weight + row * 3;
            const float* P = points2 + (size_t)bi * m * c2 + l;
            const float w0 = w[0], w1 = w[1], w2 = w[2];
            const float4 a = *reinterpret_cast<const float4*>(P + (size_t)ii[0] * c2);
            const float4 bq = *reinterpret_cast<const float4*>(P + (size_t)ii[1] * c2);
            const float4 cq = *reinterpret_cast<const float4*>(P + (size_t)ii[2] * c2);
            v.x = (a.x * w0 + bq.x * w1) + cq.x * w2;
            v.y = (a.y * w0 + bq.y * w1) + cq.y * w2;
            v.z = (a.z * w0 + bq.z * w1) + cq.z * w2;
            v.w = (a.w * w0 + bq.w * w1) + cq.w * w2;
        } else {
            const float* q = points1 + row * c1 - c2;          // column l of the output row is q[l]
            const int end = c2 + c1;
            if (l + 0 < end) v.x = q[l + 0];
            if (l + 1 < end) v.y = q[l + 1];
            if (l + 2 < end) v.z = q[l + 2];
            if (l + 3 < end) v.w = q[l + 3];
        }
        *reinterpret_cast<float4*>(out + i * 4) = v;
    }
}
// gradient: grad_points2[i_t, l] += g[row, l] * w_t (hardware atomics, like three_interpolate_grad), grad_points1[row, l] = g[row, c2 + l]
__global__ void fp_concat_grad_kernel(long total, int n, int m, int c2, int c1, int ld, const float* __restrict__ g, const int* __restrict__ idx,
                                      const float* __restrict__ weight, float* __restrict__ grad_points2, float* __restrict__ grad_points1) {
    const int cc = c2 + c1;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / cc;
        const int l = (int)(i - row * cc);
        const float gv = g[row * ld + l];
        if (l < c2) {
            if (grad_points2) {
                const long bi = row / n;
                const int* ii = idx + row * 3;
                const float* w = weight + row * 3;
                float* G = grad_points2 + (size_t)bi * m * c2 + l;
                atomicAdd(G + (size_t)ii[0] * c2, gv * w[0]);
                atomicAdd(G + (size_t)ii[1] * c2, gv * w[1]);
                atomicAdd(G + (size_t)ii[2] * c2, gv * w[2]);
            }
        } else if (grad_points1) {
            grad_points1[row * c1 + (l - c2)] = gv;
        }
    }
}
extern "C" int gspn_fp_concat(int b, int n, int m, int c2, int c1, const float* points2, const int* idx, const float* weight,
                              const float* points1, int ld, float* out, void* stream) {
    if (b < 0 || n < 0 || m <= 0 || c2 <= 0 || c1 < 0 || ld < c2 + c1 || (c1 > 0 && !points1)) return GSPN_ERR_ARG;
    const long total = (long)b * n * ld;
    if (total == 0) return 0;
    if (ld % 4 == 0 && c2 % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(points2) & 15) == 0)
        hipLaunchKernelGGL(fp_concat4_kernel, dim3(grid_for(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, total / 4, n, m, c2, c1, ld, points2, idx, weight,
                           points1, out);
    else
        hipLaunchKernelGGL(fp_concat_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, total, n, m, c2, c1, ld, points2, idx, weight, points1, out);
    return gspn_launch_status();
}
extern "C" int gspn_fp_concat_grad(int b, int n, int m, int c2, int c1, int ld, const float* grad_out, const int* idx, const float* weight,
                                   float* grad_points2, float* grad_points1, void* stream) {
    if (b < 0 || n < 0 || m <= 0 || c2 <= 0 || c1 < 0 || ld < c2 + c1) return GSPN_ERR_ARG;
    if (b == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (grad_points2) {
        hipError_t e = hipMemsetAsync(grad_points2, 0, sizeof(float) * (size_t)b * m * c2, st);
        if (e != hipSuccess) return (int)e;
    }
    const long total = (long)b * n * (c2 + c1);
    if (total == 0) return 0;
    hipLaunchKernelGGL(fp_concat_grad_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, total, n, m, c2, c1, ld, grad_out, idx, weight, grad_points2, grad_points1);
    return gspn_launch_status();
}

// fp_concat gradient without atomics: the caller supplies, per scene, the (dense point, neighbour slot) pairs grouped by sparse point
// -- `order` (b, 3n): positions p = 3*i + t into the flattened idx array, sorted by idx[p] with ties in ascending p; `offsets` (b, m+1):
// the range of `order` that belongs to each sparse point (coordinate-only data, built once per batch next to the 3-NN search).
// One wave per (scene, sparse point, 64-channel chunk), lane = channel: the contributions of a sparse point are added in ascending
// (i, t), which is exactly the order of the reference's sequential loop (tf_interpolate.cpp:131-153) -- bit-identical sums, no
// atomics, and each 256-byte row segment of grad_out is read once per use.
__global__ __launch_bounds__(256) void fp_concat_grad_csr_kernel(int n, int m, int c2, int c1, int ld, const float* __restrict__ g,
                                                                 const int* __restrict__ order, const int* __restrict__ offsets,
                                                                 const float* __restrict__ weight, float* __restrict__ grad_points2,
                                                                 float* __restrict__ grad_points1, long nwaves2, long copy_total) {
    const int lane = threadIdx.x & 63;
    const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int chunks = (c2 + 63) / 64;
    if (wv < nwaves2) {
        if (!grad_points2) return;
        const int ch = (int)(wv % chunks);
        const long sj = wv / chunks;                        // scene * m + j
        const int scene = (int)(sj / m), j = (int)(sj - (long)scene * m);
        const int l = ch * 64 + lane;
        const int* off = offsets + (size_t)scene * (m + 1);
        const int e0 = off[j], e1 = off[j + 1];
        const int* ord = order + (size_t)scene * 3 * n;
        const float* w = weight + (size_t)scene * 3 * n;
        const float* gs = g + (size_t)scene * n * ld;
        const bool act = l < c2;
        const int lc = act ? l : 0;
        float acc = 0.f;
        int e = e0;
        for (; e + 3 < e1; e += 4) {                         // 4 independent row loads in flight, added in order
            const int p0 = ord[e], p1 = ord[e + 1], p2 = ord[e + 2], p3 = ord[e + 3];
            const float v0 = gs[(size_t)(p0 / 3) * ld + lc], v1 = gs[(size_t)(p1 / 3) * ld + lc];
            const float v2 = gs[(size_t)(p2 / 3) * ld + lc], v3 = gs[(size_t)(p3 / 3) * ld + lc];
            acc += v0 * w[p0];
            acc += v1 * w[p1];
            acc += v2 * w[p2];
            acc += v3 * w[p3];
        }
        for (; e < e1; ++e) {
            const int p = ord[e];
            acc += gs[(size_t)(p / 3) * ld + lc] * w[p];
        }
        if (act) grad_points2[((size_t)scene * m + j) * c2 + l] = acc;
    } else if (grad_points1) {
        // the points1 columns are a plain slice of grad_out
        const long nthreads = ((long)gridDim.x * 4 - nwaves2) * 64;
        for (long i = (wv - nwaves2) * 64 + lane; i < copy_total; i += nthreads) {
            const long row = i / c1;
            const int l = (int)(i - row * c1);
            grad_points1[i] = g[row * ld + c2 + l];
        }
    }
}
static int fp_concat_grad_csr_impl(int b, int n, int m, int c2, int c1, int ld, const float* grad_out, const int* order, const int* offsets,
                                   const float* weight, float* grad_points2, float* grad_points1, int split_t, void* stream) {
    if (b < 0 || n < 0 || m <= 0 || c2 <= 0 || c1 < 0 || ld < c2 + c1 || !order || !offsets || !weight || split_t < 0) return GSPN_ERR_ARG;
    if (b == 0 || n == 0) return 0;
    const long copy_total = grad_points1 ? (long)b * n * c1 : 0;
    if (grad_points2) {                                              // sixteen lanes per sparse point (csr_gather.h); same sums, same order
        const CsrCopy cp{grad_out, grad_points1, ld, c2, c1, copy_total};
        const int rc = csr_gather16(true, b, m, 3 * n, n, c2, ld, 0, grad_out, order, offsets, weight, grad_points2, cp, (hipStream_t)stream, split_t);
        if (rc != GSPN_ERR_UNSUPPORTED) return rc;
    }
    const long nwaves2 = (long)b * m * ((c2 + 63) / 64);
    long copy_waves = (copy_total + 64 * 16 - 1) / (64 * 16);          // ~16 elements per lane
    if (copy_waves > 8192) copy_waves = 8192;
    const long blocks = (nwaves2 + copy_waves + 3) / 4;
    if (blocks > 0x7FFFFFFFl) return GSPN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(fp_concat_grad_csr_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, m, c2, c1, ld, grad_out, order, offsets,
                       weight, grad_points2, grad_points1, nwaves2, copy_total);
    return gspn_launch_status();
}
extern "C" int gspn_fp_concat_grad_csr(int b, int n, int m, int c2, int c1, int ld, const float* grad_out, const int* order, const int* offsets,
                                       const float* weight, float* grad_points2, float* grad_points1, void* stream) {
    return fp_concat_grad_csr_impl(b, n, m, c2, c1, ld, grad_out, order, offsets, weight, grad_points2, grad_points1, 0, stream);
}
// the same sums in a FIXED BUT DIFFERENT order for lists longer than split_t entries (csr_gather.h: SPLIT): for callers whose order is not the reference's
// (the transposed aggregation of a pre-aggregated first layer); split_t = 0 is gspn_fp_concat_grad_csr
extern "C" int gspn_fp_concat_grad_csr_split(int b, int n, int m, int c2, int c1, int ld, const float* grad_out, const int* order, const int* offsets,
                                             const float* weight, float* grad_points2, float* grad_points1, int split_t, void* stream) {
    return fp_concat_grad_csr_impl(b, n, m, c2, c1, ld, grad_out, order, offsets, weight, grad_points2, grad_points1, split_t, stream);
}

// ============================================================================================
// Inverse-distance weights of the three neighbours (utils/pointnet_util.py:157-160):
//   dist = max(dist, 1e-10); norm = sum_k 1/dist_k; weight_k = (1/dist_k) / norm          -- one kernel instead of five element-wise ones
// ============================================================================================
__global__ void three_nn_weights_kernel(long total, const float* __restrict__ dist, float* __restrict__ weight) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const float d0 = fmaxf(dist[i * 3 + 0], 1e-10f), d1 = fmaxf(dist[i * 3 + 1], 1e-10f), d2 = fmaxf(dist[i * 3 + 2], 1e-10f);
        const float r0 = 1.0f / d0, r1 = 1.0f / d1, r2 = 1.0f / d2;
        const float norm = (r0 + r1) + r2;
        weight[i * 3 + 0] = r0 / norm;
        weight[i * 3 + 1] = r1 / norm;
        weight[i * 3 + 2] = r2 / norm;
    }
}
extern "C" int gspn_three_nn_weights(long total, const float* dist, float* weight, void* stream) {
    if (total < 0) return GSPN_ERR_ARG;
    if (total == 0) return 0;
    if (!dist || !weight) return GSPN_ERR_ARG;
    hipLaunchKernelGGL(three_nn_weights_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, total, dist, weight);
    return gspn_launch_status();
}
