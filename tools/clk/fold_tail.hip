// What would folding bn_finalize into its producer buy?  
// A chain of LAYERS dependent "layers" replayed from a hipGraph; a layer = a persistent producer kernel (NP workgroups; each streams its
// share of a (rows x c) matrix, scaled by the PREVIOUS layer's scale/shift, and leaves one partial row [2][c] of column sums) followed by
// the finalisation of those partial rows into scale/shift:
//   V0  a kernel of its own: c workgroups x 256 threads, double accumulation (what mlp.hip does: bn_finalize_kernel)
//   V1  folded, one level: every workgroup takes a ticket after its partial row is visible; the last arriver sums all NP rows itself
//   V2  folded, two levels: the last arriver of each bucket of 32 workgroups sums its bucket (double) and takes a second ticket; the
//       last of those sums the bucket rows and writes scale/shift
// Prints microseconds per layer for each variant: the difference to V0 is what the fold can save per layer.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

// FULL = 1: HIP's __threadfence() on both sides (agent-scope release = L2 write-back, acquire = L2 invalidate) and plain loads / stores;
// FULL = 0: the partial rows travel as relaxed agent-scope atomic stores / loads (write-through, cache-bypassing), ordered by s_waitcnt only
#ifndef FULL
#define FULL 0
#endif
#if FULL
#define PUB(p, v) (*(p) = (v))
#define GET(p) (*(p))
#define FENCE() __threadfence()
#define ACQ() __threadfence()
#else
#define PUB(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define GET(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define FENCE() __builtin_amdgcn_s_waitcnt(0)
#define ACQ() ((void)0)
#endif
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}
// final math of one channel from (sum, sum of squares)
__device__ __forceinline__ void finish(int j, double a0, double a1, long rows, float* scale, float* shift) {
    const double mu = a0 / (double)rows;
    double v = a1 / (double)rows - mu * mu;
    if (v < 0.0) v = 0.0;
    const float inv = (float)(1.0 / sqrt(v + 1e-3));
    scale[j] = inv;
    shift[j] = -(float)mu * inv;
}
__global__ __launch_bounds__(256) void finalize_kernel(long rows, int c, const float* __restrict__ stats, int nparts, float* scale, float* shift) {
    __shared__ double sh2[2][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = blockIdx.x;
    double a0 = 0.0, a1 = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 256) { a0 += (double)stats[(size_t)p * 2 * c + j]; a1 += (double)stats[(size_t)p * 2 * c + c + j]; }
    a0 = wave_sum(a0); a1 = wave_sum(a1);
    if (lane == 0) { sh2[0][wv] = a0; sh2[1][wv] = a1; }
    __syncthreads();
    if (threadIdx.x == 0) finish(j, (sh2[0][0] + sh2[0][1]) + (sh2[0][2] + sh2[0][3]), (sh2[1][0] + sh2[1][1]) + (sh2[1][2] + sh2[1][3]), rows, scale, shift);
}

template <int MODE>
__global__ __launch_bounds__(256) void producer(long rows, int c, const float* __restrict__ X, const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                float* __restrict__ stats, float* scale, float* shift, unsigned* tickets, double* brow, float* __restrict__ Yo) {
    extern __shared__ float sh[];                    // [2][16][c] (c <= 128)
    const int cq = c >> 2, rpi = 256 / cq, q = threadIdx.x % cq, rr = threadIdx.x / cq;
    const float4 sc = *reinterpret_cast<const float4*>(in_scale + 4 * q), sf = *reinterpret_cast<const float4*>(in_shift + 4 * q);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    for (long r = (long)blockIdx.x * rpi + rr; r < rows; r += (long)gridDim.x * rpi) {
        float4 x = *reinterpret_cast<const float4*>(X + (size_t)r * c + 4 * q);
        x.x = fmaxf(x.x * sc.x + sf.x, 0.f); x.y = fmaxf(x.y * sc.y + sf.y, 0.f); x.z = fmaxf(x.z * sc.z + sf.z, 0.f); x.w = fmaxf(x.w * sc.w + sf.w, 0.f);
        if (Yo) *reinterpret_cast<float4*>(Yo + (size_t)r * c + 4 * q) = x;          // (the real producers leave a large dirty output behind)
        s1.x += x.x; s1.y += x.y; s1.z += x.z; s1.w += x.w;
        s2.x += x.x * x.x; s2.y += x.y * x.y; s2.z += x.z * x.z; s2.w += x.w * x.w;
    }
    *reinterpret_cast<float4*>(sh + ((size_t)0 * rpi + rr) * c + 4 * q) = s1;
    *reinterpret_cast<float4*>(sh + ((size_t)1 * rpi + rr) * c + 4 * q) = s2;
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * c; j += 256) {
        const int h = j / c, cc = j - h * c;
        float v = 0.f;
        for (int k = 0; k < rpi; ++k) v += sh[((size_t)h * rpi + k) * c + cc];
        if (MODE == 0) stats[(size_t)blockIdx.x * 2 * c + j] = v;
        else PUB(stats + (size_t)blockIdx.x * 2 * c + j, v);         // agent-scope write-through: visible to every XCD once vmcnt drains
    }
    if (MODE == 0) return;
    // ---- the fold ----
    __shared__ unsigned s_last;
    FENCE();                                          // this workgroup's partial row has reached the coherence point before its ticket
    __syncthreads();
    const int np = gridDim.x;
    if (MODE == 1) {
        if (threadIdx.x == 0) s_last = atomicAdd(&tickets[0], 1u) == (unsigned)(np - 1);
        __syncthreads();
        if (!s_last) return;
        ACQ();
        // 256 threads: column j = t % (2c), slice = t / (2c); coalesced rows, 8 independent loads in flight, double accumulation, fixed order
        const int cols = 2 * c, nsl = 256 / cols > 0 ? 256 / cols : 1;
        double* dsh = reinterpret_cast<double*>(sh);
        if (threadIdx.x < cols * nsl) {
            const int col = threadIdx.x % cols, sl = threadIdx.x / cols;
            double a = 0.0;
            int p = sl;
            for (; p + 7 * nsl < np; p += 8 * nsl) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = GET(stats + (size_t)(p + u * nsl) * cols + col);
#pragma unroll
                for (int u = 0; u < 8; ++u) a += (double)v[u];
            }
            for (; p < np; p += nsl) a += (double)GET(stats + (size_t)p * cols + col);
            dsh[sl * cols + col] = a;
        }
        __syncthreads();
        if (threadIdx.x < c) {
            double a0 = 0.0, a1 = 0.0;
            for (int sl = 0; sl < nsl; ++sl) { a0 += dsh[sl * cols + threadIdx.x]; a1 += dsh[sl * cols + c + threadIdx.x]; }
            finish(threadIdx.x, a0, a1, rows, scale, shift);
        }
        if (threadIdx.x == 0) tickets[0] = 0;
        return;
    }
    // two levels: buckets of BK workgroups; both reductions use all 256 threads (column x slice), every thread's loads independent
    constexpr int BK = 32;
    const int bucket = blockIdx.x / BK, nb = (np + BK - 1) / BK;
    const int in_bucket = min(BK, np - bucket * BK);
    if (threadIdx.x == 0) s_last = atomicAdd(&tickets[1 + bucket], 1u) == (unsigned)(in_bucket - 1);
    __syncthreads();
    if (!s_last) return;
    ACQ();
    const int cols = 2 * c, nsl = 256 / cols > 0 ? 256 / cols : 1;
    double* dsh = reinterpret_cast<double*>(sh);
    for (int col0 = 0; col0 < cols; col0 += 256) {
        const int col = col0 + threadIdx.x % min(cols, 256), sl = threadIdx.x / min(cols, 256);
        float v[BK];
#pragma unroll
        for (int u = 0; u < BK; ++u) { const int p = sl + u * nsl; v[u] = p < in_bucket ? GET(stats + (size_t)(bucket * BK + p) * cols + col) : 0.f; }
        double a = 0.0;
#pragma unroll
        for (int u = 0; u < BK; ++u) if (u * nsl < BK) a += (double)v[u];
        if (nsl > 1) {
            dsh[sl * cols + col] = a;
            __syncthreads();
            if (sl == 0) { for (int k = 1; k < nsl; ++k) a += dsh[k * cols + col]; }
            __syncthreads();
        }
        if (sl == 0) PUB(brow + (size_t)bucket * cols + col, a);
    }
    FENCE();
    __syncthreads();
    if (threadIdx.x == 0) { tickets[1 + bucket] = 0; s_last = atomicAdd(&tickets[0], 1u) == (unsigned)(nb - 1); }
    __syncthreads();
    if (!s_last) return;
    ACQ();
    for (int col0 = 0; col0 < cols; col0 += 256) {
        const int col = col0 + threadIdx.x % min(cols, 256), sl = threadIdx.x / min(cols, 256);
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) { const int b = sl + u * nsl; v[u] = b < nb ? GET(brow + (size_t)b * cols + col) : 0.0; }
        double a = 0.0;
#pragma unroll
        for (int u = 0; u < 32; ++u) a += v[u];
        dsh[sl * cols + col] = a;
    }
    __syncthreads();
    if (threadIdx.x < c) {
        double a0 = 0.0, a1 = 0.0;
        for (int k = 0; k < nsl; ++k) { a0 += dsh[k * cols + threadIdx.x]; a1 += dsh[k * cols + c + threadIdx.x]; }
        finish(threadIdx.x, a0, a1, rows, scale, shift);
    }
    if (threadIdx.x == 0) tickets[0] = 0;
}

int main(int argc, char** argv) {
    const int c = argc > 1 ? atoi(argv[1]) : 64;
    const int NP = argc > 2 ? atoi(argv[2]) : 896;
    const long rows = argc > 3 ? atol(argv[3]) : 131072;
    const int LAYERS = 16;
    float *X, *stats, *scale, *shift, *Y; unsigned* tickets; double* brow;
    (void)hipMalloc(&Y, rows * c * 4);
    (void)hipMalloc(&X, rows * c * 4); (void)hipMemset(X, 0, rows * c * 4);
    (void)hipMalloc(&stats, (size_t)NP * 2 * c * 4);
    (void)hipMalloc(&scale, (LAYERS + 1) * c * 4); (void)hipMalloc(&shift, (LAYERS + 1) * c * 4);
    (void)hipMemset(scale, 0, (LAYERS + 1) * c * 4); (void)hipMemset(shift, 0, (LAYERS + 1) * c * 4);
    (void)hipMalloc(&tickets, 4096); (void)hipMemset(tickets, 0, 4096);
    (void)hipMalloc(&brow, 64 * 2 * c * 8);
    hipStream_t st; (void)hipStreamCreate(&st);
    const size_t shb = 8192;
    for (int mode = 0; mode < 3; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int l = 0; l < LAYERS; ++l) {
            float *isc = scale + l * c, *ish = shift + l * c, *osc = scale + (l + 1) * c, *osh = shift + (l + 1) * c;
            if (mode == 0) {
                hipLaunchKernelGGL((producer<0>), dim3(NP), dim3(256), shb, st, rows, c, X, isc, ish, stats, osc, osh, tickets, brow, Y);
                hipLaunchKernelGGL(finalize_kernel, dim3(c), dim3(256), 0, st, rows, c, stats, NP, osc, osh);
            } else if (mode == 1) hipLaunchKernelGGL((producer<1>), dim3(NP), dim3(256), shb, st, rows, c, X, isc, ish, stats, osc, osh, tickets, brow, Y);
            else hipLaunchKernelGGL((producer<2>), dim3(NP), dim3(256), shb, st, rows, c, X, isc, ish, stats, osc, osh, tickets, brow, Y);
        }
        (void)hipStreamEndCapture(st, &g);
        (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int w = 0; w < 3; ++w) (void)hipGraphLaunch(ge, st);
        (void)hipStreamSynchronize(st);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, st);
        const int reps = 20;
        for (int r = 0; r < reps; ++r) (void)hipGraphLaunch(ge, st);
        (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<float> h(c);
        (void)hipMemcpy(h.data(), scale + LAYERS * c, c * 4, hipMemcpyDeviceToHost);
        printf("c=%d parts=%d rows=%ld  %s: %.2f us per layer  (scale[0] of the last layer %.6f)\n", c, NP, rows,
               mode == 0 ? "V0 separate finalize kernel " : (mode == 1 ? "V1 fold, one level          " : "V2 fold, buckets of 32      "), ms * 1e3 / reps / LAYERS, h[0]);
    }
    return 0;
}
