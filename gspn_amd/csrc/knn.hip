// knn.hip -- direct k-nearest-neighbour search for knn_point (tf_ops/grouping/tf_grouping.py:71-96).
//
// The reference builds a dense (b,m,n) squared-distance tensor in TensorFlow (tile / subtract / square / reduce_sum, :85-87), runs
// the partial selection sort of tf_grouping_g.cu:144-184 over every row and slices the first k columns: O(b*m*n) memory (2 GiB at
// b=8, m=2048, n=32768) and O(k*n) dependent global-memory steps per row.  Here: no matrix.  One wave per query streams the data
// cloud through LDS tiles once and keeps its 64 best candidates, one per lane, sorted by (distance, index); then it replays the
// selection sort on those 64 alone.
//
// Why 64 candidates reproduce the reference bit for bit, ties included.  The selection sort picks, at step s, the minimum over the
// CURRENT positions >= s with strict '<' (lowest current position wins ties) and swaps it with the element at position s -- so an
// element that sat at a position < k can be moved behind its equals.  Let v_k be the k-th smallest distance.  Only elements with
// distance <= v_k are ever picked.  All elements below v_k are picked; of those equal to v_k the sort takes the ones with the lowest
// current positions, and an element moves only when it sits at a position < k.  Hence the ones that can matter are: everything below
// v_k (< k elements), the ties at v_k that start at a position < k (<= k elements) and the first few ties behind -- a prefix of at
// most 2k elements in (distance, original index) order.  With k <= 32 that prefix is inside the wave's 64 candidates, and the replay
// needs nothing else: when the element at position s is not a candidate its destination is irrelevant (it can never be picked), when
// it is, it moves to the picked candidate's position, which is known.
//
// Distances are TensorFlow's fp32 arithmetic: (data - query)^2 summed x, y, z left to right, unfused.
#include "common.h"

#define KNN_WAVES 8
#define KNN_TILE 2048
#define KNN_KMAX 32

__device__ __forceinline__ int wave_min_i32(int v) { return -wave_max_i32(-v); }

__global__ __launch_bounds__(KNN_WAVES * 64) void knn_direct_kernel(int b, int n, int m, int k, const float* __restrict__ xyz1,
                                                                   const float* __restrict__ xyz2, float* __restrict__ val, int* __restrict__ idx) {
    __shared__ float tile[KNN_TILE * 3];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int scene = blockIdx.x % b;                    // scene <-> XCD affinity: a scene's data stays in one L2
    const int j = (blockIdx.x / b) * KNN_WAVES + wave;
    const bool active = j < m;
    const float* data = xyz1 + (size_t)scene * n * 3;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (active) {
        const float* q = xyz2 + ((size_t)scene * m + j) * 3;
        qx = q[0]; qy = q[1]; qz = q[2];
    }
    float lv = INFINITY;          // candidate list, lane i = i-th smallest (distance, index)
    int li = -1;
    float tau = INFINITY;         // distance of the 64th candidate: only strictly smaller distances enter
    for (int base = 0; base < n; base += KNN_TILE) {
        const int cnt = min(KNN_TILE, n - base);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt * 3; i += KNN_WAVES * 64) tile[i] = data[(size_t)base * 3 + i];
        __syncthreads();
        if (!active) continue;
        for (int c0 = 0; c0 < cnt; c0 += 64) {
            const int p = c0 + lane;
            float d = INFINITY;
            if (p < cnt) {
                const float dx = tile[p * 3 + 0] - qx, dy = tile[p * 3 + 1] - qy, dz = tile[p * 3 + 2] - qz;      // data - query (:85-86)
                d = (dx * dx + dy * dy) + dz * dz;
            }
            unsigned long long hits = __ballot(d < tau);
            while (hits) {                                // ascending index: a later equal distance goes behind the earlier one
                const int src = __builtin_ctzll(hits);
                hits &= hits - 1;
                const float nv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), src));
                if (!(nv < tau)) continue;
                const int ni = base + c0 + src;
                const int pos = __builtin_popcountll(__ballot(lv <= nv));
                const float upv = __shfl_up(lv, 1, 64);
                const int upi = __shfl_up(li, 1, 64);
                if (lane > pos) { lv = upv; li = upi; }
                else if (lane == pos) { lv = nv; li = ni; }
                tau = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lv), 63));
            }
        }
    }
    if (!active) return;
    // replay of the selection sort (tf_grouping_g.cu:162-183) on the candidates
    int cur = li;                                        // current position in the row
    bool gone = li < 0;                                  // picked already, or an empty slot
    float ov = 0.f;
    int oi = 0;
    for (int s = 0; s < k; ++s) {
        const int vb = gone ? 0x7FFFFFFF : __float_as_int(lv);           // distances are >= +0: integer order == float order
        const int minv = wave_min_i32(vb);
        const bool cand = !gone && vb == minv;
        const int minp = wave_min_i32(cand ? cur : 0x7FFFFFFF);
        const bool sel = cand && cur == minp;
        const int sl = __builtin_ctzll(__ballot(sel));
        const float sv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lv), sl));
        const int si = __builtin_amdgcn_readlane(li, sl);
        if (!gone && !sel && cur == s) cur = minp;       // the swap of :170-178 moves the element at position s to the minimum's place
        if (lane == s) { ov = sv; oi = si; }
        if (sel) gone = true;
    }
    if (lane < k) {
        val[((size_t)scene * m + j) * k + lane] = ov;
        idx[((size_t)scene * m + j) * k + lane] = oi;
    }
}

// knn_point(k, xyz1, xyz2) of tf_grouping.py:71-96 for 3-D points: xyz1 (b,n,3) data, xyz2 (b,m,3) queries ->
// val (b,m,k) squared distances, idx (b,m,k); identical to dense matrix + selection_sort_gpu + slice.  k <= 32 and k <= n.
extern "C" int gspn_knn_point(int b, int n, int m, int k, const float* xyz1, const float* xyz2, float* val, int* idx, void* stream) {
    if (b < 0 || n <= 0 || m < 0 || k <= 0) return GSPN_ERR_ARG;
    if (k > n) return GSPN_ERR_ARG;                      // the slice [:, :, :k] of an (b,m,n) tensor (tf_grouping.py:91-92)
    if (b == 0 || m == 0) return 0;
    if (!xyz1 || !xyz2 || !val || !idx) return GSPN_ERR_ARG;
    if (k > KNN_KMAX) return GSPN_ERR_UNSUPPORTED;
    const long blocks = (long)((m + KNN_WAVES - 1) / KNN_WAVES) * b;
    if (blocks > 0x7FFFFFFFl) return GSPN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(knn_direct_kernel, dim3((unsigned)blocks), dim3(KNN_WAVES * 64), 0, (hipStream_t)stream, b, n, m, k, xyz1, xyz2, val, idx);
    return gspn_launch_status();
}
