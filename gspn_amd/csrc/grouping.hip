// grouping.hip -- tf_ops/grouping on gfx950: ball query, group_point (+grad), group_maxpool
// (+grad), selection sort, and the fused sample_and_group tail used by pointnet_util.
// Reference semantics: tf_ops/grouping/tf_grouping_g.cu (cited per kernel).
#include <math.h>

#include <stdlib.h>

#include "common.h"
#include "csr_gather.h"

// ============================================================================================
// Ball query (reference: tf_grouping_g.cu:6-39)
//
// Reference: one THREAD per query scanning k = 0..n-1 serially with a divergent early break;
// b*256 threads in flight.  Here: one WAVE per query.  The 64 lanes test 64 consecutive data
// points at once, a ballot gives the hit mask, mbcnt gives each hit its slot so hits are written
// in ascending k exactly as the serial scan would, and the wave stops as soon as nsample hits
// exist.  Four 64-point chunks are in flight per iteration to hide the L2 latency of the AoS
// point loads.  The scene (<= 384 KiB) stays L2 resident; the block -> scene map keeps one scene
// on one XCD's L2 (block b runs on XCD b % 8).
//
// Hit test: the reference evaluates  max(sqrtf(s), 1e-20f) < radius  (:27-28).  sqrtf is
// correctly rounded and monotone, so this equals  s < T  with T = the smallest float whose
// sqrtf is >= radius (and "no hit at all" when radius <= 1e-20f).  T is found on the host with
// IEEE sqrtf; the kernel needs no square root and stays bit-exact.
// ============================================================================================
#define BQ_WAVES 4
#define BQ_UNROLL 4

// n_scan < n: only the first n_scan data points are scanned here; a query that has not found nsample points by then is left OPEN --
// its hits so far in row[0 .. cnt), cnt (< nsample) in pts_cnt, no padding -- for ball_query_cont_kernel to resume at n_scan.
__global__ __launch_bounds__(BQ_WAVES * 64) void ball_query_kernel(int b, int n, int m, int n_scan, float thresh, int nsample,
                                                                  const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                                  int* __restrict__ idx, int* __restrict__ pts_cnt, const unsigned char* __restrict__ resolved) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int scene = blockIdx.x % b;            // scene <-> XCD affinity
    const int chunk = blockIdx.x / b;
    const int j = chunk * BQ_WAVES + wave;
    if (j >= m) return;
    if (resolved && resolved[(size_t)scene * m + j]) return;      // answered by the cell-grid kernel (sparse clouds)
    const float* data = xyz1 + (size_t)scene * n * 3;
    const float* qp = xyz2 + ((size_t)scene * m + j) * 3;
    int* row = idx + ((size_t)scene * m + j) * nsample;
    const float qx = qp[0], qy = qp[1], qz = qp[2];

    int cnt = 0, first = 0;
    for (int base = 0; base < n_scan && cnt < nsample; base += 64 * BQ_UNROLL) {
        float s[BQ_UNROLL];
#pragma unroll
        for (int u = 0; u < BQ_UNROLL; ++u) {
            const int k = base + u * 64 + lane;
            float px = 0.f, py = 0.f, pz = 0.f;
            if (k < n_scan) {
                px = data[k * 3 + 0];
                py = data[k * 3 + 1];
                pz = data[k * 3 + 2];
            }
            s[u] = dist2_cuda(qx - px, qy - py, qz - pz);      // (x2-x1): query minus data, :27
        }
#pragma unroll
        for (int u = 0; u < BQ_UNROLL; ++u) {
            const int k = base + u * 64 + lane;
            const bool hit = (k < n_scan) && (s[u] < thresh);
            const unsigned long long mask = __ballot(hit);
            if (mask != 0ull && cnt < nsample) {
                if (cnt == 0) first = base + u * 64 + __builtin_ctzll(mask);
                const int pos = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                if (hit && pos < nsample) row[pos] = k;          // :33
                cnt += __builtin_popcountll(mask);
            }
        }
    }
    cnt = cnt < nsample ? cnt : nsample;
    if (n_scan < n && cnt < nsample) {                           // open: the continuation kernel pads and finishes it
        if (lane == 0) pts_cnt[(size_t)scene * m + j] = cnt;
        return;
    }
    // :29-32 -- unfilled slots repeat the first hit; rows without a hit are zero-filled
    for (int l = cnt + lane; l < nsample; l += 64) row[l] = first;
    if (lane == 0) pts_cnt[(size_t)scene * m + j] = cnt;         // :37
}

// The rest of the scan for the queries the prefix left open -- the regime of sparse clouds: on SURVEY 8(d)'s room scenes a 0.2 m ball
// around a centre on a wall holds fewer than nsample points, so the reference's scan (and a wave per query re-reading the scene from
// L2: 225 us at 8 x 32768 <- 2048) runs to the end of the cloud for 57 % of the queries.  Here a wave owns BQM_QW queries (their
// coordinates, counts and first hits one per lane, fetched with v_readlane) and walks the cloud in steps of 512 points, 8 per lane
// (six 16-byte loads of the AoS cloud, the next step in flight): 3 readlanes + 56 vector instructions per (query, 512 points), the
// cloud read once per BQM_QW queries.  Hits are appended in ascending k exactly as the serial scan appends them (eight ballots give
// every hit its slot).  A wave whose queries are all complete -- every wave, on dense clouds -- leaves after one load.
// Measured (tools/ball_bench.py, 8 x 32768 <- 2048, r = 0.2): S 225 -> 170 us, U 25.9 -> 28.2 us with the 8192-point prefix.  What was
// tried on the way (r04): tiles of 1024 points staged in LDS for 64 queries of a workgroup -- 4.6 us per tile of load latency + two
// barriers that nothing hides (129 us with the tests compiled out); 8 or 16 queries per wave -- a wave issues one instruction per ~5
// cycles, so the serial chain of its queries sets the time (16: 205 us, 8: 167, 4: 156 with a 4096-point prefix).
#ifndef BQM_WAVES
#define BQM_WAVES 4
#endif
#ifndef BQM_QW
#define BQM_QW 4
#endif
__global__ __launch_bounds__(BQM_WAVES * 64) void ball_query_cont_kernel(int b, int n, int m, int n0, float thresh, int nsample,
                                                                        const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                                        int* __restrict__ idx, int* __restrict__ pts_cnt, const unsigned char* __restrict__ resolved) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int scene = blockIdx.x % b;
    const int j0 = ((blockIdx.x / b) * BQM_WAVES + wave) * BQM_QW;
    const float* data = xyz1 + (size_t)scene * n * 3;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    int cnt = nsample, first = 0;
    if (lane < BQM_QW && j0 + lane < m && !(resolved && resolved[(size_t)scene * m + j0 + lane])) {
        const size_t qi = (size_t)scene * m + j0 + lane;
        qx = xyz2[qi * 3 + 0]; qy = xyz2[qi * 3 + 1]; qz = xyz2[qi * 3 + 2];
        cnt = pts_cnt[qi];
        if (cnt > 0 && cnt < nsample) first = idx[qi * nsample];
    }
    const unsigned open0 = (unsigned)__ballot(lane < BQM_QW && cnt < nsample);
    unsigned live = open0;
    if (open0 == 0u) return;                                      // nothing open in this wave: the usual case on dense clouds
    // a lane's 4 consecutive points of a 256-point sub-tile are 48 consecutive bytes of the AoS cloud: three 16-byte loads when the scene
    // starts on a 16-byte boundary (n % 4 == 0 or scene 0), twelve 4-byte loads otherwise; the next sub-tile is in flight while this one
    // is tested.  No LDS, no barrier: the waves drift apart instead of hammering the same cache lines in step.
    const bool vec = (((uintptr_t)data) & 15) == 0;
    // a step = 512 points: lane l holds points k0 + 4l .. + 3 (chunk 0) and k0 + 256 + 4l .. + 3 (chunk 1); the next step's 24 floats
    // are in flight while this step is tested (a wave's steps are a serial chain: 56 of them for 28672 points)
    float4 nb[2][3];                                              // per chunk the 12 floats as they lie in memory: (x0 y0 z0 x1) (y1 z1 x2 y2) (z2 x3 y3 z3)
    auto fetch = [&](int k0) {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const int k = k0 + 256 * ch + 4 * lane;
            if (vec && k + 3 < n) {
                const float4* p = reinterpret_cast<const float4*>(data + (size_t)k * 3);
                nb[ch][0] = p[0]; nb[ch][1] = p[1]; nb[ch][2] = p[2];
            } else {
                float v[12];
#pragma unroll
                for (int e = 0; e < 12; ++e) v[e] = (k + e / 3 < n) ? data[(size_t)k * 3 + e] : INFINITY;      // (a point at infinity is never a hit)
                nb[ch][0] = make_float4(v[0], v[1], v[2], v[3]); nb[ch][1] = make_float4(v[4], v[5], v[6], v[7]); nb[ch][2] = make_float4(v[8], v[9], v[10], v[11]);
            }
        }
    };
    fetch(n0);
    for (int base = n0; base < n && live != 0u; base += 512) {
        float4 px[2], py[2], pz[2];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            px[ch] = make_float4(nb[ch][0].x, nb[ch][0].w, nb[ch][1].z, nb[ch][2].y);
            py[ch] = make_float4(nb[ch][0].y, nb[ch][1].x, nb[ch][1].w, nb[ch][2].z);
            pz[ch] = make_float4(nb[ch][0].z, nb[ch][1].y, nb[ch][2].x, nb[ch][2].w);
        }
        if (base + 512 < n) fetch(base + 512);
        for (unsigned mk = live; mk != 0u; mk &= mk - 1u) {
            const int q = __builtin_ctz(mk);
            const float ax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qx), q)), ay = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qy), q)),
                        az = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qz), q));
            bool h[2][4];
            unsigned long long mm[2][4];
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                h[ch][0] = dist2_cuda(ax - px[ch].x, ay - py[ch].x, az - pz[ch].x) < thresh;       // (x2-x1): query minus data, :27
                h[ch][1] = dist2_cuda(ax - px[ch].y, ay - py[ch].y, az - pz[ch].y) < thresh;
                h[ch][2] = dist2_cuda(ax - px[ch].z, ay - py[ch].z, az - pz[ch].z) < thresh;
                h[ch][3] = dist2_cuda(ax - px[ch].w, ay - py[ch].w, az - pz[ch].w) < thresh;
#pragma unroll
                for (int u = 0; u < 4; ++u) mm[ch][u] = __ballot(h[ch][u]);
            }
            const unsigned long long any0 = (mm[0][0] | mm[0][1]) | (mm[0][2] | mm[0][3]), any1 = (mm[1][0] | mm[1][1]) | (mm[1][2] | mm[1][3]);
            if ((any0 | any1) == 0ull) continue;
            int c = __builtin_amdgcn_readlane(cnt, q);
            int* row = idx + ((size_t)scene * m + j0 + q) * nsample;
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {                      // chunk 0's points all precede chunk 1's
                const unsigned long long any = ch == 0 ? any0 : any1;
                if (any == 0ull) continue;
                const int kb = base + 256 * ch + 4 * lane;
                int pos = c;
#pragma unroll
                for (int u = 0; u < 4; ++u) pos += __builtin_amdgcn_mbcnt_hi((unsigned)(mm[ch][u] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mm[ch][u], 0u));
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (h[ch][u]) { if (pos < nsample) row[pos] = kb + u; ++pos; }          // :33, ascending k
                if (c == 0) {
                    const int fl = __builtin_ctzll(any);
                    const int u0 = ((mm[ch][0] >> fl) & 1ull) ? 0 : (((mm[ch][1] >> fl) & 1ull) ? 1 : (((mm[ch][2] >> fl) & 1ull) ? 2 : 3));
                    if (lane == q) first = base + 256 * ch + 4 * fl + u0;
                }
                c += __builtin_popcountll(mm[ch][0]) + __builtin_popcountll(mm[ch][1]) + __builtin_popcountll(mm[ch][2]) + __builtin_popcountll(mm[ch][3]);
            }
            if (lane == q) cnt = c;
            if (c >= nsample) live &= ~(1u << q);
        }
    }
    for (unsigned mk = open0; mk != 0u; mk &= mk - 1u) {
        const int q = __builtin_ctz(mk);
        int c = __builtin_amdgcn_readlane(cnt, q);
        c = c < nsample ? c : nsample;
        const int f = __builtin_amdgcn_readlane(first, q);
        int* row = idx + ((size_t)scene * m + j0 + q) * nsample;
        for (int l = c + lane; l < nsample; l += 64) row[l] = f;   // :29-32
        if (lane == 0) pts_cnt[(size_t)scene * m + j0 + q] = c;    // :37
    }
}

static float ball_threshold(float radius) {
    // smallest float T with sqrtf(T) >= radius
    if (!(radius > 1e-20f)) return 0.0f;                  // max(d,1e-20f) < radius is never true
    if (isinf(radius)) return INFINITY;
    float t = radius * radius;
    if (isinf(t)) t = 3.402823466e+38f;
    while (t > 0.0f && sqrtf(t) >= radius) t = nextafterf(t, 0.0f);
    while (sqrtf(t) < radius) t = nextafterf(t, INFINITY);
    return t;
}

// exported so the host-side threshold search can be tested without a GPU
extern "C" float gspn_ball_threshold(float radius) { return ball_threshold(radius); }

// ============================================================================================
// Ball query through a cell grid over the DATA points -- the sparse regime (r04).  On room scenes at the model's radii a ball holds
// fewer than nsample points, so the reference's scan reads the whole cloud for most queries (170 us at 8 x 32768 <- 2048 even with the
// continuation kernel above).  When few points are inside a ball, "the first nsample hits in index order" is "ALL hits, sorted by index":
//   build  (one workgroup per scene) bounding box, cells of edge >= 1.01 r (<= 32 per axis), counting sort of the points by cell in LDS:
//          cell_start[] and the points as (x, y, z, index) in cell order -- and the decision whether the cloud is sparse at all: with the
//          volumetric estimate n * (4/3 pi r^3) / box volume >= 8 nsample (a dense cloud: the scan's early exit wins there) the
//          grid is not used and the sort is skipped;
//   query  (one wave per query) the 3 x 3 x 3 block of cells around the query's cell holds every point closer than r: nine contiguous
//          runs of the sorted array, fetched 64 candidates per load; the reference's test (dist2_cuda < T, bit-exact, see above) picks the
//          hits, a ballot appends their indices to a per-wave LDS list, a rank sort puts them in ascending index order, the first nsample
//          are the row.  More than BQG_CAP hits or candidates: the query is left to the scan kernels (resolved[q] stays 0).
// Output identical to the scan (tests/test_gpu_fullsize.py on the S clouds; tools/ball_bench.py compares all kernels).
// ============================================================================================
#define BQG_MAXG 32
#define BQG_CELLS (BQG_MAXG * BQG_MAXG * BQG_MAXG)
#ifndef BQG_CAP
#define BQG_CAP 256
#endif
//                 // hits a wave sorts; more: the query is dense, the scan's early exit is the better tool
#ifndef BQG_MAXCAND
#define BQG_MAXCAND 2048
#endif
//            // candidates of a block a wave is willing to test
struct BallGridHdr { float lo[3]; float inv[3]; int G[3]; int use; int pad[6]; };                  // 64 bytes per scene
static inline size_t bqg_scene_bytes(int n) {       // per scene: header, cell_start, sorted points; the (b, m) `resolved` bytes follow the b scene blocks
    return sizeof(BallGridHdr) + ((sizeof(int) * (size_t)(BQG_CELLS + 1) + 15) / 16) * 16 + 16 * (size_t)n;
}
#define BQG_PTS_OFF (sizeof(BallGridHdr) + ((sizeof(int) * (size_t)(BQG_CELLS + 1) + 15) / 16) * 16)
__global__ __launch_bounds__(1024) void ball_grid_build_kernel(int n, int m, float radius, int nsample, const float* __restrict__ xyz1, char* __restrict__ ws, size_t scene_bytes,
                                                               unsigned char* __restrict__ resolved_all) {
    extern __shared__ int bg_cnt[];                  // [cells]
    __shared__ float s_red[6][16];
    __shared__ int s_wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* data = xyz1 + (size_t)blockIdx.x * n * 3;
    char* base = ws + (size_t)blockIdx.x * scene_bytes;
    BallGridHdr* hdr = reinterpret_cast<BallGridHdr*>(base);
    int* cell_start = reinterpret_cast<int*>(base + sizeof(BallGridHdr));
    float4* pts = reinterpret_cast<float4*>(base + BQG_PTS_OFF);
    unsigned char* resolved = resolved_all + (size_t)blockIdx.x * m;
    for (int j = t; j < m; j += 1024) resolved[j] = 0;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = t; k < n; k += 1024)
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = data[(size_t)k * 3 + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int sft = 32; sft >= 1; sft >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], sft, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], sft, 64)); }
        if (lane == 0) { s_red[a][wave] = lo[a]; s_red[3 + a][wave] = hi[a]; }
    }
    __syncthreads();
    int G[3];
    float inv[3];
    double vol = 1.0;
    bool finite = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = s_red[a][0]; hi[a] = s_red[3 + a][0];
        for (int w = 1; w < 16; ++w) { lo[a] = fminf(lo[a], s_red[a][w]); hi[a] = fmaxf(hi[a], s_red[3 + a][w]); }
        const float ext = hi[a] - lo[a];
        finite = finite && isfinite(ext);
        int g = (ext > 0.f && isfinite(ext)) ? (int)floorf(ext / (1.01f * radius)) : 1;      // cell edge = ext / g >= 1.01 r
        g = g < 1 ? 1 : (g > BQG_MAXG ? BQG_MAXG : g);
        G[a] = g;
        inv[a] = (ext > 0.f && isfinite(ext)) ? (float)g / ext : 0.f;
        vol *= (double)fmaxf(ext, radius);
    }
    // dense or sparse?  expected points per ball under a uniform density in the box (surfaces hold more than this says: the cap below
    // catches those queries one by one)
    const double expect = (double)n * 4.18879 * (double)radius * radius * radius / vol;
    const int use = (finite && n >= 1024 && expect < 8.0 * nsample) ? 1 : 0;
    if (t == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { hdr->lo[a] = lo[a]; hdr->inv[a] = inv[a]; hdr->G[a] = G[a]; }
        hdr->use = use;
    }
    if (!use) return;                                // (uniform across the workgroup)
    const int cells = G[0] * G[1] * G[2];
    for (int c = t; c < cells; c += 1024) bg_cnt[c] = 0;
    __syncthreads();
    auto cell_of = [&](float x, float y, float z) {
        const int cx = min(max((int)floorf((x - lo[0]) * inv[0]), 0), G[0] - 1);
        const int cy = min(max((int)floorf((y - lo[1]) * inv[1]), 0), G[1] - 1);
        const int cz = min(max((int)floorf((z - lo[2]) * inv[2]), 0), G[2] - 1);
        return (cz * G[1] + cy) * G[0] + cx;
    };
    for (int k = t; k < n; k += 1024) atomicAdd(&bg_cnt[cell_of(data[(size_t)k * 3], data[(size_t)k * 3 + 1], data[(size_t)k * 3 + 2])], 1);
    __syncthreads();
    {
        const int per = (cells + 1023) / 1024;
        int local = 0;
        for (int c = t * per; c < min((t + 1) * per, cells); ++c) local += bg_cnt[c];
        int incl = local;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) { const int u = __shfl_up(incl, sft, 64); if (lane >= sft) incl += u; }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        int run = incl - local;
        for (int w = 0; w < wave; ++w) run += s_wsum[w];
        for (int c = t * per; c < min((t + 1) * per, cells); ++c) { const int v = bg_cnt[c]; cell_start[c] = run; bg_cnt[c] = run; run += v; }
        if (t == 1023) cell_start[cells] = run;
    }
    __syncthreads();
    for (int k = t; k < n; k += 1024) {
        const float x = data[(size_t)k * 3], y = data[(size_t)k * 3 + 1], z = data[(size_t)k * 3 + 2];
        pts[atomicAdd(&bg_cnt[cell_of(x, y, z)], 1)] = make_float4(x, y, z, __int_as_float(k));
    }
}
__global__ __launch_bounds__(256) void ball_grid_query_kernel(int b, int n, int m, float thresh, int nsample, const float* __restrict__ xyz2,
                                                              char* __restrict__ ws, size_t scene_bytes, int* __restrict__ idx, int* __restrict__ pts_cnt,
                                                              unsigned char* __restrict__ resolved_all) {
    __shared__ int s_hits[4][BQG_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int scene = blockIdx.x % b;                // scene <-> XCD affinity for the sorted cloud
    const int j = (blockIdx.x / b) * 4 + wave;
    if (j >= m) return;
    char* base = ws + (size_t)scene * scene_bytes;
    const BallGridHdr* hdr = reinterpret_cast<const BallGridHdr*>(base);
    if (!hdr->use) return;
    const int* cell_start = reinterpret_cast<const int*>(base + sizeof(BallGridHdr));
    const float4* pts = reinterpret_cast<const float4*>(base + BQG_PTS_OFF);
    unsigned char* resolved = resolved_all + (size_t)scene * m;
    const float* qp = xyz2 + ((size_t)scene * m + j) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const int G0 = hdr->G[0], G1 = hdr->G[1], G2 = hdr->G[2];
    const int cx = min(max((int)floorf((qx - hdr->lo[0]) * hdr->inv[0]), 0), G0 - 1);
    const int cy = min(max((int)floorf((qy - hdr->lo[1]) * hdr->inv[1]), 0), G1 - 1);
    const int cz = min(max((int)floorf((qz - hdr->lo[2]) * hdr->inv[2]), 0), G2 - 1);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, G0 - 1);
    // lane r < 9: run r = (dz, dy) of the block; its range of the sorted array (empty when the row is outside the grid)
    int rs = 0, re = 0;
    if (lane < 9) {
        const int yy = cy + (lane % 3) - 1, zz = cz + (lane / 3) - 1;
        if (yy >= 0 && yy < G1 && zz >= 0 && zz < G2) {
            const int row = (zz * G1 + yy) * G0;
            rs = cell_start[row + x0];
            re = cell_start[row + x1 + 1];
        }
    }
    // exclusive prefix of the run lengths over lanes 0..8
    const int len = re - rs;
    int incl = len;
#pragma unroll
    for (int sft = 1; sft < 16; sft <<= 1) { const int u = __shfl_up(incl, sft, 64); if (lane >= sft) incl += u; }
    const int total = __builtin_amdgcn_readlane(incl, 8);
    if (total > BQG_MAXCAND) return;                 // a crowded block: leave it to the scan
    int pre[10], st[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) { pre[r] = __builtin_amdgcn_readlane(incl - len, r); st[r] = __builtin_amdgcn_readlane(rs, r); }
    pre[9] = total;
    int* hits = s_hits[wave];
    int cnt = 0;
    for (int t0 = 0; t0 < total; t0 += 64) {
        const int tt = t0 + lane;
        bool hit = false;
        int k = 0;
        if (tt < total) {
            int e = 0;
#pragma unroll
            for (int r = 0; r < 9; ++r) if (tt >= pre[r] && tt < pre[r + 1]) e = st[r] + (tt - pre[r]);
            const float4 p = pts[e];
            hit = dist2_cuda(qx - p.x, qy - p.y, qz - p.z) < thresh;                        // (x2-x1): query minus data, :27
            k = __float_as_int(p.w);
        }
        const unsigned long long mask = __ballot(hit);
        if (mask != 0ull) {
            const int pos = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            if (hit && pos < BQG_CAP) hits[pos] = k;
            cnt += __builtin_popcountll(mask);
        }
    }
    if (cnt > BQG_CAP) return;                       // too many for the sort: a dense ball, the scan's early exit handles it
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    int* row = idx + ((size_t)scene * m + j) * nsample;
    int first = 0;
    if (cnt <= 64) {
        const int v = lane < cnt ? hits[lane] : 0x7FFFFFFF;
        int rank = 0;
        for (int i = 0; i < cnt; ++i) rank += (__builtin_amdgcn_readlane(v, i) < v) ? 1 : 0;
        if (lane < cnt && rank < nsample) row[rank] = v;                                      // :33: ascending k, the first nsample
        const int vmin = wave_max_i32(lane < cnt ? -v : (int)0x80000001);                     // min index = -(max of -v)
        first = cnt > 0 ? -vmin : 0;
    } else {
        int vmin = 0x7FFFFFFF;
        for (int i = lane; i < cnt; i += 64) {
            const int v = hits[i];
            int rank = 0;
            for (int q = 0; q < cnt; ++q) rank += (hits[q] < v) ? 1 : 0;
            if (rank < nsample) row[rank] = v;
            vmin = min(vmin, v);
        }
        first = -wave_max_i32(-vmin);
    }
    const int c = cnt < nsample ? cnt : nsample;
    for (int l = c + lane; l < nsample; l += 64) row[l] = first;                              // :29-32 (rows without a hit are zero-filled)
    if (lane == 0) { pts_cnt[(size_t)scene * m + j] = c; resolved[j] = 1; }
}
extern "C" long gspn_ball_ws_bytes(int b, int n, int m) {
    if (b < 0 || n <= 0 || m < 0) return GSPN_ERR_ARG;
    return (long)((size_t)b * bqg_scene_bytes(n) + (((size_t)b * m + 15) / 16) * 16);
}

// (The LDS-tiled variant north_star sketches -- point tiles staged in LDS, shared by the 8 queries of a workgroup -- was built, is index-exact and
// slower: with the reference's early exit the queries of a workgroup need very different prefixes of an L2-resident cloud.  DESIGN.md 4.2 has the
// numbers; the kernel lives in tools/patches/r06_pruned_alternates.patch.)

static int ball_query_impl(int b, int n, int m, float radius, int nsample, const float* xyz1, const float* xyz2, void* ws,
                           int* idx, int* pts_cnt, void* stream) {
    if (!(radius > 0.0f) || nsample <= 0) return GSPN_ERR_ARG;        // tf_grouping.cpp:101,104
    if (b < 0 || n <= 0 || m < 0) return GSPN_ERR_ARG;
    if (b == 0 || m == 0) return 0;
    const long long blocks = (long long)b * ((m + BQ_WAVES - 1) / BQ_WAVES);
    if (blocks > 0x7FFFFFFFll || (long long)n * 3 > 0x7FFFFFFFll) return GSPN_ERR_UNSUPPORTED;
    // sparse clouds: the cell grid answers the queries whose ball holds few points, the scan kernels below skip those (resolved[])
    const unsigned char* resolved = nullptr;
    static const int cells_on = getenv("GSPN_BALL_CELLS") ? atoi(getenv("GSPN_BALL_CELLS")) : 1;      // (A/B hook)
    if (ws && cells_on && n >= 8192 && ((uintptr_t)ws % 16) == 0 && nsample <= BQG_CAP) {
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&ball_grid_build_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(int) * BQG_CELLS));
            if (ea != hipSuccess) return (int)ea;
            attr_done = true;
        }
        const size_t sb = bqg_scene_bytes(n);
        unsigned char* res = reinterpret_cast<unsigned char*>((char*)ws + (size_t)b * sb);
        hipLaunchKernelGGL(ball_grid_build_kernel, dim3(b), dim3(1024), sizeof(int) * BQG_CELLS, (hipStream_t)stream, n, m, radius, nsample, xyz1, (char*)ws, sb, res);
        hipLaunchKernelGGL(ball_grid_query_kernel, dim3((unsigned)((long long)b * ((m + 3) / 4))), dim3(256), 0, (hipStream_t)stream, b, n, m, ball_threshold(radius), nsample,
                           xyz2, (char*)ws, sb, idx, pts_cnt, res);
        resolved = res;
    }
    // clouds longer than the prefix: wave-per-query scan of the first BQ_PREFIX points (all of a dense cloud's queries end there), then
    // the register-blocked continuation for whatever is still open (GSPN_BALL_PREFIX=0: one pass over the whole cloud, round 3's form)
    static const int prefix = getenv("GSPN_BALL_PREFIX") ? atoi(getenv("GSPN_BALL_PREFIX")) : 8192;
    const int n_scan = (prefix > 0 && n > prefix) ? (prefix + 255) / 256 * 256 : n;
    hipLaunchKernelGGL(ball_query_kernel, dim3((unsigned)blocks), dim3(BQ_WAVES * 64), 0, (hipStream_t)stream,
                       b, n, m, n_scan, ball_threshold(radius), nsample, xyz1, xyz2, idx, pts_cnt, resolved);
    if (n_scan < n) {
        const int per_wg = BQM_WAVES * BQM_QW;
        const long long cblocks = (long long)b * ((m + per_wg - 1) / per_wg);
        hipLaunchKernelGGL(ball_query_cont_kernel, dim3((unsigned)cblocks), dim3(BQM_WAVES * 64), 0, (hipStream_t)stream,
                           b, n, m, n_scan, ball_threshold(radius), nsample, xyz1, xyz2, idx, pts_cnt, resolved);
    }
    return gspn_launch_status();
}
// the drop-in symbol (the reference's argument list: no workspace): prefix scan + continuation
extern "C" int gspn_queryballpoint(int b, int n, int m, float radius, int nsample, const float* xyz1, const float* xyz2,
                                   int* idx, int* pts_cnt, void* stream) {
    return ball_query_impl(b, n, m, radius, nsample, xyz1, xyz2, nullptr, idx, pts_cnt, stream);
}
// the same with a workspace of gspn_ball_ws_bytes(b, n, m) bytes (16-byte aligned): sparse clouds are answered through a cell grid over
// the data points, dense ones (and the queries of a sparse cloud whose ball is crowded) by the scan; identical output
extern "C" int gspn_queryballpoint_ws(int b, int n, int m, float radius, int nsample, const float* xyz1, const float* xyz2, void* ws,
                                      int* idx, int* pts_cnt, void* stream) {
    return ball_query_impl(b, n, m, radius, nsample, xyz1, xyz2, ws, idx, pts_cnt, stream);
}

// ============================================================================================
// group_point / group_point_grad  (tf_grouping_g.cu:43-83)
// The reference gives each query to one thread (stride-ns*c stores between lanes).  Here one
// thread per OUTPUT element, so stores (and the grad loads) are fully coalesced; the gathered
// source rows are c contiguous floats each.
// ============================================================================================
template <int VEC>
__global__ void group_point_kernel(long total, int n, int c, int m_ns, const float* __restrict__ points, const int* __restrict__ idx, float* __restrict__ out) {
    // total = b*m*ns*(c/VEC) work items; item -> (row, chunk)
    const int cv = c / VEC;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / cv;
        const int ch = (int)(i - row * cv) * VEC;
        const long bi = row / m_ns;
        const int ii = idx[row];
        const float* src = points + ((size_t)bi * n + ii) * c + ch;
        float* dst = out + (size_t)row * c + ch;
        if (VEC == 4) *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
        else dst[0] = src[0];
    }
}
template <int VEC>
__global__ void group_point_grad_kernel(long total, int n, int c, int m_ns, const float* __restrict__ grad_out, const int* __restrict__ idx, float* __restrict__ grad_points) {
    const int cv = c / VEC;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / cv;
        const int ch = (int)(i - row * cv) * VEC;
        const long bi = row / m_ns;
        const int ii = idx[row];
        float* dst = grad_points + ((size_t)bi * n + ii) * c + ch;
        const float* src = grad_out + (size_t)row * c + ch;
        if (VEC == 4) {
            const float4 v = *reinterpret_cast<const float4*>(src);
            atomicAdd(dst + 0, v.x); atomicAdd(dst + 1, v.y); atomicAdd(dst + 2, v.z); atomicAdd(dst + 3, v.w);
        } else {
            atomicAdd(dst, src[0]);
        }
    }
}
extern "C" int gspn_grouppoint(int b, int n, int c, int m, int nsample, const float* points, const int* idx, float* out, void* stream) {
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0) return GSPN_ERR_ARG;
    const long rows = (long)b * m * nsample;
    if (rows == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const bool v4 = (c % 4 == 0) && (((uintptr_t)points | (uintptr_t)out) % 16 == 0);
    if (v4) {
        const long total = rows * (c / 4);
        hipLaunchKernelGGL(group_point_kernel<4>, dim3(grid_for(total, 256)), dim3(256), 0, st, total, n, c, m * nsample, points, idx, out);
    } else {
        const long total = rows * c;
        hipLaunchKernelGGL(group_point_kernel<1>, dim3(grid_for(total, 256)), dim3(256), 0, st, total, n, c, m * nsample, points, idx, out);
    }
    return gspn_launch_status();
}
extern "C" int gspn_grouppoint_grad(int b, int n, int c, int m, int nsample, const float* grad_out, const int* idx, float* grad_points, void* stream) {
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0) return GSPN_ERR_ARG;
    if (b == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * n * c, st);     // tf_grouping.cpp:234
    if (e != hipSuccess) return (int)e;
    const long rows = (long)b * m * nsample;
    if (rows == 0) return 0;
    const bool v4 = (c % 4 == 0) && (((uintptr_t)grad_out) % 16 == 0);
    if (v4) {
        const long total = rows * (c / 4);
        hipLaunchKernelGGL(group_point_grad_kernel<4>, dim3(grid_for(total, 256)), dim3(256), 0, st, total, n, c, m * nsample, grad_out, idx, grad_points);
    } else {
        const long total = rows * c;
        hipLaunchKernelGGL(group_point_grad_kernel<1>, dim3(grid_for(total, 256)), dim3(256), 0, st, total, n, c, m * nsample, grad_out, idx, grad_points);
    }
    return gspn_launch_status();
}

// ============================================================================================
// Fused tail of sample_and_group (utils/pointnet_util.py:41-52):
//   out[row, :]  = [ xyz[idx]-new_xyz | points[idx] ]   (xyz_first)   or  [ points[idx] | xyz[idx]-new_xyz ]
// written once, with row pitch ld_out (pad columns zeroed) so the MLP GEMM can read it directly.
// The reference materialises grouped_xyz, a tiled new_xyz, grouped_points and the concat.
// ============================================================================================
__global__ void sa_group_concat_kernel(long total, int n, int c, int m, int nsample, const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                       const float* __restrict__ points, const int* __restrict__ idx, int xyz_first, int ld, float* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / ld;
        const int col = (int)(i - row * ld);
        const long q = row / nsample;              // b*m + j
        const long bi = q / m;
        const int ii = idx[row];
        float v = 0.f;
        const int xc = xyz_first ? col : col - c;          // column inside the xyz part
        const int fc = xyz_first ? col - 3 : col;          // column inside the feature part
        if (xc >= 0 && xc < 3) v = xyz[((size_t)bi * n + ii) * 3 + xc] - new_xyz[q * 3 + xc];   // :41-42
        else if (fc >= 0 && fc < c) v = points[((size_t)bi * n + ii) * c + fc];                  // :46
        out[i] = v;
    }
}
__global__ void sa_group_concat_grad_kernel(long total, int n, int c, int m_ns, const int* __restrict__ idx, int xyz_first, int ld,
                                            const float* __restrict__ grad_out, float* __restrict__ grad_points) {
    // total = rows*c : only the feature columns carry gradient to `points`
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / c;
        const int fc = (int)(i - row * c);
        const long bi = row / m_ns;
        const int ii = idx[row];
        const float g = grad_out[(size_t)row * ld + (xyz_first ? 3 + fc : fc)];
        atomicAdd(grad_points + ((size_t)bi * n + ii) * c + fc, g);
    }
}
// (rows, c) -> (rows, ld) with zero padding columns, one launch (the 16-byte feature rows the gathering first layer reads)
__global__ void pad_rows_kernel(long total, int c, int ld, const float* __restrict__ src, float* __restrict__ dst) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / ld;
        const int k = (int)(i - r * ld);
        dst[i] = k < c ? src[r * c + k] : 0.f;
    }
}
extern "C" int gspn_pad_rows(long rows, int c, int ld, const float* src, float* dst, void* stream) {
    if (rows < 0 || c <= 0 || ld < c || !src || !dst) return GSPN_ERR_ARG;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_for(rows * ld, 256)), dim3(256), 0, (hipStream_t)stream, rows * ld, c, ld, src, dst);
    return gspn_launch_status();
}
extern "C" int gspn_sa_group_concat(int b, int n, int c, int m, int nsample, const float* xyz, const float* new_xyz, const float* points,
                                    const int* idx, int xyz_first, int ld_out, float* out, void* stream) {
    if (b < 0 || n <= 0 || c < 0 || m < 0 || nsample < 0 || ld_out < 3 + c) return GSPN_ERR_ARG;
    if (c > 0 && !points) return GSPN_ERR_ARG;
    const long total = (long)b * m * nsample * ld_out;
    if (total == 0) return 0;
    hipLaunchKernelGGL(sa_group_concat_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       total, n, c, m, nsample, xyz, new_xyz, points, idx, xyz_first, ld_out, out);
    return gspn_launch_status();
}
extern "C" int gspn_sa_group_concat_grad(int b, int n, int c, int m, int nsample, const int* idx, int xyz_first, int ld_out,
                                         const float* grad_out, float* grad_points, void* stream) {
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0 || ld_out < (xyz_first ? 3 : 0) + c) return GSPN_ERR_ARG;
    if (b == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * n * c, st);
    if (e != hipSuccess) return (int)e;
    const long total = (long)b * m * nsample * c;
    if (total == 0) return 0;
    hipLaunchKernelGGL(sa_group_concat_grad_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, total, n, c, m * nsample, idx, xyz_first, ld_out, grad_out, grad_points);
    return gspn_launch_status();
}

// The same gradient as a gather: `order` (b, m*ns) = positions of the flattened idx row sorted by data-point index (ties ascending),
// `offsets` (b, n+1) = each data point's range in it (coordinate-only data, built next to the ball query: geometry.py).  One wave per
// (scene, data point, 64-channel chunk), lane = channel; fixed summation order, no atomics, no zero fill -- the reference's atomicAdd
// has no defined order, so any fixed one is as faithful, and this one makes the whole backward pass reproducible.
__global__ __launch_bounds__(256) void sa_group_concat_grad_csr_kernel(int n, int c, int m_ns, int xyz_first, int ld, const float* __restrict__ grad_out,
                                                                       const int* __restrict__ order, const int* __restrict__ offsets,
                                                                       float* __restrict__ grad_points, long nwaves) {
    const int lane = threadIdx.x & 63;
    const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= nwaves) return;
    const int chunks = (c + 63) / 64;
    const int ch = (int)(wv % chunks);
    const long sp = wv / chunks;                            // scene * n + p
    const int scene = (int)(sp / n), p = (int)(sp - (long)scene * n);
    const int l = ch * 64 + lane;
    const int lc = (l < c ? l : 0) + (xyz_first ? 3 : 0);
    const int* off = offsets + (size_t)scene * (n + 1);
    const int e0 = off[p], e1 = off[p + 1];
    const int* ord = order + (size_t)scene * m_ns;
    const float* gs = grad_out + (size_t)scene * m_ns * ld;
    float acc = 0.f;
    int e = e0;
    for (; e + 3 < e1; e += 4) {
        const float v0 = gs[(size_t)ord[e] * ld + lc], v1 = gs[(size_t)ord[e + 1] * ld + lc];
        const float v2 = gs[(size_t)ord[e + 2] * ld + lc], v3 = gs[(size_t)ord[e + 3] * ld + lc];
        acc += v0; acc += v1; acc += v2; acc += v3;
    }
    for (; e < e1; ++e) acc += gs[(size_t)ord[e] * ld + lc];
    if (l < c) grad_points[((size_t)scene * n + p) * c + l] = acc;
}
// narrow rows (c <= 8: the coordinate / colour gradients of group_point and gather_point): one THREAD per data point walks its list and sums
// the c columns itself -- a wave per data point would run 3 of its 64 lanes.  Same fixed order (ascending grouped position).
__global__ __launch_bounds__(256) void sa_group_concat_grad_csr_narrow_kernel(long total, int n, int c, int m_ns, int col0, int ld, const float* __restrict__ grad_out,
                                                                              const int* __restrict__ order, const int* __restrict__ offsets,
                                                                              float* __restrict__ grad_points) {
    for (long sp = blockIdx.x * (long)blockDim.x + threadIdx.x; sp < total; sp += (long)gridDim.x * blockDim.x) {
        const int scene = (int)(sp / n), p = (int)(sp - (long)scene * n);
        const int* off = offsets + (size_t)scene * (n + 1);
        const int e0 = off[p], e1 = off[p + 1];
        const int* ord = order + (size_t)scene * m_ns;
        const float* gs = grad_out + (size_t)scene * m_ns * ld + col0;
        float acc[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) acc[l] = 0.f;
        for (int e = e0; e < e1; ++e) {
            const float* row = gs + (size_t)ord[e] * ld;
#pragma unroll
            for (int l = 0; l < 8; ++l) if (l < c) acc[l] += row[l];
        }
        float* o = grad_points + (size_t)sp * c;
#pragma unroll
        for (int l = 0; l < 8; ++l) if (l < c) o[l] = acc[l];
    }
}
extern "C" int gspn_sa_group_concat_grad_csr(int b, int n, int c, int m, int nsample, const int* order, const int* offsets, int xyz_first, int ld_out,
                                             const float* grad_out, float* grad_points, void* stream) {
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample < 0 || ld_out < (xyz_first ? 3 : 0) + c || !order || !offsets) return GSPN_ERR_ARG;
    if (b == 0) return 0;
    if (c <= 8 && grad_out && grad_points) {
        const long total = (long)b * n;
        hipLaunchKernelGGL(sa_group_concat_grad_csr_narrow_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, total, n, c, m * nsample,
                           xyz_first ? 3 : 0, ld_out, grad_out, order, offsets, grad_points);
        return gspn_launch_status();
    }
    if (!xyz_first && grad_out && grad_points) {                     // sixteen lanes per data point (csr_gather.h); same sums, same order
        const CsrCopy cp{nullptr, nullptr, 0, 0, 0, 0};
        const int rc = csr_gather16(false, b, n, m * nsample, (long)m * nsample, c, ld_out, 0, grad_out, order, offsets, nullptr, grad_points, cp, (hipStream_t)stream);
        if (rc != GSPN_ERR_UNSUPPORTED) return rc;
    }
    const long nwaves = (long)b * n * ((c + 63) / 64);
    const long blocks = (nwaves + 3) / 4;
    if (blocks > 0x7FFFFFFFl) return GSPN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(sa_group_concat_grad_csr_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, c, m * nsample, xyz_first, ld_out,
                       grad_out, order, offsets, grad_points, nwaves);
    return gspn_launch_status();
}

// ============================================================================================
// group_maxpool / grad  (tf_grouping_g.cu:88-134): fused gather + max over nsample.
// init -10000, strict '>' (first maximum wins).  One thread per (query, channel): consecutive
// lanes read consecutive channels of the same gathered row.
// When no element exceeds -10000 the reference stores a stale index; this build stores the
// first index of the group instead (documented divergence on an input the model never produces).
// ============================================================================================
__global__ void group_maxpool_kernel(long total, int n, int c, int m, int nsample, const float* __restrict__ points, const int* __restrict__ idx,
                                     float* __restrict__ out, int* __restrict__ max_idx) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long q = i / c;                      // b*m + j
        const int l = (int)(i - q * c);
        const long bi = q / m;
        const int* row = idx + q * nsample;
        const float* P = points + (size_t)bi * n * c;
        float best = -10000.0f;
        int besti = row[0];
        for (int k = 0; k < nsample; ++k) {
            const int ii = row[k];
            const float t = P[(size_t)ii * c + l];
            if (t > best) { best = t; besti = ii; }
        }
        out[i] = best;
        max_idx[i] = besti;
    }
}
__global__ void group_maxpool_grad_kernel(long total, int n, int c, int m, const float* __restrict__ grad_out, const int* __restrict__ max_idx, float* __restrict__ grad_points) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long q = i / c;
        const int l = (int)(i - q * c);
        const long bi = q / m;
        atomicAdd(grad_points + ((size_t)bi * n + max_idx[i]) * c + l, grad_out[i]);
    }
}
extern "C" int gspn_groupmaxpool(int b, int n, int c, int m, int nsample, const float* points, const int* idx, float* out, int* max_idx, void* stream) {
    if (b < 0 || n <= 0 || c <= 0 || m < 0 || nsample <= 0) return GSPN_ERR_ARG;
    const long total = (long)b * m * c;
    if (total == 0) return 0;
    hipLaunchKernelGGL(group_maxpool_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, total, n, c, m, nsample, points, idx, out, max_idx);
    return gspn_launch_status();
}
extern "C" int gspn_groupmaxpool_grad(int b, int n, int c, int m, const float* grad_out, const int* max_idx, float* grad_points, void* stream) {
    if (b < 0 || n <= 0 || c <= 0 || m < 0) return GSPN_ERR_ARG;
    if (b == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * n * c, st);     // tf_grouping.cpp:307
    if (e != hipSuccess) return (int)e;
    const long total = (long)b * m * c;
    if (total == 0) return 0;
    hipLaunchKernelGGL(group_maxpool_grad_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, total, n, c, m, grad_out, max_idx, grad_points);
    return gspn_launch_status();
}

// ============================================================================================
// selection_sort (tf_grouping_g.cu:144-184): copy each (b,m) row of n distances, then a partial
// selection sort of its first k slots, swapping values and indices; strict '<' so the lowest
// POSITION wins ties (positions, not original indices: earlier swaps move elements).
// The reference runs one thread per row; here one wave per row: the arg-min over the tail is a
// strided scan + DPP/shuffle reduction on (value, position) keys, the swap is done by lane 0.
// ============================================================================================
__global__ __launch_bounds__(256) void selection_sort_kernel(long rows, int n, int k, const float* __restrict__ dist, int* __restrict__ outi, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long r = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* src = dist + r * n;
    float* p = out + r * n;
    int* pi = outi + r * n;
    for (int s = lane; s < n; s += 64) { p[s] = src[s]; pi[s] = s; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const int kk = k < n ? k : n;
    for (int s = 0; s < kk; ++s) {
        // arg-min over positions s..n-1, lowest position on ties (== the serial strict-'<' scan
        // that starts with min=s)
        float bv = INFINITY;
        int bp = 0x7FFFFFFF;
        bool have = false;
        for (int t = s + lane; t < n; t += 64) {
            const float v = p[t];
            if (!have || v < bv) { bv = v; bp = t; have = true; }
        }
        // reduce (value, position); NaNs are not ordered by '<' in the reference either
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
            const float ov = __shfl_xor(bv, sft, 64);
            const int op = __shfl_xor(bp, sft, 64);
            const int oh = __shfl_xor((int)have, sft, 64);
            const bool take = oh && (!have || ov < bv || (ov == bv && op < bp));
            if (take) { bv = ov; bp = op; have = true; }
        }
        if (lane == 0 && bp != s) {
            const float tv = p[bp]; p[bp] = p[s]; p[s] = tv;
            const int ti = pi[bp]; pi[bp] = pi[s]; pi[s] = ti;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}
extern "C" int gspn_selectionsort(int b, int n, int m, int k, const float* dist, int* outi, float* out, void* stream) {
    if (k <= 0) return GSPN_ERR_ARG;                                     // tf_grouping.cpp:143
    if (b < 0 || n <= 0 || m < 0) return GSPN_ERR_ARG;
    const long rows = (long)b * m;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(selection_sort_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rows, n, k, dist, outi, out);
    return gspn_launch_status();
}

// ============================================================================================
// Inverse lists (CSR) of an index tensor: for idx (b, L) with values in [0, n), order (b, L) = the positions 0..L-1 grouped by value,
// ascending inside a group, and offsets (b, n+1) = where each value's group starts.  This is what the gather-form gradients
// (gspn_sa_group_concat_grad_csr, gspn_fp_concat_grad_csr) walk -- the same result as a stable sort of idx plus a searchsorted, in
// four small kernels instead of a few dozen:
//   count   (one thread per position)  histogram with global atomics
//   scan    (one workgroup per scene)  exclusive scan of the n counts -> offsets, cursors
//   fill    (one thread per position)  position -> its group, in whatever order the atomics produce
//   sort    (one wave per value)       rank sort of the group (a handful to a few dozen entries): the final order is unique, so the
//                                      result does not depend on the atomics' order
// Positions whose value lies outside [0, n) are dropped (offsets[n] is then smaller than L).
// ============================================================================================
__global__ void csr_count_kernel(long total, int L, int n, const int* __restrict__ idx, int* __restrict__ cnt) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = idx[i];
        if ((unsigned)k < (unsigned)n) atomicAdd(&cnt[(i / L) * n + k], 1);
    }
}
__global__ __launch_bounds__(1024) void csr_scan_kernel(int n, int* __restrict__ cnt, int* __restrict__ offsets) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int* c = cnt + (size_t)blockIdx.x * n;
    int* o = offsets + (size_t)blockIdx.x * (n + 1);
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int k = base + t;
        const int v = k < n ? c[k] : 0;
        int incl = v;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) { const int u = __shfl_up(incl, s, 64); if (lane >= s) incl += u; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        const int excl = before + incl - v;
        if (k < n) { o[k] = excl; c[k] = excl; }                  // cnt becomes the fill cursor
        __syncthreads();
        if (t == 1023) carry = excl + v;
        __syncthreads();
    }
    if (t == 0) o[n] = carry;
}
__global__ void csr_fill_kernel(long total, int L, int n, const int* __restrict__ idx, int* __restrict__ cursor, int* __restrict__ tmp) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = idx[i];
        if ((unsigned)k >= (unsigned)n) continue;
        const long scene = i / L;
        const int pos = atomicAdd(&cursor[scene * n + k], 1);
        tmp[scene * L + pos] = (int)(i - scene * L);
    }
}
// count + scan + fill of ONE scene in one workgroup, the histogram and the cursors in LDS (n <= CSR_LDS_MAX_N): no global atomics, one
// launch instead of a memset and three kernels (the default; GSPN_CSR_GLOBAL=1 selects the kernels above for comparison).  Beside the
// captured layers both forms cost about the same (0.07 ms per step for the five lists of a batch, whatever the grid size).
#define CSR_LDS_MAX_N 32768
__global__ __launch_bounds__(1024) void csr_build_lds_kernel(int L, int n, const int* __restrict__ idx, int* __restrict__ offsets, int* __restrict__ tmp) {
    extern __shared__ int csr_cnt[];                 // [n]
    __shared__ int wsum[16];
    __shared__ int carry;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int* ix = idx + (size_t)blockIdx.x * L;
    int* o = offsets + (size_t)blockIdx.x * (n + 1);
    int* tp = tmp + (size_t)blockIdx.x * L;
    for (int k = t; k < n; k += 1024) csr_cnt[k] = 0;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int p = t; p < L; p += 1024) {
        const int k = ix[p];
        if ((unsigned)k < (unsigned)n) atomicAdd(&csr_cnt[k], 1);
    }
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int k = base + t;
        const int v = k < n ? csr_cnt[k] : 0;
        int incl = v;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) { const int u = __shfl_up(incl, s, 64); if (lane >= s) incl += u; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        const int excl = before + incl - v;
        if (k < n) { o[k] = excl; csr_cnt[k] = excl; }            // the count becomes the fill cursor
        __syncthreads();
        if (t == 1023) carry = excl + v;
        __syncthreads();
    }
    if (t == 0) o[n] = carry;
    for (int p = t; p < L; p += 1024) {
        const int k = ix[p];
        if ((unsigned)k >= (unsigned)n) continue;
        tp[atomicAdd(&csr_cnt[k], 1)] = p;
    }
}
// ---- r06: the same build spread over G workgroups per scene (a counting sort by POSITION slices) ------------------------------------------
// csr_build_lds_kernel walks a scene's L positions twice with ONE workgroup: 96 dependent (load, LDS atomic) rounds per pass at the dense
// feature-propagation level (L = 3 x 32768), 80-100 us on 8 CUs -- the longest part of the drop-in gradient symbols with a workspace and of the
// five list builds per batch.  Here workgroup (g, scene) owns the position slice [L g / G, L (g+1) / G):
//   hist   its slice's histogram over the n keys (LDS atomics), written out as hist[scene][g][.]
//   scan   per scene: offsets[k] = sum over keys < k of all slices' counts; hist[scene][g][k] becomes the first slot of slice g inside key k's
//          group (slices in ascending g, so groups are position-ordered BETWEEN slices; within a slice the atomics' order is arbitrary)
//   fill   its slice again: position -> tmp[slot++] (LDS cursors loaded from hist)
// and the per-group sort (csr_sort_kernel) makes the order unique as before: identical output.  Taken when a scene has enough positions to share
// out and the G x n histogram stays small against L (csr_slices).
__global__ __launch_bounds__(1024) void csr_hist_kernel(int L, int n, int G, const int* __restrict__ idx, int* __restrict__ hist) {
    extern __shared__ int csr_cnt[];                 // [n]
    const int t = threadIdx.x, g = blockIdx.x, scene = blockIdx.y;
    const int* ix = idx + (size_t)scene * L;
    const int p0 = (int)((long)L * g / G), p1 = (int)((long)L * (g + 1) / G);
    for (int k = t; k < n; k += 1024) csr_cnt[k] = 0;
    __syncthreads();
    for (int p = p0 + t; p < p1; p += 1024) {
        const int k = ix[p];
        if ((unsigned)k < (unsigned)n) atomicAdd(&csr_cnt[k], 1);
    }
    __syncthreads();
    int* h = hist + ((size_t)scene * G + g) * n;
    for (int k = t; k < n; k += 1024) h[k] = csr_cnt[k];
}
__global__ __launch_bounds__(1024) void csr_slice_scan_kernel(int n, int G, int* __restrict__ hist, int* __restrict__ offsets) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int* h = hist + (size_t)blockIdx.x * G * n;
    int* o = offsets + (size_t)blockIdx.x * (n + 1);
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int k = base + t;
        int c[32];                                   // the G <= 32 slices' counts of key k: all loads in flight at once (coalesced over k for every g)
#pragma unroll
        for (int g = 0; g < 32; ++g) c[g] = (g < G && k < n) ? h[(size_t)g * n + k] : 0;
        int v = 0;
#pragma unroll
        for (int g = 0; g < 32; ++g) v += c[g];
        int incl = v;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) { const int u = __shfl_up(incl, s, 64); if (lane >= s) incl += u; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        const int excl = before + incl - v;
        if (k < n) {
            o[k] = excl;
            int run = excl;
#pragma unroll
            for (int g = 0; g < 32; ++g) if (g < G) { h[(size_t)g * n + k] = run; run += c[g]; }
        }
        __syncthreads();
        if (t == 1023) carry = excl + v;
        __syncthreads();
    }
    if (t == 0) o[n] = carry;
}
__global__ __launch_bounds__(1024) void csr_slice_fill_kernel(int L, int n, int G, const int* __restrict__ idx, const int* __restrict__ hist, int* __restrict__ tmp) {
    extern __shared__ int csr_cnt[];                 // [n]: this slice's next free slot per key
    const int t = threadIdx.x, g = blockIdx.x, scene = blockIdx.y;
    const int* ix = idx + (size_t)scene * L;
    int* tp = tmp + (size_t)scene * L;
    const int* h = hist + ((size_t)scene * G + g) * n;
    const int p0 = (int)((long)L * g / G), p1 = (int)((long)L * (g + 1) / G);
    for (int k = t; k < n; k += 1024) csr_cnt[k] = h[k];
    __syncthreads();
    for (int p = p0 + t; p < p1; p += 1024) {
        const int k = ix[p];
        if ((unsigned)k >= (unsigned)n) continue;
        tp[atomicAdd(&csr_cnt[k], 1)] = p;
    }
}
// slices per scene of the spread-out build: about 4096 positions each, at most 32, and the G x n histogram no larger than 2 L ints; 1 = the one-workgroup kernel
static inline int csr_slices(int L, int n) {
    static const int on = getenv("GSPN_CSR_SLICES") ? atoi(getenv("GSPN_CSR_SLICES")) : -1;      // tuning hook: 0 / 1 = off, k > 1 = force k
    if (on == 0 || on == 1) return 1;
    long G = on > 1 ? on : L / 4096;
    if (G > 32) G = 32;
    while (G > 1 && G * (long)n > 2L * L) --G;
    return G < 8 ? 1 : (int)G;                   // (a few slices do not pay for the two extra launches: measured 24.9 -> 24.0 us with 4 at SA level 2, 108 -> 122 at SA level 1 where n = 32768)
}
// one WAVE per value: rank sort of its group (the positions are distinct, so rank = number of smaller entries), tmp -> order.
// Groups of up to 64 entries (the usual case: a handful to a few dozen) never touch memory again -- one entry per lane, compared through
// v_readlane; longer groups count against the group re-read from L2.
__global__ __launch_bounds__(256) void csr_sort_kernel(long nwaves, int L, int n, const int* __restrict__ offsets, int* __restrict__ tmp,
                                                       int* __restrict__ order) {
    const int lane = threadIdx.x & 63;
    const long w = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (w >= nwaves) return;
    const long scene = w / n;
    const int k = (int)(w - scene * n);
    const int* o = offsets + scene * (n + 1);
    const int e0 = __builtin_amdgcn_readfirstlane(o[k]), e1 = __builtin_amdgcn_readfirstlane(o[k + 1]);
    const int cnt = e1 - e0;
    const int* src = tmp + scene * L + e0;
    int* dst = order + scene * L + e0;
    if (cnt <= 64) {
        const int v = lane < cnt ? src[lane] : 0x7FFFFFFF;
        int rank = 0;
        for (int j = 0; j < cnt; ++j) rank += (__builtin_amdgcn_readlane(v, j) < v) ? 1 : 0;
        if (lane < cnt) dst[rank] = v;
        return;
    }
    // Longer groups (clustered clouds: a sparse point that is the nearest neighbour of thousands of dense points -- on SURVEY 8d's room
    // scenes one group in ten is longer than 64 and the longest hold a few thousand positions; round 3's count-the-smaller-entries
    // loop was quadratic there: 420 us per launch): a stable LSD radix sort by the wave, 6 bits per pass.  The 64 bins of a pass live
    // in LDS, one per lane; a tile's 64 entries find their stable rank among the entries with the same digit with six ballots (no
    // loop over digits), so a pass costs ~40 instructions per 64 entries.  The passes ping-pong between the two buffers the caller
    // already provides (tmp and order); an even number of passes ends with a copy.
    __shared__ int s_bins[4][64];
    int* bins = s_bins[threadIdx.x >> 6];
    int* bufA = const_cast<int*>(src);
    int* bufB = dst;
    const int nbits = 32 - __builtin_clz((unsigned)(L > 1 ? L - 1 : 1));
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    // The phases below exchange data between lanes through bins[] (LDS) and the ping-pong buffers (global): every phase boundary is an explicit
    // wave barrier + workgroup-scope fence (lockstep execution of a wave is an implementation detail a compiler may reschedule around).
#define CSR_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
    for (int shift = 0; shift < nbits; shift += 6) {
        bins[lane] = 0;
        CSR_WAVE_SYNC();
        for (int i0 = 0; i0 < cnt; i0 += 64) {
            const int i = i0 + lane;
            if (i < cnt) atomicAdd(&bins[(bufA[i] >> shift) & 63], 1);
        }
        CSR_WAVE_SYNC();
        // exclusive scan of the 64 bins (one per lane)
        const int c = bins[lane];
        int incl = c;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) { const int u = __shfl_up(incl, sft, 64); if (lane >= sft) incl += u; }
        CSR_WAVE_SYNC();
        bins[lane] = incl - c;
        CSR_WAVE_SYNC();
        for (int i0 = 0; i0 < cnt; i0 += 64) {
            const int i = i0 + lane;
            const bool valid = i < cnt;
            const int v = valid ? bufA[i] : 0;
            const int d = (v >> shift) & 63;
            unsigned long long eq = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < 6; ++bit) {
                const unsigned long long m = __ballot((d >> bit) & 1);
                eq &= ((d >> bit) & 1) ? m : ~m;
            }
            int base = 0;
            if (valid) base = bins[d];
            CSR_WAVE_SYNC();                          // every lane has read its digit's cursor before the digit's highest lane moves it
            if (valid) {
                const int rank = __builtin_popcountll(eq & lt);
                bufB[base + rank] = v;
                if ((eq >> lane) == 1ull) bins[d] = base + __builtin_popcountll(eq);      // the highest lane of the digit moves its cursor on
            }
            CSR_WAVE_SYNC();
        }
        __threadfence_block();                        // this pass's stores are complete before other lanes of the wave read them back
        CSR_WAVE_SYNC();
        int* t = bufA; bufA = bufB; bufB = t;
    }
#undef CSR_WAVE_SYNC
    if (bufA != dst)                                                                      // (bufA holds the sorted group after the last swap)
        for (int i = lane; i < cnt; i += 64) dst[i] = bufA[i];
}
// work: b*n ints (counts, then cursors) followed by b*L ints (the unsorted groups) followed by b*G*n ints (the slices' histograms, r06)
extern "C" long gspn_inverse_lists_work_ints(int b, int L, int n) {
    if (b < 0 || L < 0 || n <= 0) return GSPN_ERR_ARG;
    const int G = n <= CSR_LDS_MAX_N ? csr_slices(L, n) : 1;
    return (long)b * n + (long)b * L + (G > 1 ? (long)b * G * n : 0L);
}
extern "C" int gspn_inverse_lists(int b, int L, int n, const int* idx, int* work, int* order, int* offsets, void* stream) {
    if (b < 0 || L < 0 || n <= 0) return GSPN_ERR_ARG;
    if (b == 0) return 0;
    if (!idx || !work || !order || !offsets) return GSPN_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)b * L;
    static const bool global_atomics = getenv("GSPN_CSR_GLOBAL") != nullptr;          // comparison switch, read once
    if (n <= CSR_LDS_MAX_N && !global_atomics) {
        int* tmp = work + (size_t)b * n;
        const long nwaves = (long)b * n;
        if ((nwaves + 3) / 4 > 0x7FFFFFFFl) return GSPN_ERR_UNSUPPORTED;
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&csr_build_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                (int)(sizeof(int) * CSR_LDS_MAX_N));
            if (ea != hipSuccess) return (int)ea;
            attr_done = true;
        }
        const int G = csr_slices(L, n);
        if (G > 1) {
            int* hist = tmp + (size_t)b * L;
            static bool attr2_done = false;
            if (!attr2_done) {
                hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&csr_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(int) * CSR_LDS_MAX_N));
                hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&csr_slice_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(int) * CSR_LDS_MAX_N));
                if (e1 != hipSuccess) return (int)e1;
                if (e2 != hipSuccess) return (int)e2;
                attr2_done = true;
            }
            hipLaunchKernelGGL(csr_hist_kernel, dim3(G, b), dim3(1024), sizeof(int) * (size_t)n, st, L, n, G, idx, hist);
            hipLaunchKernelGGL(csr_slice_scan_kernel, dim3(b), dim3(1024), 0, st, n, G, hist, offsets);
            hipLaunchKernelGGL(csr_slice_fill_kernel, dim3(G, b), dim3(1024), sizeof(int) * (size_t)n, st, L, n, G, idx, hist, tmp);
            if (total > 0) hipLaunchKernelGGL(csr_sort_kernel, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, st, nwaves, L, n, offsets, tmp, order);
            return gspn_launch_status();
        }
        const size_t lds = gspn_claim_lds(4, reinterpret_cast<const void*>(&csr_build_lds_kernel), sizeof(int) * (size_t)n);
        hipLaunchKernelGGL(csr_build_lds_kernel, dim3(b), dim3(1024), lds, st, L, n, idx, offsets, tmp);
        if (total > 0) hipLaunchKernelGGL(csr_sort_kernel, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, st, nwaves, L, n, offsets, tmp, order);
        return gspn_launch_status();
    }
    hipError_t e = hipMemsetAsync(work, 0, sizeof(int) * (size_t)b * n, st);
    if (e != hipSuccess) return (int)e;
    if (total > 0) hipLaunchKernelGGL(csr_count_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, total, L, n, idx, work);
    hipLaunchKernelGGL(csr_scan_kernel, dim3(b), dim3(1024), 0, st, n, work, offsets);
    if (total > 0) {
        int* tmp = work + (size_t)b * n;
        const long nwaves = (long)b * n;
        if ((nwaves + 3) / 4 > 0x7FFFFFFFl) return GSPN_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(csr_fill_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, total, L, n, idx, work, tmp);
        hipLaunchKernelGGL(csr_sort_kernel, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, st, nwaves, L, n, offsets, tmp, order);
    }
    return gspn_launch_status();
}

// ============================================================================================
// Many small device-to-device copies in ONE launch (refilling the persistent geometry buffers a captured step reads: ~30 tensors of a
// few KB to a few MB per batch -- 30 copy kernels otherwise).  Segment table by value in the kernel arguments; every segment is split
// into 64 KB pieces and the pieces are spread over the grid.  16-byte aligned segments move as uint4, the rest bytewise.
// ============================================================================================
#define MC_MAX 40
struct McTable {
    const char* src[MC_MAX];
    char* dst[MC_MAX];
    long bytes[MC_MAX];
    long first_piece[MC_MAX + 1];          // prefix sum of ceil(bytes / 65536)
    int n;
};
__global__ __launch_bounds__(256) void multi_copy_kernel(McTable t) {
    const long npieces = t.first_piece[t.n];
    for (long p = blockIdx.x; p < npieces; p += gridDim.x) {
        int s = 0;
        while (s + 1 < t.n && t.first_piece[s + 1] <= p) ++s;
        const long off = (p - t.first_piece[s]) * 65536L;
        const long len = t.bytes[s] - off < 65536L ? t.bytes[s] - off : 65536L;
        const char* a = t.src[s] + off;
        char* d = t.dst[s] + off;
        if ((((uintptr_t)a | (uintptr_t)d) & 15) == 0) {
            const long nv = len >> 4;
            for (long i = threadIdx.x; i < nv; i += 256) reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(a)[i];
            for (long i = (nv << 4) + threadIdx.x; i < len; i += 256) d[i] = a[i];
        } else {
            for (long i = threadIdx.x; i < len; i += 256) d[i] = a[i];
        }
    }
}
extern "C" int gspn_multi_copy(int n, const void* const* src, void* const* dst, const long* bytes, void* stream) {
    if (n < 0 || (n > 0 && (!src || !dst || !bytes))) return GSPN_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n; base += MC_MAX) {
        McTable t;
        t.n = 0;
        t.first_piece[0] = 0;
        for (int i = base; i < n && i < base + MC_MAX; ++i) {
            if (bytes[i] < 0 || (bytes[i] > 0 && (!src[i] || !dst[i]))) return GSPN_ERR_ARG;
            if (bytes[i] == 0) continue;
            t.src[t.n] = static_cast<const char*>(src[i]);
            t.dst[t.n] = static_cast<char*>(dst[i]);
            t.bytes[t.n] = bytes[i];
            t.first_piece[t.n + 1] = t.first_piece[t.n] + (bytes[i] + 65535) / 65536;
            ++t.n;
        }
        if (t.n == 0) continue;
        const long np = t.first_piece[t.n];
        hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)(np < 2048 ? np : 2048)), dim3(256), 0, st, t);
    }
    return gspn_launch_status();
}
