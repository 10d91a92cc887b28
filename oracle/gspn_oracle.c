/*
 * gspn_oracle.c -- CPU restatement of the reference (ericyi/GSPN) set-abstraction ops.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gspn_amd/ may import, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / the timed CPU baseline, never as the product path.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - three_interpolate / three_interpolate_grad : PINNED against the reference's own
 *     compiled code (oracle/_ref, built from tf_ops/3d_interpolation/interpolate.cpp whose
 *     loop bodies are byte-identical to tf_interpolate.cpp:107-153) and against
 *     tests/golden/interp_ref_*.npz generated from it.
 *   - every other function: PARITY UNPINNED by the reference.  The reference ships no golden
 *     vector, its ops need TensorFlow + nvcc (absent here) and cannot be executed.  They are
 *     pinned only by (a) this line-by-line restatement, (b) an independent NumPy brute force
 *     (tests/ref_numpy.py) and (c) the invariants listed in SURVEY.md section 8(c).
 *
 * Arithmetic policy (SURVEY.md Appendix A "FMA policy"):
 *   CUDA-derived ops (FPS, ball query, nn_distance GPU twin) were built by nvcc -O2 with the
 *   default --fmad=true.  nvcc/NVVM contracts  a*a + b*b + c*c  left operand first:
 *       t = b*b;  t = fma(a,a,t);  t = fma(c,c,t)
 *   (LLVM DAGCombiner::visitFADD folds (fadd (fmul x y) z) -> fma(x,y,z) before the commuted
 *   form).  GSPN_DIST_POLICY selects:  2 = that form (default), 1 = fma(c,c,fma(b,b,a*a))
 *   (the form SURVEY.md guessed), 0 = unfused, 3 = fma(a,a,b*b)+c*c (what hipcc makes of the reference's
 *   expression: the form that is bit-equal to oracle/_ref's hipcc builds of the reference sources).  The HIP kernels use the same switch.
 *   Host-derived ops (three_nn, three_interpolate, nnsearch CPU twin) were built by g++ -O2
 *   without -mfma: unfused fp32, left to right.
 *
 * Build:  gcc -O2 -ffp-contract=off -fPIC -shared gspn_oracle.c -o libgspn_oracle.so -lm
 *         (-fopenmp optional: only the *_mt entry points use it)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef GSPN_DIST_POLICY
#define GSPN_DIST_POLICY 2
#endif

/* squared distance as the nvcc-built kernels compute it (a=dx, b=dy, c=dz) */
static inline float dist2_cuda(float a, float b, float c) {
#if GSPN_DIST_POLICY == 2
    return fmaf(c, c, fmaf(a, a, b * b));
#elif GSPN_DIST_POLICY == 1
    return fmaf(c, c, fmaf(b, b, a * a));
#elif GSPN_DIST_POLICY == 3
    return fmaf(a, a, b * b) + c * c;      /* hipcc's contraction of the reference expression on gfx950 (oracle/_ref builds) */
#else
    return (a * a + b * b) + c * c;
#endif
}
/* squared distance as the g++-built host loops compute it */
static inline float dist2_host(float a, float b, float c) {
    return (a * a + b * b) + c * c;
}

int oracle_dist_policy(void) { return GSPN_DIST_POLICY; }

/* bench.py's cpu_baseline "all cores" leg: OpenMP over scene x query.  Off by default (the faithful single-thread form of
 * the reference's own CPU loops); every output element is computed by one thread with the same arithmetic, so results do not change.
 * The scatter-add gradients keep their sequential order inside a scene and spread over scenes only. */
static int g_mt = 0;
void oracle_set_mt(int on) { g_mt = on ? 1 : 0; }

/* debug: expose the two distance forms so GPU arithmetic can be bit-compared */
void oracle_dist2(int cnt, const float *p, const float *q, float *out_cuda, float *out_host) {
    for (int i = 0; i < cnt; i++) {
        float a = q[i * 3 + 0] - p[i * 3 + 0];
        float b = q[i * 3 + 1] - p[i * 3 + 1];
        float c = q[i * 3 + 2] - p[i * 3 + 2];
        out_cuda[i] = dist2_cuda(a, b, c);
        out_host[i] = dist2_host(a, b, c);
    }
}

/* ------------------------------------------------------------------------------------------
 * A1  farthest point sampling -- tf_ops/sampling/tf_sampling_g.cu:105-170
 * Literal simulation of the 512-thread block: per-thread strided scan with strict '>'
 * (:146-149), then the 9-level tree in which the LOWER slot keeps ties (:153-164).
 * Winner key = (d2 desc, k mod 512 asc, k asc).
 * ---------------------------------------------------------------------------------------- */
#define FPS_BLOCK 512
static void fps_one(int n, int m, const float *xyz, float *temp, int *idxs) {
    float dists[FPS_BLOCK];
    int dists_i[FPS_BLOCK];
    if (m <= 0) return;                                   /* :106-107 */
    int old = 0;
    idxs[0] = old;                                        /* :114-116 */
    for (int j = 0; j < n; j++) temp[j] = 1e38f;          /* :117-119 */
    for (int j = 1; j < m; j++) {
        float x1 = xyz[old * 3 + 0], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2]; /* :127-129 */
        for (int t = 0; t < FPS_BLOCK; t++) {
            int besti = 0;                                /* :125-126 */
            float best = -1.f;
            for (int k = t; k < n; k += FPS_BLOCK) {      /* :130 */
                float td = temp[k];
                float x2 = xyz[k * 3 + 0], y2 = xyz[k * 3 + 1], z2 = xyz[k * 3 + 2];
                float d = dist2_cuda(x2 - x1, y2 - y1, z2 - z1);   /* :142 */
                float d2 = fminf(d, td);                  /* :143 (CUDA min(float,float)) */
                if (d2 != td) temp[k] = d2;               /* :144-145 */
                if (d2 > best) { best = d2; besti = k; }  /* :146-149 */
            }
            dists[t] = best;
            dists_i[t] = besti;
        }
        for (int u = 0; (1 << u) < FPS_BLOCK; u++) {      /* :153-164 */
            for (int t = 0; t < (FPS_BLOCK >> (u + 1)); t++) {
                int i1 = (t * 2) << u, i2 = (t * 2 + 1) << u;
                if (dists[i1] < dists[i2]) { dists[i1] = dists[i2]; dists_i[i1] = dists_i[i2]; }
            }
        }
        old = dists_i[0];                                 /* :166 */
        idxs[j] = old;
    }
}
void oracle_farthest_point_sample(int b, int n, int m, const float *inp, int *out) {
    float *temp = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < b; i++) fps_one(n, m, inp + (size_t)i * n * 3, temp, out + (size_t)i * m);
    free(temp);
}
/* indices AND the reference's scratch: temp_out (b, n) = min squared distance of every point to the chosen set after the run -- what
 * tf_sampling_g.cu:117-145 leaves in `temp + blockIdx.x*n` (compared with oracle/_ref's scratch in tests/test_gpu_policy3.py) */
void oracle_farthest_point_sample_temp(int b, int n, int m, const float *inp, int *out, float *temp_out) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < b; i++) fps_one(n, m, inp + (size_t)i * n * 3, temp_out + (size_t)i * n, out + (size_t)i * m);
}
/* same result, scenes spread over OpenMP threads (cpu_baseline "all cores") */
void oracle_farthest_point_sample_mt(int b, int n, int m, const float *inp, int *out) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < b; i++) {
        float *temp = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
        fps_one(n, m, inp + (size_t)i * n * 3, temp, out + (size_t)i * m);
        free(temp);
    }
}

/* A2 gather_point -- tf_sampling_g.cu:172-181 */
void oracle_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out) {
    for (int i = 0; i < b; i++)
        for (int j = 0; j < m; j++) {
            int a = idx[i * m + j];
            for (int l = 0; l < 3; l++) out[((size_t)i * m + j) * 3 + l] = inp[((size_t)i * n + a) * 3 + l];
        }
}
/* A3 gather_point_grad -- tf_sampling_g.cu:183-192 (+ memset tf_sampling.cpp:174).
 * The reference adds with atomics (order free); this sums in ascending j. */
void oracle_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx, float *inp_g) {
    memset(inp_g, 0, sizeof(float) * (size_t)b * n * 3);
    for (int i = 0; i < b; i++)
        for (int j = 0; j < m; j++) {
            int a = idx[i * m + j];
            for (int l = 0; l < 3; l++) inp_g[((size_t)i * n + a) * 3 + l] += out_g[((size_t)i * m + j) * 3 + l];
        }
}

/* ------------------------------------------------------------------------------------------
 * A4 query_ball_point -- tf_ops/grouping/tf_grouping_g.cu:6-39
 * visited (may be NULL) receives, per query, the number of data points the reference loop
 * reads before its break (index of the nsample-th hit + 1, else n): the L of SURVEY 8(d).
 * Rows with no hit are zero-filled (the reference leaves them uninitialised).
 * ---------------------------------------------------------------------------------------- */
static void ball_one(int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2,
                     int *idx, int *pts_cnt, int *visited) {
#pragma omp parallel for schedule(dynamic, 16) if (g_mt)
    for (int j = 0; j < m; j++) {
        int cnt = 0, k;
        for (int l = 0; l < nsample; l++) idx[j * nsample + l] = 0;
        for (k = 0; k < n; ++k) {
            if (cnt == nsample) break;                                        /* :19-20 */
            float x2 = xyz2[j * 3 + 0], y2 = xyz2[j * 3 + 1], z2 = xyz2[j * 3 + 2];
            float x1 = xyz1[k * 3 + 0], y1 = xyz1[k * 3 + 1], z1 = xyz1[k * 3 + 2];
            float d = fmaxf(sqrtf(dist2_cuda(x2 - x1, y2 - y1, z2 - z1)), 1e-20f);   /* :27 */
            if (d < radius) {                                                 /* :28 */
                if (cnt == 0)
                    for (int l = 0; l < nsample; ++l) idx[j * nsample + l] = k;   /* :29-32 */
                idx[j * nsample + cnt] = k;                                   /* :33 */
                cnt += 1;
            }
        }
        pts_cnt[j] = cnt;                                                     /* :37 */
        if (visited) visited[j] = k;
    }
}
void oracle_query_ball_point(int b, int n, int m, float radius, int nsample, const float *xyz1,
                             const float *xyz2, int *idx, int *pts_cnt, int *visited) {
    for (int i = 0; i < b; i++)
        ball_one(n, m, radius, nsample, xyz1 + (size_t)i * n * 3, xyz2 + (size_t)i * m * 3,
                 idx + (size_t)i * m * nsample, pts_cnt + (size_t)i * m, visited ? visited + (size_t)i * m : 0);
}
void oracle_query_ball_point_mt(int b, int n, int m, float radius, int nsample, const float *xyz1,
                                const float *xyz2, int *idx, int *pts_cnt, int *visited) {
#pragma omp parallel for schedule(dynamic, 1) if (!g_mt)
    for (int i = 0; i < b; i++)
        ball_one(n, m, radius, nsample, xyz1 + (size_t)i * n * 3, xyz2 + (size_t)i * m * 3,
                 idx + (size_t)i * m * nsample, pts_cnt + (size_t)i * m, visited ? visited + (size_t)i * m : 0);
}

/* A5 group_point -- tf_grouping_g.cu:43-60 */
void oracle_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx, float *out) {
#pragma omp parallel for collapse(2) schedule(static) if (g_mt)
    for (int i = 0; i < b; i++)
        for (int j = 0; j < m; j++)
            for (int k = 0; k < nsample; k++) {
                int ii = idx[((size_t)i * m + j) * nsample + k];
                for (int l = 0; l < c; l++)
                    out[(((size_t)i * m + j) * nsample + k) * c + l] = points[((size_t)i * n + ii) * c + l];
            }
}
/* A6 group_point_grad -- tf_grouping_g.cu:66-83 (+ memset tf_grouping.cpp:234); sums in (j,k) order */
void oracle_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx, float *grad_points) {
    memset(grad_points, 0, sizeof(float) * (size_t)b * n * c);
#pragma omp parallel for schedule(static) if (g_mt)
    for (int i = 0; i < b; i++)
        for (int j = 0; j < m; j++)
            for (int k = 0; k < nsample; k++) {
                int ii = idx[((size_t)i * m + j) * nsample + k];
                for (int l = 0; l < c; l++)
                    grad_points[((size_t)i * n + ii) * c + l] += grad_out[(((size_t)i * m + j) * nsample + k) * c + l];
            }
}

/* A7 group_maxpool -- tf_grouping_g.cu:88-114: init -10000, strict '>', first max wins.
 * one_max_idx is not reset between (j,l) iterations in the reference (:97-98,103-110): when no
 * element exceeds -10000 the previous value leaks through.  Restated literally per "thread":
 * thread `index` visits j = index, index+256, ... and carries one_max_idx across them. */
void oracle_group_maxpool(int b, int n, int c, int m, int nsample, const float *points, const int *idx, float *out, int *max_idx) {
    for (int i = 0; i < b; i++) {
        const float *P = points + (size_t)i * n * c;
        const int *I = idx + (size_t)i * m * nsample;
        float *O = out + (size_t)i * m * c;
        int *MI = max_idx + (size_t)i * m * c;
        for (int index = 0; index < 256; index++) {
            int one_max_idx = 0; /* uninitialised in the reference; 0 here */
            for (int j = index; j < m; j += 256)
                for (int l = 0; l < c; l++) {
                    float max_feat = -10000.0f;
                    for (int k = 0; k < nsample; k++) {
                        int ii = I[j * nsample + k];
                        float t = P[(size_t)ii * c + l];
                        if (t > max_feat) { max_feat = t; one_max_idx = ii; }
                    }
                    O[j * c + l] = max_feat;
                    MI[j * c + l] = one_max_idx;
                }
        }
    }
}
/* group_maxpool_grad -- tf_grouping_g.cu:119-134 (+ memset tf_grouping.cpp:307) */
void oracle_group_maxpool_grad(int b, int n, int c, int m, const float *grad_out, const int *max_idx, float *grad_points) {
    memset(grad_points, 0, sizeof(float) * (size_t)b * n * c);
    for (int i = 0; i < b; i++)
        for (int j = 0; j < m; j++)
            for (int l = 0; l < c; l++) {
                int ii = max_idx[((size_t)i * m + j) * c + l];
                grad_points[((size_t)i * n + ii) * c + l] += grad_out[((size_t)i * m + j) * c + l];
            }
}

/* A7 selection_sort -- tf_grouping_g.cu:144-184: copy, then partial selection sort of the
 * first k slots per row (strict '<' => lowest position wins ties). */
void oracle_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out) {
    for (size_t r = 0; r < (size_t)b * m; r++) {
        float *p = out + r * n;
        int *pi = outi + r * n;
        for (int s = 0; s < n; s++) { p[s] = dist[r * n + s]; pi[s] = s; }
        for (int s = 0; s < k && s < n; s++) {
            int mn = s;
            for (int t = s + 1; t < n; t++)
                if (p[t] < p[mn]) mn = t;
            if (mn != s) {
                float tmp = p[mn]; p[mn] = p[s]; p[s] = tmp;
                int ti = pi[mn]; pi[mn] = pi[s]; pi[s] = ti;
            }
        }
    }
}
/* knn_point -- tf_grouping.py:71-96: dist = reduce_sum((xyz1-xyz2)**2,-1) (TF fp32, operands
 * data - query, summed x,y,z left to right, unfused) then selection_sort, first k columns. */
void oracle_knn_dist(int b, int n, int c, int m, const float *xyz1, const float *xyz2, float *dist) {
    for (int i = 0; i < b; i++)
        for (int j = 0; j < m; j++)
            for (int s = 0; s < n; s++) {
                float acc = 0.f;
                for (int l = 0; l < c; l++) {
                    float d = xyz1[((size_t)i * n + s) * c + l] - xyz2[((size_t)i * m + j) * c + l];
                    float sq = d * d;
                    acc = (l == 0) ? sq : acc + sq;
                }
                dist[((size_t)i * m + j) * n + s] = acc;
            }
}

/* ------------------------------------------------------------------------------------------
 * A8 three_nn -- tf_ops/3d_interpolation/tf_interpolate.cpp:60-103 (host arithmetic: unfused
 * fp32, compared as double; strict '<' cascade; init 1e40 -> +inf when cast to float)
 * ---------------------------------------------------------------------------------------- */
void oracle_three_nn(int b, int n, int m, const float *xyz1_, const float *xyz2_, float *dist_, int *idx_) {
#pragma omp parallel for collapse(2) schedule(static) if (g_mt)
    for (int i = 0; i < b; ++i) {
        for (int j = 0; j < n; ++j) {
            const float *xyz1 = xyz1_ + (size_t)i * n * 3, *xyz2 = xyz2_ + (size_t)i * m * 3;
            float *dist = dist_ + (size_t)i * n * 3;
            int *idx = idx_ + (size_t)i * n * 3;
            float x1 = xyz1[j * 3 + 0], y1 = xyz1[j * 3 + 1], z1 = xyz1[j * 3 + 2];
            double best1 = 1e40, best2 = 1e40, best3 = 1e40;
            int besti1 = 0, besti2 = 0, besti3 = 0;
            for (int k = 0; k < m; ++k) {
                float x2 = xyz2[k * 3 + 0], y2 = xyz2[k * 3 + 1], z2 = xyz2[k * 3 + 2];
                double d = dist2_host(x2 - x1, y2 - y1, z2 - z1);             /* :74 */
                if (d < best1) { best3 = best2; besti3 = besti2; best2 = best1; besti2 = besti1; best1 = d; besti1 = k; }
                else if (d < best2) { best3 = best2; besti3 = besti2; best2 = d; besti2 = k; }
                else if (d < best3) { best3 = d; besti3 = k; }
            }
            dist[j * 3] = (float)best1; idx[j * 3] = besti1;
            dist[j * 3 + 1] = (float)best2; idx[j * 3 + 1] = besti2;
            dist[j * 3 + 2] = (float)best3; idx[j * 3 + 2] = besti3;
        }
    }
}
/* A9 three_interpolate -- tf_interpolate.cpp:107-127 */
void oracle_three_interpolate(int b, int m, int c, int n, const float *points_, const int *idx_, const float *weight_, float *out_) {
#pragma omp parallel for collapse(2) schedule(static) if (g_mt)
    for (int i = 0; i < b; ++i) {
        for (int j = 0; j < n; ++j) {
            const float *points = points_ + (size_t)i * m * c, *weight = weight_ + (size_t)i * n * 3;
            const int *idx = idx_ + (size_t)i * n * 3;
            float *out = out_ + (size_t)i * n * c;
            float w1 = weight[j * 3], w2 = weight[j * 3 + 1], w3 = weight[j * 3 + 2];
            int i1 = idx[j * 3], i2 = idx[j * 3 + 1], i3 = idx[j * 3 + 2];
            for (int l = 0; l < c; ++l)
                out[(size_t)j * c + l] = points[(size_t)i1 * c + l] * w1 + points[(size_t)i2 * c + l] * w2 + points[(size_t)i3 * c + l] * w3;
        }
    }
}
/* A10 three_interpolate_grad -- tf_interpolate.cpp:131-153 (+ memset :258) */
void oracle_three_interpolate_grad(int b, int n, int c, int m, const float *grad_out_, const int *idx_, const float *weight_, float *grad_points_) {
    memset(grad_points_, 0, sizeof(float) * (size_t)b * m * c);
#pragma omp parallel for schedule(static) if (g_mt)
    for (int i = 0; i < b; ++i) {
        const float *grad_out = grad_out_ + (size_t)i * n * c, *weight = weight_ + (size_t)i * n * 3;
        const int *idx = idx_ + (size_t)i * n * 3;
        float *grad_points = grad_points_ + (size_t)i * m * c;
        for (int j = 0; j < n; ++j) {
            float w1 = weight[j * 3], w2 = weight[j * 3 + 1], w3 = weight[j * 3 + 2];
            int i1 = idx[j * 3], i2 = idx[j * 3 + 1], i3 = idx[j * 3 + 2];
            for (int l = 0; l < c; ++l) {
                grad_points[(size_t)i1 * c + l] += grad_out[(size_t)j * c + l] * w1;
                grad_points[(size_t)i2 * c + l] += grad_out[(size_t)j * c + l] * w2;
                grad_points[(size_t)i3 * c + l] += grad_out[(size_t)j * c + l] * w3;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * A11 nn_distance
 *   GPU twin  tf_ops/nn_distance/tf_nndistance_g.cu:5-127: 512-point tiles; inside a tile
 *             strict '<' in ascending k with a reset at k==0 (:27,:103); across tiles the
 *             stored result is replaced only when strictly greater (:119).  Net effect:
 *             global argmin, lowest index wins.  Arithmetic: p2-p1, FMA-contracted sum.
 *   CPU twin  tf_nndistance.cpp:21-43 (nnsearch): same rule, unfused arithmetic.
 * ---------------------------------------------------------------------------------------- */
static void nn_dir(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx, int cuda) {
    for (int i = 0; i < b; i++)
        for (int j = 0; j < n; j++) {
            float x1 = xyz1[((size_t)i * n + j) * 3 + 0], y1 = xyz1[((size_t)i * n + j) * 3 + 1], z1 = xyz1[((size_t)i * n + j) * 3 + 2];
            float best = 0; int besti = 0;
            for (int k = 0; k < m; k++) {
                float x2 = xyz2[((size_t)i * m + k) * 3 + 0] - x1;
                float y2 = xyz2[((size_t)i * m + k) * 3 + 1] - y1;
                float z2 = xyz2[((size_t)i * m + k) * 3 + 2] - z1;
                float d = cuda ? dist2_cuda(x2, y2, z2) : dist2_host(x2, y2, z2);
                if (k == 0 || d < best) { best = d; besti = k; }
            }
            dist[(size_t)i * n + j] = best;
            idx[(size_t)i * n + j] = besti;
        }
}
void oracle_nn_distance(int b, int n, const float *xyz1, int m, const float *xyz2, float *dist1, int *idx1, float *dist2, int *idx2) {
    nn_dir(b, n, m, xyz1, xyz2, dist1, idx1, 1);
    nn_dir(b, m, n, xyz2, xyz1, dist2, idx2, 1);
}
void oracle_nn_distance_cputwin(int b, int n, const float *xyz1, int m, const float *xyz2, float *dist1, int *idx1, float *dist2, int *idx2) {
    nn_dir(b, n, m, xyz1, xyz2, dist1, idx1, 0);
    nn_dir(b, m, n, xyz2, xyz1, dist2, idx2, 0);
}
/* nn_distance_grad -- tf_nndistance_g.cu:132-157 / CPU twin tf_nndistance.cpp:126-163 (sequential) */
void oracle_nn_distance_grad(int b, int n, const float *xyz1, int m, const float *xyz2, const float *grad_dist1, const int *idx1,
                             const float *grad_dist2, const int *idx2, float *grad_xyz1, float *grad_xyz2) {
    memset(grad_xyz1, 0, sizeof(float) * (size_t)b * n * 3);
    memset(grad_xyz2, 0, sizeof(float) * (size_t)b * m * 3);
    for (int i = 0; i < b; i++) {
        for (int j = 0; j < n; j++) {
            const float *p1 = xyz1 + ((size_t)i * n + j) * 3;
            int j2 = idx1[(size_t)i * n + j];
            const float *p2 = xyz2 + ((size_t)i * m + j2) * 3;
            float g = grad_dist1[(size_t)i * n + j] * 2;
            for (int l = 0; l < 3; l++) {
                grad_xyz1[((size_t)i * n + j) * 3 + l] += g * (p1[l] - p2[l]);
                grad_xyz2[((size_t)i * m + j2) * 3 + l] -= (g * (p1[l] - p2[l]));
            }
        }
        for (int j = 0; j < m; j++) {
            const float *p1 = xyz2 + ((size_t)i * m + j) * 3;
            int j2 = idx2[(size_t)i * m + j];
            const float *p2 = xyz1 + ((size_t)i * n + j2) * 3;
            float g = grad_dist2[(size_t)i * m + j] * 2;
            for (int l = 0; l < 3; l++) {
                grad_xyz2[((size_t)i * m + j) * 3 + l] += g * (p1[l] - p2[l]);
                grad_xyz1[((size_t)i * n + j2) * 3 + l] -= (g * (p1[l] - p2[l]));
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * prob_sample -- tf_ops/sampling/tf_sampling_g.cu:7-104.
 * cumsumKernel: per row, tiles of 2048*4 = 8192 elements.  Inside a tile the prefix sums are
 * formed by a work-efficient tree over groups of 4 consecutive elements; the tile's running
 * offset `runningsum` is carried with a Kahan-style correction `runningsum2` (:81-84).
 * binarysearchKernel (:90-104): r = inpr * dataset[n-1] (row total), then the lowest index
 * whose cumulative sum is >= r using the power-of-two descent (:96-101).
 * The exact rounding of the in-tile tree is restated in oracle_cumsum below.
 * ---------------------------------------------------------------------------------------- */
void oracle_cumsum(int b, int n, const float *inp, float *out) {
    enum { BlockSize = 2048, Pad = 32 };
    static float buffer4[BlockSize * 4];
    static float buffer[BlockSize + (BlockSize >> 5)];
    for (int i = 0; i < b; i++) {
        float runningsum = 0, runningsum2 = 0;
        for (int j = 0; j < n; j += BlockSize * 4) {
            int n24_i = (BlockSize * 4 < n - j) ? BlockSize * 4 : n - j;
            int n24 = (n24_i + 3) & ~3;
            int n2 = n24 >> 2;
            /* :19-37 : per group of 4, running sums v1..v4 stored; buffer[group] = group total */
            for (int k4 = 0; k4 < n24_i; k4 += 4) {
                int k = k4;
                if (k + 3 < n24_i) {
                    float v1 = inp[(size_t)i * n + j + k];
                    float v2 = inp[(size_t)i * n + j + k + 1]; v2 += v1;
                    float v3 = inp[(size_t)i * n + j + k + 2]; float v4 = inp[(size_t)i * n + j + k + 3]; v4 += v3;
                    v3 += v2; v4 += v2;
                    buffer4[k] = v1; buffer4[k + 1] = v2; buffer4[k + 2] = v3; buffer4[k + 3] = v4;
                    buffer[(k >> 2) + (k >> (2 + 5))] = v4;
                } else {
                    float v = 0;
                    for (int k2 = k; k2 < n24_i; k2++) { v += inp[(size_t)i * n + j + k2]; buffer4[k2] = v; }
                    for (int k2 = n24_i; k2 < n24; k2++) buffer4[k2] = v;
                    buffer[(k >> 2) + (k >> (2 + 5))] = v;
                }
            }
            /* :38-49 : up-sweep */
            int u = 0;
            for (; (2 << u) <= n2; u++) {
                for (int k = 0; k < (n2 >> (u + 1)); k++) {
                    int i1 = (((k << 1) + 2) << u) - 1;
                    int i2 = (((k << 1) + 1) << u) - 1;
                    i1 += i1 >> 5; i2 += i2 >> 5;
                    buffer[i1] += buffer[i2];
                }
            }
            u--;
            /* :51-62 : down-sweep */
            for (; u >= 0; u--) {
                for (int k = 0; k < ((n2 - (1 << u)) >> (u + 1)); k++) {
                    int i1 = (((k << 1) + 3) << u) - 1;
                    int i2 = (((k << 1) + 2) << u) - 1;
                    i1 += i1 >> 5; i2 += i2 >> 5;
                    buffer[i1] += buffer[i2];
                }
            }
            /* :64-72 */
            for (int k = 0; k < n24; k++) {
                if (k != 0) {
                    int k2 = ((k >> 2) - 1) + (((k >> 2) - 1) >> 5);
                    if ((k >> 2) >= 1) buffer4[k] += buffer[k2];
                }
            }
            /* :74-76 */
            for (int k = 0; k < n24_i; k++) out[(size_t)i * n + j + k] = buffer4[k] + runningsum;
            /* :77-84 */
            float t = buffer[(n2 - 1) + ((n2 - 1) >> 5)] + runningsum2;
            float r2 = runningsum + t;
            runningsum2 = t - (r2 - runningsum);
            runningsum = r2;
        }
    }
}
void oracle_binary_search(int b, int n, int m, const float *dataset, const float *query, int *result) {
    int base = 1;
    while (base < n) base <<= 1;                                     /* :91-93 */
    for (int i = 0; i < b; i++)
        for (int j = 0; j < m; j++) {
            float q = query[(size_t)i * m + j] * dataset[(size_t)i * n + n - 1];   /* :96 */
            int r = n - 1;
            for (int k = base; k >= 1; k >>= 1)                       /* :98-100 */
                if (r >= k && dataset[(size_t)i * n + r - k] >= q) r -= k;
            result[(size_t)i * m + j] = r;
        }
}
