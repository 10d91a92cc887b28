// sampling.hip -- tf_ops/sampling on gfx950: farthest point sampling, gather_point (+grad),
// prob_sample.  Reference semantics: tf_ops/sampling/tf_sampling_g.cu (cited per kernel).
#include "fps_common.h"

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

#ifdef FPS_PROFILE
__device__ long long g_fps_prof[8];
__device__ long long g_fps_tl[16 * 8];
#define FPS_TICK(i) do { const long long _n = clock64(); prof[i] += _n - tprev; tprev = _n; if (j == 100 && blockIdx.x == 0 && lane == 0) g_fps_tl[wave * 8 + i] = _n; } while (0)
#else
#define FPS_TICK(i) do {} while (0)
#endif

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
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);     // provably wave-uniform (keeps loop state in SGPRs)
    const int rho = t >> 1;                       // residue class k mod 512 of this thread
    const int kbase = (t & 1) * P * 512 + rho;    // slot p holds point kbase + p*512
    const float* xyz = inp + (size_t)blockIdx.x * n * 3;
    int* o = out + (size_t)blockIdx.x * m;

    v2f x[P / 2], y[P / 2], z[ZLDS ? 1 : P / 2], td[P / 2];
    // clamped, unconditional loads first (all in flight together), masking afterwards
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int k = kbase + p * 512;
        const int kc = k < n ? k : n - 1;
        float px = xyz[kc * 3 + 0], py = xyz[kc * 3 + 1], pz = xyz[kc * 3 + 2];
        const float d0 = k < n ? 1e38f : -1.0f;   // :117-119; padding never wins (real candidates are >= 0)
        px = k < n ? px : 0.f; py = k < n ? py : 0.f; pz = k < n ? pz : 0.f;
        // detach x/y/z from the dwordx3 load tuple so the allocator may place them independently
        asm("" : "+v"(px), "+v"(py), "+v"(pz));
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
                const v2f d = dist2_cuda_v2(dx, dy, dz);                                   // contraction policy: fps_common.h, :142
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

// ============================================================================================
// Cell-based FPS: same result as fps_resident_kernel, far fewer serial rounds.
//
// Host pre-pass (gspn_amd/tf_sampling.py): the scene is sorted into 16 spatial cells of equal size
// (Morton order), and inside a cell by the reference tie rank (k mod 512, k).  Wave w owns cell w;
// lane l / slot p hold the cell's (l*P+p)-th point, so "lowest (lane, slot)" is still the reference
// tie order inside a wave, and waves are compared through rank(k) of their candidates.
//
// Two exact shortcuts on top of the resident design:
//   * culling  -- a wave skips a new centre c entirely when its bounding box is farther from c than
//     its current max min-distance: no point of it can change (conservative fp32 margin);
//   * batching -- one synchronisation round publishes every wave's best point and an upper bound on
//     its runner-up; all waves then accept the longest prefix c1 > c2 > ... (by key) such that each
//     c_i beats the runner-up bounds of the accepted waves and its min-distance is untouched by the
//     earlier picks of the round (dist2(c_i, c_j) >= td[c_i], evaluated exactly as the update would).
//     Those are provably the next picks of the sequential algorithm, so several picks cost one
//     barrier.  The degenerate tail (max min-distance == 0: every point already chosen or a
//     duplicate of one) repeats the rank-minimal point, like the reference.
// ============================================================================================
#ifndef FPS_AMAX
#define FPS_AMAX 8        // max picks accepted per round (<= 16: four bits per accepted wave in a 64-bit list)
#endif
#ifdef FPS_PROFILE
__device__ long long g_cell_prof[32];
__device__ int g_cell_waves[16 * 4];      // per wave of scene 0: {applies, refreshes, -, -}
__device__ long long g_cell_tl[3 * 16 * 8];   // timestamps of rounds 200, 300, 400 of scene 0: [round][wave][tick]
#if FPS_PROFILE == 2     // timeline only: no per-phase accumulators (they cost registers the P=32 kernel then spills)
#define CELL_TICK(i) do { if (blockIdx.x == 0 && lane == 0 && (tlround == 200 || tlround == 300 || tlround == 400)) g_cell_tl[((tlround / 100 - 2) * 16 + wave) * 8 + i] = clock64(); } while (0)
#else
#define CELL_TICK(i) do { const long long _n = clock64(); cprof[i] += _n - ctprev; ctprev = _n; \
        if (blockIdx.x == 0 && lane == 0 && (tlround == 200 || tlround == 300 || tlround == 400)) g_cell_tl[((tlround / 100 - 2) * 16 + wave) * 8 + i] = _n; } while (0)
#endif
#else
#define CELL_TICK(i) do {} while (0)
#endif

template <int P, bool ZLDS>
__global__ __launch_bounds__(FPS_T) void fps_cell_kernel(int n, int m, int csz, const float* __restrict__ sxyz, const int* __restrict__ perm,
                                                         const float* __restrict__ inp0, int inp0_stride, int* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [0,1024)   : candidates, 2 buffers x 16 waves x {int4 {v bits, x, y, z}, int4 {sorted position, bound bits, -, -}}
    // [1024,1056): batch record written by wave 0 (two int4)
    // [1536,1536+64*P): per-wave row for the winning lane's min-distances (refresh)
    // [4096,..)  : z plane, float4 [P/4][1024]   (ZLDS only)
    int4* s_cand = reinterpret_cast<int4*>(smem);
    float* s_trow = reinterpret_cast<float*>(smem + 1536);
    v4f* s_z = reinterpret_cast<v4f*>(smem + 4096);
    constexpr int G = FpsGroup<P>::G;
    constexpr int NG = FpsGroup<P>::NG;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);     // provably wave-uniform (keeps loop state in SGPRs)
    const float* xyz = sxyz + (size_t)blockIdx.x * n * 3;
    const int* pm = perm + (size_t)blockIdx.x * n;
    int* o = out + (size_t)blockIdx.x * m;
    const int cbase = wave * csz;                         // first sorted position of this wave's cell

    v2f x[P / 2], y[P / 2], z[ZLDS ? 1 : P / 2], td[P / 2];
    float bx0 = 3e38f, bx1 = -3e38f, by0 = 3e38f, by1 = -3e38f, bz0 = 3e38f, bz1 = -3e38f;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int q = lane * P + p;
        const int pos = cbase + q;
        const bool live = q < csz && pos < n;
        const int pc = live ? pos : 0;           // clamped, unconditional loads: all in flight together
        float px = xyz[pc * 3 + 0], py = xyz[pc * 3 + 1], pz = xyz[pc * 3 + 2];
        // branch-free masking keeps the prologue one basic block, so the scheduler can issue all loads before the first use
        bx0 = fminf(bx0, live ? px : 3e38f); bx1 = fmaxf(bx1, live ? px : -3e38f);
        by0 = fminf(by0, live ? py : 3e38f); by1 = fmaxf(by1, live ? py : -3e38f);
        bz0 = fminf(bz0, live ? pz : 3e38f); bz1 = fmaxf(bz1, live ? pz : -3e38f);
        const float d0 = live ? 1e38f : -1.0f;
        px = live ? px : 0.f; py = live ? py : 0.f; pz = live ? pz : 0.f;
        asm("" : "+v"(px), "+v"(py), "+v"(pz));
        x[p >> 1][p & 1] = px;
        y[p >> 1][p & 1] = py;
        td[p >> 1][p & 1] = d0;
        if (ZLDS) reinterpret_cast<float*>(s_z)[((p >> 2) * FPS_T + t) * 4 + (p & 3)] = pz;
        else z[p >> 1][p & 1] = pz;
    }
    // wave bounding box (uniform)
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        bx0 = fminf(bx0, __shfl_xor(bx0, s, 64)); bx1 = fmaxf(bx1, __shfl_xor(bx1, s, 64));
        by0 = fminf(by0, __shfl_xor(by0, s, 64)); by1 = fmaxf(by1, __shfl_xor(by1, s, 64));
        bz0 = fminf(bz0, __shfl_xor(bz0, s, 64)); bz1 = fmaxf(bz1, __shfl_xor(bz1, s, 64));
    }
    bx0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bx0))); bx1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bx1)));
    by0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(by0))); by1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(by1)));
    bz0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bz0))); bz1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bz1)));
    if (t == 0) o[0] = 0;                                          // tf_sampling_g.cu:114-116
    __syncthreads();

    // Centres to apply at the top of a round are NOT kept in registers (VGPR budget): they are re-read from the candidate
    // records of the previous exchange.  acc_list packs the accepted wave ids (4 bits each), abuf is that exchange's buffer.
    // Round 0 applies original point 0, parked in record 0 of buffer 1.
    const float* p0 = inp0 + (size_t)blockIdx.x * inp0_stride;
    if (t == 0) s_cand[2 * FPS_W + 0] = make_int4(0, __float_as_int(p0[0]), __float_as_int(p0[1]), __float_as_int(p0[2]));
    __syncthreads();
    unsigned long long acc_list = 0;       // (4 bits per accepted wave: up to 16 picks a round)
    int abuf = 2 * FPS_W;
    int nacc = 1;
    int j = 1;                      // number of outputs written so far
    int wmax = __float_as_int(1e38f);          // wave's current max min-distance (bits); padding-only waves settle at -1.0f
    bool need = true;                          // the cached candidate must be recomputed
    unsigned myconf = 0;                       // candidates of the last exchange that would lower the cached candidate's min-distance
    // cached candidate of this wave
    int cv = 0, ck = 0, cpos = 0, cbound = NEG_ONE_BITS;
    float cfx = 0.f, cfy = 0.f, cfz = 0.f;
    int round = 0;
    int termk = 0;
#ifdef FPS_PROFILE
#if FPS_PROFILE != 2
    long long cprof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long ctprev = clock64();
#endif
    int napplied = 0, nrefresh = 0;
    int tlround = 0;
#endif

    int4* s_batch = s_cand + 4 * FPS_W;       // batch record broadcast by wave 0: {nacc, acc_list, terminal, j_new}
    while (j < m) {
#ifdef FPS_PROFILE
        tlround = round;
#endif
        // ---- apply the accepted centres (culled per wave) ----
        // culling, all centres of the round at once: lane u (< nacc) tests centre u against this wave's bounding box.
        // Conservative in fp32: every point of the wave has dist2 >= L*(1-1e-5); a wave whose max min-distance is below that cannot change.
        unsigned todo;
        {
            const int u = lane < nacc ? lane : 0;
            const int4 cc = s_cand[abuf + (int)((acc_list >> (4 * u)) & 15ull) * 2];
            const float ccx = __int_as_float(cc.y), ccy = __int_as_float(cc.z), ccz = __int_as_float(cc.w);
            const float ex = fmaxf(fmaxf(bx0 - ccx, ccx - bx1), 0.f);
            const float ey = fmaxf(fmaxf(by0 - ccy, ccy - by1), 0.f);
            const float ez = fmaxf(fmaxf(bz0 - ccz, ccz - bz1), 0.f);
            const float L = (ex * ex + ey * ey + ez * ez) * 0.99999f;
            todo = (unsigned)__ballot(lane < nacc && !(L > __int_as_float(wmax)) && wmax >= 0);
        }
        while (todo) {
            const int ci = __builtin_ctz(todo);
            todo &= todo - 1;
            const int4 cc = s_cand[abuf + (int)((acc_list >> (4 * ci)) & 15ull) * 2];          // uniform address: LDS broadcast
            const float cx = __int_as_float(__builtin_amdgcn_readfirstlane(cc.y));
            const float cy = __int_as_float(__builtin_amdgcn_readfirstlane(cc.z));
            const float cz = __int_as_float(__builtin_amdgcn_readfirstlane(cc.w));
            if constexpr (ZLDS) {
                // z quads are double-buffered: the ds_read_b128 of quad i+1 is issued before quad i is consumed, otherwise
                // a lone active wave exposes the full LDS latency 8 times per pass
                v4f zq = s_z[t];
#pragma unroll
                for (int qd = 0; qd < P / 4; ++qd) {
                    v4f zn = zq;
                    if (qd + 1 < P / 4) zn = s_z[(qd + 1) * FPS_T + t];
                    asm("" : "+v"(zq));
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int pp = qd * 2 + h;
                        const v2f zz = h == 0 ? zq.xy : zq.zw;
                        const v2f dx = x[pp] - cx, dy = y[pp] - cy, dz = zz - cz;
                        const v2f d = dist2_cuda_v2(dx, dy, dz);                                   // contraction policy: fps_common.h
                        td[pp][0] = vmin_f32(d[0], td[pp][0]);
                        td[pp][1] = vmin_f32(d[1], td[pp][1]);
                    }
                    zq = zn;
                    __builtin_amdgcn_sched_barrier(0);     // keeps the z double-buffer from being hoisted across quads (VGPR budget: no spill)
                }
            } else {
#pragma unroll
                for (int pp = 0; pp < P / 2; ++pp) {
                    const v2f dx = x[pp] - cx, dy = y[pp] - cy, dz = z[pp] - cz;
                    const v2f d = dist2_cuda_v2(dx, dy, dz);                                       // contraction policy: fps_common.h
                    td[pp][0] = vmin_f32(d[0], td[pp][0]);
                    td[pp][1] = vmin_f32(d[1], td[pp][1]);
                }
            }
#ifdef FPS_PROFILE
            ++napplied;
#endif
        }
        CELL_TICK(0);

        // ---- refresh this wave's candidate (best point + runner-up bound) -- only when the cached one is gone: it was picked, or
        //      one of the centres just applied lies closer to it than its min-distance.  Otherwise it is still the wave's best point
        //      with the same value (the other points only went down), and the cached runner-up bound is still an upper bound ----
        if (need) {
            int best = NEG_ONE_BITS;             // per-lane max over its slots
#pragma unroll
            for (int pp = 0; pp < P / 2; ++pp) best = vmax3_i32(best, __float_as_int(td[pp][0]), __float_as_int(td[pp][1]));
            wmax = wave_max_i32(best);
            const int lw = __builtin_ctzll(__ballot(best == wmax));
            // runner-up bound, part 1: best of the other lanes
            const int s1 = wave_max_i32(lane == lw ? NEG_ONE_BITS : best);
            // The winning lane spills its P min-distances to a per-wave LDS row; the wave then finds the winning slot (lowest
            // slot on ties) and the runner-up inside that lane with one value per lane -- no per-lane compare chains, which
            // the 128-VGPR budget of P=32 cannot afford.
            float* trow = s_trow + wave * P;
            if (lane == lw) {
#pragma unroll
                for (int p = 0; p < P; p += 2) *reinterpret_cast<v2f*>(trow + p) = td[p >> 1];
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            const int tv = lane < P ? __float_as_int(trow[lane]) : NEG_ONE_BITS;
            const int fp = __builtin_ctzll(__ballot(tv == wmax));
            const int s2 = wave_max_i32(lane == fp ? NEG_ONE_BITS : tv);
            const int qw = fp / G, iw = fp % G;
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
            if (ZLDS) fz = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(
                        reinterpret_cast<const float*>(s_z)[((fp >> 2) * FPS_T + wave * 64 + lw) * 4 + (fp & 3)])));
            const int pos = cbase + lw * P + fp;
            cv = wmax;
            cpos = (wmax >= 0 && pos < n) ? pos : 0;
            {
                // a VECTOR load on purpose: a uniform address would be scalarised to s_load, whose completion is counted on
                // lgkmcnt together with every LDS access -- each ds_read/ds_write wait below would then also wait for HBM
                int voff = 0;
                asm volatile("" : "+v"(voff));
                ck = pm[cpos + voff];      // original index: consumed only when this candidate is accepted (latency hidden behind the exchange)
            }
            cbound = max(s1, s2);
            cfx = fx; cfy = fy; cfz = fz;
            need = false;
#ifdef FPS_PROFILE
            ++nrefresh;
#endif
        }
        CELL_TICK(1);
        const int buf = (round & 1) * 2 * FPS_W;
        ++round;
        if (lane == 0) {
            s_cand[buf + wave * 2 + 0] = make_int4(cv, __float_as_int(cfx), __float_as_int(cfy), __float_as_int(cfz));
            s_cand[buf + wave * 2 + 1] = make_int4(cpos, cbound, 0, 0);
        }
        __syncthreads();
        CELL_TICK(2);

        // ---- batch selection, distributed.  The accepted picks are a prefix of the candidates in value order, so whether the
        //      candidate of rank p is acceptable GIVEN that ranks 0..p-1 are accepted can be decided by its own wave: none of the
        //      better candidates may spoil it (dist2(own, other) < own min-distance, evaluated exactly as the update would) and it
        //      must beat the runner-up bounds of all of them.  Every wave publishes {rank, acceptable}; after a second barrier the
        //      batch is the run of acceptable ranks 0, 1, ... -- two 16-lane OR reductions, done redundantly by every wave ----
        int2* s_info = reinterpret_cast<int2*>(smem + 1280) + (round & 1) * FPS_W;
        const int l15 = lane & 15;
        const int4 mine = s_cand[buf + l15 * 2];                  // candidate l15: {v, x, y, z}
        const int4 mine2 = s_cand[buf + l15 * 2 + 1];             // {sorted position, runner-up bound, -, -}
        int myrank;
        {
            const unsigned better = (unsigned)(__ballot(mine.x > cv) & 0xFFFFull);
            const unsigned equal = (unsigned)(__ballot(mine.x == cv && l15 != wave) & 0xFFFFull);
            // dist2(point = own candidate, centre = candidate l15), exactly as the update would evaluate it
            const float dd = dist2_cuda(cfx - __int_as_float(mine.y), cfy - __int_as_float(mine.z), cfz - __int_as_float(mine.w));
            const unsigned conf = (unsigned)(__ballot(l15 != wave && mine.x >= 0 && dd < __int_as_float(cv)) & 0xFFFFull);
            myconf = conf;
            // largest runner-up bound among the better candidates
            const int bb = __builtin_amdgcn_readfirstlane(row_max_i32(((better >> l15) & 1u) ? mine2.y : NEG_ONE_BITS));
            int cnt = __builtin_popcount(better);
            if (cv < 0) cnt = 64;                                 // empty / padding-only waves never rank
            myrank = cnt;
            const bool okw = cnt == 0 || ((conf & better) == 0u && cv > bb);
            if (lane == 0) s_info[wave] = make_int2(cnt | ((equal != 0u && cv >= 0) ? 256 : 0) | (okw ? 512 : 0), (int)conf);
        }
        __syncthreads();
        CELL_TICK(4);
        const int2 info = s_info[l15];
        const int rk = info.x & 255;
        const bool slow = __ballot((info.x & 256) && rk < FPS_AMAX) != 0ull;    // equal values among the leaders (rare)
        if (!slow) {
            const int okb = __builtin_amdgcn_readfirstlane(row_or_i32((rk < FPS_AMAX && (info.x & 512)) ? (1 << rk) : 0));
            const unsigned wl = (unsigned)__builtin_amdgcn_readfirstlane(row_or_i32(rk < (FPS_AMAX < 8 ? FPS_AMAX : 8) ? (l15 << (4 * rk)) : 0));       // wave of every rank
            const unsigned wh = FPS_AMAX > 8 ? (unsigned)__builtin_amdgcn_readfirstlane(row_or_i32((rk >= 8 && rk < FPS_AMAX) ? (l15 << (4 * (rk - 8))) : 0)) : 0u;
            int na = __builtin_ctz(~(unsigned)okb);               // run of acceptable ranks 0, 1, ...
            na = min(na, min(FPS_AMAX, m - j));
            int term = 0;
            const unsigned zero0 = (unsigned)(__ballot(rk == 0 && mine.x == 0) & 0xFFFFull);
            if (zero0 != 0u) {                                    // everything is covered: the sequential algorithm repeats this pick forever
                int voff = 0;
                asm volatile("" : "+v"(voff));
                term = 1 + __builtin_amdgcn_readfirstlane(pm[__builtin_amdgcn_readlane(mine2.x, __builtin_ctz(zero0)) + voff]);
                na = 0;
            }
            nacc = na;
            acc_list = na > 0 ? ((((unsigned long long)wh << 32) | wl) & (na >= 16 ? ~0ull : ((1ull << (4 * na)) - 1ull))) : 0ull;
            abuf = buf;
            termk = term;
            if (myrank < na) {                                     // own candidate accepted as pick number j+rank: write it, mark consumed
                if (lane == 0) o[j + myrank] = ck;
                need = true;
            }
            if ((myconf & (unsigned)(__ballot(rk < na) & 0xFFFFull)) != 0u) need = true;      // an accepted centre reaches the cached candidate
            j += na;
            CELL_TICK(3);
            if (termk != 0) break;
            continue;
        }
        // ---- slow path: wave 0 extracts serially with reference-rank tie-breaks, then broadcasts ----
        if (wave == 0) {
            int na = 0, term = 0, jn = j;
            unsigned long long alist = 0;
            int remaining = mine.x;               // value bits; -1.0f once consumed
            unsigned acc_mask = 0;                // bit w: wave w's candidate accepted this round
            int bound = NEG_ONE_BITS;             // max runner-up bound among accepted waves
#pragma unroll 1
            for (int i = 0; i < FPS_AMAX && jn < m; ++i) {
                const int M = __builtin_amdgcn_readfirstlane(row_max_i32(remaining));
                if (M < 0) break;                                   // no candidate left
                unsigned long long eq = __ballot(remaining == M) & 0xFFFFull;
                int wsel = __builtin_ctzll(eq);
                if (__builtin_popcountll(eq) > 1) {                 // equal values in several waves: lowest reference rank wins
                    unsigned rmin = 0xFFFFFFFFu;
#pragma unroll 1
                    for (int w = 0; w < 16; ++w)
                        if ((eq >> w) & 1ull) {
                            const int kw = pm[__builtin_amdgcn_readfirstlane(s_cand[buf + w * 2 + 1].x)];
                            const unsigned rw = ((unsigned)(kw & 511) << 22) | (unsigned)(kw >> 9);
                            if (rw < rmin) { rmin = rw; wsel = w; }
                        }
                }
                if (i == 0 && M == 0) {                             // everything is covered
                    term = 1 + pm[__builtin_amdgcn_readfirstlane(s_cand[buf + wsel * 2 + 1].x)];
                    break;
                }
                const int4 sa = s_cand[buf + wsel * 2];
                const int sbound = s_cand[buf + wsel * 2 + 1].y;
                if (i > 0) {
                    if (!(M > bound)) break;
                    const float dd = dist2_cuda(__int_as_float(sa.y) - __int_as_float(mine.y), __int_as_float(sa.z) - __int_as_float(mine.z),
                                                __int_as_float(sa.w) - __int_as_float(mine.w));
                    const bool hit = ((acc_mask >> l15) & 1u) && (dd < __int_as_float(M));
                    if (__ballot(hit) != 0ull) break;
                }
                alist |= (unsigned long long)wsel << (4 * na);
                ++na;
                acc_mask |= 1u << wsel;
                bound = max(bound, __builtin_amdgcn_readfirstlane(sbound));
                remaining = (l15 == wsel) ? NEG_ONE_BITS : remaining;
                ++jn;
            }
            if (lane == 0) { s_batch[0] = make_int4(na, (int)(unsigned)alist, term, jn); s_batch[1] = make_int4((int)(unsigned)(alist >> 32), 0, 0, 0); }
        }
        __syncthreads();
        {
            const int4 br = *s_batch;
            nacc = __builtin_amdgcn_readfirstlane(br.x);
            acc_list = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(s_batch[1].x) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(br.y);
            abuf = buf;
            termk = __builtin_amdgcn_readfirstlane(br.z);
            const int jnew = __builtin_amdgcn_readfirstlane(br.w);
            for (int u = 0; u < nacc; ++u) {
                const int wu = (int)((acc_list >> (4 * u)) & 15ull);
                if (wu == wave) {
                    if (lane == 0) o[j + u] = ck;
                    need = true;
                }
                if ((myconf >> wu) & 1u) need = true;
            }
            j = jnew;
            CELL_TICK(3);
            if (termk != 0) break;
        }
    }
    // degenerate tail (max min-distance == 0): the reference keeps returning the rank-minimal covered point
    if (termk != 0)
        for (int jj = j + t; jj < m; jj += FPS_T) o[jj] = termk - 1;
#ifdef FPS_PROFILE
    if (blockIdx.x == 0 && lane == 0) { g_cell_waves[wave * 4] = napplied; g_cell_waves[wave * 4 + 1] = nrefresh; }
#if FPS_PROFILE != 2
    if (blockIdx.x == 0 && (t == 0 || t == FPS_T - 64)) {
        long long* d = g_cell_prof + (t ? 16 : 0);
        for (int i = 0; i < 4; ++i) d[i] = cprof[i];
        d[4] = round; d[5] = napplied; d[6] = nrefresh;
        for (int i = 4; i < 10; ++i) d[4 + i] = cprof[i];
    }
#endif
#endif
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

// ============================================================================================
// Pre-pass of the cell-based FPS: spatial partition of every scene into 16 cells of csz = ceil(n/16)
// points, reference tie rank (k mod 512, k) ascending inside a cell.  Any partition gives the same
// FPS result (the kernel is exact for every assignment of points to cells); compact cells only make
// culling and batching effective.
//   K1 fps_bin_kernel   (one 1024-thread workgroup per scene): bounding box -> 12-bit Morton voxel id
//      (4 bits per axis) -> LDS histogram -> exclusive scan -> counting-sort scatter of the points'
//      rank keys into voxel order (order inside a voxel is whatever the LDS atomics produce).
//   K2 fps_cellsort_kernel (one workgroup per (cell, scene)): bitonic sort of the cell's <= 2048 rank
//      keys in LDS, then perm[] (rank -> original index) and the gathered coordinates are written out.
// ============================================================================================
__device__ __forceinline__ unsigned spread4(unsigned v) { return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6); }

__global__ __launch_bounds__(1024) void fps_bin_kernel(int n, const float* __restrict__ inp, unsigned* __restrict__ keys, int* __restrict__ vorder) {
    __shared__ int hist[4096];
    __shared__ float red[6][16];
    __shared__ int wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* xyz = inp + (size_t)blockIdx.x * n * 3;
    unsigned* out = keys + (size_t)blockIdx.x * n;
    // ---- bounding box ----
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    for (int k = t; k < n; k += 1024)
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = xyz[k * 3 + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int s2 = 32; s2 >= 1; s2 >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], s2, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], s2, 64)); }
        if (lane == 0) { red[a][wave] = lo[a]; red[3 + a][wave] = hi[a]; }
    }
    for (int i = t; i < 4096; i += 1024) hist[i] = 0;
    __syncthreads();
    float inv[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = red[a][0], h = red[3 + a][0];
        for (int w = 1; w < 16; ++w) { l = fminf(l, red[a][w]); h = fmaxf(h, red[3 + a][w]); }
        lo[a] = l;
        inv[a] = (h > l) ? 16.0f / (h - l) : 0.0f;
    }
    auto voxel = [&](int k) {
        unsigned q[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            int v = (int)((xyz[k * 3 + a] - lo[a]) * inv[a]);
            q[a] = (unsigned)(v < 0 ? 0 : (v > 15 ? 15 : v));
        }
        return spread4(q[0]) | (spread4(q[1]) << 1) | (spread4(q[2]) << 2);
    };
    // ---- histogram ----
    for (int k = t; k < n; k += 1024) atomicAdd(&hist[voxel(k)], 1);
    __syncthreads();
    // ---- exclusive scan of 4096 bins: 4 bins per thread, wave scan, 16 wave totals ----
    int c0 = hist[4 * t], c1 = hist[4 * t + 1], c2 = hist[4 * t + 2], c3 = hist[4 * t + 3];
    const int tot = c0 + c1 + c2 + c3;
    int incl = tot;
#pragma unroll
    for (int s2 = 1; s2 < 64; s2 <<= 1) { const int o = __shfl_up(incl, s2, 64); if (lane >= s2) incl += o; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    const int excl = base + incl - tot;
    hist[4 * t] = excl; hist[4 * t + 1] = excl + c0; hist[4 * t + 2] = excl + c0 + c1; hist[4 * t + 3] = excl + c0 + c1 + c2;
    __syncthreads();
    // ---- scatter the reference tie ranks into voxel order ----
    for (int k = t; k < n; k += 1024) {
        const int pos = atomicAdd(&hist[voxel(k)], 1);
        out[pos] = fps_tie_rank(k);
        if (vorder) vorder[(size_t)blockIdx.x * n + pos] = k;      // the scene in 16^3-voxel Morton order (three_nn's scan order)
    }
}

// ---- r06: K1 spread over G workgroups per scene (point slices) ----------------------------------------------------------------------------------
// fps_bin_kernel walks a scene three times with ONE workgroup (bounding box, histogram, scatter: 3 x 32 dependent rounds at 32768 points): 74 us
// stand-alone, 115 us beside the layers, on 8 CUs whose co-resident layer workgroups starve meanwhile (HISTORY 4.6).  The same counting sort by
// slices of 4096 points: bbox partials -> per-slice voxel histograms -> per-scene scan (slices in ascending order inside a voxel) -> per-slice scatter.
// Which points of a voxel end up where inside it is as arbitrary as before (LDS atomics), and as irrelevant: the sampling kernel is exact for every
// assignment of points to cells, and fps_cellsort_kernel orders every cell by reference tie rank.
struct FpsBox { float lo[3], hi[3]; };
__global__ __launch_bounds__(1024) void fps_bbox_kernel(int n, int G, const float* __restrict__ inp, FpsBox* __restrict__ part) {
    __shared__ float red[6][16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = blockIdx.x, scene = blockIdx.y;
    const float* xyz = inp + (size_t)scene * n * 3;
    const int k0 = (int)((long)n * g / G), k1 = (int)((long)n * (g + 1) / G);
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    for (int k = k0 + t; k < k1; k += 1024)
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = xyz[k * 3 + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int s2 = 32; s2 >= 1; s2 >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], s2, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], s2, 64)); }
        if (lane == 0) { red[a][wave] = lo[a]; red[3 + a][wave] = hi[a]; }
    }
    __syncthreads();
    if (t < 3) {
        float l = red[t][0], h = red[3 + t][0];
        for (int w = 1; w < 16; ++w) { l = fminf(l, red[t][w]); h = fmaxf(h, red[3 + t][w]); }
        part[(size_t)scene * G + g].lo[t] = l;
        part[(size_t)scene * G + g].hi[t] = h;
    }
}
// the scene's box from the G partial boxes (min / max are exact: the same box whatever the slicing), then this point's 12-bit Morton voxel
struct FpsVoxelizer {
    float lo[3], inv[3];
    __device__ void init(const FpsBox* part, int G) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float l = part[0].lo[a], h = part[0].hi[a];
            for (int g = 1; g < G; ++g) { l = fminf(l, part[g].lo[a]); h = fmaxf(h, part[g].hi[a]); }
            lo[a] = l;
            inv[a] = (h > l) ? 16.0f / (h - l) : 0.0f;
        }
    }
    __device__ unsigned operator()(const float* xyz, int k) const {
        unsigned q[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            int v = (int)((xyz[k * 3 + a] - lo[a]) * inv[a]);
            q[a] = (unsigned)(v < 0 ? 0 : (v > 15 ? 15 : v));
        }
        return spread4(q[0]) | (spread4(q[1]) << 1) | (spread4(q[2]) << 2);
    }
};
__global__ __launch_bounds__(1024) void fps_vhist_kernel(int n, int G, const float* __restrict__ inp, const FpsBox* __restrict__ part, int* __restrict__ hist) {
    __shared__ int h[4096];
    const int t = threadIdx.x, g = blockIdx.x, scene = blockIdx.y;
    const float* xyz = inp + (size_t)scene * n * 3;
    FpsVoxelizer vox;
    vox.init(part + (size_t)scene * G, G);
    for (int i = t; i < 4096; i += 1024) h[i] = 0;
    __syncthreads();
    const int k0 = (int)((long)n * g / G), k1 = (int)((long)n * (g + 1) / G);
    for (int k = k0 + t; k < k1; k += 1024) atomicAdd(&h[vox(xyz, k)], 1);
    __syncthreads();
    int* o = hist + ((size_t)scene * G + g) * 4096;
    for (int i = t; i < 4096; i += 1024) o[i] = h[i];
}
// per scene: hist[g][v] <- first slot of slice g inside voxel v (voxels ascending, slices ascending inside a voxel); G <= 32
__global__ __launch_bounds__(1024) void fps_vscan_kernel(int G, int* __restrict__ hist) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int* h = hist + (size_t)blockIdx.x * G * 4096;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < 4096; base += 1024) {
        const int v = base + t;
        int c[32];
#pragma unroll
        for (int g = 0; g < 32; ++g) c[g] = g < G ? h[(size_t)g * 4096 + v] : 0;
        int tot = 0;
#pragma unroll
        for (int g = 0; g < 32; ++g) tot += c[g];
        int incl = tot;
#pragma unroll
        for (int s2 = 1; s2 < 64; s2 <<= 1) { const int u = __shfl_up(incl, s2, 64); if (lane >= s2) incl += u; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        int run = before + incl - tot;
#pragma unroll
        for (int g = 0; g < 32; ++g) if (g < G) { h[(size_t)g * 4096 + v] = run; run += c[g]; }
        __syncthreads();
        if (t == 1023) carry = run;
        __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void fps_vfill_kernel(int n, int G, const float* __restrict__ inp, const FpsBox* __restrict__ part, const int* __restrict__ hist,
                                                         unsigned* __restrict__ keys, int* __restrict__ vorder) {
    __shared__ int h[4096];
    const int t = threadIdx.x, g = blockIdx.x, scene = blockIdx.y;
    const float* xyz = inp + (size_t)scene * n * 3;
    unsigned* out = keys + (size_t)scene * n;
    FpsVoxelizer vox;
    vox.init(part + (size_t)scene * G, G);
    const int* hs = hist + ((size_t)scene * G + g) * 4096;
    for (int i = t; i < 4096; i += 1024) h[i] = hs[i];
    __syncthreads();
    const int k0 = (int)((long)n * g / G), k1 = (int)((long)n * (g + 1) / G);
    for (int k = k0 + t; k < k1; k += 1024) {
        const int pos = atomicAdd(&h[vox(xyz, k)], 1);
        out[pos] = fps_tie_rank(k);
        if (vorder) vorder[(size_t)scene * n + pos] = k;
    }
}
// slices per scene of the spread-out pre-pass (1 = fps_bin_kernel); scratch: G x (4096 ints + 6 floats) per scene, taken from the front of the scene's
// slice of sxyz (n x 3 floats, written only by the cell sort that follows)
static inline int fps_bin_slices(int n) {
    static const int on = getenv("GSPN_FPS_BIN_SLICES") ? atoi(getenv("GSPN_FPS_BIN_SLICES")) : -1;      // tuning hook: 0 / 1 = off, k > 1 = force k
    if (on == 0 || on == 1) return 1;
    long G = on > 1 ? on : n / 4096;
    if (G > 32) G = 32;
    while (G > 1 && G * (4096L + 8) * 4 > (long)n * 12) --G;
    return G < 4 ? 1 : (int)G;
}

template <int SZ>      // SZ = power of two >= csz, <= 2048; block = SZ/2 threads
__global__ void fps_cellsort_kernel(int n, int csz, const float* __restrict__ inp, int* perm, float* __restrict__ sxyz) {
    __shared__ unsigned sk[SZ];
    const int cell = blockIdx.x, scene = blockIdx.y;
    const int t = threadIdx.x;
    const int begin = cell * csz;
    const int cnt = min(csz, n - begin) > 0 ? min(csz, n - begin) : 0;
    // the rank keys of this cell sit in the cell's own slice of perm (fps_bin_kernel): read all of them, then overwrite the slice
    const unsigned* src = reinterpret_cast<const unsigned*>(perm) + (size_t)scene * n + begin;
    for (int i = t; i < SZ; i += SZ / 2) sk[i] = i < cnt ? src[i] : 0xFFFFFFFFu;
    __syncthreads();
    for (int k = 2; k <= SZ; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));       // lower index of the t-th pair at distance j
            const int p = i | j;
            const bool up = (i & k) == 0;
            const unsigned a = sk[i], b2 = sk[p];
            if ((a > b2) == up) { sk[i] = b2; sk[p] = a; }
            __syncthreads();
        }
    }
    const float* xyz = inp + (size_t)scene * n * 3;
    for (int i = t; i < cnt; i += SZ / 2) {
        const int k = fps_tie_rank_inv(sk[i]);
        const size_t pos = (size_t)scene * n + begin + i;
        perm[pos] = k;
        sxyz[pos * 3 + 0] = xyz[k * 3 + 0];
        sxyz[pos * 3 + 1] = xyz[k * 3 + 1];
        sxyz[pos * 3 + 2] = xyz[k * 3 + 2];
    }
}

int gspn_fps_prepass_cells(int b, int n, int ncell, int csz, const float* inp, int* perm, float* sxyz, hipStream_t st, int* vorder) {
    if (csz > 2048 || (long long)ncell * csz < n || b > 65535) return GSPN_ERR_UNSUPPORTED;
    const int G = fps_bin_slices(n);
    if (G > 1) {
        // scratch inside sxyz: every scene's own slice holds its G histograms, then its G partial boxes
        // (stride n*3 floats per scene; the pointers below are per-scene bases folded into the kernels' scene * G indexing through a common layout)
        int* hist = reinterpret_cast<int*>(sxyz);                                        // [b][G][4096] ints, packed from the front of sxyz
        FpsBox* part = reinterpret_cast<FpsBox*>(hist + (size_t)b * G * 4096);           // [b][G] boxes behind them: b*G*(4096+6)*4 bytes <= b*n*12 (fps_bin_slices)
        hipLaunchKernelGGL(fps_bbox_kernel, dim3(G, b), dim3(1024), 0, st, n, G, inp, part);
        hipLaunchKernelGGL(fps_vhist_kernel, dim3(G, b), dim3(1024), 0, st, n, G, inp, part, hist);
        hipLaunchKernelGGL(fps_vscan_kernel, dim3(b), dim3(1024), 0, st, G, hist);
        hipLaunchKernelGGL(fps_vfill_kernel, dim3(G, b), dim3(1024), 0, st, n, G, inp, part, hist, reinterpret_cast<unsigned*>(perm), vorder);
    } else {
        hipLaunchKernelGGL(fps_bin_kernel, dim3(b), dim3(1024), 0, st, n, inp, reinterpret_cast<unsigned*>(perm), vorder);
    }
    const dim3 g2(ncell, b);
    if (csz <= 128) hipLaunchKernelGGL(fps_cellsort_kernel<128>, g2, dim3(64), 0, st, n, csz, inp, perm, sxyz);
    else if (csz <= 256) hipLaunchKernelGGL(fps_cellsort_kernel<256>, g2, dim3(128), 0, st, n, csz, inp, perm, sxyz);
    else if (csz <= 512) hipLaunchKernelGGL(fps_cellsort_kernel<512>, g2, dim3(256), 0, st, n, csz, inp, perm, sxyz);
    else if (csz <= 1024) hipLaunchKernelGGL(fps_cellsort_kernel<1024>, g2, dim3(512), 0, st, n, csz, inp, perm, sxyz);
    else hipLaunchKernelGGL(fps_cellsort_kernel<2048>, g2, dim3(1024), 0, st, n, csz, inp, perm, sxyz);
    return gspn_launch_status();
}

template <int P, bool ZLDS>
static int launch_fps_cell(int b, int n, int m, int csz, const float* sxyz, const int* perm, const float* inp0, int stride0, int* out, hipStream_t st) {
    size_t lds = 4096 + (ZLDS ? (size_t)P * FPS_T * sizeof(float) : 0);
    if (ZLDS) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_cell_kernel<P, ZLDS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    lds = gspn_claim_lds(1, reinterpret_cast<const void*>(&fps_cell_kernel<P, ZLDS>), lds);
    hipLaunchKernelGGL((fps_cell_kernel<P, ZLDS>), dim3(b), dim3(FPS_T), lds, st, n, m, csz, sxyz, perm, inp0, stride0, out);
    return gspn_launch_status();
}
// FPS on a spatially pre-sorted scene (see fps_cell_kernel): sxyz (b,n,3) = inp gathered by perm (b,n) [sorted position -> original
// index]; the sort is 16 equal cells of csz = ceil(n/16) points in Morton order, reference tie rank (k mod 512, k) inside a cell;
// inp0 (b,3) = coordinates of original point 0 of every scene.  Same output as gspn_farthestpointsampling.
static int gspn_fps_cells_strided(int b, int n, int m, int csz, const float* sxyz, const int* perm, const float* inp0, int stride0, int* out, void* stream) {
    if (b < 0 || n <= 0 || m <= 0 || csz <= 0 || (long long)csz * 16 < n) return GSPN_ERR_ARG;
    if (b == 0) return 0;
    if (!sxyz || !perm || !inp0 || !out) return GSPN_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (csz <= 64 * 2) return launch_fps_cell<2, false>(b, n, m, csz, sxyz, perm, inp0, stride0, out, st);
    if (csz <= 64 * 4) return launch_fps_cell<4, false>(b, n, m, csz, sxyz, perm, inp0, stride0, out, st);
    if (csz <= 64 * 8) return launch_fps_cell<8, false>(b, n, m, csz, sxyz, perm, inp0, stride0, out, st);
    if (csz <= 64 * 16) return launch_fps_cell<16, false>(b, n, m, csz, sxyz, perm, inp0, stride0, out, st);
    if (csz <= 64 * 32) {
        // (r05 measured the alternative -- every cell striped over all 16 waves -- 1.6x slower: tools/patches/r06_pruned_alternates.patch, profiles/r05_experiments.txt item 3)
        return launch_fps_cell<32, true>(b, n, m, csz, sxyz, perm, inp0, stride0, out, st);
    }
    return GSPN_ERR_UNSUPPORTED;
}
extern "C" int gspn_fps_cells(int b, int n, int m, int csz, const float* sxyz, const int* perm, const float* inp0, int* out, void* stream) {
    return gspn_fps_cells_strided(b, n, m, csz, sxyz, perm, inp0, 3, out, stream);
}

// workspace of gspn_farthestpointsampling_cells: [perm: b*n i32][sxyz: b*n*3 f32] = 16 bytes per point.  The reference's own FPS
// scratch, temp (32,n) f32 (tf_sampling.cpp:111-115), is 128*n bytes: large enough for b <= 8 scenes.
extern "C" long gspn_fps_cells_ws_bytes(int b, int n) {
    if (b < 0 || n <= 0) return GSPN_ERR_ARG;
    return (long)b * n * 16;
}
// The two halves of gspn_farthestpointsampling_cells, separately callable (so that a caller can time the sampling kernel alone):
// the spatial pre-pass that fills `ws` ...
extern "C" int gspn_fps_cells_prepass(int b, int n, const float* inp, void* ws, void* stream) {
    if (b < 0 || n <= 0) return GSPN_ERR_ARG;
    if (b == 0) return 0;
    if (!inp || !ws) return GSPN_ERR_ARG;
    if (n > GSPN_FPS_RESIDENT_MAX || b > 65535) return GSPN_ERR_UNSUPPORTED;
    int* perm = reinterpret_cast<int*>(ws);
    float* sxyz = reinterpret_cast<float*>(perm + (size_t)b * n);
    return gspn_fps_prepass_cells(b, n, 16, (n + 15) / 16, inp, perm, sxyz, (hipStream_t)stream);
}
// the same pre-pass, also leaving the scene in its 16^3-voxel Morton order: vorder (b, n) int32, original point indices (the order inside
// a voxel is arbitrary).  A finer spatial order than the 16 cells of `ws` -- what gspn_threenn_ordered wants for its `order`.
extern "C" int gspn_fps_cells_prepass_order(int b, int n, const float* inp, void* ws, int* vorder, void* stream) {
    if (b < 0 || n <= 0) return GSPN_ERR_ARG;
    if (b == 0) return 0;
    if (!inp || !ws || !vorder) return GSPN_ERR_ARG;
    if (n > GSPN_FPS_RESIDENT_MAX || b > 65535) return GSPN_ERR_UNSUPPORTED;
    int* perm = reinterpret_cast<int*>(ws);
    float* sxyz = reinterpret_cast<float*>(perm + (size_t)b * n);
    return gspn_fps_prepass_cells(b, n, 16, (n + 15) / 16, inp, perm, sxyz, (hipStream_t)stream, vorder);
}
// ... and the sampling kernel on a workspace the pre-pass has filled for the same (b, n, inp)
extern "C" int gspn_fps_cells_sample(int b, int n, int m, const float* inp, const void* ws, int* out, void* stream) {
    if (b < 0 || n <= 0 || m <= 0) return GSPN_ERR_ARG;
    if (b == 0) return 0;
    if (!inp || !ws || !out) return GSPN_ERR_ARG;
    if (n > GSPN_FPS_RESIDENT_MAX || b > 65535) return GSPN_ERR_UNSUPPORTED;
    const int* perm = reinterpret_cast<const int*>(ws);
    const float* sxyz = reinterpret_cast<const float*>(perm + (size_t)b * n);
    // original point 0 of scene i is inp[i*n*3 ..]: a strided view, the kernel only needs a pointer + stride -> pass inp with stride n*3
    return gspn_fps_cells_strided(b, n, m, (n + 15) / 16, sxyz, perm, inp, n * 3, out, stream);
}
// Drop-in for gspn_farthestpointsampling (identical output) for 64 <= n <= 32768: pre-pass + fps_cell_kernel, all on `stream`.
extern "C" int gspn_farthestpointsampling_cells(int b, int n, int m, const float* inp, void* ws, int* out, void* stream) {
    if (b < 0 || n <= 0 || m <= 0) return GSPN_ERR_ARG;
    if (b == 0) return 0;
    if (!inp || !ws || !out) return GSPN_ERR_ARG;
    const int rc = gspn_fps_cells_prepass(b, n, inp, ws, stream);
    if (rc) return rc;
    return gspn_fps_cells_sample(b, n, m, inp, ws, out, stream);
}

// ============================================================================================
// Small scenes (n <= 2048: the second and third SA level): the same on-chip scheme with FOUR waves instead of sixteen.  A round of
// fps_resident_kernel at these sizes is all synchronisation (8 VALU instructions of distance update per wave against a 16-wave
// barrier and a 16-candidate exchange); 256 threads with 2C points each (C = ceil(n/512) <= 4) halve the round time and leave the rest
// of the CU to other workgroups.  Thread t, slot p hold the point of reference tie rank q = t*2C + p, i.e. k = (q % C)*512 + q / C, so
// "lowest (wave, lane, slot)" is again (k mod 512 asc, k asc).  Each lane carries its best slot's coordinates and index along with
// the maximum, so the winner needs no second look-up.
// ============================================================================================
template <int C>
__global__ __launch_bounds__(256) void fps_small_kernel(int n, int m, const float* __restrict__ inp, int* __restrict__ out) {
    constexpr int P = 2 * C;
    __shared__ int4 s_cand[2][4];            // {max bits, x, y, z} per wave, double buffered
    __shared__ int s_k[2][4];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const float* xyz = inp + (size_t)blockIdx.x * n * 3;
    int* o = out + (size_t)blockIdx.x * m;
    float x[P], y[P], z[P], td[P];
    int kk[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int q = t * P + p;
        const int k = (q % C) * 512 + q / C;
        const int kc = k < n ? k : n - 1;
        kk[p] = k;
        x[p] = xyz[kc * 3 + 0]; y[p] = xyz[kc * 3 + 1]; z[p] = xyz[kc * 3 + 2];
        td[p] = k < n ? 1e38f : -1.0f;        // tf_sampling_g.cu:117-119; padding never wins (real candidates are >= 0)
    }
    if (t == 0) o[0] = 0;                      // :114-116
    float cx = xyz[0], cy = xyz[1], cz = xyz[2];
    for (int j = 1; j < m; ++j) {
        int best = NEG_ONE_BITS, bk = 0;
        float bx = 0.f, by = 0.f, bz = 0.f;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const float d = dist2_cuda(x[p] - cx, y[p] - cy, z[p] - cz);          // :139-142
            td[p] = vmin_f32(d, td[p]);                                          // :143
            const int v = __float_as_int(td[p]);
            const bool up = v > best;                                             // strict: the lowest slot keeps a tie (:146-149)
            best = up ? v : best;
            bx = up ? x[p] : bx; by = up ? y[p] : by; bz = up ? z[p] : bz;
            bk = up ? kk[p] : bk;
        }
        const int wmax = wave_max_i32(best);
        const int lw = __builtin_ctzll(__ballot(best == wmax));                  // lowest lane on ties
        const int buf = j & 1;
        if (lane == lw) {
            s_cand[buf][wave] = make_int4(wmax, __float_as_int(bx), __float_as_int(by), __float_as_int(bz));
            s_k[buf][wave] = bk;
        }
        __syncthreads();
        const int4 c0 = s_cand[buf][0], c1 = s_cand[buf][1], c2 = s_cand[buf][2], c3 = s_cand[buf][3];
        int w = 0, mv = c0.x;
        if (c1.x > mv) { mv = c1.x; w = 1; }                                      // lowest wave on ties
        if (c2.x > mv) { mv = c2.x; w = 2; }
        if (c3.x > mv) { mv = c3.x; w = 3; }
        const int4 cw = w == 0 ? c0 : (w == 1 ? c1 : (w == 2 ? c2 : c3));
        cx = __int_as_float(cw.y); cy = __int_as_float(cw.z); cz = __int_as_float(cw.w);
        if (t == 0) o[j] = s_k[buf][w];                                          // :166-168
    }
}
template <int C>
static int launch_fps_small(int b, int n, int m, const float* inp, int* out, hipStream_t st) {
    const size_t lds = gspn_claim_lds(2, reinterpret_cast<const void*>(&fps_small_kernel<C>), 0);
    hipLaunchKernelGGL((fps_small_kernel<C>), dim3(b), dim3(256), lds, st, n, m, inp, out);
    return gspn_launch_status();
}

// The drop-in symbol.  `temp` is the reference's own FPS scratch, (32,n) f32 = 128*n bytes (tf_sampling.cpp:111-115): it is used as
// the workspace of the cell kernels (8192 <= n <= 32768: 8 scenes per pass; n > 32768: as many scenes per pass as fit), so a caller
// that binds this symbol the way the reference binds farthestpointsamplingLauncher gets the fast paths.  temp == NULL is accepted
// for n <= 32768 (plain on-chip kernel, no scratch).
extern "C" int gspn_farthestpointsampling(int b, int n, int m, const float* inp, float* temp, int* out, void* stream) {
    if (b < 0 || n <= 0 || m <= 0) return GSPN_ERR_ARG;          // tf_sampling.cpp:99,105
    if (b == 0) return 0;
    if (!inp || !out) return GSPN_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (n <= 512) return launch_fps_small<1>(b, n, m, inp, out, st);
    if (n <= 1024) return launch_fps_small<2>(b, n, m, inp, out, st);
    if (n <= 1536) return launch_fps_small<3>(b, n, m, inp, out, st);
    if (n <= 2048) return launch_fps_small<4>(b, n, m, inp, out, st);
    if (n >= 8192 && n <= GSPN_FPS_RESIDENT_MAX && temp) {
        for (int s0 = 0; s0 < b; s0 += 8) {                     // 16 bytes per point: 8 scenes fit the 128*n bytes of temp
            const int bs = b - s0 < 8 ? b - s0 : 8;
            const int rc = gspn_farthestpointsampling_cells(bs, n, m, inp + (size_t)s0 * n * 3, temp, out + (size_t)s0 * m, stream);
            if (rc) return rc;
        }
        return 0;
    }
    if (n <= 4096) return launch_fps_resident<4, false>(b, n, m, inp, out, st);
    if (n <= 8192) return launch_fps_resident<8, false>(b, n, m, inp, out, st);
    if (n <= 16384) return launch_fps_resident<16, false>(b, n, m, inp, out, st);
    if (n <= GSPN_FPS_RESIDENT_MAX) return launch_fps_resident<32, true>(b, n, m, inp, out, st);
    if (!temp) return GSPN_ERR_ARG;
    int per = 8;
    while (per > 1 && gspn_fps_multi_ws_bytes(per, n) > 128l * n) --per;
    if (gspn_fps_multi_ws_bytes(per, n) > 128l * n) return GSPN_ERR_UNSUPPORTED;
    for (int s0 = 0; s0 < b; s0 += per) {
        const int bs = b - s0 < per ? b - s0 : per;
        const int rc = gspn_farthestpointsampling_multi(bs, n, m, 0, inp + (size_t)s0 * n * 3, temp, out + (size_t)s0 * m, stream);
        if (rc) return rc;
    }
    return 0;
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
// prob_sample (tf_sampling_g.cu:7-104): per-row inclusive prefix sum + inverse-CDF lookup.  Not reached from the model.
//
// What has to match the reference is the ROUNDING of the prefix sums, i.e. the association order of its scan:
//   * a row is cut into tiles of 8192 values, a tile into groups of 4: (v1, v1+v2, (v1+v2)+v3, (v3+v4)+(v1+v2)); a ragged last group
//     is summed left to right (:19-42);
//   * over the group totals the reference runs a Brent-Kung scan (:43-65), which amounts to: the sum of an aligned block of 2^u groups
//     is the sum of its two halves, and the prefix over c groups adds those block sums in the order of the set bits of c, from the
//     most significant down;
//   * element = group-local prefix + prefix of the groups before it, + carry of the earlier tiles; the carry is compensated
//     (Kahan-style pair) across tiles (:72-78).
// This kernel builds exactly those quantities in a wave64-native way: a wave owns 64 consecutive groups, forms the block sums of
// 2..64 groups with xor-shuffles (a + b == b + a bit for bit, so both partners hold the same value), one wave finishes the five levels
// above, every level is kept in LDS (4095 floats), and each lane then folds the <= 12 block sums its group count selects.  Group-local
// prefixes stay in registers; there is no element buffer in LDS.
// ============================================================================================
#define PS_THREADS 512
#define PS_GROUPS 2048                   // groups of 4 per tile
#define PS_LEVELS 12                     // block sizes 1 .. 2048 groups
__device__ __forceinline__ int ps_level_off(int u) { return 2 * PS_GROUPS - ((2 * PS_GROUPS) >> u); }      // 0, 2048, 3072, ...

// prefix over the first c groups (1 <= c <= PS_GROUPS) from the per-level block sums
__device__ __forceinline__ float ps_fold(const float* lv, int c) {
    float acc = 0.f;
    bool first = true;
#pragma unroll
    for (int u = PS_LEVELS - 1; u >= 0; --u) {
        if ((c >> u) & 1) {
            const float blk = lv[ps_level_off(u) + (c >> u) - 1];
            acc = first ? blk : acc + blk;
            first = false;
        }
    }
    return acc;
}

__global__ __launch_bounds__(PS_THREADS) void prefix_rows_kernel(int b, int n, const float* __restrict__ inp, float* __restrict__ out) {
    __shared__ float lv[2 * PS_GROUPS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    constexpr int CH = PS_GROUPS / 64 / (PS_THREADS / 64);        // 64-group chunks per wave and tile (4)
    for (int row = blockIdx.x; row < b; row += gridDim.x) {
        const float* src = inp + (size_t)row * n;
        float* dst = out + (size_t)row * n;
        float carry = 0.f, comp = 0.f;
        for (int base = 0; base < n; base += 4 * PS_GROUPS) {
            const int cnt = min(n - base, 4 * PS_GROUPS);          // values in this tile
            const int ng = (cnt + 3) >> 2;                         // groups in this tile
            float g[CH][4];
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const int grp = (wave + q * (PS_THREADS / 64)) * 64 + lane;
                const int e0 = grp * 4;
                float tot = 0.f;
                if (e0 + 3 < cnt) {
                    const float a0 = src[base + e0], a1 = src[base + e0 + 1], a2 = src[base + e0 + 2], a3 = src[base + e0 + 3];
                    const float s01 = a1 + a0;
                    g[q][0] = a0;
                    g[q][1] = s01;
                    g[q][2] = a2 + s01;
                    g[q][3] = (a3 + a2) + s01;
                    tot = g[q][3];
                } else {
                    float v = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (e0 + e < cnt) v += src[base + e0 + e];
                        g[q][e] = v;
                    }
                    tot = v;
                }
                // block sums of 1, 2, 4, ... 64 groups inside the wave
                lv[grp] = tot;
#pragma unroll
                for (int u = 1; u <= 6; ++u) {
                    tot = tot + __shfl_xor(tot, 1 << (u - 1), 64);
                    if ((lane & ((1 << u) - 1)) == 0) lv[ps_level_off(u) + (grp >> u)] = tot;
                }
            }
            __syncthreads();
            if (wave == 0) {                                       // levels 7..11 over the 32 chunk sums
                float tot = lane < PS_GROUPS / 64 ? lv[ps_level_off(6) + lane] : 0.f;
#pragma unroll
                for (int u = 7; u < PS_LEVELS; ++u) {
                    tot = tot + __shfl_xor(tot, 1 << (u - 7), 64);
                    if (lane < PS_GROUPS / 64 && (lane & ((1 << (u - 6)) - 1)) == 0) lv[ps_level_off(u) + (lane >> (u - 6))] = tot;
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const int grp = (wave + q * (PS_THREADS / 64)) * 64 + lane;
                if (grp < ng) {
                    const float before = grp > 0 ? ps_fold(lv, grp) : 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (grp * 4 + e < cnt) {
                            const float v = grp > 0 ? g[q][e] + before : g[q][e];
                            dst[base + grp * 4 + e] = v + carry;
                        }
                    }
                }
            }
            // compensated carry into the next tile (tf_sampling_g.cu:72-78)
            const float tt = ps_fold(lv, ng) + comp;
            const float nc = carry + tt;
            comp = tt - (nc - carry);
            carry = nc;
            __syncthreads();
        }
    }
}

// Inverse-CDF lookup (tf_sampling_g.cu:82-99): descending power-of-two steps from the last index.  The probe sequence is part of the
// result (the rounded prefix sums need not be monotone), so it is kept: step = 2^ceil(log2 n), ..., 1; move down while cdf >= q.
__global__ void inverse_cdf_kernel(int n, int m, int top, const float* __restrict__ cdf, const float* __restrict__ u, int* __restrict__ pick) {
    const int row = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const float* c = cdf + (size_t)row * n;
    const float q = u[(size_t)row * m + j] * c[n - 1];
    int pos = n - 1;
    for (int step = top; step > 0; step >>= 1) {
        const int cand = pos - step;
        if (cand >= 0 && c[cand] >= q) pos = cand;
    }
    pick[(size_t)row * m + j] = pos;
}
extern "C" int gspn_probsample(int b, int n, int m, const float* inp_p, const float* inp_r, float* temp, int* out, void* stream) {
    if (b < 0 || n <= 0 || m < 0) return GSPN_ERR_ARG;
    if (b == 0 || m == 0) return 0;
    if (b > 65535) return GSPN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(prefix_rows_kernel, dim3(b < 2048 ? b : 2048), dim3(PS_THREADS), 0, (hipStream_t)stream, b, n, inp_p, temp);
    int top = 1;
    while (top < n) top <<= 1;
    hipLaunchKernelGGL(inverse_cdf_kernel, dim3((m + 255) / 256, b), dim3(256), 0, (hipStream_t)stream, n, m, top, temp, inp_r, out);
    return gspn_launch_status();
}

extern "C" int gspn_dist_policy(void) { return GSPN_DIST_POLICY; }
extern "C" int gspn_abi_version(void) { return GSPN_ABI_VERSION; }
extern "C" int gspn_fill_zero(void* ptr, long bytes, void* stream) {
    if (bytes < 0) return GSPN_ERR_ARG;
    if (bytes == 0) return 0;
    hipError_t e = hipMemsetAsync(ptr, 0, (size_t)bytes, (hipStream_t)stream);
    return e == hipSuccess ? 0 : (int)e;
}
