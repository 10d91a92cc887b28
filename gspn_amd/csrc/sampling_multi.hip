// sampling_multi.hip -- farthest point sampling for scenes that do not fit one CU (n > 32768): G workgroups (CUs) per scene.
// Reference semantics: tf_ops/sampling/tf_sampling_g.cu:105-170 (its global-memory path :137-141 handles any n); callers at scan
// scale: data_prep.py:64-83 (n ~ 1e5, m = 30000), dataset.py:38-45.  Output is index-exact, ties included.
//
// Same on-chip scheme as fps_cell_kernel (sampling.hip): the scene is cut into 16*G compact spatial cells of <= 2048 points, one wave
// per cell, x / y / running min-distance in VGPRs, z in LDS; a wave skips a new centre that provably cannot change any of its points
// (bounding-box test) and several picks are accepted per synchronisation round.  What is new is the exchange between the G CUs:
//
//   round r (epoch r+1):
//     1. every wave applies the centres accepted in round r-1 (culled) and refreshes its cached candidate if it was hit;
//     2. local step, one LDS barrier: the 16 candidates of the workgroup are ranked by value; the best A = min(6, 64/G) are published
//        to the other workgroups as 8-byte {epoch, payload} granules (relaxed agent-scope stores = write-through `sc1`; the data is
//        its own flag, so there is no fence, no counter and no dependence on workgroup placement: MI355X guide, G16 recipe R2);
//     3. wave 0 sweeps the G*A <= 64 published records (relaxed agent-scope loads until every tag shows this epoch) into LDS;
//     4. the acceptance test is spread over all 16 waves: record i is acceptable, GIVEN that every better record is accepted, iff
//        no better record lies within its min-distance (evaluated exactly as the update would) and it beats the runner-up bound of
//        every better record's cell.  The batch is the run of acceptable ranks 0,1,...  (at most A picks, provably the next picks of
//        the sequential algorithm -- same argument as fps_cell_kernel; a record outside a workgroup's best A has global rank >= A);
//     5. equal values are never ordered by guesswork: a record tied with any other (published or not) is not acceptable, and when
//        rank 0 itself is tied (or the maximum is 0: everything is covered) a second, rarely taken exchange finds the point of
//        lowest reference tie rank (k mod 512, k) among ALL cells holding the maximum -- exactly the reference's block arg-max.
//   Buffers are double-buffered by epoch parity (a workgroup cannot run two rounds ahead of another: it needs its records first).
//
// Co-residency: the G workgroups of a scene wait for each other, so they must be resident together.  A workgroup takes a whole CU
// (1024 threads, > 80 KiB LDS); block -> (scene, g) keeps a scene's workgroups in consecutive dispatch order on ONE XCD (block % 8),
// which is a latency bonus only.  The launcher caps a launch at FPS_MULTI_MAX_WG workgroups and loops over scene groups.  Every
// wait is bounded (FPS_MULTI_TIMEOUT_TICKS of the 100 MHz wall clock): on expiry the kernel stores 1 to *status and returns.
#include "fps_common.h"

typedef __attribute__((address_space(1))) unsigned long long gu64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

#define FPSM_AMAX 6               // most picks per round
#define FPSM_RMAX 64              // most published records per scene and round (G * A)
#define FPSM_GMAX 32              // most workgroups per scene
#define FPSM_REC_GRANULES 6       // {value, x, y, z, position | tied << 31, runner-up bound}
#define FPSM_TIE_GRANULES 4       // {tie rank, x, y, z}
#define FPSM_HALF (FPSM_RMAX * FPSM_REC_GRANULES + FPSM_GMAX * FPSM_TIE_GRANULES)      // granules per epoch parity
#define FPSM_XCH_GRANULES (2 * FPSM_HALF)                                            // per scene
#define FPS_MULTI_MAX_WG 128
#ifndef FPS_MULTI_TIMEOUT_TICKS
#define FPS_MULTI_TIMEOUT_TICKS 200000000ll      // 2 s
#endif
#define TIE_NONE 0x7FFFFFFF

// LDS map (bytes)
#define L_LOC 0            // 2 x 16 x {int4,int4}   local candidates                    1024
#define L_GLOB 1024        // 2 x 65 x {int4,int4}   swept records (+ slot 64: tie-phase winner)   4160
#define L_INFO 5184        // 2 x 64 x int           {rank | ok << 9} per record           512
#define L_MISC 5696        // tie minimum, fail flag, winner k                              64
#define L_TROW 6144        // 16 x P floats          refresh rows                         <= 2048
#define L_Z 8192           // z plane

__device__ __forceinline__ void granule_store(unsigned long long* p, unsigned epoch, unsigned v) {
    __hip_atomic_store((gu64*)p, ((unsigned long long)epoch << 32) | v, RLX_AGENT);
}
__device__ __forceinline__ unsigned long long granule_load(const unsigned long long* p) {
    return __hip_atomic_load((gu64*)p, RLX_AGENT);
}

template <int P, bool ZLDS>
__global__ __launch_bounds__(FPS_T) void fps_multi_kernel(int b, int n, int m, int csz, int G, int A, const float* __restrict__ sxyz,
                                                          const int* __restrict__ perm, const float* __restrict__ inp0, int inp0_stride,
                                                          int* __restrict__ out, unsigned long long* xch, int* status) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int4* s_loc = reinterpret_cast<int4*>(smem + L_LOC);
    int4* s_glob = reinterpret_cast<int4*>(smem + L_GLOB);
    int* s_info = reinterpret_cast<int*>(smem + L_INFO);
    int* s_misc = reinterpret_cast<int*>(smem + L_MISC);      // [0] tie minimum, [1] fail, [2] winner's original index
    float* s_trow = reinterpret_cast<float*>(smem + L_TROW);
    v4f* s_z = reinterpret_cast<v4f*>(smem + L_Z);
    constexpr int GP = FpsGroup<P>::G;
    constexpr int NG = FpsGroup<P>::NG;

    // block -> (scene, g): the G workgroups of a scene are G consecutive blocks of one residue class mod 8 (one XCD)
    const int xcd = blockIdx.x & 7, q8 = blockIdx.x >> 3;
    const int g = q8 % G, scene = (q8 / G) * 8 + xcd;
    if (scene >= b) return;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const float* xyz = sxyz + (size_t)scene * n * 3;
    const int* pm = perm + (size_t)scene * n;
    int* o = out + (size_t)scene * m;
    unsigned long long* xs = xch + (size_t)scene * FPSM_XCH_GRANULES;
    const int cbase = (g * FPS_W + wave) * csz;            // first sorted position of this wave's cell
    const int R = G * A;                                   // published records per round

    v2f x[P / 2], y[P / 2], z[ZLDS ? 1 : P / 2], td[P / 2];
    float bx0 = 3e38f, bx1 = -3e38f, by0 = 3e38f, by1 = -3e38f, bz0 = 3e38f, bz1 = -3e38f;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int q = lane * P + p;
        const int pos = cbase + q;
        const bool live = q < csz && pos < n;
        const int pc = live ? pos : 0;
        float px = xyz[pc * 3 + 0], py = xyz[pc * 3 + 1], pz = xyz[pc * 3 + 2];
        bx0 = fminf(bx0, live ? px : 3e38f); bx1 = fmaxf(bx1, live ? px : -3e38f);
        by0 = fminf(by0, live ? py : 3e38f); by1 = fmaxf(by1, live ? py : -3e38f);
        bz0 = fminf(bz0, live ? pz : 3e38f); bz1 = fmaxf(bz1, live ? pz : -3e38f);
        const float d0 = live ? 1e38f : -1.0f;             // tf_sampling_g.cu:117-119; padding never wins
        px = live ? px : 0.f; py = live ? py : 0.f; pz = live ? pz : 0.f;
        asm("" : "+v"(px), "+v"(py), "+v"(pz));
        x[p >> 1][p & 1] = px;
        y[p >> 1][p & 1] = py;
        td[p >> 1][p & 1] = d0;
        if (ZLDS) reinterpret_cast<float*>(s_z)[((p >> 2) * FPS_T + t) * 4 + (p & 3)] = pz;
        else z[p >> 1][p & 1] = pz;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        bx0 = fminf(bx0, __shfl_xor(bx0, s, 64)); bx1 = fmaxf(bx1, __shfl_xor(bx1, s, 64));
        by0 = fminf(by0, __shfl_xor(by0, s, 64)); by1 = fmaxf(by1, __shfl_xor(by1, s, 64));
        bz0 = fminf(bz0, __shfl_xor(bz0, s, 64)); bz1 = fmaxf(bz1, __shfl_xor(bz1, s, 64));
    }
    bx0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bx0))); bx1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bx1)));
    by0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(by0))); by1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(by1)));
    bz0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bz0))); bz1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bz1)));

    // round 0 applies original point 0 (tf_sampling_g.cu:114-116), parked in record 0 of buffer 1
    const float* p0 = inp0 + (size_t)scene * inp0_stride;
    if (t == 0) {
        if (g == 0) o[0] = 0;
        s_glob[130 + 0] = make_int4(0, __float_as_int(p0[0]), __float_as_int(p0[1]), __float_as_int(p0[2]));
        s_misc[0] = TIE_NONE;
        s_misc[1] = 0;
    }
    __syncthreads();

    unsigned long long acc_list = 0;           // accepted record ids of the previous round, 7 bits each
    int abuf = 130;                            // int4 offset of the record table they refer to
    int nacc = 1;
    int j = 1;                                 // outputs written so far (same value in every workgroup of the scene)
    int wmax = __float_as_int(1e38f);
    bool need = true;
    int cv = 0, ck = 0, cpos = 0, cbound = NEG_ONE_BITS;
    float cfx = 0.f, cfy = 0.f, cfz = 0.f;
    int round = 0;
    int termk = 0;

    while (j < m) {
        // ---- 1a. apply the accepted centres, culled per wave (see fps_cell_kernel) ----
        unsigned todo;
        {
            const int u = lane < nacc ? lane : 0;
            const int4 cc = s_glob[abuf + (int)((acc_list >> (7 * u)) & 127) * 2];
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
            const int4 cc = s_glob[abuf + (int)((acc_list >> (7 * ci)) & 127) * 2];
            const float cx = __int_as_float(__builtin_amdgcn_readfirstlane(cc.y));
            const float cy = __int_as_float(__builtin_amdgcn_readfirstlane(cc.z));
            const float cz = __int_as_float(__builtin_amdgcn_readfirstlane(cc.w));
            if constexpr (ZLDS) {
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
                        const v2f d = dist2_cuda_v2(dx, dy, dz);                                   // contraction policy: fps_common.h (tf_sampling_g.cu:142)
                        td[pp][0] = vmin_f32(d[0], td[pp][0]);             // :143
                        td[pp][1] = vmin_f32(d[1], td[pp][1]);
                    }
                    zq = zn;
                    __builtin_amdgcn_sched_barrier(0);     // keeps the z double-buffer from being hoisted across quads (VGPR budget)
                }
            } else {
#pragma unroll
                for (int pp = 0; pp < P / 2; ++pp) {
                    const v2f dx = x[pp] - cx, dy = y[pp] - cy, dz = z[pp] - cz;
                    const v2f d = dist2_cuda_v2(dx, dy, dz);
                    td[pp][0] = vmin_f32(d[0], td[pp][0]);
                    td[pp][1] = vmin_f32(d[1], td[pp][1]);
                }
            }
        }

        // ---- 1b. refresh the cached candidate {best point, runner-up bound} only when it was picked or hit ----
        if (need) {
            int best = NEG_ONE_BITS;
#pragma unroll
            for (int pp = 0; pp < P / 2; ++pp) best = vmax3_i32(best, __float_as_int(td[pp][0]), __float_as_int(td[pp][1]));
            wmax = wave_max_i32(best);
            const int lw = __builtin_ctzll(__ballot(best == wmax));
            const int s1 = wave_max_i32(lane == lw ? NEG_ONE_BITS : best);
            float* trow = s_trow + wave * P;
            if (lane == lw) {
#pragma unroll
                for (int p = 0; p < P; p += 2) *reinterpret_cast<v2f*>(trow + p) = td[p >> 1];
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            const int tv = lane < P ? __float_as_int(trow[lane]) : NEG_ONE_BITS;
            const int fp = __builtin_ctzll(__ballot(tv == wmax));
            const int s2 = wave_max_i32(lane == fp ? NEG_ONE_BITS : tv);
            const int qw = fp / GP, iw = fp % GP;
            float fx = 0.f, fy = 0.f, fz = 0.f;
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                if (qw == q) {
#pragma unroll
                    for (int i = 0; i < GP; ++i) {
                        if (iw == i) {
                            const int p = q * GP + i;
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
                int voff = 0;                    // vector load on purpose (keeps the load off lgkmcnt, see fps_cell_kernel)
                asm volatile("" : "+v"(voff));
                ck = pm[cpos + voff];
            }
            cbound = max(s1, s2);
            cfx = fx; cfy = fy; cfz = fz;
            need = false;
        }

        // ---- 2. local ranking; the best A candidates of this workgroup go out as granules ----
        const int par = round & 1;
        const unsigned epoch = (unsigned)round + 1u;
        ++round;
        const int lbuf = par * 2 * FPS_W;
        if (lane == 0) {
            s_loc[lbuf + wave * 2 + 0] = make_int4(cv, __float_as_int(cfx), __float_as_int(cfy), __float_as_int(cfz));
            s_loc[lbuf + wave * 2 + 1] = make_int4(cpos, cbound, 0, 0);
        }
        __syncthreads();
        unsigned long long* xw = xs + (size_t)par * FPSM_HALF;
        int myid = -1;                            // id of this wave's published record
        {
            const int l15 = lane & 15;
            const int lv = s_loc[lbuf + l15 * 2].x;
            const unsigned better = (unsigned)(__ballot(lv > cv) & 0xFFFFull);
            const unsigned equal = (unsigned)(__ballot(lv == cv && l15 != wave) & 0xFFFFull);
            const unsigned valid = (unsigned)(__ballot(lv >= 0) & 0xFFFFull);
            const int lr = cv >= 0 ? __builtin_popcount(better) + __builtin_popcount(equal & ((1u << wave) - 1u)) : 64;
            if (lr < A) {
                myid = g * A + lr;
                const unsigned pay = lane == 0 ? (unsigned)cv
                                   : lane == 1 ? (unsigned)__float_as_int(cfx)
                                   : lane == 2 ? (unsigned)__float_as_int(cfy)
                                   : lane == 3 ? (unsigned)__float_as_int(cfz)
                                   : lane == 4 ? ((unsigned)cpos | (equal != 0u ? 0x80000000u : 0u))
                                               : (unsigned)cbound;
                if (lane < FPSM_REC_GRANULES) granule_store(xw + (size_t)myid * FPSM_REC_GRANULES + lane, epoch, pay);
            }
            if (wave == 0) {                      // fewer than A live candidates: the empty slots still have to show this epoch
                for (int sl = __builtin_popcount(valid); sl < A; ++sl)
                    if (lane < FPSM_REC_GRANULES)
                        granule_store(xw + (size_t)(g * A + sl) * FPSM_REC_GRANULES + lane, epoch, lane == 0 ? (unsigned)NEG_ONE_BITS : 0u);
            }
        }

        // ---- 3. wave 0 sweeps the R published records of the scene into LDS ----
        const int gbuf = par * 130;
        if (wave == 0) {
            unsigned pv[FPSM_REC_GRANULES] = {(unsigned)NEG_ONE_BITS, 0u, 0u, 0u, 0u, (unsigned)NEG_ONE_BITS};
            const unsigned long long* src = xw + (size_t)lane * FPSM_REC_GRANULES;
            const long long t0 = wall_clock64();
            bool fail = false;
            for (;;) {
                bool okl = true;
                if (lane < R) {
#pragma unroll
                    for (int k = 0; k < FPSM_REC_GRANULES; ++k) {
                        const unsigned long long v = granule_load(src + k);
                        pv[k] = (unsigned)v;
                        okl &= (unsigned)(v >> 32) == epoch;
                    }
                }
                if (__ballot(!okl) == 0ull) break;
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > FPS_MULTI_TIMEOUT_TICKS) { fail = true; break; }
            }
            s_glob[gbuf + lane * 2 + 0] = make_int4((int)pv[0], (int)pv[1], (int)pv[2], (int)pv[3]);
            s_glob[gbuf + lane * 2 + 1] = make_int4((int)pv[4], (int)pv[5], 0, 0);
            if (fail && lane == 0) { s_misc[1] = 1; *status = 1; }
        }
        __syncthreads();
        if (s_misc[1]) return;

        // ---- 4. acceptance test, spread over the waves: wave w judges records 4w..4w+3, 16 lanes per record ----
        const int ibuf = par * 64;
        {
            const int i = wave * 4 + (lane >> 4), sub = lane & 15, sh = lane & 48;
            const int4 ri = s_glob[gbuf + i * 2];
            const int rflag = s_glob[gbuf + i * 2 + 1].x;
            int cnt = 0, bb = NEG_ONE_BITS;
            bool tie = false, conf = false;
            const int T = (R + 15) >> 4;
            for (int tt = 0; tt < T; ++tt) {
                const int jx = sub + 16 * tt;
                const int4 rj = s_glob[gbuf + jx * 2];
                const int rjb = s_glob[gbuf + jx * 2 + 1].y;
                const bool btr = rj.x > ri.x;
                const bool eq = rj.x == ri.x && jx != i;
                // point = record i, centre = record jx, exactly as the update evaluates it
                const float dd = dist2_cuda(__int_as_float(ri.y) - __int_as_float(rj.y), __int_as_float(ri.z) - __int_as_float(rj.z),
                                            __int_as_float(ri.w) - __int_as_float(rj.w));
                const bool cf = btr && dd < __int_as_float(ri.x);
                const unsigned long long mb = __ballot(btr), me = __ballot(eq), mc = __ballot(cf);
                cnt += __builtin_popcount((unsigned)(mb >> sh) & 0xFFFFu);
                tie |= ((unsigned)(me >> sh) & 0xFFFFu) != 0u;
                conf |= ((unsigned)(mc >> sh) & 0xFFFFu) != 0u;
                bb = max(bb, btr ? rjb : NEG_ONE_BITS);
            }
            bb = row_max_i32(bb);
            const bool okr = ri.x >= 0 && !tie && rflag >= 0 && (cnt == 0 || (!conf && ri.x > bb));
            if (sub == 0) s_info[ibuf + i] = (ri.x >= 0 ? cnt : 64) | (okr ? 512 : 0);
        }
        __syncthreads();

        // ---- the batch: run of acceptable ranks 0, 1, ... (every wave, redundantly) ----
        const int4 mine = s_glob[gbuf + lane * 2];                 // record `lane`: {value, x, y, z}
        const int info = s_info[ibuf + lane];
        const int rk = info & 255;
        const int vtop = wave_max_i32(mine.x);
        int na = 0;
        unsigned long long alist = 0, amask = 0;
        {
            const int lim = min(A, m - j);
            bool run = vtop != 0;                                  // maximum 0: everything is covered -> tie phase + terminate
#pragma unroll
            for (int r = 0; r < FPSM_AMAX; ++r) {
                const unsigned long long mr = __ballot(rk == r && (info & 512));
                if (run && r < lim && mr != 0ull) {
                    const int id = __builtin_ctzll(mr);
                    alist |= (unsigned long long)id << (7 * r);
                    amask |= 1ull << id;
                    ++na;
                } else {
                    run = false;
                }
            }
        }
        // which published centres would lower the cached candidate's min-distance (own record excluded)
        const float ddo = dist2_cuda(cfx - __int_as_float(mine.y), cfy - __int_as_float(mine.z), cfz - __int_as_float(mine.w));
        const unsigned long long confm = __ballot(mine.x >= 0 && lane != myid && ddo < __int_as_float(cv));
        if (na > 0) {
            if (myid >= 0 && ((amask >> myid) & 1ull)) {           // own candidate accepted as pick number j + rank
                const int myrk = s_info[ibuf + myid] & 255;
                if (lane == 0) o[j + myrk] = ck;                   // tf_sampling_g.cu:166-168
                need = true;
            }
            if ((confm & amask) != 0ull) need = true;
            nacc = na;
            acc_list = alist;
            abuf = gbuf;
            j += na;
            continue;
        }

        // ---- 5. tie phase: the maximum is shared (or is 0).  One pick: lowest reference tie rank among ALL cells holding it ----
        {
            unsigned long long* xt = xw + FPSM_RMAX * FPSM_REC_GRANULES;
            const bool holder = cv == vtop && cv >= 0;
            const int myrank = (int)fps_tie_rank(ck);
            if (holder && lane == 0) atomicMin(&s_misc[0], myrank);
            __syncthreads();
            const int lmin = s_misc[0];
            if (holder && myrank == lmin) {
                const unsigned pay = lane == 0 ? (unsigned)lmin : lane == 1 ? (unsigned)__float_as_int(cfx)
                                   : lane == 2 ? (unsigned)__float_as_int(cfy) : (unsigned)__float_as_int(cfz);
                if (lane < FPSM_TIE_GRANULES) granule_store(xt + (size_t)g * FPSM_TIE_GRANULES + lane, epoch, pay);
            }
            if (lmin == TIE_NONE && wave == 0 && lane < FPSM_TIE_GRANULES)
                granule_store(xt + (size_t)g * FPSM_TIE_GRANULES + lane, epoch, lane == 0 ? (unsigned)TIE_NONE : 0u);
            if (wave == 0) {
                unsigned tvv[FPSM_TIE_GRANULES] = {(unsigned)TIE_NONE, 0u, 0u, 0u};
                const unsigned long long* src = xt + (size_t)lane * FPSM_TIE_GRANULES;
                const long long t0 = wall_clock64();
                bool fail = false;
                for (;;) {
                    bool okl = true;
                    if (lane < G) {
#pragma unroll
                        for (int k = 0; k < FPSM_TIE_GRANULES; ++k) {
                            const unsigned long long v = granule_load(src + k);
                            tvv[k] = (unsigned)v;
                            okl &= (unsigned)(v >> 32) == epoch;
                        }
                    }
                    if (__ballot(!okl) == 0ull) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (wall_clock64() - t0 > FPS_MULTI_TIMEOUT_TICKS) { fail = true; break; }
                }
                const int gmin = -wave_max_i32(-(int)tvv[0]);                       // ranks are < 2^31
                const int wl = __builtin_ctzll(__ballot((int)tvv[0] == gmin));
                const int wx = __builtin_amdgcn_readlane((int)tvv[1], wl);
                const int wy = __builtin_amdgcn_readlane((int)tvv[2], wl);
                const int wz = __builtin_amdgcn_readlane((int)tvv[3], wl);
                if (lane == 0) {
                    s_glob[gbuf + 128] = make_int4(vtop, wx, wy, wz);
                    s_misc[2] = gmin;
                    if (fail) { s_misc[1] = 1; *status = 1; }
                }
            }
            __syncthreads();
            if (s_misc[1]) return;
            if (t == 0) s_misc[0] = TIE_NONE;            // re-armed after every reader has passed; the next atomicMin is three barriers away
            const int gmin = s_misc[2];
            const int kwin = fps_tie_rank_inv((unsigned)gmin);
            const int4 wc = s_glob[gbuf + 128];
            if (holder && myrank == gmin) need = true;                               // it was this wave's candidate
            {
                const float dw = dist2_cuda(cfx - __int_as_float(wc.y), cfy - __int_as_float(wc.z), cfz - __int_as_float(wc.w));
                if (cv >= 0 && dw < __int_as_float(cv)) need = true;
            }
            if (g == 0 && t == 0) o[j] = kwin;
            nacc = 1;
            acc_list = 64;
            abuf = gbuf;
            j += 1;
            if (vtop == 0) { termk = kwin + 1; break; }
        }
    }
    // degenerate tail (max min-distance == 0): the reference keeps returning the rank-minimal point
    if (termk != 0 && g == 0)
        for (int jj = j + t; jj < m; jj += FPS_T) o[jj] = termk - 1;
}

template <int P, bool ZLDS>
static int launch_fps_multi(int b, int n, int m, int csz, int G, const float* sxyz, const int* perm, const float* inp0, int stride0,
                            int* out, unsigned long long* xch, int* status, hipStream_t st) {
    const size_t lds = L_Z + (ZLDS ? (size_t)P * FPS_T * sizeof(float) : 0);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_multi_kernel<P, ZLDS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    const int A = min(FPSM_AMAX, FPSM_RMAX / G);
    // scene groups: at most FPS_MULTI_MAX_WG co-resident workgroups per launch; whole multiples of 8 scenes (one per XCD) while that
    // fits, fewer than 8 scenes per launch beyond G = 16 (the blocks of the absent scenes' residue classes exit at once, so
    // bs * G <= FPS_MULTI_MAX_WG workgroups stay resident -- never 8 * G = up to every CU of the chip)
    int per = FPS_MULTI_MAX_WG / G;
    if (per >= 8) per &= ~7;
    if (per < 1) per = 1;
    for (int s0 = 0; s0 < b; s0 += per) {
        const int bs = min(per, b - s0);
        const int grid = ((bs + 7) / 8) * G * 8;
        hipLaunchKernelGGL((fps_multi_kernel<P, ZLDS>), dim3(grid), dim3(FPS_T), lds, st, bs, n, m, csz, G, A,
                           sxyz + (size_t)s0 * n * 3, perm + (size_t)s0 * n, inp0 + (size_t)s0 * stride0, stride0,
                           out + (size_t)s0 * m, xch + (size_t)s0 * FPSM_XCH_GRANULES, status);
    }
    return gspn_launch_status();
}

// G workgroups per scene.  A workgroup holds at most 32768 points (2048 per wave: x, y, min-distance in 96 VGPRs, z in LDS), but the
// 1024-point-per-wave instance keeps everything in registers with room to spare and the 1536-point one (z in LDS) nearly so.  Every
// workgroup more makes a round's exchange longer, so the default is ceil(n / 16384) while that is <= 10, then ceil(n / 24576) while
// that is <= 32, and the fewest that hold the scene beyond (measured on MI355X, ms for m = 30000: n = 150000: G 7..10 26.1-26.6, 16: 36;
// n = 250000: 11: 31.2, 16: 35.7; n = 400000: 17: 42.6, 25: 60.9).  The caller may ask for more (finer cells), never for fewer.
static int fps_multi_pick_g(int n, int G) {
    const int gmin = (n + 32767) / 32768;
    if (G <= 0) {
        G = (n + 16383) / 16384;
        if (G > 10) G = (n + 24575) / 24576;
        if (G > FPSM_GMAX) G = gmin;
    }
    if (G < gmin) G = gmin;
    return G;
}

extern "C" long gspn_fps_multi_ws_bytes(int b, int n) {
    if (b < 0 || n <= 0) return GSPN_ERR_ARG;
    return (long)b * n * 16 + (long)b * FPSM_XCH_GRANULES * 8 + 64;
}

// workspace: [perm b*n i32][sxyz b*n*3 f32][exchange granules b*FPSM_XCH_GRANULES u64][status: 16 i32]
static void fps_multi_carve(int b, int n, void* ws, int** perm, float** sxyz, unsigned long long** xch, int** status) {
    *perm = reinterpret_cast<int*>(ws);
    *sxyz = reinterpret_cast<float*>(*perm + (size_t)b * n);
    *xch = reinterpret_cast<unsigned long long*>(*sxyz + (size_t)b * n * 3);
    *status = reinterpret_cast<int*>(*xch + (size_t)b * FPSM_XCH_GRANULES);
}

extern "C" int gspn_fps_multi_prepass(int b, int n, int G, const float* inp, void* ws, void* stream) {
    if (b < 0 || n <= 0 || G < 0) return GSPN_ERR_ARG;
    if (b == 0) return 0;
    if (!inp || !ws) return GSPN_ERR_ARG;
    G = fps_multi_pick_g(n, G);
    if (G > FPSM_GMAX || (long long)n >= (1ll << 31) / 3 || b > 65535) return GSPN_ERR_UNSUPPORTED;
    int* perm; float* sxyz; unsigned long long* xch; int* status;
    fps_multi_carve(b, n, ws, &perm, &sxyz, &xch, &status);
    const int ncell = G * FPS_W;
    return gspn_fps_prepass_cells(b, n, ncell, (n + ncell - 1) / ncell, inp, perm, sxyz, (hipStream_t)stream);
}

extern "C" int gspn_fps_multi_sample(int b, int n, int m, int G, const float* inp, void* ws, int* out, void* stream) {
    if (b < 0 || n <= 0 || m <= 0 || G < 0) return GSPN_ERR_ARG;
    if (b == 0) return 0;
    if (!inp || !ws || !out) return GSPN_ERR_ARG;
    G = fps_multi_pick_g(n, G);
    if (G > FPSM_GMAX || (long long)n >= (1ll << 31) / 3 || b > 65535) return GSPN_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    int* perm; float* sxyz; unsigned long long* xch; int* status;
    fps_multi_carve(b, n, ws, &perm, &sxyz, &xch, &status);
    // every polled word is zeroed before every launch (tags of an earlier call must not look like this call's epochs)
    hipError_t e = hipMemsetAsync(xch, 0, (size_t)b * FPSM_XCH_GRANULES * 8 + 64, st);
    if (e != hipSuccess) return (int)e;
    // a bounded wait that expires leaves the status word at 1 and `out` partly written: zero it first, so that even then every entry
    // is a VALID index (point 0) and a caller that gathers before looking at the status word cannot read out of bounds
    e = hipMemsetAsync(out, 0, (size_t)b * m * sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    const int ncell = G * FPS_W;
    const int csz = (n + ncell - 1) / ncell;
    if (csz <= 64 * 2) return launch_fps_multi<2, false>(b, n, m, csz, G, sxyz, perm, inp, n * 3, out, xch, status, st);
    if (csz <= 64 * 4) return launch_fps_multi<4, false>(b, n, m, csz, G, sxyz, perm, inp, n * 3, out, xch, status, st);
    if (csz <= 64 * 8) return launch_fps_multi<8, false>(b, n, m, csz, G, sxyz, perm, inp, n * 3, out, xch, status, st);
    if (csz <= 64 * 16) return launch_fps_multi<16, false>(b, n, m, csz, G, sxyz, perm, inp, n * 3, out, xch, status, st);
    if (csz <= 64 * 24) return launch_fps_multi<24, true>(b, n, m, csz, G, sxyz, perm, inp, n * 3, out, xch, status, st);
    if (csz <= 64 * 32) return launch_fps_multi<32, true>(b, n, m, csz, G, sxyz, perm, inp, n * 3, out, xch, status, st);
    return GSPN_ERR_UNSUPPORTED;
}

extern "C" int gspn_farthestpointsampling_multi(int b, int n, int m, int G, const float* inp, void* ws, int* out, void* stream) {
    const int rc = gspn_fps_multi_prepass(b, n, G, inp, ws, stream);
    if (rc) return rc;
    return gspn_fps_multi_sample(b, n, m, G, inp, ws, out, stream);
}

// device address of the status word inside a workspace (for a caller that copies it asynchronously and checks it at its own next
// synchronisation point: gspn_amd/tf_sampling.py)
extern "C" long gspn_fps_multi_status_offset(int b, int n) {
    if (b <= 0 || n <= 0) return GSPN_ERR_ARG;
    return (long)b * n * 16 + (long)b * FPSM_XCH_GRANULES * 8;
}

// status word of a workspace the sampling kernel has finished with: 0 = ok, 1 = a bounded wait expired (output invalid).
// Synchronises `stream`.
extern "C" int gspn_fps_multi_status(const void* ws, int b, int n, void* stream) {
    if (!ws || b <= 0 || n <= 0) return GSPN_ERR_ARG;
    int* perm; float* sxyz; unsigned long long* xch; int* status;
    fps_multi_carve(b, n, const_cast<void*>(ws), &perm, &sxyz, &xch, &status);
    int h = 0;
    hipError_t e = hipMemcpyAsync(&h, status, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    return h;
}
