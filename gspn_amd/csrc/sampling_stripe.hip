// sampling_stripe.hip -- farthest point sampling of a 16385..32768-point scene on one CU, every spatial cell STRIPED over all 16 waves
// (r05).  OPT-IN (GSPN_FPS_STRIPE=1): index-exact, but measured 1.5-1.6x SLOWER than fps_cell_kernel (sampling.hip: the dispatch site has
// the numbers and the reason) -- kept as the record of the experiment VERDICT r04 item 5 asked for, with a test.  Reference semantics: tf_ops/sampling/tf_sampling_g.cu:105-170; same outputs as fps_cell_kernel / the oracle, bit for bit.
//
// fps_cell_kernel (sampling.hip) gives each of the 16 waves one compact spatial cell: culling then removes whole waves from a pick's
// update, but what is left is unbalanced -- a round waits for the wave whose cell the accepted centres hit (2-3 applies of 16 dependent
// iterations each + a refresh, at the ~5 cycles per instruction a lone wave issues at), while the other waves of its SIMD idle
// (profiles/r04_fps_round_profile.txt: the busiest wave applies 2.3x the mean; DESIGN 4.1: "a different partition, not a tuning step").
//
// Here the same 16 cells (same pre-pass, same workspace) are laid ACROSS the waves: thread t holds points 2t and 2t+1 of EVERY cell
// (slot pair c <-> cell c).  A centre that reaches h cells costs every wave h one-instruction-group iterations (packed fp32: two points)
// instead of one wave 16 h; all four waves of every SIMD issue, and the apply segment follows the mean, not the maximum.
//   * culling is per (centre, cell) and wave-uniform: lane c (< 16) tests the centre against cell c's bounding box and the cell's max
//     min-distance as of its last refresh (an upper bound: conservative, exact), one ballot per centre -> a 16-bit cell mask;
//   * a cell's candidate (its best point + an upper bound on its runner-up) is a 16-way combine: every wave leaves its partial for the
//     cells that need a refresh in LDS (value, runner-up, position, coordinates), one barrier, then wave c -- still the owner of cell
//     c's candidate in the exchange -- combines the 16 partials inside one DPP row.  Ties go to the lowest wave, lane, slot = the lowest
//     sorted position = the reference's (k mod 512, k) order inside a cell (the pre-pass sorts cells by that rank);
//   * the exchange (publish, rank, accept the longest acceptable prefix, slow path for equal values) is fps_cell_kernel's, unchanged;
//     which cells need a refresh next round is computed by every wave from the published conflict masks (uniform).
#include "fps_common.h"

#ifndef FPS_AMAX
#define FPS_AMAX 8
#endif

namespace gspn_k {

constexpr int NC = 16;            // cells
// LDS map (bytes)
constexpr int OFF_CAND = 0;       // 2 buffers x 16 cells x 2 int4 (fps_cell_kernel's candidate records) + batch record
constexpr int OFF_INFO = 1280;    // 2 x 16 int2
constexpr int OFF_WMAX = 1536;    // 16 ints: max min-distance of every cell as of its last refresh
constexpr int OFF_BB = 1600;      // 16 x 8 floats: bounding boxes {x0,x1,y0,y1,z0,z1,-,-}
constexpr int OFF_PART = 2304;    // 16 cells x 16 waves x 2 int4 = 8 KB: refresh partials (prologue: the waves' bounding boxes)
constexpr int OFF_Z = 2304 + 8192;      // z plane: v2f [16][1024]
constexpr int LDS_BYTES = OFF_Z + NC * FPS_T * 8;

__global__ __launch_bounds__(FPS_T) void fps_stripe_kernel(int n, int m, int csz, const float* __restrict__ sxyz, const int* __restrict__ perm,
                                                           const float* __restrict__ inp0, int inp0_stride, int* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int4* s_cand = reinterpret_cast<int4*>(smem + OFF_CAND);
    int* s_wmax = reinterpret_cast<int*>(smem + OFF_WMAX);
    float* s_bb = reinterpret_cast<float*>(smem + OFF_BB);
    int4* s_part = reinterpret_cast<int4*>(smem + OFF_PART);
    v2f* s_z = reinterpret_cast<v2f*>(smem + OFF_Z);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15;
    const float* xyz = sxyz + (size_t)blockIdx.x * n * 3;
    const int* pm = perm + (size_t)blockIdx.x * n;
    int* o = out + (size_t)blockIdx.x * m;

    v2f x[NC], y[NC], td[NC];
    // ---- load: slot pair c = points 2t, 2t+1 of cell c; per-wave bounding boxes of every cell through LDS ----
    {
        float* s_wbb = reinterpret_cast<float*>(smem + OFF_PART);          // [cell][wave][8]
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float b0 = 3e38f, b1 = -3e38f, b2 = 3e38f, b3 = -3e38f, b4 = 3e38f, b5 = -3e38f;
            v2f zz;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int q = 2 * t + e;
                const int pos = c * csz + q;
                const bool live = q < csz && pos < n;
                const int pc = live ? pos : 0;
                float px = xyz[pc * 3 + 0], py = xyz[pc * 3 + 1], pz = xyz[pc * 3 + 2];
                b0 = fminf(b0, live ? px : 3e38f); b1 = fmaxf(b1, live ? px : -3e38f);
                b2 = fminf(b2, live ? py : 3e38f); b3 = fmaxf(b3, live ? py : -3e38f);
                b4 = fminf(b4, live ? pz : 3e38f); b5 = fmaxf(b5, live ? pz : -3e38f);
                x[c][e] = live ? px : 0.f;
                y[c][e] = live ? py : 0.f;
                zz[e] = live ? pz : 0.f;
                td[c][e] = live ? 1e38f : -1.0f;
            }
            s_z[c * FPS_T + t] = zz;
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) {
                b0 = fminf(b0, __shfl_xor(b0, s, 64)); b1 = fmaxf(b1, __shfl_xor(b1, s, 64));
                b2 = fminf(b2, __shfl_xor(b2, s, 64)); b3 = fmaxf(b3, __shfl_xor(b3, s, 64));
                b4 = fminf(b4, __shfl_xor(b4, s, 64)); b5 = fmaxf(b5, __shfl_xor(b5, s, 64));
            }
            if (lane == 0) {
                float* d = s_wbb + (c * 16 + wave) * 8;
                d[0] = b0; d[1] = b1; d[2] = b2; d[3] = b3; d[4] = b4; d[5] = b5;
            }
        }
        __syncthreads();
        if (t < NC * 6) {                          // thread (cell, component): the 16 waves' values
            const int c = t / 6, k = t % 6;
            float v = s_wbb[(c * 16) * 8 + k];
            for (int w = 1; w < 16; ++w) { const float o2 = s_wbb[(c * 16 + w) * 8 + k]; v = (k & 1) ? fmaxf(v, o2) : fminf(v, o2); }
            s_bb[c * 8 + k] = v;
        }
        if (t < NC) s_wmax[t] = __float_as_int(1e38f);
        if (t == 0) o[0] = 0;                                            // tf_sampling_g.cu:114-116
        const float* p0 = inp0 + (size_t)blockIdx.x * inp0_stride;
        if (t == 0) s_cand[2 * FPS_W + 0] = make_int4(0, __float_as_int(p0[0]), __float_as_int(p0[1]), __float_as_int(p0[2]));
        __syncthreads();
    }

    unsigned long long acc_list = 0;       // accepted cells of the last exchange, 4 bits each
    int abuf = 2 * FPS_W;
    int nacc = 1;
    int j = 1;
    unsigned needmask = 0xFFFFu;           // cells whose candidate must be recomputed (uniform)
    int cv = 0, ck = 0, cpos = 0, cbound = NEG_ONE_BITS;       // cached candidate of cell `wave`
    float cfx = 0.f, cfy = 0.f, cfz = 0.f;
    int round = 0;
    int termk = 0;
    int4* s_batch = s_cand + 4 * FPS_W;

    while (j < m) {
        // ---- apply the accepted centres: per centre a 16-bit mask of the cells it can change, then one packed update per set bit ----
        for (int u = 0; u < nacc; ++u) {
            const int4 cc = s_cand[abuf + (int)((acc_list >> (4 * u)) & 15ull) * 2];          // uniform address: LDS broadcast
            const float cx = __int_as_float(__builtin_amdgcn_readfirstlane(cc.y));
            const float cy = __int_as_float(__builtin_amdgcn_readfirstlane(cc.z));
            const float cz = __int_as_float(__builtin_amdgcn_readfirstlane(cc.w));
            unsigned hit;
            {
                // conservative in fp32: every point of cell l15 has dist2 >= L*(1-1e-5); a cell whose max min-distance is below that cannot change
                const float4 ba = *reinterpret_cast<const float4*>(s_bb + l15 * 8);
                const float2 bz = *reinterpret_cast<const float2*>(s_bb + l15 * 8 + 4);
                const int wm = s_wmax[l15];
                const float ex = fmaxf(fmaxf(ba.x - cx, cx - ba.y), 0.f);
                const float ey = fmaxf(fmaxf(ba.z - cy, cy - ba.w), 0.f);
                const float ez = fmaxf(fmaxf(bz.x - cz, cz - bz.y), 0.f);
                const float L = (ex * ex + ey * ey + ez * ez) * 0.99999f;
                hit = (unsigned)(__ballot(lane < NC && !(L > __int_as_float(wm)) && wm >= 0) & 0xFFFFull);
            }
            // (unrolled uniform tests, not a bit scan + switch: the register allocator spills a third of the resident points around a 16-way switch)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if ((hit >> c) & 1u) {
                    const v2f zz = s_z[c * FPS_T + t];
                    const v2f dx = x[c] - cx, dy = y[c] - cy, dz = zz - cz;
                    const v2f d = dist2_cuda_v2(dx, dy, dz);                                   // contraction policy: fps_common.h
                    td[c][0] = vmin_f32(d[0], td[c][0]);
                    td[c][1] = vmin_f32(d[1], td[c][1]);
                }
            }
        }
        // ---- refresh, part 1: every wave's partial for the cells whose candidate is gone ----
        if (needmask) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (!((needmask >> c) & 1u)) continue;
                const int b0 = __float_as_int(td[c][0]), b1 = __float_as_int(td[c][1]);
                const int best = max(b0, b1);
                const int wm = wave_max_i32(best);
                const int lw = __builtin_ctzll(__ballot(best == wm));                       // lowest lane that holds the maximum
                const int w0 = __builtin_amdgcn_readlane(b0, lw), w1 = __builtin_amdgcn_readlane(b1, lw);
                const int e = (w0 == wm) ? 0 : 1;                                            // lowest slot on ties
                const int s1 = wave_max_i32(lane == lw ? NEG_ONE_BITS : best);              // runner-up: the other lanes ...
                const int s2 = e == 0 ? w1 : w0;                                             // ... and the winning lane's other point
                const float fx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e == 0 ? x[c][0] : x[c][1]), lw));
                const float fy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e == 0 ? y[c][0] : y[c][1]), lw));
                if (lane == 0) {
                    const float fz = reinterpret_cast<const float*>(s_z)[(c * FPS_T + wave * 64 + lw) * 2 + e];
                    s_part[(c * 16 + wave) * 2 + 0] = make_int4(wm, __float_as_int(fx), __float_as_int(fy), __float_as_int(fz));
                    s_part[(c * 16 + wave) * 2 + 1] = make_int4(2 * (wave * 64 + lw) + e, max(s1, s2), 0, 0);
                }
            }
            __syncthreads();
            // ---- refresh, part 2: wave c combines the 16 partials of cell c (lanes 0..15: one DPP row) ----
            if ((needmask >> wave) & 1u) {
                const int4 pa = s_part[(wave * 16 + l15) * 2 + 0];
                const int4 pb = s_part[(wave * 16 + l15) * 2 + 1];
                const int M = __builtin_amdgcn_readfirstlane(row_max_i32(pa.x));
                const int win = __builtin_ctz((unsigned)(__ballot(pa.x == M) & 0xFFFFull));     // lowest wave on ties = lowest position
                const int bound = __builtin_amdgcn_readfirstlane(row_max_i32(l15 == win ? pb.y : pa.x));
                const int q = __builtin_amdgcn_readlane(pb.x, win);
                const int pos = wave * csz + q;
                cv = M;
                cfx = __int_as_float(__builtin_amdgcn_readlane(pa.y, win));
                cfy = __int_as_float(__builtin_amdgcn_readlane(pa.z, win));
                cfz = __int_as_float(__builtin_amdgcn_readlane(pa.w, win));
                cpos = (M >= 0 && pos < n) ? pos : 0;
                {
                    int voff = 0;
                    asm volatile("" : "+v"(voff));          // a VECTOR load: a scalar load's completion would share lgkmcnt with every LDS access
                    ck = pm[cpos + voff];
                }
                cbound = bound;
                if (lane == 0) s_wmax[wave] = M;
            }
        }
        const int buf = (round & 1) * 2 * FPS_W;
        ++round;
        if (lane == 0) {
            s_cand[buf + wave * 2 + 0] = make_int4(cv, __float_as_int(cfx), __float_as_int(cfy), __float_as_int(cfz));
            s_cand[buf + wave * 2 + 1] = make_int4(cpos, cbound, 0, 0);
        }
        __syncthreads();

        // ---- batch selection (fps_cell_kernel's): whether the candidate of rank p is acceptable GIVEN that ranks 0..p-1 are accepted is decided by its
        //      own wave; after a second barrier the batch is the run of acceptable ranks 0, 1, ... ----
        int2* s_info = reinterpret_cast<int2*>(smem + OFF_INFO) + (round & 1) * FPS_W;
        const int4 mine = s_cand[buf + l15 * 2];                  // candidate of cell l15: {v, x, y, z}
        const int4 mine2 = s_cand[buf + l15 * 2 + 1];             // {sorted position, runner-up bound, -, -}
        int myrank;
        {
            const unsigned better = (unsigned)(__ballot(mine.x > cv) & 0xFFFFull);
            const unsigned equal = (unsigned)(__ballot(mine.x == cv && l15 != wave) & 0xFFFFull);
            const float dd = dist2_cuda(cfx - __int_as_float(mine.y), cfy - __int_as_float(mine.z), cfz - __int_as_float(mine.w));
            const unsigned conf = (unsigned)(__ballot(l15 != wave && mine.x >= 0 && dd < __int_as_float(cv)) & 0xFFFFull);
            const int bb = __builtin_amdgcn_readfirstlane(row_max_i32(((better >> l15) & 1u) ? mine2.y : NEG_ONE_BITS));
            int cnt = __builtin_popcount(better);
            if (cv < 0) cnt = 64;                                 // empty / padding-only cells never rank
            myrank = cnt;
            const bool okw = cnt == 0 || ((conf & better) == 0u && cv > bb);
            if (lane == 0) s_info[wave] = make_int2(cnt | ((equal != 0u && cv >= 0) ? 256 : 0) | (okw ? 512 : 0), (int)conf);
        }
        __syncthreads();
        const int2 info = s_info[l15];
        const int rk = info.x & 255;
        const bool slow = __ballot((info.x & 256) && rk < FPS_AMAX) != 0ull;    // equal values among the leaders (rare)
        if (!slow) {
            const int okb = __builtin_amdgcn_readfirstlane(row_or_i32((rk < FPS_AMAX && (info.x & 512)) ? (1 << rk) : 0));
            const unsigned wl = (unsigned)__builtin_amdgcn_readfirstlane(row_or_i32(rk < (FPS_AMAX < 8 ? FPS_AMAX : 8) ? (l15 << (4 * rk)) : 0));
            const unsigned wh = FPS_AMAX > 8 ? (unsigned)__builtin_amdgcn_readfirstlane(row_or_i32((rk >= 8 && rk < FPS_AMAX) ? (l15 << (4 * (rk - 8))) : 0)) : 0u;
            int na = __builtin_ctz(~(unsigned)okb);
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
            if (myrank < na && lane == 0) o[j + myrank] = ck;      // own candidate accepted as pick number j + rank
            // cells to refresh next round: the accepted ones, and those whose cached candidate an accepted centre reaches
            const unsigned accm = (unsigned)(__ballot(rk < na) & 0xFFFFull);
            needmask = (unsigned)(__ballot(rk < na || ((unsigned)info.y & accm) != 0u) & 0xFFFFull);
            j += na;
            if (termk != 0) break;
            continue;
        }
        // ---- slow path: wave 0 extracts serially with reference-rank tie-breaks, then broadcasts ----
        if (wave == 0) {
            int na = 0, term = 0, jn = j;
            unsigned long long alist = 0;
            int remaining = mine.x;
            unsigned acc_mask = 0;
            int bound = NEG_ONE_BITS;
#pragma unroll 1
            for (int i = 0; i < FPS_AMAX && jn < m; ++i) {
                const int M = __builtin_amdgcn_readfirstlane(row_max_i32(remaining));
                if (M < 0) break;
                unsigned long long eq = __ballot(remaining == M) & 0xFFFFull;
                int wsel = __builtin_ctzll(eq);
                if (__builtin_popcountll(eq) > 1) {                 // equal values in several cells: lowest reference rank wins
                    unsigned rmin = 0xFFFFFFFFu;
#pragma unroll 1
                    for (int w = 0; w < 16; ++w)
                        if ((eq >> w) & 1ull) {
                            const int kw = pm[__builtin_amdgcn_readfirstlane(s_cand[buf + w * 2 + 1].x)];
                            const unsigned rw = ((unsigned)(kw & 511) << 22) | (unsigned)(kw >> 9);
                            if (rw < rmin) { rmin = rw; wsel = w; }
                        }
                }
                if (i == 0 && M == 0) {
                    term = 1 + pm[__builtin_amdgcn_readfirstlane(s_cand[buf + wsel * 2 + 1].x)];
                    break;
                }
                const int4 sa = s_cand[buf + wsel * 2];
                const int sbound = s_cand[buf + wsel * 2 + 1].y;
                if (i > 0) {
                    if (!(M > bound)) break;
                    const float dd = dist2_cuda(__int_as_float(sa.y) - __int_as_float(mine.y), __int_as_float(sa.z) - __int_as_float(mine.z),
                                                __int_as_float(sa.w) - __int_as_float(mine.w));
                    const bool hitp = ((acc_mask >> l15) & 1u) && (dd < __int_as_float(M));
                    if (__ballot(hitp) != 0ull) break;
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
            unsigned accm = 0;
            for (int u = 0; u < nacc; ++u) {
                const int wu = (int)((acc_list >> (4 * u)) & 15ull);
                accm |= 1u << wu;
                if (wu == wave && lane == 0) o[j + u] = ck;
            }
            needmask = (unsigned)(__ballot(((accm >> l15) & 1u) != 0u || ((unsigned)info.y & accm) != 0u) & 0xFFFFull);
            j = jnew;
            if (termk != 0) break;
        }
    }
    // degenerate tail (max min-distance == 0): the reference keeps returning the rank-minimal covered point
    if (termk != 0)
        for (int jj = j + t; jj < m; jj += FPS_T) o[jj] = termk - 1;
}

}  // namespace gspn_k
using namespace gspn_k;

// launcher (sampling.hip: gspn_fps_cells_strided): cells of 1025..2048 points, i.e. 16385 <= n <= 32768
int gspn_fps_stripe_launch(int b, int n, int m, int csz, const float* sxyz, const int* perm, const float* inp0, int stride0, int* out, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_stripe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    const size_t lds = gspn_claim_lds(1, reinterpret_cast<const void*>(&fps_stripe_kernel), (size_t)LDS_BYTES);
    hipLaunchKernelGGL(fps_stripe_kernel, dim3(b), dim3(FPS_T), lds, st, n, m, csz, sxyz, perm, inp0, stride0, out);
    return gspn_launch_status();
}
