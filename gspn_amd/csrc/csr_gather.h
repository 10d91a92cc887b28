// csr_gather.h -- the gather-form gradients of grouping and interpolation (gspn_sa_group_concat_grad_csr, gspn_fp_concat_grad_csr) on
// sixteen lanes per target row (r04).
//
//   out[s][j][0 : c) = sum over e in [offsets[s][j], offsets[s][j+1]), ASCENDING e, of  wt(e) * src[s][row(order[s][e])][col0 : col0 + c)
//
// with row(p) = p, wt = 1 for the grouping (tf_grouping_g.cu:66-83; any fixed order is as faithful as the reference's atomicAdd) and
// row(p) = p / 3, wt = weight[s][p] for the interpolation -- ascending (i, t): the order of the reference's sequential loop
// (tf_interpolate.cpp:131-153), so the sums are bit-identical to it.
//
// Round 3's kernels gave every target a whole wave, lane = channel: one 256-byte row per load instruction, four in flight, every list
// entry's index fetched by all 64 lanes -- 7-12 % of the wave cycles issued an instruction, the rest waited on a chain of dependent
// loads (profiles/r03_sq_pmc_by_kernel.txt).  Here a target owns one DPP row of 16 lanes, a lane owns 4 * CPL channels (float4 loads), so
// a wave walks FOUR lists at once; the 16 lanes fetch the next 16 list entries (and their weights) with one coalesced load each and
// hand them round with row_newbcast DPP moves -- no LDS, no per-entry index load -- and up to 8 row loads per lane are in flight
// before the first is consumed.  The accumulation order per target is unchanged: ascending e, one rounding per product and per sum
// (the translation unit is built -ffp-contract=off).
// Targets are dealt to the workgroups so that XCD x (= blockIdx % 8) takes the x-th eighth of the (scene, target) space -- with eight
// scenes, one scene per XCD: every re-read of a source row (3 per row for the interpolation) finds the row in that XCD's L2 or not
// at all, instead of all eight L2s streaming all scenes.
#pragma once
#include <type_traits>

#include "common.h"

template <int K>
__device__ __forceinline__ int row_bcast_i32(int v) {          // lane K of every 16-lane row to all lanes of that row
    return __builtin_amdgcn_update_dpp(v, v, 0x150 + K, 0xF, 0xF, false);
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// row loads issued together per 16-lane row (CPL = 1; halved / quartered for wider rows).  r04 measured 8 vs 16 on the uniform clouds (no
// difference) and kept 8; on the room scenes (S: skewed list lengths, the longest lists ARE the kernel's time) 16 takes csr_gather16<1,true>
// from 64 to ~50 us and the captured layers from 1.605 to 1.585 ms, U unchanged (tools/r05_csr_u.sh).
#ifndef GSPN_CSR_U1
#define GSPN_CSR_U1 16
#endif
struct CsrCopy {                      // the skip-link columns of fp_concat's gradient are a plain slice of grad_out: trailing workgroups copy it
    const float* g; float* dst; int ld, c0, c1; long total;
};

// SPLIT (r06; CPL = 1 only; callers whose summation order is NOT pinned to the reference's loop): a 16-lane row walks at most `split_t` entries of its list alone;
// what lies beyond -- on clustered clouds one sparse point is the nearest neighbour of over a thousand dense points, and that one row used to BE the kernel's time
// (63.8 us on the room scenes against 17.6 on uniform clouds) -- is then walked by all sixteen rows of the workgroup together (row r takes entries r, r + 16, ...),
// the sixteen partial sums meet in LDS and are added to the owner's sum in row order.  A fixed order, so run-to-run identical bits; a DIFFERENT order from the
// sequential one for lists longer than split_t, which is why the reference-pinned gradient (gspn_fp_concat_grad_csr: three_interpolate_grad) never takes it.
template <int CPL, bool WEIGHTED, bool SPLIT = false>
__global__ __launch_bounds__(256) void csr_gather16_kernel(int nt, int L, long src_scene_floats, int ld, int col0, const float* __restrict__ src,
                                                           const int* __restrict__ order, const int* __restrict__ offsets,
                                                           const float* __restrict__ weight, float* __restrict__ out, long ntot, long part,
                                                           unsigned gather_blocks, CsrCopy cp, int chunks, int split_t = 0) {
    static_assert(!SPLIT || CPL == 1, "the split walk is written for one float4 per lane");
    constexpr int U = CPL == 1 ? GSPN_CSR_U1 : (CPL == 2 ? GSPN_CSR_U1 / 2 : GSPN_CSR_U1 / 4);          // list entries whose row loads are issued together
    if (blockIdx.x >= gather_blocks) {
        const long nthreads = (long)(gridDim.x - gather_blocks) * 256;
        for (long i = (long)(blockIdx.x - gather_blocks) * 256 + threadIdx.x; i < cp.total; i += nthreads) {
            const long row = i / cp.c1;
            const int l = (int)(i - row * cp.c1);
            cp.dst[i] = cp.g[row * cp.ld + cp.c0 + l];
        }
        return;
    }
    const int q = threadIdx.x & 15, sub = threadIdx.x >> 4;
    // a work item = (target, chunk of 64 * CPL channels); `chunks` > 1 spreads a wide row over several 16-lane rows, each with its own
    // loads in flight (CPL = 1 then: 8 per lane) -- the walks are latency-bound, so more rows in flight beat fewer, fatter loads
    const long lt = (long)(blockIdx.x >> 3) * 16 + sub;
    const long it0 = (long)(blockIdx.x & 7) * part + lt;
    const bool live = !(lt >= part || it0 >= ntot * chunks);
    if (!SPLIT && !live) return;                                 // (whole 16-lane rows leave together: the DPP moves below stay inside live rows)
    const long it = live ? it0 : 0;                              // (SPLIT: idle rows stay for the workgroup's barriers and help with the long lists)
    const long tg = it / chunks;
    const int ch = (int)(it - tg * chunks);
    const int s = (int)(tg / nt), j = (int)(tg - (long)s * nt);
    const int* off = offsets + (size_t)s * (nt + 1);
    const int e0 = live ? off[j] : 0, e1_full = live ? off[j + 1] : 0;
    const int e1 = (SPLIT && e1_full - e0 > split_t) ? e0 + split_t : e1_full;
    const int* ord = order + (size_t)s * L;
    const float* w = WEIGHTED ? weight + (size_t)s * L : nullptr;
    const char* gs = reinterpret_cast<const char*>(src + (size_t)s * src_scene_floats + col0 + 64 * CPL * ch + 4 * q);
    float4 acc[CPL];
#pragma unroll
    for (int h = 0; h < CPL; ++h) acc[h] = make_float4(0.f, 0.f, 0.f, 0.f);
    // this lane's entry of the first 16: byte offset of its source row, its weight
    int ob = 0;
    float wt = 1.f;
    if (e0 + q < e1) {
        const int p = ord[e0 + q];
        ob = (WEIGHTED ? p / 3 : p) * (ld * 4);
        if (WEIGHTED) wt = w[p];
    }
    for (int eb = e0; eb < e1; eb += 16) {
        const int cnt = min(16, e1 - eb);
        int obn = 0;
        float wtn = 1.f;
        const int men = eb + 16 + q;
        int pn = 0;
        if (men < e1) pn = ord[men];                              // the next 16 entries' indices: in flight under this round's row loads
        static_for<0, 16 / U>([&](auto gi) {
            constexpr int K0 = decltype(gi)::value * U;
            if (K0 < cnt) {                                      // (uniform per 16-lane row)
                const int ob0 = row_bcast_i32<K0>(ob);
                float4 v[U][CPL];
                float ws[U];
                static_for<0, U>([&](auto ui) {
                    constexpr int u = decltype(ui)::value, K = K0 + u;
                    int o = row_bcast_i32<K>(ob);
                    ws[u] = __int_as_float(row_bcast_i32<K>(__float_as_int(wt)));
                    o = K < cnt ? o : ob0;                       // past the list's end: re-read the group's first row (a cache hit), not added
#pragma unroll
                    for (int h = 0; h < CPL; ++h) v[u][h] = *reinterpret_cast<const float4*>(gs + (size_t)(unsigned)o + 256 * h);
                });
                static_for<0, U>([&](auto ui) {
                    constexpr int u = decltype(ui)::value, K = K0 + u;
                    if (K < cnt) {
#pragma unroll
                        for (int h = 0; h < CPL; ++h) {
                            if (WEIGHTED) {
                                acc[h].x += v[u][h].x * ws[u]; acc[h].y += v[u][h].y * ws[u]; acc[h].z += v[u][h].z * ws[u]; acc[h].w += v[u][h].w * ws[u];
                            } else {
                                acc[h].x += v[u][h].x; acc[h].y += v[u][h].y; acc[h].z += v[u][h].z; acc[h].w += v[u][h].w;
                            }
                        }
                    }
                });
            }
        });
        if (men < e1) {
            obn = (WEIGHTED ? pn / 3 : pn) * (ld * 4);
            if (WEIGHTED) wtn = w[pn];
        }
        ob = obn;
        wt = wtn;
    }
    if constexpr (SPLIT) {
        __shared__ int s_t0[16], s_t1[16], s_sc[16], s_chn[16];
        __shared__ float4 s_part[16][16];
        if (q == 0) { s_t0[sub] = e1; s_t1[sub] = e1_full; s_sc[sub] = s; s_chn[sub] = ch; }
        __syncthreads();
        for (int R = 0; R < 16; ++R) {
            const int t0 = s_t0[R], t1 = s_t1[R];
            if (t1 <= t0) continue;                              // (uniform over the workgroup: nearly every list ends inside split_t)
            const int sc = s_sc[R];
            const int* ordR = order + (size_t)sc * L;
            const float* wR = WEIGHTED ? weight + (size_t)sc * L : nullptr;
            const char* gR = reinterpret_cast<const char*>(src + (size_t)sc * src_scene_floats + col0 + 64 * s_chn[R] + 4 * q);
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int e = t0 + sub; e < t1; e += 64) {            // this row's entries e, e + 16, e + 32, e + 48 in flight together, added in ascending order
                int pp[4];
                float ww[4];
                float4 vv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ee = e + 16 * u;
                    pp[u] = ee < t1 ? ordR[ee] : ordR[t0];
                    ww[u] = WEIGHTED ? wR[pp[u]] : 1.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) vv[u] = *reinterpret_cast<const float4*>(gR + (size_t)(unsigned)((WEIGHTED ? pp[u] / 3 : pp[u]) * (ld * 4)));
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (e + 16 * u < t1) {
                        if (WEIGHTED) { a.x += vv[u].x * ww[u]; a.y += vv[u].y * ww[u]; a.z += vv[u].z * ww[u]; a.w += vv[u].w * ww[u]; }
                        else { a.x += vv[u].x; a.y += vv[u].y; a.z += vv[u].z; a.w += vv[u].w; }
                    }
            }
            s_part[sub][q] = a;
            __syncthreads();
            if (sub == R) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float4 p4 = s_part[r][q]; acc[0].x += p4.x; acc[0].y += p4.y; acc[0].z += p4.z; acc[0].w += p4.w; }
            }
            __syncthreads();
        }
        if (!live) return;
    }
    float* o = out + ((size_t)tg * chunks + ch) * (64 * CPL) + 4 * q;
#pragma unroll
    for (int h = 0; h < CPL; ++h) *reinterpret_cast<float4*>(o + 64 * h) = acc[h];
}

// c == 64 * CPL, 16-byte aligned rows, byte offsets inside a scene's source below 2^31; returns GSPN_ERR_UNSUPPORTED otherwise (the caller
// then takes the wave-per-target kernel)
static int csr_gather16(bool weighted, int b, int nt, int L, long src_rows_per_scene, int c, int ld, int col0, const float* src, const int* order,
                        const int* offsets, const float* weight, float* out, const CsrCopy& cp, hipStream_t st, int split_t = 0) {
    static const bool off = getenv("GSPN_CSR_GATHER16") && atoi(getenv("GSPN_CSR_GATHER16")) == 0;      // comparison switch, read once
    if (off) return GSPN_ERR_UNSUPPORTED;
    if (c != 64 && c != 128 && c != 256) return GSPN_ERR_UNSUPPORTED;
    if ((ld & 3) || (col0 & 3) || ((uintptr_t)src % 16) || ((uintptr_t)out % 16)) return GSPN_ERR_UNSUPPORTED;
    if (src_rows_per_scene * (long)ld * 4 >= (1L << 31)) return GSPN_ERR_UNSUPPORTED;
    const long ntot = (long)b * nt;
    static const int wide = getenv("GSPN_CSR_WIDE") ? atoi(getenv("GSPN_CSR_WIDE")) : 0;      // (A/B hook: 1 = one 16-lane row per target whatever its width)
    const int chunks = (!wide && c == 128) ? 2 : 1;              // measured (tools/r04_gather_family.py): 128 channels as two rows 13.5 -> 9.0 us; 256 as four 7.8 -> 9.7
    const long part = ((ntot * chunks + 7) / 8 + 15) / 16 * 16;   // work items per XCD slice, whole workgroups
    const long gb = 8 * (part / 16);
    long cb = 0;
    if (cp.total > 0) {
        cb = (cp.total + 256 * 16 - 1) / (256 * 16);               // ~16 elements per thread
        if (cb > 2048) cb = 2048;
    }
    if (gb + cb > 0x7FFFFFFFl) return GSPN_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)(gb + cb));
    const long ssf = src_rows_per_scene * (long)ld;
#define CSR_GO(CPL_, W_) hipLaunchKernelGGL((csr_gather16_kernel<CPL_, W_>), grid, dim3(256), 0, st, nt, L, ssf, ld, col0, src, order, offsets, weight, out, ntot, part, (unsigned)gb, cp, chunks)
    if (split_t > 0 && (c == 64 || chunks > 1)) {                // long lists shared out inside the workgroup (callers without a pinned summation order)
        if (weighted) hipLaunchKernelGGL((csr_gather16_kernel<1, true, true>), grid, dim3(256), 0, st, nt, L, ssf, ld, col0, src, order, offsets, weight, out, ntot, part, (unsigned)gb, cp, chunks, split_t);
        else hipLaunchKernelGGL((csr_gather16_kernel<1, false, true>), grid, dim3(256), 0, st, nt, L, ssf, ld, col0, src, order, offsets, weight, out, ntot, part, (unsigned)gb, cp, chunks, split_t);
        return gspn_launch_status();
    }
    if (weighted) { if (c == 64 || chunks > 1) CSR_GO(1, true); else if (c == 128) CSR_GO(2, true); else CSR_GO(4, true); }
    else { if (c == 64 || chunks > 1) CSR_GO(1, false); else if (c == 128) CSR_GO(2, false); else CSR_GO(4, false); }
#undef CSR_GO
    return gspn_launch_status();
}
