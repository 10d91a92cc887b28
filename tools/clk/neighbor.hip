// Who pays for a busy workgroup?  (DESIGN 4.6 open question; VERDICT r03 item 2, step iii.)
// Victim: a one-wave-of-workgroups grid (one 256-thread workgroup per CU and a bit), every workgroup runs the SAME fixed MFMA + VALU
// workload and records its hardware position (HW_ID / XCC_ID) and its own duration (wall clock, 100 MHz).  Aggressor: NA 1024-thread
// workgroups in a busy VALU loop on a second stream (they record their position too), optionally holding the CU's whole LDS.
// Output: the victim workgroups' durations alone and beside the aggressor, grouped by their relation to the aggressors' CUs:
// same CU / the other CU of the same (xcc, se, sh, cu>>1) pair / same shader engine / same XCC / elsewhere.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned hw_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 4); }        // HW_REG_HW_ID
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 20); }      // HW_REG_XCC_ID (gfx94x/95x)

template <int KIND>       // 0: MFMA chain + a few VALU (a GEMM-like victim), 1: pure VALU
__global__ __launch_bounds__(256) void victim(int iters, unsigned* pos, unsigned long long* dur, float* out) {
    const unsigned long long t0 = wall_clock64();
    f32x16 acc = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0f, v = a;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < (KIND == 0 ? 16 : 64); ++u) v = __builtin_fmaf(v, 1.0001f, 1e-3f);
    }
    if (acc[0] == 123.456f || v == 123.456f) out[0] = acc[1] + v;
    __syncthreads();
    if (threadIdx.x == 0) {
        dur[blockIdx.x] = wall_clock64() - t0;
        pos[2 * blockIdx.x] = hw_id();
        pos[2 * blockIdx.x + 1] = xcc_id();
    }
}
template <int KIND>       // 0: v_fma loop, 1: quarter-rate transcendental loop (same busy time, a quarter of the instruction fetches)
__global__ __launch_bounds__(1024) void aggressor(long long iters, unsigned* pos, float* out) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) { lds[0] = 1.f; pos[2 * blockIdx.x] = hw_id(); pos[2 * blockIdx.x + 1] = xcc_id(); }
    float a = threadIdx.x * 1e-3f + 1.0f;
    for (long long it = 0; it < iters; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 64; ++i) a = __builtin_fmaf(a, 1.0001f, 0.5f);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) a = __builtin_amdgcn_sqrtf(a) + 1.0f;
        }
    }
    if (a == 1.2345f) out[0] = a + lds[0];
}

struct Pos { unsigned xcc, se, sh, cu; };
static Pos decode(unsigned hw, unsigned xcc) {
    Pos p; p.cu = (hw >> 8) & 15; p.sh = (hw >> 12) & 1; p.se = (hw >> 13) & 7; p.xcc = xcc & 15; return p;
}

int main(int argc, char** argv) {
    const int NA = argc > 1 ? atoi(argv[1]) : 1;              // aggressor workgroups
    const int claim = argc > 2 ? atoi(argv[2]) : 0;           // 1: the aggressor asks for the CU's whole LDS
    const int akind = argc > 3 ? atoi(argv[3]) : 0;
    const int vkind = argc > 4 ? atoi(argv[4]) : 0;
    const int NV = argc > 5 ? atoi(argv[5]) : 448;            // victim workgroups (256 threads each)
    const int viters = 6000;
    unsigned *vpos, *apos; unsigned long long* vdur; float* out;
    (void)hipMalloc(&vpos, 8 * NV); (void)hipMalloc(&apos, 8 * 64); (void)hipMalloc(&vdur, 8 * NV); (void)hipMalloc(&out, 64);
    hipStream_t s1, s2; (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const size_t lds = claim ? 163840 : 0;
    if (claim) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&aggressor<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&aggressor<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    auto launch_v = [&]() {
        if (vkind == 0) hipLaunchKernelGGL((victim<0>), dim3(NV), dim3(256), 0, s1, viters, vpos, vdur, out);
        else hipLaunchKernelGGL((victim<1>), dim3(NV), dim3(256), 0, s1, viters, vpos, vdur, out);
    };
    std::vector<unsigned> hp(2 * NV), ha(2 * 64);
    std::vector<unsigned long long> hd(NV);
    // alone
    launch_v(); (void)hipStreamSynchronize(s1);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s1); launch_v(); (void)hipEventRecord(e1, s1); (void)hipStreamSynchronize(s1);
    float ms_alone; (void)hipEventElapsedTime(&ms_alone, e0, e1);
    (void)hipMemcpy(hd.data(), vdur, 8 * NV, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> d0 = hd; std::sort(d0.begin(), d0.end());
    printf("victim alone: kernel %.3f ms; workgroup durations (us) min %.1f median %.1f max %.1f\n", ms_alone, d0[0] / 100.0, d0[NV / 2] / 100.0, d0[NV - 1] / 100.0);
    // beside the aggressor: start it first, give it time to be resident, then the victim
    const long long aiters = akind == 0 ? 12000 : 12000;
    if (akind == 0) hipLaunchKernelGGL((aggressor<0>), dim3(NA), dim3(1024), lds, s2, aiters, apos, out);
    else hipLaunchKernelGGL((aggressor<1>), dim3(NA), dim3(1024), lds, s2, aiters, apos, out);
    hipEvent_t a0, a1; (void)hipEventCreate(&a0); (void)hipEventCreate(&a1);
    (void)hipStreamSynchronize(s2);
    (void)hipEventRecord(a0, s2);
    if (akind == 0) hipLaunchKernelGGL((aggressor<0>), dim3(NA), dim3(1024), lds, s2, aiters, apos, out);
    else hipLaunchKernelGGL((aggressor<1>), dim3(NA), dim3(1024), lds, s2, aiters, apos, out);
    (void)hipEventRecord(a1, s2);
    (void)hipEventRecord(e0, s1); launch_v(); (void)hipEventRecord(e1, s1);
    (void)hipStreamSynchronize(s1); (void)hipStreamSynchronize(s2);
    float ms_with, ms_aggr; (void)hipEventElapsedTime(&ms_with, e0, e1); (void)hipEventElapsedTime(&ms_aggr, a0, a1);
    (void)hipMemcpy(hd.data(), vdur, 8 * NV, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hp.data(), vpos, 8 * NV, hipMemcpyDeviceToHost);
    (void)hipMemcpy(ha.data(), apos, 8 * NA, hipMemcpyDeviceToHost);
    printf("victim beside %d aggressor workgroup(s) [%s, %s LDS]: kernel %.3f ms (aggressor kernel %.3f ms)\n", NA, akind ? "quarter-rate loop" : "v_fma loop",
           claim ? "160 KB of" : "no", ms_with, ms_aggr);
    const char* names[5] = {"same CU", "other CU of the pair", "same SE, other pair", "same XCC, other SE", "other XCC"};
    std::vector<double> g[5];
    for (int i = 0; i < NV; ++i) {
        Pos p = decode(hp[2 * i], hp[2 * i + 1]);
        int rel = 4;
        for (int j = 0; j < NA; ++j) {
            Pos q = decode(ha[2 * j], ha[2 * j + 1]);
            int r = 4;
            if (p.xcc == q.xcc) {
                r = 3;
                if (p.se == q.se) { r = 2; if (p.sh == q.sh && (p.cu >> 1) == (q.cu >> 1)) r = (p.cu == q.cu) ? 0 : 1; }
            }
            rel = std::min(rel, r);
        }
        g[rel].push_back(hd[i] / 100.0);
    }
    for (int r = 0; r < 5; ++r) {
        if (g[r].empty()) { printf("  %-22s  (none)\n", names[r]); continue; }
        std::sort(g[r].begin(), g[r].end());
        printf("  %-22s  n=%3zu  min %.1f  median %.1f  max %.1f us\n", names[r], g[r].size(), g[r][0], g[r][g[r].size() / 2], g[r].back());
    }
    for (int j = 0; j < std::min(NA, 4); ++j) { Pos q = decode(ha[2 * j], ha[2 * j + 1]); printf("  aggressor %d at xcc %u se %u sh %u cu %u\n", j, q.xcc, q.se, q.sh, q.cu); }
    return 0;
}
