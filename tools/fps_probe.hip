// fps_probe: per-phase cycle breakdown of fps_resident_kernel (build with -DFPS_PROFILE)
#include "../gspn_amd/csrc/sampling.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
int main(int argc, char** argv) {
    int n = argc > 1 ? atoi(argv[1]) : 32768, m = argc > 2 ? atoi(argv[2]) : 1024, b = 8;
    std::vector<float> h((size_t)b * n * 3);
    srand(1);
    for (auto& v : h) v = rand() / (float)RAND_MAX;
    float* d; int* o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, (size_t)b * m * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        int rc = gspn_farthestpointsampling(b, n, m, d, nullptr, o, nullptr);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long p[8]; hipMemcpyFromSymbol(p, HIP_SYMBOL(g_fps_prof), sizeof(p));
        printf("n=%d m=%d rc=%d: %.3f ms = %.0f cyc/round @2.4GHz\n", n, m, rc, ms, ms * 2.4e6 / (m - 1));
        for (int w = 0; w < 2; ++w)
            printf("  wave %d cycles/round: update %.0f | wave-argmax+resolve %.0f | lds write+barrier %.0f | read cands+reduce %.0f\n", w * 7,
                   p[w * 4 + 0] / (double)(m - 1), p[w * 4 + 1] / (double)(m - 1), p[w * 4 + 2] / (double)(m - 1), p[w * 4 + 3] / (double)(m - 1));
    }
    long long tl[128]; hipMemcpyFromSymbol(tl, HIP_SYMBOL(g_fps_tl), sizeof(tl));
    long long t0 = tl[0];
    for (int w = 0; w < 16; ++w) for (int i = 0; i < 4; ++i) if (tl[w * 8 + i] < t0) t0 = tl[w * 8 + i];
    printf("round 100 timeline (cycles since first tick): wave: end-update end-resolve after-barrier end-reduce\n");
    for (int w = 0; w < 16; ++w) printf("  w%02d: %6lld %6lld %6lld %6lld\n", w, tl[w*8]-t0, tl[w*8+1]-t0, tl[w*8+2]-t0, tl[w*8+3]-t0);
    return 0;
}
