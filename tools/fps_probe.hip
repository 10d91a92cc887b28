// fps_probe: per-phase cycle breakdown of fps_resident_kernel (build with -DFPS_PROFILE)
#include "../gspn_amd/csrc/sampling.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <numeric>
static unsigned spread10(unsigned v) { v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v; }
static void cell_probe(int n, int m, int ncell) {
    const int b = 8, csz = (n + ncell - 1) / ncell;
    std::vector<float> h((size_t)b * n * 3), sx((size_t)b * n * 3), p0(b * 3);
    std::vector<int> perm((size_t)b * n);
    srand(1);
    for (auto& v : h) v = rand() / (float)RAND_MAX;
    for (int s = 0; s < b; ++s) {
        const float* x = &h[(size_t)s * n * 3];
        std::vector<unsigned long long> key(n);
        for (int k = 0; k < n; ++k) {
            unsigned qx = std::min(1023u, (unsigned)(x[k*3]*1024)), qy = std::min(1023u, (unsigned)(x[k*3+1]*1024)), qz = std::min(1023u, (unsigned)(x[k*3+2]*1024));
            key[k] = ((unsigned long long)(spread10(qx) | (spread10(qy) << 1) | (spread10(qz) << 2)) << 32) | (unsigned)k;
        }
        std::sort(key.begin(), key.end());
        std::vector<unsigned long long> k2(n);
        for (int i = 0; i < n; ++i) { unsigned k = (unsigned)key[i]; unsigned rank = ((k & 511) << 22) | (k >> 9); k2[i] = ((unsigned long long)(i / csz) << 32) | rank; }
        std::sort(k2.begin(), k2.end());
        for (int i = 0; i < n; ++i) { unsigned rank = (unsigned)k2[i]; int k = (int)(((rank & 0x3FFFFF) << 9) | (rank >> 22)); perm[(size_t)s*n+i] = k; for (int l = 0; l < 3; ++l) sx[((size_t)s*n+i)*3+l] = x[k*3+l]; }
        for (int l = 0; l < 3; ++l) p0[s*3+l] = x[l];
    }
    float *d, *d0; int *dp, *o, *o2; float* dh;
    hipMalloc(&d, sx.size()*4); hipMalloc(&dh, h.size()*4); hipMalloc(&d0, p0.size()*4); hipMalloc(&dp, perm.size()*4); hipMalloc(&o, (size_t)b*m*4); hipMalloc(&o2, (size_t)b*m*4);
    hipMemcpy(d, sx.data(), sx.size()*4, hipMemcpyHostToDevice); hipMemcpy(dh, h.data(), h.size()*4, hipMemcpyHostToDevice);
    hipMemcpy(d0, p0.data(), p0.size()*4, hipMemcpyHostToDevice); hipMemcpy(dp, perm.data(), perm.size()*4, hipMemcpyHostToDevice);
    gspn_farthestpointsampling(b, n, m, dh, nullptr, o2, nullptr);
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        int rc = gspn_fps_cells(b, n, m, csz, d, dp, d0, o, nullptr);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
#ifdef FPS_PROFILE
        long long p[32]; hipMemcpyFromSymbol(p, HIP_SYMBOL(g_cell_prof), sizeof(p));
#endif
        std::vector<int> a((size_t)b*m), c((size_t)b*m); hipMemcpy(a.data(), o, a.size()*4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), o2, c.size()*4, hipMemcpyDeviceToHost);
        printf("CELL n=%d m=%d rc=%d: %.3f ms (%.0f cyc/pick) match_resident=%d\n", n, m, rc, ms, ms*2.4e6/(m-1), (int)(a == c));
#ifdef FPS_PROFILE
        int pw[64]; hipMemcpyFromSymbol(pw, HIP_SYMBOL(g_cell_waves), sizeof(pw));
        printf("  applies per wave:"); for (int w = 0; w < 16; ++w) printf(" %d", pw[w*4]); printf("\n  refreshes per wave:"); for (int w = 0; w < 16; ++w) printf(" %d", pw[w*4+1]); printf("\n");
        if (rep == 1) { long long tl[3*16*8]; hipMemcpyFromSymbol(tl, HIP_SYMBOL(g_cell_tl), sizeof(tl));
            for (int r = 0; r < 3; ++r) { long long t0 = tl[r*128]; for (int w = 0; w < 16; ++w) t0 = std::min(t0, tl[(r*16+w)*8]);
                printf("  round %d (cycles since the first wave finished its applies): wave: apply-end refresh-end barrier1 barrier2 scan-end\n", 200 + 100*r);
                for (int w = 0; w < 16; ++w) { long long* q = tl + (r*16+w)*8; printf("    w%02d %6lld %6lld %6lld %6lld %6lld\n", w, q[0]-t0, q[1]-t0, q[2]-t0, q[4]-t0, q[3]-t0); } } }
        if (FPS_PROFILE != 2) for (int w = 0; w < 2; ++w) { long long* q = p + w*16; double R = (double)q[4];
            if (w == 0) printf("  own-rank+barrier2 %.0f cycles/round\n", q[8]/R);
            printf("  wave %2d: rounds %lld (%.2f picks/round) applied %lld refreshed %lld | cycles/round: apply %.0f refresh %.0f publish+barrier %.0f batch %.0f\n", w*15, q[4], (m-1)/R, q[5], q[6], q[0]/R, q[1]/R, q[2]/R, q[3]/R); }
#endif
    }
}
int main(int argc, char** argv) {
    if (argc > 3) { const int nc = atoi(argv[3]); cell_probe(atoi(argv[1]), atoi(argv[2]), nc > 0 ? nc : 16); return 0; }   // argv[3] = cells per scene (16 | 64 | 128)
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
        printf("n=%d m=%d rc=%d: %.3f ms = %.0f cyc/round @2.4GHz\n", n, m, rc, ms, ms * 2.4e6 / (m - 1));
#ifdef FPS_PROFILE
        long long p[8]; hipMemcpyFromSymbol(p, HIP_SYMBOL(g_fps_prof), sizeof(p));
        for (int w = 0; w < 2; ++w)
            printf("  wave %d cycles/round: update %.0f | wave-argmax+resolve %.0f | lds write+barrier %.0f | read cands+reduce %.0f\n", w * 7,
                   p[w * 4 + 0] / (double)(m - 1), p[w * 4 + 1] / (double)(m - 1), p[w * 4 + 2] / (double)(m - 1), p[w * 4 + 3] / (double)(m - 1));
#endif
    }
#ifdef FPS_PROFILE
    long long tl[128]; hipMemcpyFromSymbol(tl, HIP_SYMBOL(g_fps_tl), sizeof(tl));
    long long t0 = tl[0];
    for (int w = 0; w < 16; ++w) for (int i = 0; i < 4; ++i) if (tl[w * 8 + i] < t0) t0 = tl[w * 8 + i];
    printf("round 100 timeline (cycles since first tick): wave: end-update end-resolve after-barrier end-reduce\n");
    for (int w = 0; w < 16; ++w) printf("  w%02d: %6lld %6lld %6lld %6lld\n", w, tl[w*8]-t0, tl[w*8+1]-t0, tl[w*8+2]-t0, tl[w*8+3]-t0);
#endif
    return 0;
}
