// Drop-in gradient launchers WITH a caller-provided workspace (ABI 9).
//
// The reference's gradient launchers -- groupPointGradLauncher (tf_grouping.cpp:203, tf_grouping_g.cu:66-83,198-202), scatteraddpointLauncher
// (tf_sampling.cpp:150, tf_sampling_g.cu:183-192), threeinterpolate_grad_cpu (tf_interpolate.cpp:131-153), NmDistanceGradKernelLauncher
// (tf_nndistance.cpp:208, tf_nndistance_g.cu:132-157) -- are scatter-adds; their signatures have no slot for scratch memory, so the symbols that
// keep those signatures exactly (gspn_grouppoint_grad, ...) can only scatter with atomics.  The gather form this library's own op wrappers use
// (gspn_amd/invlists.py: inverse lists of the index tensor, then one pass that reads every gradient row once and adds in a FIXED order) is 4-11x
// faster on wide rows and deterministic, but needs ~ (L + n) ints of scratch per scene.  These entry points are the same launchers with ONE more
// argument before the stream -- `void* ws` of gspn_<op>_ws_bytes(...) bytes, which an OpKernel gets from allocate_temp -- and do both steps on the
// caller's stream: gspn_inverse_lists into ws, then the gather.  Nothing is cached, nothing outlives the call; the index tensor is read only.
//
// Summation order = ascending position in the flattened index tensor: for three_interpolate exactly the order of the reference's sequential loop
// (bit-identical to tf_interpolate.cpp:131-153 as compiled); for nn_distance the order of the sequential CPU twin (tf_nndistance.cpp:126-163); for
// group_point / gather_point the reference defines no order (atomicAdd) and this is one valid order, the same on every run.
#include <hip/hip_runtime.h>
#include "../../include/gspn_hip.h"

namespace {
inline long align64(long ints) { return (ints + 15) / 16 * 16; }      // 64-byte granules between the three regions
struct Lists { int* order; int* offsets; int* work; };
inline long lists_ints(int b, long L, int n) { return align64((long)b * L) + align64((long)b * (n + 1)) + align64(gspn_inverse_lists_work_ints(b, (int)L, n)); }
inline Lists carve(int* base, int b, long L, int n) {
    Lists l;
    l.order = base;
    l.offsets = l.order + align64((long)b * L);
    l.work = l.offsets + align64((long)b * (n + 1));
    return l;
}
inline bool fits_int(long v) { return v >= 0 && v < (1L << 31); }
}

extern "C" {

long gspn_grouppoint_grad_ws_bytes(int b, int n, int c, int m, int nsample) {
    if (b < 0 || n <= 0 || m < 0 || nsample < 0 || c < 0) return -1;
    return 4 * lists_ints(b, (long)m * nsample, n);
}

int gspn_grouppoint_grad_ws(int b, int n, int c, int m, int nsample, const float* grad_out, const int* idx, float* grad_points, void* ws, void* stream) {
    if (b < 0 || n <= 0 || c < 0 || m < 0 || nsample < 0 || !fits_int((long)m * nsample)) return -1;
    if (b == 0 || c == 0) return 0;
    if ((long)m * nsample == 0) return gspn_fill_zero(grad_points, (long)b * n * c * 4, stream);
    if (!grad_out || !idx || !grad_points || !ws) return -1;
    const int L = m * nsample;
    Lists l = carve((int*)ws, b, L, n);
    int rc = gspn_inverse_lists(b, L, n, idx, l.work, l.order, l.offsets, stream);
    if (rc) return rc;
    return gspn_sa_group_concat_grad_csr(b, n, c, m, nsample, l.order, l.offsets, 0, c, grad_out, grad_points, stream);
}

long gspn_scatteraddpoint_ws_bytes(int b, int n, int m) {
    if (b < 0 || n <= 0 || m < 0) return -1;
    return 4 * lists_ints(b, m, n);
}

int gspn_scatteraddpoint_ws(int b, int n, int m, const float* out_g, const int* idx, float* inp_g, void* ws, void* stream) {
    if (b < 0 || n <= 0 || m < 0) return -1;
    if (b == 0) return 0;
    if (m == 0) return gspn_fill_zero(inp_g, (long)b * n * 3 * 4, stream);
    if (!out_g || !idx || !inp_g || !ws) return -1;
    Lists l = carve((int*)ws, b, m, n);
    int rc = gspn_inverse_lists(b, m, n, idx, l.work, l.order, l.offsets, stream);
    if (rc) return rc;
    return gspn_sa_group_concat_grad_csr(b, n, 3, m, 1, l.order, l.offsets, 0, 3, out_g, inp_g, stream);
}

long gspn_threeinterpolate_grad_ws_bytes(int b, int n, int c, int m) {
    if (b < 0 || n < 0 || m <= 0 || c < 0) return -1;
    return 4 * lists_ints(b, 3L * n, m);
}

int gspn_threeinterpolate_grad_ws(int b, int n, int c, int m, const float* grad_out, const int* idx, const float* weight, float* grad_points, void* ws,
                                  void* stream) {
    if (b < 0 || n < 0 || c < 0 || m <= 0 || !fits_int(3L * n)) return -1;
    if (b == 0 || c == 0) return 0;
    if (n == 0) return gspn_fill_zero(grad_points, (long)b * m * c * 4, stream);
    if (!grad_out || !idx || !weight || !grad_points || !ws) return -1;
    Lists l = carve((int*)ws, b, 3L * n, m);
    int rc = gspn_inverse_lists(b, 3 * n, m, idx, l.work, l.order, l.offsets, stream);
    if (rc) return rc;
    return gspn_fp_concat_grad_csr(b, n, m, c, 0, c, grad_out, l.order, l.offsets, weight, grad_points, nullptr, stream);
}

long gspn_nmdistance_grad_ws_bytes(int b, int n, int m) {
    if (b < 0 || n <= 0 || m <= 0) return -1;
    return 4 * (lists_ints(b, n, m) + lists_ints(b, m, n));
}

int gspn_nmdistance_grad_ws(int b, int n, const float* xyz1, int m, const float* xyz2, const float* grad_dist1, const int* idx1, const float* grad_dist2,
                            const int* idx2, float* grad_xyz1, float* grad_xyz2, void* ws, void* stream) {
    if (b < 0 || n <= 0 || m <= 0) return -1;
    if (b == 0) return 0;
    if (!xyz1 || !xyz2 || !grad_dist1 || !idx1 || !grad_dist2 || !idx2 || !grad_xyz1 || !grad_xyz2 || !ws) return -1;
    Lists l1 = carve((int*)ws, b, n, m);                                       // idx1 (b,n): values in [0,m)
    Lists l2 = carve((int*)ws + lists_ints(b, n, m), b, m, n);                  // idx2 (b,m): values in [0,n)
    int rc = gspn_inverse_lists(b, n, m, idx1, l1.work, l1.order, l1.offsets, stream);
    if (rc) return rc;
    rc = gspn_inverse_lists(b, m, n, idx2, l2.work, l2.order, l2.offsets, stream);
    if (rc) return rc;
    return gspn_nmdistance_grad_csr(b, n, xyz1, m, xyz2, grad_dist1, idx1, grad_dist2, idx2, l1.order, l1.offsets, l2.order, l2.offsets, grad_xyz1, grad_xyz2,
                                    stream);
}

}  // extern "C"
