/*
 * gspn_hip.h -- C ABI of libgspn_hip.so: the MI355X (gfx950) implementation of the GSPN /
 * PointNet++ set-abstraction hot path.
 *
 * Drop-in boundary.  Every entry point below replaces one C launcher (or CPU loop) that the
 * reference's TensorFlow OpKernels call; the scalar and pointer order is the reference
 * launcher's, followed by one extra `void* stream` (a hipStream_t; NULL = the null stream).
 * The reference launchers go to the legacy default stream and return void
 * (tf_ops/sampling/tf_sampling_g.cu:194-211); here every call is stream-ordered and returns
 *      0                      success
 *      GSPN_ERR_ARG  (-1)     a size/attribute the reference OP_REQUIRES would reject
 *      GSPN_ERR_UNSUPPORTED   (-2)  valid input outside what this build supports
 *      > 0                    a hipError_t from the launch
 * Nothing throws across the ABI.  No entry point allocates or frees device memory: the caller
 * owns every buffer (as TF's allocate_output/allocate_temp do in the reference).  Gradient
 * entry points zero their output themselves (stream-ordered), replacing the reference's
 * cudaMemset calls (tf_sampling.cpp:174, tf_grouping.cpp:234,307, tf_nndistance_g.cu:153-154).
 * All pointers are device pointers; tensors are dense row-major float32 / int32.
 * Thread safety: the library holds no mutable global state (its tuning hooks only READ the environment).
 */
#ifndef GSPN_HIP_H
#define GSPN_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define GSPN_ERR_ARG (-1)
#define GSPN_ERR_UNSUPPORTED (-2)

/* Build facts: squared-distance contraction policy (see oracle/gspn_oracle.c header) and ABI rev.
 * GSPN_ABI_VERSION is bumped whenever an entry point is added, removed or changes its argument list or workspace layout; a binder
 * compares it with gspn_abi_version() of the library it loaded (gspn_amd/_lib.py raises on a mismatch: a stale .so fails loudly).
 *   1: round 1.   2: round 2 (gspn_fps_background removed, ~30 entry points added, finalize / workspace layouts changed).
 *   3: round 3 (gspn_sa_rel_shift, gspn_bn_apply, gspn_mlp_gemm_*, status word of the multi-CU FPS checked).
 *   4: round 3 (gspn_mlp_bwd_fused, gspn_mlp_bwd_fused_work_bytes).   5: round 3 (gspn_dense_rsum; the fused launch's pooled form).
 *   6: round 3 (gspn_fps_cells_prepass_order, gspn_bn_colsum / gspn_bn_apply_grad of tf_util's stand-alone batch norm).
 *   8: round 4 (gspn_pool32_select_groups).
 *   7: round 4 (gspn_nmdistance_grad_csr, gspn_bn_finalize_parts_pivot, gspn_mlp_bwd_fused_coef, gspn_dot, gspn_queryballpoint_ws; gspn_queryballpoint now launches a prefix scan + a continuation kernel -- same output). */
#define GSPN_ABI_VERSION 9
int gspn_dist_policy(void);
int gspn_abi_version(void);

/* ---------------- tf_ops/sampling ---------------------------------------------------- */

/* farthestpointsamplingLauncher(b,n,m,inp,temp,out)  tf_sampling.cpp:94, tf_sampling_g.cu:203-205.
 * inp (b,n,3) f32 -> out (b,m) i32.  temp: the reference's scratch, (32,n) f32 = 128*n bytes (tf_sampling.cpp:111-115).  It is
 * the workspace of the cell kernels below (n >= 8192); required for n > GSPN_FPS_RESIDENT_MAX, may be NULL below that (the plain
 * on-chip kernel then runs: same output, about half the speed at n = 32768). */
#define GSPN_FPS_RESIDENT_MAX 32768
int gspn_farthestpointsampling(int b, int n, int m, const float* inp, float* temp, int* out, void* stream);

/* Same result as gspn_farthestpointsampling, several times fewer serial rounds: FPS on a scene that the caller has sorted into
 * 16 spatial cells (csz = ceil(n/16) points each, Morton order; inside a cell by the reference tie rank (k mod 512, k)):
 *   sxyz (b,n,3) = inp gathered by perm;  perm (b,n) i32: sorted position -> original index;  inp0 (b,3) = original point 0.
 * One wave per cell; exact wave culling by bounding box and several provably-sequential picks per barrier
 * (gspn_amd/csrc/sampling.hip: fps_cell_kernel).  Requires csz <= 2048 (n <= 32768). */
int gspn_fps_cells(int b, int n, int m, int csz, const float* sxyz, const int* perm, const float* inp0, int* out, void* stream);
/* Drop-in for gspn_farthestpointsampling with identical output (n <= 32768): runs the spatial pre-pass (voxel counting sort +
 * per-cell rank sort, two small kernels) and the cell kernel on `stream`.  ws: gspn_fps_cells_ws_bytes(b,n) bytes of scratch. */
long gspn_fps_cells_ws_bytes(int b, int n);
int gspn_farthestpointsampling_cells(int b, int n, int m, const float* inp, void* ws, int* out, void* stream);
/* The two halves of gspn_farthestpointsampling_cells on the same workspace: the spatial pre-pass (counting sort into 16 cells + rank
 * sort inside each cell) and the sampling kernel proper.  Calling them one after the other on one stream equals the combined call. */
int gspn_fps_cells_prepass(int b, int n, const float* inp, void* ws, void* stream);
/* the same, also leaving the scene's points in 16^3-voxel Morton order in vorder (b, n) int32 (original indices; arbitrary order inside
 * a voxel): a finer spatial order than the 16 cells of ws, for gspn_threenn_ordered. */
int gspn_fps_cells_prepass_order(int b, int n, const float* inp, void* ws, int* vorder, void* stream);
int gspn_fps_cells_sample(int b, int n, int m, const float* inp, const void* ws, int* out, void* stream);

/* Scenes that do not fit one CU (n > 32768; tf_sampling_g.cu:137-141 is the reference's any-n path, data_prep.py:64-83 its caller
 * at n ~ 1e5, m = 30000): the same cell scheme on G workgroups (CUs) per scene, 16*G cells, candidates exchanged between the CUs
 * once per round (gspn_amd/csrc/sampling_multi.hip).  Output identical to gspn_farthestpointsampling.  G = 0 picks the
 * workgroup count measured fastest for the scene size (ceil(n/16384) up to 10, ceil(n/24576) up to 32, ceil(n/32768) beyond); the caller
 * may ask for more (<= 32: smaller cells, but every workgroup lengthens a round's exchange), never for fewer than hold the scene.  Works for any
 * n >= 1 (also below 32768, e.g. one large scene spread over several CUs).  ws: gspn_fps_multi_ws_bytes(b,n) bytes.
 * prepass + sample = the combined call, as for the single-CU cell kernel.  gspn_fps_multi_status synchronises the stream and returns
 * 0, or 1 if a bounded inter-workgroup wait expired (the workgroups of a scene were not co-resident; output invalid). */
long gspn_fps_multi_ws_bytes(int b, int n);
int gspn_fps_multi_prepass(int b, int n, int G, const float* inp, void* ws, void* stream);
int gspn_fps_multi_sample(int b, int n, int m, int G, const float* inp, void* ws, int* out, void* stream);
int gspn_farthestpointsampling_multi(int b, int n, int m, int G, const float* inp, void* ws, int* out, void* stream);
int gspn_fps_multi_status(const void* ws, int b, int n, void* stream);
/* A launch is capped at 128 co-resident workgroups (fewer than 8 scenes per launch beyond G = 16).  The sampling call zeroes `out`
 * before it launches, so after an expired wait every entry is still a valid index (0); the status word -- one int32 at byte offset
 * gspn_fps_multi_status_offset(b, n) of ws -- is then 1.  A caller MUST look at it before trusting `out`: synchronously with
 * gspn_fps_multi_status, or by copying the word asynchronously and checking it at its next synchronisation point (what
 * gspn_amd/tf_sampling.py does: a non-zero word raises GspnHipError there). */
long gspn_fps_multi_status_offset(int b, int n);

/* gatherpointLauncher(b,n,m,inp,idx,out)  tf_sampling.cpp:125, tf_sampling_g.cu:206-208 */
int gspn_gatherpoint(int b, int n, int m, const float* inp, const int* idx, float* out, void* stream);

/* scatteraddpointLauncher(b,n,m,out_g,idx,inp_g)  tf_sampling.cpp:150, tf_sampling_g.cu:209-211.
 * inp_g (b,n,3) is zeroed here first. */
int gspn_scatteraddpoint(int b, int n, int m, const float* out_g, const int* idx, float* inp_g, void* stream);

/* probsampleLauncher(b,n,m,inp_p,inp_r,temp,out)  tf_sampling.cpp:65, tf_sampling_g.cu:198-201.
 * temp: (b,n) floats (the cumulative sums). */
int gspn_probsample(int b, int n, int m, const float* inp_p, const float* inp_r, float* temp, int* out, void* stream);

/* ---------------- tf_ops/grouping ---------------------------------------------------- */

/* queryBallPointLauncher(b,n,m,radius,nsample,xyz1,xyz2,idx,pts_cnt)  tf_grouping.cpp:96,
 * tf_grouping_g.cu:186-189.  Rows without any hit are zero-filled (uninitialised in the reference). */
int gspn_queryballpoint(int b, int n, int m, float radius, int nsample, const float* xyz1, const float* xyz2,
                        int* idx, int* pts_cnt, void* stream);
/* the same with a workspace (gspn_ball_ws_bytes(b, n, m) bytes, 16-byte aligned): on sparse clouds -- a ball holds few points, the
 * reference's scan reads the whole cloud -- the queries are answered through a cell grid over the data points (all hits of the 3 x 3 x 3
 * block around the query's cell, sorted by index); dense clouds and crowded balls take the scan.  Identical output. */
long gspn_ball_ws_bytes(int b, int n, int m);
int gspn_queryballpoint_ws(int b, int n, int m, float radius, int nsample, const float* xyz1, const float* xyz2, void* ws,
                           int* idx, int* pts_cnt, void* stream);

/* Host helper (no GPU work): the squared-distance threshold T the ball-query kernel compares against,
 * i.e. the smallest float with sqrtf(T) >= radius, so that  s < T  <=>  max(sqrtf(s),1e-20f) < radius
 * (tf_grouping_g.cu:27-28) bit-exactly; 0 when radius <= 1e-20f. */
float gspn_ball_threshold(float radius);

/* selectionSortLauncher(b,n,m,k,dist,outi,out)  tf_grouping.cpp:138, tf_grouping_g.cu:190-193 */
int gspn_selectionsort(int b, int n, int m, int k, const float* dist, int* outi, float* out, void* stream);

/* knn_point(k, xyz1, xyz2) of tf_grouping.py:71-96 as ONE call, for 3-D points: the reference composes it in TensorFlow from a dense
 * (b,m,n) squared-distance tensor (:85-87), selectionSortLauncher and a slice (:88-92).  Same val (b,m,k) / idx (b,m,k), ties
 * included, without the matrix (gspn_amd/csrc/knn.hip).  k <= 32 here (GSPN_ERR_UNSUPPORTED beyond: use gspn_selectionsort);
 * k > n is rejected like the slice would be. */
int gspn_knn_point(int b, int n, int m, int k, const float* xyz1, const float* xyz2, float* val, int* idx, void* stream);

/* knn_point (tf_grouping.py:71-96) is composed on the host side exactly as the reference does:
 * dense squared-distance matrix + gspn_selectionsort + slice (see gspn_amd/tf_grouping.py). */

/* groupPointLauncher(b,n,c,m,nsample,points,idx,out)  tf_grouping.cpp:172, tf_grouping_g.cu:194-197 */
int gspn_grouppoint(int b, int n, int c, int m, int nsample, const float* points, const int* idx, float* out, void* stream);

/* groupPointGradLauncher(b,n,c,m,nsample,grad_out,idx,grad_points)  tf_grouping.cpp:203,
 * tf_grouping_g.cu:198-202.  grad_points (b,n,c) is zeroed here first. */
int gspn_grouppoint_grad(int b, int n, int c, int m, int nsample, const float* grad_out, const int* idx, float* grad_points, void* stream);

/* groupMaxpoolLauncher / groupMaxpoolGradLauncher  tf_grouping.cpp:241,277, tf_grouping_g.cu:203-210 */
int gspn_groupmaxpool(int b, int n, int c, int m, int nsample, const float* points, const int* idx, float* out, int* max_idx, void* stream);
int gspn_groupmaxpool_grad(int b, int n, int c, int m, const float* grad_out, const int* max_idx, float* grad_points, void* stream);

/* ---------------- tf_ops/3d_interpolation (CPU-only in the reference) ----------------- */

/* threenn_cpu(b,n,m,xyz1,xyz2,dist,idx)  tf_interpolate.cpp:60-103 */
int gspn_threenn(int b, int n, int m, const float* xyz1, const float* xyz2, float* dist, int* idx, void* stream);
/* the same, with the unknown points handed to the threads in the given order: order (b,n) int32, a permutation of 0..n-1 per scene
 * (e.g. the first b*n words of the workspace gspn_fps_cells_prepass filled for xyz1).  Identical output for every order; a spatially
 * coherent one makes the exact re-evaluations of a wave coincide (8 x 32768 <- 2048: 171 us in the given order, 108 us in the FPS
 * pre-pass order, 81 us in an 8^3 voxel order). */
int gspn_threenn_ordered(int b, int n, int m, const float* xyz1, const float* xyz2, const int* order, float* dist, int* idx, void* stream);
/* threeinterpolate_cpu(b,m,c,n,points,idx,weight,out)  tf_interpolate.cpp:107-127 */
int gspn_threeinterpolate(int b, int m, int c, int n, const float* points, const int* idx, const float* weight, float* out, void* stream);
/* threeinterpolate_grad_cpu(b,n,c,m,grad_out,idx,weight,grad_points)  tf_interpolate.cpp:131-153;
 * grad_points (b,m,c) is zeroed here first. */
int gspn_threeinterpolate_grad(int b, int n, int c, int m, const float* grad_out, const int* idx, const float* weight, float* grad_points, void* stream);

/* Input matrix of a feature-propagation MLP in one pass -- three_interpolate + tf.concat of utils/pointnet_util.py:161-166
 * (interpolated features FIRST, then points1), written with row pitch ld >= c2+c1 (pad columns zero):
 *   out (b*n, ld);  points2 (b,m,c2), idx/weight (b,n,3), points1 (b,n,c1) or NULL when c1 == 0.
 * gspn_fp_concat_grad: grad_out (b*n, ld) -> grad_points2 (b,m,c2) (zeroed here, then scatter-added: tf_interpolate.cpp:131-153)
 * and grad_points1 (b,n,c1); either output may be NULL. */
int gspn_fp_concat(int b, int n, int m, int c2, int c1, const float* points2, const int* idx, const float* weight,
                   const float* points1, int ld, float* out, void* stream);
int gspn_fp_concat_grad(int b, int n, int m, int c2, int c1, int ld, const float* grad_out, const int* idx, const float* weight,
                        float* grad_points2, float* grad_points1, void* stream);
/* The same gradient without atomics, in the reference's summation order (bit-identical to its sequential loop): `order` (b, 3n) holds the
 * positions p = 3*i + t of the flattened idx array sorted by idx[p] (ties in ascending p), `offsets` (b, m+1) the range of each sparse
 * point in it -- coordinate-only data a caller builds once per batch (gspn_amd/geometry.py: fp_geometry). */
int gspn_fp_concat_grad_csr(int b, int n, int m, int c2, int c1, int ld, const float* grad_out, const int* order, const int* offsets,
                            const float* weight, float* grad_points2, float* grad_points1, void* stream);
/* The same sums with lists longer than split_t entries shared out over the sixteen rows of a workgroup (ABI 9): a fixed order -- run-to-run identical bits -- that
 * DIFFERS from the reference loop's for those lists, so not for three_interpolate's gradient; for internal gathers whose order is this library's own (the transposed
 * aggregation of a pre-aggregated first layer).  On clustered clouds one sparse point is the nearest neighbour of > 1000 dense points.  split_t = 0: as above. */
int gspn_fp_concat_grad_csr_split(int b, int n, int m, int c2, int c1, int ld, const float* grad_out, const int* order, const int* offsets,
                                  const float* weight, float* grad_points2, float* grad_points1, int split_t, void* stream);

/* ---------------- tf_ops/nn_distance -------------------------------------------------- */

/* NmDistanceKernelLauncher(b,n,xyz,m,xyz2,result,result_i,result2,result2_i)  tf_nndistance.cpp:168,
 * tf_nndistance_g.cu:128-131 */
int gspn_nmdistance(int b, int n, const float* xyz, int m, const float* xyz2, float* result, int* result_i,
                    float* result2, int* result2_i, void* stream);
/* NmDistanceGradKernelLauncher(...)  tf_nndistance.cpp:208, tf_nndistance_g.cu:152-157;
 * both gradients are zeroed here first. */
int gspn_nmdistance_grad(int b, int n, const float* xyz1, int m, const float* xyz2, const float* grad_dist1, const int* idx1,
                         const float* grad_dist2, const int* idx2, float* grad_xyz1, float* grad_xyz2, void* stream);
/* The same gradient as a gather through inverse lists (gspn_inverse_lists of idx1 over cloud 2's m points: order1 (b,n), offsets1 (b,m+1);
 * of idx2 over cloud 1's n points: order2 (b,m), offsets2 (b,n+1)): terms added in the order of the reference's sequential CPU twin
 * (tf_nndistance.cpp:126-163) -- bit-identical to it, no memset, no atomics.  Any cloud size. */
int gspn_nmdistance_grad_csr(int b, int n, const float* xyz1, int m, const float* xyz2, const float* grad_dist1, const int* idx1,
                             const float* grad_dist2, const int* idx2, const int* order1, const int* offsets1, const int* order2,
                             const int* offsets2, float* grad_xyz1, float* grad_xyz2, void* stream);

/* ---------------- the four gradient launchers again, WITH a workspace (ABI 9) ----------------
 * The reference's gradient launchers are scatter-adds whose signatures have no slot for scratch memory, so the symbols above that keep those
 * signatures exactly can only scatter with hardware atomics (unordered sums, every gradient row read through an atomic).  The same launchers with
 * ONE more argument before the stream -- `void* ws`, gspn_<op>_ws_bytes(...) bytes, 16-byte aligned (an OpKernel's allocate_temp) -- build the
 * inverse lists of the index tensor in ws (gspn_inverse_lists) and GATHER: every gradient row is read once, sums are formed in a fixed order
 * (ascending position of the flattened index tensor), no atomics, no memset.  Both steps run on the caller's stream; nothing is cached.
 *   gspn_threeinterpolate_grad_ws : bit-identical to the reference's sequential loop (tf_interpolate.cpp:131-153) as compiled by g++
 *   gspn_nmdistance_grad_ws       : terms in the order of the sequential CPU twin (tf_nndistance.cpp:126-163), any cloud size
 *   gspn_grouppoint_grad_ws, gspn_scatteraddpoint_ws : one fixed order where the reference (atomicAdd, tf_grouping_g.cu:66-83,
 *       tf_sampling_g.cu:183-192) defines none.  Rows narrower than 16 floats (coordinates, colours) are FASTER through the atomic symbols;
 *       these are for callers that need run-to-run identical bits.
 * Measured against the plain symbols: profiles/r06_dropin_ws.txt; INTEGRATION.md section B shows the OpKernel side. */
long gspn_grouppoint_grad_ws_bytes(int b, int n, int c, int m, int nsample);
int gspn_grouppoint_grad_ws(int b, int n, int c, int m, int nsample, const float* grad_out, const int* idx, float* grad_points, void* ws, void* stream);
long gspn_scatteraddpoint_ws_bytes(int b, int n, int m);
int gspn_scatteraddpoint_ws(int b, int n, int m, const float* out_g, const int* idx, float* inp_g, void* ws, void* stream);
long gspn_threeinterpolate_grad_ws_bytes(int b, int n, int c, int m);
int gspn_threeinterpolate_grad_ws(int b, int n, int c, int m, const float* grad_out, const int* idx, const float* weight, float* grad_points, void* ws,
                                  void* stream);
long gspn_nmdistance_grad_ws_bytes(int b, int n, int m);
int gspn_nmdistance_grad_ws(int b, int n, const float* xyz1, int m, const float* xyz2, const float* grad_dist1, const int* idx1, const float* grad_dist2,
                            const int* idx2, float* grad_xyz1, float* grad_xyz2, void* ws, void* stream);

/* ---------------- utils/pointnet_util.py composition helpers --------------------------- */

/* (rows, c) -> (rows, ld >= c), the extra columns zero: the 16-byte feature rows gspn_mlp_fwd_gather reads */
int gspn_pad_rows(long rows, int c, int ld, const float* src, float* dst, void* stream);

/* sample_and_group's tail in one pass (pointnet_util.py:41-52): out (b,m,ns,cx+c) where the cx=3
 * xyz channels are xyz[idx]-new_xyz (:41-42) and the c feature channels are points[idx] (:46);
 * xyz_first=1 gives concat([grouped_xyz, grouped_points]) (:48), 0 gives the features-first order
 * of models/model_rpointnet.py:61.  points may be NULL (c=0).  ld_out >= 3+c is the row pitch
 * of out in floats (rows may be padded for the MLP kernels; pad columns are written as 0). */
int gspn_sa_group_concat(int b, int n, int c, int m, int nsample, const float* xyz, const float* new_xyz, const float* points,
                         const int* idx, int xyz_first, int ld_out, float* out, void* stream);
/* gradient of the above w.r.t. points: grad_points (b,n,c) += grad_out[..., feature channels];
 * zeroed here first. */
int gspn_sa_group_concat_grad(int b, int n, int c, int m, int nsample, const int* idx, int xyz_first, int ld_out,
                              const float* grad_out, float* grad_points, void* stream);
/* The same gradient as a gather (no atomics, fixed summation order): `order` (b, m*nsample) = positions of the flattened idx rows
 * sorted by data-point index (ties ascending), `offsets` (b, n+1) = each data point's range in it. */
int gspn_sa_group_concat_grad_csr(int b, int n, int c, int m, int nsample, const int* order, const int* offsets, int xyz_first, int ld_out,
                                  const float* grad_out, float* grad_points, void* stream);

/* ---------------- utils/tf_util.py conv2d 1x1 (+bias +BN +ReLU) : the shared MLP ------- */
/* The reference delegates this arithmetic to TensorFlow (tf_util.py:120-185, 515-534):
 *      y = x.W + bias ;  z = relu(gamma*(y-mean)*rsqrt(var+eps)+beta)
 * per layer over `rows` = b*m*nsample rows (NHWC, 1x1 kernel == a row-major GEMM).  Here each
 * layer is an fp32-MFMA GEMM (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate) with the
 * element-wise work fused into the operand staging (prologue) / accumulator store (epilogue), so an
 * activation tensor is written once (pre-BN) and read once by its consumer.
 *
 * gspn_mlp_fwd:  Y(rows,ldy)[:, :cout] = act(X)(rows,ldx)[:, :cin] . W(cin,cout) + bias
 *   act(X) = X                           if in_scale == NULL
 *          = relu(X*in_scale+in_shift)   otherwise (per input channel: the previous layer's BN+ReLU)
 *   stats (may be NULL; gspn_mlp_fwd_stats_bytes(rows,cout) bytes, need not be zeroed): every row-block writes its own
 *   partial column sums of Y and Y^2 (no hot-spot atomics); gspn_bn_finalize adds the partials in double.
 * Limits of this build: cin, cout <= GSPN_MLP_MAX_CHANNELS (GSPN_ERR_UNSUPPORTED beyond; the widest layer of the reference's
 * networks has 768 input channels, model_rpointnet.py:109,181), rows < 2^31.
 */
#define GSPN_MLP_MAX_CHANNELS 1024
int gspn_mlp_fwd(long rows, int cin, int cout, const float* X, int ldx, const float* in_scale, const float* in_shift,
                 const float* W, const float* bias, float* Y, int ldy, float* stats, void* stream);
/* gspn_mlp_fwd + the first half of a max-pool over groups of 32 consecutive rows (pointnet_util.py:123-124, nsample = 32): per group
 * and channel the largest raw output and its row offset, taken from the accumulators (vmax, amax: each (rows/32, cout)).  BN+ReLU is
 * increasing in y for scale >= 0, so gspn_pool32_select finishes the pool from these once scale/shift exist:
 * out = relu(scale*vmax + shift), arg = amax -- the (rows, cout) tensor is not read again, except for channels with a negative scale
 * (their group minimum is taken from Y by the select kernel, which also stores it back: on return vmax holds y at the arg row for every
 * channel -- what gspn_pool_rsum needs in backward, so that pass need not gather from Y either). */
int gspn_mlp_fwd_pool32(long rows, int cin, int cout, const float* X, int ldx, const float* in_scale, const float* in_shift,
                        const float* W, const float* bias, float* Y, int ldy, float* stats, float* vmax, int* amax, void* stream);
int gspn_pool32_select(long groups, int c, float* vmax, const int* amax, const float* Y, int ldy,
                       const float* scale, const float* shift, float* out, int* arg, void* stream);
/* The same for groups of ns = 32 * sub rows (model_rpointnet.py:68: max over 256 / 512 grouped rows): vmax / amax (groups * sub, c) as
 * gspn_mlp_fwd_pool32 left them (not modified); out, arg (row offset inside the group, 0 .. ns-1), yarg (y at the arg row) are (groups, c).
 * c a multiple of 4, 16-byte aligned arrays (GSPN_ERR_UNSUPPORTED otherwise: the caller then runs gspn_bnrelu_maxpool over Y). */
int gspn_pool32_select_groups(long groups, int sub, int c, const float* vmax, const int* amax, const float* Y, int ldy,
                              const float* scale, const float* shift, float* out, int* arg, float* yarg, void* stream);
/* bytes of the `stats` workspace for a (rows, cout) layer */
long gspn_mlp_fwd_stats_bytes(long rows, int cout);

/* BN finalize (tf.contrib.layers.batch_norm, tf_util.py:529-534): from stats = (sum, sumsq) over
 * `rows` rows produce the batch mean / biased variance (saved for backward), scale = gamma*rsqrt(var+eps),
 * shift = beta - mean*scale, and update the moving averages in place
 * (moving = moving*decay + batch*(1-decay)).  `stats` is the workspace gspn_mlp_fwd filled for the same (rows, c).  is_training==0: scale/shift come from the moving
 * statistics, mean/var are set to them and nothing is updated (stats may be NULL). */
int gspn_bn_finalize(long rows, int c, const float* stats, const float* gamma, const float* beta, float eps, float decay,
                     int is_training, float* moving_mean, float* moving_var, float* mean, float* var,
                     float* scale, float* shift, void* stream);
/* the same on nparts partial rows [nparts][2][c] of column sums from any producer (e.g. gspn_preagg_fwd) */
int gspn_bn_finalize_parts(long rows, int c, const float* stats, int nparts, const float* gamma, const float* beta, float eps, float decay,
                           int is_training, float* moving_mean, float* moving_var, float* mean, float* var, float* scale, float* shift,
                           void* stream);
/* the same with the partial sums taken about a pivot row (gspn_bn_colsum(dZ = NULL, mean = pivot)): mean = pivot + sum/rows; pivot may be NULL */
int gspn_bn_finalize_parts_pivot(long rows, int c, const float* stats, int nparts, const float* gamma, const float* beta, float eps, float decay,
                                 int is_training, float* moving_mean, float* moving_var, float* mean, float* var, float* scale, float* shift,
                                 const float* pivot, void* stream);

/* out(groups,c) = max over the ns rows of each group of relu(Y*scale+shift)   (tf.reduce_max,
 * pointnet_util.py:123-124); arg (groups,c) gets the row offset (0..ns-1) of the first maximum. */
int gspn_bnrelu_maxpool(long groups, int ns, int c, const float* Y, int ldy, const float* scale, const float* shift,
                        float* out, int* arg, void* stream);
/* out = relu(Y*scale+shift) materialised (last layer of an FP module) */
int gspn_bnrelu_apply(long rows, int c, const float* Y, int ldy, const float* scale, const float* shift, float* out, int ldo, void* stream);

/* ---- stand-alone batch normalisation over the rows of a (rows, c) matrix (gspn_amd/csrc/batchnorm.hip): the device side of
 * tf_util.batch_norm_for_fc / _conv1d / _conv2d (utils/tf_util.py:515-580: tf.contrib.layers.batch_norm over all leading axes).
 *   forward : gspn_bn_colsum(dZ = NULL) -> gspn_bn_finalize_parts -> gspn_bn_apply(relu = 0)
 *   backward: gspn_bn_colsum(dZ)        -> gspn_mlp_bwd_coef      -> gspn_bn_backward_apply
 * gspn_bn_colsum leaves *nparts_out partial rows [2][c] in part (gspn_bn_colsum_part_floats(rows, c) floats): column sums of
 * (x, x^2) when dZ is NULL, of (dz, dz * xhat) with xhat = (x - mean) * rsqrt(var + eps) otherwise. */
long gspn_bn_colsum_part_floats(long rows, int c);
int gspn_bn_colsum(long rows, int c, const float* X, int ldx, const float* dZ, int ldz, const float* mean, const float* var, float eps,
                   float* part, int* nparts_out, void* stream);
/* out = x*scale + shift (two roundings: tf.nn.batch_normalization, tf_util.py:511), through a ReLU when relu != 0 */
int gspn_bn_apply(long rows, int c, const float* X, int ldx, const float* scale, const float* shift, int relu, float* out, int ldo, void* stream);
/* dX = cA*dZ + cB*X + cC per column (training mode: the coefficients of gspn_mlp_bwd_coef; inference: cA = scale, cB = cC = 0) */
int gspn_bn_backward_apply(long rows, int c, const float* dZ, int ldz, const float* X, int ldx, const float* cA, const float* cB, const float* cC,
                           float* dX, int lddx, void* stream);

/* ---- backward of one layer -------------------------------------------------------------
 * With z = relu(s*y+t) and upstream gradient dz (dense, or the scatter of a pooled gradient to its
 * arg-max rows), the gradient w.r.t. the pre-BN output is
 *      dyh = dz * [s*y+t > 0]
 *      dY  = cA*dyh + cB*y + cC                     (per output channel coefficients)
 * which covers BN-training (cA=gamma*rstd, cB/cC from the two batch reductions), BN-inference
 * (cA=scale, cB=cC=0) and no-BN (cA=1).  dY is never materialised. */
typedef struct gspn_dy_args {
    const float* Y;        /* (rows, ldy) pre-BN output saved by the forward pass */
    int ldy;
    const float* dZ;       /* dense upstream gradient (rows, ldz), or NULL when pooled */
    int ldz;
    const float* dPool;    /* pooled upstream gradient (rows/ns, c), or NULL */
    const int* pool_arg;   /* (rows/ns, c) arg-max row offsets from gspn_bnrelu_maxpool */
    int ns;
    const float* scale;    /* c : forward scale/shift (relu mask) */
    const float* shift;
    const float* cA;       /* c each: coefficients written by gspn_mlp_bwd_wgrad (read by gspn_mlp_bwd_data only) */
    const float* cB;
    const float* cC;
} gspn_dy_args;

/* Pass A -- ONE read of (X, Y, dz):  the BN reductions r0 = sum(dyh), r1 = sum(dyh*xhat) and the raw
 * weight-gradient products G1 = act(X)^T.dyh, Gx = act(X)^T.xhat, g3 = act(X)^T.1 are accumulated into
 * per-row-chunk partials in `work` (gspn_mlp_bwd_work_bytes(rows,cin,cout) bytes; no hot-spot atomics, deterministic
 * second-stage sum in double), then finalised on the device into
 *   dW(cin,cout) = cA (.) (G1 - r0/R g3 1^T - r1/R (.) Gx)            (training-mode BN)
 *                = cA (.) G1                                           (BN on moving statistics, or no BN)
 *   cA=gamma*rstd (1 without BN), cB=-gamma*rstd^2*r1/R, cC=-gamma*rstd*(r0/R - mean*rstd*r1/R)   (0 outside training)
 *   dgamma=r1, dbeta=r0 (under BN), dbias=sum(dY).
 * mean/var are the statistics the forward pass normalised with (batch, or moving when !is_training).
 * Any of cA..dbias may be NULL.  dW may be NULL too: the final sum over the partial tiles is then left to
 * gspn_mlp_bwd_dw (same rows/cin/cout/a/X/ldx, same `work`), which nothing downstream of the layer waits for -- a caller can
 * put it on another stream, ordered after this call, and overlap it with the next layer's kernels. */
int gspn_mlp_bwd_wgrad(long rows, int cin, int cout, const gspn_dy_args* a, const float* X, int ldx,
                       const float* in_scale, const float* in_shift, const float* mean, const float* var, const float* gamma,
                       float eps, int use_bn, int is_training, float* work, float* cA, float* cB, float* cC,
                       float* dgamma, float* dbeta, float* dbias, float* dW, void* stream);
long gspn_mlp_bwd_work_bytes(long rows, int cin, int cout);
int gspn_mlp_bwd_dw(long rows, int cin, int cout, const gspn_dy_args* a, const float* X, int ldx, const float* var, const float* gamma,
                    float eps, int use_bn, int is_training, const float* work, float* dW, void* stream);
/* Pass B -- dX(rows,ldx)[:, :cin] = dY . W^T with dY = cA*dyh + cB*y + cC rebuilt on the fly */
int gspn_mlp_bwd_data(long rows, int cin, int cout, const gspn_dy_args* a, const float* W, float* dX, int ldx, void* stream);
/* ... restricted to columns [col0, col0+ncols) of dX (the other columns are left untouched): for inputs whose leading or trailing
 * channels need no gradient -- the xyz columns of a set-abstraction input, the raw colours of the last FP level. */
int gspn_mlp_bwd_data_cols(long rows, int cin, int cout, const gspn_dy_args* a, const float* W, int col0, int ncols, float* dX, int ldx,
                           void* stream);
/* gspn_mlp_bwd_data_cols + gspn_mlp_bwd_dw in one launch: the dW reduction that gspn_mlp_bwd_wgrad(..., dW = NULL) left undone runs in
 * spare workgroups of pass B instead of a kernel of its own (X / ldx_in / var / gamma / eps / use_bn / is_training / work as in that
 * call).  dX is bit-identical to gspn_mlp_bwd_data_cols'; dW is the same sum taken over 16 instead of 64 slot slices per output. */
int gspn_mlp_bwd_data_dw(long rows, int cin, int cout, const gspn_dy_args* a, const float* W, int col0, int ncols, float* dX, int ldx,
                         const float* X, int ldx_in, const float* var, const float* gamma, float eps, int use_bn, int is_training,
                         const float* work, float* dW, void* stream);
/* the same with a second, plain reduction riding along: dW2 (cin2, cout) = the sum of nslots2 partial tiles at part2 ([slot][2][cin2*cout],
 * first half) -- the side-column weight gradient gspn_preagg_bwd_dy leaves behind */
int gspn_mlp_bwd_data_dw2(long rows, int cin, int cout, const gspn_dy_args* a, const float* W, int col0, int ncols, float* dX, int ldx,
                          const float* X, int ldx_in, const float* var, const float* gamma, float eps, int use_bn, int is_training,
                          const float* work, float* dW, const float* part2, int cin2, int nslots2, float* dW2, void* stream);
/* Pass B of a POOLED TOP layer (pool groups of 32 rows) as a streaming GEMM on the layer's input: with dY = cA*dyh + cB*y + cC and
 * y = xhat.W + b,   dX = xhat.(W diag(cB) W^T) + S.W^T + const,   S = cA*dyh (one non-zero per pool group and channel) -- the layer's own
 * (rows, cout) output is not read.  a: the layer's gspn_dy_args (dPool, pool_arg, ns = 32, cA/cB/cC); pooled: the (rows/32, cout) pooled
 * output of the forward pass; scratch: gspn_pooltop_scratch_floats(rows, cin, cout) floats.  The layer's dW reduction rides in the small
 * launch that prepares the operands; the previous layer's BN reductions come out of the epilogue (as gspn_mlp_bwd_data_ex: both required).
 * GSPN_ERR_UNSUPPORTED outside cin <= 64, cout <= 128 (multiples of 4), ns = 32, 16-byte aligned rows: call gspn_mlp_bwd_data_ex then. */
long gspn_pooltop_scratch_floats(long rows, int cin, int cout);
int gspn_mlp_bwd_data_pooltop(long rows, int cin, int cout, const gspn_dy_args* a, const float* W, const float* bias, const float* pooled,
                              float* scratch, float* dX, int ldx,
                              const float* X, int ldx_in, const float* var, const float* gamma, float eps, int use_bn, int is_training,
                              const float* work, float* dW,
                              const float* Yp, int ldyp, const float* scale_p, const float* shift_p, const float* mean_p, const float* var_p,
                              float eps_p, float* part, int* nparts_out, void* stream);

/* ---- early coefficients: pass A as ONE GEMM ------------------------------------------------------------------------------------
 * Training-mode BN's backward needs r0 = sum(dyh) and r1 = sum(dyh*xhat) over all rows before dY exists; gspn_mlp_bwd_wgrad side-steps
 * that with a second product (Gx) inside the GEMM -- twice the matrix work.  When the two sums are taken BEFORE pass A, the coefficients
 * cA/cB/cC are final and gspn_mlp_bwd_wgrad_known runs one GEMM dW = act(X)^T . dY (g: optional gspn_gather_args of a fused first layer,
 * defined below; dW may be NULL, the sum over partial tiles then rides in gspn_mlp_bwd_data_ex / _dw called with use_bn = 0):
 *   - top layer of a pooled stack: gspn_pool_rsum takes the sums from (dPool, pool_arg, Y) -- (groups x c) work; ldy == 0 says Y is
 *     already the (groups, c) tensor of y at the arg row (vmax after gspn_pool32_select): no gather from the (rows, c) output;
 *   - any other layer l: its dz is the dX of layer l+1's pass B, whose epilogue takes them (gspn_mlp_bwd_data_ex, Yp = Y of layer l);
 *   - gspn_mlp_bwd_coef(rows, c, nparts, part, ...) sums the partial rows [nparts][2][c] in double and writes cA, cB, cC, dgamma, dbeta,
 *     dbias (the same formulas as gspn_mlp_bwd_wgrad's).  part: gspn_rsum_part_floats(rows, c) floats. */
struct gspn_gather_args;
long gspn_rsum_part_floats(long rows, int c);
int gspn_pool_rsum(long groups, int ns, int c, const float* dPool, const int* arg, const float* Y, int ldy, const float* scale,
                   const float* shift, const float* mean, const float* var, float eps, float* part, int* nparts_out, void* stream);
/* the same sums for the top layer of a stack with a DENSE upstream gradient dZ (rows, ldz): one streaming pass over (dZ, Y).
 * c a multiple of 4, c <= 1024, 16-byte aligned pitches (GSPN_ERR_UNSUPPORTED otherwise: use gspn_mlp_bwd_wgrad's two-product form). */
int gspn_dense_rsum(long rows, int c, const float* dZ, int ldz, const float* Y, int ldy, const float* scale, const float* shift,
                    const float* mean, const float* var, float eps, float* part, int* nparts_out, void* stream);
int gspn_mlp_bwd_coef(long rows, int c, int nparts, const float* part, const float* mean, const float* var, const float* gamma, float eps,
                      float* cA, float* cB, float* cC, float* dgamma, float* dbeta, float* dbias, void* stream);
int gspn_mlp_bwd_wgrad_known(long rows, int cin, int cout, const gspn_dy_args* a, const float* X, int ldx, const float* in_scale,
                             const float* in_shift, const struct gspn_gather_args* g, float* work, float* dW, void* stream);
int gspn_mlp_bwd_data_ex(long rows, int cin, int cout, const gspn_dy_args* a, const float* W, int col0, int ncols, float* dX, int ldx,
                         const float* X, int ldx_in, const float* var, const float* gamma, float eps, int use_bn, int is_training,
                         const float* work, float* dW,
                         const float* Yp, int ldyp, const float* scale_p, const float* shift_p, const float* mean_p, const float* var_p,
                         float eps_p, float* part, int* nparts_out, void* stream);
/* Pass A and pass B of one layer in ONE launch (r03).  With known coefficients (a->cA/cB/cC final) and either a dense upstream gradient
 * (a->dZ) or the gradient of a max-pool over groups of 32 rows (a->dZ NULL, a->dPool / a->pool_arg, a->ns == 32),
 *     dW (cin, cout)        = relu(Xp*in_scale + in_shift)^T . dY
 *     dX (rows, ldx >= cin) = dY . W^T
 * come from the same staged dY tile.  Xp (rows, ldxp) is the previous layer's raw output, in_scale / in_shift its forward scale / shift.
 * part (optional, with that layer's mean_p / var_p): its BN reductions, exactly what gspn_mlp_bwd_data_ex leaves (*nparts_out rows for
 * gspn_mlp_bwd_coef).  work: the gspn_mlp_bwd_work_bytes(rows, cin, cout) buffer.
 * GSPN_ERR_UNSUPPORTED outside cin in {32, 64}, cout in {32, 64, 128}, rows >= 65536 a multiple of 128, 16-byte aligned pitches
 * (GSPN_BWD_FUSED=0: always): run gspn_mlp_bwd_wgrad_known + gspn_mlp_bwd_data_ex then. */
long gspn_mlp_bwd_fused_work_bytes(long rows, int cin, int cout);
int gspn_mlp_bwd_fused(long rows, int cin, int cout, const gspn_dy_args* a, const float* W, const float* Xp, int ldxp, const float* in_scale,
                       const float* in_shift, float* dX, int ldx, float* work, float* dW, const float* mean_p, const float* var_p,
                       float eps_p, float* part, int* nparts_out, void* stream);
/* gspn_mlp_bwd_fused + gspn_mlp_bwd_coef of the PREVIOUS layer (cin channels; gamma_p and the outputs as gspn_mlp_bwd_coef takes them) with this
 * layer's dW reduction riding in the coefficient launch: two launches instead of three (r04).  part / nparts_out / mean_p / var_p required. */
int gspn_mlp_bwd_fused_coef(long rows, int cin, int cout, const gspn_dy_args* a, const float* W, const float* Xp, int ldxp, const float* in_scale,
                            const float* in_shift, float* dX, int ldx, float* work, float* dW, const float* mean_p, const float* var_p,
                            float eps_p, float* part, int* nparts_out, const float* gamma_p, float* cA_p, float* cB_p, float* cC_p,
                            float* dgamma_p, float* dbeta_p, float* dbias_p, void* stream);

/* Pre-aggregated first layer of an SA / FP module (gspn_amd/csrc/mlp.hip, "Pre-aggregated first layer"): the layer is linear, so its
 * feature part is multiplied on the SOURCE points (F = feat.W_feat, a small GEMM through gspn_mlp_fwd) and the grouped / interpolated
 * rows are then formed from F:   Y[r] = sum_t w[r,t] * F[idx[r,t]] + side[r,:side_n] . Wside + bias,   T = 1 (grouping, w = NULL) or 3
 * (3-NN interpolation).  idx are global source rows, or scene-local ones when per_scene_rows > 0 (output rows / source rows per scene);
 * with global idx (per_scene_rows == 0) per_scene_src is the total number of source rows of F (0: not given, taken as <= rows).
 * stats: the column sums of Y, gspn_preagg_fwd_parts(rows, cout) partial rows for gspn_bn_finalize_parts, or NULL.
 * cout must be 4 * 2^k (gspn_preagg_ok); F, Y 16-byte aligned.  Same result as the GEMM over materialised rows up to fp32 rounding
 * (a different order of additions). */
int gspn_preagg_ok(int cout);
long gspn_preagg_fwd_parts(long rows, int cout);     /* partial rows [parts][2][cout] gspn_preagg_fwd writes into stats */
int gspn_preagg_fwd(long rows, int cout, int T, const float* F, const int* idx, const float* w, int per_scene_rows, int per_scene_src,
                    const float* side, int side_ld, int side_n, const float* Wside, const float* bias, float* Y, float* stats, void* stream);
/* backward half that touches the (rows, cout) tensors: dY = cA*relu'(y*scale+shift)*dz + cB*y + cC written to dY (rows, cout), and
 * dWside (side_n, cout) = side^T . dY (per-workgroup partials in part, gspn_preagg_part_floats(cout, side_n) floats, summed in double in a
 * fixed order).  The rest of the layer's backward runs on source rows: G = transpose-gather of dY (gspn_sa_group_concat_grad_csr /
 * gspn_fp_concat_grad_csr on dY), dW_feat = feat^T . G, d(feat) = G . W_feat^T (gspn_mlp_bwd_wgrad / gspn_mlp_bwd_data, no BN). */
long gspn_preagg_part_floats(int cout, int side_n);
int gspn_preagg_bwd_dy(long rows, int cout, const gspn_dy_args* a, const float* side, int side_ld, int side_n, float* dY, float* part,
                       float* dWside, int* nslots_out, void* stream);
/* (dWside == NULL: the partial tiles stay at part, *nslots_out of them, for gspn_mlp_bwd_data_dw2 to sum in a launch that exists anyway) */

/* ---- fused set-abstraction front end: sample_and_group's concat (pointnet_util.py:36-52) + the first conv2d (:109-113)
 * without the grouped (b, npoint, nsample, 3+c) tensor.  gspn_sa_rel writes, per grouped row r = ((i*m + j)*ns + k), its centred
 * coordinates rel[r] = (xyz[i, idx[r]] - new_xyz[i, j], 0) and its source row gidx[r] = i*n + idx[r] (20 bytes per row).  The first
 * layer then reads VIRTUAL input rows [ feat[gidx[r]][0..ldf) | rel[r][0..4) ] -- the LDS-DMA of the streaming kernels takes each quad
 * from where it lives -- and W's rows are picked to match: xyz_first = 1 for W = [xyz(3); features(c)] (sample_and_group, :48),
 * 0 for W = [features(c); xyz(3)] (multi_encoding_net, model_rpointnet.py:61).
 *   feat: (b*n, ldf) rows, ldf % 4 == 0, ldf >= c (columns >= c are ignored), 16-byte aligned.
 * gspn_mlp_fwd_gather / gspn_mlp_bwd_wgrad_gather are gspn_mlp_fwd / gspn_mlp_bwd_wgrad for such a first layer (no input activation);
 * they return GSPN_ERR_UNSUPPORTED for shapes the streaming kernels do not take (the caller then materialises the rows with
 * gspn_sa_group_concat).  Workspace of the backward call: gspn_mlp_bwd_work_bytes(rows, gspn_mlp_gather_cin(g), cout).
 * Pass B of that layer is the ordinary gspn_mlp_bwd_data_cols (it does not read the layer's input). */
typedef struct gspn_gather_args {
    const float* feat;
    int ldf;
    int c;
    const int* gidx;
    const float* rel;
    int xyz_first;
} gspn_gather_args;
int gspn_sa_rel(int b, int n, int m, int ns, const float* xyz, const float* new_xyz, const int* idx, float* rel, int* gidx, void* stream);
/* the same with the per-seed shift of multi_encoding_net (model_rpointnet.py:56-57, called with a stop_gradient shift at :377):
 * rel[r] = ((xyz[i, idx[r]] - new_xyz[i, j]) - shift[i, j], 0); shift (b, m, 3) or NULL (= gspn_sa_rel) */
int gspn_sa_rel_shift(int b, int n, int m, int ns, const float* xyz, const float* new_xyz, const float* shift, const int* idx,
                      float* rel, int* gidx, void* stream);
int gspn_mlp_gather_cin(const gspn_gather_args* g);
int gspn_mlp_fwd_gather(long rows, const gspn_gather_args* g, int cout, const float* W, const float* bias, float* Y, int ldy,
                        float* stats, void* stream);
int gspn_mlp_bwd_wgrad_gather(long rows, const gspn_gather_args* g, int cout, const gspn_dy_args* a, const float* mean, const float* var,
                              const float* gamma, float eps, int use_bn, int is_training, float* work, float* cA, float* cB, float* cC,
                              float* dgamma, float* dbeta, float* dbias, float* dW, void* stream);

/* Inverse lists (CSR) of an index tensor -- what the gather-form gradients walk.  idx (b,L) int32 with values in [0,n):
 * order (b,L) = positions 0..L-1 grouped by value, ascending inside a group; offsets (b,n+1) = start of every group (offsets[n] = L).
 * Same result as a stable sort of idx + searchsorted.  work: gspn_inverse_lists_work_ints(b,L,n) ints.  Positions with a value
 * outside [0,n) are dropped.
 * (Extension: the reference scatters with atomicAdd -- tf_grouping_g.cu:78-86, tf_interpolate.cpp:119-143 -- and needs no such lists.) */
long gspn_inverse_lists_work_ints(int b, int L, int n);
int gspn_inverse_lists(int b, int L, int n, const int* idx, int* work, int* order, int* offsets, void* stream);

/* utils/pointnet_util.py:157-160 in one kernel: dist (total,3) squared distances of three_nn -> weight (total,3):
 * d = max(d, 1e-10); weight_k = (1/d_k) / ((1/d_0 + 1/d_1) + 1/d_2) */
int gspn_three_nn_weights(long total, const float* dist, float* weight, void* stream);

/* n device-to-device copies (src[i] -> dst[i], bytes[i] bytes; the three arrays live on the HOST) in one launch per 40 segments:
 * refills the persistent buffers a captured step reads (extension: scheduling helper, no reference counterpart). */
int gspn_multi_copy(int n, const void* const* src, void* const* dst, const long* bytes, void* stream);

/* Adam (torch.optim.Adam's rule; the reference trains with tf.train.AdamOptimizer) over one flat fp32 buffer of n parameters:
 * p, g (gradients), m, v (moments) all flat; g is multiplied by grad_scale first (1/world after a SUM all-reduce); step >= 1 = number of
 * this update.  One launch for the whole model. */
int gspn_adam_flat(long n, float* p, const float* g, float* m, float* v, float lr, float b1, float b2, float eps, float weight_decay,
                   float grad_scale, long step, void* stream);

/* The same update with the step number kept on the DEVICE: state = two 8-byte words, zero-initialised once by the caller (state[0] = updates done so far,
 * state[1] = scratch).  No argument changes from step to step, so the launch can be a node of a captured hipGraph (ABI 9). */
int gspn_adam_flat_dev(long n, float* p, const float* g, float* m, float* v, float lr, float b1, float b2, float eps, float weight_decay,
                       float grad_scale, unsigned long long* state, void* stream);

/* out[0] = <a, b> over n floats, deterministic (1024 per-workgroup partials added in index order, in double).  work: gspn_dot_work_floats() floats. */
long gspn_dot_work_floats(void);
int gspn_dot(long n, const float* a, const float* b, float* work, float* out, void* stream);

int gspn_fill_zero(void* ptr, long bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
