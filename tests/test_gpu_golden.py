"""The HIP path against the COMMITTED vectors of tests/golden/*.npz, directly -- until r05 only the CPU suite read them
(oracle -> golden) and the GPU suite compared HIP -> oracle: transitive, but no GPU test touched the stored bytes.

 * interp_ref_demo.npz / interp_ref_optest.npz: inputs and outputs of the REFERENCE'S OWN compiled code (tf_ops/3d_interpolation/interpolate.cpp
   built by g++ -O2, tests/make_golden.py) on the shapes of tf_interpolate.py:39-48 and tf_interpolate_op_test.py:11-16 -- three_interpolate
   bit-exact; its gradient bit-exact through the op API (ordered sums) and 1e-5 through the drop-in atomic symbol.
 * oracle_c1_fps_ball.npz: BASELINE configs[0] (1 x 4096 -> 512, r 0.2, ns 32); oracle_dup_fps_ball.npz: duplicate-heavy clouds where the
   (k mod 512, k) tie rule decides; oracle_nn.npz: nn_distance and 3-NN on the demo seeds -- indices and floats bit-exact.
 * threenn_ref.npz / nnsearch_ref.npz (r06): vectors of the REFERENCE'S OWN compiled `threenn_cpu` (tf_interpolate.cpp:60-103) and `nnsearch`
   (tf_nndistance.cpp:21-43), cut out of the reference at build time by oracle/Makefile: slices.  three_nn: indices and squared distances bit-exact
   through all three kernels (wave / cell grid / tiled); nn_distance: the product (CUDA-derived contraction, policy 2) picks the same neighbours and
   its distances are within one rounding; the policy-0 (unfused, = what g++ -O2 emits for the CPU twin) variant library is bit-exact."""
import ctypes
import glob
import os

import numpy as np
import pytest
import torch

from tests import data as D

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_every_golden_file_is_consumed_here():
    names = sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLD, "*.npz")))
    assert names == ["interp_ref_demo.npz", "interp_ref_optest.npz", "nnsearch_ref.npz", "oracle_c1_fps_ball.npz", "oracle_dup_fps_ball.npz", "oracle_nn.npz",
                     "threenn_ref.npz"], names


@pytest.mark.parametrize("name", ["interp_ref_demo.npz", "interp_ref_optest.npz"])
def test_three_interpolate_equals_the_reference_binary_vectors(name):
    from gspn_amd import _lib as L
    from gspn_amd.tf_interpolate import three_interpolate
    g = np.load(os.path.join(GOLD, name))
    pts = dev(g["points"]).requires_grad_(True)
    idx, w = dev(g["idx"]), dev(g["weight"])
    out = three_interpolate(pts, idx, w)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), g["out"])
    out.backward(dev(g["grad_out"]))
    np.testing.assert_array_equal(pts.grad.cpu().numpy(), g["grad_points"])        # op API: sums in the reference loop's own order
    if "weight2" in g.files:
        np.testing.assert_array_equal(three_interpolate(pts.detach(), idx, dev(g["weight2"])).cpu().numpy(), g["out2"])
    # the drop-in symbols with the reference's argument order (tf_interpolate.cpp:107,131)
    b, m, c = g["points"].shape
    n = g["idx"].shape[1]
    lib = L.lib()
    o2 = torch.empty((b, n, c), device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    assert lib.gspn_threeinterpolate(b, m, c, n, P(pts.detach()), P(idx), P(w), P(o2), st) == 0
    gp = torch.full((b, m, c), float("nan"), device="cuda")
    go = dev(g["grad_out"])
    assert lib.gspn_threeinterpolate_grad(b, n, c, m, P(go), P(idx), P(w), P(gp), st) == 0
    torch.cuda.synchronize()
    np.testing.assert_array_equal(o2.cpu().numpy(), g["out"])
    np.testing.assert_allclose(gp.cpu().numpy(), g["grad_points"], rtol=1e-5, atol=1e-5)      # atomics: unordered sums of up to ~12 terms of O(1)


def test_fps_and_ball_query_equal_the_config0_vectors():
    from gspn_amd.tf_grouping import query_ball_point
    from gspn_amd.tf_sampling import farthest_point_sample, gather_point
    g = np.load(os.path.join(GOLD, "oracle_c1_fps_ball.npz"))
    xyz = dev(D.batch("U", 1, 4096))
    fps = farthest_point_sample(512, xyz)
    np.testing.assert_array_equal(fps.cpu().numpy(), g["fps"])
    idx, cnt = query_ball_point(0.2, 32, xyz, gather_point(xyz, fps))
    np.testing.assert_array_equal(idx.cpu().numpy(), g["idx"])
    np.testing.assert_array_equal(cnt.cpu().numpy(), g["cnt"])


def test_fps_and_ball_query_equal_the_duplicate_cloud_vectors():
    from gspn_amd.tf_grouping import query_ball_point
    from gspn_amd.tf_sampling import farthest_point_sample, gather_point
    g = np.load(os.path.join(GOLD, "oracle_dup_fps_ball.npz"))
    xd = dev(D.batch("D", 2, 3000, 40))
    fd = farthest_point_sample(700, xd)
    np.testing.assert_array_equal(fd.cpu().numpy(), g["fps"])
    bi, bc = query_ball_point(0.15, 16, xd, gather_point(xd, fd))
    np.testing.assert_array_equal(bi.cpu().numpy(), g["idx"])
    np.testing.assert_array_equal(bc.cpu().numpy(), g["cnt"])


def test_nn_distance_and_three_nn_equal_the_vectors():
    from gspn_amd.tf_interpolate import three_nn
    from gspn_amd.tf_nndistance import nn_distance
    h = np.load(os.path.join(GOLD, "oracle_nn.npz"))
    a, b = dev(h["a"]), dev(h["b"])
    d1, i1, d2, i2 = nn_distance(a, b)
    for got, key in ((d1, "d1"), (i1, "i1"), (d2, "d2"), (i2, "i2")):
        np.testing.assert_array_equal(got.cpu().numpy(), h[key])
    # the reference's CPU twin (unfused arithmetic, tf_nndistance.cpp:21-43) picks the same neighbours on these clouds
    np.testing.assert_array_equal(i1.cpu().numpy(), h["j1"])
    np.testing.assert_array_equal(i2.cpu().numpy(), h["j2"])
    td, ti = three_nn(a, b)
    np.testing.assert_array_equal(td.cpu().numpy(), h["t_d"])
    np.testing.assert_array_equal(ti.cpu().numpy(), h["t_i"])


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("case", ["demo", "lattice", "dups", "m2", "m1", "fp_level", "grid"])
def test_three_nn_equals_the_reference_binary_vectors(case):
    """HIP three_nn (op API and the drop-in symbol with threenn_cpu's argument order) == the reference's compiled threenn_cpu, bit for bit"""
    from gspn_amd import _lib as L
    from gspn_amd.tf_interpolate import three_nn
    g = np.load(os.path.join(GOLD, "threenn_ref.npz"))
    x1, x2 = dev(g[case + "_xyz1"]), dev(g[case + "_xyz2"])
    d, i = three_nn(x1, x2)
    np.testing.assert_array_equal(i.cpu().numpy(), g[case + "_idx"])
    np.testing.assert_array_equal(_bits(d.cpu().numpy()), _bits(g[case + "_dist"]))
    b, n, _ = x1.shape
    m = x2.shape[1]
    d2 = torch.full((b, n, 3), float("nan"), device="cuda")
    i2 = torch.full((b, n, 3), -1, dtype=torch.int32, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    assert L.lib().gspn_threenn(b, n, m, P(x1), P(x2), P(d2), P(i2), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    np.testing.assert_array_equal(i2.cpu().numpy(), g[case + "_idx"])
    np.testing.assert_array_equal(_bits(d2.cpu().numpy()), _bits(g[case + "_dist"]))


def _nm(h, a, b):
    bb, n, _ = a.shape
    m = b.shape[1]
    d1 = torch.empty((bb, n), device="cuda")
    d2 = torch.empty((bb, m), device="cuda")
    i1 = torch.empty((bb, n), dtype=torch.int32, device="cuda")
    i2 = torch.empty((bb, m), dtype=torch.int32, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    assert h.gspn_nmdistance(bb, n, P(a), m, P(b), P(d1), P(i1), P(d2), P(i2), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in (d1, i1, d2, i2)]


@pytest.mark.parametrize("case", ["demo", "lattice", "single", "ins"])
def test_nn_distance_equals_the_reference_binary_vectors(case):
    from gspn_amd import _lib as L
    from tests.test_gpu_policy import _variant
    g = np.load(os.path.join(GOLD, "nnsearch_ref.npz"))
    a, b = dev(g[case + "_xyz1"]), dev(g[case + "_xyz2"])
    want = [g[case + "_" + k] for k in ("d1", "i1", "d2", "i2")]
    d1, i1, d2, i2 = _nm(L.lib(), a, b)                                  # the product: NmDistanceKernel's contraction (policy 2)
    np.testing.assert_array_equal(i1, want[1])
    np.testing.assert_array_equal(i2, want[3])
    np.testing.assert_allclose(d1, want[0], rtol=2.5e-7, atol=0)
    np.testing.assert_allclose(d2, want[2], rtol=2.5e-7, atol=0)
    for got, w in zip(_nm(_variant(0), a, b), want):                     # unfused build == the CPU twin as g++ -O2 compiled it
        np.testing.assert_array_equal(_bits(got), _bits(w))


def test_three_nn_and_nn_distance_equal_the_live_reference_binary_at_the_model_shapes():
    """oracle/_ref/libslices_ref.so travels with the snapshot (like every built .so): the dense FP level 32768 <- 2048 (cell-grid kernel) on one U and one S
    scene, and the Chamfer shape 64 x (512, 512), against the reference's compiled loops run on this box's host"""
    from oracle import oracle as O
    from gspn_amd.tf_interpolate import three_nn
    from tests.test_gpu_policy import _variant
    if O.slices_lib() is None:
        pytest.skip("oracle/_ref/libslices_ref.so absent")
    from gspn_amd.tf_sampling import farthest_point_sample, gather_point
    for kind in ("U", "S"):
        x1 = dev(D.batch(kind, 1, 32768, 3))
        x2 = gather_point(x1, farthest_point_sample(2048, x1))
        d, i = three_nn(x1, x2)
        rd, ri = O.ref_three_nn(x1.cpu().numpy(), x2.cpu().numpy())
        np.testing.assert_array_equal(i.cpu().numpy(), ri)
        np.testing.assert_array_equal(_bits(d.cpu().numpy()), _bits(rd))
    a, b = D.batch("U", 64, 512, 100), D.batch("U", 64, 512, 300)
    want = O.ref_nnsearch(a, b)
    for got, w in zip(_nm(_variant(0), dev(a), dev(b)), want):
        np.testing.assert_array_equal(_bits(got), _bits(w))
