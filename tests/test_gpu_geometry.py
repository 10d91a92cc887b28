"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle on identical seeded inputs.
Index outputs must match bit-exactly; scatter-add gradients within fp32 rounding."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import data as D

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("kind,b,n,m", [
    ("U", 1, 4096, 512),      # BASELINE config 1
    ("U", 3, 512, 128), ("U", 2, 300, 64), ("U", 2, 1000, 333), ("U", 2, 2048, 512),
    ("D", 2, 4096, 1024), ("D", 2, 8192, 256), ("U", 2, 16384, 128), ("D", 1, 20000, 300),
    ("U", 2, 32768, 64), ("D", 1, 32768, 600), ("S", 1, 6000, 700),
    ("U", 1, 7, 7), ("U", 1, 5, 9), ("U", 2, 1, 3), ("U", 40, 600, 16),
])
def test_fps_matches_oracle(kind, b, n, m):
    from gspn_amd.tf_sampling import farthest_point_sample
    xyz = D.batch(kind, b, n)
    ref = O.farthest_point_sample(m, xyz)
    got = farthest_point_sample(m, dev(xyz)).cpu().numpy()
    assert got.dtype == np.int32 and got.shape == (b, m)
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("mode,min_n", [("cells", 64), ("cells_torch", 64), ("resident", 0)])
@pytest.mark.parametrize("kind,b,n,m", [("U", 2, 4096, 700), ("D", 2, 9000, 1200), ("S", 1, 20000, 900), ("D", 1, 32768, 1500),
                                        ("U", 3, 1000, 1100), ("D", 2, 300, 64), ("U", 1, 64, 64)])
def test_fps_all_kernels_agree_with_oracle(mode, min_n, kind, b, n, m, monkeypatch):
    """the cell kernel (HIP pre-pass or an arbitrary torch-side partition) and the resident kernel return the same indices"""
    from gspn_amd import tf_sampling
    monkeypatch.setattr(tf_sampling, "FPS_MODE", mode)
    monkeypatch.setattr(tf_sampling, "FPS_CELLS_MIN_N", min_n)
    xyz = D.batch(kind, b, n, 3)
    ref = O.farthest_point_sample(m, xyz)
    got = tf_sampling.farthest_point_sample(m, dev(xyz)).cpu().numpy()
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("mode", ["cells", "cells_torch", "resident"])
@pytest.mark.parametrize("b,n,m,grid", [(2, 32768, 1500, 24), (1, 32768, 3000, 12), (2, 20000, 700, 16), (1, 9000, 500, 10), (1, 32768, 40, 2)])
def test_fps_lattice_ties(mode, b, n, m, grid, monkeypatch):
    """points on a small integer lattice: exact ties between the maxima of different lanes, sub-cells and waves, duplicates, and (for
    m > grid^3) the degenerate tail -- the reference's tie order (k mod 512, k) decides every one of them"""
    from gspn_amd import tf_sampling
    monkeypatch.setattr(tf_sampling, "FPS_MODE", mode)
    monkeypatch.setattr(tf_sampling, "FPS_CELLS_MIN_N", 64)
    rng = np.random.default_rng(grid * 1000 + n)
    xyz = (rng.integers(0, grid, size=(b, n, 3)).astype(np.float32) / np.float32(8.0)).astype(np.float32)
    ref = O.farthest_point_sample(m, xyz)
    got = tf_sampling.farthest_point_sample(m, dev(xyz)).cpu().numpy()
    np.testing.assert_array_equal(got, ref)


def test_fps_all_points_identical_full_size():
    from gspn_amd.tf_sampling import farthest_point_sample
    xyz = np.full((2, 32768, 3), 0.5, np.float32)
    got = farthest_point_sample(50, dev(xyz)).cpu().numpy()
    assert (got == 0).all()


def test_fps_streaming_large_n():
    from gspn_amd.tf_sampling import farthest_point_sample
    xyz = D.batch("D", 2, 40000)
    ref = O.farthest_point_sample(200, xyz)
    got = farthest_point_sample(200, dev(xyz)).cpu().numpy()
    np.testing.assert_array_equal(got, ref)


def test_fps_all_points_identical():
    from gspn_amd.tf_sampling import farthest_point_sample
    xyz = np.ones((2, 1500, 3), np.float32) * 0.25
    ref = O.farthest_point_sample(20, xyz)
    got = farthest_point_sample(20, dev(xyz)).cpu().numpy()
    np.testing.assert_array_equal(got, ref)
    assert (got == 0).all()


@pytest.mark.parametrize("kind,b,n,m,r,ns", [
    ("U", 1, 4096, 512, 0.2, 32),     # BASELINE config 1
    ("U", 2, 8192, 256, 0.1, 32), ("D", 2, 4096, 300, 0.15, 64), ("U", 2, 2048, 512, 0.4, 32),
    ("U", 2, 1000, 77, 0.05, 16), ("U", 1, 5000, 128, 1.5, 512), ("U", 2, 300, 300, 1e-3, 8),
    ("S", 1, 6000, 256, 0.4, 32), ("U", 1, 33, 5, 0.5, 100),
])
def test_ball_query_matches_oracle(kind, b, n, m, r, ns):
    from gspn_amd.tf_grouping import query_ball_point
    xyz = D.batch(kind, b, n)
    q = O.gather_point(xyz, O.farthest_point_sample(m, xyz))
    ridx, rcnt = O.query_ball_point(r, ns, xyz, q)
    idx, cnt = query_ball_point(r, ns, dev(xyz), dev(q))
    np.testing.assert_array_equal(cnt.cpu().numpy(), rcnt)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)


def test_ball_query_no_hit_rows_zero_filled():
    from gspn_amd.tf_grouping import query_ball_point
    xyz = D.batch("U", 1, 500)
    q = (xyz[:, :10] + 10.0).astype(np.float32)
    idx, cnt = query_ball_point(0.1, 16, dev(xyz), dev(q))
    assert (cnt.cpu().numpy() == 0).all() and (idx.cpu().numpy() == 0).all()


def test_ball_query_radius_boundary():
    """hit test is sqrtf(d2) < radius, not d2 < radius^2: probe radii at exact distances"""
    from gspn_amd.tf_grouping import query_ball_point
    xyz = D.batch("U", 1, 2048)
    q = xyz[:, :4].copy()
    d = np.sqrt(((xyz[0][None] - q[0][:, None]).astype(np.float64) ** 2).sum(-1))
    for r in [float(np.float32(d[0, 100])), float(np.float32(d[1, 7])), float(np.nextafter(np.float32(d[2, 900]), np.float32(1)))]:
        ridx, rcnt = O.query_ball_point(r, 2048, xyz, q)
        idx, cnt = query_ball_point(r, 2048, dev(xyz), dev(q))
        np.testing.assert_array_equal(cnt.cpu().numpy(), rcnt)
        np.testing.assert_array_equal(idx.cpu().numpy(), ridx)


@pytest.mark.parametrize("c", [3, 6, 64, 67])
def test_group_point_and_grad(c):
    from gspn_amd.tf_grouping import group_point
    rng = np.random.default_rng(5)
    b, n, m, ns = 2, 1024, 128, 16
    pts = rng.standard_normal((b, n, c)).astype(np.float32)
    idx = rng.integers(0, n, size=(b, m, ns)).astype(np.int32)
    idx[:, :, 8:] = idx[:, :, :1]           # padded rows repeat one index (atomic contention)
    p = dev(pts).requires_grad_(True)
    out = group_point(p, dev(idx))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), O.group_point(pts, idx))
    go = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(dev(go))
    np.testing.assert_allclose(p.grad.cpu().numpy(), O.group_point_grad(pts, idx, go), rtol=1e-5, atol=1e-5)


def test_gather_point_and_grad():
    from gspn_amd.tf_sampling import gather_point
    rng = np.random.default_rng(6)
    b, n, m = 3, 2000, 500
    xyz = rng.standard_normal((b, n, 3)).astype(np.float32)
    idx = rng.integers(0, n, size=(b, m)).astype(np.int32)
    x = dev(xyz).requires_grad_(True)
    out = gather_point(x, dev(idx))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), O.gather_point(xyz, idx))
    go = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(dev(go))
    np.testing.assert_allclose(x.grad.cpu().numpy(), O.gather_point_grad(xyz, idx, go), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("b,n,m", [(2, 2048, 512), (1, 1000, 2), (2, 700, 1500), (1, 50, 1), (1, 300, 3)])
def test_three_nn_matches_oracle(b, n, m):
    from gspn_amd.tf_interpolate import three_nn
    dense = D.batch("D", b, n, 10)
    sparse = D.batch("D", b, m, 50)
    rd, ri = O.three_nn(dense, sparse)
    d, i = three_nn(dev(dense), dev(sparse))
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(d.cpu().numpy(), rd)


@pytest.mark.parametrize("b,n,m", [(2, 777, 128), (3, 2048, 129), (1, 100, 256), (2, 513, 257), (1, 4099, 512), (2, 600, 513), (1, 333, 1024), (1, 50, 1025),
                                   (8, 2048, 512), (8, 512, 128)])
def test_three_nn_small_known_clouds_on_a_lattice(b, n, m):
    """the wave-per-query kernel (m <= 1024: every lane keeps its own three best, three wave-wide (distance, index) minima pick the
    result) against the oracle's sequential cascade (tf_interpolate.cpp:60-103) on integer lattices -- equal distances everywhere, so the
    tie rule (the LOWER index stays) decides most slots -- at every template-instance boundary and just past the kernel's range"""
    from gspn_amd.tf_interpolate import three_nn
    rng = np.random.default_rng(b * 1000 + m)
    dense = rng.integers(0, 6, size=(b, n, 3)).astype(np.float32)
    sparse = rng.integers(0, 6, size=(b, m, 3)).astype(np.float32)
    rd, ri = O.three_nn(dense, sparse)
    d, i = three_nn(dev(dense), dev(sparse))
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(d.cpu().numpy(), rd)
    # and on continuous clouds with duplicated points
    dense, sparse = D.batch("D", b, n, 3), D.batch("D", b, m, 4)
    rd, ri = O.three_nn(dense, sparse)
    d, i = three_nn(dev(dense), dev(sparse))
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(d.cpu().numpy(), rd)


@pytest.mark.parametrize("kind,b,n,m", [("D", 2, 9000, 700), ("S", 3, 12000, 300), ("U", 1, 8192, 3)])
def test_three_nn_in_a_given_order(kind, b, n, m):
    """three_nn(..., order=): the order the unknown points are handed to the threads in (a random permutation, and the spatial order
    the FPS pre-pass leaves behind) changes nothing in the result"""
    from gspn_amd.tf_interpolate import three_nn
    from gspn_amd.tf_sampling import farthest_point_sample
    dense = D.batch(kind, b, n, 10)
    sparse = D.batch(kind, b, m, 50)
    rd, ri = O.three_nn(dense, sparse)
    rng = np.random.default_rng(n)
    perm = np.stack([rng.permutation(n) for _ in range(b)]).astype(np.int32)
    _, fps_order = farthest_point_sample(16, dev(dense), return_order=True)
    assert fps_order is not None and fps_order.shape == (b, n)
    assert (np.sort(fps_order.cpu().numpy(), axis=1) == np.arange(n)).all()
    for order in (dev(perm), fps_order):
        d, i = three_nn(dev(dense), dev(sparse), order=order)
        np.testing.assert_array_equal(i.cpu().numpy(), ri)
        np.testing.assert_array_equal(d.cpu().numpy(), rd)
    with pytest.raises(ValueError):
        three_nn(dev(dense), dev(sparse), order=dev(perm[:, :-1]))


@pytest.mark.parametrize("c", [1, 16, 64, 131])
def test_three_interpolate_and_grad(c):
    from gspn_amd.tf_interpolate import three_interpolate
    rng = np.random.default_rng(7)
    b, n, m = 2, 900, 128
    pts = rng.standard_normal((b, m, c)).astype(np.float32)
    idx = rng.integers(0, m, size=(b, n, 3)).astype(np.int32)
    w = rng.random((b, n, 3)).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    p = dev(pts).requires_grad_(True)
    out = three_interpolate(p, dev(idx), dev(w))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), O.three_interpolate(pts, idx, w))   # same op order -> bitwise
    go = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(dev(go))
    # r04: the op's gradient is a gather through the inverse lists of idx in the reference's own summation order (tf_interpolate.cpp:131-153):
    # BIT-exact against the oracle's restatement and -- where oracle/_ref exists -- against the reference's own compiled loop
    np.testing.assert_array_equal(p.grad.cpu().numpy(), O.three_interpolate_grad(pts, idx, w, go))
    if O.ref_lib() is not None:
        np.testing.assert_array_equal(p.grad.cpu().numpy(), O.ref_three_interpolate_grad(pts, idx, w, go))


def test_three_interpolate_grad_bit_exact_vs_reference_binary_at_the_fp_level_shape():
    """the reference's own compiled interpolate_grad_cpu (oracle/_ref, built from tf_ops/3d_interpolation/interpolate.cpp) on the shape of
    the model's last FP level (2048 sparse points, 3-NN of a 32768-point cloud; one scene, 64 channels), real 3-NN indices and weights"""
    from gspn_amd.geometry import fp_geometry
    from gspn_amd.tf_interpolate import three_interpolate
    from gspn_amd.tf_sampling import farthest_point_sample, gather_point
    xyz = dev(D.batch("S", 1, 32768, 3))
    sparse = gather_point(xyz, farthest_point_sample(2048, xyz))
    g = fp_geometry(xyz, sparse)
    rng = np.random.default_rng(11)
    pts = rng.standard_normal((1, 2048, 64)).astype(np.float32)
    go = rng.standard_normal((1, 32768, 64)).astype(np.float32)
    p = dev(pts).requires_grad_(True)
    three_interpolate(p, g.idx, g.weight).backward(dev(go))
    idx_np, w_np = g.idx.cpu().numpy(), g.weight.cpu().numpy()
    np.testing.assert_array_equal(p.grad.cpu().numpy(), O.three_interpolate_grad(pts, idx_np, w_np, go))
    if O.ref_lib() is not None:
        np.testing.assert_array_equal(p.grad.cpu().numpy(), O.ref_three_interpolate_grad(pts, idx_np, w_np, go))
    # default (r05): nothing is kept on the index tensor between calls
    assert getattr(g.idx, "_gspn_inv", None) is None
    # opt-in cache: a second backward through the same idx builds nothing and gives the same bits; a refill of the buffer BEHIND torch's
    # back (no version bump: what a graph replay or a raw kernel does) is caught by invalidate(), and consuming on another stream is ordered
    from gspn_amd import invlists
    prev = invlists.enable_cache(True)
    try:
        p2 = dev(pts).requires_grad_(True)
        three_interpolate(p2, g.idx, g.weight).backward(dev(go))
        assert torch.equal(p.grad, p2.grad)
        ent = g.idx._gspn_inv[2048]
        assert ent[0] == (g.idx._version, g.idx.data_ptr(), invlists._generation[0])
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            p3 = dev(pts).requires_grad_(True)
            three_interpolate(p3, g.idx, g.weight).backward(dev(go))
        torch.cuda.current_stream().wait_stream(side)
        assert torch.equal(p.grad, p3.grad) and g.idx._gspn_inv[2048] is ent          # a hit, served across streams
        new_idx = torch.roll(g.idx, 1, dims=1).contiguous()
        v = g.idx._version
        g.idx.data.copy_(new_idx)                                                     # .data: no version bump
        assert g.idx._version == v
        invlists.invalidate(g.idx)
        p4 = dev(pts).requires_grad_(True)
        three_interpolate(p4, g.idx, g.weight).backward(dev(go))
        np.testing.assert_array_equal(p4.grad.cpu().numpy(), O.three_interpolate_grad(pts, new_idx.cpu().numpy(), w_np, go))
    finally:
        invlists.enable_cache(prev)


def test_standalone_scatter_gradients_are_deterministic_gathers():
    """group_point / gather_point gradients through the inverse lists (invlists.py): equal to the oracle's sequential scatter-add within
    rounding (the reference's atomicAdd defines no order), and bit-identical from run to run even with heavily repeated indices"""
    from gspn_amd.tf_grouping import group_point
    from gspn_amd.tf_sampling import gather_point
    rng = np.random.default_rng(12)
    b, n, m, ns, c = 2, 700, 90, 16, 64
    pts = rng.standard_normal((b, n, c)).astype(np.float32)
    idx = rng.integers(0, 40, size=(b, m, ns)).astype(np.int32)            # 40 distinct values: every group is long
    go = rng.standard_normal((b, m, ns, c)).astype(np.float32)
    grads = []
    for _ in range(3):
        p = dev(pts).requires_grad_(True)
        group_point(p, dev(idx)).backward(dev(go))
        grads.append(p.grad.clone())
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])
    np.testing.assert_allclose(grads[0].cpu().numpy(), O.group_point_grad(pts, idx, go), rtol=1e-5, atol=2e-5)
    # in ascending grouped position the sums are exactly the sequential loop's
    ref = np.zeros((b, n, c), np.float32)
    for s in range(b):
        for j in range(m):
            for k in range(ns):
                ref[s, idx[s, j, k]] += go[s, j, k]
    np.testing.assert_array_equal(grads[0].cpu().numpy(), ref)
    xyz = rng.standard_normal((b, n, 3)).astype(np.float32)
    gi = rng.integers(0, 5, size=(b, 300)).astype(np.int32)                # a point sampled many times (npoint > distinct points)
    g3 = rng.standard_normal((b, 300, 3)).astype(np.float32)
    ref = np.zeros((b, n, 3), np.float32)
    for s in range(b):
        for j in range(300):
            ref[s, gi[s, j]] += g3[s, j]
    # narrow rows (c < 16) take the atomic kernel by default since r05 (order-free sums, like the reference's atomicAdd: 43 us against
    # 170 + 89 us at the bench shape) ...
    x = dev(xyz).requires_grad_(True)
    gather_point(x, dev(gi)).backward(dev(g3))
    np.testing.assert_allclose(x.grad.cpu().numpy(), ref, rtol=1e-5, atol=2e-5)
    # ... and the fixed-order gather when determinism is asked for (GSPN_DETERMINISTIC_GRADS=1)
    from gspn_amd import invlists
    prev, invlists.DETERMINISTIC = invlists.DETERMINISTIC, True
    try:
        outs = []
        for _ in range(3):
            x = dev(xyz).requires_grad_(True)
            gather_point(x, dev(gi)).backward(dev(g3))
            outs.append(x.grad.clone())
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        np.testing.assert_array_equal(outs[0].cpu().numpy(), ref)
        p3 = dev(xyz).requires_grad_(True)                                   # group_point on 3 columns: the narrow list walk
        i3 = rng.integers(0, 40, size=(b, m, ns)).astype(np.int32)
        go3 = rng.standard_normal((b, m, ns, 3)).astype(np.float32)
        group_point(p3, dev(i3)).backward(dev(go3))
        r3 = np.zeros((b, n, 3), np.float32)
        for s in range(b):
            for j in range(m):
                for k in range(ns):
                    r3[s, i3[s, j, k]] += go3[s, j, k]
        np.testing.assert_array_equal(p3.grad.cpu().numpy(), r3)
    finally:
        invlists.DETERMINISTIC = prev


@pytest.mark.parametrize("b,n,m", [(4, 512, 512), (2, 1500, 700), (3, 100, 2500), (1, 1, 1), (2, 3000, 1500)])     # the last one: beyond the LDS kernel (global atomics)
def test_nn_distance_and_grad(b, n, m):
    from gspn_amd.tf_nndistance import nn_distance
    a = D.batch("D", b, n, 3)
    c = D.batch("D", b, m, 30)
    rd1, ri1, rd2, ri2 = O.nn_distance(a, c)
    ta, tc = dev(a).requires_grad_(True), dev(c).requires_grad_(True)
    d1, i1, d2, i2 = nn_distance(ta, tc)
    np.testing.assert_array_equal(i1.cpu().numpy(), ri1)
    np.testing.assert_array_equal(i2.cpu().numpy(), ri2)
    np.testing.assert_array_equal(d1.detach().cpu().numpy(), rd1)
    np.testing.assert_array_equal(d2.detach().cpu().numpy(), rd2)
    rng = np.random.default_rng(8)
    g1 = rng.standard_normal(rd1.shape).astype(np.float32)
    g2 = rng.standard_normal(rd2.shape).astype(np.float32)
    (d1 * dev(g1)).sum().add((d2 * dev(g2)).sum()).backward()
    rg1, rg2 = O.nn_distance_grad(a, c, g1, ri1, g2, ri2)
    np.testing.assert_allclose(ta.grad.cpu().numpy(), rg1, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(tc.grad.cpu().numpy(), rg2, rtol=1e-5, atol=1e-5)
    if n + m > 4096:          # r04: beyond the LDS kernel the gradient is a gather in the order of the reference's sequential CPU twin: bit-exact
        np.testing.assert_array_equal(ta.grad.cpu().numpy(), rg1)
        np.testing.assert_array_equal(tc.grad.cpu().numpy(), rg2)


def test_group_maxpool_and_selection_sort():
    from gspn_amd.tf_grouping import group_maxpool, select_top_k, knn_point
    rng = np.random.default_rng(9)
    b, n, m, ns, c = 2, 600, 70, 12, 10
    pts = rng.standard_normal((b, n, c)).astype(np.float32)
    idx = rng.integers(0, n, size=(b, m, ns)).astype(np.int32)
    p = dev(pts).requires_grad_(True)
    out, mi = group_maxpool(p, dev(idx))
    ro, rmi = O.group_maxpool(pts, idx)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), ro)
    np.testing.assert_array_equal(mi.cpu().numpy(), rmi)
    go = rng.standard_normal(ro.shape).astype(np.float32)
    out.backward(dev(go))
    np.testing.assert_allclose(p.grad.cpu().numpy(), O.group_maxpool_grad(pts, rmi, go), rtol=1e-5, atol=1e-5)
    dist = rng.integers(0, 20, size=(2, 9, 150)).astype(np.float32)    # many ties
    oi, od = select_top_k(17, dev(dist))
    roi, rod = O.select_top_k(17, dist)
    np.testing.assert_array_equal(oi.cpu().numpy(), roi)
    np.testing.assert_array_equal(od.cpu().numpy(), rod)
    x1 = D.batch("D", 2, 400, 1)
    x2 = D.batch("D", 2, 30, 2)
    v, i = knn_point(8, dev(x1), dev(x2))
    rv, ri = O.knn_point(8, x1, x2)
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(v.cpu().numpy(), rv)


def test_prob_sample():
    from gspn_amd.tf_sampling import prob_sample
    rng = np.random.default_rng(11)
    for (b, n, m) in [(3, 1000, 64), (2, 20000, 500), (1, 5, 9)]:
        w = rng.random((b, n)).astype(np.float32)
        r = rng.random((b, m)).astype(np.float32)
        got = prob_sample(dev(w), dev(r)).cpu().numpy()
        np.testing.assert_array_equal(got, O.prob_sample(w, r))


@pytest.mark.parametrize("b,n", [(2, 1), (2, 3), (3, 4), (2, 5), (3, 1000), (2, 8191), (2, 8192), (2, 8193), (2, 20000), (1, 8192 * 3 + 6), (5, 70001)])
def test_prob_sample_prefix_sums_round_like_the_reference(b, n):
    """the cumulative sums (the op's temp buffer) carry the reference's association order bit for bit (tf_sampling_g.cu:7-80):
    groups of 4, Brent-Kung tree over the group totals, compensated carry across 8192-element tiles"""
    from gspn_amd import _lib as L
    rng = np.random.default_rng(n)
    w = (rng.random((b, n)) ** 3).astype(np.float32)
    r = rng.random((b, 7)).astype(np.float32)
    tw, tr = dev(w), dev(r)
    temp = torch.empty((b, n), dtype=torch.float32, device="cuda")
    out = torch.empty((b, 7), dtype=torch.int32, device="cuda")
    L.check(L.lib().gspn_probsample(b, n, 7, L.ptr(tw), L.ptr(tr), L.ptr(temp), L.ptr(out), L.stream()), "prob_sample")
    np.testing.assert_array_equal(temp.cpu().numpy().view(np.uint32), O.cumsum(w).view(np.uint32))
    np.testing.assert_array_equal(out.cpu().numpy(), O.prob_sample(w, r))


def test_cpu_tensor_is_rejected_loudly():
    from gspn_amd import _lib
    from gspn_amd.tf_sampling import farthest_point_sample
    with pytest.raises(_lib.GspnHipError):
        farthest_point_sample(4, torch.zeros(1, 16, 3))


@pytest.mark.parametrize("kind,b,n,m,k", [("U", 2, 8192, 512, 32), ("D", 2, 3000, 100, 16), ("U", 1, 70, 33, 32), ("U", 2, 40, 9, 7),
                                          ("S", 1, 5000, 64, 1), ("U", 3, 2048, 2048, 3), ("U", 1, 33, 40, 32)])
def test_knn_direct_matches_reference_construction(kind, b, n, m, k):
    """knn_point without the (b,m,n) matrix == dense matrix + selection sort + slice of the oracle (tf_grouping.py:71-96)"""
    from gspn_amd.tf_grouping import knn_point
    x1 = D.batch(kind, b, n)
    x2 = D.batch(kind, b, m, 50)
    v, i = knn_point(k, dev(x1), dev(x2))
    rv, ri = O.knn_point(k, x1, x2)
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(v.cpu().numpy(), rv)


@pytest.mark.parametrize("n,m,k,grid", [(500, 64, 32, 3), (4000, 128, 32, 5), (4000, 50, 9, 2), (100, 30, 32, 2), (64, 10, 32, 4), (70, 10, 20, 1)])
def test_knn_direct_ties_follow_the_selection_sort(n, m, k, grid):
    """integer lattice: most distances are tied, so the order inside the first k depends on where the in-place selection sort has
    moved the elements it displaced (tf_grouping_g.cu:162-183) -- the direct kernel replays exactly that"""
    from gspn_amd.tf_grouping import knn_point
    rng = np.random.default_rng(n + k)
    x1 = rng.integers(0, grid, size=(2, n, 3)).astype(np.float32)
    x2 = rng.integers(0, grid, size=(2, m, 3)).astype(np.float32)
    v, i = knn_point(k, dev(x1), dev(x2))
    rv, ri = O.knn_point(k, x1, x2)
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(v.cpu().numpy(), rv)


def test_knn_dense_fallback_and_validation():
    from gspn_amd.tf_grouping import knn_point
    x1 = D.batch("U", 1, 300)
    x2 = D.batch("U", 1, 20, 9)
    v, i = knn_point(40, dev(x1), dev(x2))                  # k > 32: the reference's own construction
    rv, ri = O.knn_point(40, x1, x2)
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(v.cpu().numpy(), rv)
    with pytest.raises(ValueError):
        knn_point(301, dev(x1), dev(x2))
    with pytest.raises(ValueError):
        knn_point(0, dev(x1), dev(x2))


def test_knn_direct_random_shapes():
    """seeded sweep of knn_point (direct kernel) over sizes, k and tie-heavy lattices against the oracle's dense construction"""
    from gspn_amd.tf_grouping import knn_point
    rng = np.random.default_rng(77)
    for trial in range(20):
        n = int(rng.integers(33, 3000))
        m = int(rng.integers(1, 200))
        k = int(rng.integers(1, min(32, n) + 1))
        b = int(rng.integers(1, 3))
        if trial % 2:
            x1 = rng.integers(0, 4, size=(b, n, 3)).astype(np.float32)
            x2 = rng.integers(0, 4, size=(b, m, 3)).astype(np.float32)
        else:
            x1 = rng.random((b, n, 3), dtype=np.float32)
            x2 = rng.random((b, m, 3), dtype=np.float32)
        v, i = knn_point(k, dev(x1), dev(x2))
        rv, ri = O.knn_point(k, x1, x2)
        np.testing.assert_array_equal(i.cpu().numpy(), ri, err_msg="trial %d n=%d m=%d k=%d" % (trial, n, m, k))
        np.testing.assert_array_equal(v.cpu().numpy(), rv)


def test_fps_prepass_voxel_order_is_a_permutation_in_voxel_order():
    """gspn_fps_cells_prepass_order: the scan order handed to three_nn is a permutation of every scene's points, sorted by the 16^3-voxel
    Morton id of the pre-pass (non-decreasing along the order; the order inside a voxel is arbitrary), and FPS itself is unaffected"""
    from gspn_amd.tf_sampling import farthest_point_sample
    xyz = D.batch("U", 3, 20000, 11)
    t = dev(xyz)
    idx, order = farthest_point_sample(512, t, return_order=True)
    np.testing.assert_array_equal(idx.cpu().numpy(), O.farthest_point_sample(512, xyz, mt=True))
    order = order.cpu().numpy()
    spread = lambda v: (v & 1) | ((v & 2) << 2) | ((v & 4) << 4) | ((v & 8) << 6)
    for b in range(3):
        np.testing.assert_array_equal(np.sort(order[b]), np.arange(20000))
        p = xyz[b]
        lo, hi = p.min(0), p.max(0)
        inv = np.where(hi > lo, np.float32(16.0) / (hi - lo), np.float32(0.0)).astype(np.float32)
        q = np.clip(((p - lo) * inv).astype(np.int32), 0, 15)
        vox = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
        along = vox[order[b]]
        assert (np.diff(along) >= 0).all()


@pytest.mark.parametrize("case", ["lattice", "plane", "outside", "clusters", "duplicates", "m1025", "far_origin", "far_plane"])
def test_three_nn_cell_grid_is_exact(case):
    """r04: known clouds of more than 1024 points go through three_nn_grid_kernel (the known points sorted into cells in LDS, the 3 x 3 x 3
    block around the query's cell, acceptance by the distance to the block's faces, a restart over the ball's cells otherwise).  Exact
    against the oracle -- indices and distances bit for bit -- where that logic is stressed: integer lattices (ties everywhere, third
    distances exactly ON a cell face), a flat axis, queries far outside the known bounding box, tight clusters with empty space between
    them (restart path, fewer than three points in the block), duplicated points, the smallest cloud that takes this kernel."""
    from gspn_amd.tf_interpolate import three_nn
    rng = np.random.default_rng(31)
    b, n = 2, 3000
    if case == "lattice":
        g = np.stack(np.meshgrid(np.arange(13), np.arange(13), np.arange(13), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)     # 2197 points
        sparse = np.stack([g[rng.permutation(len(g))] for _ in range(b)])
        dense = rng.integers(-2, 15, size=(b, n, 3)).astype(np.float32)
        dense[:, ::3] += 0.5                                                  # some queries on cell faces / between lattice points
    elif case == "plane":
        sparse = rng.random((b, 1500, 3)).astype(np.float32)
        sparse[..., 2] = 0.25                                                 # zero extent along z
        dense = rng.random((b, n, 3)).astype(np.float32)
    elif case == "outside":
        sparse = rng.random((b, 2048, 3)).astype(np.float32)
        dense = (rng.random((b, n, 3)).astype(np.float32) - 0.5) * 40.0       # most queries far outside [0, 1)^3
    elif case == "clusters":
        centres = rng.random((b, 6, 3)).astype(np.float32) * 10.0
        sparse = (centres[:, rng.integers(0, 6, size=2500)] + 0.01 * rng.standard_normal((b, 2500, 3))).astype(np.float32)
        sparse[:, :2] += 50.0                                                 # two stragglers stretch the bounding box: almost all cells empty
        dense = (rng.random((b, n, 3)).astype(np.float32)) * 12.0
    elif case == "duplicates":
        sparse = D.batch("D", b, 4096, 40)
        sparse[:, 2000:] = sparse[:, :2096]                                   # every point twice: each nearest neighbour is a tie
        dense = D.batch("U", b, n, 41)
    elif case in ("far_origin", "far_plane"):
        #  large ABSOLUTE coordinates with a small extent -- metres around 5e5, coordinates on a 1/16 m raster so that the inputs
        # are exact: the acceptance bound must be formed relative to the box (q - lo), not as lo + c*h at the scale of the coordinates
        sparse = (np.float32(5e5) + rng.integers(0, 64, size=(b, 3000, 3)).astype(np.float32) / 16).astype(np.float32)
        dense = (np.float32(5e5) + rng.integers(-8, 72, size=(b, n, 3)).astype(np.float32) / 16).astype(np.float32)
        if case == "far_plane":
            sparse[..., 1] = np.float32(5e5 + 1.0)
    else:
        sparse = D.batch("U", b, 1025, 50)
        dense = D.batch("U", b, n, 51)
    rd, ri = O.three_nn(dense, sparse)
    d, i = three_nn(dev(dense), dev(sparse))
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(d.cpu().numpy(), rd)
    perm = np.stack([rng.permutation(n) for _ in range(b)]).astype(np.int32)
    d2, i2 = three_nn(dev(dense), dev(sparse), order=dev(perm))                # the result never depends on the scan order
    assert torch.equal(i2, i) and torch.equal(d2, d)
