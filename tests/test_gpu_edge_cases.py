"""Edge cases of the op API against the oracle: empty batches / queries, single points, more samples than points,
fewer than three known points, nsample larger than the cloud, clouds of one point.  tf_sampling_g.cu:105-170 (m > n keeps emitting the
arg-max of an all-zero field), tf_grouping_g.cu:6-39, tf_interpolate.cpp:60-103 (best = 1e40, besti = 0 when m < 3), tf_nndistance_g.cu:5-127."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("b,n,m", [(1, 1, 1), (2, 1, 5), (3, 7, 20), (2, 64, 64), (1, 65, 200), (4, 513, 513), (2, 2049, 4000), (1, 9000, 9500)])
def test_fps_more_samples_than_points_and_tiny_clouds(b, n, m):
    from gspn_amd.tf_sampling import farthest_point_sample
    xyz = np.random.default_rng(n * 31 + m).random((b, n, 3)).astype(np.float32)
    got = farthest_point_sample(m, dev(xyz)).cpu().numpy()
    np.testing.assert_array_equal(got, O.farthest_point_sample(m, xyz))


def test_empty_batches_and_queries():
    from gspn_amd.tf_grouping import group_point, query_ball_point
    from gspn_amd.tf_interpolate import three_interpolate, three_nn
    from gspn_amd.tf_nndistance import nn_distance
    from gspn_amd.tf_sampling import farthest_point_sample, gather_point
    z = torch.zeros((0, 16, 3), device="cuda")
    assert tuple(farthest_point_sample(4, z).shape) == (0, 4)
    assert tuple(gather_point(z, torch.zeros((0, 4), dtype=torch.int32, device="cuda")).shape) == (0, 4, 3)
    idx, cnt = query_ball_point(0.2, 8, z, torch.zeros((0, 4, 3), device="cuda"))
    assert tuple(idx.shape) == (0, 4, 8) and tuple(cnt.shape) == (0, 4)
    x = dev(np.random.default_rng(0).random((2, 50, 3)).astype(np.float32))
    idx, cnt = query_ball_point(0.2, 8, x, torch.zeros((2, 0, 3), device="cuda"))          # no queries
    assert tuple(idx.shape) == (2, 0, 8) and tuple(cnt.shape) == (2, 0)
    g = group_point(x, torch.zeros((2, 0, 8), dtype=torch.int32, device="cuda"))
    assert tuple(g.shape) == (2, 0, 8, 3)
    d, i = three_nn(torch.zeros((2, 0, 3), device="cuda"), x)
    assert tuple(d.shape) == (2, 0, 3)
    out = three_interpolate(torch.zeros((2, 50, 5), device="cuda"), torch.zeros((2, 0, 3), dtype=torch.int32, device="cuda"), torch.zeros((2, 0, 3), device="cuda"))
    assert tuple(out.shape) == (2, 0, 5)
    d1, i1, d2, i2 = nn_distance(torch.zeros((0, 5, 3), device="cuda"), torch.zeros((0, 7, 3), device="cuda"))
    assert tuple(d1.shape) == (0, 5) and tuple(d2.shape) == (0, 7)
    torch.cuda.synchronize()


@pytest.mark.parametrize("m", [1, 2, 3])
def test_three_nn_with_fewer_than_three_known_points(m):
    """tf_interpolate.cpp:66-89: best = 1e40 (+inf in float), besti = 0 for the slots no point fills"""
    from gspn_amd.tf_interpolate import three_nn
    rng = np.random.default_rng(m)
    dense = rng.random((2, 300, 3)).astype(np.float32)
    sparse = rng.random((2, m, 3)).astype(np.float32)
    rd, ri = O.three_nn(dense, sparse)
    d, i = three_nn(dev(dense), dev(sparse))
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(d.cpu().numpy(), rd)
    if m < 3:
        assert np.isinf(rd[..., m:]).all() and (ri[..., m:] == 0).all()


@pytest.mark.parametrize("n,ns,radius", [(5, 32, 10.0), (1, 4, 0.5), (40, 64, 0.3), (33, 33, 100.0)])
def test_ball_query_nsample_larger_than_the_cloud(n, ns, radius):
    from gspn_amd.tf_grouping import query_ball_point
    rng = np.random.default_rng(n)
    xyz = rng.random((2, n, 3)).astype(np.float32)
    q = xyz[:, : max(1, n // 2)].copy()
    ridx, rcnt = O.query_ball_point(radius, ns, xyz, q)
    idx, cnt = query_ball_point(radius, ns, dev(xyz), dev(q))
    np.testing.assert_array_equal(cnt.cpu().numpy(), rcnt)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)


@pytest.mark.parametrize("n,m", [(1, 1), (1, 40), (40, 1), (513, 2)])
def test_nn_distance_single_point_clouds(n, m):
    from gspn_amd.tf_nndistance import nn_distance
    rng = np.random.default_rng(n + m)
    a = rng.standard_normal((3, n, 3)).astype(np.float32)
    c = rng.standard_normal((3, m, 3)).astype(np.float32)
    r = O.nn_distance(a, c)
    ta, tc = dev(a).requires_grad_(True), dev(c).requires_grad_(True)
    got = nn_distance(ta, tc)
    for x, y in zip(got, r):
        np.testing.assert_array_equal(x.detach().cpu().numpy(), y)
    g1 = rng.standard_normal((3, n)).astype(np.float32)
    g2 = rng.standard_normal((3, m)).astype(np.float32)
    (got[0] * dev(g1)).sum().add((got[2] * dev(g2)).sum()).backward()
    ra, rc = O.nn_distance_grad(a, c, g1, r[1], g2, r[3])
    np.testing.assert_allclose(ta.grad.cpu().numpy(), ra, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(tc.grad.cpu().numpy(), rc, rtol=1e-5, atol=1e-5)


def test_knn_and_top_k_at_the_limits():
    from gspn_amd.tf_grouping import knn_point, select_top_k
    rng = np.random.default_rng(2)
    xyz1 = rng.random((2, 9, 3)).astype(np.float32)
    xyz2 = rng.random((2, 4, 3)).astype(np.float32)
    val, idx = knn_point(9, dev(xyz1), dev(xyz2))                       # k == ndataset
    rv, ri = O.knn_point(9, xyz1, xyz2) if hasattr(O, "knn_point") else (None, None)
    if rv is not None:
        np.testing.assert_array_equal(idx.cpu().numpy(), ri)
        np.testing.assert_array_equal(val.cpu().numpy(), rv)
    dist = rng.integers(0, 3, size=(1, 2, 5)).astype(np.float32)
    oi, od = select_top_k(5, dev(dist))                                  # k == n: a full selection sort
    ri, rd = O.select_top_k(5, dist)
    np.testing.assert_array_equal(oi.cpu().numpy(), ri)
    np.testing.assert_array_equal(od.cpu().numpy(), rd)
    with pytest.raises(ValueError):
        knn_point(10, dev(xyz1), dev(xyz2))                             # k > ndataset
