"""CPU suite 1: the oracle is pinned -- against the reference's own compiled code where that exists
(oracle/_ref + tests/golden/interp_ref_*.npz), against the committed oracle fixtures, against an independent
NumPy brute force, and against the invariants of SURVEY.md section 8(c)."""
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import oracle as O
from tests import data as D
from tests import ref_numpy as N

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_dist_policy_is_the_documented_one():
    assert O.dist_policy() == 2


# ---- the real reference code (three_interpolate / grad) --------------------------------------
@pytest.mark.parametrize("name", ["interp_ref_demo.npz", "interp_ref_optest.npz"])
def test_interpolate_restatement_matches_reference_golden(name):
    g = np.load(os.path.join(GOLD, name))
    np.testing.assert_array_equal(O.three_interpolate(g["points"], g["idx"], g["weight"]), g["out"])
    np.testing.assert_array_equal(O.three_interpolate_grad(g["points"], g["idx"], g["weight"], g["grad_out"]), g["grad_points"])
    if "weight2" in g:
        np.testing.assert_array_equal(O.three_interpolate(g["points"], g["idx"], g["weight2"]), g["out2"])


def test_interpolate_restatement_matches_live_reference_build():
    if O.ref_lib() is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    rng = np.random.default_rng(3)
    pts = rng.standard_normal((3, 40, 19)).astype(np.float32)
    idx = rng.integers(0, 40, size=(3, 333, 3)).astype(np.int32)
    w = rng.random((3, 333, 3)).astype(np.float32)
    go = rng.standard_normal((3, 333, 19)).astype(np.float32)
    np.testing.assert_array_equal(O.three_interpolate(pts, idx, w), O.ref_three_interpolate(pts, idx, w))
    np.testing.assert_array_equal(O.three_interpolate_grad(pts, idx, w, go), O.ref_three_interpolate_grad(pts, idx, w, go))


# ---- the real reference code (threenn_cpu, nnsearch): free functions cut out of TF-dependent files at build time ----------
THREENN_CASES = ["demo", "lattice", "dups", "m2", "m1", "fp_level", "grid"]
NNSEARCH_CASES = ["demo", "lattice", "single", "ins"]


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("case", THREENN_CASES)
def test_three_nn_restatement_matches_reference_golden(case):
    """oracle three_nn == vectors of the reference's compiled threenn_cpu (tf_interpolate.cpp:60-103): indices AND squared distances bit for bit
    (inf tails where m < 3 included)"""
    g = np.load(os.path.join(GOLD, "threenn_ref.npz"))
    d, i = O.three_nn(g[case + "_xyz1"], g[case + "_xyz2"])
    np.testing.assert_array_equal(i, g[case + "_idx"])
    np.testing.assert_array_equal(_bits(d), _bits(g[case + "_dist"]))


@pytest.mark.parametrize("case", NNSEARCH_CASES)
def test_nn_distance_cpu_twin_matches_reference_golden(case):
    """oracle nn_distance(cpu_twin=True) == vectors of the reference's compiled nnsearch (tf_nndistance.cpp:21-43, both directions :79-80) bit for bit;
    the CUDA-derived form (policy-2 contraction) picks the same neighbours on these clouds and differs from it by at most one rounding of the distance"""
    g = np.load(os.path.join(GOLD, "nnsearch_ref.npz"))
    a, b = g[case + "_xyz1"], g[case + "_xyz2"]
    for got, key in zip(O.nn_distance(a, b, cpu_twin=True), ("d1", "i1", "d2", "i2")):
        np.testing.assert_array_equal(_bits(got), _bits(g[case + "_" + key]))
    d1, i1, d2, i2 = O.nn_distance(a, b)
    np.testing.assert_array_equal(i1, g[case + "_i1"])
    np.testing.assert_array_equal(i2, g[case + "_i2"])
    np.testing.assert_allclose(d1, g[case + "_d1"], rtol=2.5e-7, atol=0)
    np.testing.assert_allclose(d2, g[case + "_d2"], rtol=2.5e-7, atol=0)


def test_three_nn_and_nnsearch_restatements_match_live_reference_build():
    if O.slices_lib() is None:
        pytest.skip("oracle/_ref/libslices_ref.so not built (no /root/reference on this box)")
    for seed, (b, n, m) in enumerate([(2, 700, 333), (1, 64, 3), (3, 129, 1000)]):
        x1, x2 = D.batch("S", b, n, 50 + seed), D.batch("U", b, m, 60 + seed) * 3.0
        d, i = O.three_nn(x1, x2)
        rd, ri = O.ref_three_nn(x1, x2)
        np.testing.assert_array_equal(i, ri)
        np.testing.assert_array_equal(_bits(d), _bits(rd))
        for got, want in zip(O.nn_distance(x1, x2, cpu_twin=True), O.ref_nnsearch(x1, x2)):
            np.testing.assert_array_equal(_bits(got), _bits(want))


# ---- committed oracle fixtures ------------------------------------------------------------------
def test_oracle_fixture_config1():
    g = np.load(os.path.join(GOLD, "oracle_c1_fps_ball.npz"))
    xyz = D.batch("U", 1, 4096)
    fps = O.farthest_point_sample(512, xyz)
    np.testing.assert_array_equal(fps, g["fps"])
    idx, cnt, vis = O.query_ball_point(0.2, 32, xyz, O.gather_point(xyz, fps), return_visited=True)
    np.testing.assert_array_equal(idx, g["idx"])
    np.testing.assert_array_equal(cnt, g["cnt"])
    np.testing.assert_array_equal(vis, g["visited"])


def test_oracle_fixture_duplicates_and_nn():
    g = np.load(os.path.join(GOLD, "oracle_dup_fps_ball.npz"))
    xd = D.batch("D", 2, 3000, 40)
    fd = O.farthest_point_sample(700, xd)
    np.testing.assert_array_equal(fd, g["fps"])
    bi, bc = O.query_ball_point(0.15, 16, xd, O.gather_point(xd, fd))
    np.testing.assert_array_equal(bi, g["idx"])
    np.testing.assert_array_equal(bc, g["cnt"])
    h = np.load(os.path.join(GOLD, "oracle_nn.npz"))
    d1, i1, d2, i2 = O.nn_distance(h["a"], h["b"])
    for got, key in ((d1, "d1"), (i1, "i1"), (d2, "d2"), (i2, "i2")):
        np.testing.assert_array_equal(got, h[key])
    td, ti = O.three_nn(h["a"], h["b"])
    np.testing.assert_array_equal(td, h["t_d"])
    np.testing.assert_array_equal(ti, h["t_i"])


# ---- independent NumPy brute force ----------------------------------------------------------------
@pytest.mark.parametrize("kind,n,m", [("U", 1500, 200), ("D", 1300, 300), ("U", 600, 600), ("U", 5, 9)])
def test_fps_vs_numpy(kind, n, m):
    x = D.batch(kind, 1, n, 11)
    np.testing.assert_array_equal(O.farthest_point_sample(m, x)[0], N.fps(x[0], m))


def test_fps_tie_rule_is_kmod512_then_k():
    """equal maxima at k=5 and k=513: the reference block reduction returns 513"""
    x = np.zeros((1, 1024, 3), np.float32)
    x[0, 5] = x[0, 513] = [1, 0, 0]
    assert O.farthest_point_sample(2, x)[0, 1] == 513


def test_fps_tie_rule_details():
    x = np.zeros((1, 1024, 3), np.float32)
    x[0, 1] = x[0, 513] = x[0, 5] = [1, 0, 0]
    assert O.farthest_point_sample(2, x)[0, 1] == 1         # (k mod 512, k): (1,1) < (1,513) < (5,5)


def test_fps_m_greater_than_n_repeats_zero():
    x = D.batch("U", 1, 6, 2)
    out = O.farthest_point_sample(10, x)[0]
    assert sorted(out[:6]) == list(range(6)) and (out[6:] == 0).all()


@pytest.mark.parametrize("r,ns", [(0.2, 32), (0.05, 8), (0.5, 300)])
def test_ball_query_vs_numpy(r, ns):
    x = D.batch("D", 1, 900, 5)
    q = x[:, :40].copy()
    idx, cnt = O.query_ball_point(r, ns, x, q)
    ri, rc = N.ball_query(r, ns, x[0], q[0])
    np.testing.assert_array_equal(idx[0], ri)
    np.testing.assert_array_equal(cnt[0], rc)


def test_three_nn_and_nn_distance_vs_numpy():
    a = D.batch("U", 1, 300, 1)[0]
    b = D.batch("U", 1, 77, 2)[0]
    d, i = O.three_nn(a[None], b[None])
    rd, ri = N.three_nn(a, b)
    np.testing.assert_array_equal(i[0], ri)
    np.testing.assert_array_equal(d[0], rd)
    d1, i1, d2, i2 = O.nn_distance(a[None], b[None])
    e1, j1, e2, j2 = N.nn_bruteforce(a, b)
    np.testing.assert_array_equal(i1[0], j1)
    np.testing.assert_array_equal(i2[0], j2)
    np.testing.assert_allclose(d1[0], e1, rtol=1e-5)
    np.testing.assert_allclose(d2[0], e2, rtol=1e-5)
    # CPU twin (unfused) agrees with the GPU twin up to the last bit of the distance
    c1, k1, c2, k2 = O.nn_distance(a[None], b[None], cpu_twin=True)
    np.testing.assert_allclose(c1, d1, rtol=3e-7)
    assert (k1 == i1).mean() > 0.99


def test_three_nn_fewer_than_three_candidates():
    a = D.batch("U", 1, 10, 1)
    b = D.batch("U", 1, 2, 2)
    d, i = O.three_nn(a, b)
    assert np.isinf(d[0, :, 2]).all() and (i[0, :, 2] == 0).all()


# ---- gradient consistency, as the reference's own op tests do (tolerance 1e-4) -----------------------
def test_group_point_grad_is_adjoint():
    """tf_grouping_op_test.py:9-27: compute_gradient_error(points -> group_point(points, ball idx)) < 1e-4.
    group_point is linear in `points`, so gradient consistency == adjointness <G(p), g> = <p, G^T(g)>."""
    rng = np.random.default_rng(0)
    pts = rng.random((4, 256, 8)).astype(np.float32)
    xyz = rng.random((4, 256, 3)).astype(np.float32)
    idx, _ = O.query_ball_point(0.3, 64, xyz, xyz[:, :32].copy())
    g = rng.standard_normal((4, 32, 64, 8)).astype(np.float32)
    lhs = float((O.group_point(pts, idx).astype(np.float64) * g).sum())
    rhs = float((pts.astype(np.float64) * O.group_point_grad(pts, idx, g)).sum())
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))


def test_three_interpolate_grad_is_adjoint():
    """tf_interpolate_op_test.py:9-21: (1,8,16) -> (1,128,16), weights 1/3, error < 1e-4"""
    rng = np.random.default_rng(1)
    pts = rng.random((1, 8, 16)).astype(np.float32)
    _, idx = O.three_nn(rng.random((1, 128, 3)).astype(np.float32), rng.random((1, 8, 3)).astype(np.float32))
    w = np.full((1, 128, 3), 1 / 3, np.float32)
    g = rng.standard_normal((1, 128, 16)).astype(np.float32)
    lhs = float((O.three_interpolate(pts, idx, w).astype(np.float64) * g).sum())
    rhs = float((pts.astype(np.float64) * O.three_interpolate_grad(pts, idx, w, g)).sum())
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))


def test_nn_distance_grad_matches_finite_difference():
    rng = np.random.default_rng(2)
    a = rng.standard_normal((1, 20, 3)).astype(np.float32)
    b = rng.standard_normal((1, 15, 3)).astype(np.float32)
    d1, i1, d2, i2 = O.nn_distance(a, b)
    g1, g2 = O.nn_distance_grad(a, b, np.ones_like(d1), i1, np.ones_like(d2), i2)
    f = lambda a_, b_: float(N.nn_bruteforce(a_[0], b_[0])[0].sum() + N.nn_bruteforce(a_[0], b_[0])[2].sum())
    eps = 1e-3
    for (arr, grad, which) in ((a, g1, 0), (b, g2, 1)):
        for j in (0, 3, 7):
            for l in range(3):
                p = arr.copy(); p[0, j, l] += eps
                m = arr.copy(); m[0, j, l] -= eps
                fd = (f(p, b) - f(m, b)) / (2 * eps) if which == 0 else (f(a, p) - f(a, m)) / (2 * eps)
                assert abs(fd - grad[0, j, l]) < 2e-2 * max(1.0, abs(fd))


# ---- invariants (property tests) ---------------------------------------------------------------------
@settings(max_examples=25, deadline=None)
@given(st.integers(1, 700), st.integers(1, 60), st.integers(0, 10_000))
def test_fps_invariants(n, m, seed):
    x = D.batch("D", 1, n, seed)
    idx = O.farthest_point_sample(m, x)[0]
    assert idx[0] == 0 and idx.min() >= 0 and idx.max() < n
    # every pick maximises the min-distance to the already chosen set
    temp = np.full(n, 1e38, np.float32)
    for j in range(1, m):
        temp = np.minimum(temp, N.dist2_cuda(x[0, idx[j - 1]][None], x[0]))
        assert temp[idx[j]] == temp.max()


@settings(max_examples=25, deadline=None)
@given(st.integers(1, 500), st.integers(1, 40), st.integers(1, 50), st.floats(0.01, 0.8), st.integers(0, 10_000))
def test_ball_query_invariants(n, m, ns, r, seed):
    x = D.batch("U", 1, n, seed)
    q = D.batch("U", 1, m, seed + 1)
    idx, cnt, vis = O.query_ball_point(r, ns, x, q, return_visited=True)
    for j in range(m):
        c = cnt[0, j]
        row = idx[0, j]
        assert 0 <= c <= ns
        assert (np.diff(row[:c]) > 0).all()                         # first-cnt strictly ascending
        if c:
            d = np.sqrt(N.dist2_cuda(x[0, row[:c]], q[0, j][None]))
            assert (d < np.float32(r)).all() and (row[c:] == row[0]).all()     # all inside, tail = first hit
        else:
            assert (row == 0).all()
        assert vis[0, j] == (row[c - 1] + 1 if c == ns else n)      # L of SURVEY 8(d)


def test_prob_sample_matches_float64_cdf():
    rng = np.random.default_rng(4)
    w = rng.random((3, 20000)).astype(np.float32)
    r = rng.random((3, 50)).astype(np.float32)
    cs = O.cumsum(w)
    np.testing.assert_allclose(cs, np.cumsum(w.astype(np.float64), 1), rtol=2e-6)
    out = O.prob_sample(w, r)
    for i in range(3):
        q = r[i] * cs[i, -1]
        np.testing.assert_array_equal(out[i], np.searchsorted(cs[i], q, side="left").clip(0, 19999))


def test_selection_sort_and_knn():
    rng = np.random.default_rng(5)
    dist = rng.integers(0, 9, size=(2, 5, 40)).astype(np.float32)
    oi, od = O.select_top_k(7, dist)
    for b in range(2):
        for j in range(5):
            assert (np.diff(od[b, j, :7]) >= 0).all()
            np.testing.assert_array_equal(np.sort(oi[b, j]), np.arange(40))           # a permutation
            np.testing.assert_array_equal(dist[b, j][oi[b, j]], od[b, j])
            assert od[b, j, 6] <= od[b, j, 7:].min()


def test_openmp_mode_does_not_change_results():
    """oracle_set_mt (bench.py's all-cores baseline): same bits with and without OpenMP over scene x query"""
    rng = np.random.default_rng(8)
    xyz = rng.random((3, 700, 3)).astype(np.float32)
    q = O.gather_point(xyz, O.farthest_point_sample(90, xyz))
    pts = rng.standard_normal((3, 700, 5)).astype(np.float32)
    res = []
    for on in (False, True):
        O.set_mt(on)
        try:
            idx, cnt = O.query_ball_point(0.2, 16, xyz, q)
            d, i3 = O.three_nn(xyz, q)
            gp = O.group_point(pts, idx)
            gg = O.group_point_grad(pts, idx, gp)
            w = np.full(d.shape, 1 / 3, np.float32)
            ip = O.three_interpolate(pts[:, :90].copy(), i3, w)
            ig = O.three_interpolate_grad(pts[:, :90].copy(), i3, w, ip)
            res.append((idx, cnt, d, i3, gp, gg, ip, ig))
        finally:
            O.set_mt(False)
    for a, b_ in zip(*res):
        np.testing.assert_array_equal(a, b_)
