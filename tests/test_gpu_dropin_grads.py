"""The reference-signature GRADIENT symbols of the C ABI, called the way a TensorFlow maintainer would bind them:
raw device pointers, the reference launchers' argument order, a stream -- through ctypes, not through the Python ops (which since r04
take the inverse-list gathers instead: gspn_amd/tf_grouping.py, tf_interpolate.py, tf_sampling.py).  VERDICT r04 item 3a.

  gspn_grouppoint_grad        <-> groupPointGradLauncher        tf_grouping.cpp:203, tf_grouping_g.cu:66-83,198-202
  gspn_scatteraddpoint        <-> scatteraddpointLauncher       tf_sampling.cpp:150, tf_sampling_g.cu:183-192,209-211
  gspn_threeinterpolate_grad  <-> threeinterpolate_grad_cpu     tf_interpolate.cpp:131-153
  gspn_nmdistance_grad        <-> NmDistanceGradKernelLauncher  tf_nndistance.cpp:208, tf_nndistance_g.cu:132-157
  gspn_groupmaxpool_grad      <-> groupMaxpoolGradLauncher      tf_grouping.cpp:277, tf_grouping_g.cu:112-134

Indices are CONTENDED on purpose (padded ball-query rows repeat one index up to nsample times; a handful of hot points collect thousands of
terms): atomic adds have no order, so the bar is 1e-5 relative to the magnitude a sum went through -- against the oracle's sequential loops,
against the reference's own compiled sources where oracle/_ref has them (R.group_point_grad / R.gather_point_grad launched from the hipcc build
of tf_grouping_g.cu / tf_sampling_g.cu, O.ref_three_interpolate_grad from g++'s interpolate.cpp), and EXACTLY where an index occurs once.
Outputs are pre-filled with garbage: the ABI promises to zero them itself (the reference's cudaMemset, tf_grouping.cpp:234, tf_sampling.cpp:174)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import oracle as O
from oracle import ref_hip as R

pytestmark = pytest.mark.gpu

P = ctypes.c_void_p


def _p(t):
    return P(t.data_ptr())


def _st():
    return P(torch.cuda.current_stream().cuda_stream)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def garbage(*shape):
    return torch.full(shape, float("nan"), dtype=torch.float32, device="cuda")


def close(got, ref, scale, tol=1e-5):
    """|got - ref| <= tol * scale elementwise; scale = sum of |terms| that went into each output element (the magnitude the sum passed through)"""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    err = np.abs(got - ref)
    bound = tol * np.maximum(scale, 1e-30)
    assert np.isfinite(got).all()
    assert (err <= bound + 1e-12).all(), "max err/bound %.3g" % float((err / (bound + 1e-12)).max())


def contended_group_idx(rng, b, n, m, ns):
    """ball-query-like rows: a few distinct neighbours, the rest padded with the first hit (tf_grouping_g.cu:29-32), plus hot points"""
    idx = np.empty((b, m, ns), np.int32)
    hot = rng.integers(0, n, size=8)
    for i in range(b):
        for j in range(m):
            k = int(rng.integers(1, ns + 1))
            row = np.sort(rng.choice(n, size=k, replace=False)).astype(np.int32)
            if rng.random() < 0.3:
                row[0] = hot[rng.integers(0, 8)]
            idx[i, j, :k] = row
            idx[i, j, k:] = row[0]
    return idx


@pytest.mark.parametrize("b,n,c,m,ns", [(2, 500, 3, 128, 32), (3, 2048, 64, 256, 32), (1, 64, 131, 40, 16), (2, 4096, 6, 512, 64)])
def test_grouppoint_grad_symbol(b, n, c, m, ns):
    from gspn_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(b * 1000 + c)
    idx = contended_group_idx(rng, b, n, m, ns)
    go = rng.standard_normal((b, m, ns, c)).astype(np.float32)
    g = garbage(b, n, c)
    assert lib.gspn_grouppoint_grad(b, n, c, m, ns, _p(dev(go)), _p(dev(idx)), _p(g), _st()) == 0
    torch.cuda.synchronize()
    got = g.cpu().numpy()
    ref = O.group_point_grad(np.zeros((b, n, c), np.float32), idx, go)
    scale = O.group_point_grad(np.zeros((b, n, c), np.float32), idx, np.abs(go)).astype(np.float64)
    close(got, ref, scale)
    # exactly equal where a point collects a single term (and exactly zero where it collects none)
    cnt = np.zeros((b, n), np.int64)
    for i in range(b):
        np.add.at(cnt[i], idx[i].reshape(-1), 1)
    np.testing.assert_array_equal(got[cnt <= 1], ref[cnt <= 1])
    if R.available():                                   # the reference's own kernel (hipcc build of tf_grouping_g.cu), same unordered atomics
        rg = R.group_point_grad(n, dev(idx), dev(go)).cpu().numpy()
        close(got, rg, scale)
        np.testing.assert_array_equal(got[cnt <= 1], rg[cnt <= 1])


@pytest.mark.parametrize("b,n,m", [(2, 300, 1000), (8, 32768, 2048), (1, 5, 64)])
def test_scatteraddpoint_symbol(b, n, m):
    from gspn_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(n)
    idx = rng.integers(0, n, size=(b, m)).astype(np.int32)
    idx[:, : m // 4] = idx[:, :1]                        # a quarter of the samples hit one point
    og = rng.standard_normal((b, m, 3)).astype(np.float32)
    g = garbage(b, n, 3)
    assert lib.gspn_scatteraddpoint(b, n, m, _p(dev(og)), _p(dev(idx)), _p(g), _st()) == 0
    torch.cuda.synchronize()
    got = g.cpu().numpy()
    ref = O.gather_point_grad(np.zeros((b, n, 3), np.float32), idx, og)
    scale = O.gather_point_grad(np.zeros((b, n, 3), np.float32), idx, np.abs(og)).astype(np.float64)
    close(got, ref, scale)
    cnt = np.zeros((b, n), np.int64)
    for i in range(b):
        np.add.at(cnt[i], idx[i], 1)
    np.testing.assert_array_equal(got[cnt <= 1], ref[cnt <= 1])
    if R.available():
        rg = R.gather_point_grad(n, dev(idx), dev(og)).cpu().numpy()
        close(got, rg, scale)
        np.testing.assert_array_equal(got[cnt <= 1], rg[cnt <= 1])


@pytest.mark.parametrize("b,n,c,m", [(2, 512, 64, 128), (1, 128, 16, 8), (3, 4096, 37, 50), (2, 2048, 256, 512)])
def test_threeinterpolate_grad_symbol(b, n, c, m):
    """argument order of threeinterpolate_grad_cpu(b, n, c, m, grad_out, idx, weight, grad_points)"""
    from gspn_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(c)
    idx = rng.integers(0, m, size=(b, n, 3)).astype(np.int32)
    idx[:, ::7, :] = idx[:, :1, :1]                      # hot sparse points; also i1 == i2 == i3 rows
    w = rng.random((b, n, 3)).astype(np.float32)
    w /= w.sum(2, keepdims=True)
    go = rng.standard_normal((b, n, c)).astype(np.float32)
    g = garbage(b, m, c)
    assert lib.gspn_threeinterpolate_grad(b, n, c, m, _p(dev(go)), _p(dev(idx)), _p(dev(w)), _p(g), _st()) == 0
    torch.cuda.synchronize()
    got = g.cpu().numpy()
    pts = np.zeros((b, m, c), np.float32)
    ref = O.three_interpolate_grad(pts, idx, w, go)
    scale = O.three_interpolate_grad(pts, idx, w, np.abs(go)).astype(np.float64)
    close(got, ref, scale)
    if O.ref_lib() is not None:                          # the reference's own compiled loop (g++ -O2 on interpolate.cpp)
        close(got, O.ref_three_interpolate_grad(pts, idx, w, go), scale)


@pytest.mark.parametrize("b,n,m", [(4, 512, 512), (2, 16384, 1024), (3, 100, 7)])
def test_nmdistance_grad_symbol(b, n, m):
    from gspn_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(m)
    a = rng.standard_normal((b, n, 3)).astype(np.float32)
    c = rng.standard_normal((b, m, 3)).astype(np.float32)
    d1, i1, d2, i2 = O.nn_distance(a, c)
    g1 = rng.standard_normal((b, n)).astype(np.float32)
    g2 = rng.standard_normal((b, m)).astype(np.float32)
    ga, gc = garbage(b, n, 3), garbage(b, m, 3)
    assert lib.gspn_nmdistance_grad(b, n, _p(dev(a)), m, _p(dev(c)), _p(dev(g1)), _p(dev(i1)), _p(dev(g2)), _p(dev(i2)), _p(ga), _p(gc), _st()) == 0
    torch.cuda.synchronize()
    r1, r2 = O.nn_distance_grad(a, c, g1, i1, g2, i2)
    # magnitude of the sums: the same loop on |g| gives sum |g| * |p1 - p2| only up to sign patterns -- bound it from above instead
    s1 = np.zeros((b, n, 3)); s2 = np.zeros((b, m, 3))
    for i in range(b):
        t = 2 * np.abs(g1[i])[:, None].astype(np.float64) * np.abs(a[i] - c[i][i1[i]])
        s1[i] += t
        np.add.at(s2[i], i1[i], t)
        t = 2 * np.abs(g2[i])[:, None].astype(np.float64) * np.abs(c[i] - a[i][i2[i]])
        s2[i] += t
        np.add.at(s1[i], i2[i], t)
    close(ga.cpu().numpy(), r1, s1)
    close(gc.cpu().numpy(), r2, s2)


def test_groupmaxpool_grad_symbol():
    from gspn_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(5)
    b, n, c, m, ns = 2, 700, 24, 90, 12
    pts = (np.round(rng.standard_normal((b, n, c)) * 2) / 2).astype(np.float32)        # tied maxima
    idx = rng.integers(0, n, size=(b, m, ns)).astype(np.int32)
    out, mi = O.group_maxpool(pts, idx)
    go = rng.standard_normal((b, m, c)).astype(np.float32)
    g = garbage(b, n, c)
    assert lib.gspn_groupmaxpool_grad(b, n, c, m, _p(dev(go)), _p(dev(mi)), _p(g), _st()) == 0
    torch.cuda.synchronize()
    ref = O.group_maxpool_grad(pts, mi, go)
    scale = O.group_maxpool_grad(pts, mi, np.abs(go)).astype(np.float64)
    close(g.cpu().numpy(), ref, scale)


# ---- ABI 9: the same launchers with a workspace -- fixed-order gathers through inverse lists built inside the call (csrc/dropin_ws.hip) -------------------
def _ws(nbytes):
    assert nbytes >= 0
    return torch.randint(0, 255, (max(int(nbytes), 16),), dtype=torch.uint8, device="cuda")       # garbage: the call must not rely on its content


@pytest.mark.parametrize("b,n,c,m,ns", [(2, 500, 3, 128, 32), (3, 2048, 64, 256, 32), (1, 64, 131, 40, 16), (2, 4096, 6, 512, 64), (8, 2048, 128, 512, 32)])
def test_grouppoint_grad_ws_symbol_is_the_sequential_loop_bit_for_bit(b, n, c, m, ns):
    """groupPointGradLauncher's arguments + `void* ws`: sums in ascending grouped position = the oracle's sequential loop, exactly, on contended indices"""
    from gspn_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(b * 1000 + c)
    idx = contended_group_idx(rng, b, n, m, ns) if b * m < 3000 else rng.integers(0, n, size=(b, m, ns)).astype(np.int32)
    go = rng.standard_normal((b, m, ns, c)).astype(np.float32)
    ref = O.group_point_grad(np.zeros((b, n, c), np.float32), idx, go)
    tgo, tidx = dev(go), dev(idx)
    for _ in range(2):
        g, ws = garbage(b, n, c), _ws(lib.gspn_grouppoint_grad_ws_bytes(b, n, c, m, ns))
        assert lib.gspn_grouppoint_grad_ws(b, n, c, m, ns, _p(tgo), _p(tidx), _p(g), _p(ws), _st()) == 0
        torch.cuda.synchronize()
        np.testing.assert_array_equal(g.cpu().numpy(), ref)


@pytest.mark.parametrize("b,n,m", [(2, 300, 1000), (8, 32768, 2048), (1, 5, 64)])
def test_scatteraddpoint_ws_symbol_is_the_sequential_loop_bit_for_bit(b, n, m):
    from gspn_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(n)
    idx = rng.integers(0, n, size=(b, m)).astype(np.int32)
    idx[:, : m // 4] = idx[:, :1]
    og = rng.standard_normal((b, m, 3)).astype(np.float32)
    g, ws = garbage(b, n, 3), _ws(lib.gspn_scatteraddpoint_ws_bytes(b, n, m))
    assert lib.gspn_scatteraddpoint_ws(b, n, m, _p(dev(og)), _p(dev(idx)), _p(g), _p(ws), _st()) == 0
    torch.cuda.synchronize()
    np.testing.assert_array_equal(g.cpu().numpy(), O.gather_point_grad(np.zeros((b, n, 3), np.float32), idx, og))


@pytest.mark.parametrize("b,n,c,m", [(2, 512, 64, 128), (1, 128, 16, 8), (3, 4096, 37, 50), (2, 2048, 256, 512), (8, 32768, 64, 2048)])
def test_threeinterpolate_grad_ws_symbol_is_the_reference_loop_bit_for_bit(b, n, c, m):
    """threeinterpolate_grad_cpu's arguments + `void* ws`: bit-identical to the reference's OWN compiled loop (g++ -O2 on interpolate.cpp, tf_interpolate.cpp:131-153)
    where oracle/_ref has it, and to the restatement -- incl. the dense feature-propagation level 8 x 32768 -> 2048 at c = 64"""
    from gspn_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(c)
    idx = rng.integers(0, m, size=(b, n, 3)).astype(np.int32)
    idx[:, ::7, :] = idx[:, :1, :1]
    w = rng.random((b, n, 3)).astype(np.float32)
    w /= w.sum(2, keepdims=True)
    go = rng.standard_normal((b, n, c)).astype(np.float32)
    g, ws = garbage(b, m, c), _ws(lib.gspn_threeinterpolate_grad_ws_bytes(b, n, c, m))
    assert lib.gspn_threeinterpolate_grad_ws(b, n, c, m, _p(dev(go)), _p(dev(idx)), _p(dev(w)), _p(g), _p(ws), _st()) == 0
    torch.cuda.synchronize()
    got = g.cpu().numpy()
    pts = np.zeros((b, m, c), np.float32)
    np.testing.assert_array_equal(got, O.three_interpolate_grad(pts, idx, w, go))
    if O.ref_lib() is not None:
        np.testing.assert_array_equal(got, O.ref_three_interpolate_grad(pts, idx, w, go))


@pytest.mark.parametrize("b,n,m", [(4, 512, 512), (2, 16384, 1024), (3, 100, 7)])
def test_nmdistance_grad_ws_symbol_is_the_cpu_twin_s_order_bit_for_bit(b, n, m):
    """NmDistanceGradKernelLauncher's arguments + `void* ws`: == gspn_nmdistance_grad_csr on lists built by the caller (what the op API runs beyond the LDS size,
    bit-exact against the sequential CPU twin tf_nndistance.cpp:126-163), and within 1e-5 of the oracle's loop"""
    from gspn_amd import _lib as L
    from gspn_amd.invlists import inverse_lists
    lib = L.lib()
    rng = np.random.default_rng(m)
    a = rng.standard_normal((b, n, 3)).astype(np.float32)
    c = rng.standard_normal((b, m, 3)).astype(np.float32)
    d1, i1, d2, i2 = O.nn_distance(a, c)
    g1 = rng.standard_normal((b, n)).astype(np.float32)
    g2 = rng.standard_normal((b, m)).astype(np.float32)
    ta, tc, tg1, tg2, ti1, ti2 = dev(a), dev(c), dev(g1), dev(g2), dev(i1), dev(i2)
    ga, gc, ws = garbage(b, n, 3), garbage(b, m, 3), _ws(lib.gspn_nmdistance_grad_ws_bytes(b, n, m))
    assert lib.gspn_nmdistance_grad_ws(b, n, _p(ta), m, _p(tc), _p(tg1), _p(ti1), _p(tg2), _p(ti2), _p(ga), _p(gc), _p(ws), _st()) == 0
    o1, f1 = inverse_lists(ti1, m)
    o2, f2 = inverse_lists(ti2, n)
    ha, hc = garbage(b, n, 3), garbage(b, m, 3)
    assert lib.gspn_nmdistance_grad_csr(b, n, _p(ta), m, _p(tc), _p(tg1), _p(ti1), _p(tg2), _p(ti2), _p(o1), _p(f1), _p(o2), _p(f2), _p(ha), _p(hc), _st()) == 0
    torch.cuda.synchronize()
    assert torch.equal(ga, ha) and torch.equal(gc, hc)
    r1, r2 = O.nn_distance_grad(a, c, g1, i1, g2, i2)
    np.testing.assert_allclose(ga.cpu().numpy(), r1, rtol=1e-5, atol=1e-5 * float(np.abs(r1).max()))
    np.testing.assert_allclose(gc.cpu().numpy(), r2, rtol=1e-5, atol=1e-5 * float(np.abs(r2).max()))


def test_ws_symbols_reject_what_the_reference_ops_reject_and_size_their_workspace():
    from gspn_amd import _lib as L
    lib = L.lib()
    assert lib.gspn_grouppoint_grad_ws_bytes(8, 2048, 64, 512, 32) == 4 * (8 * 512 * 32 + (8 * 2049 + 15) // 16 * 16 + (int(lib.gspn_inverse_lists_work_ints(8, 512 * 32, 2048)) + 15) // 16 * 16)
    assert lib.gspn_grouppoint_grad_ws_bytes(1, 0, 3, 4, 4) == -1 and lib.gspn_threeinterpolate_grad_ws_bytes(1, 4, 3, 0) == -1 and lib.gspn_nmdistance_grad_ws_bytes(1, 0, 4) == -1
    g = garbage(2, 10, 4)
    assert lib.gspn_grouppoint_grad_ws(2, 10, 4, 5, 3, None, None, _p(g), None, _st()) == -1            # null pointers
    assert lib.gspn_grouppoint_grad_ws(2, 10, 4, 0, 3, None, None, _p(g), None, _st()) == 0             # nothing grouped: zero gradient
    torch.cuda.synchronize()
    assert float(g.abs().max()) == 0.0
    assert lib.gspn_grouppoint_grad_ws(0, 10, 4, 5, 3, None, None, None, None, _st()) == 0
