"""NumPy front-end of the CPU oracle (oracle/gspn_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py.  gspn_amd/ never imports this module.

Every function takes/returns C-contiguous numpy arrays (float32 / int32) with the argument
order of the reference Python wrappers (tf_ops/*/tf_*.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    so = os.path.join(_HERE, "libgspn_oracle.so")
    src = os.path.join(_HERE, "gspn_oracle.c")
    stale = (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src)
    ref_missing = os.path.isdir("/root/reference") and not all(
        os.path.exists(os.path.join(_HERE, "_ref", f)) for f in ("libinterp_ref.so", "libslices_ref.so"))
    if force or stale or ref_missing:
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


class use_policy:
    """context manager: the oracle built with GSPN_DIST_POLICY = policy (0 unfused, 1 fma(c,c,fma(b,b,a*a)), 2 = the default, 3 = fma(a,a,b*b)+c*c: hipcc's own contraction of the reference expression)
    behind every function of this module -- tests/test_gpu_policy.py"""

    def __init__(self, policy):
        self.policy = int(policy)

    def __enter__(self):
        global _LIB
        self.prev = _LIB
        if self.policy != 2:
            so = os.path.join(_HERE, "libgspn_oracle_p%d.so" % self.policy)
            if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "gspn_oracle.c")):
                subprocess.check_call(["make", "-C", _HERE, "policies"], stdout=subprocess.DEVNULL)
            _LIB = ctypes.CDLL(so)
        else:
            _LIB = None
            lib()
        assert _LIB.oracle_dist_policy() == self.policy
        return self

    def __exit__(self, *a):
        global _LIB
        _LIB = self.prev


def ref_lib():
    """oracle/_ref/libinterp_ref.so, compiled from the reference's interpolate.cpp (or None)."""
    global _REF
    if _REF is None:
        p = os.path.join(_HERE, "_ref", "libinterp_ref.so")
        if not os.path.exists(p):
            build()
        if os.path.exists(p):
            _REF = ctypes.CDLL(p)
    return _REF


_SLICES = None


def slices_lib():
    """oracle/_ref/libslices_ref.so: the reference's own `threenn_cpu` (tf_interpolate.cpp:60-103) and `nnsearch`
    (tf_nndistance.cpp:21-43), cut out of /root/reference at build time and compiled by g++ -O2 (oracle/Makefile: slices); or None."""
    global _SLICES
    if _SLICES is None:
        p = os.path.join(_HERE, "_ref", "libslices_ref.so")
        if not os.path.exists(p):
            build()
        if os.path.exists(p):
            _SLICES = ctypes.CDLL(p)
    return _SLICES


def _fp(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_f)


def _ip(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_i)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def set_mt(on):
    """bench.py's cpu_baseline "all cores" leg: OpenMP over scene x query inside the oracle loops (results unchanged)"""
    lib().oracle_set_mt(int(bool(on)))


def dist_policy():
    return lib().oracle_dist_policy()


def dist2(p, q):
    p, q = _f32(p), _f32(q)
    cnt = p.shape[0]
    oc = np.empty(cnt, np.float32)
    oh = np.empty(cnt, np.float32)
    lib().oracle_dist2(cnt, _fp(p), _fp(q), _fp(oc), _fp(oh))
    return oc, oh


# ---- tf_sampling -------------------------------------------------------------------------
def farthest_point_sample_temp(npoint, inp):
    """(idx, temp): the indices and the reference kernel's scratch after the run (min squared distance of every point to the chosen set)"""
    inp = _f32(inp)
    b, n, _ = inp.shape
    out = np.zeros((b, npoint), np.int32)
    temp = np.empty((b, n), np.float32)
    lib().oracle_farthest_point_sample_temp(b, n, npoint, _fp(inp), _ip(out), _fp(temp))
    return out, temp


def farthest_point_sample(npoint, inp, mt=False):
    inp = _f32(inp)
    b, n, _ = inp.shape
    out = np.zeros((b, npoint), np.int32)
    fn = lib().oracle_farthest_point_sample_mt if mt else lib().oracle_farthest_point_sample
    fn(b, n, npoint, _fp(inp), _ip(out))
    return out


def gather_point(inp, idx):
    inp, idx = _f32(inp), _i32(idx)
    b, n, _ = inp.shape
    m = idx.shape[1]
    out = np.empty((b, m, 3), np.float32)
    lib().oracle_gather_point(b, n, m, _fp(inp), _ip(idx), _fp(out))
    return out


def gather_point_grad(inp, idx, out_g):
    inp, idx, out_g = _f32(inp), _i32(idx), _f32(out_g)
    b, n, _ = inp.shape
    m = idx.shape[1]
    g = np.empty((b, n, 3), np.float32)
    lib().oracle_gather_point_grad(b, n, m, _fp(out_g), _ip(idx), _fp(g))
    return g


def cumsum(inp):
    inp = _f32(inp)
    b, n = inp.shape
    out = np.empty((b, n), np.float32)
    lib().oracle_cumsum(b, n, _fp(inp), _fp(out))
    return out


def prob_sample(inp, inpr):
    inp, inpr = _f32(inp), _f32(inpr)
    b, n = inp.shape
    m = inpr.shape[1]
    cs = cumsum(inp)
    out = np.empty((b, m), np.int32)
    lib().oracle_binary_search(b, n, m, _fp(cs), _fp(inpr), _ip(out))
    return out


# ---- tf_grouping -------------------------------------------------------------------------
def query_ball_point(radius, nsample, xyz1, xyz2, return_visited=False, mt=False):
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = np.empty((b, m, nsample), np.int32)
    cnt = np.empty((b, m), np.int32)
    vis = np.empty((b, m), np.int32)
    fn = lib().oracle_query_ball_point_mt if mt else lib().oracle_query_ball_point
    fn(b, n, m, ctypes.c_float(radius), nsample, _fp(xyz1), _fp(xyz2), _ip(idx), _ip(cnt), _ip(vis))
    if return_visited:
        return idx, cnt, vis
    return idx, cnt


def group_point(points, idx):
    points, idx = _f32(points), _i32(idx)
    b, n, c = points.shape
    _, m, ns = idx.shape
    out = np.empty((b, m, ns, c), np.float32)
    lib().oracle_group_point(b, n, c, m, ns, _fp(points), _ip(idx), _fp(out))
    return out


def group_point_grad(points, idx, grad_out):
    points, idx, grad_out = _f32(points), _i32(idx), _f32(grad_out)
    b, n, c = points.shape
    _, m, ns = idx.shape
    g = np.empty((b, n, c), np.float32)
    lib().oracle_group_point_grad(b, n, c, m, ns, _fp(grad_out), _ip(idx), _fp(g))
    return g


def group_maxpool(points, idx):
    points, idx = _f32(points), _i32(idx)
    b, n, c = points.shape
    _, m, ns = idx.shape
    out = np.empty((b, m, c), np.float32)
    mi = np.empty((b, m, c), np.int32)
    lib().oracle_group_maxpool(b, n, c, m, ns, _fp(points), _ip(idx), _fp(out), _ip(mi))
    return out, mi


def group_maxpool_grad(points, max_idx, grad_out):
    points, max_idx, grad_out = _f32(points), _i32(max_idx), _f32(grad_out)
    b, n, c = points.shape
    m = max_idx.shape[1]
    g = np.empty((b, n, c), np.float32)
    lib().oracle_group_maxpool_grad(b, n, c, m, _fp(grad_out), _ip(max_idx), _fp(g))
    return g


def select_top_k(k, dist):
    dist = _f32(dist)
    b, m, n = dist.shape
    outi = np.empty((b, m, n), np.int32)
    out = np.empty((b, m, n), np.float32)
    lib().oracle_selection_sort(b, n, m, k, _fp(dist), _ip(outi), _fp(out))
    return outi, out


def knn_point(k, xyz1, xyz2):
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, c = xyz1.shape
    m = xyz2.shape[1]
    dist = np.empty((b, m, n), np.float32)
    lib().oracle_knn_dist(b, n, c, m, _fp(xyz1), _fp(xyz2), _fp(dist))
    outi, out = select_top_k(k, dist)
    return np.ascontiguousarray(out[:, :, :k]), np.ascontiguousarray(outi[:, :, :k])


# ---- tf_interpolate ----------------------------------------------------------------------
def three_nn(xyz1, xyz2):
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = np.empty((b, n, 3), np.float32)
    idx = np.empty((b, n, 3), np.int32)
    lib().oracle_three_nn(b, n, m, _fp(xyz1), _fp(xyz2), _fp(dist), _ip(idx))
    return dist, idx


def three_interpolate(points, idx, weight):
    points, idx, weight = _f32(points), _i32(idx), _f32(weight)
    b, m, c = points.shape
    n = idx.shape[1]
    out = np.empty((b, n, c), np.float32)
    lib().oracle_three_interpolate(b, m, c, n, _fp(points), _ip(idx), _fp(weight), _fp(out))
    return out


def three_interpolate_grad(points, idx, weight, grad_out):
    points, idx, weight, grad_out = _f32(points), _i32(idx), _f32(weight), _f32(grad_out)
    b, m, c = points.shape
    n = idx.shape[1]
    g = np.empty((b, m, c), np.float32)
    lib().oracle_three_interpolate_grad(b, n, c, m, _fp(grad_out), _ip(idx), _fp(weight), _fp(g))
    return g


# the REAL reference code (oracle/_ref), only where it exists
def ref_three_interpolate(points, idx, weight):
    r = ref_lib()
    points, idx, weight = _f32(points), _i32(idx), _f32(weight)
    b, m, c = points.shape
    n = idx.shape[1]
    out = np.empty((b, n, c), np.float32)
    getattr(r, "_Z15interpolate_cpuiiiiPKfPKiS0_Pf")(b, m, c, n, _fp(points), _ip(idx), _fp(weight), _fp(out))
    return out


def ref_three_interpolate_grad(points, idx, weight, grad_out):
    r = ref_lib()
    points, idx, weight, grad_out = _f32(points), _i32(idx), _f32(weight), _f32(grad_out)
    b, m, c = points.shape
    n = idx.shape[1]
    g = np.zeros((b, m, c), np.float32)  # caller memset, tf_interpolate.cpp:258
    getattr(r, "_Z20interpolate_grad_cpuiiiiPKfPKiS0_Pf")(b, n, c, m, _fp(grad_out), _ip(idx), _fp(weight), _fp(g))
    return g


def ref_three_nn(xyz1, xyz2):
    """the reference's compiled threenn_cpu (tf_interpolate.cpp:60-103) -> (dist (b,n,3) squared, idx (b,n,3))"""
    r = slices_lib()
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = np.empty((b, n, 3), np.float32)
    idx = np.empty((b, n, 3), np.int32)
    r.threenn_cpu(b, n, m, _fp(xyz1), _fp(xyz2), _fp(dist), _ip(idx))
    return dist, idx


def ref_nnsearch(xyz1, xyz2):
    """the reference's compiled nnsearch (tf_nndistance.cpp:21-43), run in both directions as NnDistanceOp::Compute does
    (tf_nndistance.cpp:79-80) -> (dist1, idx1, dist2, idx2)"""
    r = slices_lib()
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1 = np.empty((b, n), np.float32)
    i1 = np.empty((b, n), np.int32)
    d2 = np.empty((b, m), np.float32)
    i2 = np.empty((b, m), np.int32)
    r.nnsearch_ref(b, n, m, _fp(xyz1), _fp(xyz2), _fp(d1), _ip(i1))
    r.nnsearch_ref(b, m, n, _fp(xyz2), _fp(xyz1), _fp(d2), _ip(i2))
    return d1, i1, d2, i2


# ---- tf_nndistance -----------------------------------------------------------------------
def nn_distance(xyz1, xyz2, cpu_twin=False):
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1 = np.empty((b, n), np.float32)
    i1 = np.empty((b, n), np.int32)
    d2 = np.empty((b, m), np.float32)
    i2 = np.empty((b, m), np.int32)
    fn = lib().oracle_nn_distance_cputwin if cpu_twin else lib().oracle_nn_distance
    fn(b, n, _fp(xyz1), m, _fp(xyz2), _fp(d1), _ip(i1), _fp(d2), _ip(i2))
    return d1, i1, d2, i2


def nn_distance_grad(xyz1, xyz2, grad_dist1, idx1, grad_dist2, idx2):
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    grad_dist1, grad_dist2 = _f32(grad_dist1), _f32(grad_dist2)
    idx1, idx2 = _i32(idx1), _i32(idx2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = np.empty((b, n, 3), np.float32)
    g2 = np.empty((b, m, 3), np.float32)
    lib().oracle_nn_distance_grad(b, n, _fp(xyz1), m, _fp(xyz2), _fp(grad_dist1), _ip(idx1), _fp(grad_dist2), _ip(idx2), _fp(g1), _fp(g2))
    return g1, g2
