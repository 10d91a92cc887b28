"""TEST INFRASTRUCTURE ONLY -- a second opinion, not a pin (oracle/Makefile: ref_hip).

The reference's own CUDA kernel sources (tf_ops/sampling/tf_sampling_g.cu, tf_ops/grouping/tf_grouping_g.cu) compiled as they lie for
gfx950 by hipcc; this module calls their launchers -- the very symbols the reference's OpKernels call (tf_sampling.cpp:65,94,125,150,
tf_grouping.cpp:96,138,172,203,241,277) -- on torch ROCm tensors.  They launch on the legacy default stream (the reference passes none),
so every call is bracketed by device synchronisations.  Only tests/ may import this."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}


def lib(name, nofma=False):
    """name: 'sampling' | 'grouping'; nofma: False = hipcc's default contraction, True = -ffp-contract=off, "fast" = -ffp-contract=fast (the
    back end contracts); None when the prebuilt library is absent (no reference checkout at build time)"""
    key = (name, nofma)
    if key not in _libs:
        p = os.path.join(_HERE, "_ref", "libtf_%s_g_hip%s.so" % (name, "_fast" if nofma == "fast" else ("_nofma" if nofma else "")))
        _libs[key] = ctypes.CDLL(p) if os.path.exists(p) else None
    return _libs[key]


def available():
    return lib("sampling") is not None and lib("grouping") is not None and torch.cuda.is_available()


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _call(L, sym, *args):
    torch.cuda.synchronize()
    f = getattr(L, sym)
    f.restype = None
    f(*args)
    torch.cuda.synchronize()


def farthest_point_sample(npoint, xyz, nofma=False):
    """farthestpointsamplingLauncher(b, n, m, inp, temp (32, n), out)  -- tf_sampling_g.cu:203-205"""
    b, n, _ = xyz.shape
    temp = torch.empty((32, n), dtype=torch.float32, device=xyz.device)
    out = torch.zeros((b, npoint), dtype=torch.int32, device=xyz.device)
    _call(lib("sampling", nofma), "_Z29farthestpointsamplingLauncheriiiPKfPfPi", b, n, npoint, _p(xyz), _p(temp), _p(out))
    return out


def gather_point(inp, idx):
    b, n, _ = inp.shape
    m = idx.shape[1]
    out = torch.empty((b, m, 3), dtype=torch.float32, device=inp.device)
    _call(lib("sampling"), "_Z19gatherpointLauncheriiiPKfPKiPf", b, n, m, _p(inp), _p(idx), _p(out))
    return out


def gather_point_grad(n, idx, out_g):
    b, m = idx.shape
    inp_g = torch.zeros((b, n, 3), dtype=torch.float32, device=out_g.device)      # cudaMemset in the OpKernel, tf_sampling.cpp:174
    _call(lib("sampling"), "_Z23scatteraddpointLauncheriiiPKfPKiPf", b, n, m, _p(out_g), _p(idx), _p(inp_g))
    return inp_g


def prob_sample(inp, inpr):
    b, n = inp.shape
    m = inpr.shape[1]
    temp = torch.empty((b, n), dtype=torch.float32, device=inp.device)
    out = torch.empty((b, m), dtype=torch.int32, device=inp.device)
    _call(lib("sampling"), "_Z18probsampleLauncheriiiPKfS0_PfPi", b, n, m, _p(inp), _p(inpr), _p(temp), _p(out))
    return out


def query_ball_point(radius, nsample, xyz1, xyz2, nofma=False):
    """queryBallPointLauncher(b, n, m, radius, nsample, xyz1, xyz2, idx, pts_cnt)  -- tf_grouping_g.cu:186-189; rows without a hit stay as
    allocated (zeros here)"""
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = torch.zeros((b, m, nsample), dtype=torch.int32, device=xyz1.device)
    cnt = torch.zeros((b, m), dtype=torch.int32, device=xyz1.device)
    _call(lib("grouping", nofma), "_Z22queryBallPointLauncheriiifiPKfS0_PiS1_", b, n, m, ctypes.c_float(radius), nsample, _p(xyz1), _p(xyz2), _p(idx), _p(cnt))
    return idx, cnt


def group_point(points, idx):
    b, n, c = points.shape
    _, m, ns = idx.shape
    out = torch.empty((b, m, ns, c), dtype=torch.float32, device=points.device)
    _call(lib("grouping"), "_Z18groupPointLauncheriiiiiPKfPKiPf", b, n, c, m, ns, _p(points), _p(idx), _p(out))
    return out


def group_point_grad(n, idx, grad_out):
    b, m, ns = idx.shape
    c = grad_out.shape[3]
    g = torch.zeros((b, n, c), dtype=torch.float32, device=grad_out.device)      # cudaMemset in the OpKernel, tf_grouping.cpp:234
    _call(lib("grouping"), "_Z22groupPointGradLauncheriiiiiPKfPKiPf", b, n, c, m, ns, _p(grad_out), _p(idx), _p(g))
    return g


def group_maxpool(points, idx):
    b, n, c = points.shape
    _, m, ns = idx.shape
    out = torch.empty((b, m, c), dtype=torch.float32, device=points.device)
    mi = torch.empty((b, m, c), dtype=torch.int32, device=points.device)
    _call(lib("grouping"), "_Z20groupMaxpoolLauncheriiiiiPKfPKiPfPi", b, n, c, m, ns, _p(points), _p(idx), _p(out), _p(mi))
    return out, mi


def select_top_k(k, dist):
    b, m, n = dist.shape
    outi = torch.empty((b, m, n), dtype=torch.int32, device=dist.device)
    out = torch.empty((b, m, n), dtype=torch.float32, device=dist.device)
    _call(lib("grouping"), "_Z21selectionSortLauncheriiiiPKfPiPf", b, n, m, k, _p(dist), _p(outi), _p(out))
    return outi, out


def fps_min_distances(npoint, xyz, nofma=False):
    """the reference kernel's scratch after the run: temp[i][k] = min squared distance of point k of scene i (< 32) to the chosen set, computed
    by the reference's own expression (tf_sampling_g.cu:139-145) as THIS compiler contracted it"""
    b, n, _ = xyz.shape
    assert b <= 32
    temp = torch.empty((32, n), dtype=torch.float32, device=xyz.device)
    out = torch.zeros((b, npoint), dtype=torch.int32, device=xyz.device)
    _call(lib("sampling", nofma), "_Z29farthestpointsamplingLauncheriiiPKfPfPi", b, n, npoint, _p(xyz), _p(temp), _p(out))
    return out, temp[:b]
