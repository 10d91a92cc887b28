"""GPU parity of the multi-CU farthest point sampling (gspn_amd/csrc/sampling_multi.hip) against the CPU oracle: index-exact, at
BASELINE configs[4]'s scene size (65536 points) and at the scan scale of data_prep.py:64-83 (n ~ 1.5e5, m = 30000), with every
workgroup count the launcher can pick, on tie-heavy inputs, and through the reference-shaped drop-in symbol."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import data as D

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def multi_fps(m, xyz, G, monkeypatch):
    from gspn_amd import tf_sampling
    monkeypatch.setattr(tf_sampling, "FPS_MULTI_FORCE", True)
    monkeypatch.setattr(tf_sampling, "FPS_MULTI_G", G)
    return tf_sampling.farthest_point_sample(m, dev(xyz)).cpu().numpy()


@pytest.mark.parametrize("kind,b,n,m,G", [
    ("U", 2, 4096, 300, 1), ("U", 2, 4096, 300, 2), ("D", 3, 9000, 700, 3), ("S", 1, 20000, 900, 5), ("D", 2, 32768, 1500, 2),
    ("U", 9, 12000, 200, 4), ("U", 1, 30000, 2500, 8), ("D", 1, 50000, 1200, 11), ("U", 1, 40000, 600, 16), ("D", 1, 70000, 500, 32),
    ("U", 2, 1000, 1100, 2), ("U", 1, 100, 64, 3), ("U", 17, 3000, 100, 2),
])
def test_fps_multi_matches_oracle(kind, b, n, m, G, monkeypatch):
    xyz = D.batch(kind, b, n, 5)
    ref = O.farthest_point_sample(m, xyz, mt=True)
    got = multi_fps(m, xyz, G, monkeypatch)
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("b,n,m,grid,G", [(2, 40000, 1500, 24, 0), (1, 65536, 3000, 12, 0), (2, 20000, 700, 16, 3), (1, 9000, 500, 10, 2),
                                          (1, 32768, 40, 2, 4), (1, 66000, 300, 6, 7)])
def test_fps_multi_lattice_ties(b, n, m, grid, G, monkeypatch):
    """integer lattice: equal maxima inside a cell, between the cells of a workgroup and between workgroups, duplicates, and the
    degenerate tail for m > grid^3 -- the reference's tie order (k mod 512, k) decides every one of them"""
    rng = np.random.default_rng(grid * 1000 + n)
    xyz = (rng.integers(0, grid, size=(b, n, 3)).astype(np.float32) / np.float32(8.0)).astype(np.float32)
    ref = O.farthest_point_sample(m, xyz, mt=True)
    got = multi_fps(m, xyz, G, monkeypatch)
    np.testing.assert_array_equal(got, ref)


def test_fps_multi_all_points_identical(monkeypatch):
    xyz = np.full((2, 40000, 3), 0.5, np.float32)
    got = multi_fps(50, xyz, 0, monkeypatch)
    assert (got == 0).all()


@pytest.mark.parametrize("G", [0, 2, 3])
def test_fps_config5_scene_size_index_exact(G, monkeypatch):
    """BASELINE configs[4]: 8 scenes per GPU of 65536 points, SA1 of pn2_fea_extractor samples 2048 (model_rpointnet.py:224)"""
    xyz = D.batch("U", 8, 65536)
    ref = O.farthest_point_sample(2048, xyz, mt=True)
    got = multi_fps(2048, xyz, G, monkeypatch)
    np.testing.assert_array_equal(got, ref)
    assert (got[:, 0] == 0).all() and all(len(np.unique(r)) == 2048 for r in got)


def test_fps_scan_scale_index_exact(monkeypatch):
    """data_prep.py:64-83: one ScanNet-sized scene (n ~ 1.5e5) down to 30000 points"""
    xyz = D.batch("D", 1, 150000, 11)
    xyz[0] *= np.array([8.0, 6.0, 3.0], np.float32)           # metre-scale room
    ref = O.farthest_point_sample(30000, xyz)
    got = multi_fps(30000, xyz, 0, monkeypatch)
    np.testing.assert_array_equal(got, ref)


def test_fps_default_path_above_32768():
    """no forcing: n > 32768 takes the multi-CU kernel through the Python surface of tf_sampling.py"""
    from gspn_amd.tf_sampling import farthest_point_sample
    xyz = D.batch("D", 2, 40000)
    ref = O.farthest_point_sample(200, xyz)
    got = farthest_point_sample(200, dev(xyz)).cpu().numpy()
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("b,n,m", [(8, 32768, 512), (11, 9000, 300), (3, 40000, 256), (9, 66000, 128), (2, 5000, 100)])
def test_dropin_symbol_uses_reference_scratch(b, n, m):
    """gspn_farthestpointsampling(b,n,m,inp,temp,out) bound the way the reference binds farthestpointsamplingLauncher: temp is the
    reference's (32,n) float scratch (tf_sampling.cpp:111-115) and is all the workspace the fast kernels get"""
    from gspn_amd import _lib as L
    xyz = D.batch("U", b, n, 2)
    ref = O.farthest_point_sample(m, xyz, mt=True)
    t = dev(xyz)
    temp = torch.empty((32, n), dtype=torch.float32, device="cuda")
    guard = torch.full((1024,), 7.0, device="cuda")           # allocated right behind: a workspace overrun would likely land here
    out = torch.empty((b, m), dtype=torch.int32, device="cuda")
    L.check(L.lib().gspn_farthestpointsampling(b, n, m, L.ptr(t), L.ptr(temp), L.ptr(out), L.stream()), "fps")
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    assert (guard == 7.0).all()


def test_fps_multi_status_word_is_clean():
    from gspn_amd import _lib as L
    b, n, m = 2, 50000, 64
    t = dev(D.batch("U", b, n))
    ws = torch.empty((int(L.lib().gspn_fps_multi_ws_bytes(b, n)) + 3) // 4, dtype=torch.float32, device="cuda")
    out = torch.empty((b, m), dtype=torch.int32, device="cuda")
    L.check(L.lib().gspn_farthestpointsampling_multi(b, n, m, 0, L.ptr(t), L.ptr(ws), L.ptr(out), L.stream()), "fps multi")
    assert L.lib().gspn_fps_multi_status(L.ptr(ws), b, n, L.stream()) == 0


def test_fps_multi_random_shapes(monkeypatch):
    """seeded sweep over scene size, sample count, workgroup count and cloud kind (uniform / duplicated points / coarse lattice with ties
    everywhere / clustered): every combination index-exact against the oracle"""
    rng = np.random.default_rng(2024)
    for trial in range(24):
        n = int(rng.integers(40, 60000))
        m = int(rng.integers(1, min(n + 50, 1500)))
        b = int(rng.integers(1, 4))
        gmin = (n + 32767) // 32768
        G = int(rng.integers(gmin, min(32, gmin + 9) + 1))
        kind = trial % 4
        if kind == 0:
            xyz = rng.random((b, n, 3), dtype=np.float32)
        elif kind == 1:
            xyz = rng.random((b, n, 3), dtype=np.float32)
            k = n // 3
            xyz[:, n - k:] = xyz[:, :k]                                   # a third of the points are duplicates
        elif kind == 2:
            xyz = (rng.integers(0, 7, size=(b, n, 3)) / 4.0).astype(np.float32)      # <= 343 distinct points: ties and the degenerate tail
        else:
            centres = rng.random((b, 5, 3)).astype(np.float32)
            xyz = (centres[:, rng.integers(0, 5, size=n)] + 0.02 * rng.standard_normal((b, n, 3))).astype(np.float32)
        ref = O.farthest_point_sample(m, xyz, mt=True)
        got = multi_fps(m, np.ascontiguousarray(xyz), G, monkeypatch)
        np.testing.assert_array_equal(got, ref, err_msg="trial %d: b=%d n=%d m=%d G=%d kind=%d" % (trial, b, n, m, G, kind))
