"""The split-K kernels of the short layers (csrc/mlp_short.hip, r05): the forward of layers with <= 8192 rows (default path) against an fp64
product -- outputs, the per-workgroup partial column sums the batch norm is finalised from, the 32-row pool epilogue (maximum AND the first
row that reaches it).
utils/pointnet_util.py:109-113,165-169 (the FP1 / SA3 shapes of models/model_rpointnet.py:226-230)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("rows,cin,cout,act,pool", [(4096, 384, 256, False, False), (4096, 256, 128, True, False), (4096, 128, 128, False, False),
                                                    (8192, 64, 64, True, True), (2048, 192, 96, True, True), (1024, 128, 256, True, False),
                                                    (64, 64, 32, False, True), (4032, 256, 64, True, False)])
def test_short_forward_against_fp64(rows, cin, cout, act, pool):
    from gspn_amd import _lib as L
    lib = L.lib()
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(rows + cin)
    X = torch.randn(rows, cin + 4, device=dev, generator=gen)                 # padded pitch: ldx > cin
    W = torch.randn(cin, cout, device=dev, generator=gen) / cin ** 0.5
    bias = torch.randn(cout, device=dev, generator=gen) * 0.1
    sc = (torch.rand(cin, device=dev, generator=gen) + 0.5) if act else None
    sh = (torch.randn(cin, device=dev, generator=gen) * 0.3) if act else None
    if act:
        sc[::7] = -sc[::7]                                                    # negative scales are legal (gamma < 0)
    Y = torch.full((rows, cout + 4), float("nan"), device=dev)               # ldy > cout: the pad columns must stay untouched
    nst = int(lib.gspn_mlp_fwd_stats_bytes(rows, cout)) // 4
    stats = torch.full((nst,), float("nan"), device=dev)
    if pool:
        Xq = torch.round(X * 2) / 2                                           # quantised inputs and weights: tied maxima inside a pool group occur
        X = Xq.contiguous()
        W = (torch.round(W * 8) / 8).contiguous()
    vmax = torch.full((rows // 32, cout), float("nan"), device=dev) if pool else None
    amax = torch.full((rows // 32, cout), -1, dtype=torch.int32, device=dev) if pool else None
    st = L.stream()
    if pool:
        L.check(lib.gspn_mlp_fwd_pool32(rows, cin, cout, L.ptr(X), cin + 4, L.ptr(sc), L.ptr(sh), L.ptr(W), L.ptr(bias), L.ptr(Y), cout + 4, L.ptr(stats),
                                        L.ptr(vmax), L.ptr(amax), st), "fwd")
    else:
        L.check(lib.gspn_mlp_fwd(rows, cin, cout, L.ptr(X), cin + 4, L.ptr(sc), L.ptr(sh), L.ptr(W), L.ptr(bias), L.ptr(Y), cout + 4, L.ptr(stats), st), "fwd")
    torch.cuda.synchronize()
    A = X[:, :cin]
    A = torch.relu((A * sc + sh).double()) if act else A.double()            # two fp32 roundings, as tf.nn.batch_normalization's x*scale + shift
    ref = A @ W.double() + bias.double()
    got = Y[:, :cout]
    assert bool(torch.isnan(Y[:, cout:]).all())
    assert float((got.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    parts = stats.view(-1, 2, cout).double()
    assert bool(torch.isfinite(parts).all())
    tot = parts.sum(0)
    assert float((tot[0] - ref.sum(0)).abs().max()) <= 1e-5 * float(ref.abs().sum(0).max())
    assert float((tot[1] - (ref * ref).sum(0)).abs().max()) <= 1e-5 * float((ref * ref).sum(0).max())
    if pool:
        g = got.reshape(rows // 32, 32, cout)
        mx = g.max(1).values
        first = (g == mx.unsqueeze(1)).float().argmax(1).int()                # the FIRST row that reaches the maximum (the reference's strict '>')
        assert torch.equal(vmax, mx) and torch.equal(amax, first)


def test_short_forward_is_deterministic():
    from gspn_amd import _lib as L
    lib = L.lib()
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(1)
    rows, cin, cout = 4096, 256, 128
    X = torch.randn(rows, cin, device=dev, generator=gen)
    W = torch.randn(cin, cout, device=dev, generator=gen)
    outs = []
    for _ in range(3):
        Y = torch.empty(rows, cout, device=dev)
        stats = torch.empty(int(lib.gspn_mlp_fwd_stats_bytes(rows, cout)) // 4, device=dev)
        L.check(lib.gspn_mlp_fwd(rows, cin, cout, L.ptr(X), cin, None, None, L.ptr(W), None, L.ptr(Y), cout, L.ptr(stats), L.stream()), "fwd")
        outs.append((Y, stats))
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
