"""The forward kernels of the long thin layers against fp64 at ONE tolerance (1e-5 of max |y|, BASELINE.json north_star): the default
(exact fp32 products) and the opt-in kernel on 3 x bf16 operand pieces (GSPN_MFMA_SPLIT=1, fwd_split_kernel).  The switch
is read once per process, so each check runs in a child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r'''
import ctypes, sys, torch
from gspn_amd import _lib as L
lib = L.lib()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
worst = 0.0
for (rows, cin, cout, act, pool) in ((65536, 32, 32, True, False), (131072, 64, 128, True, True), (65536, 64, 64, False, False), (262144, 32, 64, True, True),
                                     (65536 + 128, 64, 32, True, False), (262144 + 384, 64, 64, True, False), (524288, 64, 128, True, True), (262144, 32, 32, False, True)):
    X = torch.randn(rows, cin, device=dev, generator=g) * 1.5 + 0.2
    W = torch.randn(cin, cout, device=dev, generator=g) * 0.1
    bias = torch.rand(cout, device=dev, generator=g) - 0.5
    sc = torch.rand(cin, device=dev, generator=g) + 0.5
    sh = torch.rand(cin, device=dev, generator=g) - 0.5
    Y = torch.empty(rows, cout, device=dev)
    stats = torch.empty(int(lib.gspn_mlp_fwd_stats_bytes(rows, cout)) // 4, device=dev)
    a = (L.ptr(sc), L.ptr(sh)) if act else (None, None)
    if pool:
        vmax = torch.empty(rows // 32, cout, device=dev); amax = torch.empty(rows // 32, cout, dtype=torch.int32, device=dev)
        L.check(lib.gspn_mlp_fwd_pool32(rows, cin, cout, L.ptr(X), cin, a[0], a[1], L.ptr(W), L.ptr(bias), L.ptr(Y), cout, L.ptr(stats), L.ptr(vmax), L.ptr(amax), L.stream()), "fwd")
    else:
        L.check(lib.gspn_mlp_fwd(rows, cin, cout, L.ptr(X), cin, a[0], a[1], L.ptr(W), L.ptr(bias), L.ptr(Y), cout, L.ptr(stats), L.stream()), "fwd")
    torch.cuda.synchronize()
    A = X
    if act:
        A = torch.relu((X * sc) + sh)                     # two fp32 roundings, the operand every forward kernel forms
    ref = A.double() @ W.double() + bias.double()
    err = float((Y.double() - ref).abs().max() / ref.abs().max())
    worst = max(worst, err)
    assert err < 1e-5, (rows, cin, cout, err)
    nparts = stats.numel() // (2 * cout)
    st = stats.view(nparts, 2, cout).double().sum(0)
    assert float((st[0] - ref.sum(0)).abs().max() / ref.abs().sum(0).max()) < 1e-5
    assert float((st[1] - (ref * ref).sum(0)).abs().max() / (ref * ref).sum(0).max()) < 1e-5
    if pool:
        gmax = Y.view(rows // 32, 32, cout).max(1)
        assert torch.equal(vmax, gmax.values)              # the pool epilogue reports the tile's own maxima ...
        assert torch.equal(Y.view(rows // 32, 32, cout).gather(1, amax.long().unsqueeze(1)).squeeze(1), vmax)     # ... and a row that holds them
print("WORST %.3g" % worst)
'''


@pytest.mark.parametrize("switches", [{}, {"GSPN_MFMA_SPLIT": "1"}], ids=["default", "bf16_pieces"])
def test_long_layer_forward_kernels_meet_the_fp32_tolerance(switches):
    env = dict(os.environ, **switches)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", CHILD], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "WORST" in r.stdout
