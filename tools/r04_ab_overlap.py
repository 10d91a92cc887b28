"""Would pass A (weight gradient, known coefficients) and pass B (data gradient) of a SHORT layer overlap if they ran side by side?
Each pass is captured 20x into its own linear hipGraph; the two graphs are replayed (a) one after the other on one stream, (b) at the
same time on two streams.  (b) / (a) is the fraction a dual launch could reach."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gspn_amd import _lib as L

lib = L.lib()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(0)
BN_EPS = 1e-3


def graph_of(fn, stream, reps=20):
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
            for _ in range(reps):
                fn()
    return g


for (rows, cin, cout) in ((4096, 384, 256), (16384, 192, 128), (32768, 128, 128), (32768, 128, 256), (16384, 64, 64), (4096, 128, 128)):
    X = torch.randn(rows, cin, device=dev, generator=gen)
    Y = torch.randn(rows, cout, device=dev, generator=gen)
    dZ = torch.randn(rows, cout, device=dev, generator=gen)
    W = torch.randn(cin, cout, device=dev, generator=gen) * 0.1
    vec = [torch.rand(c, device=dev, generator=gen) + 0.5 for c in (cout,) * 5 + (cin,) * 2]
    a = L.DyArgs()
    a.Y, a.ldy, a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = Y.data_ptr(), cout, dZ.data_ptr(), cout, None, None, 0
    a.scale, a.shift, a.cA, a.cB, a.cC = (v.data_ptr() for v in vec[:5])
    work = torch.empty(int(lib.gspn_mlp_bwd_work_bytes(rows, cin, cout)) // 4 + 4, device=dev)
    dW = torch.empty(cin, cout, device=dev)
    dX = torch.empty(rows, cin, device=dev)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def passA():
        L.check(lib.gspn_mlp_bwd_wgrad_known(rows, cin, cout, ctypes.byref(a), L.ptr(X), cin, L.ptr(vec[5]), L.ptr(vec[6]), None, L.ptr(work), None, L.stream()), "A")

    def passB():
        L.check(lib.gspn_mlp_bwd_data_cols(rows, cin, cout, ctypes.byref(a), L.ptr(W), 0, cin, L.ptr(dX), cin, L.stream()), "B")

    try:
        gA, gB = graph_of(passA, s1), graph_of(passB, s2)
    except NotImplementedError as e:
        print(rows, cin, cout, "unsupported", e)
        continue
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))

    def seq():
        with torch.cuda.stream(s1):
            e0.record(); gA.replay(); e1.record(); gB.replay(); e2.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3, e1.elapsed_time(e2) / 20 * 1e3

    def par():
        with torch.cuda.stream(s1):
            e0.record()
        s2.wait_event(e0)
        with torch.cuda.stream(s1):
            gA.replay()
        with torch.cuda.stream(s2):
            gB.replay()
            eb = s2.record_event()
        with torch.cuda.stream(s1):
            s1.wait_event(eb)
            e2.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e2) / 20 * 1e3

    seq(); par()
    ta, tb = seq()
    tp = min(par() for _ in range(3))
    print("%6d x %3d <- %3d : pass A %5.1f us  pass B %5.1f us  sum %5.1f | side by side %5.1f us per pair  (%.2f of the sum)" % (rows, cin, cout, ta, tb, ta + tb, tp, tp / (ta + tb)), flush=True)
