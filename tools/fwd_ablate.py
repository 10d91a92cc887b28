"""forward GEMM kernel alone at chosen layer shapes: microseconds, TB/s of algorithmic bytes, TFLOP/s (library chosen by GSPN_HIP_LIB)"""
import os, sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gspn_amd import _lib as L
lib = L.lib(); dev = torch.device('cuda', 0)
shapes = [(262144, 64, 64), (524288, 32, 64), (524288, 32, 32), (131072, 64, 128), (131072, 64, 64), (32768, 128, 128), (32768, 128, 256), (16384, 192, 128), (4096, 384, 256), (4096, 256, 128), (524288, 64, 128), (1048576, 64, 128), (524288, 128, 256), (1048576, 128, 256)]
def run(rows, cin, cout, reps=20):
    X = torch.randn(rows, cin, device=dev); Y = torch.empty(rows, cout, device=dev)
    W = torch.randn(cin, cout, device=dev) * 0.1; bias = torch.zeros(cout, device=dev)
    sc = torch.rand(cin, device=dev) + 0.5; sh = torch.randn(cin, device=dev) * 0.1
    stats = torch.empty(int(lib.gspn_mlp_fwd_stats_bytes(rows, cout)) // 4, device=dev)
    f = lambda: L.check(lib.gspn_mlp_fwd(rows, cin, cout, L.ptr(X), cin, L.ptr(sc), L.ptr(sh), L.ptr(W), L.ptr(bias), L.ptr(Y), cout, L.ptr(stats), L.stream()), "fwd")
    for _ in range(3): f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("lib:", os.path.basename(L.LIB_PATH))
for rows, cin, cout in shapes:
    us = run(rows, cin, cout)
    by = 4.0 * rows * (cin + cout); fl = 2.0 * rows * cin * cout
    print("fwd %7d x %3d -> %3d : %7.1f us  %5.2f TB/s  %5.1f TF" % (rows, cin, cout, us, by / us / 1e6, fl / us / 1e6), flush=True)
