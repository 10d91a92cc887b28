"""which layer kernels does a concurrent FPS launch slow down?  chains of one MLP kernel (graph-replayed) with / without FPS on a side stream"""
import ctypes, sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gspn_amd import _lib as L
from gspn_amd.tf_sampling import farthest_point_sample
lib = L.lib(); dev = torch.device('cuda', 0)
side = torch.cuda.Stream()
xyz = torch.rand(8, 32768, 3, device=dev)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def chain(fn, N):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(N): fn()
    return g
def report(name, g):
    alone = timeit(lambda: g.replay())
    res = []
    for bg in (0, 1):
        def both():
            ev = torch.cuda.current_stream().record_event()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                prev = lib.gspn_fps_background(bg)
                farthest_point_sample(2048, xyz)
                lib.gspn_fps_background(prev)
                done = side.record_event()
            g.replay()
            torch.cuda.current_stream().wait_event(done)
        res.append(timeit(both))
    print("%-44s alone %.3f ms | + FPS: %.3f ms | + FPS(background): %.3f ms" % (name, alone, res[0], res[1]), flush=True)
rows, cin, cout, ldx = 262144, 64, 64, 64
X = torch.randn(rows, ldx, device=dev); Y = torch.randn(rows, cout, device=dev); dZ = torch.randn(rows, cout, device=dev)
W = torch.randn(cin, cout, device=dev); bias = torch.zeros(cout, device=dev)
mean = torch.zeros(cout, device=dev); var = torch.ones(cout, device=dev); gamma = torch.ones(cout, device=dev)
scale = torch.ones(cout, device=dev); shift = torch.zeros(cout, device=dev)
isc = torch.ones(cin, device=dev); ish = torch.zeros(cin, device=dev)
cA = torch.ones(cout, device=dev); cB = torch.zeros(cout, device=dev); cC = torch.zeros(cout, device=dev)
a = L.DyArgs(); a.Y, a.ldy = Y.data_ptr(), cout
a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = dZ.data_ptr(), cout, None, None, 0
a.scale, a.shift, a.cA, a.cB, a.cC = scale.data_ptr(), shift.data_ptr(), cA.data_ptr(), cB.data_ptr(), cC.data_ptr()
work = torch.empty(int(lib.gspn_mlp_bwd_work_bytes(rows, cin, cout)) // 4 + 4, device=dev)
dW = torch.empty(cin, cout, device=dev); dX = torch.empty(rows, ldx, device=dev)
stats = torch.empty(int(lib.gspn_mlp_fwd_stats_bytes(rows, cout)) // 4, device=dev)
st = lambda: L.stream()
def f_w(): L.check(lib.gspn_mlp_bwd_wgrad(rows, cin, cout, ctypes.byref(a), L.ptr(X), ldx, L.ptr(isc), L.ptr(ish), L.ptr(mean), L.ptr(var), L.ptr(gamma), 1e-3, 1, 1,
                                          L.ptr(work), L.ptr(cA), L.ptr(cB), L.ptr(cC), None, None, None, L.ptr(dW), st()), "w")
def f_d(): L.check(lib.gspn_mlp_bwd_data(rows, cin, cout, ctypes.byref(a), L.ptr(W), L.ptr(dX), ldx, st()), "d")
def f_f(): L.check(lib.gspn_mlp_fwd(rows, cin, cout, L.ptr(X), ldx, L.ptr(isc), L.ptr(ish), L.ptr(W), L.ptr(bias), L.ptr(Y), cout, L.ptr(stats), st()), "f")
big = torch.zeros(1 << 24, device=dev)
report("60 x wgrad (3 kernels each) 64->64 x 262144", chain(f_w, 60))
report("100 x bwd_data 64->64 x 262144", chain(f_d, 100))
report("120 x fwd(stream) 64->64 x 262144", chain(f_f, 120))
report("250 x add_ 16M floats", chain(lambda: big.add_(1.0), 250))
