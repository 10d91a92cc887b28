"""pass A with known coefficients (gspn_mlp_bwd_wgrad_known, incl. its dW reduction) at the bench's layer shapes: microseconds, TB/s of algorithmic
bytes, TFLOP/s.  GSPN_WGRAD_LEAN=0 selects the streaming kernel."""
import ctypes, os, sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gspn_amd import _lib as L
lib = L.lib(); dev = torch.device('cuda', 0)
# (rows, cin, cout, pool_ns)
shapes = [(262144, 64, 64, 0), (131072, 64, 128, 32), (131072, 64, 64, 0), (524288, 32, 64, 32), (524288, 32, 32, 0), (32768, 128, 256, 32), (32768, 128, 128, 0),
          (16384, 192, 128, 0), (16384, 128, 64, 0), (4096, 384, 256, 0), (4096, 256, 128, 0), (16384, 64, 64, 0), (1048576, 128, 256, 512), (1048576, 64, 128, 0)]
def run(rows, cin, cout, ns, reps=10):
    X = torch.randn(rows, cin, device=dev); Y = torch.randn(rows, cout, device=dev)
    one = lambda c, v=1.0: torch.full((c,), v, device=dev)
    a = L.DyArgs(); a.Y, a.ldy = Y.data_ptr(), cout
    if ns:
        g = rows // ns
        dP = torch.randn(g, cout, device=dev); arg = torch.randint(0, ns, (g, cout), device=dev, dtype=torch.int32)
        a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = None, 0, dP.data_ptr(), arg.data_ptr(), ns
    else:
        dZ = torch.randn(rows, cout, device=dev)
        a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = dZ.data_ptr(), cout, None, None, 0
    sc, sh, cA, cB, cC = one(cout), one(cout, 0.1), one(cout), one(cout, 0.01), one(cout, 0.0)
    a.scale, a.shift, a.cA, a.cB, a.cC = sc.data_ptr(), sh.data_ptr(), cA.data_ptr(), cB.data_ptr(), cC.data_ptr()
    isc, ish = one(cin), one(cin, 0.1)
    work = torch.empty(int(lib.gspn_mlp_bwd_work_bytes(rows, cin, cout)) // 4 + 4, device=dev)
    dW = torch.empty(cin, cout, device=dev)
    f = lambda: L.check(lib.gspn_mlp_bwd_wgrad_known(rows, cin, cout, ctypes.byref(a), L.ptr(X), cin, L.ptr(isc), L.ptr(ish), None, L.ptr(work), L.ptr(dW), L.stream()), "wgrad")
    for _ in range(3): f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, float(dW.abs().sum())
for rows, cin, cout, ns in shapes:
    us, chk = run(rows, cin, cout, ns)
    by = 4.0 * rows * (cin + cout * (1 if ns else 2)); fl = 2.0 * rows * cin * cout
    print("wgrad %8d x %3d^T %3d %s: %7.1f us  %5.2f TB/s  %5.1f TF  (|dW| %.4e)" % (rows, cin, cout, "pool%-3d" % ns if ns else "dense  ", us, by / us / 1e6, fl / us / 1e6, chk), flush=True)
