"""gspn_mlp_bwd_fused alone (dW and dX of a layer in one launch + the dW reduction kernel) at the bench's shapes: microseconds, TB/s of the four
row streams (Y, dZ, previous Y, dX), TFLOP/s of both products (library chosen by GSPN_HIP_LIB; GSPN_BWD_FUSED_BPC sets workgroups per CU)"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gspn_amd import _lib as L
lib = L.lib(); dev = torch.device('cuda', 0)
shapes = [(262144, 64, 64), (131072, 64, 64), (524288, 32, 32), (262144, 32, 64), (262144, 64, 32), (1048576, 64, 64)]
def run(rows, cin, cout, reps=10):
    Y = torch.randn(rows, cout, device=dev); W = torch.randn(cin, cout, device=dev) * 0.1
    Yp = torch.randn(rows, cin, device=dev); dZ = torch.randn(rows, cout, device=dev)
    dX = torch.empty(rows, cin, device=dev); dW = torch.empty(cin, cout, device=dev)
    one = lambda c, v=1.0: torch.full((c,), v, device=dev)
    a = L.DyArgs(); a.Y, a.ldy = Y.data_ptr(), cout
    a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = dZ.data_ptr(), cout, None, None, 0
    sc, sh, cA, cB, cC = one(cout), one(cout, 0.1), one(cout), one(cout, 0.01), one(cout, 0.0)
    a.scale, a.shift, a.cA, a.cB, a.cC = sc.data_ptr(), sh.data_ptr(), cA.data_ptr(), cB.data_ptr(), cC.data_ptr()
    psc, psh, pm, pv = one(cin), one(cin, 0.1), one(cin, 0.0), one(cin)
    part = torch.empty(int(lib.gspn_rsum_part_floats(rows, cin)), device=dev); npart = ctypes.c_int(0)
    work = torch.empty(int(lib.gspn_mlp_bwd_work_bytes(rows, cin, cout)) // 4 + 4, device=dev)
    f = lambda: L.check(lib.gspn_mlp_bwd_fused(rows, cin, cout, ctypes.byref(a), L.ptr(W), L.ptr(Yp), cin, L.ptr(psc), L.ptr(psh), L.ptr(dX), cin, L.ptr(work),
                                               L.ptr(dW), L.ptr(pm), L.ptr(pv), 1e-3, L.ptr(part), ctypes.byref(npart), L.stream()), "fused")
    for _ in range(3): f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("lib:", os.path.basename(L.LIB_PATH), "bpc", os.environ.get("GSPN_BWD_FUSED_BPC", "default"))
for rows, cin, cout in shapes:
    us = run(rows, cin, cout)
    by = 4.0 * rows * (2 * cout + 2 * cin); fl = 4.0 * rows * cin * cout
    print("fused %8d x %3d <- %3d: %7.1f us  %5.2f TB/s  %5.1f TF" % (rows, cin, cout, us, by / us / 1e6, fl / us / 1e6), flush=True)
