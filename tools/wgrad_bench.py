"""time gspn_mlp_bwd_wgrad / bwd_data / fwd at the bench layer shapes (torch events), report effective GB/s"""
import ctypes, sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gspn_amd import _lib as L
lib = L.lib(); st = L.stream()
shapes = [("SA1-L1", 524288, 8, 6, 32, False), ("SA1-L2", 524288, 32, 32, 32, False), ("SA1-L3p", 524288, 32, 32, 64, True),
          ("SA2-L1", 131072, 68, 67, 64, False), ("SA2-L2", 131072, 64, 64, 64, False), ("SA2-L3p", 131072, 64, 64, 128, True),
          ("SA3-L1", 32768, 132, 131, 128, False), ("SA3-L2", 32768, 128, 128, 128, False), ("SA3-L3p", 32768, 128, 128, 256, True),
          ("FP1-L1", 4096, 384, 384, 256, False), ("FP1-L2", 4096, 256, 256, 128, False),
          ("FP2-L1", 16384, 192, 192, 128, False), ("FP2-L2", 16384, 128, 128, 64, False),
          ("FP3-L1", 262144, 68, 67, 64, False), ("FP3-L2", 262144, 64, 64, 64, False), ("FP3-L3", 262144, 64, 64, 64, False)]
for name, rows, ldx, cin, cout, pooled in shapes:
    dev = 'cuda'
    X = torch.randn(rows, ldx, device=dev); Y = torch.randn(rows, cout, device=dev)
    W = torch.randn(cin, cout, device=dev)
    mean = torch.zeros(cout, device=dev); var = torch.ones(cout, device=dev); gamma = torch.ones(cout, device=dev)
    scale = torch.ones(cout, device=dev); shift = torch.zeros(cout, device=dev)
    isc = torch.ones(cin, device=dev); ish = torch.zeros(cin, device=dev)
    cA = torch.ones(cout, device=dev); cB = torch.zeros(cout, device=dev); cC = torch.zeros(cout, device=dev)
    a = L.DyArgs(); a.Y, a.ldy = Y.data_ptr(), cout
    if pooled:
        ns = 32; dP = torch.randn(rows // ns, cout, device=dev); arg = torch.randint(0, ns, (rows // ns, cout), device=dev, dtype=torch.int32)
        a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = None, 0, dP.data_ptr(), arg.data_ptr(), ns
        dzb = 0
    else:
        dZ = torch.randn(rows, cout, device=dev); a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = dZ.data_ptr(), cout, None, None, 0
        dzb = rows * cout * 4
    a.scale, a.shift, a.cA, a.cB, a.cC = scale.data_ptr(), shift.data_ptr(), cA.data_ptr(), cB.data_ptr(), cC.data_ptr()
    work = torch.empty(int(lib.gspn_mlp_bwd_work_bytes(rows, cin, cout)) // 4 + 4, device=dev)
    dW = torch.empty(cin, cout, device=dev); dX = torch.empty(rows, ldx, device=dev)
    TR = int(__import__("os").environ.get("WB_TRAIN", "1"))
    def run_w():
        L.check(lib.gspn_mlp_bwd_wgrad(rows, cin, cout, ctypes.byref(a), L.ptr(X), ldx, L.ptr(isc), L.ptr(ish), L.ptr(mean), L.ptr(var), L.ptr(gamma), 1e-3, 1, TR,
                                       L.ptr(work), L.ptr(cA), L.ptr(cB), L.ptr(cC), None, None, None, L.ptr(dW), st), "w")
    def run_d():
        L.check(lib.gspn_mlp_bwd_data(rows, cin, cout, ctypes.byref(a), L.ptr(W), L.ptr(dX), ldx, st), "d")
    def run_f():
        L.check(lib.gspn_mlp_fwd(rows, cin, cout, L.ptr(X), ldx, L.ptr(isc), L.ptr(ish), L.ptr(W), L.ptr(mean), L.ptr(Y), cout, None, st), "f")
    res = []
    for fn, byts in ((run_w, rows * (ldx + cout) * 4 + dzb), (run_d, rows * (cout + ldx) * 4 + dzb), (run_f, rows * (ldx + cout) * 4)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        res.append("%6.1f us %5.2f TB/s" % (us, byts / us / 1e6))
    print("%-8s rows %6d %3d->%3d  wgrad %s | bwd_data %s | fwd %s" % (name, rows, cin, cout, *res))
