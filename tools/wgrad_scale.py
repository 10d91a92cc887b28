"""wgrad main-kernel time vs rows for a few (cin, cout, pooled) shapes: slope (per-row cost) and intercept (fixed overhead)"""
import ctypes, sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gspn_amd import _lib as L
lib = L.lib(); st = L.stream()
def run(rows, ldx, cin, cout, pooled, tr=1):
    dev = 'cuda'
    X = torch.randn(rows, ldx, device=dev); Y = torch.randn(rows, cout, device=dev)
    mean = torch.zeros(cout, device=dev); var = torch.ones(cout, device=dev); gamma = torch.ones(cout, device=dev)
    scale = torch.ones(cout, device=dev); shift = torch.zeros(cout, device=dev)
    isc = torch.ones(cin, device=dev); ish = torch.zeros(cin, device=dev)
    cA = torch.ones(cout, device=dev); cB = torch.zeros(cout, device=dev); cC = torch.zeros(cout, device=dev)
    a = L.DyArgs(); a.Y, a.ldy = Y.data_ptr(), cout
    if pooled:
        ns = 32; dP = torch.randn(rows // ns, cout, device=dev); arg = torch.randint(0, ns, (rows // ns, cout), device=dev, dtype=torch.int32)
        a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = None, 0, dP.data_ptr(), arg.data_ptr(), ns
    else:
        dZ = torch.randn(rows, cout, device=dev); a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = dZ.data_ptr(), cout, None, None, 0
    a.scale, a.shift, a.cA, a.cB, a.cC = scale.data_ptr(), shift.data_ptr(), cA.data_ptr(), cB.data_ptr(), cC.data_ptr()
    work = torch.empty(int(lib.gspn_mlp_bwd_work_bytes(rows, cin, cout)) // 4 + 4, device=dev)
    dW = torch.empty(cin, cout, device=dev)
    def f():
        L.check(lib.gspn_mlp_bwd_wgrad(rows, cin, cout, ctypes.byref(a), L.ptr(X), ldx, L.ptr(isc), L.ptr(ish), L.ptr(mean), L.ptr(var), L.ptr(gamma), 1e-3, 1, tr,
                                       L.ptr(work), L.ptr(cA), L.ptr(cB), L.ptr(cC), None, None, None, L.ptr(dW), st), "w")
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 100
for (ldx, cin, cout, pooled) in [(128, 128, 128, False), (128, 128, 256, True), (128, 128, 256, False), (64, 64, 64, False), (64, 64, 128, True), (68, 67, 64, False), (32, 32, 64, True), (32, 32, 64, False), (32, 32, 32, False)]:
    res = []
    for rows in (16384, 32768, 65536, 131072, 262144, 524288):
        res.append("%7.1f" % run(rows, ldx, cin, cout, pooled))
    ev = run(262144, ldx, cin, cout, pooled, tr=0)
    print("%3d->%3d pooled=%d  us(total of 3 kernels) @rows 16k..512k: %s   | eval-mode @256k: %.1f" % (cin, cout, pooled, " ".join(res), ev), flush=True)
