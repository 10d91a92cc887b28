"""sweep the column-tile width of the register-staged forward / pass-B kernels (GSPN_FWD_FORCE_BN / GSPN_BWD_FORCE_BN) at the bench layer shapes"""
import ctypes, os, sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gspn_amd import _lib as L
lib = L.lib(); st = L.stream()
shapes = [("SA2-L1", 131072, 68, 67, 64), ("SA2-L3p", 131072, 64, 64, 128),
          ("SA3-L1", 32768, 132, 131, 128), ("SA3-L2", 32768, 128, 128, 128), ("SA3-L3p", 32768, 128, 128, 256),
          ("FP1-L1", 4096, 384, 384, 256), ("FP1-L2", 4096, 256, 256, 128),
          ("FP2-L1", 16384, 192, 192, 128), ("FP2-L2", 16384, 128, 128, 64), ("FP3-L1", 262144, 68, 67, 64)]
dev = 'cuda'
def timeit(fn):
    if fn() != 0: return None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 50
for name, rows, ldx, cin, cout in shapes:
    X = torch.randn(rows, ldx, device=dev); Y = torch.empty(rows, cout, device=dev); W = torch.randn(cin, cout, device=dev)
    isc = torch.rand(cin, device=dev) + 0.5; ish = torch.randn(cin, device=dev); bias = torch.randn(cout, device=dev)
    stats = torch.empty(int(lib.gspn_mlp_fwd_stats_bytes(rows, cout)) // 4 + 4, device=dev)
    def fwd():
        return lib.gspn_mlp_fwd(rows, cin, cout, L.ptr(X), ldx, L.ptr(isc), L.ptr(ish), L.ptr(W), L.ptr(bias), L.ptr(Y), cout, L.ptr(stats), st)
    dZ = torch.randn(rows, cout, device=dev); Yr = torch.randn(rows, cout, device=dev)
    sc = torch.rand(cout, device=dev) + 0.5; sh = torch.randn(cout, device=dev); cA, cB, cC = (torch.randn(cout, device=dev) for _ in range(3))
    a = L.DyArgs(); a.Y, a.ldy = Yr.data_ptr(), cout; a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = dZ.data_ptr(), cout, None, None, 0
    a.scale, a.shift = sc.data_ptr(), sh.data_ptr(); a.cA, a.cB, a.cC = cA.data_ptr(), cB.data_ptr(), cC.data_ptr()
    dX = torch.empty(rows, ldx, device=dev)
    def bwd():
        return lib.gspn_mlp_bwd_data(rows, cin, cout, ctypes.byref(a), L.ptr(W), L.ptr(dX), ldx, st)
    out = []
    for env, fn, ref_t in (("GSPN_FWD_FORCE_BN", fwd, Y), ("GSPN_BWD_FORCE_BN", bwd, dX)):
        os.environ.pop(env, None)
        base = timeit(fn); ref = ref_t.clone()
        res = []
        for bn in (32, 64, 128):
            os.environ[env] = str(bn)
            us = timeit(fn)
            same = bool(torch.equal(ref_t[:, :min(cin, ref_t.shape[1])], ref[:, :min(cin, ref.shape[1])])) if fn is bwd else bool(torch.equal(ref_t, ref))
            res.append("BN%d %.1f%s" % (bn, us, "" if same else "(!)"))
        os.environ.pop(env, None)
        out.append("%s default %.1f | %s" % ("fwd" if fn is fwd else "bwd", base, "  ".join(res)))
    print("%-8s rows %6d %3d->%3d  %s" % (name, rows, cin, cout, "   ||   ".join(out)), flush=True)
