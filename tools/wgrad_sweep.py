"""sweep the wgrad tile shape / chunk count (GSPN_WGRAD_FORCE) for the small-row layers of the bench stack; prints us per gspn_mlp_bwd_wgrad"""
import ctypes, os, sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gspn_amd import _lib as L
lib = L.lib(); st = L.stream()
shapes = [("SA3-L1", 32768, 132, 131, 128, False), ("SA3-L2", 32768, 128, 128, 128, False), ("SA3-L3p", 32768, 128, 128, 256, True),
          ("FP1-L1", 4096, 384, 384, 256, False), ("FP1-L2", 4096, 256, 256, 128, False),
          ("FP2-L1", 16384, 192, 192, 128, False), ("FP2-L2", 16384, 128, 128, 64, False),
          ("SA1-L1", 524288, 8, 6, 32, False), ("SA1-L2", 524288, 32, 32, 32, False), ("SA1-L3p", 524288, 32, 32, 64, True),
          ("FP3-L1", 262144, 68, 67, 64, False), ("FP3-L2", 262144, 64, 64, 64, False),
          ("SA2-L1", 131072, 68, 67, 64, False), ("SA2-L2", 131072, 64, 64, 64, False), ("SA2-L3p", 131072, 64, 64, 128, True)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if s[0] in sys.argv[1:]]
tiles = [(1, 1), (1, 2), (1, 4), (2, 1), (2, 2), (2, 4), (3, 1), (3, 2), (4, 1), (4, 2)]
dev = 'cuda'
for name, rows, ldx, cin, cout, pooled in shapes:
    X = torch.randn(rows, ldx, device=dev); Y = torch.randn(rows, cout, device=dev)
    isc = torch.rand(cin, device=dev) + 0.5; ish = torch.randn(cin, device=dev)
    mean = torch.randn(cout, device=dev); var = torch.rand(cout, device=dev) + 0.5; gamma = torch.rand(cout, device=dev) + 0.5
    sc = torch.rand(cout, device=dev) + 0.5; sh = torch.randn(cout, device=dev)
    cA, cB, cC = (torch.empty(cout, device=dev) for _ in range(3))
    a = L.DyArgs(); a.Y, a.ldy = Y.data_ptr(), cout
    if pooled:
        ns = 32; dP = torch.randn(rows // ns, cout, device=dev); arg = torch.randint(0, ns, (rows // ns, cout), device=dev, dtype=torch.int32)
        a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = None, 0, dP.data_ptr(), arg.data_ptr(), ns
    else:
        dZ = torch.randn(rows, cout, device=dev); a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = dZ.data_ptr(), cout, None, None, 0
    a.scale, a.shift = sc.data_ptr(), sh.data_ptr(); a.cA, a.cB, a.cC = cA.data_ptr(), cB.data_ptr(), cC.data_ptr()
    dW = torch.empty(cin, cout, device=dev)
    work = torch.empty(64 * 1024 * 1024 + 2 * 1024 * 1024, device=dev)        # 256 MB: any plan fits
    def run():
        return lib.gspn_mlp_bwd_wgrad(rows, cin, cout, ctypes.byref(a), L.ptr(X), ldx, L.ptr(isc), L.ptr(ish), L.ptr(mean), L.ptr(var), L.ptr(gamma), 1e-3, 1, 1,
                                      L.ptr(work), L.ptr(cA), L.ptr(cB), L.ptr(cC), None, None, None, L.ptr(dW), st)
    def timeit():
        if run() != 0: return None
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 100
    os.environ.pop("GSPN_WGRAD_FORCE", None)
    base = timeit()
    ref = dW.clone()
    print("%-8s rows %6d %3d->%3d default %6.1f us" % (name, rows, cin, cout, base), flush=True)
    best = []
    for (mt, nt) in tiles:
        for ch in (0, 4, 8, 16, 32, 64, 112, 224, 448, 672):
            os.environ["GSPN_WGRAD_FORCE"] = "%d,%d,%d" % (mt, nt, ch)
            us = timeit()
            if us is None: continue
            err = float((dW - ref).abs().max() / ref.abs().max())
            best.append((us, mt, nt, ch, err))
    best.sort()
    for us, mt, nt, ch, err in best[:6]:
        print("    %6.1f us  MT %d NT %d chunks %3d  (rel diff to default %.1e)" % (us, mt, nt, ch, err), flush=True)
os.environ.pop("GSPN_WGRAD_FORCE", None)
