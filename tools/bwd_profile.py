"""phase breakdown of the register-staged pass B kernel (library built with -DABL_BWD_PROFILE): cycles per (tile, chunk) step in the MFMA
loop, the tile epilogue, commit (transform + LDS writes; waits for the loads), the fetch issue and the barrier"""
import ctypes, os, sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gspn_amd import _lib as L
lib = L.lib(); dev = torch.device('cuda', 0)
for rows, cin, cout in [(262144, 64, 64), (131072, 64, 64), (32768, 128, 128), (16384, 192, 128), (4096, 384, 256), (4096, 256, 128)]:
    Y = torch.randn(rows, cout, device=dev); W = torch.randn(cin, cout, device=dev) * 0.1; Yp = torch.randn(rows, cin, device=dev)
    dX = torch.empty(rows, cin, device=dev); dZ = torch.randn(rows, cout, device=dev)
    one = lambda c, v=1.0: torch.full((c,), v, device=dev)
    a = L.DyArgs(); a.Y, a.ldy = Y.data_ptr(), cout
    a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = dZ.data_ptr(), cout, None, None, 0
    sc, sh, cA, cB, cC = one(cout), one(cout, 0.1), one(cout), one(cout, 0.01), one(cout, 0.0)
    a.scale, a.shift, a.cA, a.cB, a.cC = sc.data_ptr(), sh.data_ptr(), cA.data_ptr(), cB.data_ptr(), cC.data_ptr()
    psc, psh, pm, pv = one(cin), one(cin, 0.1), one(cin, 0.0), one(cin)
    part = torch.zeros(int(lib.gspn_rsum_part_floats(rows, cin)), device=dev); npart = ctypes.c_int(0)
    for _ in range(3):
        L.check(lib.gspn_mlp_bwd_data_ex(rows, cin, cout, ctypes.byref(a), L.ptr(W), 0, cin, L.ptr(dX), cin, None, 0, None, None, 1e-3, 0, 1, None, None,
                                         L.ptr(Yp), cin, L.ptr(psc), L.ptr(psh), L.ptr(pm), L.ptr(pv), 1e-3, L.ptr(part), ctypes.byref(npart), L.stream()), "bwd")
    torch.cuda.synchronize()
    n = min(npart.value, 64)
    v = part[512 * 2 * cin:].view(torch.int64)[:64 * 4 * 8].view(64, 4, 8)[:n].double()
    st = v[..., 7].mean()
    m = v.mean(dim=(0, 1))
    print("bwd %7d x %3d <- %3d: WGs(row) %d steps/WG %.1f | per step: mfma %.0f  epilogue %.0f  commit %.0f  fetch %.0f  barrier %.0f | prologue %.0f  total %.0f cycles" %
          (rows, cin, cout, npart.value, st, m[0] / st, m[1] / st, m[2] / st, m[3] / st, m[4] / st, m[5], m[6]))
