"""the small dependent kernels of the captured step, stand-alone in a hipGraph chain (20 launches, best of 3): what each costs against the
1.6 us a trivial dependent node costs (tools/r05_launch_floor.py)"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gspn_amd import _lib as L
lib = L.lib()
dev = torch.device("cuda", 0)

def ev_time(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3

print("%-60s %8s" % ("kernel", "us"))
for rows, c in ((524288, 32), (524288, 64), (131072, 128), (32768, 256), (4096, 128), (262144, 64)):
    nst = int(lib.gspn_mlp_fwd_stats_bytes(rows, c)) // 4
    stats = torch.randn(nst, device=dev)
    gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    mm, mv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    mean, var, scale, shift = (torch.empty(c, device=dev) for _ in range(4))
    fn = lambda: L.check(lib.gspn_bn_finalize(rows, c, L.ptr(stats), L.ptr(gamma), L.ptr(beta), 1e-3, 0.5, 1, L.ptr(mm), L.ptr(mv), L.ptr(mean), L.ptr(var), L.ptr(scale), L.ptr(shift), L.stream()), "fin")
    print("%-60s %8.2f" % ("bn_finalize rows %d c %d (nparts %d)" % (rows, c, nst // (2 * c)), ev_time(fn)))
for rows, c in ((262144, 64), (32768, 128), (4096, 256)):
    npf = int(lib.gspn_rsum_part_floats(rows, c))
    nparts = npf // (2 * c)
    part = torch.randn(npf, device=dev)
    mean, var, gamma = torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.ones(c, device=dev)
    outs = [torch.empty(c, device=dev) for _ in range(6)]
    fn = lambda: L.check(lib.gspn_mlp_bwd_coef(rows, c, nparts, L.ptr(part), L.ptr(mean), L.ptr(var), L.ptr(gamma), 1e-3, *[L.ptr(o) for o in outs], L.stream()), "coef")
    print("%-60s %8.2f" % ("bwd_coef rows %d c %d (nparts %d)" % (rows, c, nparts), ev_time(fn)))
for groups, c in ((16384, 64), (4096, 128), (1024, 256)):
    vmax = torch.randn(groups, c, device=dev); amax = torch.randint(0, 32, (groups, c), dtype=torch.int32, device=dev)
    Y = torch.randn(groups * 32, c, device=dev)
    sc, sh = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    out = torch.empty(groups, c, device=dev); arg = torch.empty(groups, c, dtype=torch.int32, device=dev)
    fn = lambda: L.check(lib.gspn_pool32_select(groups, c, L.ptr(vmax), L.ptr(amax), L.ptr(Y), c, L.ptr(sc), L.ptr(sh), L.ptr(out), L.ptr(arg), L.stream()), "sel")
    print("%-60s %8.2f" % ("pool32_select groups %d c %d" % (groups, c), ev_time(fn)))
    dP = torch.randn(groups, c, device=dev)
    mean, var = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    part = torch.empty(int(lib.gspn_rsum_part_floats(groups * 32, c)), device=dev)
    npart = ctypes.c_int(0)
    fn = lambda: L.check(lib.gspn_pool_rsum(groups, 32, c, L.ptr(dP), L.ptr(arg), L.ptr(vmax), 0, L.ptr(sc), L.ptr(sh), L.ptr(mean), L.ptr(var), 1e-3, L.ptr(part), ctypes.byref(npart), L.stream()), "rsum")
    print("%-60s %8.2f" % ("pool_rsum groups %d c %d" % (groups, c), ev_time(fn)))
for rows, c in ((4096, 128), (16384, 64)):
    Y = torch.randn(rows, c, device=dev); sc, sh = torch.rand(c, device=dev), torch.randn(c, device=dev); out = torch.empty(rows, c, device=dev)
    fn = lambda: L.check(lib.gspn_bnrelu_apply(rows, c, L.ptr(Y), c, L.ptr(sc), L.ptr(sh), L.ptr(out), c, L.stream()), "apply")
    print("%-60s %8.2f" % ("bnrelu_apply rows %d c %d" % (rows, c), ev_time(fn)))
