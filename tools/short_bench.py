"""The short layers' GEMM launches one by one (r05): correctness against an fp64 product and graph-timed microseconds, for the library as
the environment configures it (GSPN_FWD_SHORT=0/1, GSPN_FWD_SHORT_MT/_NT, GSPN_BWD_SHORT..., GSPN_WGRAD_SHORT...).
usage: short_bench.py [fwd|bwd|wgrad|all]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gspn_amd import _lib as L
lib = L.lib()
dev = torch.device("cuda", 0)
what = sys.argv[1] if len(sys.argv) > 1 else "all"

def ev_time(fn, warm=3, reps=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
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

# (rows, cin, cout, act, pool)   the short forward launches of one bench step
FWD = [(32768, 128, 128, True, False), (32768, 128, 256, True, True), (4096, 384, 256, False, False), (4096, 256, 128, True, False),
       (16384, 192, 128, False, False), (16384, 128, 64, True, False), (4096, 128, 128, False, False), (16384, 64, 64, False, False)]
gen = torch.Generator(device=dev).manual_seed(3)

def run_fwd():
    print("%-28s %9s %9s %10s %10s" % ("fwd rows x cin -> cout", "us", "floor_us", "max_err", "stat_err"))
    tot = 0.0
    for rows, cin, cout, act, pool in FWD:
        X = torch.randn(rows, cin, device=dev, generator=gen)
        W = torch.randn(cin, cout, device=dev, generator=gen) / cin ** 0.5
        bias = torch.randn(cout, device=dev, generator=gen) * 0.1
        sc = (torch.rand(cin, device=dev, generator=gen) + 0.5) if act else None
        sh = (torch.randn(cin, device=dev, generator=gen) * 0.3) if act else None
        Y = torch.empty(rows, cout, device=dev)
        nst = int(lib.gspn_mlp_fwd_stats_bytes(rows, cout)) // 4
        stats = torch.full((nst,), float("nan"), device=dev)
        vmax = torch.empty(rows // 32, cout, device=dev) if pool else None
        amax = torch.empty(rows // 32, cout, dtype=torch.int32, device=dev) if pool else None
        def fn():
            if pool:
                L.check(lib.gspn_mlp_fwd_pool32(rows, cin, cout, L.ptr(X), cin, L.ptr(sc), L.ptr(sh), L.ptr(W), L.ptr(bias), L.ptr(Y), cout, L.ptr(stats), L.ptr(vmax), L.ptr(amax), L.stream()), "fwd")
            else:
                L.check(lib.gspn_mlp_fwd(rows, cin, cout, L.ptr(X), cin, L.ptr(sc), L.ptr(sh), L.ptr(W), L.ptr(bias), L.ptr(Y), cout, L.ptr(stats), L.stream()), "fwd")
        fn(); torch.cuda.synchronize()
        A = X.double()
        if act:
            A = torch.relu((X * sc + sh).double())       # two fp32 roundings, as the kernel
        ref = A @ W.double() + bias.double()
        err = float((Y.double() - ref).abs().max() / ref.abs().max())
        st = stats.view(-1, 2, cout).double().sum(0)
        serr = max(float((st[0] - ref.sum(0)).abs().max() / ref.abs().sum(0).max()), float((st[1] - (ref * ref).sum(0)).abs().max() / (ref * ref).sum(0).max()))
        perr = ""
        if pool:
            g = Y.view(rows // 32, 32, cout)
            mx, am = g.max(1)
            ok = bool(torch.equal(mx, vmax)) and bool(torch.equal(g.gather(1, amax.long().unsqueeze(1)).squeeze(1), vmax))
            first = bool((amax.long() <= am).all())      # torch's max returns some arg; ours must be the FIRST row reaching the maximum
            eq = (g == mx.unsqueeze(1))
            firstrow = eq.float().argmax(1)
            perr = " pool_ok=%s first=%s" % (ok, bool(torch.equal(firstrow.int(), amax)))
        us = ev_time(fn)
        floor = max(2.0 * rows * cin * cout / 157.3e12, 4.0 * rows * (cin + cout) / 6.3e12) * 1e6
        tot += us
        print("%-28s %9.1f %9.1f %10.2e %10.2e%s" % ("%d x %d -> %d%s%s" % (rows, cin, cout, " act" if act else "", " pool" if pool else ""), us, floor, err, serr, perr))
    print("fwd total %.1f us" % tot)

if what in ("fwd", "all"):
    run_fwd()

# (rows, cin, cout, act, pool_ns)   the known-coefficient pass A launches of the short layers
WGRAD = [(4096, 384, 256, False, 0), (16384, 192, 128, False, 0), (32768, 128, 128, True, 0), (32768, 128, 256, True, 32),
         (4096, 128, 128, False, 0), (16384, 64, 64, False, 0), (8192, 64, 128, True, 0)]

def run_wgrad():
    print("%-30s %9s %9s %10s" % ("wgrad rows x cin^T x cout", "us", "floor_us", "max_err"))
    tot = 0.0
    for rows, cin, cout, act, ns in WGRAD:
        X = torch.randn(rows, cin, device=dev, generator=gen)
        Y = torch.randn(rows, cout, device=dev, generator=gen)
        sc = (torch.rand(cin, device=dev, generator=gen) + 0.5) if act else None
        sh = (torch.randn(cin, device=dev, generator=gen) * 0.3) if act else None
        scale = torch.rand(cout, device=dev, generator=gen) + 0.5
        shift = torch.randn(cout, device=dev, generator=gen) * 0.3
        cA = torch.randn(cout, device=dev, generator=gen)
        cB = torch.randn(cout, device=dev, generator=gen) * 0.1
        cC = torch.randn(cout, device=dev, generator=gen) * 0.1
        a = L.DyArgs()
        a.Y, a.ldy = Y.data_ptr(), cout
        if ns:
            dP = torch.randn(rows // ns, cout, device=dev, generator=gen)
            arg = torch.randint(0, ns, (rows // ns, cout), device=dev, generator=gen, dtype=torch.int32)
            a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = None, 0, dP.data_ptr(), arg.data_ptr(), ns
            dz = torch.zeros(rows // ns, ns, cout, device=dev)
            dz.scatter_(1, arg.long().unsqueeze(1), dP.unsqueeze(1))
            dz = dz.view(rows, cout)
        else:
            dz = torch.randn(rows, cout, device=dev, generator=gen)
            a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = dz.data_ptr(), cout, None, None, 0
        a.scale, a.shift = scale.data_ptr(), shift.data_ptr()
        a.cA, a.cB, a.cC = cA.data_ptr(), cB.data_ptr(), cC.data_ptr()
        work = torch.empty(int(lib.gspn_mlp_bwd_work_bytes(rows, cin, cout)) // 4 + 4, device=dev)
        dW = torch.empty(cin, cout, device=dev)
        def fn():
            L.check(lib.gspn_mlp_bwd_wgrad_known(rows, cin, cout, ctypes.byref(a), L.ptr(X), cin, L.ptr(sc), L.ptr(sh), None, L.ptr(work), L.ptr(dW), L.stream()), "wgrad_known")
        fn(); torch.cuda.synchronize()
        A = torch.relu((X * sc + sh).double()) if act else X.double()
        dyh = torch.where(Y * scale > -shift, dz, torch.zeros_like(dz)).double()
        dY = cA.double() * dyh + cB.double() * Y.double() + cC.double()
        ref = A.t() @ dY
        err = float((dW.double() - ref).abs().max() / ref.abs().max())
        us = ev_time(fn)
        floor = max(2.0 * rows * cin * cout / 157.3e12, 4.0 * rows * (cin + 2 * cout) / 6.3e12) * 1e6
        tot += us
        print("%-30s %9.1f %9.1f %10.2e" % ("%d x %d^T x %d%s%s (+dW sum)" % (rows, cin, cout, " act" if act else "", " pool%d" % ns if ns else ""), us, floor, err))
    print("wgrad total %.1f us (each figure = pass A + the stand-alone dW reduction launch)" % tot)

if what in ("wgrad", "all"):
    run_wgrad()
