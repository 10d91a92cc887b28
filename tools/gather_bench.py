"""first SA layer, materialised rows (sa_group_concat + mlp_fwd / wgrad) vs gathered rows (mlp_fwd_gather / wgrad_gather), stand-alone"""
import ctypes, sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gspn_amd import _lib as L
from gspn_amd.geometry import sa_geometry
lib = L.lib()


def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (b, n, c, m, r, ns, cout) in [(8, 2048, 64, 512, 0.4, 32, 64), (8, 32768, 3, 2048, 0.2, 32, 32), (8, 512, 128, 128, 0.8, 32, 128)]:
    g = torch.Generator(device="cuda").manual_seed(0)
    xyz = torch.rand(b, n, 3, device="cuda", generator=g)
    pts = torch.randn(b, n, c, device="cuda", generator=g)
    geo = sa_geometry(xyz, m, r, ns)
    rows = b * m * ns
    cin = 3 + c
    ld = (cin + 3) // 4 * 4
    W = torch.randn(cin, cout, device="cuda") * 0.1
    bias = torch.zeros(cout, device="cuda")
    Y = torch.empty(rows, cout, device="cuda")
    X = torch.empty(rows, ld, device="cuda")
    stats = torch.empty(int(lib.gspn_mlp_fwd_stats_bytes(rows, cout)) // 4, device="cuda")
    st = L.stream()
    feat = pts if c % 4 == 0 else torch.nn.functional.pad(pts, (0, 4 - c % 4))
    feat = feat.reshape(b * n, -1).contiguous()
    ga = L.GatherArgs(feat.data_ptr(), feat.shape[1], c, geo.gidx.data_ptr(), geo.rel.data_ptr(), 1)

    def concat():
        L.check(lib.gspn_sa_group_concat(b, n, c, m, ns, L.ptr(xyz), L.ptr(geo.new_xyz), L.ptr(pts), L.ptr(geo.idx), 1, ld, L.ptr(X), st), "c")

    def fwd():
        L.check(lib.gspn_mlp_fwd(rows, cin, cout, L.ptr(X), ld, None, None, L.ptr(W), L.ptr(bias), L.ptr(Y), cout, L.ptr(stats), st), "f")

    def fwdg():
        return lib.gspn_mlp_fwd_gather(rows, ctypes.byref(ga), cout, L.ptr(W), L.ptr(bias), L.ptr(Y), cout, L.ptr(stats), st)

    concat(); fwd(); Y0 = Y.clone()
    rc = fwdg()
    line = "b=%d n=%d c=%d rows=%d %d->%d | concat %.1f us  fwd %.1f us" % (b, n, c, rows, cin, cout, timeit(concat), timeit(fwd))
    if rc == 0:
        err = float((Y - Y0).abs().max() / Y0.abs().max())
        line += " | fwd_gather %.1f us (rel diff %.1e)" % (timeit(fwdg), err)
    else:
        line += " | fwd_gather rc=%d" % rc
    print(line, flush=True)
