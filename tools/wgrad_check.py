"""direct check of gspn_mlp_bwd_wgrad against fp64 torch: r0, r1, g3 (read from the workspace) and dW"""
import ctypes, sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gspn_amd import _lib as L
lib = L.lib(); st = L.stream()
torch.manual_seed(0)
def run(rows, ldx, cin, cout, ns, act, training=1):
    dev = 'cuda'
    X = torch.randn(rows, ldx, device=dev) + 0.5
    Y = torch.randn(rows, cout, device=dev) * 1.5 + 0.3
    mean = Y.double().mean(0).float(); var = Y.double().var(0, unbiased=False).float()
    gamma = torch.rand(cout, device=dev) + 0.5; beta = torch.rand(cout, device=dev) - 0.5
    eps = 1e-3
    rstd = 1.0 / torch.sqrt(var.double() + eps)
    scale = (gamma.double() * rstd).float(); shift = (beta.double() - mean.double() * gamma.double() * rstd).float()
    isc = (torch.rand(cin, device=dev) + 0.5) if act else None
    ish = (torch.rand(cin, device=dev) - 0.3) if act else None
    cA = torch.empty(cout, device=dev); cB = torch.empty(cout, device=dev); cC = torch.empty(cout, device=dev)
    a = L.DyArgs(); a.Y, a.ldy = Y.data_ptr(), cout
    if ns:
        dP = torch.randn(rows // ns, cout, device=dev); arg = torch.randint(0, ns, (rows // ns, cout), device=dev, dtype=torch.int32)
        a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = None, 0, dP.data_ptr(), arg.data_ptr(), ns
        dz = torch.zeros(rows // ns, ns, cout, device=dev, dtype=torch.float64)
        dz.scatter_(1, arg.long().unsqueeze(1), dP.double().unsqueeze(1))
        dz = dz.reshape(rows, cout)
    else:
        dZ = torch.randn(rows, cout, device=dev); a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = dZ.data_ptr(), cout, None, None, 0
        dz = dZ.double()
    a.scale, a.shift, a.cA, a.cB, a.cC = scale.data_ptr(), shift.data_ptr(), cA.data_ptr(), cB.data_ptr(), cC.data_ptr()
    nw = int(lib.gspn_mlp_bwd_work_bytes(rows, cin, cout)) // 4 + 4
    work = torch.empty(nw, device=dev)
    dW = torch.empty(cin, cout, device=dev); dg = torch.empty(cout, device=dev); db = torch.empty(cout, device=dev); dbias = torch.empty(cout, device=dev)
    L.check(lib.gspn_mlp_bwd_wgrad(rows, cin, cout, ctypes.byref(a), L.ptr(X), ldx, L.ptr(isc), L.ptr(ish), L.ptr(mean), L.ptr(var), L.ptr(gamma), eps, 1, training,
                                   L.ptr(work), L.ptr(cA), L.ptr(cB), L.ptr(cC), L.ptr(dg), L.ptr(db), L.ptr(dbias), L.ptr(dW), st), "w")
    torch.cuda.synchronize()
    A = X[:, :cin].double()
    if act: A = torch.relu(A * isc.double() + ish.double())
    z = Y.double() * scale.double() + shift.double()
    dyh = dz * (z > 0)
    xh = (Y.double() - mean.double()) * rstd
    r0 = dyh.sum(0); r1 = (dyh * xh).sum(0); g3 = A.sum(0)
    G1 = A.t() @ dyh; Gx = A.t() @ xh
    R = float(rows)
    if training:
        dWr = (gamma.double() * rstd) * (G1 - torch.outer(g3, r0) / R - Gx * (r1 / R))
    else:
        dWr = (gamma.double() * rstd) * G1
    red = work[:4 * cout].view(torch.float64)[: 2 * cout] if False else work.view(torch.uint8)[: 16 * cout].view(torch.float64)
    g3w = work.view(torch.uint8)[16 * cout: 16 * cout + 4 * cin].view(torch.float32)
    e = lambda x, y: float((x.double() - y).abs().max() / (y.abs().max() + 1e-30))
    if rows == 4096 and cin == 32 and cout == 64:
        pp = work.view(torch.uint8)[21632: 21632 + 4 * 32 * 2 * cin * cout].view(torch.float32).view(32, 2, cin, cout).double().sum(0)
        print("   G1 err %.2e  Gx err %.2e" % (e(pp[0], G1), e(pp[1], Gx)), "Gx max", float(Gx.abs().max()), "worst idx", (pp[1]-Gx).abs().argmax().item(), float((pp[1]-Gx).abs().max()))
        d = (pp[1] - Gx).abs(); print("   Gx err by row m:", d.max(1).values[:8].tolist(), " by col n:", d.max(0).values[::8].tolist())
    print("rows %6d ld %3d %3d->%3d ns %s act %d tr %d | r0 %.1e r1 %.1e g3 %.1e dW %.1e dgamma %.1e" % (rows, ldx, cin, cout, ns, act, training,
          e(red[:cout], r0), e(red[cout:], r1), e(g3w, g3), e(dW, dWr), e(dg, r1)), flush=True)
for cfg in [(4096, 32, 32, 64, 32, 1), (4096, 32, 32, 64, 32, 0), (4096, 32, 32, 64, 0, 1)]: run(*cfg)
for cfg in [             (1024, 128, 128, 128, 32, 1), (1000, 64, 64, 64, 0, 1), (1000, 64, 64, 64, 0, 0), (2048, 64, 64, 128, 32, 1), (2048, 64, 64, 128, 16, 1), (4096, 32, 32, 64, 16, 1),
            (4096, 32, 32, 32, 16, 1), (4096, 32, 32, 32, 64, 1), (520, 68, 67, 64, 0, 1), (520, 8, 6, 32, 0, 0), (200, 384, 384, 256, 0, 1), (4100, 96, 96, 32, 0, 1), (4100, 96, 96, 64, 0, 1), (4100, 96, 96, 96, 0, 1)]:
    run(*cfg)
    if cfg[5] == 1 and cfg[0] == 4096: run(*cfg, training=0)
