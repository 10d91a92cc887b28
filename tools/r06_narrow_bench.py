"""the gathered first layer of an SA module with <= 4 feature columns (K = 3 + 3), graph-timed: GSPN_FWD_NARROW=0 (streaming GEMM) / 1 (r06 narrow kernel).
Prints us per launch and the effective TB/s on (rows x cout x 4 B written + 36 B read per row)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gspn_amd import _lib as L
lib = L.lib()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1)
for (b, n, m, ns, cout) in ((8, 32768, 2048, 32, 32), (8, 32768, 1024, 32, 64), (8, 32768, 256, 256, 64), (8, 32768, 256, 512, 64)):
    rows = b * m * ns
    feat = torch.rand(b * n, 4, device=dev, generator=gen)
    # grouped rows of a query come from a neighbourhood: indices clustered like a ball query's (sorted, within a window of the scene)
    base = torch.randint(0, n - 4096, (b, m, 1), device=dev, generator=gen)
    gidx = (base + torch.randint(0, 4096, (b, m, ns), device=dev, generator=gen).sort(dim=2).values + torch.arange(b, device=dev)[:, None, None] * n).int().reshape(-1).contiguous()
    rel = torch.randn(rows, 4, device=dev, generator=gen); rel[:, 3] = 0
    W = torch.randn(6, cout, device=dev, generator=gen); bias = torch.randn(cout, device=dev, generator=gen)
    Y = torch.empty(rows, cout, device=dev)
    stats = torch.empty(int(lib.gspn_mlp_fwd_stats_bytes(rows, cout)) // 4, device=dev)
    ga = L.GatherArgs(feat.data_ptr(), 4, 3, gidx.data_ptr(), rel.data_ptr(), 1)
    run = lambda: L.check(lib.gspn_mlp_fwd_gather(rows, ctypes.byref(ga), cout, L.ptr(W), L.ptr(bias), L.ptr(Y), cout, L.ptr(stats), L.stream()), "fwd_gather")
    ms = min(bench._ev_time(run, 3, 20) for _ in range(3))
    # the same launch writing a DIFFERENT output buffer every time (16 buffers, > 1 GB: nothing of Y stays in the 256 MB Infinity Cache -- the situation inside a step)
    Ys = [torch.empty(rows, cout, device=dev) for _ in range(16)]
    state = [0]
    def run_rot():
        y = Ys[state[0] % 16]; state[0] += 1
        L.check(lib.gspn_mlp_fwd_gather(rows, ctypes.byref(ga), cout, L.ptr(W), L.ptr(bias), L.ptr(y), cout, L.ptr(stats), L.stream()), "fwd_gather")
    ms_rot = min(bench._ev_time(run_rot, 16, 32) for _ in range(3))
    X = torch.cat([rel[:, :3], feat[gidx.long(), :3]], 1).double()
    ref = X @ W.double() + bias.double()
    err = float((Y.double() - ref).abs().max() / ref.abs().max())
    nparts = stats.numel() // (2 * cout)
    st = stats.view(nparts, 2, cout).double().sum(0)
    serr = max(float((st[0] - ref.sum(0)).abs().max() / ref.sum(0).abs().max()), float((st[1] - (ref * ref).sum(0)).abs().max() / (ref * ref).sum(0).abs().max()))
    print("rows %8d x 6 -> %2d : %7.1f us (rotating outputs %7.1f us)  %.2f TB/s  max rel err %.1e  stats err %.1e" % (rows, cout, ms * 1e3, ms_rot * 1e3, rows * (4.0 * cout + 36) / (ms * 1e-3) / 1e12, err, serr), flush=True)
