"""phase breakdown of the streaming forward kernel (library built with -DABL_FWD_PROFILE): cycles per tile iteration spent waiting for the
tile (vmcnt + barrier), issuing the next tile's DMA, in the previous tile's epilogue (stores + statistics) and in the k loop"""
import os, sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gspn_amd import _lib as L
lib = L.lib(); dev = torch.device('cuda', 0)
for rows, cin, cout in [(262144, 64, 64), (524288, 32, 64), (524288, 32, 32), (131072, 64, 128), (131072, 64, 64)]:
    X = torch.randn(rows, cin, device=dev); Y = torch.empty(rows, cout, device=dev)
    W = torch.randn(cin, cout, device=dev) * 0.1; bias = torch.zeros(cout, device=dev)
    sc = torch.rand(cin, device=dev) + 0.5; sh = torch.randn(cin, device=dev) * 0.1
    nst = int(lib.gspn_mlp_fwd_stats_bytes(rows, cout)) // 4
    stats = torch.zeros(max(nst, 1 << 20), device=dev)
    for _ in range(3):
        L.check(lib.gspn_mlp_fwd(rows, cin, cout, L.ptr(X), cin, L.ptr(sc), L.ptr(sh), L.ptr(W), L.ptr(bias), L.ptr(Y), cout, L.ptr(stats), L.stream()), "fwd")
    torch.cuda.synchronize()
    v = stats.view(torch.int64)[:64 * 4 * 8].view(64, 4, 8).double()      # first 64 workgroups x 4 waves
    it = v[..., 5].mean()
    tot = v[..., 4].mean()
    ph = v[..., :4].mean(dim=(0, 1))
    print("fwd %7d x %3d -> %3d: tiles/WG %.1f, cycles per tile: wait %.0f  issue %.0f  epilogue %.0f  k-loop %.0f  | total/tile %.0f (100 MHz ticks x? see ratio)" %
          (rows, cin, cout, it, *(ph / it).tolist(), tot / it))
