"""ball query: wave-per-query over the L2-resident scene vs LDS-staged tiles shared by a workgroup, on U / S / D clouds at the
model's radii (model_rpointnet.py:224-231); same output required.  Graph-timed (20 launches between two HIP events)."""
import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from gspn_amd import _lib as L
from gspn_amd import synth
from gspn_amd.tf_sampling import farthest_point_sample, gather_point
lib = L.lib()
kinds = sys.argv[1:] or ["U", "S", "D"]
for kind in kinds:
    xyz = torch.from_numpy(synth.batch(kind, 8, 32768, 0)).cuda()
    cur = xyz
    for (m, r, ns) in [(2048, 0.2, 32), (512, 0.4, 32), (128, 0.8, 32)]:
        n = cur.shape[1]
        q = gather_point(cur, farthest_point_sample(m, cur))
        out = []
        ws = torch.empty(int(lib.gspn_ball_ws_bytes(8, n, m)), dtype=torch.uint8, device="cuda")
        for name, fn in (("scan (prefix + continuation)", lib.gspn_queryballpoint), ("cell grid + scan", None)):
            idx = torch.empty(8, m, ns, dtype=torch.int32, device="cuda"); cnt = torch.empty(8, m, dtype=torch.int32, device="cuda")
            if fn is None:
                run = lambda: L.check(lib.gspn_queryballpoint_ws(8, n, m, r, ns, L.ptr(cur), L.ptr(q), L.ptr(ws), L.ptr(idx), L.ptr(cnt), L.stream()), name)
            else:
                run = lambda: L.check(fn(8, n, m, r, ns, L.ptr(cur), L.ptr(q), L.ptr(idx), L.ptr(cnt), L.stream()), name)
            us = bench._ev_time(run) * 1e3
            out.append((name, us, idx.clone(), cnt.clone()))
        same = all(torch.equal(out[0][2], o[2]) and torch.equal(out[0][3], o[3]) for o in out[1:])
        full = float((out[0][3] < ns).float().mean())
        print("%s n=%5d m=%4d r=%.1f ns=%d (%.0f%% of the queries find < ns points: full scan) : %s | identical %s" % (
            kind, n, m, r, ns, 100 * full, "  ".join("%s %.1f us" % (o[0], o[1]) for o in out), same), flush=True)
        cur = q
