"""ball query: wave-per-query over the L2-resident scene (default) vs LDS-staged tiles shared by a workgroup; same output required"""
import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from gspn_amd import _lib as L
from gspn_amd.tf_sampling import farthest_point_sample, gather_point
lib = L.lib()
for (n, m, r, ns) in [(32768, 2048, 0.2, 32), (32768, 1024, 0.1, 32), (2048, 512, 0.4, 32), (32768, 256, 1.5, 512)]:
    xyz_np, _ = bench.synth(8, n, 0)
    xyz = torch.from_numpy(xyz_np).cuda()
    q = gather_point(xyz, farthest_point_sample(m, xyz))
    out = []
    for name, fn in (("l2-wave", lib.gspn_queryballpoint), ("lds-tile", lib.gspn_queryballpoint_lds)):
        idx = torch.empty(8, m, ns, dtype=torch.int32, device="cuda"); cnt = torch.empty(8, m, dtype=torch.int32, device="cuda")
        run = lambda: L.check(fn(8, n, m, r, ns, L.ptr(xyz), L.ptr(q), L.ptr(idx), L.ptr(cnt), L.stream()), name)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        out.append((name, e0.elapsed_time(e1) / 20 * 1e3, idx.clone(), cnt.clone()))
    same = torch.equal(out[0][2], out[1][2]) and torch.equal(out[0][3], out[1][3])
    print("n=%d m=%d r=%.2f ns=%d : %s %.1f us | %s %.1f us | identical %s" % (n, m, r, ns, out[0][0], out[0][1], out[1][0], out[1][1], same), flush=True)
