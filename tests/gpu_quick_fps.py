"""ad-hoc: time the FPS kernel at the bench shapes (hipEvent timing through torch)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from gspn_amd.tf_sampling import farthest_point_sample
from gspn_amd.tf_grouping import query_ball_point
from gspn_amd.tf_sampling import gather_point
from tests import data as D
for (b, n, m) in [(8, 32768, 1024), (8, 32768, 2048), (8, 16384, 1024), (8, 2048, 512), (8, 512, 128)]:
    x = torch.from_numpy(D.batch("U", b, n)).cuda()
    farthest_point_sample(m, x); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): idx = farthest_point_sample(m, x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    alg = 20.0 * b * (m - 1) * n + 4 * b * m
    print(f"FPS b={b} n={n} m={m}: {ms:.3f} ms  {ms*1e3/(m-1):.3f} us/round  alg {alg/1e9:.2f} GB -> {alg/ms/1e9:.2f} TB/s ({alg/ms/1e9/8:.1%} of 8TB/s)")
    q = gather_point(x, idx)
    for r in (0.1, 0.2):
        query_ball_point(r, 32, x, q); torch.cuda.synchronize()
        e0.record()
        for _ in range(5): query_ball_point(r, 32, x, q)
        e1.record(); torch.cuda.synchronize()
        print(f"   ball r={r} ns=32: {e0.elapsed_time(e1)/5*1e3:.1f} us")
