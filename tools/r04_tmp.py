import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gspn_amd import tf_sampling as TS
xyz_np, _ = bench.synth(8, 32768, 0)
xyz = torch.from_numpy(xyz_np).cuda()
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for m in (256, 1024, 2048):
    TS.FPS_MULTI_FORCE = False
    ref = TS.farthest_point_sample(m, xyz)
    us = t(lambda: TS.farthest_point_sample(m, xyz))
    print("8 x 32768 -> %4d  cell kernel (one CU per scene) %7.1f us  (%.2f us per pick)" % (m, us, us / m), flush=True)
    for G in (2, 4, 8, 16):
        TS.FPS_MULTI_FORCE, TS.FPS_MULTI_G = True, G
        try:
            out = TS.farthest_point_sample(m, xyz)
            us = t(lambda: TS.farthest_point_sample(m, xyz))
            print("                   multi-CU G = %2d                  %7.1f us  (%.2f us per pick)  identical: %s" % (G, us, us / m, bool(torch.equal(out, ref))), flush=True)
        except Exception as e:
            print("                   multi-CU G = %2d: %s" % (G, str(e)[:100]))
