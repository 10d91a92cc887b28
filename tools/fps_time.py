"""wall time of the FPS of each SA level of the bench (torch events, 20 launches each)"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from gspn_amd.tf_sampling import farthest_point_sample
for n, m in ((32768, 2048), (2048, 512), (512, 128), (16384, 1024), (8192, 2048)):
    xyz_np, _ = bench.synth(8, n, 0)
    xyz = torch.from_numpy(xyz_np).cuda()
    farthest_point_sample(m, xyz); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        farthest_point_sample(m, xyz)
    e1.record(); torch.cuda.synchronize()
    print("fps 8x%d -> %d: %.1f us" % (n, m, e0.elapsed_time(e1) * 50))
