"""the roofline kernel alone: FPS of SA level 1 (8 x n -> 2048), a few launches (target of the PMC passes).
usage: fps_only.py [launches] [n]   (n = 32768: single-CU cell kernel; n > 32768: multi-CU kernel)"""
import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from gspn_amd.tf_sampling import farthest_point_sample
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
xyz_np, _ = bench.synth(8, n, 0)
xyz = torch.from_numpy(xyz_np).cuda()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    idx = farthest_point_sample(2048, xyz)
torch.cuda.synchronize()
print("fps ok", int(idx.sum()))
