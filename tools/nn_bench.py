"""three_nn at the bench's FP shapes (torch events after a clock warm-up, us per call)"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gspn_amd import tf_interpolate
from gspn_amd.tf_sampling import farthest_point_sample, gather_point
import bench
dev = 'cuda'
_w = torch.randn(4096, 4096, device=dev)
for _ in range(200): _w = (_w @ _w).clamp_(-1, 1)
torch.cuda.synchronize()
xyz_np, _ = bench.synth(8, 32768, 0)
xyz = torch.from_numpy(xyz_np).to(dev)
l1 = gather_point(xyz, farthest_point_sample(2048, xyz))
l2 = gather_point(l1, farthest_point_sample(512, l1))
l3 = gather_point(l2, farthest_point_sample(128, l2))
for name, dense, sparse in (("FP3 32768 <- 2048", xyz, l1), ("FP2 2048 <- 512", l1, l2), ("FP1 512 <- 128", l2, l3)):
    tf_interpolate.three_nn(dense, sparse); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): tf_interpolate.three_nn(dense, sparse)
    e1.record(); torch.cuda.synchronize()
    print("%-20s %.1f us" % (name, e0.elapsed_time(e1) * 20))
