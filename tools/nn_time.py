"""three_nn of the bench's dense feature-propagation level (8 x 32768 dense points <- 2048 sampled ones, the spatial scan order of the
sampling level, as gspn_amd.geometry runs it) and of the two small levels: graph-timed microseconds; indices checked against the plain
(unordered) entry point"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gspn_amd import _lib as L
from gspn_amd.fea_extractor import pn2_geometry
from gspn_amd.tf_interpolate import three_nn
dev = torch.device('cuda', 0)
xyz_np, _ = bench.synth(8, 32768, 0)
xyz = torch.from_numpy(xyz_np).to(dev)
def timeit(f, reps=10):
    for _ in range(3): f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
from gspn_amd import tf_sampling as S
fps1, order = S.farthest_point_sample(2048, xyz, return_order=True)      # the spatial scan order the sampling pre-pass leaves
new1 = S.gather_point(xyz, fps1)
new2 = S.gather_point(new1, S.farthest_point_sample(512, new1))
new3 = S.gather_point(new2, S.farthest_point_sample(128, new2))
for name, a, b_, o in (("32768 <- 2048", xyz, new1, order), ("32768 <- 2048 (no order)", xyz, new1, None), ("2048 <- 512", new1, new2, None), ("512 <- 128", new2, new3, None)):
    d0, i0 = three_nn(a, b_, order=o)
    d1, i1 = three_nn(a, b_)
    torch.cuda.synchronize()
    assert torch.equal(i0, i1) and torch.equal(d0, d1)
    print("three_nn %-26s %8.1f us" % (name, timeit(lambda: three_nn(a, b_, order=o))), flush=True)
