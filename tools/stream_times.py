"""how long does each stream need per step? geometry alone, layers alone (graph replay + eager tail), and both overlapped"""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from gspn_amd import parallel, tf_util
from gspn_amd.fea_extractor import pn2_fea_extractor, pn2_geometry
from gspn_amd.geometry import GeometryStream
from gspn_amd.graph import CapturedStep, copy_into
dev = torch.device('cuda', 0)
xyz_np, col_np = bench.synth(8, 32768, 0)
xyz = torch.from_numpy(xyz_np).to(dev); col = torch.from_numpy(col_np).to(dev)
gout = torch.randn(8, 32768, 64, device=dev)
store = tf_util.set_variable_store(tf_util.VariableStore(device=dev, seed=1234))
G = pn2_geometry(xyz)
st = {}
def fwd_bwd():
    for p in store.parameters(): p.grad = None
    out = pn2_fea_extractor(xyz, col, 'fea', True, 0.5, geometry=G)
    loss = (out * gout).sum() * (1.0 / out.numel())
    loss.backward()
    if "bucket" not in st:
        st["bucket"] = parallel.FlatGradBucket(store.parameters()); st["opt"] = torch.optim.Adam(store.parameters(), lr=1e-3, fused=True)
    st["bucket"].flatten()
    return loss
fwd_bwd(); st["opt"].step()
cap = CapturedStep(fwd_bwd)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("geometry alone (eager, main stream): %.3f ms" % timeit(lambda: pn2_geometry(xyz)))
print("layers graph replay alone:           %.3f ms" % timeit(lambda: cap.replay()))
print("adam step alone:                     %.3f ms" % timeit(lambda: st["opt"].step()))
print("graph + adam:                        %.3f ms" % timeit(lambda: (cap.replay(), st["opt"].step())))
geo = GeometryStream(dev)
def both():
    p = geo.submit(pn2_geometry, xyz); cap.replay(); st["opt"].step(); p.get()
print("geometry || (graph + adam):          %.3f ms" % timeit(both))
from gspn_amd.tf_sampling import farthest_point_sample, gather_point
from gspn_amd.tf_interpolate import three_nn
from gspn_amd.tf_grouping import query_ball_point
l1 = G["sa"][0].new_xyz; l2 = G["sa"][1].new_xyz; l3 = G["sa"][2].new_xyz
def side(fn):
    def f():
        p = geo.submit(fn, xyz); cap.replay(); st["opt"].step(); p.get()
    return f
print("layers || FPS1 only:                 %.3f ms" % timeit(side(lambda x: farthest_point_sample(2048, x))))
print("layers || FPS1+FPS2+FPS3:            %.3f ms" % timeit(side(lambda x: (farthest_point_sample(2048, x), farthest_point_sample(512, l1), farthest_point_sample(128, l2)))))
print("layers || 3 x three_nn:              %.3f ms" % timeit(side(lambda x: (three_nn(l2, l3), three_nn(l1, l2), three_nn(x, l1)))))
print("layers || 3 x ball query:            %.3f ms" % timeit(side(lambda x: (query_ball_point(0.2, 32, x, l1), query_ball_point(0.4, 32, l1, l2), query_ball_point(0.8, 32, l2, l3)))))
print("three_nn x3 alone:                   %.3f ms" % timeit(lambda: (three_nn(l2, l3), three_nn(l1, l2), three_nn(xyz, l1))))
print("FPS1 alone:                          %.3f ms" % timeit(lambda: farthest_point_sample(2048, xyz)))
x1 = xyz[:1].contiguous(); x4 = xyz[:4].contiguous()
print("layers || FPS1 of 1 scene:           %.3f ms" % timeit(side(lambda x: farthest_point_sample(2048, x1))))
print("layers || FPS1 of 4 scenes:          %.3f ms" % timeit(side(lambda x: farthest_point_sample(2048, x4))))
big = torch.empty(64 * 1024 * 1024, device=dev)
print("layers || 10 x fill 256MB:           %.3f ms" % timeit(side(lambda x: [big.fill_(1.0) for _ in range(10)])))
print("10 x fill 256MB alone:               %.3f ms" % timeit(lambda: [big.fill_(1.0) for _ in range(10)]))
def eager_layers():
    for p in store.parameters(): p.grad = None
    fwd_bwd()
print("layers eager (no graph) alone:       %.3f ms" % timeit(eager_layers))
def eager_side():
    p = geo.submit(lambda x: farthest_point_sample(2048, x), xyz); eager_layers(); p.get()
print("layers eager || FPS1:                %.3f ms" % timeit(eager_side))
lo_pri, hi_pri = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range (lowest, highest):", lo_pri, hi_pri)
hi = torch.cuda.Stream(device=dev, priority=-1)
torch.cuda.synchronize()
with torch.cuda.stream(hi):
    cap_hi = CapturedStep(fwd_bwd)
    def side_hi(fn):
        def f():
            p = geo.submit(fn, xyz); cap_hi.replay(); st["opt"].step(); p.get()
        return f
    print("hi-pri layers alone:                 %.3f ms" % timeit(lambda: (cap_hi.replay(), st["opt"].step())))
    print("hi-pri layers || FPS1:               %.3f ms" % timeit(side_hi(lambda x: farthest_point_sample(2048, x))))
    print("hi-pri layers || full geometry:      %.3f ms" % timeit(side_hi(pn2_geometry)))
import ctypes
spin = ctypes.CDLL('tools/libspin_probe.so')
spin.launch_spin.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
outb = torch.zeros(4, device=dev)
def spin_side(which, blocks, threads, iters):
    def fn(x):
        spin.launch_spin(which, blocks, threads, iters, outb.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return fn
print("--- layers (graph + adam) beside synthetic ~2.5 ms side kernels ---")
for name, which, blocks, threads, iters in (("hold 128 KB LDS, sleeping, 8 wg x 64", 9, 8, 64, 6000000), ("hold 128 KB LDS, sleeping, 16 wg x 64", 9, 16, 64, 6000000),
                                            ("hold all VGPRs, sleeping, 8 wg x 1024", 10, 8, 1024, 6000000), ("s_sleep 8 wg x 64", 1, 8, 64, 6000000), ("lds+barrier 8 wg x 1024 (128 KB LDS each)", 2, 8, 1024, 6000000),
                                            ("valu loop + barrier 8 wg x 1024", 4, 8, 1024, 5200), ("valu loop + s_sleep, 128 VGPRs, 8 wg x 1024", 6, 8, 1024, 5200),
                                            ("valu loop 8 wg x 256", 3, 8, 256, 20000)):
    print("  side = %-46s alone %.3f ms; layers beside it: %.3f ms" % (name, timeit(lambda: spin_side(which, blocks, threads, iters)(None), 3), timeit(side(spin_side(which, blocks, threads, iters)))))
