"""three_nn with the dense points taken in Morton order vs the given order (kernel time by torch events, 50 launches)"""
import sys, ctypes, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gspn_amd import _lib as L
from gspn_amd.tf_sampling import farthest_point_sample, gather_point
import bench
lib = L.lib()
lib.gspn_threenn_ordered.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 6
xyz_np, _ = bench.synth(8, 32768, 0)
xyz = torch.from_numpy(xyz_np).cuda()
l1 = gather_point(xyz, farthest_point_sample(2048, xyz))
def morton(x):
    lo = x.amin(1, keepdim=True); hi = x.amax(1, keepdim=True)
    q = ((x - lo) / (hi - lo + 1e-9) * 1023).long().clamp_(0, 1023)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[..., 0]) | (spread(q[..., 1]) << 1) | (spread(q[..., 2]) << 2)
order = torch.argsort(morton(xyz), dim=1).int().contiguous()
b, n, m = 8, 32768, 2048
def run(o):
    d = torch.empty(b, n, 3, device='cuda'); i = torch.empty(b, n, 3, dtype=torch.int32, device='cuda')
    f = lambda: L.check(lib.gspn_threenn_ordered(b, n, m, L.ptr(xyz), L.ptr(l1), L.ptr(o) if o is not None else None, L.ptr(d), L.ptr(i), L.stream()), "nn")
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 20, d, i
for _ in range(2):
    t0, d0, i0 = run(None)
    t1, d1, i1 = run(order)
    print("given order %.1f us   morton order %.1f us   equal %s %s" % (t0, t1, bool((d0 == d1).all()), bool((i0 == i1).all())))
# coarser orders: voxel grids of g^3 cells (arbitrary order inside a voxel), and the 16-cell order of the FPS pre-pass
for g in (2, 4, 8, 16, 32):
    lo = xyz.amin(1, keepdim=True); hi = xyz.amax(1, keepdim=True)
    q = ((xyz - lo) / (hi - lo + 1e-9) * g).long().clamp_(0, g - 1)
    key = (q[..., 0] * g + q[..., 1]) * g + q[..., 2]
    o = torch.argsort(key, dim=1).int().contiguous()
    t1, d1, i1 = run(o)
    print("voxel grid %2d^3 (row-major cells): %.1f us  equal %s" % (g, t1, bool((i0 == i1).all())))
ws = torch.empty(int(lib.gspn_fps_cells_ws_bytes(b, n)) // 4, dtype=torch.int32, device='cuda')
L.check(lib.gspn_fps_cells_prepass(b, n, L.ptr(xyz), L.ptr(ws), L.stream()), "pre")
perm = ws[:b * n].view(b, n)
t1, d1, i1 = run(perm)
print("FPS pre-pass order (16 cells): %.1f us  equal %s" % (t1, bool((i0 == i1).all())))
