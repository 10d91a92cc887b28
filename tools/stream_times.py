"""how long does each stream need per step? geometry alone, layers alone (graph replay + eager tail), and both overlapped"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
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
