"""cProfile of the host side of one training step (where does the enqueue time go?)"""
import cProfile, pstats, sys, io, time
import numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from gspn_amd import parallel, tf_util
from gspn_amd.fea_extractor import pn2_fea_extractor, pn2_geometry
from gspn_amd.geometry import GeometryStream
dev = torch.device('cuda', 0)
xyz_np, col_np = bench.synth(8, 32768, 0)
xyz = torch.from_numpy(xyz_np).to(dev); col = torch.from_numpy(col_np).to(dev)
gout = torch.randn(8, 32768, 64, device=dev)
store = tf_util.set_variable_store(tf_util.VariableStore(device=dev, seed=1234))
geo = GeometryStream(dev)
st = {"pend": geo.submit(pn2_geometry, xyz), "opt": None, "bucket": None}
def step():
    g = st["pend"].get()
    st["pend"] = geo.submit(pn2_geometry, xyz)
    out = pn2_fea_extractor(xyz, col, 'fea', True, 0.5, geometry=g)
    loss = (out * gout).sum() * (1.0 / out.numel())
    if st["opt"] is not None: st["opt"].zero_grad(set_to_none=True)
    loss.backward()
    if st["bucket"] is None:
        params = store.parameters(); st["bucket"] = parallel.FlatGradBucket(params); st["opt"] = torch.optim.Adam(params, lr=1e-3, foreach=True)
    st["bucket"].all_reduce_mean(); st["opt"].step()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter()
pr.disable(); torch.cuda.synchronize()
print("host ms/step (under cProfile): %.2f" % ((t1 - t0) * 100))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
