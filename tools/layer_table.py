"""per-layer table of the shared-MLP GEMM launches of one bench step (eager, HIP events around every launch; gspn_amd.mlp.PROFILE):
kind, shape, microseconds, algorithmic bytes / flops, TB/s, TFLOP/s, and the ratio to the larger of its two floors (6.3 TB/s measured
streaming rate, 157.3 TF).  usage: layer_table.py [reps]"""
import sys
import numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from gspn_amd import mlp as M, tf_util
from gspn_amd.fea_extractor import pn2_fea_extractor, pn2_geometry
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device('cuda', 0)
xyz_np, col_np = bench.synth(8, 32768, 0)
xyz, col = torch.from_numpy(xyz_np).to(dev), torch.from_numpy(col_np).to(dev)
tf_util.set_variable_store(tf_util.VariableStore(device=dev, seed=1234))
G = pn2_geometry(xyz)
gout = torch.randn(8, 32768, 64, device=dev) / (8 * 32768 * 64)
def step():
    for p in tf_util.get_variable_store().parameters():
        p.grad = None
    (pn2_fea_extractor(xyz, col, 'fea', True, 0.5, geometry=G) * gout).sum().backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
M.PROFILE = []
for _ in range(reps):
    step()
torch.cuda.synchronize()
prof, M.PROFILE = M.PROFILE, None
n = len(prof) // reps
tot = {"fwd": 0.0, "wgrad": 0.0, "bwd": 0.0, "fused": 0.0}
print("%-6s %8s %5s %5s %9s %9s %7s %7s %6s" % ("kind", "rows", "cin", "cout", "us", "floor_us", "TB/s", "TF", "x"))
for i in range(n):
    kind, rows, cin, cout = prof[i][:4]
    us = np.median([prof[i + r * n][4].elapsed_time(prof[i + r * n][5]) for r in range(reps)]) * 1e3
    by = 4.0 * rows * {"fwd": cin + cout, "wgrad": cin + 2 * cout, "bwd": 2 * cout + cin, "fused": 2 * cout + 2 * cin}[kind]
    fl = prof[i][6]
    floor = max(by / 6.3e12, fl / 157.3e12) * 1e6
    tot[kind] += us
    print("%-6s %8d %5d %5d %9.1f %9.1f %7.2f %7.1f %6.2f" % (kind, rows, cin, cout, us, floor, by / us / 1e6, fl / us / 1e6, us / floor))
print("totals (us):", {k: round(v, 1) for k, v in tot.items()}, "sum", round(sum(tot.values()), 1))
