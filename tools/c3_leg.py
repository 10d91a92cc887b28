"""BASELINE configs[3] per-GPU shard (multi_encoding_net + Chamfer, fwd+bwd) alone, a few steps: target of rocprofv3 --stats"""
import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from gspn_amd import tf_util
from gspn_amd.proposal_head import chamfer_recons_loss, multi_encoding_net
dev = torch.device("cuda", 0)
xyz_np, col_np = bench.synth(8, 32768, 0)
xyz, col = torch.from_numpy(xyz_np).to(dev), torch.from_numpy(col_np).to(dev)
tf_util.set_variable_store(tf_util.VariableStore(device=dev, seed=3))
b = 8
gen = torch.Generator(device=dev).manual_seed(9)
pred0 = torch.randn(b * 256, 512, 3, device=dev, generator=gen)
gt = torch.randn(b * 256, 512, 3, device=dev, generator=gen)
mask = (torch.rand(b * 256, device=dev, generator=gen) > 0.2).float()
col_g = col.clone().requires_grad_(True)
import os
SIDE = torch.cuda.Stream() if os.environ.get("C3_SIDE") == "1" else None
def step():
    for p_ in tf_util.get_variable_store().parameters():
        p_.grad = None
    pred = pred0.clone().requires_grad_(True)
    if SIDE is not None:                      # the Chamfer term on its own stream, beside the encoder (its backward follows it there)
        SIDE.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(SIDE):
            ch = chamfer_recons_loss(pred, gt, mask)
    _, new_points, _, _ = multi_encoding_net(xyz, col_g, 256, [0.5, 1.0, 1.5], [256, 256, 512], [[64, 128, 256]] * 3, [], True, 0.5, 'c3', use_xyz=True)
    if SIDE is not None:
        torch.cuda.current_stream().wait_stream(SIDE)
    else:
        ch = chamfer_recons_loss(pred, gt, mask)
    loss = new_points.mean() + ch
    loss.backward()
    col_g.grad = None
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for _ in range(2): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize()
print("ms per step %.3f" % ((time.perf_counter() - t0) / n * 1e3))
