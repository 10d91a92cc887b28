"""debug: where does the dX error of a bench-size stack sit? (rows near a ReLU kink, or spread over all rows)"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from tests import test_gpu_mlp as T
from oracle import mlp_ref as R
from gspn_amd.mlp import mlp_stack
rows, ld, cin, chans, ns = [(524288, 8, 6, [32, 32, 64], 32), (131072, 68, 67, [64, 64, 128], 32), (32768, 132, 131, [128, 128, 256], 32),
                            (262144, 8, 6, [64, 64, 128], 32)][int(sys.argv[1])]
g = torch.Generator().manual_seed(rows + cin)
x64 = torch.randn(rows, ld, generator=g, dtype=torch.float64); x64[:, cin:] = 0
ps = T.make_params(chans, cin, seed=cin)
layers = T.to_layers(ps)
x = x64.float().cuda().requires_grad_(True)
out = mlp_stack(x, cin, layers, True, 0.7, pool_ns=ns)
xr = x64[:, :cin].cuda().clone().requires_grad_(True)
for p in ps:
    for k in p:
        if torch.is_tensor(p[k]): p[k] = p[k].cuda()
    for k in ("w", "b", "gamma", "beta"): p[k] = p[k].clone().requires_grad_(True)
h = xr; minz = torch.full((rows,), 1e9, device='cuda', dtype=torch.float64)
for p in ps:
    z, _, _ = R.layer(h, p["w"], p["b"], p["gamma"], p["beta"], p["moving_mean"], p["moving_var"], True, 0.7, True, relu=False)
    minz = torch.minimum(minz, z.detach().abs().min(dim=1).values)
    h = torch.relu(z)
full = h.view(rows // ns, ns, -1)
top2 = full.detach().topk(2, dim=1).values
gap = ((top2[:, 0] - top2[:, 1]) / (top2[:, 0].abs() + 1e-3)); gap[top2[:, 0] <= 0] = 1.0
mingap = gap.min(dim=1).values
ref = full.max(dim=1).values
fr = (minz < 2e-5).view(rows // ns, ns).any(1) | (mingap < 1e-5)
go = torch.randn(ref.shape, generator=g, dtype=torch.float64).cuda(); go[fr] = 0
ref.backward(go); out.backward(go.float())
err = (x.grad[:, :cin].double() - xr.grad).abs()
scale = xr.grad.abs().max()
rowerr = err.max(dim=1).values / scale
print("max rel err", float(rowerr.max()), "rows > 1e-4:", int((rowerr > 1e-4).sum()), " > 3e-5:", int((rowerr > 3e-5).sum()), "median", float(rowerr.median()))
bad = torch.nonzero(rowerr > 5e-5).flatten()[:10]
for r in bad.tolist():
    grp = r // ns
    print("row", r, "err", float(rowerr[r]), "min|z| row", float(minz[r]), "group min|z|", float(minz.view(-1, ns)[grp].min()), "group min gap", float(mingap[grp]), "go zeroed", bool(fr[grp]))
