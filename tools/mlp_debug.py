"""per-layer gradient errors of mlp_stack vs the fp64 restatement (debug aid)"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))); sys.path.insert(0, 'tests')
from oracle import mlp_ref as R
from test_gpu_mlp import make_params, to_layers, rel_err
from gspn_amd.mlp import mlp_stack
cases = [(4096, 6, 6, [32, 32, 64], 32), (2048, 67, 67, [64, 64, 128], 32), (1024, 131, 131, [128, 128, 256], 32), (768, 384, 384, [256, 128], None),
         (1000, 67, 67, [64, 64, 64], None), (4096, 8, 6, [32, 32, 64], 32), (4096, 32, 32, [32], None), (4096, 32, 32, [64], 32), (4096, 32, 32, [64], 64), (4096, 32, 32, [64], 16)]
for rows, ld, cin, chans, ns in cases:
    for training in (True, False):
        g = torch.Generator().manual_seed(rows + cin)
        x64 = torch.randn(rows, ld, generator=g, dtype=torch.float64); x64[:, cin:] = 0
        ps = make_params(chans, cin, seed=cin)
        layers = to_layers(ps)
        x = x64.float().cuda().requires_grad_(True)
        out = mlp_stack(x, cin, layers, training, 0.7, pool_ns=ns)
        xr = x64[:, :cin].clone().requires_grad_(True)
        for p in ps:
            for k in ("w", "b", "gamma", "beta"):
                p[k] = p[k].clone().requires_grad_(True)
        ref, moving = R.stack(xr, ps, training, 0.7, ns)
        go = torch.randn(ref.shape, generator=g, dtype=torch.float64)
        ref.backward(go); out.backward(go.float().cuda())
        msg = ["out %.1e dx %.1e" % (rel_err(out, ref), rel_err(x.grad[:, :cin], xr.grad))]
        for lp, p in zip(layers, ps):
            msg.append("[dW %.1e dg %.1e db %.1e]" % (rel_err(lp.weights.grad, p["w"].grad), rel_err(lp.gamma.grad, p["gamma"].grad), rel_err(lp.beta.grad, p["beta"].grad)))
        print(rows, ld, cin, chans, ns, training, " ".join(msg), flush=True)
