"""sha256 of the outputs and gradients of a few shared-MLP stacks (dense, pooled 32 / 256, pre-BN statistics paths) and of one whole extractor step:
run under two builds of the library (GSPN_HIP_LIB) to show that a kernel change left every bit in place"""
import hashlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_mlp import make_params, to_layers
from gspn_amd import mlp as M
h = hashlib.sha256()
for (rows, cin, chans, ns) in ((8192, 6, [32, 32, 64], 32), (65536, 20, [64, 64, 64], None), (16384, 35, [64, 128, 256], 256), (4096, 131, [128, 256], 32), (70000, 8, [32, 36], None)):
    g = torch.Generator().manual_seed(rows)
    ld = (cin + 3) // 4 * 4
    x = torch.randn(rows, ld, generator=g); x[:, cin:] = 0
    layers = to_layers(make_params(chans, cin, seed=7))
    xx = x.cuda().requires_grad_(True)
    out = M.mlp_stack(xx, cin, layers, True, 0.7, pool_ns=ns)
    out.square().sum().backward()
    for t in [out.detach(), xx.grad] + [lp.weights.grad for lp in layers] + [lp.gamma.grad for lp in layers] + [lp.beta.grad for lp in layers] + [lp.moving_mean for lp in layers] + [lp.moving_variance for lp in layers]:
        h.update(t.detach().cpu().numpy().tobytes())
    print(rows, cin, chans, ns, h.hexdigest()[:16], flush=True)
