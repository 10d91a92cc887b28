"""CPU composition of the benchmark graph (pn2_fea_extractor, models/model_rpointnet.py:209-233) for
bench.py's `cpu_baseline` leg and for tests.  TEST / BASELINE INFRASTRUCTURE ONLY.

Geometry ops (FPS, gather, ball query, grouping, 3-NN, interpolation and their gradients) run through
the C oracle (oracle/gspn_oracle.c: the restatement of the reference kernels; the reference's own FPS /
ball query have no CPU kernel at all).  The shared MLP is a stand-in: torch CPU fp32 autograd of
relu(BN(xW+b)) + max-pool, because the reference delegates that arithmetic to TensorFlow, which is not
available here -- hence `kind: "port"`.
"""
import time

import numpy as np
import torch

from . import oracle as O

SA_SPEC = [(2048, 0.2, 32, [32, 32, 64]), (512, 0.4, 32, [64, 64, 128]), (128, 0.8, 32, [128, 128, 256])]
FP_SPEC = [[256, 128], [128, 64], [64, 64, 64]]


def _mlp_params(chans, cin, gen):
    ps = []
    for c in chans:
        lim = (6.0 / (cin + c)) ** 0.5
        ps.append([((torch.rand(cin, c, generator=gen) * 2 - 1) * lim).requires_grad_(True), torch.zeros(c, requires_grad=True),
                   torch.ones(c, requires_grad=True), torch.zeros(c, requires_grad=True)])
        cin = c
    return ps


def _mlp(x, ps):
    for w, b, gamma, beta in ps:
        y = x @ w + b
        mean = y.mean(0)
        var = y.var(0, unbiased=False)
        inv = torch.rsqrt(var + 1e-3) * gamma
        x = torch.relu(y * inv + (beta - mean * inv))
    return x


class _Group(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx_np):
        ctx.idx, ctx.shape = idx_np, points.shape
        return torch.from_numpy(O.group_point(points.numpy(), idx_np))

    @staticmethod
    def backward(ctx, g):
        return torch.from_numpy(O.group_point_grad(np.zeros(ctx.shape, np.float32), ctx.idx, g.contiguous().numpy())), None


class _Interp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx_np, w_np):
        ctx.idx, ctx.w, ctx.shape = idx_np, w_np, points.shape
        return torch.from_numpy(O.three_interpolate(points.numpy(), idx_np, w_np))

    @staticmethod
    def backward(ctx, g):
        return torch.from_numpy(O.three_interpolate_grad(np.zeros(ctx.shape, np.float32), ctx.idx, ctx.w, g.contiguous().numpy())), None, None


def run_step(xyz, feat, seed=1234, mt=False):
    """one fwd+bwd pass of the 3xSA + 3xFP stack on CPU; xyz (b,n,3), feat (b,n,c) float32 numpy. Returns seconds.
    mt: OpenMP inside the oracle loops (scenes for FPS and the scatter-add gradients, scene x query for ball query / grouping /
    3-NN / interpolation); the torch thread count for the MLP stand-in is the caller's (torch.set_num_threads)."""
    O.set_mt(mt)
    try:
        return _run_step(xyz, feat, seed, mt)
    finally:
        O.set_mt(False)


def _run_step(xyz, feat, seed, mt):
    gen = torch.Generator().manual_seed(seed)
    t0 = time.perf_counter()
    b = xyz.shape[0]
    l_xyz, l_pts = [xyz], [torch.from_numpy(feat)]
    for (npoint, radius, ns, mlp) in SA_SPEC:
        cur, pts = l_xyz[-1], l_pts[-1]
        new_xyz = O.gather_point(cur, O.farthest_point_sample(npoint, cur, mt=mt))
        idx, _ = O.query_ball_point(radius, ns, cur, new_xyz, mt=mt)
        gx = torch.from_numpy(O.group_point(cur, idx) - new_xyz[:, :, None, :])
        gp = _Group.apply(pts, idx)
        rows = torch.cat([gx, gp], -1).reshape(-1, 3 + pts.shape[2])
        ps = _mlp_params(mlp, rows.shape[1], gen)
        out = _mlp(rows, ps).view(b * npoint, ns, mlp[-1]).max(1).values.view(b, npoint, mlp[-1])
        l_xyz.append(new_xyz)
        l_pts.append(out)
    feats = l_pts[3]
    for lvl, mlp in zip((2, 1, 0), FP_SPEC):
        dist, idx = O.three_nn(l_xyz[lvl], l_xyz[lvl + 1])
        dist = np.maximum(dist, 1e-10)
        w = ((1.0 / dist) / (1.0 / dist).sum(2, keepdims=True)).astype(np.float32)
        interp = _Interp.apply(feats, idx, w)
        cat = torch.cat([interp, l_pts[lvl]], 2)
        ps = _mlp_params(mlp, cat.shape[2], gen)
        feats = _mlp(cat.reshape(-1, cat.shape[2]), ps).view(b, cat.shape[1], mlp[-1])
    feats.square().mean().backward()
    return time.perf_counter() - t0
