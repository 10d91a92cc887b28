"""Parity at BASELINE.json's full sizes (configs[1] and [2]: 8 scenes x 32768 points): direct comparison with the C oracle where it
finishes in seconds (multi-threaded over scenes), plus size-independent properties of the outputs."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import data as D

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

B, N = 8, 32768


@pytest.fixture(scope="module")
def scenes():
    xyz = D.batch("U", B, N)
    return xyz, torch.from_numpy(xyz).cuda()


@pytest.mark.parametrize("m", [1024, 2048])          # config 1 (SA(1024, ...)) and config 2 (SA1 of pn2_fea_extractor)
def test_fps_full_size_index_exact(scenes, m):
    from gspn_amd.tf_sampling import farthest_point_sample
    xyz, t = scenes
    got = farthest_point_sample(m, t).cpu().numpy()
    ref = O.farthest_point_sample(m, xyz, mt=True)
    np.testing.assert_array_equal(got, ref)
    # properties that do not need the oracle: starts at point 0, no repeats on distinct points, and the greedy invariant --
    # the min-distance of every later pick to the earlier picks never increases
    assert (got[:, 0] == 0).all()
    assert all(len(np.unique(r)) == m for r in got)
    p = torch.from_numpy(xyz[0][got[0]]).cuda().double()
    d = torch.cdist(p, p)
    tri = torch.tril(torch.ones(m, m, dtype=torch.bool, device=d.device), diagonal=-1)
    mind = torch.where(tri, d, torch.full_like(d, float("inf"))).min(dim=1).values[1:]     # pick j vs picks < j
    assert (mind[1:] <= mind[:-1] + 1e-6).all()


@pytest.mark.parametrize("m,radius", [(1024, 0.1), (2048, 0.2)])
def test_ball_query_full_size_index_exact(scenes, m, radius):
    from gspn_amd.tf_grouping import query_ball_point
    from gspn_amd.tf_sampling import farthest_point_sample, gather_point
    xyz, t = scenes
    new_xyz = gather_point(t, farthest_point_sample(m, t))
    idx, cnt = query_ball_point(radius, 32, t, new_xyz)
    ridx, rcnt = O.query_ball_point(radius, 32, xyz, new_xyz.cpu().numpy(), mt=True)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    np.testing.assert_array_equal(cnt.cpu().numpy(), rcnt)
    # properties: the first cnt entries ascend strictly, the tail repeats the first hit, every hit is inside the ball
    i64 = idx.long()
    k = torch.arange(32, device=idx.device)[None, None, :]
    live = k < cnt[..., None]
    asc = (i64[..., 1:] > i64[..., :-1]) | ~live[..., 1:]
    assert asc.all()
    assert (torch.where(live, i64, i64[..., :1]) == i64).all()
    g = torch.gather(t.unsqueeze(1).expand(-1, m, -1, -1), 2, i64.unsqueeze(-1).expand(-1, -1, -1, 3))
    assert ((g - new_xyz.unsqueeze(2)).double().norm(dim=-1) < radius + 1e-6).all()


def test_three_nn_full_size_exact(scenes):
    from gspn_amd.tf_interpolate import three_nn
    from gspn_amd.tf_sampling import farthest_point_sample, gather_point
    xyz, t = scenes
    l1 = gather_point(t, farthest_point_sample(2048, t))
    dist, idx = three_nn(t, l1)
    rd, ri = O.three_nn(xyz[:2], l1[:2].cpu().numpy())                 # the oracle is single-threaded: two scenes
    np.testing.assert_array_equal(idx[:2].cpu().numpy(), ri)
    np.testing.assert_array_equal(dist[:2].cpu().numpy(), rd)
    assert (dist[..., 0] <= dist[..., 1]).all() and (dist[..., 1] <= dist[..., 2]).all()
    # every sampled point is its own nearest neighbour at distance 0
    assert (dist.amin(dim=-1) >= 0).all()


def test_sa_module_full_size_batch_split_invariance(scenes):
    """config 1: SA(1024, 0.1, 32, [64, 64, 128]) forward on 8 x 32768 points; with moving statistics every scene is independent, so the
    batch of 8 must equal the two halves run separately, bit for bit"""
    from gspn_amd import tf_util
    from gspn_amd.pointnet_util import pointnet_sa_module
    xyz, t = scenes
    feat = torch.rand(B, N, 3, device="cuda")
    tf_util.set_variable_store(tf_util.VariableStore(seed=77))
    with torch.no_grad():
        nx, npts, idx = pointnet_sa_module(t, feat, 1024, 0.1, 32, [64, 64, 128], None, False, False, None, 'sa')
        outs = [pointnet_sa_module(t[h:h + 4], feat[h:h + 4], 1024, 0.1, 32, [64, 64, 128], None, False, False, None, 'sa') for h in (0, 4)]
    assert npts.shape == (B, 1024, 128) and torch.isfinite(npts).all()
    assert torch.equal(torch.cat([o[1] for o in outs]), npts)
    assert torch.equal(torch.cat([o[2] for o in outs]), idx)


def test_nn_distance_config4_shape_exact():
    """BASELINE config 3 (C4 of SURVEY 8): Chamfer nn_distance on (B*NUM_SAMPLE, 512, 3) pairs -- 2048 clouds per GPU.  The oracle
    (tf_nndistance_g.cu:5-127 restated) checks a slab of clouds bit for bit; the rest is checked through the definition's invariants."""
    from gspn_amd.tf_nndistance import nn_distance
    rng = np.random.default_rng(11)
    nb, n = 2048, 512
    a = rng.standard_normal((nb, n, 3)).astype(np.float32)
    c = (a[:, rng.permutation(n)] + 0.05 * rng.standard_normal((nb, n, 3))).astype(np.float32)
    d1, i1, d2, i2 = nn_distance(torch.from_numpy(a).cuda(), torch.from_numpy(c).cuda())
    r1, ri1, r2, ri2 = O.nn_distance(a[:96], c[:96])
    np.testing.assert_array_equal(i1[:96].cpu().numpy(), ri1)
    np.testing.assert_array_equal(i2[:96].cpu().numpy(), ri2)
    np.testing.assert_array_equal(d1[:96].cpu().numpy(), r1)
    np.testing.assert_array_equal(d2[:96].cpu().numpy(), r2)
    # invariants on all 2048 clouds: the reported distance is the distance to the reported index, and no other point is closer
    ta, tc = torch.from_numpy(a).cuda().double(), torch.from_numpy(c).cuda().double()
    full = torch.cdist(ta, tc) ** 2
    got = torch.gather(full, 2, i1.long().unsqueeze(-1)).squeeze(-1)
    assert torch.allclose(got, d1.double(), rtol=1e-5, atol=1e-7)
    assert (full.min(dim=2).values >= d1.double() * (1 - 1e-5) - 1e-7).all()
    assert (full.min(dim=1).values >= d2.double() * (1 - 1e-5) - 1e-7).all()


def test_multi_encoding_net_config4_shape(scenes):
    """the context encoder of the proposal head at its real shape (model_rpointnet.py:377: 256 seeds, radii .5/1/1.5, nsample
    256/256/512, mlp [64,128,256] x 3, use_xyz) on 32768-point scenes: ball-query indices exact vs the oracle, outputs finite, and with
    moving statistics the batch splits exactly (every scene independent)"""
    from gspn_amd import tf_util
    from gspn_amd.proposal_head import multi_encoding_net
    from gspn_amd.tf_grouping import query_ball_point
    xyz, t = scenes
    t4, x4 = t[:4].contiguous(), xyz[:4]
    feat = torch.rand(4, N, 3, device="cuda")
    tf_util.set_variable_store(tf_util.VariableStore(seed=5))
    args = (256, [0.5, 1.0, 1.5], [256, 256, 512], [[64, 128, 256]] * 3, [], False, None, 'enc')
    with torch.no_grad():
        new_xyz, new_points, _, fps_idx = multi_encoding_net(t4, feat, *args, use_xyz=True)
        halves = [multi_encoding_net(t4[h:h + 2], feat[h:h + 2], *args, use_xyz=True) for h in (0, 2)]
    assert new_points.shape == (4, 256, 768) and torch.isfinite(new_points).all()
    np.testing.assert_array_equal(fps_idx.cpu().numpy(), O.farthest_point_sample(256, x4, mt=True))
    assert torch.equal(torch.cat([h[1] for h in halves]), new_points)
    idx, cnt = query_ball_point(1.5, 512, t4, new_xyz)
    ridx, rcnt = O.query_ball_point(1.5, 512, x4, new_xyz.cpu().numpy(), mt=True)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    np.testing.assert_array_equal(cnt.cpu().numpy(), rcnt)


def test_knn_full_size_without_the_matrix(scenes):
    """(8, 32768, 2048, k=32): the reference's construction would need a 2 GiB distance tensor.  Every query is independent, so a
    sample of them is checked against the oracle's dense construction; all of them against properties of a k-NN result."""
    from gspn_amd.tf_grouping import knn_point
    from gspn_amd.tf_sampling import farthest_point_sample, gather_point
    xyz, t = scenes
    q = gather_point(t, farthest_point_sample(2048, t))
    val, idx = knn_point(32, t, q)
    assert val.shape == (B, 2048, 32) and idx.shape == (B, 2048, 32)
    qn = q.cpu().numpy()
    pick = np.arange(0, 2048, 97)
    for s in (0, 5):
        rv, ri = O.knn_point(32, xyz[s:s + 1], qn[s:s + 1, pick])
        np.testing.assert_array_equal(idx[s, pick].cpu().numpy(), ri[0])
        np.testing.assert_array_equal(val[s, pick].cpu().numpy(), rv[0])
    # queries are data points: the nearest neighbour is the query itself at distance 0; distances ascend; indices are distinct
    assert (val[..., 0] == 0).all()
    assert (val[..., 1:] >= val[..., :-1]).all()
    srt = idx.long().sort(dim=-1).values
    assert (srt[..., 1:] != srt[..., :-1]).all()
    g = torch.gather(t.unsqueeze(1).expand(-1, 2048, -1, -1), 2, idx.long().unsqueeze(-1).expand(-1, -1, -1, 3))
    d = ((g - q.unsqueeze(2)) ** 2)
    np.testing.assert_array_equal(((d[..., 0] + d[..., 1]) + d[..., 2]).cpu().numpy(), val.cpu().numpy())


def _ref_layers(store, scope, names, h, device):
    """inference-mode conv2d+BN+ReLU stack of oracle/mlp_ref.py (float64) with the store's parameters and moving statistics"""
    from oracle import mlp_ref as R
    from tests.test_gpu_modules import ref_params
    for p in ref_params(store, scope, names):
        q = {k: (v.detach().to(device) if torch.is_tensor(v) else v) for k, v in p.items()}
        h, _, _ = R.layer(h, q["w"], q["b"], q["gamma"], q["beta"], q["moving_mean"], q["moving_var"], False, 0.5)
    return h


def test_fea_extractor_full_size_forward_backward(scenes):
    """BASELINE configs[2] as bench.py runs it: 8 x 32768 points through pn2_fea_extractor (3 SA + 3 FP), forward + backward.
    (a) every index the geometry produces equals the oracle's; (b) one training step is bit-reproducible and all gradients are
    finite; (c) in inference mode (moving statistics: scenes independent) the output of one scene equals the float64 composition
    of oracle/mlp_ref.py over the same indices within 1e-5."""
    from oracle import mlp_ref as R
    from gspn_amd import tf_util
    from gspn_amd.fea_extractor import PN2_SA_SPEC, pn2_fea_extractor, pn2_geometry
    xyz, t = scenes
    col = torch.rand(B, N, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    geo = pn2_geometry(t)
    # (a) geometry vs oracle
    cur, levels = xyz, [xyz]
    for lvl, (npoint, radius, ns) in enumerate(PN2_SA_SPEC):
        new = O.gather_point(cur, O.farthest_point_sample(npoint, cur, mt=True))
        ridx, rcnt = O.query_ball_point(radius, ns, cur, new, mt=True)
        np.testing.assert_array_equal(geo["sa"][lvl].new_xyz.cpu().numpy(), new)
        np.testing.assert_array_equal(geo["sa"][lvl].idx.cpu().numpy(), ridx)
        cur = new
        levels.append(new)
    for fpg, (dense, sparse, nsc) in zip(geo["fp"], [(levels[2], levels[3], B), (levels[1], levels[2], B), (levels[0], levels[1], 1)]):
        rd, ri = O.three_nn(dense[:nsc], sparse[:nsc])                    # the 32768 <- 2048 level: one scene (single-threaded oracle)
        np.testing.assert_array_equal(fpg.idx[:nsc].cpu().numpy(), ri)
        w = R.fp_weights(torch.from_numpy(rd).double())
        assert float((fpg.weight[:nsc].cpu().double() - w).abs().max()) < 1e-6
    # (b) one training step, twice from the same initial state
    runs = []
    for _ in range(2):
        store = tf_util.set_variable_store(tf_util.VariableStore(seed=21))
        out = pn2_fea_extractor(t, col, 'fea', True, 0.5, geometry=geo)
        assert out.shape == (B, N, 64)
        out.square().mean().backward()
        torch.cuda.synchronize()
        runs.append((out.detach().clone(), {n: p.grad.detach().clone() for n, p in store.named_parameters()}))
    assert torch.isfinite(runs[0][0]).all()
    assert torch.equal(runs[0][0], runs[1][0])
    for n, g in runs[0][1].items():
        assert torch.isfinite(g).all(), n
        assert torch.equal(g, runs[1][1][n]), n
    # (c) inference mode on scene 2 vs the float64 composition (parameters + the moving statistics the training step left)
    s = 2
    with torch.no_grad():
        got = pn2_fea_extractor(t[s:s + 1].contiguous(), col[s:s + 1].contiguous(), 'fea', False, None)
    dev = t.device
    pts = [t[s].double(), None, None, None]
    feats = [col[s].double()]
    for lvl, (npoint, radius, ns) in enumerate(PN2_SA_SPEC):
        gi = geo["sa"][lvl].idx[s].long()
        new = geo["sa"][lvl].new_xyz[s].double()
        rows = torch.cat([pts[lvl][gi] - new[:, None, :], feats[lvl][gi]], -1).reshape(npoint * ns, -1)
        h = _ref_layers(store, 'fea/layer%d' % (lvl + 1), ['conv0', 'conv1', 'conv2'], rows, dev)
        feats.append(h.view(npoint, ns, -1).max(1).values)
        pts[lvl + 1] = new
    up = feats[3]
    for k, (dl, names) in enumerate([(2, ['conv_0', 'conv_1']), (1, ['conv_0', 'conv_1']), (0, ['conv_0', 'conv_1', 'conv_2'])]):
        fpg = geo["fp"][k]
        w = fpg.weight[s].double()
        interp = (up[fpg.idx[s].long()] * w[..., None]).sum(1)
        up = _ref_layers(store, 'fea/fa_layer%d' % (k + 1), names, torch.cat([interp, feats[dl]], -1), dev)
    err = float((got[0].double() - up).abs().max() / up.abs().max())
    assert err < 1e-5, err


def _mlp_nodes(root):
    """every _MlpStack autograd node reachable from `root`, keyed by the data_ptr of its first layer's weights"""
    seen, todo, found = {}, [root], {}
    while todo:
        f = todo.pop()
        if f is None or id(f) in seen:
            continue
        seen[id(f)] = f                                  # (keeps the wrapper alive: a collected wrapper's id would be reused)
        if type(f).__name__.startswith("_MlpStack"):
            found[f.spec["layers"][0].weights.data_ptr()] = f
        todo.extend(nf for nf, _ in f.next_functions)
    return found


def test_fea_extractor_training_mode_values_and_gradients_at_full_size(scenes):
    """BASELINE configs[2] in TRAINING mode -- 8 x 32768 points through all 16 layers with batch statistics over the whole batch --
    against the float64 composition of oracle/mlp_ref.py evaluated on the device over the same (oracle-checked, test above) geometry.
      (i)  every module (3 SA, 3 FP) on IDENTICAL INPUTS -- the float64 module fed the float32 tensors the GPU module was fed -- within 1e-5: north_star's bar;
      (ii) the whole 16-layer composition chained in float64 from the raw inputs: the output differs by what six float32 modules compound to
           (measured 2.0e-5 of max |out|; bound 5e-5 -- ANY float32 evaluation, the reference's included, sits this far from float64);
           and EVERY parameter gradient of the module (weights, gamma, beta) under the upstream gradient the GPU backward handed that module within 1e-5,
           nothing silenced: the float64 backward is routed by the ReLU masks / pool winners of the GPU forward's own float32 values
           (tests/test_gpu_mlp.py: check_stack_routed), so what is compared is arithmetic;
      (iii) the chained composition's 48 parameter gradients, through the grouping, pooling, interpolation and skip-link gradients between the stacks
           (measured 6e-5; bound 2e-4: the forward difference of (ii) propagated)."""
    from oracle import mlp_ref as R
    from gspn_amd import tf_util
    from gspn_amd.fea_extractor import PN2_SA_SPEC, pn2_geometry
    from gspn_amd.pointnet_util import pointnet_fp_module, pointnet_sa_module
    xyz, t = scenes
    dev = t.device
    col = torch.rand(B, N, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    geo = pn2_geometry(t)
    store = tf_util.set_variable_store(tf_util.VariableStore(seed=21))
    # pn2_fea_extractor's own body (fea_extractor.py / model_rpointnet.py:209-233), kept open so that the modules' outputs can be looked at
    gpu_feats, cur_xyz, cur_pts = [col], t, col
    with tf_util.variable_scope('fea'):
        lx = [t]
        for lvl, ((npoint, radius, ns), mlp) in enumerate(zip(PN2_SA_SPEC, ([32, 32, 64], [64, 64, 128], [128, 128, 256]))):
            cur_xyz, cur_pts, _ = pointnet_sa_module(cur_xyz, cur_pts, npoint=npoint, radius=radius, nsample=ns, mlp=mlp, mlp2=None, group_all=False,
                                                     is_training=True, bn_decay=0.5, scope='layer%d' % (lvl + 1), geometry=geo["sa"][lvl])
            lx.append(cur_xyz)
            gpu_feats.append(cur_pts)
        gpu_up = [gpu_feats[3]]
        for k, (dl, mlp) in enumerate([(2, [256, 128]), (1, [128, 64]), (0, [64, 64, 64])]):
            gpu_up.append(pointnet_fp_module(lx[dl], lx[dl + 1], gpu_feats[dl], gpu_up[-1], mlp, True, 0.5, scope='fa_layer%d' % (k + 1), geometry=geo["fp"][k]))
    out = gpu_up[-1]
    nodes = _mlp_nodes(out.grad_fn)
    assert len(nodes) == 6

    for tns in gpu_feats[1:] + gpu_up[1:]:
        tns.retain_grad()                                  # the upstream gradient every module receives in the GPU backward

    def stack(scope, names, h, pool_ns, keep):
        node = nodes[store.vars["%s/%s/weights" % (scope, names[0])].data_ptr()]
        assert len(node.saved) == len(names)
        for nm, sv in zip(names, node.saved):
            y, scale, shift = sv[5], sv[8], sv[9]
            g = lambda k: store.vars["%s/%s/%s" % (scope, nm, k)].detach().double()
            w = g("weights")
            q = {"w": w.view(w.shape[-2], w.shape[-1]).clone().requires_grad_(True), "b": g("biases").clone().requires_grad_(True),
                 "gamma": g("bn/gamma").clone().requires_grad_(True), "beta": g("bn/beta").clone().requires_grad_(True)}
            if keep is not None:
                keep["%s/%s" % (scope, nm)] = q
            z, _, _ = R.layer(h, q["w"], q["b"], q["gamma"], q["beta"], None, None, True, 0.5, True, relu=False)
            h = z * ((y.double() * scale.double() + shift.double()) > 0)          # the GPU forward's own decision: sign of fma(y, scale, shift)
        if pool_ns:
            arg = node.arg.long()
            h = h.view(-1, pool_ns, h.shape[1]).gather(1, arg[:, None, :]).squeeze(1)
        return h

    bidx = torch.arange(B, device=dev)[:, None, None]

    def sa(lvl, pts_in, feat_in, keep):
        npoint, radius, ns = PN2_SA_SPEC[lvl]
        gi = geo["sa"][lvl].idx.long()                                              # (B, npoint, ns)
        new = geo["sa"][lvl].new_xyz.double()
        rows = torch.cat([pts_in[bidx, gi] - new[:, :, None, :], feat_in[bidx, gi]], -1).reshape(B * npoint * ns, -1)
        return stack('fea/layer%d' % (lvl + 1), ['conv0', 'conv1', 'conv2'], rows, ns, keep).view(B, npoint, -1)

    def fp(k, names, up_in, skip, keep):
        fpg = geo["fp"][k]
        interp = (up_in[bidx, fpg.idx.long()] * fpg.weight.double()[..., None]).sum(2)      # (B, n_dense, c)
        h = stack('fea/fa_layer%d' % (k + 1), names, torch.cat([interp, skip], -1).reshape(-1, interp.shape[-1] + skip.shape[-1]), None, keep)
        return h.view(B, -1, h.shape[-1])

    def grad_errors(ref_p):
        """per parameter tensor: max |GPU - float64| over the tensor's largest float64 element; a gradient that is ~0 by construction (the beta of a pooled
        top layer whose winners all pass the ReLU shifts an input column of batch-normalised layers by a constant: no effect) is measured against the
        largest gradient element of its own layer instead -- its float32 value is the rounding residue of sums of that size"""
        e = {}
        for key, q in ref_p.items():
            layer_scale = max(float(q[k].grad.abs().max()) for k in ("w", "gamma", "beta"))
            for k, name in (("w", "weights"), ("gamma", "bn/gamma"), ("beta", "bn/beta")):
                got = store.vars["%s/%s" % (key, name)].grad.double().reshape(q[k].shape)
                den = float(q[k].grad.abs().max())
                e["%s/%s" % (key, name)] = float((got - q[k].grad).abs().max()) / (den if den > 1e-6 * layer_scale else layer_scale)
        return e

    rel = lambda a, b_: float((a.detach().double() - b_.detach()).abs().max() / b_.detach().abs().max())
    fp_names = [['conv_0', 'conv_1'], ['conv_0', 'conv_1'], ['conv_0', 'conv_1', 'conv_2']]
    go = torch.randn(out.shape, dtype=torch.float64, device=dev, generator=torch.Generator(device="cuda").manual_seed(9))
    out.backward(go.float())
    # (i) module by module on identical inputs: outputs, and parameter gradients under the upstream gradient the GPU backward handed the module
    per_module, per_module_g = {}, {}
    for lvl in range(3):
        ref_p = {}
        r = sa(lvl, lx[lvl].double(), gpu_feats[lvl].detach().double(), ref_p)
        per_module["sa%d" % (lvl + 1)] = rel(gpu_feats[lvl + 1], r)
        r.backward(gpu_feats[lvl + 1].grad.double())
        per_module_g.update(grad_errors(ref_p))
    for k, dl in enumerate((2, 1, 0)):
        ref_p = {}
        r = fp(k, fp_names[k], gpu_up[k].detach().double(), gpu_feats[dl].detach().double(), ref_p)
        per_module["fp%d" % (k + 1)] = rel(gpu_up[k + 1], r)
        r.backward(go if k == 2 else gpu_up[k + 1].grad.double())
        per_module_g.update(grad_errors(ref_p))
    worst_m = max(per_module_g.items(), key=lambda kv: kv[1])
    print("training-mode full size, per module on identical inputs: outputs %s; worst of %d parameter gradients %s %.2e"
          % ({k: "%.1e" % v for k, v in per_module.items()}, len(per_module_g), worst_m[0], worst_m[1]))
    assert max(per_module.values()) < 1e-5, per_module
    assert worst_m[1] < 1e-5, sorted(per_module_g.items(), key=lambda kv: -kv[1])[:5]
    # (ii) + (iii) the chained composition
    ref_p = {}
    feats = [col.double()]
    for lvl in range(3):
        feats.append(sa(lvl, lx[lvl].double(), feats[lvl], ref_p))
    up = feats[3]
    for k, dl in enumerate((2, 1, 0)):
        up = fp(k, fp_names[k], up, feats[dl], ref_p)
    err = rel(out, up)
    up.backward(go)
    gerr = grad_errors(ref_p)
    worst = max(gerr.items(), key=lambda kv: kv[1])
    print("training-mode full size, chained in float64 from the raw inputs: output rel err %.2e; worst parameter gradient %s %.2e" % (err, worst[0], worst[1]))
    assert err < 5e-5, err
    assert worst[1] < 2e-4, sorted(gerr.items(), key=lambda kv: -kv[1])[:5]


def test_config5_scene_size_through_the_extractor():
    """BASELINE configs[4]'s per-GPU shard: 8 scenes x 65536 points through pn2_fea_extractor, forward + backward.  FPS takes the
    multi-CU kernel (one scene no longer fits a CU); every index equals the oracle's; the step is finite and bit-reproducible."""
    from gspn_amd import tf_util
    from gspn_amd.fea_extractor import PN2_SA_SPEC, pn2_fea_extractor, pn2_geometry
    b, n = 8, 65536
    xyz = D.batch("U", b, n, 300)
    t = torch.from_numpy(xyz).cuda()
    col = torch.rand(b, n, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    geo = pn2_geometry(t)
    cur, levels = xyz, [xyz]
    for lvl, (npoint, radius, ns) in enumerate(PN2_SA_SPEC):
        new = O.gather_point(cur, O.farthest_point_sample(npoint, cur, mt=True))
        ridx, _ = O.query_ball_point(radius, ns, cur, new, mt=True)
        np.testing.assert_array_equal(geo["sa"][lvl].new_xyz.cpu().numpy(), new)
        np.testing.assert_array_equal(geo["sa"][lvl].idx.cpu().numpy(), ridx)
        cur = new
        levels.append(new)
    rd, ri = O.three_nn(levels[0][:1], levels[1][:1])               # the 65536 <- 2048 level, one scene (single-threaded oracle)
    np.testing.assert_array_equal(geo["fp"][2].idx[:1].cpu().numpy(), ri)
    runs = []
    for _ in range(2):
        store = tf_util.set_variable_store(tf_util.VariableStore(seed=8))
        out = pn2_fea_extractor(t, col, 'fea', True, 0.5, geometry=geo)
        assert out.shape == (b, n, 64)
        out.square().mean().backward()
        torch.cuda.synchronize()
        runs.append((out.detach().clone(), {nm: p.grad.detach().clone() for nm, p in store.named_parameters()}))
    assert torch.isfinite(runs[0][0]).all() and torch.equal(runs[0][0], runs[1][0])
    for nm, g in runs[0][1].items():
        assert torch.isfinite(g).all() and torch.equal(g, runs[1][1][nm]), nm


# ---- the clustered regime (SURVEY 8d's S clouds: room surfaces at metre scale, what north_star's "ScanNet scenes" look like) ----
@pytest.fixture(scope="module")
def rooms():
    xyz = D.batch("S", B, N)
    return xyz, torch.from_numpy(xyz).cuda()


def test_ball_query_sparse_regime_continuation_index_exact(rooms):
    """r04: on room scenes most balls hold fewer than nsample points, the reference's scan runs to the end of the cloud and the prefix
    kernel hands the open queries to ball_query_cont_kernel (grouping.hip) -- index-exact against the oracle, counts included, at the
    model's three radii (model_rpointnet.py:224-231), and with a prefix that ends inside the cloud at every level"""
    from gspn_amd.tf_grouping import query_ball_point
    from gspn_amd.tf_sampling import farthest_point_sample, gather_point
    xyz, t = rooms
    cur_np, cur = xyz, t
    for m, radius in ((2048, 0.2), (512, 0.4), (128, 0.8)):
        new_xyz = gather_point(cur, farthest_point_sample(m, cur))
        idx, cnt = query_ball_point(radius, 32, cur, new_xyz)
        ridx, rcnt = O.query_ball_point(radius, 32, cur_np, new_xyz.cpu().numpy(), mt=True)
        np.testing.assert_array_equal(cnt.cpu().numpy(), rcnt)
        np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
        if m == 2048:
            assert float((cnt < 32).float().mean()) > 0.3          # the regime this test is about: open queries that scan all n points
        cur_np, cur = new_xyz.cpu().numpy(), new_xyz


def test_inverse_lists_long_groups_on_clustered_clouds(rooms):
    """r04: a sparse point inside a dense cluster is the nearest neighbour of thousands of dense points -- groups far longer than a wave
    take csr_sort_kernel's radix path (round 3's quadratic loop needed 420 us per launch here); result = stable argsort + bincount"""
    from gspn_amd.fea_extractor import pn2_geometry
    from gspn_amd.geometry import inverse_lists
    _, t = rooms
    G = pn2_geometry(t)
    idx2d = G["fp"][2].idx.reshape(B, -1)
    n = 2048
    order, offsets = inverse_lists(idx2d, n)
    cnt = torch.stack([torch.bincount(idx2d[s].long(), minlength=n) for s in range(B)])
    assert int(cnt.max()) > 256                                     # long groups exist (1291 on these seeds)
    ref = torch.argsort(idx2d.long(), dim=1, stable=True).int()
    off_ref = torch.cat([torch.zeros(B, 1, dtype=torch.long, device=t.device), cnt.cumsum(1)], 1).int()
    assert torch.equal(order, ref) and torch.equal(offsets, off_ref)
    # an adversarial case: every position of a scene in ONE group, and two groups split down the middle
    L = 5000
    one = torch.zeros(2, L, dtype=torch.int32, device=t.device)
    one[1, L // 2:] = 7
    order, offsets = inverse_lists(one, 16)
    assert torch.equal(order, torch.arange(L, dtype=torch.int32, device=t.device).repeat(2, 1))
    assert offsets[0].tolist() == [0] + [L] * 16 and offsets[1].tolist() == [0] + [L // 2] * 7 + [L] * 9


@pytest.mark.parametrize("radius,nsample", [(0.2, 32), (0.05, 8), (0.35, 64), (3.0, 16)])
def test_ball_query_cell_grid_path_equals_the_reference_scan(radius, nsample):
    """r04: clouds of >= 8192 points take gspn_queryballpoint_ws -- a cell grid over the data points answers the queries whose ball holds few
    points (all hits of the 3 x 3 x 3 block, sorted by index), the scan the rest.  Index-exact against the oracle on a cloud built to hit every
    branch: a sparse background (grid answers), two tight clusters (more hits than the grid sorts: handed to the scan), queries that are
    not data points -- some far outside the bounding box (no hit: zero row, count 0) --, a radius larger than the cloud (dense estimate: no grid)."""
    from gspn_amd.tf_grouping import query_ball_point
    rng = np.random.default_rng(77)
    b, n, m = 2, 16384, 700
    xyz = (rng.random((b, n, 3)).astype(np.float32) * np.array([8.0, 6.0, 3.0], np.float32))
    xyz[:, 1000:3000] = (np.array([2.0, 2.0, 1.0], np.float32) + 0.03 * rng.standard_normal((b, 2000, 3))).astype(np.float32)      # a crowded ball
    xyz[:, 9000:9600] = (np.array([6.5, 4.0, 2.0], np.float32) + 0.10 * rng.standard_normal((b, 600, 3))).astype(np.float32)
    q = np.concatenate([np.stack([xyz[s][rng.integers(0, n, size=500)] for s in range(b)]),
                        (rng.random((b, 150, 3)).astype(np.float32) * np.array([8.0, 6.0, 3.0], np.float32)),
                        (rng.random((b, 50, 3)).astype(np.float32) * 40.0 - 15.0)], axis=1).astype(np.float32)
    idx, cnt = query_ball_point(radius, nsample, torch.from_numpy(xyz).cuda(), torch.from_numpy(q).cuda())
    ridx, rcnt = O.query_ball_point(radius, nsample, xyz, q, mt=True)
    np.testing.assert_array_equal(cnt.cpu().numpy(), rcnt)
    got = idx.cpu().numpy()
    hit = rcnt > 0                                          # (rows without a hit are zero-filled here, uninitialised in the reference)
    np.testing.assert_array_equal(got[hit], ridx[hit])
    assert (got[~hit] == 0).all()
