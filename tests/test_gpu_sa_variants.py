"""GPU parity of the branches of utils/pointnet_util.py that the fused SA path does not take: sample_and_group (:17-54),
sample_and_group_all (:57-82), group_all / knn / pooling in {avg, weighted_avg, min, max_and_avg} / mlp2 of pointnet_sa_module
(:103-139), coordinates that carry a gradient, the 'losses' collection of tf_util (:24-49).  Geometry against the C oracle
(bit-exact), layers against the float64 restatement (1e-5 forward, 1e-4 gradients)."""
import numpy as np
import pytest
import torch

from oracle import mlp_ref as R
from oracle import oracle as O
from tests import data as D
from tests.test_gpu_modules import fresh_store, ref_params, rel_err

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("kind,b,n,c,npoint,radius,ns,knn,use_xyz", [
    ("U", 2, 2048, 5, 128, 0.2, 16, False, True), ("D", 2, 1500, 3, 100, 0.3, 32, False, False), ("U", 1, 1000, 0, 64, 0.25, 8, False, True),
    ("U", 2, 2048, 4, 128, None, 16, True, True), ("S", 1, 3000, 2, 50, None, 32, True, False),
])
def test_sample_and_group_matches_oracle(kind, b, n, c, npoint, radius, ns, knn, use_xyz):
    from gspn_amd.pointnet_util import sample_and_group
    xyz = D.batch(kind, b, n, 4)
    pts = np.random.default_rng(3).standard_normal((b, n, c)).astype(np.float32) if c else None
    new_xyz, new_points, idx, grouped_xyz = sample_and_group(npoint, radius, ns, dev(xyz), dev(pts) if c else None, None, knn, use_xyz)
    rnew = O.gather_point(xyz, O.farthest_point_sample(npoint, xyz))
    ridx = O.knn_point(ns, xyz, rnew)[1] if knn else O.query_ball_point(radius, ns, xyz, rnew)[0]
    rgx = O.group_point(xyz, ridx) - rnew[:, :, None, :]                       # pointnet_util.py:41-42
    if c:
        rgp = O.group_point(pts, ridx)
        rnp = np.concatenate([rgx, rgp], -1) if use_xyz else rgp               # :48 xyz FIRST
    else:
        rnp = rgx                                                              # :50
    np.testing.assert_array_equal(new_xyz.cpu().numpy(), rnew)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    np.testing.assert_array_equal(grouped_xyz.cpu().numpy(), rgx)
    np.testing.assert_array_equal(new_points.cpu().numpy(), rnp)


@pytest.mark.parametrize("c,use_xyz", [(4, True), (4, False), (0, True)])
def test_sample_and_group_all(c, use_xyz):
    from gspn_amd.pointnet_util import sample_and_group_all
    b, n = 3, 257
    xyz = D.batch("U", b, n)
    pts = np.random.default_rng(1).random((b, n, c)).astype(np.float32) if c else None
    new_xyz, new_points, idx, grouped_xyz = sample_and_group_all(dev(xyz), dev(pts) if c else None, use_xyz)
    assert new_xyz.shape == (b, 1, 3) and float(new_xyz.abs().max()) == 0.0                         # :70
    np.testing.assert_array_equal(idx.cpu().numpy(), np.tile(np.arange(n, dtype=np.int32).reshape(1, 1, n), (b, 1, 1)))   # :71
    np.testing.assert_array_equal(grouped_xyz.cpu().numpy(), xyz.reshape(b, 1, n, 3))                # :72
    want = xyz if not c else (np.concatenate([xyz, pts], 2) if use_xyz else pts)                     # :73-80
    np.testing.assert_array_equal(new_points.cpu().numpy(), want.reshape(b, 1, n, -1))


def ref_sa(store, scope, xyz, pts, npoint, radius, ns, mlp, mlp2, group_all, pooling, knn, use_xyz, decay, xyz_grad=False, bn=True):
    """float64 composition of pointnet_util.py:103-139 on oracle geometry; returns (new_xyz, out, idx, leaves)"""
    b, n, _ = xyz.shape
    x64 = torch.from_numpy(xyz).double().requires_grad_(xyz_grad)
    p64 = torch.from_numpy(pts).double().requires_grad_(True) if pts is not None else None
    if group_all:
        ns = n
        ridx = np.tile(np.arange(n, dtype=np.int32).reshape(1, 1, n), (b, 1, 1))
        rnew = np.zeros((b, 1, 3), np.float32)
        gx = x64.view(b, 1, n, 3)
        rows = gx if p64 is None else (torch.cat([x64, p64], 2) if use_xyz else p64).unsqueeze(1)
        npoint = 1
    else:
        fidx = O.farthest_point_sample(npoint, xyz)
        rnew = O.gather_point(xyz, fidx)
        ridx = O.knn_point(ns, xyz, rnew)[1] if knn else O.query_ball_point(radius, ns, xyz, rnew)[0]
        gi = torch.from_numpy(ridx.astype(np.int64))
        bi = torch.arange(b)[:, None, None].expand_as(gi)
        new64 = x64[torch.arange(b)[:, None], torch.from_numpy(fidx.astype(np.int64))]
        gx = x64[bi, gi] - new64[:, :, None, :]
        rows = gx if p64 is None else (torch.cat([gx, p64[bi, gi]], -1) if use_xyz else p64[bi, gi])
    cin = rows.shape[-1]
    h = rows.reshape(-1, cin)

    def params(names):
        if bn:
            ps_ = ref_params(store, scope, names)
            for p in ps_:
                p["moving_mean"] = torch.zeros_like(p["moving_mean"])
                p["moving_var"] = torch.ones_like(p["moving_var"])
            return ps_
        ps_ = []
        for nm in names:
            w = store.vars["%s/%s/weights" % (scope, nm)].detach().double().cpu()
            ps_.append({"w": w.view(w.shape[-2], w.shape[-1]).clone().requires_grad_(True),
                        "b": store.vars["%s/%s/biases" % (scope, nm)].detach().double().cpu().clone().requires_grad_(True),
                        "gamma": None, "beta": None, "moving_mean": None, "moving_var": None})
        return ps_

    ps = params(['conv%d' % i for i in range(len(mlp))])
    for p in ps:
        h, _, _ = R.layer(h, p["w"], p["b"], p["gamma"], p["beta"], p["moving_mean"], p["moving_var"], True, decay, bn)
    h = h.view(b, npoint, ns, -1)
    if pooling == 'avg':
        h = h.mean(2, keepdim=True)
    elif pooling == 'weighted_avg':
        d = gx.norm(dim=-1, keepdim=True)
        e = torch.exp(-d * 5)
        h = (h * (e / e.sum(2, keepdim=True))).sum(2, keepdim=True)
    elif pooling == 'max':
        h = h.max(2, keepdim=True).values
    elif pooling == 'min':
        h = (-h).max(2, keepdim=True).values                                   # :126 (the reference does not negate back)
    elif pooling == 'max_and_avg':
        h = torch.cat([h.max(2, keepdim=True).values, h.mean(2, keepdim=True)], -1)   # :128-130
    ps2 = params(['conv_post_%d' % i for i in range(len(mlp2 or []))])
    c2 = h.shape[-1]
    h2 = h.reshape(-1, c2)
    for p in ps2:
        h2, _, _ = R.layer(h2, p["w"], p["b"], p["gamma"], p["beta"], p["moving_mean"], p["moving_var"], True, decay, bn)
    out = h2.view(b, npoint, -1)
    return rnew, out, ridx, dict(xyz=x64, pts=p64, ps=ps, ps2=ps2)


@pytest.mark.parametrize("pooling,knn,group_all,mlp2,use_xyz,c,bn", [
    ("avg", False, False, None, True, 6, True), ("weighted_avg", False, False, None, True, 6, True), ("min", False, False, None, True, 0, True),
    ("max_and_avg", False, False, [24], True, 6, True), ("max", True, False, None, True, 6, True),
    # group_all leaves ONE row per scene for mlp2: batch statistics over b rows amplify any rounding by ~1/sqrt(eps) per layer, in the
    # reference as much as here -- so mlp2 behind group_all is checked without BN, and group_all with BN without an mlp2
    ("max", False, True, [32, 16], True, 6, False), ("max", False, True, None, True, 6, True), ("max", False, True, None, False, 6, True),
    ("max", False, False, [20], True, 6, True), ("max", False, False, None, False, 6, True), ("avg", True, False, [8], False, 5, True),
    ("max", False, False, [12], True, 6, False),
])
def test_sa_module_unfused_branches_match_oracle(pooling, knn, group_all, mlp2, use_xyz, c, bn):
    from gspn_amd.pointnet_util import pointnet_sa_module
    store = fresh_store(77)
    b, n, npoint, radius, ns, mlp = 2, 1024, 64, 0.3, 16, [16, 32]
    xyz = D.batch("U", b, n, 8)
    pts = np.random.default_rng(2).standard_normal((b, n, c)).astype(np.float32) if c else None
    tp = dev(pts).requires_grad_(True) if c else None
    new_xyz, new_points, idx = pointnet_sa_module(dev(xyz), tp, npoint, radius, ns, mlp, mlp2, group_all, True, 0.5, 'sa', bn=bn, pooling=pooling,
                                                  knn=knn, use_xyz=use_xyz)
    rnew, ref, ridx, leaves = ref_sa(store, 'sa', xyz, pts, npoint, radius, ns, mlp, mlp2, group_all, pooling, knn, use_xyz, 0.5, bn=bn)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    np.testing.assert_array_equal(new_xyz.cpu().numpy(), rnew)
    assert new_points.shape == ref.shape
    assert rel_err(new_points, ref) < 1e-5
    g = torch.from_numpy(np.random.default_rng(5).standard_normal(tuple(ref.shape)))
    ref.backward(g)
    new_points.backward(g.float().cuda())
    for i, p in enumerate(leaves["ps"]):
        assert rel_err(store.vars['sa/conv%d/weights' % i].grad.view(p["w"].shape), p["w"].grad) < 1e-4
        if bn:
            assert rel_err(store.vars['sa/conv%d/bn/gamma' % i].grad, p["gamma"].grad) < 1e-4
    for i, p in enumerate(leaves["ps2"]):
        assert rel_err(store.vars['sa/conv_post_%d/weights' % i].grad.view(p["w"].shape), p["w"].grad) < 1e-4
    if c:
        assert rel_err(tp.grad, leaves["pts"].grad) < 1e-4


def test_sa_module_propagates_the_gradient_of_coordinates():
    """xyz.requires_grad (predicted / shifted coordinates): the module must not take the fused path, whose coordinates are
    constants; d(new_points)/d(xyz) flows through group_point_grad and gather_point_grad like in the reference"""
    from gspn_amd.pointnet_util import pointnet_sa_module, group_concat
    store = fresh_store(31)
    b, n, c, npoint, radius, ns, mlp = 2, 512, 4, 32, 0.4, 16, [16, 16]
    xyz = D.batch("U", b, n, 21)
    pts = np.random.default_rng(6).standard_normal((b, n, c)).astype(np.float32)
    tx = dev(xyz).requires_grad_(True)
    tp = dev(pts).requires_grad_(True)
    new_xyz, new_points, idx = pointnet_sa_module(tx, tp, npoint, radius, ns, mlp, None, False, True, 0.5, 'sa')
    rnew, ref, ridx, leaves = ref_sa(store, 'sa', xyz, pts, npoint, radius, ns, mlp, None, False, 'max', False, True, 0.5, xyz_grad=True)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    assert rel_err(new_points, ref) < 1e-5
    g = torch.from_numpy(np.random.default_rng(7).standard_normal(tuple(ref.shape)))
    ref.backward(g)
    new_points.backward(g.float().cuda())
    assert tx.grad is not None and float(tx.grad.abs().max()) > 0
    assert rel_err(tx.grad, leaves["xyz"].grad) < 1e-4
    assert rel_err(tp.grad, leaves["pts"].grad) < 1e-4
    with pytest.raises(ValueError):
        group_concat(tx, new_xyz.detach(), tp, idx)


def test_weight_decay_collection_holds_each_variable_once():
    """tf_util.py:24-49: one l2_loss(var) * wd term per decayed variable, however many forward passes have run"""
    from gspn_amd import tf_util
    store = fresh_store(3)
    x = torch.randn(2, 8, 4, 5, device="cuda")
    for _ in range(3):
        with tf_util.variable_scope('net'):
            y = tf_util.conv2d(x, 7, [1, 1], scope='c0', weight_decay=0.01, bn=True, is_training=True)
            tf_util.conv2d(y, 3, [1, 1], scope='c1', weight_decay=0.1, bn=False, is_training=True)
            tf_util.conv2d(y, 3, [1, 1], scope='c2', bn=False, is_training=True)
    w0, w1 = store.vars['net/c0/weights'], store.vars['net/c1/weights']
    want = 0.5 * 0.01 * float((w0.double() ** 2).sum()) + 0.5 * 0.1 * float((w1.double() ** 2).sum())
    assert len(store.losses) == 2
    assert abs(float(tf_util.weight_decay_loss()) - want) < 1e-6 * max(1.0, want)
    fresh_store(4)
    assert float(tf_util.weight_decay_loss()) == 0.0          # a new store starts with an empty collection


@pytest.mark.parametrize("n,m,offset", [(6000, 700, 0.0), (20000, 1500, 0.0), (6000, 700, 100.0), (3000, 2, 0.0), (5000, 64, -37.5)])
def test_three_nn_metre_scale_rooms(n, m, offset):
    """cloud S: an 8 x 6 x 3 m room (ScanNet scale, not the unit cube), optionally far from the origin: the kernel's early-rejection
    margin must stay conservative when |p|^2 - 2 p.q cancels badly; exact indices and distances against the oracle"""
    from gspn_amd.tf_interpolate import three_nn
    xyz1 = D.batch("S", 2, n, 40) + np.float32(offset)
    xyz2 = O.gather_point(xyz1, O.farthest_point_sample(m, xyz1))
    d, i = three_nn(dev(xyz1), dev(xyz2))
    rd, ri = O.three_nn(xyz1, xyz2)
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(d.cpu().numpy(), rd)


@pytest.mark.parametrize("kind,b,n,c,npoint,radius,ns,mlp,expect_gather", [
    ("U", 2, 4096, 3, 512, 0.2, 32, [32, 32, 64], True),         # SA1-shaped: 3 colour channels, padded to a 16-byte feature row
    ("D", 2, 2048, 64, 256, 0.4, 32, [64, 64, 128], True),       # SA2-shaped
    ("U", 3, 1500, 8, 100, 0.3, 16, [32, 48], True),
    ("U", 2, 1024, 20, 64, 0.3, 32, [64, 32], True),
    ("U", 2, 512, 128, 64, 0.8, 32, [128, 128, 256], False),     # SA3-shaped: 132 input columns do not fit the streaming forward -> materialised rows
])
def test_fused_sa_front_end_equals_the_materialised_path(kind, b, n, c, npoint, radius, ns, mlp, expect_gather, monkeypatch):
    """the first conv2d gathers its rows from (b,n,c) features + 20 bytes per grouped row (gspn_sa_rel) instead of reading
    a (b,npoint,nsample,3+c) tensor -- or, with >= 16 feature columns, is pre-aggregated: its feature part is multiplied on the b*n points
    and the grouped rows are formed from the product (mlp.PREAGG).  Same module output and gradients as the materialised path to fp32
    rounding (the order of the first layer's additions differs), and all against the float64 composition on oracle geometry."""
    from gspn_amd import mlp as M
    from gspn_amd import pointnet_util as PU
    from gspn_amd.geometry import sa_geometry
    xyz = D.batch(kind, b, n, 6)
    pts = np.random.default_rng(12).standard_normal((b, n, c)).astype(np.float32)
    tx = dev(xyz)
    res = {}
    for mode in ("preagg", "gather", "rows"):
        monkeypatch.setattr(PU, "FUSE_SA_FRONT", mode != "rows")
        monkeypatch.setattr(M, "PREAGG", mode == "preagg")
        store = fresh_store(55)
        tp = dev(pts).requires_grad_(True)
        if mode != "rows":       # the kernels must really take this shape (or really decline it)
            geo = sa_geometry(tx, npoint, radius, ns)
            layers = PU._mlp_layers(mlp, 3 + c, 'probe', True)
            assert M.preagg_ok(layers, True, c) == (mode == "preagg" and c >= 16)
            got = PU._sa_stack_gathered(tp.detach(), geo, True, 3 + c, layers, True, 0.5, ns)
            assert (got is not None) == (expect_gather or M.preagg_ok(layers, True, c))
            store = fresh_store(55)
        new_xyz, new_points, idx = PU.pointnet_sa_module(tx, tp, npoint, radius, ns, mlp, None, False, True, 0.5, 'sa')
        g = torch.from_numpy(np.random.default_rng(5).standard_normal(tuple(new_points.shape)).astype(np.float32)).cuda()
        new_points.backward(g)
        res[mode] = (new_points.detach(), tp.grad.clone(), {k: v.grad.clone() for k, v in store.named_parameters()}, store)
    for mode in ("preagg", "gather"):
        assert rel_err(res[mode][0], res["rows"][0]) < 2e-6, mode
        assert rel_err(res[mode][1], res["rows"][1]) < 1e-5, mode
        for k in res[mode][2]:
            assert rel_err(res[mode][2][k], res["rows"][2][k]) < 1e-5, (mode, k)
    rnew, ref, ridx, leaves = ref_sa(res["preagg"][3], 'sa', xyz, pts, npoint, radius, ns, mlp, None, False, 'max', False, True, 0.5)
    assert rel_err(res["preagg"][0], ref) < 1e-5


def test_preaggregated_first_layer_random_shapes(monkeypatch):
    """seeded sweep over batch, cloud size, centres, nsample, feature width and layer widths: the pre-aggregated first layer (SA modules
    with xyz first and features first, FP modules with 0..4 gradient-free skip columns) against the path that materialises the rows"""
    from gspn_amd import mlp as M
    from gspn_amd import pointnet_util as PU
    from gspn_amd.geometry import fp_geometry, sa_geometry
    rng = np.random.default_rng(77)
    taken = 0
    for trial in range(16):
        b = int(rng.integers(1, 4))
        n = int(rng.integers(300, 3000))
        c = int(rng.choice([16, 20, 32, 48, 64, 100]))
        widths = [int(rng.choice([32, 64, 128])), int(rng.choice([16, 32, 48, 64]))] + ([int(rng.choice([32, 64]))] if trial % 3 == 0 else [])
        xyz = D.batch("UDS"[trial % 3], b, n, trial)
        pts = rng.standard_normal((b, n, c)).astype(np.float32)
        res = {}
        if trial % 2 == 0:          # SA module
            npoint = int(rng.integers(16, max(17, n // 4)))
            ns = int(rng.choice([8, 16, 32, 64]))
            radius = float(rng.choice([0.15, 0.3, 0.6]))
            for pre in (True, False):
                monkeypatch.setattr(M, "PREAGG", pre)
                store = fresh_store(trial)
                wants = trial % 4 != 2            # every other SA trial: features without a gradient (no inverse lists, no d(feat) launch to ride in)
                tp = dev(pts).requires_grad_(wants)
                _, out, _ = PU.pointnet_sa_module(dev(xyz), tp, npoint, radius, ns, widths, None, False, True, 0.5, 'sa')
                g = torch.from_numpy(np.random.default_rng(trial).standard_normal(tuple(out.shape)).astype(np.float32)).cuda()
                out.backward(g)
                res[pre] = (out.detach(), tp.grad.clone() if wants else torch.zeros(1), {k: v.grad.clone() for k, v in store.named_parameters()})
            monkeypatch.setattr(M, "PREAGG", True)
            taken += int(M.preagg_ok(PU._mlp_layers(widths, 3 + c, 'probe%d' % trial, True), True, c))
        else:                       # FP module: dense cloud = xyz, sparse = an FPS sample, 0..4 skip columns without a gradient
            m = int(rng.integers(8, max(9, n // 5)))
            c1 = int(rng.integers(0, 5))
            xyz2 = O.gather_point(xyz, O.farthest_point_sample(m, xyz))
            p1 = rng.standard_normal((b, n, c1)).astype(np.float32) if c1 else None
            p2 = rng.standard_normal((b, m, c)).astype(np.float32)
            for pre in (True, False):
                monkeypatch.setattr(PU, "FUSE_FP_FRONT", pre)
                store = fresh_store(trial)
                t2 = dev(p2).requires_grad_(True)
                out = PU.pointnet_fp_module(dev(xyz), dev(xyz2), dev(p1) if c1 else None, t2, widths, True, 0.5, 'fa')
                g = torch.from_numpy(np.random.default_rng(trial).standard_normal(tuple(out.shape)).astype(np.float32)).cuda()
                out.backward(g)
                res[pre] = (out.detach(), t2.grad.clone(), {k: v.grad.clone() for k, v in store.named_parameters()})
            taken += 1
        what = "trial %d" % trial
        assert rel_err(res[True][0], res[False][0]) < 5e-6, what
        assert rel_err(res[True][1], res[False][1]) < 2e-5, what
        for k in res[True][2]:
            assert rel_err(res[True][2][k], res[False][2][k]) < 2e-5, (what, k)
    assert taken >= 12
