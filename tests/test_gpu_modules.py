"""GPU parity of the composed SA / FP modules (utils/pointnet_util.py) against the oracle composition:
C oracle for FPS / gather / ball query / grouping / 3-NN / interpolation, fp64 restatement for the MLP."""
import numpy as np
import pytest
import torch

from oracle import mlp_ref as R
from oracle import oracle as O
from tests import data as D

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def fresh_store(seed=1234):
    from gspn_amd import tf_util
    return tf_util.set_variable_store(tf_util.VariableStore(seed=seed))


def ref_params(store, scope, names):
    ps = []
    for nm in names:
        g = lambda k: store.vars["%s/%s/%s" % (scope, nm, k)].detach().double().cpu()
        w = g("weights")
        ps.append({"w": w.view(w.shape[-2], w.shape[-1]).clone().requires_grad_(True), "b": g("biases").clone().requires_grad_(True),
                   "gamma": g("bn/gamma").clone().requires_grad_(True), "beta": g("bn/beta").clone().requires_grad_(True),
                   "moving_mean": g("bn/moving_mean"), "moving_var": g("bn/moving_variance"), "bn": True})
    return ps


@pytest.mark.parametrize("kind,b,n,c,npoint,radius,ns,mlp", [
    ("U", 2, 4096, 3, 512, 0.2, 32, [32, 32, 64]),
    ("D", 2, 2048, 64, 256, 0.4, 32, [64, 64, 128]),
    ("U", 1, 1000, 0, 100, 0.3, 16, [16, 24]),
])
def test_sa_module_matches_oracle(kind, b, n, c, npoint, radius, ns, mlp):
    from gspn_amd import tf_util
    from gspn_amd.pointnet_util import pointnet_sa_module
    store = fresh_store()
    xyz = D.batch(kind, b, n)
    rng = np.random.default_rng(7)
    pts = rng.random((b, n, c)).astype(np.float32) if c else None
    txyz = torch.from_numpy(xyz).cuda()
    tpts = torch.from_numpy(pts).cuda().requires_grad_(True) if c else None
    new_xyz, new_points, idx = pointnet_sa_module(txyz, tpts, npoint, radius, ns, mlp, None, False, True, 0.5, 'layer1')
    # snapshot initial params before they are touched (moving stats were already updated in place: rebuild from init)
    # ---- oracle composition ----
    ridx_fps = O.farthest_point_sample(npoint, xyz)
    rnew = O.gather_point(xyz, ridx_fps)
    ridx, _ = O.query_ball_point(radius, ns, xyz, rnew)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    np.testing.assert_array_equal(new_xyz.cpu().numpy(), rnew)
    gx = O.group_point(xyz, ridx) - rnew[:, :, None, :]
    rows = gx if c == 0 else np.concatenate([gx, O.group_point(pts, ridx)], -1)
    ps = ref_params(store, 'layer1', ['conv%d' % i for i in range(len(mlp))])
    for p in ps:   # undo the in-place moving-average update for the reference run
        p["moving_mean"] = torch.zeros_like(p["moving_mean"])
        p["moving_var"] = torch.ones_like(p["moving_var"])
    x64 = torch.from_numpy(rows.reshape(-1, rows.shape[-1])).double()
    pts64 = None
    if c:
        pts64 = torch.from_numpy(pts).double().requires_grad_(True)
        gidx = torch.from_numpy(ridx.astype(np.int64))
        bi = torch.arange(b)[:, None, None].expand_as(gidx)
        x64 = torch.cat([torch.from_numpy(gx).double(), pts64[bi, gidx]], -1).reshape(-1, 3 + c)
    ref, moving = R.stack(x64, ps, True, 0.5, ns)
    ref = ref.view(b, npoint, mlp[-1])
    assert rel_err(new_points, ref) < 1e-5
    g = torch.from_numpy(rng.standard_normal(ref.shape)).double()
    ref.backward(g)
    new_points.backward(g.float().cuda())
    for i, p in enumerate(ps):
        wgrad = store.vars['layer1/conv%d/weights' % i].grad
        assert rel_err(wgrad.view(p["w"].shape), p["w"].grad) < 1e-4
        assert rel_err(store.vars['layer1/conv%d/bn/gamma' % i].grad, p["gamma"].grad) < 1e-4
        assert rel_err(store.vars['layer1/conv%d/bn/moving_mean' % i], moving[i][0]) < 1e-5
    if c:
        assert rel_err(tpts.grad, pts64.grad) < 1e-4


@pytest.mark.parametrize("b,n1,n2,c1,c2,mlp", [(2, 2048, 512, 64, 128, [128, 64]), (1, 700, 100, 0, 32, [16]), (2, 512, 128, 128, 256, [])])
def test_fp_module_matches_oracle(b, n1, n2, c1, c2, mlp):
    from gspn_amd.pointnet_util import pointnet_fp_module
    store = fresh_store(99)
    xyz1 = D.batch("D", b, n1, 3)
    xyz2 = O.gather_point(xyz1, O.farthest_point_sample(n2, xyz1))
    rng = np.random.default_rng(17)
    p1 = rng.standard_normal((b, n1, c1)).astype(np.float32) if c1 else None
    p2 = rng.standard_normal((b, n2, c2)).astype(np.float32)
    t1 = torch.from_numpy(p1).cuda().requires_grad_(True) if c1 else None
    t2 = torch.from_numpy(p2).cuda().requires_grad_(True)
    out = pointnet_fp_module(torch.from_numpy(xyz1).cuda(), torch.from_numpy(xyz2).cuda(), t1, t2, mlp, True, 0.5, 'fa')
    rd, ri = O.three_nn(xyz1, xyz2)
    w64 = R.fp_weights(torch.from_numpy(rd).double())
    p2r = torch.from_numpy(p2).double().requires_grad_(True)
    gi = torch.from_numpy(ri.astype(np.int64))
    bi = torch.arange(b)[:, None, None].expand_as(gi)
    interp = (p2r[bi, gi] * w64[..., None]).sum(2)
    p1r = torch.from_numpy(p1).double().requires_grad_(True) if c1 else None
    cat = torch.cat([interp, p1r], 2) if c1 else interp
    if mlp:
        ps = ref_params(store, 'fa', ['conv_%d' % i for i in range(len(mlp))])
        for p in ps:
            p["moving_mean"] = torch.zeros_like(p["moving_mean"])
            p["moving_var"] = torch.ones_like(p["moving_var"])
        ref, _ = R.stack(cat.reshape(b * n1, -1), ps, True, 0.5, None)
        ref = ref.view(b, n1, mlp[-1])
    else:
        ref = cat
    assert rel_err(out, ref) < 1e-5
    g = torch.from_numpy(rng.standard_normal(ref.shape)).double()
    ref.backward(g)
    out.backward(g.float().cuda())
    assert rel_err(t2.grad, p2r.grad) < 1e-4
    if c1:
        assert rel_err(t1.grad, p1r.grad) < 1e-4


@pytest.mark.parametrize("b,n1,n2,c1,c2,mlp", [(2, 4096, 512, 3, 64, [64, 64, 64]), (1, 1500, 200, 0, 32, [32, 16]), (2, 900, 128, 4, 128, [64, 32])])
def test_fp_module_preaggregated_first_layer(b, n1, n2, c1, c2, mlp, monkeypatch):
    """FP module whose skip link carries no gradient and has <= 4 columns (the last FP level: raw colours): the first layer's
    interpolated part is multiplied on the n2 sparse points (mlp.PREAGG).  Against the float64 composition and against the path that
    writes the concatenated matrix (outputs, d(points2), every parameter gradient)."""
    from gspn_amd import pointnet_util as PU
    xyz1 = D.batch("D", b, n1, 3)
    xyz2 = O.gather_point(xyz1, O.farthest_point_sample(n2, xyz1))
    rng = np.random.default_rng(23)
    p1 = rng.standard_normal((b, n1, c1)).astype(np.float32) if c1 else None
    p2 = rng.standard_normal((b, n2, c2)).astype(np.float32)
    g = rng.standard_normal((b, n1, mlp[-1])).astype(np.float32)
    res = {}
    for pre in (True, False):
        monkeypatch.setattr(PU, "FUSE_FP_FRONT", pre)
        store = fresh_store(99)
        t1 = torch.from_numpy(p1).cuda() if c1 else None
        t2 = torch.from_numpy(p2).cuda().requires_grad_(True)
        if pre:
            calls = []
            real = PU._fp_stack_preagg
            monkeypatch.setattr(PU, "_fp_stack_preagg", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        out = PU.pointnet_fp_module(torch.from_numpy(xyz1).cuda(), torch.from_numpy(xyz2).cuda(), t1, t2, mlp, True, 0.5, 'fa')
        if pre:
            assert calls, "the pre-aggregated path did not take this shape"
            monkeypatch.setattr(PU, "_fp_stack_preagg", real)
        out.backward(torch.from_numpy(g).cuda())
        res[pre] = (out.detach(), t2.grad.clone(), {k: v.grad.clone() for k, v in store.named_parameters()}, store)
    assert rel_err(res[True][0], res[False][0]) < 2e-6
    assert rel_err(res[True][1], res[False][1]) < 1e-5
    for k in res[True][2]:
        assert rel_err(res[True][2][k], res[False][2][k]) < 1e-5, k
    rd, ri = O.three_nn(xyz1, xyz2)
    w64 = R.fp_weights(torch.from_numpy(rd).double())
    p2r = torch.from_numpy(p2).double().requires_grad_(True)
    gi = torch.from_numpy(ri.astype(np.int64))
    bi = torch.arange(b)[:, None, None].expand_as(gi)
    interp = (p2r[bi, gi] * w64[..., None]).sum(2)
    cat = torch.cat([interp, torch.from_numpy(p1).double()], 2) if c1 else interp
    ps = ref_params(res[True][3], 'fa', ['conv_%d' % i for i in range(len(mlp))])
    for p in ps:
        p["moving_mean"] = torch.zeros_like(p["moving_mean"])
        p["moving_var"] = torch.ones_like(p["moving_var"])
    ref, _ = R.stack(cat.reshape(b * n1, -1), ps, True, 0.5, None)
    assert rel_err(res[True][0], ref.view(b, n1, mlp[-1])) < 1e-5
    ref.view(b, n1, mlp[-1]).backward(torch.from_numpy(g).double())
    assert rel_err(res[True][1], p2r.grad) < 1e-4


def test_fea_extractor_runs_and_backprops():
    """BASELINE config 3 graph at a reduced cloud size: shapes, finiteness, every parameter gets a gradient"""
    from gspn_amd import tf_util
    from gspn_amd.fea_extractor import pn2_fea_extractor
    store = fresh_store(5)
    xyz = torch.from_numpy(D.batch("U", 2, 8192)).cuda()
    col = torch.rand(2, 8192, 3, device="cuda")
    out = pn2_fea_extractor(xyz, col, 'fea', True, 0.5)
    assert out.shape == (2, 8192, 64) and torch.isfinite(out).all()
    out.square().mean().backward()
    for name, p in store.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name


def test_prefetched_geometry_is_identical_to_inline():
    """geometry.py: the coordinate-only half computed ahead on a side stream gives bit-identical outputs and gradients"""
    from gspn_amd.fea_extractor import pn2_fea_extractor, pn2_geometry
    from gspn_amd.geometry import GeometryStream
    xyz = torch.from_numpy(D.batch("U", 2, 8192)).cuda()
    col = torch.rand(2, 8192, 3, device="cuda")
    outs, grads = [], []
    for prefetch in (False, True):
        store = fresh_store(7)
        g = None
        if prefetch:
            gs = GeometryStream(xyz.device)
            pend = gs.submit(pn2_geometry, xyz)
            g = pend.get()
            inline = pn2_geometry(xyz)
            for a, b_ in zip(g["sa"], inline["sa"]):
                assert torch.equal(a.idx, b_.idx) and torch.equal(a.new_xyz, b_.new_xyz) and torch.equal(a.pts_cnt, b_.pts_cnt)
            for a, b_ in zip(g["fp"], inline["fp"]):
                assert torch.equal(a.idx, b_.idx) and torch.equal(a.weight, b_.weight)
        out = pn2_fea_extractor(xyz, col, 'fea', True, 0.5, geometry=g)
        out.square().mean().backward()
        outs.append(out.detach().clone())
        grads.append({n: p.grad.detach().clone() for n, p in store.named_parameters()})
    assert torch.equal(outs[0], outs[1])
    # (weight gradients are summed with fp32 atomics only in the shared-slot fallback; these shapes use owned slots -> deterministic)
    for n in grads[0]:
        assert torch.allclose(grads[0][n], grads[1][n], rtol=1e-5, atol=1e-7), n


def test_sa_module_rejects_mismatched_geometry():
    from gspn_amd.geometry import sa_geometry
    from gspn_amd.pointnet_util import pointnet_sa_module
    fresh_store(3)
    xyz = torch.from_numpy(D.batch("U", 1, 1024)).cuda()
    g = sa_geometry(xyz, 128, 0.3, 16)
    with pytest.raises(ValueError):
        pointnet_sa_module(xyz, None, 64, 0.3, 16, [16], None, False, True, None, 'l', geometry=g)


def test_captured_step_replays_the_eager_step():
    """graph.py: a hipGraph replay of fwd+bwd+flatten leaves the same loss and flat gradient as the eager step"""
    from gspn_amd import parallel
    from gspn_amd.fea_extractor import pn2_fea_extractor, pn2_geometry
    from gspn_amd.graph import CapturedStep
    xyz = torch.from_numpy(D.batch("U", 2, 8192)).cuda()
    col = torch.rand(2, 8192, 3, device="cuda")
    store = fresh_store(11)
    geo = pn2_geometry(xyz)
    st = {}

    def fwd_bwd():
        for p in store.parameters():
            p.grad = None
        out = pn2_fea_extractor(xyz, col, 'fea', True, 0.5, geometry=geo)
        loss = out.square().mean()
        loss.backward()
        if "bucket" not in st:
            st["bucket"] = parallel.FlatGradBucket(store.parameters())
        st["bucket"].flatten()
        return loss

    loss0 = fwd_bwd().detach().clone()
    flat0 = st["bucket"].flat.clone()
    cap = CapturedStep(fwd_bwd)
    st["bucket"].flat.zero_()
    loss1 = cap.replay()
    torch.cuda.synchronize()
    assert torch.allclose(loss1, loss0, rtol=1e-6)
    assert torch.allclose(st["bucket"].flat, flat0, rtol=1e-4, atol=1e-7)
    # parameters' .grad are views of the bucket: an optimiser sees the replayed gradients
    p0 = store.parameters()[0]
    assert p0.grad.data_ptr() == st["bucket"].flat.data_ptr()


@pytest.mark.parametrize("b,n1,n2,c1,c2", [(2, 1500, 300, 3, 64), (1, 700, 90, 0, 33), (2, 512, 128, 64, 128)])
def test_fp_concat_matches_interpolate_plus_concat(b, n1, n2, c1, c2):
    """the fused FP input matrix == three_interpolate + concat (+ zero padded pitch), forward bit-exact, gradients to rounding"""
    from gspn_amd.pointnet_util import fp_concat
    from gspn_amd.tf_interpolate import three_interpolate, three_nn
    g = torch.Generator(device="cpu").manual_seed(n1)
    xyz1 = torch.rand(b, n1, 3, generator=g).cuda()
    xyz2 = torch.rand(b, n2, 3, generator=g).cuda()
    dist, idx = three_nn(xyz1, xyz2)
    w = 1.0 / torch.clamp(dist, min=1e-10)
    w = w / w.sum(dim=2, keepdim=True)
    p2 = torch.randn(b, n2, c2, generator=g).cuda().requires_grad_(True)
    p1 = torch.randn(b, n1, c1, generator=g).cuda().requires_grad_(True) if c1 else None
    out = fp_concat(p2, idx, w, p1)
    ld = (c1 + c2 + 3) // 4 * 4
    assert out.shape == (b * n1, ld)
    p2r = p2.detach().clone().requires_grad_(True)
    p1r = p1.detach().clone().requires_grad_(True) if c1 else None
    ref = three_interpolate(p2r, idx, w)
    if c1:
        ref = torch.cat([ref, p1r], dim=2)
    ref = ref.reshape(b * n1, c1 + c2)
    assert torch.equal(out[:, :c1 + c2], ref)
    assert (out[:, c1 + c2:] == 0).all()
    go = torch.randn(out.shape, generator=g).cuda()
    out.backward(go)
    ref.backward(go[:, :c1 + c2].contiguous())
    assert rel_err(p2.grad, p2r.grad) < 1e-5
    if c1:
        assert torch.equal(p1.grad, p1r.grad)


def test_dw_reduction_placements_agree():
    """the dW reduction as its own kernel, on a side stream (mlp.DEFER_DW), or inside pass B's launch (mlp.FUSE_DW, the default):
    identical dX and dW for the two placements that share an algorithm, the same sum with fewer slot slices for the fused one; the
    side-stream placement runs the two-product pass A (no early coefficients) and agrees to fp32 rounding"""
    from gspn_amd import mlp as M
    from tests.test_gpu_mlp import make_params, to_layers
    g = torch.Generator().manual_seed(5)
    x64 = torch.randn(4096, 32, generator=g, dtype=torch.float64)
    res = []
    for fuse, defer in ((False, False), (False, True), (True, False)):
        layers = to_layers(make_params([32, 64, 48], 32, seed=9))
        x = x64.float().cuda().requires_grad_(True)
        old = (M.FUSE_DW, M.DEFER_DW)
        M.FUSE_DW, M.DEFER_DW = fuse, defer
        try:
            out = M.mlp_stack(x, 32, layers, True, 0.7, pool_ns=32)
            out.square().sum().backward()
            torch.cuda.synchronize()
        finally:
            M.FUSE_DW, M.DEFER_DW = old
        res.append([lp.weights.grad.clone() for lp in layers] + [x.grad.clone()])
    for a_, b_ in zip(res[0], res[1]):
        assert rel_err(b_, a_) < 1e-5
    if M.POOLTOP_STREAM:        # the fused placement then also takes the pooled top layer's dX from the layer's input: another order of additions
        assert rel_err(res[2][-1], res[0][-1]) < 1e-5
    else:
        assert torch.equal(res[0][-1], res[2][-1])
    for a_, b_ in zip(res[0][:-1], res[2][:-1]):
        assert rel_err(b_, a_) < (1e-5 if M.POOLTOP_STREAM else 1e-6)


def test_fp_module_grad_cols_shortcut_is_exact():
    """when points1 needs no gradient the first layer's dX is computed for the interpolated columns only: same gradient to points2"""
    from gspn_amd.pointnet_util import pointnet_fp_module
    g = torch.Generator().manual_seed(4)
    xyz1 = torch.rand(2, 900, 3, generator=g).cuda()
    xyz2 = torch.rand(2, 200, 3, generator=g).cuda()
    p1 = torch.randn(2, 900, 3, generator=g).cuda()
    p2v = torch.randn(2, 200, 32, generator=g).cuda()
    grads = []
    for rg in (True, False):
        fresh_store(21)
        p2 = p2v.clone().requires_grad_(True)
        out = pointnet_fp_module(xyz1, xyz2, p1.clone().requires_grad_(rg), p2, [32, 16], True, 0.5, 'fa')
        out.square().mean().backward()
        grads.append(p2.grad.clone())
    assert rel_err(grads[1], grads[0]) < 1e-6


@pytest.mark.parametrize("b,n1,n2,c1,c2", [(2, 1500, 300, 3, 64), (1, 700, 90, 0, 33), (2, 512, 128, 64, 130)])
def test_fp_concat_gather_gradient_is_bit_exact_vs_reference_order(b, n1, n2, c1, c2):
    """with the inverse lists of fp_geometry the interpolation gradient is a gather in the reference's own summation order
    (tf_interpolate.cpp:131-153): identical bits to the C oracle, no atomics"""
    from gspn_amd.geometry import fp_geometry
    from gspn_amd.pointnet_util import fp_concat
    g = torch.Generator(device="cpu").manual_seed(n1 + c2)
    xyz1 = torch.rand(b, n1, 3, generator=g).cuda()
    xyz2 = torch.rand(b, n2, 3, generator=g).cuda()
    geo = fp_geometry(xyz1, xyz2)
    assert geo.order.shape == (b, 3 * n1) and geo.offsets.shape == (b, n2 + 1)
    assert (geo.offsets[:, 0] == 0).all() and (geo.offsets[:, -1] == 3 * n1).all()
    p2 = torch.randn(b, n2, c2, generator=g).cuda().requires_grad_(True)
    p1 = torch.randn(b, n1, c1, generator=g).cuda().requires_grad_(True) if c1 else None
    out = fp_concat(p2, geo.idx, geo.weight, p1, geo.order, geo.offsets)
    go = torch.randn(out.shape, generator=g).cuda()
    out.backward(go)
    gi = go[:, :c2].reshape(b, n1, c2).contiguous().cpu().numpy()
    ref = O.three_interpolate_grad(p2.detach().cpu().numpy(), geo.idx.cpu().numpy(), geo.weight.cpu().numpy(), gi)
    np.testing.assert_array_equal(p2.grad.cpu().numpy(), ref)
    if c1:
        assert torch.equal(p1.grad, go[:, c2:c2 + c1].reshape(b, n1, c1))


def test_training_step_is_bit_reproducible():
    """no atomics on the training path (static row split, gather gradients through inverse lists): two identical steps give identical bits"""
    from gspn_amd.fea_extractor import pn2_fea_extractor
    xyz = torch.from_numpy(D.batch("D", 2, 8192)).cuda()
    col = torch.rand(2, 8192, 3, device="cuda")
    res = []
    for _ in range(2):
        store = fresh_store(13)
        out = pn2_fea_extractor(xyz, col, 'fea', True, 0.5)
        out.square().mean().backward()
        res.append((out.detach().clone(), {n: p.grad.detach().clone() for n, p in store.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0])
    for n in res[0][1]:
        assert torch.equal(res[0][1][n], res[1][1][n]), n


@pytest.mark.parametrize("b,ln,n", [(8, 98304, 2048), (8, 16384, 2048), (2, 6144, 512), (3, 1000, 128), (1, 5, 3), (2, 4096, 5000), (1, 70000, 1), (2, 30000, 9000)])
def test_inverse_lists_match_stable_sort(b, ln, n):
    """gspn_inverse_lists == stable sort of the indices + searchsorted (the order the gather-form gradients sum in)"""
    from gspn_amd.geometry import inverse_lists
    g = torch.Generator().manual_seed(ln + n)
    idx = torch.randint(0, n, (b, ln), generator=g, dtype=torch.int32)
    if n > 4:
        idx[:, : ln // 3] = idx[:, : ln // 3] % 3             # a few crowded values: long groups
    order, offsets = inverse_lists(idx.cuda(), n)
    keys, ref_order = torch.sort(idx.long(), dim=1, stable=True)
    bounds = torch.arange(n + 1).unsqueeze(0).expand(b, -1).contiguous()
    ref_off = torch.searchsorted(keys.contiguous(), bounds)
    assert torch.equal(offsets.cpu().long(), ref_off)
    assert torch.equal(order.cpu().long(), ref_order)


@pytest.mark.parametrize("b,ln,n", [(3, 33001, 100), (2, 70001, 3000), (1, 131071, 777), (5, 40000, 1), (2, 98304, 2048)])
def test_inverse_lists_spread_over_position_slices_with_dropped_positions(b, ln, n):
    """r06 (csr_hist / csr_slice_scan / csr_slice_fill: several workgroups per scene for long index tensors): ragged slice boundaries (L not a multiple of the
    slice count), key counts that are not multiples of the scan's 1024, one key for everything, and positions whose value lies OUTSIDE [0, n) -- dropped, as the
    header promises (offsets[n] < L then) -- against a stable sort of the valid positions"""
    from gspn_amd.geometry import inverse_lists
    g = torch.Generator().manual_seed(ln * 7 + n)
    idx = torch.randint(0, n, (b, ln), generator=g, dtype=torch.int32)
    bad = torch.rand(b, ln, generator=g) < 0.07
    idx[bad] = torch.where(torch.rand(int(bad.sum()), generator=g) < 0.5, torch.tensor(-1, dtype=torch.int32), torch.tensor(n + 5, dtype=torch.int32))
    order, offsets = inverse_lists(idx.cuda(), n)
    order, offsets = order.cpu().long(), offsets.cpu().long()
    for s_ in range(b):
        valid = (idx[s_] >= 0) & (idx[s_] < n)
        pos = torch.nonzero(valid).squeeze(1)
        keys, perm = torch.sort(idx[s_][valid].long(), stable=True)
        nv = int(valid.sum())
        assert int(offsets[s_, n]) == nv
        assert torch.equal(order[s_, :nv], pos[perm])
        assert torch.equal(offsets[s_], torch.searchsorted(keys.contiguous(), torch.arange(n + 1)))


def test_copy_into_multi_copy():
    """graph.copy_into refills a nested structure of persistent buffers with one gspn_multi_copy launch: odd sizes, unaligned views, > 40 tensors"""
    from gspn_amd.graph import copy_into
    g = torch.Generator().manual_seed(0)
    sizes = [1, 3, 17, 4096, 65536 // 4 + 5, 300001] + [7 + i for i in range(45)]
    src = {"a": [torch.randn(n, generator=g).cuda() for n in sizes], "b": (torch.randint(0, 1000, (8, 33, 5), generator=g, dtype=torch.int32).cuda(),)}
    base = torch.zeros(1000, device="cuda")
    src["a"].append(base[1:998])                               # 4-byte aligned only
    dst = {"a": [torch.zeros_like(t) for t in src["a"]], "b": (torch.zeros_like(src["b"][0]),)}
    dst["a"][-1] = torch.zeros(999, device="cuda")[2:999]
    src["a"][-1].copy_(torch.randn(997, generator=g))
    copy_into(dst, src)
    torch.cuda.synchronize()
    for d, s_ in zip(dst["a"], src["a"]):
        assert torch.equal(d, s_)
    assert torch.equal(dst["b"][0], src["b"][0])


def test_three_nn_weights_formula():
    from gspn_amd import _lib as L
    g = torch.Generator().manual_seed(1)
    dist = torch.rand(5000, 3, generator=g).cuda() ** 4
    dist[::7, 0] = 0.0                                         # coincident points: clamped at 1e-10
    w = torch.empty_like(dist)
    L.check(L.lib().gspn_three_nn_weights(dist.shape[0], L.ptr(dist), L.ptr(w), L.stream()), "w")
    d = torch.clamp(dist, min=1e-10)
    ref = (1.0 / d) / (1.0 / d).sum(dim=1, keepdim=True)
    assert rel_err(w, ref) < 1e-6
    assert torch.allclose(w.sum(dim=1), torch.ones(5000, device="cuda"), atol=1e-6)


def test_flat_adam_matches_torch_adam():
    """parallel.FlatAdam (one gspn_adam_flat launch over the flat parameter / gradient buffers) follows torch.optim.Adam"""
    from gspn_amd.parallel import FlatAdam, FlatGradBucket
    g = torch.Generator().manual_seed(3)
    shapes = [(6, 32), (32,), (32, 64), (64,), (1, 1, 67, 64), (5,)]
    init = [torch.randn(*s_, generator=g) for s_ in shapes]
    pa = [torch.nn.Parameter(t.clone().cuda()) for t in init]
    pb = [torch.nn.Parameter(t.clone().cuda()) for t in init]
    ref = torch.optim.Adam(pa, lr=1e-2)
    bucket = FlatGradBucket(pb)
    opt = FlatAdam(bucket, lr=1e-2)
    for step in range(6):
        grads = [torch.randn(*s_, generator=g).cuda() * (10.0 ** (step - 3)) for s_ in shapes]
        for p_, q_, gr in zip(pa, pb, grads):
            p_.grad = gr.clone()
            q_.grad = gr.clone()
        bucket.flatten()
        ref.step()
        bucket.flat.mul_(4.0)                                 # as if summed over 4 ranks: step() scales it back
        opt.step(grad_scale=0.25)
    torch.cuda.synchronize()
    for p_, q_ in zip(pa, pb):
        assert q_.data_ptr() >= opt.flat.data_ptr() and q_.data_ptr() < opt.flat.data_ptr() + opt.flat.numel() * 4
        assert rel_err(q_.detach(), p_.detach()) < 2e-6


def test_flat_adam_with_the_step_counter_on_the_device_equals_the_host_counted_one_also_from_a_graph():
    """FlatAdam(device_step=True) (gspn_adam_flat_dev): bit-identical parameters to the host-counted kernel over 7 updates -- eager, and replayed from
    a hipGraph whose arguments never change (the last-ticket workgroup publishes step + 1) -- on a buffer of several hundred workgroups"""
    from gspn_amd.parallel import FlatAdam, FlatGradBucket
    g = torch.Generator().manual_seed(8)
    shapes = [(300, 500), (64,), (1, 1, 131, 128)]
    init = [torch.randn(*s_, generator=g) for s_ in shapes]
    grads = [[torch.randn(*s_, generator=g).cuda() for s_ in shapes] for _ in range(7)]
    outs = []
    for mode in ("host", "device", "graph"):
        ps = [torch.nn.Parameter(t.clone().cuda()) for t in init]
        bucket = FlatGradBucket(ps)
        opt = FlatAdam(bucket, lr=1e-2, weight_decay=1e-3, device_step=mode != "host")
        graph = None
        for step in range(7):
            torch.cat([gr.reshape(-1) for gr in grads[step]], out=bucket.flat)
            if mode == "graph" and step >= 2:
                if graph is None:
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        opt.step(grad_scale=0.5)                  # (captured, not executed)
                graph.replay()
            else:
                opt.step(grad_scale=0.5)
        torch.cuda.synchronize()
        assert opt.t == 7
        outs.append(opt.flat.clone())
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("n", [1, 3, 4096, 1_000_003, 16_777_216])
def test_dot_kernel_is_deterministic_and_close_to_float64(n):
    """gspn_dot (r04: the bench's loss <out, g> on the library's own kernel instead of a library reduction): two launches, 1024 partials added
    in index order in double -- the same bits on every call, 1e-6 of the float64 product sum; unaligned operands take the scalar tail"""
    from gspn_amd import _lib as L
    g = torch.Generator(device="cuda").manual_seed(n)
    a = torch.randn(n + 1, device="cuda", generator=g)[1:]          # 4-byte aligned only
    b = torch.randn(n, device="cuda", generator=g)
    w = torch.empty(int(L.lib().gspn_dot_work_floats()), device="cuda")
    outs = []
    for x in (a, a.clone()):                                         # unaligned / aligned copies of the same data
        for _ in range(2):
            o = torch.empty((), device="cuda")
            L.check(L.lib().gspn_dot(n, L.ptr(x), L.ptr(b), L.ptr(w), L.ptr(o), L.stream()), "dot")
            outs.append(o.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[2], outs[3])
    ref = (a.double() * b.double()).sum()
    scale = (a.double() * b.double()).abs().sum() + 1e-30
    for o in outs:
        assert float((o.double() - ref).abs() / scale) < 1e-6


def test_gradient_sinks_leave_the_bucket_bit_identical_and_capture_without_a_cat():
    """r06 (parallel.FlatGradBucket.attach_sinks / mlp.GRAD_SINKS): the shared-MLP backward writes every parameter gradient straight into its slice of the
    flat bucket -- same bits as gathering them with `cat`, p.grad re-pointed at the slices, also after FlatAdam has moved the parameters, also when the
    step is replayed from a hipGraph; geometry carrying the pre-padded colours (sa_geometry(points=...)) gives the same result as padding inline."""
    from gspn_amd import mlp, parallel
    from gspn_amd.fea_extractor import pn2_fea_extractor, pn2_geometry
    from gspn_amd.graph import CapturedStep
    xyz = torch.from_numpy(D.batch("U", 2, 8192)).cuda()
    col = torch.rand(2, 8192, 3, device="cuda")
    go = torch.randn(2, 8192, 64, device="cuda")
    geo_plain, geo_pad = pn2_geometry(xyz), pn2_geometry(xyz, points=col)
    assert geo_plain["sa"][0].feat4 is None and tuple(geo_pad["sa"][0].feat4.shape) == (2 * 8192, 4)
    flats = []
    for sinks in (False, True):
        store = fresh_store(11)
        st = {}

        def fwd_bwd():
            for p in store.parameters():
                p.grad = None
            out = pn2_fea_extractor(xyz, col, 'fea', True, 0.5, geometry=geo_pad if sinks else geo_plain)
            out.backward(go)
            if "bucket" not in st:
                st["bucket"] = parallel.FlatGradBucket(store.parameters())
                st["opt"] = parallel.FlatAdam(st["bucket"], lr=0.0)          # moves the parameters into one flat buffer (lr 0: they keep their values)
                if sinks:
                    st["bucket"].attach_sinks()
            elif sinks:
                assert all(p.grad is None for p in store.parameters())      # nothing went through autograd: every gradient was written in place
                assert len(st["bucket"]._written) == len(store.parameters())
            st["bucket"].flatten()
            return out.detach()

        fwd_bwd()                                   # creates the variables, the bucket, the optimiser
        st["bucket"].flat.fill_(float("nan"))
        fwd_bwd()
        torch.cuda.synchronize()
        eager = st["bucket"].flat.clone()
        assert torch.isfinite(eager).all()
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(store.parameters(), st["bucket"]._views))
        cap = CapturedStep(fwd_bwd)
        st["bucket"].flat.fill_(float("nan"))
        cap.replay()
        torch.cuda.synchronize()
        assert torch.equal(st["bucket"].flat, eager)
        flats.append(eager)
        for k in [k for k, e in mlp.GRAD_SINKS.items() if e.bucket is st["bucket"]]:
            del mlp.GRAD_SINKS[k]
    assert torch.equal(flats[0], flats[1])


def test_gradient_sinks_add_a_second_use_of_a_layer():
    """a layer applied twice in one backward pass: the first gradient is written in place, the second arrives through autograd and flatten() adds it"""
    from gspn_amd import mlp, parallel
    g = torch.Generator().manual_seed(5)
    x1 = torch.randn(512, 32, generator=g).cuda()
    x2 = torch.randn(512, 32, generator=g).cuda()
    res = []
    for sinks in (False, True):
        w = torch.nn.Parameter((torch.randn(32, 48, generator=torch.Generator().manual_seed(1)) * 0.1).cuda())
        b = torch.nn.Parameter(torch.zeros(48).cuda())
        beta = torch.nn.Parameter(torch.zeros(48).cuda())
        gamma = torch.nn.Parameter(torch.ones(48).cuda())
        lp = mlp.LayerParams(w, b, True, beta, gamma, torch.zeros(48).cuda(), torch.ones(48).cuda())
        bucket = parallel.FlatGradBucket([w, b, beta, gamma])
        if sinks:
            bucket.attach_sinks()
        out = mlp.mlp_stack(x1, 32, [lp], True, 0.5).sum() + 2.0 * mlp.mlp_stack(x2, 32, [lp], True, 0.5).square().sum()
        out.backward()
        bucket.flatten()
        torch.cuda.synchronize()
        res.append(bucket.flat.clone())
        for k in [k for k, e in mlp.GRAD_SINKS.items() if e.bucket is bucket]:
            del mlp.GRAD_SINKS[k]
    assert torch.allclose(res[0], res[1], rtol=1e-6, atol=1e-6) and float(res[0].abs().max()) > 0


@pytest.mark.parametrize("c", [64, 128])
def test_fp_concat_grad_csr_split_shares_long_lists_out_and_stays_reproducible(c):
    """r06 (csr_gather.h SPLIT, gspn_fp_concat_grad_csr_split): lists longer than split_t walked by all sixteen rows of a workgroup -- same sums as the sequential walk
    to float32 rounding (a different, FIXED order: bit-identical from run to run), identical BITS where no list is longer than split_t, on index tensors with one list of
    5000 entries, a few of several hundred, many short ones, empty ones, and a target count that leaves idle rows in the last workgroup"""
    import ctypes
    from gspn_amd import _lib as L
    from gspn_amd.geometry import inverse_lists
    lib = L.lib()
    b, n, m = 3, 9000, 1003
    g = torch.Generator().manual_seed(c)
    idx = torch.randint(7, m, (b, n, 3), generator=g, dtype=torch.int32)
    idx[:, :1700, :] = 0                                  # one sparse point is the neighbour of 1700 dense points (5100 entries)
    idx[:, 1700:1900, 0] = 1
    idx[:, 1900:2300, 1] = 2
    idx[1, 2300:2400, :] = 3
    w = torch.rand(b, n, 3, generator=g)
    go = torch.randn(b * n, c, generator=g)
    idx, w, go = idx.cuda(), w.cuda(), go.cuda()
    order, offsets = inverse_lists(idx.reshape(b, 3 * n), m)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(split_t):
        out = torch.full((b, m, c), float("nan"), device="cuda")
        assert lib.gspn_fp_concat_grad_csr_split(b, n, m, c, 0, c, P(go), P(order), P(offsets), P(w), P(out), None, split_t, st) == 0
        torch.cuda.synchronize()
        return out

    seq = run(0)
    ref = torch.zeros(b, m, c, dtype=torch.float64, device="cuda")
    ref.view(b * m, c).index_add_(0, (idx.long() + torch.arange(b, device="cuda")[:, None, None] * m).reshape(-1),
                                  (go.double().view(b, n, 1, c) * w.double().unsqueeze(-1)).reshape(-1, c))
    scale = float(ref.abs().max())
    assert float((seq.double() - ref).abs().max()) / scale < 1e-5
    for t in (128, 64, 17):
        a, a2 = run(t), run(t)
        assert torch.equal(a, a2)                                               # a fixed order
        assert float((a.double() - ref).abs().max()) / scale < 1e-5
        lens = (offsets[:, 1:] - offsets[:, :-1])
        short = (lens <= t)
        assert torch.equal(a[short], seq[short])                                # lists inside split_t: the sequential walk, bit for bit
    assert torch.equal(run(10 ** 6), seq)                                       # nothing is longer: identical everywhere
