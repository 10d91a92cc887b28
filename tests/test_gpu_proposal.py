"""GPU parity of the proposal-head glue (gspn_amd/proposal_head.py: models/model_rpointnet.py:28-77, 257-267, 1346-1355) against
the oracle composition: C oracle for FPS / gather / ball query / grouping / nn_distance, fp64 restatement for the layers."""
import numpy as np
import pytest
import torch

from oracle import mlp_ref as R
from oracle import oracle as O
from tests import data as D
from tests.test_gpu_modules import fresh_store, ref_params, rel_err

pytestmark = pytest.mark.gpu


def _lin_params(store, name):
    w = store.vars[name + "/weights"].detach().double().cpu()
    return {"w": w.view(w.shape[-2], w.shape[-1]).clone().requires_grad_(True), "b": store.vars[name + "/biases"].detach().double().cpu().clone().requires_grad_(True)}


@pytest.mark.parametrize("use_shift", [False, True])
def test_multi_encoding_net_matches_oracle(use_shift):
    from gspn_amd.proposal_head import multi_encoding_net
    b, n, c, npoint = 2, 2048, 3, 64
    radius_list, nsample_list = [0.3, 0.5], [32, 64]
    mlp_list, mlp_list2 = [[16, 32], [16, 48]], [40]
    store = fresh_store(31)
    xyz = D.batch("U", b, n)
    rng = np.random.default_rng(3)
    pts = rng.random((b, n, c)).astype(np.float32)
    shift = (rng.standard_normal((b, npoint, 3)) * 0.05).astype(np.float32) if use_shift else None
    txyz = torch.from_numpy(xyz).cuda()
    tpts = torch.from_numpy(pts).cuda().requires_grad_(True)
    tshift = torch.from_numpy(shift).cuda().requires_grad_(True) if use_shift else None
    new_xyz, new_points, shift_out, fps_idx = multi_encoding_net(txyz, tpts, npoint, radius_list, nsample_list, mlp_list, mlp_list2, True, 0.5, 'ctx',
                                                                 use_xyz=True, output_shift=True, shift_pred=tshift)
    assert new_points.shape == (b, npoint, 40) and shift_out.shape == (b, npoint, 4)
    # ---- oracle composition ----
    ridx = O.farthest_point_sample(npoint, xyz)
    np.testing.assert_array_equal(fps_idx.cpu().numpy(), ridx)
    rnew = O.gather_point(xyz, ridx)
    np.testing.assert_array_equal(new_xyz.cpu().numpy(), rnew)
    pts64 = torch.from_numpy(pts).double().requires_grad_(True)
    sh64 = torch.from_numpy(shift).double().requires_grad_(True) if use_shift else None
    feats = []
    allps = []
    for i, (r, ns) in enumerate(zip(radius_list, nsample_list)):
        bidx, _ = O.query_ball_point(r, ns, xyz, rnew)
        gx = torch.from_numpy(O.group_point(xyz, bidx) - rnew[:, :, None, :]).double()     # fp32 subtraction, as the op does
        if use_shift:
            gx = torch.from_numpy((O.group_point(xyz, bidx) - rnew[:, :, None, :]).astype(np.float32)).double() - sh64[:, :, None, :]
        gidx = torch.from_numpy(bidx.astype(np.int64))
        bi = torch.arange(b)[:, None, None].expand_as(gidx)
        rows = torch.cat([pts64[bi, gidx], gx], -1).reshape(-1, c + 3)                      # FEATURES first (:61)
        ps = ref_params(store, 'ctx', ['conv_prev_%d_%d' % (i, j) for j in range(len(mlp_list[i]))])
        for p in ps:
            p["moving_mean"] = torch.zeros_like(p["moving_mean"]); p["moving_var"] = torch.ones_like(p["moving_var"])
        allps.append(ps)
        out, _ = R.stack(rows, ps, True, 0.5, ns)
        feats.append(out.view(b, npoint, mlp_list[i][-1]))
    cat = torch.cat(feats, -1).reshape(b * npoint, -1)
    ps2 = ref_params(store, 'ctx', ['conv_post_0'])
    for p in ps2:
        p["moving_mean"] = torch.zeros_like(p["moving_mean"]); p["moving_var"] = torch.ones_like(p["moving_var"])
    ref, _ = R.stack(cat, ps2, True, 0.5, None)
    lin = _lin_params(store, 'ctx/conv_shift_pred')
    ref_shift = ref @ lin["w"] + lin["b"]
    tol = 1e-5 if not use_shift else 2e-5
    assert rel_err(new_points, ref.view(b, npoint, 40)) < tol
    assert rel_err(shift_out, ref_shift.view(b, npoint, 4)) < 5e-5
    g = torch.from_numpy(rng.standard_normal((b, npoint, 40))).double()
    g4 = torch.from_numpy(rng.standard_normal((b, npoint, 4))).double()
    (ref.view(b, npoint, 40) * g).sum().add((ref_shift.view(b, npoint, 4) * g4).sum()).backward()
    (new_points * g.float().cuda()).sum().add((shift_out * g4.float().cuda()).sum()).backward()
    assert rel_err(tpts.grad, pts64.grad) < 1e-4
    if use_shift:
        assert rel_err(tshift.grad, sh64.grad) < 1e-4
    assert rel_err(store.vars['ctx/conv_shift_pred/weights'].grad.view(40, 4), lin["w"].grad) < 1e-4
    assert rel_err(store.vars['ctx/conv_shift_pred/biases'].grad, lin["b"].grad) < 1e-4
    for i, ps in enumerate(allps):
        for j, p in enumerate(ps):
            wg = store.vars['ctx/conv_prev_%d_%d/weights' % (i, j)].grad
            assert rel_err(wg.view(p["w"].shape), p["w"].grad) < 1e-4


def test_fea_trans_net_matches_fp64():
    from gspn_amd.proposal_head import fea_trans_net
    store = fresh_store(8)
    g = torch.Generator().manual_seed(2)
    x64 = torch.randn(3, 500, 24, generator=g, dtype=torch.float64)
    x = x64.float().cuda().requires_grad_(True)
    out = fea_trans_net(x, [32, 16, 5], 'ft', True, 0.9)
    ps = ref_params(store, 'ft', ['conv0', 'conv1'])
    for p in ps:
        p["moving_mean"] = torch.zeros_like(p["moving_mean"]); p["moving_var"] = torch.ones_like(p["moving_var"])
    xr = x64.clone().requires_grad_(True)
    h, _ = R.stack(xr.reshape(-1, 24), ps, True, 0.9, None)
    lin = _lin_params(store, 'ft/conv2')
    ref = (h @ lin["w"] + lin["b"]).view(3, 500, 5)
    assert rel_err(out, ref) < 1e-5
    go = torch.randn(3, 500, 5, generator=g, dtype=torch.float64)
    ref.backward(go)
    out.backward(go.float().cuda())
    assert rel_err(x.grad, xr.grad) < 1e-4
    assert rel_err(store.vars['ft/conv2/weights'].grad.view(16, 5), lin["w"].grad) < 1e-4


def test_chamfer_recons_loss_matches_oracle():
    from gspn_amd.proposal_head import chamfer_recons_loss
    rng = np.random.default_rng(5)
    B, n = 12, 512
    a = rng.standard_normal((B, n, 3)).astype(np.float32)
    c = rng.standard_normal((B, n, 3)).astype(np.float32)
    mask = (rng.random(B) > 0.4).astype(np.float32)
    ta = torch.from_numpy(a).cuda().requires_grad_(True)
    loss = chamfer_recons_loss(ta, torch.from_numpy(c).cuda(), torch.from_numpy(mask).cuda())
    d1, i1, d2, i2 = O.nn_distance(a, c)
    per = (d1.astype(np.float64) + d2.astype(np.float64)).mean(-1)
    ref = (per * mask).sum() / (mask.sum() + 1e-8)
    assert abs(float(loss) - ref) < 1e-5 * abs(ref)
    loss.backward()
    # gradient of the masked mean through both NN terms (tf_nndistance_g.cu:132-151)
    gd = (mask / (mask.sum() + 1e-8))[:, None] / n
    g1, g2 = O.nn_distance_grad(a, c, np.broadcast_to(gd, (B, n)).astype(np.float32).copy(), i1, np.broadcast_to(gd, (B, n)).astype(np.float32).copy(), i2)
    assert rel_err(ta.grad, torch.from_numpy(g1)) < 1e-4
