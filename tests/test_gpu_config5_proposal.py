"""BASELINE configs[4]'s proposal half at its per-GPU shard: the context encoder (models/model_rpointnet.py:28-77, called as at :377 --
256 seeds, radii 0.5/1.0/1.5 m, nsample 256/256/512, mlp [64,128,256] x 3, use_xyz, a stop-gradient shift, given seed indices) and the
Chamfer reconstruction loss (:1346-1355) on 8 scenes x 65536 points (metre-scale rooms, tests/data.py "S").

  (a) FPS (multi-CU kernel: one scene does not fit a CU) and all three ball queries index-exact against the C oracle;
  (b) one training step (encoder + Chamfer, forward + backward) is finite and bit-reproducible in every parameter gradient;
  (c) TRAINING mode, the widest branch (r = 1.5, nsample = 512: 1 048 576 grouped rows through the gathered first layer, the 128-wide
      forward / pass-B instances, the pooled wgrad kernel and the nsample-512 max-pool): outputs within 1e-5 and weight / gamma / beta
      gradients within 1e-4 of the float64 composition of oracle/mlp_ref.py over the same rows (fragile entries silenced as in
      tests/test_gpu_mlp.py::check_stack);
  (d) INFERENCE mode (moving statistics: scenes independent): one scene's full 768-channel output within 1e-5 of the float64
      composition with the parameters and moving statistics the training step left;
  (e) the Chamfer value and its gradient against the oracle's nn_distance(+grad) on a slab of clouds.
"""
import numpy as np
import pytest
import torch

from oracle import mlp_ref as R
from oracle import oracle as O
from tests import data as D
from tests.test_gpu_mlp import fragile_entries
from tests.test_gpu_modules import ref_params, rel_err

pytestmark = pytest.mark.gpu

B, N, NSEED = 8, 65536, 256
RADII, NS, MLPS = [0.5, 1.0, 1.5], [256, 256, 512], [[64, 128, 256]] * 3
NCLOUD, NINS = B * NSEED, 512


@pytest.fixture(scope="module")
def setup():
    from gspn_amd.tf_sampling import farthest_point_sample
    xyz = D.batch("S", B, N, 700)
    t = torch.from_numpy(xyz).cuda()
    gen = torch.Generator(device="cuda").manual_seed(41)
    col = torch.rand(B, N, 3, device="cuda", generator=gen)
    shift = (torch.randn(B, NSEED, 3, device="cuda", generator=gen) * 0.05).contiguous()
    fps_idx = farthest_point_sample(NSEED, t)
    return xyz, t, col, shift, fps_idx


def test_seed_and_ball_indices_match_the_oracle(setup):
    from gspn_amd import _lib as L
    from gspn_amd.tf_grouping import query_ball_point
    from gspn_amd.tf_sampling import gather_point
    xyz, t, col, shift, fps_idx = setup
    L.check_async(block=True)                          # the multi-CU launch's status word
    ref = O.farthest_point_sample(NSEED, xyz, mt=True)
    np.testing.assert_array_equal(fps_idx.cpu().numpy(), ref)
    new_xyz = gather_point(t, fps_idx)
    rnew = O.gather_point(xyz, ref)
    np.testing.assert_array_equal(new_xyz.cpu().numpy(), rnew)
    for r, ns in zip(RADII, NS):
        idx, cnt = query_ball_point(r, ns, t, new_xyz)
        ridx, rcnt = O.query_ball_point(r, ns, xyz, rnew, mt=True)
        np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
        np.testing.assert_array_equal(cnt.cpu().numpy(), rcnt)
        assert int(cnt.min()) >= 1 and int(cnt.max()) == ns          # metre-scale rooms: full and partly filled balls both occur


def _clouds(seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    pred = torch.randn(NCLOUD, NINS, 3, device="cuda", generator=gen)
    gt = (pred[:, torch.randperm(NINS, device="cuda", generator=gen)] + 0.05 * torch.randn(NCLOUD, NINS, 3, device="cuda", generator=gen)).contiguous()
    mask = (torch.rand(NCLOUD, device="cuda", generator=gen) > 0.2).float()
    return pred, gt, mask


def test_training_step_is_finite_and_bit_reproducible(setup):
    from gspn_amd import tf_util
    from gspn_amd.proposal_head import chamfer_recons_loss, multi_encoding_net
    xyz, t, col, shift, fps_idx = setup
    pred0, gt, mask = _clouds(9)
    gout = torch.randn(B, NSEED, 768, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)) / (B * NSEED * 768)
    runs = []
    for _ in range(2):
        store = tf_util.set_variable_store(tf_util.VariableStore(seed=55))
        pred = pred0.clone().requires_grad_(True)
        _, fea, _, _ = multi_encoding_net(t, col, NSEED, RADII, NS, MLPS, [], True, 0.5, 'context_encoder', use_xyz=True, output_shift=False,
                                          shift_pred=shift, fps_idx=fps_idx)
        assert fea.shape == (B, NSEED, 768)
        loss = (fea * gout).sum() + chamfer_recons_loss(pred, gt, mask)
        loss.backward()
        torch.cuda.synchronize()
        runs.append((fea.detach().clone(), float(loss), {n: p.grad.detach().clone() for n, p in store.named_parameters()}, pred.grad.detach().clone()))
    assert torch.isfinite(runs[0][0]).all() and np.isfinite(runs[0][1])
    assert torch.equal(runs[0][0], runs[1][0])
    assert len(runs[0][2]) == 3 * 3 * 4                                # 9 conv layers x (weights, biases, beta, gamma)
    for n, g in runs[0][2].items():
        assert torch.isfinite(g).all(), n
        assert torch.equal(g, runs[1][2][n]), n
    # the Chamfer gradient is a scatter-add with hardware atomics (tf_nndistance_g.cu:132-151): order-free, equal to rounding
    assert torch.isfinite(runs[0][3]).all()
    assert rel_err(runs[0][3], runs[1][3]) < 1e-6


def _rows64(t, col, shift, new_xyz, idx, scenes):
    """the encoder's grouped input rows [colour | (xyz[idx] - seed) - shift] (FEATURES first, :61) for `scenes`, the coordinate part
    formed in float32 exactly as the ops do (two float32 subtractions, :55-57), then widened"""
    gi = idx[scenes].long()                                            # (s, m, ns)
    s = len(scenes)
    bi = torch.arange(s, device=t.device)[:, None, None].expand_as(gi)
    rel = (t[scenes][bi, gi] - new_xyz[scenes][:, :, None, :]) - shift[scenes][:, :, None, :]
    return torch.cat([col[scenes][bi, gi].double(), rel.double()], -1).reshape(-1, 6)


def test_widest_branch_training_values_and_gradients_vs_fp64(setup):
    from gspn_amd import tf_util
    from gspn_amd.proposal_head import multi_encoding_net
    from gspn_amd.tf_grouping import query_ball_point
    xyz, t, col, shift, fps_idx = setup
    r, ns, mlp = RADII[2], NS[2], MLPS[2]
    store = tf_util.set_variable_store(tf_util.VariableStore(seed=56))
    new_xyz, fea, _, _ = multi_encoding_net(t, col, NSEED, [r], [ns], [mlp], [], True, 0.5, 'enc', use_xyz=True, shift_pred=shift, fps_idx=fps_idx)
    assert fea.shape == (B, NSEED, 256)
    idx, _ = query_ball_point(r, ns, t, new_xyz)
    rows = _rows64(t, col, shift, new_xyz, idx, list(range(B)))          # (1048576, 6) float64 on the device
    assert rows.shape[0] == 1048576
    ps = ref_params(store, 'enc', ['conv_prev_0_%d' % j for j in range(3)])
    for p in ps:
        for k in list(p):
            if torch.is_tensor(p[k]):
                p[k] = p[k].cuda()
        for k in ("w", "b", "gamma", "beta"):
            p[k] = p[k].detach().clone().requires_grad_(True)
        p["moving_mean"] = torch.zeros_like(p["moving_mean"]); p["moving_var"] = torch.ones_like(p["moving_var"])
    h, zs = rows, []
    for p in ps:
        z, _, _ = R.layer(h, p["w"], p["b"], p["gamma"], p["beta"], p["moving_mean"], p["moving_var"], True, 0.5, True, relu=False)
        zs.append(z.detach())
        h = torch.relu(z)
    fragile, _ = fragile_entries(zs, ns)
    del zs
    ref = h.view(B * NSEED, ns, 256).max(dim=1).values
    frac = float(fragile.float().mean())
    print("widest branch: fragile (silenced) gradient entries %.3f %%" % (100 * frac))
    assert frac < 0.05
    assert rel_err(fea.reshape(-1, 256), ref) < 1e-5
    g = torch.randn(ref.shape, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(8))
    g[fragile] = 0
    ref.backward(g)
    fea.backward(g.float().view(B, NSEED, 256))
    for j, p in enumerate(ps):
        nm = 'enc/conv_prev_0_%d' % j
        assert rel_err(store.vars[nm + '/weights'].grad.view(p["w"].shape), p["w"].grad) < 1e-4, nm
        assert rel_err(store.vars[nm + '/bn/gamma'].grad, p["gamma"].grad) < 1e-4, nm
        assert rel_err(store.vars[nm + '/bn/beta'].grad, p["beta"].grad) < 1e-4, nm


def test_inference_output_of_one_scene_vs_fp64(setup):
    from gspn_amd import tf_util
    from gspn_amd.proposal_head import multi_encoding_net
    from gspn_amd.tf_grouping import query_ball_point
    xyz, t, col, shift, fps_idx = setup
    store = tf_util.set_variable_store(tf_util.VariableStore(seed=57))
    with torch.no_grad():               # one training-mode forward leaves non-trivial moving statistics behind
        multi_encoding_net(t, col, NSEED, RADII, NS, MLPS, [], True, 0.5, 'context_encoder', use_xyz=True, shift_pred=shift, fps_idx=fps_idx)
    s = 5
    sl = slice(s, s + 1)
    with torch.no_grad():
        new_xyz, got, _, _ = multi_encoding_net(t[sl].contiguous(), col[sl].contiguous(), NSEED, RADII, NS, MLPS, [], False, None, 'context_encoder',
                                                use_xyz=True, shift_pred=shift[sl].contiguous(), fps_idx=fps_idx[sl].contiguous())
    assert got.shape == (1, NSEED, 768)
    outs = []
    for i, (r, ns) in enumerate(zip(RADII, NS)):
        idx, _ = query_ball_point(r, ns, t[sl].contiguous(), new_xyz)
        h = _rows64(t[sl], col[sl], shift[sl], new_xyz, idx, [0])
        for p in ref_params(store, 'context_encoder', ['conv_prev_%d_%d' % (i, j) for j in range(3)]):
            q = {k: (v.detach().cuda() if torch.is_tensor(v) else v) for k, v in p.items()}
            h, _, _ = R.layer(h, q["w"], q["b"], q["gamma"], q["beta"], q["moving_mean"], q["moving_var"], False, 0.5)
        outs.append(h.view(NSEED, ns, 256).max(dim=1).values)
    ref = torch.cat(outs, -1)
    err = float((got[0].double() - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err


def test_chamfer_on_2048_clouds_vs_oracle():
    from gspn_amd.proposal_head import chamfer_recons_loss
    pred0, gt, mask = _clouds(10)
    pred = pred0.clone().requires_grad_(True)
    loss = chamfer_recons_loss(pred, gt, mask)
    loss.backward()
    a, c, mk = pred0.cpu().numpy(), gt.cpu().numpy(), mask.cpu().numpy()
    sl = slice(0, 128)                                               # the oracle is a scalar loop: a slab of clouds
    d1, i1, d2, i2 = O.nn_distance(a[sl], c[sl])
    from gspn_amd.tf_nndistance import nn_distance
    g1, gi1, g2, gi2 = nn_distance(pred0, gt)
    np.testing.assert_array_equal(gi1[sl].cpu().numpy(), i1)
    np.testing.assert_array_equal(gi2[sl].cpu().numpy(), i2)
    np.testing.assert_array_equal(g1[sl].cpu().numpy(), d1)
    np.testing.assert_array_equal(g2[sl].cpu().numpy(), d2)
    per = (g1.double() + g2.double()).mean(-1).cpu().numpy()
    ref = (per * mk).sum() / (mk.sum() + 1e-8)
    assert abs(float(loss) - ref) < 1e-5 * abs(ref)
    gd = (mk / (mk.sum() + 1e-8))[:, None] / NINS
    gd = np.broadcast_to(gd, (NCLOUD, NINS)).astype(np.float32).copy()
    r1, _ = O.nn_distance_grad(a[sl], c[sl], gd[sl], i1, gd[sl], i2)
    assert rel_err(pred.grad[sl], torch.from_numpy(r1)) < 1e-4
