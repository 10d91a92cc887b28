"""tf_util.batch_norm_for_conv2d / _conv1d / _fc (utils/tf_util.py:515-580) on the HIP kernels (csrc/batchnorm.hip + gspn_bn_finalize_parts
+ gspn_mlp_bwd_coef) against a float64 restatement of tf.contrib.layers.batch_norm: biased variance, eps 1e-3,
y = x*inv + (beta - mean*inv), moving = moving*decay + batch*(1-decay).  Outputs 1e-5, gradients 1e-4 (tf_grouping_op_test.py:27)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("shape", [(4, 300, 7, 64), (2, 1000, 3), (513, 37), (3, 129, 1, 1000)])
@pytest.mark.parametrize("training", [True, False])
def test_batch_norm_matches_fp64(shape, training):
    from gspn_amd import tf_util
    store = tf_util.set_variable_store(tf_util.VariableStore(seed=3))
    g = torch.Generator().manual_seed(sum(shape))
    c = shape[-1]
    x64 = torch.randn(*shape, generator=g, dtype=torch.float64) * 1.7 + 0.4
    fn = {4: tf_util.batch_norm_for_conv2d, 3: tf_util.batch_norm_for_conv1d, 2: tf_util.batch_norm_for_fc}[len(shape)]
    x = x64.float().cuda().requires_grad_(True)
    # give the variables non-trivial values first (a forward call creates them)
    with torch.no_grad():
        fn(x.detach(), False, 0.6, 'bn')
        for k, t in (("beta", torch.rand(c, generator=g) - 0.5), ("gamma", torch.rand(c, generator=g) + 0.5),
                     ("moving_mean", torch.randn(c, generator=g) * 0.1), ("moving_variance", torch.rand(c, generator=g) + 0.5)):
            store.vars["bn/" + k].copy_(t.cuda())
    beta, gamma = store.vars["bn/beta"], store.vars["bn/gamma"]
    mm0, mv0 = store.vars["bn/moving_mean"].clone(), store.vars["bn/moving_variance"].clone()
    out = fn(x, training, 0.6, 'bn')
    assert out.shape == x.shape
    xr = x64.clone().requires_grad_(True)
    b64, g64 = beta.detach().double().cpu().requires_grad_(True), gamma.detach().double().cpu().requires_grad_(True)
    flat = xr.reshape(-1, c)
    if training:
        mean = flat.mean(0)
        var = ((flat - mean) ** 2).mean(0)
        assert rel_err(store.vars["bn/moving_mean"], mm0.double().cpu() * 0.6 + mean.detach() * 0.4) < 1e-5
        assert rel_err(store.vars["bn/moving_variance"], mv0.double().cpu() * 0.6 + var.detach() * 0.4) < 1e-5
    else:
        mean, var = mm0.double().cpu(), mv0.double().cpu()
        assert torch.equal(store.vars["bn/moving_mean"], mm0) and torch.equal(store.vars["bn/moving_variance"], mv0)
    inv = torch.rsqrt(var + 1e-3) * g64
    ref = (flat * inv + (b64 - mean * inv)).view(shape)
    assert rel_err(out, ref) < 1e-5
    go = torch.randn(*shape, generator=g, dtype=torch.float64)
    ref.backward(go)
    out.backward(go.float().cuda())
    assert rel_err(x.grad, xr.grad) < 1e-4
    assert rel_err(gamma.grad, g64.grad) < 1e-4
    assert rel_err(beta.grad, b64.grad) < 1e-4


def test_batch_norm_large_mean_small_spread():
    """x = 100 + 1e-3 * noise.  Sums of x and x^2 in float32 lose every digit of var = E[x^2] - mean^2 (round 3's kernel gave
    var = 0, or a negative number for rsqrt); the column sums are now taken about a pivot row, so the variance comes out to 1e-4."""
    from gspn_amd import tf_util
    store = tf_util.set_variable_store(tf_util.VariableStore(seed=5))
    g = torch.Generator().manual_seed(99)
    rows, c = 200000, 48
    x64 = 100.0 + 1e-3 * torch.randn(rows, c, generator=g, dtype=torch.float64)
    x = x64.float().cuda()
    out = tf_util.batch_norm_for_fc(x, True, 0.0, 'bn')       # decay 0: the moving statistics ARE the batch statistics
    assert bool(torch.isfinite(out).all())
    xf = x.double().cpu()                                   # the float32 inputs the kernel saw
    var = ((xf - xf.mean(0)) ** 2).mean(0)
    got_var = store.vars["bn/moving_variance"].double().cpu()
    got_mean = store.vars["bn/moving_mean"].double().cpu()
    assert float((got_var - var).abs().max() / var.abs().max()) < 1e-4
    assert float((got_mean - xf.mean(0)).abs().max()) < 1e-4
    ref = (xf - xf.mean(0)) * torch.rsqrt(var + 1e-3)
    assert float((out.double().cpu() - ref).abs().max()) < 2e-3      # |xhat| <= ~0.15 here; x*scale + shift rounds at 100 * ulp
