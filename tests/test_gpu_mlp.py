"""GPU parity of the MFMA shared-MLP stack vs the fp64 restatement (oracle/mlp_ref.py).
Tolerance: 1e-5 relative (BASELINE.json north_star) on outputs; gradients 1e-4 of their scale
(the reference's own gradient tests use 1e-4: tf_grouping_op_test.py:27, tf_interpolate_op_test.py:21)."""
import numpy as np
import pytest
import torch

from oracle import mlp_ref as R

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rel_err(a, b):
    a = a.double().cpu()
    b = b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def make_params(chans, cin, seed, bn=True):
    g = torch.Generator().manual_seed(seed)
    ps = []
    for c in chans:
        lim = (6.0 / (cin + c)) ** 0.5
        p = {"w": (torch.rand(cin, c, generator=g, dtype=torch.float64) * 2 - 1) * lim,
             "b": (torch.rand(c, generator=g, dtype=torch.float64) - 0.5) * 0.2, "bn": bn}
        if bn:
            p.update(gamma=torch.rand(c, generator=g, dtype=torch.float64) + 0.5, beta=(torch.rand(c, generator=g, dtype=torch.float64) - 0.5),
                     moving_mean=torch.zeros(c, dtype=torch.float64), moving_var=torch.ones(c, dtype=torch.float64))
        ps.append(p)
        cin = c
    return ps


def to_layers(ps):
    from gspn_amd.mlp import LayerParams
    layers = []
    for p in ps:
        f = lambda t, rg=True: torch.nn.Parameter(t.float().cuda(), requires_grad=rg)
        if p["bn"]:
            layers.append(LayerParams(f(p["w"]), f(p["b"]), True, f(p["beta"]), f(p["gamma"]),
                                      p["moving_mean"].float().cuda(), p["moving_var"].float().cuda()))
        else:
            layers.append(LayerParams(f(p["w"]), f(p["b"]), False))
    return layers


def fragile_entries(zs, ns):
    """zs: the float64 pre-activations (BN output before the ReLU) of every layer of a stack, (rows, c_l) each; ns: pool size or None.
    Returns (fragile, frag_rows): the entries of the stack's OUTPUT gradient that a float32 evaluation may route differently -- (rows,)
    bool for a dense output, (groups, c_last) bool for a pooled one -- and the (rows,) rows holding a fragile hidden element.
    See check_stack for the rule."""
    rows = zs[0].shape[0]
    dev = zs[0].device
    frag_rows = torch.zeros(rows, dtype=torch.bool, device=dev)
    for li, z in enumerate(zs):
        if ns and li == len(zs) - 1:
            continue                                 # the last layer's mask matters only on the row that wins the pool
        frag_rows |= (z.abs() < 2e-5).any(dim=1)
    if not ns:
        return frag_rows, frag_rows
    groups = rows // ns
    zl = zs[-1].view(groups, ns, -1)
    top2 = torch.relu(zl).topk(2, dim=1)
    vals, win = top2.values, top2.indices[:, 0]                                                   # (groups, 2, c), (groups, c)
    near = ((vals[:, 0] - vals[:, 1]) < 1e-5 * (vals[:, 0].abs() + 1e-3)) & (vals[:, 0] > 0)
    win_row = win + torch.arange(groups, device=dev)[:, None] * ns
    win_small = zl.gather(1, win[:, None, :]).squeeze(1).abs() < 2e-5
    return near | frag_rows[win_row] | win_small, frag_rows


def check_stack(rows, ld, cin, chans, ns, training, ref_device="cpu", neg_gamma=False, max_fragile=0.02):
    """mlp_stack forward + backward against oracle/mlp_ref.py evaluated in float64 on `ref_device`"""
    from gspn_amd.mlp import mlp_stack
    g = torch.Generator().manual_seed(rows + cin)
    x64 = torch.randn(rows, ld, generator=g, dtype=torch.float64)
    x64[:, cin:] = 0
    ps = make_params(chans, cin, seed=cin)
    if neg_gamma:
        ps[-1]["gamma"][::3] *= -1.0             # every third channel of the last layer: negative BN scale (the pool then takes the group minimum of y)
    if not training:
        for p in ps:
            p["moving_mean"] = torch.randn(p["w"].shape[1], generator=g, dtype=torch.float64) * 0.1
            p["moving_var"] = torch.rand(p["w"].shape[1], generator=g, dtype=torch.float64) + 0.5
    layers = to_layers(ps)
    x = x64.float().cuda().requires_grad_(True)
    out = mlp_stack(x, cin, layers, training, 0.7, pool_ns=ns)

    xr = x64[:, :cin].to(ref_device).clone().requires_grad_(True)
    for p in ps:
        for k in p:
            if torch.is_tensor(p[k]):
                p[k] = p[k].to(ref_device)
        for k in ("w", "b", "gamma", "beta"):
            p[k] = p[k].clone().requires_grad_(True)
    # float64 restatement, layer by layer (oracle/mlp_ref.py), keeping the pre-activations.  An element whose BN output lies within
    # fp32 rounding of the ReLU kink -- or a pool group whose two largest activations agree to fp32 rounding -- may fall on the other
    # side in float32.  One such flip moves dW by a single row's contribution, which at 1e5 rows is already 1e-3 of |dW| (the sum over
    # rows of a random-sign gradient grows like sqrt(rows), not rows).  So the upstream gradient is set to ZERO exactly where a fragile
    # decision would carry it:
    #   * dense output (no pool): on the rows that hold a fragile element in any layer;
    #   * pooled output: per (group, channel) ELEMENT -- where the two largest activations of the group nearly tie, where the winning
    #     activation itself is within rounding of 0, or where the winning ROW holds a fragile element in a hidden layer.  A row of a
    #     pooled stack receives gradient only through the channels it wins, so this removes every fragile row from the weight
    #     gradients without silencing its whole group (at nsample = 256 / 512 nearly every group holds SOME fragile element; the
    #     group-level rule of round 2 would zero the whole test there).
    # Rows with a fragile hidden element still see the flip through the batch-norm mean terms of their own dX row (a 1/sqrt(rows)
    # effect, measured 2e-4 of max|dX| at 131072 rows: tools/mlp_bigcheck.py), so those rows -- and only those -- are left out of the
    # dX comparison.  The forward comparison covers all rows.  The fraction of silenced gradient entries is printed and capped.
    h, moving, zs = xr, [], []
    for li, p in enumerate(ps):
        z, mm, mv = R.layer(h, p["w"], p["b"], p.get("gamma"), p.get("beta"), p.get("moving_mean"), p.get("moving_var"), training, 0.7,
                            p.get("bn", True), relu=False)
        zs.append(z.detach())
        h = torch.relu(z)
        moving.append((mm, mv))
    ref = h
    fragile, frag_rows = fragile_entries(zs, ns)
    del zs
    if ns:
        ref = ref.view(rows // ns, ns, -1).max(dim=1).values
    frac = float(fragile.float().mean())
    print("check_stack rows=%d cin=%d chans=%s ns=%s: fragile (silenced) gradient entries %.3f %% of %d; rows left out of the dX comparison %.3f %%"
          % (rows, cin, chans, ns, 100 * frac, fragile.numel(), 100 * float(frag_rows.float().mean())))
    # measured on MI355X at every shape of this file: <= 0.8 % (printed above; round 2's group-level rule: up to 20 %); the cap is that + margin
    assert frac <= max_fragile or int(fragile.sum()) <= 4, "too many fragile gradient entries: %.3f %% (cap %.1f %%)" % (100 * frac, 100 * max_fragile)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < 1e-5
    if training:
        for lp, (mm, mv) in zip(layers, moving):
            assert rel_err(lp.moving_mean, mm) < 1e-5 and rel_err(lp.moving_variance, mv) < 1e-5

    go = torch.randn(ref.shape, generator=g, dtype=torch.float64).to(ref_device)
    go[fragile] = 0
    ref.backward(go)
    out.backward(go.float().cuda())
    tol = 1e-4
    keep = ~frag_rows
    assert rel_err(x.grad[:, :cin][keep.to(x.device)], xr.grad[keep]) < tol
    for lp, p in zip(layers, ps):
        assert rel_err(lp.weights.grad, p["w"].grad) < tol
        assert rel_err(lp.gamma.grad, p["gamma"].grad) < tol
        assert rel_err(lp.beta.grad, p["beta"].grad) < tol
        # bias feeds a batch-normalised layer: its true gradient is ~0 in training mode
        scale = max(float(p["w"].grad.abs().max()), 1e-6)
        assert float((lp.biases.grad.double().cpu() - p["b"].grad.cpu()).abs().max()) < tol * max(scale, float(p["b"].grad.abs().max()))


def check_stack_routed(rows, ld, cin, chans, ns, ref_device="cuda", tol=1e-5):
    """Gradients at north_star's 1e-5, NOTHING silenced.  check_stack above compares against a float64 evaluation that makes its
    own ReLU / arg-max decisions and therefore has to zero the upstream gradient wherever float32 might decide differently.  Here the float64
    backward is ROUTED by the decisions the GPU forward actually took: layer l's ReLU mask is sign(fma(y, scale, shift)) of the GPU's own float32
    (y, scale, shift) -- the expression every kernel applies (csrc/mlp.hip) -- and the pool takes the row the GPU's arg-max names.  With the routing equal,
    every remaining difference is float32 arithmetic, and all of dX (every row), dW, dgamma, dbeta, dbias are held to `tol` of their largest element.
    (That the decisions themselves are right is what the forward comparison at 1e-5 and check_stack establish: a wrong mask away from the kink would show
    there.)"""
    from gspn_amd.mlp import mlp_stack
    g = torch.Generator().manual_seed(rows + cin + 1)
    x64 = torch.randn(rows, ld, generator=g, dtype=torch.float64)
    x64[:, cin:] = 0
    ps = make_params(chans, cin, seed=cin + 1)
    layers = to_layers(ps)
    x = x64.float().cuda().requires_grad_(True)
    out = mlp_stack(x, cin, layers, True, 0.7, pool_ns=ns)
    node = out.grad_fn                                    # the autograd node IS the ctx of _MlpStack.forward
    masks = []
    for (_, _, _, _, _, y, _, _, scale, shift) in node.saved:
        masks.append(((y.double() * scale.double() + shift.double()) > 0).to(ref_device))      # y*scale exact in float64: the sign of the fma
    arg = node.arg.long().to(ref_device) if ns else None
    xr = x64[:, :cin].to(ref_device).clone().requires_grad_(True)
    for p in ps:
        for k in p:
            if torch.is_tensor(p[k]):
                p[k] = p[k].to(ref_device)
        for k in ("w", "b", "gamma", "beta"):
            p[k] = p[k].clone().requires_grad_(True)
    h = xr
    flips = 0
    for p, mk in zip(ps, masks):
        z, _, _ = R.layer(h, p["w"], p["b"], p["gamma"], p["beta"], p["moving_mean"], p["moving_var"], True, 0.7, True, relu=False)
        flips += int(((z.detach() > 0) != mk).sum())
        h = z * mk
    ref = h
    if ns:
        groups = rows // ns
        assert int(arg.min()) >= 0 and int(arg.max()) < ns
        ref = ref.view(groups, ns, -1).gather(1, arg[:, None, :]).squeeze(1)
    assert rel_err(out, ref) < 1e-5
    go = torch.randn(ref.shape, generator=g, dtype=torch.float64).to(ref_device)
    ref.backward(go)
    out.backward(go.float().cuda())
    errs = {"dX": rel_err(x.grad[:, :cin], xr.grad)}
    for li, (lp, p) in enumerate(zip(layers, ps)):
        errs["dW%d" % li] = rel_err(lp.weights.grad, p["w"].grad)
        errs["dgamma%d" % li] = rel_err(lp.gamma.grad, p["gamma"].grad)
        errs["dbeta%d" % li] = rel_err(lp.beta.grad, p["beta"].grad)
        # the bias feeds a batch-normalised layer: its true gradient is 0 up to rounding; hold it to tol of the weight gradient's scale
        scale = max(float(p["w"].grad.abs().max()), 1e-6)
        errs["dbias%d" % li] = float((lp.biases.grad.double().cpu() - p["b"].grad.cpu()).abs().max()) / max(scale, float(p["b"].grad.abs().max()))
    print("check_stack_routed rows=%d cin=%d chans=%s ns=%s: %d decisions where float64 alone would have differed; worst %s"
          % (rows, cin, chans, ns, flips, max(errs.items(), key=lambda kv: kv[1])))
    bad = {k: v for k, v in errs.items() if not v < tol}
    assert not bad, bad


@pytest.mark.parametrize("rows,ld,cin,chans,ns", [
    (4096, 6, 6, [32, 32, 64], 32),
    (2048, 67, 67, [64, 64, 128], 32),
    (1000, 67, 67, [64, 64, 64], None),
    (768, 384, 384, [256, 128], None),
    (524288, 8, 6, [32, 32, 64], 32),          # the seven stacks of test_mlp_stack_at_bench_sizes
    (131072, 68, 67, [64, 64, 128], 32),
    (32768, 132, 131, [128, 128, 256], 32),
    (4096, 384, 384, [256, 128], None),
    (16384, 192, 192, [128, 64], None),
    (262144, 68, 67, [64, 64, 64], None),
    (262144, 8, 6, [64, 64, 128], 32),
])
def test_mlp_stack_gradients_at_1e5_with_the_gpu_forward_s_own_routing(rows, ld, cin, chans, ns):
    check_stack_routed(rows, ld, cin, chans, ns)


@pytest.mark.parametrize("rows,ld,cin,chans,ns", [
    (4096, 6, 6, [32, 32, 64], 32),        # SA1-shaped
    (2048, 67, 67, [64, 64, 128], 32),     # SA2-shaped
    (1024, 131, 131, [128, 128, 256], 32), # SA3-shaped
    (768, 384, 384, [256, 128], None),     # FP1-shaped
    (1000, 67, 67, [64, 64, 64], None),    # FP3-shaped, ragged rows
    (512, 8, 5, [20], 16),                 # padded pitch, odd channel counts
    (130, 3, 3, [7, 33], 2),
    (1024, 768, 768, [256, 256], None),    # fa_layer1 of the 4-level networks (model_rpointnet.py:109,181): 256 + 512 input channels
    (256, 1024, 1024, [640], None),        # the widest layer this build takes (MAXCH)
])
@pytest.mark.parametrize("training", [True, False])
def test_mlp_stack_forward_backward(rows, ld, cin, chans, ns, training):
    check_stack(rows, ld, cin, chans, ns, training)


@pytest.mark.parametrize("rows,ld,cin,chans,ns", [
    (524288, 8, 6, [32, 32, 64], 32),          # SA1 of BASELINE configs[2]: 8 scenes x 2048 groups x 32 samples
    (131072, 68, 67, [64, 64, 128], 32),       # SA2
    (32768, 132, 131, [128, 128, 256], 32),    # SA3
    (4096, 384, 384, [256, 128], None),        # FP1
    (16384, 192, 192, [128, 64], None),        # FP2
    (262144, 68, 67, [64, 64, 64], None),      # FP3
    (262144, 8, 6, [64, 64, 128], 32),         # SA of BASELINE configs[1]: SA(1024, 0.1, 32, [64,64,128])
])
def test_mlp_stack_at_bench_sizes(rows, ld, cin, chans, ns):
    """the row counts, pitches and channel widths bench.py runs (other template instances, multi-chunk partial-tile reduction,
    persistent grids than the small cases select), against the float64 restatement evaluated on the device"""
    check_stack(rows, ld, cin, chans, ns, True, ref_device="cuda")


@pytest.mark.parametrize("rows,ns", [(524288, 256), (1048576, 512)])
def test_mlp_stack_at_config3_shard_sizes(rows, ns):
    """the context encoder's stacks at BASELINE configs[3]'s per-GPU shard (model_rpointnet.py:377: 8 scenes x 256 seeds x nsample
    256 / 512 grouped rows, [colour, xyz] -> 64 -> 128 -> 256, max-pool over nsample): the kernel instances that carry that leg --
    mlp_fwd_kernel<128,.>, mlp_bwd_data_kernel<128,.,true,.>, wgrad_stream_kernel<1,4,16,...> at 0.5 M / 1 M rows and bnrelu_maxpool
    with nsample != 32 -- forward AND backward against the float64 restatement evaluated on the device"""
    check_stack(rows, 8, 6, [64, 128, 256], ns, True, ref_device="cuda")


def test_mlp_stack_no_bn():
    from gspn_amd.mlp import mlp_stack
    g = torch.Generator().manual_seed(3)
    rows, cin = 640, 19
    x64 = torch.randn(rows, cin, generator=g, dtype=torch.float64)
    ps = make_params([24, 40], cin, 5, bn=False)
    layers = to_layers(ps)
    x = x64.float().cuda().requires_grad_(True)
    out = mlp_stack(x, cin, layers, True, None, pool_ns=4)
    xr = x64.clone().requires_grad_(True)
    for p in ps:
        for k in ("w", "b"):
            p[k] = p[k].clone().requires_grad_(True)
    ref, _ = R.stack(xr, ps, True, 0.9, 4)
    assert rel_err(out, ref) < 1e-5
    go = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(go)
    out.backward(go.float().cuda())
    assert rel_err(x.grad, xr.grad) < 1e-4
    for lp, p in zip(layers, ps):
        assert rel_err(lp.weights.grad, p["w"].grad) < 1e-4
        assert rel_err(lp.biases.grad, p["b"].grad) < 1e-4


def test_mfma_layout_is_not_transposed():
    """A = I-like with an asymmetric W: catches a row/col swap in the accumulator store"""
    from gspn_amd.mlp import mlp_stack, LayerParams
    rows, cin, cout = 128, 32, 64
    x = torch.zeros(rows, cin)
    for r in range(rows):
        x[r, r % cin] = 1.0 + r
    w = torch.arange(cin * cout, dtype=torch.float32).view(cin, cout) / 100.0
    lp = LayerParams(w.cuda(), torch.zeros(cout).cuda(), False)
    out = mlp_stack(x.cuda(), cin, [lp], False, None, None).cpu()
    ref = torch.relu(x.double() @ w.double())
    assert rel_err(out, ref) < 1e-6


def test_sync_bn_form_equals_the_fused_stack_on_one_rank():
    """mlp.SYNC_BN's layer-by-layer form (MFMA linear layer + parallel.sync_bn_relu) on a single rank, where the global batch IS the
    local batch: same outputs, gradients and moving statistics as the fused kernels"""
    from gspn_amd import mlp as M
    g = torch.Generator().manual_seed(17)
    rows, cin, chans, ns = 2048, 20, [32, 48], 16
    x64 = torch.randn(rows, cin, generator=g, dtype=torch.float64)
    res = []
    for sync in (False, True):
        layers = to_layers(make_params(chans, cin, seed=4))
        x = x64.float().cuda().requires_grad_(True)
        out = M._mlp_stack_sync_bn(x, cin, layers, 0.7, ns) if sync else M.mlp_stack(x, cin, layers, True, 0.7, pool_ns=ns)
        out.square().sum().backward()
        res.append((out.detach(), x.grad, [lp.weights.grad for lp in layers], [lp.gamma.grad for lp in layers], [lp.moving_variance for lp in layers]))
    assert rel_err(res[1][0], res[0][0]) < 1e-5
    assert rel_err(res[1][1], res[0][1]) < 1e-4
    for k in (2, 3, 4):
        for a, b in zip(res[1][k], res[0][k]):
            assert rel_err(a, b) < 1e-4


@pytest.mark.parametrize("rows,cin,chans", [(4096, 32, [64]), (2048 + 32, 67, [64, 128]), (1024, 131, [128, 256]), (96, 6, [32, 32, 64])])
def test_pool_in_the_forward_epilogue_equals_the_pool_kernel(rows, cin, chans, monkeypatch):
    """nsample = 32: the max-pool taken from the accumulators (+ gspn_pool32_select) returns the same pooled values, bit for bit, as
    the stand-alone bnrelu_maxpool pass over the (rows, c) tensor, and an arg-max that selects the same value; negative BN scales take
    the group MINIMUM of the raw output (BN+ReLU decreasing in y)"""
    from gspn_amd import mlp as M
    g = torch.Generator().manual_seed(rows)
    ld = (cin + 3) // 4 * 4
    x = torch.randn(rows, ld, generator=g)
    x[:, cin:] = 0
    outs = []
    for fuse in (False, True):
        monkeypatch.setattr(M, "FUSE_POOL32", fuse)
        ps = make_params(chans, cin, seed=3)
        ps[-1]["gamma"][::3] *= -1.0                      # every third channel of the pooled layer: negative scale
        layers = to_layers(ps)
        xx = x.cuda().requires_grad_(True)
        out = M.mlp_stack(xx, cin, layers, True, 0.7, pool_ns=32)
        out.square().sum().backward()
        outs.append((out.detach().clone(), xx.grad.clone(), [lp.weights.grad.clone() for lp in layers]))
    assert torch.equal(outs[0][0], outs[1][0])
    assert rel_err(outs[1][1], outs[0][1]) < 1e-6          # equal unless two rows of a group tie exactly (then either is a valid arg-max)
    for a, b in zip(outs[1][2], outs[0][2]):
        assert rel_err(a, b) < 1e-6


@pytest.mark.parametrize("rows,cin,chans,ns", [(8192, 6, [64, 128], 256), (4096 + 512, 35, [64, 128, 256], 512), (2048, 32, [64], 64), (1920, 16, [32, 36], 96)])
def test_pool_over_a_multiple_of_32_rows_from_the_tile_maxima(rows, cin, chans, ns, monkeypatch):
    """pools over 32 * sub rows (the proposal head's 256 / 512, model_rpointnet.py:68): the first largest of the forward epilogue's tile
    maxima (gspn_pool32_select_groups) returns the same pooled values, bit for bit, as the stand-alone pass over the (rows, c) tensor,
    an arg-max that selects the same value, and the same gradients; negative BN scales take the group MINIMUM of the raw output"""
    from gspn_amd import mlp as M
    g = torch.Generator().manual_seed(rows + ns)
    ld = (cin + 3) // 4 * 4
    x = torch.randn(rows, ld, generator=g)
    x[:, cin:] = 0
    outs = []
    for fuse in (False, True):
        monkeypatch.setattr(M, "FUSE_POOLN", fuse)
        ps = make_params(chans, cin, seed=5)
        ps[-1]["gamma"][::3] *= -1.0                      # every third channel of the pooled layer: negative scale
        layers = to_layers(ps)
        xx = x.cuda().requires_grad_(True)
        out = M.mlp_stack(xx, cin, layers, True, 0.7, pool_ns=ns)
        out.square().sum().backward()
        outs.append((out.detach().clone(), xx.grad.clone(), [lp.weights.grad.clone() for lp in layers], [lp.gamma.grad.clone() for lp in layers]))
    assert outs[0][0].shape == (rows // ns, chans[-1])
    assert torch.equal(outs[0][0], outs[1][0])
    assert rel_err(outs[1][1], outs[0][1]) < 1e-6          # equal unless two rows of a group tie exactly (then either is a valid arg-max)
    for a, b in zip(outs[1][2] + outs[1][3], outs[0][2] + outs[0][3]):
        assert rel_err(a, b) < 1e-6


def test_grouped_pool_select_entry_point_rejects_what_it_cannot_take():
    """gspn_pool32_select_groups: argument errors and the shapes it leaves to gspn_bnrelu_maxpool (c not a multiple of 4), like the other entry points
    of the boundary (-1 = argument rejected, -2 = unsupported: include/gspn_hip.h)"""
    from gspn_amd import _lib as L
    lib = L.lib()
    dev = torch.device("cuda", 0)
    groups, sub, c = 8, 4, 6
    vmax = torch.zeros(groups * sub, c, device=dev); amax = torch.zeros(groups * sub, c, dtype=torch.int32, device=dev)
    Y = torch.zeros(groups * sub * 32, c, device=dev); sc = torch.ones(c, device=dev); sh = torch.zeros(c, device=dev)
    out = torch.empty(groups, c, device=dev); arg = torch.empty(groups, c, dtype=torch.int32, device=dev); yarg = torch.empty(groups, c, device=dev)
    call = lambda g_, s_, c_, ld_, ya: lib.gspn_pool32_select_groups(g_, s_, c_, L.ptr(vmax), L.ptr(amax), L.ptr(Y), ld_, L.ptr(sc), L.ptr(sh), L.ptr(out), L.ptr(arg), ya, L.stream())
    assert call(groups, sub, c, c, L.ptr(yarg)) == -2          # c % 4
    assert call(groups, 0, 8, 8, L.ptr(yarg)) == -1            # sub
    assert call(groups, sub, 8, 4, L.ptr(yarg)) == -1          # ldy < c
    assert call(groups, sub, 8, 8, None) == -1                 # yarg is required
    assert call(0, sub, 8, 8, L.ptr(yarg)) == 0                # nothing to do


@pytest.mark.parametrize("rows,cin,chans,ns", [(4096, 32, [32, 64, 48], 32), (2048, 67, [64, 64, 128], 32), (3000, 20, [24, 40, 16], None), (1024, 131, [128, 256], 32)])
def test_early_coefficients_equal_the_two_product_pass(rows, cin, chans, ns, monkeypatch):
    """mlp.EARLY_R: BN reductions taken by the previous pass-B epilogue / the pool arg-max, pass A as ONE GEMM on dY -- same gradients
    as the two-product pass A (G1, Gx, g3 combined afterwards) to fp32 rounding, for every tensor of the stack"""
    from gspn_amd import mlp as M
    g = torch.Generator().manual_seed(rows + cin)
    ld = (cin + 3) // 4 * 4
    x0 = torch.randn(rows, ld, generator=g)
    x0[:, cin:] = 0
    go = torch.randn(rows // ns if ns else rows, chans[-1], generator=g).cuda()
    res = []
    for early in (True, False):
        monkeypatch.setattr(M, "EARLY_R", early)
        layers = to_layers(make_params(chans, cin, seed=2))
        x = x0.cuda().requires_grad_(True)
        out = M.mlp_stack(x, cin, layers, True, 0.7, pool_ns=ns)
        out.backward(go)
        res.append([x.grad[:, :cin].clone()] + [t.grad.clone() for lp in layers for t in lp.tensors()])
    for a_, b_ in zip(res[0], res[1]):
        scale = float(b_.abs().max())
        assert float((a_ - b_).abs().max()) <= 2e-5 * max(scale, 1e-3)


@pytest.mark.parametrize("rows,cin,chans,ns", [(65536, 20, [64, 64, 64], None),      # 64 <- 64 twice (FP3's inner layers)
                                               (131072, 6, [32, 32, 64], 32),         # 32 <- 32 (SA1's second layer)
                                               (65536 + 128, 16, [32, 64, 32, 64], None),   # 64 <- 32 and 32 <- 64; rows not a multiple of the grid
                                               (262144, 67, [64, 64, 64], None),      # FP3 at the bench's row count
                                               (65536, 12, [64, 64], 32), (65536, 12, [32, 32], 32), (65536 + 128, 12, [48, 64, 32], 32),    # pooled tops
                                               (65536, 12, [64, 128], 32), (65536 + 256, 12, [32, 128, 64], None),                          # cout = 128: two chunks of 64 columns
                                               (131072, 67, [64, 64, 128], 32)])                                                           # SA2 at the bench's row count
def test_both_backward_passes_in_one_launch_equal_the_two_pass_form(rows, cin, chans, ns, monkeypatch):
    """gspn_mlp_bwd_fused (dW and dX of a layer from one staged dY tile, the previous layer's BN reductions in its epilogue) against
    pass A + pass B: dX takes the same products in the same order (equal to rounding of the dY form), dW sums the rows in another
    order -- every gradient of the stack within 2e-5 of its scale; the fused path must really have run"""
    from gspn_amd import mlp as M
    g = torch.Generator().manual_seed(rows + cin)
    ld = (cin + 3) // 4 * 4
    x0 = torch.randn(rows, ld, generator=g)
    x0[:, cin:] = 0
    go = (torch.randn(rows // ns if ns else rows, chans[-1], generator=g) / rows).cuda()
    res, kinds = [], []
    for fused in (True, False):
        monkeypatch.setattr(M, "FUSED_BWD", fused)
        layers = to_layers(make_params(chans, cin, seed=3))
        x = x0.cuda().requires_grad_(True)
        out = M.mlp_stack(x, cin, layers, True, 0.7, pool_ns=ns)
        M.PROFILE = []
        try:
            out.backward(go)
            torch.cuda.synchronize()
            kinds.append([e[0] for e in M.PROFILE])
        finally:
            M.PROFILE = None
        res.append([x.grad[:, :cin].clone()] + [t.grad.clone() for lp in layers for t in lp.tensors()])
    assert "fused" in kinds[0] and "fused" not in kinds[1]
    for a_, b_ in zip(res[0], res[1]):
        scale = float(b_.abs().max())
        assert float((a_ - b_).abs().max()) <= 2e-5 * max(scale, 1e-9), (float((a_ - b_).abs().max()), scale)


@pytest.mark.parametrize("rows,cin,chans", [(65536, 20, [64, 64, 64]), (65536 + 4, 12, [40, 128]), (70000, 8, [32, 36])])
def test_dense_top_layer_reductions_in_a_pre_pass_equal_the_two_product_pass(rows, cin, chans, monkeypatch):
    """gspn_dense_rsum: the top layer of a stack with a dense upstream gradient takes (sum dyh, sum dyh*xhat) in one streaming pass over
    (d_out, Y) and then runs with known coefficients (one-GEMM pass A, or the fused launch) -- same gradients as the two-product pass A"""
    from gspn_amd import mlp as M
    g = torch.Generator().manual_seed(rows + cin)
    ld = (cin + 3) // 4 * 4
    x0 = torch.randn(rows, ld, generator=g)
    x0[:, cin:] = 0
    go = (torch.randn(rows, chans[-1], generator=g) / rows).cuda()
    res = []
    for pre in (True, False):
        monkeypatch.setattr(M, "DENSE_TOP_RSUM", pre)
        layers = to_layers(make_params(chans, cin, seed=5))
        x = x0.cuda().requires_grad_(True)
        out = M.mlp_stack(x, cin, layers, True, 0.7)
        out.backward(go)
        res.append([x.grad[:, :cin].clone()] + [t.grad.clone() for lp in layers for t in lp.tensors()])
    for a_, b_ in zip(res[0], res[1]):
        scale = float(b_.abs().max())
        assert float((a_ - b_).abs().max()) <= 2e-5 * max(scale, 1e-9), (float((a_ - b_).abs().max()), scale)


@pytest.mark.parametrize("stream", [True, False])
@pytest.mark.parametrize("rows,cin,chans", [(4096, 32, [32, 64]), (2048 + 64, 6, [32, 32, 64]), (8192, 20, [24, 64, 128]), (1024, 64, [64, 128])])
def test_pooled_top_layer_backward_from_its_input(rows, cin, chans, stream, monkeypatch):
    """mlp.POOLTOP_STREAM: the pooled top layer's pass B as a streaming GEMM on the layer's input (its own output is not read), against
    the float64 restatement -- with negative BN scales on the pooled layer (arg = the group minimum there), partial last tiles, and the
    shapes outside the kernel's LDS budget that fall back to the register-staged pass B"""
    from gspn_amd import mlp as M
    monkeypatch.setattr(M, "POOLTOP_STREAM", stream)
    check_stack(rows, (cin + 3) // 4 * 4, cin, chans, 32, True, neg_gamma=True)


@pytest.mark.parametrize("T,rows,nsrc,cout,side_n,per_scene", [(1, 5000, 700, 64, 3, 0), (3, 4096, 300, 32, 4, 2), (3, 999, 128, 128, 0, 1), (1, 64, 9, 8, 1, 0)])
def test_preagg_entry_points_against_numpy(T, rows, nsrc, cout, side_n, per_scene):
    """gspn_preagg_fwd / gspn_bn_finalize_parts / gspn_preagg_bwd_dy called directly: Y = sum_t w_t F[idx_t] + side.Wside + b and its column
    sums; dY = cA*relu'(.)*dz + cB*y + cC and dWside = side^T dY -- against float64 NumPy"""
    import ctypes
    from gspn_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(rows + cout)
    if per_scene:       # scene-local indices: per_scene scenes of rows/per_scene output rows and nsrc source rows each
        assert rows % per_scene == 0 or per_scene == 1
        nscene = per_scene
        rps = rows // nscene
        rows = rps * nscene
        idx = rng.integers(0, nsrc, size=(rows, T)).astype(np.int32)
        gidx = idx + (np.arange(rows) // rps)[:, None] * nsrc
        F = rng.standard_normal((nscene * nsrc, cout)).astype(np.float32)
    else:
        rps = 0
        idx = rng.integers(0, nsrc, size=(rows, T)).astype(np.int32)
        gidx = idx
        F = rng.standard_normal((nsrc, cout)).astype(np.float32)
    w = rng.random((rows, T)).astype(np.float32) if T == 3 else None
    side_ld = max(side_n, 1) + (1 if side_n == 3 else 0)
    side = rng.standard_normal((rows, side_ld)).astype(np.float32)
    Ws = rng.standard_normal((max(side_n, 1), cout)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    ref = (F.astype(np.float64)[gidx] * (w.astype(np.float64)[..., None] if w is not None else 1.0)).sum(1)
    ref = ref + side[:, :side_n].astype(np.float64) @ Ws[:side_n].astype(np.float64) + bias
    dF, didx, dside, dWs, dbias = dev(F), dev(idx), dev(side), dev(Ws), dev(bias)
    dw = dev(w) if w is not None else None
    Y = torch.empty((rows, cout), device="cuda")
    nparts = int(lib.gspn_preagg_fwd_parts(rows, cout))
    stats = torch.full((nparts * 2 * cout,), float("nan"), device="cuda")
    L.check(lib.gspn_preagg_fwd(rows, cout, T, L.ptr(dF), L.ptr(didx), L.ptr(dw), rps, nsrc if per_scene else 0, L.ptr(dside), side_ld, side_n,
                                L.ptr(dWs), L.ptr(dbias), L.ptr(Y), L.ptr(stats), L.stream()), "preagg_fwd")
    assert rel_err(Y, torch.from_numpy(ref)) < 2e-6
    gamma = rng.random(cout).astype(np.float32) + 0.5
    beta = rng.standard_normal(cout).astype(np.float32)
    mm, mv = torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")
    mean, var, scale, shift = (torch.empty(cout, device="cuda") for _ in range(4))
    L.check(lib.gspn_bn_finalize_parts(rows, cout, L.ptr(stats), nparts, L.ptr(dev(gamma)), L.ptr(dev(beta)), 1e-3, 0.9, 1, L.ptr(mm), L.ptr(mv),
                                       L.ptr(mean), L.ptr(var), L.ptr(scale), L.ptr(shift), L.stream()), "bn_finalize_parts")
    assert rel_err(mean, torch.from_numpy(ref.mean(0))) < 1e-5
    assert rel_err(var, torch.from_numpy(ref.var(0))) < 1e-4
    # backward half
    dz = rng.standard_normal((rows, cout)).astype(np.float32)
    cA, cB, cC = (rng.standard_normal(cout).astype(np.float32) for _ in range(3))
    sc, sh = scale.cpu().numpy().astype(np.float64), shift.cpu().numpy().astype(np.float64)
    y64 = Y.cpu().numpy().astype(np.float64)
    dy_ref = cA * np.where(y64 * sc + sh > 0, dz, 0.0) + cB * y64 + cC
    a = L.DyArgs()
    ddz, dcA, dcB, dcC = dev(dz), dev(cA), dev(cB), dev(cC)
    a.Y, a.ldy, a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = Y.data_ptr(), cout, ddz.data_ptr(), cout, None, None, 0
    a.scale, a.shift, a.cA, a.cB, a.cC = scale.data_ptr(), shift.data_ptr(), dcA.data_ptr(), dcB.data_ptr(), dcC.data_ptr()
    dY = torch.empty((rows, cout), device="cuda")
    part = torch.empty(int(lib.gspn_preagg_part_floats(cout, max(side_n, 1))), device="cuda")
    dWs_out = torch.full((max(side_n, 1), cout), float("nan"), device="cuda")
    nsl = ctypes.c_int(0)
    L.check(lib.gspn_preagg_bwd_dy(rows, cout, ctypes.byref(a), L.ptr(dside), side_ld, side_n, L.ptr(dY), L.ptr(part), L.ptr(dWs_out), ctypes.byref(nsl),
                                   L.stream()), "preagg_bwd_dy")
    assert 1 <= nsl.value <= 1024
    fragile = np.abs(y64 * sc + sh) < 1e-5            # the ReLU mask of an element this close to the kink may flip between fp32 and fp64
    got = dY.cpu().numpy().astype(np.float64)
    assert np.abs(np.where(fragile, 0.0, got - dy_ref)).max() / np.abs(dy_ref).max() < 1e-5
    if side_n:
        assert rel_err(dWs_out[:side_n], torch.from_numpy(side[:, :side_n].astype(np.float64).T @ got)) < 1e-5


def test_pad_rows():
    from gspn_amd import _lib as L
    x = torch.randn(1000, 3, device="cuda")
    out = torch.full((1000, 4), 7.0, device="cuda")
    L.check(L.lib().gspn_pad_rows(1000, 3, 4, L.ptr(x), L.ptr(out), L.stream()), "pad_rows")
    assert torch.equal(out[:, :3], x) and (out[:, 3] == 0).all()


@pytest.mark.parametrize("cin,cout,pooled", [(64, 64, False), (32, 32, False), (32, 64, True), (64, 128, False), (64, 32, True), (32, 128, True)])
def test_fused_backward_entry_point_with_padded_pitches(cin, cout, pooled):
    """gspn_mlp_bwd_fused through the C ABI alone, every pitch LARGER than its channel count (the Python host only ever passes tight
    tensors): dX, dW and the BN-reduction rows against a float64 evaluation of the same formulas on the device --
    dY = cA*[relu open]*dz + cB*y + cC, dW = relu(Xp*s + t)^T . dY, dX = dY . W^T, (sum dyh, sum dyh*xhat) of the previous layer"""
    import ctypes
    from gspn_amd import _lib as L
    lib = L.lib()
    dev = torch.device("cuda", 0)
    rows = 65536 + 128
    g = torch.Generator(device=dev).manual_seed(cin * 1000 + cout)
    ldy, ldz, ldxp, ldx = cout + 4, cout + 8, cin + 12, cin + 4
    Yb = torch.randn(rows, ldy, device=dev, generator=g); Y = Yb[:, :cout]
    Xb = torch.randn(rows, ldxp, device=dev, generator=g); Xp = Xb[:, :cin]
    W = torch.randn(cin, cout, device=dev, generator=g) * 0.1
    vec = lambda c, lo, hi: torch.rand(c, device=dev, generator=g) * (hi - lo) + lo
    scale, shift = vec(cout, 0.5, 1.5), vec(cout, -0.3, 0.3)
    cA, cB, cC = vec(cout, 0.5, 1.5), vec(cout, -0.01, 0.01), vec(cout, -0.01, 0.01)
    isc, ish = vec(cin, 0.5, 1.5), vec(cin, -0.3, 0.3)
    pmean, pvar = vec(cin, -0.2, 0.2), vec(cin, 0.5, 1.5)
    a = L.DyArgs()
    a.Y, a.ldy = Yb.data_ptr(), ldy
    a.scale, a.shift, a.cA, a.cB, a.cC = scale.data_ptr(), shift.data_ptr(), cA.data_ptr(), cB.data_ptr(), cC.data_ptr()
    if pooled:
        ns = 32
        dP = torch.randn(rows // ns, cout, device=dev, generator=g)
        arg = torch.randint(0, ns, (rows // ns, cout), device=dev, dtype=torch.int32, generator=g)
        a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = None, 0, dP.data_ptr(), arg.data_ptr(), ns
        dz = torch.zeros(rows // ns, ns, cout, device=dev, dtype=torch.float64)
        dz.scatter_(1, arg.long().unsqueeze(1), dP.double().unsqueeze(1))
        dz = dz.view(rows, cout)
    else:
        Zb = torch.randn(rows, ldz, device=dev, generator=g)
        a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = Zb.data_ptr(), ldz, None, None, 0
        dz = Zb[:, :cout].double()
    dXb = torch.full((rows, ldx), 7.0, device=dev)
    dW = torch.empty(cin, cout, device=dev)
    work = torch.empty(int(lib.gspn_mlp_bwd_work_bytes(rows, cin, cout)) // 4 + 4, device=dev)
    part = torch.empty(int(lib.gspn_rsum_part_floats(rows, cin)), device=dev)
    npart = ctypes.c_int(0)
    L.check(lib.gspn_mlp_bwd_fused(rows, cin, cout, ctypes.byref(a), L.ptr(W), L.ptr(Xb), ldxp, L.ptr(isc), L.ptr(ish), L.ptr(dXb), ldx, L.ptr(work),
                                   L.ptr(dW), L.ptr(pmean), L.ptr(pvar), 1e-3, L.ptr(part), ctypes.byref(npart), L.stream()), "fused")
    torch.cuda.synchronize()
    # float64 on the device; the ReLU masks are taken with the float32 expressions the kernels use (two roundings), so no element sits
    # on the other side of a kink
    open_y = (Y * scale + shift) > 0
    dyh = torch.where(open_y, dz, torch.zeros_like(dz))
    dY = cA.double() * dyh + cB.double() * Y.double() + cC.double()
    xa = torch.relu(Xp * isc + ish).double()
    rdW = xa.t() @ dY
    rdX = dY @ W.double().t()
    assert rel_err(dW, rdW) < 2e-5
    assert rel_err(dXb[:, :cin], rdX) < 2e-5
    assert bool((dXb[:, cin:] == 7.0).all())                       # the padding columns of dX are not touched
    open_x = (Xp * isc + ish) > 0
    dxh = torch.where(open_x, rdX, torch.zeros_like(rdX))
    xhat = (Xp.double() - pmean.double()) / torch.sqrt(pvar.double() + 1e-3)
    sums = part[:npart.value * 2 * cin].view(npart.value, 2, cin).double().sum(0)
    r0, r1 = dxh.sum(0), (dxh * xhat).sum(0)
    assert float((sums[0] - r0).abs().max()) <= 2e-5 * max(float(dxh.abs().sum(0).max()), 1e-9)
    assert float((sums[1] - r1).abs().max()) <= 2e-5 * max(float((dxh * xhat).abs().sum(0).max()), 1e-9)


@pytest.mark.parametrize("rows,c", [(70000, 64), (4096 + 3, 192), (100, 8), (65536, 1024)])
def test_dense_rsum_entry_point_against_float64(rows, c):
    """gspn_dense_rsum through the C ABI with padded pitches: (sum dyh, sum dyh*xhat) over the rows, dyh = [relu open] * dz"""
    import ctypes
    from gspn_amd import _lib as L
    lib = L.lib()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(rows + c)
    ldz, ldy = c + 4, c + 8
    Zb = torch.randn(rows, ldz, device=dev, generator=g); Yb = torch.randn(rows, ldy, device=dev, generator=g)
    scale = torch.rand(c, device=dev, generator=g) + 0.5; shift = torch.randn(c, device=dev, generator=g) * 0.3
    mean = torch.randn(c, device=dev, generator=g) * 0.2; var = torch.rand(c, device=dev, generator=g) + 0.5
    part = torch.empty(int(lib.gspn_rsum_part_floats(rows, c)), device=dev)
    npart = ctypes.c_int(0)
    L.check(lib.gspn_dense_rsum(rows, c, L.ptr(Zb), ldz, L.ptr(Yb), ldy, L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(var), 1e-3, L.ptr(part),
                                ctypes.byref(npart), L.stream()), "dense_rsum")
    torch.cuda.synchronize()
    Y, Z = Yb[:, :c], Zb[:, :c]
    dyh = torch.where((Y * scale + shift) > 0, Z.double(), torch.zeros(1, device=dev, dtype=torch.float64))
    xhat = (Y.double() - mean.double()) / torch.sqrt(var.double() + 1e-3)
    sums = part[:npart.value * 2 * c].view(npart.value, 2, c).double().sum(0)
    assert float((sums[0] - dyh.sum(0)).abs().max()) <= 2e-5 * float(dyh.abs().sum(0).max())
    assert float((sums[1] - (dyh * xhat).sum(0)).abs().max()) <= 2e-5 * float((dyh * xhat).abs().sum(0).max())
