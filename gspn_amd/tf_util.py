"""Drop-in for the parts of utils/tf_util.py the set-abstraction path uses: variable helpers,
conv2d / conv1d / fully_connected (1x1 only: that is all SA/FP modules use), batch-norm wrappers,
dropout and the (1, nsample) pooling helpers.  NHWC tensors, torch on a ROCm device.

TensorFlow's variable scopes are mirrored by a small VariableStore: `with variable_scope('layer1'):`
+ `get_variable('weights', ...)` create-or-reuse parameters under 'layer1/weights', so model code
written against the reference (scope strings, `bn=`, `is_training=`, `bn_decay=`) runs unchanged.
"""
import contextlib
import math

import torch

from . import _lib as L
from .mlp import mlp_linear, LayerParams, mlp_stack


# --------------------------------------------------------------------------- variables / scopes
class VariableStore:
    def __init__(self, device=None, seed=None):
        self.vars = {}
        self.trainable = []
        self.losses = {}            # tf.get_collection('losses'): full variable name -> (variable, weight decay)
        self.device = device
        self.generator = None
        if seed is not None:
            self.generator = torch.Generator(device="cpu")
            self.generator.manual_seed(seed)

    def dev(self):
        return self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())

    def parameters(self):
        return [self.vars[n] for n in self.trainable]

    def named_parameters(self):
        return [(n, self.vars[n]) for n in self.trainable]


_store = VariableStore()
_scopes = []


def set_variable_store(store):
    global _store
    _store = store
    return store


def get_variable_store():
    return _store


@contextlib.contextmanager
def variable_scope(name, reuse=None):
    _scopes.append(str(name))
    try:
        yield "/".join(_scopes)
    finally:
        _scopes.pop()


def get_variable(name, shape, initializer, trainable=True):
    full = "/".join(_scopes + [name])
    if full in _store.vars:
        v = _store.vars[full]
        if tuple(v.shape) != tuple(shape):
            raise ValueError("variable %s exists with shape %s, requested %s" % (full, tuple(v.shape), tuple(shape)))
        return v
    data = initializer(tuple(shape), _store.generator).to(dtype=torch.float32, device=_store.dev())
    v = torch.nn.Parameter(data, requires_grad=trainable) if trainable else data
    _store.vars[full] = v
    if trainable:
        _store.trainable.append(full)
    return v


def xavier_initializer():
    """tf.contrib.layers.xavier_initializer(uniform=True): U(-l, l), l = sqrt(6/(fan_in+fan_out));
    for a (kh,kw,cin,cout) kernel fan_in = kh*kw*cin, fan_out = kh*kw*cout."""
    def init(shape, gen):
        rf = 1
        for d in shape[:-2]:
            rf *= d
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return (torch.rand(shape, generator=gen) * 2.0 - 1.0) * lim
    return init


def truncated_normal_initializer(stddev):
    def init(shape, gen):
        t = torch.empty(shape)
        torch.nn.init.trunc_normal_(t, mean=0.0, std=stddev, a=-2 * stddev, b=2 * stddev, generator=gen)
        return t
    return init


def constant_initializer(value):
    return lambda shape, gen: torch.full(shape, float(value))


def _variable_on_cpu(name, shape, initializer, use_fp16=False):
    """tf_util.py:10-22.  The reference pins variables to /cpu:0 and copies them to the GPU every
    step; here they live in HBM (fp32 only)."""
    return get_variable(name, shape, initializer)


def _variable_with_weight_decay(name, shape, stddev, wd, use_xavier=True):
    """tf_util.py:24-49.  The decay term joins the 'losses' collection once per variable (TF builds the graph once; this eager mirror
    passes here on every forward), keyed by the variable's full name in the store that owns it."""
    init = xavier_initializer() if use_xavier else truncated_normal_initializer(stddev)
    var = _variable_on_cpu(name, shape, init)
    if wd is not None:
        _store.losses["/".join(_scopes + [name])] = (var, float(wd))          # tf.add_to_collection('losses', l2_loss(var)*wd)
    return var


def weight_decay_loss():
    """sum of the 'losses' collection of the current store: wd * l2_loss(var) = wd * sum(var**2) / 2 per decayed variable"""
    return sum((0.5 * (v * v).sum() * wd for v, wd in _store.losses.values()), torch.zeros((), device=_store.dev()))


# --------------------------------------------------------------------------- layers
def _bn_variables(c):
    """tf.contrib.layers.batch_norm(center=True, scale=True) variables under scope 'bn' (tf_util.py:529-534)."""
    beta = get_variable("beta", (c,), constant_initializer(0.0))
    gamma = get_variable("gamma", (c,), constant_initializer(1.0))
    mm = get_variable("moving_mean", (c,), constant_initializer(0.0), trainable=False)
    mv = get_variable("moving_variance", (c,), constant_initializer(1.0), trainable=False)
    return beta, gamma, mm, mv


def _layer_params(scope, cin, cout, kernel_shape, use_xavier, stddev, weight_decay, bn):
    with variable_scope(scope):
        kernel = _variable_with_weight_decay("weights", shape=kernel_shape, use_xavier=use_xavier, stddev=stddev, wd=weight_decay)
        biases = _variable_on_cpu("biases", [cout], constant_initializer(0.0))
        w2d = kernel.view(cin, cout)
        if bn:
            with variable_scope("bn"):
                beta, gamma, mm, mv = _bn_variables(cout)
            return LayerParams(w2d, biases, True, beta, gamma, mm, mv)
        return LayerParams(w2d, biases, False)


def conv2d(inputs, num_output_channels, kernel_size, scope, stride=[1, 1], padding='SAME', data_format='NHWC',
           use_xavier=True, stddev=1e-3, weight_decay=None, activation_fn=torch.relu, bn=False, bn_decay=None, is_training=None):
    """tf_util.py:120-185 for the 1x1 / stride-1 kernels the SA and FP modules use (pointnet_util.py:109-113,168-172).
    inputs: (B,H,W,C) -> (B,H,W,num_output_channels) = activation(BN(inputs.W + biases))."""
    kh, kw = kernel_size
    if (kh, kw) != (1, 1) or list(stride) != [1, 1] or data_format != 'NHWC':
        raise NotImplementedError("gspn_amd.tf_util.conv2d implements the 1x1 stride-1 NHWC case of the set-abstraction path")
    inputs = L.need(inputs, torch.float32, 4, "inputs")
    cin = inputs.shape[-1]
    lp = _layer_params(scope, cin, num_output_channels, [1, 1, cin, num_output_channels], use_xavier, stddev, weight_decay, bn)
    out = _apply_layer(inputs.reshape(-1, cin), cin, lp, activation_fn, bool(is_training) if is_training is not None else False, bn_decay)
    return out.view(*inputs.shape[:-1], num_output_channels)


def conv1d(inputs, num_output_channels, kernel_size, scope, stride=1, padding='SAME', data_format='NHWC',
           use_xavier=True, stddev=1e-3, weight_decay=None, activation_fn=torch.relu, bn=False, bn_decay=None, is_training=None):
    """tf_util.py:52-115 for kernel_size 1.  inputs: (B,L,C)."""
    if kernel_size != 1 or stride != 1 or data_format != 'NHWC':
        raise NotImplementedError("gspn_amd.tf_util.conv1d implements kernel_size=1, stride=1, NHWC")
    inputs = L.need(inputs, torch.float32, 3, "inputs")
    cin = inputs.shape[-1]
    lp = _layer_params(scope, cin, num_output_channels, [1, cin, num_output_channels], use_xavier, stddev, weight_decay, bn)
    out = _apply_layer(inputs.reshape(-1, cin), cin, lp, activation_fn, bool(is_training) if is_training is not None else False, bn_decay)
    return out.view(*inputs.shape[:-1], num_output_channels)


def fully_connected(inputs, num_outputs, scope, use_xavier=True, stddev=1e-3, weight_decay=None, activation_fn=torch.relu,
                    bn=False, bn_decay=None, is_training=None):
    """tf_util.py:330-366.  inputs: (B,N)."""
    inputs = L.need(inputs, torch.float32, 2, "inputs")
    cin = inputs.shape[-1]
    lp = _layer_params(scope, cin, num_outputs, [cin, num_outputs], use_xavier, stddev, weight_decay, bn)
    return _apply_layer(inputs, cin, lp, activation_fn, bool(is_training) if is_training is not None else False, bn_decay)


def _apply_layer(x2d, cin, lp, activation_fn, is_training, bn_decay):
    if activation_fn is torch.relu or activation_fn is torch.nn.functional.relu:
        return mlp_stack(x2d, cin, [lp], is_training, bn_decay, None)
    if activation_fn is None and not lp.bn:
        return mlp_linear(x2d, cin, lp)           # plain linear head (model_rpointnet.py:71-73, 262-263)
    raise NotImplementedError("activation_fn must be relu, or None without batch-norm (the cases the set-abstraction path and its heads use)")


class _BatchNormRows(torch.autograd.Function):
    """tf.contrib.layers.batch_norm(center=True, scale=True, decay, updates_collections=None) over the rows of a (rows, c) matrix
    (tf_util.py:529-534) on the HIP kernels of csrc/batchnorm.hip + the finalize / coefficient kernels of the shared-MLP path:
    statistics from per-workgroup partial sums added in double, biased variance, eps 1e-3, y = x*inv + (beta - mean*inv),
    moving = moving*decay + batch*(1-decay) updated in place."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mm, mv, is_training, decay):
        import ctypes
        from .mlp import BN_EPS
        lib = L.lib()
        rows, c = x.shape
        dev = x.device
        mean, var, scale, shift = (torch.empty(c, dtype=torch.float32, device=dev) for _ in range(4))
        out = torch.empty_like(x)
        with torch.cuda.device(dev):
            st = L.stream()
            part, npart = None, ctypes.c_int(0)
            if is_training:
                part = torch.empty(int(lib.gspn_bn_colsum_part_floats(rows, c)), dtype=torch.float32, device=dev)
                # sums about a pivot -- x's own first row -- so that E[d^2] - E[d]^2 keeps its digits when |mean| >> std
                L.check(lib.gspn_bn_colsum(rows, c, L.ptr(x), c, None, 0, L.ptr(x), None, BN_EPS, L.ptr(part), ctypes.byref(npart), st), "bn_colsum")
            L.check(lib.gspn_bn_finalize_parts_pivot(rows, c, L.ptr(part), npart.value, L.ptr(gamma), L.ptr(beta), BN_EPS, float(decay), int(is_training),
                                                     L.ptr(mm), L.ptr(mv), L.ptr(mean), L.ptr(var), L.ptr(scale), L.ptr(shift),
                                                     L.ptr(x) if is_training else None, st), "bn_finalize")
            L.check(lib.gspn_bn_apply(rows, c, L.ptr(x), c, L.ptr(scale), L.ptr(shift), 0, L.ptr(out), c, st), "bn_apply")
        ctx.save_for_backward(x, gamma, mean, var, scale)
        ctx.is_training = bool(is_training)
        return out

    @staticmethod
    def backward(ctx, dz):
        import ctypes
        from .mlp import BN_EPS
        lib = L.lib()
        x, gamma, mean, var, scale = ctx.saved_tensors
        rows, c = x.shape
        dev = x.device
        dz = dz.contiguous()
        cA, cB, cC, dgamma, dbeta, dbias = (torch.empty(c, dtype=torch.float32, device=dev) for _ in range(6))
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        with torch.cuda.device(dev):
            st = L.stream()
            part = torch.empty(int(lib.gspn_bn_colsum_part_floats(rows, c)), dtype=torch.float32, device=dev)
            npart = ctypes.c_int(0)
            L.check(lib.gspn_bn_colsum(rows, c, L.ptr(x), c, L.ptr(dz), c, L.ptr(mean), L.ptr(var), BN_EPS, L.ptr(part), ctypes.byref(npart), st), "bn_colsum")
            # (sum dz, sum dz*xhat) -> dbeta, dgamma and, in training mode, the three coefficients of dx = cA*dz + cB*x + cC
            L.check(lib.gspn_mlp_bwd_coef(rows, c, npart.value, L.ptr(part), L.ptr(mean), L.ptr(var), L.ptr(gamma), BN_EPS,
                                          L.ptr(cA), L.ptr(cB), L.ptr(cC), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dbias), st), "bn_coef")
            if dx is not None:
                if not ctx.is_training:                  # moving statistics are constants: dx = dz * inv
                    cA, cB, cC = scale, torch.zeros_like(scale), torch.zeros_like(scale)
                L.check(lib.gspn_bn_backward_apply(rows, c, L.ptr(dz), c, L.ptr(x), c, L.ptr(cA), L.ptr(cB), L.ptr(cC), L.ptr(dx), c, st), "bn_backward_apply")
        return dx, dgamma, dbeta, None, None, None, None


def batch_norm_template(inputs, is_training, scope, moments_dims_unused, bn_decay, data_format='NHWC'):
    """tf_util.py:515-534: batch normalisation over every axis but the last (BC, BLC, BHWC ...), variables beta / gamma /
    moving_mean / moving_variance under `scope`, decay 0.9 when bn_decay is None.  Runs on the HIP kernels (_BatchNormRows)."""
    if data_format != 'NHWC':
        raise NotImplementedError("gspn_amd.tf_util batch norm: NHWC (channels last) only, like every call on the set-abstraction path")
    inputs = L.need(inputs, torch.float32, None, "inputs")
    c = inputs.shape[-1]
    with variable_scope(scope):
        beta, gamma, mm, mv = _bn_variables(c)
    decay = 0.9 if bn_decay is None else float(bn_decay)
    out = _BatchNormRows.apply(inputs.reshape(-1, c), gamma, beta, mm, mv, bool(is_training), decay)
    return out.view(inputs.shape)


def batch_norm_for_fc(inputs, is_training, bn_decay, scope):
    """tf_util.py:537-548"""
    return batch_norm_template(inputs, is_training, scope, [0, ], bn_decay)


def batch_norm_for_conv1d(inputs, is_training, bn_decay, scope, data_format='NHWC'):
    """tf_util.py:551-563"""
    return batch_norm_template(inputs, is_training, scope, [0, 1], bn_decay, data_format)


def batch_norm_for_conv2d(inputs, is_training, bn_decay, scope, data_format='NHWC'):
    """tf_util.py:568-580 (stand-alone BN over N,H,W; inside conv2d the same arithmetic rides in the MLP kernels)."""
    return batch_norm_template(inputs, is_training, scope, [0, 1, 2], bn_decay, data_format)


def max_pool2d(inputs, kernel_size, scope, stride=[2, 2], padding='VALID'):
    """tf_util.py:369-393 for the (1, nsample) window used by pointnet_util.py:126-129"""
    kh, kw = kernel_size
    if kh != 1 or kw != inputs.shape[2]:
        raise NotImplementedError("only the [1, nsample] pooling window of the SA module is implemented")
    return inputs.max(dim=2, keepdim=True).values


def avg_pool2d(inputs, kernel_size, scope, stride=[2, 2], padding='VALID'):
    """tf_util.py:395-418 for the (1, nsample) window"""
    kh, kw = kernel_size
    if kh != 1 or kw != inputs.shape[2]:
        raise NotImplementedError("only the [1, nsample] pooling window of the SA module is implemented")
    return inputs.mean(dim=2, keepdim=True)


def dropout(inputs, is_training, scope, keep_prob=0.5, noise_shape=None):
    """tf_util.py:597-618"""
    return torch.nn.functional.dropout(inputs, p=1.0 - keep_prob, training=bool(is_training))
