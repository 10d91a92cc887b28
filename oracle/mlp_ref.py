"""fp64 restatement of the shared MLP (utils/tf_util.py:120-185 conv2d 1x1 + bias + batch_norm + relu,
pointnet_util.py:123-124 reduce_max, :157-164 FP weights/concat).

TEST INFRASTRUCTURE ONLY (see oracle/gspn_oracle.c).  PARITY UNPINNED: the arithmetic lives in
TensorFlow 1.x (tf.nn.conv2d, tf.nn.bias_add, tf.contrib.layers.batch_norm, tf.reduce_max), which is
not part of /root/reference and not installable here; no reference test touches these layers.  The
restatement follows the documented TF-1.x semantics: BN over all rows (N,H,W), biased variance,
epsilon 1e-3, moving = moving*decay + batch*(1-decay), y = x*inv + (beta - mean*inv) with
inv = rsqrt(var+eps)*gamma.  Written with torch float64 on CPU so autograd supplies the
backward reference.
"""
import torch

EPS = 1e-3


def layer(x, w, b, gamma=None, beta=None, moving_mean=None, moving_var=None, is_training=True, decay=0.9, bn=True, relu=True):
    """x (rows,cin) f64; returns z (rows,cout) f64 and (new_moving_mean, new_moving_var)."""
    y = x @ w + b
    mm, mv = moving_mean, moving_var
    if bn:
        if is_training:
            mean = y.mean(0)
            var = ((y - mean) ** 2).mean(0)
            if moving_mean is not None:
                mm = moving_mean * decay + mean.detach() * (1 - decay)
                mv = moving_var * decay + var.detach() * (1 - decay)
        else:
            mean, var = moving_mean, moving_var
        inv = torch.rsqrt(var + EPS) * gamma
        y = y * inv + (beta - mean * inv)
    if relu:
        y = torch.relu(y)
    return y, mm, mv


def stack(x, params, is_training=True, decay=0.9, pool_ns=None):
    """params: list of dicts(w,b,gamma,beta,moving_mean,moving_var,bn).  Returns out, list of (mm, mv)."""
    moving = []
    for p in params:
        x, mm, mv = layer(x, p["w"], p["b"], p.get("gamma"), p.get("beta"), p.get("moving_mean"), p.get("moving_var"),
                          is_training, decay, p.get("bn", True))
        moving.append((mm, mv))
    if pool_ns:
        x = x.view(-1, pool_ns, x.shape[1]).max(dim=1).values
    return x, moving


def fp_weights(dist):
    """pointnet_util.py:157-160"""
    dist = torch.clamp(dist, min=1e-10)
    norm = (1.0 / dist).sum(dim=2, keepdim=True)
    return (1.0 / dist) / norm
