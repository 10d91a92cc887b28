"""Independent NumPy brute-force definitions (second opinion on the C oracle; pure vectorised numpy,
no code shared with oracle/gspn_oracle.c)."""
import numpy as np


def fma32(a, b, c):
    """correctly rounded float32 fma through float64 (24x24-bit products are exact in float64; the sum of
    an exact product and a float32 fits ~77 bits only in rare cancellation cases, where double rounding could
    differ -- callers compare index outputs, and use this for spot checks)"""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)


def dist2_cuda(p, q):
    d = (q - p).astype(np.float32)
    t = (d[..., 1] * d[..., 1]).astype(np.float32)
    t = fma32(d[..., 0], d[..., 0], t)
    return fma32(d[..., 2], d[..., 2], t)


def fps(x, m):
    n = len(x)
    temp = np.full(n, 1e38, np.float32)
    k = np.arange(n)
    out = [0]
    old = 0
    for _ in range(1, m):
        temp = np.minimum(temp, dist2_cuda(x[old][None, :], x))
        c = k[temp == temp.max()]
        old = int(c[np.lexsort((c, c % 512))][0])      # (value desc, k mod 512 asc, k asc)
        out.append(old)
    return np.array(out, np.int32)


def ball_query(radius, ns, xyz, q):
    m = len(q)
    idx = np.zeros((m, ns), np.int32)
    cnt = np.zeros(m, np.int32)
    for j in range(m):
        d = np.maximum(np.sqrt(dist2_cuda(xyz, q[j][None, :])).astype(np.float32), np.float32(1e-20))
        hits = np.nonzero(d < np.float32(radius))[0][:ns]
        if len(hits):
            idx[j, :] = hits[0]
            idx[j, :len(hits)] = hits
        cnt[j] = len(hits)
    return idx, cnt


def nn_bruteforce(a, b):
    """definition used by tf_ops/nn_distance/tf_nndistance_cpu.py:17-25: dense pairwise squared distances, min/argmin"""
    d = ((a[:, None, :].astype(np.float64) - b[None, :, :].astype(np.float64)) ** 2).sum(-1)
    return d.min(1), d.argmin(1), d.min(0), d.argmin(0)


def three_nn(x1, x2):
    d = (x2[None, :, :] - x1[:, None, :]).astype(np.float32)
    s = ((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(np.float32) + d[..., 2] * d[..., 2]).astype(np.float32)
    order = np.argsort(s, axis=1, kind="stable")[:, :3]
    return np.take_along_axis(s, order, 1), order.astype(np.int32)
