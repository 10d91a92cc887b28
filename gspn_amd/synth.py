"""Synthetic point clouds of SURVEY.md section 8(d): U (uniform), D (duplicates: tie stress, mirrors
dataset.py:100-105), S (room-like surfaces).  numpy.random.default_rng(seed), float32."""
import numpy as np


def cloud_u(n, seed):
    return np.random.default_rng(seed).random((n, 3), dtype=np.float32)


def cloud_d(n, seed, frac=0.1):
    rng = np.random.default_rng(seed)
    x = rng.random((n, 3), dtype=np.float32)
    k = max(1, int(n * frac))
    src = rng.integers(0, max(1, n - k), size=k)
    x[n - k:] = x[src]
    return x


def cloud_s(n, seed):
    """points on the faces of an 8x6x3 m room plus 20 random boxes"""
    rng = np.random.default_rng(seed)
    ext = np.array([8.0, 6.0, 3.0], np.float32)
    boxes = [(np.zeros(3, np.float32), ext)]
    for _ in range(20):
        lo = rng.random(3).astype(np.float32) * ext * 0.8
        sz = (rng.random(3).astype(np.float32) * 0.9 + 0.1).astype(np.float32)
        boxes.append((lo, sz))
    which = rng.integers(0, len(boxes), size=n)
    face = rng.integers(0, 6, size=n)
    uv = rng.random((n, 3), dtype=np.float32)
    los = np.stack([bx[0] for bx in boxes])[which]          # (n, 3) float32
    szs = np.stack([bx[1] for bx in boxes])[which]
    pts = (los + uv * szs).astype(np.float32)
    ax = face // 2
    rows = np.arange(n)
    pts[rows, ax] = los[rows, ax] + np.where(face % 2 == 1, szs[rows, ax], np.float32(0.0)).astype(np.float32)
    return pts


def batch(kind, b, n, seed0=0):
    f = {"U": cloud_u, "D": cloud_d, "S": cloud_s}[kind]
    return np.stack([f(n, seed0 + i) for i in range(b)]).astype(np.float32)
