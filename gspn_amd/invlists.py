"""Inverse lists (CSR) of an index tensor and a per-tensor cache of them.

The gradients of the stand-alone ops -- gather_point (tf_sampling_g.cu:183-192), group_point (tf_grouping_g.cu:66-83), three_interpolate
(tf_interpolate.cpp:131-153) -- are scatter-adds in the reference.  Here they are GATHERS through the inverse lists of the index tensor
(positions sorted by value, ties ascending): no atomics, a fixed summation order (ascending position: for three_interpolate exactly the
order of the reference's sequential loop, so its gradient is bit-identical to the reference's own compiled code), and every gradient row
is read once.  The lists depend on the indices only; they are built on first use and kept ON the index tensor object (`idx._gspn_inv`)
together with the tensor's version counter, so a training loop that reuses its geometry pays for them once.
GSPN_ATOMIC_GRADS=1 restores the atomic scatter-add kernels (order-free sums, as in the reference's CUDA ops)."""
import os

import torch

from . import _lib as L

ATOMIC_GRADS = os.environ.get("GSPN_ATOMIC_GRADS", "0") == "1"


def inverse_lists(idx2d, n):
    """idx2d (b, L) int32 with values in [0, n) -> order (b, L) int32 (positions sorted by value, ties ascending), offsets (b, n+1) int32
    -- a stable sort + searchsorted, done by gspn_inverse_lists (count / scan / fill in one workgroup per scene, then a per-value sort)."""
    idx2d = L.need(idx2d, torch.int32, 2, "idx")
    b, ln = idx2d.shape
    order = torch.empty((b, ln), dtype=torch.int32, device=idx2d.device)
    offsets = torch.empty((b, n + 1), dtype=torch.int32, device=idx2d.device)
    work = torch.empty(int(L.lib().gspn_inverse_lists_work_ints(b, ln, int(n))), dtype=torch.int32, device=idx2d.device)
    with torch.cuda.device(idx2d.device):
        L.check(L.lib().gspn_inverse_lists(b, ln, int(n), L.ptr(idx2d), L.ptr(work), L.ptr(order), L.ptr(offsets), L.stream()), "inverse_lists")
    return order, offsets


def cached_inverse_lists(idx, n):
    """(order, offsets) of idx.reshape(b, -1) for values in [0, n), cached on the tensor object `idx` (keyed by n and the tensor's version
    counter: an in-place write to idx invalidates the entry).  Under stream capture nothing is cached across captures that could
    outlive its memory pool: the lists are rebuilt inside the capture."""
    n = int(n)
    capturing = torch.cuda.is_current_stream_capturing()
    cache = getattr(idx, "_gspn_inv", None)
    if cache is not None and not capturing:
        hit = cache.get(n)
        if hit is not None and hit[0] == idx._version and hit[1].device == idx.device:
            return hit[1], hit[2]
    order, offsets = inverse_lists(idx.reshape(idx.shape[0], -1), n)
    if not capturing:
        if cache is None:
            cache = {}
            try:
                idx._gspn_inv = cache
            except Exception:
                return order, offsets
        cache[n] = (idx._version, order, offsets)
    return order, offsets
