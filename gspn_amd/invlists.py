"""Inverse lists (CSR) of an index tensor and a per-tensor cache of them.

The gradients of the stand-alone ops -- gather_point (tf_sampling_g.cu:183-192), group_point (tf_grouping_g.cu:66-83), three_interpolate
(tf_interpolate.cpp:131-153) -- are scatter-adds in the reference.  Here they are GATHERS through the inverse lists of the index tensor
(positions sorted by value, ties ascending): no atomics, a fixed summation order (ascending position: for three_interpolate exactly the
order of the reference's sequential loop, so its gradient is bit-identical to the reference's own compiled code), and every gradient row
is read once.  The lists depend on the indices only.  By DEFAULT they are rebuilt inside every backward call, on the stream that runs it (no state
outlives the call -- like the atomic kernels they replace).  Keeping them across calls is OPT-IN (r05): a caller that KNOWS its
index tensor is not rewritten behind torch's back passes the lists itself (SAGeometry / FPGeometry carry them: the bench path), or enables
the per-tensor cache with `enable_cache()` / GSPN_INVLIST_CACHE=1.  The cache entry is keyed on (n, data_ptr, torch version counter,
`generation`) -- kernels of this library, hipGraph replays into a static buffer and `.data` writes do NOT bump the version counter, so code that refills an
index buffer that way must call `invalidate(idx)` (or `bump_generation()`) -- and it carries an event of the stream that built the lists, which a
different consuming stream waits for.
GSPN_ATOMIC_GRADS=1 restores the atomic scatter-add kernels (order-free sums, as in the reference's CUDA ops)."""
import os

import torch

from . import _lib as L

ATOMIC_GRADS = os.environ.get("GSPN_ATOMIC_GRADS", "0") == "1"
CACHE = os.environ.get("GSPN_INVLIST_CACHE", "0") == "1"
# Narrow rows (c < NARROW_C: the coordinate / colour gradients of group_point, every gather_point gradient) take the ATOMIC kernels by default
# (r05): at c = 3 the list walk is one thread per point and waits for its longest list -- (2048, 32) -> 8 x 32768: 170 us + 89 us for the lists
# against 43 us for the hardware fp32 atomics (bench_detail.json: roofline_ops), and the reference's own kernel is an atomicAdd with no order
# either (tf_grouping_g.cu:66-83, tf_sampling_g.cu:183-192).  GSPN_DETERMINISTIC_GRADS=1 keeps the fixed-order gather at every width.
NARROW_C = int(os.environ.get("GSPN_NARROW_GRAD_C", "16"))
DETERMINISTIC = os.environ.get("GSPN_DETERMINISTIC_GRADS", "0") == "1"


def use_atomic(c):
    """the op wrappers' choice for a scatter-add gradient of row width c"""
    return ATOMIC_GRADS or (c < NARROW_C and not DETERMINISTIC)

_generation = [0]


def enable_cache(on=True):
    """opt in to (out of) keeping inverse lists on the index tensor across calls; returns the previous setting"""
    global CACHE
    prev, CACHE = CACHE, bool(on)
    return prev


def bump_generation():
    """drop every cached list at once (call after refilling index buffers in place by means torch does not see: a graph replay, a raw kernel)"""
    _generation[0] += 1


def invalidate(idx):
    """forget the lists cached on this index tensor"""
    if getattr(idx, "_gspn_inv", None) is not None:
        idx._gspn_inv.clear()


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
    """(order, offsets) of idx.reshape(b, -1) for values in [0, n).  Cache off (default): built now, on the current stream.  Cache on: kept
    on the tensor object `idx`, keyed by n, the tensor's data pointer, its version counter and the module's generation; an entry built on
    another stream is waited for through its event.  Under stream capture nothing is cached: the lists are rebuilt inside the capture."""
    n = int(n)
    if not CACHE or torch.cuda.is_current_stream_capturing():
        return inverse_lists(idx.reshape(idx.shape[0], -1), n)
    cache = getattr(idx, "_gspn_inv", None)
    key = (idx._version, idx.data_ptr(), _generation[0])
    cur = torch.cuda.current_stream(idx.device)
    if cache is not None:
        hit = cache.get(n)
        if hit is not None and hit[0] == key and hit[1].device == idx.device:
            if hit[3] is not None and hit[4] != cur.cuda_stream:
                cur.wait_event(hit[3])                       # built on another stream: order this stream after the build ...
                hit[1].record_stream(cur)                    # ... and tell the caching allocator that THIS stream reads both tensors: an entry dropped
                hit[2].record_stream(cur)                    # (invalidate / generation / key change) while the gather still runs must not be reused under it
            return hit[1], hit[2]
    order, offsets = inverse_lists(idx.reshape(idx.shape[0], -1), n)
    if cache is None:
        cache = {}
        try:
            idx._gspn_inv = cache
        except Exception:
            return order, offsets
    ev = torch.cuda.Event()
    ev.record(cur)
    cache[n] = (key, order, offsets, ev, cur.cuda_stream)
    return order, offsets
