"""Drop-in for tf_ops/3d_interpolation/tf_interpolate.py (CPU-only ops in the reference; on-device here)."""
import torch

from . import _lib as L
from . import invlists


def three_nn(xyz1, xyz2, order=None):
    """tf_interpolate.py:8-18 -- xyz1 (b,n,3) unknown, xyz2 (b,m,3) known ->
    dist (b,n,3) SQUARED distances, idx (b,n,3) int32.  Non-differentiable.
    order (extension): (b,n) int32, a permutation of range(n) per scene -- the order in which the unknown points are handed to the
    threads.  The result does not depend on it; a spatially coherent order (farthest_point_sample(..., return_order=True)) lets the
    64 queries of a wave agree on which known points are worth an exact evaluation (171 -> 108 us at 8 x 32768 <- 2048)."""
    xyz1 = L.need(xyz1.detach(), torch.float32, 3, "xyz1")
    xyz2 = L.need(xyz2.detach(), torch.float32, 3, "xyz2")
    if xyz1.shape[2] != 3:
        raise ValueError("ThreeNN expects (b,n,3) xyz1 shape")                              # tf_interpolate.cpp:163
    if xyz2.shape[2] != 3 or xyz2.shape[0] != xyz1.shape[0]:
        raise ValueError("ThreeNN expects (b,m,3) xyz2 shape")                              # tf_interpolate.cpp:168
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = torch.empty((b, n, 3), dtype=torch.float32, device=xyz1.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        if order is not None:
            order = L.need(order, torch.int32, 2, "order")
            if tuple(order.shape) != (b, n):
                raise ValueError("three_nn: order must be (b,n)")
            L.check(L.lib().gspn_threenn_ordered(b, n, m, L.ptr(xyz1), L.ptr(xyz2), L.ptr(order), L.ptr(dist), L.ptr(idx), L.stream()), "three_nn")
        else:
            L.check(L.lib().gspn_threenn(b, n, m, L.ptr(xyz1), L.ptr(xyz2), L.ptr(dist), L.ptr(idx), L.stream()), "three_nn")
    return dist, idx


class _ThreeInterpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx, weight):
        b, m, c = points.shape
        n = idx.shape[1]
        out = torch.empty((b, n, c), dtype=torch.float32, device=points.device)
        with torch.cuda.device(points.device):
            L.check(L.lib().gspn_threeinterpolate(b, m, c, n, L.ptr(points), L.ptr(idx), L.ptr(weight), L.ptr(out), L.stream()), "three_interpolate")
        ctx.save_for_backward(idx, weight)
        ctx.idx_obj = idx                    # the caller's tensor OBJECT: with the opt-in cache the inverse lists of the gradient are kept on it (invlists.py)
        ctx.m = m
        return out

    @staticmethod
    def backward(ctx, grad_out):
        # tf_interpolate.py:29-34 -> [three_interpolate_grad(points, idx, weight, grad_out), None, None]
        idx, weight = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        b, n, c = grad_out.shape
        g = torch.empty((b, ctx.m, c), dtype=torch.float32, device=grad_out.device)
        with torch.cuda.device(grad_out.device):
            if invlists.ATOMIC_GRADS or n == 0:
                L.check(L.lib().gspn_threeinterpolate_grad(b, n, c, ctx.m, L.ptr(grad_out), L.ptr(idx), L.ptr(weight), L.ptr(g), L.stream()),
                        "three_interpolate_grad")
            else:
                # a gather through the inverse lists of idx: contributions added in ascending (j, t) -- the order of the reference's own
                # sequential loop (tf_interpolate.cpp:131-153), bit for bit; no atomics, no zero fill
                order, offsets = invlists.cached_inverse_lists(ctx.idx_obj if ctx.idx_obj._version == idx._version else idx, ctx.m)
                L.check(L.lib().gspn_fp_concat_grad_csr(b, n, ctx.m, c, 0, c, L.ptr(grad_out), L.ptr(order), L.ptr(offsets), L.ptr(weight), L.ptr(g),
                                                        None, L.stream()), "three_interpolate_grad(csr)")
        return g, None, None


def three_interpolate(points, idx, weight):
    """tf_interpolate.py:19-28 -- points (b,m,c), idx (b,n,3) int32, weight (b,n,3) -> (b,n,c).
    Gradient flows to `points` only (no gradient to idx/weight, as in the reference)."""
    points = L.need(points, torch.float32, 3, "points")
    idx = L.need(idx, torch.int32, 3, "idx")
    weight = L.need(weight.detach(), torch.float32, 3, "weight")
    b = points.shape[0]
    if idx.shape[0] != b or idx.shape[2] != 3:
        raise ValueError("ThreeInterpolate expects (b,n,3) idx shape")                      # tf_interpolate.cpp:199
    if tuple(weight.shape) != tuple(idx.shape):
        raise ValueError("ThreeInterpolate expects (b,n,3) weight shape")                   # tf_interpolate.cpp:203
    return _ThreeInterpolate.apply(points, idx, weight)
