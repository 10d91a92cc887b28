"""Drop-in for tf_ops/nn_distance/tf_nndistance.py."""
import torch

from . import _lib as L
from . import invlists

LDS_GRAD_MAX_POINTS = 4096      # NMG_MAX_PTS of csrc/nndistance.hip: up to here one workgroup per cloud keeps both gradients in LDS


class _NnDistance(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        dev = xyz1.device
        dist1 = torch.empty((b, n), dtype=torch.float32, device=dev)
        idx1 = torch.empty((b, n), dtype=torch.int32, device=dev)
        dist2 = torch.empty((b, m), dtype=torch.float32, device=dev)
        idx2 = torch.empty((b, m), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            L.check(L.lib().gspn_nmdistance(b, n, L.ptr(xyz1), m, L.ptr(xyz2), L.ptr(dist1), L.ptr(idx1), L.ptr(dist2), L.ptr(idx2), L.stream()),
                    "nn_distance")
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, idx1, dist2, idx2

    @staticmethod
    def backward(ctx, grad_dist1, _gi1, grad_dist2, _gi2):
        # tf_nndistance.py:31-37: gradients flow through the two distance outputs only
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        dev = xyz1.device
        grad_dist1 = (grad_dist1 if grad_dist1 is not None else torch.zeros((b, n), device=dev)).contiguous()
        grad_dist2 = (grad_dist2 if grad_dist2 is not None else torch.zeros((b, m), device=dev)).contiguous()
        g1 = torch.empty((b, n, 3), dtype=torch.float32, device=dev)
        g2 = torch.empty((b, m, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            if n + m > LDS_GRAD_MAX_POINTS and n > 0 and m > 0 and not invlists.ATOMIC_GRADS:
                # beyond the one-workgroup-per-cloud kernel: a gather through the inverse lists of both index tensors, terms in the order
                # of the reference's sequential CPU twin (tf_nndistance.cpp:126-163) -- bit-identical to it, no atomics
                o1, f1 = invlists.inverse_lists(idx1, m)
                o2, f2 = invlists.inverse_lists(idx2, n)
                L.check(L.lib().gspn_nmdistance_grad_csr(b, n, L.ptr(xyz1), m, L.ptr(xyz2), L.ptr(grad_dist1), L.ptr(idx1), L.ptr(grad_dist2), L.ptr(idx2),
                                                         L.ptr(o1), L.ptr(f1), L.ptr(o2), L.ptr(f2), L.ptr(g1), L.ptr(g2), L.stream()), "nn_distance_grad(csr)")
            else:
                L.check(L.lib().gspn_nmdistance_grad(b, n, L.ptr(xyz1), m, L.ptr(xyz2), L.ptr(grad_dist1), L.ptr(idx1), L.ptr(grad_dist2), L.ptr(idx2),
                                                     L.ptr(g1), L.ptr(g2), L.stream()), "nn_distance_grad")
        return g1, g2


def nn_distance(xyz1, xyz2):
    """tf_nndistance.py:14-24 -- xyz1 (b,n,3), xyz2 (b,m,3) -> dist1 (b,n), idx1 (b,n), dist2 (b,m), idx2 (b,m);
    squared distances; differentiable w.r.t. both clouds through dist1/dist2."""
    xyz1 = L.need(xyz1, torch.float32, 3, "xyz1")
    xyz2 = L.need(xyz2, torch.float32, 3, "xyz2")
    if xyz1.shape[2] != 3:
        raise ValueError("NnDistance requires xyz1 be of shape (batch,#points,3)")          # tf_nndistance.cpp:51
    if xyz2.shape[2] != 3 or xyz2.shape[0] != xyz1.shape[0]:
        raise ValueError("NnDistance requires xyz2 be of shape (batch,#points,3)")          # tf_nndistance.cpp:56
    return _NnDistance.apply(xyz1, xyz2)
