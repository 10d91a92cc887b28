"""Drop-in for tf_ops/grouping/tf_grouping.py."""
import torch

from . import _lib as L
from . import invlists

# bench.py sets this to a list to collect (start_event, end_event, b, n, m, radius, nsample) around every ball-query launch, recorded
# on the stream the kernel is launched on
PROFILE = None
EVENT_POOL = []            # timing events handed back by the consumer of PROFILE: creating / destroying HIP events in a hot loop costs the host
PROFILE_BUDGET = [1 << 30]  # launches still to be bracketed: event pairs cost host time (on ROCm 7.2 the cost per pair grows with the pairs already recorded), so bench.py brackets the first 24 steps only


def query_ball_point(radius, nsample, xyz1, xyz2):
    """tf_grouping.py:8-21 -- xyz1 (b,n,3) data, xyz2 (b,m,3) queries ->
    idx (b,m,nsample) int32, pts_cnt (b,m) int32.  Non-differentiable."""
    radius = float(radius)
    nsample = int(nsample)
    if not radius > 0:
        raise ValueError("QueryBallPoint expects positive radius")                          # tf_grouping.cpp:101
    if nsample <= 0:
        raise ValueError("QueryBallPoint expects positive nsample")                         # tf_grouping.cpp:104
    xyz1 = L.need(xyz1.detach(), torch.float32, 3, "xyz1")
    xyz2 = L.need(xyz2.detach(), torch.float32, 3, "xyz2")
    if xyz1.shape[2] != 3:
        raise ValueError("QueryBallPoint expects (batch_size, ndataset, 3) xyz1 shape.")     # tf_grouping.cpp:109
    if xyz2.shape[2] != 3 or xyz2.shape[0] != xyz1.shape[0]:
        raise ValueError("QueryBallPoint expects (batch_size, npoint, 3) xyz2 shape.")       # tf_grouping.cpp:114
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=xyz1.device)
    cnt = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        ev = None
        if PROFILE is not None and PROFILE_BUDGET[0] > 0:
            ev = (EVENT_POOL.pop() if EVENT_POOL else torch.cuda.Event(enable_timing=True),
                      EVENT_POOL.pop() if EVENT_POOL else torch.cuda.Event(enable_timing=True))
            ev[0].record()
            PROFILE_BUDGET[0] -= 1
        if n >= 8192 and nsample <= 256:
            # a workspace lets sparse clouds take the cell-grid kernel (grouping.hip: ball_grid_*); dense clouds decide on the device to scan
            ws = torch.empty(int(L.lib().gspn_ball_ws_bytes(b, n, m)), dtype=torch.uint8, device=xyz1.device)
            L.check(L.lib().gspn_queryballpoint_ws(b, n, m, radius, nsample, L.ptr(xyz1), L.ptr(xyz2), L.ptr(ws), L.ptr(idx), L.ptr(cnt), L.stream()),
                    "query_ball_point")
        else:
            L.check(L.lib().gspn_queryballpoint(b, n, m, radius, nsample, L.ptr(xyz1), L.ptr(xyz2), L.ptr(idx), L.ptr(cnt), L.stream()),
                    "query_ball_point")
        if ev is not None:
            ev[1].record()
            PROFILE.append((ev[0], ev[1], b, n, m, radius, nsample))
    return idx, cnt


def select_top_k(k, dist):
    """tf_grouping.py:23-33 -- dist (b,m,n) -> (idx (b,m,n) int32, dist_out (b,m,n)); first k columns sorted."""
    k = int(k)
    if k <= 0:
        raise ValueError("SelectionSort expects positive k")                                # tf_grouping.cpp:143
    dist = L.need(dist.detach(), torch.float32, 3, "dist")
    b, m, n = dist.shape
    outi = torch.empty((b, m, n), dtype=torch.int32, device=dist.device)
    out = torch.empty((b, m, n), dtype=torch.float32, device=dist.device)
    with torch.cuda.device(dist.device):
        L.check(L.lib().gspn_selectionsort(b, n, m, k, L.ptr(dist), L.ptr(outi), L.ptr(out), L.stream()), "select_top_k")
    return outi, out


class _GroupMaxpool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx):
        b, n, c = points.shape
        _, m, ns = idx.shape
        out = torch.empty((b, m, c), dtype=torch.float32, device=points.device)
        max_idx = torch.empty((b, m, c), dtype=torch.int32, device=points.device)
        with torch.cuda.device(points.device):
            L.check(L.lib().gspn_groupmaxpool(b, n, c, m, ns, L.ptr(points), L.ptr(idx), L.ptr(out), L.ptr(max_idx), L.stream()), "group_maxpool")
        ctx.save_for_backward(max_idx)
        ctx.n = n
        ctx.mark_non_differentiable(max_idx)
        return out, max_idx

    @staticmethod
    def backward(ctx, grad_out, _grad_idx):
        # tf_grouping.py:45-50 -> [group_maxpool_grad(points, max_idx, grad_out), None]
        (max_idx,) = ctx.saved_tensors
        b, m, c = max_idx.shape
        grad_out = grad_out.contiguous()
        g = torch.empty((b, ctx.n, c), dtype=torch.float32, device=grad_out.device)
        with torch.cuda.device(grad_out.device):
            L.check(L.lib().gspn_groupmaxpool_grad(b, ctx.n, c, m, L.ptr(grad_out), L.ptr(max_idx), L.ptr(g), L.stream()), "group_maxpool_grad")
        return g, None


def group_maxpool(points, idx):
    """tf_grouping.py:35-44 -- points (b,n,c), idx (b,m,nsample) -> out (b,m,c), max_idx (b,m,c)."""
    points = L.need(points, torch.float32, 3, "points")
    idx = L.need(idx, torch.int32, 3, "idx")
    if idx.shape[0] != points.shape[0]:
        raise ValueError("GroupMaxpool expects (batch_size, npoints, nsample) idx shape")   # tf_grouping.cpp:254
    return _GroupMaxpool.apply(points, idx)


class _GroupPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx):
        b, n, c = points.shape
        _, m, ns = idx.shape
        out = torch.empty((b, m, ns, c), dtype=torch.float32, device=points.device)
        with torch.cuda.device(points.device):
            L.check(L.lib().gspn_grouppoint(b, n, c, m, ns, L.ptr(points), L.ptr(idx), L.ptr(out), L.stream()), "group_point")
        ctx.save_for_backward(idx)
        ctx.idx_obj = idx                    # the caller's tensor OBJECT: with the opt-in cache the inverse lists of the gradient are kept on it (invlists.py)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, grad_out):
        # tf_grouping.py:63-67 -> [group_point_grad(points, idx, grad_out), None]
        (idx,) = ctx.saved_tensors
        b, m, ns = idx.shape
        grad_out = grad_out.contiguous()
        c = grad_out.shape[3]
        g = torch.empty((b, ctx.n, c), dtype=torch.float32, device=grad_out.device)
        with torch.cuda.device(grad_out.device):
            if invlists.use_atomic(c) or m * ns == 0:
                L.check(L.lib().gspn_grouppoint_grad(b, ctx.n, c, m, ns, L.ptr(grad_out), L.ptr(idx), L.ptr(g), L.stream()), "group_point_grad")
            else:
                # a gather through the inverse lists of idx (fixed order: ascending grouped position; the reference's atomicAdd,
                # tf_grouping_g.cu:66-83, defines none): no atomics, no zero fill, every gradient row read once
                order, offsets = invlists.cached_inverse_lists(ctx.idx_obj if ctx.idx_obj._version == idx._version else idx, ctx.n)
                L.check(L.lib().gspn_sa_group_concat_grad_csr(b, ctx.n, c, m, ns, L.ptr(order), L.ptr(offsets), 0, c, L.ptr(grad_out), L.ptr(g),
                                                              L.stream()), "group_point_grad(csr)")
        return g, None


def group_point(points, idx):
    """tf_grouping.py:54-62 -- points (b,n,c), idx (b,m,nsample) -> (b,m,nsample,c)."""
    points = L.need(points, torch.float32, 3, "points")
    idx = L.need(idx, torch.int32, 3, "idx")
    if idx.shape[0] != points.shape[0]:
        raise ValueError("GroupPoint expects (batch_size, npoints, nsample) idx shape")     # tf_grouping.cpp:185
    return _GroupPoint.apply(points, idx)


KNN_DIRECT_MAX_K = 32


def knn_point(k, xyz1, xyz2):
    """tf_grouping.py:71-96 -- xyz1 (b,n,c) data, xyz2 (b,m,c) queries -> val (b,m,k), idx (b,m,k).
    3-D points and k <= 32: one direct kernel (LDS-tiled scan, 64 candidates per query, replay of the reference's selection sort on
    them: same result as the reference's dense matrix + sort + slice, ties included, without the O(b*m*n) tensor).  Anything else
    takes the reference's own construction (dense squared-distance matrix, selection sort, slice)."""
    k = int(k)
    xyz1 = L.need(xyz1.detach(), torch.float32, 3, "xyz1")
    xyz2 = L.need(xyz2.detach(), torch.float32, 3, "xyz2")
    b, n, c = xyz1.shape
    m = xyz2.shape[1]
    if xyz2.shape[0] != b or xyz2.shape[2] != c:
        raise ValueError("knn_point expects xyz1 (b,n,c) and xyz2 (b,m,c)")
    if k <= 0 or k > n:
        raise ValueError("knn_point expects 0 < k <= ndataset")       # tf.slice(outi, [0,0,0], [-1,-1,k]) of an (b,m,n) tensor
    if c == 3 and k <= KNN_DIRECT_MAX_K:
        val = torch.empty((b, m, k), dtype=torch.float32, device=xyz1.device)
        idx = torch.empty((b, m, k), dtype=torch.int32, device=xyz1.device)
        with torch.cuda.device(xyz1.device):
            L.check(L.lib().gspn_knn_point(b, n, m, k, L.ptr(xyz1), L.ptr(xyz2), L.ptr(val), L.ptr(idx), L.stream()), "knn_point")
        return val, idx
    # tf.reduce_sum((tile(xyz1)-tile(xyz2))**2, -1)  (:85-87): data minus query, summed over the last axis
    diff = xyz1[:, None, :, :] - xyz2[:, :, None, :]
    sq = diff * diff
    dist = sq[..., 0]
    for l in range(1, sq.shape[-1]):
        dist = dist + sq[..., l]
    outi, out = select_top_k(k, dist.contiguous())
    return out[:, :, :k].contiguous(), outi[:, :, :k].contiguous()
