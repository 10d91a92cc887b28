"""Drop-in for utils/pointnet_util.py: sample_and_group, sample_and_group_all,
pointnet_sa_module, pointnet_fp_module -- same argument names and return tuples as the reference,
over torch tensors on a ROCm device.

pointnet_sa_module / pointnet_fp_module run the fused device path: FPS -> gather -> ball query ->
one grouping kernel that writes concat([xyz[idx]-new_xyz, points[idx]]) straight into the MLP's
input matrix -> MFMA MLP stack with the max-pool folded into its last layer.
"""
import os

import torch

from . import _lib as L
from . import tf_util
from .geometry import fp_geometry, sa_geometry
from .mlp import mlp_stack, preagg_ok
from .tf_grouping import group_point, knn_point, query_ball_point
from .tf_interpolate import three_interpolate, three_nn
from .tf_sampling import farthest_point_sample, gather_point


class _GroupConcat(torch.autograd.Function):
    """rows = concat([xyz[idx]-new_xyz, points[idx]]) (pointnet_util.py:41-48) as a (b*m*ns, 3+c) matrix;
    gradient flows to `points` only (xyz/new_xyz are inputs of the network)."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, points, idx, xyz_first, order, offsets):
        b, n, _ = xyz.shape
        _, m, ns = idx.shape
        c = 0 if points is None else points.shape[2]
        ld = (3 + c + 3) // 4 * 4          # row pitch padded to 16 B so the MLP stages float4 (pad columns are zero)
        out = torch.empty((b * m * ns, ld), dtype=torch.float32, device=xyz.device)
        with torch.cuda.device(xyz.device):
            L.check(L.lib().gspn_sa_group_concat(b, n, c, m, ns, L.ptr(xyz), L.ptr(new_xyz), L.ptr(points), L.ptr(idx),
                                                 int(xyz_first), ld, L.ptr(out), L.stream()), "sa_group_concat")
        ctx.save_for_backward(idx, order, offsets)
        ctx.dims = (b, n, c, m, ns, int(xyz_first), ld)
        return out

    @staticmethod
    def backward(ctx, g):
        idx, order, offsets = ctx.saved_tensors
        b, n, c, m, ns, xyz_first, ld = ctx.dims
        gp = None
        if c > 0 and ctx.needs_input_grad[2]:
            g = g.contiguous()
            gp = torch.empty((b, n, c), dtype=torch.float32, device=g.device)
            with torch.cuda.device(g.device):
                if order is not None:       # gather through the inverse lists: fixed order, no atomics
                    L.check(L.lib().gspn_sa_group_concat_grad_csr(b, n, c, m, ns, L.ptr(order), L.ptr(offsets), xyz_first, ld, L.ptr(g), L.ptr(gp),
                                                                  L.stream()), "sa_group_concat_grad_csr")
                else:
                    L.check(L.lib().gspn_sa_group_concat_grad(b, n, c, m, ns, L.ptr(idx), xyz_first, ld, L.ptr(g), L.ptr(gp), L.stream()),
                            "sa_group_concat_grad")
        return None, None, gp, None, None, None, None


class _FpConcat(torch.autograd.Function):
    """rows = concat([three_interpolate(points2, idx, weight), points1]) (pointnet_util.py:161-166) as a (b*n1, ld) matrix with a
    16-byte row pitch, in one kernel; gradients: scatter-add to points2 (as three_interpolate_grad), slice to points1."""

    @staticmethod
    def forward(ctx, points2, idx, weight, points1, order, offsets):
        b, m, c2 = points2.shape
        n = idx.shape[1]
        c1 = 0 if points1 is None else points1.shape[2]
        ld = (c2 + c1 + 3) // 4 * 4
        out = torch.empty((b * n, ld), dtype=torch.float32, device=points2.device)
        with torch.cuda.device(points2.device):
            L.check(L.lib().gspn_fp_concat(b, n, m, c2, c1, L.ptr(points2), L.ptr(idx), L.ptr(weight), L.ptr(points1), ld, L.ptr(out), L.stream()),
                    "fp_concat")
        ctx.save_for_backward(idx, weight, order, offsets)
        ctx.dims = (b, n, m, c2, c1, ld)
        return out

    @staticmethod
    def backward(ctx, g):
        idx, weight, order, offsets = ctx.saved_tensors
        b, n, m, c2, c1, ld = ctx.dims
        g = g.contiguous()
        g2 = torch.empty((b, m, c2), dtype=torch.float32, device=g.device) if ctx.needs_input_grad[0] else None
        g1 = torch.empty((b, n, c1), dtype=torch.float32, device=g.device) if (c1 > 0 and ctx.needs_input_grad[3]) else None
        if g2 is not None or g1 is not None:
            with torch.cuda.device(g.device):
                if order is not None:       # gather through the inverse lists: no atomics, the reference's summation order
                    L.check(L.lib().gspn_fp_concat_grad_csr(b, n, m, c2, c1, ld, L.ptr(g), L.ptr(order), L.ptr(offsets), L.ptr(weight),
                                                            L.ptr(g2), L.ptr(g1), L.stream()), "fp_concat_grad_csr")
                else:
                    L.check(L.lib().gspn_fp_concat_grad(b, n, m, c2, c1, ld, L.ptr(g), L.ptr(idx), L.ptr(weight), L.ptr(g2), L.ptr(g1), L.stream()),
                            "fp_concat_grad")
        return g2, None, None, g1, None, None


def fp_concat(points2, idx, weight, points1, order=None, offsets=None):
    points2 = L.need(points2, torch.float32, 3, "points2")
    idx = L.need(idx, torch.int32, 3, "idx")
    weight = L.need(weight.detach(), torch.float32, 3, "weight")
    if points1 is not None:
        points1 = L.need(points1, torch.float32, 3, "points1")
    b = points2.shape[0]
    if idx.shape[0] != b or idx.shape[2] != 3:
        raise ValueError("ThreeInterpolate expects (b,n,3) idx shape")                      # tf_interpolate.cpp:199
    if tuple(weight.shape) != tuple(idx.shape):
        raise ValueError("ThreeInterpolate expects (b,n,3) weight shape")                   # tf_interpolate.cpp:203
    if points1 is not None and (points1.shape[0] != b or points1.shape[1] != idx.shape[1]):
        raise ValueError("pointnet_fp_module: points1 must be (b, n1, c1)")
    if order is not None:
        order = L.need(order, torch.int32, 2, "order")
        offsets = L.need(offsets, torch.int32, 2, "offsets")
    return _FpConcat.apply(points2, idx, weight, points1, order, offsets)


def group_concat(xyz, new_xyz, points, idx, xyz_first=True, order=None, offsets=None):
    """fused group + centre-subtract + concat.  The coordinates are treated as constants (they are inputs of the network in the
    set-abstraction path); a caller whose xyz / new_xyz carry a gradient must take the unfused ops (group_point / gather_point have
    the reference's gradients) -- refusing here keeps that gradient from being dropped silently."""
    if xyz.requires_grad or new_xyz.requires_grad:
        raise ValueError("group_concat: xyz / new_xyz require a gradient; use the unfused path (group_point, gather_point)")
    xyz = L.need(xyz.detach(), torch.float32, 3, "xyz")
    new_xyz = L.need(new_xyz.detach(), torch.float32, 3, "new_xyz")
    idx = L.need(idx, torch.int32, 3, "idx")
    if points is not None:
        points = L.need(points, torch.float32, 3, "points")
    if order is not None:
        order = L.need(order, torch.int32, 2, "order")
        offsets = L.need(offsets, torch.int32, 2, "offsets")
    return _GroupConcat.apply(xyz, new_xyz, points, idx, xyz_first, order, offsets)


# fused SA front end: first conv2d straight from (b, n, c) features + 20 bytes per grouped row, no (b,m,ns,3+c) tensor
FUSE_FP_FRONT = os.environ.get("GSPN_FUSE_FP_FRONT", "1") != "0"
FUSE_SA_FRONT = os.environ.get("GSPN_FUSE_SA_FRONT", "1") != "0"
# list length beyond which the transposed aggregation of a pre-aggregated FP layer shares a list out over the sixteen rows of a workgroup (0 = never)
PREAGG_SPLIT_T = int(os.environ.get("GSPN_PREAGG_SPLIT_T", "128"))


class _PadCols(torch.autograd.Function):
    """(rows, c) -> (rows, ld) with zero columns appended, in one launch; gradient = the first c columns"""

    @staticmethod
    def forward(ctx, x, ld):
        rows, c = x.shape
        out = torch.empty((rows, ld), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            L.check(L.lib().gspn_pad_rows(rows, c, ld, L.ptr(x), L.ptr(out), L.stream()), "pad_rows")
        ctx.c = c
        return out

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.c].contiguous(), None


def _sa_stack_gathered(points, geometry, xyz_first, cin, layers, is_training, bn_decay, nsample):
    """the SA module's conv stack + max-pool with the first layer gathering its input rows (mlp_stack(gather=)); None when the shape is
    outside what the gathering kernels take (the caller then materialises the grouped rows as before)"""
    b, n, c = points.shape
    m, ns = geometry.idx.shape[1], geometry.idx.shape[2]
    feat = L.need(points, torch.float32, 3, "points").reshape(b * n, c)
    if c % 4:
        f4 = getattr(geometry, "feat4", None)
        if f4 is not None and not points.requires_grad and tuple(f4.shape) == (b * n, (c + 3) // 4 * 4):
            feat = f4                                                                        # padded ahead of time beside the coordinates (sa_geometry(points=...))
        else:
            feat = _PadCols.apply(feat, (c + 3) // 4 * 4)                                      # 16-byte feature rows (the pad columns are ignored)
    if preagg_ok(layers, bool(is_training), c):
        # the first layer's feature part on the b*n points instead of the b*m*ns grouped rows (mlp.py: PREAGG)
        order, offsets, idx = geometry.order, geometry.offsets, geometry.idx

        def scatter(dy, cout):
            gp = torch.empty((b, n, cout), dtype=torch.float32, device=dy.device)
            if order is not None:
                L.check(L.lib().gspn_sa_group_concat_grad_csr(b, n, cout, m, ns, L.ptr(order), L.ptr(offsets), 0, cout, L.ptr(dy), L.ptr(gp), L.stream()),
                        "sa_group_concat_grad_csr")
            else:
                L.check(L.lib().gspn_sa_group_concat_grad(b, n, cout, m, ns, L.ptr(idx), 0, cout, L.ptr(dy), L.ptr(gp), L.stream()), "sa_group_concat_grad")
            return gp.view(b * n, cout)

        pre = {"rows": b * m * ns, "c": c, "T": 1, "idx": geometry.gidx, "w": None, "per_scene_rows": 0, "per_scene_src": b * n,      # (global idx: per_scene_src = ALL source rows, the library's offset guard)
               "side": geometry.rel, "side_ld": 4, "side_n": 3, "wf0": 3 if xyz_first else 0, "ws0": 0 if xyz_first else c, "scatter": scatter}
        return mlp_stack(feat, cin, layers, bool(is_training), bn_decay, pool_ns=nsample, preagg=pre)
    g = {"rows": b * m * ns, "c": c, "xyz_first": xyz_first, "gidx": geometry.gidx, "rel": geometry.rel, "dims": (b, n, m, ns),
         "idx": geometry.idx, "order": geometry.order, "offsets": geometry.offsets}
    try:
        return mlp_stack(feat, cin, layers, bool(is_training), bn_decay, pool_ns=nsample, gather=g)
    except NotImplementedError:
        return None


def sample_and_group(npoint, radius, nsample, xyz, points, tnet_spec=None, knn=False, use_xyz=True):
    """pointnet_util.py:17-54.  Returns new_xyz (b,npoint,3), new_points (b,npoint,nsample,3+c),
    idx (b,npoint,nsample), grouped_xyz (b,npoint,nsample,3)."""
    if tnet_spec is not None:
        raise NotImplementedError("tnet_spec: `tnet` is an undefined name in the reference (pointnet_util.py:44)")
    new_xyz = gather_point(xyz, farthest_point_sample(npoint, xyz))
    if knn:
        _, idx = knn_point(nsample, xyz, new_xyz)
    else:
        idx, pts_cnt = query_ball_point(radius, nsample, xyz, new_xyz)
    grouped_xyz = group_point(xyz, idx)
    grouped_xyz = grouped_xyz - new_xyz.unsqueeze(2)                      # :42 translation normalisation
    if points is not None:
        grouped_points = group_point(points, idx)
        new_points = torch.cat([grouped_xyz, grouped_points], dim=-1) if use_xyz else grouped_points   # :48
    else:
        new_points = grouped_xyz
    return new_xyz, new_points, idx, grouped_xyz


def sample_and_group_all(xyz, points, use_xyz=True):
    """pointnet_util.py:57-82"""
    b, n, _ = xyz.shape
    new_xyz = torch.zeros((b, 1, 3), dtype=torch.float32, device=xyz.device)
    idx = torch.arange(n, dtype=torch.int32, device=xyz.device).view(1, 1, n).repeat(b, 1, 1)
    grouped_xyz = xyz.reshape(b, 1, n, 3)
    if points is not None:
        new_points = torch.cat([xyz, points], dim=2) if use_xyz else points
        new_points = new_points.unsqueeze(1)
    else:
        new_points = grouped_xyz
    return new_xyz, new_points, idx, grouped_xyz


def _mlp_layers(channels, cin, prefix, bn):
    layers = []
    for i, cout in enumerate(channels):
        layers.append(tf_util._layer_params('%s%d' % (prefix, i), cin, cout, [1, 1, cin, cout], True, 1e-3, None, bn))
        cin = cout
    return layers


def pointnet_sa_module(xyz, points, npoint, radius, nsample, mlp, mlp2, group_all, is_training, bn_decay, scope, bn=True,
                       pooling='max', tnet_spec=None, knn=False, use_xyz=True, geometry=None):
    """pointnet_util.py:85-139.  Returns new_xyz (b,npoint,3), new_points (b,npoint,mlp[-1] or mlp2[-1]), idx (b,npoint,nsample).
    `geometry` (extension): an SAGeometry computed ahead of time for this xyz/npoint/radius/nsample (geometry.py);
    None computes it inline, with identical results."""
    with tf_util.variable_scope(scope):
        b = xyz.shape[0]
        # the fused path treats coordinates as constants; predicted / shifted coordinates (xyz.requires_grad) take the reference's
        # composition below, whose group_point / gather_point gradients reach xyz (tf_grouping.py:63-67, tf_sampling.py:43-47)
        fused = ((pooling == 'max') and not group_all and not knn and tnet_spec is None and len(mlp) > 0 and (points is None or use_xyz)
                 and not xyz.requires_grad)
        if geometry is not None and (not fused or geometry.npoint != npoint or geometry.nsample != nsample):
            raise ValueError("pointnet_sa_module: precomputed geometry does not match this module")
        if fused:
            if geometry is None:
                geometry = sa_geometry(xyz, npoint, radius, nsample, inverse=points is not None and points.requires_grad)
            new_xyz, idx = geometry.new_xyz, geometry.idx
            cin = 3 + (0 if points is None else points.shape[2])
            layers = _mlp_layers(mlp, cin, 'conv', bn)
            pooled = None
            if FUSE_SA_FRONT and points is not None and geometry.rel is not None and len(mlp) >= 2:
                # fused front end: the grouped tensor is never written -- the first layer gathers its rows from `points` (mlp.py, gather=)
                pooled = _sa_stack_gathered(points, geometry, True, cin, layers, is_training, bn_decay, nsample)
            if pooled is None:
                rows = group_concat(xyz, new_xyz, points, idx, True, geometry.order, geometry.offsets)      # (b*npoint*nsample, pitch >= 3+c)
                # group_concat's gradient reads the feature columns only (xyz carries no gradient): backward skips the 3 xyz columns of dX
                gcols = (3, cin - 3) if cin > 3 else None
                pooled = mlp_stack(rows, cin, layers, bool(is_training), bn_decay, pool_ns=nsample, grad_cols=gcols)     # (b*npoint, mlp[-1])
            new_points = pooled.view(b, npoint, 1, mlp[-1])
        else:
            if group_all:
                nsample = xyz.shape[1]
                new_xyz, new_points, idx, grouped_xyz = sample_and_group_all(xyz, points, use_xyz)
            else:
                new_xyz, new_points, idx, grouped_xyz = sample_and_group(npoint, radius, nsample, xyz, points, tnet_spec, knn, use_xyz)
            for i, num_out_channel in enumerate(mlp):
                new_points = tf_util.conv2d(new_points, num_out_channel, [1, 1], padding='VALID', stride=[1, 1], bn=bn,
                                            is_training=is_training, scope='conv%d' % (i), bn_decay=bn_decay)
            if pooling == 'avg':
                new_points = tf_util.avg_pool2d(new_points, [1, nsample], stride=[1, 1], padding='VALID', scope='avgpool1')
            elif pooling == 'weighted_avg':
                dists = torch.linalg.vector_norm(grouped_xyz, dim=-1, keepdim=True)
                exp_dists = torch.exp(-dists * 5)
                weights = exp_dists / exp_dists.sum(dim=2, keepdim=True)
                new_points = (new_points * weights).sum(dim=2, keepdim=True)
            elif pooling == 'max':
                new_points = new_points.max(dim=2, keepdim=True).values
            elif pooling == 'min':
                new_points = tf_util.max_pool2d(-1 * new_points, [1, nsample], stride=[1, 1], padding='VALID', scope='minpool1')
            elif pooling == 'max_and_avg':
                avg_points = tf_util.max_pool2d(new_points, [1, nsample], stride=[1, 1], padding='VALID', scope='maxpool1')
                max_points = tf_util.avg_pool2d(new_points, [1, nsample], stride=[1, 1], padding='VALID', scope='avgpool1')
                new_points = torch.cat([avg_points, max_points], dim=-1)
        if mlp2 is None:
            mlp2 = []
        for i, num_out_channel in enumerate(mlp2):
            new_points = tf_util.conv2d(new_points.contiguous(), num_out_channel, [1, 1], padding='VALID', stride=[1, 1], bn=bn,
                                        is_training=is_training, scope='conv_post_%d' % (i), bn_decay=bn_decay)
        new_points = new_points.squeeze(2)
        return new_xyz, new_points, idx


def _fp_stack_preagg(points2, points1, geometry, cin, layers, is_training, bn_decay):
    """the FP module's conv stack with a pre-aggregated first layer (mlp_stack(preagg=)): T = 3 weighted source rows per dense row"""
    points2 = L.need(points2, torch.float32, 3, "points2")
    b, m, c2 = points2.shape
    idx = L.need(geometry.idx, torch.int32, 3, "idx")
    weight = L.need(geometry.weight.detach(), torch.float32, 3, "weight")
    n1 = idx.shape[1]
    side = None if points1 is None else L.need(points1.detach(), torch.float32, 3, "points1")
    c1 = 0 if side is None else side.shape[2]
    order, offsets = geometry.order, geometry.offsets

    def scatter(dy, cout):
        g2 = torch.empty((b, m, cout), dtype=torch.float32, device=dy.device)
        if order is not None:
            # (the order of THIS sum is the library's own -- the pre-aggregated layer is not the reference's arithmetic order anyway -- so lists longer than
            #  PREAGG_SPLIT_T entries, the clustered clouds' case, are shared out over a workgroup's rows: csr_gather.h SPLIT)
            L.check(L.lib().gspn_fp_concat_grad_csr_split(b, n1, m, cout, 0, cout, L.ptr(dy), L.ptr(order), L.ptr(offsets), L.ptr(weight), L.ptr(g2), None,
                                                          PREAGG_SPLIT_T, L.stream()), "fp_concat_grad_csr_split")
        else:
            L.check(L.lib().gspn_fp_concat_grad(b, n1, m, cout, 0, cout, L.ptr(dy), L.ptr(idx), L.ptr(weight), L.ptr(g2), None, L.stream()), "fp_concat_grad")
        return g2.view(b * m, cout)

    pre = {"rows": b * n1, "c": c2, "T": 3, "idx": idx, "w": weight, "per_scene_rows": n1, "per_scene_src": m,
           "side": side, "side_ld": max(c1, 1), "side_n": c1, "wf0": 0, "ws0": c2, "scatter": scatter}
    return mlp_stack(points2.reshape(b * m, c2), cin, layers, bool(is_training), bn_decay, pool_ns=None, preagg=pre)


def pointnet_fp_module(xyz1, xyz2, points1, points2, mlp, is_training, bn_decay, scope, bn=True, reuse=False, geometry=None):
    """pointnet_util.py:142-174.  xyz1 (b,n1,3) dense, xyz2 (b,n2,3) sparse, points1 (b,n1,c1) or None,
    points2 (b,n2,c2) -> (b,n1,mlp[-1])  (or the concatenated features when mlp == []).
    `geometry` (extension): an FPGeometry (3-NN indices + weights, :155-160) computed ahead of time; None computes it inline."""
    with tf_util.variable_scope(scope, reuse=reuse):
        if geometry is None:
            geometry = fp_geometry(xyz1, xyz2)
        idx, weight = geometry.idx, geometry.weight
        if len(mlp) == 0:
            interpolated_points = three_interpolate(points2, idx, weight)
            if points1 is not None:
                return torch.cat([interpolated_points, points1], dim=2)       # :164 (interp FIRST)
            return interpolated_points
        # fused: interpolate + concat + 16-byte row pitch in one pass, straight into the MLP's input matrix
        b, n1 = idx.shape[0], idx.shape[1]
        cin = points2.shape[2] + (0 if points1 is None else points1.shape[2])
        layers = _mlp_layers(mlp, cin, 'conv_', bn)
        c1 = 0 if points1 is None else points1.shape[2]
        if (FUSE_FP_FRONT and c1 <= 4 and (points1 is None or not points1.requires_grad) and points2.shape[2] % 4 == 0
                and preagg_ok(layers, bool(is_training), points2.shape[2])):
            # the first layer's interpolated part on the n2 sparse points (linear: interpolate(points2) . W = interpolate(points2 . W)),
            # the <= 4 skip-link columns per dense row on the side (mlp.py: PREAGG) -- the (b*n1, c2 + c1) matrix is never written
            out = _fp_stack_preagg(points2, points1, geometry, cin, layers, is_training, bn_decay)
            return out.view(b, n1, mlp[-1])
        x2d = fp_concat(points2, idx, weight, points1, geometry.order, geometry.offsets)
        # fp_concat's gradient reads the points1 columns only if points1 wants a gradient (the last FP level gets raw colours)
        c2 = points2.shape[2]
        gcols = (0, c2) if (points1 is not None and not points1.requires_grad) else None
        out = mlp_stack(x2d, cin, layers, bool(is_training), bn_decay, pool_ns=None, grad_cols=gcols)
        return out.view(b, n1, mlp[-1])
