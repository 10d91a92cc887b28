"""The callers in models/model_rpointnet.py that fix the benchmark shapes:
pn2_fea_extractor (:209-233) -- the 3 x SA + 3 x FP stack of BASELINE config 3.  Same name, argument
order and scope strings as the reference so checkpoints' variable names line up."""
from . import tf_util
from .geometry import fp_geometry, sa_geometry
from .pointnet_util import pointnet_fp_module, pointnet_sa_module

# (npoint, radius, nsample) of the three SA levels, model_rpointnet.py:213-221
PN2_SA_SPEC = ((2048, 0.2, 32), (512, 0.4, 32), (128, 0.8, 32))


def pn2_first_fps(xyz):
    """the first launch chain of pn2_geometry alone (FPS of SA level 1, ~2/3 of a scene's geometry time): a caller that prefetches several
    batches enqueues this for all of them before the rest of any (pn2_geometry(xyz, fps0=...) on the same stream)"""
    from .tf_sampling import farthest_point_sample
    return farthest_point_sample(PN2_SA_SPEC[0][0], xyz.detach(), return_order=True)


def pn2_geometry(xyz, fps0=None, points=None):
    """Everything pn2_fea_extractor derives from coordinates alone: FPS + ball query of the three SA levels and the
    3-NN weights of the three FP levels.  Feed it to pn2_fea_extractor(..., geometry=...) -- typically computed for
    the next batch on a GeometryStream (geometry.py) while the current batch trains.  fps0: pn2_first_fps(xyz), already enqueued.
    points: the batch's raw input features (colours), optional -- input-only data like the coordinates: their 16-byte-row copy for the first
    gathering layer is then prepared here as well (the result is only valid for THESE features)."""
    sa, cur = [], xyz
    for level, (npoint, radius, nsample) in enumerate(PN2_SA_SPEC):
        g = sa_geometry(cur, npoint, radius, nsample, inverse=level > 0, fps=fps0 if level == 0 else None,
                        points=points if level == 0 else None)      # level 0 groups the raw colours: no gradient flows there
        sa.append(g)
        cur = g.new_xyz
    l1, l2, l3 = sa[0].new_xyz, sa[1].new_xyz, sa[2].new_xyz
    fp = [fp_geometry(l2, l3, sa[2].scan_order), fp_geometry(l1, l2, sa[1].scan_order), fp_geometry(xyz, l1, sa[0].scan_order)]
    return {"sa": sa, "fp": fp}


def pn2_fea_extractor(xyz, points, scope, is_training, bn_decay=None, geometry=None):
    """model_rpointnet.py:209-233.  xyz (b,n,3), points (b,n,c) -> (b,n,64).
    `geometry` (extension): the result of pn2_geometry(xyz); None computes it inline."""
    sa = geometry["sa"] if geometry is not None else (None, None, None)
    fp = geometry["fp"] if geometry is not None else (None, None, None)
    (p1, r1, s1), (p2, r2, s2), (p3, r3, s3) = PN2_SA_SPEC
    with tf_util.variable_scope(scope):
        l0_xyz, l0_points = xyz, points
        l1_xyz, l1_points, l1_indices = pointnet_sa_module(l0_xyz, l0_points, npoint=p1, radius=r1, nsample=s1, mlp=[32, 32, 64], mlp2=None,
                                                           group_all=False, is_training=is_training, bn_decay=bn_decay, scope='layer1', geometry=sa[0])
        l2_xyz, l2_points, l2_indices = pointnet_sa_module(l1_xyz, l1_points, npoint=p2, radius=r2, nsample=s2, mlp=[64, 64, 128], mlp2=None,
                                                           group_all=False, is_training=is_training, bn_decay=bn_decay, scope='layer2', geometry=sa[1])
        l3_xyz, l3_points, l3_indices = pointnet_sa_module(l2_xyz, l2_points, npoint=p3, radius=r3, nsample=s3, mlp=[128, 128, 256], mlp2=None,
                                                           group_all=False, is_training=is_training, bn_decay=bn_decay, scope='layer3', geometry=sa[2])
        l2_points = pointnet_fp_module(l2_xyz, l3_xyz, l2_points, l3_points, [256, 128], is_training, bn_decay, scope='fa_layer1', geometry=fp[0])
        l1_points = pointnet_fp_module(l1_xyz, l2_xyz, l1_points, l2_points, [128, 64], is_training, bn_decay, scope='fa_layer2', geometry=fp[1])
        new_points = pointnet_fp_module(l0_xyz, l1_xyz, l0_points, l1_points, [64, 64, 64], is_training, bn_decay, scope='fa_layer3', geometry=fp[2])
        return new_points
