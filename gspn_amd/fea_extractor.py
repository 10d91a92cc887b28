"""The callers in models/model_rpointnet.py that fix the benchmark shapes:
pn2_fea_extractor (:209-233) -- the 3 x SA + 3 x FP stack of BASELINE config 3.  Same name, argument
order and scope strings as the reference so checkpoints' variable names line up."""
from . import tf_util
from .pointnet_util import pointnet_fp_module, pointnet_sa_module


def pn2_fea_extractor(xyz, points, scope, is_training, bn_decay=None):
    """model_rpointnet.py:209-233.  xyz (b,n,3), points (b,n,c) -> (b,n,64)."""
    with tf_util.variable_scope(scope):
        l0_xyz, l0_points = xyz, points
        l1_xyz, l1_points, l1_indices = pointnet_sa_module(l0_xyz, l0_points, npoint=2048, radius=0.2, nsample=32, mlp=[32, 32, 64], mlp2=None,
                                                           group_all=False, is_training=is_training, bn_decay=bn_decay, scope='layer1')
        l2_xyz, l2_points, l2_indices = pointnet_sa_module(l1_xyz, l1_points, npoint=512, radius=0.4, nsample=32, mlp=[64, 64, 128], mlp2=None,
                                                           group_all=False, is_training=is_training, bn_decay=bn_decay, scope='layer2')
        l3_xyz, l3_points, l3_indices = pointnet_sa_module(l2_xyz, l2_points, npoint=128, radius=0.8, nsample=32, mlp=[128, 128, 256], mlp2=None,
                                                           group_all=False, is_training=is_training, bn_decay=bn_decay, scope='layer3')
        l2_points = pointnet_fp_module(l2_xyz, l3_xyz, l2_points, l3_points, [256, 128], is_training, bn_decay, scope='fa_layer1')
        l1_points = pointnet_fp_module(l1_xyz, l2_xyz, l1_points, l2_points, [128, 64], is_training, bn_decay, scope='fa_layer2')
        new_points = pointnet_fp_module(l0_xyz, l1_xyz, l0_points, l1_points, [64, 64, 64], is_training, bn_decay, scope='fa_layer3')
        return new_points
