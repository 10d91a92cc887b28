"""The GSPN proposal-head glue around the set-abstraction ops, same names and argument
order as models/model_rpointnet.py: multi_encoding_net (:28-77), fea_trans_net (:257-267) and the Chamfer
reconstruction loss of get_loss (:1346-1355).  Everything heavy goes through the same HIP kernels as pointnet_util.py:
one FPS shared by all radii, ball query per radius, the fused group+concat kernel (FEATURES first here, :61), the MFMA
MLP stack with the max-pool folded into its last layer, nn_distance for the Chamfer terms.
"""
import torch

from . import tf_util
from .mlp import mlp_stack
from . import pointnet_util as PU
from .geometry import SAGeometry, sa_front
from .pointnet_util import _mlp_layers, group_concat
from .tf_grouping import group_point, query_ball_point
from .tf_nndistance import nn_distance
from .tf_sampling import farthest_point_sample, gather_point


def multi_encoding_net(xyz, points, npoint, radius_list, nsample_list, mlp_list, mlp_list2, is_training, bn_decay, scope, bn=True,
                       use_xyz=False, output_shift=False, shift_pred=None, fps_idx=None):
    """model_rpointnet.py:28-77.  xyz (b,n,3), points (b,n,c) or None ->
    new_xyz (b,npoint,3), new_points (b,npoint,mlp_list2[-1] or sum_k mlp_list[k][-1]), shift_pred (b,npoint,4) or the input, fps_idx."""
    with tf_util.variable_scope(scope):
        if fps_idx is None:
            fps_idx = farthest_point_sample(npoint, xyz)                        # :46
        new_xyz = gather_point(xyz, fps_idx)                                    # :47
        b = xyz.shape[0]
        new_points_list = []
        for i in range(len(radius_list)):
            radius, nsample = radius_list[i], nsample_list[i]
            idx, pts_cnt = query_ball_point(radius, nsample, xyz, new_xyz)      # :53
            pooled = None
            # a shift without a gradient (the reference's own call, :377: shift_pred=tf.stop_gradient(shift_pred_seed)) is a constant of
            # the grouping like the centres; one that carries a gradient takes the reference's composition below
            shift_const = shift_pred is None or not shift_pred.requires_grad
            if shift_const and (points is None or use_xyz) and not xyz.requires_grad and (shift_pred is None or points is not None):
                c = 0 if points is None else points.shape[2]
                cin = c + 3
                if PU.FUSE_SA_FRONT and points is not None and len(mlp_list[i]) >= 2:
                    # fused front end (features FIRST here, :61): the grouped tensor is never written, the first conv gathers its rows
                    rel, gidx = sa_front(xyz.detach(), new_xyz.detach(), idx, shift_pred)
                    geo = SAGeometry(new_xyz, idx, pts_cnt, npoint, nsample, None, None, rel, gidx)
                    layers = _mlp_layers(mlp_list[i], cin, 'conv_prev_%d_' % i, bn)
                    pooled = PU._sa_stack_gathered(points, geo, False, cin, layers, is_training, bn_decay, nsample)
                if pooled is None and shift_pred is None:
                    # fused: concat([points[idx], xyz[idx] - new_xyz]) written straight into the MLP's input matrix (:54-63)
                    rows = group_concat(xyz, new_xyz, points, idx, xyz_first=False)
                    gcols = (0, c) if c > 0 else None                           # the xyz columns carry no gradient
                elif pooled is None:
                    rows = None                                                 # (a shape the gathering kernels do not take: composition below)
            else:
                rows = None
            if pooled is None and rows is None:
                grouped_xyz = group_point(xyz, idx) - new_xyz.unsqueeze(2)      # :54-55
                if shift_pred is not None:
                    grouped_xyz = grouped_xyz - shift_pred.unsqueeze(2)         # :56-57 (keeps the gradient to shift_pred)
                if points is not None:
                    grouped_points = group_point(points, idx)
                    if use_xyz:
                        grouped_points = torch.cat([grouped_points, grouped_xyz], dim=-1)   # :61 FEATURES first
                else:
                    grouped_points = grouped_xyz
                cin = grouped_points.shape[-1]
                rows = grouped_points.reshape(-1, cin)
                if cin % 4:
                    rows = torch.nn.functional.pad(rows, (0, 4 - cin % 4))
                gcols = None
            if pooled is None:
                layers = _mlp_layers(mlp_list[i], cin, 'conv_prev_%d_' % i, bn)     # scopes conv_prev_%d_%d (:66)
                pooled = mlp_stack(rows, cin, layers, bool(is_training), bn_decay, pool_ns=nsample, grad_cols=gcols)   # + reduce_max :68
            new_points_list.append(pooled.view(b, npoint, mlp_list[i][-1]))
        new_points = torch.cat(new_points_list, dim=-1)                         # :69
        for i, num_out_channel in enumerate(mlp_list2):
            new_points = tf_util.conv1d(new_points, num_out_channel, 1, padding='VALID', stride=1, bn=bn, is_training=is_training,
                                        scope='conv_post_%d' % i, bn_decay=bn_decay)
        if output_shift:
            shift_pred = tf_util.conv1d(new_points, 4, 1, padding='VALID', stride=1, scope='conv_shift_pred', activation_fn=None)   # :72-73
        return new_xyz, new_points, shift_pred, fps_idx


def fea_trans_net(input_fea, mlp_list, scope, is_training, bn_decay):
    """model_rpointnet.py:257-267: conv1d+BN+ReLU layers, the last one linear."""
    with tf_util.variable_scope(scope):
        net = input_fea
        nlayer = len(mlp_list)
        for i, num_out_channel in enumerate(mlp_list):
            if i < nlayer - 1:
                net = tf_util.conv1d(net, num_out_channel, 1, padding='VALID', bn=True, is_training=is_training, scope='conv%d' % i, bn_decay=bn_decay)
            else:
                net = tf_util.conv1d(net, num_out_channel, 1, padding='VALID', activation_fn=None, scope='conv%d' % i)
        return net


def chamfer_recons_loss(pc_ins_pred_normalized, pc_ins_gt_normalized, recon_valid_mask):
    """model_rpointnet.py:1350-1355: per-cloud mean of forward + backward squared NN distances, then the masked mean over clouds.
    pc_*: (B*nsmp, nsmp_ins, 3); recon_valid_mask: (B*nsmp,) 0/1 (no gradient, :1349)."""
    dists_forward, _, dists_backward, _ = nn_distance(pc_ins_pred_normalized, pc_ins_gt_normalized)
    recons_loss = (dists_forward + dists_backward).mean(dim=-1)
    mask = recon_valid_mask.detach()
    return (recons_loss * mask).sum() / (mask.sum() + 1e-8)
