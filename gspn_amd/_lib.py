"""ctypes binding of libgspn_hip.so (C ABI in include/gspn_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a tensor is not on a
ROCm device the call raises.  PyTorch is used only for device memory and streams.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSPN_HIP_LIB: load another build of the same ABI instead (tools/: kernel ablation / variant libraries built by gspn_amd.build.build(variant=...))
LIB_PATH = os.environ.get("GSPN_HIP_LIB") or os.path.join(_HERE, "lib", "libgspn_hip.so")

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_L = _c.c_long
_F = _c.c_float


class DyArgs(_c.Structure):
    """mirror of struct gspn_dy_args (include/gspn_hip.h)"""
    _fields_ = [("Y", _P), ("ldy", _I), ("dZ", _P), ("ldz", _I), ("dPool", _P), ("pool_arg", _P), ("ns", _I),
                ("scale", _P), ("shift", _P), ("cA", _P), ("cB", _P), ("cC", _P)]


class GatherArgs(_c.Structure):
    """mirror of struct gspn_gather_args (include/gspn_hip.h)"""
    _fields_ = [("feat", _P), ("ldf", _I), ("c", _I), ("gidx", _P), ("rel", _P), ("xyz_first", _I)]


# symbol -> argtypes; every entry point of include/gspn_hip.h (tests check the list against the header)
SIGNATURES = {
    "gspn_dist_policy": [],
    "gspn_abi_version": [],
    "gspn_farthestpointsampling": [_I, _I, _I, _P, _P, _P, _P],
    "gspn_fps_cells": [_I, _I, _I, _I, _P, _P, _P, _P, _P],
    "gspn_farthestpointsampling_cells": [_I, _I, _I, _P, _P, _P, _P],
    "gspn_fps_cells_prepass": [_I, _I, _P, _P, _P],
    "gspn_fps_cells_prepass_order": [_I, _I, _P, _P, _P, _P],
    "gspn_fps_cells_sample": [_I, _I, _I, _P, _P, _P, _P],
    "gspn_fps_multi_prepass": [_I, _I, _I, _P, _P, _P],
    "gspn_fps_multi_sample": [_I, _I, _I, _I, _P, _P, _P, _P],
    "gspn_farthestpointsampling_multi": [_I, _I, _I, _I, _P, _P, _P, _P],
    "gspn_fps_multi_status": [_P, _I, _I, _P],
    "gspn_gatherpoint": [_I, _I, _I, _P, _P, _P, _P],
    "gspn_scatteraddpoint": [_I, _I, _I, _P, _P, _P, _P],
    "gspn_probsample": [_I, _I, _I, _P, _P, _P, _P, _P],
    "gspn_queryballpoint": [_I, _I, _I, _F, _I, _P, _P, _P, _P, _P],
    "gspn_queryballpoint_ws": [_I, _I, _I, _F, _I, _P, _P, _P, _P, _P, _P],
    "gspn_selectionsort": [_I, _I, _I, _I, _P, _P, _P, _P],
    "gspn_knn_point": [_I, _I, _I, _I, _P, _P, _P, _P, _P],
    "gspn_grouppoint": [_I, _I, _I, _I, _I, _P, _P, _P, _P],
    "gspn_grouppoint_grad": [_I, _I, _I, _I, _I, _P, _P, _P, _P],
    "gspn_groupmaxpool": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "gspn_groupmaxpool_grad": [_I, _I, _I, _I, _P, _P, _P, _P],
    "gspn_preagg_ok": [_I],
    "gspn_preagg_fwd": [_L, _I, _I, _P, _P, _P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P],
    "gspn_preagg_bwd_dy": [_L, _I, _c.POINTER(DyArgs), _P, _I, _I, _P, _P, _P, _c.POINTER(_I), _P],
    "gspn_threenn": [_I, _I, _I, _P, _P, _P, _P, _P],
    "gspn_threenn_ordered": [_I, _I, _I, _P, _P, _P, _P, _P, _P],
    "gspn_threeinterpolate": [_I, _I, _I, _I, _P, _P, _P, _P, _P],
    "gspn_threeinterpolate_grad": [_I, _I, _I, _I, _P, _P, _P, _P, _P],
    "gspn_fp_concat": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P],
    "gspn_fp_concat_grad": [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "gspn_fp_concat_grad_csr": [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "gspn_fp_concat_grad_csr_split": [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P],
    "gspn_nmdistance": [_I, _I, _P, _I, _P, _P, _P, _P, _P, _P],
    "gspn_nmdistance_grad": [_I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "gspn_nmdistance_grad_csr": [_I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "gspn_sa_group_concat": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P],
    "gspn_pad_rows": [_L, _I, _I, _P, _P, _P],
    "gspn_sa_group_concat_grad": [_I, _I, _I, _I, _I, _P, _I, _I, _P, _P, _P],
    "gspn_sa_group_concat_grad_csr": [_I, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P, _P],
    "gspn_mlp_fwd": [_L, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P],
    "gspn_mlp_fwd_pool32": [_L, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P],
    "gspn_pool32_select": [_L, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P],
    "gspn_pool32_select_groups": [_L, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P],
    "gspn_sa_rel": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "gspn_sa_rel_shift": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "gspn_mlp_gather_cin": [_c.POINTER(GatherArgs)],
    "gspn_mlp_fwd_gather": [_L, _c.POINTER(GatherArgs), _I, _P, _P, _P, _I, _P, _P],
    "gspn_mlp_bwd_wgrad_gather": [_L, _c.POINTER(GatherArgs), _I, _c.POINTER(DyArgs), _P, _P, _P, _F, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "gspn_pool_rsum": [_L, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _F, _P, _c.POINTER(_I), _P],
    "gspn_dense_rsum": [_L, _I, _P, _I, _P, _I, _P, _P, _P, _P, _F, _P, _c.POINTER(_I), _P],
    "gspn_mlp_bwd_coef": [_L, _I, _I, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P],
    "gspn_mlp_bwd_wgrad_known": [_L, _I, _I, _c.POINTER(DyArgs), _P, _I, _P, _P, _c.POINTER(GatherArgs), _P, _P, _P],
    "gspn_mlp_bwd_data_ex": [_L, _I, _I, _c.POINTER(DyArgs), _P, _I, _I, _P, _I, _P, _I, _P, _P, _F, _I, _I, _P, _P,
                             _P, _I, _P, _P, _P, _P, _F, _P, _c.POINTER(_I), _P],
    "gspn_mlp_bwd_fused": [_L, _I, _I, _c.POINTER(DyArgs), _P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _F, _P, _c.POINTER(_I), _P],
    "gspn_mlp_bwd_fused_coef": [_L, _I, _I, _c.POINTER(DyArgs), _P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _F, _P, _c.POINTER(_I), _P, _P, _P, _P, _P, _P, _P, _P],
    "gspn_bn_finalize": [_L, _I, _P, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P],
    "gspn_bn_finalize_parts": [_L, _I, _P, _I, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P],
    "gspn_bn_finalize_parts_pivot": [_L, _I, _P, _I, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "gspn_bnrelu_maxpool": [_L, _I, _I, _P, _I, _P, _P, _P, _P, _P],
    "gspn_bnrelu_apply": [_L, _I, _P, _I, _P, _P, _P, _I, _P],
    "gspn_bn_colsum": [_L, _I, _P, _I, _P, _I, _P, _P, _F, _P, _c.POINTER(_I), _P],
    "gspn_bn_apply": [_L, _I, _P, _I, _P, _P, _I, _P, _I, _P],
    "gspn_bn_backward_apply": [_L, _I, _P, _I, _P, _I, _P, _P, _P, _P, _I, _P],
    "gspn_mlp_bwd_data_pooltop": [_L, _I, _I, _c.POINTER(DyArgs), _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _F, _I, _I, _P, _P,
                                  _P, _I, _P, _P, _P, _P, _F, _P, _c.POINTER(_I), _P],
    "gspn_mlp_bwd_wgrad": [_L, _I, _I, _c.POINTER(DyArgs), _P, _I, _P, _P, _P, _P, _P, _F, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "gspn_mlp_bwd_dw": [_L, _I, _I, _c.POINTER(DyArgs), _P, _I, _P, _P, _F, _I, _I, _P, _P, _P],
    "gspn_mlp_bwd_data": [_L, _I, _I, _c.POINTER(DyArgs), _P, _P, _I, _P],
    "gspn_mlp_bwd_data_cols": [_L, _I, _I, _c.POINTER(DyArgs), _P, _I, _I, _P, _I, _P],
    "gspn_mlp_bwd_data_dw2": [_L, _I, _I, _c.POINTER(DyArgs), _P, _I, _I, _P, _I, _P, _I, _P, _P, _F, _I, _I, _P, _P, _P, _I, _I, _P, _P],
    "gspn_mlp_bwd_data_dw": [_L, _I, _I, _c.POINTER(DyArgs), _P, _I, _I, _P, _I, _P, _I, _P, _P, _F, _I, _I, _P, _P, _P],
    "gspn_inverse_lists": [_I, _I, _I, _P, _P, _P, _P, _P],
    "gspn_grouppoint_grad_ws": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "gspn_scatteraddpoint_ws": [_I, _I, _I, _P, _P, _P, _P, _P],
    "gspn_threeinterpolate_grad_ws": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "gspn_nmdistance_grad_ws": [_I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "gspn_multi_copy": [_I, _P, _P, _P, _P],
    "gspn_three_nn_weights": [_L, _P, _P, _P],
    "gspn_adam_flat": [_L, _P, _P, _P, _P, _F, _F, _F, _F, _F, _F, _L, _P],
    "gspn_adam_flat_dev": [_L, _P, _P, _P, _P, _F, _F, _F, _F, _F, _F, _P, _P],
    "gspn_dot": [_L, _P, _P, _P, _P, _P],
    "gspn_fill_zero": [_P, _L, _P],
}

# entry points that do not return an int status: symbol -> (argtypes, restype)
SPECIAL = {
    "gspn_ball_threshold": ([_F], _F),
    "gspn_mlp_bwd_work_bytes": ([_L, _I, _I], _L),
    "gspn_mlp_bwd_fused_work_bytes": ([_L, _I, _I], _L),
    "gspn_mlp_fwd_stats_bytes": ([_L, _I], _L),
    "gspn_fps_cells_ws_bytes": ([_I, _I], _L),
    "gspn_fps_multi_ws_bytes": ([_I, _I], _L),
    "gspn_fps_multi_status_offset": ([_I, _I], _L),
    "gspn_rsum_part_floats": ([_L, _I], _L),
    "gspn_bn_colsum_part_floats": ([_L, _I], _L),
    "gspn_preagg_part_floats": ([_I, _I], _L),
    "gspn_preagg_fwd_parts": ([_L, _I], _L),
    "gspn_pooltop_scratch_floats": ([_L, _I, _I], _L),
    "gspn_inverse_lists_work_ints": ([_I, _I, _I], _L),
    "gspn_dot_work_floats": ([], _L),
    "gspn_ball_ws_bytes": ([_I, _I, _I], _L),
    "gspn_grouppoint_grad_ws_bytes": ([_I, _I, _I, _I, _I], _L),
    "gspn_scatteraddpoint_ws_bytes": ([_I, _I, _I], _L),
    "gspn_threeinterpolate_grad_ws_bytes": ([_I, _I, _I, _I], _L),
    "gspn_nmdistance_grad_ws_bytes": ([_I, _I, _I], _L),
}

ABI_VERSION = 9         # == GSPN_ABI_VERSION of include/gspn_hip.h this binding was written against

_lib = None


class GspnHipError(RuntimeError):
    pass


def lib():
    """Load libgspn_hip.so (built by gspn_amd.build); raises loudly if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GspnHipError(
                "libgspn_hip.so not found at %s -- run `python -m gspn_amd.build` (there is no CPU fallback)" % LIB_PATH)
        h = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(h, name)          # AttributeError if the ABI and the binding drift apart
            fn.argtypes = args
            fn.restype = _I
        for name, (args, res) in SPECIAL.items():
            fn = getattr(h, name)
            fn.argtypes = args
            fn.restype = res
        got = int(h.gspn_abi_version())
        if got != ABI_VERSION:
            raise GspnHipError("libgspn_hip.so at %s has ABI version %d, this binding needs %d: rebuild it (`python -m gspn_amd.build --force`)"
                               % (LIB_PATH, got, ABI_VERSION))
        _lib = h
    return _lib


def check(rc, what):
    if rc == 0:
        # every launch through the binding is also a check point for the status words of earlier multi-CU FPS launches (the
        # consumers of a failed launch -- gather_point, the ball query, the layers -- used to run on its zero-filled output unnoticed
        # until the next farthest_point_sample call).  Free when nothing is pending; skipped inside a stream capture (no event queries there).
        # ... nor while ANY capture of this process is open (CAPTURES_OPEN, kept by graph.CapturedStep): an event query from here is
        # illegal for a capture running in global mode on another stream / thread
        if _async_status and not CAPTURES_OPEN[0] and not torch.cuda.is_current_stream_capturing():
            check_async(block=len(_async_status) > 64)          # (and the list cannot grow without bound)
        return
    if rc == -1:
        raise ValueError("%s: invalid argument (rejected like the reference's OP_REQUIRES)" % what)
    if rc == -2:
        raise NotImplementedError("%s: input outside the supported range of this build" % what)
    raise GspnHipError("%s: HIP error %d" % (what, rc))


# Device-side status words that cannot be read where the kernel is launched without a synchronisation (the multi-CU FPS: a bounded
# inter-workgroup wait that expired).  The launcher copies the word to pinned host memory behind the kernel and registers it here;
# check_async() raises for every registered word whose copy has completed with a non-zero value.  It is called by the op wrappers
# before their next launch and by PendingGeometry.get() after its wait -- the next natural synchronisation points.
_async_status = []
CAPTURES_OPEN = [0]          # number of stream captures this process has open (graph.CapturedStep increments it around its capture)


def register_async_status(host_word, event, what):
    _async_status.append((host_word, event, what))


def check_async(block=False):
    """raise GspnHipError if a kernel launched earlier reported a failure through its status word (block=True: wait for all of them)"""
    keep, bad = [], None
    for host_word, event, what in _async_status:
        if block:
            event.synchronize()
        if event.query():
            if int(host_word.item()) != 0 and bad is None:
                bad = what
        else:
            keep.append((host_word, event, what))
    _async_status[:] = keep
    if bad is not None:
        raise GspnHipError("%s: a bounded inter-workgroup wait expired (the workgroups of a scene were not co-resident -- another kernel held "
                           "the CUs); the output of that call is invalid (all indices 0 past the failure)" % bad)


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def need(t, dtype, ndim, name):
    """shape/dtype/device validation shared by the op wrappers (ValueError like TF's InvalidArgument)"""
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if t.dtype != dtype:
        raise ValueError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if ndim is not None and t.dim() != ndim:
        raise ValueError("%s must be %d-D, got shape %s" % (name, ndim, tuple(t.shape)))
    if not t.is_cuda:
        raise GspnHipError("%s is on %s: gspn_amd runs on ROCm devices only (no CPU fallback)" % (name, t.device))
    return t.contiguous()
