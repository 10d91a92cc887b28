"""gspn_amd -- MI355X (gfx950) implementation of GSPN's PointNet++ set-abstraction hot path.

Module and function names mirror the reference (tf_ops/*/tf_*.py, utils/pointnet_util.py,
utils/tf_util.py) so model code calls them unchanged, over torch tensors on a ROCm device.
All compute goes through the C ABI of libgspn_hip.so (include/gspn_hip.h); there is no CPU
fallback.
"""
__version__ = "0.1.0"
