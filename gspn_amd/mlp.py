"""Shared-MLP stack on the MFMA kernels (C ABI gspn_mlp_* / gspn_bn_*).

One `conv2d(1x1) + bias + batch_norm + relu` layer of utils/tf_util.py:120-185 over a
(batch, npoint, nsample, C) tensor is a row-major GEMM over rows = batch*npoint*nsample.
`mlp_stack` runs a whole `for num_out_channel in mlp:` loop of pointnet_util.py:109-113 (and the
reduce_max of :123-124 when `pool_ns` is given) as one autograd node: pre-BN activations are written
once, the BN+ReLU of layer l is applied inside the operand load of layer l+1, and backward rebuilds
dY on the fly inside the two backward GEMMs.
"""
import ctypes
import os

import torch

from . import _lib as L

BN_EPS = 1e-3      # tf.contrib.layers.batch_norm default epsilon (tf_util.py:529-534 passes none)

# The last kernel of a layer's weight-gradient pass (the double-precision sum over partial tiles) has no consumer before the
# optimiser and cannot fill the chip (latency-bound); DEFER_DW = True puts it on a side stream where it overlaps the next layer's
# kernels.  Off by default: inside a captured step the fork becomes a multi-branch hipGraph, and ROCm 7.2 replays those far slower
# than a linear graph (measured on MI355X: 3.9 -> 6.6 ms per step), which costs more than the ~0.2 ms of overlap gains.
DEFER_DW = os.environ.get("GSPN_DEFER_DW", "0") == "1"
# max-pool over nsample = 32 rows taken from the last layer's accumulators (gspn_mlp_fwd_pool32 + gspn_pool32_select) instead of a pass
# over the (rows, c) output
FUSE_POOL32 = os.environ.get("GSPN_FUSE_POOL32", "1") != "0"
# the same for pools over a multiple of 32 rows (the proposal head's 256 / 512): tile maxima from the forward launch + gspn_pool32_select_groups
FUSE_POOLN = os.environ.get("GSPN_FUSE_POOLN", "1") != "0"
# early coefficients (mlp.hip, "Early coefficients"): the BN reductions of a layer are taken before its pass A -- by the epilogue of the
# next layer's pass B, or from the pool arg-max for the top layer of a pooled stack -- so that pass A is one GEMM instead of two
EARLY_R = os.environ.get("GSPN_EARLY_R", "1") != "0"
# pre-aggregated first layer (mlp.hip, "Pre-aggregated first layer"): the feature part of an SA / FP module's first layer is multiplied on the
# source points, the grouped / interpolated rows are formed from the product.  Training-mode BN stacks with early coefficients only.
PREAGG = os.environ.get("GSPN_PREAGG", "1") != "0"
PREAGG_MIN_C = int(os.environ.get("GSPN_PREAGG_MIN_C", "16"))      # below this many feature columns the grouped GEMM is as cheap
# pass B of a pooled top layer as a streaming GEMM on the layer's INPUT (gspn_mlp_bwd_data_pooltop): its (rows, cout) output is not read
POOLTOP_STREAM = os.environ.get("GSPN_POOLTOP_STREAM", "1") != "0"
# one launch for pass B + the dW reduction of the same layer (gspn_mlp_bwd_data_dw) instead of two
FUSE_DW = os.environ.get("GSPN_FUSE_DW", "1") != "0"
# pass A and pass B of a layer in one launch where the library has the kernel for the shape (gspn_mlp_bwd_fused: both products from one staged
# dY tile); the library's own switch is GSPN_BWD_FUSED
FUSED_BWD = os.environ.get("GSPN_FUSED_BWD", "1") != "0"
POOLTOP_FUSED = os.environ.get("GSPN_POOLTOP_FUSED", "1") != "0"
FUSED_COEF = os.environ.get("GSPN_FUSED_COEF", "1") != "0"          # r04: the coefficient kernel after a fused launch carries that launch's dW reduction       # the pooled (nsample = 32) top layer through the fused launch as well
# the top layer of a stack with a dense upstream gradient takes its BN reductions in a streaming pre-pass (gspn_dense_rsum) from this many rows on
DENSE_TOP_RSUM = os.environ.get("GSPN_DENSE_TOP_RSUM", "1") != "0"
DENSE_TOP_MIN_ROWS = int(os.environ.get("GSPN_DENSE_TOP_MIN_ROWS", "65536"))
_side_streams = {}

# Optional SyncBN: batch statistics over the global batch of all ranks instead of per replica -- what the single-GPU reference
# computes.  Off by default (standard data parallelism); when on, and torch.distributed has more than one rank, training-mode
# batch-normalised stacks run layer by layer: the GEMM kernel of the linear layer, then parallel.sync_bn_relu (two small all-reduces per
# layer).  Slower than the fused kernels; results equal a single-process run on the concatenated batch (tests/test_cpu_dist.py).
SYNC_BN = False
# r04: SyncBN ON the fused kernels.  Every batch-norm reduction of the stack already leaves the kernels as per-workgroup partial rows that a
# small kernel sums (gspn_bn_finalize*, gspn_mlp_bwd_coef): with equal shards the ranks' partial buffers have the same shape, so ONE
# all-reduce(SUM) of that buffer between the producing kernel and the summing kernel turns every local sum into the global one; the
# summing kernel then runs with rows x world.  Same kernels, same hipGraph, two small collectives per layer.  Needs: equal rows on every
# rank (weak scaling), training-mode BN on every layer, coefficients that come from partial rows everywhere (EARLY_R; the dense top
# layer's reductions always through gspn_dense_rsum).  SYNC_BN_FUSED = False restores round 2's layer-by-layer torch form.
SYNC_BN_FUSED = os.environ.get("GSPN_SYNC_BN_FUSED", "1") != "0"


def _sync_world(stack_ok=True):
    """world size over which the fused kernels' BN reductions are all-reduced (1: no SyncBN, or a single rank)"""
    if not (SYNC_BN and SYNC_BN_FUSED and stack_ok):
        return 1
    import torch.distributed as dist
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def _allreduce_sum(t):
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM)

# bench.py sets this to a list to collect (kind, rows, cin, cout, start_event, end_event) around every GEMM-kernel launch of the stack
# ("fwd", "wgrad" = pass A incl. its finalize kernels, "bwd" = pass B (+ the dW reduction riding in it)); None = no events
PROFILE = None


def _tic():
    if PROFILE is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _toc(e0, kind, rows, cin, cout, executed=None):
    """executed: multiply-add flops the launches between the two events really issue when that differs from the algorithmic
    2*rows*cin*cout of the layer (pre-aggregated first layers run their feature part on the source rows; the two-product pass A
    runs two GEMMs; pass B of a first layer may cover only the columns that carry a gradient)"""
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        PROFILE.append((kind, rows, cin, cout, e0, e1, 2.0 * rows * cin * cout if executed is None else float(executed)))


def _side_stream(dev):
    key = (dev.type, dev.index)
    s = _side_streams.get(key)
    if s is None:
        s = _side_streams[key] = torch.cuda.Stream(device=dev)
    return s


class LayerParams:
    """Variables of one conv2d scope (tf_util.py:159-183): weights (cin,cout) [= the (1,1,cin,cout) kernel],
    biases, and under bn: beta, gamma, moving_mean, moving_variance."""

    def __init__(self, weights, biases, bn=True, beta=None, gamma=None, moving_mean=None, moving_variance=None):
        self.weights, self.biases, self.bn = weights, biases, bn
        self.beta, self.gamma, self.moving_mean, self.moving_variance = beta, gamma, moving_mean, moving_variance

    def tensors(self):
        t = [self.weights, self.biases]
        if self.bn:
            t += [self.beta, self.gamma]
        return t


# ---- gradient sinks (r06) ---------------------------------------------------------------------------------------------------------
# A caller that keeps all parameter gradients in ONE flat buffer (parallel.FlatGradBucket: the operand of the gradient all-reduce and of the
# one-launch Adam) can register, per parameter, the slice of that buffer its gradient belongs in (FlatGradBucket.attach_sinks()).  The backward
# pass of a stack then lets its reduction kernels write dW / dbias / dbeta / dgamma STRAIGHT into those slices and returns None to autograd for
# them: no per-step `cat` of ~60 gradient tensors into the bucket (the last library kernel inside the captured step but autograd's two gradient
# accumulations).  Keyed by the parameter's data pointer (LayerParams.weights is a fresh VIEW of the stored kernel on every forward); an entry
# holds a weak reference to the stored parameter and dies with it.  A parameter used by two stacks in one backward pass is written once, the
# second gradient goes through autograd and FlatGradBucket.flatten() adds it.
GRAD_SINKS = {}


class _Sink:
    __slots__ = ("bucket", "index", "view", "ref")

    def __init__(self, bucket, index, view, ref):
        self.bucket, self.index, self.view, self.ref = bucket, index, view, ref


def _grad_buffer(param, sunk):
    """where the gradient of `param` is to be written: its registered bucket slice (first gradient of this backward pass) or a fresh tensor"""
    e = GRAD_SINKS.get(param.data_ptr())
    if e is not None:
        base = e.ref()
        if base is None or base.data_ptr() != param.data_ptr() or base.numel() != param.numel():
            GRAD_SINKS.pop(param.data_ptr(), None)                 # the parameter is gone (or moved): never write through a stale entry
        elif e.index not in e.bucket._written and param.dtype == torch.float32 and param.is_contiguous():
            e.bucket._written.add(e.index)
            v = e.view.view(param.shape)
            sunk.add(v.data_ptr())
            return v
    return torch.empty_like(param)


def _zeros(n, dev, dtype=torch.float32):
    return torch.zeros(n, dtype=dtype, device=dev)


class _MlpStack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cin0, spec, *params):
        """x: (rows, ld) float32 contiguous; spec: dict(layers=[LayerParams], is_training, decay, pool_ns);
        params: flat list of the differentiable tensors (w, b[, beta, gamma]) per layer, for autograd."""
        lib = L.lib()
        dev = x.device
        gather = spec.get("gather")              # fused SA front end: x is the (b*n, ldf) feature matrix, the GEMM rows are virtual
        if gather is not None:
            rows, ld = gather["rows"], 0
            ga = L.GatherArgs(x.data_ptr(), x.shape[1], gather["c"], gather["gidx"].data_ptr(), gather["rel"].data_ptr(), int(gather["xyz_first"]))
            ctx.gargs = ga
            ctx.gather_x = x                     # the backward pass gathers from it again: keep the storage alive
        else:
            rows, ld = x.shape
        pre = spec.get("preagg")                 # pre-aggregated first layer: x is the (source rows, ldf) feature matrix
        if pre is not None:
            rows, ld = pre["rows"], 0
            ctx.pre_x = x
        layers = spec["layers"]
        is_training = bool(spec["is_training"])
        decay = float(spec["decay"])
        pool_ns = spec["pool_ns"]
        sw = _sync_world(is_training) if spec.get("sync_bn") else 1          # > 1: BN statistics over the global batch (fused SyncBN)
        saved = []
        cur, cur_ld, cin = x, ld, cin0
        in_scale = in_shift = None
        pool = None                              # (vmax, amax) of the last layer when the pool rides in its forward epilogue
        with torch.cuda.device(dev):
            st = L.stream()                      # the current stream of x's device (not of whatever device was current outside)
            for li, lp in enumerate(layers):
                cout = lp.weights.shape[1]
                y = torch.empty((rows, cout), dtype=torch.float32, device=dev)
                use_stats = lp.bn and is_training
                pre0 = pre if li == 0 else None
                if pre0 is not None:
                    nparts0 = int(lib.gspn_preagg_fwd_parts(rows, cout))
                    stats = torch.empty(nparts0 * 2 * cout, dtype=torch.float32, device=dev) if use_stats else None
                else:
                    stats = torch.empty(int(lib.gspn_mlp_fwd_stats_bytes(rows, cout)) // 4, dtype=torch.float32, device=dev) if use_stats else None
                ev = _tic()
                if (FUSE_POOL32 and pool_ns and (pool_ns == 32 or (FUSE_POOLN and pool_ns % 32 == 0 and lp.weights.shape[1] % 4 == 0)) and li == len(layers) - 1 and rows % 32 == 0
                        and not (li == 0 and (pre is not None or gather is not None))):      # (a one-layer stack with a gathered / pre-aggregated input has no plain forward)
                    g32 = rows // 32
                    pool = (torch.empty((g32, cout), dtype=torch.float32, device=dev), torch.empty((g32, cout), dtype=torch.int32, device=dev))
                    try:
                        L.check(lib.gspn_mlp_fwd_pool32(rows, cin, cout, L.ptr(cur), cur_ld, L.ptr(in_scale), L.ptr(in_shift), L.ptr(lp.weights),
                                                        L.ptr(lp.biases), L.ptr(y), cout, L.ptr(stats), L.ptr(pool[0]), L.ptr(pool[1]), st),
                                "mlp_fwd_pool32")
                    except NotImplementedError:          # a shape / alignment without a pool epilogue: plain forward, the stand-alone pool below
                        pool = None
                        L.check(lib.gspn_mlp_fwd(rows, cin, cout, L.ptr(cur), cur_ld, L.ptr(in_scale), L.ptr(in_shift),
                                                 L.ptr(lp.weights), L.ptr(lp.biases), L.ptr(y), cout, L.ptr(stats), st), "mlp_fwd")
                elif li == 0 and pre is not None:
                    # F = feat . W_feat on the source rows, then the output rows from F (+ side . W_side + bias) and their column sums
                    wf = lp.weights[pre["wf0"]:pre["wf0"] + pre["c"]]
                    ws = lp.weights[pre["ws0"]:pre["ws0"] + pre["side_n"]]
                    fsrc = torch.empty((x.shape[0], cout), dtype=torch.float32, device=dev)
                    L.check(lib.gspn_mlp_fwd(x.shape[0], pre["c"], cout, L.ptr(x), x.shape[1], None, None, L.ptr(wf), None, L.ptr(fsrc), cout, None, st),
                            "mlp_fwd(pre-aggregation)")
                    L.check(lib.gspn_preagg_fwd(rows, cout, pre["T"], L.ptr(fsrc), L.ptr(pre["idx"]), L.ptr(pre["w"]), pre["per_scene_rows"],
                                                pre["per_scene_src"], L.ptr(pre["side"]), pre["side_ld"], pre["side_n"], L.ptr(ws), L.ptr(lp.biases),
                                                L.ptr(y), L.ptr(stats), st), "preagg_fwd")
                elif li == 0 and gather is not None:
                    L.check(lib.gspn_mlp_fwd_gather(rows, ctypes.byref(ga), cout, L.ptr(lp.weights), L.ptr(lp.biases), L.ptr(y), cout, L.ptr(stats), st),
                            "mlp_fwd_gather")
                else:
                    L.check(lib.gspn_mlp_fwd(rows, cin, cout, L.ptr(cur), cur_ld, L.ptr(in_scale), L.ptr(in_shift),
                                             L.ptr(lp.weights), L.ptr(lp.biases), L.ptr(y), cout, L.ptr(stats), st), "mlp_fwd")
                _toc(ev, "fwd", rows, cin, cout,
                     2.0 * cout * (x.shape[0] * pre["c"] + rows * pre["side_n"]) if (li == 0 and pre is not None) else None)
                mean = torch.empty(cout, dtype=torch.float32, device=dev)
                var = torch.empty(cout, dtype=torch.float32, device=dev)
                scale = torch.empty(cout, dtype=torch.float32, device=dev)
                shift = torch.empty(cout, dtype=torch.float32, device=dev)
                if sw > 1 and use_stats:
                    _allreduce_sum(stats)                # every partial row becomes the sum of the ranks' rows: global column sums
                rows_bn = rows * sw
                if lp.bn and pre0 is not None:
                    L.check(lib.gspn_bn_finalize_parts(rows_bn, cout, L.ptr(stats), nparts0, L.ptr(lp.gamma), L.ptr(lp.beta), BN_EPS, decay,
                                                       int(is_training), L.ptr(lp.moving_mean), L.ptr(lp.moving_variance),
                                                       L.ptr(mean), L.ptr(var), L.ptr(scale), L.ptr(shift), st), "bn_finalize")
                elif lp.bn and sw > 1:
                    # (gspn_bn_finalize derives the number of partial rows from `rows`: the local count; the statistics divide by the global one)
                    L.check(lib.gspn_bn_finalize_parts(rows_bn, cout, L.ptr(stats), int(lib.gspn_mlp_fwd_stats_bytes(rows, cout)) // (8 * cout), L.ptr(lp.gamma),
                                                       L.ptr(lp.beta), BN_EPS, decay, int(is_training), L.ptr(lp.moving_mean), L.ptr(lp.moving_variance),
                                                       L.ptr(mean), L.ptr(var), L.ptr(scale), L.ptr(shift), st), "bn_finalize(sync)")
                elif lp.bn:
                    L.check(lib.gspn_bn_finalize(rows, cout, L.ptr(stats), L.ptr(lp.gamma), L.ptr(lp.beta), BN_EPS, decay,
                                                 int(is_training), L.ptr(lp.moving_mean), L.ptr(lp.moving_variance),
                                                 L.ptr(mean), L.ptr(var), L.ptr(scale), L.ptr(shift), st), "bn_finalize")
                else:
                    scale.fill_(1.0)
                    shift.zero_()
                    mean.zero_()
                    var.fill_(1.0)
                saved.append((None if (li == 0 and (gather is not None or pre is not None)) else cur, cur_ld, cin, in_scale, in_shift, y, mean, var, scale, shift))
                cur, cur_ld, cin, in_scale, in_shift = y, cout, cout, scale, shift
            cl = cin
            arg = None
            if pool_ns:
                groups = rows // pool_ns
                out = torch.empty((groups, cl), dtype=torch.float32, device=dev)
                arg = torch.empty((groups, cl), dtype=torch.int32, device=dev)
                done = False
                if pool is not None and pool_ns == 32:             # the group extrema came out of the last forward launch: finish on (groups, c) elements
                    try:
                        L.check(lib.gspn_pool32_select(groups, cl, L.ptr(pool[0]), L.ptr(pool[1]), L.ptr(cur), cur_ld,
                                                       L.ptr(in_scale), L.ptr(in_shift), L.ptr(out), L.ptr(arg), st), "pool32_select")
                        done = True
                    except NotImplementedError:
                        pool = None
                elif pool is not None:           # groups of 32 * sub rows: the first largest of the tile maxima
                    yarg = torch.empty((groups, cl), dtype=torch.float32, device=dev)
                    try:
                        L.check(lib.gspn_pool32_select_groups(groups, pool_ns // 32, cl, L.ptr(pool[0]), L.ptr(pool[1]), L.ptr(cur), cur_ld,
                                                              L.ptr(in_scale), L.ptr(in_shift), L.ptr(out), L.ptr(arg), L.ptr(yarg), st), "pool32_select_groups")
                        pool = (yarg, None)
                        done = True
                    except NotImplementedError:  # (the library declined the shape -- the header's contract is "the caller then runs gspn_bnrelu_maxpool")
                        pool = None
                if not done:
                    L.check(lib.gspn_bnrelu_maxpool(groups, pool_ns, cl, L.ptr(cur), cur_ld, L.ptr(in_scale), L.ptr(in_shift),
                                                    L.ptr(out), L.ptr(arg), st), "bnrelu_maxpool")
            else:
                out = torch.empty((rows, cl), dtype=torch.float32, device=dev)
                L.check(lib.gspn_bnrelu_apply(rows, cl, L.ptr(cur), cur_ld, L.ptr(in_scale), L.ptr(in_shift), L.ptr(out), cl, st), "bnrelu_apply")
        ctx.gather = gather
        ctx.pre = pre
        ctx.saved = saved
        # backward re-reads the weights and gamma: an in-place update between forward and backward would silently change the result
        # (autograd's own check only covers save_for_backward tensors; these are kept as attributes so the layer list stays one object)
        ctx.versions = [(lp.weights._version, lp.gamma._version if lp.bn else 0) for lp in layers]
        ctx.arg = arg
        ctx.pool_yarg = pool[0] if pool is not None else None
        if pool_ns:
            ctx.save_for_backward(out)           # the pooled output: where it is 0 no gradient passes the ReLU (POOLTOP_STREAM)
        ctx.spec = spec
        ctx.sw = sw
        ctx.rows = rows
        ctx.x_needs_grad = x.requires_grad
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = L.lib()
        spec = ctx.spec
        layers = spec["layers"]
        for lp, (vw, vg) in zip(layers, ctx.versions):
            if lp.weights._version != vw or (lp.bn and lp.gamma._version != vg):
                raise RuntimeError("mlp_stack: a weight or gamma tensor was modified in place between forward and backward")
        is_training = bool(spec["is_training"])
        pool_ns = spec["pool_ns"]
        rows = ctx.rows
        sw = ctx.sw
        gather = ctx.gather
        d_out = d_out.contiguous()
        dev = d_out.device
        grads = []
        sunk = set()                             # data pointers of gradient buffers that ARE bucket slices (returned to autograd as None)
        dz = None if pool_ns else d_out          # dense upstream gradient of the current layer
        ldz = d_out.shape[1]
        dx0 = None
        side = main = None
        keep = []
        with torch.cuda.device(dev):
            st = L.stream()
            tr_all = is_training and EARLY_R and not DEFER_DW
            coef = {}                            # layer index -> (cA, cB, cC, dgamma, dbeta, dbias) taken BEFORE that layer's pass A
            for li in range(len(layers) - 1, -1, -1):
                lp = layers[li]
                (xin, xld, cin, in_scale, in_shift, y, mean, var, scale, shift) = ctx.saved[li]
                cout = lp.weights.shape[1]
                a = L.DyArgs()
                a.Y, a.ldy = y.data_ptr(), cout
                if dz is None:
                    a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = None, 0, d_out.data_ptr(), ctx.arg.data_ptr(), pool_ns
                else:
                    a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = dz.data_ptr(), ldz, None, None, 0
                a.scale, a.shift = scale.data_ptr(), shift.data_ptr()
                gather0 = gather if li == 0 else None
                # ---- early coefficients of the top layer of a pooled stack: (groups x c) work on (dPool, arg, Y) ----
                if tr_all and lp.bn and dz is None and li not in coef:
                    part = torch.empty(int(lib.gspn_rsum_part_floats(rows, cout)), dtype=torch.float32, device=dev)
                    npart = ctypes.c_int(0)
                    yarg = ctx.pool_yarg           # y at the arg row, left by gspn_pool32_select (None: gather it from Y)
                    L.check(lib.gspn_pool_rsum(rows // pool_ns, pool_ns, cout, L.ptr(d_out), L.ptr(ctx.arg), L.ptr(y if yarg is None else yarg),
                                               cout if yarg is None else 0, L.ptr(scale), L.ptr(shift),
                                               L.ptr(mean), L.ptr(var), BN_EPS, L.ptr(part), ctypes.byref(npart), st), "pool_rsum")
                    coef[li] = _coef_from_parts(lib, rows, cout, npart.value, part, mean, var, lp, dev, st, sw, sunk)
                # ---- early coefficients of the top layer of a DENSE stack: one streaming pass over (d_out, Y); worth its launch on the long
                #      layers only (the two-product pass A of a short layer costs less than the extra dependent kernels) ----
                if (tr_all and (DENSE_TOP_RSUM or sw > 1) and lp.bn and dz is not None and li == len(layers) - 1 and li not in coef
                        and ((li > 0 and rows >= DENSE_TOP_MIN_ROWS) or (sw > 1 and gather0 is None and not (li == 0 and ctx.pre is not None)))):
                    part = torch.empty(int(lib.gspn_rsum_part_floats(rows, cout)), dtype=torch.float32, device=dev)
                    npart = ctypes.c_int(0)
                    try:
                        L.check(lib.gspn_dense_rsum(rows, cout, L.ptr(dz), ldz, L.ptr(y), cout, L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(var),
                                                    BN_EPS, L.ptr(part), ctypes.byref(npart), st), "dense_rsum")
                        coef[li] = _coef_from_parts(lib, rows, cout, npart.value, part, mean, var, lp, dev, st, sw, sunk)
                    except NotImplementedError:
                        pass
                known = coef.get(li)
                if sw > 1 and lp.bn and known is None:
                    raise RuntimeError("mlp_stack (fused SyncBN): layer %d has no early BN coefficients -- its reductions would be per replica" % li)
                if known is not None:
                    cA, cB, cC, dgamma, dbeta, dbias = known
                else:
                    cA = torch.empty(cout, dtype=torch.float32, device=dev)
                    cB = torch.empty(cout, dtype=torch.float32, device=dev)
                    cC = torch.empty(cout, dtype=torch.float32, device=dev)
                    dgamma = _grad_buffer(lp.gamma, sunk) if lp.bn else None
                    dbeta = _grad_buffer(lp.beta, sunk) if lp.bn else None
                    dbias = _grad_buffer(lp.biases, sunk)
                a.cA, a.cB, a.cC = cA.data_ptr(), cB.data_ptr(), cC.data_ptr()
                if li == 0 and ctx.pre is not None:
                    if known is None:
                        raise RuntimeError("mlp_stack(preagg=): the first layer's BN coefficients must be known before its backward (EARLY_R)")
                    ev = _tic()
                    dW, dx0 = _preagg_backward(lib, ctx.pre, ctx.pre_x, lp, a, rows, cout, ctx.x_needs_grad, dev, st, sunk)
                    # (the whole backward of the layer, both "passes": dW_feat and d(feat) on the source rows, dW_side on the output rows)
                    _toc(ev, "bwd", rows, cin, cout, 2.0 * cout * (ctx.pre_x.shape[0] * ctx.pre["c"] * (2 if ctx.x_needs_grad else 1) + rows * ctx.pre["side_n"]))
                    g = [dW, dbias]
                    if lp.bn:
                        g += [dbeta, dgamma]
                    grads = g + grads
                    del a
                    continue
                dW = _grad_buffer(lp.weights, sunk)
                wcin = int(lib.gspn_mlp_gather_cin(ctypes.byref(ctx.gargs))) if gather0 is not None else cin
                work = torch.empty(int(lib.gspn_mlp_bwd_work_bytes(rows, wcin, cout)) // 4 + 4, dtype=torch.float32, device=dev)
                has_dx = li > 0 or ctx.x_needs_grad
                # ---- both passes in one launch (known coefficients, dense dz, an inner layer of a shape the fused kernel takes) ----
                if (FUSED_BWD and known is not None and (dz is not None or (pool_ns == 32 and POOLTOP_FUSED)) and li > 0 and not DEFER_DW
                        and int(lib.gspn_mlp_bwd_fused_work_bytes(rows, cin, cout)) > 0):
                    prev = layers[li - 1]
                    want_rsum = tr_all and prev.bn
                    (_, _, _, _, _, pY, pmean, pvar, pscale, pshift) = ctx.saved[li - 1]
                    dx = torch.empty((rows, cin), dtype=torch.float32, device=dev)
                    part = torch.empty(int(lib.gspn_rsum_part_floats(rows, cin)), dtype=torch.float32, device=dev) if want_rsum else None
                    npart = ctypes.c_int(0)
                    ev = _tic()
                    merged = None
                    try:
                        if want_rsum and sw == 1 and FUSED_COEF:
                            # the previous layer's coefficient kernel and this layer's dW reduction depend on the fused launch only: one launch
                            pc = [torch.empty(cin, dtype=torch.float32, device=dev) for _ in range(3)]
                            pg = [_grad_buffer(prev.gamma, sunk), _grad_buffer(prev.beta, sunk), _grad_buffer(prev.biases, sunk)]
                            L.check(lib.gspn_mlp_bwd_fused_coef(rows, cin, cout, ctypes.byref(a), L.ptr(lp.weights), L.ptr(xin), xld, L.ptr(in_scale),
                                                                L.ptr(in_shift), L.ptr(dx), cin, L.ptr(work), L.ptr(dW), L.ptr(pmean), L.ptr(pvar), BN_EPS,
                                                                L.ptr(part), ctypes.byref(npart), L.ptr(prev.gamma), L.ptr(pc[0]), L.ptr(pc[1]), L.ptr(pc[2]),
                                                                L.ptr(pg[0]), L.ptr(pg[1]), L.ptr(pg[2]), st), "mlp_bwd_fused_coef")
                            merged = (pc[0], pc[1], pc[2], pg[0], pg[1], pg[2])
                        else:
                            L.check(lib.gspn_mlp_bwd_fused(rows, cin, cout, ctypes.byref(a), L.ptr(lp.weights), L.ptr(xin), xld, L.ptr(in_scale), L.ptr(in_shift),
                                                           L.ptr(dx), cin, L.ptr(work), L.ptr(dW), L.ptr(pmean), L.ptr(pvar), BN_EPS, L.ptr(part),
                                                           ctypes.byref(npart), st), "mlp_bwd_fused")
                        fused = True
                    except NotImplementedError:
                        fused = False
                    if fused:
                        _toc(ev, "fused", rows, cin, cout, 4.0 * rows * cin * cout)
                        if merged is not None:
                            coef[li - 1] = merged
                        elif want_rsum:
                            coef[li - 1] = _coef_from_parts(lib, rows, cin, npart.value, part, pmean, pvar, prev, dev, st, sw, sunk)
                        g = [dW, dbias]
                        if lp.bn:
                            g += [dbeta, dgamma]
                        grads = g + grads
                        dz, ldz = dx, cin
                        del a
                        continue
                fuse_dw = FUSE_DW and has_dx and not DEFER_DW and gather0 is None     # the dW reduction rides in spare workgroups of this layer's pass B
                # ---- pass A ----
                ev = _tic()
                ran_known = False
                if known is not None:
                    try:
                        L.check(lib.gspn_mlp_bwd_wgrad_known(rows, cin, cout, ctypes.byref(a), L.ptr(xin), xld, L.ptr(in_scale), L.ptr(in_shift),
                                                             ctypes.byref(ctx.gargs) if gather0 is not None else None, L.ptr(work),
                                                             None if fuse_dw else L.ptr(dW), st), "mlp_bwd_wgrad_known")
                        ran_known = True
                    except NotImplementedError:
                        ran_known = False            # a shape the one-GEMM kernels do not take: the two-product pass recomputes the same coefficients
                if not ran_known:
                    if gather0 is not None:
                        L.check(lib.gspn_mlp_bwd_wgrad_gather(rows, ctypes.byref(ctx.gargs), cout, ctypes.byref(a), L.ptr(mean), L.ptr(var),
                                                              L.ptr(lp.gamma if lp.bn else None), BN_EPS, int(lp.bn), int(is_training), L.ptr(work),
                                                              L.ptr(cA), L.ptr(cB), L.ptr(cC), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dbias), L.ptr(dW), st),
                                "mlp_bwd_wgrad_gather")
                    else:
                        L.check(lib.gspn_mlp_bwd_wgrad(rows, cin, cout, ctypes.byref(a), L.ptr(xin), xld, L.ptr(in_scale), L.ptr(in_shift),
                                                       L.ptr(mean), L.ptr(var), L.ptr(lp.gamma if lp.bn else None), BN_EPS, int(lp.bn), int(is_training),
                                                       L.ptr(work), L.ptr(cA), L.ptr(cB), L.ptr(cC), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dbias),
                                                       None if (DEFER_DW or fuse_dw) else L.ptr(dW), st), "mlp_bwd_wgrad")
                _toc(ev, "wgrad", rows, cin, cout, None if ran_known else 2 * 2.0 * rows * cin * cout)      # the two-product form: G1 and Gx
                if DEFER_DW and gather0 is None:
                    if side is None:
                        side = _side_stream(dev)
                        main = torch.cuda.current_stream()
                    side.wait_event(main.record_event())            # after this layer's wgrad + channel sums
                    L.check(lib.gspn_mlp_bwd_dw(rows, cin, cout, ctypes.byref(a), L.ptr(xin), xld, L.ptr(var), L.ptr(lp.gamma if lp.bn else None),
                                                BN_EPS, int(lp.bn), int(is_training), L.ptr(work), L.ptr(dW),
                                                ctypes.c_void_p(side.cuda_stream)), "mlp_bwd_dw")
                    keep.append((work, a))                          # alive until the side stream has joined below
                g = [dW, dbias]
                if lp.bn:
                    g += [dbeta, dgamma]
                grads = g + grads
                # ---- pass B ----
                if has_dx and gather0 is not None:
                    # the gathered layer: dX of the feature columns in grouped-row layout, then the gather-form gradient of the grouping
                    # (inverse lists; atomics without them) straight into d(features)
                    gb, gn, gm, gns = gather0["dims"]
                    gc, xf = gather0["c"], int(gather0["xyz_first"])
                    ldp = (3 + gc + 3) // 4 * 4
                    dxg = torch.empty((rows, ldp), dtype=torch.float32, device=dev)
                    ev = _tic()
                    L.check(lib.gspn_mlp_bwd_data_cols(rows, cin, cout, ctypes.byref(a), L.ptr(lp.weights), 3 if xf else 0, gc, L.ptr(dxg), ldp, st),
                            "mlp_bwd_data_cols")
                    _toc(ev, "bwd", rows, cin, cout, 2.0 * rows * gc * cout)
                    gp = torch.empty((gb, gn, gc), dtype=torch.float32, device=dev)
                    if gather0.get("order") is not None:
                        L.check(lib.gspn_sa_group_concat_grad_csr(gb, gn, gc, gm, gns, L.ptr(gather0["order"]), L.ptr(gather0["offsets"]), xf, ldp,
                                                                  L.ptr(dxg), L.ptr(gp), st), "sa_group_concat_grad_csr")
                    else:
                        L.check(lib.gspn_sa_group_concat_grad(gb, gn, gc, gm, gns, L.ptr(gather0["idx"]), xf, ldp, L.ptr(dxg), L.ptr(gp), st),
                                "sa_group_concat_grad")
                    ldf = ctx.gargs.ldf
                    dx0 = gp.view(gb * gn, gc)
                    if ldf != gc:
                        dx0 = torch.nn.functional.pad(dx0, (0, ldf - gc))
                elif has_dx:
                    dx = torch.empty((rows, xld), dtype=torch.float32, device=dev) if li == 0 else torch.empty((rows, cin), dtype=torch.float32, device=dev)
                    if li == 0 and xld > cin and spec.get("grad_cols") is None:
                        dx.zero_()
                    # only grad_cols of the input feed a gradient upstream (e.g. not the xyz columns of an SA input)
                    gc = (spec.get("grad_cols") if li == 0 else None) or (0, cin)
                    # this dX is the dz of layer li-1: its epilogue can take that layer's BN reductions (early coefficients for its pass A)
                    prev = layers[li - 1] if li > 0 else None
                    want_rsum = tr_all and prev is not None and prev.bn
                    ev = _tic()
                    if want_rsum or fuse_dw:
                        part = npart = None
                        pY = pmean = pvar = pscale = pshift = None
                        if want_rsum:
                            (_, _, _, _, _, pY, pmean, pvar, pscale, pshift) = ctx.saved[li - 1]
                            part = torch.empty(int(lib.gspn_rsum_part_floats(rows, cin)), dtype=torch.float32, device=dev)
                            npart = ctypes.c_int(0)
                        bn_dw = 0 if ran_known else int(lp.bn)           # known coefficients: the partial tiles are dW's own (plain sum)
                        top_done = False
                        if (POOLTOP_STREAM and dz is None and pool_ns == 32 and want_rsum and fuse_dw and known is not None and cin <= 64
                                and cin % 4 == 0 and cout <= 128 and cout % 4 == 0 and tuple(gc) == (0, cin)):
                            scr = torch.empty(int(lib.gspn_pooltop_scratch_floats(rows, cin, cout)), dtype=torch.float32, device=dev)
                            try:
                                L.check(lib.gspn_mlp_bwd_data_pooltop(rows, cin, cout, ctypes.byref(a), L.ptr(lp.weights), L.ptr(lp.biases),
                                                                      L.ptr(ctx.saved_tensors[0]), L.ptr(scr), L.ptr(dx), dx.shape[1],
                                                                      L.ptr(xin), xld, L.ptr(var), L.ptr(lp.gamma if lp.bn else None), BN_EPS, bn_dw,
                                                                      int(is_training), L.ptr(work), L.ptr(dW),
                                                                      L.ptr(pY), cin, L.ptr(pscale), L.ptr(pshift), L.ptr(pmean), L.ptr(pvar), BN_EPS,
                                                                      L.ptr(part), ctypes.byref(npart), st), "mlp_bwd_data_pooltop")
                                top_done = True
                            except NotImplementedError:
                                top_done = False
                        if not top_done:
                          L.check(lib.gspn_mlp_bwd_data_ex(rows, cin, cout, ctypes.byref(a), L.ptr(lp.weights), int(gc[0]), int(gc[1]), L.ptr(dx), dx.shape[1],
                                                         L.ptr(xin), xld, L.ptr(var), L.ptr(lp.gamma if lp.bn else None), BN_EPS, bn_dw,
                                                         int(is_training), L.ptr(work) if fuse_dw else None, L.ptr(dW) if fuse_dw else None,
                                                         L.ptr(pY), cin, L.ptr(pscale), L.ptr(pshift), L.ptr(pmean), L.ptr(pvar), BN_EPS, L.ptr(part),
                                                         ctypes.byref(npart) if want_rsum else None, st), "mlp_bwd_data_ex")
                        if want_rsum:
                            coef[li - 1] = _coef_from_parts(lib, rows, cin, npart.value, part, pmean, pvar, prev, dev, st, sw, sunk)
                    else:
                        L.check(lib.gspn_mlp_bwd_data_cols(rows, cin, cout, ctypes.byref(a), L.ptr(lp.weights), int(gc[0]), int(gc[1]), L.ptr(dx),
                                                           dx.shape[1], st), "mlp_bwd_data_cols")
                    _toc(ev, "bwd", rows, cin, cout, 2.0 * rows * int(gc[1]) * cout)
                    if li == 0:
                        dx0 = dx
                    dz, ldz = dx, dx.shape[1]
                # keep the buffers referenced by `a` alive until the kernels are enqueued (same stream: ordered)
                del a
            if side is not None:
                main.wait_event(side.record_event())                # join: the gradients (and the workspaces) are complete past here
        del keep
        # gradients that were written straight into their bucket slices are not handed to autograd (it would re-assign / clone them)
        return (dx0, None, None) + tuple(None if (g is not None and g.data_ptr() in sunk) else g for g in grads)


_consts = {}


def _const_vectors(dev, c):
    """(ones(c), zeros(c)) on dev, made once: read-only operands of the no-BN GEMM calls"""
    key = (dev.type, dev.index, c)
    v = _consts.get(key)
    if v is None:
        v = _consts[key] = (torch.ones(c, dtype=torch.float32, device=dev), torch.zeros(c, dtype=torch.float32, device=dev))
    return v


def _preagg_backward(lib, pre, x, lp, a, rows, cout, need_dx, dev, st, sunk=None):
    """backward of a pre-aggregated first layer: dY written once (+ dW_side), its transpose-gather G onto the source rows, then
    dW_feat = feat^T . G and d(feat) = G . W_feat^T as small GEMMs without BN (always-open mask: scale 0, shift 1, dY = dz)"""
    c, side_n = pre["c"], pre["side_n"]
    dW = _grad_buffer(lp.weights, sunk if sunk is not None else set())
    if pre["c"] + side_n < lp.weights.shape[0]:
        dW.zero_()
    dy = torch.empty((rows, cout), dtype=torch.float32, device=dev)
    part = torch.empty(int(lib.gspn_preagg_part_floats(cout, max(side_n, 1))), dtype=torch.float32, device=dev)
    dws = dW[pre["ws0"]:pre["ws0"] + side_n]
    ride = need_dx and FUSE_DW               # the reductions of dW_side's and dW_feat's partial tiles ride in the d(feat) launch
    nsl = ctypes.c_int(0)
    L.check(lib.gspn_preagg_bwd_dy(rows, cout, ctypes.byref(a), L.ptr(pre["side"]), pre["side_ld"], side_n, L.ptr(dy), L.ptr(part),
                                   None if (ride and side_n) else L.ptr(dws), ctypes.byref(nsl), st), "preagg_bwd_dy")
    nsrc = x.shape[0]
    gsrc = pre["scatter"](dy, cout)                                 # (source rows, cout): sum over the output rows each source row fed
    one, zero = _const_vectors(dev, cout)
    a2 = L.DyArgs()
    a2.Y, a2.ldy = gsrc.data_ptr(), cout
    a2.dZ, a2.ldz, a2.dPool, a2.pool_arg, a2.ns = gsrc.data_ptr(), cout, None, None, 0
    a2.scale, a2.shift = zero.data_ptr(), one.data_ptr()           # always-open mask
    a2.cA, a2.cB, a2.cC = one.data_ptr(), zero.data_ptr(), zero.data_ptr()      # known coefficients: dY = dz
    dwf = dW[pre["wf0"]:pre["wf0"] + c]
    wf = lp.weights[pre["wf0"]:pre["wf0"] + c]
    work = torch.empty(int(lib.gspn_mlp_bwd_work_bytes(nsrc, c, cout)) // 4 + 4, dtype=torch.float32, device=dev)
    L.check(lib.gspn_mlp_bwd_wgrad_known(nsrc, c, cout, ctypes.byref(a2), L.ptr(x), x.shape[1], None, None, None, L.ptr(work),
                                         None if ride else L.ptr(dwf), st), "mlp_bwd_wgrad_known(pre-aggregation)")
    dx = None
    if need_dx:
        dx = torch.empty((nsrc, x.shape[1]), dtype=torch.float32, device=dev)
        if x.shape[1] > c:
            dx.zero_()
        if ride:
            L.check(lib.gspn_mlp_bwd_data_dw2(nsrc, c, cout, ctypes.byref(a2), L.ptr(wf), 0, c, L.ptr(dx), x.shape[1], L.ptr(x), x.shape[1], None, None,
                                              BN_EPS, 0, 0, L.ptr(work), L.ptr(dwf), L.ptr(part), side_n, nsl.value if side_n else 0, L.ptr(dws), st),
                    "mlp_bwd_data_dw2(pre-aggregation)")
        else:
            L.check(lib.gspn_mlp_bwd_data(nsrc, c, cout, ctypes.byref(a2), L.ptr(wf), L.ptr(dx), x.shape[1], st), "mlp_bwd_data(pre-aggregation)")
    return dW, dx


def _coef_from_parts(lib, rows, c, nparts, part, mean, var, lp, dev, st, sw=1, sunk=None):
    """gspn_mlp_bwd_coef: partial sums [nparts][2][c] of (dyh, dyh*xhat) -> the layer's final BN-backward coefficients and its
    dgamma / dbeta / dbias, before its pass A runs.  sw > 1 (fused SyncBN): the partial rows are all-reduced first, the coefficients are
    those of the global batch (rows x sw), and dgamma / dbeta / dbias -- global sums then -- are divided by sw so that the gradient
    bucket's SUM all-reduce restores them (an exact division for power-of-two worlds)."""
    cA = torch.empty(c, dtype=torch.float32, device=dev)
    cB = torch.empty(c, dtype=torch.float32, device=dev)
    cC = torch.empty(c, dtype=torch.float32, device=dev)
    if sunk is not None and sw == 1:
        dgamma, dbeta, dbias = _grad_buffer(lp.gamma, sunk), _grad_buffer(lp.beta, sunk), _grad_buffer(lp.biases, sunk)
        dgb = None
    else:
        dgb = torch.empty((3, c), dtype=torch.float32, device=dev)
        dgamma, dbeta, dbias = dgb[0], dgb[1], dgb[2]
    if sw > 1:
        _allreduce_sum(part[:int(nparts) * 2 * c])
    L.check(lib.gspn_mlp_bwd_coef(rows * sw, c, int(nparts), L.ptr(part), L.ptr(mean), L.ptr(var), L.ptr(lp.gamma), BN_EPS,
                                  L.ptr(cA), L.ptr(cB), L.ptr(cC), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dbias), st), "mlp_bwd_coef")
    if sw > 1:
        dgb.mul_(1.0 / sw)
    return cA, cB, cC, dgamma, dbeta, dbias


class _Linear(torch.autograd.Function):
    """y = x.W + b with no batch-norm and no activation (tf_util.conv1d(..., activation_fn=None), e.g. model_rpointnet.py:71-73,262-263)
    on the same kernels: the forward GEMM's raw output IS the result; backward runs the two passes with an always-open ReLU mask
    (scale 0, shift 1) and use_bn = 0, so dY = dz."""

    @staticmethod
    def forward(ctx, x, cin, w, b):
        lib = L.lib()
        rows, ld = x.shape
        cout = w.shape[1]
        y = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            L.check(lib.gspn_mlp_fwd(rows, cin, cout, L.ptr(x), ld, None, None, L.ptr(w), L.ptr(b), L.ptr(y), cout, None, L.stream()), "mlp_fwd")
        ctx.save_for_backward(x, w, y)
        ctx.cin = cin
        return y

    @staticmethod
    def backward(ctx, dz):
        lib = L.lib()
        x, w, y = ctx.saved_tensors
        cin = ctx.cin
        rows, ld = x.shape
        cout = w.shape[1]
        dz = dz.contiguous()
        dev = dz.device
        one = torch.ones(cout, dtype=torch.float32, device=dev)
        zero = torch.zeros(cout, dtype=torch.float32, device=dev)
        cA, cB, cC = torch.empty_like(one), torch.empty_like(one), torch.empty_like(one)
        a = L.DyArgs()
        a.Y, a.ldy = y.data_ptr(), cout
        a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = dz.data_ptr(), cout, None, None, 0
        a.scale, a.shift = zero.data_ptr(), one.data_ptr()          # 0*y + 1 > 0: every element passes
        a.cA, a.cB, a.cC = cA.data_ptr(), cB.data_ptr(), cC.data_ptr()
        dW = torch.empty_like(w)
        dbias = torch.empty(cout, dtype=torch.float32, device=dev)
        dx = None
        with torch.cuda.device(dev):
            work = torch.empty(int(lib.gspn_mlp_bwd_work_bytes(rows, cin, cout)) // 4 + 4, dtype=torch.float32, device=dev)
            L.check(lib.gspn_mlp_bwd_wgrad(rows, cin, cout, ctypes.byref(a), L.ptr(x), ld, None, None, None, None, None, BN_EPS, 0, 0,
                                           L.ptr(work), L.ptr(cA), L.ptr(cB), L.ptr(cC), None, None, L.ptr(dbias), L.ptr(dW), L.stream()), "mlp_bwd_wgrad")
            if ctx.needs_input_grad[0]:
                dx = torch.empty((rows, ld), dtype=torch.float32, device=dev)
                if ld > cin:
                    dx.zero_()
                L.check(lib.gspn_mlp_bwd_data(rows, cin, cout, ctypes.byref(a), L.ptr(w), L.ptr(dx), ld, L.stream()), "mlp_bwd_data")
        return dx, None, dW, dbias


def mlp_linear(x, cin, lp):
    """x (rows, ld >= cin) -> x[:, :cin].W + b for a LayerParams without batch-norm."""
    if lp.bn:
        raise ValueError("mlp_linear is the no-BN, no-activation layer")
    x = L.need(x, torch.float32, 2, "x")
    return _Linear.apply(x, cin, lp.weights, lp.biases)


def _mlp_stack_sync_bn(x, cin, layers, decay, pool_ns):
    """the SYNC_BN form of mlp_stack: per layer y = x.W + b on the MFMA kernel, then BN over the global batch + ReLU"""
    from .parallel import sync_bn_relu
    cur = x
    for lp in layers:
        y = _Linear.apply(cur, cin, lp.weights, lp.biases)
        cur = sync_bn_relu(y, lp.gamma, lp.beta, lp.moving_mean, lp.moving_variance, decay, BN_EPS) if lp.bn else torch.relu(y)
        cin = y.shape[1]
    if pool_ns:
        cur = cur.view(-1, pool_ns, cur.shape[1]).max(dim=1).values
    return cur


def preagg_ok(layers, is_training, c):
    """can the stack's first layer be pre-aggregated?  (training-mode BN on the first two layers -- the second layer's pass B hands the
    first its BN coefficients --, a kernel-friendly width, enough feature columns for the saved GEMM work to matter)"""
    return (PREAGG and EARLY_R and not DEFER_DW and not (SYNC_BN and not SYNC_BN_FUSED) and is_training and len(layers) >= 2 and layers[0].bn and layers[1].bn
            and c >= PREAGG_MIN_C and bool(L.lib().gspn_preagg_ok(layers[0].weights.shape[1])))


def mlp_stack(x, cin, layers, is_training, bn_decay, pool_ns=None, grad_cols=None, gather=None, preagg=None):
    """x: (rows, ld>=cin) float32 on a ROCm device.  Returns (rows/pool_ns, C_last) if pool_ns else (rows, C_last).
    grad_cols = (col0, ncols): the only columns of x whose gradient the caller will read (the rest of x.grad is left undefined).
    gather (fused SA front end): x is then the (b*n, ldf) FEATURE matrix and the stack's input rows are virtual --
    dict(rows, c, xyz_first, gidx, rel, dims=(b, n, m, ns), idx, order, offsets), see pointnet_util.pointnet_sa_module; cin = 3 + c.
    preagg (pre-aggregated first layer): x is the (source rows, ldf) feature matrix; dict(rows, c, T, idx, w, per_scene_rows,
    per_scene_src, side, side_ld, side_n, wf0, ws0 [first rows of W_feat / W_side inside the layer's weights], scatter [callable:
    (dY (rows, cout), cout) -> (source rows, cout), the transpose of the aggregation]) -- see pointnet_util.
    Raises NotImplementedError (before anything has run) when the first layer's shape is outside what the gathering kernels take."""
    if not layers:
        raise ValueError("mlp_stack needs at least one layer")
    x = L.need(x, torch.float32, 2, "x")
    if gather is not None:
        if x.shape[1] % 4 or gather["c"] > x.shape[1] or cin != 3 + gather["c"] or len(layers) < 2 or layers[0].weights.shape[1] % 4:
            raise NotImplementedError("mlp_stack(gather=): needs 16-byte feature rows, >= 2 layers and a first layer of 4k output channels")
        if SYNC_BN and not SYNC_BN_FUSED:
            raise NotImplementedError("mlp_stack(gather=) with the layer-by-layer SyncBN form")
    if preagg is not None:
        if gather is not None or not preagg_ok(layers, is_training, preagg["c"]) or cin != preagg["c"] + preagg["side_n"]:
            raise NotImplementedError("mlp_stack(preagg=): training-mode BN stacks of >= 2 layers, cin = c + side_n, no gather")
    if pool_ns and (preagg["rows"] if preagg is not None else (gather["rows"] if gather is not None else x.shape[0])) % pool_ns:
        raise ValueError("rows must be a multiple of pool_ns")
    if grad_cols is not None and not (0 <= grad_cols[0] and grad_cols[1] > 0 and grad_cols[0] + grad_cols[1] <= cin):
        raise ValueError("grad_cols must be a column range inside [0, cin)")
    sync_fused = False
    if SYNC_BN and is_training and any(lp.bn for lp in layers):
        import torch.distributed as dist
        if dist.is_initialized() and dist.get_world_size() > 1:
            # the fused kernels when every layer is batch-normalised and the early coefficients are on (see SYNC_BN_FUSED); else layer by layer
            sync_fused = SYNC_BN_FUSED and EARLY_R and not DEFER_DW and all(lp.bn for lp in layers)
            if not sync_fused:
                if gather is not None or preagg is not None:
                    raise NotImplementedError("mlp_stack(gather= / preagg=) with the layer-by-layer SyncBN form")
                return _mlp_stack_sync_bn(x, cin, layers, 0.9 if bn_decay is None else float(bn_decay), pool_ns)
    spec = {"layers": layers, "is_training": is_training, "decay": 0.9 if bn_decay is None else float(bn_decay), "pool_ns": pool_ns,
            "grad_cols": grad_cols, "gather": gather, "preagg": preagg, "sync_bn": sync_fused}
    flat = []
    for lp in layers:
        flat += lp.tensors()
    return _MlpStack.apply(x, cin, spec, *flat)
