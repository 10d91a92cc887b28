"""Data parallelism over scenes.  The reference is single-GPU; scenes are
independent in every kernel, so rank r simply owns scenes [r*B/W, (r+1)*B/W) and the only exchange
is ONE flat-bucket all-reduce of the MLP parameter gradients (~1 MB for the SA/FP stack: latency
bound, so a single bucket) over RCCL/xGMI via torch.distributed (backend 'nccl' == RCCL on ROCm;
'gloo' on CPU for tests).  Batch-norm statistics stay per replica (standard DP)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, force=False):
    """One process per GPU, launched by torch.distributed.run: RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the env.
    force=True initialises the process group even at world size 1 (a one-rank RCCL communicator: the collective path -- communicator
    set-up, the all-reduce launch, its ordering against the captured step and the side streams -- then runs on a single GPU)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 control flow on a box with ONE GPU (r03): GSPN_FORCE_DEVICE=0 GSPN_DIST_BACKEND=gloo puts every rank on
    # that device with gloo moving the bucket through the host -- slow, but the ranks run the real step, the real stream tests and the
    # real sequence of collectives (RCCL refuses two ranks on one device)
    backend = backend or os.environ.get("GSPN_DIST_BACKEND") or None
    if "GSPN_FORCE_DEVICE" in os.environ:
        local = int(os.environ["GSPN_FORCE_DEVICE"])
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:            # every rank would pick a port of its own and the rendezvous would hang until its timeout
                raise RuntimeError("init_from_env: WORLD_SIZE > 1 needs MASTER_PORT (torch.distributed.run sets it; a manual launch must)")
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        kw = {}
        if torch.cuda.is_available() and (backend or "nccl") == "nccl":
            kw["device_id"] = torch.device("cuda", local)      # bind the communicator to this rank's GPU at init (no lazy device guess)
        dist.init_process_group(backend=backend or ("nccl" if torch.cuda.is_available() else "gloo"), rank=rank, world_size=world, **kw)
    return rank, local, world


def shard_range(total, rank, world):
    """contiguous scene shard of rank `rank` (sizes differ by at most one)"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class _SyncBNReLU(torch.autograd.Function):
    """relu(batch_norm(y)) with the batch statistics taken over ALL ranks (the reference is single-GPU, so its tf.contrib.layers.batch_norm
    sees the global batch: tf_util.py:529-534).  Two tiny all-reduces per layer: (rows, sum y, sum y^2) forward, (sum dyh, sum dyh*xhat)
    backward, in double.  Same arithmetic as the fused kernels: biased variance, y*inv + (beta - mean*inv), moving = moving*decay +
    batch*(1-decay).  Pure torch, so it runs (and is tested) on CPU tensors under gloo as well."""

    @staticmethod
    def forward(ctx, y, gamma, beta, moving_mean, moving_var, decay, eps, relu):
        c = y.shape[1]
        st = torch.empty(2 * c + 1, dtype=torch.float64, device=y.device)
        yd = y.double()
        st[0] = y.shape[0]
        st[1:c + 1] = yd.sum(0)
        st[c + 1:] = (yd * yd).sum(0)
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(st, op=dist.ReduceOp.SUM)
        rows = st[0]
        mean = st[1:c + 1] / rows
        var = (st[c + 1:] / rows - mean * mean).clamp_min(0.0)
        with torch.no_grad():
            moving_mean.mul_(decay).add_(mean.to(moving_mean.dtype) * (1.0 - decay))
            moving_var.mul_(decay).add_(var.to(moving_var.dtype) * (1.0 - decay))
        rstd = torch.rsqrt(var + eps)
        inv = (rstd * gamma.double()).to(y.dtype)
        shift = (beta.double() - mean * rstd * gamma.double()).to(y.dtype)
        z = y * inv + shift
        if relu:
            z = torch.relu(z)
        ctx.save_for_backward(y, z, gamma, mean, rstd)
        ctx.rows, ctx.relu = float(rows), relu
        return z

    @staticmethod
    def backward(ctx, dz):
        y, z, gamma, mean, rstd = ctx.saved_tensors
        c = y.shape[1]
        dyh = (dz * (z > 0)) if ctx.relu else dz
        xhat = (y.double() - mean) * rstd
        r = torch.empty(2 * c, dtype=torch.float64, device=y.device)
        r[:c] = dyh.double().sum(0)
        r[c:] = (dyh.double() * xhat).sum(0)
        dbeta, dgamma = r[:c].to(gamma.dtype), r[c:].to(gamma.dtype)     # local sums: the bucket all-reduce adds the ranks' shares
        if dist.is_initialized() and dist.get_world_size() > 1:
            r = r.clone()
            dist.all_reduce(r, op=dist.ReduceOp.SUM)
        r0, r1 = r[:c] / ctx.rows, r[c:] / ctx.rows
        dy = (gamma.double() * rstd) * (dyh.double() - r0 - xhat * r1)
        return dy.to(y.dtype), dgamma, dbeta, None, None, None, None, None


def sync_bn_relu(y, gamma, beta, moving_mean, moving_var, decay=0.9, eps=1e-3, relu=True):
    """training-mode BN (+ReLU) of a (rows, c) matrix with statistics over the global batch of all ranks (optional SyncBN)"""
    return _SyncBNReLU.apply(y, gamma, beta, moving_mean, moving_var, float(decay), float(eps), bool(relu))


class FlatGradBucket:
    """Flat fp32 bucket over a fixed parameter list.  flatten(): one `cat` gathers the fresh gradients into the persistent bucket and
    every p.grad is re-pointed at its slice (views, no copies back); all_reduce(): ONE all_reduce(SUM) + scale averages the bucket
    over ranks.  Use with optimizer.zero_grad(set_to_none=True) so backward assigns gradients instead of accumulating into the views.
    The bucket's storage never changes, so a captured step (graph.py) can end in flatten() and be replayed."""

    def __init__(self, params):
        self.params = [p for p in params]
        self.sizes = [p.numel() for p in self.params]
        dev = self.params[0].device if self.params else None
        self.flat = torch.zeros(sum(self.sizes), dtype=torch.float32, device=dev)
        self._views = []
        off = 0
        for p, s in zip(self.params, self.sizes):
            self._views.append(self.flat[off:off + s].view_as(p))
            off += s
        self._written = set()        # indices of the parameters whose gradient a backward pass wrote straight into its slice since the last flatten()
        self.sinks = False

    def attach_sinks(self):
        """r06: register every parameter's slice with gspn_amd.mlp.GRAD_SINKS -- the shared-MLP backward then writes dW / dbias / dbeta / dgamma
        straight into the bucket and flatten() has nothing left to gather (no `cat` kernel in the step).  Call AFTER anything that moves the
        parameters' storage (FlatAdam re-attaches by itself).  Parameters whose gradient arrives through autograd as before (another op, a second
        use of the same layer in one backward pass) are still folded in by flatten()."""
        import weakref
        from . import mlp
        for k in [k for k, e in mlp.GRAD_SINKS.items() if e.bucket is self]:
            del mlp.GRAD_SINKS[k]
        for i, (p, v) in enumerate(zip(self.params, self._views)):
            if p.dtype == torch.float32 and p.is_contiguous() and p.numel() > 0:
                mlp.GRAD_SINKS[p.data_ptr()] = mlp._Sink(self, i, v.reshape(-1), weakref.ref(p))
        self.sinks = True
        return self

    def _flatten_sunk(self):
        """flatten() when (some) gradients already sit in their slices: only what autograd delivered separately is copied / added"""
        with torch.no_grad():
            for i, (p, v) in enumerate(zip(self.params, self._views)):
                g = p.grad
                fresh = g is not None and g.data_ptr() != v.data_ptr()
                if i in self._written:
                    if fresh:
                        v.add_(g)                   # a second gradient of a parameter whose first one was written in place
                elif fresh:
                    v.copy_(g)
                elif g is None:
                    v.zero_()                       # no gradient this step
                p.grad = v
        self._written.clear()
        return self.flat

    def flatten(self):
        if not self.params:
            return self.flat
        if self._written:
            return self._flatten_sunk()
        grads = []
        for p, v in zip(self.params, self._views):
            g = p.grad
            if g is None:
                g = torch.zeros_like(p)
            elif g.data_ptr() == v.data_ptr():
                g = g.clone()                       # already a view of the bucket (no fresh gradient this step): cat must not alias `out`
            grads.append(g.reshape(-1))
        torch.cat(grads, out=self.flat)
        for p, v in zip(self.params, self._views):
            p.grad = v
        return self.flat

    def all_reduce(self, average=True, force=False):
        """SUM all-reduce of the bucket over the ranks; average=False leaves the sum (FlatAdam.step(grad_scale=1/world) folds the division
        into the update).  At world size 1 there is nothing to reduce and no collective is issued -- unless force=True (and a process
        group exists): the call then really goes through RCCL on the one-rank communicator (identity result, bit for bit), which is
        how the collective path is exercised on a single GPU (tests/test_gpu_collective.py, bench.py --force-collective)."""
        if dist.is_initialized() and (dist.get_world_size() > 1 or force):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            if average and dist.get_world_size() > 1:
                self.flat.div_(dist.get_world_size())
        return self.flat

    def all_reduce_mean(self):
        self.flatten()
        return self.all_reduce()


class FlatAdam:
    """Adam over the bucket's parameters as ONE buffer: the parameters are moved into a flat fp32 tensor (each `p.data` becomes a view of
    it -- do this before anything captures their addresses), the gradients are the bucket's flat tensor, and step() is a single
    gspn_adam_flat launch (torch.optim.Adam's update rule; tested against it) instead of a multi-tensor launch per ~50 tensors."""

    def __init__(self, bucket, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, device_step=False):
        """device_step=True (r06): the update count lives in device memory (gspn_adam_flat_dev), no argument of step() changes between calls --
        step() can then be captured into a hipGraph together with the backward pass and replays correctly (`t` reads the counter back)"""
        self.bucket = bucket
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.flat = torch.empty_like(bucket.flat)
        off = 0
        with torch.no_grad():
            for p, s in zip(bucket.params, bucket.sizes):
                view = self.flat[off:off + s].view_as(p)
                view.copy_(p)
                p.data = view
                off += s
        if getattr(bucket, "sinks", False):
            bucket.attach_sinks()                   # the parameters have just moved: the sinks are keyed by their data pointers
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self._t = 0
        self._dev_state = torch.zeros(2, dtype=torch.int64, device=self.flat.device) if device_step else None

    @property
    def t(self):
        """number of updates done (device_step: read back from the device -- a synchronisation)"""
        return int(self._dev_state[0].item()) if self._dev_state is not None else self._t

    @t.setter
    def t(self, value):
        self._t = int(value)
        if self._dev_state is not None:
            self._dev_state[0] = int(value)

    def zero_grad(self, set_to_none=True):
        for p in self.bucket.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def step(self, grad_scale=1.0):
        from . import _lib as L
        with torch.cuda.device(self.flat.device):
            if self._dev_state is not None:
                L.check(L.lib().gspn_adam_flat_dev(self.flat.numel(), L.ptr(self.flat), L.ptr(self.bucket.flat), L.ptr(self.m), L.ptr(self.v), self.lr,
                                                   self.betas[0], self.betas[1], self.eps, self.weight_decay, float(grad_scale), L.ptr(self._dev_state),
                                                   L.stream()), "adam_flat_dev")
                return
            self._t += 1
            L.check(L.lib().gspn_adam_flat(self.flat.numel(), L.ptr(self.flat), L.ptr(self.bucket.flat), L.ptr(self.m), L.ptr(self.v), self.lr,
                                           self.betas[0], self.betas[1], self.eps, self.weight_decay, float(grad_scale), self._t, L.stream()), "adam_flat")


def average_moving_statistics(tensors):
    """Plain data parallelism (the default) normalises every replica with ITS OWN batch statistics, so the ranks' moving_mean /
    moving_variance drift apart (the parameters do not: their gradients are all-reduced).  Call this before evaluating or checkpointing:
    one flat all-reduce leaves the rank AVERAGE of every moving statistic on every rank.  `tensors`: an iterable of the moving_mean /
    moving_variance tensors (e.g. `[v for k, v in store.vars.items() if "moving_" in k]`).  The mean of the replicas' moving variances
    is an approximation of the global-batch variance (it leaves out the spread of the replicas' means): exact global-batch statistics
    are what SyncBN is for (mlp.SYNC_BN; with it the moving statistics are identical on all ranks and this call is a no-op in effect)."""
    ts = [t for t in tensors]
    if not ts or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    flat = torch.cat([t.detach().reshape(-1).float() for t in ts])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    off = 0
    with torch.no_grad():
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n

