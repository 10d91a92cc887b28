"""The weight-independent half of the set-abstraction graph, and a side stream to run it ahead of time.

Everything `pointnet_sa_module` / `pointnet_fp_module` compute from coordinates alone -- farthest point
sampling, the gather of the sampled centres, the ball query (utils/pointnet_util.py:38-40) and the 3-NN
search with its inverse-distance weights (:155-160) -- depends on `xyz` only, never on a weight or a
feature.  FPS is also the one kernel that cannot fill the chip: it is sequential in `npoint` and one
scene lives on one CU (8 scenes -> 8 of 256 CUs for 2.5 ms at 8 x 32768 -> 2048).  So the geometry of
batch k+1 is computed on its own HIP stream while the MFMA layers of batch k own the other CUs:

    geo = GeometryStream(device)
    pend = geo.submit(pn2_geometry, xyz_next)          # side stream, returns immediately
    out = pn2_fea_extractor(xyz, feats, 'fea', True, bn_decay, geometry=pend_prev.get())

Results are identical to the inline path (same kernels, same inputs); only the schedule changes.
"""
import os

import torch

from . import _lib as L
from .invlists import inverse_lists  # noqa: F401  (re-exported: tools and tests import it from here)
from .tf_grouping import knn_point, query_ball_point
from .tf_interpolate import three_nn
from .tf_sampling import farthest_point_sample, gather_point


class SAGeometry:
    """new_xyz (b,npoint,3), idx (b,npoint,nsample) int32, pts_cnt (b,npoint) int32 or None (knn), plus the inverse lists the gradient
    of the grouping gathers through: order (b, npoint*nsample) int32 = grouped positions sorted by data-point index, offsets (b, n+1)."""
    __slots__ = ("new_xyz", "idx", "pts_cnt", "npoint", "nsample", "order", "offsets", "rel", "gidx", "scan_order", "feat4")

    def __init__(self, new_xyz, idx, pts_cnt, npoint, nsample, order=None, offsets=None, rel=None, gidx=None, scan_order=None, feat4=None):
        self.new_xyz, self.idx, self.pts_cnt, self.npoint, self.nsample = new_xyz, idx, pts_cnt, npoint, nsample
        self.order, self.offsets = order, offsets
        # fused SA front end (gspn_sa_rel): per grouped row its centred coordinates (b*npoint*nsample, 4) and its source row (int32)
        self.rel, self.gidx = rel, gidx
        # spatial order of the INPUT cloud left behind by the FPS pre-pass ((b,n) int32 or None): fp_geometry scans in it
        self.scan_order = scan_order
        # r06: the module's INPUT features padded to 16-byte rows ((b*n, 4*ceil(c/4)), what the gathering first layer reads) when the caller handed
        # the raw features to sa_geometry(points=...): input-only data like the coordinates, so the pad runs ahead on the geometry stream too
        self.feat4 = feat4

    def tensors(self):
        return [t for t in (self.new_xyz, self.idx, self.pts_cnt, self.order, self.offsets, self.rel, self.gidx, self.scan_order, self.feat4) if t is not None]


class FPGeometry:
    """idx (b,n1,3) int32 and weight (b,n1,3) float32 of pointnet_util.py:155-160, plus the inverse lists the gradient of the
    interpolation gathers through: order (b, 3*n1) int32 = positions 3*i+t sorted by idx (ties ascending), offsets (b, n2+1) int32."""
    __slots__ = ("idx", "weight", "order", "offsets")

    def __init__(self, idx, weight, order=None, offsets=None):
        self.idx, self.weight, self.order, self.offsets = idx, weight, order, offsets

    def tensors(self):
        return [t for t in (self.idx, self.weight, self.order, self.offsets) if t is not None]


def sa_front(xyz, new_xyz, idx, shift=None):
    """pointnet_util.py:41-42 reduced to what depends on coordinates only: rel[r] = (xyz[idx[r]] - centre, 0) and the source row
    gidx[r] of every grouped row r -- the inputs of the fused front end (gspn_mlp_fwd_gather), 20 bytes per grouped row.
    shift (b, npoint, 3), optional: multi_encoding_net's per-seed shift (model_rpointnet.py:56-57), subtracted after the centre."""
    b, n, _ = xyz.shape
    _, m, ns = idx.shape
    rel = torch.empty((b * m * ns, 4), dtype=torch.float32, device=xyz.device)
    gidx = torch.empty((b * m * ns,), dtype=torch.int32, device=xyz.device)
    if shift is not None:
        shift = L.need(shift.detach(), torch.float32, 3, "shift_pred")
        if tuple(shift.shape) != (b, m, 3):
            raise ValueError("shift_pred must be (batch, npoint, 3)")
    with torch.cuda.device(xyz.device):
        L.check(L.lib().gspn_sa_rel_shift(b, n, m, ns, L.ptr(xyz), L.ptr(new_xyz), L.ptr(shift), L.ptr(idx), L.ptr(rel), L.ptr(gidx), L.stream()),
                "sa_rel")
    return rel, gidx


def pad_features(points):
    """(b, n, c) raw input features -> (b*n, 4*ceil(c/4)) with zero pad columns (gspn_pad_rows), or None when c is already a multiple of 4"""
    b, n, c = points.shape
    if c % 4 == 0:
        return None
    src = L.need(points.detach(), torch.float32, 3, "points").reshape(b * n, c)
    out = torch.empty((b * n, (c + 3) // 4 * 4), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        L.check(L.lib().gspn_pad_rows(b * n, c, out.shape[1], L.ptr(src), L.ptr(out), L.stream()), "pad_rows")
    return out


def sa_geometry(xyz, npoint, radius, nsample, knn=False, inverse=True, front=True, fps=None, points=None):
    """pointnet_util.py:38-40: centres by FPS, neighbours by ball query (or kNN); front: also the coordinate half of the grouping.
    fps: (fps_idx, scan_order) of farthest_point_sample(npoint, xyz, return_order=True) when the caller has already enqueued it (on the
    same stream) -- the long pole of a scene's geometry, worth starting before the host enqueues anything else.
    points: the module's raw INPUT features (no gradient flows into them), optional: their 16-byte-row copy is prepared here (feat4)"""
    xyz = xyz.detach()
    fps_idx, scan_order = fps if fps is not None else farthest_point_sample(npoint, xyz, return_order=True)
    new_xyz = gather_point(xyz, fps_idx)
    if knn:
        _, idx = knn_point(nsample, xyz, new_xyz)
        cnt = None
    else:
        idx, cnt = query_ball_point(radius, nsample, xyz, new_xyz)
    order, offsets = inverse_lists(idx.reshape(idx.shape[0], -1), xyz.shape[1]) if inverse else (None, None)
    rel, gidx = sa_front(xyz, new_xyz, idx) if front else (None, None)
    feat4 = pad_features(points) if (points is not None and front) else None
    return SAGeometry(new_xyz, idx, cnt, npoint, nsample, order, offsets, rel, gidx, scan_order, feat4)


def fp_geometry(xyz1, xyz2, scan_order=None):
    """pointnet_util.py:155-160: three nearest sparse points of every dense point and their normalised 1/d weights.
    scan_order: a spatial order of xyz1 (SAGeometry.scan_order of the level that sampled xyz1), see three_nn."""
    dist, idx = three_nn(xyz1.detach(), xyz2.detach(), order=scan_order)
    # :157-160 -- dist = max(dist, 1e-10); norm = sum(1/dist); weight = (1/dist)/norm, in one kernel
    weight = torch.empty_like(dist)
    with torch.cuda.device(dist.device):
        L.check(L.lib().gspn_three_nn_weights(dist.numel() // 3, L.ptr(dist), L.ptr(weight), L.stream()), "three_nn_weights")
    # inverse lists for the gradient (three_interpolate_grad as a gather in the reference's own summation order)
    b, n1, _ = idx.shape
    order, offsets = inverse_lists(idx.reshape(b, 3 * n1), xyz2.shape[1])
    return FPGeometry(idx, weight, order, offsets)


class PendingGeometry:
    def __init__(self, value, event, stream):
        self._value, self._event, self._stream = value, event, stream

    def get(self, host_wait=False):
        """Make the consumer's current stream wait for the geometry and hand the tensors over to it.
        host_wait=True blocks the calling thread until the geometry is complete instead of queueing a cross-stream wait: a wait between
        two hardware queues is not free on the waiting queue (0.08 ms per step measured on MI355X when the layers' stream waits every
        step), while geometry prefetched a step or two ahead has long finished when it is asked for."""
        cur = torch.cuda.current_stream()
        if cur != self._stream:
            if host_wait:
                self._event.synchronize()
            else:
                cur.wait_event(self._event)
            for t in _tensors_of(self._value):
                t.record_stream(cur)
            if host_wait:
                L.check_async()                  # the geometry is complete: so is the status word of a multi-CU FPS launch inside it
        return self._value


def _tensors_of(v):
    if isinstance(v, torch.Tensor):
        return [v]
    if hasattr(v, "tensors"):
        return v.tensors()
    if isinstance(v, dict):
        return [t for x in v.values() for t in _tensors_of(x)]
    if isinstance(v, (list, tuple)):
        return [t for x in v for t in _tensors_of(x)]
    return []


_parked = []          # streams that turned out to share a hardware queue with the layers: kept alive so that their queue slot stays taken


def runs_beside(busy, other, work=None, cycles=1500000):
    """True if work enqueued on `other` (default: one trivial kernel) completes while a long spin kernel is still running on `busy`, i.e.
    the two streams sit on different hardware queues.  (HIP multiplexes streams onto GPU_MAX_HW_QUEUES -- by default four -- hardware
    queues round-robin in creation order; kernels of two streams on one queue are serialised.)  Costs about a millisecond; call at set-up."""
    dev = busy.device
    torch.cuda.synchronize(dev)
    tiny = torch.zeros(64, device=dev)
    with torch.cuda.stream(busy):
        torch.cuda._sleep(int(cycles))
        spun = busy.record_event()
    with torch.cuda.stream(other):
        if work is None:
            tiny.fill_(1.0)
        else:
            work()
        done = other.record_event()
    done.synchronize()
    beside = not spun.query()
    torch.cuda.synchronize(dev)
    return beside


class GeometryStream:
    """A side HIP stream for coordinate-only work.  submit(fn, *tensors) runs fn on it after the tensors'
    producer (the caller's current stream) and returns a PendingGeometry."""

    def __init__(self, device=None, priority=0, beside=None, probes=(), attempts=12, agree=None):
        """beside: streams this one must run CONCURRENTLY with (default: the caller's current stream, i.e. the one the layers run on);
        probes: callables that enqueue work on the caller's current stream which must not queue up behind this stream either (e.g. a
        collective that the communication library runs on a stream of its own).  Every probe runs in EVERY attempt (no short-circuit).
        agree: with several ranks and a collective among the probes every rank must run the same number of attempts -- the verdict of an
        attempt is timing-based and could differ between ranks, which would leave them with different numbers of collectives issued.
        agree(ok) -> bool makes it common (e.g. an all-reduce MIN of the flag); it is called once per attempt on every rank.
        HIP hands its hardware queues to streams round-robin in creation order, so which queue a new stream lands on depends on every
        stream anybody created before -- torch's capture streams, and RCCL's: with a process group initialised, the geometry stream of
        round 2 landed on the LAYERS' queue and every step paid +1.1 ms (r03, rocprofv3 queue ids).  So the stream is not trusted, it
        is tested: a spin kernel on each stream of `beside` must not delay a trivial launch here (and a spin here must not delay the
        probes); a stream that fails is parked and the next one tried."""
        cur = torch.cuda.current_stream(device)
        beside = [cur] if beside is None else list(beside)
        self.stream = None
        self.tried = 0
        fallback = None
        for _ in range(max(1, attempts)):
            st = torch.cuda.Stream(device=device, priority=priority)
            with torch.cuda.stream(st):
                torch.zeros(1, device=st.device)        # first use binds the stream to its hardware queue
            self.tried += 1
            ok = all(runs_beside(b, st) for b in beside)
            beside_ok = ok
            for p in probes:                                 # (all of them, whatever the verdict so far: the same collectives on every rank)
                if agree is not None:
                    agree(True)                              # a collective of its own: the ranks enter the probe together (a rank that waits
                                                             # for a late peer inside the probe would blame the stream for the skew)
                ok = runs_beside(st, cur, work=p, cycles=6000000) and ok
            if agree is not None:
                ok = bool(agree(ok))
            if ok:
                self.stream = st
                break
            _parked.append(st)
            if beside_ok and fallback is None:
                fallback = st                            # runs beside the layers at least (only a probe objected)
        if self.stream is None:                          # fewer free hardware queues than streams that must run side by side
            self.stream = fallback if fallback is not None else _parked[-1]
            _parked.remove(self.stream)
            self.shares_queue = True
        else:
            self.shares_queue = False

    def submit(self, fn, *args, after="current", **kwargs):
        """after = "current" (default): fn starts once everything enqueued so far on the caller's current stream is done (its inputs may
        have just been produced there); an Event: once that event is reached (e.g. the loader's); None: immediately (the inputs are
        known to be ready -- e.g. resident since an earlier synchronisation).  Recording an event on the stream that replays the captured
        layers is not free (measured 0.09 ms per step on MI355X), so pass what is actually needed."""
        ready = torch.cuda.current_stream().record_event() if isinstance(after, str) else after
        with torch.cuda.stream(self.stream):
            if ready is not None:
                self.stream.wait_event(ready)
            for a in args:
                if isinstance(a, torch.Tensor):
                    a.record_stream(self.stream)
            with torch.no_grad():
                value = fn(*args, **kwargs)
            done = self.stream.record_event()
        return PendingGeometry(value, done, self.stream)
