"""Drop-in for tf_ops/sampling/tf_sampling.py: same function names, argument order and
gradients, over torch tensors on a ROCm device."""
import os

import torch

from . import _lib as L
from . import invlists

# bench.py sets this to a list to collect (start_event, end_event, b, n, m) around every FPS launch,
# recorded on the stream the kernel is launched on (roofline.achieved is measured live from these)
PROFILE = None
EVENT_POOL = []            # timing events handed back by the consumer of PROFILE: creating / destroying HIP events in a hot loop costs the host
PROFILE_BUDGET = [1 << 30]  # launches still to be bracketed: event pairs cost host time (on ROCm 7.2 the cost per pair grows with the pairs already recorded), so bench.py brackets the first 24 steps only
PROFILE_MIN_N = 0          # only launches with at least this many points per scene are bracketed


# 'cells' (default): HIP spatial pre-pass + fps_cell_kernel (batched, culled; identical output) for n >= FPS_CELLS_MIN_N;
# 'resident': always the plain on-chip kernel; 'cells_torch': cell kernel on a torch-side pre-sort (tests: arbitrary partitions)
FPS_MODE = os.environ.get("GSPN_FPS_MODE", "cells")
FPS_CELLS_MIN_N = int(os.environ.get("GSPN_FPS_CELLS_MIN_N", "8192"))
VOXEL_ORDER = os.environ.get("GSPN_FPS_VOXEL_ORDER", "1") != "0"      # return_order: the 16^3-voxel Morton order of the pre-pass (finer than its 16 cells)
# workgroups (CUs) per scene for n > 32768 (0 = the library's choice); any n goes multi-CU when FPS_MULTI_FORCE is set (tests)
FPS_MULTI_G = int(os.environ.get("GSPN_FPS_MULTI_G", "0"))
FPS_MULTI_FORCE = False
# check the multi-CU kernel's status word synchronously after every launch (costs a stream synchronisation; default: asynchronously,
# at the next op call / PendingGeometry.get / L.check_async())
FPS_MULTI_SYNC_CHECK = os.environ.get("GSPN_FPS_MULTI_SYNC_CHECK", "0") == "1"


def _spread10(v):
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


def _cell_prepass(inp):
    """Sort every scene into 16 equal Morton cells, reference tie rank (k mod 512, k) inside a cell.
    Returns sxyz (b,n,3), perm (b,n) int32 [sorted position -> original index], csz."""
    b, n, _ = inp.shape
    lo = inp.amin(dim=1, keepdim=True)
    ext = (inp.amax(dim=1, keepdim=True) - lo).clamp_min(1e-30)
    q = ((inp - lo) / ext * 1024.0).to(torch.int64).clamp_(0, 1023)
    code = _spread10(q[..., 0]) | (_spread10(q[..., 1]) << 1) | (_spread10(q[..., 2]) << 2)
    order = torch.argsort(code, dim=1, stable=True)
    csz = (n + 15) // 16
    cell = (torch.arange(n, device=inp.device, dtype=torch.int64) // csz).unsqueeze(0)
    rank = ((order & 511) << 22) | (order >> 9)
    o2 = torch.argsort((cell << 32) | rank, dim=1)
    perm = torch.gather(order, 1, o2)
    sxyz = torch.gather(inp, 1, perm.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    return sxyz, perm.to(torch.int32).contiguous(), csz


def farthest_point_sample(npoint, inp, return_order=False):
    """tf_sampling.py:48-57 -- inp (batch, ndataset, 3) float32 -> (batch, npoint) int32.
    Non-differentiable (ops.NoGradient('FarthestPointSample')).
    return_order (extension): also return the spatial order the pre-pass sorted the scene into -- (batch, ndataset) int32, a
    permutation of the point indices in which neighbours in space sit together (16 cells), or None when the launch had no pre-pass.
    three_nn(..., order=) runs markedly faster on it (geometry.py)."""
    npoint = int(npoint)
    if npoint <= 0:
        raise ValueError("FarthestPointSample expects positive npoint")               # tf_sampling.cpp:99
    L.check_async()                                      # a multi-CU launch of an earlier call that reported a failure raises here
    inp = L.need(inp.detach(), torch.float32, 3, "inp")
    if inp.shape[2] != 3:
        raise ValueError("FarthestPointSample expects (batch_size,num_points,3) inp shape")   # tf_sampling.cpp:105
    b, n, _ = inp.shape
    out = torch.empty((b, npoint), dtype=torch.int32, device=inp.device)
    lib = L.lib()
    order = None
    with torch.cuda.device(inp.device):
        ev = None

        def tic():
            nonlocal ev
            if PROFILE is not None and PROFILE_BUDGET[0] > 0 and n >= PROFILE_MIN_N:           # events around the sampling kernel alone (the pre-pass is 3 % of the call)
                ev = (EVENT_POOL.pop() if EVENT_POOL else torch.cuda.Event(enable_timing=True),
                      EVENT_POOL.pop() if EVENT_POOL else torch.cuda.Event(enable_timing=True))
                ev[0].record()
                PROFILE_BUDGET[0] -= 1

        if n > 32768 or FPS_MULTI_FORCE:
            # several CUs per scene (sampling_multi.hip); the reference's kernel takes any n (tf_sampling_g.cu:137-141)
            ws = torch.empty((int(lib.gspn_fps_multi_ws_bytes(b, n)) + 3) // 4, dtype=torch.float32, device=inp.device)
            L.check(lib.gspn_fps_multi_prepass(b, n, FPS_MULTI_G, L.ptr(inp), L.ptr(ws), L.stream()), "farthest_point_sample(multi pre-pass)")
            tic()
            L.check(lib.gspn_fps_multi_sample(b, n, npoint, FPS_MULTI_G, L.ptr(inp), L.ptr(ws), L.ptr(out), L.stream()),
                    "farthest_point_sample(multi)")
            # the kernel's status word (1 = a bounded inter-workgroup wait expired, `out` zero-filled past the failure): copied behind
            # the kernel into pinned memory and checked at the next synchronisation point (L.check_async) -- or right here when the
            # caller asked for it.  Not under stream capture (a captured copy would need a persistent host word; FPS is never captured:
            # it runs on the geometry streams).
            if not torch.cuda.is_current_stream_capturing():
                word = ws.view(torch.int32)[int(lib.gspn_fps_multi_status_offset(b, n)) // 4:][:1]
                host = torch.empty(1, dtype=torch.int32, pin_memory=True)
                host.copy_(word, non_blocking=True)
                L.register_async_status(host, torch.cuda.current_stream().record_event(), "farthest_point_sample(multi-CU, b=%d, n=%d, m=%d)" % (b, n, npoint))
                if FPS_MULTI_SYNC_CHECK:
                    L.check_async(block=True)
        elif FPS_MODE == "cells" and FPS_CELLS_MIN_N <= n:
            ws = torch.empty(int(lib.gspn_fps_cells_ws_bytes(b, n)) // 4, dtype=torch.float32, device=inp.device)
            if return_order and VOXEL_ORDER:
                # the scene in 16^3-voxel Morton order, written by the pre-pass's counting sort on its way (4 bytes per point)
                order = torch.empty((b, n), dtype=torch.int32, device=inp.device)
                L.check(lib.gspn_fps_cells_prepass_order(b, n, L.ptr(inp), L.ptr(ws), L.ptr(order), L.stream()), "farthest_point_sample(cells pre-pass)")
            else:
                L.check(lib.gspn_fps_cells_prepass(b, n, L.ptr(inp), L.ptr(ws), L.stream()), "farthest_point_sample(cells pre-pass)")
            tic()
            L.check(lib.gspn_fps_cells_sample(b, n, npoint, L.ptr(inp), L.ptr(ws), L.ptr(out), L.stream()), "farthest_point_sample(cells)")
            if order is None:
                order = ws[:b * n].view(torch.int32).view(b, n)      # perm: sorted position -> original index (first b*n words of ws): 16 cells
        elif FPS_MODE == "cells_torch" and 64 <= n:
            tic()
            sxyz, perm, csz = _cell_prepass(inp)
            inp0 = inp[:, 0, :].contiguous()
            L.check(lib.gspn_fps_cells(b, n, npoint, csz, L.ptr(sxyz), L.ptr(perm), L.ptr(inp0), L.ptr(out), L.stream()),
                    "farthest_point_sample(cells)")
        else:
            tic()
            L.check(lib.gspn_farthestpointsampling(b, n, npoint, L.ptr(inp), L.ptr(None), L.ptr(out), L.stream()),
                    "farthest_point_sample")
        if ev is not None:
            ev[1].record()
            PROFILE.append((ev[0], ev[1], b, n, npoint))
    if return_order:
        return out, order
    return out


class _GatherPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, idx):
        b, n, _ = inp.shape
        m = idx.shape[1]
        out = torch.empty((b, m, 3), dtype=torch.float32, device=inp.device)
        with torch.cuda.device(inp.device):
            L.check(L.lib().gspn_gatherpoint(b, n, m, L.ptr(inp), L.ptr(idx), L.ptr(out), L.stream()), "gather_point")
        ctx.save_for_backward(idx)
        ctx.idx_obj = idx                    # the caller's tensor OBJECT: with the opt-in cache the inverse lists of the gradient are kept on it (invlists.py)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, out_g):
        # tf_sampling.py:43-47 -> [gather_point_grad(inp, idx, out_g), None]
        (idx,) = ctx.saved_tensors
        out_g = out_g.contiguous()
        b, m = idx.shape
        inp_g = torch.empty((b, ctx.n, 3), dtype=torch.float32, device=out_g.device)
        with torch.cuda.device(out_g.device):
            if invlists.use_atomic(3) or m == 0:
                L.check(L.lib().gspn_scatteraddpoint(b, ctx.n, m, L.ptr(out_g), L.ptr(idx), L.ptr(inp_g), L.stream()), "gather_point_grad")
            else:
                # gather through the inverse lists of idx (ascending sample position): deterministic also when a point was sampled
                # several times (npoint > number of distinct points: tf_sampling_g.cu's atomicAdd leaves that order open)
                order, offsets = invlists.cached_inverse_lists(ctx.idx_obj if ctx.idx_obj._version == idx._version else idx, ctx.n)
                L.check(L.lib().gspn_sa_group_concat_grad_csr(b, ctx.n, 3, m, 1, L.ptr(order), L.ptr(offsets), 0, 3, L.ptr(out_g), L.ptr(inp_g),
                                                              L.stream()), "gather_point_grad(csr)")
        return inp_g, None


def gather_point(inp, idx):
    """tf_sampling.py:29-37 -- inp (b,n,3) float32, idx (b,m) int32 -> (b,m,3)."""
    inp = L.need(inp, torch.float32, 3, "inp")
    idx = L.need(idx, torch.int32, 2, "idx")
    if inp.shape[2] != 3:
        raise ValueError("GatherPoint expects (batch_size,num_points,3) inp shape")       # tf_sampling.cpp:131
    if idx.shape[0] != inp.shape[0]:
        raise ValueError("GatherPoint expects (batch_size,num_result) idx shape")          # tf_sampling.cpp:135
    return _GatherPoint.apply(inp, idx)


def prob_sample(inp, inpr):
    """tf_sampling.py:13-21 -- inp (b, ncategory) weights, inpr (b, npoints) uniform numbers -> (b, npoints) int32."""
    inp = L.need(inp.detach(), torch.float32, 2, "inp")
    inpr = L.need(inpr.detach(), torch.float32, 2, "inpr")
    if inpr.shape[0] != inp.shape[0]:
        raise ValueError("ProbSample expects (batch_size,num_points) inpr shape")           # tf_sampling.cpp:79
    b, n = inp.shape
    m = inpr.shape[1]
    temp = torch.empty((b, n), dtype=torch.float32, device=inp.device)
    out = torch.empty((b, m), dtype=torch.int32, device=inp.device)
    with torch.cuda.device(inp.device):
        L.check(L.lib().gspn_probsample(b, n, m, L.ptr(inp), L.ptr(inpr), L.ptr(temp), L.ptr(out), L.stream()), "prob_sample")
    return out
