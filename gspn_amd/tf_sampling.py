"""Drop-in for tf_ops/sampling/tf_sampling.py: same function names, argument order and
gradients, over torch tensors on a ROCm device."""
import torch

from . import _lib as L

# bench.py sets this to a list to collect (start_event, end_event, b, n, m) around every FPS launch,
# recorded on the stream the kernel is launched on (roofline.achieved is measured live from these)
PROFILE = None


def farthest_point_sample(npoint, inp):
    """tf_sampling.py:48-57 -- inp (batch, ndataset, 3) float32 -> (batch, npoint) int32.
    Non-differentiable (ops.NoGradient('FarthestPointSample'))."""
    npoint = int(npoint)
    if npoint <= 0:
        raise ValueError("FarthestPointSample expects positive npoint")               # tf_sampling.cpp:99
    inp = L.need(inp.detach(), torch.float32, 3, "inp")
    if inp.shape[2] != 3:
        raise ValueError("FarthestPointSample expects (batch_size,num_points,3) inp shape")   # tf_sampling.cpp:105
    b, n, _ = inp.shape
    out = torch.empty((b, npoint), dtype=torch.int32, device=inp.device)
    temp = None
    if n > 32768:   # GSPN_FPS_RESIDENT_MAX: only the streaming kernel needs the (32,n) scratch of tf_sampling.cpp:115
        temp = torch.empty((min(b, 32), n), dtype=torch.float32, device=inp.device)
    with torch.cuda.device(inp.device):
        ev = None
        if PROFILE is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        L.check(L.lib().gspn_farthestpointsampling(b, n, npoint, L.ptr(inp), L.ptr(temp), L.ptr(out), L.stream()),
                "farthest_point_sample")
        if ev is not None:
            ev[1].record()
            PROFILE.append((ev[0], ev[1], b, n, npoint))
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
            L.check(L.lib().gspn_scatteraddpoint(b, ctx.n, m, L.ptr(out_g), L.ptr(idx), L.ptr(inp_g), L.stream()), "gather_point_grad")
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
