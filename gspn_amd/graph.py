"""hipGraph capture of the launch-bound part of a training step.

One fwd+bwd of the set-abstraction stack is ~200 short kernels; enqueueing them from Python costs about as
much wall time as the GPU needs to run them once the geometry is overlapped (geometry.py).  CapturedStep
records the step once into a hipGraph (torch.cuda.CUDAGraph == hipGraph on ROCm: every C-ABI launch goes to
the capturing stream, scratch comes from the graph's private pool) and replays it with one call.

Contract: the captured callable reads its inputs from tensors whose storage does not change between
replays (copy each new batch into them), and leaves its results (e.g. the flat gradient bucket) in
persistent tensors.  Work that must stay observable per launch -- the geometry stream with its HIP events --
stays outside the graph.
"""
import torch


class CapturedStep:
    def __init__(self, fn, warmup=2, pool=None):
        """fn(): enqueue the step on the current stream, return a DETACHED tensor (e.g. loss.detach()) or None -- a result that still
        carries its autograd graph keeps that graph's AccumulateGrad nodes alive into the next capture.  fn is run `warmup` times
        eagerly on a side stream (lazy initialisation, variable creation), then captured."""
        self.graph = torch.cuda.CUDAGraph()
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        # capture_error_mode "thread_local": only THIS thread is held to the capture rules.  Under the default ("global") any other thread
        # that touches the runtime while the capture is open fails with hipErrorStreamCaptureUnsupported -- and the process-group watchdog
        # of torch.distributed polls the events of earlier collectives from its own thread: with RCCL initialised the capture of the
        # step killed the process about once in thirty runs ("operation not permitted when stream is capturing" out of
        # ProcessGroupNCCL::Watchdog; tests/test_gpu_collective.py, r03).
        from . import _lib as L
        L.CAPTURES_OPEN[0] += 1               # (no event polling from the binding's check points while a capture is open: _lib.check)
        try:
            with torch.cuda.graph(self.graph, pool=pool, capture_error_mode="thread_local"):
                self.result = fn()
        finally:
            L.CAPTURES_OPEN[0] -= 1

    def pool(self):
        return self.graph.pool()

    def replay(self):
        self.graph.replay()
        return self.result


def _pairs(dst, src, out):
    if isinstance(dst, torch.Tensor):
        out.append((dst, src))
    elif isinstance(dst, dict):
        for k in dst:
            _pairs(dst[k], src[k], out)
    elif isinstance(dst, (list, tuple)):
        for d, s in zip(dst, src):
            _pairs(d, s, out)
    elif hasattr(dst, "tensors"):
        out.extend(zip(dst.tensors(), src.tensors()))
    return out


def copy_into(dst, src):
    """Copy a (nested) geometry result into persistent buffers of the same structure, on the current stream, in ONE launch
    (gspn_multi_copy) instead of one copy kernel per tensor."""
    import ctypes
    from . import _lib as L
    pairs = _pairs(dst, src, [])
    keep = []
    ps, pd, pb = [], [], []
    for d, s in pairs:
        if d.shape != s.shape or d.dtype != s.dtype or d.device != s.device:
            raise ValueError("copy_into: mismatched tensors %s %s vs %s %s" % (tuple(d.shape), d.dtype, tuple(s.shape), s.dtype))
        if not d.is_contiguous():
            raise ValueError("copy_into: destination buffers must be contiguous")
        s = s.contiguous()
        keep.append(s)
        ps.append(s.data_ptr()); pd.append(d.data_ptr()); pb.append(s.numel() * s.element_size())
    n = len(ps)
    if n:
        with torch.cuda.device(pairs[0][0].device):
            L.check(L.lib().gspn_multi_copy(n, (ctypes.c_void_p * n)(*ps), (ctypes.c_void_p * n)(*pd), (ctypes.c_long * n)(*pb), L.stream()), "multi_copy")
    return dst
