"""cost of a busy second HW queue on a chain of dependent kernels: N launches of an elementwise kernel (graph-replayed) with and
without a long single-workgroup kernel running on another stream"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gspn_amd.tf_sampling import farthest_point_sample
dev = torch.device('cuda', 0)
xyz1 = torch.rand(1, 32768, 3, device=dev)
side = torch.cuda.Stream()
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for numel, N in ((1 << 16, 150), (1 << 24, 150), (1 << 26, 40)):
    x = torch.zeros(numel, device=dev)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): x.add_(1.0)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(N): x.add_(1.0)
    alone = timeit(lambda: g.replay())
    def both():
        ev = torch.cuda.current_stream().record_event()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            farthest_point_sample(2048, xyz1)
            done = side.record_event()
        g.replay()
        torch.cuda.current_stream().wait_event(done)
    fps = timeit(lambda: farthest_point_sample(2048, xyz1))
    b = timeit(both)
    print("%3d x add_(%8d floats): chain alone %.3f ms (%.1f us/kernel), FPS(1 scene) alone %.3f ms, overlapped %.3f ms -> +%.1f us per kernel" % (
        N, numel, alone, alone / N * 1e3, fps, b, (b - max(alone, fps)) / N * 1e3 if b > alone else 0.0))
