import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gspn_amd import tf_sampling
from gspn_amd.tf_sampling import farthest_point_sample
dev = torch.device('cuda', 0)
side = torch.cuda.Stream()
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
x = torch.zeros(1 << 26, device=dev)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): x.add_(1.0)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
with torch.cuda.graph(g):
    for _ in range(40): x.add_(1.0)
print("40 x add_(64M floats) alone: %.3f ms" % timeit(lambda: g.replay()))
for mode, b, n, m in (("cells", 8, 32768, 2048), ("cells", 1, 32768, 2048), ("resident", 8, 32768, 1300), ("resident", 1, 32768, 1300), ("resident", 8, 2048, 2048), ("resident", 8, 8192, 2048)):
    tf_sampling.FPS_MODE = mode
    xyz = torch.rand(b, n, 3, device=dev)
    def sidek(): farthest_point_sample(m, xyz)
    t_side = timeit(sidek, 3)
    def both():
        ev = torch.cuda.current_stream().record_event()
        with torch.cuda.stream(side):
            side.wait_event(ev); sidek(); done = side.record_event()
        g.replay()
        torch.cuda.current_stream().wait_event(done)
    print("  side = FPS %-8s b=%d n=%5d m=%4d  alone %.3f ms; overlapped with the chain: %.3f ms" % (mode, b, n, m, t_side, timeit(both)))
