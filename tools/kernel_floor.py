"""the cost of one more (dependent) kernel in a captured graph: N launches of a trivial kernel in one stream, replayed; microseconds per kernel.
usage: kernel_floor.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gspn_amd import _lib as L
lib = L.lib()
dev = torch.device("cuda", 0)
x = torch.zeros(256, 64, device=dev); y = torch.empty_like(x)
sc = torch.ones(64, device=dev); sh = torch.zeros(64, device=dev)
big = torch.zeros(262144, 64, device=dev); bigo = torch.empty_like(big)
def tiny(st):
    L.check(lib.gspn_bn_apply(256, 64, L.ptr(x), 64, L.ptr(sc), L.ptr(sh), 0, L.ptr(y), 64, st), "bn_apply")
def large(st):
    L.check(lib.gspn_bn_apply(262144, 64, L.ptr(big), 64, L.ptr(sc), L.ptr(sh), 0, L.ptr(bigo), 64, st), "bn_apply")
def run(n_tiny, n_large):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        st = L.stream()
        for _ in range(3):
            tiny(st); large(st)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            st = L.stream()
            for i in range(max(n_tiny, n_large)):
                if i < n_large: large(st)
                if i < n_tiny: tiny(st)
        for _ in range(3): g.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(10): g.replay()
        e1.record(s); s.synchronize()
        return e0.elapsed_time(e1) * 1e3 / 10
a = run(200, 0); b = run(0, 100); c = run(100, 100)
print("200 tiny kernels: %.2f us each" % (a / 200))
print("100 large (134 MB moved) kernels: %.2f us each" % (b / 100))
print("100 x (large + tiny): %.2f us per pair -> the tiny one adds %.2f us" % (c / 100, c / 100 - b / 100))
