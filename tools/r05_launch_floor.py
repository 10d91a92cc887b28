"""what a dependent kernel costs in a chain: N tiny kernels (a) enqueued eagerly, (b) replayed from a hipGraph; environment variants of the
HIP runtime are tried by the caller (tools/r05_launch_floor.sh)"""
import os, sys, time, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gspn_amd import _lib as L
lib = L.lib()
dev = torch.device("cuda", 0)
x = torch.zeros(1 << 20, device=dev)
N = 200
def chain(n_elems):
    for _ in range(N):
        L.check(lib.gspn_fill_zero(L.ptr(x), n_elems, L.stream()), "fill")       # one small hand-written kernel
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e3 / N
y = torch.zeros(64, device=dev)
idx = torch.zeros(64, dtype=torch.long, device=dev)
def chain_add():
    for _ in range(N):
        y.add_(1.0)                       # a real kernel: one load -> one store of data the previous node wrote (one round trip)
def chain_dep2():
    for _ in range(N):
        torch.index_select(y, 0, idx, out=y2)      # load idx, then load y[idx] (two dependent round trips), store
y2 = torch.zeros(64, device=dev)
for name, fn in (("add_ 64 elems (1 round trip)", chain_add), ("index_select 64 (2 dependent round trips)", chain_dep2)):
    eager = timed(fn)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    print("%s  %s: eager %.2f us/kernel, graph %.2f us/kernel" % (os.environ.get("TAG", "default"), name, eager, timed(g.replay)))
for n_elems in (64, 1 << 20):
    eager = timed(lambda: chain(n_elems))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain(n_elems)
    graph = timed(g.replay)
    print("%s  elems %8d: eager %.2f us/kernel, graph %.2f us/kernel" % (os.environ.get("TAG", "default"), n_elems, eager, graph))
