import sys, time, ctypes, torch
dev = torch.device('cuda', 0)
lib = ctypes.CDLL('tools/libspin_probe.so')
lib.launch_spin.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(4, device=dev)
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
alone = timeit(lambda: g.replay())
print("40 x add_(64M floats) alone: %.3f ms" % alone)
CYC = 250_000_000 // 100 * 1    # clock64 ticks at 100 MHz: 2.5 ms
for name, which, blocks, threads, iters in (("valu loop 1024 thr, 128 VGPRs", 5, 8, 1024, 5200), ("  + s_sleep 1 per 64 FMAs", 6, 8, 1024, 5200), ("  + s_sleep 4 per 64 FMAs", 7, 8, 1024, 5200),
                                            ("  + 64 s_nop cycles per 64 FMAs", 8, 8, 1024, 5200)):
    def sidek():
        lib.launch_spin(which, blocks, threads, iters, out.data_ptr(), side.cuda_stream)
    sidek(); torch.cuda.synchronize()
    t_side = timeit(sidek, 3)
    def both():
        ev = torch.cuda.current_stream().record_event()
        side.wait_event(ev)
        sidek()
        done = side.record_event()
        g.replay()
        torch.cuda.current_stream().wait_event(done)
    print("  side = %-40s alone %.3f ms; overlapped with the chain: %.3f ms" % (name, t_side, timeit(both)))
