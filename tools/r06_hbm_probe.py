"""What the memory system sustains for pure streaming writes / reads / copies on this box (torch kernels over buffers far larger than the 256 MB
Infinity Cache, and over 64 MB buffers that fit it): the denominators behind the 'floor' columns of the layer tables."""
import torch
dev = torch.device("cuda", 0)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for mb in (64, 2048):
    n = mb * 1024 * 1024 // 4
    a = torch.rand(n, device=dev); b = torch.empty_like(a); c = torch.rand(n, device=dev)
    w = t(lambda: b.fill_(1.0)); r = t(lambda: a.sum()); cp = t(lambda: b.copy_(a)); ad = t(lambda: torch.add(a, c, out=b))
    print("%5d MB buffers: write %.2f TB/s   read (sum) %.2f TB/s   copy %.2f TB/s (r+w bytes)   add %.2f TB/s (2r+w bytes)" % (
        mb, n * 4 / w / 1e12, n * 4 / r / 1e12, 2 * n * 4 / cp / 1e12, 3 * n * 4 / ad / 1e12), flush=True)
