"""does hipExtStreamCreateWithCUMask confine kernels on this stack?  times a chip-filling kernel (three_nn 8 x 32768 <- 2048) on streams created
with masks of 256 / 128 / 64 / 32 / 16 compute units, taken as the LOW bits and as every k-th bit"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gspn_amd.tf_interpolate import three_nn
from gspn_amd import tf_sampling as S
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
xyz = torch.from_numpy(bench.synth(8, 32768, 0)[0]).to(dev)
new1 = S.gather_point(xyz, S.farthest_point_sample(2048, xyz))
torch.cuda.synchronize()
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)
def timeit(stream, reps=5):
    with torch.cuda.stream(stream):
        for _ in range(2): three_nn(xyz, new1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps): three_nn(xyz, new1)
        e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("unmasked torch stream: %.1f us" % timeit(torch.cuda.Stream()))
full = (1 << 256) - 1
for n in (256, 128, 64, 32, 16):
    low = (1 << n) - 1
    step = 256 // n
    spread = sum(1 << (i * step) for i in range(n))
    print("%3d CUs: low bits %.1f us   every %d-th bit %.1f us" % (n, timeit(masked_stream(low)), step, timeit(masked_stream(spread))), flush=True)
