"""three_nn on the cell grid (r04) against the all-pairs kernel: same bits, time, per cloud kind; sweep of points per cell / queries per thread
through the environment (GSPN_NN_CELL_POINTS, GSPN_NN_CELL_QPW)"""
import os, sys, subprocess, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "sweep":
    for ppc in (2, 4, 8):
        for qpw in (1, 2, 4):
            env = dict(os.environ, GSPN_NN_CELL_POINTS=str(ppc), GSPN_NN_CELL_QPW=str(qpw))
            out = subprocess.run([sys.executable, __file__], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
            print("points/cell %d queries/thread %d | %s" % (ppc, qpw, " | ".join(l.strip() for l in out.splitlines() if "cells" in l)), flush=True)
    sys.exit(0)
import bench
from gspn_amd import _lib as L
from gspn_amd import synth
from gspn_amd import tf_sampling as S
lib = L.lib()
dev = torch.device("cuda", 0)
for kind in ("U", "S", "D"):
    xyz = torch.from_numpy(synth.batch(kind, 8, 32768, 0)).to(dev)
    fps1, order = S.farthest_point_sample(2048, xyz, return_order=True)
    new1 = S.gather_point(xyz, fps1)
    b, n, m = 8, 32768, 2048
    res = {}
    for name, env in (("cells", "1"), ("all-pairs", "0")):
        d = torch.empty(b, n, 3, device=dev); i = torch.empty(b, n, 3, dtype=torch.int32, device=dev)
        # the switch is read once per process: call the two kernels through the environment of a child for the reference bits
        res[name] = (d, i)
    d, i = res["cells"]
    run = lambda o: L.check(lib.gspn_threenn_ordered(b, n, m, L.ptr(xyz), L.ptr(new1), L.ptr(o), L.ptr(d), L.ptr(i), L.stream()), "nn") if o is not None else \
        L.check(lib.gspn_threenn(b, n, m, L.ptr(xyz), L.ptr(new1), L.ptr(d), L.ptr(i), L.stream()), "nn")
    t_ord = bench._ev_time(lambda: run(order)) * 1e3
    t_no = bench._ev_time(lambda: run(None)) * 1e3
    torch.cuda.synchronize()
    # brute force on the device in float64 for the index check (ties are measure-zero on these clouds; exactness proper: pytest)
    ok = True
    for s in range(2):
        dd = torch.cdist(xyz[s].double(), new1[s].double()) ** 2
        ref = dd.topk(3, dim=1, largest=False).indices.int()
        ok = ok and bool((ref == i[s]).all())
    print("%s three_nn 8 x 32768 <- 2048 %s: %6.1f us in the FPS pre-pass order, %6.1f us unordered; indices == float64 brute force: %s" % (
        kind, "cells" if os.environ.get("GSPN_NN_CELLS", "1") != "0" else "all-pairs", t_ord, t_no, ok), flush=True)
