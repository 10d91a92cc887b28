"""times the FPS kernels alone (HIP events around the sampling kernel, pre-pass outside): single-CU cell kernel at the bench shape and
the multi-CU kernel over scene sizes / workgroup counts.  usage: python tools/fps_multi_bench.py [quick]"""
import sys

import numpy as np
import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from gspn_amd import _lib as L  # noqa: E402
from gspn_amd import tf_sampling  # noqa: E402


def time_case(b, n, m, G, reps=3, force=True):
    xyz = torch.from_numpy(np.random.default_rng(0).random((b, n, 3), dtype=np.float32)).cuda()
    tf_sampling.FPS_MULTI_FORCE = force
    tf_sampling.FPS_MULTI_G = G
    best = 1e9
    for _ in range(reps):
        tf_sampling.PROFILE = []
        idx = tf_sampling.farthest_point_sample(m, xyz)
        torch.cuda.synchronize()
        ms = tf_sampling.PROFILE[0][0].elapsed_time(tf_sampling.PROFILE[0][1])
        best = min(best, ms)
    tf_sampling.PROFILE = None
    tf_sampling.FPS_MULTI_FORCE = False
    return best, int(idx.sum())


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    cases = [(8, 32768, 2048, -1), (8, 32768, 2048, 1), (8, 32768, 2048, 2), (8, 32768, 2048, 4),
             (8, 65536, 2048, 2), (8, 65536, 2048, 3), (8, 65536, 2048, 4), (8, 65536, 2048, 6), (8, 65536, 2048, 8),
             (1, 32768, 2048, 4), (1, 32768, 2048, 8), (1, 32768, 2048, 16)]
    if not quick:
        cases += [(1, 150000, 30000, 0), (1, 150000, 30000, 5), (1, 150000, 30000, 16), (1, 1000000, 30000, 0)]
    for b, n, m, G in cases:
        ms, chk = time_case(b, n, m, max(G, 0), force=G >= 0)
        print("b=%d n=%d m=%d G=%s : %.3f ms  (%.3f us/pick)  checksum %d" % (b, n, m, "single-CU" if G < 0 else G, ms, ms * 1e3 / m, chk), flush=True)
