import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gspn_amd.fea_extractor import pn2_geometry
for K in ("U", "S", "D"):
    xyz_np, _ = bench.synth(8, 32768, 0, K)
    G = pn2_geometry(torch.from_numpy(xyz_np).cuda())
    off = G["fp"][2].offsets.cpu().numpy()
    ln = np.diff(off, axis=1).reshape(-1)
    print(K, "FP3 lists: n", ln.size, "mean %.1f" % ln.mean(), "max", ln.max(), "p50/p90/p99/p99.9", np.percentile(ln, [50, 90, 99, 99.9]).astype(int),
          "targets >128: %d (%.1f %% of entries)  >256: %d (%.1f %%)  >512: %d (%.1f %%)" % ((ln > 128).sum(), 100.0 * ln[ln > 128].sum() / ln.sum(), (ln > 256).sum(), 100.0 * ln[ln > 256].sum() / ln.sum(), (ln > 512).sum(), 100.0 * ln[ln > 512].sum() / ln.sum()))
