"""gspn_inverse_lists at the five shapes of a bench batch (+ the group_point gradient's at SA level 1), graph-timed: GSPN_CSR_SLICES=1 (one workgroup per scene, r02-r05)
against the default (r06: position slices over several workgroups per scene).  Output checked against a stable argsort."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gspn_amd import synth
from gspn_amd.fea_extractor import pn2_geometry
from gspn_amd.invlists import inverse_lists
kind = sys.argv[1] if len(sys.argv) > 1 else "U"
xyz = torch.from_numpy(synth.batch(kind, 8, 32768, 0)).cuda()
G = pn2_geometry(xyz)
cases = [("SA1 group (2048,32) -> 32768", G["sa"][0].idx.reshape(8, -1), 32768), ("SA2 group (512,32) -> 2048", G["sa"][1].idx.reshape(8, -1), 2048),
         ("SA3 group (128,32) -> 512", G["sa"][2].idx.reshape(8, -1), 512), ("FP1 3 x 512 -> 128", G["fp"][0].idx.reshape(8, -1), 128),
         ("FP2 3 x 2048 -> 512", G["fp"][1].idx.reshape(8, -1), 512), ("FP3 3 x 32768 -> 2048", G["fp"][2].idx.reshape(8, -1), 2048)]
for name, idx, n in cases:
    idx = idx.contiguous()
    order, offsets = inverse_lists(idx, n)
    ref = torch.argsort(idx.long(), dim=1, stable=True).int()
    ok = bool(torch.equal(order, ref))
    ms = min(bench._ev_time(lambda: inverse_lists(idx, n), 3, 20) for _ in range(3))
    print("%s %-32s L %6d : %7.1f us   == stable argsort: %s" % (kind, name, idx.shape[1], ms * 1e3, ok), flush=True)
