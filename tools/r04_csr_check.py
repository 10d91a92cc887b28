"""inverse lists on clustered clouds: gspn_inverse_lists against a stable argsort, and its time, per cloud kind and list"""
import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from gspn_amd import synth
from gspn_amd.fea_extractor import pn2_geometry
from gspn_amd.geometry import inverse_lists
for kind in sys.argv[1:] or ["U", "S", "D"]:
    xyz = torch.from_numpy(synth.batch(kind, 8, 32768, 0)).cuda()
    G = pn2_geometry(xyz)
    for name, idx2d, n in [("SA2 ball", G["sa"][1].idx.reshape(8, -1), 2048), ("SA3 ball", G["sa"][2].idx.reshape(8, -1), 512),
                           ("FP1 3-NN", G["fp"][0].idx.reshape(8, -1), 128), ("FP2 3-NN", G["fp"][1].idx.reshape(8, -1), 512), ("FP3 3-NN", G["fp"][2].idx.reshape(8, -1), 2048)]:
        order, offsets = inverse_lists(idx2d, n)
        ref = torch.argsort(idx2d.long(), dim=1, stable=True).int()
        cnt = torch.stack([torch.bincount(idx2d[s].long(), minlength=n) for s in range(8)])
        off_ref = torch.cat([torch.zeros(8, 1, dtype=torch.long, device="cuda"), cnt.cumsum(1)], 1).int()
        ok = torch.equal(order, ref) and torch.equal(offsets, off_ref)
        us = bench._ev_time(lambda: inverse_lists(idx2d, n)) * 1e3
        print("%s %-9s L=%6d n=%5d longest list %6d, lists > 64: %5.1f %%  | %6.1f us | equals the stable argsort: %s" % (
            kind, name, idx2d.shape[1], n, int(cnt.max()), 100.0 * float((cnt > 64).float().mean()), us, ok), flush=True)
