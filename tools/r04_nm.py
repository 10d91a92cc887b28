"""nn_distance forward at the model's shape (2048 clouds x (512, 512)) and the reference harness shape, graph-timed"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gspn_amd.tf_nndistance import nn_distance
g = torch.Generator(device="cuda").manual_seed(1)
for (nb, n, m) in ((2048, 512, 512), (32, 16384, 1024)):
    a = torch.randn(nb, n, 3, device="cuda", generator=g); c = torch.randn(nb, m, 3, device="cuda", generator=g)
    with torch.no_grad():
        us = bench._ev_time(lambda: nn_distance(a, c)) * 1e3
    print("nn_distance %d x (%d, %d): %.1f us  (%.2f of the 78.6 T lane-instr/s at 9 per pair)" % (nb, n, m, us, 2.0 * nb * n * m * 9 / (us * 1e-6) / 78.6e12))
