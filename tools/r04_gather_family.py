"""The gather family of the captured step at the bench shapes, stand-alone, graph-timed (20 launches between two HIP
events): preagg_fwd (SA2 / SA3 / FP3), preagg_bwd_dy, sa_group_concat_grad_csr (SA2 / SA3), fp_concat_grad_csr (FP3 pre-aggregated form,
FP2, FP1).  Prints microseconds and the bytes each launch must move; with `check` as argv[1] also compares every output with the
previous library (`GSPN_LIB_OLD` = path of a .so built before the change) bit for bit."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gspn_amd import _lib as L
from gspn_amd.fea_extractor import pn2_geometry

lib = L.lib()
dev = torch.device("cuda", 0)
xyz_np, _ = bench.synth(8, 32768, 0, os.environ.get("GSPN_GATHER_KIND", "U"))
xyz = torch.from_numpy(xyz_np).to(dev)
G = pn2_geometry(xyz)
gen = torch.Generator(device=dev).manual_seed(3)

b = 8
res = []


def t(name, nbytes, fn):
    us = bench._ev_time(fn) * 1e3
    res.append((name, us, nbytes))
    print("%-58s %7.1f us   %6.1f MB  -> %5.2f TB/s" % (name, us, nbytes / 1e6, nbytes / us / 1e6), flush=True)


# ---- pre-aggregated forward / dY kernels ----
for (name, sa, fp, n_src, rows, cout, T, side_n) in (("SA2", G["sa"][1], None, 2048, 8 * 512 * 32, 64, 1, 3), ("SA3", G["sa"][2], None, 512, 8 * 128 * 32, 128, 1, 3),
                                                   ("FP3", None, G["fp"][2], 2048, 8 * 32768, 64, 3, 3)):
    F = torch.randn(b * n_src, cout, device=dev, generator=gen)
    Ws = torch.randn(4, cout, device=dev, generator=gen)
    bias = torch.randn(cout, device=dev, generator=gen)
    Y = torch.empty(rows, cout, device=dev)
    nparts = int(lib.gspn_preagg_fwd_parts(rows, cout))
    stats = torch.empty(nparts * 2 * cout, device=dev)
    if T == 1:
        idx, w, psr, pss, side, side_ld = sa.gidx, None, 0, 0, sa.rel, 4
    else:
        idx, w, psr, pss = fp.idx, fp.weight, 32768, n_src
        side, side_ld = torch.rand(rows, 3, device=dev, generator=gen), 3
    t("preagg_fwd %s rows=%d cout=%d T=%d" % (name, rows, cout, T), 4.0 * rows * cout + 4.0 * rows * (2 * T + side_n),
      lambda: L.check(lib.gspn_preagg_fwd(rows, cout, T, L.ptr(F), L.ptr(idx), L.ptr(w), psr, pss, L.ptr(side), side_ld, side_n, L.ptr(Ws), L.ptr(bias),
                                          L.ptr(Y), L.ptr(stats), L.stream()), "preagg_fwd"))
    dZ = torch.randn(rows, cout, device=dev, generator=gen)
    a = L.DyArgs()
    vec = [torch.rand(cout, device=dev, generator=gen) for _ in range(5)]
    a.Y, a.ldy, a.dZ, a.ldz, a.dPool, a.pool_arg, a.ns = Y.data_ptr(), cout, dZ.data_ptr(), cout, None, None, 0
    a.scale, a.shift, a.cA, a.cB, a.cC = (v.data_ptr() for v in vec)
    dY = torch.empty(rows, cout, device=dev)
    part = torch.empty(int(lib.gspn_preagg_part_floats(cout, side_n)), device=dev)
    nsl = ctypes.c_int(0)
    t("preagg_bwd_dy %s" % name, 12.0 * rows * cout,
      lambda: L.check(lib.gspn_preagg_bwd_dy(rows, cout, ctypes.byref(a), L.ptr(side), side_ld, side_n, L.ptr(dY), L.ptr(part), None, ctypes.byref(nsl), L.stream()), "bwd_dy"))
    if T == 1:
        n, m, ns = n_src, sa.idx.shape[1], sa.idx.shape[2]
        gp = torch.empty(b, n, cout, device=dev)
        t("sa_group_concat_grad_csr %s (n=%d, c=%d, lists of %.1f)" % (name, n, cout, m * ns / n), 4.0 * rows * cout + 4.0 * b * n * cout + 4.0 * rows,
          lambda: L.check(lib.gspn_sa_group_concat_grad_csr(b, n, cout, m, ns, L.ptr(sa.order), L.ptr(sa.offsets), 0, cout, L.ptr(dY), L.ptr(gp), L.stream()), "sa_csr"))
        res.append(("out:sa_csr_" + name, gp.clone(), None))
    else:
        n1, m = 32768, n_src
        g2 = torch.empty(b, m, cout, device=dev)
        t("fp_concat_grad_csr FP3 pre-aggregated (n=%d, m=%d, c=%d, lists of %.0f)" % (n1, m, cout, 3.0 * n1 / m), 4.0 * rows * cout + 4.0 * b * m * cout + 8.0 * 3 * rows,
          lambda: L.check(lib.gspn_fp_concat_grad_csr(b, n1, m, cout, 0, cout, L.ptr(dY), L.ptr(fp.order), L.ptr(fp.offsets), L.ptr(fp.weight), L.ptr(g2), None, L.stream()), "fp_csr"))
        res.append(("out:fp_csr_FP3", g2.clone(), None))
    res.append(("out:Y_" + name, Y.clone(), None))
    res.append(("out:dY_" + name, dY.clone(), None))
    res.append(("out:stats_" + name, stats.clone(), None))

# ---- the un-aggregated FP levels (fp_concat's gradient: c2 = 256, a skip link of c1 columns copied in the same launch) ----
for (name, fp, n1, m, c2, c1) in (("FP2", G["fp"][1], 2048, 512, 256, 64), ("FP1", G["fp"][0], 512, 128, 256, 128)):
    ld = c2 + c1
    g = torch.randn(b * n1, ld, device=dev, generator=gen)
    g2 = torch.empty(b, m, c2, device=dev)
    g1 = torch.empty(b, n1, c1, device=dev)
    t("fp_concat_grad_csr %s (n=%d, m=%d, c2=%d, c1=%d)" % (name, n1, m, c2, c1), 4.0 * b * n1 * ld + 4.0 * b * (m * c2 + n1 * c1),
      lambda: L.check(lib.gspn_fp_concat_grad_csr(b, n1, m, c2, c1, ld, L.ptr(g), L.ptr(fp.order), L.ptr(fp.offsets), L.ptr(fp.weight), L.ptr(g2), L.ptr(g1), L.stream()), "fp_csr"))
    res.append(("out:fp_csr_" + name, torch.cat([g2.reshape(-1), g1.reshape(-1)]), None))

if len(sys.argv) > 1:
    import hashlib
    import json
    path = sys.argv[1]            # a JSON of sha256 digests: written when absent (keep it under tools/ so that it travels), compared when present
    outs = {k: hashlib.sha256(v.cpu().numpy().tobytes()).hexdigest() for (k, v, z) in res if k.startswith("out:")}
    if os.path.exists(path):
        old = json.load(open(path))
        for k, v in outs.items():
            print("%-24s %s" % (k, "bit-identical to the saved run" if v == old.get(k) else "DIFFERS from the saved run"))
    else:
        json.dump(outs, open(path, "w"), indent=1)
        print("saved", path)
