"""Generates tests/golden/*.npz.  Run in the BUILD container (needs /root/reference for oracle/_ref):
    python tests/make_golden.py
 * interp_ref_*.npz : inputs + outputs of the REFERENCE's own compiled code (oracle/_ref/libinterp_ref.so, built
   from tf_ops/3d_interpolation/interpolate.cpp) for three_interpolate / three_interpolate_grad, on the demo shape
   of tf_interpolate.py:39-48 (seed 100, (32,128,64) -> (32,512,64)) and the op-test shape of
   tf_interpolate_op_test.py:11-16 ((1,8,16) -> (1,128,16), weights 1/3).
 * threenn_ref_*.npz / nnsearch_ref_*.npz : inputs + outputs of the REFERENCE's own compiled `threenn_cpu`
   (tf_interpolate.cpp:60-103) and `nnsearch` (tf_nndistance.cpp:21-43; both directions as NnDistanceOp::Compute runs it,
   :79-80) from oracle/_ref/libslices_ref.so (oracle/Makefile: slices) -- on the demo seeds (tf_interpolate.py:39-48 seed 100;
   tf_nndistance_cpu.py:29 seed 0), on integer lattices where ties decide every answer, with duplicated points, and with
   fewer than three candidates.
 * oracle_*.npz : small seeded cases of every op produced by the C oracle (pins the oracle against accidental
   edits and gives the GPU suite fixed vectors); the reference itself ships no golden vector for these.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests import data as D  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def interp_ref():
    assert O.ref_lib() is not None, "needs /root/reference (oracle/_ref)"
    # demo shape, tf_interpolate.py:39-48 (np.random.seed(100)); cut to 2 batches to keep the fixture small
    np.random.seed(100)
    pts = np.random.random((32, 128, 64)).astype('float32')[:2]
    tmp1 = np.random.random((32, 512, 3)).astype('float32')[:2]
    tmp2 = np.random.random((32, 128, 3)).astype('float32')[:2]
    _, idx = O.three_nn(tmp1, tmp2)
    w = np.ones_like(tmp1) / 3.0
    out = O.ref_three_interpolate(pts, idx, w)
    g = np.random.random(out.shape).astype('float32')
    gp = O.ref_three_interpolate_grad(pts, idx, w, g)
    np.savez_compressed(os.path.join(OUT, "interp_ref_demo.npz"), points=pts, idx=idx, weight=w, out=out, grad_out=g, grad_points=gp)
    # op-test shape, tf_interpolate_op_test.py:11-16
    rng = np.random.default_rng(0)
    pts = rng.random((1, 8, 16)).astype('float32')
    x1 = rng.random((1, 128, 3)).astype('float32')
    x2 = rng.random((1, 8, 3)).astype('float32')
    dist, idx = O.three_nn(x1, x2)
    w = (np.ones_like(dist) / 3.0).astype('float32')
    out = O.ref_three_interpolate(pts, idx, w)
    g = rng.random(out.shape).astype('float32')
    gp = O.ref_three_interpolate_grad(pts, idx, w, g)
    # inverse-squared-distance weights of pointnet_util.py:157-160 through the same reference loops
    d = np.maximum(dist, 1e-10)
    w2 = ((1.0 / d) / (1.0 / d).sum(2, keepdims=True)).astype('float32')
    out2 = O.ref_three_interpolate(pts, idx, w2)
    np.savez_compressed(os.path.join(OUT, "interp_ref_optest.npz"), points=pts, idx=idx, weight=w, out=out, grad_out=g, grad_points=gp,
                        weight2=w2, out2=out2)


def _lattice(rng, b, n, side):
    """integer lattice points (exactly representable squared distances: every contraction form agrees, ties everywhere)"""
    return rng.integers(0, side, size=(b, n, 3)).astype('float32')


def slices_ref():
    assert O.slices_lib() is not None, "needs /root/reference (oracle/_ref/libslices_ref.so)"
    out = {}
    # demo shape of tf_interpolate.py:39-48 (np.random.seed(100)): 512 dense <- 128 sparse, first 2 batches
    np.random.seed(100)
    _ = np.random.random((32, 128, 64))
    x1 = np.random.random((32, 512, 3)).astype('float32')[:2]
    x2 = np.random.random((32, 128, 3)).astype('float32')[:2]
    rng = np.random.default_rng(7)
    cases = {"demo": (x1, x2),
             "lattice": (_lattice(rng, 2, 300, 5), _lattice(rng, 2, 90, 5)),          # ties: the strict-'<' cascade keeps ascending k
             "dups": (D.batch("D", 2, 400, 3), D.batch("D", 2, 400, 3)[:, ::4].copy()),  # duplicated points, queries ON known points
             "m2": (D.batch("U", 1, 10, 1), D.batch("U", 1, 2, 2)),                     # m < 3: the tail stays (float(1e40) = inf, 0)
             "m1": (D.batch("U", 1, 5, 3), D.batch("U", 1, 1, 4)),
             "fp_level": (D.batch("U", 1, 2048, 11), D.batch("U", 1, 2048, 11)[:, ::4].copy()),   # the 2048 <- 512 feature-propagation level
             "grid": (D.batch("S", 1, 4096, 31), D.batch("S", 1, 4096, 31)[:, ::2].copy())}       # 2048 known points: the cell-grid kernel's range, room scene
    for k, (a, b) in cases.items():
        d, i = O.ref_three_nn(a, b)
        out[k + "_xyz1"], out[k + "_xyz2"], out[k + "_dist"], out[k + "_idx"] = a, b, d, i
    np.savez_compressed(os.path.join(OUT, "threenn_ref.npz"), **out)
    out = {}
    np.random.seed(0)                                                       # tf_nndistance_cpu.py:29
    a = np.random.randn(2, 60, 3).astype('float32')
    b = np.random.randn(2, 50, 3).astype('float32')
    cases = {"demo": (a, b),
             "lattice": (_lattice(rng, 2, 200, 4), _lattice(rng, 2, 150, 4)),          # ties: lowest index wins (strict '<', k ascending)
             "single": (D.batch("U", 2, 1, 5), D.batch("U", 2, 7, 6)),
             "ins": (D.batch("U", 3, 512, 21), D.batch("U", 3, 512, 22))}               # the model's Chamfer shape (NUM_POINT_INS = 512)
    for k, (a, b) in cases.items():
        d1, i1, d2, i2 = O.ref_nnsearch(a, b)
        out[k + "_xyz1"], out[k + "_xyz2"] = a, b
        out[k + "_d1"], out[k + "_i1"], out[k + "_d2"], out[k + "_i2"] = d1, i1, d2, i2
    np.savez_compressed(os.path.join(OUT, "nnsearch_ref.npz"), **out)


def oracle_cases():
    # BASELINE configs[0]: 1 scene x 4096 pts, FPS -> 512 + ball query r=0.2 k=32
    xyz = D.batch("U", 1, 4096)
    fps = O.farthest_point_sample(512, xyz)
    new_xyz = O.gather_point(xyz, fps)
    idx, cnt, vis = O.query_ball_point(0.2, 32, xyz, new_xyz, return_visited=True)
    np.savez_compressed(os.path.join(OUT, "oracle_c1_fps_ball.npz"), fps=fps, idx=idx, cnt=cnt, visited=vis)
    # duplicate-heavy cloud: the (k mod 512, k) tie rule decides
    xd = D.batch("D", 2, 3000, 40)
    fd = O.farthest_point_sample(700, xd)
    bi, bc = O.query_ball_point(0.15, 16, xd, O.gather_point(xd, fd))
    np.savez_compressed(os.path.join(OUT, "oracle_dup_fps_ball.npz"), fps=fd, idx=bi, cnt=bc)
    # 3-NN / nn_distance on the demo seeds (tf_nndistance_cpu.py:29 seed 0; tf_interpolate.py:39 seed 100)
    np.random.seed(0)
    a = np.random.randn(2, 60, 3).astype('float32')
    b = np.random.randn(2, 50, 3).astype('float32')
    d1, i1, d2, i2 = O.nn_distance(a, b)
    c1, j1, c2, j2 = O.nn_distance(a, b, cpu_twin=True)
    t_d, t_i = O.three_nn(a, b)
    np.savez_compressed(os.path.join(OUT, "oracle_nn.npz"), a=a, b=b, d1=d1, i1=i1, d2=d2, i2=i2, c1=c1, j1=j1, c2=c2, j2=j2, t_d=t_d, t_i=t_i)


if __name__ == "__main__":
    interp_ref()
    slices_ref()
    oracle_cases()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
