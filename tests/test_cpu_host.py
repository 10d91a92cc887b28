"""CPU suite 2: the C-ABI library loads and exports everything include/gspn_hip.h declares; host-side logic
(argument validation, ball-query threshold search, variable scopes/initialisers, sharding, flat bucket) without a GPU."""
import ctypes
import math
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "gspn_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|float|long)\s+(gspn_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from gspn_amd import _lib
    from gspn_amd import build
    build.build()
    h = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 28
    for s in syms:
        assert hasattr(h, s), "libgspn_hip.so does not export %s" % s
    bound = set(_lib.SIGNATURES) | set(_lib.SPECIAL)
    assert bound == set(syms), "binding table and header disagree: %s" % (bound ^ set(syms))
    lib = _lib.lib()
    hdr = int(re.search(r"#define GSPN_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "gspn_hip.h")).read()).group(1))
    assert lib.gspn_abi_version() == hdr == _lib.ABI_VERSION
    assert lib.gspn_dist_policy() == 2        # same contraction policy as the oracle


def test_missing_library_fails_loudly(monkeypatch):
    from gspn_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgspn_hip.so")
    with pytest.raises(_lib.GspnHipError):
        _lib.lib()


def test_ball_threshold_is_exact():
    """s < T  <=>  max(sqrtf(s),1e-20f) < radius for floats around the boundary (tf_grouping_g.cu:27-28)"""
    from gspn_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(0)
    radii = list(rng.random(200).astype(np.float32) * 2) + [np.float32(0.1), np.float32(0.2), np.float32(1e-19), np.float32(3e19), np.float32(1e-10)]
    for r in radii:
        r = np.float32(r)
        T = np.float32(lib.gspn_ball_threshold(ctypes.c_float(float(r))))
        cand = [T]
        lo, hi = T, T
        for _ in range(4):
            lo = np.nextafter(lo, np.float32(0), dtype=np.float32)
            hi = np.nextafter(hi, np.float32(np.inf), dtype=np.float32)
            cand += [lo, hi]
        for s in cand:
            ref = max(np.sqrt(np.float32(s), dtype=np.float32), np.float32(1e-20)) < r
            assert (np.float32(s) < T) == bool(ref), (r, s, T)
    assert lib.gspn_ball_threshold(ctypes.c_float(1e-21)) == 0.0
    assert math.isinf(lib.gspn_ball_threshold(ctypes.c_float(float("inf"))))


def test_argument_validation_mirrors_op_requires():
    from gspn_amd import _lib
    from gspn_amd.tf_grouping import query_ball_point, select_top_k
    from gspn_amd.tf_sampling import farthest_point_sample
    x = torch.zeros(1, 8, 3)
    with pytest.raises(ValueError):
        farthest_point_sample(0, x)                     # tf_sampling.cpp:99
    with pytest.raises(ValueError):
        query_ball_point(0.0, 4, x, x)                  # tf_grouping.cpp:101
    with pytest.raises(ValueError):
        query_ball_point(0.1, 0, x, x)                  # tf_grouping.cpp:104
    with pytest.raises(ValueError):
        select_top_k(0, torch.zeros(1, 2, 3))           # tf_grouping.cpp:143
    with pytest.raises(_lib.GspnHipError):              # CPU tensors never fall back to a CPU path
        farthest_point_sample(4, x)
    with pytest.raises(ValueError):
        farthest_point_sample(4, torch.zeros(1, 8, 3, dtype=torch.float64))
    # the ABI itself rejects bad sizes without touching the device
    lib = _lib.lib()
    assert lib.gspn_farthestpointsampling(1, 0, 4, None, None, None, None) == -1
    assert lib.gspn_queryballpoint(1, 8, 2, ctypes.c_float(-1.0), 4, None, None, None, None, None) == -1
    assert lib.gspn_mlp_fwd(8, 4, 4, None, 2, None, None, None, None, None, 4, None, None) == -1      # ldx < cin


def test_variable_store_scopes_and_xavier():
    from gspn_amd import tf_util
    store = tf_util.set_variable_store(tf_util.VariableStore(device=torch.device("cpu"), seed=3))
    lp = tf_util._layer_params('conv0', 6, 64, [1, 1, 6, 64], True, 1e-3, None, True)
    assert set(store.vars) == {'conv0/weights', 'conv0/biases', 'conv0/bn/beta', 'conv0/bn/gamma', 'conv0/bn/moving_mean', 'conv0/bn/moving_variance'}
    w = store.vars['conv0/weights']
    lim = math.sqrt(6.0 / (6 + 64))
    assert w.shape == (1, 1, 6, 64) and float(w.abs().max()) <= lim and float(w.abs().max()) > 0.8 * lim
    assert float(store.vars['conv0/biases'].abs().max()) == 0 and float(store.vars['conv0/bn/gamma'].min()) == 1
    assert not store.vars['conv0/bn/moving_mean'].requires_grad and store.vars['conv0/bn/moving_variance'].mean() == 1
    assert lp.weights.shape == (6, 64) and lp.weights.data_ptr() == w.data_ptr()      # the GEMM sees the same storage
    with tf_util.variable_scope('a'):
        with tf_util.variable_scope('b'):
            v = tf_util.get_variable('x', (3,), tf_util.constant_initializer(2.0))
    assert 'a/b/x' in store.vars and v is tf_util.get_variable_store().vars['a/b/x']
    lp2 = tf_util._layer_params('conv0', 6, 64, [1, 1, 6, 64], True, 1e-3, None, True)     # reuse, not re-create
    assert lp2.weights.data_ptr() == w.data_ptr() and len(store.parameters()) == 5
    with pytest.raises(ValueError):
        tf_util._layer_params('conv0', 7, 64, [1, 1, 7, 64], True, 1e-3, None, True)


def test_shard_range_and_bucket():
    from gspn_amd.parallel import FlatGradBucket, shard_range
    for total in (8, 32, 64, 13):
        for world in (1, 2, 4, 8):
            r = [shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
    ps = [torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(5))]
    ps[0].grad = torch.arange(12.).view(3, 4)
    bk = FlatGradBucket(ps)
    flat = bk.all_reduce_mean()                        # world 1: identity
    assert flat.numel() == 17 and float(flat[:12].sum()) == 66 and float(ps[1].grad.abs().sum()) == 0


def test_preagg_host_side_rules():
    """which first layers the pre-aggregated path takes (host logic only): cout = 4 * 2^k, training-mode BN on the first two layers,
    enough feature columns; the forward's partial-row count is what the finalize call is told to sum"""
    from gspn_amd import _lib, mlp
    lib = _lib.lib()
    assert [c for c in (4, 8, 12, 32, 48, 64, 96, 128, 256, 1024, 2048, 6, 0) if lib.gspn_preagg_ok(c)] == [4, 8, 32, 64, 128, 256, 1024]
    for rows, cout in ((1, 64), (100, 64), (131072, 64), (262144, 64), (32768, 128), (10 ** 7, 32)):
        p = int(lib.gspn_preagg_fwd_parts(rows, cout))
        assert 1 <= p <= 2048 and (p < 64 or p % 8 == 0)
    assert lib.gspn_preagg_fwd_parts(0, 64) < 0 and lib.gspn_preagg_fwd_parts(100, 48) < 0
    assert lib.gspn_preagg_part_floats(64, 3) >= 1024 * 2 * 3 * 64

    def layer(cout, bn=True):
        return mlp.LayerParams(torch.zeros(67, cout), torch.zeros(cout), bn=bn)
    assert mlp.preagg_ok([layer(64), layer(64)], True, 64)
    assert not mlp.preagg_ok([layer(64), layer(64)], False, 64)            # inference: the coefficients of backward do not exist
    assert not mlp.preagg_ok([layer(64)], True, 64)                        # the second layer's pass B hands the first its coefficients
    assert not mlp.preagg_ok([layer(64, bn=False), layer(64)], True, 64)
    assert not mlp.preagg_ok([layer(48), layer(64)], True, 64)
    assert not mlp.preagg_ok([layer(64), layer(64)], True, 3)              # SA level 1: three colour channels, the gathered GEMM is as cheap


def test_async_status_words_raise_at_the_next_check():
    """_lib.check_async: a registered status word whose copy has completed with a non-zero value raises GspnHipError (the multi-CU
    FPS's expired-wait flag); pending ones are kept, zero ones dropped"""
    import torch
    from gspn_amd import _lib

    class Ev:
        def __init__(self, done):
            self.done = done

        def query(self):
            return self.done

        def synchronize(self):
            self.done = True

    _lib._async_status[:] = []
    ok, pend, bad = torch.zeros(1, dtype=torch.int32), torch.ones(1, dtype=torch.int32), torch.ones(1, dtype=torch.int32)
    _lib.register_async_status(ok, Ev(True), "ok")
    _lib.register_async_status(pend, Ev(False), "pending")
    _lib.check_async()                                   # nothing completed-and-bad yet
    assert len(_lib._async_status) == 1
    _lib.register_async_status(bad, Ev(True), "farthest_point_sample(multi-CU, test)")
    with pytest.raises(_lib.GspnHipError, match="multi-CU, test"):
        _lib.check_async()
    assert len(_lib._async_status) == 1                  # the pending one is still watched
    with pytest.raises(_lib.GspnHipError, match="pending"):
        _lib.check_async(block=True)
    assert _lib._async_status == []


def test_bench_line_is_compact_strict_json_with_the_contract_keys():
    """the stdout line is ONE compact JSON object (< 4 KB, fixed key set, strict JSON) whatever the full result holds;
    r04's 23 KB line was not parsed by the driver."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    long = "prose " * 2000
    res = {"metric": bench.METRIC, "value": 4586.123456789, "unit": "scenes/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 1.7444456,
           "median_ms_per_step": 1.718, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic (U: " + long + ")",
           "config": {"workload": "BASELINE configs[2]: " + long, "scenes_per_gpu": 8, "global_batch": 8, "npoints": 32768, "parallelism": "dp1", "schedule": long,
                      "another": long},
           "roofline": {"bound": "hbm", "kernel": "fps_cell_kernel<32,true>", "achieved": float("nan"), "peak": 8000.0, "unit": "GB/s", "frac": float("inf"),
                        "traffic": 4.39e6, "algorithmic_bytes_per_launch": 1.0732e10, "avg_launch_ms": 1.793, "launches_timed": 24, "concurrent_launches": 2,
                        "achieved_is": long, "us_per_pick": 0.9},
           "roofline_ops": {"ops": [{"op": long, "bound_by": long}] * 40}, "data_kinds": {"S": long}, "reference_harness": long, "other_configs": {"x": long},
           "cpu_baseline": {"value": 1.317, "unit": "scenes/s", "cores": 1, "kind": "port", "sample": long,
                            "all_cores": {"value": 3.07, "cores": 256, "omp_threads": 256, "note": long}}}
    out = bench.compact_line(res)
    line = json.dumps(out, allow_nan=False)                   # strict: NaN / Infinity would raise
    assert "\n" not in line and len(line) < 4096
    back = json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    assert set(back) == set(bench.LINE_KEYS)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in back
    assert set(back["config"]) == set(bench.CONFIG_KEYS) and "model" not in back["config"]
    assert set(back["roofline"]) == set(bench.ROOFLINE_KEYS)
    assert set(back["cpu_baseline"]) == set(bench.CPU_KEYS) and set(back["cpu_baseline"]["all_cores"]) == {"value", "cores"}
    assert back["roofline"]["achieved"] is None and back["roofline"]["frac"] is None       # non-finite numbers become null, not NaN
    assert back["value"] == pytest.approx(4586.12, rel=1e-5) and back["detail_file"] == "bench_detail.json"
    # no CPU baseline (N > 1 runs): the key is there, null
    res.pop("cpu_baseline")
    assert bench.compact_line(res)["cpu_baseline"] is None
    assert back["collective"] is None and "traffic_source" in back["roofline"] and "geometry_pair_phase" in back["config"]
    # N > 1: the collective's cost inside the step and the value against a kept one-GPU line ride on the line
    res.update(n_gpus=4, value=4 * 4400.0, collective={"allreduce_ms_in_step": 0.031, "ms_per_step_without_collective": 1.75, "bucket_bytes": 1069056, "note": long})
    res["collective"].update(bench.n1_reference(res["value"], 4))
    c = bench.compact_line(res)["collective"]
    assert set(c) == set(bench.COLLECTIVE_KEYS) and c["allreduce_ms_in_step"] == pytest.approx(0.031)
    if c["n1_value"] is not None:
        assert c["scaling_efficiency_vs_n1"] == pytest.approx(4 * 4400.0 / (4 * c["n1_value"]), rel=1e-5) and c["n1_source"].startswith("profiles/")
