"""The N>1 path's collective, executed on ONE GPU: a one-rank RCCL communicator
(`init_process_group("nccl", world_size=1)`), `FlatGradBucket.all_reduce(force=True)` really calling RCCL, and a captured training
step + all-reduce + FlatAdam equal, bit for bit, to the same step without the collective -- with the all-reduce issued after the
graph replay (the default placement) and captured as the graph's last node.  Runs in a child process so that the process group and
RCCL's threads never touch the rest of the suite."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent(r'''
    import os, sys, json
    sys.path.insert(0, %(root)r)
    import numpy as np
    import torch
    import torch.distributed as dist
    from gspn_amd import parallel, tf_util
    from gspn_amd.fea_extractor import pn2_fea_extractor, pn2_geometry
    from gspn_amd.graph import CapturedStep

    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1")
    rank, local, world = parallel.init_from_env(force=True)
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    dev = torch.device("cuda", 0)
    out = {}

    # (1) the bucket's all-reduce really reaches RCCL: count the calls, compare bits
    calls = []
    real = dist.all_reduce
    def spy(t, *a, **k):
        calls.append(t.data_ptr())
        return real(t, *a, **k)
    dist.all_reduce = spy
    ps = [torch.nn.Parameter(torch.randn(6, 32, device=dev)), torch.nn.Parameter(torch.randn(32, device=dev))]
    for p in ps:
        p.grad = torch.randn_like(p)
    bk = parallel.FlatGradBucket(ps)
    bk.flatten()
    before = bk.flat.clone()
    bk.all_reduce(average=False)                       # world 1, not forced: no collective
    assert calls == []
    bk.all_reduce(average=False, force=True)
    torch.cuda.synchronize()
    assert calls == [bk.flat.data_ptr()]
    assert torch.equal(bk.flat, before)                # SUM over one rank is the identity, bit for bit
    bk.all_reduce(average=True, force=True)
    torch.cuda.synchronize()
    assert torch.equal(bk.flat, before)                # ... and averaging over one rank divides by nothing
    out["forced_calls"] = len(calls)

    # (2) one captured step of the real network (2 scenes x 4096 points): no collective / all-reduce after the replay / all-reduce
    #     captured as the last node.  Parameters after two Adam steps must agree bit for bit.
    xyz = torch.from_numpy(np.random.default_rng(1).random((2, 4096, 3), dtype=np.float32)).to(dev)
    col = torch.rand(2, 4096, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    gout = torch.randn(2, 4096, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) / (2 * 4096 * 64)

    def run(mode):
        store = tf_util.set_variable_store(tf_util.VariableStore(device=dev, seed=99))
        st = {"bucket": None, "opt": None}
        G = pn2_geometry(xyz)

        def fwd_bwd():
            o = pn2_fea_extractor(xyz, col, 'fea', True, 0.5, geometry=G)
            (o * gout).sum().backward()
            if st["bucket"] is None:
                st["bucket"] = parallel.FlatGradBucket(store.parameters())
                st["opt"] = parallel.FlatAdam(st["bucket"], lr=1e-3)
            st["bucket"].flatten()

        def finish(skip_collective=False):
            if mode != "none" and not skip_collective:
                st["bucket"].all_reduce(average=False, force=True)
            st["opt"].step(grad_scale=1.0)

        fwd_bwd()
        finish()

        def captured():
            st["opt"].zero_grad(set_to_none=True)
            fwd_bwd()
            if mode == "in_graph":
                st["bucket"].all_reduce(average=False, force=True)

        n0 = len(calls)
        g = CapturedStep(captured)
        ncap = len(calls) - n0
        for _ in range(2):
            g.replay()
            finish(skip_collective=(mode == "in_graph"))
        torch.cuda.synchronize()
        return st["opt"].flat.clone(), st["bucket"].flat.clone(), ncap

    p_none, g_none, _ = run("none")
    p_after, g_after, _ = run("after")
    assert torch.isfinite(p_none).all() and float(g_none.abs().max()) > 0
    assert torch.equal(p_after, p_none) and torch.equal(g_after, g_none)
    out["after_graph_bit_equal"] = True
    try:
        p_in, g_in, ncap = run("in_graph")
        out["in_graph_captured_calls"] = ncap
        out["in_graph_bit_equal"] = bool(torch.equal(p_in, p_none) and torch.equal(g_in, g_none))
    except Exception as e:                              # capture of a collective is a property of the RCCL / torch build: report, do not hide
        out["in_graph_error"] = repr(e)[:300]
    print("RESULT " + json.dumps(out), flush=True)
    try:
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        print("teardown:", repr(e)[:200], flush=True)
''')


def test_rccl_path_at_world_1_equals_the_no_collective_step():
    import json
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("MASTER_PORT", None)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    assert res["forced_calls"] == 2
    assert res["after_graph_bit_equal"] is True
    # the captured form: either it works and is bit-equal too, or the build refuses to capture a collective (then the error is shown)
    if "in_graph_error" not in res:
        assert res["in_graph_bit_equal"] is True


def test_two_ranks_rehearsal_on_one_gpu_over_gloo():
    """The N > 1 control flow of bench.py on a box with ONE GPU: two ranks launched by torch.distributed.run exactly as the driver launches
    them, both on device 0, gloo carrying the bucket through the host (RCCL refuses two ranks on one device).  Slow, but the ranks run the
    real step, the real stream tests (every probe collective and the agreed verdict on every rank) and the real per-step sequence of
    collectives; a mismatch would hang (hence the timeout).  The JSON line must be the LAST line of stdout and report both ranks."""
    import json
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.update(GSPN_DIST_BACKEND="gloo", GSPN_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-extra"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    line = json.loads(lines[-1])                        # the line is the last thing on stdout
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 16 and line["value"] > 0
    assert len(lines[-1]) < 4096 and line["detail_file"] == "bench_detail.json"      # the compact line; everything else is in the detail file
    detail = json.load(open(os.path.join(ROOT, line["detail_file"])))
    assert detail["collective"]["world"] == 2 and detail["n_gpus"] == 2


SYNC_CHILD = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
from gspn_amd import mlp as M
from gspn_amd import parallel
from gspn_amd.mlp import LayerParams, mlp_stack

rank, local, world = parallel.init_from_env()
dev = torch.device("cuda", local)
res = {}


def layers_of(chans, cin, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for c in chans:
        w = (torch.randn(cin, c, generator=g) * (1.0 / cin ** 0.5)).to(dev).requires_grad_(True)
        b = (torch.randn(c, generator=g) * 0.1).to(dev).requires_grad_(True)
        be = (torch.randn(c, generator=g) * 0.1).to(dev).requires_grad_(True)
        ga = (torch.rand(c, generator=g) + 0.5).to(dev).requires_grad_(True)
        out.append(LayerParams(w, b, True, be, ga, torch.zeros(c, device=dev), torch.ones(c, device=dev)))
        cin = c
    return out


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


for name, rows, cin, chans, ns in (("pooled", 2 * 8192, 64, [64, 64, 128], 32), ("dense", 2 * 4096, 32, [64, 32], None), ("dense1", 2 * 2048, 24, [48], None)):
    g = torch.Generator().manual_seed(5)
    X = torch.randn(rows, cin, generator=g)
    half = rows // world
    go_rows = rows // ns if ns else rows
    GO = torch.randn(go_rows, chans[-1], generator=g)
    gh = go_rows // world
    # ---- this rank's shard, BN statistics over both ranks (fused SyncBN) ----
    M.SYNC_BN = True
    lay = layers_of(chans, cin, 9)
    x = X[rank * half:(rank + 1) * half].to(dev).requires_grad_(True)
    out = mlp_stack(x, cin, lay, True, 0.6, pool_ns=ns)
    (out * GO[rank * gh:(rank + 1) * gh].to(dev)).sum().backward()
    flat = torch.cat([t.grad.reshape(-1) for lp in lay for t in lp.tensors()])
    dist.all_reduce(flat)                                # the gradient bucket's SUM all-reduce
    mv = torch.cat([torch.cat([lp.moving_mean, lp.moving_variance]) for lp in lay])
    # ---- the whole batch on one rank, plain fused stack ----
    M.SYNC_BN = False
    ref = layers_of(chans, cin, 9)
    xr = X.to(dev).requires_grad_(True)
    outr = mlp_stack(xr, cin, ref, True, 0.6, pool_ns=ns)
    (outr * GO.to(dev)).sum().backward()
    flatr = torch.cat([t.grad.reshape(-1) for lp in ref for t in lp.tensors()])
    mvr = torch.cat([torch.cat([lp.moving_mean, lp.moving_variance]) for lp in ref])
    res[name] = {"out": rel(out.detach(), outr.detach()[rank * gh:(rank + 1) * gh]), "dx": rel(x.grad, xr.grad[rank * half:(rank + 1) * half]),
                 "params": rel(flat, flatr), "moving": rel(mv, mvr)}
torch.cuda.synchronize()
gathered = [None] * world
dist.all_gather_object(gathered, res)
if rank == 0:
    print("RESULT " + json.dumps(gathered), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def test_fused_sync_bn_two_ranks_equal_the_whole_batch_on_one():
    """r04: SyncBN on the fused kernels (mlp.SYNC_BN_FUSED: the per-workgroup partial rows of every BN reduction all-reduced between the
    producing and the summing kernel).  Two ranks on the one GPU (gloo), each with half of the rows: outputs, input gradients, the
    bucket-summed parameter gradients and the moving statistics equal those of ONE process running the plain fused stack on the whole
    batch -- the reference's single-GPU batch norm (tf_util.py:529-534).  Pooled stack, dense stack, one-layer dense stack."""
    import json
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.update(GSPN_DIST_BACKEND="gloo", GSPN_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    script = os.path.join(ROOT, "gpurun_out", "_sync_child.py")
    os.makedirs(os.path.dirname(script), exist_ok=True)
    with open(script, "w") as f:
        f.write(SYNC_CHILD % {"root": ROOT})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port), script]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert len(res) == 2
    for per_rank in res:
        for name, e in per_rank.items():
            assert e["out"] < 1e-5 and e["moving"] < 1e-5, (name, e)
            assert e["dx"] < 1e-4 and e["params"] < 1e-4, (name, e)
