"""CPU suite 3: the N>1 path -- scene sharding + one flat-bucket gradient all-reduce -- with gloo, world_size 2."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from gspn_amd import parallel
    r, local, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    lo, hi = parallel.shard_range(16, r, w)           # 16 scenes over 2 ranks
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(6, 32)), torch.nn.Parameter(torch.randn(32))]    # identical init on every rank
    scenes = torch.arange(16.)
    # per-rank gradient = mean over ITS scenes of a per-scene gradient field
    local_g = [sum((s + 1) * torch.ones_like(p) for s in scenes[lo:hi]) / (hi - lo) for p in ps]
    for p, g in zip(ps, local_g):
        p.grad = g.clone()
    bk = parallel.FlatGradBucket(ps)
    bk.all_reduce_mean()
    # equal shard sizes: mean of rank means == global mean over all 16 scenes
    expect = float((scenes + 1).mean())
    ok = all(torch.allclose(p.grad, torch.full_like(p, expect)) for p in ps)
    # the form bench.py uses: SUM all-reduce, the 1/world scale is applied later (inside the Adam kernel on the GPU)
    for p, g in zip(ps, local_g):
        p.grad = g.clone()
    bk.flatten()
    bk.all_reduce(average=False)
    ok = ok and all(torch.allclose(p.grad / w, torch.full_like(p, expect)) for p in ps)
    out[rank] = (ok, lo, hi)
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_flat_bucket_allreduce():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert out[0][0] and out[1][0]
    assert (out[0][1], out[0][2], out[1][1], out[1][2]) == (0, 8, 8, 16)


def _worker4(rank, world, port, out):
    """world 4, 10 scenes -> unequal shards (3,3,2,2): a scene-weighted mean needs the SUM form with per-scene gradient sums; and the
    optional SyncBN (global-batch statistics) must reproduce a single-process batch norm over the concatenated batch."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from gspn_amd import parallel
    r, local, w = parallel.init_from_env(backend="gloo")
    total = 10
    lo, hi = parallel.shard_range(total, r, w)
    sizes = [parallel.shard_range(total, q, w) for q in range(w)]
    ok = [b - a for a, b in sizes] == [3, 3, 2, 2] and sizes[0][0] == 0 and sizes[-1][1] == total
    ok = ok and all(sizes[q][1] == sizes[q + 1][0] for q in range(w - 1))
    # SUM all-reduce of per-scene gradient SUMS, divided by the global scene count: exact global mean with unequal shards
    p = torch.nn.Parameter(torch.zeros(5))
    p.grad = sum((s + 1.0) * torch.ones(5) for s in range(lo, hi))
    bk = parallel.FlatGradBucket([p])
    bk.flatten()
    bk.all_reduce(average=False)
    ok = ok and torch.allclose(p.grad / total, torch.full((5,), sum(range(1, total + 1)) / total))
    # ---- SyncBN: rows_per_scene rows per scene, c channels; reference = one process over all scenes ----
    g = torch.Generator().manual_seed(123)
    rows_per_scene, cin, c = 7, 4, 6
    x_all = torch.randn(total * rows_per_scene, cin, generator=g, dtype=torch.float64)
    wgt = torch.randn(cin, c, generator=g, dtype=torch.float64)
    gamma0 = torch.rand(c, generator=g, dtype=torch.float64) + 0.5
    beta0 = torch.rand(c, generator=g, dtype=torch.float64) - 0.5
    coef = torch.randn(total * rows_per_scene, c, generator=g, dtype=torch.float64)

    def run(x, co, sync):
        x = x.clone().requires_grad_(True)
        W = wgt.clone().requires_grad_(True)
        ga, be = gamma0.clone().requires_grad_(True), beta0.clone().requires_grad_(True)
        mm, mv = torch.zeros(c, dtype=torch.float64), torch.ones(c, dtype=torch.float64)
        y = x @ W
        if sync:
            z = parallel.sync_bn_relu(y, ga, be, mm, mv, 0.7, 1e-3)
        else:
            mean = y.mean(0)
            var = ((y - mean) ** 2).mean(0)
            inv = torch.rsqrt(var + 1e-3) * ga
            z = torch.relu(y * inv + (be - mean * inv))
            mm = mm * 0.7 + mean.detach() * 0.3
            mv = mv * 0.7 + var.detach() * 0.3
        (z * co).sum().backward()                      # loss = SUM over rows: the ranks' losses add up to the global loss
        return z.detach(), x.grad, W.grad, ga.grad, be.grad, mm, mv

    sl = slice(lo * rows_per_scene, hi * rows_per_scene)
    z, dx, dW, dga, dbe, mm, mv = run(x_all[sl], coef[sl], True)
    flat = torch.cat([dW.reshape(-1), dga, dbe])
    dist.all_reduce(flat)                              # what the gradient bucket does
    rz, rdx, rdW, rdga, rdbe, rmm, rmv = run(x_all, coef, False)
    ok = ok and torch.allclose(z, rz[sl], rtol=1e-10, atol=1e-12) and torch.allclose(dx, rdx[sl], rtol=1e-9, atol=1e-11)
    ok = ok and torch.allclose(flat, torch.cat([rdW.reshape(-1), rdga, rdbe]), rtol=1e-9, atol=1e-11)
    ok = ok and torch.allclose(mm, rmm, rtol=1e-12) and torch.allclose(mv, rmv, rtol=1e-10)
    out[rank] = (bool(ok), lo, hi)
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world4_unequal_shards_and_sync_bn():
    world = 4
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker4, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all(out[r][0] for r in range(world)), dict(out)
    assert [(out[r][1], out[r][2]) for r in range(world)] == [(0, 3), (3, 6), (6, 8), (8, 10)]


def _worker_forced(rank, world, port, out):
    """world 1 with force=True: the process group exists and the bucket's all-reduce really calls the backend (identity result)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    from gspn_amd import parallel
    r, local, w = parallel.init_from_env(backend="gloo", force=True)
    calls = []
    real = dist.all_reduce

    def spy(t, *a, **k):
        calls.append(1)
        return real(t, *a, **k)
    dist.all_reduce = spy
    p = torch.nn.Parameter(torch.zeros(7))
    p.grad = torch.arange(7.)
    bk = parallel.FlatGradBucket([p])
    bk.flatten()
    bk.all_reduce(average=False)
    n_unforced = len(calls)
    bk.all_reduce(average=False, force=True)
    bk.all_reduce(average=True, force=True)
    out[0] = (dist.is_initialized(), w, n_unforced, len(calls), bool(torch.equal(bk.flat, torch.arange(7.))))
    dist.all_reduce = real
    dist.destroy_process_group()


def test_forced_collective_at_world_1():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_forced, args=(1, _free_port(), out), nprocs=1, join=True)
    assert out[0] == (True, 1, 0, 2, True)


def _worker_mv(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from gspn_amd import parallel
    parallel.init_from_env(backend="gloo")
    mm = torch.full((5,), float(rank + 1))                 # rank-dependent moving statistics (per-replica batch norm drifts like this)
    mv = torch.arange(3.) * (rank + 1)
    parallel.average_moving_statistics([mm, mv])
    out[rank] = (mm.tolist(), mv.tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_moving_statistics_are_averaged_over_ranks():
    """r04: per-replica batch norm leaves different moving statistics on every rank; average_moving_statistics puts
    the rank mean on all of them (one flat all-reduce) before evaluation / checkpointing"""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_mv, args=(world, _free_port(), out), nprocs=world, join=True)
    assert out[0] == out[1]
    assert out[0][0] == [1.5] * 5 and out[0][1] == [0.0, 1.5, 3.0]


def test_init_from_env_needs_a_master_port_for_several_ranks(monkeypatch):
    """without MASTER_PORT every rank of a manual launch would pick a port of its own and hang in the rendezvous"""
    import pytest
    from gspn_amd import parallel
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.delenv("MASTER_PORT", raising=False)
    with pytest.raises(RuntimeError, match="MASTER_PORT"):
        parallel.init_from_env(backend="gloo")
