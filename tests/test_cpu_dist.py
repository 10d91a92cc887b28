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
