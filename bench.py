#!/usr/bin/env python
"""bench.py -- scenes/s of the set-abstraction hot path on MI355X.

Workload (BASELINE.json configs[2], the one `metric` is quoted on): per GPU a batch of 8 synthetic scenes x
32768 points (xyz ~ U[0,1)^3 + 3 colour channels), one fwd+bwd step of the 3-level SA + 3-level FP stack
(pn2_fea_extractor, models/model_rpointnet.py:209-233), gradient all-reduce over RCCL when N>1, Adam update.
Weak scaling: every rank owns 8 scenes.  Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCENES_PER_GPU = 8
NPOINTS = 32768
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def synth(b, n, seed0):
    xyz = np.stack([np.random.default_rng(seed0 + i).random((n, 3), dtype=np.float32) for i in range(b)])
    col = np.stack([np.random.default_rng(10_000 + seed0 + i).random((n, 3), dtype=np.float32) for i in range(b)])
    return xyz, col


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from gspn_amd import parallel, tf_sampling, tf_util
    from gspn_amd.fea_extractor import pn2_fea_extractor

    rank, local, world = parallel.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    # inputs resident in HBM before the timed region; rank r owns scenes [8r, 8r+8) of the global batch
    xyz_np, col_np = synth(SCENES_PER_GPU, NPOINTS, seed0=rank * SCENES_PER_GPU)
    xyz = torch.from_numpy(xyz_np).to(dev)
    col = torch.from_numpy(col_np).to(dev)
    gout = torch.from_numpy(np.random.default_rng(777).standard_normal((SCENES_PER_GPU, NPOINTS, 64)).astype(np.float32)).to(dev)

    store = tf_util.set_variable_store(tf_util.VariableStore(device=dev, seed=1234))   # same weights on every rank
    state = {"bucket": None, "opt": None}

    def step():
        out = pn2_fea_extractor(xyz, col, 'fea', True, 0.5)
        loss = (out * gout).sum() * (1.0 / out.numel())
        if state["opt"] is not None:
            state["opt"].zero_grad(set_to_none=True)      # backward assigns fresh grads; the bucket re-points them at its slices
        loss.backward()
        if state["bucket"] is None:
            params = store.parameters()
            state["bucket"] = parallel.FlatGradBucket(params)
            state["opt"] = torch.optim.Adam(params, lr=1e-3, foreach=True)
        state["bucket"].all_reduce_mean()            # one flat RCCL all-reduce (no-op at world 1)
        state["opt"].step()
        return loss

    for _ in range(args.warmup):
        step()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    tf_sampling.PROFILE = []          # HIP-event pairs around every FPS launch on its stream
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    prof = tf_sampling.PROFILE
    tf_sampling.PROFILE = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # dominant kernel: FPS of SA level 1 (b=8, n=32768 -> m=2048), live HIP-event average over the timed region
    fps_ms = [e0.elapsed_time(e1) for (e0, e1, b, n, m) in prof if n == NPOINTS]
    b, n, m = SCENES_PER_GPU, NPOINTS, 2048
    alg_bytes = 20.0 * b * (m - 1) * n + 4.0 * b * m            # 20 B/point/round + the index output
    fps_avg_ms = float(np.mean(fps_ms)) if fps_ms else float("nan")
    achieved = alg_bytes / (fps_avg_ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_fps_pmc.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    if rank == 0:
        global_batch = SCENES_PER_GPU * world
        res = {
            "metric": "scenes/sec fwd+bwd set-abstraction, 32768 pts, 1/2/4/8 MI355X",
            "value": global_batch * args.steps / dt,
            "unit": "scenes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: batch 8 x 32768-pt scenes per GPU, 3-level SA + 3-level FP (three_nn/interpolate) fwd+bwd, "
                                   "pn2_fea_extractor layer spec, BN training mode, Adam step", "scenes_per_gpu": SCENES_PER_GPU,
                       "global_batch": global_batch, "npoints": NPOINTS, "parallelism": "dp%d (scenes sharded, one flat RCCL grad all-reduce)" % world},
            "roofline": {"bound": "hbm", "kernel": "fps_resident_kernel<32,true> (SA1: 8 x 32768 -> 2048)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_ms": fps_avg_ms, "launches_timed": len(fps_ms)},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(xyz_np, col_np)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(xyz_np, col_np):
    """the CPU port (oracle geometry + torch-CPU MLP stand-in) on a bounded sample of the same workload"""
    from oracle import cpu_pipeline
    nthr = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        sample = 2
        t1 = cpu_pipeline.run_step(xyz_np[:sample], col_np[:sample], mt=False)
    finally:
        torch.set_num_threads(nthr)
    cores = os.cpu_count() or 1
    tall = cpu_pipeline.run_step(xyz_np, col_np, mt=True)
    return {"value": sample / t1, "unit": "scenes/s", "cores": 1, "kind": "port",
            "sample": "%d of the 8 scenes (32768 pts each), one full fwd+bwd step, single thread: C oracle for FPS/ball/group/3-NN/interp, "
                      "torch-CPU fp32 stand-in for the TensorFlow MLP; %.1f s" % (sample, t1),
            "all_cores": {"value": xyz_np.shape[0] / tall, "cores": cores,
                          "note": "same step on all 8 scenes: OpenMP over scenes for FPS/ball query (<=8 threads), torch intra-op threads for the MLP; %.1f s" % tall}}


if __name__ == "__main__":
    main()
