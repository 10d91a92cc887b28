#!/usr/bin/env python
"""bench.py -- scenes/s of the set-abstraction hot path on MI355X.

Workload (BASELINE.json configs[2], the one `metric` is quoted on): per GPU a batch of 8 synthetic scenes x
32768 points (xyz ~ U[0,1)^3 + 3 colour channels), one fwd+bwd step of the 3-level SA + 3-level FP stack
(pn2_fea_extractor, models/model_rpointnet.py:209-233), gradient all-reduce over RCCL when N>1, Adam update.
Weak scaling: every rank owns 8 scenes.  Prints ONE JSON line on rank 0.

Schedule: the coordinate-only part of a step (3 x FPS + ball query, 3 x 3-NN: gspn_amd/geometry.py) is computed two
batches ahead on two side HIP streams while the MFMA layers of batch k run -- FPS is sequential in npoint and holds one
CU per scene, so it overlaps instead of serialising.  Every timed step still executes one full geometry pass and one full
fwd+bwd+update; three distinct synthetic batches rotate, nothing is cached across steps.  The ~200 launches of the
layers' fwd+bwd are captured once into a hipGraph and replayed (gspn_amd/graph.py); the geometry stream, the gradient
all-reduce and the Adam update stay eager.  --no-overlap runs the geometry inline on the main stream, --no-graph
enqueues kernel by kernel (same kernels, same results either way).

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
    ap.add_argument("--no-overlap", action="store_true", help="geometry inline on the main stream instead of prefetched on a side stream")
    ap.add_argument("--no-graph", action="store_true", help="enqueue the layers kernel by kernel instead of replaying a captured hipGraph")
    args = ap.parse_args()

    from gspn_amd import parallel, tf_sampling, tf_util
    from gspn_amd.fea_extractor import pn2_fea_extractor, pn2_geometry
    from gspn_amd.geometry import GeometryStream
    from gspn_amd.graph import CapturedStep, copy_into

    rank, local, world = parallel.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    # inputs resident in HBM before the timed region; rank r owns scenes [8r, 8r+8) of the global batch; NB distinct batches rotate
    # (NB = 3 slots: the layers of step i read slot i%3 while the geometry of steps i+1 and i+2 is being written into the other two)
    NB = 3
    DEPTH = int(os.environ.get('GSPN_BENCH_DEPTH', '2'))                   # geometry runs this many steps ahead, on DEPTH side streams (FPS throughput: one CU per scene per stream)
    batches = []
    for k in range(NB):
        xyz_np, col_np = synth(SCENES_PER_GPU, NPOINTS, seed0=(k * 1000 + rank) * SCENES_PER_GPU)
        batches.append((torch.from_numpy(xyz_np).to(dev), torch.from_numpy(col_np).to(dev)))
        if k == 0:
            xyz_np0, col_np0 = xyz_np, col_np
    gout = torch.from_numpy(np.random.default_rng(777).standard_normal((SCENES_PER_GPU, NPOINTS, 64)).astype(np.float32)).to(dev)

    gout_scaled = (gout * (1.0 / gout.numel())).contiguous()

    class _DotLoss(torch.autograd.Function):
        """loss = sum(out * g) for a constant g; d loss / d out = g -- returned as is (the incoming gradient of a scalar loss is 1)."""
        @staticmethod
        def forward(ctx, out, g):
            ctx.g = g
            return torch.dot(out.reshape(-1), g.reshape(-1))

        @staticmethod
        def backward(ctx, grad):
            return ctx.g, None

    store = tf_util.set_variable_store(tf_util.VariableStore(device=dev, seed=1234))   # same weights on every rank
    state = {"bucket": None, "opt": None, "i": 0, "pend": None, "t_wait": 0.0}
    geo = None if args.no_overlap else [GeometryStream(dev) for _ in range(DEPTH)]     # (default priority: a high-priority geometry queue starves the layers, 3.7 -> 8.1 ms per step)
    use_graph = geo is not None and not args.no_graph
    pend = {}                   # step index -> PendingGeometry
    done = {}                   # step index -> event after its layers + optimiser step

    def fwd_bwd(k, g):
        """forward + loss + backward of batch k with geometry g, gradients gathered into the flat bucket"""
        xyz, col = batches[k]
        out = pn2_fea_extractor(xyz, col, 'fea', True, 0.5, geometry=g)
        loss = _DotLoss.apply(out, gout_scaled)       # <out, gout> / numel: one reduction kernel; its gradient IS gout_scaled (no kernel)
        loss.backward()
        if state["bucket"] is None:
            params = store.parameters()
            state["bucket"] = parallel.FlatGradBucket(params)
            state["opt"] = parallel.FlatAdam(state["bucket"], lr=1e-3)       # one kernel for the whole update (parameters live in one flat buffer)
        state["bucket"].flatten()
        return loss.detach()          # (a live loss would keep the autograd graph -- and its AccumulateGrad nodes -- alive across captures)

    def finish():
        state["bucket"].all_reduce(average=False)    # one flat RCCL all-reduce (SUM; no-op at world 1)
        state["opt"].step(grad_scale=1.0 / world)    # the division by the world size rides in the Adam kernel

    graphs = None
    if use_graph:
        # persistent geometry buffers per batch slot (the captured layers read these; the geometry stream refills them every step)
        G = [pn2_geometry(batches[k][0]) for k in range(NB)]
        fwd_bwd(0, G[0])                             # creates the variables, the bucket and the optimiser
        finish()
        graphs = []
        def captured(k):
            state["opt"].zero_grad(set_to_none=True)          # host-side only: the captured backward ASSIGNS fresh gradients, flatten() re-points them
            return fwd_bwd(k, G[k])
        for k in range(NB):
            graphs.append(CapturedStep(lambda k=k: captured(k), pool=graphs[0].pool() if graphs else None))

    # the input batches are resident and never rewritten: the geometry of a later step has nothing to wait for on the main stream,
    # except (graph mode) the layers that still READ the persistent geometry buffers of its slot -- NB = DEPTH + 1 slots keep those apart
    AFTER = None

    def submit_geometry(j):
        """geometry of step j (batch slot j % NB) on side stream j % DEPTH"""
        kj = j % NB
        if use_graph:
            pend[j] = geo[j % DEPTH].submit(lambda x: copy_into(G[kj], pn2_geometry(x)), batches[kj][0], after=AFTER)
        else:
            pend[j] = geo[j % DEPTH].submit(pn2_geometry, batches[kj][0], after=AFTER)

    # diagnostic only (the line it prints is NOT a benchmark result: the geometry of every step is skipped): layers graph alone
    LAYERS_ONLY = use_graph and os.environ.get("GSPN_BENCH_LAYERS_ONLY") == "1"
    SIDE = os.environ.get("GSPN_BENCH_SIDE", "") if LAYERS_ONLY else ""      # diagnostic: a chosen part of the geometry beside the layers

    tiny = torch.zeros(64, device=dev)

    def side_load(i):
        from gspn_amd.geometry import fp_geometry, sa_geometry
        from gspn_amd.tf_grouping import query_ball_point
        kj = i % NB
        xyz = batches[kj][0]

        def part(x):
            if "fps0" in SIDE:
                tf_sampling.farthest_point_sample(2048, x)
            if "rest" in SIDE:
                l1 = G[kj]["sa"][0].new_xyz
                query_ball_point(0.2, 32, x, l1)
                s2 = sa_geometry(l1, 512, 0.4, 32)
                s3 = sa_geometry(s2.new_xyz, 128, 0.8, 32)
                fp_geometry(s2.new_xyz, s3.new_xyz); fp_geometry(l1, s2.new_xyz); fp_geometry(x, l1)
            if SIDE.startswith("empty"):                       # N trivial kernels: what does a kernel boundary on another queue cost the layers?
                for _ in range(int(SIDE[5:])):
                    tiny.fill_(1.0)
            if "inv" in SIDE:
                from gspn_amd.geometry import inverse_lists
                g = G[kj]
                for lvl, n in ((1, 2048), (2, 512)):
                    inverse_lists(g["sa"][lvl].idx.reshape(SCENES_PER_GPU, -1), n)
                for f, n in ((0, 128), (1, 512), (2, 2048)):
                    inverse_lists(g["fp"][f].idx.reshape(SCENES_PER_GPU, -1), n)
            if "small" in SIDE:
                l1 = G[kj]["sa"][0].new_xyz
                s2 = sa_geometry(l1, 512, 0.4, 32, inverse=False)
                sa_geometry(s2.new_xyz, 128, 0.8, 32, inverse=False)
            if "nn" in SIDE:
                l1 = G[kj]["sa"][0].new_xyz
                from gspn_amd.tf_interpolate import three_nn
                three_nn(x, l1)
            return None
        if SIDE.startswith("spin"):                            # spin:<which>:<blocks>:<threads>:<amount> -- resident workgroups of tools/spin_probe.hip beside the layers
            import ctypes
            if "spin" not in state:
                state["spin"] = ctypes.CDLL(os.path.join(ROOT, "tools", "libspin_probe.so"))
            _, which, blocks, threads, amount = SIDE.split(":")          # amount: shader-clock cycles (spin_*/hold_*) or iterations (loop_*)
            state["spin"].launch_spin(int(which), int(blocks), int(threads), ctypes.c_longlong(int(amount)), ctypes.c_void_p(tiny.data_ptr()),
                                      ctypes.c_void_p(geo[i % DEPTH].stream.cuda_stream))
            return
        if SIDE.startswith("raw"):                             # the same tiny kernels without the submit() events
            with torch.cuda.stream(geo[i % DEPTH].stream):
                for _ in range(int(SIDE[3:])):
                    tiny.fill_(1.0)
            return
        geo[i % DEPTH].submit(part, xyz)

    if geo is not None and not LAYERS_ONLY:
        for j in range(DEPTH):
            submit_geometry(j)

    def step():
        i = state["i"]
        state["i"] = i + 1
        k = i % NB
        g = None
        if LAYERS_ONLY and SIDE:
            side_load(i)
        if geo is not None and not LAYERS_ONLY:
            tw = time.perf_counter()
            g = pend.pop(i).get(host_wait=True)               # geometry of THIS step (submitted DEPTH steps ago: long complete)
            state["t_wait"] += time.perf_counter() - tw
        if use_graph:
            graphs[k].replay()
        else:
            if state["opt"] is not None:
                state["opt"].zero_grad(set_to_none=True)      # backward assigns fresh grads; the bucket re-points them at its slices
            fwd_bwd(k, g)
        finish()
        if geo is not None and not LAYERS_ONLY:
            # Geometry of step i+DEPTH, to run under the layers of steps i+1 .. i+DEPTH.  In graph mode it refills the persistent buffers
            # of slot (i+DEPTH) % NB = (i-1) % NB, which the layers of step i-1 read: the HOST waits for that step (step i is already
            # queued behind it, so the GPU never idles) instead of making the side stream wait on the layers' stream.
            done[i] = torch.cuda.current_stream().record_event()
            if (i - 1) in done:
                tw = time.perf_counter()
                done.pop(i - 1).synchronize()
                state["t_wait"] += time.perf_counter() - tw
            submit_geometry(i + DEPTH)

    for _ in range(args.warmup):
        step()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    tf_sampling.PROFILE = []          # HIP-event pairs around every FPS launch on its stream
    sync()
    state["t_wait"] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_host = time.perf_counter() - t0 - state["t_wait"]   # host time to enqueue the K steps, net of its waits on the GPU (launch-bound if close to dt)
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
            pj = json.load(open(pmc))
            # the timed bracket holds the sampling kernel alone: its own HBM bytes (the whole call incl. the sort pre-pass: pj["hbm_bytes_per_launch"])
            traffic = next((v["hbm_bytes_per_launch"] for k, v in pj.get("kernels", {}).items() if "fps_cell_kernel" in k), pj.get("hbm_bytes_per_launch"))
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
            "data": "synthetic" if not LAYERS_ONLY else "DIAGNOSTIC RUN, NOT A RESULT: geometry skipped (GSPN_BENCH_LAYERS_ONLY)",
            "config": {"workload": "BASELINE configs[2]: batch 8 x 32768-pt scenes per GPU, 3-level SA + 3-level FP (three_nn/interpolate) fwd+bwd, "
                                   "pn2_fea_extractor layer spec, BN training mode, Adam step", "scenes_per_gpu": SCENES_PER_GPU,
                       "schedule": "geometry inline" if args.no_overlap else ("geometry of batches k+1, k+2 on two side streams under the layers of batch k"
                                                                               + ("; fwd+bwd replayed from a hipGraph" if use_graph else "")),
                       "global_batch": global_batch, "npoints": NPOINTS, "parallelism": "dp%d (scenes sharded, one flat RCCL grad all-reduce)" % world},
            "roofline": {"bound": "hbm", "kernel": "fps_cell_kernel<32,true> (SA1: 8 x 32768 -> 2048; its sort pre-pass, 0.07 ms, is timed outside the bracket)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_ms": fps_avg_ms, "launches_timed": len(fps_ms)},
            "host_enqueue_ms_per_step": t_host / args.steps * 1e3,
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(xyz_np0, col_np0)
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
        sample, reps = 8, 2
        t1 = min(cpu_pipeline.run_step(xyz_np[:sample], col_np[:sample], mt=False) for _ in range(reps))
    finally:
        torch.set_num_threads(nthr)
    cores = os.cpu_count() or 1
    tall = cpu_pipeline.run_step(xyz_np, col_np, mt=True)
    return {"value": sample / t1, "unit": "scenes/s", "cores": 1, "kind": "port",
            "sample": "one full fwd+bwd step on the %d scenes of one batch (32768 pts each), best of %d, single thread: C oracle for "
                      "FPS/ball/group/3-NN/interp, torch-CPU fp32 stand-in for the TensorFlow MLP; %.1f s per step" % (sample, reps, t1),
            "all_cores": {"value": xyz_np.shape[0] / tall, "cores": cores,
                          "note": "same step on all 8 scenes: OpenMP over scenes for FPS/ball query (<=8 threads), torch intra-op threads for the MLP; %.1f s" % tall}}


if __name__ == "__main__":
    main()
