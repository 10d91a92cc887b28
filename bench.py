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

  python bench.py --gpus 1 --steps 100 --warmup 10      (the defaults)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

# The layers' stream, the two geometry streams and the communication library's stream must each own a hardware queue; HIP's default of
# four leaves no slack once RCCL and the capture streams have taken their round-robin slots (gspn_amd/geometry.py: GeometryStream tests
# its queue, this only gives the test room to succeed).  Read by the HIP runtime at initialisation, so it is set before torch loads.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCENES_PER_GPU = 8
NPOINTS = 32768
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


DATA_KIND = "U"            # U = uniform in the unit cube (primary), S = room surfaces at metre scale, D = U with 10 % duplicated points


def synth(b, n, seed0, kind=None):
    kind = kind or DATA_KIND
    if kind == "U":
        xyz = np.stack([np.random.default_rng(seed0 + i).random((n, 3), dtype=np.float32) for i in range(b)])
    else:
        from gspn_amd import synth as _synth            # cloud_s: faces of an 8 x 6 x 3 m room + 20 boxes; cloud_d: dataset.py:100-105's duplicates
        xyz = _synth.batch(kind, b, n, seed0)
    col = np.stack([np.random.default_rng(10_000 + seed0 + i).random((n, 3), dtype=np.float32) for i in range(b)])
    return xyz, col


def _stdout_discipline(rank, done=False):
    """The contract is ONE JSON line on stdout.  RCCL prints a version banner through C stdio when its first communicator comes up; with
    stdout redirected that text sits in the C buffer until the process exits -- AFTER rank 0's line (seen at world size 1 with
    --force-collective: the line was the first of six).  Ranks other than 0 send their file descriptor 1 to /dev/null; rank 0 flushes the C
    buffers (so the banner precedes the line) and closes the tap once the line is out."""
    import ctypes
    try:
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        if rank != 0 or done:
            fd = os.open(os.devnull, os.O_WRONLY)
            os.dup2(fd, 1)
            os.close(fd)
    except Exception:
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 0.2 s of timed steps (20 steps = 40 ms spanned 3325-3402 scenes/s between repeats in round 2; the first steps after a short
    # warm-up also run below the settled clock: 2.03 ms per step over 20 steps against 1.98 over 60-100)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--data", choices=["U", "S", "D"], default="U", help="synthetic cloud kind of SURVEY 8(d): U uniform unit cube (the headline), "
                    "S ScanNet-like room surfaces at metre scale, D uniform with 10 %% duplicated points (dataset.py:100-105)")
    ap.add_argument("--kind-leg", action="store_true", help="(internal) a short run on --data whose line carries the ball-query leg only: the "
                    "default run spawns one per non-headline kind and folds the figures into `data_kinds`")
    ap.add_argument("--shape", default=None, help="(internal, with --kind-leg) SCENES,POINTS per GPU instead of 8,32768: the reference's own operating point "
                    "2,18000 (models/config.py:14-19) through the same schedule -- a detail leg, never the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="geometry inline on the main stream instead of prefetched on a side stream")
    ap.add_argument("--no-graph", action="store_true", help="enqueue the layers kernel by kernel instead of replaying a captured hipGraph")
    ap.add_argument("--sync-bn", action="store_true", help="optional SyncBN (global-batch statistics, SURVEY 8e): unfused layers + two small "
                    "all-reduces per layer, no hipGraph; NOT the headline configuration")
    ap.add_argument("--detail", action="store_true", help="after the headline run, also run the post-run legs (ball-query / layer / per-op rooflines, "
                    "the S and D cloud kinds, the other BASELINE configs, the reference's own harness shapes: several child processes, a few minutes) "
                    "and put them into bench_detail.json.  The stdout line is the same compact object either way.")
    ap.add_argument("--no-extra", action="store_true", help="(accepted for the tools/ scripts of earlier rounds: the default run has no post-run legs any more)")
    ap.add_argument("--legs-only", action="store_true", help="(internal) run the post-run legs alone and print their JSON: roofline_mlp, other_configs, "
                    "roofline_ops, reference_harness -- bench.py runs itself with this flag in a child process")
    ap.add_argument("--force-collective", action="store_true", help="issue the gradient all-reduce through RCCL even at world size 1 (one-rank "
                    "communicator, identity result): the collective path of the N>1 runs, executed on a single GPU; the line then carries `collective`")
    ap.add_argument("--collective-in-graph", action="store_true", help="capture the all-reduce as the last node of the captured step instead of "
                    "issuing it after the replay (diagnostic: the default placement is after the graph)")
    args = ap.parse_args()

    global DATA_KIND, SCENES_PER_GPU, NPOINTS
    DATA_KIND = args.data
    if args.shape:
        if not args.kind_leg:
            raise SystemExit("--shape is a detail leg (use with --kind-leg): the headline workload is BASELINE configs[2]")
        SCENES_PER_GPU, NPOINTS = (int(v) for v in args.shape.split(","))
    if args.legs_only:
        return legs_main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: re-run under torch.distributed.run, one rank per GPU (the driver's own form)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    from gspn_amd import mlp as mlp_mod
    from gspn_amd import parallel, tf_grouping, tf_sampling, tf_util
    from gspn_amd.fea_extractor import pn2_fea_extractor, pn2_geometry, pn2_first_fps
    from gspn_amd.geometry import GeometryStream
    from gspn_amd.graph import CapturedStep, copy_into

    rank, local, world = parallel.init_from_env(force=args.force_collective)
    if args.sync_bn:
        mlp_mod.SYNC_BN = True
        # r04: SyncBN runs on the fused kernels (mlp.SYNC_BN_FUSED: the partial rows of every BN reduction all-reduced between the producing
        # and the summing kernel), so the step is captured like the default one -- RCCL collectives capture into the hipGraph (tested at world
        # 1, tests/test_gpu_collective.py); gloo's host-side collectives and the layer-by-layer torch form cannot be captured
        if (not mlp_mod.SYNC_BN_FUSED or not dist.is_initialized() or dist.get_backend() != "nccl" or os.environ.get("GSPN_SYNC_BN_NO_GRAPH") == "1"):
            args.no_graph = True
    _stdout_discipline(rank)
    FORCE_COLL = bool(args.force_collective)
    COLL_IN_GRAPH = bool(args.collective_in_graph) and (world > 1 or FORCE_COLL)
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d): the line would report n_gpus != --gpus" % (world, args.gpus))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    # diagnostic knobs (not the default): stream priorities of the layers' stream / the geometry streams
    MAIN_PRIO = os.environ.get("GSPN_BENCH_MAIN_PRIO")
    GEO_PRIO = int(os.environ.get("GSPN_BENCH_GEO_PRIO", "0"))
    if MAIN_PRIO is not None:
        torch.cuda.set_stream(torch.cuda.Stream(dev, priority=int(MAIN_PRIO)))

    # inputs resident in HBM before the timed region; rank r owns scenes [8r, 8r+8) of the global batch; NB distinct batches rotate
    # (NB = 3 slots: the layers of step i read slot i%3 while the geometry of steps i+1 and i+2 is being written into the other two)
    # PAIRED (default): the geometry of two consecutive steps is submitted TOGETHER, on the two side streams, every other step, so the
    # two 8-CU FPS kernels run side by side and the chip is free of geometry work for the rest of the two steps.  A resident FPS
    # workgroup slows the layers' kernels chip-wide while it runs and that cost saturates in the number of busy
    # workgroups -- so the same work packed into half the time costs the layers less (measured: +0.31 -> +0.21 ms per step for FPS
    # of SA level 1 alone).  Needs 4 batch slots instead of 3; every step still gets exactly one geometry pass of its own batch.
    PAIRED = os.environ.get('GSPN_BENCH_PAIRED', '1') != '0'
    GROUP = int(os.environ.get('GSPN_BENCH_GROUP', '2')) if PAIRED else 1   # steps whose geometry is submitted together (diagnostic: 4 = four FPS launches side by side every fourth step)
    NB = int(os.environ.get('GSPN_BENCH_NB', str(2 * GROUP) if PAIRED else '3'))
    DEPTH = int(os.environ.get('GSPN_BENCH_DEPTH', str(GROUP) if PAIRED else '2'))     # geometry runs this many steps ahead, on DEPTH side streams (FPS throughput: one CU per scene per stream)
    batches = []
    for k in range(NB):
        xyz_np, col_np = synth(SCENES_PER_GPU, NPOINTS, seed0=(k * 1000 + rank) * SCENES_PER_GPU)
        batches.append((torch.from_numpy(xyz_np).to(dev), torch.from_numpy(col_np).to(dev)))
        if k == 0:
            xyz_np0, col_np0 = xyz_np, col_np
    gout = torch.from_numpy(np.random.default_rng(777).standard_normal((SCENES_PER_GPU, NPOINTS, 64)).astype(np.float32)).to(dev)

    gout_scaled = (gout * (1.0 / gout.numel())).contiguous()
    one = torch.ones((), dtype=torch.float32, device=dev)

    class _DotLoss(torch.autograd.Function):
        """loss = sum(out * g) for a constant g; d loss / d out = g -- returned as is (the incoming gradient of a scalar loss is 1)."""
        @staticmethod
        def forward(ctx, out, g):
            ctx.g = g
            from gspn_amd import _lib as L
            if not out.is_contiguous():
                out = out.contiguous()
            if "dot_work" not in state:
                state["dot_work"] = torch.empty(int(L.lib().gspn_dot_work_floats()), dtype=torch.float32, device=out.device)
            res = torch.empty((), dtype=torch.float32, device=out.device)
            L.check(L.lib().gspn_dot(out.numel(), L.ptr(out), L.ptr(g), L.ptr(state["dot_work"]), L.ptr(res), L.stream()), "dot")     # (hand-written: no library kernel in the step)
            return res

        @staticmethod
        def backward(ctx, grad):
            return ctx.g, None

    store = tf_util.set_variable_store(tf_util.VariableStore(device=dev, seed=1234))   # same weights on every rank
    state = {"bucket": None, "opt": None, "i": 0, "pend": None, "t_wait": 0.0}
    geo = None
    if not args.no_overlap:
        # every geometry stream is TESTED to run beside the layers' stream, beside the other geometry streams and -- when a process
        # group exists -- not to hold up a collective issued on the layers' stream (GeometryStream: hardware-queue round-robin)
        probes = []
        agree = None
        if dist.is_initialized():
            dummy = torch.zeros(1024, device=dev)
            probes.append(lambda: dist.all_reduce(dummy))
            flag = torch.zeros(1, device=dev)

            def agree(ok):              # the verdict of an attempt is timing-based: all ranks take the same one (same number of collectives everywhere)
                flag.fill_(1.0 if ok else 0.0)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                return bool(flag.item() > 0.5)
        geo = []
        for _ in range(DEPTH):          # (default priority: a high-priority geometry queue starves the layers, 3.7 -> 8.1 ms per step)
            geo.append(GeometryStream(dev, priority=GEO_PRIO, beside=[torch.cuda.current_stream()] + [g_.stream for g_ in geo], probes=probes,
                                      agree=agree))
    use_graph = geo is not None and not args.no_graph
    # r06, OPT-IN (GSPN_BENCH_ADAM_IN_GRAPH=1): where nothing has to run between the backward pass and the optimiser on the host's clock -- one rank without
    # a forced collective, or the collective captured into the graph -- the Adam launch can be the graph's last node (gspn_adam_flat_dev: its step counter
    # lives on the device).  Measured SLOWER than the eager launch behind every replay (captured layers 1.509 -> 1.523 ms, full step 1.731 -> 1.750: a graph
    # that follows a graph directly starts later than one that follows an eager kernel; profiles/r06_experiments.txt item 4), so it stays off.
    ADAM_IN_GRAPH = use_graph and os.environ.get("GSPN_BENCH_ADAM_IN_GRAPH", "0") == "1" and ((world == 1 and not FORCE_COLL) or COLL_IN_GRAPH)
    pend = {}                   # step index -> PendingGeometry
    done = {}                   # step index -> event after its layers + optimiser step

    def fwd_bwd(k, g):
        """forward + loss + backward of batch k with geometry g, gradients gathered into the flat bucket"""
        xyz, col = batches[k]
        out = pn2_fea_extractor(xyz, col, 'fea', True, 0.5, geometry=g)
        loss = _DotLoss.apply(out, gout_scaled)       # <out, gout> / numel: one reduction kernel; its gradient IS gout_scaled (no kernel)
        loss.backward(gradient=one)                   # (a given gradient: autograd would otherwise fill a one-element tensor, a 5 us kernel)
        if state["bucket"] is None:
            params = store.parameters()
            state["bucket"] = parallel.FlatGradBucket(params)
            state["opt"] = parallel.FlatAdam(state["bucket"], lr=1e-3, device_step=ADAM_IN_GRAPH)       # one kernel for the whole update (parameters live in one flat buffer)
            if os.environ.get("GSPN_BENCH_SINKS", "1") != "0":
                state["bucket"].attach_sinks()                                   # from the next backward on the gradients are written straight into the bucket (no cat)
        state["bucket"].flatten()
        return loss.detach()          # (a live loss would keep the autograd graph -- and its AccumulateGrad nodes -- alive across captures)

    def finish(in_graph=False):
        if not in_graph and not state.get("skip_collective"):
            state["bucket"].all_reduce(average=False, force=FORCE_COLL)    # one flat RCCL all-reduce (SUM; no-op at world 1 unless --force-collective)
        if not (ADAM_IN_GRAPH and use_graph):
            state["opt"].step(grad_scale=1.0 / world)    # the division by the world size rides in the Adam kernel

    graphs = None
    if use_graph:
        # persistent geometry buffers per batch slot (the captured layers read these; the geometry stream refills them every step)
        G = [pn2_geometry(batches[k][0], points=batches[k][1]) for k in range(NB)]
        fwd_bwd(0, G[0])                             # creates the variables, the bucket and the optimiser
        finish()
        graphs = []
        def captured(k):
            state["opt"].zero_grad(set_to_none=True)          # host-side only: the captured backward ASSIGNS fresh gradients, flatten() re-points them
            r = fwd_bwd(k, G[k])
            if COLL_IN_GRAPH:
                state["bucket"].all_reduce(average=False, force=FORCE_COLL)      # the collective as the captured step's last node (--collective-in-graph)
            if ADAM_IN_GRAPH:
                state["opt"].step(grad_scale=1.0 / world)                        # r06: the optimiser as the graph's last node (its step counter lives on the device)
            return r
        for k in range(NB):
            graphs.append(CapturedStep(lambda k=k: captured(k), pool=graphs[0].pool() if graphs else None))

    # the input batches are resident and never rewritten: the geometry of a later step has nothing to wait for on the main stream,
    # except (graph mode) the layers that still READ the persistent geometry buffers of its slot -- NB = DEPTH + 1 slots keep those apart
    AFTER = None

    FPS_FIRST = os.environ.get("GSPN_BENCH_FPS_FIRST", "1") != "0"      # (A/B hook)
    first_fps = {}

    def submit_first_fps(j):
        """the long pole of step j's geometry (FPS of SA level 1) alone: enqueued for every batch of a group before the rest of any of them, so
        that the second stream's chain starts ~0.05 ms after the first's instead of the ~0.55 ms the host needs to enqueue a whole batch"""
        first_fps[j] = geo[j % DEPTH].submit(pn2_first_fps, batches[j % NB][0], after=AFTER)

    def submit_geometry(j):
        """geometry of step j (batch slot j % NB) on side stream j % DEPTH"""
        kj = j % NB
        f0 = first_fps.pop(j, None)
        f0 = f0._value if f0 is not None else None          # (same stream, in order: no wait)
        if use_graph:
            pend[j] = geo[j % DEPTH].submit(lambda x, c: copy_into(G[kj], pn2_geometry(x, fps0=f0, points=c)), batches[kj][0], batches[kj][1], after=AFTER)
        else:
            pend[j] = geo[j % DEPTH].submit(lambda x, c: pn2_geometry(x, fps0=f0, points=c), batches[kj][0], batches[kj][1], after=AFTER)

    # diagnostic only (the line it prints is NOT a benchmark result: the geometry of every step is skipped): layers graph alone
    LAYERS_ONLY = use_graph and os.environ.get("GSPN_BENCH_LAYERS_ONLY") == "1"
    GAP_EV = [] if os.environ.get("GSPN_BENCH_GAPS") == "1" else None        # diagnostic: two events around every replay
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
            if "fpsn:" in SIDE:                                # fpsn:<scenes>: FPS of SA level 1 for that many scenes in ONE launch every step (how does the tax scale with the CUs held?)
                ns_ = int(SIDE.split("fpsn:")[1].split()[0])
                if "xyzn" not in state:
                    state["xyzn"] = torch.cat([batches[j % NB][0] for j in range(4)], 0).contiguous()
                tf_sampling.farthest_point_sample(2048, state["xyzn"][:ns_].contiguous())
            if "fps32alt" in SIDE and i % 4 == 0:          # four batches' FPS in one launch every fourth step
                if "xyz32" not in state:
                    state["xyz32"] = torch.cat([batches[0][0], batches[1][0], batches[2][0], batches[0][0]], 0).contiguous()
                tf_sampling.farthest_point_sample(2048, state["xyz32"])
            if "fps16alt" in SIDE and i % 2 == 0:          # two batches' FPS in ONE launch every other step (same work per step, half the busy time)
                if "xyz16" not in state:
                    state["xyz16"] = torch.cat([batches[0][0], batches[1][0]], 0).contiguous()
                tf_sampling.farthest_point_sample(2048, state["xyz16"])
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
        geo[i % DEPTH].submit(part, xyz, after=None)          # (inputs resident; an event on the layers' stream that a side stream waits for costs 0.11 ms per step by itself)

    if PAIRED and (DEPTH != GROUP or NB != 2 * GROUP):
        raise SystemExit("GSPN_BENCH_PAIRED needs GSPN_BENCH_DEPTH = GSPN_BENCH_GROUP and 2 x GROUP batch slots")
    # Phase of the paired submissions (r05).  Pairs go out every GROUP-th step; which residue is free.  It is chosen from the run length so that
    # the LAST step of the run is never a submitting one: a pair submitted by the final step has one step's time to run a 2.65 ms chain and the
    # rest shows up after the last step as a drain (1.1-1.6 ms once per run: 3 % of the driver's 20-step form, nothing at 100 steps).  Every step
    # still gets exactly one geometry pass and any K consecutive steps contain K passes' submissions -- only WHEN inside a pair of steps changes.
    PHASE = ((args.warmup + args.steps) % GROUP) if (PAIRED and os.environ.get("GSPN_BENCH_PHASE", "auto") == "auto") else (int(os.environ.get("GSPN_BENCH_PHASE", "0")) % GROUP if PAIRED else 0)

    def is_submit(i):
        return (i - PHASE) % GROUP == 0
    if geo is not None and not LAYERS_ONLY:
        for j in range(DEPTH + PHASE):                     # steps 0 .. GROUP + PHASE - 1: the first submitting step (PHASE) sends PHASE + GROUP ...
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
            state["t_wait_geo"] = state.get("t_wait_geo", 0.0) + time.perf_counter() - tw      # (if this is not ~0 the layers' queue idled: the geometry was late)
        if use_graph:
            th = time.perf_counter()
            if GAP_EV is not None and len(GAP_EV) < 400:
                e_ = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                e_[0].record()                             # reached when the previous step's Adam is done
                graphs[k].replay()
                e_[1].record()                             # reached when this replay's last kernel is done
                GAP_EV.append(e_)
            else:
                graphs[k].replay()
            state["t_replay"] = state.get("t_replay", 0.0) + time.perf_counter() - th
        else:
            if state["opt"] is not None:
                state["opt"].zero_grad(set_to_none=True)      # backward assigns fresh grads; the bucket re-points them at its slices
            fwd_bwd(k, g)
        finish(in_graph=use_graph and COLL_IN_GRAPH)
        if geo is not None and not LAYERS_ONLY:
            # Geometry of step i+DEPTH, to run under the layers of steps i+1 .. i+DEPTH.  In graph mode it refills the persistent buffers
            # of slot (i+DEPTH) % NB = (i-1) % NB, which the layers of step i-1 read: the HOST waits for that step (step i is already
            # queued behind it, so the GPU never idles) instead of making the side stream wait on the layers' stream.
            # (only the step whose completion the host will wait for gets an event: the last of each group)
            if not PAIRED or is_submit(i + 1):
                done[i] = torch.cuda.current_stream().record_event()
            if PAIRED:
                # every GROUP-th step submits the geometry of steps i+GROUP .. i+2*GROUP-1 at once (their slots were last read by steps
                # i-GROUP .. i-1: the host waits for step i-1)
                if is_submit(i):
                    if (i - 1) in done:
                        tw = time.perf_counter()
                        done.pop(i - 1).synchronize()
                        state["t_wait"] += time.perf_counter() - tw
                    for r in range(2, GROUP + 1):
                        done.pop(i - r, None)
                    if FPS_FIRST:
                        for r in range(GROUP):
                            submit_first_fps(i + GROUP + r)
                    for r in range(GROUP):
                        submit_geometry(i + GROUP + r)
            else:
                if (i - 1) in done:
                    tw = time.perf_counter()
                    done.pop(i - 1).synchronize()
                    state["t_wait"] += time.perf_counter() - tw
                submit_geometry(i + DEPTH)

    fps_done, bq_done = [], []        # (ms, meta...) of launches whose events have completed

    def drain_events(final=False):
        """turn completed event pairs into numbers and drop the events: hundreds of live HIP events slow every later launch on the host"""
        for mod, dst in ((tf_sampling, fps_done), (tf_grouping, bq_done)):
            src = mod.PROFILE
            while src and (final or (len(src) > 4 and src[0][1].query())):
                e = src.pop(0)
                dst.append((e[0].elapsed_time(e[1]),) + tuple(e[2:]))
                mod.EVENT_POOL += [e[0], e[1]]            # reused by later launches (no event is created or destroyed in steady state)

    for _ in range(args.warmup):
        step()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if os.environ.get("GSPN_BENCH_NOPROFILE") != "1":      # (diagnostic switch: no per-launch events)
        tf_sampling.PROFILE_MIN_N = NPOINTS   # (the roofline kernel only: SA level 1)
        PSTEPS = int(os.environ.get("GSPN_BENCH_PROFILE_STEPS", "24"))     # steps of the timed region whose FPS / ball-query launches are bracketed
        tf_sampling.PROFILE_BUDGET[0] = PSTEPS
        tf_grouping.PROFILE_BUDGET[0] = 3 * PSTEPS
        tf_sampling.PROFILE = []          # HIP-event pairs around every FPS launch on its stream
        tf_grouping.PROFILE = []          # ... and around every ball-query launch
    sync()
    # Python's cyclic collector: a full (generation 2) collection walks every live object -- 40-56 ms here, once per ~70 steps, i.e.
    # +0.2..0.3 ms per step on a 100-200 step run and nothing on a 20-step one.  Collect now and move the survivors (modules, graphs,
    # parameter objects) to the permanent generation: later collections only look at what the steps themselves allocate.
    import gc
    gc.collect()
    gc.freeze()
    state["t_wait"] = 0.0
    state["t_wait_geo"] = 0.0
    if GAP_EV is not None:
        del GAP_EV[:]
    t0 = time.perf_counter()
    stamps = [] if os.environ.get("GSPN_BENCH_STEP_TIMES") == "1" else None      # (diagnostic: host clock after every step, the host trails the GPU by <= 2 steps)
    # SURVEY 8(d) asks for the MEDIAN step: one timing event per step on the layers' stream (created before the timed region; recording an
    # event that nobody waits for costs the stream nothing -- DESIGN 4.6), read after the final synchronisation
    step_ev = None
    if os.environ.get("GSPN_BENCH_NO_STEP_EVENTS") != "1":
        step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        step_ev[0].record()
    for si in range(args.steps):
        step()
        if step_ev is not None:
            step_ev[si + 1].record()
        if tf_sampling.PROFILE is not None:
            drain_events()
        if stamps is not None:
            stamps.append(time.perf_counter())
    t_host = time.perf_counter() - t0 - state["t_wait"]   # host time to enqueue the K steps, net of its waits on the GPU (launch-bound if close to dt)
    sync()
    dt = time.perf_counter() - t0
    if GAP_EV and rank == 0:              # diagnostic (GSPN_BENCH_GAPS=1): from the end of the previous step's Adam to the end of this step's replay
        sp = sorted(a.elapsed_time(b) for a, b in GAP_EV)
        print("replay span incl. the idle time in front of it, ms: median %.4f mean %.4f p90 %.4f max %.4f over %d steps; host waited %.4f ms per step for geometry"
              % (sp[len(sp) // 2], sum(sp) / len(sp), sp[int(len(sp) * 0.9)], sp[-1], len(sp), state.get("t_wait_geo", 0.0) / args.steps * 1e3), file=sys.stderr)
    if stamps is not None and rank == 0:
        w = 10
        print("ms/step by window of %d steps: %s" % (w, " ".join("%.2f" % ((stamps[min(j + w, len(stamps)) - 1] - (stamps[j - 1] if j else t0)) / (min(j + w, len(stamps)) - j) * 1e3)
                                                               for j in range(0, len(stamps), w))), file=sys.stderr)
        d = sorted(((stamps[j] - (stamps[j - 1] if j else t0)) * 1e3, j) for j in range(len(stamps)))[-3:]
        print("longest steps (ms, index): %s" % " ".join("%.2f@%d" % x for x in d), file=sys.stderr)
    step_median_ms = None
    if step_ev is not None:
        per = [step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps)]
        step_median_ms = float(np.median(per))             # GPU time between the ends of consecutive steps on the layers' stream
        if os.environ.get("GSPN_BENCH_DUMP_STEPS") == "1" and rank == 0:      # (diagnostic)
            print("per-step ms on the layers' stream: " + " ".join("%.2f" % v for v in per) + "   | sum %.2f of %.2f ms wall" % (sum(per), dt * 1e3), file=sys.stderr)
    if tf_sampling.PROFILE is not None:
        drain_events(final=True)
    tf_sampling.PROFILE = None
    tf_grouping.PROFILE = None
    prof, bq_prof = fps_done, bq_done
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # dominant kernel: FPS of SA level 1 (b=8, n=32768 -> m=2048), live HIP-event average over the timed region
    fps_ms = [ms for (ms, b, n, m) in prof if n == NPOINTS]
    b, n, m = SCENES_PER_GPU, NPOINTS, 2048
    alg_bytes = 20.0 * b * (m - 1) * n + 4.0 * b * m            # 20 B/point/round + the index output
    fps_avg_ms = float(np.mean(fps_ms)) if fps_ms else float("nan")
    achieved = alg_bytes / (fps_avg_ms * 1e-3) / 1e9
    traffic = None
    pmc = next((q for q in (os.path.join(ROOT, "profiles", f) for f in ("r06_fps_pmc.json", "r05_fps_pmc.json", "r04_fps_pmc.json", "r03_fps_pmc.json", "r02_fps_pmc.json", "r01_fps_pmc.json")) if os.path.exists(q)), None)
    traffic_src = ("profiles/" + os.path.basename(pmc)) if pmc else None
    if pmc:
        try:
            pj = json.load(open(pmc))
            # the timed bracket holds the sampling kernel alone: its own HBM bytes (the whole call incl. the sort pre-pass: pj["hbm_bytes_per_launch"])
            traffic = next((v["hbm_bytes_per_launch"] for k, v in pj.get("kernels", {}).items() if "fps_cell_kernel" in k), pj.get("hbm_bytes_per_launch"))
        except Exception:
            traffic = None

    coll = None
    if world > 1 or FORCE_COLL:         # every rank enters (a collective is entered by all ranks or by none); after the timed region
        coll = collective_leg(state["bucket"], world, FORCE_COLL, "last node of the captured step" if (use_graph and COLL_IN_GRAPH) else
                              "after the graph replay, before the Adam kernel (same stream)")
        # what the collective costs INSIDE the step: the same steps once more without it (a difference run; the replicas' parameters drift
        # apart from here on, which no longer matters -- the timed region is over).  Not possible when the all-reduce is a node of the graph.
        if not (use_graph and COLL_IN_GRAPH) and not LAYERS_ONLY:
            nd = min(args.steps, 40)
            state["skip_collective"] = True
            for _ in range(3):
                step()
            sync()
            t1 = time.perf_counter()
            for _ in range(nd):
                step()
            sync()
            dt_nc = time.perf_counter() - t1
            state["skip_collective"] = False
            if world > 1:
                tt = torch.tensor([dt_nc], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt_nc = float(tt.item())
            coll["ms_per_step_without_collective"] = dt_nc / nd * 1e3
            coll["allreduce_ms_in_step"] = dt / args.steps * 1e3 - dt_nc / nd * 1e3
            coll["difference_run_steps"] = nd
    if rank == 0:
        global_batch = SCENES_PER_GPU * world
        concurrent = GROUP if (PAIRED and geo is not None) else 1
        res = {
            "metric": METRIC,
            "value": global_batch * args.steps / dt,
            "unit": "scenes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "median_ms_per_step": step_median_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": ("synthetic (%s)" % {"U": "U: xyz ~ U[0,1)^3, SURVEY 8(d)'s primary kind", "S": "S: ScanNet-like room surfaces, 8 x 6 x 3 m + 20 boxes, metre scale",
                                         "D": "D: uniform with the last 10 % of the points duplicates of earlier ones"}[DATA_KIND])
                    if not LAYERS_ONLY else "DIAGNOSTIC RUN, NOT A RESULT: geometry skipped (GSPN_BENCH_LAYERS_ONLY)",
            "config": {"workload": ("BASELINE configs[2]: " if (SCENES_PER_GPU, NPOINTS) == (8, 32768) else "DETAIL LEG (not the headline): ") +
                                   "batch %d x %d-pt scenes per GPU, 3-level SA + 3-level FP (three_nn/interpolate) fwd+bwd, "
                                   "pn2_fea_extractor layer spec, BN training mode, Adam step" % (SCENES_PER_GPU, NPOINTS), "scenes_per_gpu": SCENES_PER_GPU,
                       "schedule": "geometry inline" if args.no_overlap else (("geometry of batches k+2, k+3 submitted together every other step on two side streams under the layers of batches k, k+1"
                                                                                if PAIRED else "geometry of batches k+1, k+2 on two side streams under the layers of batch k")
                                                                               + ("; fwd+bwd replayed from a hipGraph" if use_graph else "")),
                       "geometry_pair_phase": PHASE if (PAIRED and geo is not None) else None,      # pairs are submitted by the steps i with i % 2 == phase (the run's last step submits none)
                       "global_batch": global_batch, "npoints": NPOINTS, "parallelism": "dp%d (scenes sharded, one flat RCCL grad all-reduce)%s" % (world, "; SyncBN" if args.sync_bn else "")},
            # SURVEY 8(d)'s yardstick: `achieved` = ALGORITHMIC bytes (what the reference's kernel moves: 20 B per point per round) / time.
            # It is an effective rate, not measured bandwidth: the kernel keeps the scene on chip, `traffic` (PMC) is what really
            # crosses HBM, and what bounds the kernel is VALU issue + barrier latency -- hence us_per_pick beside it.
            # concurrent_launches: the FPS launches of `concurrent` consecutive batches run side by side (8 CUs each, PAIRED above), so one
            # launch may last longer than a step although every step contains exactly one.
            "roofline": {"bound": "hbm", "kernel": "fps_cell_kernel<32,true> (SA1: 8 x 32768 -> 2048)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg_bytes,
                         "achieved_is": "effective rate = algorithmic bytes / time (the kernel is on-chip resident; see traffic); its sort pre-pass, 0.07 ms, is timed outside the bracket",
                         "avg_launch_ms": fps_avg_ms, "us_per_pick": fps_avg_ms * 1e3 / m, "launches_timed": len(fps_ms), "concurrent_launches": concurrent},
            "geometry_streams": None if geo is None else {"hw_queues_env": os.environ.get("GPU_MAX_HW_QUEUES"), "streams_tried": [g_.tried for g_ in geo],
                                                         "shares_a_queue": [g_.shares_queue for g_ in geo]},
            "geometry_pair_phase": PHASE if (PAIRED and geo is not None) else None,      # pairs are submitted by the steps i with i % 2 == phase (chosen so that the run's last step submits none)
            "host_enqueue_ms_per_step": t_host / args.steps * 1e3,
            "host_wait_ms_per_step": state["t_wait"] / args.steps * 1e3,
            "host_replay_ms_per_step": state.get("t_replay", 0.0) / (args.steps + args.warmup) * 1e3,
        }
        if coll is not None:
            if world > 1:
                coll.update(n1_reference(res["value"], world))
            res["collective"] = coll
        _stdout_discipline(rank)                                    # whatever the C side buffered so far comes out BEFORE the line
        if world == 1 and not args.no_cpu_baseline and not args.kind_leg:
            try:
                res["cpu_baseline"] = cpu_baseline(xyz_np0, col_np0, full=args.detail)
            except Exception as e:                                  # the CPU leg never takes the headline down with it
                print("cpu_baseline failed: %r" % (e,), file=sys.stderr)
        if world == 1 and not LAYERS_ONLY and (args.detail or args.kind_leg):
            torch.cuda.synchronize()
            try:
                res["roofline_ball_query"] = ball_query_roofline(bq_prof, batches, G if use_graph else None)
            except Exception as e:                                  # the extra legs never take the headline line down with them
                res["roofline_ball_query"] = {"error": repr(e)}
        if world == 1 and not LAYERS_ONLY and args.detail and not args.kind_leg:
            # the same step on the other cloud kinds of SURVEY 8(d) (north_star names ScanNet scenes): short runs in child processes
            res["data_kinds"] = data_kinds_legs(res)
            # The remaining legs (per-kernel rooflines of the layers and of the stand-alone ops, the other configs, the reference's own
            # harness shapes) run in a CHILD process: whatever happens there -- an exception, a crash, a hang -- the headline line above
            # is printed.  (r03: a graph capture inside one of these legs segfaulted and the run printed nothing at all.)
            res.update(run_legs_in_child(args))
            if isinstance(res.get("reference_harness"), dict):
                res["reference_harness"]["operating_point_captured"] = operating_point_captured()
        if args.kind_leg:
            print(json.dumps(res), flush=True)                      # (internal child of data_kinds_legs: the parent reads the full object)
        else:
            if not LAYERS_ONLY:
                write_detail(res)
            print(json.dumps(compact_line(res)), flush=True)
        _stdout_discipline(rank, done=True)                         # ... and nothing after it
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def n1_reference(value, world):
    """whole-job value of this N-GPU run against the newest KEPT one-GPU line under profiles/ (weak scaling: efficiency = value / (N * value_1))"""
    for f in ("r06_bench_line.json", "r05_bench_line.json", "r04_bench_line.json"):
        q = os.path.join(ROOT, "profiles", f)
        try:
            v1 = float(json.load(open(q))["value"])
            return {"n1_value": v1, "n1_source": "profiles/" + f + " (a kept line of another run and box)", "scaling_efficiency_vs_n1": value / (world * v1)}
        except Exception:
            continue
    return {"n1_value": None, "n1_source": None, "scaling_efficiency_vs_n1": None}


METRIC = "scenes/sec fwd+bwd set-abstraction, 32768 pts, 1/2/4/8 MI355X"
DETAIL_FILE = "bench_detail.json"
# the keys of the stdout line: everything else lives in bench_detail.json
LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "median_ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data", "config", "roofline", "cpu_baseline", "collective", "detail_file")
CONFIG_KEYS = ("workload", "scenes_per_gpu", "global_batch", "npoints", "parallelism", "schedule", "geometry_pair_phase")
# traffic is a KEPT counter figure (a --pmc pass cannot run inside the timed command): traffic_source names the file it was read from
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes_per_launch", "avg_launch_ms", "launches_timed",
                 "concurrent_launches")
# N > 1 only (null at N = 1): what the gradient all-reduce costs inside a step and the whole-job value against a KEPT one-GPU line (n1_source) --
# so that the first multi-GPU record explains itself; the driver computes its own efficiency from its own N = 1 run
COLLECTIVE_KEYS = ("allreduce_ms_in_step", "ms_per_step_without_collective", "bucket_bytes", "n1_value", "n1_source", "scaling_efficiency_vs_n1")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "all_cores")
LINE_LIMIT = 4096


def _short(v, n):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 3] + "..."


def _num(v):
    """six significant digits are plenty for the line (the detail file keeps full precision)"""
    if isinstance(v, float) and v == v and abs(v) != float("inf"):
        return float("%.6g" % v)
    if isinstance(v, float):
        return None                      # NaN / inf are not JSON
    return v


def compact_line(res):
    """The ONE stdout line: a fixed key set, numbers at six digits, strings capped, strict JSON, < 4 KB whatever the full result holds."""
    out = {}
    for k in LINE_KEYS:
        if k == "config":
            c = res.get("config") or {}
            out[k] = {q: _short(_num(c.get(q)), 200) for q in CONFIG_KEYS}
        elif k == "roofline":
            r = res.get("roofline") or {}
            out[k] = {q: _short(_num(r.get(q)), 120) for q in ROOFLINE_KEYS}
        elif k == "cpu_baseline":
            c = res.get("cpu_baseline")
            if c is None:
                out[k] = None
            else:
                o = {q: _short(_num(c.get(q)), 200) for q in CPU_KEYS if q != "all_cores"}
                a = c.get("all_cores")
                o["all_cores"] = None if not a else {"value": _num(a.get("value")), "cores": a.get("cores")}
                out[k] = o
        elif k == "collective":
            c = res.get("collective") if (res.get("n_gpus") or 1) > 1 else None
            out[k] = None if not c else {q: _short(_num(c.get(q)), 120) for q in COLLECTIVE_KEYS}
        elif k == "detail_file":
            out[k] = DETAIL_FILE
        else:
            out[k] = _short(_num(res.get(k)), 200)
    line = json.dumps(out, allow_nan=False)
    assert len(line) < LINE_LIMIT, "bench line grew to %d bytes" % len(line)
    return out


def write_detail(res):
    """everything the run knows (full precision, prose, the legs of --detail) next to bench.py; a short digest to stderr"""
    try:
        with open(os.path.join(ROOT, DETAIL_FILE), "w") as f:
            json.dump(res, f, indent=1, default=repr)
    except OSError as e:
        print("bench_detail.json not written: %r" % (e,), file=sys.stderr)
    dig = {k: res.get(k) for k in ("host_enqueue_ms_per_step", "host_wait_ms_per_step", "host_replay_ms_per_step", "geometry_streams", "collective") if res.get(k) is not None}
    print("bench detail (full object in %s): %s" % (DETAIL_FILE, json.dumps(dig, default=repr)), file=sys.stderr)


def data_kinds_legs(res_u, kinds=("S", "D"), timeout=600):
    """U / S / D side by side: step time, FPS of SA level 1 (us per pick), ball query per SA level (sum_visited, time).  The headline kind's
    figures are copied from this run's own line; the others come from `bench.py --data K --kind-leg` children (40 timed steps)."""
    import subprocess

    def pick(r):
        bq = r.get("roofline_ball_query", {})
        return {"ms_per_step": r["ms_per_step"], "scenes_per_s": r["value"], "steps": r["steps"],
                "fps_sa1": {k: r["roofline"].get(k) for k in ("avg_launch_ms", "us_per_pick", "frac")},
                "ball_query": [{k: lv.get(k) for k in ("level", "radius", "avg_launch_ms", "sum_visited", "upper_bound_bytes", "algorithmic_bytes_per_launch", "frac")}
                               for lv in bq.get("levels", [])] if isinstance(bq, dict) else bq}
    out = {DATA_KIND: pick(res_u)}
    for k in kinds:
        if k == DATA_KIND:
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--data", k, "--kind-leg", "--no-cpu-baseline", "--steps", "40", "--warmup", "5"]
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            out[k] = pick(json.loads(lines[-1])) if (r.returncode == 0 and lines) else {"error": "child exited with code %d" % r.returncode, "stderr_tail": r.stderr[-400:]}
        except Exception as e:
            out[k] = {"error": repr(e)}
    return out


def operating_point_captured(timeout=600):
    """the reference's own operating point (models/config.py:14-19, train.py:27-28: batch 2 x 18000 points) through the headline's schedule --
    layers replayed from a hipGraph, geometry prefetched on the side streams -- in a child process"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--shape", "2,18000", "--kind-leg", "--no-cpu-baseline", "--steps", "100", "--warmup", "10"]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "child exited with code %d" % r.returncode, "stderr_tail": r.stderr[-400:]}
        d = json.loads(lines[-1])
        return {"workload": d["config"]["workload"], "schedule": d["config"]["schedule"], "ms_per_step": d["ms_per_step"], "median_ms_per_step": d["median_ms_per_step"],
                "scenes_per_s": d["value"], "steps": d["steps"], "fps_sa1": {k: d["roofline"].get(k) for k in ("avg_launch_ms", "us_per_pick")},
                "host_enqueue_ms_per_step": d.get("host_enqueue_ms_per_step"), "host_wait_ms_per_step": d.get("host_wait_ms_per_step")}
    except Exception as e:
        return {"error": repr(e)}


def run_legs_in_child(args, timeout=900):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--legs-only"] + (["--no-cpu-baseline"] if args.no_cpu_baseline else [])
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"extra_legs": {"error": "child exited with code %d" % r.returncode, "stderr_tail": r.stderr[-600:]}}
        return json.loads(lines[-1])
    except subprocess.TimeoutExpired:
        return {"extra_legs": {"error": "child timed out after %d s" % timeout}}
    except Exception as e:
        return {"extra_legs": {"error": repr(e)}}


def legs_main(args):
    """the post-run legs on a fresh process and device context (one GPU): each leg in its own try block, ONE JSON object on stdout"""
    from gspn_amd import mlp as mlp_mod
    from gspn_amd import parallel, tf_util
    from gspn_amd.fea_extractor import pn2_fea_extractor, pn2_geometry, pn2_first_fps
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    xyz_np, col_np = synth(SCENES_PER_GPU, NPOINTS, seed0=0)
    xyz, col = torch.from_numpy(xyz_np).to(dev), torch.from_numpy(col_np).to(dev)
    gout = torch.from_numpy(np.random.default_rng(777).standard_normal((SCENES_PER_GPU, NPOINTS, 64)).astype(np.float32)).to(dev) / (SCENES_PER_GPU * NPOINTS * 64)
    store = tf_util.set_variable_store(tf_util.VariableStore(device=dev, seed=1234))
    G0 = pn2_geometry(xyz)
    state = {"opt": None}

    def eager_step():
        for p_ in store.parameters():
            p_.grad = None
        (pn2_fea_extractor(xyz, col, 'fea', True, 0.5, geometry=G0) * gout).sum().backward()
    out = {}
    for key, fn in (("roofline_mlp", lambda: mlp_roofline(mlp_mod, eager_step, state)),
                    ("other_configs", lambda: other_configs(xyz, col, dev)),
                    ("roofline_ops", lambda: ops_roofline(xyz, G0, dev))) + \
                   (() if args.no_cpu_baseline else (("reference_harness", lambda: reference_harness(dev)),)):
        try:
            out[key] = fn()
        except Exception as e:
            out[key] = {"error": repr(e)}
        torch.cuda.synchronize()
    print(json.dumps(out), flush=True)
    return 0


def collective_leg(bucket, world, forced, placement, reps=50):
    """the gradient all-reduce alone: `reps` back-to-back calls on an otherwise idle stream, HIP events around the lot (rank 0's view; the
    other ranks run the same calls -- a collective is entered by everyone or by no one)"""
    flat = bucket.flat
    keep = flat.clone()
    for _ in range(3):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    e1.record()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    flat.copy_(keep)
    return {"backend": dist.get_backend(), "world": world, "forced_at_world_1": bool(forced and world == 1), "placement": placement,
            "bucket_bytes": flat.numel() * 4, "allreduce_ms_standalone": e0.elapsed_time(e1) / reps, "host_enqueue_ms_per_call": t_host / reps * 1e3,
            "note": "per-step cost inside the step = ms_per_step of this run minus that of the same command without the collective (DESIGN 6)"}


def ball_query_roofline(bq_prof, batches, G):
    """per SA level: HIP-event time of the ball-query launches of the timed region against the ALGORITHMIC bytes of SURVEY 8(d):
    12*sum(L) + 12*b*m + 4*b*m*(ns+1), L = data points the reference scan visits before its break (oracle's `visited`, computed here
    on batch slot 0's clouds and centres -- the three synthetic batches are statistically alike)."""
    from oracle import oracle as O
    from gspn_amd.fea_extractor import PN2_SA_SPEC, pn2_geometry
    g = G[0] if G is not None else pn2_geometry(batches[0][0])
    cur = batches[0][0].cpu().numpy()
    out = []
    for lvl, (npoint, radius, ns) in enumerate(PN2_SA_SPEC):
        new = g["sa"][lvl].new_xyz.cpu().numpy()
        bsz, n = cur.shape[0], cur.shape[1]
        _, _, visited = O.query_ball_point(radius, ns, cur, new, return_visited=True, mt=True)
        sum_l = float(visited.astype(np.int64).sum())
        alg = 12.0 * sum_l + 12.0 * bsz * npoint + 4.0 * bsz * npoint * (ns + 1)
        ms = [t for (t, b_, n_, m_, r_, ns_) in bq_prof if n_ == n and m_ == npoint]
        avg = float(np.mean(ms)) if ms else float("nan")
        ach = alg / (avg * 1e-3) / 1e9
        out.append({"level": "SA%d" % (lvl + 1), "n": n, "m": npoint, "radius": radius, "nsample": ns, "avg_launch_ms": avg, "launches_timed": len(ms),
                    "sum_visited": sum_l, "algorithmic_bytes_per_launch": alg, "upper_bound_bytes": 12.0 * bsz * npoint * n,
                    "achieved": ach, "frac": ach / HBM_PEAK_GBS})
        cur = new
    return {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "kernel": "ball_query_kernel (one wave per query, ballot/mbcnt compaction, early exit)",
            "achieved_is": "effective rate = algorithmic bytes / time; the scene (<= 384 KiB) is L2-resident, so this is not HBM traffic",
            "levels": out}


MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 MFMA


def mlp_roofline(mlp_mod, eager_step, state, reps=3):
    """The shared-MLP GEMM kernels of one fwd+bwd step (16 layers: forward, weight-gradient pass A, data-gradient pass B), each launch
    bracketed by HIP events on its stream in `reps` eager (un-captured) steps run after the timed region.  flops = SURVEY 8(d):
    2*rows*cin*cout per GEMM, x3 for fwd+bwd."""
    for _ in range(5):                                 # warm: variables, per-kernel attributes, the caching allocator's pool (this runs in a fresh process)
        eager_step()
    torch.cuda.synchronize()
    mlp_mod.PROFILE = []
    try:
        for _ in range(reps):
            if state["opt"] is not None:
                state["opt"].zero_grad(set_to_none=True)
            eager_step()
        torch.cuda.synchronize()
        prof = mlp_mod.PROFILE
    finally:
        mlp_mod.PROFILE = None
    by = {"fwd": 0.0, "wgrad": 0.0, "bwd": 0.0, "fused": 0.0}      # fused: pass A + pass B of a layer in one launch (gspn_mlp_bwd_fused)
    flops = 0.0
    nbytes = 0.0
    executed = 0.0
    for kind, rows, cin, cout, e0, e1, ex in prof:
        by[kind] += e0.elapsed_time(e1)
        executed += ex
        if kind == "fwd":
            flops += 3 * 2.0 * rows * cin * cout
            # one read of X and one write of Y forward; X, Y, dZ read by pass A; Y, dZ read + dX written by pass B
            nbytes += 4.0 * rows * ((cin + cout) + (cin + 2 * cout) + (2 * cout + cin))
    for k in by:
        by[k] /= reps
    flops /= reps
    nbytes /= reps
    executed /= reps
    total_ms = sum(by.values())
    tf = flops / (total_ms * 1e-3) / 1e12
    hbm_floor_ms = nbytes / (HBM_PEAK_GBS * 1e9) * 1e3
    return {"bound": "mfma_f32", "unit": "TFLOP/s", "peak": MFMA_F32_PEAK_TFLOPS, "achieved": tf, "frac": tf / MFMA_F32_PEAK_TFLOPS,
            "kernels": "fwd_* / wgrad_* (+ finalize) / bwd_* (pass B) / bwd_fused_* (both passes) of the 16 layers of pn2_fea_extractor",
            "flops_per_step": flops, "executed_flops_per_step": executed, "executed_TFLOPs": executed / (total_ms * 1e-3) / 1e12,
            "gemm_ms_per_step": total_ms, "ms_by_pass": by,
            "algorithmic_bytes_per_step": nbytes, "algorithmic_TBps": nbytes / (total_ms * 1e-3) / 1e12,
            # at 12 flop/B the stack sits left of the fp32 ridge (157.3 TF / 8 TB/s = 19.7 flop/B): its true roof is HBM
            "hbm_roof": {"floor_ms_at_peak": hbm_floor_ms, "frac": hbm_floor_ms / total_ms, "flop_per_byte": flops / nbytes},
            "note": "flops = SURVEY 8(d)'s algorithmic count of the 16 layers (the pre-aggregated first layers of SA2 / SA3 / the last FP level execute "
                    "fewer: their feature part runs on the source points); eager launches bracketed one by one (includes ~1-2 us of event overhead per "
                    "launch); profiles/ holds the rocprofv3 per-kernel table of the captured step"}


def _ev_time(fn, warm=3, reps=20):
    """average milliseconds of fn() on the GPU: `reps` calls captured into one hipGraph and replayed between two HIP events, so that the
    figure is kernel time and not the host's time to enqueue a 10-microsecond launch from Python (eager fallback if capture fails)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        del g
        return ms
    except Exception:
        torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def _pmc_ops():
    """memory-side bytes per launch of the stand-alone ops (tools/pmc_ops.sh -> profiles/r03_ops_pmc.json: separate --pmc FETCH_SIZE /
    WRITE_SIZE passes over tools/ops_only.py, FETCH_SIZE doubled per MI355X_MICROARCH.md), keyed like the entries below"""
    q = next((f for f in (os.path.join(ROOT, "profiles", n_) for n_ in ("r05_ops_pmc.json", "r04_ops_pmc.json", "r03_ops_pmc.json")) if os.path.exists(f)), None)
    _pmc_ops.source = ("profiles/" + os.path.basename(q)) if q else None
    try:
        return json.load(open(q)).get("ops", {}) if q else {}
    except Exception:
        return {}


def ops_roofline(xyz, geo, dev, timer=None):
    """SURVEY 8(d)'s remaining per-op rooflines at the bench shapes (BASELINE configs[2] levels; nn_distance at configs[3]'s 2048 x (512,512)
    clouds): three_nn, three_interpolate(+grad) and the fused FP input (fp_concat + its inverse-list gradient), group_point(+grad),
    gather_point, nn_distance(+grad).  `achieved` = 8(d)'s algorithmic bytes / HIP-event time of the stand-alone launch (20 back-to-back
    calls replayed from a hipGraph on an idle chip) -- an EFFECTIVE rate: these working sets (<= 134 MB, mostly <= 4 MB) live in L2 / Infinity Cache, so the
    fraction of the 8 TB/s HBM peak can exceed what HBM could deliver; `traffic` is the PMC memory-side byte count per launch and
    `bound_by` names what actually limits the kernel."""
    from gspn_amd.invlists import inverse_lists
    from gspn_amd.pointnet_util import fp_concat
    from gspn_amd.tf_grouping import group_point
    from gspn_amd.tf_interpolate import three_interpolate, three_nn
    from gspn_amd.tf_nndistance import nn_distance
    from gspn_amd.tf_sampling import gather_point
    from gspn_amd import _lib as L
    lib = L.lib()
    pmc = _pmc_ops()
    gen = torch.Generator(device=dev).manual_seed(21)
    b = xyz.shape[0]
    out = []

    if timer is None:
        timer = lambda key, fn: _ev_time(fn)       # (tools/ops_only.py passes one that brackets a single launch with marker kernels for the PMC passes)

    # vector-issue yardstick for the ops that work out of LDS / registers (their HBM fraction says nothing: VERDICT r04): lane-instructions per
    # second the chip can issue = 256 CUs x 4 SIMDs x 64 lanes x 2.4 GHz / 2.07 cycles per fp32 add/mul/fma (tools/valu_probe*.hip)
    VALU_PEAK = 256 * 4 * 64 * 2.4e9 / 2.07

    def add(name, shape, alg_bytes, fn, bound_by, key=None, lane_instr=None):
        ms = timer(key or name, fn)
        ach = alg_bytes / (ms * 1e-3) / 1e9
        e = {"op": name, "shape": shape, "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": ms, "achieved": ach,
             "frac": ach / HBM_PEAK_GBS, "traffic": pmc.get(key or name), "bound_by": bound_by}
        if lane_instr is not None:          # (pairs x vector instructions per pair) / time against the issue peak
            e["valu_issue_frac"] = lane_instr / (ms * 1e-3) / VALU_PEAK
        out.append(e)

    torch.set_grad_enabled(False)            # forward launches and direct C-ABI gradient launches only: nothing here records an autograd graph
    lv = [xyz, geo["sa"][0].new_xyz, geo["sa"][1].new_xyz, geo["sa"][2].new_xyz]
    # three_nn (tf_interpolate.cpp:60-103): 12*b*n*m + 36*b*n
    for d, s_ in ((0, 1), (1, 2), (2, 3)):
        n, m = lv[d].shape[1], lv[s_].shape[1]
        order = geo["sa"][d].scan_order if d < 3 else None
        add("three_nn", "%dx%d<-%d%s" % (b, n, m, " (queries in the FPS pre-pass order)" if order is not None else ""), 12.0 * b * n * m + 36.0 * b * n, lambda: three_nn(lv[d], lv[s_], order=order),
            "VALU issue + LDS broadcast: 4 instructions per (query, candidate) pair, exact re-evaluation of the survivors; the known cloud (<= 24 KB per "
            "scene) sits in LDS" if n >= 2048 else "launch latency (a few microseconds of work)", "three_nn_%d" % n, lane_instr=4.0 * b * n * m)
    # three_interpolate (+grad) (tf_interpolate.cpp:107-153): b*n*(24 + 16*c), and the fused FP input used by the bench graph
    for (d, s_, c2, c1, k) in ((0, 1, 128, 3, 2), (1, 2, 256, 64, 1), (2, 3, 256, 128, 0)):
        n, m = lv[d].shape[1], lv[s_].shape[1]
        fpg = geo["fp"][k]
        p2 = torch.randn(b, m, c2, device=dev, generator=gen)
        p1 = torch.randn(b, n, c1, device=dev, generator=gen)
        go = torch.randn(b, n, c2, device=dev, generator=gen)
        add("three_interpolate", "%dx%d<-%d, c=%d" % (b, n, m, c2), 1.0 * b * n * (24 + 16 * c2), lambda: three_interpolate(p2, fpg.idx, fpg.weight),
            "memory: one coalesced (n, c) write, three L2-resident row reads per point", "three_interpolate_%d" % n)
        gp2 = torch.empty_like(p2)
        # the op API's gradient (tf_interpolate.py, r04): a gather through the inverse lists of idx (cached on the index tensor), sums in the
        # reference's own order, bit-exact vs oracle/_ref; the lists' one-off build is its own entry below
        add("three_interpolate_grad (op API: gather over inverse lists, built per call unless the cache is opted into)", "%dx%d->%d, c=%d" % (b, n, m, c2), 1.0 * b * n * (24 + 16 * c2),
            lambda: L.check(lib.gspn_fp_concat_grad_csr(b, n, m, c2, 0, c2, L.ptr(go), L.ptr(fpg.order), L.ptr(fpg.offsets), L.ptr(fpg.weight), L.ptr(gp2), None,
                                                        L.stream()), "three_interpolate_grad(csr)"),
            "dependent-load latency of the inverse-list walk (16 lanes per sparse point, 8 rows in flight)", "three_interpolate_grad_csr_%d" % n)
        add("three_interpolate_grad (C-ABI drop-in symbol: scatter-add, hardware fp32 atomics)", "%dx%d->%d, c=%d" % (b, n, m, c2), 1.0 * b * n * (24 + 16 * c2),
            lambda: L.check(lib.gspn_threeinterpolate_grad(b, n, c2, m, L.ptr(go), L.ptr(fpg.idx), L.ptr(fpg.weight), L.ptr(gp2), L.stream()), "three_interpolate_grad"),
            "L2 atomic throughput: 3*c atomic adds per dense point onto m*c addresses", "three_interpolate_grad_%d" % n)
        ws3 = torch.empty(int(lib.gspn_threeinterpolate_grad_ws_bytes(b, n, c2, m)), dtype=torch.uint8, device=dev)
        add("three_interpolate_grad (C-ABI drop-in symbol WITH a workspace, ABI 9: inverse lists built in the call + the ordered gather)", "%dx%d->%d, c=%d" % (b, n, m, c2),
            1.0 * b * n * (24 + 16 * c2),
            lambda: L.check(lib.gspn_threeinterpolate_grad_ws(b, n, c2, m, L.ptr(go), L.ptr(fpg.idx), L.ptr(fpg.weight), L.ptr(gp2), L.ptr(ws3), L.stream()), "three_interpolate_grad_ws"),
            "the list build (one workgroup per scene) + the dependent-load latency of the walk", "three_interpolate_grad_ws_%d" % n)
        idx2d = fpg.idx.reshape(b, 3 * n)
        add("inverse_lists (once per index tensor)", "%d x %d positions -> %d lists" % (b, 3 * n, m), 1.0 * b * (3 * n * 12 + 4 * m),
            lambda: inverse_lists(idx2d, m), "one workgroup per scene (LDS histogram + cursors), then a per-list sort", "inverse_lists_%d" % n)
        ld2 = (c2 + c1 + 3) // 4 * 4
        g2 = torch.randn(b * n, ld2, device=dev, generator=gen)
        add("fp_concat (interpolate + concat, fused)", "%dx%d<-%d, c2=%d c1=%d" % (b, n, m, c2, c1), 1.0 * b * n * (24 + 16 * c2 + 8 * c1), lambda: fp_concat(p2, fpg.idx, fpg.weight, p1, fpg.order, fpg.offsets),
            "memory: one write of the (n, c2+c1) input matrix", "fp_concat_%d" % n)
        add("fp_concat_grad (gather over inverse lists, no atomics)", "%dx%d->%d, c2=%d" % (b, n, m, c2), 1.0 * b * n * (24 + 16 * c2),
            lambda: L.check(lib.gspn_fp_concat_grad_csr(b, n, m, c2, c1, ld2, L.ptr(g2), L.ptr(fpg.order), L.ptr(fpg.offsets), L.ptr(fpg.weight), L.ptr(gp2), None,
                                                        L.stream()), "fp_concat_grad_csr"),
            "dependent-load latency of the inverse-list walk (one wave per sparse point x 64 channels)", "fp_concat_grad_%d" % n)
    # group_point (+grad) (tf_grouping_g.cu:43-83): b*m*ns*(4 + 8*c); gather_point (tf_sampling_g.cu:172-192)
    feats = [3, 64, 128]
    for lvl in range(3):
        sa = geo["sa"][lvl]
        n, m, ns, c = lv[lvl].shape[1], sa.idx.shape[1], sa.idx.shape[2], feats[lvl]
        pts = torch.randn(b, n, c, device=dev, generator=gen)
        add("group_point", "%dx%d -> (%d,%d), c=%d" % (b, n, m, ns, c), 1.0 * b * m * ns * (4 + 8 * c), lambda: group_point(pts, sa.idx),
            "memory: the (m, ns, c) write; gathered rows are L2 hits" if c >= 64 else "write coalescing: 12-byte rows", "group_point_%d" % n)
        go = torch.randn(b, m, ns, c, device=dev, generator=gen)
        gpts = torch.empty(b, n, c, device=dev)
        sord, soff = (sa.order, sa.offsets) if sa.order is not None else inverse_lists(sa.idx.reshape(b, m * ns), n)
        from gspn_amd import invlists as _inv
        add("group_point_grad (gather over inverse lists%s)" % ("; the op API's choice at this width" if not _inv.use_atomic(c) else "; GSPN_DETERMINISTIC_GRADS=1"),
            "(%d,%d) -> %dx%d, c=%d" % (m, ns, b, n, c), 1.0 * b * m * ns * (4 + 8 * c),
            lambda: L.check(lib.gspn_sa_group_concat_grad_csr(b, n, c, m, ns, L.ptr(sord), L.ptr(soff), 0, c, L.ptr(go), L.ptr(gpts), L.stream()), "group_point_grad(csr)"),
            "dependent-load latency of the inverse-list walk; every gradient row read once", "group_point_grad_csr_%d" % n)
        add("group_point_grad (C-ABI drop-in symbol: scatter-add, hardware fp32 atomics%s)" % ("; the op API's choice at this width" if _inv.use_atomic(c) else ""),
            "(%d,%d) -> %dx%d, c=%d" % (m, ns, b, n, c), 1.0 * b * m * ns * (4 + 8 * c),
            lambda: L.check(lib.gspn_grouppoint_grad(b, n, c, m, ns, L.ptr(go), L.ptr(sa.idx), L.ptr(gpts), L.stream()), "group_point_grad"),
            "L2 atomic throughput", "group_point_grad_%d" % n)
        wsg = torch.empty(int(lib.gspn_grouppoint_grad_ws_bytes(b, n, c, m, ns)), dtype=torch.uint8, device=dev)
        add("group_point_grad (C-ABI drop-in symbol WITH a workspace, ABI 9: inverse lists built in the call + the ordered gather)", "(%d,%d) -> %dx%d, c=%d" % (m, ns, b, n, c),
            1.0 * b * m * ns * (4 + 8 * c),
            lambda: L.check(lib.gspn_grouppoint_grad_ws(b, n, c, m, ns, L.ptr(go), L.ptr(sa.idx), L.ptr(gpts), L.ptr(wsg), L.stream()), "group_point_grad_ws"),
            "the list build (one workgroup per scene) + the dependent-load latency of the walk", "group_point_grad_ws_%d" % n)
    fidx = geo["sa"][0].idx[:, :, 0].contiguous()
    add("gather_point", "%dx%d -> %d" % (b, xyz.shape[1], fidx.shape[1]), 1.0 * b * fidx.shape[1] * (4 + 24), lambda: gather_point(xyz, fidx), "launch latency (590 KB moved)", "gather_point")
    # nn_distance (+grad) (tf_nndistance_g.cu:5-151): 12*b*(2*n*m) + 8*b*(n+m); grad b*(n+m)*56
    for (nb, n, m) in ((256 * b, 512, 512), (32, 16384, 1024)):
        a = torch.randn(nb, n, 3, device=dev, generator=gen)
        c_ = torch.randn(nb, m, 3, device=dev, generator=gen)
        add("nn_distance", "%d clouds x (%d, %d)" % (nb, n, m), 12.0 * nb * 2 * n * m + 8.0 * nb * (n + m), lambda: nn_distance(a, c_),
            "VALU issue: 7 instructions per point pair from LDS tiles (the clouds are on-chip; 38 MB of input for 2048 clouds)", "nn_distance_%d" % n,
            lane_instr=7.0 * nb * 2 * n * m)
        with torch.no_grad():
            d1, i1, d2, i2 = nn_distance(a.detach(), c_.detach())
        g1, g2 = torch.randn_like(d1), torch.randn_like(d2)
        ga, gc = torch.empty_like(a), torch.empty_like(c_)
        add("nn_distance_grad", "%d clouds x (%d, %d) (scatter-add, hardware fp32 atomics)" % (nb, n, m), 56.0 * nb * (n + m),
            lambda: L.check(lib.gspn_nmdistance_grad(nb, n, L.ptr(a), m, L.ptr(c_), L.ptr(g1), L.ptr(i1), L.ptr(g2), L.ptr(i2), L.ptr(ga), L.ptr(gc), L.stream()),
                            "nn_distance_grad"),
            "L2 atomic throughput + two memsets", "nn_distance_grad_%d" % n)
        wsn = torch.empty(int(lib.gspn_nmdistance_grad_ws_bytes(nb, n, m)), dtype=torch.uint8, device=dev)
        add("nn_distance_grad (C-ABI drop-in symbol WITH a workspace, ABI 9: two list builds + the ordered gather)", "%d clouds x (%d, %d)" % (nb, n, m), 56.0 * nb * (n + m),
            lambda: L.check(lib.gspn_nmdistance_grad_ws(nb, n, L.ptr(a), m, L.ptr(c_), L.ptr(g1), L.ptr(i1), L.ptr(g2), L.ptr(i2), L.ptr(ga), L.ptr(gc), L.ptr(wsn), L.stream()),
                            "nn_distance_grad_ws"),
            "the list builds (one workgroup per cloud) + the walk", "nn_distance_grad_ws_%d" % n)
    torch.set_grad_enabled(True)
    return {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "achieved_is": "effective rate = SURVEY 8(d) algorithmic bytes / stand-alone launch time (HIP events, 20 calls, idle chip); traffic = PMC "
                           "memory-side bytes per launch (%s) or null" % getattr(_pmc_ops, "source", None),
            "ops": out}


def reference_harness(dev):
    """The only performance harnesses the reference ships, at their own shapes, HIP kernel beside CPU code:
      * tf_ops/3d_interpolation/interpolate.cpp:134,153-167 (b=32, n=512, m=128, c=64) / tf_interpolate.py:40-54: three_nn,
        three_interpolate, three_interpolate_grad.  CPU = the REFERENCE'S OWN compiled loops for interpolate / interpolate_grad
        (oracle/_ref/libinterp_ref.so, built from that very file) -- kind "reference"; its threenn_cpu ignores the query point
        (interpolate.cpp:34), so the 3-NN CPU time is the oracle's restatement of tf_interpolate.cpp:60-103 -- kind "port".
      * tf_ops/nn_distance/tf_nndistance.py:48-66 ((32,16384,3) x (32,1024,3), forward + gradient): CPU = the oracle's twin of the
        reference's own CPU kernel nnsearch (tf_nndistance.cpp:21-43,126-163) -- kind "port", one thread like the reference.
      * the reference's operating point (models/config.py:14-19: batch 2 x 18000 points): one fwd+bwd of pn2_fea_extractor."""
    from oracle import oracle as O
    from gspn_amd import tf_util
    from gspn_amd.fea_extractor import pn2_fea_extractor
    from gspn_amd.tf_interpolate import three_interpolate, three_nn
    from gspn_amd.tf_nndistance import nn_distance
    res = {}

    def best(fn, reps=3):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return min(ts) * 1e3

    # ---- interpolate.cpp / tf_interpolate.py ----
    rng = np.random.RandomState(100)                       # np.random.seed(100), tf_interpolate.py:39
    pts = rng.random_sample((32, 128, 64)).astype(np.float32)
    x1 = rng.random_sample((32, 512, 3)).astype(np.float32)
    x2 = rng.random_sample((32, 128, 3)).astype(np.float32)
    tp, t1, t2 = (torch.from_numpy(a).to(dev) for a in (pts, x1, x2))
    dist, idx = three_nn(t1, t2)
    w = torch.full_like(dist, 1.0 / 3.0)                   # tf_interpolate.py:48
    from gspn_amd import _lib as L
    lib = L.lib()
    with torch.no_grad():
        o = three_interpolate(tp, idx, w)
    go = torch.randn_like(o)
    gtp = torch.empty_like(tp)
    idx_np, w_np, go_np = idx.cpu().numpy(), w.cpu().numpy(), go.cpu().numpy()
    have_ref = O.ref_lib() is not None
    ref_out = O.ref_three_interpolate(pts, idx_np, w_np) if have_ref else None
    leg = {"shape": "b=32, n=512, m=128, c=64 (interpolate.cpp:134; tf_interpolate.py:40-47)",
           "hip_ms": {"three_nn": _ev_time(lambda: three_nn(t1, t2), 3, 50), "three_interpolate": _ev_time(lambda: three_interpolate(tp, idx, w), 3, 50),
                      # (gradient launches go straight through the C ABI: an autograd backward inside a stream capture crashes this torch build)
                      "three_interpolate_grad": _ev_time(lambda: L.check(lib.gspn_threeinterpolate_grad(32, 512, 64, 128, L.ptr(go), L.ptr(idx), L.ptr(w), L.ptr(gtp),
                                                                                                        L.stream()), "three_interpolate_grad"), 3, 50)},
           "cpu_ms": {"three_nn": best(lambda: O.three_nn(x1, x2)),
                      "three_interpolate": best(lambda: (O.ref_three_interpolate if have_ref else O.three_interpolate)(pts, idx_np, w_np)),
                      "three_interpolate_grad": best(lambda: (O.ref_three_interpolate_grad if have_ref else O.three_interpolate_grad)(pts, idx_np, w_np, go_np))},
           "cpu_kind": {"three_nn": "port (oracle restatement of tf_interpolate.cpp:60-103; the harness's own threenn_cpu is not the op's formula)",
                        "three_interpolate": "reference (oracle/_ref: interpolate.cpp compiled as is)" if have_ref else "port",
                        "three_interpolate_grad": "reference (oracle/_ref)" if have_ref else "port"},
           "cores": 1,
           "hip_equals_reference_bits": bool(have_ref and np.array_equal(o.cpu().numpy(), ref_out))}
    leg["tf_interpolate.py loop (100 x three_interpolate)"] = {"hip_ms": 100 * leg["hip_ms"]["three_interpolate"], "cpu_ms": 100 * leg["cpu_ms"]["three_interpolate"]}
    res["3d_interpolation"] = leg
    # ---- tf_nndistance.py ----
    rng = np.random.RandomState(100)
    a = rng.randn(32, 16384, 3).astype(np.float32)
    c = rng.randn(32, 1024, 3).astype(np.float32)
    ta = torch.from_numpy(a).to(dev)
    tc = torch.from_numpy(c).to(dev)
    one1, one2 = torch.ones(32, 16384, device=dev), torch.ones(32, 1024, device=dev)   # d loss / d dist: loss = reduce_sum(reta) + reduce_sum(retc), tf_nndistance.py:59
    ga, gc = torch.empty_like(ta), torch.empty_like(tc)

    def hip_step():
        with torch.no_grad():
            d1, i1, d2, i2 = nn_distance(ta, tc)
        L.check(lib.gspn_nmdistance_grad(32, 16384, L.ptr(ta), 1024, L.ptr(tc), L.ptr(one1), L.ptr(i1), L.ptr(one2), L.ptr(i2), L.ptr(ga), L.ptr(gc), L.stream()),
                "nn_distance_grad")

    def cpu_step():
        d1, i1, d2, i2 = O.nn_distance(a, c, cpu_twin=True)
        O.nn_distance_grad(a, c, np.ones_like(d1), i1, np.ones_like(d2), i2)
    res["nn_distance"] = {"shape": "(32,16384,3) x (32,1024,3), forward + gradient (tf_nndistance.py:48-66)",
                          "hip_ms_per_step": _ev_time(hip_step, 3, 20), "cpu_ms_per_step": best(cpu_step, 2), "cores": 1,
                          "cpu_kind": "port (oracle twin of the reference's CPU kernel nnsearch, tf_nndistance.cpp:21-43, + its gradient loop)"}
    # ---- the reference's operating point ----
    keep = tf_util.get_variable_store()
    try:
        tf_util.set_variable_store(tf_util.VariableStore(device=dev, seed=6))
        nb, npt = 2, 18000
        xyz = torch.from_numpy(np.stack([np.random.default_rng(900 + i).random((npt, 3), dtype=np.float32) for i in range(nb)])).to(dev)
        col = torch.rand(nb, npt, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(1))

        def op_step():
            for p_ in tf_util.get_variable_store().parameters():
                p_.grad = None
            pn2_fea_extractor(xyz, col, 'op', True, 0.5).square().mean().backward()
        t = _time_steps(op_step, 3, 10)
        res["operating_point"] = {"workload": "models/config.py:14-19: batch 2 x 18000 points, pn2_fea_extractor 3 x SA + 3 x FP fwd+bwd, eager, geometry inline",
                                  "ms_per_step": t * 1e3, "scenes_per_s": nb / t}
    finally:
        tf_util.set_variable_store(keep)
        torch.cuda.empty_cache()
    return res


def _latest_profiles(*suffixes):
    """profiles/rNN_<suffix> of the newest round that has it (labels follow the file actually on disk)"""
    out = []
    for suf in suffixes:
        for r in range(9, 0, -1):
            q = os.path.join(ROOT, "profiles", "r%02d_%s" % (r, suf))
            if os.path.exists(q):
                out.append("profiles/" + os.path.basename(q))
                break
    return ", ".join(out)


def _time_steps(fn, warm, reps):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def _fps_leg_roofline(step_fn, reps=3):
    """FPS launches of `reps` runs of a leg, bracketed by HIP events (tf_sampling.PROFILE): the leg's dominant geometry kernel against
    SURVEY 8(d)'s algorithmic bytes 20*b*(m-1)*n + 4*b*m"""
    from gspn_amd import tf_sampling
    tf_sampling.PROFILE, tf_sampling.PROFILE_BUDGET[0], tf_sampling.PROFILE_MIN_N = [], 1 << 30, 0
    try:
        for _ in range(reps):
            step_fn()
        torch.cuda.synchronize()
        ev = tf_sampling.PROFILE
    finally:
        tf_sampling.PROFILE = None
    if not ev:
        return None
    big = max(ev, key=lambda e: e[2] * e[3] * e[4])
    same = [e for e in ev if e[2:] == big[2:]]
    ms = float(np.mean([e[0].elapsed_time(e[1]) for e in same]))
    b_, n_, m_ = big[2:]
    alg = 20.0 * b_ * (m_ - 1) * n_ + 4.0 * b_ * m_
    ach = alg / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "farthest point sampling %d x %d -> %d (largest FPS launch of the leg)" % (b_, n_, m_), "avg_launch_ms": ms,
            "us_per_pick": ms * 1e3 / m_, "algorithmic_bytes_per_launch": alg, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "achieved_is": "effective rate = algorithmic bytes / time (on-chip resident kernel)"}


def other_configs(xyz, col, dev):
    """Timed legs at the other BASELINE configs' per-GPU shapes (not the headline metric; same kernels, eager, geometry inline)."""
    from gspn_amd import tf_util
    from gspn_amd.pointnet_util import pointnet_sa_module
    from gspn_amd.proposal_head import chamfer_recons_loss, multi_encoding_net
    keep = tf_util.get_variable_store()
    out = {}
    try:
        b = xyz.shape[0]
        # configs[1]: batch 8 x 32768, SA(1024, 0.1, 32, [64,64,128]) forward only (BN training mode, like the reference's train graph)
        tf_util.set_variable_store(tf_util.VariableStore(device=dev, seed=2))

        def c1():
            with torch.no_grad():
                pointnet_sa_module(xyz, col, 1024, 0.1, 32, [64, 64, 128], None, False, True, 0.5, 'c1')
        t = _time_steps(c1, 3, 10)
        out["configs[1]"] = {"workload": "8 x 32768 pts, SA(1024, 0.1, 32, [64,64,128]) fwd only, FPS + ball query + group + 3 layers + max-pool inline on one stream",
                             "ms_per_step": t * 1e3, "scenes_per_s": b / t, "roofline": _fps_leg_roofline(c1)}
        # configs[3], one GPU's shard (8 of the 32 scenes): multi_encoding_net (model_rpointnet.py:377) + Chamfer on 2048 x (512, 512) clouds
        tf_util.set_variable_store(tf_util.VariableStore(device=dev, seed=3))
        gen = torch.Generator(device=dev).manual_seed(9)
        pred0 = torch.randn(b * 256, 512, 3, device=dev, generator=gen)
        gt = torch.randn(b * 256, 512, 3, device=dev, generator=gen)
        mask = (torch.rand(b * 256, device=dev, generator=gen) > 0.2).float()
        col_g = col.clone().requires_grad_(True)
        side = torch.cuda.Stream(dev)

        def c3():
            for p_ in tf_util.get_variable_store().parameters():
                p_.grad = None                                   # (as c4: a step starts from cleared gradients -- r03's leg accumulated into them, 35 torch adds a step)
            pred = pred0.clone().requires_grad_(True)
            # the Chamfer term has no input in common with the encoder: it runs on its own stream beside the encoder's FPS (8 CUs busy, 248 idle),
            # and autograd runs its backward there as well (tools/c3_leg.py C3_SIDE=0 / 1: 8.69 -> 8.54 ms per step)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ch = chamfer_recons_loss(pred, gt, mask)
            _, new_points, _, _ = multi_encoding_net(xyz, col_g, 256, [0.5, 1.0, 1.5], [256, 256, 512], [[64, 128, 256]] * 3, [], True, 0.5, 'c3', use_xyz=True)
            torch.cuda.current_stream().wait_stream(side)
            loss = new_points.mean() + ch
            loss.backward()
            col_g.grad = None
        t = _time_steps(c3, 2, 5)
        rows = b * 256 * (256 + 256 + 512)
        gf = 3 * 2.0 * rows * (6 * 64 + 64 * 128 + 128 * 256)
        out["configs[3] per-GPU shard"] = {"workload": "8 x 32768 pts: multi_encoding_net(256 seeds, r .5/1/1.5, ns 256/256/512, mlp [64,128,256] x3, use_xyz) + "
                                                      "Chamfer nn_distance on 2048 x (512,512) clouds (on a second stream beside the encoder), fwd+bwd, eager, geometry inline",
                                           "ms_per_step": t * 1e3, "scenes_per_s": b / t, "grouped_rows": rows, "mlp_TFLOPs_fwd_bwd_over_whole_step": gf / t / 1e12,
                                           "roofline": {"bound": "mfma_f32", "kernels": "the three 6 -> 64 -> 128 -> 256 stacks (forward, pass A, pass B): 3 x 2 x rows x (6*64 + 64*128 + "
                                                        "128*256) flops over the WHOLE leg (geometry, pooling and Chamfer included in the time)",
                                                        "achieved": gf / t / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": gf / t / 1e12 / MFMA_F32_PEAK_TFLOPS,
                                                        "per_kernel": _latest_profiles("c3_kernel_stats.csv", "c3_sq_pmc_by_kernel.txt")}}
        # configs[4], one GPU's shard of the SA/FP part (8 of the 64 scenes): 65536-pt scenes through pn2_fea_extractor, fwd+bwd
        from gspn_amd.fea_extractor import pn2_fea_extractor
        tf_util.set_variable_store(tf_util.VariableStore(device=dev, seed=4))
        n5 = 65536
        xyz5 = torch.from_numpy(np.stack([np.random.default_rng(5000 + i).random((n5, 3), dtype=np.float32) for i in range(b)])).to(dev)
        col5 = torch.rand(b, n5, 3, device=dev, generator=gen)
        st5 = {}

        def c4():
            for p_ in tf_util.get_variable_store().parameters():
                p_.grad = None
            o = pn2_fea_extractor(xyz5, col5, 'c4', True, 0.5)
            o.square().mean().backward()
        t = _time_steps(c4, 2, 5)
        out["configs[4] per-GPU shard (SA/FP part)"] = {"workload": "8 x 65536 pts: pn2_fea_extractor 3 x SA + 3 x FP fwd+bwd, eager, geometry inline (FPS on the multi-CU kernel, "
                                                                    "4 CUs per scene)", "ms_per_step": t * 1e3, "scenes_per_s": b / t, "roofline": _fps_leg_roofline(c4)}
        # configs[4], the proposal half on the same 65536-pt scenes: the context encoder as model_rpointnet.py:377 calls it (given seeds, a
        # stop-gradient shift) + the Chamfer reconstruction loss (:1346-1355), fwd+bwd
        tf_util.set_variable_store(tf_util.VariableStore(device=dev, seed=5))
        from gspn_amd.tf_sampling import farthest_point_sample
        seeds = farthest_point_sample(256, xyz5)
        shift = torch.randn(b, 256, 3, device=dev, generator=gen) * 0.05

        def c4p():
            for p_ in tf_util.get_variable_store().parameters():
                p_.grad = None
            pred = pred0.clone().requires_grad_(True)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ch = chamfer_recons_loss(pred, gt, mask)
            _, fea, _, _ = multi_encoding_net(xyz5, col5, 256, [0.5, 1.0, 1.5], [256, 256, 512], [[64, 128, 256]] * 3, [], True, 0.5, 'c4p', use_xyz=True,
                                              shift_pred=shift, fps_idx=seeds)
            torch.cuda.current_stream().wait_stream(side)
            (fea.mean() + ch).backward()
        t = _time_steps(c4p, 2, 5)
        out["configs[4] per-GPU shard (proposal part)"] = {"workload": "8 x 65536 pts: multi_encoding_net(256 given seeds, stop-gradient shift, r .5/1/1.5, ns 256/256/512, mlp "
                                                                       "[64,128,256] x3, use_xyz) + Chamfer on 2048 x (512,512) clouds (second stream), fwd+bwd, eager", "ms_per_step": t * 1e3,
                                                           "scenes_per_s": b / t,
                                                           "roofline": {"bound": "mfma_f32", "achieved": gf / t / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                                        "frac": gf / t / 1e12 / MFMA_F32_PEAK_TFLOPS, "kernels": "as configs[3]: the layers' flops over the whole leg"}}
    finally:
        tf_util.set_variable_store(keep)
        torch.cuda.empty_cache()
    return out


def cpu_baseline(xyz_np, col_np, full=False):
    """the CPU port (oracle geometry + torch-CPU MLP stand-in) on a bounded sample of the same workload: one full fwd+bwd step,
    (i) on one thread -- how the reference's own CPU ops run (threenn_cpu / nnsearch are plain loops) -- on `sample` scenes of one batch and
    (ii) on all cores on the whole batch: OpenMP over scene x query inside the oracle loops, torch intra-op threads for the MLP stand-in.
    Default: 2 scenes / median of 3 after 1 warm-up each (~15-20 s of CPU work, so that the whole default run stays under ~30 s);
    full (--detail): SURVEY 8(d)'s 4 scenes / median of 5."""
    from oracle import cpu_pipeline
    nthr = torch.get_num_threads()
    cores = os.cpu_count() or 1
    sample, reps = (4, 5) if full else (2, 3)
    med = lambda f: float(np.median([f() for _ in range(reps + 1)][1:]))
    torch.set_num_threads(1)
    try:
        t1 = med(lambda: cpu_pipeline.run_step(xyz_np[:sample], col_np[:sample], mt=False))
        mlp_threads = min(cores, 32)                  # torch's small fp32 GEMMs stop scaling (and then slow down) far below 256 threads
        torch.set_num_threads(mlp_threads)
        tall = med(lambda: cpu_pipeline.run_step(xyz_np, col_np, mt=True))
    finally:
        torch.set_num_threads(nthr)
    return {"value": sample / t1, "unit": "scenes/s", "cores": 1, "kind": "port",
            "sample": "1 fwd+bwd step on %d of the batch's 8 scenes (32768 pts), median of %d after 1 warm-up, 1 thread: C oracle geometry + torch-CPU fp32 "
                      "MLP stand-in; %.1f s/step" % (sample, reps, t1),
            "all_cores": {"value": xyz_np.shape[0] / tall, "cores": cores, "omp_threads": cores, "mlp_threads": mlp_threads,
                          "note": "the whole batch of %d scenes, median of %d after 1 warm-up: OpenMP over scenes (FPS, scatter-add gradients) and over "
                                  "scene x query (ball query, grouping, 3-NN, interpolation); torch intra-op threads for the MLP stand-in; %.2f s per "
                                  "step" % (xyz_np.shape[0], reps, tall)}}


if __name__ == "__main__":
    main()
