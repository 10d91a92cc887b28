#!/bin/bash
# everything profiles/ holds for round 6, from the current tree (run on the GPU box; results land in gpurun_out/r06/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
# 1. the bench line the driver would record
python bench.py --detail 2> $O/r06_bench_default.log | tail -1 > $O/r06_bench_line.json; cp bench_detail.json $O/r06_bench_detail.json
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) 2> $O/r06_bench_driver_form.log | tail -1 > $O/r06_bench_line_driver_form.json
# 2. per-kernel table of the same command (kernel durations are what to read: the step is slower under the profiler)
rm -rf gpurun_out/kstats
(cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kstats -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$O/r06_bench_under_rocprof.log 2>&1)
cp $(find gpurun_out/kstats -name "*kernel_stats.csv" | head -1) $O/r06_bench_kernel_stats.csv; rm -rf gpurun_out/kstats
# 3. one captured step, kernel by kernel; per-layer table of the GEMM launches
bash tools/trace_layers.sh > /dev/null 2>&1; cp gpurun_out/layers_timeline.txt $O/r06_layers_timeline.txt; rm -rf gpurun_out/ltrace
python tools/layer_table.py 5 2>/dev/null > $O/r06_mlp_layer_table.txt
# 4. SQ counters per kernel of one eager, un-overlapped bench step: EVERY kernel (the r03 name filter missed the lean / fused kernels)
(cd /tmp; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-graph --no-overlap > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq.log 2>&1)
python tools/pmc_kernels.py gpurun_out/pmc_sq > $O/r06_sq_pmc_by_kernel.txt 2>&1; rm -rf gpurun_out/pmc_sq
# 5. memory-side bytes per step by kernel (two separate --pmc passes), eager un-overlapped steps
STEPS=4
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp; rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sb_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 0 --no-cpu-baseline --no-extra --no-graph --no-overlap > $GRAFT_REPO_ROOT/gpurun_out/pmc_sb_$c.log 2>&1)
done
python tools/pmc_step_bytes.py gpurun_out/pmc_sb_FETCH_SIZE gpurun_out/pmc_sb_WRITE_SIZE $STEPS $O/r06_step_bytes.json > $O/r06_step_bytes.txt 2>&1; rm -rf gpurun_out/pmc_sb_FETCH_SIZE gpurun_out/pmc_sb_WRITE_SIZE
# 6. memory-side bytes: the stand-alone ops, the FPS kernels
bash tools/pmc_ops.sh > $O/pmc_ops.out 2>&1; cp gpurun_out/r04_ops_pmc.json $O/r06_ops_pmc.json 2>/dev/null
bash tools/pmc_fps.sh > $O/pmc_fps.out 2>&1; cp gpurun_out/r02_fps_pmc_32768.json $O/r06_fps_pmc.json 2>/dev/null; cp gpurun_out/r02_fps_pmc_65536.json $O/r06_fps_multi_pmc_65536.json 2>/dev/null
# 7. instruction counts per kernel of one eager step and the SIMD floor they imply
(cd /tmp; rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_ins -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-graph --no-overlap > $GRAFT_REPO_ROOT/gpurun_out/pmc_ins.log 2>&1)
python tools/simd_budget.py gpurun_out/pmc_ins 6 > $O/r06_simd_budget.txt 2>&1; rm -rf gpurun_out/pmc_ins
# 8. the configs[3] shard leg: per-kernel stats + SQ counters (separate runs)
(cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c3prof -o k -- python $GRAFT_REPO_ROOT/tools/c3_leg.py > $GRAFT_REPO_ROOT/$O/r06_c3_under_rocprof.log 2>&1)
cp $(find gpurun_out/c3prof -name "*kernel_stats.csv" | head -1) $O/r06_c3_kernel_stats.csv; rm -rf gpurun_out/c3prof
(cd /tmp; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
   --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c3sq -o p -- python $GRAFT_REPO_ROOT/tools/c3_leg.py > $GRAFT_REPO_ROOT/gpurun_out/c3sq.log 2>&1)
python tools/pmc_kernels.py gpurun_out/c3sq > $O/r06_c3_sq_pmc_by_kernel.txt 2>&1; rm -rf gpurun_out/c3sq
# 9. kernel stats of the bench on the room scenes (S) and the duplicate-heavy clouds (D)
for K in S D; do bash tools/r04_kstats.sh $K > $O/r06_kstats_$K.txt 2>&1; cp gpurun_out/r04_kstats_$K.csv $O/ 2>/dev/null; done
# 10. the drop-in gradient symbols with and without a workspace, from the detail file; the geometry tax part by part
python - <<'PY' > $O/r06_dropin_ws.txt
import json
d = json.load(open("gpurun_out/r06/r06_bench_detail.json"))
print("stand-alone gradient launches at the bench shapes (bench.py --detail: roofline_ops; graph-timed, 20 launches per replay); frac = algorithmic bytes / time / 8 TB/s")
for o in d["roofline_ops"]["ops"]:
    if "grad" in o["op"] or "inverse_lists" in o["op"]:
        print("%-125s %-34s %8.1f us   frac %.3f" % (o["op"][:125], o.get("shape", "")[:34], o["avg_launch_ms"] * 1e3, o.get("frac", 0)))
PY
bash tools/side_costs.sh > $O/r06_side_costs_final.txt 2>&1
ls -la $O
