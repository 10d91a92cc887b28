#!/bin/bash
# everything profiles/ holds for round 3, from the current tree (run on the GPU box; results land in gpurun_out/r03/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
# 1. the bench line the driver would record
python bench.py 2> $O/r03_bench_default.log | tail -1 > $O/r03_bench_line.json
# 2. per-kernel table of the same command (kernel durations are what to read: the step is slower under the profiler)
rm -rf gpurun_out/kstats
(cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kstats -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$O/r03_bench_under_rocprof.log 2>&1)
cp $(find gpurun_out/kstats -name "*kernel_stats.csv" | head -1) $O/r03_bench_kernel_stats.csv; rm -rf gpurun_out/kstats
# 3. one captured step, kernel by kernel; per-layer table of the GEMM launches
bash tools/trace_layers.sh > /dev/null 2>&1; cp gpurun_out/layers_timeline.txt $O/r03_layers_timeline.txt; rm -rf gpurun_out/ltrace
python tools/layer_table.py 5 2>/dev/null > $O/r03_mlp_layer_table.txt
# 4. the configs[3] shard leg: per-kernel stats + SQ counters (separate runs)
(cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c3prof -o k -- python $GRAFT_REPO_ROOT/tools/c3_leg.py > $GRAFT_REPO_ROOT/$O/r03_c3_under_rocprof.log 2>&1)
cp $(find gpurun_out/c3prof -name "*kernel_stats.csv" | head -1) $O/r03_c3_kernel_stats.csv; rm -rf gpurun_out/c3prof
(cd /tmp; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
   --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c3sq -o p -- python $GRAFT_REPO_ROOT/tools/c3_leg.py > $GRAFT_REPO_ROOT/gpurun_out/c3sq.log 2>&1)
python tools/pmc_kernels.py gpurun_out/c3sq wgrad_stream mlp_fwd mlp_bwd_data bnrelu_maxpool nm_distance ball_query pool_rsum preagg sa_ > $O/r03_c3_sq_pmc_by_kernel.txt 2>&1; rm -rf gpurun_out/c3sq
(cd /tmp; rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
   --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c3sq2 -o p -- python $GRAFT_REPO_ROOT/tools/c3_leg.py > $GRAFT_REPO_ROOT/gpurun_out/c3sq2.log 2>&1)
python tools/pmc_kernels2.py gpurun_out/c3sq2 wgrad_stream mlp_fwd mlp_bwd_data > $O/r03_c3_sq_insts_by_kernel.txt 2>&1; rm -rf gpurun_out/c3sq2
# 5. SQ counters per kernel of one eager, un-overlapped bench step
(cd /tmp; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-graph --no-overlap > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq.log 2>&1)
python tools/pmc_kernels.py gpurun_out/pmc_sq wgrad_stream mlp_fwd mlp_bwd_data fps_ ball_query three_nn pool fp_concat sa_ bn_finalize bwd_coef preagg csr > $O/r03_sq_pmc_by_kernel.txt 2>&1; rm -rf gpurun_out/pmc_sq
# 6. memory-side bytes: the stand-alone ops, the FPS kernels
bash tools/pmc_ops.sh > $O/pmc_ops.out 2>&1; cp gpurun_out/r03_ops_pmc.json $O/
bash tools/pmc_fps.sh > $O/pmc_fps.out 2>&1; cp gpurun_out/r02_fps_pmc_32768.json $O/r03_fps_pmc.json 2>/dev/null; cp gpurun_out/r02_fps_pmc_65536.json $O/r03_fps_multi_pmc_65536.json 2>/dev/null
ls -la $O
# 7. instruction counts per kernel of one eager step and the SIMD floor they imply (fp32 MFMA and vector instructions add on a SIMD)
(cd /tmp; rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_ins -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-graph --no-overlap > $GRAFT_REPO_ROOT/gpurun_out/pmc_ins.log 2>&1)
python tools/simd_budget.py gpurun_out/pmc_ins 6 > $O/r03_simd_budget.txt 2>&1; rm -rf gpurun_out/pmc_ins
# 8. the stand-alone microbenchmarks behind DESIGN.md's statements on what a SIMD overlaps
for b in mfma_clock overlap overlap_rw coexec pattern; do [ -x tools/clk/$b ] && (echo "== $b"; tools/clk/$b) ; done > $O/r03_clk_microbenchmarks.txt 2>&1
ls -la $O
