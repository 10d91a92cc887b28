#!/bin/bash
# SQ counters per kernel of one eager, un-overlapped bench step (rocprofv3 --pmc, one run; no trace domains alongside)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --output-format csv -d gpurun_out/pmc_sq -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-graph --no-overlap > gpurun_out/pmc_sq.log 2>&1
python tools/pmc_kernels.py gpurun_out/pmc_sq wgrad_stream mlp_fwd mlp_bwd_data fps_ ball_query three_nn pool fp_concat sa_ bn_finalize bwd_coef knn > gpurun_out/r02_sq_pmc_by_kernel.txt 2>&1
tail -40 gpurun_out/r02_sq_pmc_by_kernel.txt
