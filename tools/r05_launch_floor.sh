#!/bin/bash
# per-kernel cost of a dependent chain under HIP runtime switches; every variant under its own timeout (one of them hung the first attempt)
mkdir -p gpurun_out
for e in "TAG=default" "TAG=AMD_OPT_FLUSH0 AMD_OPT_FLUSH=0" "TAG=PKTCAP0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "TAG=DEVKERNARG HIP_FORCE_DEV_KERNARG=1" \
         "TAG=ACTIVEWAIT ROC_ACTIVE_WAIT_TIMEOUT=1000" "TAG=SYSSCOPE0 ROC_SYSTEM_SCOPE_SIGNAL=0" "TAG=GRAPHBATCH DEBUG_HIP_GRAPH_BATCH_SIZE=256"; do
  timeout 90 env $e python tools/r05_launch_floor.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05_launch_floor.txt
  echo "($e rc=$?)" | tee -a gpurun_out/r05_launch_floor.txt
done
