cd /tmp && export TMPDIR=/tmp
cd /root/repo
timeout 300 python bench.py > gpurun_out/r01_bench_default.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_full -- python bench.py --no-cpu-baseline > gpurun_out/r01_bench_under_rocprof.log 2>&1
f=$(find gpurun_out/prof_full -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r01_bench_kernel_stats.csv
rm -rf gpurun_out/prof_full
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python tools/fps_only.py > gpurun_out/pmc_$c.log 2>&1
done
python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/r01_fps_pmc.json 2> gpurun_out/pmc_summary.err
cp $(find gpurun_out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) gpurun_out/r01_pmc_fetch_counters.csv
cp $(find gpurun_out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) gpurun_out/r01_pmc_write_counters.csv
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
timeout 600 python tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r01_mlp_layer_bench.txt
