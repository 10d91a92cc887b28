cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
python bench.py 2> $O/r04_bench_default.log | tail -1 > $O/r04_bench_line.json
rm -rf gpurun_out/kstats
(cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kstats -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$O/r04_bench_under_rocprof.log 2>&1)
cp $(find gpurun_out/kstats -name "*kernel_stats.csv" | head -1) $O/r04_bench_kernel_stats.csv; rm -rf gpurun_out/kstats
bash tools/trace_layers.sh > /dev/null 2>&1; cp gpurun_out/layers_timeline.txt $O/r04_layers_timeline.txt; rm -rf gpurun_out/ltrace
python tools/layer_table.py 5 2>/dev/null > $O/r04_mlp_layer_table.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
