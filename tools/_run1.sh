cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu -k "mlp or sa_module or pool or stack" 2>&1 | tail -3 > gpurun_out/gpu_tests.log
timeout 600 python tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids | grep "L3p" > gpurun_out/mlp_layer_bench.txt
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2>&1
timeout 300 python bench.py --no-cpu-baseline >> gpurun_out/bench.log 2>&1
