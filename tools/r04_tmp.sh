python -m pytest tests/test_gpu_mfma_split.py tests/test_gpu_mlp.py -q -m gpu -x 2>&1 | tail -4
for v in 0 1; do GSPN_FWD_DIRECT=$v python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/direct_bench_$v.json; done
python - <<'PY'
import json
for v in (0, 1):
    d = json.load(open("gpurun_out/direct_bench_%d.json" % v))
    print("GSPN_FWD_DIRECT=%d" % v, round(d["value"]), round(d["ms_per_step"], 4), d["roofline_mlp"]["ms_by_pass"], d["other_configs"]["configs[3] per-GPU shard"]["ms_per_step"], d["other_configs"]["configs[4] per-GPU shard (proposal part)"]["ms_per_step"])
PY
