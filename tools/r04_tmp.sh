mkdir -p gpurun_out
(cd tools/clk && for w in 3; do echo "---- $w workgroups per CU (grid = 256 x $w persistent workgroups)"; timeout 300 ./bf16_split $w; done) > gpurun_out/r04_bf16_split_microbenchmark.txt 2>&1
GSPN_MFMA_SPLIT=1 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_reference_kernels.py 2>&1 | tail -15 > gpurun_out/split_tests.txt
GSPN_MFMA_SPLIT=1 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/split_tests_all.txt
for v in 0 1; do GSPN_MFMA_SPLIT=$v python bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 > gpurun_out/split_bench_$v.json; done
python - <<'PY'
import json
for v in (0, 1):
    d = json.load(open("gpurun_out/split_bench_%d.json" % v))
    print("GSPN_MFMA_SPLIT=%d" % v, d["value"], d["ms_per_step"], {k: d[k] for k in d if "layer" in k.lower()})
PY
