timeout 600 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_sa_variants.py -x -q -k "three_nn or nn" 2>&1 | tail -2
GSPN_NN_GRID=64 timeout 600 python -m pytest tests/test_gpu_geometry.py -x -q -k "three_nn" 2>&1 | tail -1
run() { env "$@" GSPN_BENCH_LAYERS_ONLY=1 python bench.py --no-cpu-baseline --no-extra --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],3))"; }
echo "layers only: $(run A=1)"
for c in 0 512 256 128 64 32; do echo "nn side, grid cap $c: $(run GSPN_BENCH_SIDE=nn GSPN_NN_GRID=$c)"; done
