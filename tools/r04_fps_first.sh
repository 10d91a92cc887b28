# A/B of the two-phase geometry submission (bench.py: FPS of SA level 1 for every batch of a group before the rest of any): the driver's 20-step form and the default
for v in 0 1 0 1; do
  echo "GSPN_BENCH_FPS_FIRST=$v  20/3: $(GSPN_BENCH_FPS_FIRST=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],4), round(d['median_ms_per_step'],4))")   100/10: $(GSPN_BENCH_FPS_FIRST=$v python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],4), round(d['median_ms_per_step'],4))")"
done
