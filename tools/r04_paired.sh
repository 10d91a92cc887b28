# pipeline shape of the bench's geometry prefetch: paired (two batches every other step, default) against one batch per step; 20-step and 100-step forms
for e in "GSPN_BENCH_PAIRED=1" "GSPN_BENCH_PAIRED=0" "GSPN_BENCH_PAIRED=1" "GSPN_BENCH_PAIRED=0"; do
  echo "$e  20/3: $(env $e python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],4), round(d['median_ms_per_step'],4))")   100/10: $(env $e python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],4), round(d['median_ms_per_step'],4))")"
done
