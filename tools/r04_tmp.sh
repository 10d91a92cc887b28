ms() { python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],3), round(r['median_ms_per_step'],3))"; }
for rep in 1 2; do for k in S U; do for c in 1 0; do echo "$k cells=$c  $(GSPN_BALL_CELLS=$c python bench.py --data $k --no-cpu-baseline --no-extra --steps 200 2>/dev/null | tail -1 | ms)"; done; done; done
