# clocks and power while the bench runs (is the step power-limited?): rocm-smi samples during a 3000-step run, then idle
(python bench.py --steps 20000 --warmup 10 --no-cpu-baseline --no-extra > gpurun_out/power_bench.json 2>/dev/null) &
BP=$!
sleep 30
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power\|fclk" | tr '\n' ' '; echo; sleep 0.5; done
wait $BP
echo "idle:"; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | tr '\n' ' '; echo
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
python -c "import json; d=json.load(open('gpurun_out/power_bench.json')); print('20000 steps:', round(d['value']), d['ms_per_step'], d['median_ms_per_step'])"
