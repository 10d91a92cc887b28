"""does the per-step time drift over a long run? (diagnostic: runs bench.main's loop pieces via env and prints per-10-step wall)"""
import os, subprocess, sys, json
for steps in (10, 20, 40, 80):
    for prof in ("1", "0"):
        env = dict(os.environ, GSPN_BENCH_PAIRED="0", GSPN_BENCH_NOPROFILE=prof)
        out = subprocess.run([sys.executable, "bench.py", "--steps", str(steps), "--warmup", "3", "--no-cpu-baseline", "--no-extra"], env=env, capture_output=True, text=True).stdout
        r = json.loads(out.strip().splitlines()[-1])
        print("steps %3d noprofile=%s  ms/step %.3f  host %.3f" % (steps, prof, r["ms_per_step"], r["host_enqueue_ms_per_step"]), flush=True)
