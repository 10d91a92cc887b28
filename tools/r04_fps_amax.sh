# FPS of SA level 1 with the per-round batch cap FPS_AMAX = 4 / 6 (default) / 8: stand-alone time and index-exactness
cat > /tmp/fps_t.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
from gspn_amd import tf_sampling as TS
for kind in ("U", "S", "D"):
    xyz = torch.from_numpy(bench.synth(8, 32768, 0, kind)[0]).cuda()
    out = TS.farthest_point_sample(2048, xyz); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): TS.farthest_point_sample(2048, xyz)
    e1.record(); torch.cuda.synchronize()
    import hashlib
    print(kind, "8 x 32768 -> 2048: %.1f us (incl. pre-pass)  sha %s" % (e0.elapsed_time(e1) * 100, hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12]))
PY
for v in _amax6 "" _amax10 _amax12 _amax16; do echo "== libgspn_hip$v.so"; GSPN_HIP_LIB=$GRAFT_REPO_ROOT/gspn_amd/lib/libgspn_hip$v.so python /tmp/fps_t.py 2>&1 | grep "8 x"; done
