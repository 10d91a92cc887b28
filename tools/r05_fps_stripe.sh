#!/bin/bash
# FPS with every cell striped over the 16 waves (sampling_stripe.hip) against one cell per wave: index-exactness tests, then times on U / S / D
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_geometry.py tests/test_gpu_policy.py tests/test_gpu_policy3.py tests/test_gpu_reference_kernels.py tests/test_gpu_golden.py -x -q -k "fps or FPS or policy or geometry_is or golden or reference" 2>&1 | tail -5
for s in 0 1; do
  for k in U S D; do
    GSPN_FPS_STRIPE=$s python - <<PY 2>&1 | grep -v amdgpu.ids
import sys, torch, numpy as np
sys.path.insert(0, '.')
from gspn_amd import synth
from gspn_amd.tf_sampling import farthest_point_sample
for n, m in ((32768, 2048), (32768, 1024), (20000, 1024)):
    xyz = torch.from_numpy(synth.batch("$k", 8, n, 0)).cuda()
    farthest_point_sample(m, xyz); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        farthest_point_sample(m, xyz)
    e1.record(); torch.cuda.synchronize()
    print("stripe=$s $k fps 8x%d -> %d: %.1f us (incl. pre-pass)  %.3f us/pick" % (n, m, e0.elapsed_time(e1) * 100, e0.elapsed_time(e1) * 100 / m))
PY
  done
done
