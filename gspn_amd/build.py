"""Builds gspn_amd/lib/libgspn_hip.so from gspn_amd/csrc/*.hip with hipcc for gfx950.

In-tree, explicit hipcc (cross-compiles without a GPU).  ``python -m gspn_amd.build`` or
``gspn_amd.build.build()``.  Objects are rebuilt only when their sources are newer.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgspn_hip.so")

ARCH = "gfx950"
# -ffp-contract=off : the only fused multiply-adds are the ones spelled out in common.h (bit parity)
# -munsafe-fp-atomics: hardware global_atomic_add_f32 for the scatter-add gradients
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, policy=None):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(HERE, "..", "include", "*.h")))
    flags = list(FLAGS) + os.environ.get("GSPN_EXTRA_HIPCC_FLAGS", "").split()
    if policy is not None:
        flags.append("-DGSPN_DIST_POLICY=%d" % policy)
    hipcc = _hipcc()
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            jobs.append([hipcc] + flags + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stdout))
        if verbose and r.stdout.strip():
            print(r.stdout)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _newer(LIB, objs):
        run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
