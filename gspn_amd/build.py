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


# translation units whose arithmetic depends on GSPN_DIST_POLICY (common.h: dist2_cuda); a policy-variant library holds only these
POLICY_SOURCES = ("sampling.hip", "sampling_multi.hip", "grouping.hip", "nndistance.hip")


def policy_lib_path(policy):
    return os.path.join(LIBDIR, "libgspn_hip_p%d.so" % policy)


def build(force=False, verbose=False, policy=None, variant=None, extra_flags=()):
    """policy=None: the product library (GSPN_DIST_POLICY 2, common.h).  policy=0/1/2: a VARIANT library
    lib/libgspn_hip_p<policy>.so holding only the policy-dependent translation units (FPS, ball query / grouping, nn_distance), built
    with -DGSPN_DIST_POLICY=<policy> into its own object directory -- test infrastructure for tests/test_gpu_policy.py; the product
    library is never touched by it."""
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(HERE, "..", "include", "*.h")))
    flags = list(FLAGS) + os.environ.get("GSPN_EXTRA_HIPCC_FLAGS", "").split()
    obj_dir, lib_path = OBJ, LIB
    if policy is not None:
        flags.append("-DGSPN_DIST_POLICY=%d" % policy)
        obj_dir, lib_path = os.path.join(OBJ, "p%d" % policy), policy_lib_path(policy)
        srcs = [s for s in srcs if os.path.basename(s) in POLICY_SOURCES]
    if variant is not None:                      # tools/: a complete library built with extra -D switches (kernel ablations), own object directory
        flags += list(extra_flags)
        obj_dir, lib_path = os.path.join(OBJ, "v_" + variant), os.path.join(LIBDIR, "libgspn_hip_%s.so" % variant)
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(obj_dir, os.path.basename(s)[:-4] + ".o")
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
    if force or jobs or _newer(lib_path, objs):
        run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib_path] + objs)
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
