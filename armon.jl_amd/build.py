"""Build libarmon_hip.so (gfx950 only) in-tree with hipcc. No torch, no cmake: explicit commands.

    python armon.jl_amd/build.py [--force] [--verbose]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libarmon_hip.so")
# A/B build: the product library + the measured-and-rejected kernel forms (-DARMON_ALT_KERNELS: whole-cycle kernels, LDS X
# march, one-cell-per-lane DPP sweep). Only the tests and tools that exercise those load it. (Both libraries carry the exact
# arithmetic in its strict form since round 4: quotients that can fall below the normal range take the IEEE expansion,
# csrc/physics.hpp.)
OUT_ALT = os.path.join(HERE, "libarmon_hip_alt.so")
ALT_SOURCES = ("fused_sweep_f64.hip", "fused_sweep_f32.hip")     # the translation units the define changes
ALT_FLAGS = ["-DARMON_ALT_KERNELS"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# -ffp-contract=off: the "exact" kernels must evaluate one IEEE op per source op (bit parity with the
# CPU oracle); the tuned kernels ask for FMAs explicitly.
CXXFLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off",
            "-Wall", "-Wextra", "-Wno-unused-parameter", "-fvisibility=hidden",
            "-DARMON_BUILDING_LIB"] + os.environ.get("ARMON_EXTRA_FLAGS", "").split()


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hs.append(os.path.join(HERE, "..", "include", "armon_hip.h"))
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, alt=True):
    objs, objs_alt, jobs = [], [], []
    hdrs = headers()
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in sources():
        obj = os.path.join(HERE, "build", src[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [os.path.join(CSRC, src)] + hdrs + [__file__]):
            jobs.append([HIPCC, *CXXFLAGS, "-c", os.path.join(CSRC, src), "-o", obj])
        if alt and src in ALT_SOURCES:
            obj_alt = os.path.join(HERE, "build", src[:-4] + "_alt.o")
            if force or _stale(obj_alt, [os.path.join(CSRC, src)] + hdrs + [__file__]):
                jobs.append([HIPCC, *CXXFLAGS, *ALT_FLAGS, "-c", os.path.join(CSRC, src), "-o", obj_alt])
            obj = obj_alt
        objs_alt.append(obj)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    for out, group in ((OUT, objs),) + (((OUT_ALT, objs_alt),) if alt else ()):
        if jobs or force or _stale(out, group):
            run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *group,
                 "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined"])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, alt="--no-alt" not in sys.argv))
