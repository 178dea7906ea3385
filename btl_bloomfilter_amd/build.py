"""Build the HIP shared library in-tree:  python -m btl_bloomfilter_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  Translation units are compiled in parallel into
btl_bloomfilter_amd/_build/ and linked into libbtlbf.so; the .so is git-ignored but travels to the GPU
box with the repo snapshot (the objects do not: .gpurunignore)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# BTLBF_BUILD_TAG=<tag> (with BTLBF_CXXFLAGS) builds a diagnostic variant next to the product library
TAG = os.environ.get("BTLBF_BUILD_TAG", "")
OBJ = os.path.join(HERE, "_build" + ("_" + TAG if TAG else ""))
LIB = os.path.join(HERE, "libbtlbf%s.so" % ("_" + TAG if TAG else ""))
HEADERS = ["internal.hpp", "host_internal.hpp", "device_utils.hpp", "seq_core.hpp", "partition_core.hpp",
           os.path.join("..", "..", "include", "btlbf.h")]
# (object name, source, extra flags): pass A of the partitioned pipeline is one unit per hash count
UNITS = [("capi", "capi.cpp", []), ("fastx", "fastx.cpp", []), ("seq_kernels", "seq_kernels.hip", []), ("aux_kernels", "aux_kernels.hip", []),
         ("partition_kernels", "partition_kernels.hip", [])]
UNITS += [("part_hash_h%d" % h, "part_hash_inst.hip", ["-DBTLBF_PART_H=%d" % h]) for h in range(1, 9)]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _newest_header():
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS + ["../build.py"])


def _extra_flags():
    return os.environ.get("BTLBF_CXXFLAGS", "").split()


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return _newest_header() > t or any(os.path.getmtime(os.path.join(CSRC, src)) > t for _, src, _ in UNITS)


# BTLBF_BUILD_UNITS=part_hash_h4,... (diagnostic variants only): compile just these units with the variant's flags and
# link the product build's objects for the rest -- a kernel experiment then costs one translation unit, not thirteen
ONLY = [u for u in os.environ.get("BTLBF_BUILD_UNITS", "").split(",") if u] if TAG else []
MAIN_OBJ = os.path.join(HERE, "_build")


def _compile(unit, hipcc, force, verbose):
    name, src, extra = unit
    if ONLY and name not in ONLY:
        return os.path.join(MAIN_OBJ, name + ".o")
    obj = os.path.join(OBJ, name + ".o")
    srcp = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(srcp), _newest_header()):
        return obj
    cmd = [hipcc] + FLAGS + _extra_flags() + extra + ["-c", "-o", obj, srcp]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s%s" % (src, r.stdout, r.stderr))
    return obj


def build(force=False, verbose=False, jobs=None):
    if not force and not ONLY and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    jobs = jobs or min(8, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda u: _compile(u, hipcc, force, verbose), UNITS))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
