"""Builds libphonlp_heads.so (hand-written HIP, gfx950 only) in-tree with hipcc.

    python -m phonlp_amd.build            # incremental: recompiles the sources that are newer than their objects
    python -m phonlp_amd.build --force

Each .hip file is compiled to an object of its own (in parallel), then linked; a changed header recompiles all.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "build")
LIB = os.path.join(LIB_DIR, "libphonlp_heads.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-variable", "-Wno-inline-asm"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "phonlp_heads.h")]


def obj_of(src):
    return os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")


def stale():
    if not os.path.exists(LIB):
        return True
    deps = sources() + headers()
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


def build_ablation(verbose=True):
    """Timing-only variant (tools/fused_probe.py, tools/x3_abl.sh): the GEMM kernels with their ablation branches
    (-DPHONLP_ABLATION; PHONLP_X3_ABLATION / PHONLP_X6_ABLATION select them at run time).  The shipped library has none."""
    out = os.path.join(HERE, "..", "tools", "libphonlp_heads_ablation.so")
    cmd = [HIPCC] + FLAGS + ["-DPHONLP_ABLATION", "-shared", "-o", out] + sources()
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build_stamps(verbose=True):
    """Diagnostic variant (tools/mst_stamps.py): the MST kernels with per-phase cycle counters (-DMST_STAMPS)."""
    out = os.path.join(HERE, "..", "tools", "libphonlp_heads_stamps.so")
    cmd = [HIPCC] + FLAGS + ["-DMST_STAMPS", "-shared", "-o", out] + sources()
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build(force=False, verbose=True):
    if not force and not stale():
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_time = max(os.path.getmtime(h) for h in headers())
    todo = [s for s in sources()
            if force or not os.path.exists(obj_of(s)) or os.path.getmtime(obj_of(s)) < max(os.path.getmtime(s), hdr_time)]

    def compile_one(src):
        cmd = [HIPCC] + FLAGS + ["-c", "-o", obj_of(src), src]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    jobs = max(1, min(len(todo), int(os.environ.get("MAX_JOBS", "8"))))
    with ThreadPoolExecutor(jobs) as ex:
        list(ex.map(compile_one, todo))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [obj_of(s) for s in sources()]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        build_stamps()
    elif "--ablation" in sys.argv:
        build_ablation()
    else:
        build(force="--force" in sys.argv)
