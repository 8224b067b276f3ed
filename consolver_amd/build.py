"""Builds libconsolver_hip.so in-tree with hipcc for gfx950 (no JIT cache, no cmake).

``python -m consolver_amd.build`` or ``consolver_amd.build.build()``.  hipcc
cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(PKG, "libconsolver_hip.so")
ARCH = "gfx950"

# source -> extra flags
SOURCES = {
    "api.cpp": [],
    "solver.hip": ["-ffp-contract=off"],   # torch-like separate mul/add roundings
    "igemm.hip": [],
    "attention.hip": [],
    "xattn.hip": [],
    "norm.hip": [],
    "misc.hip": [],
    "gemm2.hip": [],
    "flux_ops.hip": [],
    "flux.cpp": [],
    "unet.cpp": [],
    "vae.cpp": [],
    "clip.cpp": [],
    "ppo.hip": [],
    "ops_api.cpp": [],
}
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
          "-I" + os.path.join(os.path.dirname(PKG), "include")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(src_paths, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_paths)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    inc = os.path.join(os.path.dirname(PKG), "include")
    headers += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]   # every ABI header
    jobs = []
    objs = []
    for src, extra in SOURCES.items():
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            raise FileNotFoundError(sp)
        op = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(op)
        if force or _newer([sp] + headers, op):
            cmd = [hipcc] + COMMON + extra + ["-x", "hip", "-c", sp, "-o", op]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)
    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _newer(objs, LIB):
        run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
