"""AddressSanitizer + UndefinedBehaviorSanitizer run of the library's HOST side.

The executors' host code -- weight repacking, the first-fit arena over the caller's workspace, the dry-run workspace sizing over all execution
variants, the launch sequences -- is compiled from the real sources with ``-fsanitize=address,undefined`` (hipcc ``--cuda-host-only``: no device code,
no GPU) and linked with tests/sanitize/stubs.cpp (host malloc stands in for device memory; every launch stub touches the first and last byte of each
tensor the kernel would access).  tests/sanitize/harness.cpp then drives create -> set_weight -> finalize -> workspace_bytes -> forward through the
arena with a workspace of exactly the requested size, in both residual-stream modes and every knob variant, plus the error paths.
GPU-side sanitizers are not available on this pool (xnack), so this is the sanitizer coverage the tree has.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "consolver_amd", "csrc")
HOST_SOURCES = ["api.cpp", "unet.cpp", "vae.cpp", "flux.cpp", "clip.cpp", "ops_api.cpp"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    return None


@pytest.mark.timeout(1500)
def test_host_executors_under_asan_ubsan(tmp_path):
    hipcc = _hipcc()
    if hipcc is None:
        pytest.skip("hipcc not found")
    flags = ["-x", "hip", "--cuda-host-only", "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined",
             "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-w", "-I" + os.path.join(ROOT, "include")]
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES] + [os.path.join(ROOT, "tests", "sanitize", f) for f in ("stubs.cpp", "harness.cpp")]
    objs = []
    procs = []
    for s in srcs:
        o = str(tmp_path / (os.path.basename(s) + ".o"))
        objs.append(o)
        procs.append((s, subprocess.Popen([hipcc] + flags + ["-c", s, "-o", o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        assert p.returncode == 0, f"{s}:\n{out}"
    exe = str(tmp_path / "harness")
    r = subprocess.run([hipcc, "-fsanitize=address,undefined", "-o", exe] + objs, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=1200)
    text = r.stdout + r.stderr
    assert r.returncode == 0, text[-6000:]
    assert "sanitize harness: ok" in r.stdout
    assert "AddressSanitizer" not in text and "runtime error" not in text and "LeakSanitizer" not in text, text[-6000:]
