"""Finds compiler-serialized memory round trips in the device assembly of a .hip source (round 4): on gfx9 `s_waitcnt vmcnt(0)` waits for EVERY older load, store
and LDS-DMA of the wave, and hipcc waits for a load where the source uses it -- so `load; wait; use` inside an unrolled epilogue or a small loop is one full round trip
per occurrence, each behind the acknowledgement of the stores before it.

  python tools/scan_waits.py consolver_amd/csrc/igemm.hip        (CPU container: hipcc cross-compiles to assembly; ~1 min per file)

Report 1: per kernel, the loads that are followed by `s_waitcnt vmcnt(0)` within four instructions (before any other memory instruction).
Report 2: small loops (< 400 instructions, no MFMA needed) that contain both a load and a `vmcnt(0)`: `for (...) acc += p[i]` shapes in front of a barrier."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(n):
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    return re.sub(r"\(anonymous namespace\)::", "", d)[:100]


def main(src):
    out = os.path.join(tempfile.mkdtemp(prefix="scan_"), "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-x", "hip", src, "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    cur, stats = None, {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\S+):\s", l)
        if m:
            cur = m.group(1); stats[cur] = [0, 0, 0]; continue
        if cur is None:
            continue
        t = l.strip()
        if t.startswith(("global_load", "buffer_load")) and not t.endswith("lds"):
            stats[cur][0] += 1
            for j in range(i + 1, min(i + 5, len(lines))):
                tt = lines[j].strip()
                if tt.startswith("s_waitcnt") and "vmcnt(0)" in tt:
                    stats[cur][1] += 1; break
                if tt.startswith(("global_load", "buffer_load", "global_store")):
                    break
        if t.startswith(("global_store", "buffer_store")):
            stats[cur][2] += 1
    print("== loads waited for with vmcnt(0) right where they are issued")
    for k, (nl, nw, ns) in stats.items():
        if nw >= 4:
            print(f"loads {nl:4d}  load->vmcnt(0) {nw:4d}  stores {ns:4d}  {demangle(k)}")
    print("== small loops with a load and a vmcnt(0) inside")
    cur = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\S+):\s", l)
        if m:
            cur = m.group(1)
        m = re.match(r"^(\.LBB\d+_\d+):\s*;.*Loop Header", l)
        if not (m and cur):
            continue
        lab, nl, nw, end = m.group(1), 0, 0, None
        for j in range(i + 1, min(i + 400, len(lines))):
            t = lines[j].strip()
            if t.startswith(("global_load", "buffer_load")) and not t.endswith("lds"):
                nl += 1
            if t.startswith("s_waitcnt") and "vmcnt(0)" in t:
                nw += 1
            if t.startswith("s_cbranch") and t.endswith(lab):
                end = j; break
            if re.match(r"^_Z\S+:\s", lines[j]):
                break
        if end and nl and nw:
            print(f"loop {lab:12s} {end - i:4d} instructions, loads {nl}, vmcnt(0) {nw}  {demangle(cur)}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "consolver_amd", "csrc", "igemm.hip"))
