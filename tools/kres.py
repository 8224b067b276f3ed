"""Compile one csrc/*.hip (or .cpp) for gfx950 to assembly HERE (no GPU) and print per kernel matching a pattern: VGPRs, AGPR offset, scratch bytes, LDS bytes,
and counts of v_mfma / s_waitcnt vmcnt(0) / lgkmcnt(0) / scratch instructions.   python tools/kres.py misc.hip conv_out_mfma"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
extra = ["-ffp-contract=off"] if src.startswith("solver") else []
out = "/tmp/kres_" + os.path.splitext(src)[0]
os.makedirs(out, exist_ok=True)
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(root, "include"), "-x", "hip", "-c",
       os.path.join(root, "consolver_amd", "csrc", src), "-o", out + "/o.o", "-save-temps=obj", "-w"] + extra
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-4000:]); sys.exit(1)
asm = [f for f in os.listdir(out) if f.endswith("gfx950.s")][0]
s = open(os.path.join(out, asm)).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
    name, body = m.group(1), m.group(2)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if pat and pat not in dem:
        continue
    g = lambda k: (re.findall(r"\.amdhsa_%s (\d+)" % k, body) or ["?"])[0]
    fm = re.search(r"^%s:.*?s_endpgm" % re.escape(name), s, re.S | re.M)
    b = fm.group(0) if fm else ""
    cnt = lambda rx: len(re.findall(rx, b))
    n_vm0, n_lg0 = cnt(r"vmcnt\(0\)"), cnt(r"lgkmcnt\(0\)")
    print(f"{dem[:110]}\n    vgpr {g('next_free_vgpr')} accum_offset {g('accum_offset')} scratch {g('private_segment_fixed_size')} B lds {g('group_segment_fixed_size')} B | "
          f"mfma {cnt('v_mfma')} vmcnt(0) {n_vm0} lgkmcnt(0) {n_lg0} scratch-ops {cnt('scratch_')} lines {b.count(chr(10))}")
