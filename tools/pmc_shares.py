"""Turn the per-kernel raw SQ counter means of tools/pmc_kernels.sh (gpurun_out/pmc_kernels.txt) into shares of the wave cycles:
issuing / issue-stall / waitcnt-barrier / LDS-stall, matrix-pipe busy (two waves per SIMD) and LDS bank-conflict share."""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
i = 0
while i < len(lines):
    l = lines[i]
    if l.startswith("=="):
        print(l); i += 1; continue
    if i + 1 < len(lines) and "SQ_WAVE_CYCLES" in lines[i + 1]:
        c = {k: float(v) for k, v in re.findall(r"(\w+)=([0-9.e+\-]+)", lines[i + 1])}
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        name = re.sub(r"^void \(anonymous namespace\)::", "", l.strip())
        sh = lambda k: 100.0 * c.get(k, 0) / wc
        mf = 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (wc * 4 / 2)
        cf = 100.0 * c.get("SQ_LDS_BANK_CONFLICT", 0) / (c.get("SQ_LDS_IDX_ACTIVE", 0) or 1)
        print(f"{name[:72]:72s} issuing {sh('SQ_ACTIVE_INST_ANY'):5.1f}%  issue-stall {sh('SQ_WAIT_INST_ANY'):5.1f}%  waitcnt/barrier {sh('SQ_WAIT_ANY'):5.1f}%  "
              f"LDS-stall {sh('SQ_WAIT_INST_LDS'):5.1f}%  MFMA-pipe busy ~{mf:3.0f}%  LDS conflict cycles / LDS active {cf:4.1f}%")
        i += 2; continue
    i += 1
