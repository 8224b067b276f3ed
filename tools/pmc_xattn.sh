# SQ counter passes over tools/ab_xattn_tile.py: xattn64_kernel (two workgroups per CU) against xattn_block_kernel -> gpurun_out/pmc_xattn.txt
# (counters only, separate passes, the program directly after `--`)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_x/p$i -- python3 $R/tools/ab_xattn_tile.py > $R/gpurun_out/pmc_x_p$i.log 2>&1
done
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_x | grep -A1 -E "xattn64_kernel|xattn_block_kernel" > gpurun_out/pmc_xattn_raw.txt
python3 - <<'PY' > gpurun_out/pmc_xattn.txt
import re
rows = open("gpurun_out/pmc_xattn_raw.txt").read().split("\n")
print("SQ counter means per dispatch (tools/pmc_xattn.sh); shares = counter / SQ_WAVE_CYCLES (MFMA busy: / SQ_BUSY_CYCLES x 4 SIMDs as in tools/pmc_shares.py is not applied here: raw ratios)")
for i in range(0, len(rows) - 1):
    if "kernel" in rows[i] and "=" in rows[i + 1]:
        name = rows[i].strip()
        c = dict(kv.split("=") for kv in rows[i + 1].split())
        c = {k: float(v) for k, v in c.items()}
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        sh = {k: c[k] / wc for k in c if k not in ("SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_INSTS_MFMA")}
        print(name); print("   " + "  ".join(f"{k}={v:.3g}" for k, v in sorted(c.items()))); print("   shares of wave cycles: " + "  ".join(f"{k}={v:.3f}" for k, v in sorted(sh.items())))
PY
rm -rf gpurun_out/pmc_x gpurun_out/pmc_x_p*.log
cat gpurun_out/pmc_xattn.txt | head -40
