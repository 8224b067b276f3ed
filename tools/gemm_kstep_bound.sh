# what bounds a k step of gemm_big_kernel?  k-loop-only timings (debug bit 1 = no epilogue) with the staging removed or redirected.
#   1        k loop as shipped (no epilogue)
#   32769    + no LDS-DMA inside the loop (MFMAs + fragment reads + barriers only)
#   65537    + every activation piece from the 256-byte zero page (L1/L2 hits), weights as shipped
#   131073   + the same k step staged every time (activations and weights L2-hot, DMA count unchanged)
#   196609   + both
R=$GRAFT_REPO_ROOT
for v in 1 32769 65537 131073 196609; do echo "== gemm_ring=0 debug=$v"; CS_TUNE="gemm_ring=0,debug=$v" python3 $R/tools/bench_ops.py gemm 2>&1 | grep -E "^linear"; done
