"""The tightest gated number -- step 0 of the 4-step trajectory on the full UNet (t = 999 -> 749: the per-forward eps error reaches the latents 1.08x) -- under
the round-6 precision-for-speed knobs, one oracle trajectory, same box:   python tools/parity_knobs_n4.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consolver_amd import ops
from tests import test_parity_e2e_gpu as T

n, wseed, g, B = 4, 7, 3.0, 1
c = T._oracle_case(n, wseed, g, B)
ux2, _ = T.build_full(seed=wseed, residual="f16x2")
for knobs in ({}, {"lo8": 0}, {"x2_sc_skip": 0}, {"lo8": 0, "x2_sc_skip": 0}, {"x2_sc_skip": 0x58}, {"x2_sc_skip": 0x18}, {"epi_fast": 1}, {"conv_out_mfma": 0}, {"x2_split_a": 3}):
    ops.reset_tuning()
    for k, v in knobs.items():
        ops.set_tuning(k, v)
    traj = T._hip_trajectory(ux2, c["sch"], c["idx"], c["noise"], c["ctx"].to(T.DEV), B, n, g)
    d = [T.rel_l2(traj[i], c["traj"][i]) for i in range(n)]
    print(f"{str(knobs):40s} per step " + " ".join(f"{v:.4e}" for v in d), flush=True)
ops.reset_tuning()
