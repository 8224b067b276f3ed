"""CS_RESIDUAL_F16X2: per-forward eps error vs the fp32 CPU oracle and forward time for each x2_split_a setting (which stream-consuming GEMMs read hi + lo)."""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consolver_amd import ops
from consolver_amd.unet import HipUNet2DConditionModel
from consolver_amd.synth import synthetic_unet_state_dict, synthetic_prompt_embeds
from oracle.unet_oracle import UNetOracle

dev = "cuda:0"
u = HipUNet2DConditionModel(device=dev)
sd = synthetic_unet_state_dict(u.manifest(), seed=7)
u.load_state_dict(sd)
torch.set_num_threads(os.cpu_count())
orc = UNetOracle(sd, u.config)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


out = {}
g = torch.Generator().manual_seed(5)
cases = []
for t in (999, 499):
    lat = torch.randn(1, 4, 64, 64, generator=g).half()
    ctx = synthetic_prompt_embeds(2, seed=13 + t).half()
    cases.append((t, lat, ctx, orc(torch.cat([lat.float()] * 2), t, ctx.float())))
B = 16
latb = torch.randn(B, 4, 64, 64, generator=g).half().to(dev)
ctxb = torch.cat([synthetic_prompt_embeds(B, seed=1002), synthetic_prompt_embeds(B, seed=1001)]).half().to(dev)
tt = torch.tensor([499.0], device=dev)
for mode, sa in (("f16", 0), ("f16x2", 0), ("f16x2", 1), ("f16x2", 2), ("f16x2", 3)):
    u.set_residual_precision(mode)
    ops.set_tuning("x2_split_a", sa)
    row = {}
    for t, lat, ctx, want in cases:
        row[f"err_t{t}"] = rel(u(lat.to(dev), t, encoder_hidden_states=ctx.to(dev), dup=2, reuse_kv=False)[0].float().cpu(), want)
    for _ in range(2):
        u(latb, tt, encoder_hidden_states=ctxb, dup=2, reuse_kv=False)
    torch.cuda.synchronize()
    ms = []
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(10):
            u(latb, tt, encoder_hidden_states=ctxb, dup=2, reuse_kv=(i > 0))
        b.record(); torch.cuda.synchronize()
        ms.append(a.elapsed_time(b) / 10)
    row["fwd_ms_b32"] = min(ms)
    out[f"{mode} split_a={sa}"] = row
    print(mode, sa, row, flush=True)
ops.set_tuning("x2_split_a", 1)
print(json.dumps(out, indent=1))
