"""same-box A/B of the two residual-stream modes: UNet forward ms at effective batch 32 (CFG dual of 16) and per-class profile."""
import json
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consolver_amd.unet import HipUNet2DConditionModel
from consolver_amd.synth import synthetic_unet_state_dict, synthetic_prompt_embeds

dev = "cuda:0"
u = HipUNet2DConditionModel(device=dev)
u.load_state_dict(synthetic_unet_state_dict(u.manifest(), seed=20251226))
B = 16
lat = torch.randn(B, 4, 64, 64, generator=torch.Generator().manual_seed(43)).half().to(dev)
ctx = torch.cat([synthetic_prompt_embeds(B, seed=1002), synthetic_prompt_embeds(B, seed=1001)]).half().to(dev)
t = torch.tensor([499.0], device=dev)
out = {}
for rnd in range(3):
    for mode in ("f16", "f16x2"):
        u.set_residual_precision(mode)
        for _ in range(2):
            u(lat, t, encoder_hidden_states=ctx, dup=2, reuse_kv=False)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(10):
            u(lat, t, encoder_hidden_states=ctx, dup=2, reuse_kv=(i > 0))
        b.record(); torch.cuda.synchronize()
        out.setdefault(mode, []).append(a.elapsed_time(b) / 10)
for mode in ("f16", "f16x2"):
    u.set_residual_precision(mode)
    u(lat, t, encoder_hidden_states=ctx, dup=2, reuse_kv=False)
    u.set_profiling(True)
    u(lat, t, encoder_hidden_states=ctx, dup=2, reuse_kv=False)
    out[mode + "_classes"] = {k: round(v["ms"], 3) for k, v in u.profile().items()}
    u.set_profiling(False)
    out[mode + "_ws_GB"] = u._ws.numel() / 1e9
print(json.dumps(out, indent=1))
