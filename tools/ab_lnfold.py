"""same-box A/B: UNet forward (effective batch 32) with the LayerNorms folded into their consumer GEMMs (ln_fold = 1) against LayerNorm kernels + plain GEMMs,
in both residual-stream modes; per-class profile of each."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consolver_amd import ops
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
        for fold in (0, 1):
            u.set_residual_precision(mode); ops.set_tuning("ln_fold", fold)
            for _ in range(2):
                u(lat, t, encoder_hidden_states=ctx, dup=2, reuse_kv=False)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(10):
                u(lat, t, encoder_hidden_states=ctx, dup=2, reuse_kv=(i > 0))
            b.record(); torch.cuda.synchronize()
            out.setdefault(f"{mode} ln_fold={fold}", []).append(round(a.elapsed_time(b) / 10, 3))
for mode in ("f16", "f16x2"):
    for fold in (0, 1):
        u.set_residual_precision(mode); ops.set_tuning("ln_fold", fold)
        u(lat, t, encoder_hidden_states=ctx, dup=2, reuse_kv=False)
        u.set_profiling(True)
        u(lat, t, encoder_hidden_states=ctx, dup=2, reuse_kv=False)
        out[f"{mode} ln_fold={fold} classes"] = {k: round(v["ms"], 3) for k, v in u.profile().items()}
        u.set_profiling(False)
ops.set_tuning("ln_fold", 1)
print(json.dumps(out, indent=1))
