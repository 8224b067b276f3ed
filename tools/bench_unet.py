"""UNet forward A/B tool: batch 32 (16 images x CFG) SD1.5 forward, mean of N runs + per-class profile.
CS_RESIDUAL=f16|f16x2 selects the residual-stream mode (default f16x2); CS_PROFILE_JSON=path dumps the per-class profile (ms, flops, algorithmic bytes).
CS_TUNE="key=value,..." sets library tuning knobs; CONSOLVER_HIP_LIB selects an alternative build; CS_OUT_F32=1 asks for the fp32 eps output (the engine's call)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consolver_amd import ops
from consolver_amd.unet import HipUNet2DConditionModel
from consolver_amd.synth import synthetic_unet_state_dict, synthetic_prompt_embeds
dev = "cuda:0"
for kv in os.environ.get("CS_TUNE", "").split(","):
    if "=" in kv:
        k, v = kv.split("="); ops.set_tuning(k, int(v))
u = HipUNet2DConditionModel(device=dev, residual=os.environ.get("CS_RESIDUAL", "f16x2")); u.load_state_dict(synthetic_unet_state_dict(u.manifest()))      # (host-side synthesis on purpose: the PMC traffic passes sum the counters over EVERY dispatch of this process)
NL = int(os.environ.get("CS_NLAT", "16"))
lat = torch.randn(NL, 4, 64, 64, device=dev).half()
ctx = synthetic_prompt_embeds(2 * NL).half().to(dev)
t = torch.tensor([499.0], device=dev)
OD = torch.float32 if os.environ.get("CS_OUT_F32") == "1" else None
for _ in range(3): u(lat, t, encoder_hidden_states=ctx, dup=2, out_dtype=OD)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n): u(lat, t, encoder_hidden_states=ctx, dup=2, reuse_kv=True, out_dtype=OD)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / n
print(f"forward {ms:.3f} ms  {u.flops(2 * NL) / ms / 1e9:.1f} TFLOP/s  ({u.flops(2 * NL) / ms / 1e9 / 2500 * 100:.1f} % of fp16 MFMA peak)")
u.set_profiling(True); u(lat, t, encoder_hidden_states=ctx, dup=2, reuse_kv=True, out_dtype=OD); pr = u.profile(); u.set_profiling(False)
print("  ".join(f"{k}={v['ms']:.2f}" for k, v in pr.items()))
if os.environ.get("CS_PROFILE_JSON"):
    import json
    json.dump(pr, open(os.environ["CS_PROFILE_JSON"], "w"), indent=1)
