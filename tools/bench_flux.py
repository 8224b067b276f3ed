"""configs[3]: FLUX-Kontext DiT + FMPPOScheduler 8-step bf16 edit at 1024x1024 on one MI355X (synthetic weights).

Weights (11.9 B parameters, 23.8 GB bf16) are generated on the GPU tensor by tensor; the forward is timed with
HIP events; algorithmic FLOPs come from cs_flux_flops (2*MAC of every GEMM + 4*B*S^2*D per attention)."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import consolver_amd
from consolver_amd.flux import HipFluxTransformer2DModel, FluxKontextSamplingEngine, pack_latents

dev = torch.device("cuda:0")
layers = int(os.environ.get("FLUX_LAYERS", "19")); singles = int(os.environ.get("FLUX_SINGLES", "38"))
m = HipFluxTransformer2DModel(dict(num_layers=layers, num_single_layers=singles), device=dev)
g = torch.Generator(device=dev).manual_seed(20251226)
t0 = time.time()
nparams = 0
for name, shape in m.manifest():
    if name.endswith("norm_q.weight") or name.endswith("norm_k.weight") or name.endswith("norm_added_q.weight") or name.endswith("norm_added_k.weight"):
        w = 1.0 + 0.1 * torch.randn(shape, generator=g, device=dev)
    elif name.endswith(".weight"):
        w = torch.randn(shape, generator=g, device=dev) * (1.0 / shape[1]) ** 0.5
        if ".norm" in name and name.endswith("linear.weight"): w = w * 0.5
    else:
        w = 0.05 * torch.randn(shape, generator=g, device=dev)
        if ".norm" in name and name.endswith("linear.bias"): w = w + 0.3
    nparams += w.numel()
    m.set_weight(name, w)
    del w
m.finalize()
torch.cuda.synchronize()
print(f"weights: {nparams/1e9:.2f} B params in {time.time()-t0:.1f} s", file=sys.stderr)

B, T, Lq = 1, 512, 4096
sch = consolver_amd.FMPPOScheduler.from_pretrained("black-forest-labs/FLUX.1-Kontext-dev", subfolder="scheduler", order_dim=2, scaler_dim=0, mu_dim=0,
                                                   factor_net_kwargs=dict(embedding_dim=32, hidden_dim=256, num_actions=11))
sch.factor_net.to(dev)
gc = torch.Generator().manual_seed(43)
lat = pack_latents(torch.randn(B, 16, 128, 128, generator=gc)).to(torch.bfloat16).to(dev)
img = pack_latents(torch.randn(B, 16, 128, 128, generator=gc)).to(torch.bfloat16).to(dev)
enc = torch.nn.functional.layer_norm(torch.randn(B, T, 4096, generator=gc), (4096,)).to(torch.bfloat16).to(dev)
pooled = torch.randn(B, 768, generator=gc).to(torch.bfloat16).to(dev)
eng = FluxKontextSamplingEngine(m, sch, guidance_scale=2.5)
n = 8
out = eng.generate(lat, img, enc, pooled, latent_hw=(64, 64), num_inference_steps=n)     # warm-up (also sizes the workspace)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); out = eng.generate(lat, img, enc, pooled, latent_hw=(64, 64), num_inference_steps=n); e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
fl = m.flops(B, T, 2 * Lq)
print(json.dumps({"workload": "configs[3]: FLUX-Kontext + FMPPOScheduler 8-step bf16, 1024x1024 edit, 1 MI355X", "layers": [layers, singles],
                  "params_B": round(nparams / 1e9, 2), "ms_per_edit": round(ms, 1), "edits_per_s": round(1e3 / ms, 4),
                  "tflop_per_forward": round(fl / 1e12, 2), "tflops": round(fl * n / (ms * 1e-3) / 1e12, 1), "frac_of_2500": round(fl * n / (ms * 1e-3) / 2.5e15, 3),
                  "finite": bool(torch.isfinite(out.float()).all())}))
