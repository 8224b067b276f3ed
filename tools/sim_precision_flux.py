"""WHERE the full-depth FLUX DiT's bf16 error comes from, by emulation (the FLUX twin of tools/sim_precision.py): the fp32 restatement (oracle/flux_oracle.py) run on the GPU
with explicit bf16 rounding points -- the residual STREAM only / the BRANCH tensors only / both -- against the unrounded fp32 run, next to the HIP DiT itself.  Sizes a
split-bf16 (hi + lo) hidden-state stream through the 57 gated-residual epilogues of gemm2.hip BEFORE a kernel is written: what it would buy is (both) -> (branch only).
Full depth (19 + 38 blocks, 11.9 B synthetic parameters), 64 text + 256 latent + 256 image tokens.  Runs on the GPU box (weights stay on the GPU, one fp32 tensor live at a time)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consolver_amd.flux import HipFluxTransformer2DModel, prepare_latent_image_ids
from oracle.flux_oracle import FluxOracle
DEV = "cuda:0"
depth = dict(dtype=torch.bfloat16)
if len(sys.argv) > 2:
    depth.update(num_layers=int(sys.argv[1]), num_single_layers=int(sys.argv[2]))
m = HipFluxTransformer2DModel(depth, device=DEV)
g = torch.Generator(device=DEV).manual_seed(11)
sd = {}
for name, shape in m.manifest():
    if name.endswith(("norm_q.weight", "norm_k.weight", "norm_added_q.weight", "norm_added_k.weight")):
        w = 1.0 + 0.1 * torch.randn(shape, generator=g, device=DEV)
    elif name.endswith(".weight"):
        w = torch.randn(shape, generator=g, device=DEV) * (1.0 / shape[1]) ** 0.5
        if ".norm" in name and name.endswith("linear.weight"):
            w = w * 0.5
    else:
        w = 0.05 * torch.randn(shape, generator=g, device=DEV)
        if ".norm" in name and name.endswith("linear.bias"):
            w = w + 0.3
    sd[name] = w.to(torch.bfloat16)
    m.set_weight(name, sd[name])
m.finalize()
gc = torch.Generator().manual_seed(4)
B, T, Lq = 1, 64, 256
lat = torch.randn(B, Lq, 64, generator=gc).to(torch.bfloat16); img = torch.randn(B, Lq, 64, generator=gc).to(torch.bfloat16)
enc = torch.nn.functional.layer_norm(torch.randn(B, T, 4096, generator=gc), (4096,)).to(torch.bfloat16)
pooled = torch.randn(B, 768, generator=gc).to(torch.bfloat16)
t = torch.tensor([0.9567]); guidance = torch.full((B,), 2.5)
ids = np.concatenate([prepare_latent_image_ids(16, 16), prepare_latent_image_ids(16, 16, first=1.0)], 0)
txt_ids = np.zeros((T, 3), np.float32)
rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())
rb16 = lambda x: x.to(torch.bfloat16).to(x.dtype)
def run(rs, rb, rp=False, rh=False, rs17=False):
    o = FluxOracle(sd, m.config, lazy=True, device=DEV, dtype=torch.float32)
    if rs: o.rs = rb16
    if rb: o.rb = rb16
    if rp: o.rp = rb16
    if rh: o.rh = rb16
    if rs17: o.rs = lambda x: rb16(x) + rb16(x - rb16(x))          # the split stream: hi + lo planes of bf16
    return o(torch.cat([lat, img], 1), t, guidance, pooled, enc, txt_ids, ids)[:, :Lq]
want = run(False, False)
cpu = FluxOracle(sd, m.config, lazy=True)(torch.cat([lat, img], 1).float(), t, guidance, pooled.float(), enc.float(), txt_ids, ids)[:, :Lq]
print(f"fp32 graph on the GPU vs the fp32 CPU oracle (sanity of the emulator): {rel(want, cpu):.3e}")
for name, rs, rb in (("residual stream in bf16, branches exact", True, False), ("branch tensors in bf16, stream exact", False, True), ("both (the bf16 storage class)", True, True)):
    print(f"{name:48s}: {rel(run(rs, rb), want):.3e}")
# round 6: what separates the HIP split stream from "branch tensors in bf16, stream exact"
for name, kw in (("branch + the stream as hi + lo bf16 planes", dict(rs17=True)), ("branch + softmax probabilities in bf16", dict(rp=True)),
                 ("branch + the output head's two tensors in bf16", dict(rh=True)), ("branch + all three (what the HIP DiT stores)", dict(rs17=True, rp=True, rh=True))):
    print(f"{name:48s}: {rel(run(False, True, **kw), want):.3e}", flush=True)
got = m(lat.to(DEV), t.to(DEV), guidance=guidance.to(DEV), pooled_projections=pooled.to(DEV), encoder_hidden_states=enc.to(DEV), txt_ids=txt_ids, img_ids=ids, image_latents=img.to(DEV))[0]
t16 = FluxOracle(sd, m.config, lazy=True, device=DEV, dtype=torch.bfloat16)(torch.cat([lat, img], 1), t, guidance, pooled, enc, txt_ids, ids)[:, :Lq]
print(f"{'HIP DiT (cs_flux_forward_joint)':48s}: {rel(got.float(), want):.3e}")
print(f"{'plain torch bf16 graph (the reference class)':48s}: {rel(t16.float(), want):.3e}")
