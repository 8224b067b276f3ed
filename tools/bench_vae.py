import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from consolver_amd.vae import HipAutoencoderKL, decode_latents
from consolver_amd.synth import synthetic_vae_state_dict
v = HipAutoencoderKL({}, device="cuda:0"); v.load_state_dict(synthetic_vae_state_dict(v.manifest()))
from consolver_amd import ops
for kv in os.environ.get("CS_TUNE", "").split(","):          # CS_TUNE="up_fold=0": the fused-upsample kernels instead of the sub-pixel upsamplers
    if "=" in kv:
        ops.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
for B in (1, 4, 16):
    lat = torch.randn(B, 4, 64, 64, device="cuda:0", dtype=torch.float16) * 0.18
    for _ in range(2): decode_latents(v, lat, B)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(5): decode_latents(v, lat, B)
    torch.cuda.synchronize(); dt = (time.time() - t) / 5
    print(f"B={B} {dt*1e3:.2f} ms  {dt*1e3/B:.2f} ms/img  {v.flops(B)/dt/1e12:.1f} TFLOP/s ws={v._ws.numel()/2**30:.2f} GiB")
