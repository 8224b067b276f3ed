"""configs[4]: PPO rollout at the trainer's batch (train_ppo.py:322-437): B = 80 trajectories of one prompt / noise, n solver steps
(CFG 3 -> effective UNet batch 160), decode of the 80 predictions and the 80 teacher latents (chunks of 8), image-PSNR reward,
advantages, and ppo_epochs policy updates.  Synthetic SD1.5-shaped weights.  Prints one JSON line."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import consolver_amd
from consolver_amd import ppo
from consolver_amd.unet import HipUNet2DConditionModel
from consolver_amd.vae import HipAutoencoderKL
from consolver_amd.synth import synthetic_unet_state_dict, synthetic_vae_state_dict, synthetic_prompt_embeds

dev = torch.device("cuda:0")
B = int(os.environ.get("ROLLOUT_B", "80")); n = int(os.environ.get("ROLLOUT_STEPS", "8")); epochs = 4
unet = HipUNet2DConditionModel(device=dev); unet.load_state_dict(synthetic_unet_state_dict(unet.manifest()))
vae = HipAutoencoderKL(device=dev); vae.load_state_dict(synthetic_vae_state_dict(vae.manifest()))
sch = consolver_amd.PPOScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="trailing", order_dim=4,
                                 scaler_dim=0, factor_net_kwargs=dict(embedding_dim=32, hidden_dim=256, num_actions=11))
g = torch.Generator().manual_seed(0)
with torch.no_grad():
    for p in sch.factor_net.parameters():
        p.copy_(torch.randn(p.shape, generator=g) * 0.5)
sch.factor_net.to(dev)
tr = ppo.PolicyTrainer(sch.factor_net, lr=1e-4)
pe = synthetic_prompt_embeds(1, seed=1001).half().to(dev).repeat(B, 1, 1); ne = synthetic_prompt_embeds(1, seed=1002).half().to(dev).repeat(B, 1, 1)
batch = (["p"] * B, torch.randn(1, 4, 64, 64, generator=g).half().to(dev).repeat(B, 1, 1, 1), (torch.randn(1, 4, 64, 64, generator=g) * 0.18).half().to(dev).repeat(B, 1, 1, 1))
def it():
    return ppo.train_iteration(tr, None, sch, unet, vae, batch, None, cfg=3.0, num_inference_steps=n, ppo_epochs=epochs, prompt_embeds=pe, negative_prompt_embeds=ne)
it(); torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
K = 3
ev[0].record()
for _ in range(K): out = it()
ev[1].record(); torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / K
fl = unet.flops(2 * B) * n + vae.flops(8) * (2 * B / 8)
print(json.dumps({"workload": f"configs[4]: PPO rollout B={B}, {n} steps, CFG 3, 2x{B} VAE decodes, image_psnr reward, {epochs} PPO epochs (1 x MI355X)",
                  "ms_per_iteration": ms, "trajectories_per_s": B / (ms * 1e-3), "tflops": fl / (ms * 1e-3) / 1e12,
                  "frac_of_fp16_mfma_peak": fl / (ms * 1e-3) / 1e12 / 2500, "reward_mean": float(out["reward"]), "loss": float(out["loss"])}))
