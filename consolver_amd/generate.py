"""Sharded batch generation driver: prompts -> 8-step ConsistencySolver latents -> pixels -> ``{rank}_{idx:08d}.png/.txt``.

Mirror of ``gen_ppo.generate_batch_images`` / ``generate_imgs`` (gen_ppo.py:237-379) on the native engine: the same
partition rule (contiguous ``len // P`` blocks, the last rank takes the remainder, :349-357), batching, per-batch
generator seed ``seed + batch_idx`` (:258-260) and file naming (:318-330).  One process per GPU; no data-path collective.

    python -m torch.distributed.run --nproc-per-node 8 -m consolver_amd.generate --prompt-cache prompts.safetensors \\
        --unet unet.safetensors --vae vae.safetensors --policy checkpoint-N/model.ckpt --out generation/ --steps 8 --cfg 3

Prompt embeddings come from a prompt cache (text_encoder.save_prompt_cache) or from a caller-supplied encoder; model
weights are state dicts with the diffusers names.  With ``--synthetic`` seeded random weights / embeddings of the SD1.5
shapes stand in (no checkpoints exist offline), which is what the tests and the benchmark use.
"""
import argparse
import os
import time

import torch

from . import evaluation, launch


def prepare_latents(batch, shape, seed, device, dtype=torch.float16):
    """pipeline.prepare_latents with ``generator = torch.Generator(device).manual_seed(seed + batch_idx)`` (gen_ppo.py:258-260)."""
    gen = torch.Generator(device=device).manual_seed(seed)
    return torch.randn((batch,) + tuple(shape), generator=gen, device=device, dtype=dtype)


def generate_batch_images(prompts, prompt_embeds, negative_prompt_embeds, batch_size, engine, num_inference_steps, device, device_id,
                          seed, generation_path, use_graph=False, save=True):
    """gen_ppo.py:237-330 for one rank's prompts.  Returns the number of images written."""
    shape = (engine.unet.config["in_channels"], engine.unet.config["sample_size"], engine.unet.config["sample_size"])
    total_batches = len(prompts) // batch_size + (1 if len(prompts) % batch_size != 0 else 0)
    done = 0
    for batch_idx in range(total_batches):
        sl = slice(batch_idx * batch_size, (batch_idx + 1) * batch_size)
        batch_prompts = prompts[sl]
        pe = prompt_embeds[sl].to(device)
        ne = negative_prompt_embeds[sl].to(device) if negative_prompt_embeds is not None else None
        noise = prepare_latents(len(batch_prompts), shape, seed + batch_idx, device)
        images = engine.generate(pe, ne, latents=noise, num_inference_steps=num_inference_steps,
                                 use_graph=use_graph and len(batch_prompts) == batch_size, output_type="pt")
        if save:
            for img_idx, (img, prompt) in enumerate(zip(images, batch_prompts)):
                evaluation.save_generation(generation_path, device_id, batch_idx * batch_size + img_idx, img, prompt)
        done += len(batch_prompts)
    return done


def generate_imgs(generation_path, prompts, prompt_embeds, negative_prompt_embeds, engine, num_inference_steps, device_id, num_processes,
                  seed, batch_size=32, device=None, save=True):
    """gen_ppo.py:333-379: this rank's contiguous shard, batches of 32."""
    lo, hi = launch.shard_bounds(len(prompts), num_processes, device_id)
    device = device or torch.device("cuda", torch.cuda.current_device())
    return generate_batch_images(prompts[lo:hi], prompt_embeds[lo:hi], negative_prompt_embeds[lo:hi] if negative_prompt_embeds is not None else None,
                                 batch_size, engine, num_inference_steps, device, device_id, seed, generation_path, save=save)


def _load_sd(path):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu")


def main(argv=None):
    import consolver_amd
    from .engine import SDSamplingEngine
    from .synth import synthetic_prompt_embeds, synthetic_unet_state_dict, synthetic_vae_state_dict
    from .text_encoder import load_prompt_cache
    from .unet import HipUNet2DConditionModel
    from .vae import HipAutoencoderKL
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--out", required=True)
    ap.add_argument("--prompt-cache")
    ap.add_argument("--unet"), ap.add_argument("--vae"), ap.add_argument("--policy")
    ap.add_argument("--synthetic", type=int, default=0, help="N synthetic prompts with seeded random weights")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--cfg", type=float, default=3.0)
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--seed", type=int, default=43)
    ap.add_argument("--order-dim", type=int, default=4)
    ap.add_argument("--scaler-dim", type=int, default=0)
    ap.add_argument("--num-actions", type=int, default=11)
    ap.add_argument("--use_conv", "--use-conv", dest="use_conv", action="store_true",
                    help="gen_ppo.py:399: the policy also sees the cosine-similarity features of the eps history (factor_net_ppo.py:72-73,146-149)")
    ap.add_argument("--residual", default="f16x2", choices=["f16", "f16x2", "residual_fp32"],
                    help="residual-stream storage of the UNet executor (include/consolver_hip.h): f16x2 meets the 1e-3 latent gate, f16 is ~8 %% faster")
    args = ap.parse_args(argv)
    rank, world, local, dist = launch.init_distributed()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    unet, vae = HipUNet2DConditionModel(device=dev, residual=args.residual), HipAutoencoderKL(device=dev)
    if args.synthetic:
        unet.load_state_dict(synthetic_unet_state_dict(unet.manifest()))
        vae.load_state_dict(synthetic_vae_state_dict(vae.manifest()))
        prompts = [f"synthetic prompt {i}" for i in range(args.synthetic)]
        pe, ne = synthetic_prompt_embeds(args.synthetic, seed=1001).half(), synthetic_prompt_embeds(args.synthetic, seed=1002).half()
    else:
        unet.load_state_dict(_load_sd(args.unet))
        vae.load_state_dict(_load_sd(args.vae))
        prompts, pe, ne = load_prompt_cache(args.prompt_cache)
    sch = consolver_amd.PPOScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="trailing",
                                     order_dim=args.order_dim, scaler_dim=args.scaler_dim, use_conv=args.use_conv,
                                     factor_net_kwargs=dict(embedding_dim=32, hidden_dim=256, num_actions=args.num_actions))
    if args.policy:
        sch.factor_net.load_state_dict(torch.load(args.policy, map_location="cpu"))
    sch.factor_net.to(dev)
    eng = SDSamplingEngine(unet, sch, guidance_scale=args.cfg, vae=vae)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    n = generate_imgs(args.out, prompts, pe, ne, eng, args.steps, rank, world, args.seed, batch_size=args.batch_size, device=dev)
    torch.cuda.synchronize()
    secs = launch.reduce_max_seconds(dist, time.perf_counter() - t0, device=dev)
    report = launch.gather_report(dist, n, 0.0, device=dev)
    if rank == 0:
        total = sum(c for c, _ in report)
        print(f"{total} images in {secs:.2f} s = {total / secs:.2f} images/s on {world} GPU(s); per rank: {[c for c, _ in report]}")
    if dist is not None:
        dist.destroy_process_group()
    return {"images": n, "use_conv": bool(sch.factor_net.use_conv), "residual": unet.residual, "seconds": secs}


if __name__ == "__main__":
    main()
