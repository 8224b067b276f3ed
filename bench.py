#!/usr/bin/env python
"""Headline benchmark: images/s of 8-step ConsistencySolver sampling at 512x512 on MI355X.

Workload (BASELINE.json configs[1]): SD1.5 UNet fp16 + PPOScheduler (order 4, CFG 3), batch 16
prompts per GPU, 8 solver steps, synthetic seeded weights / prompt embeddings / noise (no
checkpoints exist offline).  A "step" of this bench = one full 8-step generation of one batch
(8 CFG dual-batch UNet forwards at effective batch 32 + 8 fused solver updates); the unit of the
metric is one image's final latents ("latent-images/s", the roofline numerator).  The VAE
decode to 512x512 pixels (decode_latents, utils.py:6-34) is the step after the path:
it is timed in a second, separate K-step loop and reported as "pixel_images_per_s" (latents + decode).

Contract: python bench.py --gpus N --steps K --warmup W ; one JSON line on rank 0.
Multi-GPU: one process per GPU (torch.distributed/RCCL only for the barrier and the max-over-ranks
time); prompts are sharded with no data-path collective (gen_ppo.py:349-357) -> weak scaling.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F16_TFLOPS = 2500.0   # MI355X dense fp16 MFMA (MI355X_MICROARCH.md)


def usable_cores():
    """host cores this process may actually use: min(affinity, cgroup cpu quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def cpu_baseline(sd, cfg, steps_total=8, guidance=3.0, budget_s=40.0, vae_sd=None):
    """The oracle (CPU restatement of the same path, kind 'port') timed on the host cores:
    configs[0] = B=1, 8 steps, CFG 3, fp32 (2 UNet sample-forwards + 1 solver update per step).
    Runs all steps when they fit the time budget, otherwise the steps done are scaled up."""
    import numpy as np
    from oracle.unet_oracle import UNetOracle
    from oracle import solver_oracle as so
    from consolver_amd.synth import synthetic_prompt_embeds
    cores = usable_cores()
    torch.set_num_threads(cores)
    orc = UNetOracle(sd, cfg, round_weights_to_f16=False)
    g = torch.Generator().manual_seed(43)
    S = cfg["sample_size"]
    lat = torch.randn(1, 4, S, S, generator=g).numpy()
    ctx = torch.cat([synthetic_prompt_embeds(1, seed=1002), synthetic_prompt_embeds(1, seed=1001)])
    gw = torch.Generator().manual_seed(20251226)
    w = {}
    for k, shp in (("mlp.0.weight", (256, 2)), ("mlp.0.bias", (256,)), ("mlp.2.weight", (256, 256)), ("mlp.2.bias", (256,)),
                   ("mlp.4.weight", (33, 256)), ("mlp.4.bias", (33,))):
        w[k] = (torch.randn(shp, generator=gw) * 0.5).numpy()
    sch = so.PPOSchedulerOracle(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                timestep_spacing="trailing", order_dim=4, scaler_dim=0, num_actions=11, weights=w)
    sch.set_timesteps(steps_total)
    rng = np.random.default_rng(0)
    done, t0 = 0, time.perf_counter()
    for i, t in enumerate(sch.timesteps):
        e = orc(torch.from_numpy(np.concatenate([lat, lat])), int(t), ctx).numpy()
        lat = sch.step(so.cfg_combine(e[:1], e[1:], guidance), int(t), lat, rng.integers(0, 11, size=(1, 3)))["prev_sample"]
        done += 1
        el = time.perf_counter() - t0
        if el / done * (done + 1) > budget_s:
            break
    el = time.perf_counter() - t0
    per_image = el * steps_total / done
    out = {"value": 1.0 / per_image, "unit": "images/s", "cores": cores, "kind": "port",
           "sample": f"{done} of {steps_total} solver steps of configs[0] (B=1, CFG dual UNet forward fp32 + solver update per step) "
                     f"in {el:.1f} s on {cores} threads" + ("" if done == steps_total else f", scaled x{steps_total}/{done}")}
    if vae_sd is not None:                      # the step after the path: decode_latents (utils.py:6-34) of that image
        from oracle import vae_oracle
        vo = vae_oracle.VaeOracle(vae_sd, round_weights_to_f16=False)
        t1 = time.perf_counter()
        vae_oracle.decode_latents(vo, torch.from_numpy(lat).float(), 1)
        dec = time.perf_counter() - t1
        out["pixel_images_per_s"] = 1.0 / (per_image + dec)
        out["sample"] += f"; + 1 VAE decode fp32 ({dec:.1f} s) for pixel_images_per_s"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per generation")
    ap.add_argument("--num-inference-steps", type=int, default=8)
    ap.add_argument("--guidance", type=float, default=3.0)
    ap.add_argument("--graph", type=int, default=0, help="capture the whole generation in one hipGraph")
    ap.add_argument("--decode", type=int, default=1, help="1: also time K generations WITH the VAE decode (pixel_images_per_s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ceilings", type=int, default=1, help="1: also measure the vendor 8192^3 fp16 GEMM and a 1 GiB device copy on this box")
    ap.add_argument("--profile-kernels", type=int, default=1, help="extra untimed pass with per-kernel-class HIP events")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the one-process-per-GPU job as a CHILD (nothing has touched the GPU yet) and pass its exit code on
        import socket, subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import consolver_amd
    from consolver_amd.unet import HipUNet2DConditionModel
    from consolver_amd.engine import SDSamplingEngine
    from consolver_amd.synth import synthetic_unet_state_dict, synthetic_prompt_embeds, synthetic_vae_state_dict
    from consolver_amd.launch import shard_bounds
    from consolver_amd.vae import HipAutoencoderKL

    unet = HipUNet2DConditionModel(device=dev)
    sd = synthetic_unet_state_dict(unet.manifest(), seed=20251226)
    unet.load_state_dict(sd)
    sch = consolver_amd.PPOScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                     timestep_spacing="trailing", order_dim=4, scaler_dim=0,
                                     factor_net_kwargs=dict(embedding_dim=32, hidden_dim=256, num_actions=11))
    g = torch.Generator().manual_seed(20251226)
    with torch.no_grad():
        for p in sch.factor_net.parameters():      # seeded N(0, 0.5) policy (zero-init = uniform sampling), SURVEY 8(d)
            p.copy_(torch.randn(p.shape, generator=g) * 0.5)
    sch.factor_net.to(dev)
    vae, vae_sd = None, None
    if args.decode:
        vae = HipAutoencoderKL(device=dev)
        vae_sd = synthetic_vae_state_dict(vae.manifest(), seed=20251227)
        vae.load_state_dict(vae_sd)
    eng = SDSamplingEngine(unet, sch, guidance_scale=args.guidance, vae=vae)

    # prompts: global list sharded contiguously over ranks (gen_ppo.py:349-357); every rank gets `batch` per generation
    B, n = args.batch, args.num_inference_steps
    total_prompts = B * world
    lo, hi = shard_bounds(total_prompts, world, rank)
    pe = synthetic_prompt_embeds(total_prompts, seed=1001)[lo:hi].half().to(dev)
    ne = synthetic_prompt_embeds(total_prompts, seed=1002)[lo:hi].half().to(dev)
    gen = torch.Generator().manual_seed(43)                 # readme seed; same noise on every rank like gen_ppo.py:258-260
    noise = torch.randn(B, 4, 64, 64, generator=gen).half().to(dev)

    def one(output_type="latent"):
        return eng.generate(pe, ne, latents=noise, num_inference_steps=n, use_graph=bool(args.graph), output_type=output_type)

    for _ in range(args.warmup):
        one()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    eng.forward_events = [] if not args.graph else None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out).all()

    # ---- roofline of the dominant unit: the UNet forward (MFMA bound), HIP events on the launch stream -------
    eff_batch = 2 * B if args.guidance > 1 else B
    flops_fwd = unet.flops(eff_batch)
    if eng.forward_events:
        fwd_ms = sum(a.elapsed_time(b) for a, b in eng.forward_events) / len(eng.forward_events)
    else:
        fwd_ms = elapsed * 1e3 / (args.steps * n)
    eng.forward_events = None

    # ---- second loop: the same K generations followed by the VAE decode (pixel images) ------------------------
    decode_ms = pixel_elapsed = None
    if args.decode:
        one("pt")                                           # untimed: workspace allocation
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        eng.decode_events = []
        t1 = time.perf_counter()
        for _ in range(args.steps):
            img = one("pt")
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        pixel_elapsed = time.perf_counter() - t1
        if dist is not None:
            tt = torch.tensor([pixel_elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            pixel_elapsed = float(tt.item())
        assert torch.isfinite(img).all() and img.shape == (B, 3, 512, 512)
        decode_ms = sum(a.elapsed_time(b) for a, b in eng.decode_events) / len(eng.decode_events)
        eng.decode_events = None
    achieved = flops_fwd / (fwd_ms * 1e-3) / 1e12
    # HBM-side bytes per UNet forward: PMC counters are collected offline (separate rocprofv3 --pmc passes, tools/profile_round.sh)
    # and committed under profiles/; the newest committed measurement at this batch size is quoted here
    traffic, traffic_src = None, None
    if eff_batch == 32:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))):
            try:
                traffic, traffic_src = json.load(open(f))["traffic_bytes_per_forward"], os.path.relpath(f, ROOT)
            except Exception:
                pass
    roofline = {"bound": "mfma", "achieved": achieved, "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F16_TFLOPS,
                "traffic": traffic, "traffic_source": traffic_src, "kernel": "unet_forward (all kernels of one CFG dual-batch denoiser call)",
                "launch_ms": fwd_ms, "flops_per_launch": flops_fwd}

    kernels = None
    if args.profile_kernels and rank == 0:
        unet.set_profiling(True)
        unet(noise, torch.tensor([499.0], device=dev), encoder_hidden_states=torch.cat([ne, pe]), dup=2, reuse_kv=False)
        prof = unet.profile()
        unet.set_profiling(False)
        kernels = {k: {"ms": round(v["ms"], 3), "launches": v["launches"],
                       "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["ms"] > 0 and v["flops"] > 0 else None,
                       "gbps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else None} for k, v in prof.items()}

    # ---- same-run measured ceilings (makes the fraction of the spec-sheet peak auditable): what the vendor GEMM library and a
    # plain device copy reach on THIS box, on random data.  Reference points only - nothing on the product path uses hipBLASLt or torch math.
    ceilings = None
    if rank == 0 and world == 1 and args.ceilings:
        def _ev(fn, it):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(it): fn()
            b.record(); torch.cuda.synchronize()
            return a.elapsed_time(b) / it
        ga = (torch.randn(8192, 8192, device=dev) * 0.05).half(); gb = (torch.randn(8192, 8192, device=dev) * 0.05).half()
        gc = torch.empty(8192, 8192, device=dev, dtype=torch.float16)
        gms = _ev(lambda: torch.matmul(ga, gb, out=gc), 20)
        src = torch.empty(1 << 30, dtype=torch.uint8, device=dev).random_(0, 255); dst = torch.empty_like(src)
        cms = _ev(lambda: dst.copy_(src), 20)
        ceilings = {"vendor_gemm_f16_8192_tflops": 2.0 * 8192 ** 3 / (gms * 1e-3) / 1e12,
                    "dtod_copy_1gib_tbps": 2.0 * (1 << 30) / (cms * 1e-3) / 1e12,
                    "note": "torch.matmul (hipBLASLt/rocBLAS) 8192^3 fp16 and a 1 GiB device copy (read + write bytes) on this box, random data"}
        del ga, gb, gc, src, dst

    if rank == 0:
        images = B * world * args.steps
        rec = {
            "metric": "images/sec at 8-step ConsistencySolver 512x512 (final latents; SD1.5 UNet fp16 + PPOScheduler, CFG 3)",
            "value": images / elapsed, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic (seeded weights, prompt embeddings and noise of the SD1.5 shapes)",
            "config": {"workload": "configs[1]: SD1.5 + PPOScheduler 8-step fp16, batch 16, 512x512 on 1 MI355X",
                       "batch_per_gpu": B, "num_inference_steps": n, "guidance_scale": args.guidance, "order_dim": 4,
                       "parallelism": f"dp{world} (prompt shards, no data-path collective)", "hipgraph": bool(args.graph)},
            "roofline": roofline,
        }
        if decode_ms is not None:
            rec["pixel_images_per_s"] = images / pixel_elapsed          # latents + AutoencoderKL decode (row f-1), own timed loop
            rec["vae_decode"] = {"ms_per_batch": decode_ms, "tflops": vae.flops(B) / (decode_ms * 1e-3) / 1e12,
                                 "frac_of_mfma_peak": vae.flops(B) / (decode_ms * 1e-3) / 1e12 / PEAK_F16_TFLOPS}
        if kernels:
            rec["roofline_kernels"] = kernels
        if ceilings:
            rec["ceilings"] = ceilings
            rec["roofline"]["frac_of_vendor_gemm"] = achieved / ceilings["vendor_gemm_f16_8192_tflops"]
        if not args.no_cpu_baseline and world == 1:
            rec["cpu_baseline"] = cpu_baseline(sd, unet.config, n, args.guidance, vae_sd=vae_sd)
        elif world == 1:
            rec["cpu_baseline"] = None
        print(json.dumps(rec))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
