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
DEFAULT_RESIDUAL = "f16x2"


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
    # at most 16 threads: the same reduction order on every box of the pool with >= 16 usable cores (the fp32 oracle's own rounding noise, amplified over 8
    # free-running steps, moved `parity` by ~1 % between boxes in round 4)
    cores = min(usable_cores(), 16)
    torch.set_num_threads(cores)
    # weights rounded to fp16 once, arithmetic fp32: the reference pipeline's weights ARE fp16 (gen_ppo.py:193-195), and with them the latents this
    # leg produces are the fp32 reference the HIP engine's latents are gated against (`parity` in the JSON line).  Same CPU time either way.
    orc = UNetOracle(sd, cfg, round_weights_to_f16=True)
    g = torch.Generator().manual_seed(43)
    S = cfg["sample_size"]
    lat = torch.randn(1, 4, S, S, generator=g).half().float().numpy()      # fp16-representable inputs: both sides start from identical values
    noise0 = lat.copy()
    ctx = torch.cat([synthetic_prompt_embeds(1, seed=1002), synthetic_prompt_embeds(1, seed=1001)]).half().float()
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
    idx_used = []
    for i, t in enumerate(sch.timesteps):
        e = orc(torch.from_numpy(np.concatenate([lat, lat])), int(t), ctx).numpy()
        idx_used.append(rng.integers(0, 11, size=(1, 3)))
        lat = sch.step(so.cfg_combine(e[:1], e[1:], guidance), int(t), lat, idx_used[-1], cond_dtype="f16")["prev_sample"]
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
    # private hand-over to parity_vs_oracle (popped before the record is printed): the fp32 reference latents after `done` steps and their inputs
    out["_replay"] = {"latents": lat, "noise": noise0, "ctx": ctx, "idx": idx_used, "weights": w, "steps_done": done, "steps_total": steps_total}
    return out


def parity_vs_oracle(unet, replay, guidance, dev, modes):
    """`latent L2 vs ref`, the second half of BASELINE.json's metric: configs[0]'s inputs (1 prompt, 8 steps, CFG 3, the noise / prompt embeddings /
    replayed action indices / policy weights of the cpu_baseline leg) through the HIP engine on the GPU, relative L2 of the final latents against the
    fp32 oracle latents that leg just produced.  No extra oracle time.  north_star gate: 1e-3."""
    import consolver_amd
    import numpy as np
    n_done, n = replay["steps_done"], replay["steps_total"]
    out = {"gate": 1e-3, "config": f"configs[0] inputs (1 prompt, CFG 3, replayed action indices), {n_done} of {n} solver steps, "
                                   "fp16 HIP engine vs the fp32 CPU oracle (relative L2 of the latents)",
           "reference": "oracle/unet_oracle.py + oracle/solver_oracle.py in fp32 on the host, weights rounded to fp16 ONCE (the reference pipeline's weights are "
                        "fp16, gen_ppo.py:193-195), fp16-representable noise / prompt embeddings, conds in fp16 as scheduler_ppo.py:207 produces them; "
                        "<= 16 oracle threads (reduction order pinned)"}
    want = torch.from_numpy(replay["latents"]).double()
    keep = unet.residual
    from consolver_amd.engine import SDSamplingEngine
    for mode in modes:
        # "f16x2_scheduled" = the engine's defaults (what the timed loop ran); "f16x2" / "f16" = every forward in that mode
        base, hps = ("f16x2", "auto") if mode == "f16x2_scheduled" else (mode, "all")
        unet.set_residual_precision(base)
        sch = consolver_amd.PPOScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="trailing",
                                         order_dim=4, scaler_dim=0, factor_net_kwargs=dict(embedding_dim=32, hidden_dim=256, num_actions=11))
        sch.factor_net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in replay["weights"].items()}, strict=False)
        sch.factor_net.to(dev)
        sch.factor_net.forced_action_idx = [torch.from_numpy(i).to(dev) for i in replay["idx"]]
        ctx = replay["ctx"].half().to(dev)
        if n_done == n:
            # the product's own loop: SDSamplingEngine.generate (fp32 solver state and eps, fused CFG update, precision schedule)
            eng = SDSamplingEngine(unet, sch, guidance_scale=guidance, hi_precision_steps=hps)
            x = eng.generate(ctx[1:], ctx[:1], latents=torch.from_numpy(replay["noise"]).to(dev), num_inference_steps=n)
        else:
            # (a truncated CPU leg: the engine's loop written out for the steps the oracle got through)
            eng = SDSamplingEngine(unet, sch, guidance_scale=guidance, hi_precision_steps=hps)
            k = eng.hi_steps(n) if base == "f16x2" else None
            sch.set_timesteps(n, device=dev)
            x = torch.from_numpy(replay["noise"]).to(dev).float()
            for i, t in enumerate(sch.timesteps[:n_done]):
                kw = {} if k is None else {"residual": "f16x2" if i < k else "f16"}
                eps = unet(x.half(), t, encoder_hidden_states=ctx, dup=2, reuse_kv=(i > 0), out_dtype=torch.float32, **kw)[0]
                x = sch.step(eps[1:], t, x, return_dict=False, eps_uncond=eps[:1], guidance_scale=guidance)[0]
            unet.set_residual_precision_keep(base)
        got = x.double().cpu()
        out[mode] = float((got - want).norm() / want.norm())
    unet.set_residual_precision(keep)
    unet.invalidate_kv()
    return out


PEAK_HBM_GBPS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md; 6.29 TB/s measured float4 copy)


def live_traffic(mode, timeout_s=240):
    """HBM-side bytes per UNet forward measured IN THIS RUN: two child processes `rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python3 tools/bench_unet.py 2`
    (separate passes, counters only, the program directly after `--`; the children are started from here, nothing is exec'ed), summed over all dispatches and
    divided by the forwards of the child (conv_in runs once per forward: `conv_in_kernel`, or `latent_to_nhwc64_kernel` when it is on the MFMA conv).  gfx950 corrections as MI355X_MICROARCH.md prescribes: FETCH_SIZE counts 128-byte
    requests at 64 bytes -> x2; WRITE_SIZE exact for 16-byte-per-lane stores; unit KB.  Returns (bytes, source) or (None, reason)."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    tot, fwd = {}, {}
    tmp = tempfile.mkdtemp(prefix="cs_pmc_", dir="/tmp")
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, c)
            env = dict(os.environ, TMPDIR="/tmp", CS_RESIDUAL=mode)
            r = subprocess.run([exe, "--pmc", c, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.join(ROOT, "tools", "bench_unet.py"), "2"],
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {c} failed (rc {r.returncode})"
            t, n = 0.0, 0
            for f in files:
                for row in csv.DictReader(open(f)):
                    t += float(row["Counter_Value"])
                    n += ("conv_in_kernel" in row["Kernel_Name"]) or ("latent_to_nhwc64_kernel" in row["Kernel_Name"])
            if n == 0:
                return None, "no conv_in dispatch in the counter file: cannot count the forwards"
            tot[c], fwd[c] = t, n
        b = tot["FETCH_SIZE"] * 1024 * 2 / fwd["FETCH_SIZE"] + tot["WRITE_SIZE"] * 1024 / fwd["WRITE_SIZE"]
        return b, (f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/bench_unet.py 2 in this run, residual stream {mode}, "
                   f"{fwd['FETCH_SIZE']} forwards; FETCH x2 (gfx950: 128-B requests counted at 64 B), KB units")
    except Exception as e:                       # a profiler problem must not take the headline line down
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


class PowerSampler:
    """mean socket power / shader clock over a timed region from the amdgpu hwmon files (sysfs; the same source rocm-smi reads), sampled by a host thread every 20 ms.
    Readable as an ordinary user on most boxes; when it is not, `summary()` says so and the bench line carries null."""

    def __init__(self, device_index=0):
        import glob
        self.paths = {}
        cards = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        withp = [c for c in cards if os.path.exists(os.path.join(c, "power1_average")) or os.path.exists(os.path.join(c, "power1_input"))]
        self.matched_by = None
        if withp:
            # the card whose PCI address is the HIP device's (a box may expose eight cards in sysfs and one GPU to the process: round 6 read an idle neighbour's
            # 242 W / 99 MHz by index); else the device_index-th card that has a power file
            h = None
            try:
                pr = torch.cuda.get_device_properties(device_index)
                want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
                for c in withp:
                    if os.path.basename(os.path.realpath(os.path.join(c, "..", ".."))).lower().startswith(want):
                        h, self.matched_by = c, "pci " + want
                        break
            except Exception:
                h = None
            if h is None:
                h, self.matched_by = withp[min(device_index, len(withp) - 1)], f"index {device_index} of {len(withp)} cards (no PCI match)"
            for key, names in (("power_uW", ("power1_average", "power1_input")), ("sclk_Hz", ("freq1_input",))):
                for nme in names:
                    if os.path.exists(os.path.join(h, nme)):
                        self.paths[key] = os.path.join(h, nme)
                        break
        self.samples = {k: [] for k in self.paths}
        self._stop = None
        self._thr = None

    def _read(self):
        for k, pth in self.paths.items():
            try:
                with open(pth) as f:
                    self.samples[k].append(float(f.read().strip()))
            except Exception:
                pass

    def start(self):
        if not self.paths:
            return self
        import threading
        self._stop = threading.Event()

        def run():
            while not self._stop.is_set():
                self._read()
                self._stop.wait(0.02)
        self._thr = threading.Thread(target=run, daemon=True)
        self._thr.start()
        return self

    def stop(self):
        if self._thr is not None:
            self._stop.set()
            self._thr.join(timeout=1.0)
            self._thr = None
        return self.summary()

    def summary(self):
        if not self.paths:
            return {"available": False, "why": "no readable amdgpu hwmon power file under /sys/class/drm/card*/device/hwmon"}
        out = {"available": True, "samples": max((len(v) for v in self.samples.values()), default=0), "source": "sysfs hwmon (amdgpu), 20 ms period", "card": self.matched_by}
        if self.samples.get("power_uW"):
            v = self.samples["power_uW"]
            out["mean_socket_power_w"] = sum(v) / len(v) / 1e6
            out["max_socket_power_w"] = max(v) / 1e6
        if self.samples.get("sclk_Hz"):
            v = self.samples["sclk_Hz"]
            out["mean_sclk_mhz"] = sum(v) / len(v) / 1e6
            out["min_sclk_mhz"] = min(v) / 1e6
        return out


def _timed_ms(fn, iters, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        out = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters, out


def extra_four_step(eng, unet, pe, ne, noise, guidance, iters):
    """configs[2]'s per-GPU shape: 4-step generation, batch 16, CFG 3 (gen_ppo.py path; 128 prompts over 8 GPUs)."""
    B = noise.shape[0]
    ms, out = _timed_ms(lambda: eng.generate(pe, ne, latents=noise, num_inference_steps=4), iters)
    tf = unet.flops(2 * B) * 4 / (ms * 1e-3) / 1e12
    return {"workload": "configs[2] per-GPU shape: SD1.5 + PPOScheduler 4-step CFG 3, batch 16", "ms_per_generation": ms,
            "images_per_s": B / (ms * 1e-3), "tflops": tf, "frac_of_mfma_peak": tf / PEAK_F16_TFLOPS, "finite": bool(torch.isfinite(out).all())}


def extra_solver(dev, sch):
    """the fused solver-update kernel K1 (CFG combine + coefficient fix-up + LMS combine + DDIM, one pass) and the policy MLP (K2).
    K1 bytes per step: [x] + [eps_u, eps_c] + [m-1 history] + [x'] + [eps store] passes of B x 32 KiB."""
    import ctypes as C
    from consolver_amd import _lib as L
    lib = L.lib()

    def k1(B, iters):
        # the variant the ENGINE runs (engine.py defaults, round 6): fp32 solver state AND fp32 eps (the UNet's conv_out stores its accumulator unrounded), fp32 history
        # ring and combined-eps store, plus the fp16 copy of the result for the denoiser written in the same pass (x_out_lp).  Per element: x, eps_u, eps_c, 3 history
        # entries in (6 x 4 B); x', combined eps (2 x 4 B) and x' fp16 (2 B) out = 34 B (the fp16 kernel of rounds 1-4: 8 passes of 2 B)
        t = lambda: torch.randn(B, 16384, device=dev)
        x, eu, ec, eo, h1, h2, h3, out = (t() for _ in range(8))
        out_lp = torch.empty(B, 16384, device=dev, dtype=torch.float16)
        actions = torch.rand(B, 3, device=dev)
        a = L.CsStepArgs()
        a.x, a.eps_text, a.eps_uncond, a.guidance = x.data_ptr(), ec.data_ptr(), eu.data_ptr(), 3.0
        for k, h in enumerate((h1, h2, h3)):
            a.hist[k] = h.data_ptr()
        a.m, a.order_dim, a.scaler_dim = 4, 4, 0
        a.actions, a.actions_stride, a.B, a.elems = actions.data_ptr(), 3, B, 16384
        a.io_dtype = a.out_dtype = L.dtype_code(torch.float32)
        a.x_out, a.eps_out, a.x_out_lp, a.lp_dtype = out.data_ptr(), eo.data_ptr(), out_lp.data_ptr(), L.dtype_code(torch.float16)
        a.sqrt_at, a.sqrt_1mat, a.sqrt_ap, a.sqrt_1map = 0.3, 0.95, 0.5, 0.86
        st = L.stream_ptr(dev)
        ms, _ = _timed_ms(lambda: L.check(lib.cs_lms_ddim_step(C.byref(a), st)), iters, warm=3)
        nbytes = 34 * B * 16384
        return ms * 1e3, nbytes / (ms * 1e-3) / 1e9
    us16, gb16 = k1(16, 200)
    us4k, gb4k = k1(4096, 20)
    row = torch.tensor([[874.0, 749.0]], device=dev)
    net = sch.factor_net
    pol_ms, _ = _timed_ms(lambda: net.probs_from(row, batch=16), 200, warm=5)
    # the whole per-step solver chain as the engine drives it: scheduler.step on a CFG pair at batch 16 = policy MLP + torch.rand + inverse-CDF sample + the fused
    # update (masks / conds come from the scheduler's per-shape caches): wall per call, back to back, host launch path included
    B, n = 16, 8
    eps = torch.randn(2 * B, 4, 64, 64, device=dev)
    xs = [torch.randn(B, 4, 64, 64, device=dev), torch.empty(B, 4, 64, 64, device=dev)]
    x16 = torch.empty(B, 4, 64, 64, device=dev, dtype=torch.float16)
    ring = [torch.empty(B, 4, 64, 64, device=dev) for _ in range(4)]
    keep = (sch.num_inference_steps, list(sch.ets))

    def chain():
        sch.set_timesteps(n, device=dev)
        for i in range(n):
            sch.step(eps[B:], sch.timesteps[i], xs[i & 1], return_dict=False, eps_uncond=eps[:B], guidance_scale=3.0, eps_out=ring[i % 4], out=xs[(i & 1) ^ 1], out_lp=x16)
    chain_ms, _ = _timed_ms(chain, 20, warm=2)
    # the same chain as the GPU sees it: captured into one hipGraph (no host in the loop), replayed back to back
    chain_gpu_us = None
    try:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            chain()
        torch.cuda.current_stream(dev).wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            chain()
        gms, _ = _timed_ms(gr.replay, 50, warm=3)
        chain_gpu_us = gms * 1e3 / n
        del gr
    except Exception as e:
        chain_gpu_us = f"{type(e).__name__}: {e}"
    if keep[0] is not None:
        sch.set_timesteps(keep[0], device=dev)
    return {"k1_us_at_batch16": us16, "k1_gbps_at_batch16": gb16, "k1_us_at_1GB": us4k, "k1_gbps_at_1GB": gb4k,
            "k1_frac_of_hbm_peak_at_1GB": gb4k / PEAK_HBM_GBPS, "k1_note": "the engine's variant: fp32 state, eps and history, CFG, order 4 steady state, fp16 copy of the result in the same pass: 34 B per element; "
            "batch 16 (8.9 MB) is launch-latency bound, the 2.3 GB working set is what the kernel sustains when HBM-bound",
            "policy_mlp_us_at_batch16": pol_ms * 1e3, "policy_note": "cs_factor_probs, hidden 256, one conditioning row broadcast to 16 samples; "
            "wall per call incl. the host launch path (back-to-back launches)",
            "solver_chain_us_per_step": chain_ms * 1e3 / n, "solver_chain_gpu_us_per_step": chain_gpu_us,
            "solver_chain_note": "PPOScheduler.step on a CFG pair at batch 16, fp32 state and eps: policy MLP + torch.rand + inverse-CDF sample + fused update (incl. the fp16 copy of the "
            "latents); 4 launches per step (round 5: 8 -- masks, two conds launches and the fp32 -> fp16 cast are gone).  solver_chain_us_per_step = wall per Python call, back to back (the host "
            "path: in the sampling loop it runs ahead under the 28 ms denoiser forward); solver_chain_gpu_us_per_step = the same 8 steps captured in one hipGraph and replayed: what the chain costs the GPU"}


def extra_rollout(unet, vae, dev, B=80, n=8, epochs=4, iters=2):
    """configs[4] on one GPU: one PPO trainer iteration (train_ppo.py:322-437) = B trajectories x n steps (CFG 3 -> UNet batch 2B),
    decode of the B predictions + B teacher latents, image-PSNR reward, advantages, `epochs` policy updates."""
    import consolver_amd
    from consolver_amd import ppo
    from consolver_amd.synth import synthetic_prompt_embeds
    sch = consolver_amd.PPOScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", timestep_spacing="trailing",
                                     order_dim=4, scaler_dim=0, factor_net_kwargs=dict(embedding_dim=32, hidden_dim=256, num_actions=11))
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for p in sch.factor_net.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.5)
    sch.factor_net.to(dev)
    tr = ppo.PolicyTrainer(sch.factor_net, lr=1e-4)
    pe = synthetic_prompt_embeds(1, seed=1001).half().to(dev).repeat(B, 1, 1)
    ne = synthetic_prompt_embeds(1, seed=1002).half().to(dev).repeat(B, 1, 1)
    batch = (["p"] * B, torch.randn(1, 4, 64, 64, generator=g).half().to(dev).repeat(B, 1, 1, 1),
             (torch.randn(1, 4, 64, 64, generator=g) * 0.18).half().to(dev).repeat(B, 1, 1, 1))
    ms, out = _timed_ms(lambda: ppo.train_iteration(tr, None, sch, unet, vae, batch, None, cfg=3.0, num_inference_steps=n, ppo_epochs=epochs,
                                                    prompt_embeds=pe, negative_prompt_embeds=ne), iters)
    fl_ref = unet.flops(2 * B) * n + vae.flops(8) * (2 * B / 8)
    # executed: the B rows are copies of one sample (repeat_random_sample, data_processing.py:65-83): steps 0 and 1 of the rollout run the
    # denoiser for one row, the teacher latent is decoded once (~25 % of the reference loop's FLOPs are not run at all)
    fl_exec = unet.flops_executed(B, 2) * (n - 2) + unet.flops_executed(1, 2) * 2 + vae.flops(8) * (B / 8) + vae.flops(1)
    tf = fl_exec / (ms * 1e-3) / 1e12
    return {"workload": f"configs[4] on 1 GPU: PPO rollout B={B}, {n} steps, CFG 3, 2x{B} VAE decodes, image_psnr reward, {epochs} PPO epochs",
            "ms_per_iteration": ms, "trajectories_per_s": B / (ms * 1e-3), "tflops": tf, "frac_of_mfma_peak": tf / PEAK_F16_TFLOPS,
            "tflops_reference_equivalent": fl_ref / (ms * 1e-3) / 1e12,
            "note": "tflops / frac_of_mfma_peak count the FLOPs the hardware EXECUTED (identical-input sharing: steps 0-1 for one row, one teacher decode); "
                    "tflops_reference_equivalent = the reference loop's algorithmic FLOPs (B full trajectories, 2B decodes) / the same time",
            "loss_finite": bool(torch.isfinite(torch.as_tensor(float(out["loss"]))))}


def extra_flux(dev, n=8):
    """configs[3]: full FLUX.1-Kontext DiT (19 + 38 blocks, 11.9 B synthetic bf16 parameters generated on the GPU) + FMPPOScheduler,
    8-step 1024x1024 edit, batch 1; finiteness + run-to-run determinism of the full-size edit (fixed action indices)."""
    import consolver_amd
    from consolver_amd.flux import HipFluxTransformer2DModel, FluxKontextSamplingEngine, pack_latents
    m = HipFluxTransformer2DModel(device=dev)
    g = torch.Generator(device=dev).manual_seed(20251226)
    nparams = 0
    for name, shape in m.manifest():
        if name.endswith(("norm_q.weight", "norm_k.weight", "norm_added_q.weight", "norm_added_k.weight")):
            w = 1.0 + 0.1 * torch.randn(shape, generator=g, device=dev)
        elif name.endswith(".weight"):
            w = torch.randn(shape, generator=g, device=dev) * (1.0 / shape[1]) ** 0.5
            if ".norm" in name and name.endswith("linear.weight"):
                w = w * 0.5
        else:
            w = 0.05 * torch.randn(shape, generator=g, device=dev)
            if ".norm" in name and name.endswith("linear.bias"):
                w = w + 0.3
        nparams += w.numel()
        m.set_weight(name, w)
        del w
    m.finalize()
    B, T, Lq = 1, 512, 4096
    sch = consolver_amd.FMPPOScheduler.from_pretrained("black-forest-labs/FLUX.1-Kontext-dev", subfolder="scheduler", order_dim=2, scaler_dim=0,
                                                       mu_dim=0, factor_net_kwargs=dict(embedding_dim=32, hidden_dim=256, num_actions=11))
    sch.factor_net.to(dev)
    sch.factor_net.forced_action_idx = torch.full((B, 1), 3, dtype=torch.long, device=dev)      # fixed policy draw: determinism check
    gc = torch.Generator().manual_seed(43)
    lat = pack_latents(torch.randn(B, 16, 128, 128, generator=gc)).to(torch.bfloat16).to(dev)
    img = pack_latents(torch.randn(B, 16, 128, 128, generator=gc)).to(torch.bfloat16).to(dev)
    enc = torch.nn.functional.layer_norm(torch.randn(B, T, 4096, generator=gc), (4096,)).to(torch.bfloat16).to(dev)
    pooled = torch.randn(B, 768, generator=gc).to(torch.bfloat16).to(dev)
    eng = FluxKontextSamplingEngine(m, sch, guidance_scale=2.5)
    first = eng.generate(lat, img, enc, pooled, latent_hw=(64, 64), num_inference_steps=n).clone()       # warm-up, sizes the workspace
    ms, out = _timed_ms(lambda: eng.generate(lat, img, enc, pooled, latent_hw=(64, 64), num_inference_steps=n), 2, warm=0)
    fl = m.flops(B, T, 2 * Lq)
    tf = fl * n / (ms * 1e-3) / 1e12
    rec = {"workload": "configs[3]: FLUX-Kontext + scheduler_fmppo 8-step bf16, 1024x1024 edit on 1 MI355X", "params_B": nparams / 1e9,
           "ms_per_edit": ms, "edits_per_s": 1e3 / ms, "tflop_per_forward": fl / 1e12, "tflops": tf, "frac_of_mfma_peak": tf / PEAK_F16_TFLOPS,
           "dtype": "bf16", "finite": bool(torch.isfinite(out.float()).all()), "deterministic": bool(torch.equal(out, first))}
    del m, eng, lat, img, enc, pooled, first, out
    torch.cuda.empty_cache()
    return rec


def dry_run(args):
    """`--dry-run`: everything bench.py does around the GPU work and nothing on a GPU -- the rendezvous (gloo instead of RCCL), the shard
    bounds, the start barrier, the max-over-ranks reduction and the one JSON line -- so that the N > 1 launch path can be exercised in the
    CPU-only build container (tests/test_launch_cpu.py).  No CUDA call is made."""
    from consolver_amd.launch import shard_bounds
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    B = args.batch
    lo, hi = shard_bounds(B * world, world, rank)
    if dist is not None:
        dist.barrier()
    elapsed = 0.001 * (rank + 1)                      # stand-in for this rank's timed region
    bounds = [[lo, hi]]
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        all_b = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(all_b, torch.tensor([lo, hi], dtype=torch.int64))
        bounds = [b.tolist() for b in all_b]
    # the same per-rank record main() gathers over RCCL (rank, local rank, device stand-in, images, elapsed, shard, checksum stand-in)
    from consolver_amd.launch import gather_rank_records
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import socket, zlib
    rows = gather_rank_records(dist, [rank, local, local, B * args.steps, 0.001 * (rank + 1), lo, hi, float(lo + hi), float(zlib.crc32(socket.gethostname().encode()))])
    assert len(rows) == world
    per_rank = [{"rank": int(r[0]), "local_rank": int(r[1]), "device": int(r[2]), "images": int(r[3]), "elapsed_s": r[4],
                 "prompt_shard": [int(r[5]), int(r[6])], "latent_checksum": r[7], "host_id": int(r[8])} for r in rows]
    # what main() runs OUTSIDE the timed region on this launch: the cpu_baseline leg and the sub-record extras are N = 1 only (same conditions as in
    # main()), so an 8-GPU timed region and its barriers see nothing but the sampling loop
    side_work = {"cpu_baseline": world == 1 and not args.no_cpu_baseline, "extras": world == 1 and bool(args.extras),
                 "ceilings": world == 1 and bool(args.ceilings), "kernel_profile_pass": rank == 0 and bool(args.profile_kernels)}
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": "weak",
                          "images": B * world * args.steps, "max_elapsed_s": elapsed, "shards": bounds, "per_rank": per_rank,
                          "per_rank_distinct_devices": len({(p["host_id"], p["device"]) for p in per_rank}),
                          "local_rank_env": os.environ.get("LOCAL_RANK"), "cuda_initialised": torch.cuda.is_initialized(),
                          "side_work_after_timed_region": side_work}))
    if dist is not None:
        dist.destroy_process_group()


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
    ap.add_argument("--extras", type=int, default=1, help="1 (N = 1 only): also measure configs[2] per-GPU shape, configs[3] (full FLUX edit), "
                    "configs[4] (PPO rollout iteration) and the solver kernels, reported as sub-records of the same JSON line")
    ap.add_argument("--residual", default=DEFAULT_RESIDUAL, choices=["f16", "f16x2", "residual_fp32"],
                    help="residual-stream storage of the UNet executor for the headline number (include/consolver_hip.h CS_RESIDUAL_*): f16x2 "
                         "(split-fp16, meets the 1e-3 latent gate) or f16 (one plane, the reference pipeline's own arithmetic class); the other "
                         "mode is timed too and reported under `residual_modes`")
    ap.add_argument("--traffic", default="live", choices=["live", "committed", "none"],
                    help="roofline.traffic: 'live' (N = 1) measures FETCH_SIZE / WRITE_SIZE with two rocprofv3 --pmc child runs of tools/bench_unet.py after the "
                         "timed region (falls back to the newest committed profiles/r*_pmc_traffic.json), 'committed' quotes that file only")
    ap.add_argument("--hi-precision-steps", default="auto", help="the engine's precision schedule of the UNet's residual stream (engine.py): 'auto' (default: the first "
                    "ceil(n / 4) forwards of a generation on the split stream, the rest on one fp16 plane), 'all' (every forward in --residual's mode), or an int")
    ap.add_argument("--force-dist", action="store_true", help="initialise the RCCL process group and run the N > 1 barrier / all_reduce(MAX) / per-rank all-gather "
                    "branch even at world size 1 (a one-GPU box exercising the exact code an 8-GPU launch runs; tests/test_engine_gpu.py)")
    ap.add_argument("--dry-run", action="store_true", help="exercise the launch / rendezvous / sharding / reduction path without touching a GPU")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the one-process-per-GPU job as a CHILD (nothing has touched the GPU yet) and pass its exit code on
        import socket, subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    if args.dry_run:
        return dry_run(args)
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
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:                      # (--force-dist outside torch.distributed.run: a free local port)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import consolver_amd
    from consolver_amd.unet import HipUNet2DConditionModel
    from consolver_amd.engine import SDSamplingEngine
    from consolver_amd.synth import synthetic_unet_state_dict, synthetic_prompt_embeds, synthetic_vae_state_dict
    from consolver_amd.launch import shard_bounds
    from consolver_amd.vae import HipAutoencoderKL

    headline_mode = "f16x2" if args.residual in ("f16x2", "residual_fp32") else "f16"
    other_mode = "f16" if headline_mode == "f16x2" else "f16x2"
    unet = HipUNet2DConditionModel(device=dev, residual=headline_mode)
    # seeded synthetic weights generated ON THIS RANK'S GPU (860 M values: every rank of an 8-GPU launch synthesising them on the shared host cores was the
    # start-up cost of the N > 1 run); same seed on every rank -> identical weights, and the host copies below are what the CPU oracle legs read
    sd = synthetic_unet_state_dict(unet.manifest(), seed=20251226, device=dev)
    sd = {k: v.cpu() for k, v in sd.items()}
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
    hps = args.hi_precision_steps if args.hi_precision_steps in ("auto", "all") else int(args.hi_precision_steps)
    eng = SDSamplingEngine(unet, sch, guidance_scale=args.guidance, vae=vae, hi_precision_steps=hps)

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
    power = PowerSampler(local).start() if rank == 0 else None      # (a host thread reading two sysfs files every 20 ms; rank 0 only)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    power = power.stop() if power is not None else None
    elapsed_local = elapsed
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out).all()

    # ---- N > 1: every rank's own record, gathered over RCCL (proves the collective saw `world` ranks on `world` distinct devices) ----
    per_rank = None
    if dist is not None:
        from consolver_amd.launch import gather_rank_records
        import socket, zlib
        host_id = float(zlib.crc32(socket.gethostname().encode()))       # (exact in a double; the local device index alone repeats across nodes)
        mine = [float(rank), float(local), float(torch.cuda.current_device()), float(B * args.steps), float(elapsed_local), float(lo), float(hi),
                float(out.float().abs().sum().item()), host_id]
        rows = gather_rank_records(dist, mine, dev)
        assert len(rows) == world, (len(rows), world)
        if args.force_dist:
            # (--force-dist: also the other RCCL-side helpers of launch.py on GPU tensors -- the max over ranks the drivers report (gen_ppo.py:433-465), the
            #  count / checksum gather, the trainer's gradient mean (train_ppo.py:257,430) -- so that one process-group initialisation covers all of them in the GPU suite)
            from consolver_amd import launch as _launch
            assert _launch.reduce_max_seconds(dist, 1.5, dev) == 1.5
            rep = _launch.gather_report(dist, B * args.steps, 2.25, dev)
            assert len(rep) == world and rep[rank] == (B * args.steps, 2.25), rep
            gvec = torch.arange(75000, dtype=torch.float32, device=dev)
            assert torch.equal(_launch.average_gradients(dist, gvec.clone()), gvec)      # (identical on every rank: the mean is the vector itself)
            assert dist.get_backend() == "nccl"
        per_rank = [{"rank": int(r[0]), "local_rank": int(r[1]), "device": int(r[2]), "images": int(r[3]), "elapsed_s": r[4],
                     "prompt_shard": [int(r[5]), int(r[6])], "latent_checksum": r[7], "host_id": int(r[8])} for r in rows]
        assert sorted(p["rank"] for p in per_rank) == list(range(world)), per_rank
        # one rank per (host, device): recorded, not asserted -- a launch with two ranks per GPU is legal and must still print its line
        distinct = len({(p["host_id"], p["device"]) for p in per_rank})

    # ---- roofline of the dominant unit: the UNet forward (MFMA bound), HIP events on the launch stream -------
    eff_batch = 2 * B if args.guidance > 1 else B
    flops_fwd = unet.flops(eff_batch)
    k_hi = eng.hi_steps(n) if headline_mode == "f16x2" else None      # forwards per generation on the split stream (None: all)
    fwd_by_mode = None
    if eng.forward_events:
        ev = [a.elapsed_time(b) for a, b in eng.forward_events]
        fwd_ms = sum(ev) / len(ev)
        if k_hi is not None and k_hi < n:
            hi = [v for j, v in enumerate(ev) if j % n < k_hi]; lo = [v for j, v in enumerate(ev) if j % n >= k_hi]
            fwd_by_mode = {"f16x2": sum(hi) / len(hi), "f16": sum(lo) / len(lo), "forwards_per_generation": {"f16x2": k_hi, "f16": n - k_hi}}
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
    if eff_batch == 32 and args.traffic != "none":
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))):
            try:
                traffic, traffic_src = json.load(open(f))["traffic_bytes_per_forward"], os.path.relpath(f, ROOT)
            except Exception:
                pass
    # flops_per_launch is the ALGORITHMIC count of the reference graph (32 x 803.27 GFLOP): `achieved` / `frac` are quoted on it as the
    # contract asks.  The executor evaluates the layers in front of the first cross attention once for both CFG halves (same latents and
    # timestep: bit-identical results, tests/test_unet_gpu.py), so ~2.5 % fewer FLOPs are actually executed: `achieved_executed` / `frac_executed`
    # are what the matrix pipes did.  Round 6: the forwards on one fp16 plane also run the three upsamplers in the sub-pixel form (16 instead of 36 multiplies per
    # input pixel on pre-summed taps, knob up_fold): another 4.7 % fewer executed FLOPs in those forwards; under the precision schedule the mean of the mix is quoted.
    def executed_in(mode):
        cur = unet.residual
        unet.set_residual_precision_keep(mode)
        v = unet.flops_executed(B, 2) if args.guidance > 1 else unet.flops_executed(B, 1)
        unet.set_residual_precision_keep(cur)
        return v
    if k_hi is not None and k_hi < n:
        fx = {"f16x2": executed_in("f16x2"), "f16": executed_in("f16")}
        flops_exec = (k_hi * fx["f16x2"] + (n - k_hi) * fx["f16"]) / n
    else:
        fx = None
        flops_exec = executed_in(unet.residual)
    achieved_exec = flops_exec / (fwd_ms * 1e-3) / 1e12
    roofline = {"bound": "mfma", "achieved": achieved, "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F16_TFLOPS,
                "achieved_executed": achieved_exec, "frac_executed": achieved_exec / PEAK_F16_TFLOPS,
                "traffic": traffic, "traffic_source": traffic_src, "kernel": "unet_forward (all kernels of one CFG dual-batch denoiser call)",
                "launch_ms": fwd_ms, "flops_per_launch": flops_fwd, "flops_executed_per_launch": flops_exec}
    if fx:
        roofline["flops_executed_by_stream_mode"] = fx
    if fwd_by_mode:
        roofline["launch_ms_by_stream_mode"] = fwd_by_mode
        roofline["launch_note"] = (f"launch_ms is the mean over ALL forwards of the timed generations: the engine's precision schedule runs the first {k_hi} of {n} forwards of a "
                                   "generation on the split (hi + lo) residual stream and the rest on one fp16 plane (engine.py; the 1e-3 gate is asserted at every step "
                                   "under this schedule, tests/test_parity_e2e_gpu.py; `parity` below is measured under it); launch_ms_by_stream_mode splits the same events")

    # ---- the other residual-stream mode, same K generations, same box, same process (N = 1 only: nothing else may sit between an 8-GPU run's barriers) ----
    sched = k_hi is not None and k_hi < n
    head_key = "f16x2_scheduled" if sched else headline_mode
    modes = {head_key: {"images_per_s": B * world * args.steps / elapsed, "unet_forward_ms": fwd_ms, "roofline_frac": achieved / PEAK_F16_TFLOPS,
                        "headline": True}}
    if sched:
        modes[head_key]["forwards_on_split_stream"] = f"{k_hi} of {n}"
    if world == 1 and not args.graph and sched:
        # the same K generations with EVERY forward on the split stream (round 5's headline configuration)
        eng.hi_precision_steps = "all"
        one(); torch.cuda.synchronize()
        eng.forward_events = []
        t3 = time.perf_counter()
        for _ in range(args.steps):
            o3 = one()
        torch.cuda.synchronize()
        el3 = time.perf_counter() - t3
        f3 = sum(a.elapsed_time(b) for a, b in eng.forward_events) / len(eng.forward_events)
        eng.forward_events = None
        eng.hi_precision_steps = hps
        assert torch.isfinite(o3).all()
        modes["f16x2"] = {"images_per_s": B * args.steps / el3, "unet_forward_ms": f3, "roofline_frac": flops_fwd / (f3 * 1e-3) / 1e12 / PEAK_F16_TFLOPS, "headline": False}
    if world == 1 and not args.graph:
        unet.set_residual_precision(other_mode)
        one(); torch.cuda.synchronize()
        eng.forward_events = []
        t2 = time.perf_counter()
        for _ in range(args.steps):
            o2 = one()
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t2
        f2 = sum(a.elapsed_time(b) for a, b in eng.forward_events) / len(eng.forward_events)
        eng.forward_events = None
        assert torch.isfinite(o2).all()
        modes[other_mode] = {"images_per_s": B * args.steps / el2, "unet_forward_ms": f2, "roofline_frac": flops_fwd / (f2 * 1e-3) / 1e12 / PEAK_F16_TFLOPS,
                             "headline": False}
        unet.set_residual_precision(headline_mode)
        one(); torch.cuda.synchronize()                    # workspace of the headline mode back in place for the passes below
    modes["note"] = ("residual-stream storage of the UNet executor (include/consolver_hip.h): f16x2 = split-fp16 hi + lo planes, fp32-class adds along "
                     "the stream; f16 = one plane, the reference fp16 pipeline's own arithmetic class (1.2e-3 over 8 steps: above the 1e-3 latent gate).  GEMM operands are "
                     "fp16 in both.  f16x2_scheduled (the engine's default): the first ceil(n / 4) forwards of a generation on the split stream -- where the gate's "
                     "budget is spent -- the rest on one plane; every step of the trajectory is under the gate (parity).")

    # per-kernel-class HIP-event profile of one forward, in BOTH residual-stream modes (N = 1): what the lo planes cost per class is visible on the driver's box.
    # The executor's byte model of the `f16` mode (one fp16 plane per tensor) is the reference graph's ALGORITHMIC traffic: `traffic_ratio` below is quoted on it,
    # whatever mode the headline runs in (the lo planes are a representation cost, not algorithmic work).
    kernels, kernels_other, algo_bytes_one_plane = None, None, None

    def _profile_pass():
        unet.set_profiling(True)
        unet(noise, torch.tensor([499.0], device=dev), encoder_hidden_states=torch.cat([ne, pe]), dup=2, reuse_kv=False)
        prof = unet.profile()
        unet.set_profiling(False)
        return ({k: {"ms": round(v["ms"], 3), "launches": v["launches"],
                     "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["ms"] > 0 and v["flops"] > 0 else None,
                     "gbps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else None} for k, v in prof.items()},
                sum(v["bytes"] for v in prof.values()))
    if args.profile_kernels and rank == 0:
        kernels, b_head = _profile_pass()
        if headline_mode == "f16":
            algo_bytes_one_plane = b_head
        if world == 1 and not args.graph:
            unet.set_residual_precision(other_mode)
            kernels_other, b_other = _profile_pass()
            if other_mode == "f16":
                algo_bytes_one_plane = b_other
            unet.set_residual_precision(headline_mode)
            one(); torch.cuda.synchronize()
    roofline["traffic_algorithmic_one_plane"] = algo_bytes_one_plane

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
        # SUSTAINED ceilings: the boxes of the pool differ by up to 8 % on one build and a 20 ms GEMM burst does not see it (it is not power-limited; the 28 ms forward,
        # run back to back for seconds, is).  Two >= 250 ms back-to-back loops travel with every line as this box's index: the vendor GEMM again, and ONE fixed product
        # kernel -- the 64 x 64 320 -> 320 3x3 conv at batch 32 (conv3_lw_kernel, the forward's dominant kernel) -- so that `roofline.frac` of two driver runs can be
        # compared through `frac / sustained`.
        try:
            from consolver_amd import ops as _ops
            gms_s = _ev(lambda: torch.matmul(ga, gb, out=gc), 300)
            cx = (torch.randn(32, 64, 64, 320, device=dev)).half()
            cw = _ops.pack_conv_weight(torch.randn(320, 320, 3, 3) * (1.0 / 2880) ** 0.5).to(dev)
            cb = torch.zeros(320, device=dev, dtype=torch.float16)
            cms_s = _ev(lambda: _ops.conv2d(cx, cw, cb, taps=9), 1500)
            cfl = 2.0 * 32 * 4096 * 2880 * 320
            ceilings.update({"sustained_vendor_gemm_f16_8192_tflops": 2.0 * 8192 ** 3 / (gms_s * 1e-3) / 1e12,
                             "sustained_conv3_lw_64x64_320_tflops": cfl / (cms_s * 1e-3) / 1e12, "sustained_conv3_lw_us_per_launch": cms_s * 1e3,
                             "sustained_note": "300 back-to-back 8192^3 GEMMs (~0.3 s) and 1500 back-to-back launches of the product's own 3x3 conv kernel at 64 x 64, 320 -> 320, "
                                               "batch 32 (~0.3 s, random fp16 data): what this box sustains under the power limit"})
            del cx, cw, cb
        except Exception as e:
            ceilings["sustained_error"] = f"{type(e).__name__}: {e}"
        del ga, gb, gc, src, dst

    # ---- the other configurations of BASELINE.json, measured in the same driver-run process (sub-records; N = 1 only) -------------
    extras = None
    if rank == 0 and world == 1 and args.extras:
        extras = {}
        for name, fn in (("four_step", lambda: extra_four_step(eng, unet, pe, ne, noise, args.guidance, max(2, args.steps))),
                         ("solver_k1", lambda: extra_solver(dev, sch)),
                         ("rollout_iteration", lambda: extra_rollout(unet, vae, dev) if vae is not None else None),
                         ("flux_edit", lambda: extra_flux(dev))):
            try:
                extras[name] = fn()
            except Exception as e:                      # a failing extra must not take the headline line down with it
                extras[name] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0 and world == 1 and args.traffic == "live" and eff_batch == 32:
        live, src = live_traffic(headline_mode)
        if live is not None and sched:
            # the schedule's forwards are of two kinds: measure the one-plane forward too and quote the per-launch MEAN of the mix the timed loop ran
            live_lo, src_lo = live_traffic("f16")
            if live_lo is not None:
                roofline["traffic_by_stream_mode"] = {"f16x2": live, "f16": live_lo}
                live = (k_hi * live + (n - k_hi) * live_lo) / n
                src = src + f"; per-launch mean of the schedule's mix ({k_hi} split-stream + {n - k_hi} one-plane forwards per generation; the one-plane forward measured the same way)"
            else:
                src = src + f" (the split-stream forward only: the one-plane measurement failed: {src_lo})"
        if live is not None:
            roofline["traffic"], roofline["traffic_source"] = live, src
        else:
            roofline["traffic_source"] = f"{traffic_src} (committed; live measurement unavailable: {src})"
    if rank == 0:
        images = B * world * args.steps
        rec = {
            "metric": "images/sec at 8-step ConsistencySolver 512x512 (final latents; SD1.5 UNet fp16 + PPOScheduler, CFG 3)",
            "value": images / elapsed, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "residual_stream": (f"f16x2 for the first {k_hi} of {n} forwards of a generation, f16 for the rest (engine precision schedule)" if sched else headline_mode), "data": "synthetic (seeded weights, prompt embeddings and noise of the SD1.5 shapes)",
            "config": {"workload": "configs[1]: SD1.5 + PPOScheduler 8-step fp16, batch 16, 512x512 on 1 MI355X",
                       "batch_per_gpu": B, "num_inference_steps": n, "guidance_scale": args.guidance, "order_dim": 4,
                       "parallelism": f"dp{world} (prompt shards, no data-path collective)", "hipgraph": bool(args.graph)},
            "roofline": roofline,
            "residual_modes": modes,
        }
        if per_rank is not None:
            rec["per_rank"] = per_rank
            rec["per_rank_distinct_devices"] = distinct
        if decode_ms is not None:
            rec["pixel_images_per_s"] = images / pixel_elapsed          # latents + AutoencoderKL decode (row f-1), own timed loop
            rec["vae_decode"] = {"ms_per_batch": decode_ms, "tflops": vae.flops(B) / (decode_ms * 1e-3) / 1e12,
                                 "frac_of_mfma_peak": vae.flops(B) / (decode_ms * 1e-3) / 1e12 / PEAK_F16_TFLOPS}
        if kernels:
            rec["roofline_kernels"] = kernels
        if kernels_other:
            rec["roofline_kernels_" + other_mode] = kernels_other
        if roofline.get("traffic") and roofline.get("traffic_algorithmic_one_plane"):
            # measured HBM-side bytes of the headline mode over the reference graph's algorithmic bytes (one fp16 plane per tensor): lo planes count as excess
            roofline["traffic_ratio"] = roofline["traffic"] / roofline["traffic_algorithmic_one_plane"]
        if extras:
            rec.update({k: v for k, v in extras.items() if v is not None})
        rec["power_over_timed_region"] = power
        if ceilings:
            rec["ceilings"] = ceilings
            if ceilings.get("sustained_conv3_lw_64x64_320_tflops"):
                # the forward's algorithmic rate over what this box sustains on the product's own dominant kernel: comparable across boxes of the pool
                rec["roofline"]["achieved_over_sustained_conv3_lw"] = achieved / ceilings["sustained_conv3_lw_64x64_320_tflops"]
                rec["roofline"]["achieved_over_sustained_vendor_gemm"] = achieved / ceilings["sustained_vendor_gemm_f16_8192_tflops"]
            rec["roofline"]["frac_of_vendor_gemm"] = achieved / ceilings["vendor_gemm_f16_8192_tflops"]
        if not args.no_cpu_baseline and world == 1:
            rec["cpu_baseline"] = cpu_baseline(sd, unet.config, n, args.guidance, vae_sd=vae_sd)
            # "latent L2 vs ref" (the metric's second half): the HIP engine on the inputs of the CPU leg, against the fp32 latents it just produced
            replay = rec["cpu_baseline"].pop("_replay")
            try:
                par_modes = ([head_key] if sched else []) + [headline_mode, other_mode]
                par = parity_vs_oracle(unet, replay, args.guidance, dev, par_modes)
                par["latent_rel_l2"] = par[head_key]
                par["gate_met"] = bool(par[head_key] <= par["gate"])
                par["margin"] = 1.0 - par[head_key] / par["gate"]          # fraction of the gate left (negative: not met)
                rec["parity"] = par
            except Exception as e:
                rec["parity"] = {"error": f"{type(e).__name__}: {e}"}
        elif world == 1:
            rec["cpu_baseline"] = None
            rec["parity"] = None
        print(json.dumps(rec))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
