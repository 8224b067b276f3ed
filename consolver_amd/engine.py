"""Device-resident ConsistencySolver sampling loop for SD1.5 (the body of the hot path).

Equivalent of the loop ``StableDiffusionPipeline.__call__`` runs around ``PPOScheduler``
(in-tree copy: gen_pretrain/pipeline.py:1045-1098) and of ``gen_ppo.generate_batch_images``
(gen_ppo.py:237-330) minus prompt encoding / VAE / PNG I/O:

    for t in timesteps:  eps = unet(cat([x]*2), t, ctx)          # CFG dual batch
                         eps = u + g (c - u)
                         x   = scheduler.step(eps, t, x)[0]

MI355X-first choices: the dual batch is never materialised (conv_in reads latent b % B), the CFG
combine is fused into the solver kernel, eps history lives in a pre-allocated ring that the
solver kernel writes directly, latents ping-pong between two buffers, cross-attention K/V of the
prompt are computed once per batch, and nothing in the loop synchronises with the host -- so a
whole N-step generation can also be captured into one hipGraph (``use_graph=True``).

Round 5: the SOLVER STATE (the latents between steps) is fp32 by default (``latents_dtype``): the update kernel reads the fp32 sample next
to the fp16 eps tensors (CsStepArgs::x_is_f32) and writes fp32; the UNet reads an fp16 copy.  An fp16 state rounds the latents once per step
(2.8e-4 relative L2 each, adding in quadrature): the 8-step latents sat at 0.89e-3 of the fp32 oracle and the 12-step ones at 1.10e-3, above
north_star's 1e-3 gate (tests/test_parity_e2e_gpu.py).  64 KiB instead of 32 KiB per image and step through a 5 us kernel.
"""
import torch

from . import _lib as L


class SDSamplingEngine:
    def __init__(self, unet, scheduler, guidance_scale=3.0, vae=None, latents_dtype=torch.float32, eps_dtype=None, hi_precision_steps="auto"):
        if latents_dtype not in (torch.float32, torch.float16):
            raise ValueError("latents_dtype must be torch.float32 (default) or torch.float16")
        # Round 6: the denoiser's OUTPUT is taken in fp32 too when the state is fp32 (cs_unet_set_output_dtype: conv_out stores its accumulator unrounded) -- the eps
        # tensors, the CFG combine and the history ring then carry no fp16 rounding (2.8e-4 relative each for eps and for the combined eps).  eps_dtype=torch.float16
        # is the reference pipeline's own class (the UNet returns the model dtype).
        if eps_dtype is None:
            eps_dtype = torch.float32 if latents_dtype == torch.float32 else torch.float16
        if eps_dtype not in (torch.float32, torch.float16) or (eps_dtype == torch.float32 and latents_dtype != torch.float32):
            raise ValueError("eps_dtype must be torch.float16, or torch.float32 together with an fp32 solver state")
        self.latents_dtype = latents_dtype
        self.eps_dtype = eps_dtype
        # Precision SCHEDULE of the denoiser's residual stream (round 6).  The split (hi + lo) stream exists to meet north_star's 1e-3 latent gate, and the gate's budget
        # is spent in the first steps of a trajectory: the update x' = c1 x - c2 eps has c2 ~ 1 at t = 999 and ~ 0.1 later, and the multistep combination re-uses the early
        # eps.  Measured on the full UNet (tools/parity_schedule.py, profiles/r06_parity_schedule.txt: per-step latent drift against the fp32 oracle, first k forwards with
        # the split stream, the rest with ONE fp16 plane -- the reference pipeline's own arithmetic class):
        #     n = 8:   k = 8 (all) 0.732e-3 | k = 4 0.734 | k = 3 0.760 | k = 2 0.834 | k = 1 1.35 (gate missed) | k = 0 1.48
        #     n = 12:  all 0.801 | k = 4 0.802 | k = 3 0.838 | k = 1 0.859;   n = 15:  all 0.518 | k = 5.. 0.53 | k = 3 0.768;   n = 4:  all 0.924 (step 0) | k = 2 0.924
        # With the round's last kernels (fp32 eps, the output head on two planes, sub-pixel upsamplers in the one-plane forwards; profiles/r06_parity_schedule_final.txt):
        #     n = 4:   all 0.882 (step 0) | k = 1 0.884;   n = 8:  all 0.676 | k = 3 0.704 | k = 2 0.795 | k = 1 misses the gate;   n = 12:  all 0.767 | k = 4 0.770 | k = 3 0.806;
        #     n = 15:  all 0.492 | k = 4 0.504;   n = 5 / 6:  k = 2 is the all-split maximum (0.816 / 0.767 at step 0)
        # "auto" (default): the first ceil(n / 4) forwards run the handle's mode (f16x2), the rest `f16` -- 1 of 4, 2 of 8, 3 of 12, 4 of 15: the tightest numbers (short
        # trajectories, step 0) are the all-split ones, every longer trajectory keeps >= 19 % of air (asserted at every step: tests/test_parity_e2e_gpu.py), and 6 of 8
        # forwards of configs[1] run on one plane (-3.0 ms each: no lo planes, sub-pixel upsamplers).  Until the two-plane head the rule was ceil(n / 4) + 1 (3 of 8: 0.757e-3).
        # An int fixes k; None / "all" runs every step in the handle's mode.  Inactive on an `f16` handle.
        if not (hi_precision_steps is None or hi_precision_steps in ("auto", "all") or (isinstance(hi_precision_steps, int) and hi_precision_steps >= 0)):
            raise ValueError("hi_precision_steps must be 'auto', 'all' / None, or a non-negative int")
        self.hi_precision_steps = hi_precision_steps
        self._base_mode = None
        self._n_steps = None
        self.unet = unet
        self.vae = vae                  # HipAutoencoderKL for output_type="pt" (decode_latents, utils.py:6-34)
        self.decode_events = None       # optional list collecting (start, stop) events around the VAE decode
        self.scheduler = scheduler
        self.guidance_scale = float(guidance_scale)
        self._bufs = None
        self._graph = None
        self._graph_key = None
        self.forward_events = None      # optional list collecting (start, stop) events around UNet forwards

    def _buffers(self, B, shape, device):
        key = (B, tuple(shape), str(device), self.latents_dtype, self.eps_dtype)
        if self._bufs is None or self._bufs["key"] != key:
            order = self.scheduler.config.order_dim
            C, H, W = shape
            self._graph = self._graph_key = None      # a captured graph holds the OLD buffers' addresses (latents_dtype is a public attribute and part of the key)
            self._bufs = dict(
                key=key,
                lat=[torch.empty(B, C, H, W, dtype=self.latents_dtype, device=device) for _ in range(2)],
                lat16=torch.empty(B, C, H, W, dtype=torch.float16, device=device) if self.latents_dtype != torch.float16 else None,
                ring=[torch.empty(B, C, H, W, dtype=self.eps_dtype, device=device) for _ in range(order)],
                eps=torch.empty(2 * B, C, H, W, dtype=self.eps_dtype, device=device))
        return self._bufs

    def _stream_mode(self, i):
        """residual-stream mode of step i: the handle's mode for the first `hi_precision_steps` steps, "f16" afterwards (only when the handle runs f16x2)"""
        k = self.hi_steps(self._n_steps)
        if k is None or self._base_mode != "f16x2":
            return self._base_mode
        return "f16x2" if i < k else "f16"

    def hi_steps(self, n):
        """number of leading forwards of an n-step generation that run the split stream (None: all of them)"""
        h = self.hi_precision_steps
        if h is None or h == "all" or n is None:
            return None
        return min(n, max(1, -(-n // 4))) if h == "auto" else min(n, int(h))

    def _loop(self, ctx, bufs, n, B, do_cfg):
        sch, unet = self.scheduler, self.unet
        self._base_mode = getattr(unet, "residual", None)
        self._n_steps = n
        try:
            return self._loop_steps(ctx, bufs, n, B, do_cfg)
        finally:
            if self._base_mode is not None and getattr(unet, "residual", None) != self._base_mode:
                unet.set_residual_precision_keep(self._base_mode)

    def _loop_steps(self, ctx, bufs, n, B, do_cfg):
        sch, unet = self.scheduler, self.unet
        x = bufs["lat"][0]
        cur = 0
        t_dev = self._t_dev
        for i in range(n):
            if self.forward_events is not None:
                a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
                a.record()
            # the denoiser's fp16 view of the fp32 solver state: one cast for the initial noise, afterwards written by the update kernel itself (step(out_lp=...))
            xin = x if bufs["lat16"] is None else (bufs["lat16"].copy_(x) if i == 0 else bufs["lat16"])
            eps = unet(xin, t_dev[i:i + 1], encoder_hidden_states=ctx, dup=2 if do_cfg else 1, reuse_kv=(i > 0),
                       out=bufs["eps"] if do_cfg else bufs["eps"][:B], **({"residual": self._stream_mode(i)} if self._base_mode is not None else {}))[0]
            if self.forward_events is not None:
                b.record()
                self.forward_events.append((a, b))
            nxt = bufs["lat"][cur ^ 1]
            if do_cfg:
                sch.step(eps[B:], sch.timesteps[i], x, return_dict=False, eps_uncond=eps[:B],
                         guidance_scale=self.guidance_scale, eps_out=bufs["ring"][i % len(bufs["ring"])], out=nxt, out_lp=bufs["lat16"])
            else:
                # the history keeps a reference to the model output: copy it out of the reused buffer
                slot = bufs["ring"][i % len(bufs["ring"])]
                slot.copy_(eps)
                sch.step(slot, sch.timesteps[i], x, return_dict=False, out=nxt, out_lp=bufs["lat16"])
            x = nxt
            cur ^= 1
        return x

    @torch.no_grad()
    def generate(self, prompt_embeds, negative_prompt_embeds=None, latents=None, num_inference_steps=8, generator=None,
                 use_graph=False, output_type="latent", decode_batch_size=None):
        """prompt_embeds [B,77,768]; latents [B,4,64,64] initial noise (already scaled by
        init_noise_sigma = 1).  output_type="latent" returns the final latents [B,4,H,W] in ``latents_dtype`` (fp32 by default; a view of an
        internal buffer that the next call overwrites -- clone to keep); output_type="pt" returns the decoded
        images [B,3,8H,8W] fp16 in [0, 1] (decode_latents, utils.py:6-34; needs ``vae``)."""
        if output_type not in ("latent", "pt"):
            raise ValueError("output_type must be 'latent' or 'pt'")
        if output_type == "pt" and self.vae is None:
            raise RuntimeError("output_type='pt' needs a HIP AutoencoderKL (SDSamplingEngine(..., vae=...))")
        lat = self._generate_latents(prompt_embeds, negative_prompt_embeds, latents, num_inference_steps, generator, use_graph)
        if output_type == "latent":
            return lat
        from .vae import decode_latents
        if self.decode_events is not None:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        img = decode_latents(self.vae, lat.to(torch.float16), decode_batch_size or lat.shape[0])
        if self.decode_events is not None:
            b.record()
            self.decode_events.append((a, b))
        return img

    def _generate_latents(self, prompt_embeds, negative_prompt_embeds, latents, num_inference_steps, generator, use_graph):
        L.require_cuda(prompt_embeds, "prompt_embeds")
        dev = prompt_embeds.device
        B = prompt_embeds.shape[0]
        do_cfg = self.guidance_scale > 1.0            # denoise_ppo.py:37
        if do_cfg:
            if negative_prompt_embeds is None:
                raise ValueError("negative_prompt_embeds required when guidance_scale > 1")
            ctx = torch.cat([negative_prompt_embeds, prompt_embeds]).to(torch.float16).contiguous()
        else:
            ctx = prompt_embeds.to(torch.float16).contiguous()
        S = self.unet.config["sample_size"]
        shape = (self.unet.config["in_channels"], S, S)
        bufs = self._buffers(B, shape, dev)
        if latents is None:
            latents = torch.randn((B,) + shape, generator=generator, device=dev, dtype=torch.float16)
        bufs["lat"][0].copy_(latents.to(torch.float16) * self.scheduler.init_noise_sigma)        # (the initial noise is an fp16 tensor in the reference's pipeline)
        n = num_inference_steps

        if not use_graph:
            self.scheduler.set_timesteps(n, device=dev)
            self._t_dev = self.scheduler.timesteps.to(torch.float32)
            out = self._loop(ctx, bufs, n, B, do_cfg)
            self.scheduler.verify_timesteps()     # (no device read unless a step resolved its timestep through the host counter)
            return out

        # ---- whole-generation hipGraph: capture once per (B, n, cfg), replay afterwards ------------------
        key = (B, n, do_cfg, self.guidance_scale, self.hi_precision_steps, getattr(self.unet, "residual", None))
        if self._graph is None or self._graph_key != key:
            net = self.scheduler.factor_net
            if getattr(net, "sampler", None) == "multinomial" and net.forced_action_idx is None:
                raise RuntimeError("use_graph=True needs a graph-capturable sampler: factor_net.sampler = 'inverse_cdf' (the default)")
            self._ctx_static = torch.empty_like(ctx)
            self._ctx_static.copy_(ctx)
            self.scheduler.set_timesteps(n, device=dev)
            self._t_dev = self.scheduler.timesteps.to(torch.float32)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):             # warm-up outside capture (lazy inits, func attributes)
                self._loop(self._ctx_static, bufs, n, B, do_cfg)
                self.scheduler.set_timesteps(n, device=dev)
            torch.cuda.current_stream(dev).wait_stream(side)
            bufs["lat"][0].copy_(latents.to(torch.float16) * self.scheduler.init_noise_sigma)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._graph_out = self._loop(self._ctx_static, bufs, n, B, do_cfg)
            self._graph, self._graph_key = g, key
            bufs["lat"][0].copy_(latents.to(torch.float16) * self.scheduler.init_noise_sigma)
        else:
            self._ctx_static.copy_(ctx)
        self._graph.replay()
        return self._graph_out
