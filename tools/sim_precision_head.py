"""What the UNet's one-plane stations at its two ends cost the split-stream forward (CPU emulation on tools/sim_precision.py, the FLUX head finding applied to the UNet):
the final GroupNorm + SiLU output that conv_out reads (fp16 today), next to the floors of the other classes.    python tools/sim_precision_head.py [t ...]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from sim_precision import manifest_from_oracle_keys, rel_l2                      # noqa: E402
from sim_precision_r06 import Emu6, hi_lo8                                        # noqa: E402
from oracle.unet_oracle import UNetOracle                                         # noqa: E402
from consolver_amd.synth import synthetic_prompt_embeds, synthetic_unet_state_dict  # noqa: E402
from consolver_amd.unet import SD15_CONFIG                                        # noqa: E402

torch.set_num_threads(os.cpu_count())
cfg = dict(SD15_CONFIG)
sd = synthetic_unet_state_dict(manifest_from_oracle_keys(cfg), seed=7)
built = dict(stream=False, raw=True, norm=True, branch=True, raw_sc=False, ln_fold=True, h_store=hi_lo8)     # the f16x2 executor as shipped in round 6
for t in [int(v) for v in sys.argv[1:]] or [999, 499]:
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(1, 4, 64, 64, generator=g).half().float()
    ctx = synthetic_prompt_embeds(2, seed=13 + t).half().float()
    x2 = torch.cat([lat] * 2)
    t0 = time.time()
    want = UNetOracle(sd, cfg)(x2, t, ctx)
    print(f"t = {t}; oracle forward {time.time() - t0:.1f} s", flush=True)
    for name, kw in (("f16x2 executor as shipped", {}),
                     ("  conv_out reads the final GroupNorm + SiLU output unrounded (hi + lo operand)", dict(head=False)),
                     ("  ... and proj_out / down / upsample convs read hi + lo too (raw = exact)", dict(head=False, raw=False)),
                     ("  proj_out alone reads hi + lo (x2_split_a bit 1; needs fp16 lo planes of h: no lo8)", dict(raw_po=False, h_store=None)),
                     ("  down / upsample convs alone read hi + lo", dict(raw_ud=False)),
                     ("  every GroupNorm / LayerNorm output unrounded (floor of the norm class)", dict(norm=False, head=False)),
                     ("  every intra-branch tensor unrounded (floor of the branch class)", dict(branch=False))):
        e = Emu6(sd, cfg, **{**built, **kw})
        print(f"  {rel_l2(e(x2, t, ctx), want):.4e}  {name}", flush=True)
