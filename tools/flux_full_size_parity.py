"""configs[3] at its REAL size against the oracle, once: the complete FLUX.1-Kontext DiT (19 double + 38 single blocks, 24 heads x 128, 11.9 B synthetic bf16
parameters) on 512 text + 4096 latent + 4096 image tokens = 8704 rows (edit_ppo/pipeline.py:1082-1097 at 1024 x 1024), one forward, against the STREAMED fp32
CPU oracle (FluxOracle(lazy=True): weights read through from the GPU one tensor at a time, attention one head at a time) and against the same restatement as a
plain torch-bf16 graph on the GPU (the reference pipeline's own arithmetic class).  ~165 TFLOP of CPU fp32: minutes.  A tool, not a test: the suite checks
depth at S = 576 and length at 1 + 1 blocks (tests/test_flux_gpu.py); this run's output is kept under profiles/.

    python tools/flux_full_size_parity.py > gpurun_out/r06_flux_full_size_parity.txt
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consolver_amd.flux import HipFluxTransformer2DModel, prepare_latent_image_ids      # noqa: E402
from oracle.flux_oracle import FluxOracle                                              # noqa: E402

DEV = "cuda:0"


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


def gpu_flux_weights(m, seed):
    """tests/test_flux_gpu.py::_gpu_flux_weights: seeded synthetic weights drawn on the GPU, kept in the model dtype for the oracle to read through"""
    g = torch.Generator(device=DEV).manual_seed(seed)
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
        sd[name] = w.to(m.dtype)
        m.set_weight(name, sd[name])
    m.finalize()
    return sd


def main():
    nl, ns = int(os.environ.get("FLUX_LAYERS", "19")), int(os.environ.get("FLUX_SINGLES", "38"))
    m = HipFluxTransformer2DModel(dict(num_layers=nl, num_single_layers=ns, dtype=torch.bfloat16), device=DEV)
    sd = gpu_flux_weights(m, seed=11)
    print(f"FLUX.1-Kontext DiT: {nl} double + {ns} single blocks, {sum(v.numel() for v in sd.values()):,} parameters (bf16, synthetic, seed 11)", flush=True)
    g = torch.Generator().manual_seed(5)
    B, T, Lq = 1, 512, 4096
    lat = torch.randn(B, Lq, 64, generator=g).to(torch.bfloat16)
    img = torch.randn(B, Lq, 64, generator=g).to(torch.bfloat16)
    enc = torch.nn.functional.layer_norm(torch.randn(B, T, 4096, generator=g), (4096,)).to(torch.bfloat16)
    pooled = torch.randn(B, 768, generator=g).to(torch.bfloat16)
    t = torch.tensor([0.9567]); guidance = torch.full((B,), 2.5)
    ids = np.concatenate([prepare_latent_image_ids(64, 64), prepare_latent_image_ids(64, 64, first=1.0)], 0)
    txt_ids = np.zeros((T, 3), np.float32)
    run = lambda: m(lat.to(DEV), t.to(DEV), guidance=guidance.to(DEV), pooled_projections=pooled.to(DEV), encoder_hidden_states=enc.to(DEV),
                    txt_ids=txt_ids, img_ids=ids, image_latents=img.to(DEV))[0].clone()
    got = run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); got2 = run(); b.record(); torch.cuda.synchronize()
    assert torch.equal(got, got2)
    print(f"HIP forward at S = {T + 2 * Lq}: {a.elapsed_time(b):.1f} ms, deterministic (two runs bit-identical), residual = {m.residual}", flush=True)
    got32 = m(lat.to(DEV), t.to(DEV), guidance=guidance.to(DEV), pooled_projections=pooled.to(DEV), encoder_hidden_states=enc.to(DEV),
              txt_ids=txt_ids, img_ids=ids, image_latents=img.to(DEV), out_dtype=torch.float32)[0].clone()
    m.set_residual_precision("plain")
    got_plain = run()
    m.set_residual_precision("split")
    t16 = FluxOracle(sd, m.config, lazy=True, device=DEV, dtype=torch.bfloat16)(torch.cat([lat, img], 1), t, guidance, pooled, enc, txt_ids, ids)[:, :Lq].float().cpu()
    torch.set_num_threads(min(int(os.environ.get("CS_ORACLE_THREADS", "32")), os.cpu_count() or 1))
    t0 = time.time()
    want = FluxOracle(sd, m.config, lazy=True)(torch.cat([lat, img], 1).float(), t, guidance, pooled.float(), enc.float(), txt_ids, ids)[:, :Lq]
    dt = time.time() - t0
    err, err_plain, e_t16 = rel_l2(got.float(), want), rel_l2(got_plain.float(), want), rel_l2(t16, want)
    err_tail = rel_l2(got[:, -512:].float(), want[:, -512:])
    print(f"fp32 CPU oracle (streamed weights, {torch.get_num_threads()} threads): {dt:.0f} s")
    print(f"relative L2 of the velocity [1, 4096, 64] vs the fp32 oracle at {nl} + {ns} blocks x S = {T + 2 * Lq}:")
    print(f"  HIP, split hidden-state stream (default): {err:.3e}   (last 512 latent rows: {err_tail:.3e})")
    print(f"  HIP, split stream, fp32 output:           {rel_l2(got32, want):.3e}   (cs_flux_set_output_dtype: the sum of the output head's two planes)")
    print(f"  HIP, one-plane stream:                    {err_plain:.3e}")
    print(f"  torch-bf16 graph of the same restatement: {e_t16:.3e}   (the reference pipeline's own arithmetic class)")
    assert torch.isfinite(got.float()).all() and got.shape == (B, Lq, 64)


if __name__ == "__main__":
    main()
