"""Per-launch A/B of the UNet's three upsamplers (nearest x2 + 3x3 conv, batch 32): the fused-upsample loader-wave kernel (36 multiplies per input pixel) against the
sub-pixel form on pre-summed taps (16), alternating on one box; then the one-plane UNet forward with up_fold 0 / 1, three alternations.
python tools/ab_up_sub.py > gpurun_out/r06_ab_up_sub.txt"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consolver_amd import ops
dev = "cuda:0"
def rnd(*s, scale=1.0, dt=torch.float16): return (torch.randn(*s, device=dev) * scale).to(dt)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
B = 32
for H, C in ((8, 1280), (16, 1280), (32, 640)):
    x = rnd(B, H, H, C); w = torch.randn(C, C, 3, 3) * (9 * C) ** -0.5; bias = rnd(C, scale=0.1)
    wp = ops.pack_conv_weight(w).to(dev); ws = ops.conv_up_fold_pack(wp).to(dev)
    gf = 2.0 * B * 4 * H * H * 9 * C * C / 1e9
    ts = []
    for _ in range(2):
        ts.append((timeit(lambda: ops.conv2d(x, wp, bias, upsample=True, gn_stats=True)), timeit(lambda: ops.conv_up_sub(x, wp, ws, bias, gn_stats=True))))
    o0, o1 = ops.conv2d(x, wp, bias, upsample=True), ops.conv_up_sub(x, wp, ws, bias)
    print(f"upsampler {H}x{H} -> {2 * H}x{2 * H}, {C} channels, batch {B} ({gf:.0f} GFLOP of the graph): fused-upsample kernel {ts[0][0]:7.1f} / {ts[1][0]:7.1f} us "
          f"({gf / ts[1][0] * 1e3:.0f} TFLOP/s) | sub-pixel {ts[0][1]:7.1f} / {ts[1][1]:7.1f} us ({gf / ts[1][1] * 1e3:.0f} TFLOP/s of the graph, {gf * 4 / 9 / ts[1][1] * 1e3:.0f} executed); "
          f"rel L2 between them {rel(o1, o0):.2e}", flush=True)
    del x, wp, ws
# the UNet forward on one fp16 plane
from consolver_amd.unet import HipUNet2DConditionModel                              # noqa: E402
from consolver_amd.synth import synthetic_unet_state_dict, synthetic_prompt_embeds  # noqa: E402
u = HipUNet2DConditionModel(device=dev, residual="f16"); u.load_state_dict(synthetic_unet_state_dict(u.manifest()))
lat = torch.randn(16, 4, 64, 64, device=dev).half(); ctx = synthetic_prompt_embeds(32).half().to(dev); t = torch.tensor([499.0], device=dev)
u(lat, t, encoder_hidden_states=ctx, dup=2)
def fwd(): return u(lat, t, encoder_hidden_states=ctx, dup=2, reuse_kv=True)[0]
outs = {}
for rnd_i in range(3):
    for k in (0, 1):
        ops.set_tuning("up_fold", k)
        outs[k] = fwd().clone()
        print(f"UNet forward, one fp16 plane, up_fold = {k}: {timeit(fwd, 10) / 1e3:.3f} ms", flush=True)
print(f"eps of the two forms: rel L2 {rel(outs[1], outs[0]):.3e}")
ops.set_tuning("up_fold", 1)
