"""conv3_lw_kernel at BN = 128 (VAE widths 128 / 256 / 512; cs_set_tuning("conv_lw", 1); conv_lw = 3 keeps BN 160 only, so these widths fall back to the halo kernels) against the 8-wave halo kernels: equality of the results and time per layer."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from consolver_amd import _lib as L, ops
lib = L.lib(); dev = torch.device("cuda:0")
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for tag, B, H, cin, cout, up in [("512->512 @64", 16, 64, 512, 512, False), ("512->512 @128", 16, 128, 512, 512, False), ("512->512 up @64->128", 16, 64, 512, 512, True),
                                 ("512->256 @256", 8, 256, 512, 256, False), ("256->256 @256", 8, 256, 256, 256, False), ("256->128 @512", 4, 512, 256, 128, False),
                                 ("128->128 @512", 4, 512, 128, 128, False), ("256->256 up @256->512", 4, 256, 256, 256, True)]:
    x = torch.randn(B, H, H, cin, device=dev).half()
    w = ops.pack_conv_weight((torch.randn(cout, cin, 3, 3, device=dev) * (9 * cin) ** -0.5).half())
    b = torch.randn(cout, device=dev).half()
    outs, ms = {}, {}
    for v in (3, 1):      # 3: BN 160 only -> the VAE widths run the 8-wave halo kernels; 1: loader-wave kernel at BN 128 too
        lib.cs_set_tuning(b"conv_lw", v)
        outs[v] = ops.conv2d(x, w, b, upsample=up, splitk=False)
        ms[v] = t(lambda: ops.conv2d(x, w, b, upsample=up, splitk=False))
    lib.cs_set_tuning(b"conv_lw", 1)
    Ho = 2 * H if up else H
    fl = 2.0 * B * Ho * Ho * 9 * cin * cout
    d = (outs[3].float() - outs[1].float()).abs().max().item()
    print(f"{tag:26s} halo kernels {ms[3]:.3f} ms ({fl / ms[3] / 1e9:.0f} TFLOP/s) | loader-wave BN 128 {ms[1]:.3f} ms ({fl / ms[1] / 1e9:.0f} TFLOP/s) | equal {torch.equal(outs[3], outs[1])} max abs diff {d:.2e}")
