"""Micro-benchmark of the HIP ops on the SD1.5 UNet shapes (batch 32 = 16 images x CFG)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consolver_amd import ops

dev = "cuda:0"
def rnd(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).half()

def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

B = 32
rows = []
shapes = []      # per timed shape: tag + algorithmic bytes (operands read once, outputs written once); CS_SHAPES_JSON=path dumps it (tools/pmc_traffic.sh)
def lin(M, K, N, geglu=False, res=False, tag=""):
    x, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N)
    r = rnd(M, N) if res else None
    ms = timeit(lambda: ops.linear(x, w, b, res=r, geglu=geglu))
    fl = 2.0 * M * K * N
    rows.append((f"linear{'+geglu' if geglu else ''}{'+res' if res else ''} {tag}", M, K, N, ms, fl / ms / 1e9))
    shapes.append({"tag": rows[-1][0], "alg_bytes": 2.0 * (M * K + N * K + M * (N // 2 if geglu else N) * (2 if res else 1))})

def conv(H, cin, cout, taps=9, stride=1, up=False, tag=""):
    x = rnd(B, H, H, cin); k = 3 if taps == 9 else 1
    w = ops.pack_conv_weight(rnd(cout, cin, k, k, scale=(cin * taps) ** -0.5)); b = rnd(cout)
    ms = timeit(lambda: ops.conv2d(x, w, b, taps=taps, stride=stride, upsample=up))
    Ho = 2 * H if up else H // stride
    fl = 2.0 * B * Ho * Ho * taps * cin * cout
    rows.append((f"conv{k}x{k} s{stride}{' up' if up else ''} {tag}", B * Ho * Ho, taps * cin, cout, ms, fl / ms / 1e9))
    shapes.append({"tag": rows[-1][0], "alg_bytes": 2.0 * (B * H * H * cin + cout * taps * cin + B * Ho * Ho * cout)})

def attn(N, C, Nk=None, tag=""):
    Nk = Nk or N
    q, k, v = rnd(B, N, C), rnd(B, Nk, C), rnd(B, Nk, C)
    ms = timeit(lambda: ops.attention(q, k, v, 8))
    rows.append((f"attention dh={C // 8} {tag}", N, Nk, C, ms, 4.0 * B * N * Nk * C / ms / 1e9))
    shapes.append({"tag": rows[-1][0], "alg_bytes": 2.0 * B * (2 * N * C + 2 * Nk * C)})

which = sys.argv[1] if len(sys.argv) > 1 else "all"
for kv in os.environ.get("CS_TUNE", "").split(","):
    if "=" in kv:
        k, v = kv.split("="); ops.set_tuning(k, int(v))
if which in ("all", "gemm"):
    lin(8192, 8192, 8192, tag="square")
    lin(131072, 320, 960, tag="qkv L0"); lin(131072, 320, 320, res=True, tag="out L0"); lin(131072, 320, 2560, geglu=True, tag="ff1 L0")
    lin(131072, 1280, 320, res=True, tag="ff2 L0")
    lin(32768, 640, 1920, tag="qkv L1"); lin(32768, 640, 5120, geglu=True, tag="ff1 L1"); lin(32768, 2560, 640, res=True, tag="ff2 L1")
    lin(8192, 1280, 3840, tag="qkv L2"); lin(8192, 1280, 10240, geglu=True, tag="ff1 L2"); lin(8192, 5120, 1280, res=True, tag="ff2 L2")
    lin(2464, 768, 640, tag="cross kv")
    lin(8192, 1280, 1280, res=True, tag="out/proj L2"); lin(32768, 640, 640, res=True, tag="out/proj L1"); lin(2048, 1280, 1280, res=True, tag="out/proj L3")
if which in ("all", "conv"):
    conv(64, 320, 320, tag="L0"); conv(64, 640, 320, tag="L0 up"); conv(64, 960, 320, tag="L0 up")
    conv(32, 640, 640, tag="L1"); conv(32, 1280, 640, tag="L1 up"); conv(32, 1920, 640, tag="L1 up")
    conv(16, 1280, 1280, tag="L2"); conv(16, 2560, 1280, tag="L2 up")
    conv(8, 1280, 1280, tag="L3"); conv(8, 2560, 1280, tag="L3 up")
    conv(64, 320, 320, stride=2, tag="down"); conv(32, 640, 640, up=True, tag="upsample")
    conv(64, 320, 320, taps=1, tag="proj L0"); conv(16, 1280, 1280, taps=1, tag="proj L2")
if which in ("all", "attn"):
    attn(4096, 320, tag="self L0"); attn(1024, 640, tag="self L1"); attn(256, 1280, tag="self L2"); attn(64, 1280, tag="self L3")
    attn(4096, 320, 77, tag="cross L0")
    # FLUX shape (dh 128, 24 heads, 8704 tokens), f16 instance of the kernel the bf16 DiT uses
    qf, kf, vf = rnd(1, 8704, 3072), rnd(1, 8704, 3072), rnd(1, 8704, 3072)
    ms = timeit(lambda: ops.attention(qf, kf, vf, 24))
    rows.append(("attention dh=128 FLUX", 8704, 8704, 3072, ms, 4.0 * 8704 * 8704 * 3072 / ms / 1e9))
    shapes.append({"tag": rows[-1][0], "alg_bytes": 2.0 * 4 * 8704 * 3072})
if which in ("gemm2",):
    # FLUX-Kontext DiT linears (bf16, 8192 image + 512 text tokens, D = 3072) through the 256x256 transformer GEMM
    from consolver_amd import _lib as L
    def g2(M, K, N, tag, act=0, gated=False):
        x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
        res = torch.randn(M, N, device=dev).bfloat16() if gated else None; gate = torch.randn(1, N, device=dev) if gated else None
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16); st = L.stream_ptr(x.device)
        fn = lambda: L.check(L.lib().cs_op_gemm2(L.ptr(x), M, K, L.ptr(w), L.ptr(b), N, L.ptr(res), L.ptr(gate), N, M, act, L.ptr(out), N, 0, 2, st))
        ms = timeit(fn)
        rows.append((f"gemm2 bf16 {tag}", M, K, N, ms, 2.0 * M * K * N / ms / 1e9))
    g2(8192, 3072, 9216, "img qkv"); g2(512, 3072, 9216, "ctx qkv"); g2(8192, 3072, 3072, "img out", gated=True); g2(512, 3072, 3072, "ctx out", gated=True)
    g2(8192, 3072, 12288, "img ff1", act=1); g2(8192, 12288, 3072, "img ff2", gated=True); g2(512, 3072, 12288, "ctx ff1", act=1); g2(512, 12288, 3072, "ctx ff2", gated=True)
    g2(8704, 3072, 9216, "single qkv"); g2(8704, 3072, 12288, "single mlp", act=1); g2(8704, 15360, 3072, "single out", gated=True)
if which in ("gemm2f16",):
    # the FLUX transformer GEMM (256 x 256 tiles, software-pipelined k loop with two stages in flight) in fp16 on the UNet linear shapes whose N is a
    # multiple of 256, next to cs_op_linear (256 x 320 tiles): is its loop structure worth porting to the UNet GEMM?
    from consolver_amd import _lib as L
    for tag, M, K, N in [("square", 8192, 8192, 8192), ("qkv L2", 8192, 1280, 3840), ("ff2 L2", 8192, 5120, 1280), ("out L2", 8192, 1280, 1280),
                         ("ff1 L1 (no geglu)", 32768, 640, 5120), ("ff1 L0 (no geglu)", 131072, 320, 2560), ("ff1 L2 (no geglu)", 8192, 1280, 10240)]:
        x = rnd(M, K); w = rnd(N, K, scale=K ** -0.5); b = rnd(N)
        out = torch.empty(M, N, device=dev, dtype=torch.float16); st = L.stream_ptr(x.device)
        ms2 = timeit(lambda: L.check(L.lib().cs_op_gemm2(L.ptr(x), M, K, L.ptr(w), L.ptr(b), N, None, None, N, M, 0, L.ptr(out), N, 0, 1, st)))
        o2 = out.clone()
        ms1 = timeit(lambda: ops.linear(x, w, b, out=out))
        err = float((o2.float() - out.float()).norm() / out.float().norm())
        rows.append((f"gemm2 f16 {tag}", M, K, N, ms2, 2.0 * M * K * N / ms2 / 1e9))
        rows.append((f"linear    {tag} (diff {err:.1e})", M, K, N, ms1, 2.0 * M * K * N / ms1 / 1e9))
if which in ("xattn",):
    # fused cross-attention sub-block at the 64 x 64 level (batch 32) against the four kernels it replaces
    C, HW, Nk = 320, 4096, 77
    M = B * HW
    h = rnd(M, C); gam, bet = rnd(C), rnd(C); wq = rnd(C, C, scale=C ** -0.5); wo = rnd(C, C, scale=C ** -0.5); bo = rnd(C); kv = rnd(B, Nk, 2 * C)
    fl = 4.0 * M * C * C + 4.0 * M * Nk * C
    ms = timeit(lambda: ops.xattn_block(h, gam, bet, wq, kv, wo, bo, hw=HW))
    rows.append(("xattn_block fused L0", M, Nk, C, ms, fl / ms / 1e9))
    def unfused():
        ln = ops.layer_norm(h, gam, bet); q = ops.linear(ln, wq)
        a = ops.attention(q.view(B, HW, C), kv[..., :C], kv[..., C:], 8)
        return ops.linear(a.view(M, C), wo, bo, res=h)
    ms = timeit(unfused)
    rows.append(("LN + to_q + attn + to_out (4 kernels)", M, Nk, C, ms, fl / ms / 1e9))
if which in ("norm",):
    # GroupNorm (finalize + apply on producer statistics: the executor's common case) and LayerNorm on the UNet's shapes
    for H, c0, c1, tag in [(64, 320, 0, "L0"), (64, 640, 320, "L0 cat"), (32, 640, 0, "L1"), (32, 1280, 640, "L1 cat"), (16, 1280, 0, "L2"), (16, 1280, 1280, "L2 cat")]:
        x0 = rnd(B, H * H, c0); x1 = rnd(B, H * H, c1) if c1 else None
        C = c0 + c1
        g, b = rnd(C), rnd(C)
        s0 = torch.zeros(B, H * H // 64, c0 // 2, 2, device=dev); s1 = torch.zeros(B, H * H // 64, max(c1, 2) // 2, 2, device=dev) if c1 else None
        ms = timeit(lambda: ops.group_norm(x0, g, b, 32, 1e-5, True, x1=x1, stats0=s0, stats1=s1))
        rows.append((f"group_norm pre {tag}", B * H * H, C, 0, ms, 4.0 * B * H * H * C / ms / 1e9))
        shapes.append({"tag": rows[-1][0], "alg_bytes": 4.0 * B * H * H * C})
    for M, C, tag in [(131072, 320, "L0"), (32768, 640, "L1"), (8192, 1280, "L2")]:
        x, g, b = rnd(M, C), rnd(C), rnd(C)
        ms = timeit(lambda: ops.layer_norm(x, g, b))
        rows.append((f"layer_norm {tag}", M, C, 0, ms, 4.0 * M * C / ms / 1e9))
        shapes.append({"tag": rows[-1][0], "alg_bytes": 4.0 * M * C})
if os.environ.get("CS_SHAPES_JSON"):
    import json
    json.dump(shapes, open(os.environ["CS_SHAPES_JSON"], "w"), indent=1)
print(f"{'op':40s} {'M':>8s} {'K':>8s} {'N':>6s} {'ms':>9s} {'TFLOP/s':>9s}")
if which in ("upsub",):
    # the UNet's three upsamplers, batch 32: fused-upsample kernel and the sub-pixel form (tools/pmc_up_sub.sh reads the SQ counters of both)
    for H, C in ((8, 1280), (16, 1280), (32, 640)):
        x = rnd(B, H, H, C); w = ops.pack_conv_weight(rnd(C, C, 3, 3, scale=(9 * C) ** -0.5)); b = rnd(C)
        ws = ops.conv_up_fold_pack(w).to(dev)
        fl = 2.0 * B * 4 * H * H * 9 * C * C
        for name, fn in (("fused-upsample", lambda: ops.conv2d(x, w, b, upsample=True)), ("sub-pixel", lambda: ops.conv_up_sub(x, w, ws, b))):
            ms = timeit(fn)
            rows.append((f"upsampler {H}->{2 * H} {name}", B * 4 * H * H, 9 * C, C, ms, fl / ms / 1e9))
if which in ("convgn",):
    # cost of the GroupNorm-statistics epilogue: the same conv / 1x1 with and without gn_stats
    for H, cin, cout, taps, res, tag in [(64, 320, 320, 9, True, "L0 conv2"), (64, 960, 320, 9, False, "L0 up conv1"), (32, 640, 640, 9, True, "L1 conv2"),
                                         (16, 1280, 1280, 9, True, "L2 conv2"), (64, 320, 320, 1, True, "L0 proj_out"), (32, 640, 640, 1, True, "L1 proj_out")]:
        x = rnd(B, H, H, cin); k = 3 if taps == 9 else 1
        w = ops.pack_conv_weight(rnd(cout, cin, k, k, scale=(cin * taps) ** -0.5)); b = rnd(cout)
        r = rnd(B, H, H, cout) if res else None
        fl = 2.0 * B * H * H * taps * cin * cout
        ms0 = timeit(lambda: ops.conv2d(x, w, b, taps=taps, res=r))
        ms1 = timeit(lambda: ops.conv2d(x, w, b, taps=taps, res=r, gn_stats=True))
        rows.append((f"{tag} plain", B * H * H, taps * cin, cout, ms0, fl / ms0 / 1e9))
        rows.append((f"{tag} + gn_stats ({(ms1 / ms0 - 1) * 100:+.1f} %)", B * H * H, taps * cin, cout, ms1, fl / ms1 / 1e9))
if which in ("gn",):
    # GroupNorm + SiLU: statistics pass + finalize + apply ("full") against finalize + apply on producer-written partial sums ("pre")
    for H, c0, c1, tag in [(64, 320, 0, "L0"), (64, 320, 320, "L0 cat"), (64, 640, 320, "L0 cat"), (32, 640, 0, "L1"), (32, 1280, 640, "L1 cat"),
                           (16, 1280, 0, "L2"), (16, 1280, 1280, "L2 cat"), (8, 1280, 1280, "L3 cat")]:
        x0 = rnd(B, H * H, c0); x1 = rnd(B, H * H, c1) if c1 else None
        C = c0 + c1
        g, b = rnd(C), rnd(C)
        s0 = torch.zeros(B, H * H // 64, c0 // 2, 2, device=dev); s1 = torch.zeros(B, H * H // 64, max(c1, 2) // 2, 2, device=dev) if c1 else None
        byt = 2.0 * B * H * H * C
        ms = timeit(lambda: ops.group_norm(x0, g, b, 32, 1e-5, True, x1=x1))
        rows.append((f"group_norm full {tag}", B * H * H, C, 0, ms, 3 * byt / ms / 1e9))
        ms = timeit(lambda: ops.group_norm(x0, g, b, 32, 1e-5, True, x1=x1, stats0=s0, stats1=s1))
        rows.append((f"group_norm pre  {tag}", B * H * H, C, 0, ms, 2 * byt / ms / 1e9))
for r in rows:
    print(f"{r[0]:40s} {r[1]:8d} {r[2]:8d} {r[3]:6d} {r[4]:9.3f} {r[5] / 1e3:9.1f}")
