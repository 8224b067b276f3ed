import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consolver_amd import ops
dev = "cuda:0"
def rnd(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).half()
which = sys.argv[1]
if which == "qkv":
    x, w = rnd(131072, 320), rnd(960, 320, scale=0.05)
    for _ in range(3): ops.linear(x, w)
elif which == "conv":
    x = rnd(32, 64, 64, 320); w = ops.pack_conv_weight(rnd(320, 320, 3, 3, scale=0.02)); b = rnd(320)
    for _ in range(3): ops.conv2d(x, w, b)
elif which == "attn":
    q, k, v = rnd(32, 4096, 320), rnd(32, 4096, 320), rnd(32, 4096, 320)
    for _ in range(3): ops.attention(q, k, v, 8)
torch.cuda.synchronize()
