"""Generate tests/golden/t5_encoder.npz from the INSTALLED third-party ``transformers.T5EncoderModel`` (build container).

TEST INFRASTRUCTURE ONLY.  Reduced config (2 blocks, d_model 256, 4 heads x 64, d_ff 512, vocabulary 120, gated-gelu) with seeded
weights; the file holds the weights, token ids (a 64-token and a 200-token batch: beyond the 128-position bucket range) and the
resulting ``last_hidden_state`` -- inputs and outputs only.
"""
import os

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    from transformers import T5Config, T5EncoderModel
    cfg = T5Config(vocab_size=120, d_model=256, d_kv=64, num_heads=4, d_ff=512, num_layers=2, feed_forward_proj="gated-gelu",
                   relative_attention_num_buckets=32, relative_attention_max_distance=128, layer_norm_epsilon=1e-6, dropout_rate=0.0,
                   is_encoder_decoder=False, use_cache=False)
    m = T5EncoderModel(cfg).eval()
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if "layer_norm" in k:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif "relative_attention_bias" in k:
                p.copy_(torch.randn(p.shape, generator=g))
            elif "shared" in k or "embed_tokens" in k:
                p.copy_(torch.randn(p.shape, generator=g))
            elif k.endswith("SelfAttention.q.weight"):
                p.copy_(torch.randn(p.shape, generator=g) * (0.3 / p.shape[-1] ** 0.5))      # T5 has no 1/sqrt(d): keep the logits O(1)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / p.shape[-1] ** 0.5))
    fx = {"cfg": np.asarray([120, 256, 64, 4, 512, 2, 32, 128], np.int64)}
    seen = set()
    for k, v in m.state_dict().items():
        if k == "encoder.embed_tokens.weight" or not v.is_floating_point():
            continue                                  # tied to shared.weight
        fx["w_" + k] = v.numpy().copy()
        seen.add(k)
    for name, (B, L) in {"short": (2, 64), "long": (1, 200)}.items():
        ids = torch.randint(0, 120, (B, L), generator=g)
        with torch.no_grad():
            out = m(input_ids=ids)[0]
        fx[f"{name}_ids"] = ids.numpy()
        fx[f"{name}_out"] = out.numpy()
    np.savez_compressed(os.path.join(OUT, "t5_encoder.npz"), **fx)
    print("t5 fixtures written", sorted(seen)[:4], {k: v.shape for k, v in fx.items() if not k.startswith("w_")})


if __name__ == "__main__":
    main()
