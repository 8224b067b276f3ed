"""Generate tests/golden/clip_text.npz from the INSTALLED third-party ``transformers.CLIPTextModel`` (build container).

TEST INFRASTRUCTURE ONLY.  The model is instantiated from a reduced config (2 layers, width 128 = 2 heads of 64, MLP 256,
vocabulary 200, 77 positions, quick_gelu) with seeded weights; the file holds the weights, two batches of token ids
(full 77-token rows and a short 20-token batch) and the resulting ``last_hidden_state`` -- inputs and outputs only.
"""
import os
import sys

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    from transformers import CLIPTextConfig, CLIPTextModel
    torch.manual_seed(1234)
    cfg = CLIPTextConfig(vocab_size=200, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                         max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5, bos_token_id=198, eos_token_id=199, pad_token_id=199)
    m = CLIPTextModel(cfg).eval()
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 if "embedding" in k else 1.5 / p.shape[-1] ** 0.5))
            elif "layer_norm" in k and k.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    fx = {"cfg": np.asarray([200, 128, 256, 2, 2, 77], np.int64)}
    for k, v in m.state_dict().items():
        if v.is_floating_point():
            fx["w_" + (k[len("text_model."):] if k.startswith("text_model.") else k)] = v.numpy().copy()
    for name, (B, L) in {"full": (3, 77), "short": (2, 20)}.items():
        ids = torch.randint(0, 198, (B, L), generator=g)
        ids[:, 0] = 198
        ids[:, -1] = 199                                  # eos = highest id (pooled output position)
        if name == "full":
            ids[1, 40:] = 199                             # an early end-of-text followed by padding: pooled position 40
        with torch.no_grad():
            res = m(input_ids=ids)
        fx[f"{name}_ids"] = ids.numpy()
        fx[f"{name}_out"] = res[0].numpy()
        fx[f"{name}_pooled"] = res.pooler_output.numpy()
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "clip_text.npz"), **fx)
    print("clip fixtures written", {k: v.shape for k, v in fx.items() if not k.startswith("w_")})


if __name__ == "__main__":
    main()
