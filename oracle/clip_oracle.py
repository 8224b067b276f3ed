"""CPU restatement (plain torch ops, fp32) of the CLIP text encoder the SD1.5 path calls as
``text_encoder(input_ids)[0]`` (denoise_ppo.py:25-50, gen_pretrain/pipeline.py:402-517).

TEST INFRASTRUCTURE ONLY.

The arithmetic lives in the third-party package ``transformers`` (``CLIPTextModel``), absent from /root/reference but
INSTALLED in the build image: this restatement is **pinned** against it -- ``oracle/make_clip_golden.py`` instantiates
``transformers.CLIPTextModel`` from a reduced config with seeded weights, runs it and writes weights, token ids and
``last_hidden_state`` to ``tests/golden/clip_text.npz``; ``tests/test_oracle_golden.py`` checks this file against it.
Architecture: token + learned position embeddings; pre-LN layers ``x += out_proj(attn(LN1 x))``, ``x += fc2(quick_gelu(fc1(LN2 x)))``
with a causal mask and ``q`` scaled by ``head_dim ** -0.5``; ``final_layer_norm``.  State-dict names as in transformers
(an optional ``text_model.`` prefix is stripped).
"""
import torch
import torch.nn.functional as F

CLIP_L_CONFIG = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                     max_position_embeddings=77, layer_norm_eps=1e-5)


def clip_manifest(cfg):
    D, I = cfg["hidden_size"], cfg["intermediate_size"]
    out = [("embeddings.token_embedding.weight", (cfg["vocab_size"], D)), ("embeddings.position_embedding.weight", (cfg["max_position_embeddings"], D))]
    for l in range(cfg["num_hidden_layers"]):
        p = f"encoder.layers.{l}"
        for q in (".self_attn.k_proj", ".self_attn.v_proj", ".self_attn.q_proj", ".self_attn.out_proj"):
            out += [(p + q + ".weight", (D, D)), (p + q + ".bias", (D,))]
        out += [(p + ".layer_norm1.weight", (D,)), (p + ".layer_norm1.bias", (D,)), (p + ".mlp.fc1.weight", (I, D)), (p + ".mlp.fc1.bias", (I,)),
                (p + ".mlp.fc2.weight", (D, I)), (p + ".mlp.fc2.bias", (D,)), (p + ".layer_norm2.weight", (D,)), (p + ".layer_norm2.bias", (D,))]
    out += [("final_layer_norm.weight", (D,)), ("final_layer_norm.bias", (D,))]
    return out


def strip_prefix(sd):
    return {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in sd.items()}


class ClipTextOracle:
    def __init__(self, sd, config=None, round_weights_to_f16=True):
        self.cfg = dict(CLIP_L_CONFIG)
        self.cfg.update(config or {})
        self.sd = {k: (v.half().float() if round_weights_to_f16 else v.float()) for k, v in strip_prefix(sd).items() if v.is_floating_point()}

    @torch.no_grad()
    def __call__(self, input_ids):
        sd, c = self.sd, self.cfg
        B, L = input_ids.shape
        H, D = c["num_attention_heads"], c["hidden_size"]
        dh = D // H
        x = sd["embeddings.token_embedding.weight"][input_ids] + sd["embeddings.position_embedding.weight"][:L][None]
        mask = torch.full((L, L), float("-inf")).triu(1)
        for l in range(c["num_hidden_layers"]):
            p = f"encoder.layers.{l}"
            n = F.layer_norm(x, (D,), sd[p + ".layer_norm1.weight"], sd[p + ".layer_norm1.bias"], c["layer_norm_eps"])
            q = F.linear(n, sd[p + ".self_attn.q_proj.weight"], sd[p + ".self_attn.q_proj.bias"]) * dh ** -0.5
            k = F.linear(n, sd[p + ".self_attn.k_proj.weight"], sd[p + ".self_attn.k_proj.bias"])
            v = F.linear(n, sd[p + ".self_attn.v_proj.weight"], sd[p + ".self_attn.v_proj.bias"])
            q, k, v = (t.view(B, L, H, dh).transpose(1, 2) for t in (q, k, v))
            a = torch.softmax(q @ k.transpose(-1, -2) + mask, dim=-1) @ v
            a = a.transpose(1, 2).reshape(B, L, D)
            x = x + F.linear(a, sd[p + ".self_attn.out_proj.weight"], sd[p + ".self_attn.out_proj.bias"])
            n = F.layer_norm(x, (D,), sd[p + ".layer_norm2.weight"], sd[p + ".layer_norm2.bias"], c["layer_norm_eps"])
            h = F.linear(n, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"])
            h = h * torch.sigmoid(1.702 * h)
            x = x + F.linear(h, sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])
        return (F.layer_norm(x, (D,), sd["final_layer_norm.weight"], sd["final_layer_norm.bias"], c["layer_norm_eps"]),)
