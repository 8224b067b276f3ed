"""CPU restatement (plain torch ops, fp32) of the T5 v1.1 ENCODER that produces FLUX's ``prompt_embeds``
(edit_ppo/pipeline.py:279-330: ``text_encoder_2(text_input_ids)[0]``, 512 tokens x 4096).

TEST INFRASTRUCTURE ONLY.  **Pinned** against the installed third-party ``transformers.T5EncoderModel``:
``oracle/make_t5_golden.py`` instantiates it from a reduced config with seeded weights and writes weights, token ids and
``last_hidden_state`` to ``tests/golden/t5_encoder.npz``; ``tests/test_oracle_golden.py`` checks this file against it.

Architecture: shared token embedding (no position embedding, no scaling); per block ``h += O(softmax(Q K^T + bias) V)`` on
``RMSNorm(h)`` -- NO 1/sqrt(d) scaling, bias[h][i][j] = rel_bias_table[bucket(j - i)][h] (bidirectional log buckets, table of
block 0 shared by every block) -- then ``h += wo(gelu_new(wi_0 n) * wi_1 n)`` on ``RMSNorm(h)``; final RMSNorm.  No biases.
"""
import math

import torch
import torch.nn.functional as F

T5_XXL_CONFIG = dict(vocab_size=32128, d_model=4096, d_kv=64, num_heads=64, d_ff=10240, num_layers=24,
                     relative_attention_num_buckets=32, relative_attention_max_distance=128, layer_norm_epsilon=1e-6)


def t5_manifest(cfg):
    D, I, inner = cfg["d_model"], cfg["d_ff"], cfg["num_heads"] * cfg["d_kv"]
    out = [("shared.weight", (cfg["vocab_size"], D)),
           ("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", (cfg["relative_attention_num_buckets"], cfg["num_heads"]))]
    for l in range(cfg["num_layers"]):
        p = f"encoder.block.{l}"
        for q in ("q", "k", "v"):
            out.append((f"{p}.layer.0.SelfAttention.{q}.weight", (inner, D)))
        out += [(f"{p}.layer.0.SelfAttention.o.weight", (D, inner)), (f"{p}.layer.0.layer_norm.weight", (D,)),
                (f"{p}.layer.1.DenseReluDense.wi_0.weight", (I, D)), (f"{p}.layer.1.DenseReluDense.wi_1.weight", (I, D)),
                (f"{p}.layer.1.DenseReluDense.wo.weight", (D, I)), (f"{p}.layer.1.layer_norm.weight", (D,))]
    out.append(("encoder.final_layer_norm.weight", (D,)))
    return out


def relative_position_bucket(relative_position, num_buckets=32, max_distance=128):
    """bidirectional bucket of (key position - query position); integer tensor in, integer tensor out"""
    nb = num_buckets // 2
    ret = (relative_position > 0).long() * nb
    n = relative_position.abs()
    max_exact = nb // 2
    is_small = n < max_exact
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return ret + torch.where(is_small, n, large)


def position_bias(table, L, num_buckets=32, max_distance=128):
    """table [num_buckets, H] -> bias [H, L, L]"""
    ctx = torch.arange(L)[:, None]
    mem = torch.arange(L)[None, :]
    b = relative_position_bucket(mem - ctx, num_buckets, max_distance)
    return table[b].permute(2, 0, 1).contiguous()


def rms_norm(x, w, eps):
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3))))


class T5EncoderOracle:
    def __init__(self, sd, config=None, round_weights_to_bf16=True):
        self.cfg = dict(T5_XXL_CONFIG)
        self.cfg.update(config or {})
        self.sd = {k: (v.bfloat16().float() if round_weights_to_bf16 else v.float()) for k, v in sd.items() if v.is_floating_point()}

    @torch.no_grad()
    def __call__(self, input_ids):
        sd, c = self.sd, self.cfg
        B, L = input_ids.shape
        H, dk, eps = c["num_heads"], c["d_kv"], c["layer_norm_epsilon"]
        h = sd["shared.weight"][input_ids]
        bias = position_bias(sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], L,
                             c["relative_attention_num_buckets"], c["relative_attention_max_distance"])
        for l in range(c["num_layers"]):
            p = f"encoder.block.{l}"
            n = rms_norm(h, sd[f"{p}.layer.0.layer_norm.weight"], eps)
            q, k, v = (F.linear(n, sd[f"{p}.layer.0.SelfAttention.{t}.weight"]).view(B, L, H, dk).transpose(1, 2) for t in ("q", "k", "v"))
            a = torch.softmax(q @ k.transpose(-1, -2) + bias[None], dim=-1) @ v
            h = h + F.linear(a.transpose(1, 2).reshape(B, L, H * dk), sd[f"{p}.layer.0.SelfAttention.o.weight"])
            n = rms_norm(h, sd[f"{p}.layer.1.layer_norm.weight"], eps)
            ff = gelu_new(F.linear(n, sd[f"{p}.layer.1.DenseReluDense.wi_0.weight"])) * F.linear(n, sd[f"{p}.layer.1.DenseReluDense.wi_1.weight"])
            h = h + F.linear(ff, sd[f"{p}.layer.1.DenseReluDense.wo.weight"])
        return (rms_norm(h, sd["encoder.final_layer_norm.weight"], eps),)
