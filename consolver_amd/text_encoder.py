"""HIP-backed CLIP text encoder stand-in and the on-disk prompt-embedding cache.

``HipCLIPTextModel`` is call-compatible with what the reference does with the third-party model:
``prompt_embeds = text_encoder(input_ids)[0]`` (denoise_ppo.py:25-50; gen_pretrain/pipeline.py:402-517) ->
``[B, 77, 768]`` fp16.  Weights load by their ``transformers`` state-dict names (``text_model.`` prefix optional).

Tokenisation needs the CLIP vocabulary / merges files, which are assets, not code: ``tokenize`` wraps a
``transformers.CLIPTokenizer`` the caller constructs from those files; everything downstream takes token ids.

``save_prompt_cache`` / ``load_prompt_cache``: a safetensors file with ``prompt_embeds`` (and optionally
``negative_prompt_embeds``) ``[N, 77, 768]`` fp16 and the prompts in the JSON metadata, so batch generation
(`gen_ppo.py` path) can run on a box without tokenizer / text-encoder assets.
"""
import ctypes as C
import json

import torch

from . import _lib as L

CLIP_L_CONFIG = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                     max_position_embeddings=77, layer_norm_eps=1e-5)


class _ClipOutput(tuple):
    """tuple (last_hidden_state, pooler_output) with the transformers attribute names"""
    last_hidden_state = property(lambda self: self[0])
    pooler_output = property(lambda self: self[1])


class HipCLIPTextModel:
    is_consolver_hip = True
    dtype = torch.float16

    def __init__(self, config=None, device="cuda:0"):
        cfg = dict(CLIP_L_CONFIG)
        cfg.update(config or {})
        self.config = cfg
        self.device = torch.device(device)
        c = L.CsClipConfig(cfg["vocab_size"], cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"],
                           cfg["num_attention_heads"], cfg["max_position_embeddings"], cfg["layer_norm_eps"])
        h = C.c_void_p()
        L.check(L.lib().cs_clip_create(C.byref(c), C.byref(h)))
        self._h = h
        self._ws = None
        self._finalized = False

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.lib().cs_clip_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def manifest(self):
        lib = L.lib()
        out, shape, nd = [], (C.c_int64 * 4)(), C.c_int()
        for i in range(lib.cs_clip_num_weights(self._h)):
            name = lib.cs_clip_weight_name(self._h, i, shape, C.byref(nd)).decode()
            out.append((name, tuple(shape[k] for k in range(nd.value))))
        return out

    def load_state_dict(self, sd, strict=True):
        sd = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in sd.items()}
        lib = L.lib()
        want = dict(self.manifest())
        missing = [k for k in want if k not in sd]
        if missing:
            raise KeyError(f"missing {len(missing)} tensors, e.g. {missing[:3]}")
        for name, shape in want.items():
            t = sd[name].detach().to("cpu", torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise ValueError(f"{name}: shape {tuple(t.shape)} != {shape}")
            sh = (C.c_int64 * len(shape))(*shape)
            L.check(lib.cs_clip_set_weight(self._h, name.encode(), C.c_void_p(t.data_ptr()), sh, len(shape)))
        torch.cuda.set_device(self.device)
        L.check(lib.cs_clip_finalize(self._h))
        self._finalized = True
        return self

    def flops(self, batch, seq_len=77):
        return float(L.lib().cs_clip_flops(self._h, batch, seq_len))

    def __call__(self, input_ids, attention_mask=None, **_ignored):
        """-> (last_hidden_state [B, L, hidden] fp16, pooler_output [B, hidden] fp16): the SD path indexes [0]
        (denoise_ppo.py:31); the FLUX path reads ``.pooler_output`` of the same tower (edit_ppo/pipeline.py:330-345) = the final
        hidden state at the end-of-text position (highest token id of each row, as transformers does)."""
        if not self._finalized:
            raise RuntimeError("weights not loaded")
        L.require_cuda(input_ids, "input_ids")
        if attention_mask is not None:
            raise NotImplementedError("the reference never passes an attention mask for SD1.5 (use_attention_mask is absent from the CLIP-L config)")
        ids = input_ids.to(torch.int64).contiguous()
        B, Lq = ids.shape
        out = torch.empty(B, Lq, self.config["hidden_size"], dtype=torch.float16, device=ids.device)
        if B == 0 or Lq == 0:
            return _ClipOutput((out, out.new_empty(B, self.config["hidden_size"])))
        lib = L.lib()
        need = int(lib.cs_clip_workspace_bytes(self._h, B, Lq))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=ids.device)
        L.check(lib.cs_clip_encode(self._h, L.ptr(ids), B, Lq, L.ptr(out), L.ptr(self._ws), self._ws.numel(), L.stream_ptr(ids.device)))
        pooled = out[torch.arange(B, device=ids.device), ids.argmax(dim=-1)]
        return _ClipOutput((out, pooled))


def tokenize(tokenizer, prompts, device):
    """denoise_ppo.py:25-33: padding="max_length", truncation, max_length = tokenizer.model_max_length -> ids [B, 77] on ``device``."""
    return tokenizer(prompts, padding="max_length", max_length=tokenizer.model_max_length, truncation=True, return_tensors="pt").input_ids.to(device)


def encode_prompts(text_encoder, tokenizer, prompts, device, negative_prompt=""):
    """-> (prompt_embeds, negative_prompt_embeds), the two tensors the CFG dual batch is built from (denoise_ppo.py:25-50)."""
    pe = text_encoder(tokenize(tokenizer, prompts, device))[0]
    ne = text_encoder(tokenize(tokenizer, [negative_prompt] * len(prompts), device))[0]
    return pe, ne


def save_prompt_cache(path, prompts, prompt_embeds, negative_prompt_embeds=None):
    from safetensors.torch import save_file
    if prompt_embeds.shape[0] != len(prompts):
        raise ValueError("one embedding row per prompt")
    tensors = {"prompt_embeds": prompt_embeds.detach().to("cpu", torch.float16).contiguous()}
    if negative_prompt_embeds is not None:
        if negative_prompt_embeds.shape != prompt_embeds.shape:
            raise ValueError("negative_prompt_embeds must match prompt_embeds")
        tensors["negative_prompt_embeds"] = negative_prompt_embeds.detach().to("cpu", torch.float16).contiguous()
    save_file(tensors, path, metadata={"format": "consolver_amd.prompt_cache.v1", "prompts": json.dumps(list(prompts), ensure_ascii=False)})


def load_prompt_cache(path, device="cpu", start=None, end=None):
    """-> (prompts, prompt_embeds, negative_prompt_embeds or None); ``start:end`` selects one rank's shard without reading the rest."""
    from safetensors import safe_open
    with safe_open(path, framework="pt", device=str(device)) as f:
        meta = f.metadata() or {}
        if meta.get("format") != "consolver_amd.prompt_cache.v1":
            raise ValueError(f"{path}: not a prompt cache (format={meta.get('format')!r})")
        prompts = json.loads(meta["prompts"])
        sl = slice(start, end)
        pe = f.get_slice("prompt_embeds")[sl]
        ne = f.get_slice("negative_prompt_embeds")[sl] if "negative_prompt_embeds" in f.keys() else None
    return prompts[sl], pe, ne


# ---------------------------------------------------------------------------------------------------------------------
# T5 v1.1 encoder (FLUX ``text_encoder_2``; edit_ppo/pipeline.py:279-330 ``prompt_embeds = text_encoder_2(ids)[0]``)
# ---------------------------------------------------------------------------------------------------------------------
T5_XXL_CONFIG = dict(vocab_size=32128, d_model=4096, d_kv=64, num_heads=64, d_ff=10240, num_layers=24,
                     relative_attention_num_buckets=32, relative_attention_max_distance=128, layer_norm_epsilon=1e-6)


def _t5_bucket(rel, num_buckets, max_distance):
    import math
    nb = num_buckets // 2
    ret = (rel > 0).long() * nb
    n = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return ret + torch.where(n < max_exact, n, large)


class HipT5EncoderModel:
    """``text_encoder_2(input_ids)[0]`` -> [B, L, d_model].  Every matmul runs in the 256x256 LDS-staged MFMA GEMM (gemm2.hip), the
    attention in the flash kernel's biased head-64 form (relative-position bias [H, L, L] precomputed once per sequence length,
    no 1/sqrt(d) scaling), RMSNorm / gated GELU(tanh) / embedding gather as 16-byte-per-lane streaming kernels.  The layer
    sequence is driven from Python over the op-level C ABI (a once-per-prompt step, not part of the solver loop)."""
    is_consolver_hip = True

    def __init__(self, config=None, device="cuda:0", dtype=torch.bfloat16):
        cfg = dict(T5_XXL_CONFIG)
        cfg.update(config or {})
        if cfg["d_kv"] != 64:
            raise ValueError("the biased attention kernel is built for head dim 64")
        if cfg["d_model"] % 64 or cfg["d_ff"] % 64 or (cfg["num_heads"] * 64) % 64:
            raise ValueError("d_model and d_ff must be multiples of 64")
        self.config, self.device, self.dtype = cfg, torch.device(device), dtype
        self._dt = L.dtype_code(dtype)
        self.w = None
        self._bias = {}

    def manifest(self):
        c = self.config
        D, I, inner = c["d_model"], c["d_ff"], c["num_heads"] * c["d_kv"]
        out = [("shared.weight", (c["vocab_size"], D)),
               ("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", (c["relative_attention_num_buckets"], c["num_heads"]))]
        for l in range(c["num_layers"]):
            p = f"encoder.block.{l}"
            out += [(f"{p}.layer.0.SelfAttention.{q}.weight", (inner, D)) for q in ("q", "k", "v")]
            out += [(f"{p}.layer.0.SelfAttention.o.weight", (D, inner)), (f"{p}.layer.0.layer_norm.weight", (D,)),
                    (f"{p}.layer.1.DenseReluDense.wi_0.weight", (I, D)), (f"{p}.layer.1.DenseReluDense.wi_1.weight", (I, D)),
                    (f"{p}.layer.1.DenseReluDense.wo.weight", (D, I)), (f"{p}.layer.1.layer_norm.weight", (D,))]
        out.append(("encoder.final_layer_norm.weight", (D,)))
        return out

    @staticmethod
    def _pad_rows(t):
        """gemm2 weight panels are read in 256-row tiles: pad the row count up with zeros"""
        n = t.shape[0]
        pad = (-n) % 256
        return t if pad == 0 else torch.cat([t, t.new_zeros(pad, t.shape[1])])

    def load_state_dict(self, sd, strict=True):
        if "shared.weight" not in sd and "encoder.embed_tokens.weight" in sd:
            sd = dict(sd, **{"shared.weight": sd["encoder.embed_tokens.weight"]})
        want = dict(self.manifest())
        missing = [k for k in want if k not in sd]
        if missing:
            raise KeyError(f"missing {len(missing)} tensors, e.g. {missing[:3]}")
        for k, shape in want.items():
            if tuple(sd[k].shape) != shape:
                raise ValueError(f"{k}: shape {tuple(sd[k].shape)} != {shape}")
        dev, dt = self.device, self.dtype
        put = lambda t: t.detach().to(dev, dt).contiguous()
        w = {"shared": put(sd["shared.weight"]), "final_ln": put(sd["encoder.final_layer_norm.weight"]),
             "rel": sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].detach().float().to(dev), "blocks": []}
        for l in range(self.config["num_layers"]):
            p = f"encoder.block.{l}"
            qkv = torch.cat([sd[f"{p}.layer.0.SelfAttention.{q}.weight"] for q in ("q", "k", "v")])
            w["blocks"].append(dict(
                ln0=put(sd[f"{p}.layer.0.layer_norm.weight"]), qkv=self._pad_rows(put(qkv)), o=self._pad_rows(put(sd[f"{p}.layer.0.SelfAttention.o.weight"])),
                ln1=put(sd[f"{p}.layer.1.layer_norm.weight"]), wi0=self._pad_rows(put(sd[f"{p}.layer.1.DenseReluDense.wi_0.weight"])),
                wi1=self._pad_rows(put(sd[f"{p}.layer.1.DenseReluDense.wi_1.weight"])), wo=self._pad_rows(put(sd[f"{p}.layer.1.DenseReluDense.wo.weight"]))))
        self.w = w
        self._bias = {}
        return self

    def _position_bias(self, Lq):
        """[H, L, L] fp32, already times log2(e) (the softmax runs in base 2); cached per sequence length"""
        if Lq not in self._bias:
            c = self.config
            ctx = torch.arange(Lq, device=self.device)[:, None]
            mem = torch.arange(Lq, device=self.device)[None, :]
            b = _t5_bucket(mem - ctx, c["relative_attention_num_buckets"], c["relative_attention_max_distance"])
            self._bias[Lq] = (self.w["rel"][b].permute(2, 0, 1) * 1.4426950408889634).contiguous()
        return self._bias[Lq]

    def _gemm(self, x, w, N, out=None, res=None, act=0):
        M, K = x.shape
        if out is None:
            out = torch.empty(M, N, dtype=self.dtype, device=x.device)
        L.check(L.lib().cs_op_gemm2(L.ptr(x), M, K, L.ptr(w), L.ptr(None), N, L.ptr(res), L.ptr(None), 0, 0, act, L.ptr(out), N, 0, self._dt,
                                    L.stream_ptr(x.device)))
        return out

    def flops(self, batch, seq_len):
        c = self.config
        D, I, inner = c["d_model"], c["d_ff"], c["num_heads"] * c["d_kv"]
        rows = batch * seq_len
        return c["num_layers"] * (2.0 * rows * D * (4 * inner + 3 * I) + 4.0 * batch * seq_len * seq_len * inner)

    @torch.no_grad()
    def __call__(self, input_ids, attention_mask=None, **_ignored):
        if self.w is None:
            raise RuntimeError("weights not loaded")
        if attention_mask is not None:
            raise NotImplementedError("the FLUX pipeline calls text_encoder_2 without an attention mask (edit_ppo/pipeline.py:322)")
        L.require_cuda(input_ids, "input_ids")
        ids = input_ids.to(torch.int64).contiguous()
        B, Lq = ids.shape
        c, w, lib, st = self.config, self.w, L.lib(), L.stream_ptr(ids.device)
        D, I, H, eps = c["d_model"], c["d_ff"], c["num_heads"], c["layer_norm_epsilon"]
        inner, M = H * 64, B * Lq
        if Lq % 4:
            raise ValueError("sequence length must be a multiple of 4")
        h = torch.empty(M, D, dtype=self.dtype, device=ids.device)
        if M == 0:
            return (h.view(B, Lq, D),)
        L.check(lib.cs_op_embed_rows(L.ptr(ids), L.ptr(w["shared"]), L.ptr(h), M, D, c["vocab_size"], st))
        bias = self._position_bias(Lq)
        n = torch.empty_like(h)
        qkv = torch.empty(M, 3 * inner, dtype=self.dtype, device=ids.device)
        att = torch.empty(M, inner, dtype=self.dtype, device=ids.device)
        g = torch.empty(M, I, dtype=self.dtype, device=ids.device)
        u = torch.empty_like(g)
        for blk in w["blocks"]:
            L.check(lib.cs_op_rms_norm(L.ptr(h), L.ptr(blk["ln0"]), L.ptr(n), M, D, eps, self._dt, st))
            self._gemm(n, blk["qkv"], 3 * inner, out=qkv)
            L.check(lib.cs_op_attention_bias(L.ptr(qkv), 3 * inner, C.c_void_p(qkv.data_ptr() + 2 * inner), 3 * inner,
                                             C.c_void_p(qkv.data_ptr() + 4 * inner), 3 * inner, L.ptr(att), inner, B, H, Lq, 64, 1.0,
                                             L.ptr(bias), self._dt, st))
            self._gemm(att, blk["o"], D, out=h, res=h)
            L.check(lib.cs_op_rms_norm(L.ptr(h), L.ptr(blk["ln1"]), L.ptr(n), M, D, eps, self._dt, st))
            self._gemm(n, blk["wi0"], I, out=g, act=1)                       # gelu_new == GELU(tanh)
            self._gemm(n, blk["wi1"], I, out=u)
            L.check(lib.cs_op_gated_mul(L.ptr(g), L.ptr(u), L.ptr(g), M * I, self._dt, st))
            self._gemm(g, blk["wo"], D, out=h, res=h)
        out = torch.empty_like(h)
        L.check(lib.cs_op_rms_norm(L.ptr(h), L.ptr(w["final_ln"]), L.ptr(out), M, D, eps, self._dt, st))
        return (out.view(B, Lq, D),)
