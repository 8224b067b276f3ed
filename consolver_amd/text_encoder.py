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
