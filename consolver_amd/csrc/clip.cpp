// CLIP text encoder (the prompt front-end of the SD1.5 path).
//
// Replaces `text_encoder(input_ids)[0]` (denoise_ppo.py:25-50, gen_pretrain/pipeline.py:402-517): third-party
// transformers CLIPTextModel (CLIP ViT-L/14 text tower for SD1.5: 12 pre-LN layers, width 768, 12 heads of 64, MLP 3072
// with quick_gelu, causal mask, learned positions, final LayerNorm).  Tokens are [rows = B * 77] x width fp16;
// every linear runs through the implicit-GEMM MFMA kernels, attention through the flash kernel's causal head-64 form.
// It runs once per prompt batch (6.6 GMAC per prompt), i.e. < 0.1 % of an 8-step generation.
#include "ops.h"
#include "consolver_hip.h"

#include <map>
#include <string>
#include <vector>
#include <algorithm>
#include <cmath>

namespace {
struct HostT { std::vector<int64_t> shape; std::vector<f16> data; };
struct Layer { f16 *ln1g, *ln1b, *wqkv, *bqkv, *wo, *bo, *ln2g, *ln2b, *w1, *b1, *w2, *b2; };
}

struct CsClip {
    CsClipConfig cfg;
    std::vector<std::string> names;
    std::map<std::string, std::vector<int64_t>> expect;
    std::map<std::string, HostT> host;
    std::vector<void*> dev_allocs;
    bool finalized = false;
    f16 *tok = nullptr, *pos = nullptr, *lnfg = nullptr, *lnfb = nullptr;
    std::vector<Layer> layers;
};

namespace {

void expect_tensor(CsClip* c, const std::string& n, std::vector<int64_t> shape) { c->names.push_back(n); c->expect[n] = std::move(shape); }

void build_manifest(CsClip* c) {
    const int D = c->cfg.hidden_size, I = c->cfg.intermediate_size;
    expect_tensor(c, "embeddings.token_embedding.weight", {c->cfg.vocab_size, D});
    expect_tensor(c, "embeddings.position_embedding.weight", {c->cfg.max_position_embeddings, D});
    for (int l = 0; l < c->cfg.num_hidden_layers; ++l) {
        const std::string p = "encoder.layers." + std::to_string(l);
        for (const char* q : {".self_attn.k_proj", ".self_attn.v_proj", ".self_attn.q_proj", ".self_attn.out_proj"}) {
            expect_tensor(c, p + q + ".weight", {D, D}); expect_tensor(c, p + q + ".bias", {D});
        }
        expect_tensor(c, p + ".layer_norm1.weight", {D}); expect_tensor(c, p + ".layer_norm1.bias", {D});
        expect_tensor(c, p + ".mlp.fc1.weight", {I, D}); expect_tensor(c, p + ".mlp.fc1.bias", {I});
        expect_tensor(c, p + ".mlp.fc2.weight", {D, I}); expect_tensor(c, p + ".mlp.fc2.bias", {D});
        expect_tensor(c, p + ".layer_norm2.weight", {D}); expect_tensor(c, p + ".layer_norm2.bias", {D});
    }
    expect_tensor(c, "final_layer_norm.weight", {D}); expect_tensor(c, "final_layer_norm.bias", {D});
}

f16* upload(CsClip* c, const std::vector<f16>& h) {
    void* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(h.size() * sizeof(f16), 256)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(f16), hipMemcpyHostToDevice) != hipSuccess) { hipFree(d); return nullptr; }
    c->dev_allocs.push_back(d);
    return (f16*)d;
}
const HostT& T(CsClip* c, const std::string& n) { return c->host.at(n); }

int linear(const f16* x, int M, int K, const f16* w, const f16* b, int N, const f16* res, f16* out, hipStream_t s) {
    IgemmArgs a{};
    a.a0 = x; a.c0 = K; a.B = 1; a.Hi = M; a.Wi = 1; a.Ho = M; a.Wo = 1; a.taps = 1; a.stride = 1; a.N = N; a.w = w; a.bias = b; a.res = res; a.out = out;
    return launch_igemm(a, s);
}

}  // namespace

extern "C" {

int cs_clip_create(const CsClipConfig* cfg, CsClip** out) {
    if (!cfg || !out) CS_FAIL(CS_E_ARG, "cfg/out is NULL");
    if (cfg->hidden_size % 128 || cfg->intermediate_size % 128) CS_FAIL(CS_E_SHAPE, "clip: hidden / intermediate size must be multiples of 128");
    if (cfg->num_attention_heads < 1 || cfg->hidden_size != cfg->num_attention_heads * 64) CS_FAIL(CS_E_UNSUPPORTED, "clip: built for heads of dim 64");
    if (cfg->num_hidden_layers < 1 || cfg->vocab_size < 1 || cfg->max_position_embeddings < 1) CS_FAIL(CS_E_ARG, "clip: bad config");
    CsClip* c = new CsClip();
    c->cfg = *cfg;
    build_manifest(c);
    *out = c;
    return CS_OK;
}

void cs_clip_destroy(CsClip* c) {
    if (!c) return;
    for (void* p : c->dev_allocs) hipFree(p);
    delete c;
}

int cs_clip_num_weights(const CsClip* c) { return c ? (int)c->names.size() : 0; }

const char* cs_clip_weight_name(const CsClip* c, int i, int64_t* shape4, int* ndim) {
    if (!c || i < 0 || i >= (int)c->names.size()) return nullptr;
    const auto& sh = c->expect.at(c->names[i]);
    if (ndim) *ndim = (int)sh.size();
    if (shape4) for (size_t k = 0; k < 4; ++k) shape4[k] = k < sh.size() ? sh[k] : 1;
    return c->names[i].c_str();
}

int cs_clip_set_weight(CsClip* c, const char* name, const float* data, const int64_t* shape, int ndim) {
    if (!c || !name || !data || !shape) CS_FAIL(CS_E_ARG, "null argument");
    if (c->finalized) CS_FAIL(CS_E_STATE, "weights are already packed");
    auto it = c->expect.find(name);
    if (it == c->expect.end()) CS_FAIL(CS_E_ARG, "unexpected tensor name '%s'", name);
    if ((int)it->second.size() != ndim) CS_FAIL(CS_E_SHAPE, "%s: rank %d, expected %zu", name, ndim, it->second.size());
    int64_t n = 1;
    for (int k = 0; k < ndim; ++k) {
        if (shape[k] != it->second[k]) CS_FAIL(CS_E_SHAPE, "%s: dim %d is %lld, expected %lld", name, k, (long long)shape[k], (long long)it->second[k]);
        n *= shape[k];
    }
    HostT t; t.shape.assign(shape, shape + ndim); t.data.resize(n);
    for (int64_t i = 0; i < n; ++i) t.data[i] = (f16)data[i];
    c->host[name] = std::move(t);
    return CS_OK;
}

int cs_clip_finalize(CsClip* c) {
    if (!c) CS_FAIL(CS_E_ARG, "null");
    if (c->finalized) return CS_OK;
    for (auto& n : c->names) if (!c->host.count(n)) CS_FAIL(CS_E_STATE, "missing weight '%s'", n.c_str());
    bool ok = true;
    c->tok = upload(c, T(c, "embeddings.token_embedding.weight").data); c->pos = upload(c, T(c, "embeddings.position_embedding.weight").data);
    c->lnfg = upload(c, T(c, "final_layer_norm.weight").data); c->lnfb = upload(c, T(c, "final_layer_norm.bias").data);
    ok = c->tok && c->pos && c->lnfg && c->lnfb;
    c->layers.resize(c->cfg.num_hidden_layers);
    for (int l = 0; l < c->cfg.num_hidden_layers && ok; ++l) {
        const std::string p = "encoder.layers." + std::to_string(l);
        Layer& L = c->layers[l];
        std::vector<f16> w, b;
        for (const char* q : {".self_attn.q_proj", ".self_attn.k_proj", ".self_attn.v_proj"}) {       // fused [3D, D]
            const HostT& tw = T(c, p + q + ".weight"); w.insert(w.end(), tw.data.begin(), tw.data.end());
            const HostT& tb = T(c, p + q + ".bias"); b.insert(b.end(), tb.data.begin(), tb.data.end());
        }
        L.wqkv = upload(c, w); L.bqkv = upload(c, b);
        L.wo = upload(c, T(c, p + ".self_attn.out_proj.weight").data); L.bo = upload(c, T(c, p + ".self_attn.out_proj.bias").data);
        L.ln1g = upload(c, T(c, p + ".layer_norm1.weight").data); L.ln1b = upload(c, T(c, p + ".layer_norm1.bias").data);
        L.ln2g = upload(c, T(c, p + ".layer_norm2.weight").data); L.ln2b = upload(c, T(c, p + ".layer_norm2.bias").data);
        L.w1 = upload(c, T(c, p + ".mlp.fc1.weight").data); L.b1 = upload(c, T(c, p + ".mlp.fc1.bias").data);
        L.w2 = upload(c, T(c, p + ".mlp.fc2.weight").data); L.b2 = upload(c, T(c, p + ".mlp.fc2.bias").data);
        ok = L.wqkv && L.bqkv && L.wo && L.bo && L.ln1g && L.ln1b && L.ln2g && L.ln2b && L.w1 && L.b1 && L.w2 && L.b2;
    }
    if (!ok) CS_FAIL(CS_E_HIP, "clip: weight upload failed (hipMalloc/hipMemcpy)");
    c->host.clear();
    c->finalized = true;
    return CS_OK;
}

size_t cs_clip_workspace_bytes(const CsClip* c, int batch, int seq_len) {
    if (!c || batch <= 0 || seq_len <= 0) return 0;
    const size_t rows = (size_t)batch * seq_len, D = c->cfg.hidden_size, I = c->cfg.intermediate_size;
    return (rows * (D + D + 3 * D + I)) * sizeof(f16) + 4096;       // x, normed, qkv (attention output reuses normed), mlp
}

double cs_clip_flops(const CsClip* c, int batch, int seq_len) {
    if (!c) return 0;
    const double rows = (double)batch * seq_len, D = c->cfg.hidden_size, I = c->cfg.intermediate_size;
    return c->cfg.num_hidden_layers * (2.0 * rows * D * (4 * D + 2 * I) + 4.0 * batch * (double)seq_len * seq_len * D);
}

int cs_clip_encode(CsClip* c, const int64_t* input_ids, int batch, int seq_len, void* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!c) CS_FAIL(CS_E_ARG, "clip is NULL");
    if (!c->finalized) CS_FAIL(CS_E_STATE, "cs_clip_finalize has not been called");
    if (batch < 0 || seq_len < 0) CS_FAIL(CS_E_ARG, "negative size");
    if (batch == 0 || seq_len == 0) return CS_OK;
    if (seq_len > c->cfg.max_position_embeddings) CS_FAIL(CS_E_SHAPE, "clip: sequence of %d tokens exceeds max_position_embeddings %d", seq_len, c->cfg.max_position_embeddings);
    if (!input_ids || !out || !workspace) CS_FAIL(CS_E_ARG, "null pointer");
    if (workspace_bytes < cs_clip_workspace_bytes(c, batch, seq_len)) CS_FAIL(CS_E_ARG, "clip: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int D = c->cfg.hidden_size, I = c->cfg.intermediate_size, H = c->cfg.num_attention_heads;
    const long rows = (long)batch * seq_len;
    f16* x = (f16*)workspace; f16* n = x + rows * D; f16* qkv = n + rows * D; f16* h = qkv + rows * 3 * D;
    int rc = launch_embed_tokens(input_ids, c->tok, c->pos, x, rows, seq_len, D, c->cfg.vocab_size, s);
    for (int l = 0; l < c->cfg.num_hidden_layers && rc == CS_OK; ++l) {
        const Layer& L = c->layers[l];
        rc = launch_layer_norm(x, L.ln1g, L.ln1b, n, (int)rows, D, c->cfg.layer_norm_eps, s);
        if (rc == CS_OK) rc = linear(n, (int)rows, D, L.wqkv, L.bqkv, 3 * D, nullptr, qkv, s);
        if (rc == CS_OK) {
            AttnArgs a{};
            a.q = qkv; a.q_stride = 3 * D; a.k = qkv + D; a.k_stride = 3 * D; a.v = qkv + 2 * D; a.v_stride = 3 * D; a.out = n; a.out_stride = D;
            a.B = batch; a.H = H; a.Nq = seq_len; a.Nk = seq_len; a.dh = 64; a.scale = 0.125f; a.causal = 1;
            rc = launch_attention(a, s);
        }
        if (rc == CS_OK) rc = linear(n, (int)rows, D, L.wo, L.bo, D, x, x, s);                          // + residual
        if (rc == CS_OK) rc = launch_layer_norm(x, L.ln2g, L.ln2b, n, (int)rows, D, c->cfg.layer_norm_eps, s);
        if (rc == CS_OK) rc = linear(n, (int)rows, D, L.w1, L.b1, I, nullptr, h, s);
        if (rc == CS_OK) rc = launch_quick_gelu(h, rows * I, s);
        if (rc == CS_OK) rc = linear(h, (int)rows, I, L.w2, L.b2, D, x, x, s);                          // + residual
    }
    if (rc == CS_OK) rc = launch_layer_norm(x, c->lnfg, c->lnfb, (f16*)out, (int)rows, D, c->cfg.layer_norm_eps, s);
    return rc;
}

}  // extern "C"
