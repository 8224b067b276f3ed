// Sanitizer harness for the HOST side of libconsolver_hip (the reference has no sanitizer story; this is ours).
// Built by tests/test_sanitize_host.py with -fsanitize=address,undefined from the real unet.cpp / vae.cpp / flux.cpp / clip.cpp / ops_api.cpp / api.cpp
// plus tests/sanitize/stubs.cpp (host malloc as device memory, launch stubs that touch every tensor's first and last byte).  Drives, without a GPU:
//   weight registration and repacking, finalize, the dry-run workspace sizing (every execution variant, both residual-stream modes), forwards through the
//   first-fit arena with a workspace of EXACTLY the size the library asked for, the profiling event pool, and the error paths.
#include "../../include/consolver_hip.h"
#include "../../include/consolver_hip_ops.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static int g_fail = 0;
#define EXPECT(cond) do { if (!(cond)) { fprintf(stderr, "FAIL %s:%d: %s  (last error: %s)\n", __FILE__, __LINE__, #cond, cs_last_error()); ++g_fail; } } while (0)

static void unet_part() {
    CsUNetConfig c{};
    c.in_channels = 4; c.out_channels = 4;
    const int boc[4] = {320, 320, 640, 640};
    for (int i = 0; i < 4; ++i) { c.block_out_channels[i] = boc[i]; c.down_has_attn[i] = i < 3; c.up_has_attn[i] = i > 0; }
    c.layers_per_block = 1; c.num_heads = 8; c.cross_attention_dim = 768; c.norm_num_groups = 32; c.sample_size = 32; c.ctx_len = 77;
    CsUNet* u = nullptr;
    EXPECT(cs_unet_create(nullptr, &u) != CS_OK);
    CsUNetConfig bad = c; bad.block_out_channels[1] = 100;
    EXPECT(cs_unet_create(&bad, &u) != CS_OK);
    EXPECT(cs_unet_create(&c, &u) == CS_OK);
    EXPECT(cs_unet_workspace_bytes(u, 2) == 0);                       // not finalized
    EXPECT(cs_unet_forward(u, &c, 1, 1, nullptr, 1, &c, &c, &c, 16, 0, nullptr) != CS_OK);
    const int n = cs_unet_num_weights(u);
    EXPECT(n > 100);
    int64_t shape[4]; int nd = 0;
    EXPECT(cs_unet_weight_name(u, -1, shape, &nd) == nullptr && cs_unet_weight_name(u, n, shape, &nd) == nullptr);
    std::vector<float> buf;
    for (int i = 0; i < n; ++i) {
        const char* name = cs_unet_weight_name(u, i, shape, &nd);
        size_t cnt = 1; for (int k = 0; k < nd; ++k) cnt *= (size_t)shape[k];
        buf.assign(cnt, 0.01f);
        if (i == 3) {                                                 // error paths on a real tensor
            int64_t wrong[4] = {shape[0] + 1, shape[1], shape[2], shape[3]};
            EXPECT(cs_unet_set_weight(u, name, buf.data(), wrong, nd) != CS_OK);
            EXPECT(cs_unet_set_weight(u, name, buf.data(), shape, nd + 1) != CS_OK);
            EXPECT(cs_unet_set_weight(u, "no.such.tensor", buf.data(), shape, nd) != CS_OK);
            EXPECT(cs_unet_set_weight(u, name, nullptr, shape, nd) != CS_OK);
        }
        if (i == n - 1) EXPECT(cs_unet_finalize(u) != CS_OK);          // one tensor still missing
        EXPECT(cs_unet_set_weight(u, name, buf.data(), shape, nd) == CS_OK);
    }
    EXPECT(cs_unet_finalize(u) == CS_OK && cs_unet_finalize(u) == CS_OK);
    EXPECT(cs_unet_set_weight(u, "conv_in.bias", buf.data(), shape, 1) != CS_OK);   // packed already
    EXPECT(cs_unet_flops(u, 2) > 0 && cs_unet_flops_executed(u, 2, 2) > 0 && cs_unet_flops_executed(u, 2, 2) < cs_unet_flops_executed(u, 4, 1));
    // per-handle knobs: validated against the knob table, applied around the forwards below, dropped again
    EXPECT(cs_unet_set_tuning(u, "no_such_knob", 1) != CS_OK && cs_unet_set_tuning(u, "ln_fold", 9) != CS_OK && cs_unet_set_tuning(nullptr, "ln_fold", 0) != CS_OK);
    EXPECT(cs_unet_set_tuning(u, "ln_fold", 0) == CS_OK && cs_unet_set_tuning(u, "xattn_fused", 0) == CS_OK);
    EXPECT(cs_unet_set_residual_precision(u, 7) != CS_OK);

    const int S = c.sample_size;
    for (int mode : {CS_RESIDUAL_F16, CS_RESIDUAL_F16X2}) {
        EXPECT(cs_unet_set_residual_precision(u, mode) == CS_OK && cs_unet_get_residual_precision(u) == mode);
        for (int B : {1, 2, 3, 6}) {
            const size_t wsb = cs_unet_workspace_bytes(u, B);
            EXPECT(wsb > 0);
            char* ws = (char*)malloc(wsb);                            // exactly what the library asked for: one byte past it is an ASan error
            std::vector<char> lat((size_t)B * 4 * S * S * 2), ctx((size_t)B * 77 * 768 * 2), out((size_t)B * 4 * S * S * 2);
            std::vector<float> t(B, 499.f);
            struct K { const char* key; int v; };
            const std::vector<std::vector<K>> variants = {{}, {{"cfg_share", 0}}, {{"xattn_fused", 0}}, {{"gn_fuse", 0}}, {{"cfg_share", 0}, {"xattn_fused", 0}, {"gn_fuse", 0}},
                                                          {{"x2_split_a", 0}}, {{"x2_split_a", 3}}, {{"ln_fold", 0}}, {{"ln_fold", 0}, {"xattn_fused", 0}, {"cfg_share", 0}}, {{"conv_in_mfma", 0}}, {{"conv_in_mfma", 0}, {"cfg_share", 0}}, {{"gemm_w8", 0}, {"gemm_lw", 0}, {"conv_lw", 0}}};
            for (auto& var : variants) {
                for (auto& k : var) EXPECT(cs_set_tuning(k.key, k.v) == CS_OK);
                EXPECT(cs_unet_forward(u, lat.data(), B, 1, t.data(), 1, ctx.data(), out.data(), ws, wsb, 0, nullptr) == CS_OK);
                EXPECT(cs_unet_forward(u, lat.data(), B, 1, t.data(), B, ctx.data(), out.data(), ws, wsb, 1, nullptr) == CS_OK);      // per-sample timesteps, cached K/V
                if (B % 2 == 0) EXPECT(cs_unet_forward(u, lat.data(), B / 2, 2, t.data(), 1, ctx.data(), out.data(), ws, wsb, 0, nullptr) == CS_OK);   // CFG dual batch
                EXPECT(cs_reset_tuning() == CS_OK);
            }
            // profiling pool (events created on first use, reused afterwards)
            EXPECT(cs_unet_set_profiling(u, 1) == CS_OK);
            for (int rep = 0; rep < 2; ++rep) EXPECT(cs_unet_forward(u, lat.data(), B, 1, t.data(), 1, ctx.data(), out.data(), ws, wsb, 0, nullptr) == CS_OK);
            double ms, fl, by; int ln;
            for (int i = 0; i < cs_unet_profile_entries(u); ++i) EXPECT(cs_unet_profile_entry(u, i, &ms, &fl, &by, &ln) != nullptr);
            EXPECT(cs_unet_profile_entry(u, 99, &ms, &fl, &by, &ln) == nullptr);
            EXPECT(cs_unet_set_profiling(u, 0) == CS_OK);
            // error paths of the run itself
            EXPECT(cs_unet_forward(u, lat.data(), B, 1, t.data(), 1, ctx.data(), out.data(), ws, wsb / 2, 0, nullptr) != CS_OK);        // arena runs dry mid-forward
            EXPECT(cs_unet_forward(u, lat.data(), B, 1, t.data(), 1, ctx.data(), out.data(), ws, 1024, 0, nullptr) != CS_OK);
            EXPECT(cs_unet_forward(u, lat.data(), B, 3, t.data(), 1, ctx.data(), out.data(), ws, wsb, 0, nullptr) != CS_OK);
            EXPECT(cs_unet_forward(u, nullptr, B, 1, t.data(), 1, ctx.data(), out.data(), ws, wsb, 0, nullptr) != CS_OK);
            EXPECT(cs_unet_forward(u, lat.data(), B, 1, t.data(), B + 1, ctx.data(), out.data(), ws, wsb, 0, nullptr) != CS_OK);
            EXPECT(cs_unet_forward(u, lat.data(), 0, 1, t.data(), 1, ctx.data(), out.data(), ws, wsb, 0, nullptr) == CS_OK);            // empty batch: no-op
            free(ws);
        }
    }
    EXPECT(cs_unet_clear_tuning(u) == CS_OK);
    // a workspace sized in one residual mode is refused (not overrun) in the other
    EXPECT(cs_unet_set_residual_precision(u, CS_RESIDUAL_F16) == CS_OK);
    const size_t small = cs_unet_workspace_bytes(u, 2);
    EXPECT(cs_unet_set_residual_precision(u, CS_RESIDUAL_F16X2) == CS_OK);
    EXPECT(cs_unet_workspace_bytes(u, 2) > small);
    {
        char* ws = (char*)malloc(small);
        std::vector<char> lat(2 * 4 * S * S * 2), ctx(2 * 77 * 768 * 2), out(2 * 4 * S * S * 2); float t = 1.f;
        EXPECT(cs_unet_forward(u, lat.data(), 2, 1, &t, 1, ctx.data(), out.data(), ws, small, 0, nullptr) != CS_OK);
        free(ws);
    }
    // round 6, with the handle's own overrides cleared (folded LayerNorms on): byte lo planes of the transformer hidden state (knob lo8: the arena hands out HALF-size
    // planes, the stubs touch exactly that many bytes), per-block unfold masks (cs_unet_calibrate_ln_fold's result), the fp32 output tensor, the calibration forward
    for (int B : {2, 4}) {
        for (unsigned mask : {0u, 0x5u, 0xffffu}) {
            EXPECT(cs_unet_set_ln_unfold_mask(u, mask) == CS_OK && cs_unet_get_ln_unfold_mask(u) == mask);
            const size_t wsb = cs_unet_workspace_bytes(u, B);
            char* ws = (char*)malloc(wsb);
            std::vector<char> lat((size_t)B * 4 * S * S * 2), ctx((size_t)B * 77 * 768 * 2), out((size_t)B * 4 * S * S * 2), out32((size_t)B * 4 * S * S * 4);
            float t = 499.f;
            for (const char* k : {"", "lo8", "cfg_share", "xattn_fused", "x2_sc_skip"}) {
                if (*k) EXPECT(cs_set_tuning(k, 0) == CS_OK);
                EXPECT(cs_unet_forward(u, lat.data(), B, 1, &t, 1, ctx.data(), out.data(), ws, wsb, 0, nullptr) == CS_OK);
                EXPECT(cs_unet_forward(u, lat.data(), B / 2, 2, &t, 1, ctx.data(), out.data(), ws, wsb, 0, nullptr) == CS_OK);
                EXPECT(cs_reset_tuning() == CS_OK);
            }
            EXPECT(cs_unet_set_output_dtype(u, 9) != CS_OK && cs_unet_set_output_dtype(u, CS_F32) == CS_OK && cs_unet_get_output_dtype(u) == CS_F32);
            EXPECT(cs_unet_forward(u, lat.data(), B / 2, 2, &t, 1, ctx.data(), out32.data(), ws, wsb, 0, nullptr) == CS_OK);
            EXPECT(cs_unet_set_output_dtype(u, CS_F16) == CS_OK);
            if (mask == 0u) {
                unsigned m = 77; float worst = -1.f;
                EXPECT(cs_unet_calibrate_ln_fold(u, lat.data(), B / 2, 2, &t, 1, ctx.data(), out.data(), ws, wsb, 4.0f, nullptr, &m, &worst) == CS_OK && m != 77u && worst >= 0.f);   // (the stubs touch the statistics, they do not compute them)
                EXPECT(cs_unet_set_ln_unfold_mask(u, 0) == CS_OK);
                EXPECT(cs_unet_calibrate_ln_fold(u, lat.data(), B / 2, 2, &t, 1, ctx.data(), out.data(), ws, wsb, 0.0f, nullptr, &m, &worst) != CS_OK);
                EXPECT(cs_unet_calibrate_ln_fold(nullptr, lat.data(), B / 2, 2, &t, 1, ctx.data(), out.data(), ws, wsb, 4.0f, nullptr, &m, &worst) != CS_OK);
            }
            free(ws);
        }
    }
    EXPECT(cs_unet_set_ln_unfold_mask(u, 0) == CS_OK);
    cs_unet_destroy(u);
    cs_unet_destroy(nullptr);
}

static void vae_part() {
    CsVaeConfig c{};
    c.latent_channels = 4; c.out_channels = 3;
    const int boc[4] = {128, 128, 256, 256};
    for (int i = 0; i < 4; ++i) c.block_out_channels[i] = boc[i];
    c.layers_per_block = 1; c.norm_num_groups = 32; c.sample_size = 16; c.use_post_quant_conv = 1; c.with_encoder = 1; c.use_quant_conv = 1;
    CsVae* v = nullptr;
    EXPECT(cs_vae_create(&c, &v) == CS_OK);
    if (!v) return;
    const int n = cs_vae_num_weights(v);
    int64_t shape[4]; int nd = 0; std::vector<float> buf;
    for (int i = 0; i < n; ++i) {
        const char* name = cs_vae_weight_name(v, i, shape, &nd);
        size_t cnt = 1; for (int k = 0; k < nd; ++k) cnt *= (size_t)shape[k];
        buf.assign(cnt, 0.01f);
        EXPECT(cs_vae_set_weight(v, name, buf.data(), shape, nd) == CS_OK);
    }
    EXPECT(cs_vae_set_weight(v, "decoder.no_such", buf.data(), shape, nd) != CS_OK);
    EXPECT(cs_vae_finalize(v) == CS_OK);
    for (int gn_fuse : {1, 0})
        for (int B : {1, 3}) {
            EXPECT(cs_set_tuning("gn_fuse", 1) == CS_OK);              // the workspace is sized with the knob ON and must serve both settings
            const size_t wsb = cs_vae_workspace_bytes(v, B), web = cs_vae_encode_workspace_bytes(v, B);
            EXPECT(wsb > 0 && web > 0);
            EXPECT(cs_set_tuning("gn_fuse", gn_fuse) == CS_OK);
            EXPECT(cs_vae_workspace_bytes(v, B) == wsb);               // ... and does not depend on it
            const int S = c.sample_size;
            std::vector<char> lat((size_t)B * 4 * S * S * 2), img((size_t)B * 3 * 64 * S * S * 2);
            char* ws = (char*)malloc(wsb);
            EXPECT(cs_vae_decode(v, lat.data(), B, 1.0f / 0.18215f, 0.f, img.data(), 1, ws, wsb, nullptr) == CS_OK);
            EXPECT(cs_vae_decode(v, lat.data(), B, 1.0f, 0.f, img.data(), 1, ws, wsb / 4, nullptr) != CS_OK);
            free(ws);
            ws = (char*)malloc(web);
            EXPECT(cs_vae_encode(v, img.data(), B, 0.18215f, 0.f, lat.data(), ws, web, nullptr) == CS_OK);
            free(ws);
        }
    EXPECT(cs_reset_tuning() == CS_OK);
    cs_vae_destroy(v);
}

static void flux_part() {
    CsFluxConfig c{};
    c.in_channels = 64; c.num_layers = 2; c.num_single_layers = 2; c.num_heads = 2; c.head_dim = 128; c.joint_attention_dim = 256; c.pooled_projection_dim = 64;
    c.guidance_embeds = 1; c.axes_dims_rope[0] = 16; c.axes_dims_rope[1] = 56; c.axes_dims_rope[2] = 56; c.dtype = CS_BF16;
    CsFlux* f = nullptr;
    EXPECT(cs_flux_create(&c, &f) == CS_OK);
    if (!f) return;
    const int n = cs_flux_num_weights(f);
    int64_t shape[4]; int nd = 0; std::vector<unsigned short> buf;
    for (int i = 0; i < n; ++i) {
        const char* name = cs_flux_weight_name(f, i, shape, &nd);
        size_t cnt = 1; for (int k = 0; k < nd; ++k) cnt *= (size_t)shape[k];
        buf.assign(cnt, 0x3c00);
        if (i == 5) { int64_t wrong[2] = {shape[0] + 1, shape[1]}; EXPECT(cs_flux_set_weight(f, name, buf.data(), 0, wrong, nd) != CS_OK); }
        EXPECT(cs_flux_set_weight(f, name, buf.data(), 0, shape, nd) == CS_OK);
    }
    EXPECT(cs_flux_finalize(f) == CS_OK);
    EXPECT(cs_flux_set_residual_precision(f, 5) != CS_OK && cs_flux_get_residual_precision(f) == CS_RESIDUAL_F16X2);       // default: split hidden-state stream
    for (int mode : {CS_RESIDUAL_F16X2, CS_RESIDUAL_F16})
    for (int B : {1, 2}) {
        EXPECT(cs_flux_set_residual_precision(f, mode) == CS_OK && cs_flux_get_residual_precision(f) == mode);
        const int T = 64, Lq = 256, Li = 256, D2 = c.head_dim / 2;
        const size_t wsb = cs_flux_workspace_bytes(f, B, T, Lq + Li);
        EXPECT(wsb > 0 && cs_flux_flops(f, B, T, Lq + Li) > 0);
        char* ws = (char*)malloc(wsb);
        std::vector<char> lat((size_t)B * Lq * 64 * 2), img((size_t)B * Li * 64 * 2), enc((size_t)B * T * 256 * 2), out((size_t)B * (Lq + Li) * 64 * 2);
        std::vector<float> pooled(B * 64, 0.1f), ts(B, 0.5f), gd(B, 2.5f), rc((size_t)(T + Lq + Li) * D2, 1.f), rs((size_t)(T + Lq + Li) * D2, 0.f);
        EXPECT(cs_flux_forward_joint(f, lat.data(), Lq, img.data(), Li, B, enc.data(), T, pooled.data(), ts.data(), gd.data(), rc.data(), rs.data(), out.data(), ws, wsb, nullptr) == CS_OK);
        std::vector<char> hid((size_t)B * (Lq + Li) * 64 * 2);
        EXPECT(cs_flux_forward(f, hid.data(), B, Lq + Li, enc.data(), T, pooled.data(), ts.data(), gd.data(), rc.data(), rs.data(), out.data(), ws, wsb, nullptr) == CS_OK);
        EXPECT(cs_flux_forward(f, hid.data(), B, Lq + Li, enc.data(), T, pooled.data(), ts.data(), gd.data(), rc.data(), rs.data(), out.data(), ws, wsb / 3, nullptr) != CS_OK);
        free(ws);
        // round 6: the fp32 output (the output head's two planes summed; split stream only): its own workspace query, an fp32 `out`, refused on the one-plane stream
        EXPECT(cs_flux_set_output_dtype(f, 77) != CS_OK && cs_flux_set_output_dtype(f, CS_F32) == CS_OK && cs_flux_get_output_dtype(f) == CS_F32);
        {
            const size_t wsb32 = cs_flux_workspace_bytes(f, B, T, Lq + Li);
            char* ws32 = (char*)malloc(wsb32);
            std::vector<char> out32((size_t)B * (Lq + Li) * 64 * 4);
            const int rc32 = cs_flux_forward_joint(f, lat.data(), Lq, img.data(), Li, B, enc.data(), T, pooled.data(), ts.data(), gd.data(), rc.data(), rs.data(), out32.data(), ws32, wsb32, nullptr);
            EXPECT((rc32 == CS_OK) == (mode == CS_RESIDUAL_F16X2));
            free(ws32);
        }
        EXPECT(cs_flux_set_output_dtype(f, c.dtype) == CS_OK && cs_flux_get_output_dtype(f) == c.dtype);
    }
    cs_flux_destroy(f);
}

int main() {
    EXPECT(cs_abi_version() == 2);
    EXPECT(cs_set_tuning("conv_lw", 5) != CS_OK && cs_set_tuning("no_such_knob", 1) != CS_OK && cs_set_tuning("attn_qt40", 3) != CS_OK);
    int v = -7;
    EXPECT(cs_get_tuning("conv_lw", &v) == CS_OK && v == 1);
    unet_part();
    vae_part();
    flux_part();
    if (g_fail) { fprintf(stderr, "%d expectation(s) failed\n", g_fail); return 1; }
    printf("sanitize harness: ok\n");
    return 0;
}
