// SwinIR (network_swinir.py:618-840): weight binding, workspace, forward schedule.
#include "ira_model.h"

namespace ira_rt {

// =============================================================================================
// SwinIR (network_swinir.py:618-840): weights, workspace, forward schedule
struct SwGeo { int C, Cp, hid, hidp, qp, heads_max; };
static SwGeo sw_geo(const ira_swinir_config& c) {
    SwGeo g;
    g.C = c.embed_dim; g.Cp = up32(c.embed_dim); g.hid = c.mlp_hidden; g.hidp = up32(c.mlp_hidden); g.qp = up32(c.embed_dim / 4);
    g.heads_max = 0;
    for (int i = 0; i < c.n_layers; ++i) g.heads_max = std::max(g.heads_max, c.num_heads[i]);
    return g;
}

static bool sw_load_conv(ira_model* m, const std::string& p, bool three, const SwGeo& g, SConv& c) {
    if (!three) {
        c.w0 = wptr(m, p + "_w", true, (size_t)2 * 9 * g.Cp * g.Cp * 2);
        c.b0 = (const float*)wptr(m, p + "_b", true, (size_t)g.Cp * 4);
        return c.w0 && c.b0;
    }
    c.w0 = wptr(m, p + ".0_w", true, (size_t)2 * 9 * g.qp * g.Cp * 2);
    c.b0 = (const float*)wptr(m, p + ".0_b", true, (size_t)g.qp * 4);
    c.w2 = wptr(m, p + ".2_w", true, (size_t)2 * g.qp * g.qp * 2);
    c.b2 = (const float*)wptr(m, p + ".2_b", true, (size_t)g.qp * 4);
    c.w4 = wptr(m, p + ".4_w", true, (size_t)2 * 9 * g.Cp * g.qp * 2);
    c.b4 = (const float*)wptr(m, p + ".4_b", true, (size_t)g.Cp * 4);
    return c.w0 && c.b0 && c.w2 && c.b2 && c.w4 && c.b4;
}

// ira_set_weight_planes(1) on a SwinIR handle: the lo planes of the transformer 1x1 weights are zeroed in the handle's copy of the blob (see
// restormer_zero_lo_planes), whatever the embedding size; the frame-sized launches of K = 192 / 256 layers then skip the zero plane.
void sw_lo_planes(ira_model* m, std::vector<std::pair<void*, size_t>>& out) {
    const ira_swinir_config& c = m->scfg;
    const SwGeo g = sw_geo(c);
    auto add = [&](const void* w, size_t total) { if (w) out.emplace_back((char*)const_cast<void*>(w) + total / 2, total / 2); };
    for (int i = 0; i < c.n_layers; ++i) {
        const size_t Kh = (size_t)c.num_heads[i] * 32;
        for (SBlk& b : m->sw_blocks[i]) {
            add(b.qkv_w, (size_t)2 * 3 * Kh * g.Cp * 2);
            add(b.proj_w, (size_t)2 * g.Cp * Kh * 2);
            add(b.fc1_w, (size_t)2 * g.hidp * g.Cp * 2);
            add(b.fc2_w, (size_t)2 * g.Cp * g.hidp * 2);
        }
    }
}

bool sw_bind_weights(ira_model* m) {
    const ira_swinir_config& c = m->scfg;
    const SwGeo g = sw_geo(c);
    const bool three = c.resi_3conv != 0;
    m->err.clear();
    m->sw_first_w = (const float*)wptr(m, "first_w", true, (size_t)g.Cp * c.in_chans * 9 * 4);
    m->sw_first_b = (const float*)wptr(m, "first_b", true, (size_t)g.Cp * 4);
    m->sw_pe_g = (const float*)wptr(m, "pe_norm_g", true, (size_t)g.Cp * 4);
    m->sw_pe_b = (const float*)wptr(m, "pe_norm_b", true, (size_t)g.Cp * 4);
    m->sw_norm_g = (const float*)wptr(m, "norm_g", true, (size_t)g.Cp * 4);
    m->sw_norm_b = (const float*)wptr(m, "norm_b", true, (size_t)g.Cp * 4);
    if (!m->sw_first_w || !m->sw_first_b || !m->sw_pe_g || !m->sw_pe_b || !m->sw_norm_g || !m->sw_norm_b) return false;
    m->sw_blocks.assign(c.n_layers, {});
    m->sw_lconv.assign(c.n_layers, SConv());
    for (int i = 0; i < c.n_layers; ++i) {
        const int heads = c.num_heads[i], Kh = heads * 32;
        m->sw_blocks[i].resize(c.depths[i]);
        for (int k = 0; k < c.depths[i]; ++k) {
            const std::string p = "l" + std::to_string(i) + ".b" + std::to_string(k);
            SBlk& b = m->sw_blocks[i][k];
            b.qkv_w = wptr(m, p + ".qkv_w", true, (size_t)2 * 3 * Kh * g.Cp * 2);
            b.qkv_b = (const float*)wptr(m, p + ".qkv_b", true, (size_t)3 * Kh * 4);
            b.bias_tab = (const float*)wptr(m, p + ".bias_tab", true, (size_t)heads * 64 * 64 * 4);
            b.proj_w = wptr(m, p + ".proj_w", true, (size_t)2 * g.Cp * Kh * 2);
            b.proj_b = (const float*)wptr(m, p + ".proj_b", true, (size_t)g.Cp * 4);
            b.fc1_w = wptr(m, p + ".fc1_w", true, (size_t)2 * g.hidp * g.Cp * 2);
            b.fc1_b = (const float*)wptr(m, p + ".fc1_b", true, (size_t)g.hidp * 4);
            b.fc2_w = wptr(m, p + ".fc2_w", true, (size_t)2 * g.Cp * g.hidp * 2);
            b.fc2_b = (const float*)wptr(m, p + ".fc2_b", true, (size_t)g.Cp * 4);
            if (!b.qkv_w || !b.qkv_b || !b.bias_tab || !b.proj_w || !b.proj_b || !b.fc1_w || !b.fc1_b || !b.fc2_w || !b.fc2_b) return false;
        }
        if (!sw_load_conv(m, "l" + std::to_string(i) + ".conv", three, g, m->sw_lconv[i])) return false;
    }
    if (!sw_load_conv(m, "cab", three, g, m->sw_cab)) return false;
    const int nf = 64, oc32 = up32(c.in_chans);
    auto c3 = [&](const char* n, int K, int N) { return wptr(m, n, true, (size_t)2 * 9 * up32(N) * K * 2); };
    auto bb = [&](const char* n, int N) { return (const float*)wptr(m, n, true, (size_t)up32(N) * 4); };
    if (c.upsampler == 0) {
        m->sw_last_w = c3("last_w", g.Cp, c.in_chans); m->sw_last_b = bb("last_b", c.in_chans);
        return m->sw_last_w && m->sw_last_b;
    }
    m->sw_cbu_w = c3("cbu_w", g.Cp, nf); m->sw_cbu_b = bb("cbu_b", nf);
    const int n_up = c.upscale == 4 ? 2 : (c.upscale == 2 ? 1 : 0);
    for (int j = 0; j < n_up; ++j) {
        const std::string n = "up" + std::to_string(j);
        const int N = c.upsampler == 1 ? 4 * nf : nf;
        m->sw_up_w[j] = c3((n + "_w").c_str(), nf, N); m->sw_up_b[j] = bb((n + "_b").c_str(), N);
        if (!m->sw_up_w[j] || !m->sw_up_b[j]) return false;
    }
    if (c.upsampler == 2) {
        m->sw_hr_w = c3("hr_w", nf, nf); m->sw_hr_b = bb("hr_b", nf);
        if (!m->sw_hr_w || !m->sw_hr_b) return false;
    }
    m->sw_last_w = c3("last_w", nf, c.in_chans); m->sw_last_b = bb("last_b", c.in_chans);
    (void)oc32;
    return m->sw_cbu_w && m->sw_cbu_b && m->sw_last_w && m->sw_last_b;
}

static int sw_ensure_workspace(ira_model* m, int T, int Hp, int Wp) {
    if (m->ws && T <= m->ws_T && Hp == m->ws_H && Wp == m->ws_W) return IRA_OK;
    if (m->ws) { HIP_TRY(m, hipStreamSynchronize(m->st)); HIP_TRY(m, hipFree(m->ws)); m->ws = nullptr; }
    const ira_swinir_config& c = m->scfg;
    const SwGeo g = sw_geo(c);
    const size_t tp = (size_t)T * Hp * Wp, s2 = (size_t)c.upscale * c.upscale;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al256(bytes); return o; };
    const size_t o_xp = take(tp * c.in_chans * 4), o_xf = take(tp * g.Cp * 4), o_xa = take(tp * g.Cp * 4), o_xb = take(tp * g.Cp * 4);
    const size_t o_qkv = take(tp * 3 * g.heads_max * 32 * 2), o_att = take(tp * g.heads_max * 32 * 2), o_hid = take(tp * g.hidp * 2);
    const size_t o_c1 = take(c.resi_3conv ? tp * g.qp * 4 : 256), o_c2 = take(c.resi_3conv ? tp * g.qp * 4 : 256);
    const size_t ub = c.upsampler ? tp * s2 * 64 * 4 : 256;
    const size_t o_ua = take(ub), o_ub = take(ub), o_y = take(tp * s2 * c.in_chans * 4);
    hipError_t e = hipMalloc((void**)&m->ws, off);
    if (e != hipSuccess) {
        m->ws = nullptr; m->ws_T = 0;
        return m->fail(IRA_ERR_OOM, "workspace hipMalloc(%zu bytes) for T=%d H=%d W=%d: %s", off, T, Hp, Wp, hipGetErrorString(e));
    }
    m->ws_bytes = off; m->ws_T = T; m->ws_H = Hp; m->ws_W = Wp;
    char* b = m->ws;
    m->sw_xp = (float*)(b + o_xp); m->sw_xf = (float*)(b + o_xf); m->sw_xa = (float*)(b + o_xa); m->sw_xb = (float*)(b + o_xb);
    m->sw_qkv = b + o_qkv; m->sw_att = b + o_att; m->sw_hid = b + o_hid;
    m->sw_c1 = (float*)(b + o_c1); m->sw_c2 = (float*)(b + o_c2); m->sw_ua = (float*)(b + o_ua); m->sw_ub = (float*)(b + o_ub);
    m->sw_y = (float*)(b + o_y);
    return IRA_OK;
}

// y (may alias skip) = resi_conv(x) + skip   ('1conv' / '3conv')
static int sw_resi_conv(ira_model* m, const SConv& cv, const float* x, const float* skip, float* y, int T, int H, int W, const char* fam) {
    const ira_swinir_config& c = m->scfg;
    const SwGeo g = sw_geo(c);
    hipStream_t st = m->st;
    if (!c.resi_3conv) {
        RUN(m, fam, ira::launch_conv3x3(st, x, cv.w0, cv.b0, skip, y, 3, T, H, W, g.Cp, g.Cp, g.Cp / 32));
        return IRA_OK;
    }
    RUN(m, fam, ira::launch_conv3x3(st, x, cv.w0, cv.b0, nullptr, m->sw_c1, 4, T, H, W, g.Cp, g.qp, g.qp / 32, 0.2f));
    RUN(m, fam, ira::launch_naf_pw(st, 3, 0, m->sw_c1, g.qp, cv.w2, 0, cv.b2, nullptr, m->sw_c2, g.qp, g.qp / 32, T, H, W));
    RUN(m, fam, ira::launch_lrelu(st, m->sw_c2, (long)T * H * W * g.qp, 0.2f));
    RUN(m, fam, ira::launch_conv3x3(st, m->sw_c2, cv.w4, cv.b4, skip, y, 3, T, H, W, g.qp, g.Cp, g.Cp / 32));
    return IRA_OK;
}

int sw_forward_enqueue(ira_model* m, const float* in, int T, int H, int W, float* out) {
    const ira_swinir_config& c = m->scfg;
    const SwGeo g = sw_geo(c);
    const int ws = c.window_size, s = c.upscale, ic = c.in_chans;
    if (T <= 0 || H <= 0 || W <= 0) return m->fail(IRA_ERR_ARG, "forward: T=%d H=%d W=%d must be positive", T, H, W);
    const int Hp = (H + ws - 1) / ws * ws, Wp = (W + ws - 1) / ws * ws;                       // check_image_size :783-788
    if (Hp - H >= H || Wp - W >= W) return m->fail(IRA_ERR_ARG, "reflect padding needs pad < image size (F.pad 'reflect')");
    int rc = sw_ensure_workspace(m, T, Hp, Wp);
    if (rc != IRA_OK) return rc;
    hipStream_t st = m->st;
    const int P = Hp * Wp;
    static const float rgb_mean[3] = {0.4488f, 0.4371f, 0.4040f};                             // :659-662
    static const float zero_mean[3] = {0.f, 0.f, 0.f};
    const float* mean = ic == 3 ? rgb_mean : zero_mean;
    RUN(m, "swin_pre", ira::launch_swin_pre(st, in, m->sw_xp, T, H, W, Hp, Wp, ic, mean, c.img_range));                      // :807-810
    RUN(m, "swin_conv_first", ira::launch_embed(st, m->sw_xp, m->sw_first_w, m->sw_first_b, m->sw_xf, T, Hp, Wp, ic, g.Cp));   // :814
    RUN(m, "swin_layernorm", ira::launch_ln_affine(st, m->sw_xf, m->sw_pe_g, m->sw_pe_b, m->sw_xa, (long)T * P, g.C, g.Cp, 1e-5f));   // :792
    char k_[5][64];
    for (int i = 0; i < c.n_layers; ++i) {
        const int heads = c.num_heads[i], Kh = heads * 32;
        if (m->profile) {
            const char* fam[5] = {"swin_ln_qkv", "swin_window_attn", "swin_proj_res", "swin_ln_fc1_gelu", "swin_fc2_res"};
            for (int q = 0; q < 5; ++q) snprintf(k_[q], sizeof k_[q], "%s|C%d|%dx%d|T%d", fam[q], g.C, Hp, Wp, T);
        } else {
            for (int q = 0; q < 5; ++q) k_[q][0] = 0;
        }
        // The RSTB keeps its input (:482) for the residual behind its convolution.  Where the second half of a block is ONE kernel, the first
        // block reads that input where it lies (LN + qkv, and the projection's residual) and writes into the working copy: no copy.
        static const bool proj_fused = [] { const char* v = getenv("IRA_SWIN_PROJ_MLP"); return !(v && *v == '0'); }();
        const bool mlp1 = ira::swin_mlp_supported(g.Cp, g.hidp, (long)T * P);
        const bool one_kernel = mlp1 && proj_fused && Kh == g.Cp && c.depths[i] > 0 && m->sw_blocks[i][0].proj_b;
        if (!one_kernel) HIP_TRY(m, hipMemcpyAsync(m->sw_xb, m->sw_xa, (size_t)T * P * g.Cp * 4, hipMemcpyDeviceToDevice, st));
        float* x = m->sw_xb;
        for (int k = 0; k < c.depths[i]; ++k) {
            const SBlk& b = m->sw_blocks[i][k];
            const int shift = (k % 2 == 0) ? 0 : ws / 2;                                                             // :385-386
            const float* xsrc = (one_kernel && k == 0) ? m->sw_xa : x;
            RUN(m, k_[0], ira::launch_pw_kres(st, xsrc, true, 2, g.Cp, b.qkv_w, b.qkv_b, m->sw_qkv, 3 * Kh, 3 * Kh, 3 * heads, T, P, 1e-5f, g.C, 0, m->w_planes));
            RUN(m, k_[1], ira::launch_swin_attn(st, m->sw_qkv, b.bias_tab, m->sw_att, T, Hp, Wp, heads, ws, shift));
            if (mlp1 && proj_fused && Kh == g.Cp && b.proj_b) {            // proj + residual + LN + fc1 + GELU + fc2 + residual: the second half of the block in one kernel
                char km[64];
                if (m->profile) snprintf(km, sizeof km, "swin_proj_mlp_fused|C%d|%dx%d|T%d", g.C, Hp, Wp, T); else km[0] = 0;
                RUN(m, km, ira::launch_swin_mlp(st, x, b.fc1_w, b.fc1_b, b.fc2_w, b.fc2_b, g.Cp, g.hidp, (long)T * P, 1e-5f, g.C, m->w_planes, m->sw_att, b.proj_w, b.proj_b,
                                                xsrc == x ? nullptr : xsrc));
                continue;
            }
            RUN(m, k_[2], ira::launch_pw_nres(st, m->sw_att, Kh, nullptr, 0, false, b.proj_w, 0, b.proj_b, x, x, g.Cp, g.Cp / 32, T, P, m->w_planes));
            if (mlp1) {      // LN + fc1 + GELU + fc2 + residual in one kernel: the hidden tensor never exists
                char km[64];
                if (m->profile) snprintf(km, sizeof km, "swin_mlp_fused|C%d|%dx%d|T%d", g.C, Hp, Wp, T); else km[0] = 0;
                RUN(m, km, ira::launch_swin_mlp(st, x, b.fc1_w, b.fc1_b, b.fc2_w, b.fc2_b, g.Cp, g.hidp, (long)T * P, 1e-5f, g.C, m->w_planes));
                continue;
            }
            RUN(m, k_[3], ira::launch_pw_kres(st, x, true, 2, g.Cp, b.fc1_w, b.fc1_b, m->sw_hid, g.hidp, g.hidp, g.hidp / 32, T, P, 1e-5f, g.C, 1, m->w_planes));
            RUN(m, k_[4], ira::launch_pw_nres(st, m->sw_hid, g.hidp, nullptr, 0, false, b.fc2_w, 0, b.fc2_b, x, x, g.Cp, g.Cp / 32, T, P, m->w_planes));
        }
        if ((rc = sw_resi_conv(m, m->sw_lconv[i], x, m->sw_xa, m->sw_xa, T, Hp, Wp, "swin_rstb_conv"))) return rc;
    }
    RUN(m, "swin_layernorm", ira::launch_ln_affine(st, m->sw_xa, m->sw_norm_g, m->sw_norm_b, m->sw_xb, (long)T * P, g.C, g.Cp, 1e-5f));  // :800
    if ((rc = sw_resi_conv(m, m->sw_cab, m->sw_xb, m->sw_xf, m->sw_xa, T, Hp, Wp, "swin_conv_after_body"))) return rc;                    // + x_first
    const float* body = m->sw_xa;
    const int nb64 = 2, nbo = up32(ic) / 32;
    int Ho = Hp, Wo = Wp;
    if (c.upsampler == 0) {                                                                                          // :835-837
        RUN(m, "swin_conv_last", ira::launch_conv3x3(st, body, m->sw_last_w, m->sw_last_b, m->sw_xp, m->sw_y, 3, T, Hp, Wp, g.Cp, ic, nbo));
    } else {
        float *ua = m->sw_ua, *ub = m->sw_ub;
        RUN(m, "swin_conv_before_upsample", ira::launch_conv3x3(st, body, m->sw_cbu_w, m->sw_cbu_b, nullptr, ua, 4, T, Hp, Wp, g.Cp, 64, nb64, 0.01f));
        const int n_up = s == 4 ? 2 : (s == 2 ? 1 : 0);
        for (int j = 0; j < n_up; ++j) {
            if (c.upsampler == 1) {                                                                                  // Upsample :580-592
                RUN(m, "swin_upsample_conv_shuffle", ira::launch_conv3x3(st, ua, m->sw_up_w[j], m->sw_up_b[j], nullptr, ub, 2, T, Ho, Wo, 64, 256, 8));
                std::swap(ua, ub);
            } else {                                                                                                 // :830-833
                RUN(m, "swin_nearest2x", ira::launch_nearest2x(st, ua, ub, T, Ho, Wo, 64));
                RUN(m, "swin_conv_up", ira::launch_conv3x3(st, ub, m->sw_up_w[j], m->sw_up_b[j], nullptr, ua, 4, T, 2 * Ho, 2 * Wo, 64, 64, nb64, 0.2f));
            }
            Ho *= 2; Wo *= 2;
        }
        if (c.upsampler == 2) {
            RUN(m, "swin_conv_hr", ira::launch_conv3x3(st, ua, m->sw_hr_w, m->sw_hr_b, nullptr, ub, 4, T, Ho, Wo, 64, 64, nb64, 0.2f));
            std::swap(ua, ub);
        }
        RUN(m, "swin_conv_last", ira::launch_conv3x3(st, ua, m->sw_last_w, m->sw_last_b, nullptr, m->sw_y, 0, T, Ho, Wo, 64, ic, nbo));
    }
    RUN(m, "swin_post", ira::launch_swin_post(st, m->sw_y, out, T, Ho, Wo, H * s, W * s, ic, mean, c.img_range));   // :838-840
    return IRA_OK;
}

}  // namespace ira_rt
