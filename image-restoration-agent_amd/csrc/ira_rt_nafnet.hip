// NAFNet (NAFNet_arch.py:249-361): weight binding, workspace, forward schedule.
#include "ira_model.h"

namespace ira_rt {

// =============================================================================================
// NAFNet (NAFNet_arch.py:249-361): weights, workspace, forward schedule
static int naf_levels(const ira_nafnet_config& c) { return c.n_levels; }

static bool naf_load_block(ira_model* m, const std::string& p, int c, NBlk& b) {
    const size_t NBc = up32(c), NB2 = up32(2 * c);
    m->err.clear();
    b.c1_w = wptr(m, p + ".c1_w", true, 2 * NB2 * c * 2);
    b.c1_b = (const float*)wptr(m, p + ".c1_b", true, NB2 * 4);
    b.dw_w = (const float*)wptr(m, p + ".dw_w", true, (size_t)9 * 2 * c * 4);
    b.dw_b = (const float*)wptr(m, p + ".dw_b", true, (size_t)2 * c * 4);
    b.sca_w = (const float*)wptr(m, p + ".sca_w", true, (size_t)c * c * 4);
    b.sca_b = (const float*)wptr(m, p + ".sca_b", true, (size_t)c * 4);
    b.c3_w = wptr(m, p + ".c3_w", true, 2 * NBc * c * 2);
    b.c3_b = (const float*)wptr(m, p + ".c3_b", true, NBc * 4);
    b.c4_w = wptr(m, p + ".c4_w", true, 2 * NB2 * c * 2);
    b.c4_b = (const float*)wptr(m, p + ".c4_b", true, NB2 * 4);
    b.c5_w = wptr(m, p + ".c5_w", true, 2 * NBc * c * 2);
    b.c5_b = (const float*)wptr(m, p + ".c5_b", true, NBc * 4);
    return m->err.empty();
}

bool naf_bind_weights(ira_model* m) {
    const ira_nafnet_config& c = m->ncfg;
    const int L = c.n_levels;
    m->err.clear();
    m->intro_w = (const float*)wptr(m, "intro_w", true, (size_t)c.width * c.img_channel * 9 * 4);
    m->intro_b = (const float*)wptr(m, "intro_b", true, (size_t)c.width * 4);
    m->ending_w = wptr(m, "ending_w", true, (size_t)2 * 9 * up32(c.img_channel) * c.width * 2);
    m->ending_b = (const float*)wptr(m, "ending_b", true, (size_t)up32(c.img_channel) * 4);
    if (!m->intro_w || !m->intro_b || !m->ending_w || !m->ending_b) return false;
    m->nenc.assign(L, {}); m->ndec.assign(L, {});
    m->ndown_w.assign(L, nullptr); m->ndown_b.assign(L, nullptr); m->nup_w.assign(L, nullptr);
    int ch = c.width;
    for (int i = 0; i < L; ++i) {
        m->nenc[i].resize(c.enc_blk_nums[i]);
        for (int k = 0; k < c.enc_blk_nums[i]; ++k)
            if (!naf_load_block(m, "enc." + std::to_string(i) + "." + std::to_string(k), ch, m->nenc[i][k])) return false;
        m->ndown_w[i] = wptr(m, "down." + std::to_string(i) + "_w", true, (size_t)2 * up32(2 * ch) * 4 * ch * 2);
        m->ndown_b[i] = (const float*)wptr(m, "down." + std::to_string(i) + "_b", true, (size_t)up32(2 * ch) * 4);
        if (!m->ndown_w[i] || !m->ndown_b[i]) return false;
        ch *= 2;
    }
    m->nmid.resize(c.middle_blk_num);
    for (int k = 0; k < c.middle_blk_num; ++k)
        if (!naf_load_block(m, "mid." + std::to_string(k), ch, m->nmid[k])) return false;
    for (int i = 0; i < L; ++i) {
        m->nup_w[i] = wptr(m, "up." + std::to_string(i) + "_w", true, (size_t)2 * up32(2 * ch) * ch * 2);
        if (!m->nup_w[i]) return false;
        ch /= 2;
        m->ndec[i].resize(c.dec_blk_nums[i]);
        for (int k = 0; k < c.dec_blk_nums[i]; ++k)
            if (!naf_load_block(m, "dec." + std::to_string(i) + "." + std::to_string(k), ch, m->ndec[i][k])) return false;
    }
    return true;
}

// (H, W): padded size, multiples of 2^L
static int naf_ensure_workspace(ira_model* m, int T, int H, int W) {
    if (m->ws && T <= m->ws_T && H == m->ws_H && W == m->ws_W) return IRA_OK;
    if (m->ws) { HIP_TRY(m, hipStreamSynchronize(m->st)); HIP_TRY(m, hipFree(m->ws)); m->ws = nullptr; }
    const ira_nafnet_config& c = m->ncfg;
    const int L = c.n_levels;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al256(bytes); return o; };
    const size_t t = (size_t)T;
    std::vector<size_t> o_x(L + 1);
    size_t mt1 = 0, mg = 0, mpart = 0, ms = 0;
    for (int l = 0; l <= L; ++l) {
        const size_t ch = (size_t)c.width << l, P = (size_t)(H >> l) * (W >> l);
        o_x[l] = take(t * P * ch * 4);
        mt1 = std::max(mt1, t * P * 2 * ch * 2);
        mg = std::max(mg, t * P * ch * 2);
        mpart = std::max(mpart, t * (size_t)ira::naf_dw_sg_blocks(H >> l, W >> l, (int)ch) * ch * 4);
        ms = std::max(ms, t * ch * 4);
    }
    const size_t o_t1 = take(mt1), o_g = take(mg), o_p = take(mpart), o_s = take(ms);
    const size_t o_pi = take(t * H * W * c.img_channel * 4), o_po = take(t * H * W * c.img_channel * 4);
    hipError_t e = hipMalloc((void**)&m->ws, off);
    if (e != hipSuccess) {
        m->ws = nullptr; m->ws_T = 0;
        return m->fail(IRA_ERR_OOM, "workspace hipMalloc(%zu bytes) for T=%d H=%d W=%d: %s", off, T, H, W, hipGetErrorString(e));
    }
    m->ws_bytes = off; m->ws_T = T; m->ws_H = H; m->ws_W = W;
    char* b = m->ws;
    m->nx.resize(L + 1);
    for (int l = 0; l <= L; ++l) m->nx[l] = (float*)(b + o_x[l]);
    m->hp = b + o_t1; m->g = b + o_g; m->part = (float*)(b + o_p); m->nsca = (float*)(b + o_s);
    m->npad_in = (float*)(b + o_pi); m->npad_out = (float*)(b + o_po);
    return IRA_OK;
}

// one NAFBlock in place on x (NAFNet_arch.py:198-219)
static int naf_run_block(ira_model* m, const NBlk& b, float* x, int T, int H, int W, int c) {
    const int P = H * W, NBc = up32(c) / 32, NB2 = up32(2 * c) / 32;
    char k_[6][64];
    if (m->profile) {
        const char* fam[6] = {"naf_ln_conv1", "naf_dw_gate", "naf_sca", "naf_conv3_res", "naf_ln_conv4", "naf_gate_conv5_res"};
        for (int i = 0; i < 6; ++i) snprintf(k_[i], sizeof k_[i], "%s|C%d|%dx%d|T%d", fam[i], c, H, W, T);
    } else {
        for (int i = 0; i < 6; ++i) k_[i][0] = 0;
    }
    hipStream_t st = m->st;
    const int bpt = ira::naf_dw_sg_blocks(H, W, c);
    RUN(m, k_[0], ira::launch_pw_kres(st, x, true, 2, c, b.c1_w, b.c1_b, m->hp, 2 * c, 2 * c, NB2, T, P, 1e-6f));
    RUN(m, k_[1], ira::launch_naf_dw_sg(st, m->hp, b.dw_w, b.dw_b, m->g, m->part, T, H, W, c));
    RUN(m, k_[2], ira::launch_naf_sca(st, m->part, bpt, (long)P, b.sca_w, b.sca_b, m->nsca, T, c));
    if (m->use_fused_ffn && ira::naf_tail_supported(c)) {
        char kt[64];
        if (m->profile) snprintf(kt, sizeof kt, "naf_tail_fused|C%d|%dx%d|T%d", c, H, W, T); else kt[0] = 0;
        RUN(m, kt, ira::launch_naf_tail(st, m->g, m->nsca, x, c, b.c3_w, b.c3_b, b.c4_w, b.c4_b, b.c5_w, b.c5_b, T, P));
        return IRA_OK;
    }
    RUN(m, k_[3], ira::launch_naf_pw(st, 4, 0, m->g, c, b.c3_w, 0, b.c3_b, x, x, c, NBc, T, H, W, m->nsca));
    RUN(m, k_[4], ira::launch_pw_kres(st, x, true, 2, c, b.c4_w, b.c4_b, m->hp, 2 * c, 2 * c, NB2, T, P, 1e-6f));
    RUN(m, k_[5], ira::launch_naf_pw(st, 1, 0, m->hp, c, b.c5_w, 0, b.c5_b, x, x, c, NBc, T, H, W));
    return IRA_OK;
}

// NAFNet.forward (:301-326) on device NHWC fp32 (T, H, W, img_channel), any H, W (zero padded to 2^L here)
int naf_forward_enqueue(ira_model* m, const float* in, int T, int H, int W, float* out) {
    const ira_nafnet_config& c = m->ncfg;
    const int L = c.n_levels, ps = 1 << L;
    if (T <= 0 || H <= 0 || W <= 0) return m->fail(IRA_ERR_ARG, "forward: T=%d H=%d W=%d must be positive", T, H, W);
    const int Hp = (H + ps - 1) / ps * ps, Wp = (W + ps - 1) / ps * ps;          // check_image_size :356-361
    int rc = naf_ensure_workspace(m, T, Hp, Wp);
    if (rc != IRA_OK) return rc;
    hipStream_t st = m->st;
    const bool padded = (Hp != H) || (Wp != W);
    const float* xin = in;
    float* xout = out;
    if (padded) {
        RUN(m, "naf_pad", ira::launch_naf_pad_crop(st, in, m->npad_in, T, H, W, Hp, Wp, c.img_channel));
        xin = m->npad_in; xout = m->npad_out;
    }
    auto nb = [](int N) { return up32(N) / 32; };
    RUN(m, "naf_intro3x3", ira::launch_embed(st, xin, m->intro_w, m->intro_b, m->nx[0], T, Hp, Wp, c.img_channel, c.width));   // :305
    int ch = c.width, h = Hp, w = Wp;
    for (int l = 0; l < L; ++l) {                                                                   // :309-312
        for (const NBlk& b : m->nenc[l]) if ((rc = naf_run_block(m, b, m->nx[l], T, h, w, ch))) return rc;
        RUN(m, "naf_down2x2", ira::launch_naf_pw(st, 2, 0, m->nx[l], 4 * ch, m->ndown_w[l], 0, m->ndown_b[l], nullptr, m->nx[l + 1],
                                                 2 * ch, nb(2 * ch), T, h / 2, w / 2));
        ch *= 2; h /= 2; w /= 2;
    }
    for (const NBlk& b : m->nmid) if ((rc = naf_run_block(m, b, m->nx[L], T, h, w, ch))) return rc;  // :314
    for (int i = 0; i < L; ++i) {                                                                   // :316-321
        const int l = L - 1 - i;
        // x = PixelShuffle(conv1x1(x)) + enc_skip, written over the skip buffer
        RUN(m, "naf_up_shuffle_add", ira::launch_naf_pw(st, 3, 1, m->nx[l + 1], ch, m->nup_w[i], 0, nullptr, m->nx[l], m->nx[l],
                                                        2 * ch, nb(2 * ch), T, h, w));
        ch /= 2; h *= 2; w *= 2;
        for (const NBlk& b : m->ndec[i]) if ((rc = naf_run_block(m, b, m->nx[l], T, h, w, ch))) return rc;
    }
    RUN(m, "naf_ending3x3", ira::launch_conv3x3(st, m->nx[0], m->ending_w, m->ending_b, xin, xout, 3, T, Hp, Wp, c.width,
                                                c.img_channel, nb(c.img_channel)));                  // :323-324
    if (padded) RUN(m, "naf_crop", ira::launch_naf_pad_crop(st, m->npad_out, out, T, Hp, Wp, H, W, c.img_channel));   // :326
    return IRA_OK;
}

}  // namespace ira_rt
