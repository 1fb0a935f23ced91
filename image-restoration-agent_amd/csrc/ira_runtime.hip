// libira_hip.so runtime: model handle, weight blob, workspace, Restormer forward schedule, tiler,
// and the C ABI of include/ira.h.  No torch, no CPU compute fallback.
//
// Reference path this replaces: restormer_arch.py:245-284 (network), demo.py:109-168 /
// model_serving.py:158-209 (harness).  See DESIGN.md for layouts.
#include "../../include/ira.h"
#include "ira_kernels.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;   // errors of calls that have no handle

struct Blk {   // one TransformerBlock (restormer_arch.py:137-150), device pointers into the weight blob
    const void* qkv_w = nullptr;  const float* qkv_b = nullptr;
    const float* dw1_w = nullptr; const float* dw1_b = nullptr;
    const float* temp = nullptr;  const float* wo = nullptr; const float* wo_b = nullptr;
    const void* pin_w = nullptr;  const float* pin_b = nullptr;
    const float* dw2_w = nullptr; const float* dw2_b = nullptr;
    const void* pout_w = nullptr; const float* pout_b = nullptr;
};

struct Prof { long launches = 0; double ms = 0; };

struct SBlk {  // one SwinTransformerBlock (network_swinir.py:164-279)
    const void* qkv_w = nullptr; const float* qkv_b = nullptr; const float* bias_tab = nullptr;
    const void* proj_w = nullptr; const float* proj_b = nullptr;
    const void* fc1_w = nullptr; const float* fc1_b = nullptr;
    const void* fc2_w = nullptr; const float* fc2_b = nullptr;
};
struct SConv {  // '1conv' (w0 only) or '3conv' (3x3, 1x1, 3x3) residual connection conv (:462-469, :727-735)
    const void *w0 = nullptr, *w2 = nullptr, *w4 = nullptr;
    const float *b0 = nullptr, *b2 = nullptr, *b4 = nullptr;
};

struct NBlk {  // one NAFBlock (NAFNet_arch.py:163-219), device pointers into the weight blob
    const void* c1_w = nullptr;  const float* c1_b = nullptr;     // conv1 with norm1 folded in
    const float* dw_w = nullptr; const float* dw_b = nullptr;     // conv2 (depthwise 3x3)
    const float* sca_w = nullptr; const float* sca_b = nullptr;   // sca.1
    const void* c3_w = nullptr; const float* c3_b = nullptr;    // diag(beta) conv3, beta*b3 (the SCA scale is applied to its input)
    const void* c4_w = nullptr;  const float* c4_b = nullptr;     // conv4 with norm2 folded in
    const void* c5_w = nullptr;  const float* c5_b = nullptr;     // diag(gamma) conv5, gamma*b5
};

}  // namespace

struct ira_model {
    int device = -1;
    int arch = 0;                    // 0 Restormer, 1 NAFNet, 2 DnCNN, 3 SwinIR
    ira_restormer_config cfg{};
    ira_nafnet_config ncfg{};
    ira_dncnn_config dcfg{};
    ira_swinir_config scfg{};
    std::vector<std::vector<SBlk>> sw_blocks;
    std::vector<SConv> sw_lconv;
    SConv sw_cab;
    const float *sw_first_w = nullptr, *sw_first_b = nullptr, *sw_pe_g = nullptr, *sw_pe_b = nullptr, *sw_norm_g = nullptr, *sw_norm_b = nullptr;
    const void *sw_cbu_w = nullptr, *sw_up_w[2] = {nullptr, nullptr}, *sw_hr_w = nullptr, *sw_last_w = nullptr;
    const float *sw_cbu_b = nullptr, *sw_up_b[2] = {nullptr, nullptr}, *sw_hr_b = nullptr, *sw_last_b = nullptr;
    float *sw_xp = nullptr, *sw_xf = nullptr, *sw_xa = nullptr, *sw_xb = nullptr, *sw_c1 = nullptr, *sw_c2 = nullptr, *sw_ua = nullptr, *sw_ub = nullptr,
          *sw_y = nullptr;
    void *sw_qkv = nullptr, *sw_att = nullptr, *sw_hid = nullptr;
    const float *dn_first_w = nullptr, *dn_first_b = nullptr, *dn_last_b = nullptr;
    const void* dn_last_w = nullptr;
    std::vector<const void*> dn_mid_w;
    std::vector<const float*> dn_mid_b;
    // NAFNet weights / workspace
    std::vector<std::vector<NBlk>> nenc, ndec;
    std::vector<NBlk> nmid;
    const float *intro_w = nullptr, *intro_b = nullptr, *ending_b = nullptr;
    const void* ending_w = nullptr;
    std::vector<const void*> ndown_w, nup_w;
    std::vector<const float*> ndown_b;
    std::vector<float*> nx;          // residual stream per level (encoder output = skip, reused by the decoder)
    float *npad_in = nullptr, *npad_out = nullptr, *nsca = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, pev0 = nullptr, pev1 = nullptr;
    std::mutex mu;
    std::string err, prof_json;
    bool profile = false;
    std::map<std::string, Prof> prof;

    // weights
    char* wblob = nullptr; size_t wbytes = 0;
    std::map<std::string, std::pair<size_t, size_t>> toc;   // name -> (offset, nbytes)
    std::vector<Blk> enc1, enc2, enc3, lat, dec3, dec2, dec1, refine;
    const float* embed_w = nullptr; const float* embed_b = nullptr;
    const void *down12 = nullptr, *down23 = nullptr, *down34 = nullptr, *up43 = nullptr, *up32 = nullptr, *up21 = nullptr;
    const void *red3 = nullptr, *red2 = nullptr; const float *red3_b = nullptr, *red2_b = nullptr;
    const void* out_w = nullptr; const float* out_b = nullptr;

    // workspace
    char* ws = nullptr; size_t ws_bytes = 0;
    int ws_T = 0, ws_H = 0, ws_W = 0;
    float *xe1, *xe2, *xe3, *x4, *xd3, *xd2, *xd1, *u3, *u2, *u1, *part, *attnA;
    float *ae1, *ad1, *ae2, *ad2;     // ping-pong partners of the residual stream (fused FFN writes out of place)
    bool use_fused_ffn = true;
    void *qkvp, *qk, *v, *hp, *g, *M;
    int max_batch = 0;
    // harness scratch
    char* hs = nullptr; size_t hs_bytes = 0;

    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        err = buf;
        return code;
    }
};

namespace {

int gfail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(m, expr)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return (m)->fail(e_ == hipErrorOutOfMemory ? IRA_ERR_OOM : IRA_ERR_HIP, "%s: %s (%s:%d)", #expr, \
                             hipGetErrorString(e_), __FILE__, __LINE__);                          \
    } while (0)

inline int up32(int n) { return (n + 31) / 32 * 32; }
inline size_t al256(size_t n) { return (n + 255) / 256 * 256; }

struct Geo {   // per-level geometry for a batch of T tiles of H x W
    int H[4], W[4], C[4], heads[4], hid[4], Hp[4];
    long P[4];
};

Geo make_geo(const ira_restormer_config& c, int H, int W) {
    Geo g;
    for (int l = 0; l < 4; ++l) {
        g.H[l] = H >> l; g.W[l] = W >> l; g.P[l] = (long)g.H[l] * g.W[l];
        g.C[l] = c.dim << l; g.heads[l] = c.heads[l]; g.hid[l] = c.hidden[l]; g.Hp[l] = up32(c.hidden[l]);
    }
    return g;
}

int gram_splits(int T, int heads, long nblk) {
    // depends on the tile geometry only (never on T): a tile's result must not depend on what else is
    // in its launch-batch, and the partials are reduced in fixed order (bitwise reproducible)
    (void)T; (void)heads;
    const long nsteps = (nblk + 3) / 4;
    return (int)std::max(1L, std::min(nsteps / 64, 64L));
}

// ---- profiling wrapper: run one launch, optionally bracketed by events
template <class F>
int run(ira_model* m, const char* family, F&& f) {
    if (m->profile) {
        HIP_TRY(m, hipEventRecord(m->pev0, m->st));
        HIP_TRY(m, f());
        HIP_TRY(m, hipEventRecord(m->pev1, m->st));
        HIP_TRY(m, hipEventSynchronize(m->pev1));
        float ms = 0;
        HIP_TRY(m, hipEventElapsedTime(&ms, m->pev0, m->pev1));
        Prof& p = m->prof[family];
        p.launches++; p.ms += ms;
        return IRA_OK;
    }
    HIP_TRY(m, f());
    return IRA_OK;
}
#define RUN(m, fam, call)                                        \
    do {                                                         \
        int rc_ = run((m), (fam), [&]() { return (call); });     \
        if (rc_ != IRA_OK) return rc_;                           \
    } while (0)

// ---- weights
struct TocEntry { char name[112]; uint64_t offset; uint64_t nbytes; };
struct BlobHeader { char magic[8]; uint32_t n_entries; uint32_t reserved; };

const void* wptr(ira_model* m, const std::string& name, bool required, size_t expect_bytes = 0) {
    auto it = m->toc.find(name);
    if (it == m->toc.end()) {
        if (required) m->err = "weight blob: missing entry '" + name + "'";
        return nullptr;
    }
    if (expect_bytes && it->second.second != expect_bytes) {
        m->err = "weight blob: entry '" + name + "' has " + std::to_string(it->second.second) + " bytes, expected " +
                 std::to_string(expect_bytes);
        return nullptr;
    }
    return m->wblob + it->second.first;
}

bool load_block(ira_model* m, const std::string& p, int C, int heads, int Hp, Blk& b) {
    const size_t NBq = up32(3 * C), NBc = up32(C);
    const bool cb = m->cfg.conv_bias != 0, lb = m->cfg.ln_with_bias != 0;
    m->err.clear();
    b.qkv_w = wptr(m, p + ".qkv_w", true, 2 * NBq * C * 2);        // hi + lo planes
    b.qkv_b = (const float*)wptr(m, p + ".qkv_b", cb || lb, NBq * 4);
    b.dw1_w = (const float*)wptr(m, p + ".dw1_w", true, (size_t)9 * 3 * C * 4);
    b.dw1_b = (const float*)wptr(m, p + ".dw1_b", cb, (size_t)3 * C * 4);
    b.temp = (const float*)wptr(m, p + ".temp", true, (size_t)heads * 4);
    b.wo = (const float*)wptr(m, p + ".wo", true, (size_t)C * C * 4);
    b.wo_b = (const float*)wptr(m, p + ".wo_b", cb, NBc * 4);
    b.pin_w = wptr(m, p + ".pin_w", true, (size_t)2 * 2 * Hp * C * 2);
    b.pin_b = (const float*)wptr(m, p + ".pin_b", cb || lb, (size_t)2 * Hp * 4);
    b.dw2_w = (const float*)wptr(m, p + ".dw2_w", true, (size_t)9 * 2 * Hp * 4);
    b.dw2_b = (const float*)wptr(m, p + ".dw2_b", cb, (size_t)2 * Hp * 4);
    b.pout_w = wptr(m, p + ".pout_w", true, 2 * NBc * Hp * 2);
    b.pout_b = (const float*)wptr(m, p + ".pout_b", cb, NBc * 4);
    return m->err.empty();
}

bool load_stage(ira_model* m, const char* name, int n, int C, int heads, int Hp, std::vector<Blk>& out) {
    out.resize(n);
    for (int i = 0; i < n; ++i)
        if (!load_block(m, std::string(name) + "." + std::to_string(i), C, heads, Hp, out[i])) return false;
    return true;
}

bool bind_weights(ira_model* m) {
    const ira_restormer_config& c = m->cfg;
    const Geo g = make_geo(c, 8, 8);
    const bool cb = c.conv_bias != 0;
    m->err.clear();
    m->embed_w = (const float*)wptr(m, "embed_w", true, (size_t)c.dim * c.inp_channels * 9 * 4);
    if (!m->embed_w) return false;
    if (!load_stage(m, "encoder_level1", c.num_blocks[0], g.C[0], g.heads[0], g.Hp[0], m->enc1)) return false;
    if (!load_stage(m, "encoder_level2", c.num_blocks[1], g.C[1], g.heads[1], g.Hp[1], m->enc2)) return false;
    if (!load_stage(m, "encoder_level3", c.num_blocks[2], g.C[2], g.heads[2], g.Hp[2], m->enc3)) return false;
    if (!load_stage(m, "latent", c.num_blocks[3], g.C[3], g.heads[3], g.Hp[3], m->lat)) return false;
    if (!load_stage(m, "decoder_level3", c.num_blocks[2], g.C[2], g.heads[2], g.Hp[2], m->dec3)) return false;
    if (!load_stage(m, "decoder_level2", c.num_blocks[1], g.C[1], g.heads[1], g.Hp[1], m->dec2)) return false;
    // decoder_level1 / refinement run at dim*2 channels with heads[0] (restormer_arch.py:232-234)
    if (!load_stage(m, "decoder_level1", c.num_blocks[0], g.C[1], g.heads[0], g.Hp[1], m->dec1)) return false;
    if (!load_stage(m, "refinement", c.num_refinement_blocks, g.C[1], g.heads[0], g.Hp[1], m->refine)) return false;
    auto c3 = [&](const char* n, int K, int N) { return wptr(m, n, true, (size_t)2 * 9 * up32(N) * K * 2); };
    m->down12 = c3("down1_2_w", g.C[0], g.C[0] / 2);
    m->down23 = c3("down2_3_w", g.C[1], g.C[1] / 2);
    m->down34 = c3("down3_4_w", g.C[2], g.C[2] / 2);
    m->up43 = c3("up4_3_w", g.C[3], g.C[3] * 2);
    m->up32 = c3("up3_2_w", g.C[2], g.C[2] * 2);
    m->up21 = c3("up2_1_w", g.C[1], g.C[1] * 2);
    m->red3 = wptr(m, "reduce3_w", true, (size_t)2 * up32(g.C[2]) * g.C[3] * 2);
    m->red2 = wptr(m, "reduce2_w", true, (size_t)2 * up32(g.C[1]) * g.C[2] * 2);
    m->red3_b = (const float*)wptr(m, "reduce3_b", cb, (size_t)up32(g.C[2]) * 4);
    m->red2_b = (const float*)wptr(m, "reduce2_b", cb, (size_t)up32(g.C[1]) * 4);
    m->out_w = c3("output_w", g.C[1], c.out_channels);
    m->out_b = (const float*)wptr(m, "output_b", cb, (size_t)up32(c.out_channels) * 4);
    m->embed_b = nullptr;   // OverlapPatchEmbed is always bias=False (restormer_arch.py:157)
    return m->down12 && m->down23 && m->down34 && m->up43 && m->up32 && m->up21 && m->red3 && m->red2 && m->out_w &&
           (!cb || (m->red3_b && m->red2_b && m->out_b));
}

// ---- workspace
int ensure_workspace(ira_model* m, int T, int H, int W) {
    if (m->ws && T <= m->ws_T && H == m->ws_H && W == m->ws_W) return IRA_OK;
    if (m->ws) { HIP_TRY(m, hipStreamSynchronize(m->st)); HIP_TRY(m, hipFree(m->ws)); m->ws = nullptr; }
    const Geo g = make_geo(m->cfg, H, W);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al256(bytes); return o; };
    const size_t t = (size_t)T;
    size_t o_xe1 = take(t * g.P[0] * g.C[0] * 4), o_xe2 = take(t * g.P[1] * g.C[1] * 4), o_xe3 = take(t * g.P[2] * g.C[2] * 4),
           o_x4 = take(t * g.P[3] * g.C[3] * 4), o_xd3 = take(t * g.P[2] * g.C[2] * 4), o_xd2 = take(t * g.P[1] * g.C[1] * 4),
           o_xd1 = take(t * g.P[0] * g.C[1] * 4), o_u3 = take(t * g.P[2] * g.C[2] * 4), o_u2 = take(t * g.P[1] * g.C[1] * 4),
           o_u1 = take(t * g.P[0] * g.C[0] * 4);
    size_t o_ae1 = take(t * g.P[0] * g.C[0] * 4), o_ad1 = take(t * g.P[0] * g.C[1] * 4), o_ae2 = take(t * g.P[1] * g.C[1] * 4),
           o_ad2 = take(t * g.P[1] * g.C[1] * 4);
    // branch scratch: max over the (level, channel) pairs actually run; level-1 decoder runs at 2*dim
    size_t mq = 0, mqk = 0, mv = 0, mh = 0, mg = 0, mM = 0, mpart = 0, mattn = 0;
    for (int l = 0; l < 4; ++l) {
        const int nvar = (l == 0) ? 2 : 1;
        for (int var = 0; var < nvar; ++var) {
            const int C = (var == 1) ? g.C[1] : g.C[l];
            const int heads = g.heads[l];
            const int Hp = (var == 1) ? g.Hp[1] : g.Hp[l];
            const long nblk = (long)g.H[l] * ((g.W[l] + 7) / 8);
            const int S = gram_splits(T, heads, nblk);
            const int c = C / heads;
            mq = std::max(mq, t * g.P[l] * 3 * C * 2);
            mqk = std::max(mqk, t * nblk * 2 * C * 8 * 2);
            mv = std::max(mv, t * g.P[l] * C * 2);
            mh = std::max(mh, t * g.P[l] * 2 * Hp * 2);
            mg = std::max(mg, t * g.P[l] * Hp * 2);
            mM = std::max(mM, t * 2 * up32(C) * C * 2);
            mpart = std::max(mpart, t * S * heads * ((size_t)c * c + 2 * c) * 4);
            mattn = std::max(mattn, t * (size_t)C * c * 4);
        }
    }
    size_t o_q = take(mq), o_qk = take(mqk), o_v = take(mv), o_h = take(mh), o_g = take(mg), o_M = take(mM), o_p = take(mpart), o_a = take(mattn);
    hipError_t e = hipMalloc((void**)&m->ws, off);
    if (e != hipSuccess) {
        m->ws = nullptr; m->ws_T = 0;
        return m->fail(IRA_ERR_OOM, "workspace hipMalloc(%zu bytes) for T=%d H=%d W=%d: %s", off, T, H, W, hipGetErrorString(e));
    }
    m->ws_bytes = off; m->ws_T = T; m->ws_H = H; m->ws_W = W;
    char* b = m->ws;
    m->xe1 = (float*)(b + o_xe1); m->xe2 = (float*)(b + o_xe2); m->xe3 = (float*)(b + o_xe3); m->x4 = (float*)(b + o_x4);
    m->xd3 = (float*)(b + o_xd3); m->xd2 = (float*)(b + o_xd2); m->xd1 = (float*)(b + o_xd1);
    m->u3 = (float*)(b + o_u3); m->u2 = (float*)(b + o_u2); m->u1 = (float*)(b + o_u1);
    m->ae1 = (float*)(b + o_ae1); m->ad1 = (float*)(b + o_ad1); m->ae2 = (float*)(b + o_ae2); m->ad2 = (float*)(b + o_ad2);
    m->qkvp = b + o_q; m->qk = b + o_qk; m->v = b + o_v; m->hp = b + o_h; m->g = b + o_g; m->M = b + o_M;
    m->part = (float*)(b + o_p); m->attnA = (float*)(b + o_a);
    // M rows >= C are MFMA padding and must read as zero
    HIP_TRY(m, hipMemsetAsync(m->M, 0, al256(mM), m->st));
    return IRA_OK;
}

int ensure_scratch(ira_model* m, size_t bytes) {
    if (m->hs && bytes <= m->hs_bytes) return IRA_OK;
    if (m->hs) { HIP_TRY(m, hipStreamSynchronize(m->st)); HIP_TRY(m, hipFree(m->hs)); m->hs = nullptr; m->hs_bytes = 0; }
    hipError_t e = hipMalloc((void**)&m->hs, bytes);
    if (e != hipSuccess) { m->hs = nullptr; return m->fail(IRA_ERR_OOM, "scratch hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
    m->hs_bytes = bytes;
    return IRA_OK;
}

// ---- one TransformerBlock: x += attn(LN1(x)); x += ffn(LN2(x))   (restormer_arch.py:146-150)
// x: current residual stream buffer; alt: a second buffer of the same size (or nullptr).  The fused FFN writes
// x -> alt and the two are swapped, so on return x still names the buffer holding the block's output.
int run_block(ira_model* m, const Blk& b, float*& x, float*& alt, int T, int H, int W, int C, int heads, int Hp) {
    const int P = H * W;
    char k_[8][64];
    if (m->profile) {
        const char* fam[8] = {"ln_qkv_1x1", "dwconv_qkv", "gram", "attn_mix", "attn_apply_res", "ln_pin_1x1", "dwconv_gate", "pout_res"};
        for (int i = 0; i < 8; ++i) snprintf(k_[i], sizeof k_[i], "%s|C%d|%dx%d|T%d", fam[i], C, H, W, T);
    } else {
        for (int i = 0; i < 8; ++i) k_[i][0] = 0;
    }
    const int ln = m->cfg.ln_with_bias ? 2 : 1;
    const int NBq = up32(3 * C) / 32, NBc = up32(C) / 32;
    const long nblk = (long)H * ((W + 7) / 8);
    const int S = gram_splits(T, heads, nblk);
    hipStream_t st = m->st;
    // MDTA (:111-132)
    if (m->use_fused_ffn && ira::qkv_fused_supported(C)) {
        char kq[64];
        if (m->profile) snprintf(kq, sizeof kq, "qkv_fused|C%d|%dx%d|T%d", C, H, W, T); else kq[0] = 0;
        RUN(m, kq, ira::launch_qkv_fused(st, x, ln, C, b.qkv_w, b.qkv_b, b.dw1_w, b.dw1_b, m->qk, m->v, T, H, W));
    } else {
        RUN(m, k_[0], ira::launch_pw_kres(st, x, true, ln, C, b.qkv_w, b.qkv_b, m->qkvp, 3 * C, 3 * C, NBq, T, P));
        RUN(m, k_[1], ira::launch_dwconv_qkv(st, m->qkvp, b.dw1_w, b.dw1_b, m->qk, m->v, T, H, W, C));
    }
    RUN(m, k_[2], ira::launch_gram(st, m->qk, m->part, T, (int)nblk, C, heads, S));
    RUN(m, k_[3], ira::launch_attn_mix(st, m->part, b.wo, b.temp, m->attnA, m->M, T, C, heads, S, NBc));
    RUN(m, k_[4], ira::launch_pw_nres(st, m->v, C, nullptr, 0, false, m->M, 2L * NBc * 32 * C, b.wo_b, x, x, C, NBc, T, P));
    // GDFN (:88-93)
    if (alt != nullptr && m->use_fused_ffn && ira::ffn_fused_supported(C, Hp)) {
        char kf[64];
        if (m->profile) snprintf(kf, sizeof kf, "ffn_fused|C%d|%dx%d|T%d", C, H, W, T); else kf[0] = 0;
        RUN(m, kf, ira::launch_ffn_fused(st, x, alt, ln, C, b.pin_w, b.pin_b, b.dw2_w, b.dw2_b, b.pout_w, b.pout_b, T, H, W, Hp));
        std::swap(x, alt);
        return IRA_OK;
    }
    RUN(m, k_[5], ira::launch_pw_kres(st, x, true, ln, C, b.pin_w, b.pin_b, m->hp, 2 * Hp, 2 * Hp, 2 * Hp / 32, T, P));
    RUN(m, k_[6], ira::launch_dwconv_gate(st, m->hp, b.dw2_w, b.dw2_b, m->g, T, H, W, Hp));
    RUN(m, k_[7], ira::launch_pw_nres(st, m->g, Hp, nullptr, 0, false, b.pout_w, 0, b.pout_b, x, x, C, NBc, T, P));
    return IRA_OK;
}

int run_stage(ira_model* m, const std::vector<Blk>& blks, float*& x, float* alt, int T, int H, int W, int C, int heads, int Hp) {
    for (const Blk& b : blks) {
        int rc = run_block(m, b, x, alt, T, H, W, C, heads, Hp);
        if (rc != IRA_OK) return rc;
    }
    return IRA_OK;
}

// ---- Restormer.forward (restormer_arch.py:245-284) on device NHWC fp32, T tiles of H x W (enqueue only)
int restormer_forward_enqueue(ira_model* m, const float* in, int T, int H, int W, float* out) {
    const ira_restormer_config& c = m->cfg;
    if (T <= 0 || H <= 0 || W <= 0 || (H & 7) || (W & 7))
        return m->fail(IRA_ERR_ARG, "forward: H=%d W=%d must be positive multiples of 8 (demo.py:122-127), T=%d", H, W, T);
    int rc = ensure_workspace(m, T, H, W);
    if (rc != IRA_OK) return rc;
    const Geo g = make_geo(c, H, W);
    hipStream_t st = m->st;
    auto nb = [](int N) { return up32(N) / 32; };

    float *e1 = m->xe1, *e2 = m->xe2, *e3 = m->xe3, *l4 = m->x4, *d3 = m->xd3, *d2 = m->xd2, *d1 = m->xd1;
    RUN(m, "embed3x3", ira::launch_embed(st, in, m->embed_w, m->embed_b, e1, T, g.H[0], g.W[0], c.inp_channels, g.C[0]));            // :247
    if ((rc = run_stage(m, m->enc1, e1, m->ae1, T, g.H[0], g.W[0], g.C[0], g.heads[0], g.Hp[0]))) return rc;                         // :248
    RUN(m, "conv3x3_down", ira::launch_conv3x3(st, e1, m->down12, nullptr, nullptr, e2, 1, T, g.H[0], g.W[0], g.C[0], g.C[0] / 2, nb(g.C[0] / 2)));   // :250
    if ((rc = run_stage(m, m->enc2, e2, m->ae2, T, g.H[1], g.W[1], g.C[1], g.heads[1], g.Hp[1]))) return rc;
    RUN(m, "conv3x3_down", ira::launch_conv3x3(st, e2, m->down23, nullptr, nullptr, e3, 1, T, g.H[1], g.W[1], g.C[1], g.C[1] / 2, nb(g.C[1] / 2)));
    if ((rc = run_stage(m, m->enc3, e3, nullptr, T, g.H[2], g.W[2], g.C[2], g.heads[2], g.Hp[2]))) return rc;
    RUN(m, "conv3x3_down", ira::launch_conv3x3(st, e3, m->down34, nullptr, nullptr, l4, 1, T, g.H[2], g.W[2], g.C[2], g.C[2] / 2, nb(g.C[2] / 2)));
    if ((rc = run_stage(m, m->lat, l4, nullptr, T, g.H[3], g.W[3], g.C[3], g.heads[3], g.Hp[3]))) return rc;                         // :257

    RUN(m, "conv3x3_up", ira::launch_conv3x3(st, l4, m->up43, nullptr, nullptr, m->u3, 2, T, g.H[3], g.W[3], g.C[3], g.C[3] * 2, nb(g.C[3] * 2)));         // :259
    RUN(m, "reduce_1x1", ira::launch_pw_nres(st, m->u3, g.C[2], e3, g.C[2], true, m->red3, 0, m->red3_b, nullptr, d3, g.C[2], nb(g.C[2]), T, (int)g.P[2]));   // :260-261
    if ((rc = run_stage(m, m->dec3, d3, nullptr, T, g.H[2], g.W[2], g.C[2], g.heads[2], g.Hp[2]))) return rc;
    RUN(m, "conv3x3_up", ira::launch_conv3x3(st, d3, m->up32, nullptr, nullptr, m->u2, 2, T, g.H[2], g.W[2], g.C[2], g.C[2] * 2, nb(g.C[2] * 2)));
    RUN(m, "reduce_1x1", ira::launch_pw_nres(st, m->u2, g.C[1], e2, g.C[1], true, m->red2, 0, m->red2_b, nullptr, d2, g.C[1], nb(g.C[1]), T, (int)g.P[1]));
    if ((rc = run_stage(m, m->dec2, d2, m->ad2, T, g.H[1], g.W[1], g.C[1], g.heads[1], g.Hp[1]))) return rc;
    RUN(m, "conv3x3_up", ira::launch_conv3x3(st, d2, m->up21, nullptr, nullptr, m->u1, 2, T, g.H[1], g.W[1], g.C[1], g.C[1] * 2, nb(g.C[1] * 2)));
    RUN(m, "concat", ira::launch_concat2(st, m->u1, g.C[0], e1, g.C[0], d1, (long)T * g.P[0]));                                        // :270
    float* a1 = m->ad1;
    if ((rc = run_stage(m, m->dec1, d1, a1, T, g.H[0], g.W[0], g.C[1], g.heads[0], g.Hp[1]))) return rc;                              // :271
    a1 = (d1 == m->xd1) ? m->ad1 : m->xd1;
    if ((rc = run_stage(m, m->refine, d1, a1, T, g.H[0], g.W[0], g.C[1], g.heads[0], g.Hp[1]))) return rc;                            // :273
    RUN(m, "conv3x3_out", ira::launch_conv3x3(st, d1, m->out_w, m->out_b, in, out, 3, T, g.H[0], g.W[0], g.C[1], c.out_channels, nb(c.out_channels)));   // :281
    return IRA_OK;
}


// =============================================================================================
// NAFNet (NAFNet_arch.py:249-361): weights, workspace, forward schedule
int naf_levels(const ira_nafnet_config& c) { return c.n_levels; }

bool naf_load_block(ira_model* m, const std::string& p, int c, NBlk& b) {
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
int naf_ensure_workspace(ira_model* m, int T, int H, int W) {
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
int naf_run_block(ira_model* m, const NBlk& b, float* x, int T, int H, int W, int c) {
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

// =============================================================================================
// DnCNN (BASELINE config 1; no reference code: a build-defined plumbing model, see ira_amd/dncnn.py):
// conv3x3(in->nc)+ReLU, (nb-2) x [conv3x3(nc->nc) + BN (folded by the packer) + ReLU], conv3x3(nc->in); out = x - net(x).
bool dn_bind_weights(ira_model* m) {
    const ira_dncnn_config& c = m->dcfg;
    m->err.clear();
    m->dn_first_w = (const float*)wptr(m, "first_w", true, (size_t)c.nc * c.in_nc * 9 * 4);
    m->dn_first_b = (const float*)wptr(m, "first_b", true, (size_t)c.nc * 4);
    m->dn_last_w = wptr(m, "last_w", true, (size_t)2 * 9 * up32(c.in_nc) * c.nc * 2);
    m->dn_last_b = (const float*)wptr(m, "last_b", true, (size_t)up32(c.in_nc) * 4);
    if (!m->dn_first_w || !m->dn_first_b || !m->dn_last_w || !m->dn_last_b) return false;
    m->dn_mid_w.assign(c.nb - 2, nullptr); m->dn_mid_b.assign(c.nb - 2, nullptr);
    for (int i = 0; i < c.nb - 2; ++i) {
        m->dn_mid_w[i] = wptr(m, "mid." + std::to_string(i) + "_w", true, (size_t)2 * 9 * up32(c.nc) * c.nc * 2);
        m->dn_mid_b[i] = (const float*)wptr(m, "mid." + std::to_string(i) + "_b", true, (size_t)up32(c.nc) * 4);
        if (!m->dn_mid_w[i] || !m->dn_mid_b[i]) return false;
    }
    return true;
}

int dn_forward_enqueue(ira_model* m, const float* in, int T, int H, int W, float* out) {
    const ira_dncnn_config& c = m->dcfg;
    if (T <= 0 || H <= 0 || W <= 0) return m->fail(IRA_ERR_ARG, "forward: T=%d H=%d W=%d must be positive", T, H, W);
    if (!(m->ws && T <= m->ws_T && H == m->ws_H && W == m->ws_W)) {
        if (m->ws) { HIP_TRY(m, hipStreamSynchronize(m->st)); HIP_TRY(m, hipFree(m->ws)); m->ws = nullptr; }
        const size_t one = al256((size_t)T * H * W * c.nc * 4);
        hipError_t e = hipMalloc((void**)&m->ws, 2 * one);
        if (e != hipSuccess) { m->ws = nullptr; m->ws_T = 0; return m->fail(IRA_ERR_OOM, "workspace hipMalloc(%zu): %s", 2 * one, hipGetErrorString(e)); }
        m->ws_bytes = 2 * one; m->ws_T = T; m->ws_H = H; m->ws_W = W;
        m->nx.assign(2, nullptr);
        m->nx[0] = (float*)m->ws; m->nx[1] = (float*)(m->ws + one);
    }
    hipStream_t st = m->st;
    const int nbc = up32(c.nc) / 32;
    float *a = m->nx[0], *b = m->nx[1];
    RUN(m, "dncnn_first3x3_relu", ira::launch_embed(st, in, m->dn_first_w, m->dn_first_b, a, T, H, W, c.in_nc, c.nc, 1));
    for (int i = 0; i < c.nb - 2; ++i) {
        RUN(m, "dncnn_conv3x3_bn_relu", ira::launch_conv3x3(st, a, m->dn_mid_w[i], m->dn_mid_b[i], nullptr, b, 4, T, H, W, c.nc, c.nc, nbc));
        std::swap(a, b);
    }
    // weights and bias of the last conv are negated by the packer: out = x + (-(conv)) = x - noise
    RUN(m, "dncnn_last3x3_residual", ira::launch_conv3x3(st, a, m->dn_last_w, m->dn_last_b, in, out, 3, T, H, W, c.nc, c.in_nc, up32(c.in_nc) / 32));
    return IRA_OK;
}

// =============================================================================================
// SwinIR (network_swinir.py:618-840): weights, workspace, forward schedule
struct SwGeo { int C, Cp, hid, hidp, qp, heads_max; };
SwGeo sw_geo(const ira_swinir_config& c) {
    SwGeo g;
    g.C = c.embed_dim; g.Cp = up32(c.embed_dim); g.hid = c.mlp_hidden; g.hidp = up32(c.mlp_hidden); g.qp = up32(c.embed_dim / 4);
    g.heads_max = 0;
    for (int i = 0; i < c.n_layers; ++i) g.heads_max = std::max(g.heads_max, c.num_heads[i]);
    return g;
}

bool sw_load_conv(ira_model* m, const std::string& p, bool three, const SwGeo& g, SConv& c) {
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

int sw_ensure_workspace(ira_model* m, int T, int Hp, int Wp) {
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
int sw_resi_conv(ira_model* m, const SConv& cv, const float* x, const float* skip, float* y, int T, int H, int W, const char* fam) {
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
        HIP_TRY(m, hipMemcpyAsync(m->sw_xb, m->sw_xa, (size_t)T * P * g.Cp * 4, hipMemcpyDeviceToDevice, st));     // RSTB keeps its input (:482)
        float* x = m->sw_xb;
        for (int k = 0; k < c.depths[i]; ++k) {
            const SBlk& b = m->sw_blocks[i][k];
            const int shift = (k % 2 == 0) ? 0 : ws / 2;                                                             // :385-386
            RUN(m, k_[0], ira::launch_pw_kres(st, x, true, 2, g.Cp, b.qkv_w, b.qkv_b, m->sw_qkv, 3 * Kh, 3 * Kh, 3 * heads, T, P, 1e-5f, g.C, 0));
            RUN(m, k_[1], ira::launch_swin_attn(st, m->sw_qkv, b.bias_tab, m->sw_att, T, Hp, Wp, heads, ws, shift));
            RUN(m, k_[2], ira::launch_pw_nres(st, m->sw_att, Kh, nullptr, 0, false, b.proj_w, 0, b.proj_b, x, x, g.Cp, g.Cp / 32, T, P));
            RUN(m, k_[3], ira::launch_pw_kres(st, x, true, 2, g.Cp, b.fc1_w, b.fc1_b, m->sw_hid, g.hidp, g.hidp, g.hidp / 32, T, P, 1e-5f, g.C, 1));
            RUN(m, k_[4], ira::launch_pw_nres(st, m->sw_hid, g.hidp, nullptr, 0, false, b.fc2_w, 0, b.fc2_b, x, x, g.Cp, g.Cp / 32, T, P));
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

int forward_enqueue(ira_model* m, const float* in, int T, int H, int W, float* out) {
    if (m->arch == 3) return sw_forward_enqueue(m, in, T, H, W, out);
    if (m->arch == 2) return dn_forward_enqueue(m, in, T, H, W, out);
    return m->arch == 1 ? naf_forward_enqueue(m, in, T, H, W, out) : restormer_forward_enqueue(m, in, T, H, W, out);
}

struct Plan { int Hp, Wp, th, tw, sh, sw, n_rows, n_cols; int flip = 0, scale = 1; };

int make_plan(int h, int w, int tile, int overlap, Plan& p, std::string& err) {
    if (h <= 0 || w <= 0) { err = "image size must be positive"; return IRA_ERR_ARG; }
    // demo.py:123-126
    const int H = ((h + 8) / 8) * 8, W = ((w + 8) / 8) * 8;
    p.Hp = (h % 8 != 0) ? H : h;
    p.Wp = (w % 8 != 0) ? W : w;
    if (p.Hp - h >= h || p.Wp - w >= w) { err = "reflect padding needs pad < image size (F.pad 'reflect')"; return IRA_ERR_ARG; }
    if (tile <= 0) {   // demo.py:129-131
        p.th = p.Hp; p.tw = p.Wp; p.sh = p.Hp; p.sw = p.Wp; p.n_rows = 1; p.n_cols = 1;
        return IRA_OK;
    }
    const int t = std::min(tile, std::min(p.Hp, p.Wp));                      // demo.py:135
    if (t % 8 != 0) { err = "tile size should be multiple of 8"; return IRA_ERR_ARG; }   // demo.py:136
    const int stride = t - overlap;                                           // demo.py:139
    if (stride <= 0) { err = "tile_overlap must be smaller than the tile (range() step must be positive)"; return IRA_ERR_ARG; }
    p.th = p.tw = t; p.sh = p.sw = stride;
    // len(list(range(0, Hp - t, stride))) + 1   (demo.py:140-141)
    p.n_rows = (p.Hp - t + stride - 1) / stride + 1;
    p.n_cols = (p.Wp - t + stride - 1) / stride + 1;
    return IRA_OK;
}

// The harness of the model's architecture: Restormer's demo.py rule above, or SwinIR's (SwinIR/inference.py:76-80,294-310):
// ALWAYS pad to (h // ws + 1) * ws by flip-concatenation, tile a multiple of the window, outputs scaled by `upscale`.
int plan_for(const ira_model* m, int h, int w, int tile, int overlap, Plan& p, std::string& err) {
    if (m->arch != 3) return make_plan(h, w, tile, overlap, p, err);
    if (h <= 0 || w <= 0) { err = "image size must be positive"; return IRA_ERR_ARG; }
    const int ws = m->scfg.window_size;
    p.flip = 1; p.scale = m->scfg.upscale;
    p.Hp = (h / ws + 1) * ws; p.Wp = (w / ws + 1) * ws;
    if (p.Hp - h > h || p.Wp - w > w) { err = "flip padding needs pad <= image size (torch.cat([img, flip(img)]))"; return IRA_ERR_ARG; }
    if (tile <= 0) { p.th = p.Hp; p.tw = p.Wp; p.sh = p.Hp; p.sw = p.Wp; p.n_rows = 1; p.n_cols = 1; return IRA_OK; }
    const int t = std::min(tile, std::min(p.Hp, p.Wp));                       // inference.py:301
    if (t % ws != 0) { err = "tile size should be a multiple of window_size"; return IRA_ERR_ARG; }   // :302
    const int stride = t - overlap;
    if (stride <= 0) { err = "tile_overlap must be smaller than the tile (range() step must be positive)"; return IRA_ERR_ARG; }
    p.th = p.tw = t; p.sh = p.sw = stride;
    p.n_rows = (p.Hp - t + stride - 1) / stride + 1;                          // :306-308
    p.n_cols = (p.Wp - t + stride - 1) / stride + 1;
    return IRA_OK;
}

int blend_enqueue(ira_model* m, const float* tiles, int h, int w, int c, const Plan& p, uint8_t* out) {
    const int s = p.scale;
    RUN(m, "tile_blend", ira::launch_blend(m->st, tiles, h * s, w * s, c, p.Hp * s, p.Wp * s, p.th * s, p.tw * s, p.sh * s, p.sw * s,
                                           p.n_rows, p.n_cols, out));
    return IRA_OK;
}

int default_batch(const ira_model* m, const Plan& p) {
    if (m->max_batch > 0) return m->max_batch;
    // up to 64 tiles of 512x512 per launch-batch (about 1 GB of workspace per such tile; a 4K frame is 40):
    // the level-3/4 kernels only fill the chip with tens of tiles in flight (353 ms per 4K frame at 40 vs 397 at 8)
    const long px = (long)p.th * p.tw;
    long b = (64L * 512 * 512) / std::max(px, 1L);
    return (int)std::max(1L, std::min(b, 256L));
}

int restore_tiles_enqueue(ira_model* m, const uint8_t* frame_dev, int h, int w, int c, const Plan& p, int first,
                          int n, float* tiles_out) {
    const int batch = default_batch(m, p);
    const size_t tile_elems = (size_t)p.th * p.tw * c;
    const size_t tile_elems_out = tile_elems * p.scale * p.scale;
    int rc = ensure_scratch(m, al256((size_t)batch * tile_elems * 4));
    if (rc != IRA_OK) return rc;
    float* tin = (float*)m->hs;
    for (int t0 = 0; t0 < n; t0 += batch) {
        const int nb = std::min(batch, n - t0);
        RUN(m, "tile_extract", ira::launch_extract_tiles(m->st, frame_dev, h, w, c, p.Hp, p.Wp, p.th, p.tw, p.sh, p.sw,
                                                         p.n_rows, p.n_cols, first + t0, nb, tin, p.flip));
        rc = forward_enqueue(m, tin, nb, p.th, p.tw, tiles_out + (size_t)t0 * tile_elems_out);
        if (rc != IRA_OK) return rc;
    }
    return IRA_OK;
}

}  // namespace

// device checks, blob table of contents, stream / events, weight upload: common to every architecture
static int create_common(int device, const void* blob, size_t bytes, ira_model** out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return gfail(IRA_ERR_NO_DEVICE, "no HIP device visible: libira_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return gfail(IRA_ERR_ARG, "device %d out of range (0..%d)", device, ndev - 1);
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, device) != hipSuccess || strncmp(pr.gcnArchName, "gfx950", 6) != 0)
        return gfail(IRA_ERR_NO_DEVICE, "device %d is '%s', this library carries gfx950 code only", device, pr.gcnArchName);

    // parse blob table of contents
    if (bytes < sizeof(BlobHeader)) return gfail(IRA_ERR_WEIGHTS, "weight blob too small");
    const BlobHeader* hd = (const BlobHeader*)blob;
    if (memcmp(hd->magic, "IRAWGT1", 8) != 0) return gfail(IRA_ERR_WEIGHTS, "weight blob: bad magic");
    const size_t toc_end = sizeof(BlobHeader) + (size_t)hd->n_entries * sizeof(TocEntry);
    if (toc_end > bytes) return gfail(IRA_ERR_WEIGHTS, "weight blob: truncated table of contents");

    ira_model* m = new ira_model();
    m->device = device;
    const TocEntry* te = (const TocEntry*)((const char*)blob + sizeof(BlobHeader));
    for (uint32_t i = 0; i < hd->n_entries; ++i) {
        if (te[i].offset + te[i].nbytes > bytes || te[i].offset % 256 != 0) {
            delete m; return gfail(IRA_ERR_WEIGHTS, "weight blob: entry %u out of bounds / misaligned", i);
        }
        char nm[113]; memcpy(nm, te[i].name, 112); nm[112] = 0;
        m->toc[nm] = {(size_t)te[i].offset, (size_t)te[i].nbytes};
    }
    auto bail = [&](int code, const std::string& why) {
        g_err = why;
        if (m->wblob) hipFree(m->wblob);
        if (m->st) hipStreamDestroy(m->st);
        if (m->ev0) hipEventDestroy(m->ev0);
        if (m->ev1) hipEventDestroy(m->ev1);
        if (m->pev0) hipEventDestroy(m->pev0);
        if (m->pev1) hipEventDestroy(m->pev1);
        delete m;
        return code;
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(IRA_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    if ((e = hipStreamCreateWithFlags(&m->st, hipStreamNonBlocking)) != hipSuccess) return bail(IRA_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    if (hipEventCreate(&m->ev0) != hipSuccess || hipEventCreate(&m->ev1) != hipSuccess || hipEventCreate(&m->pev0) != hipSuccess ||
        hipEventCreate(&m->pev1) != hipSuccess)
        return bail(IRA_ERR_HIP, "hipEventCreate failed");
    if ((e = hipMalloc((void**)&m->wblob, bytes)) != hipSuccess) { m->wblob = nullptr; return bail(IRA_ERR_OOM, std::string("weights hipMalloc: ") + hipGetErrorString(e)); }
    m->wbytes = bytes;
    if ((e = hipMemcpy(m->wblob, blob, bytes, hipMemcpyHostToDevice)) != hipSuccess) return bail(IRA_ERR_HIP, std::string("weights upload: ") + hipGetErrorString(e));
    *out = m;
    return IRA_OK;
}

// =============================================================================================
extern "C" {

int ira_device_count(int* count) {
    if (!count) return gfail(IRA_ERR_ARG, "ira_device_count: count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return gfail(IRA_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, i) == hipSuccess && strncmp(pr.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    *count = ok;
    return IRA_OK;
}

int ira_model_create(int device, const ira_restormer_config* cfg, const void* blob, size_t bytes, ira_model_t** out) {
    if (!cfg || !blob || !out) return gfail(IRA_ERR_ARG, "ira_model_create: NULL argument");
    *out = nullptr;
    if (cfg->dual_pixel_task) return gfail(IRA_ERR_ARG, "dual_pixel_task is not supported (not dispatched by the agent)");
    if (cfg->dim <= 0 || cfg->dim % 16 != 0) return gfail(IRA_ERR_ARG, "dim must be a positive multiple of 16, got %d", cfg->dim);
    if (cfg->inp_channels != cfg->out_channels || (cfg->inp_channels != 1 && cfg->inp_channels != 3))
        return gfail(IRA_ERR_ARG, "inp_channels == out_channels in {1,3} required, got %d/%d", cfg->inp_channels, cfg->out_channels);
    for (int l = 0; l < 4; ++l) {
        const int C = cfg->dim << l;
        if (cfg->heads[l] <= 0 || C % cfg->heads[l] != 0 || (C / cfg->heads[l]) % 16 != 0)
            return gfail(IRA_ERR_ARG, "level %d: C=%d heads=%d needs C/heads to be a multiple of 16", l, C, cfg->heads[l]);
        const int c = C / cfg->heads[l];
        if (c != 16 && c != 32 && c != 48 && c != 96) return gfail(IRA_ERR_ARG, "level %d: per-head channels %d not in {16,32,48,96}", l, c);
        if (cfg->hidden[l] <= 0) return gfail(IRA_ERR_ARG, "hidden[%d] must be positive", l);
    }
    {   // decoder level 1 / refinement: 2*dim channels on heads[0]
        const int c = 2 * cfg->dim / cfg->heads[0];
        if ((2 * cfg->dim) % cfg->heads[0] != 0 || (c != 16 && c != 32 && c != 48 && c != 96))
            return gfail(IRA_ERR_ARG, "decoder level 1: per-head channels %d not in {16,32,48,96}", c);
    }
    ira_model* m = nullptr;
    int rc = create_common(device, blob, bytes, &m);
    if (rc != IRA_OK) return rc;
    m->arch = 0; m->cfg = *cfg;
    if (!bind_weights(m)) {
        g_err = m->err.empty() ? "weight blob: missing entries" : m->err;
        ira_model_destroy(m);
        return IRA_ERR_WEIGHTS;
    }
    *out = m;
    return IRA_OK;
}


int ira_nafnet_create(int device, const ira_nafnet_config* cfg, const void* blob, size_t bytes, ira_model_t** out) {
    if (!cfg || !blob || !out) return gfail(IRA_ERR_ARG, "ira_nafnet_create: NULL argument");
    *out = nullptr;
    if (cfg->img_channel != 1 && cfg->img_channel != 3) return gfail(IRA_ERR_ARG, "img_channel must be 1 or 3, got %d", cfg->img_channel);
    if (cfg->width < 16 || (cfg->width & (cfg->width - 1)) != 0)
        return gfail(IRA_ERR_ARG, "width must be a power of two >= 16, got %d", cfg->width);
    if (cfg->n_levels < 1 || cfg->n_levels > IRA_NAFNET_MAX_LEVELS || (cfg->width << cfg->n_levels) > 2048)
        return gfail(IRA_ERR_ARG, "n_levels=%d out of range (1..%d, width*2^levels <= 2048)", cfg->n_levels, IRA_NAFNET_MAX_LEVELS);
    if (cfg->middle_blk_num < 0) return gfail(IRA_ERR_ARG, "middle_blk_num must be >= 0");
    for (int i = 0; i < cfg->n_levels; ++i)
        if (cfg->enc_blk_nums[i] < 0 || cfg->dec_blk_nums[i] < 0) return gfail(IRA_ERR_ARG, "block counts must be >= 0");
    ira_model* m = nullptr;
    int rc = create_common(device, blob, bytes, &m);
    if (rc != IRA_OK) return rc;
    m->arch = 1; m->ncfg = *cfg;
    m->cfg.inp_channels = m->cfg.out_channels = cfg->img_channel;      // what the harness entry points check
    if (!naf_bind_weights(m)) {
        g_err = m->err.empty() ? "weight blob: missing entries" : m->err;
        ira_model_destroy(m);
        return IRA_ERR_WEIGHTS;
    }
    *out = m;
    return IRA_OK;
}

int ira_dncnn_create(int device, const ira_dncnn_config* cfg, const void* blob, size_t bytes, ira_model_t** out) {
    if (!cfg || !blob || !out) return gfail(IRA_ERR_ARG, "ira_dncnn_create: NULL argument");
    *out = nullptr;
    if (cfg->in_nc != 1 && cfg->in_nc != 3) return gfail(IRA_ERR_ARG, "in_nc must be 1 or 3, got %d", cfg->in_nc);
    if (cfg->nc < 16 || cfg->nc % 16 != 0) return gfail(IRA_ERR_ARG, "nc must be a positive multiple of 16, got %d", cfg->nc);
    if (cfg->nb < 2) return gfail(IRA_ERR_ARG, "nb must be >= 2, got %d", cfg->nb);
    ira_model* m = nullptr;
    int rc = create_common(device, blob, bytes, &m);
    if (rc != IRA_OK) return rc;
    m->arch = 2; m->dcfg = *cfg;
    m->cfg.inp_channels = m->cfg.out_channels = cfg->in_nc;
    if (!dn_bind_weights(m)) {
        g_err = m->err.empty() ? "weight blob: missing entries" : m->err;
        ira_model_destroy(m);
        return IRA_ERR_WEIGHTS;
    }
    *out = m;
    return IRA_OK;
}

int ira_swinir_create(int device, const ira_swinir_config* cfg, const void* blob, size_t bytes, ira_model_t** out) {
    if (!cfg || !blob || !out) return gfail(IRA_ERR_ARG, "ira_swinir_create: NULL argument");
    *out = nullptr;
    if (cfg->in_chans != 1 && cfg->in_chans != 3) return gfail(IRA_ERR_ARG, "in_chans must be 1 or 3, got %d", cfg->in_chans);
    if (cfg->window_size != 7 && cfg->window_size != 8) return gfail(IRA_ERR_ARG, "window_size must be 7 or 8 (the reference's configs), got %d", cfg->window_size);
    if (cfg->embed_dim < 16 || cfg->embed_dim % 4 != 0 || cfg->embed_dim > 512) return gfail(IRA_ERR_ARG, "embed_dim %d unsupported (multiple of 4, 16..512)", cfg->embed_dim);
    if (cfg->n_layers < 1 || cfg->n_layers > IRA_SWINIR_MAX_LAYERS) return gfail(IRA_ERR_ARG, "n_layers out of range");
    for (int i = 0; i < cfg->n_layers; ++i) {
        if (cfg->depths[i] < 1 || cfg->num_heads[i] < 1 || cfg->embed_dim % cfg->num_heads[i] != 0 || cfg->embed_dim / cfg->num_heads[i] > 32)
            return gfail(IRA_ERR_ARG, "layer %d: depth %d heads %d (head_dim must divide embed_dim and be <= 32)", i, cfg->depths[i], cfg->num_heads[i]);
    }
    if (cfg->mlp_hidden < 1) return gfail(IRA_ERR_ARG, "mlp_hidden must be positive");
    if (cfg->upsampler < 0 || cfg->upsampler > 2) return gfail(IRA_ERR_ARG, "upsampler must be 0 (''), 1 (pixelshuffle) or 2 (nearest+conv)");
    if (cfg->upsampler == 0 ? cfg->upscale != 1 : (cfg->upscale != 1 && cfg->upscale != 2 && cfg->upscale != 4))
        return gfail(IRA_ERR_ARG, "upscale %d unsupported for this upsampler (1 without; 1, 2 or 4 with)", cfg->upscale);
    if (!(cfg->img_range > 0.f)) return gfail(IRA_ERR_ARG, "img_range must be positive");
    ira_model* m = nullptr;
    int rc = create_common(device, blob, bytes, &m);
    if (rc != IRA_OK) return rc;
    m->arch = 3; m->scfg = *cfg;
    m->cfg.inp_channels = m->cfg.out_channels = cfg->in_chans;
    if (!sw_bind_weights(m)) {
        g_err = m->err.empty() ? "weight blob: missing entries" : m->err;
        ira_model_destroy(m);
        return IRA_ERR_WEIGHTS;
    }
    *out = m;
    return IRA_OK;
}

int ira_model_destroy(ira_model_t* m) {
    if (!m) return IRA_OK;
    hipSetDevice(m->device);
    if (m->st) hipStreamSynchronize(m->st);
    if (m->ws) hipFree(m->ws);
    if (m->hs) hipFree(m->hs);
    if (m->wblob) hipFree(m->wblob);
    if (m->ev0) hipEventDestroy(m->ev0);
    if (m->ev1) hipEventDestroy(m->ev1);
    if (m->pev0) hipEventDestroy(m->pev0);
    if (m->pev1) hipEventDestroy(m->pev1);
    if (m->st) hipStreamDestroy(m->st);
    delete m;
    return IRA_OK;
}

const char* ira_last_error(const ira_model_t* m) { return m ? m->err.c_str() : g_err.c_str(); }

#define ENTER(m)                                                     \
    if (!(m)) return gfail(IRA_ERR_ARG, "%s: model is NULL", __func__); \
    std::lock_guard<std::mutex> lock_((m)->mu);                      \
    HIP_TRY((m), hipSetDevice((m)->device))

int ira_sync(ira_model_t* m) {
    ENTER(m);
    HIP_TRY(m, hipStreamSynchronize(m->st));
    return IRA_OK;
}

int ira_forward_device(ira_model_t* m, const float* in, int B, int H, int W, float* out) {
    ENTER(m);
    if (!in || !out) return m->fail(IRA_ERR_ARG, "ira_forward_device: NULL buffer");
    int rc = forward_enqueue(m, in, B, H, W, out);
    if (rc != IRA_OK) return rc;
    HIP_TRY(m, hipStreamSynchronize(m->st));
    return IRA_OK;
}

int ira_forward(ira_model_t* m, const float* in, int B, int C, int H, int W, float* out) {
    ENTER(m);
    if (!in || !out) return m->fail(IRA_ERR_ARG, "ira_forward: NULL buffer");
    const int ic = m->cfg.inp_channels;      // set by every create function
    if (C != ic) return m->fail(IRA_ERR_ARG, "ira_forward: C=%d but the model has inp_channels=%d", C, ic);
    if (B <= 0 || H <= 0 || W <= 0 || (m->arch == 0 && ((H & 7) || (W & 7))))
        return m->fail(IRA_ERR_ARG, "ira_forward: B=%d H=%d W=%d (H, W multiples of 8)", B, H, W);
    const size_t n = (size_t)B * C * H * W;
    const int s = m->arch == 3 ? m->scfg.upscale : 1;             // SwinIR SR: the output is (B, C, H*s, W*s)
    const size_t no = n * s * s;
    int rc = ensure_scratch(m, al256(n * 4) + 2 * al256(no * 4));
    if (rc != IRA_OK) return rc;
    float* a = (float*)m->hs;                       // NCHW staging (in, then out)
    float* nhwc_out = (float*)(m->hs + al256(no * 4));
    float* nhwc_in = (float*)(m->hs + 2 * al256(no * 4));
    HIP_TRY(m, hipMemcpyAsync(a, in, n * 4, hipMemcpyHostToDevice, m->st));
    RUN(m, "layout", ira::launch_nchw_to_nhwc(m->st, a, nhwc_in, B, C, H, W));
    rc = forward_enqueue(m, nhwc_in, B, H, W, nhwc_out);
    if (rc != IRA_OK) return rc;
    RUN(m, "layout", ira::launch_nhwc_to_nchw(m->st, nhwc_out, a, B, C, H * s, W * s));
    HIP_TRY(m, hipMemcpyAsync(out, a, no * 4, hipMemcpyDeviceToHost, m->st));
    HIP_TRY(m, hipStreamSynchronize(m->st));
    return IRA_OK;
}

int ira_tile_plan(int h, int w, int tile, int overlap, int* Hp, int* Wp, int* th, int* tw, int* n_rows, int* n_cols) {
    Plan p; std::string err;
    int rc = make_plan(h, w, tile, overlap, p, err);
    if (rc != IRA_OK) return gfail(rc, "%s", err.c_str());
    if (Hp) *Hp = p.Hp; if (Wp) *Wp = p.Wp; if (th) *th = p.th; if (tw) *tw = p.tw;
    if (n_rows) *n_rows = p.n_rows; if (n_cols) *n_cols = p.n_cols;
    return IRA_OK;
}

int ira_model_tile_plan(ira_model_t* m, int h, int w, int tile, int overlap, int* Hp, int* Wp, int* th, int* tw, int* n_rows,
                        int* n_cols, int* scale) {
    if (!m) return gfail(IRA_ERR_ARG, "ira_model_tile_plan: model is NULL");
    std::lock_guard<std::mutex> lk(m->mu);
    Plan p; std::string err;
    int rc = plan_for(m, h, w, tile, overlap, p, err);
    if (rc != IRA_OK) return m->fail(rc, "%s", err.c_str());
    if (Hp) *Hp = p.Hp; if (Wp) *Wp = p.Wp; if (th) *th = p.th; if (tw) *tw = p.tw;
    if (n_rows) *n_rows = p.n_rows; if (n_cols) *n_cols = p.n_cols; if (scale) *scale = p.scale;
    return IRA_OK;
}

int ira_restore_tiles_device(ira_model_t* m, const uint8_t* frame, int h, int w, int c, int tile, int overlap,
                             int first_tile, int n_tiles, float* tiles_out) {
    ENTER(m);
    if (!frame || !tiles_out) return m->fail(IRA_ERR_ARG, "ira_restore_tiles_device: NULL buffer");
    if (c != m->cfg.inp_channels) return m->fail(IRA_ERR_ARG, "image has %d channels, model wants %d", c, m->cfg.inp_channels);
    Plan p; std::string err;
    int rc = plan_for(m, h, w, tile, overlap, p, err);
    if (rc != IRA_OK) return m->fail(rc, "%s", err.c_str());
    if (first_tile < 0 || n_tiles < 0 || first_tile + n_tiles > p.n_rows * p.n_cols)
        return m->fail(IRA_ERR_ARG, "tile range [%d,%d) outside the %dx%d plan", first_tile, first_tile + n_tiles, p.n_rows, p.n_cols);
    if (n_tiles == 0) return IRA_OK;
    rc = restore_tiles_enqueue(m, frame, h, w, c, p, first_tile, n_tiles, tiles_out);
    if (rc != IRA_OK) return rc;
    HIP_TRY(m, hipStreamSynchronize(m->st));
    return IRA_OK;
}

int ira_blend_tiles_device(ira_model_t* m, const float* tiles, int h, int w, int c, int tile, int overlap, uint8_t* out) {
    ENTER(m);
    if (!tiles || !out) return m->fail(IRA_ERR_ARG, "ira_blend_tiles_device: NULL buffer");
    Plan p; std::string err;
    int rc = plan_for(m, h, w, tile, overlap, p, err);
    if (rc != IRA_OK) return m->fail(rc, "%s", err.c_str());
    if ((rc = blend_enqueue(m, tiles, h, w, c, p, out)) != IRA_OK) return rc;
    HIP_TRY(m, hipStreamSynchronize(m->st));
    return IRA_OK;
}

int ira_restore_u8(ira_model_t* m, const uint8_t* in, int h, int w, int c, int tile, int overlap, uint8_t* out) {
    if (!m) return gfail(IRA_ERR_ARG, "ira_restore_u8: model is NULL");
    Plan p; std::string err;
    {
        std::lock_guard<std::mutex> lk(m->mu);
        if (!in || !out) return m->fail(IRA_ERR_ARG, "ira_restore_u8: NULL buffer");
        if (c != m->cfg.inp_channels) return m->fail(IRA_ERR_ARG, "image has %d channels, model wants %d", c, m->cfg.inp_channels);
        int rc = plan_for(m, h, w, tile, overlap, p, err);
        if (rc != IRA_OK) return m->fail(rc, "%s", err.c_str());
    }
    ENTER(m);
    const int nt = p.n_rows * p.n_cols;
    const size_t fbytes = (size_t)h * w * c;
    const size_t obytes = fbytes * p.scale * p.scale;
    const size_t tbytes = (size_t)nt * p.th * p.tw * c * 4 * p.scale * p.scale;
    uint8_t *fin = nullptr, *fout = nullptr; float* tiles = nullptr;
    auto cleanup = [&]() { if (fin) hipFree(fin); if (fout) hipFree(fout); if (tiles) hipFree(tiles); };
    hipError_t e;
    if ((e = hipMalloc((void**)&fin, fbytes)) != hipSuccess || (e = hipMalloc((void**)&fout, obytes)) != hipSuccess ||
        (e = hipMalloc((void**)&tiles, tbytes)) != hipSuccess) {
        cleanup();
        return m->fail(IRA_ERR_OOM, "ira_restore_u8: hipMalloc: %s", hipGetErrorString(e));
    }
    int rc = IRA_OK;
    do {
        if ((e = hipMemcpyAsync(fin, in, fbytes, hipMemcpyHostToDevice, m->st)) != hipSuccess) { rc = m->fail(IRA_ERR_HIP, "upload: %s", hipGetErrorString(e)); break; }
        if ((rc = restore_tiles_enqueue(m, fin, h, w, c, p, 0, nt, tiles)) != IRA_OK) break;
        if ((rc = blend_enqueue(m, tiles, h, w, c, p, fout)) != IRA_OK) break;
        if ((e = hipMemcpyAsync(out, fout, obytes, hipMemcpyDeviceToHost, m->st)) != hipSuccess) { rc = m->fail(IRA_ERR_HIP, "download: %s", hipGetErrorString(e)); break; }
        if ((e = hipStreamSynchronize(m->st)) != hipSuccess) { rc = m->fail(IRA_ERR_HIP, "sync: %s", hipGetErrorString(e)); break; }
    } while (0);
    if (rc != IRA_OK) hipStreamSynchronize(m->st);
    cleanup();
    return rc;
}

int ira_restore_padcrop_u8(ira_model_t* m, const uint8_t* in, int h, int w, int c, int tile_size, int tile_pad, int padding_size,
                           uint8_t* out) {
    ENTER(m);
    if (!in || !out) return m->fail(IRA_ERR_ARG, "ira_restore_padcrop_u8: NULL buffer");
    if (c != m->cfg.inp_channels) return m->fail(IRA_ERR_ARG, "image has %d channels, model wants %d", c, m->cfg.inp_channels);
    if (h <= 0 || w <= 0 || padding_size <= 0 || tile_pad < 0) return m->fail(IRA_ERR_ARG, "bad size / padding arguments");
    // pre_process (xrestormer_model.py:53-63)
    const int mph = (h % padding_size) ? padding_size - h % padding_size : 0;
    const int mpw = (w % padding_size) ? padding_size - w % padding_size : 0;
    if (mph >= h || mpw >= w) return m->fail(IRA_ERR_ARG, "reflect padding needs pad < image size (F.pad 'reflect')");
    const int Hp = h + mph, Wp = w + mpw;
    // tile_process (:65-133); tile_size <= 0: the untiled branch of test() (:138-150)
    struct D { int v[6]; };
    std::map<std::pair<int, int>, std::vector<D>> classes;
    const int ts = tile_size > 0 ? tile_size : std::max(Hp, Wp);
    const int tiles_x = (Wp + ts - 1) / ts, tiles_y = (Hp + ts - 1) / ts;
    for (int y = 0; y < tiles_y; ++y)
        for (int x = 0; x < tiles_x; ++x) {
            const int sx = x * ts, ex = std::min(sx + ts, Wp), sy = y * ts, ey = std::min(sy + ts, Hp);
            const int sxp = std::max(sx - tile_pad, 0), exp_ = std::min(ex + tile_pad, Wp);
            const int syp = std::max(sy - tile_pad, 0), eyp = std::min(ey + tile_pad, Hp);
            D d{{syp, sxp, sy, sx, ey - sy, ex - sx}};
            classes[{eyp - syp, exp_ - sxp}].push_back(d);
        }
    if (m->arch == 0)
        for (auto& kv : classes)
            if ((kv.first.first & 7) || (kv.first.second & 7))
                return m->fail(IRA_ERR_ARG, "tile %dx%d is not a multiple of 8: choose tile_size, tile_pad, padding_size as multiples of 8",
                               kv.first.first, kv.first.second);
    const size_t fbytes = (size_t)h * w * c;
    uint8_t *fin = nullptr, *fout = nullptr; float *tin = nullptr, *tout = nullptr; int* ddev = nullptr;
    auto cleanup = [&]() { if (fin) hipFree(fin); if (fout) hipFree(fout); if (tin) hipFree(tin); if (tout) hipFree(tout); if (ddev) hipFree(ddev); };
    size_t max_batch_elems = 0, max_desc = 0;
    std::map<std::pair<int, int>, int> batch_of;
    for (auto& kv : classes) {
        const long px = (long)kv.first.first * kv.first.second;
        int b = m->max_batch > 0 ? m->max_batch : (int)std::max(1L, std::min(256L, (64L * 512 * 512) / std::max(px, 1L)));
        b = std::min<int>(b, (int)kv.second.size());
        batch_of[kv.first] = b;
        max_batch_elems = std::max(max_batch_elems, (size_t)b * px * c);
        max_desc = std::max(max_desc, kv.second.size());
    }
    hipError_t e;
    if ((e = hipMalloc((void**)&fin, fbytes)) != hipSuccess || (e = hipMalloc((void**)&fout, fbytes)) != hipSuccess ||
        (e = hipMalloc((void**)&tin, max_batch_elems * 4)) != hipSuccess || (e = hipMalloc((void**)&tout, max_batch_elems * 4)) != hipSuccess ||
        (e = hipMalloc((void**)&ddev, max_desc * sizeof(D))) != hipSuccess) {
        cleanup();
        return m->fail(IRA_ERR_OOM, "ira_restore_padcrop_u8: hipMalloc: %s", hipGetErrorString(e));
    }
    int rc = IRA_OK;
    do {
        if ((e = hipMemcpyAsync(fin, in, fbytes, hipMemcpyHostToDevice, m->st)) != hipSuccess) { rc = m->fail(IRA_ERR_HIP, "upload: %s", hipGetErrorString(e)); break; }
        for (auto& kv : classes) {
            const int th = kv.first.first, tw = kv.first.second, n = (int)kv.second.size(), batch = batch_of[kv.first];
            // the descriptor buffer is reused by the next class: drain the stream before overwriting it
            if ((e = hipStreamSynchronize(m->st)) != hipSuccess) { rc = m->fail(IRA_ERR_HIP, "sync: %s", hipGetErrorString(e)); break; }
            if ((e = hipMemcpy(ddev, kv.second.data(), (size_t)n * sizeof(D), hipMemcpyHostToDevice)) != hipSuccess) { rc = m->fail(IRA_ERR_HIP, "descriptor upload: %s", hipGetErrorString(e)); break; }
            int max_core = 0;
            for (const D& d : kv.second) max_core = std::max(max_core, d.v[4] * d.v[5]);
            for (int t0 = 0; t0 < n && rc == IRA_OK; t0 += batch) {
                const int nbt = std::min(batch, n - t0);
                rc = run(m, "padcrop_extract", [&]() { return ira::launch_extract_list(m->st, fin, h, w, c, th, tw, ddev + (size_t)t0 * 6, nbt, tin); });
                if (rc != IRA_OK) break;
                if ((rc = forward_enqueue(m, tin, nbt, th, tw, tout)) != IRA_OK) break;
                rc = run(m, "padcrop_scatter", [&]() { return ira::launch_scatter_cores(m->st, tout, th, tw, c, ddev + (size_t)t0 * 6, nbt, max_core, h, w, fout); });
            }
            if (rc != IRA_OK) break;
        }
        if (rc != IRA_OK) break;
        if ((e = hipMemcpyAsync(out, fout, fbytes, hipMemcpyDeviceToHost, m->st)) != hipSuccess) { rc = m->fail(IRA_ERR_HIP, "download: %s", hipGetErrorString(e)); break; }
        if ((e = hipStreamSynchronize(m->st)) != hipSuccess) { rc = m->fail(IRA_ERR_HIP, "sync: %s", hipGetErrorString(e)); break; }
    } while (0);
    if (rc != IRA_OK) hipStreamSynchronize(m->st);
    cleanup();
    return rc;
}

int ira_set_max_batch(ira_model_t* m, int max_tiles) {
    ENTER(m);
    if (max_tiles < 0) return m->fail(IRA_ERR_ARG, "max_tiles must be >= 0");
    m->max_batch = max_tiles;
    return IRA_OK;
}

int ira_timer_start(ira_model_t* m) {
    ENTER(m);
    HIP_TRY(m, hipEventRecord(m->ev0, m->st));
    return IRA_OK;
}

int ira_timer_stop(ira_model_t* m, float* ms) {
    ENTER(m);
    if (!ms) return m->fail(IRA_ERR_ARG, "ira_timer_stop: NULL");
    HIP_TRY(m, hipEventRecord(m->ev1, m->st));
    HIP_TRY(m, hipEventSynchronize(m->ev1));
    HIP_TRY(m, hipEventElapsedTime(ms, m->ev0, m->ev1));
    return IRA_OK;
}

int ira_profile_enable(ira_model_t* m, int on) {
    ENTER(m);
    m->profile = on != 0;
    if (on) m->prof.clear();
    return IRA_OK;
}

const char* ira_profile_json(ira_model_t* m) {
    if (!m) return "{}";
    std::lock_guard<std::mutex> lk(m->mu);
    std::string s = "{";
    bool first = true;
    for (auto& kv : m->prof) {
        char buf[256];
        snprintf(buf, sizeof buf, "%s\"%s\": {\"launches\": %ld, \"ms\": %.6f}", first ? "" : ", ", kv.first.c_str(), kv.second.launches, kv.second.ms);
        s += buf; first = false;
    }
    s += "}";
    m->prof_json = s;
    return m->prof_json.c_str();
}

int ira_memory_bytes(ira_model_t* m, size_t* bytes) {
    ENTER(m);
    if (!bytes) return m->fail(IRA_ERR_ARG, "NULL");
    *bytes = m->wbytes + m->ws_bytes + m->hs_bytes;
    return IRA_OK;
}

// ------------------------------------------------------------------ single-kernel entry points
#define OP_BODY(m, fam, call)             \
    ENTER(m);                             \
    RUN(m, fam, call);                    \
    HIP_TRY(m, hipStreamSynchronize(m->st)); \
    return IRA_OK

int ira_op_pw_kres(ira_model_t* m, const void* x, int in_f32, int ln_mode, int K, const void* W, const float* bias,
                   void* y, int y_stride, int n_store, int NB, int T, int P) {
    OP_BODY(m, "op", ira::launch_pw_kres(m->st, x, in_f32 != 0, ln_mode, K, W, bias, y, y_stride, n_store, NB, T, P));
}

int ira_op_pw_kres_eps(ira_model_t* m, const void* x, int in_f32, int ln_mode, int K, const void* W, const float* bias,
                       void* y, int y_stride, int n_store, int NB, int T, int P, float eps) {
    OP_BODY(m, "op", ira::launch_pw_kres(m->st, x, in_f32 != 0, ln_mode, K, W, bias, y, y_stride, n_store, NB, T, P, eps));
}

int ira_op_pw_nres(ira_model_t* m, const void* x0, int K0, const void* x1, int K1, int in_f32, const void* W,
                   long w_tile_stride, const float* bias, const float* resid, float* y, int N, int NB, int T, int P) {
    OP_BODY(m, "op", ira::launch_pw_nres(m->st, x0, K0, x1, K1, in_f32 != 0, W, w_tile_stride, bias, resid, y, N, NB, T, P));
}

int ira_op_dwconv_qkv(ira_model_t* m, const void* in, const float* w, const float* bias, void* qk, void* v, int T,
                      int H, int W, int C) {
    OP_BODY(m, "op", ira::launch_dwconv_qkv(m->st, in, w, bias, qk, v, T, H, W, C));
}

int ira_op_dwconv_gate(ira_model_t* m, const void* in, const float* w, const float* bias, void* out, int T, int H,
                       int W, int Hp) {
    OP_BODY(m, "op", ira::launch_dwconv_gate(m->st, in, w, bias, out, T, H, W, Hp));
}

int ira_op_gram(ira_model_t* m, const void* qk, float* part, int T, int nblk, int C, int heads, int S) {
    OP_BODY(m, "op", ira::launch_gram(m->st, qk, part, T, nblk, C, heads, S));
}

int ira_op_attn_mix(ira_model_t* m, const float* part, const float* wo, const float* temp, void* M, int T, int C,
                    int heads, int S, int NB) {
    if (!m) return gfail(IRA_ERR_ARG, "ira_op_attn_mix: model is NULL");
    if (heads <= 0 || C % heads != 0) return gfail(IRA_ERR_ARG, "ira_op_attn_mix: bad C/heads");
    {
        std::lock_guard<std::mutex> lk(m->mu);
        hipSetDevice(m->device);
        int rc = ensure_scratch(m, al256((size_t)T * C * (C / heads) * 4));
        if (rc != IRA_OK) return rc;
    }
    OP_BODY(m, "op", ira::launch_attn_mix(m->st, part, wo, temp, (float*)m->hs, M, T, C, heads, S, NB));
}

int ira_op_conv3x3(ira_model_t* m, const float* in, const void* W, const float* bias, const float* skip, float* out,
                   int mode, int T, int H, int Wd, int K, int N, int NB) {
    OP_BODY(m, "op", ira::launch_conv3x3(m->st, in, W, bias, skip, out, mode, T, H, Wd, K, N, NB));
}

int ira_op_ffn_fused(ira_model_t* m, const float* x, float* xo, int ln_mode, int C, const void* Win, const float* bin,
                     const float* dww, const float* dwb, const void* Wout, const float* bout, int T, int H, int W, int Hp) {
    OP_BODY(m, "op", ira::launch_ffn_fused(m->st, x, xo, ln_mode, C, Win, bin, dww, dwb, Wout, bout, T, H, W, Hp));
}

int ira_op_qkv_fused(ira_model_t* m, const float* x, int ln_mode, int C, const void* Wq, const float* bq, const float* dww,
                     const float* dwb, void* qk, void* v, int T, int H, int W) {
    OP_BODY(m, "op", ira::launch_qkv_fused(m->st, x, ln_mode, C, Wq, bq, dww, dwb, qk, v, T, H, W));
}

int ira_set_fused_ffn(ira_model_t* m, int on) {
    ENTER(m);
    m->use_fused_ffn = on != 0;
    return IRA_OK;
}

int ira_op_embed(ira_model_t* m, const float* in, const float* w, const float* bias, float* out, int T, int H, int W,
                 int ic, int C) {
    OP_BODY(m, "op", ira::launch_embed(m->st, in, w, bias, out, T, H, W, ic, C));
}

int ira_op_concat2(ira_model_t* m, const float* a, int Ca, const float* b, int Cb, float* out, long npx) {
    OP_BODY(m, "op", ira::launch_concat2(m->st, a, Ca, b, Cb, out, npx));
}

int ira_op_naf_pw(ira_model_t* m, int inmode, int outmode, const void* x, int K, const void* W, long w_tile_stride,
                  const float* bias, const float* resid, float* y, int N, int NBtot, int T, int Hg, int Wg, const float* scale) {
    OP_BODY(m, "op", ira::launch_naf_pw(m->st, inmode, outmode, x, K, W, w_tile_stride, bias, resid, y, N, NBtot, T, Hg, Wg, scale));
}

int ira_op_naf_dw_sg(ira_model_t* m, const void* in, const float* w, const float* bias, void* out, float* part, int T, int H,
                     int W, int c, int* blocks_per_image) {
    if (blocks_per_image) *blocks_per_image = ira::naf_dw_sg_blocks(H, W, c);
    OP_BODY(m, "op", ira::launch_naf_dw_sg(m->st, in, w, bias, out, part, T, H, W, c));
}

int ira_op_swin_attn(ira_model_t* m, const void* qkv, const float* bias, void* out, int T, int H, int W, int heads, int ws, int shift) {
    OP_BODY(m, "op", ira::launch_swin_attn(m->st, qkv, bias, out, T, H, W, heads, ws, shift));
}

int ira_op_ln_affine(ira_model_t* m, const float* x, const float* g, const float* b, float* y, long npx, int C, int Cp, float eps) {
    OP_BODY(m, "op", ira::launch_ln_affine(m->st, x, g, b, y, npx, C, Cp, eps));
}

int ira_op_pw_kres_ex(ira_model_t* m, const void* x, int in_f32, int ln_mode, int K, const void* W, const float* bias, void* y,
                      int y_stride, int n_store, int NB, int T, int P, float eps, int ln_n, int act) {
    OP_BODY(m, "op", ira::launch_pw_kres(m->st, x, in_f32 != 0, ln_mode, K, W, bias, y, y_stride, n_store, NB, T, P, eps, ln_n, act));
}

int ira_op_naf_tail(ira_model_t* m, const void* g, const float* scale, float* x, int c, const void* W3, const float* b3,
                    const void* W4, const float* b4, const void* W5, const float* b5, int T, int P) {
    OP_BODY(m, "op", ira::launch_naf_tail(m->st, g, scale, x, c, W3, b3, W4, b4, W5, b5, T, P));
}

int ira_op_naf_sca(ira_model_t* m, const float* part, int blocks_per_image, long npx, const float* wsca, const float* bsca,
                   float* s, int T, int c) {
    OP_BODY(m, "op", ira::launch_naf_sca(m->st, part, blocks_per_image, npx, wsca, bsca, s, T, c));
}

}  // extern "C"
