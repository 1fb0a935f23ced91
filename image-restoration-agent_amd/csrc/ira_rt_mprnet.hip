// MPRNet (reference AgenticIR/executor/denoising/tools/MPRNet/Deblurring/MPRNet.py -- the Denoising and Deraining copies
// differ only in the constructor defaults): weight binding and the three-stage forward schedule.
//
// Layout: fp32 NHWC feature streams with the channel count padded to a multiple of 16 (padded channels are zero).  The top / bottom
// halves of stage 1 and 2 (:265-277) are the SAME memory read as 2T images of half the height, so only the left | right split and
// the width concatenation of stage-1 features (:294-295) move data.  Every 3x3 convolution is k_conv3x3 (PReLU = its LeakyReLU epilogue;
// the single nn.PReLU() parameter is shared by every CAB, :242); every 1x1 is k_naf_pw on hi+lo split fp32 pixels.  DownSample (:155-163,
// bilinear x0.5 = the 2x2 mean, then 1x1) is ONE space-to-depth contraction with W/4 per tap; UpSample (bilinear x2 then bias-free 1x1)
// runs the 1x1 at the low resolution first (both are linear and the interpolation weights sum to 1, so they commute) and the ORSNet
// chains up -> 1x1 (-> up -> 1x1) -> 1x1 (:229-231) are folded by the packer into one matrix per source.
#include "ira_model.h"

namespace ira_rt {

namespace {

inline int up16(int n) { return (n + 15) / 16 * 16; }

struct MCtx {
    ira_model* m;
    hipStream_t st;
    char* base = nullptr;
    size_t off = 0;
    bool dry = true;              // dry: resolve weights and size the arena, launch nothing
    float *t1 = nullptr, *rs = nullptr, *part = nullptr, *z = nullptr;   // CAB scratch
    float* alloc(size_t floats) {
        const size_t o = off;
        off += al256(floats * sizeof(float));
        return dry ? nullptr : (float*)(base + o);
    }
};

#define MRUN(c, fam, call)            \
    do {                              \
        if (!(c).dry) RUN((c).m, fam, call); \
    } while (0)
#define MTRY(expr)                    \
    do {                              \
        int rc__ = (expr);            \
        if (rc__ != IRA_OK) return rc__; \
    } while (0)

int need(MCtx& c, const std::string& name, size_t bytes, const void** out) {
    *out = wptr(c.m, name, true, bytes);
    if (*out == nullptr) return c.m->fail(IRA_ERR_WEIGHTS, "%s", c.m->err.c_str());
    return IRA_OK;
}

// 3x3 convolution cin -> cout on padded-channel tensors; mode 0 plain, 3 + skip, 4 PReLU
int conv3(MCtx& c, const float* x, const std::string& name, int kp, int cout, int n_stride, int T, int H, int W, int mode, const float* skip,
          float* out, const char* fam) {
    const void* w;
    MTRY(need(c, name, (size_t)2 * 9 * up32(cout) * kp * 2, &w));
    MRUN(c, fam, ira::launch_conv3x3(c.st, x, w, nullptr, skip, out, mode, T, H, W, kp, n_stride, up32(cout) / 32, c.m->mcfg.prelu));
    return IRA_OK;
}

// 1x1 convolution (s2d: the 2x2-mean DownSample contraction on the (H, W) OUTPUT grid) + optional residual
int pw(MCtx& c, const float* x, const std::string& name, int cin, int cout, int T, int H, int W, const float* resid, float* out, bool s2d,
       const char* fam) {
    const int kp = (s2d ? 4 : 1) * up16(cin), np = up16(cout);
    const void* w;
    MTRY(need(c, name, (size_t)2 * up32(np) * kp * 2, &w));
    MRUN(c, fam, ira::launch_naf_pw(c.st, s2d ? 6 : 5, 0, x, kp, w, 0, nullptr, resid, out, np, up32(np) / 32, T, H, W));
    return IRA_OK;
}

// CAB (:42-57): out = CA(conv(PReLU(conv(x)))) + x; out may alias x
int cab(MCtx& c, const float* x, const std::string& p, int C, int T, int H, int W, float* out) {
    const int Cp = up16(C), Cr = C / c.m->mcfg.reduction;
    const void *w1, *w2;
    MTRY(need(c, p + ".du0", (size_t)Cr * C * 4, &w1));
    MTRY(need(c, p + ".du2", (size_t)C * Cr * 4, &w2));
    MTRY(conv3(c, x, p + ".c0", Cp, C, Cp, T, H, W, 4, nullptr, c.t1, "mpr_cab_conv3x3_prelu"));
    MTRY(conv3(c, c.t1, p + ".c2", Cp, C, Cp, T, H, W, 0, nullptr, c.rs, "mpr_cab_conv3x3"));
    MRUN(c, "mpr_cab_ca", ira::launch_mpr_ca(c.st, c.rs, c.part, (const float*)w1, (const float*)w2, c.z, T, (long)H * W, C, Cp, Cr));
    MRUN(c, "mpr_cab_scale_add", ira::launch_mpr_scale_add(c.st, c.rs, c.z, x, out, T, (long)H * W, Cp));
    return IRA_OK;
}

struct Feats { float* l[3] = {nullptr, nullptr, nullptr}; };

// Encoder.forward (:104-121); x is consumed (level 1 runs in place)
int encoder(MCtx& c, float* x, const std::string& p, const int* ch, int T, int H, int W, const Feats* enc, const Feats* dec, Feats& outs) {
    for (int l = 0; l < 3; ++l) {
        const int h = H >> l, w = W >> l, C = ch[l];
        const std::string lv = std::to_string(l + 1);
        for (int k = 0; k < 2; ++k) MTRY(cab(c, x, p + ".encoder_level" + lv + "." + std::to_string(k), C, T, h, w, x));
        if (enc != nullptr) {
            MTRY(pw(c, enc->l[l], p + ".csff_enc" + lv, C, C, T, h, w, x, x, false, "mpr_csff_1x1"));
            MTRY(pw(c, dec->l[l], p + ".csff_dec" + lv, C, C, T, h, w, x, x, false, "mpr_csff_1x1"));
        }
        outs.l[l] = x;
        if (l < 2) {
            float* xn = c.alloc((size_t)T * (h / 2) * (w / 2) * up16(ch[l + 1]));
            MTRY(pw(c, x, p + ".down" + lv + std::to_string(l + 2), C, ch[l + 1], T, h / 2, w / 2, nullptr, xn, true, "mpr_down_mean2x2_1x1"));
            x = xn;
        }
    }
    return IRA_OK;
}

// Decoder.forward (:141-151); the encoder features are preserved
int decoder(MCtx& c, const Feats& e, const std::string& p, const int* ch, int T, int H, int W, Feats& d) {
    for (int l = 2; l >= 0; --l) {
        const int h = H >> l, w = W >> l, C = ch[l], Cp = up16(C);
        const std::string lv = std::to_string(l + 1);
        d.l[l] = c.alloc((size_t)T * h * w * Cp);
        if (l == 2) {
            MTRY(cab(c, e.l[2], p + ".decoder_level3.0", C, T, h, w, d.l[2]));
        } else {
            float* lo = c.alloc((size_t)T * (h / 2) * (w / 2) * Cp);
            MTRY(pw(c, d.l[l + 1], p + ".up" + std::to_string(l + 2) + lv, ch[l + 1], C, T, h / 2, w / 2, nullptr, lo, false, "mpr_up_1x1"));
            MTRY(cab(c, e.l[l], p + ".skip_attn" + lv, C, T, h, w, d.l[l]));
            MRUN(c, "mpr_bilinear_up2_add", ira::launch_mpr_up2(c.st, lo, d.l[l], d.l[l], T, h / 2, w / 2, Cp));
            MTRY(cab(c, d.l[l], p + ".decoder_level" + lv + ".0", C, T, h, w, d.l[l]));
        }
        MTRY(cab(c, d.l[l], p + ".decoder_level" + lv + ".1", C, T, h, w, d.l[l]));
    }
    return IRA_OK;
}

// SAM (:61-74) + shallow_feat{i} + concat{..} (:302-307 / :327-332): the next stage's input features.  img_out: the stage image.
int sam_concat(MCtx& c, const float* x, const float* ximg, const std::string& sp, const std::string& shallow, const std::string& cat_name,
               int n, int cout, int T, int H, int W, float** next, float** img_out) {
    const int np = up16(n);
    const long px = (long)T * H * W;
    const void *w3, *ws;
    MTRY(need(c, sp + ".conv3", (size_t)n * 3 * 4, &w3));
    MTRY(need(c, shallow + ".0", (size_t)np * 3 * 9 * 4, &ws));
    float* x1 = c.alloc(px * np);
    float* y8 = c.alloc(px * 8);
    float* sf = c.alloc(px * np);
    float* img = c.alloc(px * 3);
    float* sh = c.alloc(px * np);
    float* cat = c.alloc(px * 2 * np);
    float* nx = c.alloc(px * up16(cout));
    MTRY(pw(c, x, sp + ".conv1", n, n, T, H, W, nullptr, x1, false, "mpr_sam_1x1"));
    {   // conv2: n -> 3, stored in rows of 8
        const void* w;
        MTRY(need(c, sp + ".conv2", (size_t)2 * 32 * np * 2, &w));
        MRUN(c, "mpr_sam_1x1", ira::launch_naf_pw(c.st, 5, 0, x, np, w, 0, nullptr, nullptr, y8, 8, 1, T, H, W));
    }
    MRUN(c, "mpr_sam_gate", ira::launch_mpr_sam(c.st, x1, y8, ximg, (const float*)w3, x, sf, img, n, np, px));
    MRUN(c, "mpr_shallow3x3", ira::launch_embed(c.st, ximg, (const float*)ws, nullptr, sh, T, H, W, 3, np, 0));
    MTRY(cab(c, sh, shallow + ".1", n, T, H, W, sh));
    MRUN(c, "mpr_concat", ira::launch_concat2(c.st, sh, np, sf, np, cat, px));
    MTRY(conv3(c, cat, cat_name, 2 * np, cout, up16(cout), T, H, W, 0, nullptr, nx, "mpr_concat_conv3x3"));
    *next = nx;
    *img_out = img;
    return IRA_OK;
}

int schedule(MCtx& c, const float* in, int T, int H, int W, float* out) {
    const ira_mprnet_config& g = c.m->mcfg;
    const int n = g.n_feat, su = g.scale_unetfeats, so = g.scale_orsnetfeats;
    const int ch[3] = {n, n + su, n + 2 * su}, co = n + so;
    const int np = up16(n), cop = up16(co);
    const long px = (long)T * H * W;
    // CAB scratch: the widest full-resolution tensor
    const size_t big = (size_t)px * std::max(np, cop);
    c.t1 = c.alloc(big);
    c.rs = c.alloc(big);
    c.part = c.alloc((size_t)4 * T * 2048 * 256);
    c.z = c.alloc((size_t)4 * T * 256);

    // ---- stage 1 (:279-300): four quadrants = 4T images of (H/2, W/2)
    const int H2 = H / 2, W2 = W / 2;
    Feats f1, r1;
    {
        const void* ws;
        MTRY(need(c, "shallow_feat1.0", (size_t)np * 3 * 9 * 4, &ws));
        float* q = c.alloc(px * 3);
        float* sh = c.alloc(px * np);
        MRUN(c, "mpr_split_lr", ira::launch_mpr_lr(c.st, const_cast<float*>(in), q, 2 * T, H2, W2, 3, 0));
        MRUN(c, "mpr_shallow3x3", ira::launch_embed(c.st, q, (const float*)ws, nullptr, sh, 4 * T, H2, W2, 3, np, 0));
        MTRY(cab(c, sh, "shallow_feat1.1", n, 4 * T, H2, W2, sh));
        Feats fq;
        MTRY(encoder(c, sh, "stage1_encoder", ch, 4 * T, H2, W2, nullptr, nullptr, fq));
        for (int l = 0; l < 3; ++l) {
            const int h = H2 >> l, w = W2 >> l, Cp = up16(ch[l]);
            f1.l[l] = c.alloc((size_t)2 * T * h * 2 * w * Cp);
            MRUN(c, "mpr_concat_lr", ira::launch_mpr_lr(c.st, f1.l[l], fq.l[l], 2 * T, h, w, Cp, 1));
        }
        MTRY(decoder(c, f1, "stage1_decoder", ch, 2 * T, H2, W, r1));
    }
    // ---- stage 2 (:302-322): top / bottom halves = 2T images of (H/2, W)
    float *x2, *img1;
    MTRY(sam_concat(c, r1.l[0], in, "sam12", "shallow_feat2", "concat12", n, n, 2 * T, H2, W, &x2, &img1));
    Feats f2, r2;
    MTRY(encoder(c, x2, "stage2_encoder", ch, 2 * T, H2, W, &f1, &r1, f2));
    MTRY(decoder(c, f2, "stage2_decoder", ch, T, H, W, r2));       // the height concatenation (:319) is the same memory as T images
    // ---- stage 3 (:324-345)
    float *x, *img2;
    MTRY(sam_concat(c, r2.l[0], in, "sam23", "shallow_feat3", "concat23", n, co, T, H, W, &x, &img2));
    float* orb_r = c.alloc(px * cop);
    float* lo2 = c.alloc((size_t)T * (H / 2) * (W / 2) * cop);
    float* lo4 = c.alloc((size_t)T * (H / 4) * (W / 4) * cop);
    for (int i = 1; i <= 3; ++i) {
        const std::string p = "stage3_orsnet.orb" + std::to_string(i) + ".body.";
        const float* src = x;
        for (int k = 0; k < g.num_cab; ++k) { MTRY(cab(c, src, p + std::to_string(k), co, T, H, W, orb_r)); src = orb_r; }
        MTRY(conv3(c, src, p + std::to_string(g.num_cab), cop, co, cop, T, H, W, 3, x, x, "mpr_orb_conv3x3_skip"));
        for (int kind = 0; kind < 2; ++kind) {
            const Feats& f = kind == 0 ? f2 : r2;
            const std::string wn = std::string("stage3_orsnet.") + (kind == 0 ? "enc" : "dec") + std::to_string(i);
            if (i == 1) {
                MTRY(pw(c, f.l[0], wn, ch[0], co, T, H, W, x, x, false, "mpr_ors_1x1"));
            } else if (i == 2) {
                MTRY(pw(c, f.l[1], wn, ch[1], co, T, H / 2, W / 2, nullptr, lo2, false, "mpr_ors_1x1"));
                MRUN(c, "mpr_bilinear_up2_add", ira::launch_mpr_up2(c.st, lo2, x, x, T, H / 2, W / 2, cop));
            } else {
                MTRY(pw(c, f.l[2], wn, ch[2], co, T, H / 4, W / 4, nullptr, lo4, false, "mpr_ors_1x1"));
                MRUN(c, "mpr_bilinear_up2", ira::launch_mpr_up2(c.st, lo4, nullptr, lo2, T, H / 4, W / 4, cop));
                MRUN(c, "mpr_bilinear_up2_add", ira::launch_mpr_up2(c.st, lo2, x, x, T, H / 2, W / 2, cop));
            }
        }
    }
    MTRY(conv3(c, x, "tail", cop, g.out_c, g.out_c, T, H, W, 3, in, out, "mpr_tail_conv3x3_skip"));
    return IRA_OK;
}

}  // namespace

bool mpr_bind_weights(ira_model* m) {
    MCtx c{m, m->st};
    c.dry = true;
    m->err.clear();
    return schedule(c, nullptr, 1, 8, 8, nullptr) == IRA_OK;
}

int mpr_forward_enqueue(ira_model* m, const float* in, int T, int H, int W, float* out) {
    if (T <= 0 || H <= 0 || W <= 0) return m->fail(IRA_ERR_ARG, "forward: T=%d H=%d W=%d must be positive", T, H, W);
    if ((H % 8) || (W % 8)) return m->fail(IRA_ERR_ARG, "forward: H=%d W=%d must be multiples of 8 (two patch splits and two 2x levels)", H, W);
    if (!(m->ws && T == m->ws_T && H == m->ws_H && W == m->ws_W)) {
        MCtx d{m, m->st};
        d.dry = true;
        int rc = schedule(d, nullptr, T, H, W, nullptr);
        if (rc != IRA_OK) return rc;
        if (m->ws) { HIP_TRY(m, hipStreamSynchronize(m->st)); HIP_TRY(m, hipFree(m->ws)); m->ws = nullptr; }
        hipError_t e = hipMalloc((void**)&m->ws, d.off);
        if (e != hipSuccess) { m->ws = nullptr; m->ws_T = 0; return m->fail(IRA_ERR_OOM, "workspace hipMalloc(%zu): %s", d.off, hipGetErrorString(e)); }
        m->ws_bytes = d.off; m->ws_T = T; m->ws_H = H; m->ws_W = W;
    }
    MCtx c{m, m->st};
    c.base = m->ws;
    c.dry = false;
    return schedule(c, in, T, H, W, out);
}

}  // namespace ira_rt
