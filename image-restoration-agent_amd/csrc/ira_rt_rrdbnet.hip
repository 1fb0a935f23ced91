// RRDBNet / Real-ESRGAN generator (reference AgenticIR/executor/super_resolution/tools/DiffBIR/utils/realesrgan/rrdbnet.py:74-182,
// built by set_realesrgan, realesrganer.py:304-339): weight binding and forward schedule.
//
// A ResidualDenseBlock (:74-105) convolves torch.cat((x, x1, ..)) five times.  Here a block owns ONE buffer of nf + 4 * gc floats per pixel:
// conv k reads the first nf + (k - 1) * gc channels of every pixel and writes its gc outputs (bias + LeakyReLU 0.2 in the epilogue) into the
// next channel slice of the same pixels (disjoint 128-byte lines, so no hazard), and conv5 (weights and bias pre-scaled by 0.2) writes
// x5 * 0.2 + x into the first nf channels of the next block's buffer.  No concatenation ever moves data.
#include "ira_model.h"

namespace ira_rt {

namespace {

struct RCtx { ira_model* m; hipStream_t st; char* base = nullptr; size_t off = 0; bool dry = true;
    float* alloc(size_t floats) { const size_t o = off; off += al256(floats * sizeof(float)); return dry ? nullptr : (float*)(base + o); } };

#define RRUN(c, fam, call) do { if (!(c).dry) RUN((c).m, fam, call); } while (0)
#define RTRY(expr) do { int rc__ = (expr); if (rc__ != IRA_OK) return rc__; } while (0)

int conv(RCtx& c, const std::string& name, const float* in, int K, int ldin, float* out, int N, int ldout, int mode, const float* skip,
         int T, int H, int W, const char* fam) {
    const void* w = wptr(c.m, name + "_w", true, (size_t)2 * 9 * up32(N) * K * 2);
    const float* b = (const float*)wptr(c.m, name + "_b", true, (size_t)up32(N) * 4);
    if (!w || !b) return c.m->fail(IRA_ERR_WEIGHTS, "%s", c.m->err.c_str());
    RRUN(c, fam, ira::launch_conv3x3(c.st, in, w, b, skip, out, mode, T, H, W, K, N, up32(N) / 32, 0.2f, ldin, ldout));
    return IRA_OK;
}

int schedule(RCtx& c, const float* in, int T, int H, int W, float* out) {
    const ira_rrdbnet_config& g = c.m->rcfg;
    const int f = g.scale == 2 ? 2 : (g.scale == 1 ? 4 : 1);                       // :167-172
    const int h = H / f, w = W / f, ic = g.num_in_ch * f * f, nf = g.num_feat, gc = g.num_grow_ch, LD = nf + 4 * gc;
    const long px = (long)T * h * w;
    const float* x = in;
    if (f > 1) {
        float* xs = c.alloc(px * ic);
        RRUN(c, "rrdb_pixel_unshuffle", ira::launch_pixel_unshuffle(c.st, in, xs, T, H, W, g.num_in_ch, f));
        x = xs;
    }
    float* feat = c.alloc(px * nf);
    if (ic % 16 == 0) {
        RTRY(conv(c, "first", x, ic, ic, feat, nf, nf, 0, nullptr, T, h, w, "rrdb_conv_first"));
    } else {
        const float* fw = (const float*)wptr(c.m, "first_w", true, (size_t)nf * ic * 9 * 4);
        const float* fb = (const float*)wptr(c.m, "first_b", true, (size_t)nf * 4);
        if (!fw || !fb) return c.m->fail(IRA_ERR_WEIGHTS, "%s", c.m->err.c_str());
        RRUN(c, "rrdb_conv_first", ira::launch_embed(c.st, x, fw, fb, feat, T, h, w, ic, nf, 0));
    }
    float* D[3] = {c.alloc(px * LD), c.alloc(px * LD), c.alloc(px * LD)};
    RRUN(c, "rrdb_copy", ira::launch_axpy_strided(c.st, nullptr, 0, feat, nf, D[0], LD, 0.f, nf, px));
    for (int i = 0; i < g.num_block; ++i) {                                         // RRDB (:108-128)
        float* src[3] = {D[0], D[1], D[2]};
        float* dst[3] = {D[1], D[2], D[1]};
        for (int j = 0; j < 3; ++j) {
            const std::string p = "b" + std::to_string(i) + ".r" + std::to_string(j + 1) + ".c";
            for (int k = 0; k < 4; ++k)
                RTRY(conv(c, p + std::to_string(k + 1), src[j], nf + k * gc, LD, src[j] + nf + k * gc, gc, LD, 4, nullptr, T, h, w, "rrdb_dense_conv3x3_lrelu"));
            RTRY(conv(c, p + "5", src[j], LD, LD, dst[j], nf, LD, 3, src[j], T, h, w, "rrdb_dense_conv5_res"));
        }
        RRUN(c, "rrdb_block_residual", ira::launch_axpy_strided(c.st, D[1], LD, D[0], LD, D[1], LD, 0.2f, nf, px));   // out * 0.2 + x (:128)
        std::swap(D[0], D[1]);
    }
    float* t0 = c.alloc(px * nf);
    RTRY(conv(c, "body", D[0], nf, LD, t0, nf, nf, 3, feat, T, h, w, "rrdb_conv_body_res"));                            // :175-176
    float* u1 = c.alloc(px * 4 * nf);
    float* f1 = c.alloc(px * 4 * nf);
    float* u2 = c.alloc(px * 16 * nf);
    float* f2 = c.alloc(px * 16 * nf);
    RRUN(c, "rrdb_nearest2x", ira::launch_nearest2x(c.st, t0, u1, T, h, w, nf));
    RTRY(conv(c, "up1", u1, nf, nf, f1, nf, nf, 4, nullptr, T, 2 * h, 2 * w, "rrdb_conv_up_lrelu"));                    // :178
    RRUN(c, "rrdb_nearest2x", ira::launch_nearest2x(c.st, f1, u2, T, 2 * h, 2 * w, nf));
    RTRY(conv(c, "up2", u2, nf, nf, f2, nf, nf, 4, nullptr, T, 4 * h, 4 * w, "rrdb_conv_up_lrelu"));                    // :179
    RTRY(conv(c, "hr", f2, nf, nf, u2, nf, nf, 4, nullptr, T, 4 * h, 4 * w, "rrdb_conv_hr_lrelu"));                     // :180
    RTRY(conv(c, "last", u2, nf, nf, out, g.num_out_ch, g.num_out_ch, 0, nullptr, T, 4 * h, 4 * w, "rrdb_conv_last"));
    return IRA_OK;
}

}  // namespace

bool rr_bind_weights(ira_model* m) {
    RCtx c{m, m->st};
    m->err.clear();
    const int f = m->rcfg.scale == 2 ? 2 : (m->rcfg.scale == 1 ? 4 : 1);
    return schedule(c, nullptr, 1, 4 * f, 4 * f, nullptr) == IRA_OK;
}

int rr_forward_enqueue(ira_model* m, const float* in, int T, int H, int W, float* out) {
    const int f = m->rcfg.scale == 2 ? 2 : (m->rcfg.scale == 1 ? 4 : 1);
    if (T <= 0 || H <= 0 || W <= 0) return m->fail(IRA_ERR_ARG, "forward: T=%d H=%d W=%d must be positive", T, H, W);
    if (H % f || W % f) return m->fail(IRA_ERR_ARG, "forward: H=%d W=%d must be multiples of %d for scale %d (pixel_unshuffle asserts, rrdbnet.py:66)", H, W, f, m->rcfg.scale);
    if (!(m->ws && T == m->ws_T && H == m->ws_H && W == m->ws_W)) {
        RCtx d{m, m->st};
        int rc = schedule(d, nullptr, T, H, W, nullptr);
        if (rc != IRA_OK) return rc;
        if (m->ws) { HIP_TRY(m, hipStreamSynchronize(m->st)); HIP_TRY(m, hipFree(m->ws)); m->ws = nullptr; }
        hipError_t e = hipMalloc((void**)&m->ws, d.off);
        if (e != hipSuccess) { m->ws = nullptr; m->ws_T = 0; return m->fail(IRA_ERR_OOM, "workspace hipMalloc(%zu): %s", d.off, hipGetErrorString(e)); }
        m->ws_bytes = d.off; m->ws_T = T; m->ws_H = H; m->ws_W = W;
    }
    RCtx c{m, m->st};
    c.base = m->ws; c.dry = false;
    return schedule(c, in, T, H, W, out);
}

}  // namespace ira_rt
