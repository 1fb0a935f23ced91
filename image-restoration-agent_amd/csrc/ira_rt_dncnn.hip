// DnCNN (build-defined, BASELINE config 1): weight binding and forward schedule.
#include "ira_model.h"

namespace ira_rt {

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

}  // namespace ira_rt
