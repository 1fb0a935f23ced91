// Internal header of libira_hip.so's runtime: the model handle and the helpers shared by the per-architecture schedules
// (ira_rt_restormer.hip, ira_rt_nafnet.hip, ira_rt_dncnn.hip, ira_rt_swinir.hip) and the C ABI (ira_runtime.hip).
#pragma once
#include "../../include/ira.h"
#include "ira_kernels.h"
#include "ira_comm.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace ira_rt {

struct Blk {   // one TransformerBlock (restormer_arch.py:137-150), device pointers into the weight blob
    const void* qkv_w = nullptr;  const float* qkv_b = nullptr;
    const float* dw1_w = nullptr; const float* dw1_b = nullptr;
    const float* temp = nullptr;  const float* wo = nullptr; const float* wo_b = nullptr;
    const void* pin_w = nullptr;  const float* pin_b = nullptr;
    const float* dw2_w = nullptr; const float* dw2_b = nullptr;
    const float* dw2_u = nullptr; const float* dw2_ub = nullptr;   // the same weights per vector-work unit (ira_ffn2.hip; pack.py::_dw_unit_pack)
    const void* pout_w = nullptr; const float* pout_b = nullptr;
    int ffn_lo1 = 3;   // lo planes the fused GDFN multiplies under weight_planes == 1 (bit 0 project_in, bit 1 project_out; ira_rt_restormer.hip::ffn_lo_rule)
    // X-Restormer only (xrestormer_arch.py:239-262): OCAB and the second GDFN; level = index into spatial_heads
    int level = 0;
    const void* oc_qkv_w = nullptr; const float* oc_qkv_b = nullptr; const float* oc_rel = nullptr;
    const void* oc_proj_w = nullptr; const float* oc_proj_b = nullptr;
    const void* pin2_w = nullptr; const float* pin2_b = nullptr;
    const float* dw3_w = nullptr; const float* dw3_b = nullptr;
    const float* dw3_u = nullptr; const float* dw3_ub = nullptr;
    const void* pout2_w = nullptr; const float* pout2_b = nullptr;
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


}  // namespace ira_rt

using ira_rt::Blk; using ira_rt::Prof; using ira_rt::SBlk; using ira_rt::SConv; using ira_rt::NBlk;

struct ira_model {
    int device = -1;
    int arch = 0;                    // 0 Restormer, 1 NAFNet, 2 DnCNN, 3 SwinIR, 4 MPRNet, 5 RRDBNet
    ira_restormer_config cfg{};
    bool xr = false;                 // arch 0 with X-Restormer blocks
    int xr_spatial_heads[4] = {0, 0, 0, 0};
    int xr_scale = 1;                // X-Restormer SR preset: bilinear x4 of the input before the network
    float* xr_up = nullptr; size_t xr_up_bytes = 0;
    ira_nafnet_config ncfg{};
    ira_dncnn_config dcfg{};
    ira_swinir_config scfg{};
    ira_mprnet_config mcfg{};
    ira_rrdbnet_config rcfg{};
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
    int ffn_gen = 2;
    float* ffn_out = nullptr; int ffn_out_ld = 0;     // one-shot output override of the next fused GDFN launch (ira_rt_restormer.hip: level-1 concat without a copy)
    char* lo_backup = nullptr;           // device copy of the lo planes that w_planes == 1 zeroed (ira_set_weight_planes(2) puts them back)
    int w_planes = 2;          // ira_set_weight_planes: 1 = the 1x1 layers of this model use the hi plane alone where a kernel for that exists                  // 2: ira_ffn2.hip (attention apply folded in); 1: ira_ffn_fused.hip + separate apply (IRA_FFN_GEN=1)
    // strip-parallel whole-frame forward (ira_strip_*): this rank's band of rows with halo rows of its neighbours
    bool strip_set = false, strip_on = false;
    ira_strip_comm scomm{};
    int strip_top = 0, strip_bot = 0;     // halo rows at level 1 (0 or 8)
    // RCCL communicator of this handle (ira_comm_init): the tile all-gather and, when strip_rccl is set, the band exchange steps run on
    // the model's stream with no host synchronisation
    ira_comm::Comm* comm = nullptr;
    bool strip_rccl = false;
    char* strip_buf = nullptr; size_t strip_buf_bytes = 0;      // library-owned staging of the band exchange (RCCL mode)
    float* dist_tiles = nullptr; size_t dist_tiles_bytes = 0;   // all tiles of a frame (all-gather destination)
    // grow-only device buffers of the host-image entry points (ira_restore_u8, ira_restore_padcrop_u8): frame in, frame out, tiles, ...
    void* io[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t io_bytes[5] = {0, 0, 0, 0, 0};
    int* strip_desc = nullptr;                                  // device copy of the band descriptor of ira_strip_restore_u8_device
    int strip_desc_host[6] = {0, 0, 0, 0, 0, 0};
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

namespace ira_rt {

extern thread_local std::string g_err;   // errors of calls that have no handle (ira_runtime.hip)

inline int gfail(int code, const char* fmt, ...) {
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
// output pixels per input pixel along each axis (SR models)
inline int model_scale(const ira_model* m) {
    return m->arch == 3 ? m->scfg.upscale : (m->arch == 5 ? m->rcfg.scale : (m->xr ? m->xr_scale : 1));
}
inline size_t al256(size_t n) { return (n + 255) / 256 * 256; }

// ---- profiling wrapper: run one launch, optionally bracketed by events
template <class F>
inline int run(ira_model* m, const char* family, F&& f) {
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

inline const void* wptr(ira_model* m, const std::string& name, bool required, size_t expect_bytes = 0) {
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

int ensure_scratch(ira_model* m, size_t bytes);                                   // ira_runtime.hip

// per-architecture: bind the device pointers of the weight blob; enqueue one forward on float32 NHWC device tensors
bool restormer_bind_weights(ira_model* m);
int restormer_forward_enqueue(ira_model* m, const float* in, int T, int H, int W, float* out);
int restormer_strip_forward_enqueue(ira_model* m, const float* in, int H, int W, int halo_top, int halo_bot, float* out);
bool naf_bind_weights(ira_model* m);
int naf_forward_enqueue(ira_model* m, const float* in, int T, int H, int W, float* out);
bool dn_bind_weights(ira_model* m);
int dn_forward_enqueue(ira_model* m, const float* in, int T, int H, int W, float* out);
bool sw_bind_weights(ira_model* m);
void sw_lo_planes(ira_model* m, std::vector<std::pair<void*, size_t>>& out);          // the lo-plane regions ira_set_weight_planes(1) zeroes on a SwinIR handle
void restormer_lo_planes(ira_model* m, std::vector<std::pair<void*, size_t>>& out);   // the same for Restormer / X-Restormer (levels with C = 192 / 384)
int sw_forward_enqueue(ira_model* m, const float* in, int T, int H, int W, float* out);
bool rr_bind_weights(ira_model* m);
int rr_forward_enqueue(ira_model* m, const float* in, int T, int H, int W, float* out);
bool mpr_bind_weights(ira_model* m);
int mpr_forward_enqueue(ira_model* m, const float* in, int T, int H, int W, float* out);

}  // namespace ira_rt
