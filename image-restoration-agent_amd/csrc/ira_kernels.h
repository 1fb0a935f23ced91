// Launchers of the hand-written gfx950 kernels (ira_kernels.hip). Host-side C++ interface used by
// ira_runtime.hip; the public boundary is include/ira.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace ira {

// hipFuncAttributeMaxDynamicSharedMemorySize is per (function, device): raise it once per device (a process may hold handles on
// several GPUs, driven from several host threads).  `mask` is one static word per kernel instantiation, bit d = done on device d.
inline hipError_t ensure_dynamic_lds(const void* fn, size_t bytes, std::atomic<uint64_t>& mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (mask.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) mask.fetch_or(bit, std::memory_order_release);
    return e;
}

// All launchers enqueue on `st` and return hipGetLastError(). T = tiles, P = pixels per tile.

hipError_t launch_pw_kres(hipStream_t st, const void* x, bool in_f32, int ln_mode, int K,
                          const void* W, const float* bias, void* y, int y_stride, int n_store,
                          int NB, int T, int P, float eps = 1e-5f, int ln_n = 0, int act = 0, int planes = 2);   // ln_n: true channel count when K is zero-padded; act 1 = GELU;
                          // planes 1 = use the fp16 hi plane of W alone where a kernel for that exists (ira_set_weight_planes), else both

hipError_t launch_pw_nres(hipStream_t st, const void* x0, int K0, const void* x1, int K1, bool in_f32,
                          const void* W, long w_tile_stride, const float* bias, const float* resid,
                          float* y, int N, int NB, int T, int P, int planes = 2);

hipError_t launch_dwconv_qkv(hipStream_t st, const void* in, const float* w, const float* bias,
                             void* qk, void* v, int T, int H, int W, int C);

hipError_t launch_dwconv_gate(hipStream_t st, const void* in, const float* w, const float* bias,
                              void* out, int T, int H, int W, int Hp);

hipError_t launch_gram(hipStream_t st, const void* qk, float* part, int T, int nblk, int C, int heads, int S);

// attn_scratch: fp32 [T][heads][c][c]
// red: room for T x attn_mix_reduce_groups(S) records when that is non-zero (S > 256: frame-wide Grams), or NULL = single-level sum
int attn_mix_reduce_groups(int S);
hipError_t launch_attn_mix(hipStream_t st, const float* part, const float* wo, const float* temp,
                           float* attn_scratch, void* M, int T, int C, int heads, int S, int NB, bool m_lo_zero = false, float* red = nullptr);

// Fused attention front end (ira_qkv_fused.hip): LN + qkv 1x1 + depthwise 3x3 -> q,k (pixel-minor) and v (NHWC).
bool qkv_fused_supported(int C);
hipError_t launch_qkv_fused(hipStream_t st, const float* x, int ln_mode, int C, const void* Wq, const float* bq,
                            const float* dww, const float* dwb, void* qk, void* v, int T, int H, int W);
// Third generation (ira_qkv3.hip): 4-wave workgroups, two per CU; pixel-stationary TRANSPOSED 1x1 (a depthwise lane owns whole pixel-minor
// q,k elements); weights staged once per workgroup in LDS by DMA; MFMAs and depthwise FMAs interleaved.  Same arguments, layouts and
// rounding points as generation 1.  launch_qkv_front picks the generation: 3 unless IRA_QKV_GEN=1 (read once per process).
hipError_t launch_qkv_front(hipStream_t st, const float* x, int ln_mode, int C, const void* Wq, const float* bq, const float* dww,
                            const float* dwb, void* qk, void* v, int T, int H, int W, int planes = 2);   // planes 1: generation 3 skips the lo weight plane
bool qkv3_supported(int C);
hipError_t launch_qkv3(hipStream_t st, const float* x, int ln_mode, int C, const void* Wq, const float* bq, const float* dww,
                       const float* dwb, void* qk, void* v, int T, int H, int W, int planes = 2);

// Fused GDFN (ira_ffn_fused.hip): x_out = x + ffn(LN(x)); x_out must not alias x.  Supported for C in {48, 96}.
bool ffn_fused_supported(int C, int Hp);
hipError_t launch_ffn_fused(hipStream_t st, const float* x, float* x_out, int ln_mode, int C, const void* Win,
                            const float* bin, const float* dww, const float* dwb, const void* Wout, const float* bout,
                            int T, int H, int W, int Hp);

// Fused attention apply + GDFN, second generation (ira_ffn2.hip): x_out = x1 + ffn(LN(x1)), x1 = x + M_t . v (+ wo_b); v == M == nullptr: x1 = x.
// M: the per-tile matrices of launch_attn_mix (fp16 hi + lo planes, fragment-major), m_tile_stride in halfs.
// ira_pw_wide.hip: LayerNorm + 1x1 for K = 192 with 512-pixel workgroups (arguments of launch_pw_kres)
bool pw_wide_supported(int K, long npx, int planes = 2);
hipError_t launch_pw_wide(hipStream_t st, const void* x, bool in_f32, int ln_mode, int K, const void* W, const float* bias, void* y, int y_stride,
                          int n_store, int NB, int T, int P, float eps, int ln_n, int act, int planes = 2);

// ira_nres_wide.hip: 1x1 + residual for N = 192 / 384 (fp16 pixels, one input tensor) in 256-pixel workgroups
bool nres_wide_supported(int K, int N, int NB, int T, int P, long w_tile_stride);
hipError_t launch_nres_wide(hipStream_t st, const void* x, int K, const void* W, long w_tile_stride, const float* bias, const float* resid, float* y,
                            int N, int NB, int T, int P, int planes = 2);

// ira_swin_mlp.hip: x += fc2(GELU(fc1(LayerNorm(x)))) in one kernel for 256 padded channels (SwinIR-L), in place on the fp32 token stream
bool swin_mlp_supported(int Cp, int hidp, long npx);
hipError_t launch_swin_mlp(hipStream_t st, float* x, const void* W1, const float* b1, const void* W2, const float* b2, int Cp, int hidp, long npx,
                           float eps, int ln_n, int planes,
                           const void* attn = nullptr, const void* Wp = nullptr, const float* bp = nullptr,    // attn != NULL: x = xin + W_p attn + b_p first (attn fp16 (npx, 256))
                           const float* xin = nullptr);                                                        // NULL = x (in place); else x is only written: the block's input stays intact

// ira_brighten.hip: tool 0 constant shift, 1 gamma correction, 2 CLAHE (hist_scratch: 64 x 256 u32, lut_scratch: 64 x 256 bytes, tool 2 only)
hipError_t launch_brighten(hipStream_t st, int tool, const uint8_t* in_rgb, uint8_t* out_rgb, int h, int w, uint32_t* hist_scratch, uint8_t* lut_scratch);

// ira_ffn2.hip: the (Phi, slope) table of the GELU look-up, 1024 entries of float2 over [0, 6), one copy per device
// Phi table (one device copy per GPU, filled on first use): tab[i] = (Phi(i h), Phi((i + 1) h) - Phi(i h)), h = PHI_RANGE / PHI_NTAB;
// gelu(a) = |a| Phi(|a|) + min(a, 0) with Phi interpolated linearly (error < 1.1e-6 |a|, far below the fp16 rounding of what it feeds)
constexpr int PHI_NTAB = 1024;
constexpr float PHI_RANGE = 6.0f;
hipError_t phi_table(const float2** out);
bool ffn2_supported(int C, int Hp);
hipError_t launch_ffn2(hipStream_t st, const float* x, float* x_out, const void* v, const void* M, long m_tile_stride, const float* wo_b,
                       int ln_mode, int C, const void* Win, const float* bin, const float* dww, const float* dwb, const void* Wout,
                       const float* bout, int T, int H, int W, int Hp, int ldo = 0, int lo = 3);
// ldo: floats per pixel of x_out (0 = C; a concat buffer's width otherwise).  lo: which 1x1 weights carry a lo plane that is multiplied
// (bit 0 W_in, bit 1 W_out; ffn2_lo_supported(C, lo): 3 everywhere, 1 and 0 at C = 96): the others are read as plain fp16.
bool ffn2_lo_supported(int C, int lo);

hipError_t launch_conv3x3(hipStream_t st, const float* in, const void* W, const float* bias,
                          const float* skip, float* out, int mode, int T, int H, int Wd, int K, int N, int NB, float slope = 0.f,
                          int ldin = 0, int ldout = 0);   // floats per pixel of in / of out and skip (0 = K / N)

hipError_t launch_embed(hipStream_t st, const float* in, const float* w, const float* bias, float* out,
                        int T, int H, int W, int ic, int C, int relu = 0);

hipError_t launch_sum_records(hipStream_t st, const float* part, int S, int reclen, float* out);   // out[i] = sum_s part[s][i]
hipError_t launch_concat2(hipStream_t st, const float* a, int Ca, const float* b, int Cb, float* out, long npx);
hipError_t launch_copy_slice(hipStream_t st, const float* b, int Cb, float* out, int ldo, int off, long npx);

// tiler
hipError_t launch_extract_tiles(hipStream_t st, const uint8_t* frame, int h, int w, int c, int Hp, int Wp,
                                int th, int tw, int stride_h, int stride_w, int n_rows, int n_cols,
                                int first_tile, int n_tiles, float* tiles, int flip = 0);   // flip 1: flip-concat padding (SwinIR harness)

hipError_t launch_blend(hipStream_t st, const float* tiles, int h, int w, int c, int Hp, int Wp, int th, int tw,
                        int stride_h, int stride_w, int n_rows, int n_cols, uint8_t* out);

// pad-crop tiler (xrestormer_model.py:53-136): desc[n][6] = {tile y0, x0, core y0, x0, core h, w} in padded-frame coordinates
hipError_t launch_extract_list(hipStream_t st, const uint8_t* frame, int h, int w, int c, int th, int tw, const int* desc,
                               int n_tiles, float* tiles);
hipError_t launch_scatter_cores(hipStream_t st, const float* tiles, int th, int tw, int c, const int* desc, int n_tiles,
                                int max_core, int h, int w, uint8_t* out, int scale = 1);   // tiles / out on the scale-times finer grid

hipError_t launch_nchw_to_nhwc(hipStream_t st, const float* in, float* out, int B, int C, int H, int W);
hipError_t launch_nhwc_to_nchw(hipStream_t st, const float* in, float* out, int B, int C, int H, int W);

// ---- NAFNet (ira_nafnet.hip).  inmode: 0 fp16, 1 fp16 SimpleGate pair, 2 fp32 space-to-depth (2x2 stride-2 conv), 3 fp32;
// outmode: 0 plain (+resid), 1 PixelShuffle(2) + add resid at the destination.  (Hg, Wg) = the MFMA pixel grid.
hipError_t launch_naf_pw(hipStream_t st, int inmode, int outmode, const void* x, int K, const void* W, long w_tile_stride,
                         const float* bias, const float* resid, float* y, int N, int NBtot, int T, int Hg, int Wg,
                         const float* scale = nullptr);   // inmode 4: fp16 x scaled by scale[t][k] (fp32 [T][K])
// the same contraction with both operands through LDS (k_pw_gemm); bitwise equal to k_naf_pw / k_pw_nres_lds on their common cases
bool pw_gemm_supported(int K, int N, int NBtot);
hipError_t launch_pw_gemm(hipStream_t st, const void* x, bool x_f32, int K, const void* W, long w_tile_stride, const float* bias,
                          const float* resid, float* y, int N, int NBtot, int T, int P);
int naf_dw_sg_blocks(int H, int W, int c);      // partial-sum records per image written by launch_naf_dw_sg
hipError_t launch_naf_dw_sg(hipStream_t st, const void* in, const float* w, const float* bias, void* out, float* part,
                            int T, int H, int W, int c);
hipError_t launch_naf_sca(hipStream_t st, const float* part, int bpt, long npx, const float* wsca, const float* bsca, float* s,
                          int T, int c);
// Fused NAFBlock tail (conv3 with the channel attention + residual, LN, conv4, SimpleGate, conv5 + residual), in place on x.
bool naf_tail_supported(int c);
hipError_t launch_naf_tail(hipStream_t st, const void* g, const float* scale, float* x, int c, const void* W3, const float* b3,
                           const void* W4, const float* b4, const void* W5, const float* b5, int T, int P);
hipError_t launch_naf_pad_crop(hipStream_t st, const float* in, float* out, int T, int Hi, int Wi, int Ho, int Wo, int C);

// ---- SwinIR (ira_swinir.hip)
// (shifted-)window attention: qkv fp16 [T][H][W][3*heads*32] ([q|k|v][head][32]), bias fp32 [heads][64][64], out fp16 [T][H][W][heads*32]
hipError_t launch_swin_attn(hipStream_t st, const void* qkv, const float* bias, void* out, int T, int H, int W, int heads,
                            int ws, int shift);
hipError_t launch_ln_affine(hipStream_t st, const float* x, const float* g, const float* b, float* y, long npx, int C, int Cp,
                            float eps);
hipError_t launch_swin_pre(hipStream_t st, const float* in, float* out, int T, int H, int W, int Hp, int Wp, int c,
                           const float* mean3, float range);
hipError_t launch_swin_post(hipStream_t st, const float* in, float* out, int T, int Hi, int Wi, int Ho, int Wo, int c,
                            const float* mean3, float range);
hipError_t launch_nearest2x(hipStream_t st, const float* in, float* out, int T, int H, int W, int C);
hipError_t launch_lrelu(hipStream_t st, float* x, long n, float slope);

// ---- X-Restormer (ira_xrestormer.hip): overlapping cross-attention, window 8 / overlap window 12 / head_dim 16.
// qkv fp16 [T][H][W][3*heads*16] (q | k | v), rel fp32 [2][32][16] (rel_width rows 0..22, then rel_height), out fp16 [T][H][W][heads*16]
hipError_t launch_bilinear_up(hipStream_t st, const float* in, float* out, int T, int H, int W, int C, int s);
hipError_t launch_ocab_attn(hipStream_t st, const void* qkv, const float* rel, void* out, int T, int H, int W, int heads);

// ---- MPRNet (ira_mprnet.hip): fp32 NHWC, channels padded to a multiple of 16
int mpr_ca_blocks(long P);          // partial records per image used by launch_mpr_ca
hipError_t launch_mpr_ca(hipStream_t st, const float* x, float* part, const float* w1, const float* w2, float* z, int T, long P, int C, int Cp, int Cr);
hipError_t launch_mpr_scale_add(hipStream_t st, const float* res, const float* z, const float* x, float* out, int T, long P, int Cp);
hipError_t launch_mpr_sam(hipStream_t st, const float* x1, const float* y8, const float* ximg, const float* w3, const float* x, float* out,
                          float* img_out, int C, int Cp, long npx);
// wide [N][h][2w][C] <-> halves [2N][h][w][C] (half image 2n + lr); merge 0: split, 1: concatenate
hipError_t launch_mpr_lr(hipStream_t st, float* wide, float* halves, int N, int h, int w, int C, int merge);
hipError_t launch_mpr_up2(hipStream_t st, const float* in, const float* add, float* out, int T, int H, int W, int C);

// ---- RRDBNet (ira_rrdb.hip)
hipError_t launch_pixel_unshuffle(hipStream_t st, const float* in, float* out, int T, int H, int W, int c, int s);
hipError_t launch_axpy_strided(hipStream_t st, const float* u, int ldu, const float* v, int ldv, float* y, int ldy, float a, int C, long npx);

}  // namespace ira
