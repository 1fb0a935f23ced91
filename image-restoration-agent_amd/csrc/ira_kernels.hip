// Hand-written gfx950 (CDNA4, wave64) kernels for the Restormer hot path.
//
// Reference being re-implemented (never copied): restormer_arch.py:19-284 and demo.py:120-161 of
// AgenticIR/executor/defocus_deblurring/tools/Restormer/.  Layouts and the numerics contract are in
// DESIGN.md.  Short version:
//   residual stream x      fp32 NHWC  [T][H][W][C]
//   branch activations     fp16 NHWC  (qkv_pre, v, hidden_pre, gated hidden)
//   q,k after the dwconv   fp16 "pixel-minor" [T][H][ceil(W/8)][2C][8]   (K-dim of the Gram = pixels)
//   all contractions       v_mfma_f32_32x32x16_f16 / 16x16x32_f16, fp32 accumulate
//   weights                fp16 hi + fp16 lo planes (W ~= hi + lo to ~2^-22): every weight MFMA is issued
//                          twice on the same pixel fragment.  A fixed weight-rounding error does not
//                          average out over pixels, and it was the largest term in the 1e-3 dB PSNR
//                          budget (DESIGN.md "Numerics contract")
//   LN statistics, dwconv, softmax, GELU, residual adds: fp32 VALU
//
// MFMA operand convention used by every 1x1 / 3x3 conv here ("weights are A, pixels are B"):
//   D[32 ch][32 px] += A[32 ch][16 k] * B[16 k][32 px]
//   lane l: r = l & 31, h = l >> 5.
//   B fragment of lane (r,h): 8 consecutive input channels 16s+8h .. +7 of pixel r   (one 16-byte load, NHWC)
//   A fragment of lane (r,h): W[row 32nb + rowperm(r)][16s+8h .. +7].  Weights are stored FRAGMENT-MAJOR,
//                             [plane][nb][s][lane][8], so one A load is one contiguous 1 KB per wave (a row-major
//                             [N][K] matrix costs 32-64 cache lines per load instruction: TA-bound)
//   D: lane (r,h), register t holds MFMA row (t&3) + 8(t>>2) + 4h.  With rowperm() below that row is
//      output channel 32nb + 16(t>>3) + 8h + (t&7): each lane owns two runs of 8 consecutive
//      channels of ITS pixel -> 16/32-byte NHWC stores, and the registers are already the B fragment
//      (k = 16s + 8h + j) of a following MFMA that contracts over these channels.
#include "ira_kernels.h"
#include "ira_device.h"

#include <algorithm>
#include <cstdlib>

namespace ira {

// ---------------------------------------------------------------------------------------------
// Pointwise (1x1) conv, "K-resident": all K input channels of the wave's 32 pixels stay in
// registers (optionally LayerNorm-ed there), the wave loops over 32-row blocks of W.
// Used for LN + qkv 1x1 (restormer_arch.py:105,114 after :146) and LN + project_in (:82,89).
//   x: [T*P][K] fp32 or fp16.   W: [NB*32][K] fp16 (rows >= N zero).   y: [T*P][y_stride] fp16.
// LN: 0 none; 1 BiasFree core x*rsqrt(var+1e-5) (:37-39); 2 WithBias core (x-mu)*rsqrt(var+1e-5)
// (:54-57). The gain (and W@ln_bias) are folded into W / bias by the packer.
// PG = 32-pixel groups per wave: every A (weight) fragment fetched from L1/L2 is reused on PG pixel
// groups, which divides the weight traffic per pixel by PG (at PG = 1 it exceeded the HBM traffic 4x).
template <int KS, int LN, bool IN_F32, int PG>
__global__ __launch_bounds__(256) void k_pw_kres(const void* __restrict__ xin, const f16* __restrict__ W,
                                                 const float* __restrict__ bias, f16* __restrict__ y,
                                                 int y_stride, int n_store, int NB, int T, int P, float eps, int ln_n, int act) {
    constexpr int K = KS * 16;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int chunks = (P + 32 * PG - 1) / (32 * PG);
    const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= (long)T * chunks) return;
    const int t = (int)(wv / chunks);
    const int pbase = (int)(wv % chunks) * 32 * PG + r;

    half8 b[PG][KS];
    bool valid[PG];
    long px[PG];
#pragma unroll
    for (int g = 0; g < PG; ++g) {
        const int p = pbase + 32 * g;
        valid[g] = p < P;
        px[g] = (long)t * P + (valid[g] ? p : P - 1);
        if constexpr (IN_F32) {
            const float* src = (const float*)xin + px[g] * K + 8 * h;
            float xv[KS][8];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const f32x4 lo = *(const f32x4*)(src + 16 * s);
                const f32x4 hi = *(const f32x4*)(src + 16 * s + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { xv[s][j] = lo[j]; xv[s][4 + j] = hi[j]; }
            }
            if constexpr (LN != 0) {
                float sum = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int j = 0; j < 8; ++j) sum += xv[s][j];
                sum += __shfl_xor(sum, 32);
                // ln_n < K: channels [ln_n, K) are zero padding (SwinIR's 180 -> 192): statistics over ln_n channels
                const float inv_n = 1.0f / (float)ln_n;
                const float mu = sum * inv_n;
                float sq = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float d = xv[s][j] - mu; sq += d * d; }
                sq += __shfl_xor(sq, 32);
                sq -= (float)(K - ln_n) * mu * mu;
                const float rstd = 1.0f / sqrtf(sq * inv_n + eps);
                const float sub = (LN == 2) ? mu : 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[s][j] = (xv[s][j] - sub) * rstd;
            }
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) b[g][s][j] = (f16)xv[s][j];
        } else {
            const f16* src = (const f16*)xin + px[g] * K + 8 * h;
#pragma unroll
            for (int s = 0; s < KS; ++s) b[g][s] = *(const half8*)(src + 16 * s);
        }
    }

    // W: fragment-major [2][NB][KS][64][8]
    const f16* wrow = W + lane * 8;
    for (int nb = 0; nb < NB; ++nb) {
        f32x16 acc[PG];
        if (bias != nullptr) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float bv = bias[32 * nb + 16 * u + 8 * h + j];
#pragma unroll
                    for (int g = 0; g < PG; ++g) acc[g][8 * u + j] = bv;
                }
        } else {
#pragma unroll
            for (int g = 0; g < PG; ++g) acc[g] = zero16();
        }
        const f16* wp = wrow + (long)nb * KS * 512;
        const f16* wl = wp + (long)NB * KS * 512;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const half8 a = *(const half8*)(wp + s * 512);
            const half8 al = *(const half8*)(wl + s * 512);
#pragma unroll
            for (int g = 0; g < PG; ++g) {
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[g][s], acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b[g][s], acc[g], 0, 0, 0);
            }
        }
#pragma unroll
        for (int g = 0; g < PG; ++g)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n0 = 32 * nb + 16 * u + 8 * h;
                if (valid[g] && n0 + 8 <= n_store) {
                    half8 o;
                    if (act == 1) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[g][8 * u + j] = gelu_erf(acc[g][8 * u + j]);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (f16)acc[g][8 * u + j];
                    *(half8*)(y + px[g] * y_stride + n0) = o;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-shared-weight variants for the wide levels (C = 192 / 384).  There a weight block is 24-48 KB and each
// wave of k_pw_kres / k_pw_nres streamed the whole matrix (up to 3 MB) from L2 for its 32 pixels: the kernels were
// L2-bandwidth bound at ~0.5 TB/s of useful traffic.  Here the 4 waves of a workgroup stage every weight block in
// LDS cooperatively (each wave fetches a quarter, one block ahead, through registers) and all read it back with
// conflict-free ds_read_b128: 4x less L2 traffic.  Waves never exit early (barriers): out-of-range waves clamp.
template <int KS, int LN, bool IN_F32, int NW, int MINW>
__global__ __launch_bounds__(64 * NW, MINW) void k_pw_kres_lds(const void* __restrict__ xin, const f16* __restrict__ W,
                                                     const float* __restrict__ bias, f16* __restrict__ y,
                                                     int y_stride, int n_store, int NB, int T, int P, float eps, int ln_n, int act) {
    constexpr int K = KS * 16;
    constexpr int F = 2 * KS;                 // fragments (1 KB each) per 32-row weight block: hi plane then lo plane
    constexpr int SH = (F + NW - 1) / NW;     // fragments each of the NW waves stages per block
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half8* wl = (half8*)smem;                 // [2][F][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int chunks = (P + 31) >> 5;
    const long total = (long)T * chunks;
    long wv = (long)blockIdx.x * NW + wave;
    const bool live = wv < total;
    if (!live) wv = total - 1;
    const int t = (int)(wv / chunks);
    const int p = (int)(wv % chunks) * 32 + r;
    const bool valid = live && p < P;
    const long px = (long)t * P + (p < P ? p : P - 1);

    half8 b[KS];
    if constexpr (IN_F32) {
        const float* src = (const float*)xin + px * K + 8 * h;
        float sum = 0.f, sq = 0.f;
        // two passes over registers would need K/2 fp32 registers per lane; for K = 384 that is 192: keep fp32
        // only for the statistics (re-read from L1/L2 for the normalisation pass)
        if constexpr (LN != 0) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const f32x4 lo = *(const f32x4*)(src + 16 * s), hi = *(const f32x4*)(src + 16 * s + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) sum += lo[j] + hi[j];
            }
            sum += __shfl_xor(sum, 32);
        }
        const float inv_n = 1.0f / (float)ln_n;
        const float mu = sum * inv_n;
        if constexpr (LN != 0) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const f32x4 lo = *(const f32x4*)(src + 16 * s), hi = *(const f32x4*)(src + 16 * s + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d0 = lo[j] - mu, d1 = hi[j] - mu; sq += d0 * d0 + d1 * d1; }
            }
            sq += __shfl_xor(sq, 32);
            sq -= (float)(K - ln_n) * mu * mu;
        }
        const float rstd = (LN != 0) ? 1.0f / sqrtf(sq * inv_n + eps) : 1.0f;
        const float sub = (LN == 2) ? mu : 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f32x4 lo = *(const f32x4*)(src + 16 * s), hi = *(const f32x4*)(src + 16 * s + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { b[s][j] = (f16)((lo[j] - sub) * rstd); b[s][4 + j] = (f16)((hi[j] - sub) * rstd); }
        }
    } else {
        const f16* src = (const f16*)xin + px * K + 8 * h;
#pragma unroll
        for (int s = 0; s < KS; ++s) b[s] = *(const half8*)(src + 16 * s);
    }

    // fragment f of block nb: f < KS -> hi plane k-step f, else lo plane k-step f-KS
    auto gsrc = [&](int nb, int f) -> const half8* {
        const long plane = (f < KS) ? 0 : (long)NB * KS * 512;
        return (const half8*)(W + plane + ((long)nb * KS + (f < KS ? f : f - KS)) * 512 + lane * 8);
    };
    // blockIdx.y splits the NB row blocks (small pixel counts: more workgroups instead of longer ones)
    const int nb_per = (NB + gridDim.y - 1) / gridDim.y;
    const int nb_lo = blockIdx.y * nb_per, nb_hi = min(NB, nb_lo + nb_per);
    half8 stg[SH];
#pragma unroll
    for (int i = 0; i < SH; ++i) if (wave + NW * i < F) stg[i] = *gsrc(nb_lo, wave + NW * i);
#pragma unroll
    for (int i = 0; i < SH; ++i) if (wave + NW * i < F) wl[(wave + NW * i) * 64 + lane] = stg[i];
    __syncthreads();

    f32x16 binit = zero16();
    if (bias != nullptr) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) binit[8 * u + j] = bias[32 * nb_lo + 16 * u + 8 * h + j];
    }
    for (int nb = nb_lo; nb < nb_hi; ++nb) {
        const half8* cur = wl + ((nb - nb_lo) & 1) * F * 64;
        // The accumulators start from the bias of THIS block, fetched one iteration ago together with the weight prefetch.  Loaded here
        // instead, the wait for the bias in front of the first MFMA was vmcnt(0) (vector-memory results return in order): every iteration
        // of every layer with a bias (SwinIR, X-Restormer, WithBias Restormer) waited for the weights of the NEXT block - an L2 round trip.
        f32x16 acc = binit;
        if (nb + 1 < nb_hi) {
#pragma unroll
            for (int i = 0; i < SH; ++i) if (wave + NW * i < F) stg[i] = *gsrc(nb + 1, wave + NW * i);     // prefetch, issued before the stores below
            if (bias != nullptr) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j) binit[8 * u + j] = bias[32 * (nb + 1) + 16 * u + 8 * h + j];
            }
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[s * 64 + lane], b[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[(KS + s) * 64 + lane], b[s], acc, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int n0 = 32 * nb + 16 * u + 8 * h;
            if (valid && n0 + 8 <= n_store) {
                half8 o;
                if (act == 1) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[8 * u + j] = gelu_erf(acc[8 * u + j]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (f16)acc[8 * u + j];
                *(half8*)(y + px * y_stride + n0) = o;
            }
        }
        if (nb + 1 < nb_hi) {
            half8* nxt = wl + ((nb + 1 - nb_lo) & 1) * F * 64;
#pragma unroll
            for (int i = 0; i < SH; ++i) if (wave + NW * i < F) nxt[(wave + NW * i) * 64 + lane] = stg[i];
        }
        __syncthreads();
    }
}

template <int NB, bool IN_F32, int PG>
__global__ __launch_bounds__(256) void k_pw_nres_lds(const void* __restrict__ x0, int K0, const void* __restrict__ x1, int K1,
                                                     const f16* __restrict__ W, long w_tile_stride,
                                                     const float* __restrict__ bias, const float* resid, float* y,
                                                     int N, int T, int P, int NBtot) {
    // NB = 32-row blocks per workgroup, starting at block blockIdx.y * NB of the matrix's NBtot (N-split: a smaller register
    // tile lets three waves share a SIMD, which hides the per-k-step load / LDS-write / barrier phases that one wave per
    // SIMD exposed: SwinIR-L fc2 at NB = 8 in one piece 36.6 ms per frame, as 2 x 4 blocks 34.2; the pixel operand is
    // re-read once per split, from L2).
    // one weight "slab" = the NB row blocks of ONE k-step, hi and lo planes: F = 2*NB fragments of 1 KB.
    // Per-tile weights (attention apply): all four waves of a workgroup must be in the same tile, so the wave ->
    // pixel map is per tile: ceil(chunks / (4*PG)) workgroups per tile.  PG = 32-pixel groups per wave: every weight
    // fragment read from LDS feeds PG MFMAs (at PG = 1 the LDS read rate equals the MFMA rate: 1 KB per 32 cycles).
    constexpr int F = 2 * NB;
    constexpr int SH = (F + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half8* wl = (half8*)smem;                 // [2][F][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int chunks = (P + 31) >> 5;
    const int wg_per_tile = (chunks + 4 * PG - 1) / (4 * PG);
    const int t = blockIdx.x / wg_per_tile;
    const int ch0 = ((blockIdx.x % wg_per_tile) * 4 + wave) * PG;
    bool valid[PG];
    long px[PG];
#pragma unroll
    for (int g = 0; g < PG; ++g) {
        const int ch = ch0 + g;
        const bool live = ch < chunks;
        const int p = (live ? ch : chunks - 1) * 32 + r;
        valid[g] = live && p < P;
        px[g] = (long)t * P + (p < P ? p : P - 1);
    }
    const int K = K0 + K1, KT = K >> 4;
    const int nb0 = blockIdx.y * NB;

    f32x16 acc[PG][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        f32x16 init = zero16();
        if (bias != nullptr) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = 32 * (nb0 + nb) + 16 * u + 8 * h + j;
                    init[8 * u + j] = (n < N) ? bias[n] : 0.f;
                }
        }
#pragma unroll
        for (int g = 0; g < PG; ++g) acc[g][nb] = init;
    }
    const f16* wt = W + (long)t * w_tile_stride + lane * 8;
    const long lo_off = (long)NBtot * KT * 512;
    auto gsrc = [&](int s, int f) -> const half8* {     // f < NB: hi plane block nb0+f, else lo plane block nb0+f-NB
        int nb = nb0 + (f < NB ? f : f - NB);
        if (nb >= NBtot) nb = NBtot - 1;                 // partial last split: staged, never stored
        return (const half8*)(wt + (f < NB ? 0 : lo_off) + ((long)nb * KT + s) * 512);
    };
    half8 stg[SH];
#pragma unroll
    for (int i = 0; i < SH; ++i) if (wave + 4 * i < F) stg[i] = *gsrc(0, wave + 4 * i);
#pragma unroll
    for (int i = 0; i < SH; ++i) if (wave + 4 * i < F) wl[(wave + 4 * i) * 64 + lane] = stg[i];
    __syncthreads();

    // B (pixel) fragments are fetched BD k-steps ahead: a load issued in the k-step that uses it exposed one full memory
    // latency per k-step (SQ_WAIT_ANY 72 % of wave cycles on the SwinIR fc2 shape), and loads cannot be hoisted over the
    // barrier by the compiler
    constexpr int BD = (NB <= 3 && PG == 1) ? 4 : 1;      // deeper only where the registers are free (<= 152 -> 3 waves per SIMD)
    auto load_b = [&](int s, int g) -> half8 {
        const int c0 = 16 * s;
        if constexpr (IN_F32) {
            const float* src = (c0 < K0) ? ((const float*)x0 + px[g] * K0 + c0 + 8 * h) : ((const float*)x1 + px[g] * K1 + (c0 - K0) + 8 * h);
            return cvt8(*(const f32x4*)src, *(const f32x4*)(src + 4));
        } else {
            const f16* src = (c0 < K0) ? ((const f16*)x0 + px[g] * K0 + c0 + 8 * h) : ((const f16*)x1 + px[g] * K1 + (c0 - K0) + 8 * h);
            return *(const half8*)src;
        }
    };
    half8 bq[PG][BD];
#pragma unroll
    for (int j = 0; j < BD; ++j)
        if (j < KT) {
#pragma unroll
            for (int g = 0; g < PG; ++g) bq[g][j] = load_b(j, g);
        }
    for (int s0 = 0; s0 < KT; s0 += BD) {
#pragma unroll
        for (int j = 0; j < BD; ++j) {
            const int s = s0 + j;
            if (s >= KT) break;
            const half8* cur = wl + (s & 1) * F * 64;
            if (s + 1 < KT) {
#pragma unroll
                for (int i = 0; i < SH; ++i) if (wave + 4 * i < F) stg[i] = *gsrc(s + 1, wave + 4 * i);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const half8 a = cur[nb * 64 + lane], al = cur[(NB + nb) * 64 + lane];
#pragma unroll
                for (int g = 0; g < PG; ++g) {
                    acc[g][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq[g][j], acc[g][nb], 0, 0, 0);
                    acc[g][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bq[g][j], acc[g][nb], 0, 0, 0);
                }
            }
            if (s + BD < KT) {
#pragma unroll
                for (int g = 0; g < PG; ++g) bq[g][j] = load_b(s + BD, g);
            }
            if (s + 1 < KT) {
                half8* nxt = wl + ((s + 1) & 1) * F * 64;
#pragma unroll
                for (int i = 0; i < SH; ++i) if (wave + 4 * i < F) nxt[(wave + 4 * i) * 64 + lane] = stg[i];
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int g = 0; g < PG; ++g) {
        if (!valid[g]) continue;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n0 = 32 * (nb0 + nb) + 16 * u + 8 * h;
                if (n0 + 8 <= N) {
                    f32x4 lo, hi;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { lo[j] = acc[g][nb][8 * u + j]; hi[j] = acc[g][nb][8 * u + 4 + j]; }
                    if (resid != nullptr) {
                        lo += *(const f32x4*)(resid + px[g] * N + n0);
                        hi += *(const f32x4*)(resid + px[g] * N + n0 + 4);
                    }
                    *(f32x4*)(y + px[g] * N + n0) = lo;
                    *(f32x4*)(y + px[g] * N + n0 + 4) = hi;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// Depthwise 3x3, zero padding (restormer_arch.py:84,106).  One thread = 8 consecutive pixels of one
// image row x 8 channels (an 8x8 register block): 3 rows x 10 input vectors (16 B each) per block.
//   in: fp16 NHWC [T][H][W][Cs];  w: fp32 [9][Cs] (tap-major), bias fp32 [Cs] or null.
// MODE 0 (qkv, Cs = 3C): channel groups of q,k -> pixel-minor [T][H][WB][2C][8] (rows padded with
//         zeros to WB*8 pixels so the Gram ignores them); v -> NHWC [T][H][W][C].
// MODE 1 (GDFN gate, Cs = 2*Hp): out[T][H][W][Hp] = gelu(dw(in[.., :Hp])) * dw(in[.., Hp:]) (:90-91).
template <int PX>
__device__ __forceinline__ void dw_block(const f16* __restrict__ in, const float* __restrict__ w,
                                         const float* __restrict__ bias, int H, int W, int Cs, int t,
                                         int y, int x0, int ch0, float (&acc)[PX][8]) {
#pragma unroll
    for (int p = 0; p < PX; ++p)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[p][e] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = y + dy - 1;
        if (yy < 0 || yy >= H) continue;
        const f16* rowp = in + ((long)(t * H + yy) * W) * Cs + ch0;
        half8 v[PX + 2];
#pragma unroll
        for (int i = 0; i < PX + 2; ++i) {
            const int xx = x0 - 1 + i;
            v[i] = (xx >= 0 && xx < W) ? *(const half8*)(rowp + (long)xx * Cs) : zero8();
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const float* wp = w + (long)(dy * 3 + dx) * Cs + ch0;
            const f32x4 wl = *(const f32x4*)wp, wh = *(const f32x4*)(wp + 4);
            // (v_fma_mix_f32: as (float)half * w + a the compiler emitted v_cvt_f32_f16 + v_pk_fma_f32 + moves, twice the vector instructions)
#pragma unroll
            for (int p = 0; p < PX; ++p) dw_tap(acc[p], __builtin_bit_cast(u32x4, v[p + dx]), wl, wh);
        }
    }
    if (bias != nullptr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float bv = bias[ch0 + e];
#pragma unroll
            for (int p = 0; p < PX; ++p) acc[p][e] += bv;
        }
    }
}

// Two output rows per thread: the four input rows y - 1 .. y + 2 are each loaded once and feed both outputs (3 row loads per output
// row become 2).  Per output the taps are still accumulated dy-major, dx-minor: bitwise the same as dw_block.
template <int PX>
__device__ __forceinline__ void dw_block2(const f16* __restrict__ in, const float* __restrict__ w,
                                          const float* __restrict__ bias, int H, int W, int Cs, int t,
                                          int y, int x0, int ch0, float (&acc)[2][PX][8]) {
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int p = 0; p < PX; ++p)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[o][p][e] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int yy = y + r - 1;
        if (yy < 0 || yy >= H) continue;
        const f16* rowp = in + ((long)(t * H + yy) * W) * Cs + ch0;
        half8 v[PX + 2];
#pragma unroll
        for (int i = 0; i < PX + 2; ++i) {
            const int xx = x0 - 1 + i;
            v[i] = (xx >= 0 && xx < W) ? *(const half8*)(rowp + (long)xx * Cs) : zero8();
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int dy = r - o;
            if (dy < 0 || dy > 2) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float* wp = w + (long)(dy * 3 + dx) * Cs + ch0;
                const f32x4 wl = *(const f32x4*)wp, wh = *(const f32x4*)(wp + 4);
#pragma unroll
                for (int p = 0; p < PX; ++p) dw_tap(acc[o][p], __builtin_bit_cast(u32x4, v[p + dx]), wl, wh);
            }
        }
    }
    if (bias != nullptr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float bv = bias[ch0 + e];
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int p = 0; p < PX; ++p) acc[o][p][e] += bv;
        }
    }
}

constexpr int DW_PX = 4;   // pixels per thread: 4 x 8 register block (8 x 8 ran at 1 wave/SIMD)

__global__ __launch_bounds__(256) void k_dwconv_qkv(const f16* __restrict__ in, const float* __restrict__ w,
                                                    const float* __restrict__ bias, f16* __restrict__ qk,
                                                    f16* __restrict__ v, int T, int H, int W, int C, int g_lo) {
    // channel groups g_lo .. 3C/8 - 1 (g_lo = 2C/8: the v third only, q and k being written by k_dwconv_qk_lines)
    const int WB = (W + 7) >> 3, XB = 2 * WB, G = ((3 * C) >> 3) - g_lo, GQK = (2 * C) >> 3, HP = (H + 1) >> 1;
    const long total = (long)T * HP * XB * G;
    const long idx = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int g = g_lo + (int)(idx % G);
    long rest = idx / G;
    const int xq = (int)(rest % XB); rest /= XB;     // 4-pixel block index along the (padded) row
    const int y0 = 2 * (int)(rest % HP);             // two output rows per thread
    const int t = (int)(rest / HP);
    const int x0 = xq * DW_PX;
    float acc2[2][DW_PX][8];
    dw_block2<DW_PX>(in, w, bias, H, W, 3 * C, t, y0, x0, g * 8, acc2);
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int y = y0 + o;
        if (y >= H) break;
        float (&acc)[DW_PX][8] = acc2[o];
        if (g < GQK) {
            // pixel-minor [T][H][WB][2C][8]: this thread owns half (4 pixels) of each 8-pixel vector
            f16* dst = qk + ((((long)(t * H + y) * WB + (xq >> 1)) * (2 * C)) + g * 8) * 8 + 4 * (xq & 1);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                half4 o4;
#pragma unroll
                for (int p = 0; p < DW_PX; ++p) o4[p] = (x0 + p < W) ? (f16)acc[p][e] : (f16)0.f;
                *(half4*)(dst + e * 8) = o4;
            }
        } else {
            f16* dst = v + ((long)(t * H + y) * W + x0) * C + (g - GQK) * 8;
#pragma unroll
            for (int p = 0; p < DW_PX; ++p) {
                if (x0 + p < W) {
                    half8 o8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o8[e] = (f16)acc[p][e];
                    *(half8*)(dst + (long)p * C) = o8;
                }
            }
        }
    }
}

// q and k of the depthwise 3x3 in the pixel-minor layout [T][H][WB][2C][8], one thread = one (row, 8-pixel block, channel octet) = eight
// 16-byte vectors = exactly ONE 128-byte line of the output, and consecutive work items own consecutive lines: the tensor is a linear
// function of the item index.  k_dwconv_qkv's 4-pixel threads wrote that line as sixteen 8-byte pieces from two threads (64 lines per store
// instruction); here a wave's 64 lines (8 KB) are transposed through LDS (slot XOR-swizzled by the lane's low bits: conflict-free both
// ways) and leave as eight fully contiguous 1 KB stores.  Same tap order as dw_block: bitwise the same values.
__global__ __launch_bounds__(256) void k_dwconv_qk_lines(const f16* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                                                         f16* __restrict__ qk, int T, int H, int W, int C) {
    __shared__ __attribute__((aligned(16))) half8 stage[4][512];
    const int WB = (W + 7) >> 3, GQK = (2 * C) >> 3;
    const long total = (long)T * H * WB * GQK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long item = xcd_swizzle(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    const long it = item < total ? item : total - 1;
    const int g = (int)(it % GQK);
    long rest = it / GQK;
    const int xb = (int)(rest % WB); rest /= WB;
    const int y = (int)(rest % H);
    const int t = (int)(rest / H);
    float acc[8][8];
    dw_block<8>(in, w, bias, H, W, 3 * C, t, y, 8 * xb, g * 8, acc);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        half8 o;
#pragma unroll
        for (int p = 0; p < 8; ++p) o[p] = (8 * xb + p < W) ? (f16)acc[p][e] : (f16)0.f;
        stage[wave][lane * 8 + (e ^ (lane & 7))] = o;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long item0 = item - lane;                                  // the wave's first item: its lines start at half8 index 8 * item0
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int slot = 64 * i + lane, l2 = slot >> 3, e2 = slot & 7;
        const half8 o = stage[wave][l2 * 8 + (e2 ^ (l2 & 7))];
        if (item0 + l2 < total) ((half8*)qk)[item0 * 8 + slot] = o;
    }
}

template <bool TAB>
__global__ __launch_bounds__(256) void k_dwconv_gate(const f16* __restrict__ in, const float* __restrict__ w,
                                                     const float* __restrict__ bias, f16* __restrict__ out,
                                                     int T, int H, int W, int Hp, const float2* __restrict__ phitab) {
    // TAB: GELU as |a| Phi(|a|) + min(a, 0) with Phi from the 1024-entry (value, slope) table of ira_ffn2.hip staged in LDS: 10 vector
    // instructions per value instead of the ~20 (one v_rcp, one v_exp) of gelu_erf, which cost as much as the nine taps
    __shared__ float2 tab[TAB ? 1024 : 1];
    if constexpr (TAB) {
        for (int i = threadIdx.x; i < 512; i += 256) ((f32x4*)tab)[i] = ((const f32x4*)phitab)[i];
        __syncthreads();
    }
    // A workgroup = 64 consecutive (4-pixel block, channel octet) items of FOUR consecutive rows, one row per wave: the six input rows the
    // band needs are then shared through L1 by waves that run together (with one row per workgroup every input row came from L2 three
    // times: 9 KB of L2 reads per pixel for 2 KB of data - the kernel ran at the L2's bandwidth, not HBM's)
    const int XB = (W + DW_PX - 1) / DW_PX, G = Hp >> 3;
    const int row_items = XB * G, chunks = (row_items + 63) >> 6, bands = (H + 3) >> 2;
    const long bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int chunk = (int)(bid % chunks);
    const int band = (int)((bid / chunks) % bands);
    const int t = (int)(bid / ((long)chunks * bands));
    const int item = chunk * 64 + (threadIdx.x & 63);
    const int y = band * 4 + (threadIdx.x >> 6);
    if (item >= row_items || y >= H) return;
    const int g = item % G, xq = item / G;
    const int x0 = xq * DW_PX;
    float a1[DW_PX][8], a2[DW_PX][8];
    dw_block<DW_PX>(in, w, bias, H, W, 2 * Hp, t, y, x0, g * 8, a1);
    dw_block<DW_PX>(in, w, bias, H, W, 2 * Hp, t, y, x0, Hp + g * 8, a2);
    f16* dst = out + ((long)(t * H + y) * W + x0) * Hp + g * 8;
#pragma unroll
    for (int p = 0; p < DW_PX; ++p) {
        if (x0 + p < W) {
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if constexpr (TAB) {
                    const float a = a1[p][e], aa = fabsf(a);
                    const float tt = fminf(aa * (1024.0f / 6.0f), 1024.0f - 0.004f);
                    const float2 en = tab[(int)tt];
                    const float ph = fmaf(__builtin_amdgcn_fractf(tt), en.y, en.x);
                    o[e] = (f16)(fmaf(aa, ph, fminf(a, 0.f)) * a2[p][e]);
                } else {
                    o[e] = (f16)(gelu_erf(a1[p][e]) * a2[p][e]);
                }
            }
            *(half8*)(dst + (long)p * Hp) = o;
        }
    }
}

// The same gate with the depthwise WEIGHTS in LDS.  k_dwconv_gate's lanes are channel octets, so every tap's 16 weights are two 16-byte
// loads per lane: as many load instructions as the pixels themselves (36 + 36 per 4 output pixels), and 36 KB of weights (2 Hp = 1024
// channels) do not stay in the 32 KB L1.  Here a workgroup stages the weights of its 64 channel octets once (18 taps x 64 x 32 B = 36 KB,
// lane-contiguous: conflict-free reads) and its four waves (one image row each) walk along the row in 4-pixel steps.
template <bool TAB>
__global__ __launch_bounds__(256) void k_dwconv_gate_row(const f16* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                                                         f16* __restrict__ out, int T, int H, int W, int Hp, int nseg, const float2* __restrict__ phitab) {
    __shared__ __attribute__((aligned(16))) float wsm[18][64][8];
    __shared__ float2 tab[TAB ? 1024 : 1];
    const int G = Hp >> 3, gchunks = (G + 63) >> 6, bands = (H + 3) >> 2, Cs = 2 * Hp;
    const long bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int seg = (int)(bid % nseg);
    const int gchunk = (int)((bid / nseg) % gchunks);
    const int band = (int)((bid / ((long)nseg * gchunks)) % bands);
    const int t = (int)(bid / ((long)nseg * gchunks * bands));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 18 * 64 * 2; i += 256) {               // 16-byte pieces: [tap18][octet][half of the octet]
        const int hf = i & 1, gl = (i >> 1) & 63, tp = i >> 7;
        const int gg = min(gchunk * 64 + gl, G - 1);
        const int ch = (tp < 9 ? 0 : Hp) + 8 * gg + 4 * hf;
        *(f32x4*)&wsm[tp][gl][4 * hf] = *(const f32x4*)(w + (long)(tp < 9 ? tp : tp - 9) * Cs + ch);
    }
    if constexpr (TAB) {
        for (int i = threadIdx.x; i < 512; i += 256) ((f32x4*)tab)[i] = ((const f32x4*)phitab)[i];
    }
    __syncthreads();
    const int g = gchunk * 64 + lane, y = band * 4 + wave;
    if (g >= G || y >= H) return;
    const int XB = (W + 3) >> 2, per = (XB + nseg - 1) / nseg;
    const int xq_lo = seg * per, xq_hi = min(XB, xq_lo + per);
    float b1[8], b2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { b1[e] = bias ? bias[8 * g + e] : 0.f; b2[e] = bias ? bias[Hp + 8 * g + e] : 0.f; }
    for (int xq = xq_lo; xq < xq_hi; ++xq) {
        const int x0 = 4 * xq;
        float a1[4][8], a2[4][8];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int e = 0; e < 8; ++e) { a1[p][e] = 0.f; a2[p][e] = 0.f; }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
            if (yy < 0 || yy >= H) continue;
            const f16* rowp = in + ((long)(t * H + yy) * W) * Cs + 8 * g;
            half8 v1[6], v2[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int xx = x0 - 1 + i;
                const bool ok = xx >= 0 && xx < W;
                v1[i] = ok ? *(const half8*)(rowp + (long)xx * Cs) : zero8();
                v2[i] = ok ? *(const half8*)(rowp + (long)xx * Cs + Hp) : zero8();
            }
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const f32x4 w1l = *(const f32x4*)&wsm[dy * 3 + dx][lane][0], w1h = *(const f32x4*)&wsm[dy * 3 + dx][lane][4];
                const f32x4 w2l = *(const f32x4*)&wsm[9 + dy * 3 + dx][lane][0], w2h = *(const f32x4*)&wsm[9 + dy * 3 + dx][lane][4];
                // v_fma_mix_f32 (fp16 source, fp32 weight and sum: the same arithmetic): written as (float)half * w + a the compiler made
                // v_cvt_f32_f16 + v_pk_fma_f32 + moves of it, twice the vector instructions (r3: 0.53 -> see DESIGN 5 ms per level-3 launch)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    dw_tap(a1[p], __builtin_bit_cast(u32x4, v1[p + dx]), w1l, w1h);
                    dw_tap(a2[p], __builtin_bit_cast(u32x4, v2[p + dx]), w2l, w2h);
                }
            }
        }
        f16* dst = out + ((long)(t * H + y) * W + x0) * Hp + g * 8;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (x0 + p < W) {
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = a1[p][e] + b1[e], bb = a2[p][e] + b2[e];
                    if constexpr (TAB) {
                        const float aa = fabsf(a);
                        const float tt = fminf(aa * (1024.0f / 6.0f), 1024.0f - 0.004f);
                        const float2 en = tab[(int)tt];
                        const float ph = fmaf(__builtin_amdgcn_fractf(tt), en.y, en.x);
                        o[e] = (f16)(fmaf(aa, ph, fminf(a, 0.f)) * bb);
                    } else {
                        o[e] = (f16)(gelu_erf(a) * bb);
                    }
                }
                *(half8*)(dst + (long)p * Hp) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Partial Gram q k^T and squared norms over pixels (restormer_arch.py:121-124), K-dim = pixels.
// qk: pixel-minor fp16 [T][nblk][2C][8]: for every channel 8 consecutive pixels are one 16-byte vector,
// which is exactly the A/B fragment of v_mfma_f32_16x16x32_f16 (lane l: row/col l&15, k = 8(l>>4)+j).
// One wave = (tile, head, split s): strides over the tile's 32-pixel steps s, s+S, ...
// part: fp32 [T][S][heads][c*c + 2c]: gram[qi][kj], then sum q^2 [c], then sum k^2 [c].
template <int CT, int PER>
__global__ __launch_bounds__(256) void k_gram(const f16* __restrict__ qk, float* __restrict__ part, int T,
                                              int nblk, int C, int heads, int S, int bands) {
    // bands = CT / PER: the q-channel tiles of one (tile, head, split) are shared out over `bands` waves, PER tiles each.  Every
    // output element still accumulates the same pixels in the same order, so the result does not depend on `bands`; the launcher
    // uses it to fill the chip when there are few tiles (a whole 720p image at heads = 1 had 64 waves for 1024 SIMDs).  PER is a
    // template parameter so that a wave only holds the accumulators of its own rows: with the full CT x CT set (144 registers
    // at c = 96) the banded waves of a 5-tile shard ran two per SIMD, and the step's load -> MFMA chain had nothing to hide behind.
    constexpr int c = CT * 16;
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const long wv0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv0 >= (long)T * heads * S * bands) return;
    const int band = (int)(wv0 % bands);
    const long wv = wv0 / bands;
    const int s = (int)(wv % S);
    const int hd = (int)((wv / S) % heads);
    const int t = (int)(wv / ((long)S * heads));
    const int i_lo = band * PER;

    f32x4 acc[PER][CT];
    float nq[PER], nk[CT];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        nq[i] = 0.f;
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < CT; ++j) nk[j] = 0.f;
    const int nsteps = (nblk + 3) >> 2;
    const f16* base = qk + (long)t * nblk * (2 * C) * 8;
    for (int step = s; step < nsteps; step += S) {
        const int blk = step * 4 + lg;
        const bool ok = blk < nblk;
        const f16* bp = base + ((long)(ok ? blk : 0) * (2 * C) + hd * c + li) * 8;
        half8 a[PER], b[CT];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            a[i] = ok ? *(const half8*)(bp + (long)(16 * (i_lo + i)) * 8) : zero8();
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float qa = (float)a[i][j]; nq[i] = fmaf(qa, qa, nq[i]); }
        }
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            b[i] = ok ? *(const half8*)(bp + (long)(C + 16 * i) * 8) : zero8();
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float kb = (float)b[i][j]; nk[i] = fmaf(kb, kb, nk[i]); }
        }
#pragma unroll
        for (int i = 0; i < PER; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float* dst = part + (((long)t * S + s) * heads + hd) * (c * c + 2 * c);
    // D layout 16x16: col = lane & 15 (k channel), row = 4*(lane>>4) + reg (q channel)
#pragma unroll
    for (int i = 0; i < PER; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(16 * (i_lo + i) + 4 * lg + e) * c + 16 * j + li] = acc[i][j][e];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        float a = nq[i];
        a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
        if (lg == 0) dst[c * c + 16 * (i_lo + i) + li] = a;
    }
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        float b = nk[i];
        b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
        if (lg == 0 && band == 0) dst[c * c + c + 16 * i + li] = b;
    }
}

__global__ __launch_bounds__(256) void k_attn_softmax(const float* __restrict__ part, const float* __restrict__ temp,
                                                      float* __restrict__ attn, int T, int C, int heads, int S) {
    __shared__ float L[8 * 96];
    __shared__ float nks[96];
    __shared__ float nqs[8];
    const int c = C / heads;
    const int hd = blockIdx.x % heads, t = blockIdx.x / heads;
    const int i0 = blockIdx.y * 8;
    const int nrows = min(8, c - i0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rec = c * c + 2 * c;
    const float* pb = part + ((long)t * S * heads + hd) * rec;
    const long sstride = (long)heads * rec;
    // sum of the S partial records in a FIXED order (bitwise reproducible), eight independent chains so that the loads of a
    // long reduction (S up to 512) overlap instead of waiting for one another
    auto sum_parts = [&](long off) -> float {
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 8 <= S; s += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += pb[(long)(s + u) * sstride + off];
        }
        for (; s < S; ++s) a[0] += pb[(long)s * sstride + off];
        return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    };
    for (int idx = tid; idx < nrows * c; idx += 256) L[idx] = sum_parts((long)i0 * c + idx);
    for (int idx = tid; idx < c; idx += 256) nks[idx] = sum_parts((long)c * c + c + idx);
    if (tid < nrows) nqs[tid] = sum_parts((long)c * c + i0 + tid);
    __syncthreads();
    const float tmp = temp[hd];
    for (int row = wave; row < nrows; row += 4) {
        const float qn = fmaxf(sqrtf(nqs[row]), 1e-12f);
        const int j0 = lane, j1 = lane + 64;
        float l0 = -INFINITY, l1 = -INFINITY;
        if (j0 < c) l0 = L[row * c + j0] / (qn * fmaxf(sqrtf(nks[j0]), 1e-12f)) * tmp;
        if (j1 < c) l1 = L[row * c + j1] / (qn * fmaxf(sqrtf(nks[j1]), 1e-12f)) * tmp;
        const float mx = wave_max(fmaxf(l0, l1));
        const float e0 = (j0 < c) ? expf(l0 - mx) : 0.f;
        const float e1 = (j1 < c) ? expf(l1 - mx) : 0.f;
        const float inv = 1.0f / wave_sum(e0 + e1);
        float* dst = attn + (((long)t * heads + hd) * c + i0 + row) * c;
        if (j0 < c) dst[j0] = e0 * inv;
        if (j1 < c) dst[j1] = e1 * inv;
    }
}

__global__ __launch_bounds__(256) void k_attn_fold(const float* __restrict__ attn, const float* __restrict__ wo,
                                                   f16* __restrict__ M, int T, int C, int heads, int NB, int lo_zero) {
    extern __shared__ __attribute__((aligned(16))) float A[];   // [c][c]
    const int c = C / heads;
    const int hd = blockIdx.x % heads, t = blockIdx.x / heads;
    const int o0 = blockIdx.y * 16;
    const int tid = threadIdx.x;
    const float* src = attn + ((long)t * heads + hd) * c * c;
    for (int idx = tid; idx < c * c; idx += 256) A[idx] = src[idx];
    __syncthreads();
    // M is written fragment-major [2][NB][C/16][64][8]: element (row o, col k) lives at
    // nb = o/32, lane = 32*((k%16)/8) + rowperm(o%32) (rowperm is an involution), s = k/16, j = k%8
    const int KT = C >> 4;
    f16* Mt = M + (long)t * 2 * NB * KT * 512;
    f16* Ml = Mt + (long)NB * KT * 512;
    for (int idx = tid; idx < 16 * c; idx += 256) {
        const int o = o0 + idx / c, j = idx % c;
        if (o >= C) continue;
        const float* wr = wo + (long)o * C + hd * c;
        float a = 0.f;
        for (int i = 0; i < c; ++i) a = fmaf(wr[i], A[i * c + j], a);
        const int k = hd * c + j;
        const long off = (((long)(o >> 5) * KT + (k >> 4)) * 64 + 32 * ((k >> 3) & 1) + rowperm(o & 31)) * 8 + (k & 7);
        const f16 hi = (f16)a;
        Mt[off] = hi;
        Ml[off] = lo_zero ? (f16)0.f : (f16)(a - (float)hi);      // lo_zero: the apply of this level runs on the hi plane alone (ira_set_weight_planes)
    }
}

// ---------------------------------------------------------------------------------------------
// Dense 3x3 conv, zero padding, as 9 shifted 1x1 MFMA contractions (implicit GEMM):
// OverlapPatchEmbed is k_embed below; this one serves Downsample / Upsample / output
// (restormer_arch.py:171-189,243,281).   in fp32 NHWC [T][H][Wd][K]; W fp16 [9][NBtot*32][K]; out fp32.
// MODE 0 plain [T][H][Wd][N]; 1 PixelUnshuffle(2) -> [T][H/2][Wd/2][4N]; 2 PixelShuffle(2) ->
// [T][2H][2Wd][N/4]; 3 plain + skip (the "+ inp_img" of :281); 4 plain + LeakyReLU(slope) (slope 0 = ReLU: DnCNN body; 0.2 / 0.01: SwinIR).
// W: fragment-major [2][9][NBtot][K/16][64][8] (hi, lo planes).  The fp32 input is split into fp16 hi + lo in registers as well
// (three MFMAs per step: Wh.xh + Wl.xh + Wh.xl), so the conv sees ~fp32 operands at no extra traffic.
template <int NBW, int MODE>
__global__ __launch_bounds__(256) void k_conv3x3(const float* __restrict__ in, const f16* __restrict__ W,
                                                 const float* __restrict__ bias, const float* __restrict__ skip,
                                                 float* __restrict__ out, int T, int H, int Wd, int K, int N,
                                                 int NBtot, float slope, int ldin, int ldout) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int P = H * Wd;
    const int chunks = (P + 31) >> 5;
    const long wv = xcd_swizzle(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    if (wv >= (long)T * chunks) return;
    const int nb0 = blockIdx.y * NBW;
    const int t = (int)(wv / chunks);
    const int p = (int)(wv % chunks) * 32 + r;
    const bool valid = p < P;
    const int pc = valid ? p : P - 1;
    const int y = pc / Wd, x = pc % Wd;

    f32x16 acc[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        if (bias != nullptr) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = 32 * (nb0 + nb) + 16 * u + 8 * h + j;
                    acc[nb][8 * u + j] = (n < N) ? bias[n] : 0.f;
                }
        } else {
            acc[nb] = zero16();
        }
    }
    const int KT = K >> 4;
    const f16* wbase = W + lane * 8;                 // fragment-major [2][9][NBtot][KT][64][8]
    const long lo_off = 9L * NBtot * KT * 512;
    for (int s = 0; s < KT; ++s) {                   // k-step outermost, taps innermost: see k_conv3x3_lds
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            const bool inb = yy >= 0 && yy < H && xx >= 0 && xx < Wd;
            const float* src = in + ((long)(t * H + (inb ? yy : y)) * Wd + (inb ? xx : x)) * ldin + 8 * h;
            const f16* wt = wbase + (long)tap * NBtot * KT * 512;
            f32x4 lo4 = *(const f32x4*)(src + 16 * s), hi4 = *(const f32x4*)(src + 16 * s + 4);
            if (!inb) { lo4 = f32x4{0.f, 0.f, 0.f, 0.f}; hi4 = lo4; }
            const half8 b = cvt8(lo4, hi4);
            half8 bl;
#pragma unroll
            for (int j = 0; j < 4; ++j) { bl[j] = (f16)(lo4[j] - (float)b[j]); bl[4 + j] = (f16)(hi4[j] - (float)b[4 + j]); }
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                if (nb0 + nb < NBtot) {
                    const half8 a = *(const half8*)(wt + ((long)(nb0 + nb) * KT + s) * 512);
                    const half8 al = *(const half8*)(wt + lo_off + ((long)(nb0 + nb) * KT + s) * 512);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b, acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bl, acc[nb], 0, 0, 0);
                }
            }
        }
    }
    if (!valid) return;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = 32 * (nb0 + nb) + 16 * u + 8 * h + j;
                if (n >= N) continue;
                const float val = acc[nb][8 * u + j];
                if constexpr (MODE == 0) {
                    out[((long)t * P + p) * ldout + n] = val;
                } else if constexpr (MODE == 4) {
                    out[((long)t * P + p) * ldout + n] = val > 0.f ? val : slope * val;
                } else if constexpr (MODE == 1) {
                    const int oc = n * 4 + (y & 1) * 2 + (x & 1);
                    out[(((long)t * (H >> 1) + (y >> 1)) * (Wd >> 1) + (x >> 1)) * (4 * N) + oc] = val;
                } else if constexpr (MODE == 2) {
                    const int cc = n >> 2, sy = (n >> 1) & 1, sx = n & 1;
                    out[(((long)t * (2 * H) + 2 * y + sy) * (2 * Wd) + 2 * x + sx) * ldout + cc] = val;      // MODE 2: ldout = floats per OUTPUT pixel (>= N / 4)
                } else {
                    const long o = ((long)t * P + p) * ldout + n;
                    out[o] = val + skip[o];
                }
            }
}

// LDS-shared-weight variant of k_conv3x3 for the wide convs (Upsample 384->768 etc.): per (tap, k-step) the
// 4 waves of a workgroup stage the slab of NBW row blocks x {hi, lo} fragments in LDS once (one step ahead) instead of
// each wave streaming the whole 9*K*N weight tensor (up to 10 MB per 32 pixels) from L2.
template <int NBW, int MODE>
__global__ __launch_bounds__(256) void k_conv3x3_lds(const float* __restrict__ in, const f16* __restrict__ W,
                                                     const float* __restrict__ bias, const float* __restrict__ skip,
                                                     float* __restrict__ out, int T, int H, int Wd, int K, int N,
                                                     int NBtot, float slope, int ldin, int ldout) {
    constexpr int F = 2 * NBW;
    constexpr int SH = (F + 3) / 4;
    __shared__ __attribute__((aligned(16))) half8 wl[2][F][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int P = H * Wd;
    const int chunks = (P + 31) >> 5;
    const long total = (long)T * chunks;
    long wv = xcd_swizzle(blockIdx.x, gridDim.x) * 4 + wave;
    const bool live = wv < total;
    if (!live) wv = total - 1;
    const int nb0 = blockIdx.y * NBW;
    const int t = (int)(wv / chunks);
    const int p = (int)(wv % chunks) * 32 + r;
    const bool valid = live && p < P;
    const int pc = (p < P) ? p : P - 1;
    const int y = pc / Wd, x = pc % Wd;

    f32x16 acc[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        acc[nb] = zero16();
        if (bias != nullptr) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = 32 * (nb0 + nb) + 16 * u + 8 * h + j;
                    acc[nb][8 * u + j] = (n < N) ? bias[n] : 0.f;
                }
        }
    }
    const int KT = K >> 4;
    const int nsteps = 9 * KT;
    const long lo_off = 9L * NBtot * KT * 512;
    // fragment f of step (tap, s): f < NBW -> hi plane, row block nb0+f; else lo plane, row block nb0+f-NBW
    // step order: k-step outermost, the 9 taps innermost.  With taps outermost every tap walked all K input channels of
    // its shifted pixel before the next tap touched the neighbouring pixel's lines again: 9 passes over rows of 32 px x K
    // fp32 (128 KB per wave at K = 256) that neither L1 nor, for wide images, L2 could hold (PMC: 4.9 GB read per launch for
    // a 0.54 GB input on the SwinIR 256 -> 64 conv).  Taps innermost touch 3 rows x 34 px x 64 B per k-step.
    auto gsrc = [&](int step, int f) -> const half8* {
        const int s = step / 9, tap = step - 9 * s;
        int nb = nb0 + (f < NBW ? f : f - NBW);
        if (nb >= NBtot) nb = NBtot - 1;                      // partial last chunk: staged but never used
        return (const half8*)(W + (f < NBW ? 0 : lo_off) + (((long)tap * NBtot + nb) * KT + s) * 512 + lane * 8);
    };
    half8 stg[SH];
#pragma unroll
    for (int i = 0; i < SH; ++i) if (wave + 4 * i < F) stg[i] = *gsrc(0, wave + 4 * i);
#pragma unroll
    for (int i = 0; i < SH; ++i) if (wave + 4 * i < F) wl[0][wave + 4 * i][lane] = stg[i];
    __syncthreads();

    for (int step = 0; step < nsteps; ++step) {
        const int s = step / 9, tap = step - 9 * s;
        if (step + 1 < nsteps) {
#pragma unroll
            for (int i = 0; i < SH; ++i) if (wave + 4 * i < F) stg[i] = *gsrc(step + 1, wave + 4 * i);
        }
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        const bool inb = yy >= 0 && yy < H && xx >= 0 && xx < Wd;
        const float* src = in + ((long)(t * H + (inb ? yy : y)) * Wd + (inb ? xx : x)) * ldin + 8 * h + 16 * s;
        f32x4 lo4 = *(const f32x4*)src, hi4 = *(const f32x4*)(src + 4);
        if (!inb) { lo4 = f32x4{0.f, 0.f, 0.f, 0.f}; hi4 = lo4; }
        const half8 b = cvt8(lo4, hi4);
        half8 bl;
#pragma unroll
        for (int j = 0; j < 4; ++j) { bl[j] = (f16)(lo4[j] - (float)b[j]); bl[4 + j] = (f16)(hi4[j] - (float)b[4 + j]); }
        const int cur = step & 1;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            if (nb0 + nb < NBtot) {
                const half8 a = wl[cur][nb][lane], al = wl[cur][NBW + nb][lane];
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bl, acc[nb], 0, 0, 0);
            }
        }
        if (step + 1 < nsteps) {
#pragma unroll
            for (int i = 0; i < SH; ++i) if (wave + 4 * i < F) wl[cur ^ 1][wave + 4 * i][lane] = stg[i];
        }
        __syncthreads();
    }
    if (!valid) return;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = 32 * (nb0 + nb) + 16 * u + 8 * h + j;
                if (n >= N) continue;
                const float val = acc[nb][8 * u + j];
                if constexpr (MODE == 0) {
                    out[((long)t * P + p) * ldout + n] = val;
                } else if constexpr (MODE == 4) {
                    out[((long)t * P + p) * ldout + n] = val > 0.f ? val : slope * val;
                } else if constexpr (MODE == 1) {
                    const int oc = n * 4 + (y & 1) * 2 + (x & 1);
                    out[(((long)t * (H >> 1) + (y >> 1)) * (Wd >> 1) + (x >> 1)) * (4 * N) + oc] = val;
                } else if constexpr (MODE == 2) {
                    const int cc = n >> 2, sy = (n >> 1) & 1, sx = n & 1;
                    out[(((long)t * (2 * H) + 2 * y + sy) * (2 * Wd) + 2 * x + sx) * ldout + cc] = val;      // MODE 2: ldout = floats per OUTPUT pixel (>= N / 4)
                } else {
                    const long o = ((long)t * P + p) * ldout + n;
                    out[o] = val + skip[o];
                }
            }
}

// ---------------------------------------------------------------------------------------------
// Dense 3x3 with the PIXEL operand staged through LDS as well.  k_conv3x3_lds loads and splits (fp32 -> fp16 hi + lo) every
// pixel's 16 channels of a k-step once per TAP: nine times the loads and nine times the conversion VALU work of the 3 * NBW MFMAs
// they feed.  Here a workgroup owns an 8 x 16 pixel tile (wave w: rows 2w, 2w + 1): per k-step the 10 x 18 halo tile is read and
// split ONCE into LDS ([plane][k half][pixel] x 16 B: a wave's B fragment of any tap is 2 runs of 16 consecutive 16-byte slots,
// conflict-free), loaded a k-step ahead of its use; the nine taps then read their fragments at shifted tile offsets.  Same
// operands, same MFMA order as k_conv3x3_lds: bitwise the same output.
template <int NBW, int PG, int MODE>
__global__ __launch_bounds__(256) void k_conv3x3_tile(const float* __restrict__ in, const f16* __restrict__ W,
                                                      const float* __restrict__ bias, const float* __restrict__ skip,
                                                      float* __restrict__ out, int T, int H, int Wd, int K, int N,
                                                      int NBtot, float slope, int ldin, int ldout) {
    constexpr int F = 2 * NBW;
    constexpr int SH = (F + 3) / 4;
    constexpr int TH = 8 * PG, TW = 16, HX = TW + 2, NQ = (TH + 2) * HX;   // PG pixel groups per wave: rows 2w, 2w + 1 of each 8-row band
    constexpr int NE = NQ * 4, BI = (NE + 255) / 256;                  // float4 elements of one k-step of the halo tile
    typedef half8 (*WlT)[F][64];
    typedef half8 (*BtT)[2][2][NQ];
    constexpr size_t WL_BYTES = sizeof(half8) * 2 * F * 64, BT_BYTES = sizeof(half8) * 2 * 2 * 2 * NQ;
    __shared__ __attribute__((aligned(16))) char lds_all[WL_BYTES + BT_BYTES];
    WlT wl = (WlT)lds_all;                                             // [2][F][64]
    BtT bt = (BtT)(lds_all + WL_BYTES);                                // [buffer][plane hi / lo][k half][halo pixel]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int tiles_x = (Wd + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const long tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int t = (int)(tile / ((long)tiles_y * tiles_x));
    const int trem = (int)(tile - (long)t * tiles_y * tiles_x);
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
    const int py = 2 * wave + (r >> 4), pxx = r & 15;
    const int x = x0 + pxx;
    const int P = H * Wd;
    const int nb0 = blockIdx.y * NBW;

    f32x16 acc[PG][NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        acc[0][nb] = zero16();
        if (bias != nullptr) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = 32 * (nb0 + nb) + 16 * u + 8 * h + j;
                    acc[0][nb][8 * u + j] = (n < N) ? bias[n] : 0.f;
                }
        }
#pragma unroll
        for (int g = 1; g < PG; ++g) acc[g][nb] = acc[0][nb];
    }
    // The accumulators were just filled by global loads (bias).  Without an explicit wait here the compiler's wait-count
    // analysis carries "accumulators may be pending" around the loop's back edge and puts s_waitcnt vmcnt(0) in front of
    // the first MFMA of EVERY step: each step then waited for the weight fragments it had just requested for the NEXT
    // step (a full L2 round trip per 12 MFMAs; SQ counters: MFMA pipe 37 % busy, waves 39 % in s_waitcnt).
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), expcnt / lgkmcnt untouched
    const int KT = K >> 4;
    const int nsteps = 9 * KT;
    const long lo_off = 9L * NBtot * KT * 512;
    auto gsrc = [&](int step, int f) -> const half8* {
        const int s = step / 9, tap = step - 9 * s;
        int nb = nb0 + (f < NBW ? f : f - NBW);
        if (nb >= NBtot) nb = NBtot - 1;
        return (const half8*)(W + (f < NBW ? 0 : lo_off) + (((long)tap * NBtot + nb) * KT + s) * 512 + lane * 8);
    };
    // halo-tile element e of a k-step: pixel q = e / 4 (row q / 18 - 1, column q % 18 - 1 of the tile), channels 4 * (e % 4) .. + 3
    f32x4 bst[BI];
    auto bload = [&](int s) {
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int e = threadIdx.x + 256 * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < NE) {
                const int q = e >> 2, c4 = e & 3;
                const int yy = y0 + q / HX - 1, xx = x0 + q % HX - 1;
                if (yy >= 0 && yy < H && xx >= 0 && xx < Wd) v = *(const f32x4*)(in + ((long)(t * H + yy) * Wd + xx) * ldin + 16 * s + 4 * c4);
            }
            bst[i] = v;
        }
    };
    auto bstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int e = threadIdx.x + 256 * i;
            if (e < NE) {
                const int q = e >> 2, c4 = e & 3;
                half4 hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) { hi[j] = (f16)bst[i][j]; lo[j] = (f16)(bst[i][j] - (float)hi[j]); }
                half4* dh = (half4*)&bt[buf][0][c4 >> 1][q] + (c4 & 1);
                half4* dl = (half4*)&bt[buf][1][c4 >> 1][q] + (c4 & 1);
                *dh = hi; *dl = lo;
            }
        }
    };
    // weight fragments travel global -> registers -> LDS TWO steps ahead of their use: one step of MFMAs (~400 cycles) is shorter
    // than an L2 round trip, so with a one-step distance the ds_write at the end of every step waited for the loads of its start
    half8 nxt[SH], ld[SH];
#pragma unroll
    for (int i = 0; i < SH; ++i) if ((F % 4 == 0) || wave + 4 * i < F) ld[i] = *gsrc(0, wave + 4 * i);
#pragma unroll
    for (int i = 0; i < SH; ++i) if ((F % 4 == 0) || wave + 4 * i < F) nxt[i] = *gsrc(nsteps > 1 ? 1 : 0, wave + 4 * i);
    bload(0);
#pragma unroll
    for (int i = 0; i < SH; ++i) if ((F % 4 == 0) || wave + 4 * i < F) wl[0][wave + 4 * i][lane] = ld[i];
    bstore(0);
    __syncthreads();

    for (int s = 0; s < KT; ++s) {
        const int bb = s & 1;
        if (s + 1 < KT) bload(s + 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int step = 9 * s + tap;
            if (step + 2 < nsteps) {
#pragma unroll
                for (int i = 0; i < SH; ++i) if ((F % 4 == 0) || wave + 4 * i < F) ld[i] = *gsrc(step + 2, wave + 4 * i);
            }
            const int q = (py + tap / 3) * HX + pxx + tap % 3;
            half8 b[PG], bl[PG];
#pragma unroll
            for (int g = 0; g < PG; ++g) { b[g] = bt[bb][0][h][q + 8 * g * HX]; bl[g] = bt[bb][1][h][q + 8 * g * HX]; }
            const int cur = step & 1;
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {       // row blocks past NBtot (partial last chunk) compute on a clamped copy, never stored:
                const half8 a = wl[cur][nb][lane], al = wl[cur][NBW + nb][lane];   // no branch in the loop, so reads and MFMAs interleave
#pragma unroll
                for (int g = 0; g < PG; ++g) {
                    acc[g][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[g], acc[g][nb], 0, 0, 0);
                    acc[g][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b[g], acc[g][nb], 0, 0, 0);
                    acc[g][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bl[g], acc[g][nb], 0, 0, 0);
                }
            }
            if (step + 1 < nsteps) {
#pragma unroll
                for (int i = 0; i < SH; ++i) if ((F % 4 == 0) || wave + 4 * i < F) wl[cur ^ 1][wave + 4 * i][lane] = nxt[i];
            }
#pragma unroll
            for (int i = 0; i < SH; ++i) nxt[i] = ld[i];
            if (tap == 8 && s + 1 < KT) bstore(bb ^ 1);
            __syncthreads();
        }
    }
    if constexpr (MODE == 2 && PG == 1 && NBW % 2 == 0) if ((N & 15) == 0) {
        // PixelShuffle(2) epilogue through LDS.  Output channel n of input pixel (y, x) is channel n >> 2 of output pixel (2y + (n >> 1 & 1),
        // 2x + (n & 1)): straight from the accumulators that is one 4-byte store per value, 64 different cache lines per store
        // instruction (96 instructions per lane at 6 row blocks - the up-convolutions spent most of their time here).  Each wave owns
        // 2 x 16 input pixels = 4 x 32 output pixels; two row blocks at a time (16 output channels = 64 B per output pixel) go through the
        // wave's own 9 KB of the (now dead) weight / pixel buffers and leave as 16-byte stores, 4 lanes per output pixel.
        constexpr int PS = 18;                                   // floats per staged output pixel (16 + 2: the 16 lanes of a row hit 8 bank groups)
        static_assert(WL_BYTES + BT_BYTES >= 4 * 128 * PS * 4, "staging must fit the main loop's buffers");
        float* stg = (float*)lds_all + wave * 128 * PS;
        const int N4 = N >> 2;
        const int yo0 = 2 * (y0 + 2 * wave), xo0 = 2 * x0;       // first output row / column of this wave
#pragma unroll
        for (int pr = 0; pr < NBW / 2; ++pr) {
#pragma unroll
            for (int nbb = 0; nbb < 2; ++nbb)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub) {           // sub = 2 sy + sx = n & 3; the lane's two values of this sub-pixel are channels cc, cc + 1
                        const int orow = 2 * (r >> 4) + (sub >> 1), opx = 2 * pxx + (sub & 1);
                        float2 v2;
                        v2.x = acc[0][2 * pr + nbb][8 * u + sub]; v2.y = acc[0][2 * pr + nbb][8 * u + 4 + sub];
                        *(float2*)(stg + (orow * 32 + opx) * PS + 8 * nbb + 4 * u + 2 * h) = v2;
                    }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int piece = lane & 3;
            const int cc0 = 8 * (nb0 + 2 * pr) + 4 * piece;       // first of this lane's 4 output channels
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int op = (lane >> 2) + 16 * i;              // output pixel 0..127 of the wave: row op >> 5, column op & 31
                const int yo = yo0 + (op >> 5), xo = xo0 + (op & 31);
                const float* sp = stg + op * PS + 4 * piece;
                f32x4 v4; v4[0] = sp[0]; v4[1] = sp[1]; v4[2] = sp[2]; v4[3] = sp[3];
                if (yo < 2 * H && xo < 2 * Wd && cc0 + 4 <= N4) *(f32x4*)(out + (((long)t * (2 * H) + yo) * (2 * Wd) + xo) * ldout + cc0) = v4;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        return;
    }
    if constexpr (MODE == 1) if (gridDim.y == 1) {
        // PixelUnshuffle(2) epilogue through LDS.  A wave's 2 x 16 input pixels are exactly 8 output pixels with all their 4 N channels
        // (channel 4 n + 2 (y & 1) + (x & 1)), and those 8 pixels are consecutive in memory: 8 x 4N floats in one run.  From the
        // accumulators that is one 4-byte store per value at a 16-byte stride; staged, it is whole 16-byte pieces of consecutive lines.
        const int N4 = 4 * N, PSU = N4 + 4;                       // floats per staged output pixel (+4: the 8 pixels start in different banks)
        float* stg = (float*)lds_all + wave * 8 * PSU;
        const bool fits = (size_t)4 * 8 * PSU * 4 <= WL_BYTES + BT_BYTES;
        if (fits) {
#pragma unroll
            for (int g = 0; g < PG; ++g) {
                const int sub = 2 * (r >> 4) + (pxx & 1), opx = pxx >> 1;
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int n = 32 * nb + 16 * u + 8 * h + j;
                            if (n < N) stg[opx * PSU + 4 * n + sub] = acc[g][nb][8 * u + j];
                        }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int yo = (y0 + 8 * g) / 2 + wave, xo0 = x0 / 2;
                const int npiece = 8 * N;                         // float4 pieces of the wave's 8 output pixels
                for (int e = lane; e < npiece; e += 64) {
                    const int op = e / N, c4 = e - op * N;        // output pixel, piece within its 4N floats
                    const float* sp = stg + op * PSU + 4 * c4;
                    f32x4 v4; v4[0] = sp[0]; v4[1] = sp[1]; v4[2] = sp[2]; v4[3] = sp[3];
                    if (yo < (H >> 1) && xo0 + op < (Wd >> 1)) *(f32x4*)(out + (((long)t * (H >> 1) + yo) * (Wd >> 1) + xo0 + op) * N4 + 4 * c4) = v4;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            return;
        }
    }
#pragma unroll
    for (int g = 0; g < PG; ++g) {
    const int y = y0 + py + 8 * g;
    if (!(y < H && x < Wd)) continue;
    const int p = y * Wd + x;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = 32 * (nb0 + nb) + 16 * u + 8 * h + j;
                if (n >= N) continue;
                const float val = acc[g][nb][8 * u + j];
                if constexpr (MODE == 0) {
                    out[((long)t * P + p) * ldout + n] = val;
                } else if constexpr (MODE == 4) {
                    out[((long)t * P + p) * ldout + n] = val > 0.f ? val : slope * val;
                } else if constexpr (MODE == 1) {
                    const int oc = n * 4 + (y & 1) * 2 + (x & 1);
                    out[(((long)t * (H >> 1) + (y >> 1)) * (Wd >> 1) + (x >> 1)) * (4 * N) + oc] = val;
                } else if constexpr (MODE == 2) {
                    const int cc = n >> 2, sy = (n >> 1) & 1, sx = n & 1;
                    out[(((long)t * (2 * H) + 2 * y + sy) * (2 * Wd) + 2 * x + sx) * ldout + cc] = val;      // MODE 2: ldout = floats per OUTPUT pixel (>= N / 4)
                } else {
                    const long o = ((long)t * P + p) * ldout + n;
                    out[o] = val + skip[o];
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// OverlapPatchEmbed (restormer_arch.py:156-165): 3x3 conv from 1/3 input channels, fp32 VALU (exact
// fp32 like the reference; 0.1% of the FLOPs).  One thread = one pixel x 8 output channels.
__global__ __launch_bounds__(256) void k_embed(const float* __restrict__ in, const float* __restrict__ w,
                                               const float* __restrict__ bias, float* __restrict__ out, int T,
                                               int H, int W, int ic, int C, int relu) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [9*ic][C]: (tap, ci) major, channel minor
    for (int i = threadIdx.x; i < C * ic * 9; i += blockDim.x) {
        const int o = i / (ic * 9), rem = i % (ic * 9), ci = rem / 9, tap = rem % 9;
        wl[(tap * ic + ci) * C + o] = w[i];
    }
    __syncthreads();
    const int G = C >> 3;
    const long total = (long)T * H * W * G;
    // a workgroup stages the weights once and then walks `gridDim.x`-strided over 256-item slices (one slice per workgroup meant 5 KB of
    // weights fetched, written to LDS and a barrier for every 42 pixels)
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int g = (int)(idx % G);
    const long px = idx / G;
    const int x = (int)(px % W);
    const int y = (int)((px / W) % H);
    const int t = (int)(px / ((long)W * H));
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = bias ? bias[g * 8 + e] : 0.f;
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
        if (yy < 0 || yy >= H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = x + kx - 1;
            if (xx < 0 || xx >= W) continue;
            const float* src = in + ((long)(t * H + yy) * W + xx) * ic;
            for (int ci = 0; ci < ic; ++ci) {
                const float v = src[ci];
                const float* wp = wl + ((ky * 3 + kx) * ic + ci) * C + g * 8;
                const f32x4 w0 = *(const f32x4*)wp, w1 = *(const f32x4*)(wp + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[e] = fmaf(v, w0[e], acc[e]); acc[4 + e] = fmaf(v, w1[e], acc[4 + e]); }
            }
        }
    }
    if (relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaxf(acc[e], 0.f);
    }
    float* dst = out + px * C + g * 8;
    *(f32x4*)dst = f32x4{acc[0], acc[1], acc[2], acc[3]};
    *(f32x4*)(dst + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
    }
}

__global__ __launch_bounds__(256) void k_concat2(const float* __restrict__ a, int Ca, const float* __restrict__ b,
                                                 int Cb, float* __restrict__ out, long npx) {
    const int Cv = (Ca + Cb) >> 2;   // float4 per pixel
    const long total = npx * Cv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long px = idx / Cv;
        const int cv = (int)(idx % Cv) * 4;
        const f32x4 v = (cv < Ca) ? *(const f32x4*)(a + px * Ca + cv) : *(const f32x4*)(b + px * Cb + (cv - Ca));
        *(f32x4*)(out + px * (Ca + Cb) + cv) = v;
    }
}

// out[px][off .. off + Cb) = b[px][0 .. Cb): the skip half of torch.cat when the other half was written in place by its producer
__global__ __launch_bounds__(256) void k_copy_slice(const float* __restrict__ b, int Cb, float* __restrict__ out, int ldo, int off, long npx) {
    const int Cv = Cb >> 2;
    const long total = npx * Cv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long px = idx / Cv;
        const int cv = (int)(idx % Cv) * 4;
        *(f32x4*)(out + px * ldo + off + cv) = *(const f32x4*)(b + px * Cb + cv);
    }
}

// ---------------------------------------------------------------------------------------------
// Tiler (demo.py:120-153).
IRA_DEV int tile_origin(int i, int n, int stride, int full, int tile) { return (i == n - 1) ? (full - tile) : i * stride; }

// uint8 HWC frame -> float32 NHWC tiles:  x/255 (demo.py:120), reflect pad to (Hp,Wp) (demo.py:127),
// crop tile (demo.py:147).
__global__ __launch_bounds__(256) void k_extract_tiles(const uint8_t* __restrict__ frame, int h, int w, int c,
                                                       int Hp, int Wp, int th, int tw, int stride_h, int stride_w,
                                                       int n_rows, int n_cols, int first_tile, int n_tiles,
                                                       float* __restrict__ tiles, int flip) {
    const long total = (long)n_tiles * th * tw * c;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % c);
        long rest = idx / c;
        const int xx = (int)(rest % tw); rest /= tw;
        const int yy = (int)(rest % th);
        const int tl = (int)(rest / th) + first_tile;
        const int ri = tl / n_cols, ci = tl % n_cols;
        int Y = tile_origin(ri, n_rows, stride_h, Hp, th) + yy;
        int X = tile_origin(ci, n_cols, stride_w, Wp, tw) + xx;
        // bottom / right padding: F.pad(..., 'reflect') (demo.py:127), or with flip = 1 the flip-concatenation
        // torch.cat([img, flip(img)]) of SwinIR/inference.py:79-80 (row h + i is row h - 1 - i)
        if (Y >= h) Y = 2 * (h - 1) - Y + flip;
        if (X >= w) X = 2 * (w - 1) - X + flip;
        tiles[idx] = __fdiv_rn((float)frame[((long)Y * w + X) * c + ch], 255.0f);
    }
}

// ---------------------------------------------------------------------------------------------
// Pad-crop tiler of the X-Restormer / HAT style harness (X-Restormer/xrestormer/models/xrestormer_model.py:53-136):
// reflect-pad the frame on the bottom/right to a multiple of padding_size (pre_process :53-63), cut NON-overlapping
// tile_size cores, run each with a tile_pad halo clipped at the padded frame (:83-95), keep the core (:119-133), crop
// (:134-136).  Tiles are grouped by shape on the host; desc[n][6] = {y0, x0 of the haloed tile, core y0, x0, core h, w},
// all in padded-frame coordinates.
__global__ __launch_bounds__(256) void k_extract_list(const uint8_t* __restrict__ frame, int h, int w, int c, int th, int tw,
                                                      const int* __restrict__ desc, int n_tiles, float* __restrict__ tiles) {
    const long total = (long)n_tiles * th * tw * c;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % c);
        long rest = idx / c;
        const int xx = (int)(rest % tw); rest /= tw;
        const int yy = (int)(rest % th);
        const int tl = (int)(rest / th);
        int Y = desc[tl * 6 + 0] + yy, X = desc[tl * 6 + 1] + xx;
        if (Y >= h) Y = 2 * (h - 1) - Y;     // F.pad(..., 'reflect') :63
        if (X >= w) X = 2 * (w - 1) - X;
        tiles[idx] = __fdiv_rn((float)frame[((long)Y * w + X) * c + ch], 255.0f);
    }
}

// core of every tile -> uint8 frame: clamp(0,1), rint(x*255); pixels in the pre_process padding are dropped (:134-136).
// scale s > 1 (SR presets): tiles and the output frame live on the s-times finer grid, desc stays in input coordinates (:117-133).
__global__ __launch_bounds__(256) void k_scatter_cores(const float* __restrict__ tiles, int th, int tw, int c,
                                                       const int* __restrict__ desc, int n_tiles, int max_core, int h, int w,
                                                       uint8_t* __restrict__ out, int s) {
    const long total = (long)n_tiles * max_core * s * s * c;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % c);
        long rest = idx / c;
        const long q = rest % ((long)max_core * s * s);
        const int tl = (int)(rest / ((long)max_core * s * s));
        const int* d = desc + tl * 6;
        const int cw = d[5] * s, chh = d[4] * s;
        if (q >= (long)cw * chh) continue;
        const int dy = (int)(q / cw), dx = (int)(q % cw);
        const int Y = d[2] * s + dy, X = d[3] * s + dx;
        if (Y >= h * s || X >= w * s) continue;
        const float v = tiles[(((long)tl * th * s + ((d[2] - d[0]) * s + dy)) * (tw * s) + ((d[3] - d[1]) * s + dx)) * c + ch];
        out[((long)Y * (w * s) + X) * c + ch] = (uint8_t)__float2int_rn(fminf(fmaxf(v, 0.f), 1.f) * 255.0f);
    }
}

// E += tile, W += 1 in row-major tile order; E/W; clamp; crop; rint(x*255) (demo.py:145-161).
__global__ __launch_bounds__(256) void k_blend(const float* __restrict__ tiles, int h, int w, int c, int Hp, int Wp,
                                               int th, int tw, int stride_h, int stride_w, int n_rows, int n_cols,
                                               uint8_t* __restrict__ out) {
    const long total = (long)h * w * c;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % c);
        const int x = (int)((idx / c) % w);
        const int y = (int)(idx / ((long)c * w));
        float E = 0.f, Wt = 0.f;
        for (int ri = 0; ri < n_rows; ++ri) {
            const int y0 = tile_origin(ri, n_rows, stride_h, Hp, th);
            if (y < y0 || y >= y0 + th) continue;
            for (int ci = 0; ci < n_cols; ++ci) {
                const int x0 = tile_origin(ci, n_cols, stride_w, Wp, tw);
                if (x < x0 || x >= x0 + tw) continue;
                const long tl = (long)ri * n_cols + ci;
                E += tiles[((tl * th + (y - y0)) * tw + (x - x0)) * c + ch];
                Wt += 1.0f;
            }
        }
        float v = __fdiv_rn(E, Wt);
        v = fminf(fmaxf(v, 0.0f), 1.0f);
        out[idx] = (uint8_t)rintf(v * 255.0f);
    }
}

__global__ __launch_bounds__(256) void k_nchw_to_nhwc(const float* __restrict__ in, float* __restrict__ out, int B,
                                                      int C, int H, int W) {
    const long total = (long)B * C * H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % C);
        const long px = idx / C;
        const long hw = px % ((long)H * W);
        const long b = px / ((long)H * W);
        out[idx] = in[(b * C + ch) * (long)H * W + hw];
    }
}

__global__ __launch_bounds__(256) void k_nhwc_to_nchw(const float* __restrict__ in, float* __restrict__ out, int B,
                                                      int C, int H, int W) {
    const long total = (long)B * C * H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long hw = idx % ((long)H * W);
        const int ch = (int)((idx / ((long)H * W)) % C);
        const long b = idx / ((long)C * H * W);
        out[idx] = in[(b * (long)H * W + hw) * C + ch];
    }
}

// =============================================================================================
// launchers
static inline unsigned grid1d(long total, int block, long cap = 1L << 30) {
    long g = (total + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

template <int KS, int PG>
static hipError_t pw_kres_launch(hipStream_t st, const void* x, bool in_f32, int ln, const f16* W, const float* bias,
                                 f16* y, int ys, int ns, int NB, int T, int P, float eps, int ln_n, int act) {
    const long waves = (long)T * ((P + 32 * PG - 1) / (32 * PG));
    const dim3 grid(grid1d(waves, 4)), block(256);
#define IRA_KRES(LN_, F32_) hipLaunchKernelGGL((k_pw_kres<KS, LN_, F32_, PG>), grid, block, 0, st, x, W, bias, y, ys, ns, NB, T, P, eps, ln_n, act)
    if (in_f32) {
        if (ln == 0) IRA_KRES(0, true); else if (ln == 1) IRA_KRES(1, true); else IRA_KRES(2, true);
    } else {
        if (ln != 0) return hipErrorInvalidValue;
        IRA_KRES(0, false);
    }
#undef IRA_KRES
    return hipGetLastError();
}

template <int KS, int NW, int MINW>
static hipError_t pw_kres_lds_launch_nw(hipStream_t st, const void* x, bool in_f32, int ln, const f16* W, const float* bias,
                                        f16* y, int ys, int ns, int NB, int T, int P, float eps, int ln_n, int act) {
    const long waves = (long)T * ((P + 31) / 32);
    const long wgs = (waves + NW - 1) / NW;
    int splits = 1;
    if (wgs < 512 && NB >= 8) splits = (int)std::min<long>(NB / 4, (1024 + wgs - 1) / wgs);
    const dim3 grid((unsigned)wgs, splits), block(64 * NW);
    const size_t smem = (size_t)2 * 2 * KS * 1024;
#define IRA_KL(LN_, F32_)                                                                                               \
    do {                                                                                                                \
        static std::atomic<uint64_t> done_{0};                                                                          \
        if (hipError_t e_ = ensure_dynamic_lds((const void*)k_pw_kres_lds<KS, LN_, F32_, NW, MINW>, smem, done_); e_ != hipSuccess) return e_; \
        hipLaunchKernelGGL((k_pw_kres_lds<KS, LN_, F32_, NW, MINW>), grid, block, smem, st, x, W, bias, y, ys, ns, NB, T, P, eps, ln_n, act);      \
    } while (0)
    if (in_f32) {
        if (ln == 0) IRA_KL(0, true); else if (ln == 1) IRA_KL(1, true); else IRA_KL(2, true);
    } else {
        if (ln != 0) return hipErrorInvalidValue;
        IRA_KL(0, false);
    }
#undef IRA_KL
    return hipGetLastError();
}

template <int KS>
static hipError_t pw_kres_lds_launch(hipStream_t st, const void* x, bool in_f32, int ln, const f16* W, const float* bias,
                                     f16* y, int ys, int ns, int NB, int T, int P, float eps, int ln_n, int act) {
    static const int occ_env = env_int("IRA_KRES_OCC", 1);
    if constexpr (KS == 12) {   // 48 KB per workgroup: three workgroups per CU if the wave fits 170 registers
        if (occ_env) return pw_kres_lds_launch_nw<KS, 4, 3>(st, x, in_f32, ln, W, bias, y, ys, ns, NB, T, P, eps, ln_n, act);
    }
    if constexpr (KS == 24) {   // 96 KB per workgroup = one per CU: eight waves share it (two per SIMD instead of one)
        if (occ_env) return pw_kres_lds_launch_nw<KS, 8, 2>(st, x, in_f32, ln, W, bias, y, ys, ns, NB, T, P, eps, ln_n, act);
    }
    return pw_kres_lds_launch_nw<KS, 4, 1>(st, x, in_f32, ln, W, bias, y, ys, ns, NB, T, P, eps, ln_n, act);
}

template <int KS>
static hipError_t pw_kres_dispatch(hipStream_t st, const void* x, bool in_f32, int ln, const f16* W,
                                   const float* bias, f16* y, int ys, int ns, int NB, int T, int P, float eps, int ln_n, int act) {
    static const int lds_env = env_int("IRA_PW_LDS", 1);
    if constexpr (KS >= 16) return pw_kres_lds_launch<KS>(st, x, in_f32, ln, W, bias, y, ys, ns, NB, T, P, eps, ln_n, act);
    else {
    if constexpr (KS >= 12) { if (lds_env) return pw_kres_lds_launch<KS>(st, x, in_f32, ln, W, bias, y, ys, ns, NB, T, P, eps, ln_n, act); }
    static const int pg_env = env_int("IRA_KRES_PG", 0);
    int pg = pg_env > 0 ? pg_env : (KS <= 4 ? 4 : (KS <= 12 ? 2 : 1));
    while (pg > 1 && P <= 32 * (pg / 2)) pg /= 2;        // tiny tiles: do not pad a wave with empty groups
    if constexpr (KS <= 6) { if (pg >= 4) return pw_kres_launch<KS, 4>(st, x, in_f32, ln, W, bias, y, ys, ns, NB, T, P, eps, ln_n, act); }
    if constexpr (KS <= 12) { if (pg >= 2) return pw_kres_launch<KS, 2>(st, x, in_f32, ln, W, bias, y, ys, ns, NB, T, P, eps, ln_n, act); }
    return pw_kres_launch<KS, 1>(st, x, in_f32, ln, W, bias, y, ys, ns, NB, T, P, eps, ln_n, act);
    }
}

hipError_t launch_pw_kres(hipStream_t st, const void* x, bool in_f32, int ln_mode, int K, const void* W,
                          const float* bias, void* y, int y_stride, int n_store, int NB, int T, int P, float eps, int ln_n, int act, int planes) {
    if (ln_n <= 0 || ln_n > K) ln_n = K;          // 0 = "all K channels are real"
    if (K % 16 != 0 || n_store % 8 != 0 || y_stride % 8 != 0 || ln_mode < 0 || ln_mode > 2) return hipErrorInvalidValue;
    if (planes == 1 && pw_wide_supported(K, (long)T * P, 1)) return launch_pw_wide(st, x, in_f32, ln_mode, K, W, bias, y, y_stride, n_store, NB, T, P, eps, ln_n, act, 1);
    if (pw_wide_supported(K, (long)T * P)) return launch_pw_wide(st, x, in_f32, ln_mode, K, W, bias, y, y_stride, n_store, NB, T, P, eps, ln_n, act);
    const f16* Wh = (const f16*)W; f16* yh = (f16*)y;
    switch (K / 16) {
#define IRA_CASE(KS_) case KS_: return pw_kres_dispatch<KS_>(st, x, in_f32, ln_mode, Wh, bias, yh, y_stride, n_store, NB, T, P, eps, ln_n, act);
        IRA_CASE(1) IRA_CASE(2) IRA_CASE(3) IRA_CASE(4) IRA_CASE(6) IRA_CASE(8) IRA_CASE(12) IRA_CASE(16) IRA_CASE(24) IRA_CASE(32)
#undef IRA_CASE
        default: return hipErrorInvalidValue;
    }
}

template <int NBW, int PG>
static hipError_t pw_nres_lds_launch_w(hipStream_t st, const void* x0, int K0, const void* x1, int K1, bool in_f32,
                                       const f16* W, long wts, const float* bias, const float* resid, float* y, int N,
                                       int T, int P, int NBtot) {
    const int chunks = (P + 31) / 32;
    const size_t smem = (size_t)2 * 2 * NBW * 1024;
    const dim3 grid((unsigned)((long)T * ((chunks + 4 * PG - 1) / (4 * PG))), (unsigned)((NBtot + NBW - 1) / NBW)), block(256);
    if (in_f32) hipLaunchKernelGGL((k_pw_nres_lds<NBW, true, PG>), grid, block, smem, st, x0, K0, x1, K1, W, wts, bias, resid, y, N, T, P, NBtot);
    else hipLaunchKernelGGL((k_pw_nres_lds<NBW, false, PG>), grid, block, smem, st, x0, K0, x1, K1, W, wts, bias, resid, y, N, T, P, NBtot);
    return hipGetLastError();
}

template <int NB>
static hipError_t pw_nres_lds_launch(hipStream_t st, const void* x0, int K0, const void* x1, int K1, bool in_f32,
                                     const f16* W, long wts, const float* bias, const float* resid, float* y, int N,
                                     int T, int P) {
    static const int nbw_env = env_int("IRA_NRES_NBW", 0);     // 0 = rule below; otherwise the cap on blocks per workgroup
    // measured per 4K / SwinIR frame: NB = 8 as 2 x 4 blocks 62.7 -> 34-37 ms (fc2); NB = 6 as 2 x 3 and NB = 12 as 3 x 4 lose (pout 10.2 -> 11.6)
    const int cap = nbw_env > 0 ? nbw_env : (NB == 8 ? 4 : 16);
    if constexpr (NB > 4) {
        if (cap <= 2) return pw_nres_lds_launch_w<2, 1>(st, x0, K0, x1, K1, in_f32, W, wts, bias, resid, y, N, T, P, NB);
        if (cap == 3 || (cap == 4 && NB % 4 != 0 && NB % 3 == 0))
            return pw_nres_lds_launch_w<3, 1>(st, x0, K0, x1, K1, in_f32, W, wts, bias, resid, y, N, T, P, NB);
        if (cap == 4) return pw_nres_lds_launch_w<4, 1>(st, x0, K0, x1, K1, in_f32, W, wts, bias, resid, y, N, T, P, NB);
    }
    return pw_nres_lds_launch_w<NB, 1>(st, x0, K0, x1, K1, in_f32, W, wts, bias, resid, y, N, T, P, NB);
}

template <int NB>
static hipError_t pw_nres_dispatch(hipStream_t st, const void* x0, int K0, const void* x1, int K1, bool in_f32,
                                   const f16* W, long wts, const float* bias, const float* resid, float* y, int N,
                                   int T, int P) {
    // weights of every shape through LDS (the per-wave streaming kernel k_pw_nres of round 1 is retired: k_pw_nres_lds superseded it on every
    // shape, and single-input calls go to k_pw_gemm / k_nres_wide before they get here)
    return pw_nres_lds_launch<NB>(st, x0, K0, x1, K1, in_f32, W, wts, bias, resid, y, N, T, P);
}

hipError_t launch_pw_nres(hipStream_t st, const void* x0, int K0, const void* x1, int K1, bool in_f32, const void* W,
                          long w_tile_stride, const float* bias, const float* resid, float* y, int N, int NB, int T,
                          int P, int planes) {
    if (K0 % 16 != 0 || K1 % 16 != 0 || N % 8 != 0 || NB * 32 < N) return hipErrorInvalidValue;
    if ((x1 == nullptr || K1 == 0) && !in_f32 && nres_wide_supported(K0, N, NB, T, P, w_tile_stride))
        return launch_nres_wide(st, x0, K0, W, w_tile_stride, bias, resid, y, N, NB, T, P, planes == 1 ? 1 : 2);
    if ((x1 == nullptr || K1 == 0) && pw_gemm_supported(K0, N, NB))      // one input tensor: both operands through LDS (ira_nafnet.hip)
        return launch_pw_gemm(st, x0, in_f32, K0, W, w_tile_stride, bias, resid, y, N, NB, T, P);
    const f16* Wh = (const f16*)W;
    switch (NB) {
#define IRA_CASE(NB_) case NB_: return pw_nres_dispatch<NB_>(st, x0, K0, x1, K1, in_f32, Wh, w_tile_stride, bias, resid, y, N, T, P);
        IRA_CASE(1) IRA_CASE(2) IRA_CASE(3) IRA_CASE(4) IRA_CASE(6) IRA_CASE(8) IRA_CASE(12)
#undef IRA_CASE
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_dwconv_qkv(hipStream_t st, const void* in, const float* w, const float* bias, void* qk, void* v,
                             int T, int H, int W, int C) {
    if (C % 8 != 0) return hipErrorInvalidValue;
    static const int lines_env = env_int("IRA_DWQK_LINES", 1);
    const int g_lo = lines_env ? 2 * C / 8 : 0;
    if (lines_env) {
        const long items = (long)T * H * ((W + 7) / 8) * (2 * C / 8);
        hipLaunchKernelGGL(k_dwconv_qk_lines, dim3(grid1d(items, 256)), dim3(256), 0, st, (const f16*)in, w, bias, (f16*)qk, T, H, W, C);
    }
    const long total = (long)T * ((H + 1) / 2) * (2 * ((W + 7) / 8)) * (3 * C / 8 - g_lo);      // two output rows per thread
    hipLaunchKernelGGL(k_dwconv_qkv, dim3(grid1d(total, 256)), dim3(256), 0, st, (const f16*)in, w, bias, (f16*)qk,
                       (f16*)v, T, H, W, C, g_lo);
    return hipGetLastError();
}

hipError_t launch_dwconv_gate(hipStream_t st, const void* in, const float* w, const float* bias, void* out, int T,
                              int H, int W, int Hp) {
    if (Hp % 8 != 0) return hipErrorInvalidValue;
    const long total = (long)T * ((H + 3) / 4) * ((((W + DW_PX - 1) / DW_PX) * (Hp / 8) + 63) / 64) * 256;      // threads: workgroups x 256
    static const int tab_env = env_int("IRA_GATE_TAB", 1), row_env = env_int("IRA_GATE_ROW", 1);
    if (row_env && Hp >= 64) {
        const float2* tab = nullptr;
        if (hipError_t e = phi_table(&tab); e != hipSuccess) return e;
        const int gchunks = (Hp / 8 + 63) / 64, bands = (H + 3) / 4, XB = (W + 3) / 4;
        const long base = (long)T * bands * gchunks;
        int nseg = (int)std::min<long>(std::max<long>(1, (1024 + base - 1) / base), std::max(1, XB / 4));    // >= 4 steps per segment: the weight staging is per workgroup
        hipLaunchKernelGGL(k_dwconv_gate_row<true>, dim3((unsigned)(base * nseg)), dim3(256), 0, st, (const f16*)in, w, bias, (f16*)out, T, H, W, Hp, nseg, tab);
        return hipGetLastError();
    }
    if (tab_env) {
        const float2* tab = nullptr;
        if (hipError_t e = phi_table(&tab); e != hipSuccess) return e;
        hipLaunchKernelGGL(k_dwconv_gate<true>, dim3(grid1d(total, 256)), dim3(256), 0, st, (const f16*)in, w, bias, (f16*)out, T, H, W, Hp, tab);
    } else {
        hipLaunchKernelGGL(k_dwconv_gate<false>, dim3(grid1d(total, 256)), dim3(256), 0, st, (const f16*)in, w, bias, (f16*)out, T, H, W, Hp,
                           (const float2*)nullptr);
    }
    return hipGetLastError();
}

hipError_t launch_gram(hipStream_t st, const void* qk, float* part, int T, int nblk, int C, int heads, int S) {
    if (heads <= 0 || C % heads != 0 || (C / heads) % 16 != 0 || S <= 0) return hipErrorInvalidValue;
    const int CT = C / heads / 16;
    const long waves1 = (long)T * heads * S;
    int bands = 1;                                   // largest divisor of CT that keeps the launch under ~4096 waves
    for (int d = 2; d <= CT; ++d)
        if (CT % d == 0 && waves1 * d <= 4096) bands = d;
    const long waves = waves1 * bands;
    const dim3 grid(grid1d(waves, 4)), block(256);
    const f16* q = (const f16*)qk;
#define IRA_GRAM(CT_, PER_) hipLaunchKernelGGL((k_gram<CT_, PER_>), grid, block, 0, st, q, part, T, nblk, C, heads, S, bands)
    const int per = CT / bands;
    switch (CT * 8 + per) {
        case 1 * 8 + 1: IRA_GRAM(1, 1); break;
        case 2 * 8 + 1: IRA_GRAM(2, 1); break;
        case 2 * 8 + 2: IRA_GRAM(2, 2); break;
        case 3 * 8 + 1: IRA_GRAM(3, 1); break;
        case 3 * 8 + 3: IRA_GRAM(3, 3); break;
        case 6 * 8 + 1: IRA_GRAM(6, 1); break;
        case 6 * 8 + 2: IRA_GRAM(6, 2); break;
        case 6 * 8 + 3: IRA_GRAM(6, 3); break;
        case 6 * 8 + 6: IRA_GRAM(6, 6); break;
        default: return hipErrorInvalidValue;
    }
#undef IRA_GRAM
    return hipGetLastError();
}

// out[i] = sum over the S records of part[s][i], fixed order (the strip-parallel whole-frame path ships ONE Gram record per rank)
__global__ __launch_bounds__(256) void k_sum_records(const float* __restrict__ part, int S, int reclen, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= reclen) return;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 4 <= S; s += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] += part[(long)(s + u) * reclen + i];
    }
    for (; s < S; ++s) a[0] += part[(long)s * reclen + i];
    out[i] = (a[0] + a[1]) + (a[2] + a[3]);
}

hipError_t launch_sum_records(hipStream_t st, const float* part, int S, int reclen, float* out) {
    if (S <= 0 || reclen <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_sum_records, dim3((reclen + 255) / 256), dim3(256), 0, st, part, S, reclen, out);
    return hipGetLastError();
}

// out[t][g][i] = sum over the records s in [g G, min(S, (g + 1) G)) of part[t][s][i], in the fixed order of k_attn_softmax's sum_parts (eight
// chains).  A frame-wide Gram has up to 2048 partial records per image; k_attn_softmax runs heads x ceil(c / 8) workgroups per image, each
// adding all of them up with dependent loads - 0.2 ms per block of an untiled 4K frame (9.6 ms of its 194).  Here the first 32-fold of that
// sum runs on reclen / 256 x S / 32 workgroups.
__global__ __launch_bounds__(256) void k_reduce_records(const float* __restrict__ part, int S, int G, int reclen, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y, t = blockIdx.z, Sp = gridDim.y;
    if (i >= reclen) return;
    const int s0 = g * G, n = min(G, S - s0);
    const float* pb = part + ((long)t * S + s0) * reclen + i;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 8 <= n; s += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += pb[(long)(s + u) * reclen];
    }
    for (; s < n; ++s) a[0] += pb[(long)s * reclen];
    out[((long)t * Sp + g) * reclen + i] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

int attn_mix_reduce_groups(int S) { return S > 256 ? (S + 31) / 32 : 0; }

hipError_t launch_attn_mix(hipStream_t st, const float* part, const float* wo, const float* temp, float* attn_scratch,
                           void* M, int T, int C, int heads, int S, int NB, bool m_lo_zero, float* red) {
    const int c = C / heads;
    if (c > 96) return hipErrorInvalidValue;
    const int Sp = attn_mix_reduce_groups(S);
    if (red != nullptr && Sp > 0) {       // frame-wide Grams only (S depends on the tile geometry alone, never on the batch): two-level sum
        const int reclen = heads * (c * c + 2 * c);
        hipLaunchKernelGGL(k_reduce_records, dim3((reclen + 255) / 256, Sp, T), dim3(256), 0, st, part, S, 32, reclen, red);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
        part = red;
        S = Sp;
    }
    hipLaunchKernelGGL(k_attn_softmax, dim3(T * heads, (c + 7) / 8), dim3(256), 0, st, part, temp, attn_scratch, T, C, heads, S);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_attn_fold, dim3(T * heads, (C + 15) / 16), dim3(256), (size_t)c * c * sizeof(float), st, attn_scratch, wo,
                       (f16*)M, T, C, heads, NB, m_lo_zero ? 1 : 0);
    return hipGetLastError();
}

template <int MODE>
static hipError_t conv3x3_dispatch(hipStream_t st, const float* in, const f16* W, const float* bias,
                                   const float* skip, float* out, int T, int H, int Wd, int K, int N, int NB, float slope, int ldin, int ldout) {
    const long waves = (long)T * (((long)H * Wd + 31) / 32);
    // row blocks per workgroup: the fewest passes over the pixel operand with at most 6 blocks in registers, split evenly
    // (NB = 4 ran as 3 + 1 before: the second pass re-read and re-converted every pixel for a third of the MFMA work)
    const int passes = (NB + 5) / 6;
    int NBW = (NB + passes - 1) / passes;
    if (NBW == 5) NBW = 6;
    const dim3 grid(grid1d(waves, 4), (NB + NBW - 1) / NBW), block(256);
    static const int lds_env = env_int("IRA_CONV_LDS", 1);
    static const int tile_env = env_int("IRA_CONV_TILE", 1);
    // weights AND pixels through LDS (8 x 16 pixel tiles) where a k-step carries enough MFMAs to pay for the tile staging and its
    // barriers: measured faster for >= 3 row blocks and K >= 96 (MPRNet CABs, Restormer up-convolutions), slower for the 64-channel
    // SwinIR upsampler and the 2-block convolutions
    static const int tile_min_nbw = env_int("IRA_CONV_TILE_MINNBW", 3), tile_min_kt = env_int("IRA_CONV_TILE_MINKT", 6);
    static const int tile_small = env_int("IRA_CONV_TILE_SMALL", 1);     // 1-2 row blocks: 2 pixel groups per wave
    static const int small_min_kt = env_int("IRA_CONV_TILE_SMALL_MINKT", 3);
    const bool big = NBW >= tile_min_nbw && (K >> 4) >= tile_min_kt, small = tile_small && NBW <= 2 && (K >> 4) >= small_min_kt;
    if (tile_env && lds_env && (big || small)) {
        static const int pg34 = env_int("IRA_CONV_TILE_PG34", 1);
        const int pg = NBW <= 2 ? 2 : ((NBW == 3 || NBW == 4) ? pg34 : 1);
        const long tiles = (long)T * ((H + 8 * pg - 1) / (8 * pg)) * ((Wd + 15) / 16);
        const dim3 tgrid((unsigned)tiles, (NB + NBW - 1) / NBW);
        switch (NBW) {
            case 1: hipLaunchKernelGGL((k_conv3x3_tile<1, 2, MODE>), tgrid, block, 0, st, in, W, bias, skip, out, T, H, Wd, K, N, NB, slope, ldin, ldout); break;
            case 2: hipLaunchKernelGGL((k_conv3x3_tile<2, 2, MODE>), tgrid, block, 0, st, in, W, bias, skip, out, T, H, Wd, K, N, NB, slope, ldin, ldout); break;
            case 3:
                if (pg == 2) hipLaunchKernelGGL((k_conv3x3_tile<3, 2, MODE>), tgrid, block, 0, st, in, W, bias, skip, out, T, H, Wd, K, N, NB, slope, ldin, ldout);
                else hipLaunchKernelGGL((k_conv3x3_tile<3, 1, MODE>), tgrid, block, 0, st, in, W, bias, skip, out, T, H, Wd, K, N, NB, slope, ldin, ldout);
                break;
            case 4:
                if (pg == 2) hipLaunchKernelGGL((k_conv3x3_tile<4, 2, MODE>), tgrid, block, 0, st, in, W, bias, skip, out, T, H, Wd, K, N, NB, slope, ldin, ldout);
                else hipLaunchKernelGGL((k_conv3x3_tile<4, 1, MODE>), tgrid, block, 0, st, in, W, bias, skip, out, T, H, Wd, K, N, NB, slope, ldin, ldout);
                break;
            default: hipLaunchKernelGGL((k_conv3x3_tile<6, 1, MODE>), tgrid, block, 0, st, in, W, bias, skip, out, T, H, Wd, K, N, NB, slope, ldin, ldout); break;
        }
        return hipGetLastError();
    }
    if (lds_env && NB >= 2) {      // every multi-block conv: weights through LDS
        switch (NBW) {
            case 2: hipLaunchKernelGGL((k_conv3x3_lds<2, MODE>), grid, block, 0, st, in, W, bias, skip, out, T, H, Wd, K, N, NB, slope, ldin, ldout); break;
            case 3: hipLaunchKernelGGL((k_conv3x3_lds<3, MODE>), grid, block, 0, st, in, W, bias, skip, out, T, H, Wd, K, N, NB, slope, ldin, ldout); break;
            case 4: hipLaunchKernelGGL((k_conv3x3_lds<4, MODE>), grid, block, 0, st, in, W, bias, skip, out, T, H, Wd, K, N, NB, slope, ldin, ldout); break;
            default: hipLaunchKernelGGL((k_conv3x3_lds<6, MODE>), grid, block, 0, st, in, W, bias, skip, out, T, H, Wd, K, N, NB, slope, ldin, ldout); break;
        }
        return hipGetLastError();
    }
    switch (NBW) {
        case 1: hipLaunchKernelGGL((k_conv3x3<1, MODE>), grid, block, 0, st, in, W, bias, skip, out, T, H, Wd, K, N, NB, slope, ldin, ldout); break;
        case 2: hipLaunchKernelGGL((k_conv3x3<2, MODE>), grid, block, 0, st, in, W, bias, skip, out, T, H, Wd, K, N, NB, slope, ldin, ldout); break;
        case 3: hipLaunchKernelGGL((k_conv3x3<3, MODE>), grid, block, 0, st, in, W, bias, skip, out, T, H, Wd, K, N, NB, slope, ldin, ldout); break;
        default: hipLaunchKernelGGL((k_conv3x3<6, MODE>), grid, block, 0, st, in, W, bias, skip, out, T, H, Wd, K, N, NB, slope, ldin, ldout); break;
    }
    return hipGetLastError();
}

hipError_t launch_conv3x3(hipStream_t st, const float* in, const void* W, const float* bias, const float* skip,
                          float* out, int mode, int T, int H, int Wd, int K, int N, int NB, float slope, int ldin, int ldout) {
    if (K % 16 != 0 || NB * 32 < N) return hipErrorInvalidValue;
    // ldin / ldout: floats per pixel of the input / of the output and skip tensors (0 = K / N): a convolution can read the first K
    // channels of a wider pixel and write N channels into a slice of one (dense blocks, RRDBNet); float4 loads need ldin % 4 == 0
    if (ldin == 0) ldin = K;
    // mode 2 (PixelShuffle): ldout counts floats per OUTPUT pixel (default N / 4), so that the shuffled result can land in the first
    // N / 4 channels of a wider pixel (the level-1 concat buffer of the decoder: no separate copy of the up-sampled half)
    if (ldout == 0) ldout = (mode == 2) ? N / 4 : N;
    if (ldin < K || (ldin & 3) || (mode == 2 ? (ldout < N / 4 || (ldout & 3)) : ldout < N) || (mode == 1 && ldout != N)) return hipErrorInvalidValue;
    if (mode == 1 && ((H | Wd) & 1)) return hipErrorInvalidValue;
    if (mode == 2 && (N & 3)) return hipErrorInvalidValue;
    if (mode == 3 && skip == nullptr) return hipErrorInvalidValue;
    const f16* Wh = (const f16*)W;
    switch (mode) {
        case 0: return conv3x3_dispatch<0>(st, in, Wh, bias, skip, out, T, H, Wd, K, N, NB, slope, ldin, ldout);
        case 1: return conv3x3_dispatch<1>(st, in, Wh, bias, skip, out, T, H, Wd, K, N, NB, slope, ldin, ldout);
        case 2: return conv3x3_dispatch<2>(st, in, Wh, bias, skip, out, T, H, Wd, K, N, NB, slope, ldin, ldout);
        case 3: return conv3x3_dispatch<3>(st, in, Wh, bias, skip, out, T, H, Wd, K, N, NB, slope, ldin, ldout);
        case 4: return conv3x3_dispatch<4>(st, in, Wh, bias, skip, out, T, H, Wd, K, N, NB, slope, ldin, ldout);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_embed(hipStream_t st, const float* in, const float* w, const float* bias, float* out, int T, int H,
                        int W, int ic, int C, int relu) {
    if (C % 8 != 0) return hipErrorInvalidValue;
    const long total = (long)T * H * W * (C / 8);
    hipLaunchKernelGGL(k_embed, dim3(grid1d(total, 256, 256L * 16)), dim3(256), (size_t)C * ic * 9 * sizeof(float), st, in, w, bias, out, T, H, W, ic, C, relu);
    return hipGetLastError();
}

hipError_t launch_concat2(hipStream_t st, const float* a, int Ca, const float* b, int Cb, float* out, long npx) {
    if ((Ca | Cb) & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_concat2, dim3(grid1d(npx * ((Ca + Cb) / 4), 256, 65536)), dim3(256), 0, st, a, Ca, b, Cb, out, npx);
    return hipGetLastError();
}

hipError_t launch_copy_slice(hipStream_t st, const float* b, int Cb, float* out, int ldo, int off, long npx) {
    if ((Cb | ldo | off) & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_copy_slice, dim3(grid1d(npx * (Cb / 4), 256, 65536)), dim3(256), 0, st, b, Cb, out, ldo, off, npx);
    return hipGetLastError();
}

hipError_t launch_extract_tiles(hipStream_t st, const uint8_t* frame, int h, int w, int c, int Hp, int Wp, int th,
                                int tw, int stride_h, int stride_w, int n_rows, int n_cols, int first_tile,
                                int n_tiles, float* tiles, int flip) {
    const long total = (long)n_tiles * th * tw * c;
    hipLaunchKernelGGL(k_extract_tiles, dim3(grid1d(total, 256, 65536)), dim3(256), 0, st, frame, h, w, c, Hp, Wp, th, tw,
                       stride_h, stride_w, n_rows, n_cols, first_tile, n_tiles, tiles, flip);
    return hipGetLastError();
}

hipError_t launch_extract_list(hipStream_t st, const uint8_t* frame, int h, int w, int c, int th, int tw, const int* desc,
                               int n_tiles, float* tiles) {
    hipLaunchKernelGGL(k_extract_list, dim3(grid1d((long)n_tiles * th * tw * c, 256, 65536)), dim3(256), 0, st, frame, h, w, c, th, tw,
                       desc, n_tiles, tiles);
    return hipGetLastError();
}

hipError_t launch_scatter_cores(hipStream_t st, const float* tiles, int th, int tw, int c, const int* desc, int n_tiles,
                                int max_core, int h, int w, uint8_t* out, int scale) {
    hipLaunchKernelGGL(k_scatter_cores, dim3(grid1d((long)n_tiles * max_core * scale * scale * c, 256, 65536)), dim3(256), 0, st, tiles, th, tw, c,
                       desc, n_tiles, max_core, h, w, out, scale);
    return hipGetLastError();
}

hipError_t launch_blend(hipStream_t st, const float* tiles, int h, int w, int c, int Hp, int Wp, int th, int tw,
                        int stride_h, int stride_w, int n_rows, int n_cols, uint8_t* out) {
    const long total = (long)h * w * c;
    hipLaunchKernelGGL(k_blend, dim3(grid1d(total, 256, 65536)), dim3(256), 0, st, tiles, h, w, c, Hp, Wp, th, tw, stride_h,
                       stride_w, n_rows, n_cols, out);
    return hipGetLastError();
}

hipError_t launch_nchw_to_nhwc(hipStream_t st, const float* in, float* out, int B, int C, int H, int W) {
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3(grid1d((long)B * C * H * W, 256, 65536)), dim3(256), 0, st, in, out, B, C, H, W);
    return hipGetLastError();
}

hipError_t launch_nhwc_to_nchw(hipStream_t st, const float* in, float* out, int B, int C, int H, int W) {
    hipLaunchKernelGGL(k_nhwc_to_nchw, dim3(grid1d((long)B * C * H * W, 256, 65536)), dim3(256), 0, st, in, out, B, C, H, W);
    return hipGetLastError();
}

}  // namespace ira
