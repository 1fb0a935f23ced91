// gfx950 kernels specific to NAFNet (reference X-Restormer/xrestormer/archs/NAFNet_arch.py:158-326).
// Everything GEMM-shaped reuses the conventions of ira_kernels.hip (weights = MFMA A operand, fragment-major hi/lo
// fp16 planes; 32 pixels = B operand; fp32 accumulators; fp32 NHWC residual stream, fp16 branch activations).
//
// One NAFBlock (:198-219) is scheduled as
//   t1 = conv1(LN(x))                       k_pw_kres / k_pw_kres_lds  (LN gain/bias folded into W1, eps 1e-6)
//   g  = SimpleGate(dw3x3(t1)), sum_px g    k_naf_dw_sg                 (per-workgroup partial channel sums)
//   s  = sca(mean g)                        k_naf_sca                   (AdaptiveAvgPool2d(1) + 1x1, per image)
//   x += (diag(beta) W3)(g * s) + beta*b3   k_naf_pw<.., IN_SCALED, OUT_RES>  (channel attention applied while loading B)
//   t1 = conv4(LN(x))                       k_pw_kres
//   x += (diag(gamma) W5)(t1a * t1b) + ..   k_naf_pw<.., IN_GATED, OUT_RES>   (SimpleGate applied while loading B)
// and the U-Net plumbing: downs (2x2 stride-2 conv, :275-277) = k_naf_pw<.., IN_S2D, OUT_PLAIN> (K = 4c gathered from
// the four source pixels), ups (1x1 c->2c + PixelShuffle(2), :286-291) plus the skip add (:320) =
// k_naf_pw<.., IN_F32, OUT_SHUFFLE_ADD> (weight rows permuted by the packer so that 8 consecutive rows are 8
// consecutive channels of ONE sub-pixel).
#include "ira_kernels.h"
#include "ira_device.h"

#include <cstdlib>

namespace ira {

namespace {


enum { IN_F16 = 0, IN_GATED = 1, IN_S2D = 2, IN_F32 = 3, IN_SCALED = 4, IN_F32X2 = 5, IN_S2DX2 = 6 };
enum { OUT_RES = 0, OUT_SHUFFLE_ADD = 1 };

}  // namespace

// ---------------------------------------------------------------------------------------------
// N-resident pointwise contraction with LDS-shared weight slabs (the skeleton of k_pw_nres_lds) and NAFNet's
// operand forms.  MFMA pixel grid: T images of Hg x Wg pixels; one wave = 32 pixels, one workgroup = 128 pixels of
// one image (per-image weights for the channel-attention fold).  blockIdx.y selects NBW of the NBtot 32-row blocks.
//   IN_F16    x: fp16 [T][P][K]
//   IN_GATED  x: fp16 [T][P][2K], B = x[.., k] * x[.., K + k]                       (SimpleGate :158-161)
//   IN_S2D    x: fp32 [T][2Hg][2Wg][K/4], k = (dy*2+dx)*(K/4) + ci                  (2x2 stride-2 conv)
//   IN_F32    x: fp32 [T][P][K]
//   IN_SCALED x: fp16 [T][P][K], B = x[.., k] * scale[t][k]   (the channel attention x * sca(x), :205, without building a
//             per-image matrix)
//   IN_F32X2 / IN_S2DX2: as IN_F32 / IN_S2D with the fp32 pixel operand split into fp16 hi + lo (third MFMA per block): the
//             1x1 convolutions of MPRNet's fp32 feature streams
//   OUT_RES          y fp32 [T][P][N] = acc (+ resid)
//   OUT_SHUFFLE_ADD  y fp32 [T][2Hg][2Wg][N/4]: row n = q*(N/4) + cc -> sub-pixel q = 2sy+sx, channel cc; + resid there
template <int NBW, int INMODE, int OUTMODE>
__global__ __launch_bounds__(256) void k_naf_pw(const void* __restrict__ xin, int K, const f16* __restrict__ W,
                                                long w_tile_stride, const float* __restrict__ bias, const float* resid,
                                                float* y, int N, int NBtot, int T, int Hg, int Wg,
                                                const float* __restrict__ scale) {
    constexpr int F = 2 * NBW;
    constexpr int SH = (F + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half8* wl = (half8*)smem;                 // [2][F][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int P = Hg * Wg;
    const int chunks = (P + 31) >> 5;
    const int wg_per_tile = (chunks + 3) >> 2;
    const int t = blockIdx.x / wg_per_tile;
    const int ch = (blockIdx.x % wg_per_tile) * 4 + wave;
    const bool live = ch < chunks;
    int p = (live ? ch : chunks - 1) * 32 + r;
    const bool valid = live && p < P;
    if (p >= P) p = P - 1;
    const long px = (long)t * P + p;
    const int KT = K >> 4;
    const int nb0 = blockIdx.y * NBW;
    const int Y = p / Wg, X = p - Y * Wg;

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
    const f16* wt = W + (long)t * w_tile_stride + lane * 8;
    const long lo_off = (long)NBtot * KT * 512;
    auto gsrc = [&](int s, int f) -> const half8* {     // f < NBW: hi plane block nb0+f, else lo plane block nb0+f-NBW
        return (const half8*)(wt + (f < NBW ? 0 : lo_off) + ((long)(nb0 + (f < NBW ? f : f - NBW)) * KT + s) * 512);
    };
    half8 stg[SH];
#pragma unroll
    for (int i = 0; i < SH; ++i) if (wave + 4 * i < F) stg[i] = *gsrc(0, wave + 4 * i);
#pragma unroll
    for (int i = 0; i < SH; ++i) if (wave + 4 * i < F) wl[(wave + 4 * i) * 64 + lane] = stg[i];
    __syncthreads();

    const int cs = K >> 2;                    // IN_S2D: source channels
    const int spt = (KT >> 2) > 0 ? (KT >> 2) : 1;   // IN_S2D: k-steps per tap
    for (int s = 0; s < KT; ++s) {
        const half8* cur = wl + (s & 1) * F * 64;
        if (s + 1 < KT) {
#pragma unroll
            for (int i = 0; i < SH; ++i) if (wave + 4 * i < F) stg[i] = *gsrc(s + 1, wave + 4 * i);
        }
        const int c0 = 16 * s + 8 * h;
        half8 b;
        [[maybe_unused]] half8 bl;
        if constexpr (INMODE == IN_F16) {
            b = *(const half8*)((const f16*)xin + px * K + c0);
        } else if constexpr (INMODE == IN_SCALED) {
            const half8 a = *(const half8*)((const f16*)xin + px * K + c0);
            const float* sp = scale + (long)t * K + c0;
            const f32x4 s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { b[j] = (f16)((float)a[j] * s0[j]); b[4 + j] = (f16)((float)a[4 + j] * s1[j]); }
        } else if constexpr (INMODE == IN_GATED) {
            const f16* src = (const f16*)xin + px * 2 * K + c0;
            const half8 a1 = *(const half8*)src, a2 = *(const half8*)(src + K);
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = (f16)((float)a1[j] * (float)a2[j]);
        } else {
            const float* src;
            if constexpr (INMODE == IN_S2D || INMODE == IN_S2DX2) {
                const int tap = s / spt, ci0 = (s - tap * spt) * 16 + 8 * h;
                src = (const float*)xin + (((long)t * 2 * Hg + 2 * Y + (tap >> 1)) * (2 * Wg) + 2 * X + (tap & 1)) * cs + ci0;
            } else {
                src = (const float*)xin + px * K + c0;
            }
            const f32x4 lo4 = *(const f32x4*)src, hi4 = *(const f32x4*)(src + 4);
            b = cvt8(lo4, hi4);
            if constexpr (INMODE == IN_F32X2 || INMODE == IN_S2DX2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { bl[j] = (f16)(lo4[j] - (float)b[j]); bl[4 + j] = (f16)(hi4[j] - (float)b[4 + j]); }
            }
        }
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[nb * 64 + lane], b, acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[(NBW + nb) * 64 + lane], b, acc[nb], 0, 0, 0);
            if constexpr (INMODE == IN_F32X2 || INMODE == IN_S2DX2)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[nb * 64 + lane], bl, acc[nb], 0, 0, 0);
        }
        if (s + 1 < KT) {
            half8* nxt = wl + ((s + 1) & 1) * F * 64;
#pragma unroll
            for (int i = 0; i < SH; ++i) if (wave + 4 * i < F) nxt[(wave + 4 * i) * 64 + lane] = stg[i];
        }
        __syncthreads();
    }
    if (!valid) return;
    const int Nq = N >> 2;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int n0 = 32 * (nb0 + nb) + 16 * u + 8 * h;
            if (n0 + 8 <= N) {
                f32x4 lo, hi;
#pragma unroll
                for (int j = 0; j < 4; ++j) { lo[j] = acc[nb][8 * u + j]; hi[j] = acc[nb][8 * u + 4 + j]; }
                long dst;
                if constexpr (OUTMODE == OUT_RES) {
                    dst = px * N + n0;
                } else {
                    const int q = n0 / Nq, cc = n0 - q * Nq;
                    dst = (((long)t * 2 * Hg + 2 * Y + (q >> 1)) * (2 * Wg) + 2 * X + (q & 1)) * Nq + cc;
                }
                if (resid != nullptr) {
                    lo += *(const f32x4*)(resid + dst);
                    hi += *(const f32x4*)(resid + dst + 4);
                }
                *(f32x4*)(y + dst) = lo;
                *(f32x4*)(y + dst + 4) = hi;
            }
        }
}

// ---------------------------------------------------------------------------------------------
// The same contraction with BOTH operands staged through LDS in stages of 32 input channels (two k-steps), two stages ahead of
// their use in registers.  k_naf_pw feeds its MFMAs with pixel fragments loaded from global memory by the wave that consumes
// them; in that loop the compiler waits vmcnt(0) at the first use of every step, which also retires the weight loads just
// issued for the next step: every step costs a memory round trip.  Here every MFMA operand comes from LDS (counted lgkmcnt
// waits, reads interleaved with MFMAs), the only vmcnt waits are in front of the ds_writes and leave the newest stage's loads in
// flight, one barrier covers 4 * NBW MFMAs per wave, and the pixel rows are read as 64 / 128 contiguous bytes.
// IN_F32 = false: x fp16 [T][P][K]; true: x fp32 [T][P][K] rounded to fp16 (k_naf_pw's IN_F32).  y fp32 [T][P][N] = acc (+ resid).
// Same operands and MFMA order per accumulator as k_naf_pw: bitwise the same output.
template <int NBW, int XMODE>   // XMODE: IN_F16, IN_F32, IN_GATED or IN_SCALED of k_naf_pw
__global__ __launch_bounds__(256) void k_pw_gemm(const void* __restrict__ xin, int K, const f16* __restrict__ W, long w_tile_stride,
                                                 const float* __restrict__ bias, const float* resid, float* y, int N, int NBtot,
                                                 int T, int P, const float* __restrict__ scale) {
    constexpr int F = 2 * NBW;                    // fragments per k-step: NBW hi + NBW lo
    constexpr int FS = 2 * F;                     // per stage (2 k-steps)
    constexpr int SHA = (FS + 3) / 4;             // weight fragments staged per wave
    __shared__ __attribute__((aligned(16))) half8 wl[2][FS][64];
    __shared__ __attribute__((aligned(16))) half8 bl[2][2][2][128];     // [buffer][k-step][k half][pixel]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int chunks = (P + 127) >> 7;
    const int t = blockIdx.x / chunks;
    const int p0 = (blockIdx.x % chunks) * 128;
    const int nb0 = blockIdx.y * NBW;
    const int pl = 32 * wave + r;                 // this lane's pixel within the tile
    const bool valid = p0 + pl < P;
    const long px = (long)t * P + (valid ? p0 + pl : P - 1);

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
    __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): the bias loads retire here, not at the loop's first MFMA
    const int KT = K >> 4, NS = KT >> 1;          // K % 32 == 0
    const f16* wt = W + (long)t * w_tile_stride + lane * 8;
    const long lo_off = (long)NBtot * KT * 512;
    // weight fragment f of stage g: k-step 2g + f / F, within it f % F < NBW -> hi plane block nb0 + f % F, else lo plane
    auto wsrc = [&](int g, int f) -> const half8* {
        const int s = 2 * g + f / F, ff = f % F;
        int nb = nb0 + (ff < NBW ? ff : ff - NBW);
        if (nb >= NBtot) nb = NBtot - 1;
        return (const half8*)(wt + (ff < NBW ? 0 : lo_off) + ((long)nb * KT + s) * 512);
    };
    // pixel element e = tid + 256 i (i < 2): pixel e / 4, 8-channel part e % 4 of the stage's 32 channels
    auto bsrc = [&](int g, int i) -> half8 {
        const int e = tid + 256 * i, q = e >> 2, part = e & 3;
        const int pp = p0 + q < P ? p0 + q : P - 1;
        const int c0 = 32 * g + 8 * part;
        const long off = ((long)t * P + pp) * K + c0;
        if constexpr (XMODE == IN_F32) {
            const float* src = (const float*)xin + off;
            return cvt8(*(const f32x4*)src, *(const f32x4*)(src + 4));
        } else if constexpr (XMODE == IN_GATED) {
            const f16* src = (const f16*)xin + ((long)t * P + pp) * 2 * K + c0;
            const half8 a1 = *(const half8*)src, a2 = *(const half8*)(src + K);
            half8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f16)((float)a1[j] * (float)a2[j]);
            return o;
        } else if constexpr (XMODE == IN_SCALED) {
            const half8 a = *(const half8*)((const f16*)xin + off);
            const float* sp = scale + (long)t * K + c0;
            const f32x4 s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
            half8 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) { o[j] = (f16)((float)a[j] * s0[j]); o[4 + j] = (f16)((float)a[4 + j] * s1[j]); }
            return o;
        } else {
            return *(const half8*)((const f16*)xin + off);
        }
    };
    half8 wa[SHA], wb[SHA], ba[2], bb[2];         // stage + 1 (a) and stage + 2 (b) in registers
    auto wload = [&](int g, half8 (&dst)[SHA]) {
#pragma unroll
        for (int i = 0; i < SHA; ++i) if ((FS % 4 == 0) || wave + 4 * i < FS) dst[i] = *wsrc(g, wave + 4 * i);
    };
    auto store = [&](int buf, const half8 (&wsrc_)[SHA], const half8 (&bsrc_)[2]) {
#pragma unroll
        for (int i = 0; i < SHA; ++i) if ((FS % 4 == 0) || wave + 4 * i < FS) wl[buf][wave + 4 * i][lane] = wsrc_[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + 256 * i, q = e >> 2, part = e & 3;
            bl[buf][part >> 1][part & 1][q] = bsrc_[i];
        }
    };
    wload(0, wb);
    bb[0] = bsrc(0, 0); bb[1] = bsrc(0, 1);
    wload(NS > 1 ? 1 : 0, wa);
    ba[0] = bsrc(NS > 1 ? 1 : 0, 0); ba[1] = bsrc(NS > 1 ? 1 : 0, 1);
    store(0, wb, bb);
    __syncthreads();
    for (int g = 0; g < NS; ++g) {
        const int cur = g & 1;
        if (g + 2 < NS) {
            wload(g + 2, wb);
            bb[0] = bsrc(g + 2, 0); bb[1] = bsrc(g + 2, 1);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const half8 b = bl[cur][ks][h][pl];
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                const half8 a = wl[cur][ks * F + nb][lane], al = wl[cur][ks * F + NBW + nb][lane];
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b, acc[nb], 0, 0, 0);
            }
        }
        if (g + 1 < NS) store(cur ^ 1, wa, ba);
#pragma unroll
        for (int i = 0; i < SHA; ++i) wa[i] = wb[i];
        ba[0] = bb[0]; ba[1] = bb[1];
        __syncthreads();
    }
    if (!valid) return;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int n0 = 32 * (nb0 + nb) + 16 * u + 8 * h;
            if (n0 + 8 <= N) {
                f32x4 lo, hi;
#pragma unroll
                for (int j = 0; j < 4; ++j) { lo[j] = acc[nb][8 * u + j]; hi[j] = acc[nb][8 * u + 4 + j]; }
                const long dst = px * N + n0;
                if (resid != nullptr) {
                    lo += *(const f32x4*)(resid + dst);
                    hi += *(const f32x4*)(resid + dst + 4);
                }
                *(f32x4*)(y + dst) = lo;
                *(f32x4*)(y + dst + 4) = hi;
            }
        }
}

// ---------------------------------------------------------------------------------------------
// conv2 (depthwise 3x3 on 2c channels, zero padding, bias) + SimpleGate + per-workgroup partial sums of the gated
// tensor for the channel attention's global average pool (:202-205).  One thread = 4 pixels x 8 gated channels.
//   in fp16 [T][H][W][2c];  w fp32 [9][2c];  out fp16 [T][H][W][c];  part fp32 [T][bpt][c]
// The grid is per image (bpt workgroups each) and the partials are reduced in a fixed order by k_naf_sca, so an
// image's result does not depend on its batch.
constexpr int NDW_PX = 4;

template <int PX>
IRA_DEV void naf_dw_block(const f16* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias, int H,
                          int W, int Cs, int t, int y, int x0, int ch0, float (&acc)[PX][8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float bv = bias ? bias[ch0 + e] : 0.f;
#pragma unroll
        for (int p = 0; p < PX; ++p) acc[p][e] = bv;
    }
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
}

__global__ __launch_bounds__(256) void k_naf_dw_sg(const f16* __restrict__ in, const float* __restrict__ w,
                                                   const float* __restrict__ bias, f16* __restrict__ out,
                                                   float* __restrict__ part, int T, int H, int W, int c, int bpt) {
    __shared__ float sm[256][9];              // +1: conflict-free column reads
    const int XB = (W + NDW_PX - 1) / NDW_PX, G = c >> 3;
    const long bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int t = (int)(bid / bpt), blk = (int)(bid % bpt);
    const int tid = threadIdx.x;
    const long idx = (long)blk * 256 + tid;
    const long per_tile = (long)H * XB * G;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    if (idx < per_tile) {
        const int g = (int)(idx % G);
        const long rest = idx / G;
        const int xq = (int)(rest % XB), y = (int)(rest / XB);
        const int x0 = xq * NDW_PX;
        float a1[NDW_PX][8], a2[NDW_PX][8];
        naf_dw_block<NDW_PX>(in, w, bias, H, W, 2 * c, t, y, x0, g * 8, a1);
        naf_dw_block<NDW_PX>(in, w, bias, H, W, 2 * c, t, y, x0, c + g * 8, a2);
        f16* dst = out + ((long)(t * H + y) * W + x0) * c + g * 8;
#pragma unroll
        for (int p = 0; p < NDW_PX; ++p) {
            if (x0 + p < W) {
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float gv = a1[p][e] * a2[p][e]; s[e] += gv; o[e] = (f16)gv; }
                *(half8*)(dst + (long)p * c) = o;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) sm[tid][e] = s[e];
    __syncthreads();
    // G divides 256 (c is a power of two <= 2048): threads tid, tid+G, ... hold the same channel group
    for (int j = tid; j < c; j += 256) {
        const int gi = j >> 3, e = j & 7;
        float a = 0.f;
        for (int k = gi; k < 256; k += G) a += sm[k][e];
        part[((long)t * bpt + blk) * c + j] = a;
    }
}

// s[t][o] = sca_b[o] + sum_k sca_w[o][k] * mean_px g[t][.., k]   (:175-179, :205).  grid (T, ceil(c/64)), 1024 threads.
// The bpt partial records are summed by 1024/c slices of threads (4 independent accumulators each), then combined in a
// fixed order: deterministic, and short dependency chains (the first version walked all records serially per channel:
// 80 us per launch at c = 512).
__global__ __launch_bounds__(1024) void k_naf_sca(const float* __restrict__ part, int bpt, float inv_px,
                                                  const float* __restrict__ wsca, const float* __restrict__ bsca,
                                                  float* __restrict__ s, int c) {
    extern __shared__ __attribute__((aligned(16))) float lds[];     // mean[c] + slices[1024 or c]
    float* mean = lds;
    float* sl = lds + c;
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pb = part + (long)t * bpt * c;
    const int nsl = c >= 1024 ? 1 : 1024 / c;                       // c is a power of two
    for (int j0 = 0; j0 < c; j0 += 1024) {
        const int j = j0 + tid % (c < 1024 ? c : 1024), sli = c < 1024 ? tid / c : 0;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int b = sli;
        for (; b + 3 * nsl < bpt; b += 4 * nsl) {
            a0 += pb[(long)b * c + j]; a1 += pb[(long)(b + nsl) * c + j];
            a2 += pb[(long)(b + 2 * nsl) * c + j]; a3 += pb[(long)(b + 3 * nsl) * c + j];
        }
        for (; b < bpt; b += nsl) a0 += pb[(long)b * c + j];
        sl[tid] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (tid < c - j0 && tid < 1024) {
            float m = 0.f;
            for (int k = 0; k < nsl; ++k) m += sl[k * (c < 1024 ? c : 1024) + tid];
            mean[j0 + tid] = m * inv_px;
        }
        __syncthreads();
    }
    for (int i = 0; i < 4; ++i) {
        const int o = blockIdx.y * 64 + wave * 4 + i;
        if (o >= c) break;
        float a = 0.f;
        for (int k = lane; k < c; k += 64) a = fmaf(wsca[(long)o * c + k], mean[k], a);
        a = wave_sum(a);
        if (lane == 0) s[(long)t * c + o] = a + bsca[o];
    }
}

// ---------------------------------------------------------------------------------------------
// Fused NAFBlock tail for C = 32 / 64 (everything after the pooled statistics is per pixel, :205-219):
//     y = x + (diag(beta) W3)(g * s[t]) + beta*b3
//     x_out = y + (diag(gamma) W5)( SimpleGate( W4' LN(y) + b4' ) ) + gamma*b5             written in place over x
// The three unfused kernels moved 3c + 3c + 4c elements per pixel plus the 2c-channel fp16 intermediate; this one reads
// g (fp16) and x and writes x: 320 B/px at C = 32 instead of 960.  All three weight matrices (hi/lo planes, fragment
// major) live in LDS for the life of the workgroup (16 KB at C = 32, 64 KB at C = 64) and every wave walks ITER groups
// of 32 pixels.  The MFMA accumulators never leave registers: D-layout registers are, by construction of rowperm(),
// the B fragments of the next contraction (see the header of ira_kernels.hip), LayerNorm statistics are a per-lane sum
// + one cross-half shuffle, and SimpleGate pairs block nb with block NB+nb in the same lane and register.
constexpr int TAIL_ITER = 8;

template <int C>
__global__ __launch_bounds__(256) void k_naf_tail(const f16* __restrict__ g, const float* __restrict__ scale, float* x,
                                                  const f16* __restrict__ W3, const float* __restrict__ b3,
                                                  const f16* __restrict__ W4, const float* __restrict__ b4,
                                                  const f16* __restrict__ W5, const float* __restrict__ b5, int T, int P) {
    constexpr int NB = C / 32, KS = C / 16;
    constexpr int F3 = 2 * NB * KS, F4 = 2 * 2 * NB * KS;           // fragments (1 KB) of W3 / W5 and of W4
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half8* w3 = (half8*)smem;                 // [2][NB][KS][64]
    half8* w4 = w3 + F3 * 64;                 // [2][2NB][KS][64]
    half8* w5 = w4 + F4 * 64;                 // [2][NB][KS][64]
    float* bl = (float*)(w5 + F3 * 64);       // b3[C] b4[2C] b5[C]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < F3 * 64; i += 256) { w3[i] = ((const half8*)W3)[i]; w5[i] = ((const half8*)W5)[i]; }
    for (int i = tid; i < F4 * 64; i += 256) w4[i] = ((const half8*)W4)[i];
    for (int i = tid; i < C; i += 256) { bl[i] = b3[i]; bl[3 * C + i] = b5[i]; }
    for (int i = tid; i < 2 * C; i += 256) bl[C + i] = b4[i];
    __syncthreads();

    const int chunks = (P + 31) >> 5;
    const int wg_per_tile = (chunks + 4 * TAIL_ITER - 1) / (4 * TAIL_ITER);
    const long bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int t = (int)(bid / wg_per_tile);
    const int ch0 = (int)(bid % wg_per_tile) * 4 * TAIL_ITER + wave * TAIL_ITER;
    const float* sc = scale + (long)t * C;
    f32x4 sv[KS][2];
#pragma unroll
    for (int s = 0; s < KS; ++s) { sv[s][0] = *(const f32x4*)(sc + 16 * s + 8 * h); sv[s][1] = *(const f32x4*)(sc + 16 * s + 8 * h + 4); }

    for (int it = 0; it < TAIL_ITER; ++it) {
        const int ch = ch0 + it;
        if (ch >= chunks) break;
        const int p = ch * 32 + r;
        const bool valid = p < P;
        const long px = (long)t * P + (valid ? p : P - 1);
        // B fragments of conv3: g * s
        half8 b[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const half8 a = *(const half8*)(g + px * C + 16 * s + 8 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) { b[s][j] = (f16)((float)a[j] * sv[s][0][j]); b[s][4 + j] = (f16)((float)a[4 + j] * sv[s][1][j]); }
        }
        // y = x + b3 + W3 (g*s)
        f32x16 y[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n0 = 32 * nb + 16 * u + 8 * h;
                const f32x4 xl = *(const f32x4*)(x + px * C + n0), xh = *(const f32x4*)(x + px * C + n0 + 4);
                const f32x4 bl0 = *(const f32x4*)(bl + n0), bl1 = *(const f32x4*)(bl + n0 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { y[nb][8 * u + j] = xl[j] + bl0[j]; y[nb][8 * u + 4 + j] = xh[j] + bl1[j]; }
            }
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                y[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3[((0 * NB + nb) * KS + s) * 64 + lane], b[s], y[nb], 0, 0, 0);
                y[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3[((1 * NB + nb) * KS + s) * 64 + lane], b[s], y[nb], 0, 0, 0);
            }
        // LayerNorm2d statistics over the C channels of the pixel (lane pair r, r+32)
        float sum = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int j = 0; j < 16; ++j) sum += y[nb][j];
        sum += __shfl_xor(sum, 32);
        const float mu = sum * (1.0f / C);
        float sq = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int j = 0; j < 16; ++j) { const float d = y[nb][j] - mu; sq += d * d; }
        sq += __shfl_xor(sq, 32);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / C) + 1e-6f);
        // register (nb, 8u + j) = channel 32nb + 16u + 8h + j = element j of the B fragment of k-step 2nb + u
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) b[2 * nb + u][j] = (f16)((y[nb][8 * u + j] - mu) * rstd);
        // out = y + b5 + W5 gate(W4 n + b4)
        f32x16 o[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n0 = 3 * C + 32 * nb + 16 * u + 8 * h;
                const f32x4 bl0 = *(const f32x4*)(bl + n0), bl1 = *(const f32x4*)(bl + n0 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { o[nb][8 * u + j] = y[nb][8 * u + j] + bl0[j]; o[nb][8 * u + 4 + j] = y[nb][8 * u + 4 + j] + bl1[j]; }
            }
#pragma unroll
        for (int pr = 0; pr < NB; ++pr) {
            f32x16 a1, a2;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n1 = C + 32 * pr + 16 * u + 8 * h, n2 = n1 + C;
                const f32x4 p0 = *(const f32x4*)(bl + n1), p1 = *(const f32x4*)(bl + n1 + 4);
                const f32x4 q0 = *(const f32x4*)(bl + n2), q1 = *(const f32x4*)(bl + n2 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { a1[8 * u + j] = p0[j]; a1[8 * u + 4 + j] = p1[j]; a2[8 * u + j] = q0[j]; a2[8 * u + 4 + j] = q1[j]; }
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w4[((0 * 2 * NB + pr) * KS + s) * 64 + lane], b[s], a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w4[((1 * 2 * NB + pr) * KS + s) * 64 + lane], b[s], a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w4[((0 * 2 * NB + NB + pr) * KS + s) * 64 + lane], b[s], a2, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w4[((1 * 2 * NB + NB + pr) * KS + s) * 64 + lane], b[s], a2, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                half8 gf;
#pragma unroll
                for (int j = 0; j < 8; ++j) gf[j] = (f16)(a1[8 * u + j] * a2[8 * u + j]);
                const int s5 = 2 * pr + u;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w5[((0 * NB + nb) * KS + s5) * 64 + lane], gf, o[nb], 0, 0, 0);
                    o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w5[((1 * NB + nb) * KS + s5) * 64 + lane], gf, o[nb], 0, 0, 0);
                }
            }
        }
        if (valid) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int n0 = 32 * nb + 16 * u + 8 * h;
                    f32x4 lo, hi;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { lo[j] = o[nb][8 * u + j]; hi[j] = o[nb][8 * u + 4 + j]; }
                    *(f32x4*)(x + px * C + n0) = lo;
                    *(f32x4*)(x + px * C + n0 + 4) = hi;
                }
        }
    }
}

// check_image_size (:356-361): zero pad to the right/bottom; and the final crop (:326).  fp32 NHWC, C channels.
__global__ __launch_bounds__(256) void k_naf_pad_crop(const float* __restrict__ in, float* __restrict__ out, int T,
                                                      int Hi, int Wi, int Ho, int Wo, int C) {
    const long total = (long)T * Ho * Wo * C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % C);
        long rest = idx / C;
        const int x = (int)(rest % Wo); rest /= Wo;
        const int y = (int)(rest % Ho);
        const int t = (int)(rest / Ho);
        out[idx] = (y < Hi && x < Wi) ? in[(((long)t * Hi + y) * Wi + x) * C + ch] : 0.f;
    }
}

// =============================================================================================
template <int NBW, int INMODE, int OUTMODE>
static hipError_t naf_pw_launch(hipStream_t st, const void* x, int K, const f16* W, long wts, const float* bias,
                                const float* resid, float* y, int N, int NBtot, int T, int Hg, int Wg, const float* scale) {
    const int chunks = (Hg * Wg + 31) / 32;
    const dim3 grid((unsigned)((long)T * ((chunks + 3) / 4)), (unsigned)(NBtot / NBW)), block(256);
    const size_t smem = (size_t)2 * 2 * NBW * 1024;
    hipLaunchKernelGGL((k_naf_pw<NBW, INMODE, OUTMODE>), grid, block, smem, st, x, K, W, wts, bias, resid, y, N, NBtot, T, Hg, Wg, scale);
    return hipGetLastError();
}

template <int INMODE, int OUTMODE>
static hipError_t naf_pw_nbw(hipStream_t st, const void* x, int K, const f16* W, long wts, const float* bias,
                             const float* resid, float* y, int N, int NBtot, int T, int Hg, int Wg, const float* scale) {
    // at most 4 blocks per workgroup: the 8-block register tile ran one wave per SIMD (measured on the same kernel with the
    // SwinIR-L fc2 shape: 46.7 ms per frame at 8 blocks, 34.2 at 2 x 4); the pixel operand is re-read per split, from L2
    if (NBtot % 4 == 0) return naf_pw_launch<4, INMODE, OUTMODE>(st, x, K, W, wts, bias, resid, y, N, NBtot, T, Hg, Wg, scale);
    if (NBtot % 3 == 0) return naf_pw_launch<3, INMODE, OUTMODE>(st, x, K, W, wts, bias, resid, y, N, NBtot, T, Hg, Wg, scale);   // 96 / 192 channels (MPRNet)
    if (NBtot % 2 == 0) return naf_pw_launch<2, INMODE, OUTMODE>(st, x, K, W, wts, bias, resid, y, N, NBtot, T, Hg, Wg, scale);
    return naf_pw_launch<1, INMODE, OUTMODE>(st, x, K, W, wts, bias, resid, y, N, NBtot, T, Hg, Wg, scale);
}

bool pw_gemm_supported(int K, int N, int NBtot) {
    static const int gemm_env = getenv("IRA_PW_GEMM") ? atoi(getenv("IRA_PW_GEMM")) : 1;
    return gemm_env && K > 0 && K % 32 == 0 && N % 8 == 0 && NBtot * 32 >= N;
}

// y fp32 [T][P][N] = W x (+ bias) (+ resid): x fp16 or fp32 [T][P][K]; W fragment-major hi + lo planes, w_tile_stride halfs per image
static hipError_t pw_gemm_mode(hipStream_t st, int xmode, const void* x, int K, const void* W, long w_tile_stride, const float* bias,
                               const float* resid, float* y, int N, int NBtot, int T, int P, const float* scale) {
    if (!pw_gemm_supported(K, N, NBtot) || T <= 0 || P <= 0) return hipErrorInvalidValue;
    const int passes = (NBtot + 3) / 4, nbw = (NBtot + passes - 1) / passes;       // <= 4 row blocks per workgroup, split evenly
    const dim3 grid((unsigned)((long)T * ((P + 127) / 128)), (unsigned)((NBtot + nbw - 1) / nbw)), block(256);
    const f16* Wh = (const f16*)W;
#define IRA_GEMM_M(NBW_, M_) hipLaunchKernelGGL((k_pw_gemm<NBW_, M_>), grid, block, 0, st, x, K, Wh, w_tile_stride, bias, resid, y, N, NBtot, T, P, scale)
#define IRA_GEMM(NBW_)                                           \
    do {                                                         \
        switch (xmode) {                                         \
            case IN_F32: IRA_GEMM_M(NBW_, IN_F32); break;        \
            case IN_GATED: IRA_GEMM_M(NBW_, IN_GATED); break;    \
            case IN_SCALED: IRA_GEMM_M(NBW_, IN_SCALED); break;  \
            default: IRA_GEMM_M(NBW_, IN_F16); break;            \
        }                                                        \
    } while (0)
    switch (nbw) {
        case 1: IRA_GEMM(1); break;
        case 2: IRA_GEMM(2); break;
        case 3: IRA_GEMM(3); break;
        default: IRA_GEMM(4); break;
    }
#undef IRA_GEMM
#undef IRA_GEMM_M
    return hipGetLastError();
}

hipError_t launch_pw_gemm(hipStream_t st, const void* x, bool x_f32, int K, const void* W, long w_tile_stride, const float* bias,
                          const float* resid, float* y, int N, int NBtot, int T, int P) {
    return pw_gemm_mode(st, x_f32 ? IN_F32 : IN_F16, x, K, W, w_tile_stride, bias, resid, y, N, NBtot, T, P, nullptr);
}

hipError_t launch_naf_pw(hipStream_t st, int inmode, int outmode, const void* x, int K, const void* W, long w_tile_stride,
                         const float* bias, const float* resid, float* y, int N, int NBtot, int T, int Hg, int Wg,
                         const float* scale) {
    if (inmode == IN_SCALED && scale == nullptr) return hipErrorInvalidValue;
    if (outmode == OUT_RES && (inmode == IN_F16 || inmode == IN_F32 || inmode == IN_GATED || inmode == IN_SCALED) && pw_gemm_supported(K, N, NBtot) &&
        T > 0 && Hg > 0 && Wg > 0)
        return pw_gemm_mode(st, inmode, x, K, W, w_tile_stride, bias, resid, y, N, NBtot, T, Hg * Wg, scale);
    if (K <= 0 || K % 16 != 0 || N % 8 != 0 || NBtot * 32 < N || T <= 0 || Hg <= 0 || Wg <= 0) return hipErrorInvalidValue;
    if ((inmode == IN_S2D || inmode == IN_S2DX2) && (K % 64 != 0)) return hipErrorInvalidValue;           // 4 taps x multiples of 16 channels
    if (outmode == OUT_SHUFFLE_ADD && (N % 32 != 0)) return hipErrorInvalidValue;  // N/4 channels, runs of 8
    const f16* Wh = (const f16*)W;
#define IRA_NAF(IM, OM) return naf_pw_nbw<IM, OM>(st, x, K, Wh, w_tile_stride, bias, resid, y, N, NBtot, T, Hg, Wg, scale)
    if (outmode == OUT_RES) {
        switch (inmode) {
            case IN_F16: IRA_NAF(IN_F16, OUT_RES);
            case IN_GATED: IRA_NAF(IN_GATED, OUT_RES);
            case IN_S2D: IRA_NAF(IN_S2D, OUT_RES);
            case IN_F32: IRA_NAF(IN_F32, OUT_RES);
            case IN_SCALED: IRA_NAF(IN_SCALED, OUT_RES);
            case IN_F32X2: IRA_NAF(IN_F32X2, OUT_RES);
            case IN_S2DX2: IRA_NAF(IN_S2DX2, OUT_RES);
            default: return hipErrorInvalidValue;
        }
    } else if (outmode == OUT_SHUFFLE_ADD && inmode == IN_F32) {
        IRA_NAF(IN_F32, OUT_SHUFFLE_ADD);
    }
#undef IRA_NAF
    return hipErrorInvalidValue;
}

int naf_dw_sg_blocks(int H, int W, int c) {
    const long per_tile = (long)H * ((W + NDW_PX - 1) / NDW_PX) * (c / 8);
    return (int)((per_tile + 255) / 256);
}

hipError_t launch_naf_dw_sg(hipStream_t st, const void* in, const float* w, const float* bias, void* out, float* part,
                            int T, int H, int W, int c) {
    if (c < 16 || (c & (c - 1)) != 0 || c > 2048) return hipErrorInvalidValue;
    const int bpt = naf_dw_sg_blocks(H, W, c);
    hipLaunchKernelGGL(k_naf_dw_sg, dim3((unsigned)((long)T * bpt)), dim3(256), 0, st, (const f16*)in, w, bias, (f16*)out, part,
                       T, H, W, c, bpt);
    return hipGetLastError();
}

bool naf_tail_supported(int c) { return c == 32 || c == 64; }

template <int C>
static hipError_t naf_tail_launch(hipStream_t st, const void* g, const float* scale, float* x, const void* W3, const float* b3,
                                  const void* W4, const float* b4, const void* W5, const float* b5, int T, int P) {
    constexpr int NB = C / 32, KS = C / 16;
    const size_t smem = (size_t)(2 * (2 * NB * KS) + 4 * NB * KS) * 1024 + (size_t)4 * C * sizeof(float);
    static std::atomic<uint64_t> done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)k_naf_tail<C>, smem, done); e != hipSuccess) return e;
    const int chunks = (P + 31) / 32;
    const int wg_per_tile = (chunks + 4 * TAIL_ITER - 1) / (4 * TAIL_ITER);
    hipLaunchKernelGGL((k_naf_tail<C>), dim3((unsigned)((long)T * wg_per_tile)), dim3(256), smem, st, (const f16*)g, scale, x,
                       (const f16*)W3, b3, (const f16*)W4, b4, (const f16*)W5, b5, T, P);
    return hipGetLastError();
}

hipError_t launch_naf_tail(hipStream_t st, const void* g, const float* scale, float* x, int c, const void* W3, const float* b3,
                           const void* W4, const float* b4, const void* W5, const float* b5, int T, int P) {
    if (c == 32) return naf_tail_launch<32>(st, g, scale, x, W3, b3, W4, b4, W5, b5, T, P);
    if (c == 64) return naf_tail_launch<64>(st, g, scale, x, W3, b3, W4, b4, W5, b5, T, P);
    return hipErrorInvalidValue;
}

hipError_t launch_naf_sca(hipStream_t st, const float* part, int bpt, long npx, const float* wsca, const float* bsca, float* s,
                          int T, int c) {
    if (c < 16 || (c & (c - 1)) != 0 || c > 2048) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_naf_sca, dim3(T, (c + 63) / 64), dim3(1024), (size_t)(c + 1024) * sizeof(float), st, part, bpt,
                       1.0f / (float)npx, wsca, bsca, s, c);
    return hipGetLastError();
}

hipError_t launch_naf_pad_crop(hipStream_t st, const float* in, float* out, int T, int Hi, int Wi, int Ho, int Wo, int C) {
    long g = ((long)T * Ho * Wo * C + 255) / 256;
    if (g > 65536) g = 65536;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_naf_pad_crop, dim3((unsigned)g), dim3(256), 0, st, in, out, T, Hi, Wi, Ho, Wo, C);
    return hipGetLastError();
}

}  // namespace ira
