// Fused GDFN block (restormer_arch.py:88-93 + the residual of :149), one kernel:
//     x_out = x + project_out( gelu(dw3x3(project_in(LN(x)))[:Hf]) * dw3x3(project_in(LN(x)))[Hf:] )
// The 2*Hf-channel hidden tensor (the largest tensor of the network: 510 channels at 512x512) and the gated
// Hf tensor never leave the CU: HBM traffic of the FFN drops from ~4.2 KB/px (three kernels) to
// ~0.9 KB/px (x in with a 1-pixel halo, x out), at the price of recomputing project_in on the halo.
//
// One workgroup (512 threads = 8 waves) owns an 8 x 32 output patch of one tile:
//   step 1   LN(x) of the 10 x 34 halo patch -> fp16 MFMA B-fragments in LDS (fragment-major, 1 KB / (group, k-step))
//   per hidden chunk j (32 channels of x1 + the matching 32 of x2):
//     2a     hp[halo px][64] = W_in[chunk] . LN(x)        MFMA, 11 pixel groups over 8 waves -> LDS (fp16, zero outside
//                                                         the image = the dwconv's zero padding)
//     2b     wave w = output row w: depthwise 3x3 on x1/x2 from LDS, exact-erf GELU gate, result straight into the
//            B-fragment registers of
//     2c     acc_out += W_out[:, chunk] . g               MFMA
//   step 3   x_out = acc_out (+bias) + x                   fp32 store
// x_out must not alias x (neighbouring patches read each other's halo).
// Numerics contract identical to the three-kernel path (ira_kernels.hip): fp16 LN output, fp16 hidden (hp) and gated
// (g) values, weights as fp16 hi+lo planes, fp32 accumulation / LN statistics / dwconv / GELU.
#include "ira_kernels.h"

namespace ira {

typedef _Float16 f16;
typedef f16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define IRA_DEV __device__ __forceinline__

namespace {

IRA_DEV long xcd_swz(long bid, long nblk) {
    const long q = nblk >> 3, rem = nblk & 7, x = bid & 7, j = bid >> 3;
    return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + j;
}

IRA_DEV float gelu_erf_f(float x) {   // same formulation as ira_kernels.hip
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    const float e = p * t * __expf(-z * z);
    return 0.5f * x * ((x >= 0.f) ? (2.0f - e) : e);
}

constexpr int TH = 8, TW = 32, PH = TH + 2, PW = TW + 2, PHW = PH * PW;   // 340 halo pixels
constexpr int NG1 = (PHW + 31) / 32;                                        // 11 pixel groups
constexpr int HPS = 72;                                                     // hp row stride in halfs (64 + 8 pad)

template <int C>
struct FfnLds {
    static constexpr int KS = C / 16;
    static constexpr size_t xn_bytes = (size_t)NG1 * KS * 64 * 16;
    static constexpr size_t hp_bytes = (size_t)NG1 * 32 * HPS * 2;
    static constexpr size_t dw_bytes = 9 * 64 * 4;
    static constexpr size_t total = xn_bytes + hp_bytes + dw_bytes;
};

}  // namespace

template <int C, int LN>
__global__ __launch_bounds__(512) void k_ffn_fused(const float* __restrict__ x, float* __restrict__ xo,
                                                   const f16* __restrict__ Win, const float* __restrict__ bin,
                                                   const float* __restrict__ dww, const float* __restrict__ dwb,
                                                   const f16* __restrict__ Wout, const float* __restrict__ bout,
                                                   int T, int H, int W, int Hp) {
    constexpr int KS = C / 16;            // k-steps of project_in
    constexpr int NBC = (C + 31) / 32;    // 32-row blocks of project_out
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half8* xn = (half8*)smem;                                              // [NG1][KS][64]
    f16* hp = (f16*)(smem + FfnLds<C>::xn_bytes);                          // [NG1*32][HPS]
    float* dwl = (float*)(smem + FfnLds<C>::xn_bytes + FfnLds<C>::hp_bytes);   // [9][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int npy = (H + TH - 1) / TH, npx = (W + TW - 1) / TW;
    const long pid = xcd_swz(blockIdx.x, gridDim.x);
    const int t = (int)(pid / ((long)npy * npx));
    const int py = (int)((pid / npx) % npy), pxx = (int)(pid % npx);
    const int y0 = py * TH, x0 = pxx * TW;
    const float* xt = x + (long)t * H * W * C;

    // ---- step 1: LN of the halo patch -> B fragments in LDS
    for (int g = wave; g < NG1; g += 8) {
        const int q = 32 * g + r;
        const int hy = q / PW, hx = q - hy * PW;
        const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
        const bool in = (q < PHW) && yy >= 0 && yy < H && xx >= 0 && xx < W;
        const float* src = xt + ((long)(in ? yy : 0) * W + (in ? xx : 0)) * C + 8 * h;
        float xv[KS][8];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f32x4 lo = *(const f32x4*)(src + 16 * s), hi = *(const f32x4*)(src + 16 * s + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { xv[s][j] = lo[j]; xv[s][4 + j] = hi[j]; }
        }
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += xv[s][j];
        sum += __shfl_xor(sum, 32);
        const float mu = sum * (1.0f / C);
        float sq = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = xv[s][j] - mu; sq += d * d; }
        sq += __shfl_xor(sq, 32);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / C) + 1e-5f);
        const float sub = (LN == 2) ? mu : 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            half8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f16)((xv[s][j] - sub) * rstd);
            xn[(g * KS + s) * 64 + lane] = o;
        }
    }

    // this wave's output row (step 2b/2c/3): pixels (y0 + wave, x0 + r)
    const int oy = y0 + wave, ox = x0 + r;
    const bool ovalid = oy < H && ox < W;
    f32x16 acc_out[NBC];
#pragma unroll
    for (int nb = 0; nb < NBC; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc_out[nb][j] = 0.f;

    const int NBH = Hp >> 5;                       // 32-channel chunks of x1 (= of x2)
    const int KTO = Hp >> 4;                       // k-steps of project_out
    const long win_lo = (long)(2 * NBH) * KS * 512;
    const long wout_lo = (long)NBC * KTO * 512;
    const f16* winl = Win + lane * 8;
    const f16* woutl = Wout + lane * 8;

    for (int j = 0; j < NBH; ++j) {
        __syncthreads();     // xn complete (first pass) / previous chunk's readers of hp and dwl done
        // depthwise weights of this chunk: [9][64] (x1 channels 0..31, x2 channels 32..63)
        for (int i = tid; i < 9 * 64; i += 512) {
            const int tap = i >> 6, cc = i & 63;
            const int ch = (cc < 32) ? (32 * j + cc) : (Hp + 32 * j + cc - 32);
            dwl[i] = dww[(long)tap * 2 * Hp + ch];
        }
        // ---- 2a: hp chunk = W_in[chunk] . LN(x) on every halo pixel group
        for (int g = wave; g < NG1; g += 8) {
            f32x16 a1, a2;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int cl = 16 * u + 8 * h + e;
                    a1[8 * u + e] = bin ? bin[32 * j + cl] : 0.f;
                    a2[8 * u + e] = bin ? bin[Hp + 32 * j + cl] : 0.f;
                }
            const f16* w1 = winl + (long)j * KS * 512;
            const f16* w2 = winl + (long)(NBH + j) * KS * 512;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const half8 b = xn[(g * KS + s) * 64 + lane];
                const half8 a1h = *(const half8*)(w1 + s * 512), a1l = *(const half8*)(w1 + win_lo + s * 512);
                const half8 a2h = *(const half8*)(w2 + s * 512), a2l = *(const half8*)(w2 + win_lo + s * 512);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b, a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, b, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2h, b, a2, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2l, b, a2, 0, 0, 0);
            }
            const int q = 32 * g + r;
            const int hy = q / PW, hx = q - hy * PW;
            const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
            const bool in = (q < PHW) && yy >= 0 && yy < H && xx >= 0 && xx < W;
            f16* dst = hp + (long)q * HPS + 8 * h;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                half8 o1, o2;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    o1[e] = in ? (f16)a1[8 * u + e] : (f16)0.f;     // zero padding of the depthwise conv
                    o2[e] = in ? (f16)a2[8 * u + e] : (f16)0.f;
                }
                *(half8*)(dst + 16 * u) = o1;
                *(half8*)(dst + 32 + 16 * u) = o2;
            }
        }
        __syncthreads();
        // ---- 2b + 2c: depthwise 3x3 + gate on this wave's output row, then project_out accumulation
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int cl0 = 16 * s + 8 * h;
            float d1[8], d2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                d1[e] = dwb ? dwb[32 * j + cl0 + e] : 0.f;
                d2[e] = dwb ? dwb[Hp + 32 * j + cl0 + e] : 0.f;
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int q = (wave + tap / 3) * PW + r + tap % 3;       // halo index of (oy + dy, ox + dx)
                const half8 v1 = *(const half8*)(hp + (long)q * HPS + cl0);
                const half8 v2 = *(const half8*)(hp + (long)q * HPS + 32 + cl0);
                const f32x4 wa = *(const f32x4*)(dwl + tap * 64 + cl0), wb = *(const f32x4*)(dwl + tap * 64 + cl0 + 4);
                const f32x4 wc = *(const f32x4*)(dwl + tap * 64 + 32 + cl0), wd = *(const f32x4*)(dwl + tap * 64 + 32 + cl0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    d1[e] = fmaf((float)v1[e], wa[e], d1[e]);
                    d1[4 + e] = fmaf((float)v1[4 + e], wb[e], d1[4 + e]);
                    d2[e] = fmaf((float)v2[e], wc[e], d2[e]);
                    d2[4 + e] = fmaf((float)v2[4 + e], wd[e], d2[4 + e]);
                }
            }
            half8 bg;
#pragma unroll
            for (int e = 0; e < 8; ++e) bg[e] = (f16)(gelu_erf_f(d1[e]) * d2[e]);
            const int ks = 2 * j + s;
#pragma unroll
            for (int nb = 0; nb < NBC; ++nb) {
                const half8 ah = *(const half8*)(woutl + ((long)nb * KTO + ks) * 512);
                const half8 al = *(const half8*)(woutl + wout_lo + ((long)nb * KTO + ks) * 512);
                acc_out[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bg, acc_out[nb], 0, 0, 0);
                acc_out[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bg, acc_out[nb], 0, 0, 0);
            }
        }
    }

    // ---- step 3: residual + store
    if (!ovalid) return;
    const long po = ((long)t * H + oy) * W + ox;
#pragma unroll
    for (int nb = 0; nb < NBC; ++nb)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int n0 = 32 * nb + 16 * u + 8 * h;
            if (n0 + 8 <= C) {
                f32x4 lo = *(const f32x4*)(x + po * C + n0), hi = *(const f32x4*)(x + po * C + n0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] += acc_out[nb][8 * u + e] + (bout ? bout[n0 + e] : 0.f);
                    hi[e] += acc_out[nb][8 * u + 4 + e] + (bout ? bout[n0 + 4 + e] : 0.f);
                }
                *(f32x4*)(xo + po * C + n0) = lo;
                *(f32x4*)(xo + po * C + n0 + 4) = hi;
            }
        }
}

bool ffn_fused_supported(int C, int Hp) { return (C == 48 || C == 96) && Hp % 32 == 0; }

template <int C>
static hipError_t ffn_launch(hipStream_t st, const float* x, float* xo, int ln_mode, const f16* Win, const float* bin,
                             const float* dww, const float* dwb, const f16* Wout, const float* bout, int T, int H, int W,
                             int Hp) {
    const long nblk = (long)T * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    const size_t smem = FfnLds<C>::total;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ffn_fused<C, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)k_ffn_fused<C, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (ln_mode == 1)
        hipLaunchKernelGGL((k_ffn_fused<C, 1>), dim3((unsigned)nblk), dim3(512), smem, st, x, xo, Win, bin, dww, dwb, Wout, bout, T, H, W, Hp);
    else
        hipLaunchKernelGGL((k_ffn_fused<C, 2>), dim3((unsigned)nblk), dim3(512), smem, st, x, xo, Win, bin, dww, dwb, Wout, bout, T, H, W, Hp);
    return hipGetLastError();
}

hipError_t launch_ffn_fused(hipStream_t st, const float* x, float* xo, int ln_mode, int C, const void* Win, const float* bin,
                            const float* dww, const float* dwb, const void* Wout, const float* bout, int T, int H, int W,
                            int Hp) {
    if (x == xo || (ln_mode != 1 && ln_mode != 2) || !ffn_fused_supported(C, Hp)) return hipErrorInvalidValue;
    if (C == 48) return ffn_launch<48>(st, x, xo, ln_mode, (const f16*)Win, bin, dww, dwb, (const f16*)Wout, bout, T, H, W, Hp);
    return ffn_launch<96>(st, x, xo, ln_mode, (const f16*)Win, bin, dww, dwb, (const f16*)Wout, bout, T, H, W, Hp);
}

}  // namespace ira
