// LayerNorm + 1x1 conv for the deep levels (K = 192 / 384 input channels: restormer_arch.py:105,114 after :146 and :82,89), second version of
// k_pw_kres_lds (ira_kernels.hip).  What that kernel measured at the headline shape (40 tiles, level 3: 655 360 px, K = 192, N = 1024;
// tools/time_pw.py): 1.01 ms = 20 % of the MFMA peak; 0.47 ms still with the stores, the LayerNorm and the lo-plane MFMAs all removed.  The
// reason is the weight stream: a workgroup of 4 waves x 32 px fetches the whole [N][K] hi+lo matrix (786 KB) from L2 for its 128 pixels =
// 4 GB of L2 reads per launch, and every MFMA reads its own 1 KB A fragment from LDS.  Here
//   * a wave keeps PG = 2 pixel groups (64 px) of LayerNorm-ed B fragments in registers and a workgroup has NW = 8 waves: 512 px per pass
//     over the weights (4x less L2 traffic), and every A fragment read from LDS feeds PG MFMAs;
//   * an iteration covers NBI = 2 row blocks (64 output channels): 4 independent accumulator chains per wave, half the barriers, and the
//     fp16 results of a group leave through a per-wave LDS transposition as whole 128-byte lines (8 lanes x 16 B per pixel) instead of
//     32-byte pieces of 32 different lines per store instruction (that alone cost 22 % of the old kernel);
//   * weights (and the bias of the iteration's row blocks) are staged one iteration ahead by LDS-DMA into an LDS double buffer: no staging
//     registers, and no vector-memory result is consumed inside the loop (the old kernel's accumulator init from `bias` made the compiler wait
//     vmcnt(0) - i.e. for the just-issued weight prefetch - in front of the first MFMA of every iteration).  A phase stagger of the first
//     workgroup on each CU (tried: 0 / 8k / 16k / 32k cycles per class) changes nothing: the kernel is not HBM-burst-bound in its prologue.
//     Nor is the prologue bound by its fragment-shaped loads (32 cache lines per instruction): a variant that copied each group's pixel rows
//     with fully contiguous loads through wave-private LDS rows and picked the fragments out of LDS measured the same (0.73-0.75 vs 0.72 ms).
// Numerics are those of k_pw_kres_lds: same LayerNorm arithmetic, same fragment order of the hi and lo MFMAs per accumulator.
#include "ira_kernels.h"
#include "ira_device.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>

namespace ira {

namespace {

template <int KS, int LN, bool IN_F32, bool HAS_BIAS, int NW, int PG, int NBI, int MINW, bool LO>
__global__ __launch_bounds__(64 * NW, MINW) void k_pw_wide(const void* __restrict__ xin, const f16* __restrict__ W, const float* __restrict__ bias,
                                                           f16* __restrict__ y, int y_stride, int n_store, int NB, int T, int P, float eps, int ln_n,
                                                           int act) {
    constexpr int K = KS * 16;
    constexpr int F = (LO ? 2 : 1) * KS;         // 1 KB fragments per 32-row block: hi plane k-steps, then (LO) lo plane k-steps; !LO: the layer runs on
                                                 // fp16 weights (the hi plane alone), for nets whose goldens allow it (ira_set_weight_planes)
    constexpr int FI = NBI * F;                  // fragments per iteration
    constexpr int SH = (FI + NW - 1) / NW;       // fragments each wave stages per iteration
    constexpr int RS = NBI * 64 + 16;            // transposition row stride in bytes (one pixel: NBI x 32 fp16 + pad)
    constexpr int CPR = NBI * 4;                 // 16-byte pieces per pixel row
    constexpr int RPI = 64 / CPR;                // pixel rows per store instruction
    constexpr int NST = PG * (32 / RPI);         // store instructions per iteration
    constexpr int WBUF = FI * 1024, BBUF = 256;  // one stage: weight fragments, then 64 bias floats (NBI x 32 used)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* tr = smem + 2 * (WBUF + BBUF) + wave * 32 * RS;                 // this wave's [32][RS]
    const int chunks = (P + 31) >> 5;
    const long total = (long)T * chunks;

    long pxbase[PG]; int nvalid[PG];
    half8 b[PG][KS];
#pragma unroll
    for (int g = 0; g < PG; ++g) {
        long gi = ((long)blockIdx.x * NW + wave) * PG + g;
        const bool live = gi < total;
        if (!live) gi = total - 1;
        const int t = (int)(gi / chunks), p0 = (int)(gi % chunks) * 32;
        pxbase[g] = (long)t * P + p0;
        nvalid[g] = live ? min(32, P - p0) : 0;
        const long px = pxbase[g] + min(r, P - p0 - 1);
        if constexpr (IN_F32) {
            // the lane's 8 of every 16 channels of its pixel, read ONCE into registers (K / 2 fp32 values: 96 at K = 192); the statistics are
            // the two-pass ones of k_pw_kres_lds, which re-read the row from memory for each pass (3 x 24 loads of 32 cache lines each)
            const float* src = (const float*)xin + px * K + 8 * h;
            f32x4 raw[KS][2];
#pragma unroll
            for (int s = 0; s < KS; ++s) { raw[s][0] = *(const f32x4*)(src + 16 * s); raw[s][1] = *(const f32x4*)(src + 16 * s + 4); }
            float sum = 0.f, sq = 0.f;
            if constexpr (LN != 0) {
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) sum += raw[s][0][j] + raw[s][1][j];
                sum += __shfl_xor(sum, 32);
            }
            const float inv_n = 1.0f / (float)ln_n;
            const float mu = sum * inv_n;
            if constexpr (LN != 0) {
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float d0 = raw[s][0][j] - mu, d1 = raw[s][1][j] - mu; sq += d0 * d0 + d1 * d1; }
                sq += __shfl_xor(sq, 32);
                sq -= (float)(K - ln_n) * mu * mu;
            }
            const float rstd = (LN != 0) ? 1.0f / sqrtf(sq * inv_n + eps) : 1.0f;
            const float sub = (LN == 2) ? mu : 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) { b[g][s][j] = (f16)((raw[s][0][j] - sub) * rstd); b[g][s][4 + j] = (f16)((raw[s][1][j] - sub) * rstd); }
        } else {
            const f16* src = (const f16*)xin + px * K + 8 * h;
#pragma unroll
            for (int s = 0; s < KS; ++s) b[g][s] = *(const half8*)(src + 16 * s);
        }
    }
    int wf = 1;
#pragma unroll
    for (int g = 0; g < PG; ++g) wf &= (nvalid[g] == 32);
    const bool wave_full = __builtin_amdgcn_readfirstlane(wf) != 0;

    const int nb_per = ((NB + gridDim.y - 1) / gridDim.y + NBI - 1) / NBI * NBI;
    const int nb_lo = blockIdx.y * nb_per, nb_hi = min(NB, nb_lo + nb_per);
    if (nb_lo >= nb_hi) return;                                         // whole workgroup (blockIdx.y is uniform)
    // Stage of the iteration that starts at block nb -> LDS buffer `buf`, by LDS-DMA (no registers, no ds_write): fragment f is block
    // nb + f / F (clamped), hi k-steps then lo k-steps; one global_load_lds_dwordx4 moves one 1 KB fragment (dest = base + 16 lane).
    const char* wlane = (const char*)W + lane * 16;
    auto stage = [&](int nb, int buf) {
#pragma unroll
        for (int i = 0; i < SH; ++i) {
            const int f = wave + NW * i;
            if (f < FI) {
                const int j = f / F, ff = f - j * F;
                const int blk = min(nb + j, nb_hi - 1);
                const long off = ((ff < KS ? 0 : (long)NB * KS) + (long)blk * KS + (ff < KS ? ff : ff - KS)) * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wlane + off),
                                                 (__attribute__((address_space(3))) void*)(smem + buf * (WBUF + BBUF) + f * 1024), 16, 0, 0);
            }
        }
        if (HAS_BIAS && wave == NW - 1) {                                 // 64 floats: bias of blocks nb, nb + 1 (reads stay inside [NB * 32] + 32 pad: clamped)
            const int idx = min(32 * nb + lane, 32 * NB - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bias + idx),
                                             (__attribute__((address_space(3))) void*)(smem + buf * (WBUF + BBUF) + WBUF), 4, 0, 0);
        }
    };
    stage(nb_lo, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int it = 0;
    for (int nb = nb_lo; nb < nb_hi; nb += NBI, ++it) {
        const char* stg = smem + (it & 1) * (WBUF + BBUF);
        const half8* cur = (const half8*)stg;
        const bool more = nb + NBI < nb_hi;
        if (more) stage(nb + NBI, (it + 1) & 1);
        f32x16 acc[PG][NBI];
#pragma unroll
        for (int j = 0; j < NBI; ++j) {
#pragma unroll
            for (int g = 0; g < PG; ++g) {
                if constexpr (HAS_BIAS) {
                    const float* bl = (const float*)(stg + WBUF) + 32 * j + 8 * h;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const f32x4 lo = *(const f32x4*)(bl + 16 * u), hi = *(const f32x4*)(bl + 16 * u + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { acc[g][j][8 * u + e] = lo[e]; acc[g][j][8 * u + 4 + e] = hi[e]; }
                    }
                } else {
                    acc[g][j] = zero16();
                }
            }
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int j = 0; j < NBI; ++j) {
                const half8 ah = cur[(j * F + s) * 64 + lane];
#pragma unroll
                for (int g = 0; g < PG; ++g) acc[g][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b[g][s], acc[g][j], 0, 0, 0);
                if constexpr (LO) {
                    const half8 al = cur[(j * F + KS + s) * 64 + lane];
#pragma unroll
                    for (int g = 0; g < PG; ++g) acc[g][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b[g][s], acc[g][j], 0, 0, 0);
                }
            }
        }
        // results: fp16, transposed through this wave's LDS rows, stored as whole pixel rows of NBI x 64 bytes
#pragma unroll
        for (int g = 0; g < PG; ++g) {
#pragma unroll
            for (int j = 0; j < NBI; ++j)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    half8 o;
                    if (act == 1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[g][j][8 * u + e] = gelu_erf(acc[g][j][8 * u + e]);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (f16)acc[g][j][8 * u + e];
                    *(half8*)(tr + r * RS + j * 64 + u * 32 + h * 16) = o;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int piece = lane % CPR, row0 = lane / CPR;
            const int n0 = 32 * nb + 8 * piece;
            const bool col_ok = n0 + 8 <= n_store && nb + piece / 4 < nb_hi;
#pragma unroll
            for (int i = 0; i < 32 / RPI; ++i) {
                const int row = row0 + RPI * i;
                const half8 o = *(const half8*)(tr + row * RS + piece * 16);
                if (col_ok && row < nvalid[g]) *(half8*)(y + (pxbase[g] + row) * y_stride + n0) = o;
            }
            asm volatile("" ::: "memory");
        }
        // The DMA of the next stage is OLDER than this iteration's NST stores on the wave's vector-memory counter: vmcnt(NST) retires it
        // without waiting for the stores - but only if all NST store instructions were issued, which holds when no lane is masked off
        // (a wholly masked store may be branched over).  Any other wave waits for everything.
        if (wave_full && 32 * (nb + NBI) <= n_store && nb + NBI <= nb_hi) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

int env_int_w(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

template <int KS, int NW, int PG, int NBI, int MINW, bool LO>
hipError_t pw_wide_launch(hipStream_t st, const void* x, bool in_f32, int ln, const f16* W, const float* bias, f16* y, int ys, int ns, int NB,
                          int T, int P, float eps, int ln_n, int act) {
    const long groups = (long)T * ((P + 31) / 32);
    const long wgs = (groups + (long)NW * PG - 1) / ((long)NW * PG);
    int splits = 1;
    if (wgs < 256 && NB >= 8) splits = (int)std::min<long>(NB / (2 * NBI), (512 + wgs - 1) / wgs);
    const dim3 grid((unsigned)wgs, std::max(splits, 1)), block(64 * NW);
    const size_t smem = (size_t)2 * (NBI * (LO ? 2 : 1) * KS * 1024 + 256) + (size_t)NW * 32 * (NBI * 64 + 16);
#define IRA_KW2(LN_, F32_, B_)                                                                                                                          \
    do {                                                                                                                                                \
        static std::atomic<uint64_t> done_{0};                                                                                                          \
        if (hipError_t e_ = ensure_dynamic_lds((const void*)k_pw_wide<KS, LN_, F32_, B_, NW, PG, NBI, MINW, LO>, smem, done_); e_ != hipSuccess) return e_; \
        hipLaunchKernelGGL((k_pw_wide<KS, LN_, F32_, B_, NW, PG, NBI, MINW, LO>), grid, block, smem, st, x, W, bias, y, ys, ns, NB, T, P, eps, ln_n, act);   \
    } while (0)
#define IRA_KW(LN_, F32_) do { if (bias != nullptr) IRA_KW2(LN_, F32_, true); else IRA_KW2(LN_, F32_, false); } while (0)
    if (in_f32) {
        if (ln == 0) IRA_KW(0, true); else if (ln == 1) IRA_KW(1, true); else IRA_KW(2, true);
    } else {
        if (ln != 0) return hipErrorInvalidValue;
        IRA_KW(0, false);
    }
#undef IRA_KW
#undef IRA_KW2
    return hipGetLastError();
}

}  // namespace

bool pw_wide_supported(int K, long npx, int planes) {
    static const int on = env_int_w("IRA_PW_WIDE", 1);
    if (!on) return false;
    // K = 256 (SwinIR-L, 240 channels padded) with both planes was tried with one row block per iteration (two need 2 x 64 KB of weight
    // stages): LN + qkv 31.3 -> 31.2 ms per frame, LN + fc1 25.8 -> 27.9: not dispatched.  With the hi plane alone two row blocks fit.
    if (planes == 1) return ((K == 192 || K == 256) && npx >= 512L * 64) || (K == 384 && npx >= 256L * 64);
    return (K == 192 && npx >= 512L * 64) || (K == 384 && npx >= 256L * 64);
}

hipError_t launch_pw_wide(hipStream_t st, const void* x, bool in_f32, int ln_mode, int K, const void* W, const float* bias, void* y, int y_stride,
                          int n_store, int NB, int T, int P, float eps, int ln_n, int act, int planes) {
    if (ln_n <= 0 || ln_n > K) ln_n = K;
    if (n_store % 8 != 0 || y_stride % 8 != 0 || ln_mode < 0 || ln_mode > 2) return hipErrorInvalidValue;
#define IRA_PWW(KS_, NW_, PG_, NBI_, LO_) \
    return pw_wide_launch<KS_, NW_, PG_, NBI_, 2, LO_>(st, x, in_f32, ln_mode, (const f16*)W, bias, (f16*)y, y_stride, n_store, NB, T, P, eps, ln_n, act)
    if (planes == 1) {
        if (K == 192) IRA_PWW(12, 8, 2, 2, false);
        if (K == 256) IRA_PWW(16, 8, 2, 2, false);
        if (K == 384) IRA_PWW(24, 8, 1, 2, false);      // two row blocks per iteration fit with one plane: whole-line stores
        return hipErrorInvalidValue;
    }
    if (K == 192) IRA_PWW(12, 8, 2, 2, true);
    if (K == 384) IRA_PWW(24, 8, 1, 1, true);
#undef IRA_PWW
    return hipErrorInvalidValue;
}

}  // namespace ira
