// 1x1 conv + residual for the deep levels: y[px][N] (fp32) = resid[px][N] + W[N][K] . x[px][K] (x fp16), N = 192 / 384 - the GDFN
// project_out (restormer_arch.py:92,149) and the attention apply with its per-tile matrix M_t (:127-131,148).  Second version of k_pw_gemm
// (ira_nafnet.hip) for these shapes, built like k_pw_wide (ira_pw_wide.hip), whose measurements apply here too: k_pw_gemm synchronises its
// workgroup once per 16-channel k-step with 6-8 MFMAs per wave in between, re-reads x once per 96 output channels and stores the result
// as 32-byte pieces.  Here
//   * a workgroup is 8 waves x 32 pixels with the accumulators of SIX row blocks (192 output channels) per wave; N = 384 runs as two
//     halves (blockIdx.y), x (1-2 KB per pixel) being re-read from L2;
//   * the weights stream through an LDS double buffer in stages of FOUR k-steps (48 KB: 48 MFMAs per wave between barriers), staged by
//     LDS-DMA one stage ahead; the pixel fragments of the next stage are requested at the same time, straight into registers;
//   * the epilogue transposes the accumulators through LDS (the weight buffers are dead by then) two row blocks at a time and adds the
//     residual in the transposed layout: residual loads and result stores are whole 256-byte runs per pixel, 16 lanes each.
// Numerics: the same hi / lo MFMA order per accumulator and k-step as k_pw_gemm, bias in the accumulator init, residual added last.
#include "ira_kernels.h"
#include "ira_device.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace ira {

namespace {

template <int NBW, int NW, int KSI, bool HAS_BIAS, bool LO>
__global__ __launch_bounds__(64 * NW, 2) void k_nres_wide(const f16* __restrict__ x, int K, const f16* __restrict__ W, long w_tile_stride,
                                                          const float* __restrict__ bias, const float* resid, float* y, int N, int NBtot, int T,
                                                          int P) {
    static_assert(NBW % 2 == 0, "the epilogue moves two row blocks at a time");
    constexpr int NPL = LO ? 2 : 1;              // weight planes used: hi + lo, or (ira_set_weight_planes(1)) the fp16 hi plane alone
    constexpr int FS = KSI * NBW * NPL;          // 1 KB fragments per stage: [k-step][plane][row block]
    constexpr int WBUF = FS * 1024;
    constexpr int RS = 64 * 4 + 16;              // transposition row stride in bytes: 64 floats + pad

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg_per_tile = (P + 32 * NW - 1) / (32 * NW);
    const int t = blockIdx.x / wg_per_tile;
    const int p0 = ((blockIdx.x % wg_per_tile) * NW + wave) * 32;       // this wave's first pixel inside the tile
    const int nvalid = p0 < P ? min(32, P - p0) : 0;
    const long px = (long)t * P + min(p0 + r, P - 1);
    const int nb0 = blockIdx.y * NBW;
    const int KT = K >> 4, nst = KT / KSI;

    f32x16 acc[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        acc[nb] = zero16();
        if constexpr (HAS_BIAS) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int n = 32 * (nb0 + nb) + 16 * u + 8 * h + e;
                    acc[nb][8 * u + e] = n < N ? bias[n] : 0.f;
                }
        }
    }
    const char* wlane = (const char*)(W + (long)t * w_tile_stride) + lane * 16;
    const long lo_off = (long)NBtot * KT * 1024;                        // bytes between the planes
    auto stage = [&](int sti, int buf) {
#pragma unroll
        for (int i = 0; i < (FS + NW - 1) / NW; ++i) {
            const int f = wave + NW * i;
            if (f < FS) {
                const int ks = f / (NPL * NBW), rem = f - ks * NPL * NBW, plane = rem / NBW, nb = rem - plane * NBW;
                const int blk = min(nb0 + nb, NBtot - 1);
                const long off = (plane ? lo_off : 0) + ((long)blk * KT + sti * KSI + ks) * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wlane + off),
                                                 (__attribute__((address_space(3))) void*)(smem + buf * WBUF + f * 1024), 16, 0, 0);
            }
        }
    };
    half8 bq[2][KSI];
    const f16* xrow = x + px * K + 8 * h;
    auto xload = [&](int sti, int set) {
#pragma unroll
        for (int ks = 0; ks < KSI; ++ks) bq[set][ks] = *(const half8*)(xrow + 16 * (sti * KSI + ks));
    };
    stage(0, 0);
    xload(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    auto compute = [&](int sti, auto SETc) {
        constexpr int set = decltype(SETc)::value;
        if (sti + 1 < nst) { stage(sti + 1, set ^ 1); xload(sti + 1, set ^ 1); }
        const half8* cur = (const half8*)(smem + set * WBUF);
#pragma unroll
        for (int ks = 0; ks < KSI; ++ks)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                const half8 ah = cur[(ks * NPL * NBW + nb) * 64 + lane];
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bq[set][ks], acc[nb], 0, 0, 0);
                if constexpr (LO) {
                    const half8 al = cur[(ks * NPL * NBW + NBW + nb) * 64 + lane];
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bq[set][ks], acc[nb], 0, 0, 0);
                }
            }
        // no store is issued in this loop: everything outstanding on the vector-memory counter is the next stage (DMA + pixel fragments)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    for (int sti = 0; sti < nst; sti += 2) {                            // two stages per trip: the register set of the pixel fragments is static
        compute(sti, std::integral_constant<int, 0>{});
        if (sti + 1 < nst) compute(sti + 1, std::integral_constant<int, 1>{});
    }

    // epilogue (every wave is past the last barrier: the weight buffers are free)
    char* tr = smem + wave * 32 * RS;
    const int piece = lane & 15, row0 = lane >> 4;
#pragma unroll
    for (int pr = 0; pr < NBW / 2; ++pr) {
        const int n0 = 32 * (nb0 + 2 * pr) + 4 * piece;
        f32x4 rs[8];
        if (resid != nullptr) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = row0 + 4 * i;
                rs[i] = (row < nvalid && n0 + 4 <= N) ? *(const f32x4*)(resid + ((long)t * P + p0 + row) * N + n0) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 lo, hi;
#pragma unroll
                for (int e = 0; e < 4; ++e) { lo[e] = acc[2 * pr + j][8 * u + e]; hi[e] = acc[2 * pr + j][8 * u + 4 + e]; }
                float* dst = (float*)(tr + r * RS) + 32 * j + 16 * u + 8 * h;
                *(f32x4*)dst = lo;
                *(f32x4*)(dst + 4) = hi;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = row0 + 4 * i;
            f32x4 v = *(const f32x4*)(tr + row * RS + piece * 16);
            if (resid != nullptr) v += rs[i];
            if (row < nvalid && n0 + 4 <= N) *(f32x4*)(y + ((long)t * P + p0 + row) * N + n0) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

int env_int_n(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

}  // namespace

bool nres_wide_supported(int K, int N, int NB, int T, int P, long w_tile_stride) {
    static const int on = env_int_n("IRA_NRES_WIDE", 1);
    if (!on || N % 4 != 0 || N <= 32 * (NB - 1) || (long)T * P < 32768 || T <= 0 || P <= 0 || K < 128) return false;
    if (NB == 6 || NB == 12) return K % 64 == 0;          // Restormer / X-Restormer levels 3-4: six row blocks, stages of four k-steps
    if (NB == 8) return K % 64 == 0 || K % 48 == 0;        // SwinIR-L proj / fc2 (N = 256 padded, K = 256 / 480): eight row blocks, stages of four / three k-steps
    return false;
}

template <int NBW, int KSI, bool LO>
static hipError_t nres_wide_launch(hipStream_t st, const void* x, int K, const void* W, long w_tile_stride, const float* bias, const float* resid,
                                   float* y, int N, int NB, int T, int P) {
    constexpr int NW = 8;
    const dim3 grid((unsigned)((long)T * ((P + 32 * NW - 1) / (32 * NW))), (unsigned)(NB / NBW)), block(64 * NW);
    // two weight stages, and at least the epilogue's transposition rows (8 waves x 32 x 272 B), which reuse the same LDS
    const size_t smem = std::max((size_t)2 * KSI * NBW * (LO ? 2 : 1) * 1024, (size_t)NW * 32 * (64 * 4 + 16));
    static std::atomic<uint64_t> done_b{0}, done_n{0};
    if (bias != nullptr) {
        if (hipError_t e = ensure_dynamic_lds((const void*)k_nres_wide<NBW, NW, KSI, true, LO>, smem, done_b); e != hipSuccess) return e;
        hipLaunchKernelGGL((k_nres_wide<NBW, NW, KSI, true, LO>), grid, block, smem, st, (const f16*)x, K, (const f16*)W, w_tile_stride, bias, resid, y, N, NB, T, P);
    } else {
        if (hipError_t e = ensure_dynamic_lds((const void*)k_nres_wide<NBW, NW, KSI, false, LO>, smem, done_n); e != hipSuccess) return e;
        hipLaunchKernelGGL((k_nres_wide<NBW, NW, KSI, false, LO>), grid, block, smem, st, (const f16*)x, K, (const f16*)W, w_tile_stride, bias, resid, y, N, NB, T, P);
    }
    return hipGetLastError();
}

hipError_t launch_nres_wide(hipStream_t st, const void* x, int K, const void* W, long w_tile_stride, const float* bias, const float* resid, float* y,
                            int N, int NB, int T, int P, int planes) {
    if (NB == 8) {
        if (planes == 1) {
            if (K % 64 == 0) return nres_wide_launch<8, 4, false>(st, x, K, W, w_tile_stride, bias, resid, y, N, NB, T, P);
            return nres_wide_launch<8, 3, false>(st, x, K, W, w_tile_stride, bias, resid, y, N, NB, T, P);
        }
        if (K % 64 == 0) return nres_wide_launch<8, 4, true>(st, x, K, W, w_tile_stride, bias, resid, y, N, NB, T, P);
        return nres_wide_launch<8, 3, true>(st, x, K, W, w_tile_stride, bias, resid, y, N, NB, T, P);
    }
    if (planes == 1) return nres_wide_launch<6, 4, false>(st, x, K, W, w_tile_stride, bias, resid, y, N, NB, T, P);
    return nres_wide_launch<6, 4, true>(st, x, K, W, w_tile_stride, bias, resid, y, N, NB, T, P);
}

}  // namespace ira
