// SwinIR MLP half-block in one kernel (network_swinir.py:274-277 with Mlp :14-30):   x += fc2( GELU( fc1( LayerNorm(x) ) ) )
// for token rows of 256 padded channels (240 real) and 480 hidden channels (SwinIR-L, mlp_ratio 2).  The two kernels it replaces
// (k_pw_kres_lds / k_pw_wide with GELU, then k_nres_wide) write and re-read the hidden tensor: 1.9 KB of the 3.9 KB they move per token.
// Structure = k_pw_wide's pixel-stationary GEMM chained into k_nres_wide's accumulator-stationary one:
//   * a workgroup is 8 waves x 32 tokens; a wave keeps the LayerNorm-ed fp16 B fragments of its tokens in registers (16 k-steps) and the
//     fc2 accumulators of all 8 output row blocks (128 registers);
//   * the hidden dimension is walked in chunks of 32 channels = one MFMA row block of fc1: 16 (x2 planes) MFMAs give the chunk's hidden
//     values in the accumulator layout, which IS the B-fragment layout of the two fc2 k-steps that contract over those 32 channels
//     (ira_kernels.hip header): GELU, round to fp16, 2 x 8 (x2) MFMAs into the fc2 accumulators.  The hidden tensor never exists;
//   * per chunk the fc1 row block (16 KB per plane), the fc2 slab of the chunk's two k-steps (16 KB per plane) and the 32 fc1 biases are
//     staged one chunk ahead by LDS-DMA into a double buffer; one barrier per chunk;
//   * epilogue as k_nres_wide: accumulators (started from the fc2 bias) through LDS two row blocks at a time, + residual in 256-byte runs,
//     in place.
// Same MFMA order per accumulator and the same rounding points (fp16 LayerNorm output, GELU then fp16 hidden, fp32 accumulation) as the
// two-kernel path: bitwise the same result (tests/test_gpu_swinir.py).  LO = false: plain fp16 weights (ira_set_weight_planes).
#include "ira_kernels.h"
#include "ira_device.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>

namespace ira {

namespace {

constexpr int MLP_KS = 16;      // fc1 k-steps: 256 padded input channels
constexpr int MLP_NB2 = 8;      // fc2 output row blocks: 256 padded output channels
constexpr int MLP_NW = 8;

// DBG (IRA_SWIN_MLP_DBG, measurement only, wrong results): 1 no GELU, 2 fc1 only, 3 fc2 + GELU only, 4 prologue + epilogue, 5 staging + barriers only
template <bool LO, int DBG = 0>
__global__ __launch_bounds__(64 * MLP_NW, 2) void k_swin_mlp(float* __restrict__ x, const f16* __restrict__ W1, const float* __restrict__ b1,
                                                             const f16* __restrict__ W2, const float* __restrict__ b2, int NB1, long npx,
                                                             float eps, int ln_n) {
    constexpr int KS = MLP_KS, NB2 = MLP_NB2, NW = MLP_NW, K = KS * 16;
    constexpr int NPL = LO ? 2 : 1;
    constexpr int F1 = NPL * KS, F2 = NPL * 2 * NB2, FS = F1 + F2;          // 1 KB fragments per chunk: fc1 [plane][k-step], fc2 [k-step][plane][row block]
    constexpr int WBUF = FS * 1024, BBUF = 256;
    constexpr int RS = 64 * 4 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long p0 = ((long)blockIdx.x * NW + wave) * 32;                     // this wave's first token
    const int nvalid = p0 < npx ? (int)((npx - p0) < 32 ? (npx - p0) : 32) : 0;
    const long px = (p0 + r < npx) ? p0 + r : npx - 1;
    const int KT2 = 2 * NB1;                                                  // fc2 k-steps (hidden / 16)

    // LayerNorm (the arithmetic of k_pw_wide / k_pw_kres_lds) -> fp16 B fragments of the wave's 32 tokens
    half8 b[KS];
    {
        const float* src = x + px * K + 8 * h;
        f32x4 raw[KS][2];
#pragma unroll
        for (int s = 0; s < KS; ++s) { raw[s][0] = *(const f32x4*)(src + 16 * s); raw[s][1] = *(const f32x4*)(src + 16 * s + 4); }
        float sum = 0.f, sq = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += raw[s][0][j] + raw[s][1][j];
        sum += __shfl_xor(sum, 32);
        const float inv_n = 1.0f / (float)ln_n;
        const float mu = sum * inv_n;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d0 = raw[s][0][j] - mu, d1 = raw[s][1][j] - mu; sq += d0 * d0 + d1 * d1; }
        sq += __shfl_xor(sq, 32);
        sq -= (float)(K - ln_n) * mu * mu;
        const float rstd = 1.0f / sqrtf(sq * inv_n + eps);
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) { b[s][j] = (f16)((raw[s][0][j] - mu) * rstd); b[s][4 + j] = (f16)((raw[s][1][j] - mu) * rstd); }
    }

    f32x16 acc2[NB2];                                                        // start from the fc2 bias, as k_nres_wide does (same summation order)
#pragma unroll
    for (int nb = 0; nb < NB2; ++nb)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc2[nb][8 * u + e] = b2[32 * nb + 16 * u + 8 * h + e];

    const char* w1lane = (const char*)W1 + lane * 16;
    const char* w2lane = (const char*)W2 + lane * 16;
    const long lo1 = (long)NB1 * KS * 1024, lo2 = (long)NB2 * KT2 * 1024;    // bytes between the planes
    auto stage = [&](int c, int buf) {
        char* dst = smem + buf * (WBUF + BBUF);
#pragma unroll
        for (int i = 0; i < (FS + NW - 1) / NW; ++i) {
            const int f = wave + NW * i;
            if (f < FS) {
                const char* src;
                if (f < F1) {
                    const int plane = f / KS, s = f - plane * KS;
                    src = w1lane + (plane ? lo1 : 0) + ((long)c * KS + s) * 1024;
                } else {
                    const int g = f - F1, ks = g / (NPL * NB2), rem = g - ks * NPL * NB2, plane = rem / NB2, nb = rem - plane * NB2;
                    src = w2lane + (plane ? lo2 : 0) + ((long)nb * KT2 + 2 * c + ks) * 1024;
                }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
            }
        }
        if (wave == NW - 1) {                                                // the chunk's 32 fc1 biases (64 floats moved, the array is clamped)
            const int idx = (32 * c + lane < 32 * NB1) ? 32 * c + lane : 32 * NB1 - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b1 + idx),
                                             (__attribute__((address_space(3))) void*)(dst + WBUF), 4, 0, 0);
        }
    };
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    for (int c = 0; c < (DBG == 4 ? 0 : NB1); ++c) {
        const char* cur = smem + (c & 1) * (WBUF + BBUF);
        const half8* w1 = (const half8*)cur;
        const half8* w2 = (const half8*)(cur + F1 * 1024);
        if (c + 1 < NB1) stage(c + 1, (c + 1) & 1);
        // fc1 row block c: hidden channels 32c .. 32c+31 of the wave's 32 tokens
        f32x16 acc1;
        {
            const float* bl = (const float*)(cur + WBUF) + 8 * h;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x4 lo = *(const f32x4*)(bl + 16 * u), hi = *(const f32x4*)(bl + 16 * u + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc1[8 * u + e] = lo[e]; acc1[8 * u + 4 + e] = hi[e]; }
            }
        }
        if constexpr (DBG != 3 && DBG != 5)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[s * 64 + lane], b[s], acc1, 0, 0, 0);
            if constexpr (LO) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[(KS + s) * 64 + lane], b[s], acc1, 0, 0, 0);
        }
        // GELU (exact-erf form of the two-kernel path), fp16: registers 0-7 / 8-15 are the B fragments of fc2 k-steps 2c / 2c+1
        half8 hf[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) hf[u][e] = (DBG == 1 || DBG == 5) ? (f16)acc1[8 * u + e] : (f16)gelu_erf(acc1[8 * u + e]);
        if constexpr (DBG == 2 || DBG == 5) {
            if (hf[0][0] == (f16)12345.f && hf[1][7] == (f16)-4321.f) acc2[0][0] += 1.f;   // keeps fc1 alive
        } else
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nb = 0; nb < NB2; ++nb) {
                acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[(ks * NPL * NB2 + nb) * 64 + lane], hf[ks], acc2[nb], 0, 0, 0);
                if constexpr (LO) acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[(ks * NPL * NB2 + NB2 + nb) * 64 + lane], hf[ks], acc2[nb], 0, 0, 0);
            }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // no store in the loop: everything outstanding is the next stage
        __builtin_amdgcn_s_barrier();
    }

    // epilogue: x += acc2 + b2, two row blocks (64 channels = 256 bytes per token) at a time through this wave's LDS rows
    char* tr = smem + wave * 32 * RS;
    const int piece = lane & 15, row0 = lane >> 4;
#pragma unroll
    for (int pr = 0; pr < NB2 / 2; ++pr) {
        const int n0 = 64 * pr + 4 * piece;
        f32x4 rs[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = row0 + 4 * i;
            rs[i] = row < nvalid ? *(const f32x4*)(x + (p0 + row) * K + n0) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 lo, hi;
#pragma unroll
                for (int e = 0; e < 4; ++e) { lo[e] = acc2[2 * pr + j][8 * u + e]; hi[e] = acc2[2 * pr + j][8 * u + 4 + e]; }
                float* dst = (float*)(tr + r * RS) + 32 * j + 16 * u + 8 * h;
                *(f32x4*)dst = lo;
                *(f32x4*)(dst + 4) = hi;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = row0 + 4 * i;
            f32x4 v = *(const f32x4*)(tr + row * RS + piece * 16);
            v += rs[i];                                                      // residual last, as in the two-kernel path
            if (row < nvalid) *(f32x4*)(x + (p0 + row) * K + n0) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

int env_int_m(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

}  // namespace

bool swin_mlp_supported(int Cp, int hidp, long npx) {
    static const int on = env_int_m("IRA_SWIN_MLP", 1);
    return on && Cp == 256 && hidp % 32 == 0 && hidp >= 64 && hidp <= 1024 && npx >= 32768;
}

hipError_t launch_swin_mlp(hipStream_t st, float* x, const void* W1, const float* b1, const void* W2, const float* b2, int Cp, int hidp, long npx,
                           float eps, int ln_n, int planes) {
    if (!swin_mlp_supported(Cp, hidp, npx) || !b1 || !b2) return hipErrorInvalidValue;
    const int NB1 = hidp / 32;
    const dim3 grid((unsigned)((npx + 32 * MLP_NW - 1) / (32 * MLP_NW))), block(64 * MLP_NW);
    auto smem_of = [](int npl) { return std::max((size_t)2 * ((size_t)(npl * MLP_KS + npl * 2 * MLP_NB2) * 1024 + 256), (size_t)MLP_NW * 32 * (64 * 4 + 16)); };
    static std::atomic<uint64_t> d1{0}, d2{0};
    static const int dbg = env_int_m("IRA_SWIN_MLP_DBG", 0);
    if (planes == 1 && dbg >= 1 && dbg <= 5) {
        const size_t smem = smem_of(1);
        const void* fn = dbg == 1 ? (const void*)k_swin_mlp<false, 1> : dbg == 2 ? (const void*)k_swin_mlp<false, 2> : dbg == 3 ? (const void*)k_swin_mlp<false, 3>
                       : dbg == 4 ? (const void*)k_swin_mlp<false, 4> : (const void*)k_swin_mlp<false, 5>;
        static std::atomic<uint64_t> dd{0};
        if (hipError_t e = ensure_dynamic_lds(fn, smem, dd); e != hipSuccess) return e;
        const f16 *w1 = (const f16*)W1, *w2 = (const f16*)W2;
        void* args[] = {&x, &w1, &b1, &w2, &b2, (void*)&NB1, &npx, &eps, &ln_n};
        return hipLaunchKernel(fn, grid, block, args, smem, st);
    }
    if (planes == 1) {
        const size_t smem = smem_of(1);
        if (hipError_t e = ensure_dynamic_lds((const void*)k_swin_mlp<false>, smem, d1); e != hipSuccess) return e;
        hipLaunchKernelGGL(k_swin_mlp<false>, grid, block, smem, st, x, (const f16*)W1, b1, (const f16*)W2, b2, NB1, npx, eps, ln_n);
    } else {
        const size_t smem = smem_of(2);
        if (hipError_t e = ensure_dynamic_lds((const void*)k_swin_mlp<true>, smem, d2); e != hipSuccess) return e;
        hipLaunchKernelGGL(k_swin_mlp<true>, grid, block, smem, st, x, (const f16*)W1, b1, (const f16*)W2, b2, NB1, npx, eps, ln_n);
    }
    return hipGetLastError();
}

}  // namespace ira
