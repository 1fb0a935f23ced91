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
#include <type_traits>

#ifdef IRA_MLP_STAMPS                                                        // timeline build (tools/mlp_stamps.py), never the product library
__device__ unsigned long long g_mlp_stamps[160];
#define MLP_STAMP(i) do { if (blockIdx.x == 777 && threadIdx.x == 0 && (i) < 160) g_mlp_stamps[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int ira_debug_mlp_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_stamps), sizeof(unsigned long long) * (n < 160 ? n : 160));
}
#define MLP_RSTAMP(i) do { if (blockIdx.x == 777 && threadIdx.x == 0) g_mlp_stamps[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)   // 100 MHz
#else
#define MLP_RSTAMP(i) do {} while (0)
#define MLP_STAMP(i) do {} while (0)
#endif

namespace ira {

namespace {

typedef f16 half2 __attribute__((ext_vector_type(2)));

constexpr int MLP_KS = 16;      // fc1 k-steps: 256 padded input channels
constexpr int MLP_NB2 = 8;      // fc2 output row blocks: 256 padded output channels

// DBG (IRA_SWIN_MLP_DBG, measurement only, wrong results): 1 no GELU, 2 fc1 only, 3 fc2 + GELU only, 4 prologue + epilogue, 5 staging + barriers only
// PROJ: the attention projection + residual in front (network_swinir.py:263-271 after the window attention):  x += W_p attn + b_p  first, with
// the arithmetic of k_nres_wide<8, 8, 4> (bias in the accumulators, k-step-major MFMA order, residual last); x is written back once (the
// epilogue's residual) and normalised from the accumulators, whose layout is the B-fragment layout fc1 wants.
template <bool LO, int NW, bool PROJ, int DBG = 0>
__global__ __launch_bounds__(64 * NW, 2) void k_swin_mlp(float* __restrict__ x, const f16* __restrict__ W1, const float* __restrict__ b1,
                                                         const f16* __restrict__ W2, const float* __restrict__ b2, const float2* __restrict__ phi,
                                                         const f16* __restrict__ attn, const f16* __restrict__ Wp, const float* __restrict__ bp,
                                                         const float* xin, int NB1, long npx, float eps, int ln_n) {
    constexpr int KS = MLP_KS, NB2 = MLP_NB2, K = KS * 16;
    constexpr int NPL = LO ? 2 : 1;
    constexpr int F1 = NPL * KS, F2 = NPL * 2 * NB2, FS = F1 + F2;          // 1 KB fragments per stage: fc1 [plane][k-step], fc2 [k-step][plane][row block]
    constexpr int WBUF = FS * 1024, BBUF = 256;
    static_assert(PHI_NTAB * 8 == 8192, "the table is moved as eight 1 KB pieces");
    constexpr int RS = 64 * 4 + 16;
    constexpr int PF = 4;                                                     // weight fragments requested from LDS ahead of the MFMA that uses them
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long p0 = ((long)blockIdx.x * NW + wave) * 32;                     // this wave's first token
    const int nvalid = p0 < npx ? (int)((npx - p0) < 32 ? (npx - p0) : 32) : 0;
    const long px = (p0 + r < npx) ? p0 + r : npx - 1;
    const int KT2 = 2 * NB1;                                                  // fc2 k-steps (hidden / 16)

    MLP_STAMP(0);
    MLP_RSTAMP(5);
    const char* w1lane = (const char*)W1 + lane * 16;
    const char* w2lane = (const char*)W2 + lane * 16;
    const long lo1 = (long)NB1 * KS * 1024, lo2 = (long)NB2 * KT2 * 1024;    // bytes between the planes
    // stage c = what trip c of the loop below consumes: fc1 row block c (c < NB1), its 32 biases, and the fc2 slab of chunk c - 1 (c >= 1)
    auto stage = [&](int c, int buf) {
        char* dst = smem + buf * (WBUF + BBUF);
#pragma unroll
        for (int i = 0; i < (FS + NW - 1) / NW; ++i) {
            const int f = wave + NW * i;
            if (f < F1) {
                if (c < NB1) {
                    const int plane = f / KS, s = f - plane * KS;
                    const char* src = w1lane + (plane ? lo1 : 0) + ((long)c * KS + s) * 1024;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
                }
            } else if (f < FS) {
                if (c >= 1) {
                    const int g = f - F1, ks = g / (NPL * NB2), rem = g - ks * NPL * NB2, plane = rem / NB2, nb = rem - plane * NB2;
                    const char* src = w2lane + (plane ? lo2 : 0) + ((long)nb * KT2 + 2 * (c - 1) + ks) * 1024;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
                }
            }
        }
        if (wave == NW - 1 && c < NB1) {                                     // 64 floats moved, 32 used: the index is clamped to the array
            const int idx = (32 * c + lane < 32 * NB1) ? 32 * c + lane : 32 * NB1 - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b1 + idx),
                                             (__attribute__((address_space(3))) void*)(dst + WBUF), 4, 0, 0);
        }
    };
    // DMA pieces of stage cn without conditions, for the steady-state trips (2 <= cn <= NB1; the fc1 block of stage NB1 does not exist:
    // block NB1 - 1 is fetched again and never read).  Piece i of wave w is fragment f = w + NW i of the stage; the two source bases of
    // the trip go through an empty asm so that the eight piece addresses are one add each from them, not sixteen hoisted registers.
    static_assert(FS % NW == 0 && F1 % NW == 0 && KS % NW == 0 && NB2 % NW == 0, "every wave issues the same pieces, of the same kinds");
    constexpr int NPC = FS / NW;                                             // DMA pieces per wave and stage
    static_assert(NPC == KS / 2, "one DMA piece rides behind each k-step of the second half of fc1");
    unsigned dma_l16 = 0;                                                    // lane * 16, the only vector part of a piece address (SGPR base + 32-bit offset form)
    auto dma_bases = [&]() {
        dma_l16 = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) << 4;
        asm volatile("" : "+v"(dma_l16));                                    // opaque: no per-piece address pair is precomputed and kept (spilled) across trips
    };
    auto dma_piece = [&](int cn, int i) {
        char* dst = smem + (cn & 1) * (WBUF + BBUF) + (wave + NW * i) * 1024;
        const char* src;
        if (i < F1 / NW) {
            const int plane = i / (KS / NW), sp = NW * (i % (KS / NW));      // k-step = wave + sp
            const int blk = cn < NB1 ? cn : NB1 - 1;
            src = (const char*)W1 + (plane ? lo1 : 0) + ((long)blk * KS + wave + sp) * 1024;
        } else {
            const int g0 = NW * (i - F1 / NW);                               // fragment g0 + wave of the fc2 part: [k-step][plane][row block]
            const int ks = g0 / (NPL * NB2), rem0 = g0 % (NPL * NB2), plane = rem0 / NB2, nb0 = rem0 % NB2;   // row block = nb0 + wave
            src = (const char*)W2 + (plane ? lo2 : 0) + ((long)(nb0 + wave) * KT2 + 2 * (cn - 1) + ks) * 1024;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + dma_l16), (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    auto bias_piece_all = [&](int cn) {                                      // every wave moves the same 256 bytes to the same place: no branch in the trip
        const int blk = cn < NB1 ? cn : NB1 - 1;
        const int idx = (32 * blk + lane < 32 * NB1) ? 32 * blk + lane : 32 * NB1 - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b1 + idx),
                                         (__attribute__((address_space(3))) void*)(smem + (cn & 1) * (WBUF + BBUF) + WBUF), 4, 0, 0);
    };
    // first the requests that do not depend on the tokens: stage 0 by DMA and the Phi table (8 KB; a separate LDS object that no DMA ever
    // writes: its look-ups then need no vmcnt wait - inside the dynamic array, or filled by LDS-DMA, every table read waited for vmcnt(0),
    // i.e. for all the DMA pieces in flight); they travel while the tokens are loaded and normalised
    if constexpr (!PROJ) stage(0, 0);
    __shared__ float2 tab[PHI_NTAB];
    f32x4 treg[512 / (64 * NW)];
#pragma unroll
    for (int i = 0; i < 512 / (64 * NW); ++i) treg[i] = ((const f32x4*)phi)[threadIdx.x + 64 * NW * i];

    half8 b[KS];
    f32x16 acc2[NB2];
    if constexpr (PROJ) {
        // ---- x1 = x + W_p attn + b_p.  W_p travels in four stages of four k-steps, [k-step][plane][row block] like k_nres_wide, through the
        // two stage buffers (a stage is FS fragments there too); the MLP's stage 0 is requested during the last of them
        constexpr int KSI = 4, PST = KS / KSI;
        static_assert(KSI * NPL * NB2 == FS, "a W_p stage fills a stage buffer exactly");
        const char* wplane = (const char*)Wp + lane * 16;
        const long lop = (long)NB2 * KS * 1024;
        auto ppiece = [&](int sti, int i) {                                  // DMA piece i of this wave for W_p stage sti
            const int f = wave + NW * i;
            const int ks = f / (NPL * NB2), rem = f - ks * NPL * NB2, plane = rem / NB2, nb = rem - plane * NB2;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wplane + (plane ? lop : 0) + ((long)nb * KS + sti * KSI + ks) * 1024),
                                             (__attribute__((address_space(3))) void*)(smem + (sti & 1) * (WBUF + BBUF) + f * 1024), 16, 0, 0);
        };
#pragma unroll
        for (int i = 0; i < NPC; ++i) ppiece(0, i);
        const f16* arow = attn + px * K + 8 * h;
#pragma unroll
        for (int s = 0; s < KS; ++s) b[s] = *(const half8*)(arow + 16 * s);
#pragma unroll
        for (int nb = 0; nb < NB2; ++nb)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc2[nb][8 * u + e] = bp[32 * nb + 16 * u + 8 * h + e];
        dma_bases();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int sti = 0; sti < PST; ++sti) {
            const half8* cur = (const half8*)(smem + (sti & 1) * (WBUF + BBUF)) + lane;
            // MFMA order (k-step, row block, plane) as k_nres_wide; fragment (ks, plane, nb) of the stage sits at ks NPL NB2 + plane NB2 + nb
            auto offp = [](int q) { const int plane = q % NPL, nb = (q / NPL) % NB2, ks = q / (NPL * NB2); return (ks * NPL * NB2 + plane * NB2 + nb) * 64; };
            constexpr int PFP = 6;                                           // plenty of registers here: no fc1 accumulators yet
            half8 a[PFP];
#pragma unroll
            for (int q = 0; q < PFP; ++q) a[q] = cur[offp(q)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < FS; ++q) {                                   // the next stage's DMA pieces ride behind every NW-th MFMA, in this order
                const int nb = (q / NPL) % NB2, ks = q / (NPL * NB2);
                acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q % PFP], b[sti * KSI + ks], acc2[nb], 0, 0, 0);
                if (q + PFP < FS) a[q % PFP] = cur[offp(q + PFP)];
                if (q % NW == 0) {
                    const int i = q / NW;
                    if (sti + 1 < PST) ppiece(sti + 1, i);
                    else {                                                   // after the last W_p stage: the MLP's stage 0 (fc1 row block 0 and its biases)
                        if (i == 0) bias_piece_all(0);
                        if (i < F1 / NW) dma_piece(0, i);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (sti + 1 < PST) {                                              // the last stage's successor is awaited below, with the table
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
        // residual in accumulator layout, the sums of the LayerNorm in the order of the load-based prologue, x1 back to memory
        const float* xr = xin + px * K + 8 * h;                             // xin == x, or the block's input kept elsewhere (x then only receives)
        float* xw = x + (p0 + r) * K + 8 * h;
        float sum = 0.f, sq = 0.f;
#pragma unroll
        for (int q4 = 0; q4 < KS / 4; ++q4) {
            f32x4 raw[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q) { raw[q][0] = *(const f32x4*)(xr + 16 * (4 * q4 + q)); raw[q][1] = *(const f32x4*)(xr + 16 * (4 * q4 + q) + 4); }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = 4 * q4 + q, nb = s >> 1, u = s & 1;
                f32x4 lo, hi;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lo[j] = acc2[nb][8 * u + j] + raw[q][0][j]; hi[j] = acc2[nb][8 * u + 4 + j] + raw[q][1][j];
                    acc2[nb][8 * u + j] = lo[j]; acc2[nb][8 * u + 4 + j] = hi[j];
                }
                if (r < nvalid) { *(f32x4*)(xw + 16 * s) = lo; *(f32x4*)(xw + 16 * s + 4) = hi; }
            }
        }
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += acc2[s >> 1][8 * (s & 1) + j] + acc2[s >> 1][8 * (s & 1) + 4 + j];
        sum += __shfl_xor(sum, 32);
        const float inv_n = 1.0f / (float)ln_n;
        const float mu = sum * inv_n;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d0 = acc2[s >> 1][8 * (s & 1) + j] - mu, d1 = acc2[s >> 1][8 * (s & 1) + 4 + j] - mu; sq += d0 * d0 + d1 * d1; }
        sq += __shfl_xor(sq, 32);
        sq -= (float)(K - ln_n) * mu * mu;
        const float rstd = 1.0f / sqrtf(sq * inv_n + eps);
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                b[s][j] = (f16)((acc2[s >> 1][8 * (s & 1) + j] - mu) * rstd); b[s][4 + j] = (f16)((acc2[s >> 1][8 * (s & 1) + 4 + j] - mu) * rstd);
            }
    } else {
        // LayerNorm (the arithmetic of k_pw_wide / k_pw_kres_lds) -> fp16 B fragments of the wave's 32 tokens
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

#pragma unroll
    for (int nb = 0; nb < NB2; ++nb)                                         // fc2 accumulators start from the fc2 bias, as k_nres_wide does (same summation order)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc2[nb][8 * u + e] = b2[32 * nb + 16 * u + 8 * h + e];

    MLP_STAMP(1);
#pragma unroll
    for (int i = 0; i < 512 / (64 * NW); ++i) ((f32x4*)tab)[threadIdx.x + 64 * NW * i] = treg[i];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    MLP_STAMP(2);

    // Trip c: fc1 of chunk c (hidden channels 32c .. 32c+31 of the wave's tokens) into acc_n, while the VALU turns chunk c-1's fc1 result
    // (acc_o) into fp16 GELU values - the B fragments of fc2 k-steps 2(c-1), 2(c-1)+1 - and then those 2 x 8 fc2 MFMAs.  NB1 + 1 trips.
    f32x16 acc_o = zero16();
    auto trip = [&](int c, auto FC1c, auto FC2c) {
        constexpr bool fc1 = decltype(FC1c)::value && DBG != 3 && DBG != 5, has1 = decltype(FC1c)::value;
        constexpr bool has2 = decltype(FC2c)::value, fc2 = has2 && DBG != 2 && DBG != 5;
        const char* cur = smem + (c & 1) * (WBUF + BBUF);
        const half8* w1 = (const half8*)cur + lane;
        const half8* w2 = (const half8*)(cur + F1 * 1024) + lane;
        constexpr bool steady = decltype(FC1c)::value && decltype(FC2c)::value && DBG == 0;   // DMA pieces ride between the fc1 MFMAs
        MLP_STAMP(8 + 8 * c);
        if constexpr (!steady) { if (c + 1 <= NB1) stage(c + 1, (c + 1) & 1); }
        MLP_STAMP(8 + 8 * c + 1);
        f32x16 acc_n = zero16();
        if constexpr (has1) {
            const float* bl = (const float*)(cur + WBUF) + 8 * h;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x4 lo = *(const f32x4*)(bl + 16 * u), hi = *(const f32x4*)(bl + 16 * u + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc_n[8 * u + e] = lo[e]; acc_n[8 * u + 4 + e] = hi[e]; }
            }
        }
        half2 hp[8];                                                         // GELU(acc_o) as fp16 pairs: the B fragments of fc2 k-steps 2(c-1), 2(c-1)+1
        // GELU through the Phi table (as the fused GDFN kernel): gelu(a) = |a| Phi(|a|) + min(a, 0).  The look-up of pair j (index, fraction,
        // ds_read_b64 of (Phi, slope)) is issued one MFMA group before the arithmetic that uses it.
        float gfr[2]; float2 gte[2];
        auto gelu_lookup = [&](int j) {
            if constexpr (DBG == 1 || DBG == 5) return;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float tt = fminf(fabsf(acc_o[2 * j + e]) * ((float)PHI_NTAB / PHI_RANGE), (float)PHI_NTAB - 0.004f);
                gfr[e] = __builtin_amdgcn_fractf(tt);
                gte[e] = tab[(int)tt];
            }
        };
        auto gelu_finish = [&](int j) {
            float g[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float a = acc_o[2 * j + e];
                g[e] = (DBG == 1 || DBG == 5) ? a : fmaf(fabsf(a), fmaf(gfr[e], gte[e].y, gte[e].x), fminf(a, 0.f));
            }
            hp[j] = half2{(f16)g[0], (f16)g[1]};
            asm volatile("" : "+v"(hp[j]));                                  // the pair is computed HERE (in the MFMA shadow it was written into), not sunk below the loop
        };
        if constexpr (fc1) {
            constexpr int PF = 3;                                            // (register budget: 256 with b, acc2, acc_o, acc_n, the GELU pairs live)
            half8 a[PF];                                                     // order (k-step, plane): the hi / lo order of k_pw_wide
#pragma unroll
            for (int q = 0; q < PF; ++q) a[q] = w1[((q % NPL) * KS + q / NPL) * 64];
            if constexpr (steady) dma_bases();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KS; ++s) {                                   // per k-step: its MFMA(s), then in their shadow a DMA piece (even s) or a GELU pair (odd s)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    const int q = s * NPL + pl;
                    acc_n = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q % PF], b[s], acc_n, 0, 0, 0);
                    if (q + PF < F1) a[q % PF] = w1[(((q + PF) % NPL) * KS + (q + PF) / NPL) * 64];
                }
                // first half: GELU pair s looked up, pair s - 1 finished - the table reads happen while no DMA is in flight (the compiler
                // puts vmcnt(0) in front of every LDS read it cannot tell from a DMA target); second half: the next stage's DMA pieces
                if constexpr (has2) {
                    if (s >= 1 && s <= KS / 2) gelu_finish(s - 1);
                    if (s < KS / 2) gelu_lookup(s);
                }
                if constexpr (steady) {
                    if (s == KS / 2) bias_piece_all(c + 1);
                    if (s >= KS / 2) dma_piece(c + 1, s - KS / 2);
                }
                __builtin_amdgcn_sched_barrier(0);                           // keeps that interleaving
            }
        } else if constexpr (has2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { gelu_lookup(j); gelu_finish(j); }
        }
        half8 hf[2];
#pragma unroll
        for (int j = 0; j < 8; ++j) { hf[j >> 2][2 * (j & 3)] = hp[j][0]; hf[j >> 2][2 * (j & 3) + 1] = hp[j][1]; }
        MLP_STAMP(8 + 8 * c + 2);
        if constexpr (has2) {
            if constexpr (fc2) {
                half8 a[PF];                                                 // order (k-step, row block, plane) as k_nres_wide
                auto off2 = [](int q) { const int ks = q / (NPL * NB2), rem = q % (NPL * NB2), nb = rem / NPL, plane = rem % NPL; return (ks * NPL * NB2 + plane * NB2 + nb) * 64; };
#pragma unroll
                for (int q = 0; q < PF; ++q) a[q] = w2[off2(q)];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < F2; ++q) {
                    const int ks = q / (NPL * NB2), nb = (q % (NPL * NB2)) / NPL;
                    acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q % PF], hf[ks], acc2[nb], 0, 0, 0);
                    if (q + PF < F2) a[q % PF] = w2[off2(q + PF)];
                    __builtin_amdgcn_sched_barrier(0);                       // PF reads stay in flight
                }
            } else {
                if (hf[0][0] == (f16)12345.f && hf[1][7] == (f16)-4321.f) acc2[0][0] += 1.f;      // keeps fc1 / GELU alive in the measurement variants
            }
        }
        acc_o = acc_n;
        MLP_STAMP(8 + 8 * c + 3);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // no store in the loop: everything outstanding is the next stage
        MLP_STAMP(8 + 8 * c + 4);
        __builtin_amdgcn_s_barrier();
        MLP_STAMP(8 + 8 * c + 5);
    };
    if constexpr (DBG != 4) {
        trip(0, std::true_type{}, std::false_type{});
        for (int c = 1; c < NB1; ++c) trip(c, std::true_type{}, std::true_type{});
        trip(NB1, std::false_type{}, std::true_type{});
    }

    MLP_STAMP(3);
    // epilogue: x += acc2 + b2, two row blocks (64 channels = 256 bytes per token) at a time through this wave's LDS rows
    char* tr = smem + wave * 32 * RS;
    const int piece = lane & 15, row0 = lane >> 4;
    // the residual rows are requested two iterations ahead (the registers of b[], then of the accumulators already written out, hold them):
    // one memory latency for the whole epilogue instead of one per iteration
    f32x4 rs[NB2 / 2][8];
    auto rs_load = [&](int pr) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = row0 + 4 * i;
            rs[pr][i] = row < nvalid ? *(const f32x4*)(x + (p0 + row) * K + 64 * pr + 4 * piece) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    rs_load(0);
    rs_load(1);
#pragma unroll
    for (int pr = 0; pr < NB2 / 2; ++pr) {
        const int n0 = 64 * pr + 4 * piece;
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
            v += rs[pr][i];                                                  // residual last, as in the two-kernel path
            if (row < nvalid) *(f32x4*)(x + (p0 + row) * K + n0) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (pr + 2 < NB2 / 2) rs_load(pr + 2);
    }
    MLP_STAMP(4);
    MLP_RSTAMP(6);
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

struct MlpArgs {
    float* x; const void* W1; const float* b1; const void* W2; const float* b2;
    const void* attn; const void* Wp; const float* bp; const float* xin;    // attention projection in front (PROJ), or all NULL; xin: where PROJ reads x
    int NB1; long npx; float eps; int ln_n;
};

template <bool LO, int NW, bool PROJ, int DBG>
static hipError_t swin_mlp_launch(hipStream_t st, const MlpArgs& a) {
    const float2* phi = nullptr;
    if (hipError_t e = phi_table(&phi); e != hipSuccess) return e;
    constexpr int NPL = LO ? 2 : 1;
    const dim3 grid((unsigned)((a.npx + 32 * NW - 1) / (32 * NW))), block(64 * NW);
    // two stages (fc1 row block + fc2 slab + biases) or, if larger, the epilogue's transposition rows, which reuse the same LDS (the Phi table is a static 8 KB on top)
    const size_t smem = std::max((size_t)2 * ((size_t)(NPL * MLP_KS + NPL * 2 * MLP_NB2) * 1024 + 256), (size_t)NW * 32 * (64 * 4 + 16));
    static std::atomic<uint64_t> done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)k_swin_mlp<LO, NW, PROJ, DBG>, smem, done); e != hipSuccess) return e;
    hipLaunchKernelGGL((k_swin_mlp<LO, NW, PROJ, DBG>), grid, block, smem, st, a.x, (const f16*)a.W1, a.b1, (const f16*)a.W2, a.b2, phi, (const f16*)a.attn,
                       (const f16*)a.Wp, a.bp, a.xin, a.NB1, a.npx, a.eps, a.ln_n);
    return hipGetLastError();
}

hipError_t launch_swin_mlp(hipStream_t st, float* x, const void* W1, const float* b1, const void* W2, const float* b2, int Cp, int hidp, long npx,
                           float eps, int ln_n, int planes, const void* attn, const void* Wp, const float* bp, const float* xin) {
    if (!swin_mlp_supported(Cp, hidp, npx) || !b1 || !b2) return hipErrorInvalidValue;
    const bool proj = attn != nullptr;
    if (proj && (!Wp || !bp)) return hipErrorInvalidValue;
    if (xin != nullptr && !proj) return hipErrorInvalidValue;              // only the projection step reads x before anything is written
    const MlpArgs a{x, W1, b1, W2, b2, attn, Wp, bp, xin ? xin : x, hidp / 32, npx, eps, ln_n};
    // plain fp16 weights: 64.5 KB of LDS per workgroup of FOUR waves, two workgroups per CU (their load / MFMA / store phases overlap);
    // hi + lo planes: 128.5 KB, one workgroup of eight waves
    if (planes == 1) {
#ifdef IRA_KERNEL_DBG                      // measurement variants (wrong results): only in the `make dbg` library, never in the product one
        static const int dbg = env_int_m("IRA_SWIN_MLP_DBG", 0);
        if (!proj) switch (dbg) {
            case 1: return swin_mlp_launch<false, 4, false, 1>(st, a);
            case 2: return swin_mlp_launch<false, 4, false, 2>(st, a);
            case 3: return swin_mlp_launch<false, 4, false, 3>(st, a);
            case 4: return swin_mlp_launch<false, 4, false, 4>(st, a);
            case 5: return swin_mlp_launch<false, 4, false, 5>(st, a);
            default: break;
        }
#endif
        return proj ? swin_mlp_launch<false, 4, true, 0>(st, a) : swin_mlp_launch<false, 4, false, 0>(st, a);
    }
    return proj ? swin_mlp_launch<true, 8, true, 0>(st, a) : swin_mlp_launch<true, 8, false, 0>(st, a);
}

}  // namespace ira
