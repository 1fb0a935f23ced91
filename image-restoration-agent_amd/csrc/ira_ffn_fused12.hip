// Fused GDFN block, 12-wave variant (see ira_ffn_fused.hip for the algorithm and the numerics contract).
//
// Why a second variant: s_memtime stamps of the 8-wave kernel showed the consumers (depthwise conv + GELU on the
// VALU) bound every chunk at 9.8k cycles while the producers waited 4k of them, because ONE VALU-heavy wave per
// SIMD can only issue an instruction every 4 cycles.  Here every SIMD runs one producer and TWO consumer waves
// (768 threads, waves i, i+4, i+8 share SIMD i), each consumer owning one output row.  Twelve waves cap the
// register budget at 168 per wave, so the producers no longer hold W_in fragments in registers: each chunk's
// fragments ({x1,x2} x {hi,lo} x K/16, 24 KB at C = 96) are staged in an LDS double buffer one chunk ahead by
// the four producers and read back with ds_read_b128 (LDS use: hp 2 x 50.7 KB + W_in 2 x 24 KB + dw 4.6 KB = 155 KB).
#include "ira_kernels.h"

#include <cstdlib>

namespace ira {

typedef _Float16 f16;
typedef f16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define IRA_DEV __device__ __forceinline__

namespace ffn12 {

IRA_DEV void lds_barrier() {      // LDS hand-offs only need lgkmcnt(0) + s_barrier (see ira_ffn_fused.hip)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

IRA_DEV long xcd_swz(long bid, long nblk) {
    const long q = nblk >> 3, rem = nblk & 7, x = bid & 7, j = bid >> 3;
    return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + j;
}

IRA_DEV float gelu_erf_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    const float e = p * t * __expf(-z * z);
    return 0.5f * x * ((x >= 0.f) ? (2.0f - e) : e);
}

IRA_DEV void fma_mix_lo(float& acc, unsigned packed, float w) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(packed), "v"(w));
}
IRA_DEV void fma_mix_hi(float& acc, unsigned packed, float w) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(packed), "v"(w));
}
IRA_DEV void dw_tap(float (&acc)[8], const u32x4 v, const f32x4 wl, const f32x4 wh) {
    fma_mix_lo(acc[0], v[0], wl[0]); fma_mix_hi(acc[1], v[0], wl[1]);
    fma_mix_lo(acc[2], v[1], wl[2]); fma_mix_hi(acc[3], v[1], wl[3]);
    fma_mix_lo(acc[4], v[2], wh[0]); fma_mix_hi(acc[5], v[2], wh[1]);
    fma_mix_lo(acc[6], v[3], wh[2]); fma_mix_hi(acc[7], v[3], wh[3]);
}

constexpr int TH = 8, TW = 32, PH = TH + 2, PW = TW + 2, PHW = PH * PW;   // 340 halo pixels
constexpr int NG1 = (PHW + 31) / 32;                                        // 11 pixel groups
constexpr int HPS = 72;                                                     // hp row stride in halfs (64 + 8 pad)
constexpr int HP_ROWS = NG1 * 32;
constexpr size_t HP_BYTES = (size_t)HP_ROWS * HPS * 2;                      // 50 688 B
constexpr size_t DW_BYTES = 9 * 64 * 4;
constexpr size_t lds_bytes(int KS) { return 2 * HP_BYTES + 2 * DW_BYTES + 2 * (size_t)(4 * KS) * 1024; }

}  // namespace ffn12
using namespace ffn12;

template <int C, int LN, bool BIN, bool CB>
__global__ __launch_bounds__(768) void k_ffn_fused12(const float* __restrict__ x, float* __restrict__ xo,
                                                     const f16* __restrict__ Win, const float* __restrict__ bin,
                                                     const float* __restrict__ dww, const float* __restrict__ dwb,
                                                     const f16* __restrict__ Wout, const float* __restrict__ bout,
                                                     int T, int H, int W, int Hp) {
    constexpr int KS = C / 16;
    constexpr int NBC = (C + 31) / 32;
    constexpr int NF = 4 * KS;                    // W_in fragments per chunk: [x1 hi KS][x1 lo KS][x2 hi KS][x2 lo KS]
    constexpr int FPP = NF / 4;                   // fragments each producer stages
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f16* hpbuf = (f16*)smem;                                          // [2][HP_ROWS][HPS]
    float* dwbuf = (float*)(smem + 2 * HP_BYTES);                     // [2][9][64]
    half8* winbuf = (half8*)(smem + 2 * HP_BYTES + 2 * DW_BYTES);     // [2][NF][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int npy = (H + TH - 1) / TH, npx = (W + TW - 1) / TW;
    const long pid = xcd_swz(blockIdx.x, gridDim.x);
    const int t = (int)(pid / ((long)npy * npx));
    const int py = (int)((pid / npx) % npy), pxx = (int)(pid % npx);
    const int y0 = py * TH, x0 = pxx * TW;
    const float* xt = x + (long)t * H * W * C;
    const int NBH = Hp >> 5;

    // ---- prologue: waves 0..10 LayerNorm one halo group each -> fp16 B-fragments staged in LDS (hp region)
    half8* xnl = (half8*)smem;                    // [NG1][KS][64]
    if (wave < NG1) {
        const int g = wave;
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
            xnl[(g * KS + s) * 64 + lane] = o;
        }
    }

    if (wave < 4) {
        // =========================== producers ===========================
        constexpr int MAXG = 3;
        half8 b[MAXG][KS];
        bool inimg[MAXG];
        const long win_lo = (long)(2 * NBH) * KS * 512;
        const f16* winl = Win + lane * 8;
        const int ngroups = (wave + 8 < NG1) ? 3 : 2;

        // fragment f of chunk j in LDS order: f / KS selects {x1 hi, x1 lo, x2 hi, x2 lo}
        auto gfrag = [&](int j, int f) -> const half8* {
            const int part = f / KS, s = f - part * KS;
            const int nbrow = (part < 2) ? j : NBH + j;
            return (const half8*)(winl + ((part & 1) ? win_lo : 0) + ((long)nbrow * KS + s) * 512);
        };
        half8 stg[FPP];
        float dwreg[3];
        auto load_stage = [&](int j) {
#pragma unroll
            for (int i = 0; i < FPP; ++i) stg[i] = *gfrag(j, wave + 4 * i);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int i = tid + 256 * k;              // tid < 256: the four producer waves
                const int ii = (i < 576) ? i : 575;
                const int tap = ii >> 6, cc = ii & 63;
                const int ch = (cc < 32) ? (32 * j + cc) : (Hp + 32 * j + cc - 32);
                dwreg[k] = dww[(long)tap * 2 * Hp + ch];
            }
        };
        // W_in[j+1] is written at the END of producer iteration j (only producers read it, two chunks later);
        // dw[j] at the START of iteration j: the consumers are then still reading dw[j-1] from the other buffer
        auto store_win = [&](int j) {
            half8* wb = winbuf + (j & 1) * NF * 64;
#pragma unroll
            for (int i = 0; i < FPP; ++i) wb[(wave + 4 * i) * 64 + lane] = stg[i];
        };
        auto store_dw = [&](int j) {
            float* dwl = dwbuf + (j & 1) * 576;
#pragma unroll
            for (int k = 0; k < 3; ++k) if (tid + 256 * k < 576) dwl[tid + 256 * k] = dwreg[k];
        };
        load_stage(0);
        lds_barrier();                            // (A) every group's B-fragments are in LDS
#pragma unroll
        for (int gi = 0; gi < MAXG; ++gi) {
            const int g = wave + 4 * gi;
            const int q = 32 * g + r;
            const int hy = q / PW, hx = q - hy * PW;
            const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
            inimg[gi] = (g < NG1) && (q < PHW) && yy >= 0 && yy < H && xx >= 0 && xx < W;
#pragma unroll
            for (int s = 0; s < KS; ++s) b[gi][s] = xnl[((g < NG1 ? g : 0) * KS + s) * 64 + lane];
        }
        store_win(0);                             // the W_in region does not alias the prologue staging
        lds_barrier();                            // (B) fragments in registers: the staging region becomes the hp buffers;
                                                  //     chunk 0's W_in / dw weights are visible to all producers

        for (int j = 0; j <= NBH; ++j) {
            if (j < NBH) {
                store_dw(j);
                if (j + 1 < NBH) load_stage(j + 1);               // global -> registers, lands during this chunk
                f16* hp = hpbuf + (size_t)(j & 1) * HP_ROWS * HPS;
                const half8* wb = winbuf + (j & 1) * NF * 64 + lane;
#pragma unroll
                for (int xsel = 0; xsel < 2; ++xsel) {
#pragma unroll
                    for (int gi = 0; gi < MAXG; ++gi) {
                        if (gi >= ngroups) break;
                        f32x16 acc;
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[8 * u + e] = BIN ? bin[xsel * Hp + 32 * j + 16 * u + 8 * h + e] : 0.f;
#pragma unroll
                        for (int s = 0; s < KS; ++s) {
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[((2 * xsel) * KS + s) * 64], b[gi][s], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[((2 * xsel + 1) * KS + s) * 64], b[gi][s], acc, 0, 0, 0);
                        }
                        const int q = 32 * (wave + 4 * gi) + r;
                        f16* dst = hp + (long)q * HPS + 32 * xsel + 8 * h;
                        const bool in = inimg[gi];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            half8 o;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = in ? (f16)acc[8 * u + e] : (f16)0.f;   // zero padding of the dwconv
                            *(half8*)(dst + 16 * u) = o;
                        }
                    }
                }
                if (j + 1 < NBH) store_win(j + 1);                // buffer (j+1)&1: last read by producers one chunk ago
            }
            lds_barrier();        // hp[j] (+ W_in / dw of chunk j+1) visible; consumers have finished with hp[j-1]
        }
        return;
    }

    // =========================== consumers (waves 4..11): one output row each ===========================
    const int cw = wave - 4;                      // row of the patch
    const int KTO = Hp >> 4;
    const long wout_lo = (long)NBC * KTO * 512;
    const f16* woutl = Wout + lane * 8;
    f32x16 acc[NBC];                              // lane (r, h) owns channels 32nb + 16u + 8h + e of pixel (row cw, x0 + r)
#pragma unroll
    for (int nb = 0; nb < NBC; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nb][e] = 0.f;

    lds_barrier();                                // (A)
    lds_barrier();                                // (B)
    lds_barrier();                                // hp[0] ready
    for (int j = 0; j < NBH; ++j) {
        const f16* hp = hpbuf + (size_t)(j & 1) * HP_ROWS * HPS;
        const float* dwl = dwbuf + (j & 1) * 576;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int cl0 = 16 * s + 8 * h;
            half8 woh[NBC], wol[NBC];
#pragma unroll
            for (int nb = 0; nb < NBC; ++nb) {
                woh[nb] = *(const half8*)(woutl + ((long)nb * KTO + 2 * j + s) * 512);
                wol[nb] = *(const half8*)(woutl + wout_lo + ((long)nb * KTO + 2 * j + s) * 512);
            }
            float a1[8], a2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a1[e] = CB ? dwb[32 * j + cl0 + e] : 0.f;
                a2[e] = CB ? dwb[Hp + 32 * j + cl0 + e] : 0.f;
            }
            struct TapIn { f32x4 w1l, w1h, w2l, w2h; u32x4 v1, v2; };
            auto load_tap = [&](int tap) {
                TapIn ti;
                ti.w1l = *(const f32x4*)(dwl + tap * 64 + cl0);      ti.w1h = *(const f32x4*)(dwl + tap * 64 + cl0 + 4);
                ti.w2l = *(const f32x4*)(dwl + tap * 64 + 32 + cl0); ti.w2h = *(const f32x4*)(dwl + tap * 64 + 32 + cl0 + 4);
                const int q = (cw + tap / 3) * PW + r + tap % 3;
                ti.v1 = *(const u32x4*)(hp + (long)q * HPS + cl0);
                ti.v2 = *(const u32x4*)(hp + (long)q * HPS + 32 + cl0);
                return ti;
            };
            TapIn cur = load_tap(0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                TapIn nxt = cur;
                if (tap < 8) nxt = load_tap(tap + 1);
                __builtin_amdgcn_sched_barrier(0);
                dw_tap(a1, cur.v1, cur.w1l, cur.w1h);
                dw_tap(a2, cur.v2, cur.w2l, cur.w2h);
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            }
            half8 fA;
#pragma unroll
            for (int e = 0; e < 8; ++e) fA[e] = (f16)(gelu_erf_f(a1[e]) * a2[e]);
#pragma unroll
            for (int nb = 0; nb < NBC; ++nb) {
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(woh[nb], fA, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wol[nb], fA, acc[nb], 0, 0, 0);
            }
        }
        lds_barrier();            // done with hp[j] / dw[j]; hp[j+1] / dw[j+1] are ready
    }

    // ---- epilogue: x_out = x + acc (+ bias), fp32
    const int oy = y0 + cw, ox = x0 + r;
    if (oy >= H || ox >= W) return;
    const long po = ((long)t * H + oy) * W + ox;
    f32x4 xr[NBC][2][2];
#pragma unroll
    for (int nb = 0; nb < NBC; ++nb)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int n0 = min(32 * nb + 16 * u + 8 * h, C - 8);
            xr[nb][u][0] = *(const f32x4*)(x + po * C + n0);
            xr[nb][u][1] = *(const f32x4*)(x + po * C + n0 + 4);
        }
#pragma unroll
    for (int nb = 0; nb < NBC; ++nb)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int n0 = 32 * nb + 16 * u + 8 * h;
            if (n0 + 8 <= C) {
                f32x4 lo = xr[nb][u][0], hi = xr[nb][u][1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] += acc[nb][8 * u + e] + (CB ? bout[n0 + e] : 0.f);
                    hi[e] += acc[nb][8 * u + 4 + e] + (CB ? bout[n0 + 4 + e] : 0.f);
                }
                *(f32x4*)(xo + po * C + n0) = lo;
                *(f32x4*)(xo + po * C + n0 + 4) = hi;
            }
        }
}

template <int C, int LN, bool BIN, bool CB>
static hipError_t ffn12_launch2(hipStream_t st, const float* x, float* xo, const f16* Win, const float* bin, const float* dww,
                                const float* dwb, const f16* Wout, const float* bout, int T, int H, int W, int Hp) {
    const long nblk = (long)T * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    constexpr size_t smem = lds_bytes(C / 16);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ffn_fused12<C, LN, BIN, CB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((k_ffn_fused12<C, LN, BIN, CB>), dim3((unsigned)nblk), dim3(768), smem, st, x, xo, Win, bin, dww, dwb, Wout, bout, T, H, W, Hp);
    return hipGetLastError();
}

template <int C>
static hipError_t ffn12_launch(hipStream_t st, const float* x, float* xo, int ln_mode, const f16* Win, const float* bin,
                               const float* dww, const float* dwb, const f16* Wout, const float* bout, int T, int H, int W, int Hp) {
    const bool cb = dwb != nullptr && bout != nullptr;
    if ((dwb != nullptr) != (bout != nullptr)) return hipErrorInvalidValue;
#define IRA_FFN(LN_, BIN_, CB_) return ffn12_launch2<C, LN_, BIN_, CB_>(st, x, xo, Win, bin, dww, dwb, Wout, bout, T, H, W, Hp)
    if (ln_mode == 1) {
        if (bin) { if (cb) IRA_FFN(1, true, true); else IRA_FFN(1, true, false); }
        else { if (cb) IRA_FFN(1, false, true); else IRA_FFN(1, false, false); }
    } else {
        if (bin) { if (cb) IRA_FFN(2, true, true); else IRA_FFN(2, true, false); }
        else { if (cb) IRA_FFN(2, false, true); else IRA_FFN(2, false, false); }
    }
#undef IRA_FFN
}

hipError_t launch_ffn_fused12(hipStream_t st, const float* x, float* xo, int ln_mode, int C, const void* Win, const float* bin,
                              const float* dww, const float* dwb, const void* Wout, const float* bout, int T, int H, int W,
                              int Hp) {
    if (x == xo || (ln_mode != 1 && ln_mode != 2) || !(C == 48 || C == 96) || Hp % 32 != 0) return hipErrorInvalidValue;
    if (C == 48) return ffn12_launch<48>(st, x, xo, ln_mode, (const f16*)Win, bin, dww, dwb, (const f16*)Wout, bout, T, H, W, Hp);
    return ffn12_launch<96>(st, x, xo, ln_mode, (const f16*)Win, bin, dww, dwb, (const f16*)Wout, bout, T, H, W, Hp);
}

}  // namespace ira
