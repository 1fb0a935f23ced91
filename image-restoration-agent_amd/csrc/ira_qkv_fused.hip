// Fused attention front end (restormer_arch.py:114 after :146's norm1), one kernel:
//     qkv = dwconv3x3( conv1x1( LN(x) ) )            q,k -> fp16 pixel-minor [T][H][ceil(W/8)][2C][8], v -> fp16 NHWC
// The 3C-channel pre-depthwise tensor never leaves the CU (it was written and re-read by two kernels before:
// 2.1 KB/px -> 1.1 KB/px of HBM traffic at C = 96), at the price of recomputing the 1x1 on the 1-pixel halo.
//
// Same skeleton as ira_ffn_fused.hip: one workgroup = 8 waves per 8 x 32 output patch (10 x 34 halo = 11 groups of
// 32 pixels); producers (waves 0-3) keep the LayerNorm-ed fp16 B-fragments of their halo groups in registers and
// compute one 32-channel chunk of the 1x1 per step into an LDS double buffer, one chunk ahead, with the next
// chunk's weight fragments prefetched; consumers (waves 4-7) run the depthwise 3x3 out of LDS with
// v_fma_mix_f32 on 4-pixel x 8-channel register blocks (256 blocks per chunk = one per consumer lane) and
// store q,k / v directly in the layouts the Gram and attention-apply kernels read.
#include "ira_kernels.h"
#include "ira_device.h"

namespace ira {

namespace {

constexpr int TH = 8, TW = 32, PH = TH + 2, PW = TW + 2, PHW = PH * PW;   // 340 halo pixels
constexpr int NG1 = (PHW + 31) / 32;                                        // 11 pixel groups
constexpr int HPS = 40;                                                     // row stride in halfs (32 + 8 pad): conflict-free
constexpr int HP_ROWS = NG1 * 32;
constexpr size_t HP_BYTES = (size_t)HP_ROWS * HPS * 2;                      // 28 160 B
constexpr size_t DW_BYTES = 9 * 32 * 4;
constexpr size_t QKV_LDS_MAIN = 2 * HP_BYTES + 2 * DW_BYTES;                // 58 624 B
constexpr size_t QKV_LDS = (QKV_LDS_MAIN > (size_t)NG1 * 6 * 1024 ? QKV_LDS_MAIN : (size_t)NG1 * 6 * 1024) + 1024;   // >= 66 KB prologue staging (C = 96)

}  // namespace

// BIN: the 1x1 has a bias (LayerNorm bias folded and/or conv bias); CB: depthwise bias present (cfg.bias).
template <int C, int LN, bool BIN, bool CB>
__global__ __launch_bounds__(512) void k_qkv_fused(const float* __restrict__ x, const f16* __restrict__ Wq,
                                                   const float* __restrict__ bq, const float* __restrict__ dww,
                                                   const float* __restrict__ dwb, f16* __restrict__ qk,
                                                   f16* __restrict__ vout, int T, int H, int W) {
    constexpr int KS = C / 16;
    constexpr int NQ = 3 * C;                     // real qkv channels
    constexpr int NCH = (NQ + 31) / 32;           // 32-channel chunks (the last one may be half padding)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f16* hpbuf = (f16*)smem;                                  // [2][HP_ROWS][HPS]
    float* dwbuf = (float*)(smem + 2 * HP_BYTES);             // [2][9][32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int npy = (H + TH - 1) / TH, npx = (W + TW - 1) / TW;
    const long pid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int t = (int)(pid / ((long)npy * npx));
    const int py = (int)((pid / npx) % npy), pxx = (int)(pid % npx);
    const int y0 = py * TH, x0 = pxx * TW;
    const float* xt = x + (long)t * H * W * C;

    // ---- prologue, all 8 waves: LayerNorm of the halo patch -> fp16 B-fragments staged in LDS (region reused by the
    // hp buffers afterwards): one HBM round trip for the workgroup instead of three sequential ones in the producers.
    half8* xnl = (half8*)smem;                    // [NG1][KS][64]: 66 KB at C = 96 (needs QKV_LDS >= that)
    {
        // a wave owns halo groups `wave` and `wave + 8` (the second only for waves 0-2): the loads of BOTH are issued
        // before either LayerNorm, so the prologue costs one HBM round trip, not two
        constexpr int NPG = 2;
        float xv[NPG][KS][8];
        bool have[NPG];
#pragma unroll
        for (int gi = 0; gi < NPG; ++gi) {
            const int g = wave + 8 * gi;
            have[gi] = g < NG1;
            const int q = 32 * (have[gi] ? g : 0) + r;
            const int hy = q / PW, hx = q - hy * PW;
            const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
            const bool in = (q < PHW) && yy >= 0 && yy < H && xx >= 0 && xx < W;
            const float* src = xt + ((long)(in ? yy : 0) * W + (in ? xx : 0)) * C + 8 * h;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const f32x4 lo = *(const f32x4*)(src + 16 * s), hi = *(const f32x4*)(src + 16 * s + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { xv[gi][s][j] = lo[j]; xv[gi][s][4 + j] = hi[j]; }
            }
        }
#pragma unroll
        for (int gi = 0; gi < NPG; ++gi) {
            float sum = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += xv[gi][s][j];
            sum += __shfl_xor(sum, 32);
            const float mu = sum * (1.0f / C);
            float sq = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = xv[gi][s][j] - mu; sq += d * d; }
            sq += __shfl_xor(sq, 32);
            const float rstd = 1.0f / sqrtf(sq * (1.0f / C) + 1e-5f);
            const float sub = (LN == 2) ? mu : 0.f;
            if (have[gi]) {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    half8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (f16)((xv[gi][s][j] - sub) * rstd);
                    xnl[((wave + 8 * gi) * KS + s) * 64 + lane] = o;
                }
            }
        }
    }

    if (wave < 4) {
        // =========================== producers ===========================
        constexpr int MAXG = 3;
        half8 b[MAXG][KS];
        bool inimg[MAXG];
        const long w_lo = (long)NCH * KS * 512;
        const f16* wl_ = Wq + lane * 8;
        const int ngroups = (wave + 8 < NG1) ? 3 : 2;

        half8 f_h[KS], f_l[KS], n_h[KS], n_l[KS];     // this chunk's / the next chunk's weight fragments (hi, lo)
        auto load_frags = [&](half8 (&fh)[KS], half8 (&fl)[KS], int nbrow) {
            const f16* w = wl_ + (long)nbrow * KS * 512;
#pragma unroll
            for (int s = 0; s < KS; ++s) { fh[s] = *(const half8*)(w + s * 512); fl[s] = *(const half8*)(w + w_lo + s * 512); }
        };
        float dwreg[2];
        auto load_dw = [&](int j) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int i = tid + 256 * k;              // tid < 256: the four producer waves; 288 entries
                const int ii = (i < 288) ? i : 287;
                const int tap = ii >> 5, cc = ii & 31;
                const int ch = 32 * j + cc;
                const float wv = dww[(long)tap * NQ + (ch < NQ ? ch : NQ - 1)];      // unconditional load, clamped address
                dwreg[k] = (ch < NQ) ? wv : 0.f;
            }
        };
        load_frags(f_h, f_l, 0);                  // requested before the barrier: their latency hides behind it
        load_dw(0);
        lds_barrier();                          // (A) every group's B-fragments are in LDS
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
        lds_barrier();                          // (B) fragments are in registers: the LDS region becomes the hp buffers

        // chunk j computes from (ch, cl) while the fragments of chunk j+1 stream into (nh, nl): the two register sets
        // ping-pong (loop unrolled by two, no copies), so a prefetch has a whole chunk period to land
        auto do_chunk = [&](int j, const half8 (&ch)[KS], const half8 (&cl)[KS], half8 (&nh)[KS], half8 (&nl)[KS]) {
            f16* hp = hpbuf + (size_t)(j & 1) * HP_ROWS * HPS;
            float* dwl = dwbuf + (j & 1) * 288;
#pragma unroll
            for (int k = 0; k < 2; ++k) if (tid + 256 * k < 288) dwl[tid + 256 * k] = dwreg[k];
            if (j + 1 < NCH) { load_dw(j + 1); load_frags(nh, nl, j + 1); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gi = 0; gi < MAXG; ++gi) {
                if (gi >= ngroups) break;
                f32x16 acc;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[8 * u + e] = BIN ? bq[32 * j + 16 * u + 8 * h + e] : 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch[s], b[gi][s], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl[s], b[gi][s], acc, 0, 0, 0);
                }
                const int q = 32 * (wave + 4 * gi) + r;
                f16* dst = hp + (long)q * HPS + 8 * h;
                const bool in = inimg[gi];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    half8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = in ? (f16)acc[8 * u + e] : (f16)0.f;   // zero padding of the dwconv
                    *(half8*)(dst + 16 * u) = o;
                }
            }
        };
        for (int j = 0; j <= NCH; j += 2) {
            if (j < NCH) do_chunk(j, f_h, f_l, n_h, n_l);
            lds_barrier();
            if (j + 1 <= NCH) {
                if (j + 1 < NCH) do_chunk(j + 1, n_h, n_l, f_h, f_l);
                lds_barrier();
            }
        }
        return;
    }

    // =========================== consumers ===========================
    // lane -> (channel octet o of the chunk, 4-pixel block blk of the row, row of the patch)
    const int o = lane & 3, blk = (lane >> 2) & 7, prow = 2 * (wave - 4) + (lane >> 5);
    const int oy = y0 + prow, ox = x0 + 4 * blk;
    const int WB = (W + 7) >> 3;
    lds_barrier();                              // (A)
    lds_barrier();                              // (B)
    lds_barrier();                              // chunk 0 ready
    for (int j = 0; j < NCH; ++j) {
        const f16* hp = hpbuf + (size_t)(j & 1) * HP_ROWS * HPS;
        const float* dwl = dwbuf + (j & 1) * 288;
        const int n0 = 32 * j + 8 * o;            // first qkv channel of this lane's octet
        float acc[4][8];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[p][e] = (CB && n0 < NQ) ? dwb[n0 + e] : 0.f;
        // all 36 LDS reads of the chunk are issued before the first FMA: with one consumer wave per SIMD nothing else
        // hides LDS latency (18 exposed round trips per chunk otherwise: 4.1k -> cycles per chunk measured)
        u32x4 v[3][6];
        f32x4 wl[9], wh[9];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int i = 0; i < 6; ++i) v[dy][i] = *(const u32x4*)(hp + (long)((prow + dy) * PW + 4 * blk + i) * HPS + 8 * o);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            wl[tap] = *(const f32x4*)(dwl + tap * 32 + 8 * o);
            wh[tap] = *(const f32x4*)(dwl + tap * 32 + 8 * o + 4);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int p = 0; p < 4; ++p) dw_tap(acc[p], v[dy][p + dx], wl[dy * 3 + dx], wh[dy * 3 + dx]);
        // q,k rows are padded to WB*8 pixels and the padding must read as zero for the Gram: store every 4-pixel
        // block below WB*8 (patches cover x < ceil(W/32)*32 >= WB*8), zeros for x >= W
        if (oy < H && n0 < NQ && (n0 < 2 * C ? ox < WB * 8 : ox < W)) {
            if (n0 < 2 * C) {
                // q / k: pixel-minor, this lane owns 4 pixels (half of an 8-pixel vector) of 8 channels
                f16* dst = qk + ((((long)(t * H + oy) * WB + (ox >> 3)) * (2 * C)) + n0) * 8 + (ox & 7);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    half4 q4;
#pragma unroll
                    for (int p = 0; p < 4; ++p) q4[p] = (ox + p < W) ? (f16)acc[p][e] : (f16)0.f;
                    *(half4*)(dst + e * 8) = q4;
                }
            } else {
                f16* dst = vout + ((long)(t * H + oy) * W + ox) * C + (n0 - 2 * C);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    if (ox + p < W) {
                        half8 o8;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o8[e] = (f16)acc[p][e];
                        *(half8*)(dst + (long)p * C) = o8;
                    }
                }
            }
        }
        lds_barrier();
    }
}

bool qkv_fused_supported(int C) { return C == 48 || C == 96; }

template <int C, int LN, bool BIN, bool CB>
static hipError_t qkv_launch2(hipStream_t st, const float* x, const f16* Wq, const float* bq, const float* dww,
                              const float* dwb, f16* qk, f16* v, int T, int H, int W) {
    const long nblk = (long)T * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)k_qkv_fused<C, LN, BIN, CB>, QKV_LDS, attr_done); e != hipSuccess) return e;
    hipLaunchKernelGGL((k_qkv_fused<C, LN, BIN, CB>), dim3((unsigned)nblk), dim3(512), QKV_LDS, st, x, Wq, bq, dww, dwb, qk, v, T, H, W);
    return hipGetLastError();
}

template <int C>
static hipError_t qkv_launch(hipStream_t st, const float* x, int ln_mode, const f16* Wq, const float* bq, const float* dww,
                             const float* dwb, f16* qk, f16* v, int T, int H, int W) {
#define IRA_QKV(LN_, BIN_, CB_) return qkv_launch2<C, LN_, BIN_, CB_>(st, x, Wq, bq, dww, dwb, qk, v, T, H, W)
    if (ln_mode == 1) {
        if (bq) { if (dwb) IRA_QKV(1, true, true); else IRA_QKV(1, true, false); }
        else { if (dwb) IRA_QKV(1, false, true); else IRA_QKV(1, false, false); }
    } else {
        if (bq) { if (dwb) IRA_QKV(2, true, true); else IRA_QKV(2, true, false); }
        else { if (dwb) IRA_QKV(2, false, true); else IRA_QKV(2, false, false); }
    }
#undef IRA_QKV
}

hipError_t launch_qkv_fused(hipStream_t st, const float* x, int ln_mode, int C, const void* Wq, const float* bq,
                            const float* dww, const float* dwb, void* qk, void* v, int T, int H, int W) {
    if ((ln_mode != 1 && ln_mode != 2) || !qkv_fused_supported(C)) return hipErrorInvalidValue;
    // rows of the pixel-minor q,k tensor are padded to 8 pixels; the padding must read as zero for the Gram:
    // patches cover x < ceil(W/32)*32 >= WB*8, and out-of-image pixels are stored as zeros above
    if (C == 48) return qkv_launch<48>(st, x, ln_mode, (const f16*)Wq, bq, dww, dwb, (f16*)qk, (f16*)v, T, H, W);
    return qkv_launch<96>(st, x, ln_mode, (const f16*)Wq, bq, dww, dwb, (f16*)qk, (f16*)v, T, H, W);
}

}  // namespace ira
