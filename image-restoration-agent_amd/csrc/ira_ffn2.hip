// Fused attention-apply + GDFN block (restormer_arch.py:127-131 applied as a per-tile matrix, :146-150, :88-93), one kernel:
//     x1    = x + M_t . v (+ b_o)                                  M_t = W_o . blockdiag(softmax): the attention of tile t as a 1x1 conv on v
//     x_out = x1 + project_out( gelu(dw3x3(project_in(LN(x1)))[:Hf]) * dw3x3(project_in(LN(x1)))[Hf:] )
// (v == nullptr: x1 = x, the plain GDFN block.)  Neither x1, nor the 2*Hf-channel hidden tensor, nor the gated tensor ever leaves
// the CU: per block and pixel the kernel reads x (fp32) and v (fp16) with a 1-pixel halo and writes x_out (fp32).
//
// Second generation of ira_ffn_fused.hip.  What the first one measured (s_memtime stamps): a wave-specialised
// workgroup whose consumer wave is ALONE with its VALU work on its SIMD issues one vector instruction per 4 cycles (half of what the
// SIMD can take), reads every depthwise weight from LDS (4 of its 8 ds_read_b128 per tap) and spends as long in the erf as in the
// convolution.  Here:
//   * one workgroup = 8 waves = an 8 x 32 output patch (10 x 34 halo); a CHUNK is one k-step of project_out: 16 channels of x1's
//     half + the matching 16 of the gate half = ONE 32-row MFMA block of W_in (its rows are picked out of the [x1 | x2] layout by
//     the per-lane fragment address: no repacking);
//   * MFMA work is specialised as before - waves 0-3 ("P") compute the chunk's hidden values hp = W_in . LN(x1) for the 340 halo
//     pixels into an LDS double buffer, waves 4-7 ("C") own two output rows each and accumulate W_out . g - but the VECTOR work
//     (depthwise 3x3, GELU, gate) is shared by ALL 8 waves in units of (row pair, channel octet): 8 units per chunk, one per wave,
//     so every SIMD has two waves issuing VALU; the gated values travel to the C waves as ready-made MFMA B fragments through a
//     16 KB LDS double buffer;
//   * a unit's lanes are 64 PIXELS of one channel octet, so its 144 depthwise weights are wave-uniform: they come through the
//     scalar cache into SGPRs (s_load_dwordx8) and enter v_fma_mix_f32 as its one scalar operand - no LDS weight reads at all,
//     and 18 ds_read_b128 per 144 FMAs instead of 72 per 288;
//   * GELU = |a| * Phi(|a|) + min(a, 0) with Phi from a 1024-entry (value, slope) table in LDS: linear interpolation on
//     h = 6/1024 is exact to 1.0e-6 (h^2/8 * max|Phi''|), below fp32 round-off of the A&S 7.1.26 form it replaces, at 9 vector
//     instructions and no transcendental instead of 17 with two;
//   * the attention apply is folded into the prologue: the C waves compute x1 of their own two rows straight into their
//     project_out accumulators (which therefore START as the residual: the epilogue is a plain store), the P waves that of the
//     84 border pixels of the halo; M_t is staged once per workgroup in LDS.  LayerNorm-ed fp16 B fragments of all 11 pixel
//     groups live in LDS for the whole kernel (66 KB at C = 96), so the P waves keep only weight fragments in registers.
// LDS: xn 66 KB + hp 2 x 26.6 KB + g 2 x 8 KB + Phi table 8 KB = 143 KB (C = 96).  One barrier per chunk; pipeline per interval i:
// P: hp[i] | all: unit(i-1) | C: project_out(i-2).
// Numerics contract unchanged: fp16 LN output / hidden / gated values, fp16 hi+lo weight planes, fp32 accumulation, LN statistics,
// depthwise conv and GELU; x_out must not alias x (neighbouring patches read each other's halo).
#include "ira_kernels.h"
#include "ira_device.h"

#include <cmath>
#include <mutex>
#include <type_traits>
#include <vector>

namespace ira {

namespace {

constexpr int TH = 8, TW = 32, PH = TH + 2, PW = TW + 2, PHW = PH * PW;   // 340 halo pixels
constexpr int NGI = TH, NGB = 3, NG = NGI + NGB;                            // 8 interior groups (patch rows) + 3 groups of border pixels
constexpr int NBORDER = 2 * PW + 2 * TH;                                    // 84
constexpr int HPS = 40;                                                     // hp row stride in halfs: 16 (x1) + 16 (x2) + 8 pad
constexpr size_t HP_BYTES = (size_t)PHW * HPS * 2;                          // 27 200 B
constexpr size_t G_BYTES = (size_t)4 * 2 * 64 * 16;                         // [row pair][row][lane] half8: 8 192 B
constexpr int NTAB = 1024;
constexpr float TAB_RANGE = 6.0f;
constexpr size_t TAB_BYTES = NTAB * 8;

typedef float f32x8 __attribute__((ext_vector_type(8)));

// halo-linear index (row-major in the 10 x 34 halo patch) of pixel r of group g; -1 for the padding lanes of the last border group
IRA_DEV int group_q(int g, int r) {
    if (g < NGI) return (g + 1) * PW + 1 + r;
    const int i = 32 * (g - NGI) + r;
    if (i < PW) return i;                                   // top halo row
    if (i < 2 * PW) return (PH - 1) * PW + (i - PW);        // bottom halo row
    if (i < 2 * PW + TH) return (i - 2 * PW + 1) * PW;      // left halo column
    if (i < NBORDER) return (i - 2 * PW - TH + 1) * PW + PW - 1;
    return -1;
}

// wave-uniform 8 floats through the scalar cache (constant address space => s_load_dwordx8 when the address is uniform)
IRA_DEV f32x8 sload8(const float* p) {
    return *(const __attribute__((address_space(4))) f32x8*)(unsigned long long)p;
}

typedef float f32x16s __attribute__((ext_vector_type(16)));
IRA_DEV f32x16 sload16(const float* p) {
    return *(const __attribute__((address_space(4))) f32x16*)(unsigned long long)p;
}

// acc += (float)half(lo / hi 16 bits of `packed`) * w with w in an SGPR (the one constant-bus operand of the instruction)
IRA_DEV void fma_mix_lo_s(float& acc, unsigned packed, float w) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(packed), "s"(w));
}
IRA_DEV void fma_mix_hi_s(float& acc, unsigned packed, float w) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(packed), "s"(w));
}
IRA_DEV void dw_tap_s(float (&acc)[8], const u32x4 v, const f32x8 w) {
    fma_mix_lo_s(acc[0], v[0], w[0]); fma_mix_hi_s(acc[1], v[0], w[1]);
    fma_mix_lo_s(acc[2], v[1], w[2]); fma_mix_hi_s(acc[3], v[1], w[3]);
    fma_mix_lo_s(acc[4], v[2], w[4]); fma_mix_hi_s(acc[5], v[2], w[5]);
    fma_mix_lo_s(acc[6], v[3], w[6]); fma_mix_hi_s(acc[7], v[3], w[7]);
}

// gelu(a) * b, gelu(a) = a Phi(a) = |a| Phi(|a|) + min(a, 0); tab[i] = (Phi(i h), Phi((i+1) h) - Phi(i h)), h = 6 / 1024, in LDS
IRA_DEV float gelu_gate_tab(float a, float b, const float2* tab) {
    const float aa = fabsf(a);
    const float tt = fminf(aa * ((float)NTAB / TAB_RANGE), (float)NTAB - 0.004f);
    const int idx = (int)tt;
    const float fr = tt - (float)idx;
    const float2 e = tab[idx];
    const float p = fmaf(fr, e.y, e.x);
    return fmaf(aa, p, fminf(a, 0.f)) * b;
}

}  // namespace

template <int C, int LN, bool BIN, bool CB, bool FOLD>
__global__ __launch_bounds__(512) void k_ffn2(const float* __restrict__ x, float* __restrict__ xo, const f16* __restrict__ v,
                                              const f16* __restrict__ M, long m_stride, const float* __restrict__ wob,
                                              const f16* __restrict__ Win, const float* __restrict__ bin,
                                              const float* __restrict__ dwu, const float* __restrict__ dwub,
                                              const f16* __restrict__ Wout, const float* __restrict__ bout,
                                              const float2* __restrict__ phitab, int T, int H, int W, int Hp) {
    constexpr int KS = C / 16;            // k-steps of project_in / of the attention apply
    constexpr int NBC = (C + 31) / 32;    // 32-row blocks of project_out / of M
    constexpr size_t XN_BYTES = (size_t)NG * KS * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half8* xnl = (half8*)smem;                                              // [NG][KS][64]   LayerNorm-ed B fragments
    f16* hpbuf = (f16*)(smem + XN_BYTES);                                   // [2][PHW][HPS]
    half8* gbuf = (half8*)(smem + XN_BYTES + 2 * HP_BYTES);                 // [2][4][2][64]
    const float2* tab = (const float2*)(smem + XN_BYTES + 2 * HP_BYTES + 2 * G_BYTES);
    half8* Mlds = (half8*)(smem + XN_BYTES);                                // prologue only: [2][NBC][KS][64], overlays hp / g

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int npy = (H + TH - 1) / TH, npx = (W + TW - 1) / TW;
    const long pid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int t = (int)(pid / ((long)npy * npx));
    const int py = (int)((pid / npx) % npy), pxx = (int)(pid % npx);
    const int y0 = py * TH, x0 = pxx * TW;
    const float* xt = x + (long)t * H * W * C;
    const f16* vt = FOLD ? v + (long)t * H * W * C : nullptr;
    const int NCH = Hp >> 4;              // chunks = k-steps of project_out
    const int NBH = Hp >> 5;              // 32-row blocks of x1 in W_in (x2's blocks follow)
    const bool isC = wave >= 4;
    const int cw = wave - 4;

    // pixel of (group g, lane r): in-image flag and pixel offset inside the tile (clamped to pixel 0 when not loadable)
    auto pixel_of = [&](int g, bool& in) -> long {
        const int q = group_q(g, r);
        const int hy = (q < 0 ? 0 : q) / PW, hx = (q < 0 ? 0 : q) - hy * PW;
        const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
        in = q >= 0 && yy >= 0 && yy < H && xx >= 0 && xx < W;
        return in ? (long)yy * W + xx : 0;
    };

    // ------------------------------------------------------------------------------------------------ prologue
    // groups of this wave: C waves their two output rows (interior groups 2cw, 2cw+1; x1 stays in acc as the residual), P waves
    // 0-2 one group of border pixels each
    f32x16 accA[NBC], accB[NBC];
    const int ga = isC ? 2 * cw : NGI + wave, gb = 2 * cw + 1;
    const bool haveA = isC || wave < NGB;
    {
        half8 vA[KS], vB[KS];
        auto load_group = [&](f32x16 (&acc)[NBC], half8 (&vb)[KS], int g) {
            bool in;
            const long pix = pixel_of(g, in);
            const float* src = xt + pix * C;
#pragma unroll
            for (int nb = 0; nb < NBC; ++nb)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (32 * nb + 16 * u + 16 <= C) {
                        const f32x4 lo = *(const f32x4*)(src + 32 * nb + 16 * u + 8 * h), hi = *(const f32x4*)(src + 32 * nb + 16 * u + 8 * h + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { acc[nb][8 * u + e] = lo[e]; acc[nb][8 * u + 4 + e] = hi[e]; }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[nb][8 * u + e] = 0.f;
                    }
                }
            if (FOLD) {
#pragma unroll
                for (int s = 0; s < KS; ++s) vb[s] = *(const half8*)(vt + pix * C + 16 * s + 8 * h);
            }
        };
        if (haveA) load_group(accA, vA, ga);
        if (isC) load_group(accB, vB, gb);
        // stage the tile's attention matrix and the Phi table (LDS writes of all 8 waves; the x / v loads above are in flight)
        if (FOLD) {
            const half8* Mg = (const half8*)(M + (long)t * m_stride);
            for (int i = tid; i < 2 * NBC * KS * 64; i += 512) Mlds[i] = Mg[i];
        }
        ((f32x4*)tab)[tid] = ((const f32x4*)phitab)[tid];
        __syncthreads();

        auto fold_ln_store = [&](f32x16 (&acc)[NBC], const half8 (&vb)[KS], int g) {
            if (FOLD) {
#pragma unroll
                for (int nb = 0; nb < NBC; ++nb) {
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Mlds[((0 * NBC + nb) * KS + s) * 64 + lane], vb[s], acc[nb], 0, 0, 0);
                        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Mlds[((1 * NBC + nb) * KS + s) * 64 + lane], vb[s], acc[nb], 0, 0, 0);
                    }
                    if (wob != nullptr) {
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            if (32 * nb + 16 * u + 16 <= C)
#pragma unroll
                                for (int e = 0; e < 8; ++e) acc[nb][8 * u + e] += wob[32 * nb + 16 * u + 8 * h + e];
                    }
                }
            }
            float sum = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += acc[s >> 1][8 * (s & 1) + e];
            sum += __shfl_xor(sum, 32);
            const float mu = sum * (1.0f / C);
            float sq = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = acc[s >> 1][8 * (s & 1) + e] - mu; sq += d * d; }
            sq += __shfl_xor(sq, 32);
            const float rstd = 1.0f / sqrtf(sq * (1.0f / C) + 1e-5f);
            const float sub = (LN == 2) ? mu : 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (f16)((acc[s >> 1][8 * (s & 1) + e] - sub) * rstd);
                xnl[(g * KS + s) * 64 + lane] = o;
            }
        };
        if (haveA) fold_ln_store(accA, vA, ga);
        if (isC) fold_ln_store(accB, vB, gb);
    }

    // W_in fragment of chunk c for this lane: MFMA rows 0-15 = x1 channels 16c.., rows 16-31 = the gate channels 16c.. (see header)
    const long win_lo = (long)(2 * NBH) * KS * 512;
    auto win_ptr = [&](int c) {
        const int blk = ((r < 16) ? 0 : NBH) + (c >> 1);
        return Win + ((long)blk * KS * 64 + 16 * (c & 1) + (r & 15) + 32 * h) * 8;
    };
    half8 fh[KS], fl[KS];
    auto load_win = [&](int c) {
        const f16* p = win_ptr(c);
#pragma unroll
        for (int s = 0; s < KS; ++s) { fh[s] = *(const half8*)(p + s * 512); fl[s] = *(const half8*)(p + win_lo + s * 512); }
    };
    if (!isC) load_win(0);
    // W_out fragments of chunk c (C waves)
    const int KTO = NCH;
    const long wout_lo = (long)NBC * KTO * 512;
    half8 woh[NBC], wol[NBC];
    auto load_wout = [&](int c) {
#pragma unroll
        for (int nb = 0; nb < NBC; ++nb) {
            woh[nb] = *(const half8*)(Wout + (((long)nb * KTO + c) * 64 + lane) * 8);
            wol[nb] = *(const half8*)(Wout + wout_lo + (((long)nb * KTO + c) * 64 + lane) * 8);
        }
    };
    __syncthreads();        // xn fragments complete; M no longer read: its LDS region becomes the hp / g buffers

    // ------------------------------------------------------------------------------------------------ pipeline pieces
    // An interval is nine SLOTS, one per depthwise tap.  Slot t of a wave: wait for the operands requested in slot t-1 (the only
    // wait in the loop: scalar loads return out of order, so it has to be lgkmcnt(0), and it must come BEFORE the next requests),
    // request slot t+1's operands (tap weights: one s_load_dwordx16; hp values: two ds_read_b128; P: the B fragments of its next
    // MFMAs), then issue this slot's MFMAs with the tap's 16 FMAs dealt between them (a wave's VALU issues in the shadow of its
    // own MFMAs: tools/mb/mb_valu.hip, 1 MFMA + 6 FMAs = 38 cycles).  P: 36 MFMAs per chunk = 9 slots x 4 (C = 48: x 2);
    // C: 12 MFMAs = slots 0-5 x 2.
    constexpr int KPS = KS / 3;           // project_in k-steps per slot
    int pq[3]; bool pin[3];
#pragma unroll
    for (int gi = 0; gi < 3; ++gi) {
        const int g = (wave & 3) + 4 * gi;
        pq[gi] = g < NG ? group_q(g, r) : -1;
        (void)pixel_of(g < NG ? g : 0, pin[gi]);
    }
    const int urp = wave & 3, uoc = wave >> 2;
    f32x16 uw[2];                          // tap weights: [0..7] x1 channels, [8..15] gate channels (SGPRs)
    u32x4 uv1[2], uv2[2];                  // hp values of the tap: x1 octet, gate octet
    float a1[8], a2[8];
    half8 bq[2][KPS];
    auto unit_request = [&](int c, int tap, int slot) {
        const f16* hb = hpbuf + (size_t)(c & 1) * PHW * HPS + ((long)(2 * urp + h + tap / 3) * PW + r + tap % 3) * HPS + 8 * uoc;
        uv1[slot] = *(const u32x4*)hb;
        uv2[slot] = *(const u32x4*)(hb + 16);
        uw[slot] = sload16(dwu + ((long)(c * 2 + uoc) * 9 + tap) * 16);
    };
    auto unit_init = [&](int c) {
        if (CB) {
            const f32x16 bb = sload16(dwub + (long)(c * 2 + uoc) * 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) { a1[e] = bb[e]; a2[e] = bb[8 + e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
        }
    };
    // quarter q (0-3) of a tap's 16 FMAs
    auto unit_fma4 = [&](int slot, int q) {
        float* acc = (q < 2) ? a1 + 4 * q : a2 + 4 * (q - 2);
        const unsigned va = (q < 2) ? uv1[slot][2 * q] : uv2[slot][2 * (q - 2)];
        const unsigned vb = (q < 2) ? uv1[slot][2 * q + 1] : uv2[slot][2 * (q - 2) + 1];
        fma_mix_lo_s(acc[0], va, uw[slot][4 * q + 0]); fma_mix_hi_s(acc[1], va, uw[slot][4 * q + 1]);
        fma_mix_lo_s(acc[2], vb, uw[slot][4 * q + 2]); fma_mix_hi_s(acc[3], vb, uw[slot][4 * q + 3]);
    };
    auto unit_finish = [&](int c) {
        half8 gg;
#pragma unroll
        for (int e = 0; e < 8; ++e) gg[e] = (f16)gelu_gate_tab(a1[e], a2[e], tab);
        gbuf[(((c & 1) * 4 + urp) * 2 + h) * 64 + r + 32 * uoc] = gg;
    };
#define IRA_PIN() __builtin_amdgcn_sched_barrier(0)
#define IRA_WAIT_LGKM() __builtin_amdgcn_s_waitcnt(0xC07F)

    // ------------------------------------------------------------------------------------------------ main loop
    // interval i: P computes hp[i], every wave runs its unit of chunk i-1, C accumulates chunk i-2; one barrier per interval
    // The (produce, unit) / (unit, consume) activity of an interval is a compile-time property of its body: the first and last
    // intervals are peeled so that the steady-state body carries no branches.
    if (!isC) {
        f32x16 acc;
        auto interval = [&](auto PRc, auto UNc, int i) {
            constexpr bool pr = decltype(PRc)::value, un = decltype(UNc)::value;
            f16* hp = hpbuf + (size_t)(i & 1) * PHW * HPS;
            if (un) { unit_init(i - 1); unit_request(i - 1, 0, 0); }
            if (pr) {
#pragma unroll
                for (int k = 0; k < KPS; ++k) bq[0][k] = xnl[(wave * KS + k) * 64 + lane];
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int gi = t / 3, sl = t & 1;
                const bool pv = pr && (gi < 2 || wave < 3);            // wave 3 has two pixel groups only (wave-uniform)
                IRA_WAIT_LGKM();
                IRA_PIN();
                if (un && t < 8) unit_request(i - 1, t + 1, sl ^ 1);
                if (pr && t < 8 && ((t + 1) / 3 < 2 || wave < 3)) {
#pragma unroll
                    for (int k = 0; k < KPS; ++k) bq[sl ^ 1][k] = xnl[((wave + 4 * ((t + 1) / 3)) * KS + KPS * ((t + 1) % 3) + k) * 64 + lane];
                }
                IRA_PIN();
                if (pv) {
                    if (t % 3 == 0) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            acc[e] = BIN ? bin[16 * i + 8 * h + e] : 0.f;
                            acc[8 + e] = BIN ? bin[Hp + 16 * i + 8 * h + e] : 0.f;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < KPS; ++k) {
                        const int ks = KPS * (t % 3) + k;
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ks], bq[sl][k], acc, 0, 0, 0);
                        if (un) { unit_fma4(sl, (KPS == 2) ? 2 * k : 0); if (KPS == 1) unit_fma4(sl, 1); }
                        IRA_PIN();
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[ks], bq[sl][k], acc, 0, 0, 0);
                        if (un) { unit_fma4(sl, (KPS == 2) ? 2 * k + 1 : 2); if (KPS == 1) unit_fma4(sl, 3); }
                        IRA_PIN();
                    }
                    if (t % 3 == 2) {
                        const bool in = pin[gi];
                        const int q = pq[gi];
                        if (q >= 0) {
                            half8 o1, o2;
#pragma unroll
                            for (int e = 0; e < 8; ++e) { o1[e] = in ? (f16)acc[e] : (f16)0.f; o2[e] = in ? (f16)acc[8 + e] : (f16)0.f; }   // zero padding of the dwconv
                            *(half8*)(hp + (long)q * HPS + 8 * h) = o1;
                            *(half8*)(hp + (long)q * HPS + 16 + 8 * h) = o2;
                        }
                    }
                } else if (un) {
                    unit_fma4(sl, 0); unit_fma4(sl, 1); unit_fma4(sl, 2); unit_fma4(sl, 3);
                }
            }
            if (pr && i + 1 < NCH) load_win(i + 1);
            if (un) unit_finish(i - 1);
            lds_barrier();
        };
        interval(std::true_type{}, std::false_type{}, 0);
        for (int i = 1; i < NCH; ++i) interval(std::true_type{}, std::true_type{}, i);
        interval(std::false_type{}, std::true_type{}, NCH);
        return;
    }
    half8 fA, fB;
    auto cinterval = [&](auto UNc, auto COc, int i) {
        constexpr bool un = decltype(UNc)::value, co = decltype(COc)::value;
        if (un) { unit_init(i - 1); unit_request(i - 1, 0, 0); }
        if (co) {
            fA = gbuf[((((i - 2) & 1) * 4 + cw) * 2 + 0) * 64 + lane];
            fB = gbuf[((((i - 2) & 1) * 4 + cw) * 2 + 1) * 64 + lane];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            constexpr int dummy = 0; (void)dummy;
            const int sl = t & 1, nb = (t >> 1) < NBC ? (t >> 1) : 0;
            const bool cv = co && (t >> 1) < NBC;
            IRA_WAIT_LGKM();
            IRA_PIN();
            if (un && t < 8) unit_request(i - 1, t + 1, sl ^ 1);
            IRA_PIN();
            if (cv) { if (t & 1) accB[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(woh[nb], fB, accB[nb], 0, 0, 0);
                      else accA[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(woh[nb], fA, accA[nb], 0, 0, 0); }
            if (un) { unit_fma4(sl, 0); unit_fma4(sl, 1); }
            IRA_PIN();
            if (cv) { if (t & 1) accB[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wol[nb], fB, accB[nb], 0, 0, 0);
                      else accA[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wol[nb], fA, accA[nb], 0, 0, 0); }
            if (un) { unit_fma4(sl, 2); unit_fma4(sl, 3); }
            IRA_PIN();
        }
        if (un) load_wout(i - 1);                           // consumed in the next interval
        if (un) unit_finish(i - 1);
        if (un) lds_barrier();
    };
    lds_barrier();                                          // the barrier of interval 0 (P only)
    cinterval(std::true_type{}, std::false_type{}, 1);
    for (int i = 2; i <= NCH; ++i) cinterval(std::true_type{}, std::true_type{}, i);
    cinterval(std::false_type{}, std::true_type{}, NCH + 1);
#undef IRA_PIN
#undef IRA_WAIT_LGKM

    // ------------------------------------------------------------------------------------------------ epilogue: the accumulators are x_out
#pragma unroll
    for (int rowsel = 0; rowsel < 2; ++rowsel) {
        const int oyy = y0 + 2 * cw + rowsel, oxx = x0 + r;
        if (oyy >= H || oxx >= W) continue;
        const long po = ((long)t * H + oyy) * W + oxx;
#pragma unroll
        for (int nb = 0; nb < NBC; ++nb)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n0 = 32 * nb + 16 * u + 8 * h;
                if (32 * nb + 16 * u + 16 <= C) {
                    f32x4 lo, hi;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        lo[e] = (rowsel ? accB[nb][8 * u + e] : accA[nb][8 * u + e]) + (CB ? bout[n0 + e] : 0.f);
                        hi[e] = (rowsel ? accB[nb][8 * u + 4 + e] : accA[nb][8 * u + 4 + e]) + (CB ? bout[n0 + 4 + e] : 0.f);
                    }
                    *(f32x4*)(xo + po * C + n0) = lo;
                    *(f32x4*)(xo + po * C + n0 + 4) = hi;
                }
            }
    }
}

// ---- Phi table: one copy per device, filled on first use
static hipError_t phi_table(const float2** out) {
    static float2* tabs[64] = {};
    static std::atomic<uint64_t> done{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        static std::mutex mu;
        std::lock_guard<std::mutex> lk(mu);
        if (!(done.load(std::memory_order_acquire) & bit)) {
            std::vector<float2> hst(NTAB);
            const double hstep = (double)TAB_RANGE / NTAB;
            auto phi = [](double z) { return 0.5 * std::erfc(-z * 0.70710678118654752440); };
            for (int i = 0; i < NTAB; ++i) {
                const double p0 = phi(i * hstep), p1 = phi((i + 1) * hstep);
                hst[i].x = (float)p0; hst[i].y = (float)(p1 - p0);
            }
            float2* d = nullptr;
            if ((e = hipMalloc((void**)&d, TAB_BYTES)) != hipSuccess) return e;
            if ((e = hipMemcpy(d, hst.data(), TAB_BYTES, hipMemcpyHostToDevice)) != hipSuccess) return e;
            tabs[dev & 63] = d;
            done.fetch_or(bit, std::memory_order_release);
        }
    }
    *out = tabs[dev & 63];
    return hipSuccess;
}

bool ffn2_supported(int C, int Hp) { return (C == 48 || C == 96) && Hp % 32 == 0; }

template <int C, int LN, bool BIN, bool CB, bool FOLD>
static hipError_t ffn2_launch3(hipStream_t st, const float* x, float* xo, const f16* v, const f16* M, long m_stride, const float* wob,
                               const f16* Win, const float* bin, const float* dww, const float* dwb, const f16* Wout, const float* bout,
                               int T, int H, int W, int Hp) {
    constexpr size_t LDS = (size_t)NG * (C / 16) * 1024 + 2 * HP_BYTES + 2 * G_BYTES + TAB_BYTES;
    static_assert(2 * ((C + 31) / 32) * (C / 16) * 1024 <= 2 * HP_BYTES + 2 * G_BYTES, "M staging must fit the hp / g region");
    const float2* tab = nullptr;
    if (hipError_t e = phi_table(&tab); e != hipSuccess) return e;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)k_ffn2<C, LN, BIN, CB, FOLD>, LDS, attr_done); e != hipSuccess) return e;
    const long nblk = (long)T * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    hipLaunchKernelGGL((k_ffn2<C, LN, BIN, CB, FOLD>), dim3((unsigned)nblk), dim3(512), LDS, st, x, xo, v, M, m_stride, wob, Win, bin, dww, dwb,
                       Wout, bout, tab, T, H, W, Hp);
    return hipGetLastError();
}

template <int C, int LN, bool FOLD>
static hipError_t ffn2_launch2(hipStream_t st, const float* x, float* xo, const f16* v, const f16* M, long m_stride, const float* wob,
                               const f16* Win, const float* bin, const float* dww, const float* dwb, const f16* Wout, const float* bout,
                               int T, int H, int W, int Hp) {
    const bool cb = dwb != nullptr && bout != nullptr;
    if ((dwb != nullptr) != (bout != nullptr)) return hipErrorInvalidValue;
#define IRA_F2(BIN_, CB_) return ffn2_launch3<C, LN, BIN_, CB_, FOLD>(st, x, xo, v, M, m_stride, wob, Win, bin, dww, dwb, Wout, bout, T, H, W, Hp)
    if (bin) { if (cb) IRA_F2(true, true); else IRA_F2(true, false); }
    else { if (cb) IRA_F2(false, true); else IRA_F2(false, false); }
#undef IRA_F2
}

template <int C>
static hipError_t ffn2_launch1(hipStream_t st, const float* x, float* xo, const f16* v, const f16* M, long m_stride, const float* wob,
                               int ln_mode, const f16* Win, const float* bin, const float* dww, const float* dwb, const f16* Wout,
                               const float* bout, int T, int H, int W, int Hp) {
#define IRA_F1(LN_, FOLD_) return ffn2_launch2<C, LN_, FOLD_>(st, x, xo, v, M, m_stride, wob, Win, bin, dww, dwb, Wout, bout, T, H, W, Hp)
    if (ln_mode == 1) { if (v) IRA_F1(1, true); else IRA_F1(1, false); }
    else { if (v) IRA_F1(2, true); else IRA_F1(2, false); }
#undef IRA_F1
}

hipError_t launch_ffn2(hipStream_t st, const float* x, float* xo, const void* v, const void* M, long m_tile_stride, const float* wo_b,
                       int ln_mode, int C, const void* Win, const float* bin, const float* dww, const float* dwb, const void* Wout,
                       const float* bout, int T, int H, int W, int Hp) {
    if (x == xo || (ln_mode != 1 && ln_mode != 2) || !ffn2_supported(C, Hp) || ((v != nullptr) != (M != nullptr))) return hipErrorInvalidValue;
    if (C == 48) return ffn2_launch1<48>(st, x, xo, (const f16*)v, (const f16*)M, m_tile_stride, wo_b, ln_mode, (const f16*)Win, bin, dww, dwb, (const f16*)Wout, bout, T, H, W, Hp);
    return ffn2_launch1<96>(st, x, xo, (const f16*)v, (const f16*)M, m_tile_stride, wo_b, ln_mode, (const f16*)Win, bin, dww, dwb, (const f16*)Wout, bout, T, H, W, Hp);
}

}  // namespace ira
