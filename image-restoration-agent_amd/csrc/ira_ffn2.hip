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
#include <cstdlib>
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
constexpr int NTAB = PHI_NTAB;
constexpr float TAB_RANGE = PHI_RANGE;
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
IRA_DEV float sload1(const float* p) {
    return *(const __attribute__((address_space(4))) float*)(unsigned long long)p;
}
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
template <int N> using IC = std::integral_constant<int, N>;

// ds_read_b128 with the constant part of the address in the instruction's offset field (hipcc kept one address REGISTER per tap
// instead: 18 loop-invariant VGPRs).  The destination is valid only after the caller's s_waitcnt lgkmcnt(0) (cdna_hip_programming.md
// 5.7 item 1, form iii: the wait is a builtin followed by sched_barrier(0), the only consumers are asm statements after it).
template <int OFF>
IRA_DEV u32x4 lds_read128(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}

// two fp32 -> one word of two fp16 (round to nearest even)
IRA_DEV unsigned cvt_pk_f16(float lo, float hi) {
    typedef f16 half2v __attribute__((ext_vector_type(2)));
    half2v t; t[0] = (f16)lo; t[1] = (f16)hi;
    return __builtin_bit_cast(unsigned, t);
}
// first tap: acc = (float)half * w (no accumulator to clear beforehand)
IRA_DEV void mul_mix_lo_s(float& acc, unsigned packed, float w) {
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(acc) : "v"(packed), "s"(w));
}
IRA_DEV void mul_mix_hi_s(float& acc, unsigned packed, float w) {
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(acc) : "v"(packed), "s"(w));
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

// DBG (diagnostic builds of ONE instantiation, selected with IRA_FFN2_DBG at launch; results are wrong or slower, never shipped on a
// product path): 1 = unit weights are constants (no scalar loads), 2 = no vector work at all, 3 = no project_in MFMAs, 4 = s_memtime stamps,
// 12 = 2 + no hidden write-out, 13 = 12 + no project_out MFMAs, 14 = 13 + no project_in MFMAs (barriers, weight loads and LDS fragment reads only)
// They exist only in a -DIRA_KERNEL_DBG build (tools/time_ffn2.py, tools/pmc_ffn2.sh: `make dbg` -> libira_hip_dbg.so); the product library
// instantiates DBG = 0 alone and reads no IRA_FFN2_DBG, so an inherited environment cannot change its results.
#ifdef IRA_KERNEL_DBG
__device__ unsigned long long g_ffn2_stamps[2][64];
#endif

// LOI / LOO: W_in / W_out carry an fp16 lo plane (hi + lo reproduce the fp32 weight to 2^-22) that the kernel multiplies as well.  false: the
// plain-fp16 form for the stages whose lo planes the handle has zeroed (weight_planes == 1): half the MFMAs of that 1x1, and
// bit for bit what the two-plane form computes on the zeroed plane (x + 0 * w = x).
template <int C, int LN, bool BIN, bool CB, bool FOLD, int DBG = 0, bool LOI = true, bool LOO = true>
__global__ __launch_bounds__(512) void k_ffn2(const float* __restrict__ x, float* __restrict__ xo, const f16* __restrict__ v,
                                              const f16* __restrict__ M, long m_stride, const float* __restrict__ wob,
                                              const f16* __restrict__ Win, const float* __restrict__ bin,
                                              const float* __restrict__ dwu, const float* __restrict__ dwub,
                                              const f16* __restrict__ Wout, const float* __restrict__ bout,
                                              const float2* __restrict__ phitab, int T, int H, int W, int Hp, int ldo) {
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
    [[maybe_unused]] int stamp_n = 0;
    auto STAMP = [&]() {
#ifdef IRA_KERNEL_DBG
        if ((DBG == 4 || DBG == 9) && blockIdx.x == 777 && lane == 0 && (wave == 0 || wave == 4) && stamp_n < (DBG == 9 ? 32 : 64)) g_ffn2_stamps[wave >> 2][stamp_n] = __builtin_amdgcn_s_memtime();
        ++stamp_n;
#endif
    };
    [[maybe_unused]] int sstamp_n = 32;
    auto SLOTSTAMP = [&](int i) {
#ifdef IRA_KERNEL_DBG
        if (DBG == 9 && i == 5 && blockIdx.x == 777 && lane == 0 && (wave == 0 || wave == 4) && sstamp_n < 64) g_ffn2_stamps[wave >> 2][sstamp_n++] = __builtin_amdgcn_s_memtime();
#endif
    };
    STAMP();
    // (r1-r2 delayed the first workgroup of every CU by 0-3 quarter patch times so that the CUs' prologues would not coincide; since the
    // weights travel through LDS (r3) the stagger measures nothing: 7.82 / 7.84 / 7.85 / 7.83 ms at 0 / 1 / 3 / 5 units, and it is gone.)
    // pixel of (group g, lane r): in-image flag and pixel offset inside the tile (clamped to pixel 0 when not loadable)
    auto pixel_of = [&](int g, bool& in) -> long {
        const int q = group_q(g, r);
        const int hy = (q < 0 ? 0 : q) / PW, hx = (q < 0 ? 0 : q) - hy * PW;
        const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
        in = q >= 0 && yy >= 0 && yy < H && xx >= 0 && xx < W;
        return in ? (long)yy * W + xx : 0;
    };

    // ------------------------------------------------------------------------------------------------ prologue
    // (1) the 10 x 34 halo patch of x travels HBM -> LDS in whole rows (every wave instruction = 1 KB of consecutive bytes: the
    //     per-lane "fragment shaped" loads of the first generation touched 64 cache lines per instruction and kept the texture
    //     path busy for ~15k cycles per patch), pixel-major with a stride of C + 4 floats (conflict-free for the reads of (2));
    // (2) every wave picks the pixels of its groups out of the staging area in accumulator layout (lane (r, h): channels
    //     32 nb + 16 u + 8 h ..): C waves their two output rows (interior groups 2cw, 2cw+1: x1 stays in the accumulators as the
    //     residual), P waves 0-2 one group of border pixels each;
    // (3) x1 = x + M_t . v with M_t staged in LDS, LayerNorm, fp16 B fragments of all 11 groups -> LDS; (4) the P waves take the
    //     fragments of their three groups into registers for the whole kernel.
    constexpr int XS = C + 4;                                               // staging stride in floats
    static_assert((size_t)PHW * XS * 4 <= XN_BYTES + 2 * HP_BYTES + 2 * G_BYTES, "x staging must end below the Phi table");
    float* xs = (float*)smem;
    // ROLE SEPARATION.  The P and C halves of the kernel are two disjoint branches from here to their ends, each with its own copy of
    // the prologue: the register allocator then never has to keep the C waves' 96 accumulator registers reserved while it allocates
    // the P waves' loop (it did, when the prologue was shared: the P loop ran on 160 registers and spilled).
    auto prologue = [&](auto ISC, f32x16 (&accA)[NBC], f32x16 (&accB)[NBC]) {
        constexpr bool isCc = decltype(ISC)::value;
        const int ga = isCc ? 2 * cw : NGI + wave, gb = 2 * cw + 1;
        const bool haveA = isCc || wave < NGB;
        constexpr int PPR = PW * C / 4;                                     // 16-byte pieces per halo row
        constexpr int NPIECE = PH * PPR, NIT = (NPIECE + 511) / 512;
        f32x4 piece[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int id = it * 512 + tid;
            const int hy = id / PPR, wi = id - hy * PPR, hx = wi / (C / 4);
            const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
            const bool ok = id < NPIECE && yy >= 0 && yy < H && xx >= 0 && xx < W;
            const float* src = xt + ((long)(ok ? yy : 0) * W + (ok ? x0 - 1 : 0)) * C + (ok ? wi * 4 : 0);
            piece[it] = *(const f32x4*)src;
        }
        half8 vA[KS], vB[KS];
        if (FOLD) {
            bool in;
            if (haveA) {
                const long pix = pixel_of(ga, in);
#pragma unroll
                for (int s = 0; s < KS; ++s) vA[s] = *(const half8*)(vt + pix * C + 16 * s + 8 * h);
            }
            if (isCc) {
                const long pix = pixel_of(gb, in);
#pragma unroll
                for (int s = 0; s < KS; ++s) vB[s] = *(const half8*)(vt + pix * C + 16 * s + 8 * h);
            }
        }
        ((f32x4*)tab)[tid] = ((const f32x4*)phitab)[tid];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int id = it * 512 + tid;
            const int hy = id / PPR, wi = id - hy * PPR, hx = wi / (C / 4), ch = wi - hx * (C / 4);
            if (id < NPIECE) *(f32x4*)(xs + (long)(hy * PW + hx) * XS + 4 * ch) = piece[it];
        }
        STAMP();
        __syncthreads();                                                    // (1) done
        STAMP();
        auto take_group = [&](f32x16 (&acc)[NBC], int g) {
            const int q = group_q(g, r);
            const float* src = xs + (long)(q < 0 ? 0 : q) * XS;
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
        };
        if (haveA) take_group(accA, ga);
        if (isCc) take_group(accB, gb);
        STAMP();
        __syncthreads();                                                    // (2) done: the staging area is dead
        STAMP();
        if (FOLD) {
            const half8* Mg = (const half8*)(M + (long)t * m_stride);
            for (int i = tid; i < 2 * NBC * KS * 64; i += 512) Mlds[i] = Mg[i];
            __syncthreads();
        }

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
        if (isCc) fold_ln_store(accB, vB, gb);
    };

    // W_in fragment of chunk c for this lane: MFMA rows 0-15 = x1 channels 16c.., rows 16-31 = the gate channels 16c.. (see header)
    const long win_lo = (long)(2 * NBH) * KS * 512;
    auto win_ptr = [&](int c) {
        const int blk = ((r < 16) ? 0 : NBH) + (c >> 1);
        return Win + ((long)blk * KS * 64 + 16 * (c & 1) + (r & 15) + 32 * h) * 8;
    };
    // r3: the chunk's 2 KS W_in fragments (hi plane, then lo plane; 1 KB each) travel L2 -> LDS by LDS-DMA ONCE per workgroup, one interval
    // ahead (each P wave moves its share; per-lane source addresses: the row gather above), into stage (chunk & 1); the P waves read them from
    // there, ONE ds_read_b128 per fragment, and use each for the MFMAs of their three pixel groups (fragment-major order, three accumulators;
    // a group still sees its products in the old order, so the numbers do not move).  The B operands - the LayerNorm-ed pixels of a P wave's
    // three groups - stay in 72 registers for the whole patch.  Before (r2): weights in two register sets of 48 per P wave (four private copies
    // from L2 per chunk) and one ds_read_b128 of a pixel fragment per MFMA pair.  Stage 0 lies behind the Phi table, stage 1 at the start of
    // the xn region (dead once the P waves hold their fragments).
    constexpr int NF = LOI ? 2 * KS : KS;
    constexpr unsigned WST0 = (unsigned)(XN_BYTES + 2 * HP_BYTES + 2 * G_BYTES + TAB_BYTES);       // byte offset of stage 0 (stage 1: offset 0)
    auto stage_win = [&](int c) {                      // P waves only
        const f16* p = win_ptr(c);
#pragma unroll
        for (int i = 0; i < (NF + 3) / 4; ++i) {
            const int f = wave + 4 * i;
            if (f < NF) {
                const int plane = f >= KS ? 1 : 0, sk = f - plane * KS;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (plane ? win_lo : 0) + sk * 512),
                                                 (__attribute__((address_space(3))) void*)(smem + ((c & 1) ? 0u : WST0) + f * 1024), 16, 0, 0);
            }
        }
    };
    // W_out fragments of chunk c (C waves)
    const int KTO = NCH;
    const long wout_lo = (long)NBC * KTO * 512;
    half8 woh2[2][NBC]; [[maybe_unused]] half8 wol2[2][NBC];
    auto load_wout = [&](int c, int set) {
#pragma unroll
        for (int nb = 0; nb < NBC; ++nb) {
            woh2[set][nb] = *(const half8*)(Wout + (((long)nb * KTO + c) * 64 + lane) * 8);
            if (LOO) wol2[set][nb] = *(const half8*)(Wout + wout_lo + (((long)nb * KTO + c) * 64 + lane) * 8);
        }
    };
    // (the two barrier-separated halves of "(3) done" sit in the role branches below)
    // ------------------------------------------------------------------------------------------------ pipeline pieces
    // An interval is nine SLOTS, one per depthwise tap.  Slot t of a wave: wait for the operands requested in slot t-1 (the only
    // wait in the loop: scalar loads return out of order, so it has to be lgkmcnt(0), and it must come BEFORE the next requests),
    // request slot t+1's operands (tap weights: one s_load_dwordx16; hp values: two ds_read_b128; P: the B fragments of its next
    // MFMAs), then issue this slot's MFMAs with the tap's 16 FMAs dealt between them (a wave's VALU issues in the shadow of its
    // own MFMAs: tools/mb/mb_valu.hip, 1 MFMA + 6 FMAs = 38 cycles).  P: 36 MFMAs per chunk = 9 slots x 4 (C = 48: x 2);
    // C: 12 MFMAs = slots 0-5 x 2.
    int pq[3]; bool pin[3];
#pragma unroll
    for (int gi = 0; gi < 3; ++gi) {
        const int g = (wave & 3) + 4 * gi;
        pq[gi] = g < NG ? group_q(g, r) : -1;
        (void)pixel_of(g < NG ? g : 0, pin[gi]);
    }
    const int urp = wave & 3, uoc = wave >> 2;
    // The unit's 18 HALF TAPS (tap k of the 8 x1 channels for k < 9, tap k - 9 of the 8 gate channels after that) are walked two per
    // slot: x1 is complete in the middle of slot 4, its eight Phi look-ups are issued in slot 5 and finished in slot 7, all in the
    // shadow of the gate's FMAs (and, in P, of the MFMAs) - only the last multiply and the fp16 pack are left behind slot 8.
    f32x16 uw[2];                          // the slot's weights: [0..7] first half tap, [8..15] second half tap (SGPRs)
    u32x4 uva[2], uvb[2];                  // hp values of the two half taps (8 fp16 each)
    float a1[8], a2[8];
    // LDS byte address of this lane's pixel / octet in hp buffer 0 (all of this kernel's LDS is dynamic: smem starts at LDS offset 0)
    const unsigned hb_lane = (unsigned)XN_BYTES + (unsigned)((((2 * urp + h) * PW + r) * HPS + 8 * uoc) * 2);
    auto unit_request = [&](unsigned hbase, const float* wunit, auto TC, int slot) {       // operands of slot t (half taps 2t, 2t+1)
        constexpr int t = decltype(TC)::value;
        constexpr int k0 = 2 * t, k1 = 2 * t + 1, tap0 = k0 < 9 ? k0 : k0 - 9, tap1 = k1 < 9 ? k1 : k1 - 9;
        constexpr int off0 = ((tap0 / 3) * PW + tap0 % 3) * HPS * 2 + (k0 < 9 ? 0 : 32), off1 = ((tap1 / 3) * PW + tap1 % 3) * HPS * 2 + (k1 < 9 ? 0 : 32);
        if (DBG == 5) { uva[slot] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}; uvb[slot] = uva[slot]; }
        else { uva[slot] = lds_read128<off0>(hbase); uvb[slot] = lds_read128<off1>(hbase); }
        if (DBG == 1) { for (int e = 0; e < 16; ++e) uw[slot][e] = 0.01f * (e + 1); }
        else if (t > 0) uw[slot] = sload16(wunit + 16 * t);
    };
    // Slot 0's weights are NOT requested with its hp values at the start of the interval (r3 end): a unit's 576 bytes are nine fresh 64-byte
    // lines for the scalar cache every interval, an s_load that misses there comes back from L2 in about as long as a whole slot lasts, and
    // requested right behind the barrier its latency lay open once per interval (diagnostic build with constant weights: 7.22 -> 6.43 ms per
    // 40-tile launch).  They are requested one interval ahead instead - in slot 8, into the register set slot 7 has just finished with: nine
    // slots per interval, so the sets swap roles from one interval to the next: slot t of interval i uses set (t + i) & 1.
    // ... and the unit's other eight lines are TOUCHED (one dword each) between the wait and the s_barrier of the interval's barrier: the loads
    // fly while the wave sits in the barrier, whichever of the four waves that share the unit's octet arrives first pulls the lines into the
    // scalar cache for all, and the per-slot requests of the next interval hit (7.69 -> 7.39 ms per 40-tile launch at C = 96; at C = 48,
    // where an interval is half as long, 3.04 -> 3.07: not done there).  Ordinary loads into registers that stay reserved until slot 0's wait
    // has passed (unit_touch_done): a scalar load the compiler does not know of would land in a register it has meanwhile given away.
    constexpr bool TOUCH = C == 96;
    float utch[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto unit_touch = [&](const float* wunit) {
        if (DBG == 1 || !TOUCH) return;
#pragma unroll
        for (int k = 0; k < 8; ++k) utch[k] = sload1(wunit + 16 * (k + 1));
    };
    auto unit_touch_done = [&]() {
        if (TOUCH) asm volatile("" :: "s"(utch[0]), "s"(utch[1]), "s"(utch[2]), "s"(utch[3]), "s"(utch[4]), "s"(utch[5]), "s"(utch[6]), "s"(utch[7]));
    };
    auto unit_w0 = [&](const float* wunit, int set) {
        if (DBG == 1) { for (int e = 0; e < 16; ++e) uw[set][e] = 0.01f * (e + 1); }
        else uw[set] = sload16(wunit);
    };
    f32x16 ubias;
    auto unit_init = [&](int c) {
        if (CB) ubias = sload16(dwub + (long)(c * 2 + uoc) * 16);
    };
    // quarter q (0-3) of slot t's 16 FMAs: q < 2 first half tap, q >= 2 second
    auto unit_fma4 = [&](int slot, int q, int t) {
        const int k = 2 * t + (q >> 1);                      // half tap
        float* acc = (k < 9 ? a1 : a2) + 4 * (q & 1);
        const u32x4& vv = (q < 2) ? uva[slot] : uvb[slot];
        const unsigned va = vv[2 * (q & 1)], vb = vv[2 * (q & 1) + 1];
        const int w0 = 8 * (q >> 1) + 4 * (q & 1);
        if (k == 0 || k == 9) {        // first tap of x1 / of the gate: a product (+ bias), not an accumulation
            if (CB) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = ubias[(k == 9 ? 8 : 0) + 4 * (q & 1) + e];
                fma_mix_lo_s(acc[0], va, uw[slot][w0 + 0]); fma_mix_hi_s(acc[1], va, uw[slot][w0 + 1]);
                fma_mix_lo_s(acc[2], vb, uw[slot][w0 + 2]); fma_mix_hi_s(acc[3], vb, uw[slot][w0 + 3]);
            } else {
                mul_mix_lo_s(acc[0], va, uw[slot][w0 + 0]); mul_mix_hi_s(acc[1], va, uw[slot][w0 + 1]);
                mul_mix_lo_s(acc[2], vb, uw[slot][w0 + 2]); mul_mix_hi_s(acc[3], vb, uw[slot][w0 + 3]);
            }
        } else {
            fma_mix_lo_s(acc[0], va, uw[slot][w0 + 0]); fma_mix_hi_s(acc[1], va, uw[slot][w0 + 1]);
            fma_mix_lo_s(acc[2], vb, uw[slot][w0 + 2]); fma_mix_hi_s(acc[3], vb, uw[slot][w0 + 3]);
        }
    };
    // GELU gate, four channels at a time: (a) index / fraction / ds_read_b64 of (Phi, slope) in slots 5 and 6; (b) Phi(|a1|) one slot
    // later; (c) after slot 8: gate, pack, store
    float ft[8]; float2 fe[4];
    auto unit_gelu_a = [&](int e0) {                 // look-ups of channels e0 .. e0 + 3
        if (DBG == 6) return;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float tt = fminf(fabsf(a1[e0 + e]) * ((float)NTAB / TAB_RANGE), (float)NTAB - 0.004f);
            ft[e0 + e] = __builtin_amdgcn_fractf(tt);
            fe[e] = tab[(int)tt];
        }
    };
    auto unit_gelu_b = [&](int e0) {
        if (DBG == 6) return;
#pragma unroll
        for (int e = 0; e < 4; ++e) ft[e0 + e] = fmaf(ft[e0 + e], fe[e].y, fe[e].x);       // Phi(|a1|)
    };
    auto unit_finish = [&](int c) {
        u32x4 gg;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float g0, g1;
            if (DBG == 6) { g0 = a1[2 * j] * a2[2 * j]; g1 = a1[2 * j + 1] * a2[2 * j + 1]; }
            else {
                g0 = fmaf(fabsf(a1[2 * j]), ft[2 * j], fminf(a1[2 * j], 0.f)) * a2[2 * j];
                g1 = fmaf(fabsf(a1[2 * j + 1]), ft[2 * j + 1], fminf(a1[2 * j + 1], 0.f)) * a2[2 * j + 1];
            }
            gg[j] = cvt_pk_f16(g0, g1);
        }
        *(u32x4*)&gbuf[(((c & 1) * 4 + urp) * 2 + h) * 64 + r + 32 * uoc] = gg;
    };
#define IRA_PIN() do { if (DBG != 8) __builtin_amdgcn_sched_barrier(0); } while (0)
#define IRA_WAIT_LGKM() __builtin_amdgcn_s_waitcnt(0xC07F)

    // ------------------------------------------------------------------------------------------------ main loop
    // interval i: P computes hp[i], every wave runs its unit of chunk i-1, C accumulates chunk i-2; one barrier per interval
    // The (produce, unit) / (unit, consume) activity of an interval is a compile-time property of its body: the first and last
    // intervals are peeled so that the steady-state body carries no branches.
    if (!isC) {
        stage_win(0);                              // chunk 0's weights -> stage 0 (they land under the prologue; __syncthreads below waits vmcnt(0))
        {
            f32x16 accT[NBC];                      // P: x1 of one group of border pixels, only to LayerNorm it
            prologue(std::false_type{}, accT, accT);
        }
        STAMP();
        __syncthreads();    // (3) done: xn fragments complete; M no longer read: its LDS region becomes the hp / g buffers
        STAMP();
        half8 xb[3][KS];                           // the B fragments of this wave's pixel groups wave, wave + 4, wave + 8 (wave 3: the third is discarded)
#pragma unroll
        for (int gi = 0; gi < 3; ++gi)
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) xb[gi][s2] = xnl[((wave + 4 * gi) * KS + s2) * 64 + lane];
#pragma unroll
        for (int gi = 0; gi < 3; ++gi) {                 // zero padding: this lane's slots of its out-of-image halo pixels, both hp buffers, once
            if (pq[gi] >= 0 && !pin[gi]) {
                const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int bsel = 0; bsel < 2; ++bsel) {
                    f16* hpz = hpbuf + (size_t)bsel * PHW * HPS + (long)pq[gi] * HPS;
                    *(u32x4*)(hpz + 8 * h) = z;
                    *(u32x4*)(hpz + 16 + 8 * h) = z;
                }
            }
        }
        auto write_out = [&](const f32x16& acc, f16* hp, int gi) {
            if (DBG == 7 || DBG == 12 || DBG == 13 || DBG == 14 || DBG == 15 || DBG == 16) { asm volatile("" :: "v"(acc)); return; }
            const int q = pq[gi];
            if (q >= 0 && pin[gi]) {
                // convert in pairs (v_cvt_pk_f16_f32, round to nearest even).  Pixels outside the image (the zero padding of the depthwise
                // conv) are never written: their slots of both buffers were zeroed once, before interval 0 (8 vector instructions per
                // group instead of 16 with a mask)
                u32x4 o1, o2;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o1[j] = cvt_pk_f16(acc[2 * j], acc[2 * j + 1]);
                    o2[j] = cvt_pk_f16(acc[8 + 2 * j], acc[8 + 2 * j + 1]);
                }
                *(u32x4*)(hp + (long)q * HPS + 8 * h) = o1;
                *(u32x4*)(hp + (long)q * HPS + 16 + 8 * h) = o2;
            }
        };
        lds_barrier();      // every P wave holds its fragments: the start of the xn region becomes weight stage 1
        f32x16 accp[3];
        constexpr int NMS = NF / 3;                // MFMAs per slot (two planes: 4 at C = 96, 2 at C = 48; one plane: 2 / 1)
        auto interval = [&](auto PRc, auto UNc, auto PARc, int i) {
            constexpr bool pr = decltype(PRc)::value, un = decltype(UNc)::value && DBG != 2 && DBG != 12 && DBG != 13 && DBG != 14 && DBG != 15 && DBG != 16;
            if (pr && i + 1 < NCH) stage_win(i + 1);                   // chunk i + 1 -> stage (i + 1) & 1, last read in interval i - 1
            f16* hp = hpbuf + (size_t)(i & 1) * PHW * HPS;
            const char* wcur = smem + ((i & 1) ? 0u : WST0) + lane * 16;
            const unsigned hbase = hb_lane + (unsigned)((i - 1) & 1) * (unsigned)HP_BYTES;       // the unit reads chunk i-1's buffer
            const float* wunit = dwu + (long)((i - 1) * 2 + uoc) * 144;
            constexpr int par = decltype(PARc)::value;                 // = i & 1
            if (un) { unit_init(i - 1); unit_request(hbase, wunit, IC<0>{}, par); }
            // MFMA q of the interval (q = 0 .. 3 NF - 1): fragment ms = q / 3 = (k-step ms >> 1, plane ms & 1: hi, then lo; one plane: k-step ms),
            // group q % 3.
            // fr[ms] is requested one slot before the slot of its first MFMA.
            half8 fr[NF];
            auto frag_ptr = [&](int ms) { return (const half8*)(wcur + (LOI ? (ms & 1) * KS + (ms >> 1) : ms) * 1024); };
            if (pr) {
#pragma unroll
                for (int ms = 0; ms <= (NMS - 1) / 3; ++ms) fr[ms] = *frag_ptr(ms);
            }
            auto slot = [&](auto TC) {
                constexpr int t = decltype(TC)::value, sl = (t + par) & 1;
                SLOTSTAMP(i);
                IRA_WAIT_LGKM();
                SLOTSTAMP(i);
                if constexpr (t == 0) unit_touch_done();
                IRA_PIN();
                if constexpr (un && t < 8) unit_request(hbase, wunit, IC<(t < 8 ? t + 1 : 8)>{}, sl ^ 1);
                if constexpr (t == 8 && DBG != 2 && DBG < 12) { if (i < NCH) unit_w0(dwu + (long)(i * 2 + uoc) * 144, sl ^ 1); }   // the next interval's slot 0
                if (pr && t < 8) {
#pragma unroll
                    for (int ms = (NMS * t + NMS - 1) / 3 + 1; ms <= (NMS * (t + 1) + NMS - 1) / 3; ++ms) fr[ms] = *frag_ptr(ms);
                }
                IRA_PIN();
                if (un && t == 5) unit_gelu_a(0);
                if (un && t == 6) { unit_gelu_b(0); unit_gelu_a(4); }
                if (un && t == 7) unit_gelu_b(4);
                if (pr) {
#pragma unroll
                    for (int m = 0; m < NMS; ++m) {
                        const int q = NMS * t + m, ms = q / 3, gi = q - 3 * ms, ks = LOI ? ms >> 1 : ms;
                        if (ms == 0) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                accp[gi][e] = BIN ? bin[16 * i + 8 * h + e] : 0.f;
                                accp[gi][8 + e] = BIN ? bin[Hp + 16 * i + 8 * h + e] : 0.f;
                            }
                        }
                        if (DBG != 3 && DBG != 14) accp[gi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[ms], xb[gi][ks], accp[gi], 0, 0, 0);
                        if (un) {
#pragma unroll
                            for (int qq = 0; qq < 4 / NMS; ++qq) unit_fma4(sl, m * (4 / NMS) + qq, t);
                        }
                        IRA_PIN();
                    }
                } else {
                    if (un) { unit_fma4(sl, 0, t); unit_fma4(sl, 1, t); unit_fma4(sl, 2, t); unit_fma4(sl, 3, t); }
                }
            };
            slot(IC<0>{}); slot(IC<1>{}); slot(IC<2>{}); slot(IC<3>{}); slot(IC<4>{}); slot(IC<5>{}); slot(IC<6>{}); slot(IC<7>{}); slot(IC<8>{});
            if (pr) {
#pragma unroll
                for (int gi = 0; gi < 3; ++gi) write_out(accp[gi], hp, gi);
            }
            STAMP();
            if (un) unit_finish(i - 1);
            STAMP();
            if (pr) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of chunk i + 1's weights has landed (the P waves have no other vector-memory traffic)
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (DBG != 2 && DBG < 12 && i < NCH) unit_touch(dwu + (long)(i * 2 + uoc) * 144);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            STAMP();
        };
        using I0 = IC<0>; using I1 = IC<1>;
        interval(std::true_type{}, std::false_type{}, I0{}, 0);      // (its slot 8 requests chunk 0's first weights for interval 1)
        for (int i = 1; i + 1 < NCH; i += 2) {                          // NCH is even (Hp % 32 == 0): intervals 1 .. NCH-2 in pairs
            interval(std::true_type{}, std::true_type{}, I1{}, i);
            interval(std::true_type{}, std::true_type{}, I0{}, i + 1);
        }
        interval(std::true_type{}, std::true_type{}, I1{}, NCH - 1);
        interval(std::false_type{}, std::true_type{}, I0{}, NCH);
        return;
    }
    f32x16 accA[NBC], accB[NBC];                   // C: x1 of this wave's two rows = the residual the block's output accumulates on
    prologue(std::true_type{}, accA, accB);
    STAMP();
    __syncthreads();        // (3) done (see the P branch)
    STAMP();
    lds_barrier();          // (the P waves have taken their pixel fragments out of the xn region)
    half8 fA, fB;
    auto cinterval = [&](auto UNc, auto COc, auto PARc, int i) {
        constexpr bool un_ = decltype(UNc)::value, un = un_ && DBG != 2 && DBG != 12 && DBG != 13 && DBG != 14 && DBG != 15 && DBG != 16, co = decltype(COc)::value && DBG != 13 && DBG != 14 && DBG != 15 && DBG != 16;
        constexpr int par = decltype(PARc)::value;                     // (i - 2) & 1 = i & 1: the weight register set chunk i - 2 was loaded into
        if (un_) load_wout(i - 1, par ^ 1);                            // chunk i - 1 is consumed in the next interval
        const unsigned hbase = hb_lane + (unsigned)((i - 1) & 1) * (unsigned)HP_BYTES;
        const float* wunit = dwu + (long)((i - 1) * 2 + uoc) * 144;
        constexpr int upar = par;                                      // i & 1 as well: the unit's register sets, see unit_w0
        if (un) { unit_init(i - 1); unit_request(hbase, wunit, IC<0>{}, upar); }
        if (co) {
            fA = gbuf[((((i - 2) & 1) * 4 + cw) * 2 + 0) * 64 + lane];
            fB = gbuf[((((i - 2) & 1) * 4 + cw) * 2 + 1) * 64 + lane];
        }
        auto slot = [&](auto TC) {
            constexpr int t = decltype(TC)::value, sl = (t + upar) & 1, nb = (t >> 1) < NBC ? (t >> 1) : 0;
            constexpr bool cv = co && (t >> 1) < NBC;
            SLOTSTAMP(i);
            IRA_WAIT_LGKM();
            SLOTSTAMP(i);
            if constexpr (t == 0) unit_touch_done();
            IRA_PIN();
            if constexpr (un && t < 8) unit_request(hbase, wunit, IC<(t < 8 ? t + 1 : 8)>{}, sl ^ 1);
            if constexpr (t == 8 && DBG != 2 && DBG < 12) { if (un_ && i < NCH) unit_w0(dwu + (long)(i * 2 + uoc) * 144, sl ^ 1); }
            IRA_PIN();
            if (un && t == 5) unit_gelu_a(0);
            if (un && t == 6) { unit_gelu_b(0); unit_gelu_a(4); }
            if (un && t == 7) unit_gelu_b(4);
            if (cv) { if (t & 1) accB[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(woh2[par][nb], fB, accB[nb], 0, 0, 0);
                      else accA[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(woh2[par][nb], fA, accA[nb], 0, 0, 0); }
            if (un) { unit_fma4(sl, 0, t); unit_fma4(sl, 1, t); }
            IRA_PIN();
            if (cv && LOO) { if (t & 1) accB[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wol2[par][nb], fB, accB[nb], 0, 0, 0);
                             else accA[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wol2[par][nb], fA, accA[nb], 0, 0, 0); }
            if (un) { unit_fma4(sl, 2, t); unit_fma4(sl, 3, t); }
            IRA_PIN();
        };
        slot(IC<0>{}); slot(IC<1>{}); slot(IC<2>{}); slot(IC<3>{}); slot(IC<4>{}); slot(IC<5>{}); slot(IC<6>{}); slot(IC<7>{}); slot(IC<8>{});
        if (un) unit_finish(i - 1);
        STAMP();
        if (un_) {
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (DBG != 2 && DBG < 12 && i < NCH) unit_touch(dwu + (long)(i * 2 + uoc) * 144);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        STAMP();
    };
    unit_w0(dwu + (long)uoc * 144, 1);                      // chunk 0's unit runs in interval 1: set (0 + 1) & 1
    lds_barrier();                                          // the barrier of interval 0 (P only)
    STAMP();
    {
        using I0 = IC<0>; using I1 = IC<1>;
        cinterval(std::true_type{}, std::false_type{}, I1{}, 1);
        for (int i = 2; i < NCH; i += 2) {
            cinterval(std::true_type{}, std::true_type{}, I0{}, i);
            cinterval(std::true_type{}, std::true_type{}, I1{}, i + 1);
        }
        cinterval(std::true_type{}, std::true_type{}, I0{}, NCH);
        cinterval(std::false_type{}, std::true_type{}, I1{}, NCH + 1);
    }
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
                    *(f32x4*)(xo + po * ldo + n0) = lo;                 // ldo floats per output pixel (C, or the width of a concat buffer)
                    *(f32x4*)(xo + po * ldo + n0 + 4) = hi;
                }
            }
    }
}

// ---- Phi table: one copy per device, filled on first use
hipError_t phi_table(const float2** out) {
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

#ifdef IRA_KERNEL_DBG
extern "C" int ira_debug_ffn2_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ffn2_stamps), sizeof(unsigned long long) * 128);
}
#endif

bool ffn2_supported(int C, int Hp) { return (C == 48 || C == 96) && Hp % 32 == 0; }
// lo-plane masks with an instantiation (bit 0: W_in has a lo plane, bit 1: W_out has one): both everywhere; C = 96 also W_in only and neither
bool ffn2_lo_supported(int C, int lo) { return lo == 3 || (C == 96 && (lo == 1 || lo == 0)); }

template <int C, int LN, bool BIN, bool CB, bool FOLD, bool LOI = true, bool LOO = true>
static hipError_t ffn2_launch4(hipStream_t st, const float* x, float* xo, const f16* v, const f16* M, long m_stride, const float* wob,
                               const f16* Win, const float* bin, const float* dww, const float* dwb, const f16* Wout, const float* bout,
                               int T, int H, int W, int Hp, int ldo) {
    constexpr size_t LDS = (size_t)NG * (C / 16) * 1024 + 2 * HP_BYTES + 2 * G_BYTES + TAB_BYTES + (size_t)2 * (C / 16) * 1024;   // + weight stage 0
    static_assert(2 * ((C + 31) / 32) * (C / 16) * 1024 <= 2 * HP_BYTES + 2 * G_BYTES, "M staging must fit the hp / g region");
    const float2* tab = nullptr;
    if (hipError_t e = phi_table(&tab); e != hipSuccess) return e;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)k_ffn2<C, LN, BIN, CB, FOLD, 0, LOI, LOO>, LDS, attr_done); e != hipSuccess) return e;
    const long nblk = (long)T * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
#ifdef IRA_KERNEL_DBG
    if constexpr (C == 96 && LN == 1 && !BIN && !CB && !FOLD && LOI && LOO) {
        static const int dbg = getenv("IRA_FFN2_DBG") ? atoi(getenv("IRA_FFN2_DBG")) : 0;
#define IRA_DBG_LAUNCH(D_)                                                                                                                     \
    if (dbg == D_) {                                                                                                                            \
        static std::atomic<uint64_t> ad{0};                                                                                                     \
        if (hipError_t e = ensure_dynamic_lds((const void*)k_ffn2<C, LN, BIN, CB, FOLD, D_>, LDS, ad); e != hipSuccess) return e;                \
        hipLaunchKernelGGL((k_ffn2<C, LN, BIN, CB, FOLD, D_>), dim3((unsigned)nblk), dim3(512), LDS, st, x, xo, v, M, m_stride, wob, Win, bin, \
                           dww, dwb, Wout, bout, tab, T, H, W, Hp, ldo);                                                                    \
        return hipGetLastError();                                                                                                               \
    }
        IRA_DBG_LAUNCH(1) IRA_DBG_LAUNCH(2) IRA_DBG_LAUNCH(3) IRA_DBG_LAUNCH(4) IRA_DBG_LAUNCH(5) IRA_DBG_LAUNCH(6) IRA_DBG_LAUNCH(7) IRA_DBG_LAUNCH(8) IRA_DBG_LAUNCH(9) IRA_DBG_LAUNCH(12) IRA_DBG_LAUNCH(13) IRA_DBG_LAUNCH(14) IRA_DBG_LAUNCH(15) IRA_DBG_LAUNCH(16)
#undef IRA_DBG_LAUNCH
    }
#endif
    hipLaunchKernelGGL((k_ffn2<C, LN, BIN, CB, FOLD, 0, LOI, LOO>), dim3((unsigned)nblk), dim3(512), LDS, st, x, xo, v, M, m_stride, wob, Win, bin, dww, dwb,
                       Wout, bout, tab, T, H, W, Hp, ldo);
    return hipGetLastError();
}

template <int C, int LN, bool BIN, bool CB, bool FOLD>
static hipError_t ffn2_launch3(hipStream_t st, const float* x, float* xo, const f16* v, const f16* M, long m_stride, const float* wob,
                               const f16* Win, const float* bin, const float* dww, const float* dwb, const f16* Wout, const float* bout,
                               int T, int H, int W, int Hp, int ldo, int lo) {
#define IRA_F3(LOI_, LOO_) return ffn2_launch4<C, LN, BIN, CB, FOLD, LOI_, LOO_>(st, x, xo, v, M, m_stride, wob, Win, bin, dww, dwb, Wout, bout, T, H, W, Hp, ldo)
    if constexpr (C == 96) {
        if (lo == 0) IRA_F3(false, false);
        if (lo == 1) IRA_F3(true, false);
    }
    IRA_F3(true, true);
#undef IRA_F3
}

template <int C, int LN, bool FOLD>
static hipError_t ffn2_launch2(hipStream_t st, const float* x, float* xo, const f16* v, const f16* M, long m_stride, const float* wob,
                               const f16* Win, const float* bin, const float* dww, const float* dwb, const f16* Wout, const float* bout,
                               int T, int H, int W, int Hp, int ldo, int lo) {
    const bool cb = dwb != nullptr && bout != nullptr;
    if ((dwb != nullptr) != (bout != nullptr)) return hipErrorInvalidValue;
#define IRA_F2(BIN_, CB_) return ffn2_launch3<C, LN, BIN_, CB_, FOLD>(st, x, xo, v, M, m_stride, wob, Win, bin, dww, dwb, Wout, bout, T, H, W, Hp, ldo, lo)
    if (bin) { if (cb) IRA_F2(true, true); else IRA_F2(true, false); }
    else { if (cb) IRA_F2(false, true); else IRA_F2(false, false); }
#undef IRA_F2
}

template <int C>
static hipError_t ffn2_launch1(hipStream_t st, const float* x, float* xo, const f16* v, const f16* M, long m_stride, const float* wob,
                               int ln_mode, const f16* Win, const float* bin, const float* dww, const float* dwb, const f16* Wout,
                               const float* bout, int T, int H, int W, int Hp, int ldo, int lo) {
#define IRA_F1(LN_, FOLD_) return ffn2_launch2<C, LN_, FOLD_>(st, x, xo, v, M, m_stride, wob, Win, bin, dww, dwb, Wout, bout, T, H, W, Hp, ldo, lo)
    if (ln_mode == 1) { if (v) IRA_F1(1, true); else IRA_F1(1, false); }
    else { if (v) IRA_F1(2, true); else IRA_F1(2, false); }
#undef IRA_F1
}

hipError_t launch_ffn2(hipStream_t st, const float* x, float* xo, const void* v, const void* M, long m_tile_stride, const float* wo_b,
                       int ln_mode, int C, const void* Win, const float* bin, const float* dww, const float* dwb, const void* Wout,
                       const float* bout, int T, int H, int W, int Hp, int ldo, int lo) {
    if (ldo == 0) ldo = C;
    if (!ffn2_lo_supported(C, lo)) return hipErrorInvalidValue;
    if (ldo < C || (ldo & 3)) return hipErrorInvalidValue;
    if (x == xo || (ln_mode != 1 && ln_mode != 2) || !ffn2_supported(C, Hp) || ((v != nullptr) != (M != nullptr))) return hipErrorInvalidValue;
    if (C == 48) return ffn2_launch1<48>(st, x, xo, (const f16*)v, (const f16*)M, m_tile_stride, wo_b, ln_mode, (const f16*)Win, bin, dww, dwb, (const f16*)Wout, bout, T, H, W, Hp, ldo, lo);
    return ffn2_launch1<96>(st, x, xo, (const f16*)v, (const f16*)M, m_tile_stride, wo_b, ln_mode, (const f16*)Win, bin, dww, dwb, (const f16*)Wout, bout, T, H, W, Hp, ldo, lo);
}

}  // namespace ira
