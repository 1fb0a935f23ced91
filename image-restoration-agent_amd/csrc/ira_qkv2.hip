// Fused attention front end, second generation (restormer_arch.py:114 after :146's norm1), one PERSISTENT kernel:
//     qkv = dwconv3x3( conv1x1( LN(x) ) )            q,k -> fp16 pixel-minor [T][H][ceil(W/8)][2C][8], v -> fp16 NHWC
// Same contract, layouts and rounding points as ira_qkv_fused.hip (generation 1, kept for A/B: IRA_QKV_GEN=1); what changed is how a CU
// spends its time.  Generation 1 (r2 profile: 4.09 ms per 40-tile C = 96 launch for 10.7 GB of HBM traffic = 2.6 TB/s, MFMA floor 0.6 ms,
// vector floor 0.7 ms) ran, per 8 x 32 patch, a 13k-cycle prologue in which every wave waits for its own fragment-shaped loads of x, and
// then nine 4.1k-cycle chunks in which a lone consumer wave per SIMD issues one vector instruction per 4 cycles, reads 54 16-byte LDS
// words per lane and chunk, and stores q,k as 8-byte pieces.  Here:
//   * the workgroup is persistent (one per CU, patches in the XCD-contiguous order of xcd_swizzle): the four C waves request the NEXT
//     patch's 10 x 34 halo rows of x (whole rows: every wave instruction = 1 KB of consecutive bytes) into registers right after the
//     current patch's LayerNorm, so HBM latency is covered by the whole main loop instead of being waited for;
//   * the 1x1 runs TRANSPOSED: D[pixel][channel] = LN(x)[pixel][k] . W[k][channel] (A = pixel fragments, resident in the P waves'
//     registers; B = the packed weight fragments as they are).  With the row permutation of ira_device.h applied to the PIXELS of a
//     group, a lane of the result owns two runs of 8 consecutive pixels of ONE channel: exactly one 16-byte element of the pixel-minor
//     q,k tensor.  The hidden tile goes to LDS as [halo row][channel][pixel] (two ds_write_b128 per lane and group);
//   * the depthwise 3x3 runs on lanes = (channel, 8-pixel block): per halo row one ds_read_b128 (the block) and two ds_read_b64 (the
//     pixel to its left and right: the left halo column sits in the last slot of the previous record so that both are plain offsets),
//     72 v_fma_mix_f32 with the lane's nine weights in registers (no weight reads in the loop), and the result IS the q,k element:
//     one 16-byte store per lane, 512 consecutive bytes per 32 lanes (generation 1: eight 8-byte stores).  v chunks go through a
//     1.25 KB per-wave LDS transposition into [pixel][channel] and are stored as 16-byte pieces as before;
//   * depthwise weights and biases of all chunks are staged ONCE per workgroup in LDS (in the lane order of the transposed result).
//   * x staging is the halo patch as it lies in memory (10 rows x 34 pixels x C floats = 16-byte piece n at LDS piece n ^ ((n >> SH) & 15):
//     an XOR swizzle instead of padding, conflict-free for the P waves' fragment-shaped reads, and a pure function of the LINEAR piece
//     index, so that the C waves' addresses need no division: wave w owns pieces [w N/4, (w+1) N/4) = 2.5 halo rows, piece
//     w N/4 + 64 it + lane in iteration it; for a patch inside the image the global address of an iteration is a wave-uniform base + 16 lane).
// LDS (C = 96): x staging 8 160 pieces = 130 560 B (overlaid by the two hidden buffers, 2 x 25 616 B, and the transposition
// scratch, 4 x 1 280 B, while the main loop runs) + weights / biases 12.7 KB = 143.3 KB: one workgroup per CU, eight waves.
// Barriers per patch: 3 + number of chunks (C = 96: 12).  Numerics: bit-identical to generation 1 (same products in the same order:
// tests/test_gpu_ops.py::test_qkv2_equals_generation1).
#include "ira_kernels.h"
#include "ira_device.h"

#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace ira {

namespace {

constexpr int TH = 8, TW = 32, PH = TH + 2, PW = TW + 2;                  // 10 x 34 halo patch
constexpr int NG = 11;                                                      // 8 interior rows + top row + bottom row + the two halo columns
constexpr int RS = 40;                                                      // halfs per (halo row, channel) record: 32 interior pixels, [32] right halo, [39] left halo of the NEXT record
constexpr int REC0 = 8;                                                     // halfs in front of record 0 (so that record 0 has a left-halo slot)
constexpr size_t HPT_BYTES = (size_t)(REC0 + PH * 32 * RS) * 2;             // 25 616 B (a multiple of 16)
constexpr int SCR_STRIDE = 40;                                              // halfs per pixel row of the v transposition scratch
constexpr size_t SCR_BYTES = (size_t)16 * SCR_STRIDE * 2;                   // 1 280 B per wave
static_assert(HPT_BYTES % 16 == 0, "hidden buffers must keep 16-byte alignment");

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

IRA_DEV unsigned pack2(float lo, float hi) {
    typedef f16 half2v __attribute__((ext_vector_type(2)));
    half2v t; t[0] = (f16)lo; t[1] = (f16)hi;
    return __builtin_bit_cast(unsigned, t);
}

// halo-linear index (hy * 34 + hx) of pixel p of group g, -1 for the padding of the last group
IRA_DEV int group_px(int g, int p) {
    if (g < TH) return (g + 1) * PW + 1 + p;
    if (g == TH) return 1 + p;
    if (g == TH + 1) return (PH - 1) * PW + 1 + p;
    if (p < PH) return p * PW;
    if (p < 2 * PH) return (p - PH) * PW + PW - 1;
    return -1;
}

template <int C> struct Q2 {
    static constexpr int KS = C / 16;
    static constexpr int NQ = 3 * C;
    static constexpr int NCH = (NQ + 31) / 32;
    static constexpr int NQK = 2 * C / 32;                                  // chunks 0 .. NQK-1 are q,k; the rest is v
    static constexpr int PP = C / 4;                                        // 16-byte pieces per pixel
    static constexpr int PPR = PW * PP;                                     // ... per halo row
    static constexpr int NPIECE = PH * PPR;
    static constexpr int PERW = NPIECE / 4;                                 // ... per C wave: 2.5 halo rows
    static constexpr int NIT = (PERW + 63) / 64;                            // ... per C-wave thread
    static constexpr int SH = (C == 96) ? 3 : 2;                            // swizzle shift (found by enumeration: tools/lds_swizzle_search.py)
    static constexpr size_t STG_BYTES = (((size_t)NPIECE + 15) / 16 * 16) * 16;
    static_assert(NPIECE % 4 == 0 && (PERW * 2) % PPR == 0, "a C wave owns 2.5 halo rows");
    static constexpr size_t MAIN_BYTES = 2 * HPT_BYTES + 4 * SCR_BYTES;
    static constexpr size_t DYN_BYTES = STG_BYTES > MAIN_BYTES ? STG_BYTES : MAIN_BYTES;
    static constexpr size_t WDW_BYTES = (size_t)NCH * 9 * 32 * 4;
    static constexpr size_t B_BYTES = (size_t)NCH * 32 * 4;
    static constexpr size_t LDS = DYN_BYTES + WDW_BYTES + 2 * B_BYTES;
    static_assert(2 * C % 32 == 0, "a chunk is either q,k or v");
};

}  // namespace

template <int C, int LN, bool BIN, bool CB>
__global__ __launch_bounds__(512) void k_qkv2(const float* __restrict__ x, const f16* __restrict__ Wq, const float* __restrict__ bq,
                                              const float* __restrict__ dww, const float* __restrict__ dwb, f16* __restrict__ qk,
                                              f16* __restrict__ vout, int T, int H, int W, long NP) {
    using G = Q2<C>;
    constexpr int KS = G::KS, NQ = G::NQ, NCH = G::NCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* xs = (f32x4*)smem;                                               // staging of the halo patch (fp32): piece n at index n ^ ((n >> SH) & 15)
    f16* hpb = (f16*)smem;                                                  // [2] hidden buffers, see header
    f16* scr_all = (f16*)(smem + 2 * HPT_BYTES);
    float* wdw = (float*)(smem + G::DYN_BYTES);                             // [NCH][9][32]   depthwise weights, lane order
    float* bdw = (float*)(smem + G::DYN_BYTES + G::WDW_BYTES);              // [NCH][32]      depthwise bias
    float* b1 = (float*)(smem + G::DYN_BYTES + G::WDW_BYTES + G::B_BYTES);  // [NCH][32]      1x1 bias

    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int npy = (H + TH - 1) / TH, npx = (W + TW - 1) / TW;
    const int WB = (W + 7) >> 3;
    const long G_ = gridDim.x;

    // ---- once per workgroup: depthwise weights / biases in the lane order of the transposed product (column c = channel 32 j + rowperm(c))
    for (int i = tid; i < NCH * 9 * 32; i += 512) {
        const int j = i / 288, rem = i - j * 288, tap = rem >> 5, cc = rem & 31;
        const int ch = 32 * j + rowperm(cc);
        wdw[i] = ch < NQ ? dww[(long)tap * NQ + ch] : 0.f;
    }
    for (int i = tid; i < NCH * 32; i += 512) {
        const int ch = 32 * (i >> 5) + rowperm(i & 31);
        bdw[i] = (CB && ch < NQ) ? dwb[ch] : 0.f;
        b1[i] = (BIN && ch < NQ) ? bq[ch] : 0.f;
    }

    auto coords = [&](long vb, int& t, int& y0, int& x0) {
        const long pid = xcd_swizzle(vb, NP);
        t = (int)(pid / ((long)npy * npx));
        y0 = (int)((pid / npx) % npy) * TH;
        x0 = (int)(pid % npx) * TW;
    };

    if (wave < 4) {
        // ========================================================================= P waves: LayerNorm + transposed 1x1
        const long w_lo = (long)NCH * KS * 1024;       // bytes between the hi and the lo plane
        const unsigned lane16 = lane * 16;             // a fragment = 64 lanes x 16 bytes: wave-uniform base (scalar registers) + this
        using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
        const int ngroups = (wave + 8 < NG) ? 3 : 2;
        const int pr = rowperm(c);                     // A row c of a group holds pixel rowperm(c): see header
        for (long vb = blockIdx.x; vb < NP; vb += G_) {
            int t, y0, x0;
            coords(vb, t, y0, x0);
            const bool inside = y0 >= 1 && y0 + TH < H && x0 >= 1 && x0 + TW < W;       // the whole halo patch lies in the image
            // (declared per patch: nothing of them is live across the patch loop's back edge)
            half8 xa[3][KS];                           // A fragments (pixels x k) of this wave's groups wave, wave + 4, wave + 8
            half8 wh[2][KS], wl[2][KS];                // B fragments (weights, hi / lo planes) of this chunk / the next one
            auto load_w = [&](int j, auto SETc) {
                constexpr int set = decltype(SETc)::value;
                const char* w = (const char*)Wq + (long)j * KS * 1024;
#pragma unroll
                for (int s = 0; s < KS; ++s) { wh[set][s] = *(const half8*)(w + s * 1024 + lane16); wl[set][s] = *(const half8*)(w + w_lo + s * 1024 + lane16); }
            };
            lds_barrier();                             // B0: the previous patch's units are done with the hidden buffers
            load_w(0, S0{});
            lds_barrier();                             // B1: the staging area holds this patch
#pragma unroll
            for (int gi = 0; gi < 3; ++gi) {
                if (gi >= ngroups) break;
                int q = group_px(wave + 4 * gi, pr);
                q = q < 0 ? 0 : q;
                asm volatile("" : "+v"(q));            // recompute the LDS addresses per patch: hoisted out of the patch loop they would cost 36 registers
                const int n0 = q * G::PP + 2 * h;      // first piece of this lane's 8 channels of k-step 0
                float xv[KS][8];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int na = n0 + 4 * s, nb = na + 1;
                    const f32x4 lo = xs[na ^ ((na >> G::SH) & 15)], hi = xs[nb ^ ((nb >> G::SH) & 15)];
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
                    xa[gi][s] = o;
                }
            }
            lds_barrier();                             // B2: every P wave has its pixels: the staging area becomes the hidden buffers

            auto produce = [&](int j, auto SETc) {
                constexpr int set = decltype(SETc)::value;
                f16* hp = hpb + (size_t)(j & 1) * (HPT_BYTES / 2);
                int lofs = c * RS + 8 * h;             // this lane's slot inside a halo row's block of records (halfs)
                asm volatile("" : "+v"(lofs));         // addresses are rebuilt from one register per call: as loop invariants they were kept in ~40 (spilled)
                const float bias = BIN ? b1[32 * j + c] : 0.f;
#pragma unroll
                for (int gi = 0; gi < 3; ++gi) {
                    if (gi >= ngroups) break;
                    const int g = wave + 4 * gi;
                    f32x16 acc;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] = bias;
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[gi][s], wh[set][s], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[gi][s], wl[set][s], acc, 0, 0, 0);
                    }
                    if (g < NG - 1) {
                        // a halo row's 32 interior pixels: this lane owns pixels 8h .. 8h+7 (registers 0-7) and 16+8h .. 16+8h+7 (8-15) of channel c
                        const int hy = g < TH ? g + 1 : (g == TH ? 0 : PH - 1);
                        u32x4 o1, o2;
#pragma unroll
                        for (int k = 0; k < 4; ++k) { o1[k] = pack2(acc[2 * k], acc[2 * k + 1]); o2[k] = pack2(acc[8 + 2 * k], acc[8 + 2 * k + 1]); }
                        if (!inside) {
                            // zero padding of the depthwise conv: pixels outside the image hold zeros, whatever the 1x1 (and its bias) made of them
                            const int yy = y0 - 1 + hy;
                            const bool rowok = yy >= 0 && yy < H;
                            const int n1 = rowok ? W - x0 - 8 * h : 0, n2 = rowok ? W - x0 - 16 - 8 * h : 0;      // valid pixels of the two runs
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                o1[k] &= (2 * k < n1 ? 0xFFFFu : 0u) | (2 * k + 1 < n1 ? 0xFFFF0000u : 0u);
                                o2[k] &= (2 * k < n2 ? 0xFFFFu : 0u) | (2 * k + 1 < n2 ? 0xFFFF0000u : 0u);
                            }
                        }
                        f16* rec = hp + REC0 + hy * 32 * RS + lofs;
                        *(u32x4*)rec = o1;
                        *(u32x4*)(rec + 16) = o2;
                    } else {
                        // the two halo columns.  Group pixel p < 10 = left column (slot -1 of the record) of halo row p, p < 20 = right column (slot 32)
                        // of halo row p - 10; a lane holds p = 8h + e (registers e = 0..7) and 16 + 8h + e (8..15):
                        //   e = 0, 1:  h = 0: left rows 0, 1          h = 1: left rows 8, 9
                        //   e = 2..7:  h = 0: left rows 2..7          h = 1: right rows 0..5
                        //   e = 8..11: h = 0: right rows 6..9         h = 1: nothing (p >= 24)
                        constexpr int ROWH = 32 * RS;                  // halfs per halo row
                        f16* col = hp + REC0 + (lofs - 8 * h);         // record (halo row 0, channel c)
                        f16* bA = col - 1 + (h ? 8 * ROWH : 0);
                        f16* bB = h ? col + 32 - 2 * ROWH : col - 1;
                        f16* bC = col + 32 + 6 * ROWH;
                        auto val = [&](int e, int hy, bool right) -> f16 {
                            const int yy = y0 - 1 + hy, xx = right ? x0 + TW : x0 - 1;
                            const bool ok = inside || (yy >= 0 && yy < H && xx >= 0 && xx < W);
                            return ok ? (f16)acc[e] : (f16)0.f;
                        };
#pragma unroll
                        for (int e = 0; e < 2; ++e) bA[e * ROWH] = val(e, e + 8 * h, false);
#pragma unroll
                        for (int e = 2; e < 8; ++e) bB[e * ROWH] = val(e, h ? e - 2 : e, h != 0);
                        if (h == 0) {
#pragma unroll
                            for (int e = 8; e < 12; ++e) bC[(e - 8) * ROWH] = val(e, e - 2, true);
                        }
                    }
                }
            };
#pragma unroll 1
            for (int j = 0; j < NCH; j += 2) {
                if (j + 1 < NCH) load_w(j + 1, S1{});
                produce(j, S0{});
                lds_barrier();
                if (j + 1 < NCH) {
                    if (j + 2 < NCH) load_w(j + 2, S0{});
                    produce(j + 1, S1{});
                    lds_barrier();
                }
            }
        }
        return;
    }

    // ============================================================================= C waves: halo rows of the next patch + depthwise 3x3 + stores
    const int cw = wave - 4;
    f32x4 piece[G::NIT];
    const int par = cw & 1, hy_start = (5 * cw) >> 1;      // this wave's pieces start at halo row hy_start (+ half a row for odd waves)
    // Patch inside the image: iteration it of this wave reads 64 consecutive pieces of a halo row (or the tail of one row and the head of
    // the next): the address is a wave-uniform row base + 16 bytes per lane.  Everything but the lane term is scalar arithmetic.
    auto prefetch_inside = [&](auto PARc, const float* row0, long wrow) {      // row0: first float of halo row hy_start, wrow: floats per image row
        constexpr int PAR = decltype(PARc)::value, S0 = PAR * (G::PPR / 2);
#pragma unroll
        for (int it = 0; it < G::NIT; ++it) {
            const int n0 = S0 + 64 * it, hy = n0 / G::PPR, wi0 = n0 - hy * G::PPR, thr = G::PPR - wi0;     // compile-time after unrolling
            int ln = lane;
            if (64 * it + 63 >= G::PERW) ln = ln < G::PERW - 1 - 64 * it ? ln : G::PERW - 1 - 64 * it;     // the last iteration is short (56 / 60 pieces): extra lanes repeat the last one
            const float* b0 = row0 + hy * wrow + 4 * wi0;
            if (thr >= 64) piece[it] = *(const f32x4*)(b0 + 4 * ln);
            else {
                const float* b1 = row0 + (hy + 1) * wrow - 4 * thr;
                piece[it] = *(const f32x4*)((ln >= thr ? b1 : b0) + 4 * ln);
            }
        }
    };
    // Patch at the image border: per-piece clamped coordinates (any finite values do: the hidden values of pixels outside the image are zeroed).
    auto prefetch_border = [&](const float* xt, int y0, int x0) {
        int nb = cw * G::PERW + lane;
        asm volatile("" : "+v"(nb));                   // keep this path's per-piece arithmetic out of the patch loop's preheader (it would be kept in registers)
#pragma unroll
        for (int it = 0; it < G::NIT; ++it) {
            int n = nb + 64 * it;
            n = n < (cw + 1) * G::PERW ? n : (cw + 1) * G::PERW - 1;
            const int hy = n / G::PPR, wi = n - hy * G::PPR, hx = wi / G::PP, chq = wi - hx * G::PP;
            int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
            yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
            xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            piece[it] = *(const f32x4*)(xt + ((long)yy * W + xx) * C + 4 * chq);
        }
    };
    auto prefetch = [&](long vb) {
        int t, y0, x0;
        coords(vb, t, y0, x0);
        const float* xt = x + (long)t * H * W * C;
        const bool inside = y0 >= 1 && y0 + TH < H && x0 >= 1 && x0 + TW < W;
        if (inside) {
            const float* row0 = xt + ((long)(y0 - 1 + hy_start) * W + (x0 - 1)) * C;
            if (par) prefetch_inside(std::integral_constant<int, 1>{}, row0, (long)W * C);
            else prefetch_inside(std::integral_constant<int, 0>{}, row0, (long)W * C);
        } else prefetch_border(xt, y0, x0);
    };
    f16* scr = scr_all + (size_t)cw * (SCR_BYTES / 2);                     // this wave's transposition scratch
    prefetch(blockIdx.x);
    for (long vb = blockIdx.x; vb < NP; vb += G_) {
        int t, y0, x0;
        coords(vb, t, y0, x0);
        lds_barrier();                                 // B0
        {
            int nb = cw * G::PERW + lane;              // linear piece index of iteration 0 (halo row * PPR + piece of the row)
            asm volatile("" : "+v"(nb));
#pragma unroll
            for (int it = 0; it < G::NIT; ++it) {
                const int n = nb + 64 * it;
                if (64 * it + 63 < G::PERW || n < (cw + 1) * G::PERW) xs[n ^ ((n >> G::SH) & 15)] = piece[it];
            }
        }
        lds_barrier();                                 // B1
        if (vb + G_ < NP) prefetch(vb + G_);
        lds_barrier();                                 // B2
        lds_barrier();                                 // interval 0: the P waves produce chunk 0
#pragma unroll 1
        for (int j = 0; j < NCH; ++j) {
            const f16* hp = hpb + (size_t)(j & 1) * (HPT_BYTES / 2);
            float wv[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) wv[k] = wdw[(j * 9 + k) * 32 + c];
            const float bias = CB ? bdw[32 * j + c] : 0.f;
            const int ch = 32 * j + rowperm(c);
#pragma unroll
            for (int ui = 0; ui < 4; ++ui) {
                const int u = 4 * cw + ui;             // unit = (patch row, half row): this wave's four units are rows 2cw, 2cw + 1
                const int ry = u >> 1, b = 2 * (u & 1) + h;
                const int oy = y0 + ry;
                float acc[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) acc[p] = bias;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const f16* rec = hp + REC0 + ((ry + dy) * 32 + c) * RS + 8 * b;
                    const u32x4 m = *(const u32x4*)rec;
                    const u32x2 lft = *(const u32x2*)(rec - 4), rgt = *(const u32x2*)(rec + 8);
                    // element e of the row segment (e = -1 .. 8): -1 = lft[1] high half, 0..7 = m, 8 = rgt[0] low half
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float w = wv[dy * 3 + dx];
#pragma unroll
                        for (int p = 0; p < 8; ++p) {
                            const int e = p + dx - 1;
                            if (e < 0) fma_mix_hi(acc[p], lft[1], w);
                            else if (e == 8) fma_mix_lo(acc[p], rgt[0], w);
                            else if (e & 1) fma_mix_hi(acc[p], m[e >> 1], w);
                            else fma_mix_lo(acc[p], m[e >> 1], w);
                        }
                    }
                }
                const int ox = x0 + 8 * b;
                if (j < G::NQK) {
                    // q / k: the lane's 8 pixels of channel ch are one element of the pixel-minor tensor.  Rows are padded to WB * 8 pixels and the
                    // padding must read as zero for the Gram: store every block below WB * 8, zeros for x >= W
                    if (oy < H && ox < WB * 8) {
                        u32x4 o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float lo = (ox + 2 * k < W) ? acc[2 * k] : 0.f, hi = (ox + 2 * k + 1 < W) ? acc[2 * k + 1] : 0.f;
                            o[k] = pack2(lo, hi);
                        }
                        *(u32x4*)(qk + ((((long)t * H + oy) * WB + (ox >> 3)) * (2 * C) + ch) * 8) = o;
                    }
                } else {
                    // v (NHWC): transpose the wave's 16 pixels x 32 channels through LDS, then 16-byte pieces of 8 channels of one pixel
#pragma unroll
                    for (int p = 0; p < 8; ++p) scr[(8 * h + p) * SCR_STRIDE + rowperm(c)] = (f16)acc[p];
                    // the read below takes what OTHER lanes of this wave wrote: LDS operations of one wave execute in order, the compiler
                    // must not move the read above the writes
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    const int p16 = lane >> 2, o8 = lane & 3;
                    const u32x4 o = *(const u32x4*)(scr + p16 * SCR_STRIDE + 8 * o8);
                    const int vx = x0 + 16 * (u & 1) + p16, vch = 32 * j - 2 * C + 8 * o8;
                    if (oy < H && vx < W && vch < C) *(u32x4*)(vout + (((long)t * H + oy) * W + vx) * C + vch) = o;
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if (j + 1 < NCH) lds_barrier();            // (the last interval's barrier is B0 of the next patch)
        }
    }
}

bool qkv2_supported(int C) { return C == 48 || C == 96; }

template <int C, int LN, bool BIN, bool CB>
static hipError_t qkv2_launch2(hipStream_t st, const float* x, const f16* Wq, const float* bq, const float* dww, const float* dwb, f16* qk,
                               f16* v, int T, int H, int W) {
    const long NP = (long)T * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)k_qkv2<C, LN, BIN, CB>, Q2<C>::LDS, attr_done); e != hipSuccess) return e;
    static std::atomic<int> ncu_cache{0};
    int ncu = ncu_cache.load(std::memory_order_relaxed);
    if (ncu == 0) {
        int dev = 0;
        if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
        if (hipError_t e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess) return e;
        ncu_cache.store(ncu, std::memory_order_relaxed);
    }
    // persistent: one workgroup per CU; a multiple of 8 so that block id & 7 (the XCD under round-robin placement) stays the XCD of every patch
    // the workgroup walks (xcd_swizzle keys on the low three bits)
    long grid = NP < ncu ? NP : ncu;
    if (grid >= 8) grid &= ~7L;
    hipLaunchKernelGGL((k_qkv2<C, LN, BIN, CB>), dim3((unsigned)grid), dim3(512), Q2<C>::LDS, st, x, Wq, bq, dww, dwb, qk, v, T, H, W, NP);
    return hipGetLastError();
}

template <int C>
static hipError_t qkv2_launch(hipStream_t st, const float* x, int ln_mode, const f16* Wq, const float* bq, const float* dww, const float* dwb,
                              f16* qk, f16* v, int T, int H, int W) {
#define IRA_QKV2(LN_, BIN_, CB_) return qkv2_launch2<C, LN_, BIN_, CB_>(st, x, Wq, bq, dww, dwb, qk, v, T, H, W)
    if (ln_mode == 1) {
        if (bq) { if (dwb) IRA_QKV2(1, true, true); else IRA_QKV2(1, true, false); }
        else { if (dwb) IRA_QKV2(1, false, true); else IRA_QKV2(1, false, false); }
    } else {
        if (bq) { if (dwb) IRA_QKV2(2, true, true); else IRA_QKV2(2, true, false); }
        else { if (dwb) IRA_QKV2(2, false, true); else IRA_QKV2(2, false, false); }
    }
#undef IRA_QKV2
}

hipError_t launch_qkv2(hipStream_t st, const float* x, int ln_mode, int C, const void* Wq, const float* bq, const float* dww,
                       const float* dwb, void* qk, void* v, int T, int H, int W) {
    if ((ln_mode != 1 && ln_mode != 2) || !qkv2_supported(C)) return hipErrorInvalidValue;
    if (C == 48) return qkv2_launch<48>(st, x, ln_mode, (const f16*)Wq, bq, dww, dwb, (f16*)qk, (f16*)v, T, H, W);
    return qkv2_launch<96>(st, x, ln_mode, (const f16*)Wq, bq, dww, dwb, (f16*)qk, (f16*)v, T, H, W);
}

hipError_t launch_qkv_front(hipStream_t st, const float* x, int ln_mode, int C, const void* Wq, const float* bq, const float* dww,
                            const float* dwb, void* qk, void* v, int T, int H, int W) {
    static const int gen = getenv("IRA_QKV_GEN") ? atoi(getenv("IRA_QKV_GEN")) : 2;
    if (gen == 1) return launch_qkv_fused(st, x, ln_mode, C, Wq, bq, dww, dwb, qk, v, T, H, W);
    return launch_qkv2(st, x, ln_mode, C, Wq, bq, dww, dwb, qk, v, T, H, W);
}

}  // namespace ira
