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
// Barriers per patch: 3 + number of chunks (C = 96: 12).  Numerics: the rounding points and products of generation 1; the transposed MFMA
// forms its fp32 sums in another order, so about 6e-6 of the fp16 hidden values land on the other neighbour (measured; everything else is
// bit-identical: tests/test_gpu_ops.py::test_qkv2_equals_generation1).  Deterministic, and a patch's result does not depend on the launch.
#include "ira_kernels.h"
#include "ira_device.h"

#include <atomic>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace ira {

namespace {

constexpr int TH = 8, TW = 32, PH = TH + 2, PW = TW + 2;                  // 10 x 34 halo patch
constexpr int NWP = 8, NWC = 8, NTHR = 64 * (NWP + NWC);                    // waves 0-7: LayerNorm + 1x1 (P); waves 8-15: halo rows, depthwise 3x3, stores (C)
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
    static constexpr int PERW = NPIECE / NWC;                               // ... per C wave: 1.25 halo rows
    static constexpr int NIT = (PERW + 63) / 64;                            // ... per C-wave thread
    static constexpr int SH = (C == 96) ? 3 : 2;                            // swizzle shift (found by enumeration: tools/lds_swizzle_search.py)
    static constexpr size_t STG_BYTES = (((size_t)NPIECE + 15) / 16 * 16) * 16;
    static_assert(NPIECE % NWC == 0 && (PERW * 4) % PPR == 0 && PPR % 4 == 0, "a C wave owns 1.25 halo rows");
    static constexpr size_t MAIN_BYTES = 2 * HPT_BYTES + NWC * SCR_BYTES;
    static constexpr size_t DYN_BYTES = STG_BYTES > MAIN_BYTES ? STG_BYTES : MAIN_BYTES;
    static constexpr size_t WDW_BYTES = (size_t)NCH * 9 * 32 * 4;
    static constexpr size_t B_BYTES = (size_t)NCH * 32 * 4;
    static constexpr size_t LDS = DYN_BYTES + WDW_BYTES + 2 * B_BYTES;
    static_assert(2 * C % 32 == 0, "a chunk is either q,k or v");
};

}  // namespace

#ifdef IRA_KERNEL_DBG
__device__ unsigned long long g_qkv2_stamps[2][64];     // [P wave 0 / C wave 8][event] of workgroup 17 (make dbg; tools/time_qkv.py)
#define QSTAMP(role) do { if (blockIdx.x == 17 && lane == 0 && (wave == 0 || wave == NWP) && stamp_n < 64) g_qkv2_stamps[role][stamp_n] = __builtin_amdgcn_s_memtime(); ++stamp_n; } while (0)
#else
#define QSTAMP(role) do {} while (0)
#endif
// Measurement variants (wrong results): 1 no global stores, 2 no halo-row loads, 4 no MFMAs, 8 no depthwise FMAs.  Compiled only with
// -DIRA_KERNEL_DBG_FLAGS (not part of `make dbg`: the one build that had them faulted on the GPU and the cause was not found).
#ifdef IRA_KERNEL_DBG_FLAGS
__device__ int g_qkv2_dbg;
#define QDBG(bit) (g_dbg & (bit))
#else
#define QDBG(bit) false
#endif

template <int C, int LN, bool BIN, bool CB>
__global__ __launch_bounds__(NTHR) void k_qkv2(const float* __restrict__ x, const f16* __restrict__ Wq, const float* __restrict__ bq,
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
    [[maybe_unused]] int stamp_n = 0;
#ifdef IRA_KERNEL_DBG_FLAGS
    const int g_dbg = __builtin_amdgcn_readfirstlane(g_qkv2_dbg);
#endif

    // ---- once per workgroup: depthwise weights / biases in the lane order of the transposed product (column c = channel 32 j + rowperm(c))
    for (int i = tid; i < NCH * 9 * 32; i += NTHR) {
        const int j = i / 288, rem = i - j * 288, tap = rem >> 5, cc = rem & 31;
        const int ch = 32 * j + rowperm(cc);
        wdw[i] = ch < NQ ? dww[(long)tap * NQ + ch] : 0.f;
    }
    for (int i = tid; i < NCH * 32; i += NTHR) {
        const int ch = 32 * (i >> 5) + rowperm(i & 31);
        bdw[i] = (CB && ch < NQ) ? dwb[ch] : 0.f;
        b1[i] = (BIN && ch < NQ) ? bq[ch] : 0.f;
    }

    auto coords = [&](long vb, int& t, int& y0, int& x0) __attribute__((always_inline)) {
        const long pid = xcd_swizzle(vb, NP);
        t = (int)(pid / ((long)npy * npx));
        y0 = (int)((pid / npx) % npy) * TH;
        x0 = (int)(pid % npx) * TW;
    };

    if (wave < NWP) {
        // ========================================================================= P waves: LayerNorm + transposed 1x1
        const long w_lo = (long)NCH * KS * 1024;       // bytes between the hi and the lo plane
        const unsigned lane16 = lane * 16;             // a fragment = 64 lanes x 16 bytes: wave-uniform base (scalar registers) + this
        using S0 = std::integral_constant<int, 0>;
        const int ngroups = (wave + NWP < NG) ? 2 : 1;     // groups wave and wave + 8 (waves 0-2)
        const int pr = rowperm(c);                     // A row c of a group holds pixel rowperm(c): see header
        for (long vb = blockIdx.x; vb < NP; vb += G_) {
            int t, y0, x0;
            coords(vb, t, y0, x0);
            const bool inside = y0 >= 1 && y0 + TH < H && x0 >= 1 && x0 + TW < W;       // the whole halo patch lies in the image
            // (declared per patch: nothing of them is live across the patch loop's back edge)
            half8 xa[2][KS];                           // A fragments (pixels x k) of this wave's groups wave, wave + 8
            half8 wh[1][KS], wl[1][KS];                // B fragments (weights, hi / lo planes) of one chunk: the next chunk's are requested into the
                                                       // same registers right behind the last MFMA (four waves per SIMD cover the latency; 128 registers per wave)
            auto load_w = [&](int j, auto SETc) __attribute__((always_inline)) {
                constexpr int set = decltype(SETc)::value;
                const char* w = (const char*)Wq + (long)j * KS * 1024;
                unsigned l16 = lane16;
                asm volatile("" : "+v"(l16));          // scalar base + this one register: as loop invariants the twelve per-lane addresses were kept (and spilled)
#pragma unroll
                for (int s = 0; s < KS; ++s) { wh[set][s] = *(const half8*)(w + s * 1024 + l16); wl[set][s] = *(const half8*)(w + w_lo + s * 1024 + l16); }
            };
            QSTAMP(0);
            lds_barrier();                             // B0: the previous patch's units are done with the hidden buffers
            QSTAMP(0);
            lds_barrier();                             // B1: the staging area holds this patch
            QSTAMP(0);
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                if (gi >= ngroups) break;
                int q = group_px(wave + NWP * gi, pr);
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
            load_w(0, S0{});                           // (requested here, not before the LayerNorm: 48 more live registers there meant spills)
            QSTAMP(0);
            lds_barrier();                             // B2: every P wave has its pixels: the staging area becomes the hidden buffers
            QSTAMP(0);

            auto produce = [&](int j, auto SETc) __attribute__((always_inline)) {
                constexpr int set = decltype(SETc)::value;
                f16* hp = hpb + (size_t)(j & 1) * (HPT_BYTES / 2);
                int lofs = c * RS + 8 * h;             // this lane's slot inside a halo row's block of records (halfs)
                asm volatile("" : "+v"(lofs));         // addresses are rebuilt from one register per call: as loop invariants they were kept in ~40 (spilled)
                const float bias = BIN ? b1[32 * j + c] : 0.f;
#pragma unroll
                for (int gi = 0; gi < 2; ++gi) {
                    if (gi >= ngroups) break;
                    const int g = wave + NWP * gi;
                    f32x16 acc;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] = bias;
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        if (QDBG(4)) { asm volatile("" :: "v"(xa[gi][s]), "v"(wh[set][s]), "v"(wl[set][s])); continue; }
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
                        auto val = [&](int e, int hy, bool right) __attribute__((always_inline)) -> f16 {
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
            for (int j = 0; j < NCH; ++j) {
                produce(j, S0{});
                __builtin_amdgcn_sched_barrier(0);     // the requests below reuse the weight registers: they must stay behind the last MFMA
                if (j + 1 < NCH) load_w(j + 1, S0{});
                __builtin_amdgcn_sched_barrier(0);
                QSTAMP(0);
                lds_barrier();
                QSTAMP(0);
            }
        }
        return;
    }

    // ============================================================================= C waves: halo rows of the next patch + depthwise 3x3 + stores
    const int cw = wave - NWP;
    f32x4 piece[G::NIT];
    const int par = cw & 3, hy_start = (5 * cw) >> 2;      // this wave's pieces start at halo row hy_start, par quarter rows in
    // The NEXT patch's pieces are requested ONE PER UNIT of the current patch's main loop (36 units for 32 pieces at C = 96): a wave cannot
    // issue vector-memory instructions faster than the CU's share of HBM bandwidth drains them (~10 bytes per cycle: 130 KB in one burst kept the
    // C waves at the issue for 13.7k cycles, measured), so the requests are spread over the whole patch and ride beside the stores.
    // Patch inside the image: piece `it` of this wave = 64 consecutive pieces of a halo row (or the tail of one row and the head of the
    // next): a wave-uniform row base + 16 bytes per lane; everything but the lane term is scalar arithmetic.
    // Patch at the image border: per-piece clamped coordinates (any finite values do: the hidden values of pixels outside the image are zeroed).
    struct Next { const float* xt; const float* row0; long wrow; int y0, x0; bool inside, any; } nx{};
    auto set_next = [&](long vb) __attribute__((always_inline)) {
        nx.any = vb < NP;
        if (!nx.any) return;
        int t, y0, x0;
        coords(vb, t, y0, x0);
        nx.xt = x + (long)t * H * W * C;
        nx.y0 = y0; nx.x0 = x0;
        nx.inside = y0 >= 1 && y0 + TH < H && x0 >= 1 && x0 + TW < W;
        nx.row0 = nx.xt + ((long)(y0 - 1 + hy_start) * W + (x0 - 1)) * C;
        nx.wrow = (long)W * C;
    };
    auto fetch_inside = [&](auto PARc, auto ITc) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PARc)::value, it = decltype(ITc)::value, S0 = PAR * (G::PPR / 4);
        constexpr int n0 = S0 + 64 * it, hy = n0 / G::PPR, wi0 = n0 - hy * G::PPR, thr = G::PPR - wi0;
        int ln = lane;
        if (64 * it + 63 >= G::PERW) ln = ln < G::PERW - 1 - 64 * it ? ln : G::PERW - 1 - 64 * it;     // the last piece is short (56 / 60 lanes): extra lanes repeat the last one
        const float* b0 = nx.row0 + hy * nx.wrow + 4 * wi0;
        if (thr >= 64) piece[it] = *(const f32x4*)(b0 + 4 * ln);
        else {
            const float* b1 = nx.row0 + (hy + 1) * nx.wrow - 4 * thr;
            piece[it] = *(const f32x4*)((ln >= thr ? b1 : b0) + 4 * ln);
        }
    };
    auto fetch_border = [&](auto ITc) __attribute__((always_inline)) {
        constexpr int it = decltype(ITc)::value;
        int n = cw * G::PERW + lane + 64 * it;
        asm volatile("" : "+v"(n));                    // (keeps this rarely used arithmetic from being hoisted out of the patch loop into registers)
        n = n < (cw + 1) * G::PERW ? n : (cw + 1) * G::PERW - 1;
        const int hy = n / G::PPR, wi = n - hy * G::PPR, hx = wi / G::PP, chq = wi - hx * G::PP;
        int yy = nx.y0 - 1 + hy, xx = nx.x0 - 1 + hx;
        yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
        xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
        piece[it] = *(const f32x4*)(nx.xt + ((long)yy * W + xx) * C + 4 * chq);
    };
    auto fetch = [&](auto ITc) __attribute__((always_inline)) {
        constexpr int it = decltype(ITc)::value;
        if constexpr (it < G::NIT) {
            if (!nx.any || QDBG(2)) return;
            if (nx.inside) {
                if (par == 0) fetch_inside(std::integral_constant<int, 0>{}, ITc);
                else if (par == 1) fetch_inside(std::integral_constant<int, 1>{}, ITc);
                else if (par == 2) fetch_inside(std::integral_constant<int, 2>{}, ITc);
                else fetch_inside(std::integral_constant<int, 3>{}, ITc);
            } else fetch_border(ITc);
        }
    };
    auto fetch_all_from = [&](auto FROMc) __attribute__((always_inline)) {            // pieces FROM .. NIT-1 (all of them before the first patch; the rest after the last unit)
        constexpr int from = decltype(FROMc)::value;
        [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) { (fetch(std::integral_constant<int, from + I>{}), ...); }(std::make_integer_sequence<int, (G::NIT > from ? G::NIT - from : 0)>{});
    };
    f16* scr = scr_all + (size_t)cw * (SCR_BYTES / 2);                     // this wave's transposition scratch
    set_next(blockIdx.x);
    fetch_all_from(std::integral_constant<int, 0>{});
    for (long vb = blockIdx.x; vb < NP; vb += G_) {
        int t, y0, x0;
        coords(vb, t, y0, x0);
        QSTAMP(1);
        lds_barrier();                                 // B0
        QSTAMP(1);
        {
            int nb = cw * G::PERW + lane;              // linear piece index of piece 0 (halo row * PPR + piece of the row)
            asm volatile("" : "+v"(nb));
#pragma unroll
            for (int it = 0; it < G::NIT; ++it) {
                const int n = nb + 64 * it;
                if (64 * it + 63 < G::PERW || n < (cw + 1) * G::PERW) xs[n ^ ((n >> G::SH) & 15)] = piece[it];
            }
        }
        QSTAMP(1);
        lds_barrier();                                 // B1
        QSTAMP(1);
        set_next(vb + G_);
        QSTAMP(1);
        lds_barrier();                                 // B2
        lds_barrier();                                 // interval 0: the P waves produce chunk 0
        QSTAMP(1);
        auto chunk = [&](auto Jc) __attribute__((always_inline)) {
            constexpr int j = decltype(Jc)::value;
            const f16* hp = hpb + (size_t)(j & 1) * (HPT_BYTES / 2);
            float wv[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) wv[k] = wdw[(j * 9 + k) * 32 + c];
            const float bias = CB ? bdw[32 * j + c] : 0.f;
            const int ch = 32 * j + rowperm(c);
            auto unit = [&](auto UIc) __attribute__((always_inline)) {
                constexpr int ui = decltype(UIc)::value;
                fetch(std::integral_constant<int, 2 * j + ui>{});          // one piece of the next patch per unit
                const int u = 2 * cw + ui;             // unit = (patch row, half row): this wave's two units are the halves of patch row cw
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
                            if (QDBG(8)) { asm volatile("" :: "v"(m), "v"(lft), "v"(rgt)); continue; }
                            if (e < 0) fma_mix_hi(acc[p], lft[1], w);
                            else if (e == 8) fma_mix_lo(acc[p], rgt[0], w);
                            else if (e & 1) fma_mix_hi(acc[p], m[e >> 1], w);
                            else fma_mix_lo(acc[p], m[e >> 1], w);
                        }
                    }
                }
                const int ox = x0 + 8 * b;
                if constexpr (j < G::NQK) {
                    // q / k: the lane's 8 pixels of channel ch are one element of the pixel-minor tensor.  Rows are padded to WB * 8 pixels and the
                    // padding must read as zero for the Gram: store every block below WB * 8, zeros for x >= W
                    if (oy < H && ox < WB * 8 && !QDBG(1)) {
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
                    if (oy < H && vx < W && vch < C && !QDBG(1)) *(u32x4*)(vout + (((long)t * H + oy) * W + vx) * C + vch) = o;
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                }
            };
            unit(std::integral_constant<int, 0>{}); unit(std::integral_constant<int, 1>{});
            if constexpr (j + 1 == NCH) fetch_all_from(std::integral_constant<int, 2 * NCH>{});      // (none left: 16 pieces for 18 units at C = 96, 8 for 10 at C = 48)
            QSTAMP(1);
            if constexpr (j + 1 < NCH) lds_barrier();  // (the last interval's barrier is B0 of the next patch)
            QSTAMP(1);
        };
        [&]<int... J>(std::integer_sequence<int, J...>) __attribute__((always_inline)) { (chunk(std::integral_constant<int, J>{}), ...); }(std::make_integer_sequence<int, NCH>{});
    }
}

#ifdef IRA_KERNEL_DBG
#ifdef IRA_KERNEL_DBG_FLAGS
extern "C" int ira_debug_qkv2_flags(int flags) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_qkv2_dbg), &flags, sizeof flags); }
#endif
extern "C" int ira_debug_qkv2_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qkv2_stamps), sizeof(unsigned long long) * 128);
}
#endif

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
    hipLaunchKernelGGL((k_qkv2<C, LN, BIN, CB>), dim3((unsigned)grid), dim3(NTHR), Q2<C>::LDS, st, x, Wq, bq, dww, dwb, qk, v, T, H, W, NP);
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
    static const int gen = getenv("IRA_QKV_GEN") ? atoi(getenv("IRA_QKV_GEN")) : 3;
    if (gen == 1) return launch_qkv_fused(st, x, ln_mode, C, Wq, bq, dww, dwb, qk, v, T, H, W);
    if (gen == 3) return launch_qkv3(st, x, ln_mode, C, Wq, bq, dww, dwb, qk, v, T, H, W);
    return launch_qkv2(st, x, ln_mode, C, Wq, bq, dww, dwb, qk, v, T, H, W);
}

}  // namespace ira
