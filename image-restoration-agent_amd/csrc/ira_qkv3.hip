// Fused attention front end, third generation (restormer_arch.py:114 after :146's norm1):
//     qkv = dwconv3x3( conv1x1( LN(x) ) )            q,k -> fp16 pixel-minor [T][H][ceil(W/8)][2C][8], v -> fp16 NHWC
// Same contract, layouts and rounding points as ira_qkv_fused.hip (generation 1).
//
// What the first two measured: a CU's share of HBM bandwidth is ~10 bytes per cycle, a patch moves 277 KB (28k cycles), and
// both generations take 51-64k cycles per patch because ONE workgroup per CU walks through its phases in series: while it waits for its
// halo patch nothing computes, while it computes nothing is in flight, and any wave that issues loads under the main loop stalls AT THE
// ISSUE for as long as the memory pipeline is backed up (generation 2: 13.7k cycles for 32 requests).  The remedy is the ordinary one:
// TWO workgroups per CU, each small and simple, so that one's memory phase lies under the other's arithmetic:
//   * a workgroup = 4 waves = one 8 x 32 output patch (10 x 34 halo = 11 pixel groups); at most 256 registers and 69 KB of LDS, two per CU;
//   * no roles.  Wave w loads the pixels of its groups w, w + 4, w + 8 straight into registers, LayerNorms them (statistics over a lane
//     pair) and keeps them as MFMA A fragments for the whole kernel (pixel-stationary: no LDS round trip, no barrier before the first MFMA);
//   * the 1x1 runs TRANSPOSED (generation 2): D[pixel][channel] = LN(x) . W, the packed weight fragments are the B operand as they are, and
//     with the row permutation applied to a group's PIXELS a lane of the result owns two runs of 8 consecutive pixels of one channel.  The
//     hidden tile goes to LDS as [halo row][channel][pixel] (two ds_write_b128 per lane and group), double buffered, one barrier per chunk;
//   * the depthwise 3x3 runs on lanes = (channel, 8-pixel block): per halo row one ds_read_b128 and two ds_read_b64 (the left halo column
//     sits in the last slot of the previous record, so both neighbours are plain offsets), 72 v_fma_mix_f32 with the lane's nine weights in
//     registers, and the 8 results ARE one 16-byte element of the pixel-minor q,k tensor (one store per lane, 512 consecutive bytes per 32
//     lanes).  v chunks go through a 1.25 KB per-wave LDS transposition and are stored as 16-byte pieces of 8 channels;
//   * every wave runs the chunk's MFMAs for its groups AND four of its sixteen depthwise units; chunk j + 1's MFMAs and chunk j's units
//     are independent instruction streams of the same wave (the vector work issues in the shadow of the MFMAs), and the other workgroup's
//     waves on the same SIMDs fill what is left.
// LDS: hidden buffers 2 x 25 616 B + transposition scratch 4 x 1 280 B + two weight stages of 2 KS KB = 80 928 B at C = 96 (two
// workgroups per CU: 2 x 81 920 B are there), 68 640 B at C = 48.
#include "ira_kernels.h"
#include "ira_device.h"

#include <atomic>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace ira {

namespace {

constexpr int TH = 8, TW = 32, PH = TH + 2, PW = TW + 2;                  // 10 x 34 halo patch
constexpr int NW = 4;                                                       // waves per workgroup
constexpr int NG = 11;                                                      // 8 interior rows + top row + bottom row + the two halo columns
constexpr int RS = 40;                                                      // halfs per (halo row, channel) record: 32 interior pixels, [32] right halo, [39] left halo of the NEXT record
constexpr int REC0 = 8;                                                     // halfs in front of record 0 (so that record 0 has a left-halo slot)
constexpr size_t HPT_BYTES = (size_t)(REC0 + PH * 32 * RS) * 2;             // 25 616 B (a multiple of 16)
constexpr int SCR_STRIDE = 40;                                              // halfs per pixel row of the v transposition scratch
constexpr size_t SCR_BYTES = (size_t)16 * SCR_STRIDE * 2;                   // 1 280 B per wave
constexpr size_t Q3_MAIN = 2 * HPT_BYTES + NW * SCR_BYTES;              // + two weight stages of 2 KS fragments (k_qkv3)
static_assert(HPT_BYTES % 16 == 0, "hidden buffers must keep 16-byte alignment");

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

IRA_DEV unsigned pack2(float lo, float hi) {
    typedef f16 half2v __attribute__((ext_vector_type(2)));
    half2v t; t[0] = (f16)lo; t[1] = (f16)hi;
    return __builtin_bit_cast(unsigned, t);
}

// (halo row, halo column) of pixel p of group g; hy < 0 for the padding of the last group
IRA_DEV void group_px(int g, int p, int& hy, int& hx) {
    if (g < TH) { hy = g + 1; hx = 1 + p; }
    else if (g == TH) { hy = 0; hx = 1 + p; }
    else if (g == TH + 1) { hy = PH - 1; hx = 1 + p; }
    else if (p < PH) { hy = p; hx = 0; }
    else if (p < 2 * PH) { hy = p - PH; hx = PW - 1; }
    else { hy = -1; hx = 0; }
}

template <int N> using IC = std::integral_constant<int, N>;

// acc = (float)half(lo / hi 16 bits of `packed`) * w: the first tap of a depthwise sum without a bias (no accumulator to clear beforehand)
IRA_DEV void mul_mix_lo(float& acc, unsigned packed, float w) {
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(acc) : "v"(packed), "v"(w));
}
IRA_DEV void mul_mix_hi(float& acc, unsigned packed, float w) {
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(acc) : "v"(packed), "v"(w));
}

}  // namespace

#ifdef IRA_KERNEL_DBG
__device__ unsigned long long g_qkv3_stamps[64];        // s_memtime of wave 0 of workgroup 1777 (make dbg; tools/time_qkv.py)
#define Q3STAMP() do { if (blockIdx.x == 1777 && tid == 0 && stamp_n < 64) g_qkv3_stamps[stamp_n] = __builtin_amdgcn_s_memtime(); ++stamp_n; } while (0)
#else
#define Q3STAMP() do {} while (0)
#endif

// DBG (measurement variants, wrong results; instantiated and selectable - IRA_QKV3_DBG - only in the `make dbg` library): 1 = no global
// stores, 2 = no pixel loads, 4 = no MFMAs, 8 = no depthwise FMAs, 16 = no LDS traffic of the hidden tile (bits may be combined)
// LO: the 1x1 weights have a lo plane (fp16 hi + lo, two MFMAs per fragment); false = plain fp16 weights, half the MFMAs and half the staged
// bytes - the handle's default since r3 (ira_set_weight_planes(1); the qkv 1x1 only feeds the Gram / softmax: dropping its lo plane moves the
// worst |dPSNR| of the 27 harness / sweep / bench-tile cases from 5.5e-4 to 5.7e-4 dB, tools/ablate_lo_restormer.py)
template <int C, int LN, bool BIN, bool CB, int DBG = 0, bool LO = true>
__global__ __launch_bounds__(64 * NW, 2) void k_qkv3(const float* __restrict__ x, const f16* __restrict__ Wq, const float* __restrict__ bq,
                                                     const float* __restrict__ dww, const float* __restrict__ dwb, f16* __restrict__ qk,
                                                     f16* __restrict__ vout, int T, int H, int W) {
    constexpr int KS = C / 16, NQ = 3 * C, NCH = (NQ + 31) / 32, NQK = 2 * C / 32;
    static_assert(2 * C % 32 == 0, "a chunk is either q,k or v");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f16* hpb = (f16*)smem;                                                  // [2] hidden buffers, see header
    f16* scr_all = (f16*)(smem + 2 * HPT_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int npy = (H + TH - 1) / TH, npx = (W + TW - 1) / TW;
    const int WB = (W + 7) >> 3;
    const long pid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int t = (int)(pid / ((long)npy * npx));
    const int y0 = (int)((pid / npx) % npy) * TH, x0 = (int)(pid % npx) * TW;
    const float* xt = x + (long)t * H * W * C;
    const bool inside = y0 >= 1 && y0 + TH < H && x0 >= 1 && x0 + TW < W;       // the whole halo patch lies in the image
    const int ngroups = (wave + 2 * NW < NG) ? 3 : 2;
    [[maybe_unused]] int stamp_n = 0;
    Q3STAMP();
    const int chl = rowperm(c);                        // column c of the transposed product is channel 32 j + rowperm(c) (the packed weights' row order)

    // ---- weights of chunk j: its 2 KS MFMA B fragments (hi plane, then lo plane; 1 KB each) travel L2 -> LDS by LDS-DMA, ONCE per workgroup
    // (each wave moves its share), one interval ahead, into stage j & 1; every wave reads them from there, one ds_read_b128 per MFMA.  In
    // registers (one private copy per wave, generation 1 / 2 and the first form of this kernel) they were 48 registers per set, four times
    // the L2 traffic, and with a single set the request could only be issued behind the last MFMA: its latency (~5k cycles under load) lay open.
    constexpr int NF = LO ? 2 * KS : KS;
    constexpr size_t WST = (size_t)NF * 1024;
    char* wst = smem + Q3_MAIN;                        // [2][NF][64 lanes][16 B]
    float bias1[2];
    auto stage_w = [&](int j) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < (NF + NW - 1) / NW; ++i) {
            const int f = wave + NW * i;
            if (f < NF) {
                const int plane = f >= KS ? 1 : 0, sk = f - plane * KS;
                const char* src = (const char*)Wq + ((long)plane * NCH * KS + (long)j * KS + sk) * 1024 + lane * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(wst + (size_t)(j & 1) * WST + f * 1024), 16, 0, 0);
            }
        }
    };
    auto load_b1 = [&](int j, auto SETc) __attribute__((always_inline)) {
        constexpr int set = decltype(SETc)::value;
        const int ch = 32 * j + chl;
        const int chc = ch < NQ ? ch : NQ - 1;
        bias1[set] = (BIN && ch < NQ) ? bq[chc] : 0.f;
    };
    float wv[2][9], biasd[2];
    auto load_dw = [&](int j, auto SETc) __attribute__((always_inline)) {
        constexpr int set = decltype(SETc)::value;
        const int ch = 32 * j + chl;
        const int chc = ch < NQ ? ch : NQ - 1;
#pragma unroll
        for (int k = 0; k < 9; ++k) { const float w = dww[(long)k * NQ + chc]; wv[set][k] = ch < NQ ? w : 0.f; }
        biasd[set] = (CB && ch < NQ) ? dwb[chc] : 0.f;
    };

    // ---- prologue: this wave's pixels -> LayerNorm -> A fragments (A row c of a group holds pixel rowperm(c)).
    // A group's 32 pixels (C floats each) are fetched as WHOLE 128-byte lines - instruction i of a group: lanes 8k .. 8k+7 = line i / 4 of pixel
    // 8 (i % 4) + k, 8 lines per instruction - parked in this wave's 12.5 KB of the (still idle) hidden-buffer region with a pixel stride of
    // C + 4 floats, and read back in fragment shape.  Fetched in fragment shape straight from memory (16 bytes per lane at the pixel stride:
    // 32 lines per instruction, each touched by four instructions) the texture path needed 8.9k cycles just to take the 24 requests.
    half8 xa[3][KS];
    {
        constexpr int XS = C + 4;                      // staging stride in floats: conflict-free 16-byte reads at pixel granularity
        constexpr int PP = C / 4;                      // 16-byte pieces per pixel
        constexpr int NSG = (PP + 7) / 8;              // 128-byte segments per pixel (C = 48: the second one is half used)
        constexpr int NLD = 4 * NSG;                   // requests per group
        static_assert((size_t)NW * 32 * XS * 4 <= 2 * HPT_BYTES, "pixel staging must fit the hidden-buffer region");
        float* stg = (float*)smem + (size_t)wave * 32 * XS;
        f32x4 raw[2][NLD];
        // lane part of a request's address (floats): pixel lane / 8 of the instruction's pixel octet, piece lane % 8 of the segment (the last
        // segment of a C = 48 pixel is half used: its upper lanes repeat piece PP - 1 and are not written to the staging area)
        const int voff = (lane >> 3) * C + 4 * (lane & 7);
        const int pcl = 8 * (NSG - 1) + (lane & 7);
        const int voff_last = (lane >> 3) * C + 4 * (pcl < PP ? pcl : PP - 1);
        auto request = [&](int gi, auto SETc) __attribute__((always_inline)) {
            constexpr int set = decltype(SETc)::value;
            const int g = wave + NW * gi;
            if (inside && g < NG - 1) {
                // a halo row's 32 interior pixels of a patch inside the image: a wave-uniform base + the lane part, no clamping
                const int hy = g < TH ? g + 1 : (g == TH ? 0 : PH - 1);
                const float* gb = xt + ((long)(y0 - 1 + hy) * W + x0) * C;
#pragma unroll
                for (int i = 0; i < NLD; ++i) raw[set][i] = (DBG & 2) ? f32x4{0.1f * (lane & 7), 0.3f, -0.2f, 0.05f * i}
                                                                      : *(const f32x4*)(gb + 8 * (i % 4) * C + ((i / 4 == NSG - 1) ? voff_last : 32 * (i / 4) + voff));
                return;
            }
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int p = 8 * (i % 4) + (lane >> 3);
                int hy, hx;
                group_px(g, p, hy, hx);
                int yy = y0 - 1 + (hy < 0 ? 0 : hy), xx = x0 - 1 + hx;
                yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);                  // pixels outside the image: any finite values (their hidden values are zeroed)
                xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
                const int pc = 8 * (i / 4) + (lane & 7);
                if constexpr (DBG & 2) { raw[set][i] = f32x4{0.1f * (lane & 7), 0.3f, -0.2f, 0.05f * i}; (void)yy; (void)xx; (void)pc; }
                else raw[set][i] = *(const f32x4*)(xt + ((long)yy * W + xx) * C + 4 * (pc < PP ? pc : PP - 1));
            }
        };
        auto normalise = [&](int gi, auto SETc) __attribute__((always_inline)) {
            constexpr int set = decltype(SETc)::value;
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int pc = 8 * (i / 4) + (lane & 7);
                if (8 * (i / 4) + 8 <= PP || pc < PP) *(f32x4*)(stg + (8 * (i % 4) + (lane >> 3)) * XS + 4 * pc) = raw[set][i];
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();           // (the reads below take what other lanes of this wave wrote: LDS operations of a wave execute in order)
            const float* src = stg + chl * XS + 8 * h;
            f32x4 xlo[KS], xhi[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) { xlo[s] = *(const f32x4*)(src + 16 * s); xhi[s] = *(const f32x4*)(src + 16 * s + 4); }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            float sum = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int j = 0; j < 4; ++j) sum += xlo[s][j];
#pragma unroll
                for (int j = 0; j < 4; ++j) sum += xhi[s][j];
            }
            sum += __shfl_xor(sum, 32);
            const float mu = sum * (1.0f / C);
            float sq = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = xlo[s][j] - mu; sq += d * d; }
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = xhi[s][j] - mu; sq += d * d; }
            }
            sq += __shfl_xor(sq, 32);
            const float rstd = 1.0f / sqrtf(sq * (1.0f / C) + 1e-5f);
            const float sub = (LN == 2) ? mu : 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                half8 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) { o[j] = (f16)((xlo[s][j] - sub) * rstd); o[4 + j] = (f16)((xhi[s][j] - sub) * rstd); }
                xa[gi][s] = o;
            }
        };
        stage_w(0);
        load_b1(0, IC<0>{});
        request(0, IC<0>{}); request(1, IC<1>{});
        Q3STAMP();
        normalise(0, IC<0>{});
        Q3STAMP();
        if (ngroups > 2) request(2, IC<0>{});
        normalise(1, IC<1>{});
        if (ngroups > 2) normalise(2, IC<0>{});
        Q3STAMP();
    }

    // wave 3 has two groups only; it runs a third, discarded one (a copy of its second) rather than branch around twelve MFMAs: the main loop
    // is then the same straight-line code for every wave (only the few stores behind a group's MFMAs differ)
    if (ngroups < 3) {
#pragma unroll
        for (int s = 0; s < KS; ++s) xa[2][s] = xa[1][s];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of chunk 0's weights has landed (the pixels were waited for anyway)
    lds_barrier();                                     // ... and everybody else's

    // ---- main loop.  Interval j = chunk j's 1x1 (into hidden buffer j & 1) + chunk j-1's depthwise units (out of buffer (j-1) & 1), one barrier
    // per interval.  The two are independent instruction streams of the same wave and are INTERLEAVED BY HAND: a group's 2 KS MFMAs pair with
    // the three halo rows of one depthwise unit, 2 KS / 3 MFMAs per row, each followed by its share of the row's 24 v_fma_mix (6 at C = 96): the
    // vector instructions issue in the shadow of the MFMAs (an MFMA holds the SIMD's issue for 8 of its 32 cycles) instead of after them.
    // A row's three LDS reads are requested one row ahead.  The fourth unit has no MFMAs beside it; the next chunk's weight fragments are
    // requested in front of it, into the registers the last MFMA has just read (one weight set: a second one meant spills at C = 96).
    struct Row { u32x4 m; u32x2 l, r; };
    f16* scr = scr_all + (size_t)wave * (SCR_BYTES / 2);
    const int lofs = c * RS + 8 * h;                   // this lane's slot inside a halo row's block of records (halfs)
#define Q3_PIN() __builtin_amdgcn_sched_barrier(0)

    auto row_request = [&](const f16* hp, int ui, int dy) __attribute__((always_inline)) -> Row {
        const int ry = 2 * wave + (ui >> 1), b = 2 * (ui & 1) + h;
        const f16* rec = hp + REC0 + ((ry + dy) * 32 + c) * RS + 8 * b;
        Row rw;
        // (records are 80 bytes apart and start 16 bytes into the buffer: rec is 16-byte aligned, its neighbours' words 8-byte aligned; told
        // explicitly, because with 4-byte alignment the compiler reads them as ds_read2_b32, twice the LDS cycles of a ds_read_b64)
        rw.m = *(const u32x4*)__builtin_assume_aligned(rec, 16);
        rw.l = *(const u32x2*)__builtin_assume_aligned(rec - 4, 8);
        rw.r = *(const u32x2*)__builtin_assume_aligned(rec + 8, 8);
        return rw;
    };
    // FMAs n0 .. n0+cnt-1 of a row's 24 (n = 8 dx + p): element e = p + dx - 1 of the row segment: -1 = l[1] high half, 0..7 = m, 8 = r[0] low half
    auto row_fma = [&](float (&acc)[8], const Row& rw, const float (&w)[9], int dy, int n0, int cnt) __attribute__((always_inline)) {
#pragma unroll
        for (int n = n0; n < n0 + cnt; ++n) {
            const int dx = n >> 3, p = n & 7, e = p + dx - 1;
            const float wt = w[dy * 3 + dx];
            if (!CB && dy == 0 && dx == 0) {                                   // first tap, no bias: a product (e = p - 1: never 8)
                if (e < 0) mul_mix_hi(acc[p], rw.l[1], wt);
                else if (e & 1) mul_mix_hi(acc[p], rw.m[e >> 1], wt);
                else mul_mix_lo(acc[p], rw.m[e >> 1], wt);
            }
            else if (e < 0) fma_mix_hi(acc[p], rw.l[1], wt);
            else if (e == 8) fma_mix_lo(acc[p], rw.r[0], wt);
            else if (e & 1) fma_mix_hi(acc[p], rw.m[e >> 1], wt);
            else fma_mix_lo(acc[p], rw.m[e >> 1], wt);
        }
    };
    // a finished unit: q,k as one pixel-minor element per lane, v through the per-wave transposition
    auto unit_store = [&](int j, int ui, const float (&acc)[8], auto INc) __attribute__((always_inline)) {
        constexpr bool in = decltype(INc)::value;
        const int ry = 2 * wave + (ui >> 1), b = 2 * (ui & 1) + h;
        const int oy = y0 + ry, ox = x0 + 8 * b;
        if (j < NQK) {
            // rows of the pixel-minor tensor are padded to WB * 8 pixels and the padding must read as zero for the Gram: store every block
            // below WB * 8, zeros for x >= W
            if (in || (oy < H && ox < WB * 8)) {
                u32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float lo = (in || ox + 2 * k < W) ? acc[2 * k] : 0.f, hi = (in || ox + 2 * k + 1 < W) ? acc[2 * k + 1] : 0.f;
                    o[k] = pack2(lo, hi);
                }
                if constexpr (DBG & 1) asm volatile("" :: "v"(o));
                else *(u32x4*)(qk + ((((long)t * H + oy) * WB + (ox >> 3)) * (2 * C) + 32 * j + chl) * 8) = o;
            }
        } else {
#pragma unroll
            for (int p = 0; p < 8; ++p) scr[(8 * h + p) * SCR_STRIDE + chl] = (f16)acc[p];
            // the read below takes what OTHER lanes of this wave wrote: LDS operations of one wave execute in order, the compiler must
            // not move the read above the writes
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int p16 = lane >> 2, o8 = lane & 3;
            const u32x4 o = *(const u32x4*)(scr + p16 * SCR_STRIDE + 8 * o8);
            const int vx = x0 + 16 * (ui & 1) + p16, vch = 32 * j - 2 * C + 8 * o8;
            if constexpr (DBG & 1) asm volatile("" :: "v"(o));
            else if ((in || (oy < H && vx < W)) && vch < C) *(u32x4*)(vout + (((long)t * H + oy) * W + vx) * C + vch) = o;
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    };
    // a finished group of the 1x1 -> hidden buffer (zeros for pixels outside the image: the zero padding of the depthwise conv)
    auto group_store = [&](f16* hp, int gi, const f32x16& acc, auto INc) __attribute__((always_inline)) {
        constexpr bool in = decltype(INc)::value;
        const int g = wave + NW * gi;
        if (g < NG - 1) {
            // a halo row's 32 interior pixels: this lane owns pixels 8h .. 8h+7 (registers 0-7) and 16+8h .. 16+8h+7 (8-15) of its channel
            const int hy = g < TH ? g + 1 : (g == TH ? 0 : PH - 1);
            u32x4 o1, o2;
#pragma unroll
            for (int k = 0; k < 4; ++k) { o1[k] = pack2(acc[2 * k], acc[2 * k + 1]); o2[k] = pack2(acc[8 + 2 * k], acc[8 + 2 * k + 1]); }
            if constexpr (!in) {
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
        } else if (g == NG - 1) {
            // the two halo columns.  Group pixel p < 10 = left column (slot -1 of the record) of halo row p, p < 20 = right column (slot 32)
            // of halo row p - 10; a lane holds p = 8h + e (registers e = 0..7) and 16 + 8h + e (8..15):
            //   e = 0, 1:  h = 0: left rows 0, 1          h = 1: left rows 8, 9
            //   e = 2..7:  h = 0: left rows 2..7          h = 1: right rows 0..5
            //   e = 8..11: h = 0: right rows 6..9         h = 1: nothing (p >= 24)
            constexpr int ROWH = 32 * RS;                  // halfs per halo row
            f16* col = hp + REC0 + (lofs - 8 * h);         // record (halo row 0, this lane's channel)
            f16* bA = col - 1 + (h ? 8 * ROWH : 0);
            f16* bB = h ? col + 32 - 2 * ROWH : col - 1;
            f16* bC = col + 32 + 6 * ROWH;
            auto val = [&](int e, int hy, bool right) __attribute__((always_inline)) -> f16 {
                const int yy = y0 - 1 + hy, xx = right ? x0 + TW : x0 - 1;
                const bool ok = in || (yy >= 0 && yy < H && xx >= 0 && xx < W);
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
        }                                                  // (g == NG: wave 3's discarded group)
    };

    // one interval; PR: it has a 1x1 (chunk j), UN: it has depthwise units (chunk j-1); PAR = (j - 1) & 1 = the depthwise weight set of the units
    auto interval = [&](int j, auto PRc, auto UNc, auto PARc, auto INc) __attribute__((always_inline)) {
        constexpr bool pr = decltype(PRc)::value, un = decltype(UNc)::value;
        constexpr int par = decltype(PARc)::value;     // (j - 1) & 1: the units' depthwise weight set, the next chunk's weight stage and bias slot
        constexpr bool in = decltype(INc)::value;
        f16* hpw = hpb + (size_t)(j & 1) * (HPT_BYTES / 2);
        const f16* hpr = hpb + (size_t)((j - 1) & 1) * (HPT_BYTES / 2);
        const char* wcur = wst + (size_t)(j & 1) * WST + lane * 16;
        if constexpr (pr) {
            load_dw(j, IC<par ^ 1>{});                                          // chunk j's depthwise weights: used by the NEXT interval's units
            if (j + 1 < NCH) { stage_w(j + 1); load_b1(j + 1, IC<par>{}); }     // chunk j + 1's weights -> stage (j + 1) & 1 (read last in interval j - 1)
        }
        // FRAGMENT-MAJOR order: a weight fragment is read from LDS once and used by the three groups' MFMAs (three accumulators live); the
        // interval's 3 NF MFMAs are paired with the 288 v_fma_mix of the four depthwise units (12 halo rows of 24), FPM per MFMA, in order.
        constexpr int NM = 3 * NF, FPM2 = 288 / NM;
        static_assert(288 % NM == 0, "FMAs per MFMA");
        f32x16 acc[3];
        float ua[8];
        Row cur{}, nxt{};
        half8 bw{}, bwn{};
        if constexpr (pr) {
#pragma unroll
            for (int gi = 0; gi < 3; ++gi)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[gi][e] = bias1[par ^ 1];
            bw = *(const half8*)wcur;
        }
        if constexpr (un) cur = row_request(hpr, 0, 0);
        auto fma_slot = [&](auto Qc) __attribute__((always_inline)) {            // FMAs Q * FPM2 .. of the interval's 288
            constexpr int q = decltype(Qc)::value;
#pragma unroll
            for (int n = q * FPM2; n < (q + 1) * FPM2; ++n) {
                const int rowi = n / 24, k = n - rowi * 24, ui = rowi / 3, dy = rowi - ui * 3;
                if (k == 0) {
                    if (rowi > 0) cur = nxt;
                    if (rowi + 1 < 12) nxt = row_request(hpr, (rowi + 1) / 3, (rowi + 1) % 3);                    // one row ahead
                    if (CB && dy == 0) {
#pragma unroll
                        for (int p = 0; p < 8; ++p) ua[p] = biasd[par];
                    }
                }
                if constexpr (DBG & 8) asm volatile("" :: "v"(cur.m), "v"(cur.l), "v"(cur.r));
                else row_fma(ua, cur, wv[par], dy, k, 1);
                if (k == 23 && dy == 2) unit_store(j - 1, ui, ua, INc);
            }
        };
        [&]<int... Q>(std::integer_sequence<int, Q...>) __attribute__((always_inline)) {
            ([&] {
                constexpr int q = Q, ms = q / 3, gi = q % 3, sk = LO ? ms >> 1 : ms;       // fragment ms: k-step sk, plane ms & 1 (hi, then lo: the order of the other generations)
                if constexpr (pr) {
                    if constexpr (gi == 0 && ms + 1 < NF) { constexpr int msn = ms + 1, fn = LO ? (msn & 1) * KS + (msn >> 1) : msn; bwn = *(const half8*)(wcur + fn * 1024); }
                    if constexpr (DBG & 4) asm volatile("" :: "v"(xa[gi][sk]), "v"(bw));
                    else acc[gi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[gi][sk], bw, acc[gi], 0, 0, 0);
                    if constexpr (gi == 2) bw = bwn;
                }
                if constexpr (un) fma_slot(IC<q>{});
                Q3_PIN();
            }(), ...);
        }(std::make_integer_sequence<int, NM>{});
        if constexpr (pr) {
#pragma unroll
            for (int gi = 0; gi < 3; ++gi) group_store(hpw, gi, acc[gi], INc);
        }
        Q3STAMP();
        if constexpr (pr) {
            // the DMA pieces of chunk j + 1 are older than this interval's stores on the wave's vector-memory counter: inside the image the four
            // unit stores are unconditional, so vmcnt(4) retires the pieces and leaves the stores in flight
            if (in && un && !(DBG & 1)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
        }
        Q3STAMP();
    };
    auto run = [&](auto INc) __attribute__((always_inline)) {
        using Tt = std::true_type; using Ff = std::false_type;
        interval(0, Tt{}, Ff{}, IC<1>{}, INc);             // chunk 0's 1x1 alone (its depthwise weights go to set 0, its bias came in slot 0)
        static_assert((NCH - 1) % 2 == 0, "steady intervals come in (odd, even) pairs");
#pragma unroll 1
        for (int j = 1; j < NCH; j += 2) {
            interval(j, Tt{}, Tt{}, IC<0>{}, INc);         // units of chunk j - 1 (even): weight set 0
            interval(j + 1, Tt{}, Tt{}, IC<1>{}, INc);
        }
        interval(NCH, Ff{}, Tt{}, IC<(NCH - 1) & 1>{}, INc);
    };
    if (inside) run(std::true_type{});
    else run(std::false_type{});
#undef Q3_PIN
}

#ifdef IRA_KERNEL_DBG
extern "C" int ira_debug_qkv3_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qkv3_stamps), sizeof(unsigned long long) * 64);
}
#endif

bool qkv3_supported(int C) { return C == 48 || C == 96; }

template <int C, int LN, bool BIN, bool CB, bool LO>
static hipError_t qkv3_launch3(hipStream_t st, const float* x, const f16* Wq, const float* bq, const float* dww, const float* dwb, f16* qk,
                               f16* v, int T, int H, int W) {
    const long nblk = (long)T * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    constexpr size_t LDS = Q3_MAIN + 2 * ((LO ? 2 : 1) * (C / 16)) * 1024;
#ifdef IRA_KERNEL_DBG
    if constexpr (C == 96 && LN == 1 && !BIN && !CB && LO) {
        static const int dbg = getenv("IRA_QKV3_DBG") ? atoi(getenv("IRA_QKV3_DBG")) : 0;
#define IRA_Q3_DBG(D_)                                                                                                                   \
    if (dbg == D_) {                                                                                                                      \
        static std::atomic<uint64_t> ad{0};                                                                                               \
        if (hipError_t e = ensure_dynamic_lds((const void*)k_qkv3<C, LN, BIN, CB, D_>, LDS, ad); e != hipSuccess) return e;               \
        hipLaunchKernelGGL((k_qkv3<C, LN, BIN, CB, D_>), dim3((unsigned)nblk), dim3(64 * NW), LDS, st, x, Wq, bq, dww, dwb, qk, v, T, H, W); \
        return hipGetLastError();                                                                                                         \
    }
        IRA_Q3_DBG(1) IRA_Q3_DBG(2) IRA_Q3_DBG(3) IRA_Q3_DBG(4) IRA_Q3_DBG(8) IRA_Q3_DBG(12) IRA_Q3_DBG(15)
#undef IRA_Q3_DBG
    }
#endif
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = ensure_dynamic_lds((const void*)k_qkv3<C, LN, BIN, CB, 0, LO>, LDS, attr_done); e != hipSuccess) return e;
    hipLaunchKernelGGL((k_qkv3<C, LN, BIN, CB, 0, LO>), dim3((unsigned)nblk), dim3(64 * NW), LDS, st, x, Wq, bq, dww, dwb, qk, v, T, H, W);
    return hipGetLastError();
}

template <int C, int LN, bool BIN, bool CB>
static hipError_t qkv3_launch2(hipStream_t st, const float* x, const f16* Wq, const float* bq, const float* dww, const float* dwb, f16* qk,
                               f16* v, int T, int H, int W, int planes) {
    if (planes == 1) return qkv3_launch3<C, LN, BIN, CB, false>(st, x, Wq, bq, dww, dwb, qk, v, T, H, W);
    return qkv3_launch3<C, LN, BIN, CB, true>(st, x, Wq, bq, dww, dwb, qk, v, T, H, W);
}

template <int C>
static hipError_t qkv3_launch(hipStream_t st, const float* x, int ln_mode, const f16* Wq, const float* bq, const float* dww, const float* dwb,
                              f16* qk, f16* v, int T, int H, int W, int planes) {
#define IRA_QKV3(LN_, BIN_, CB_) return qkv3_launch2<C, LN_, BIN_, CB_>(st, x, Wq, bq, dww, dwb, qk, v, T, H, W, planes)
    if (ln_mode == 1) {
        if (bq) { if (dwb) IRA_QKV3(1, true, true); else IRA_QKV3(1, true, false); }
        else { if (dwb) IRA_QKV3(1, false, true); else IRA_QKV3(1, false, false); }
    } else {
        if (bq) { if (dwb) IRA_QKV3(2, true, true); else IRA_QKV3(2, true, false); }
        else { if (dwb) IRA_QKV3(2, false, true); else IRA_QKV3(2, false, false); }
    }
#undef IRA_QKV3
}

hipError_t launch_qkv3(hipStream_t st, const float* x, int ln_mode, int C, const void* Wq, const float* bq, const float* dww,
                       const float* dwb, void* qk, void* v, int T, int H, int W, int planes) {
    if ((ln_mode != 1 && ln_mode != 2) || !qkv3_supported(C) || (planes != 1 && planes != 2)) return hipErrorInvalidValue;
    if (C == 48) return qkv3_launch<48>(st, x, ln_mode, (const f16*)Wq, bq, dww, dwb, (f16*)qk, (f16*)v, T, H, W, planes);
    return qkv3_launch<96>(st, x, ln_mode, (const f16*)Wq, bq, dww, dwb, (f16*)qk, (f16*)v, T, H, W, planes);
}

// The network schedule's entry: generation 3 unless IRA_QKV_GEN=1 (A/B on the same box; read once per process).
hipError_t launch_qkv_front(hipStream_t st, const float* x, int ln_mode, int C, const void* Wq, const float* bq, const float* dww,
                            const float* dwb, void* qk, void* v, int T, int H, int W, int planes) {
    static const int gen = getenv("IRA_QKV_GEN") ? atoi(getenv("IRA_QKV_GEN")) : 3;
    if (gen == 1) return launch_qkv_fused(st, x, ln_mode, C, Wq, bq, dww, dwb, qk, v, T, H, W);
    return launch_qkv3(st, x, ln_mode, C, Wq, bq, dww, dwb, qk, v, T, H, W, planes);
}

}  // namespace ira
