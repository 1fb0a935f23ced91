// gfx950 kernels specific to SwinIR (reference SwinIR/models/network_swinir.py).
//
// Token stream: fp32 NHWC [T][H][W][Cp], Cp = embed_dim rounded up to 32 (180 -> 192, 240 -> 256), padded channels zero.
// One SwinTransformerBlock (:239-279) is scheduled as
//   qkv = Linear(LN1(x))            k_pw_kres      (LN over the true 180/240 channels; gain/bias and the q scale folded;
//                                                   rows laid out per head with head_dim 30 padded to 32: [q|k|v][head][32])
//   a   = window attention          k_swin_attn    (this file: cyclic shift, window partition, q k^T + relative position
//                                                   bias + shift mask, softmax, P v, window reverse, un-shift: one kernel)
//   x  += proj(a)                   k_pw_nres
//   h   = GELU(fc1(LN2(x)))         k_pw_kres      (act = GELU)
//   x  += fc2(h)                    k_pw_nres
// plus LayerNorm alone (patch_embed.norm, the final norm), the mean/range pre/post maps, nearest-neighbour x2 and LeakyReLU.
#include "ira_kernels.h"
#include "ira_device.h"

#include <algorithm>
#include <atomic>

namespace ira {

// ---------------------------------------------------------------------------------------------
// (Shifted-)window multi-head self attention (:114-145 inside :239-279), one wave per (image, window, head).
//   qkv  fp16 [T][H][W][3*heads*32]: q of head hd at channel hd*32, k at (heads+hd)*32, v at (2*heads+hd)*32
//   bias fp32 [heads][64][64]: relative_position_bias_table[relative_position_index] as [query][key] (:92-103,124-127)
//   out  fp16 [T][H][W][heads*32]
// Window token n = ty*WS + tx lives at rolled coordinates (wy*WS+ty, wx*WS+tx), i.e. at source pixel
// ((Y+shift) mod H, (X+shift) mod W) (torch.roll by -shift, :249-252; the reverse roll :268-271 is the same address on the
// way out).  The mask of calculate_mask (:216-237) is recomputed from the rolled coordinates (regions split at H-WS and
// H-shift): -100 where query and key fall in different regions.
// MFMA mapping: S^T[key][query] = K Q^T with the 32 keys of a tile as MFMA rows (loaded through rowperm so that a lane's
// 16 registers are two runs of 8 CONSECUTIVE keys) and 32 queries as columns: every lane owns ONE query, so the softmax
// over keys is a reduction over its own registers + one cross-half shuffle, and the probabilities are already the
// B fragments (k = key) of O^T[d][query] = V^T P^T.  V^T comes from an LDS transpose (4 KB per wave).
// A wave keeps ONE head and walks windows (grid stride a multiple of heads / gcd(heads, 4) workgroups, so wv % heads is the same on every
// trip): its 2 x 32 relative-position-bias values per lane - query r / 32 + r against the lane's 32 keys - stay in 64 registers for the
// whole kernel.  Loaded per window they were 16 KB per wave from L2, more than its q, k and v together, and a quarter of the kernel's time
// (measured by leaving them out: 19.6 -> 14.9 ms per SwinIR-L x4 frame).  V^T is private to the wave: no workgroup barrier anywhere.
template <int WS>
__global__ __launch_bounds__(256, 2) void k_swin_attn(const f16* __restrict__ qkv, const float* __restrict__ bias,
                                                   f16* __restrict__ out, int T, int H, int W, int heads, int shift) {
    constexpr int N = WS * WS;                      // tokens per window (<= 64)
    constexpr int VT_STRIDE = 72;                   // halfs per d-row of V^T: 64 keys + 8 (144 B: 16-byte aligned, conflict-free)
    __shared__ __attribute__((aligned(16))) f16 vt_all[4][32 * VT_STRIDE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int nwx = W / WS, nwy = H / WS;
    const long total = (long)T * nwy * nwx * heads;
    long wv = (long)blockIdx.x * 4 + wave;
    if (wv >= total) return;
    const int hd = (int)(wv % heads);
    const int CS = 3 * heads * 32;
    f16* vt = vt_all[wave];

    // bias[hd][query][key] for this lane's queries and keys: bq[qt][kt][8 u + j] = bias of (query 32 qt + r, key 32 kt + 16 u + 8 h + j)
    f32x16 bq[2][2];
    {
        const float* bh = bias + (long)hd * 64 * 64;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int nq = 32 * qt + r;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float* bp = bh + (long)(nq < N ? nq : 0) * 64 + 32 * kt + 16 * u + 8 * h;
                    const f32x4 b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { bq[qt][kt][8 * u + j] = b0[j]; bq[qt][kt][8 * u + 4 + j] = b1[j]; }
                }
        }
    }

    for (; wv < total; wv += (long)gridDim.x * 4) {
        long rest = wv / heads;
        const int wx = (int)(rest % nwx); rest /= nwx;
        const int wy = (int)(rest % nwy);
        const int t = (int)(rest / nwy);

        auto pix = [&](int n) -> long {                  // source pixel of window token n
            const int ty = n / WS, tx = n - ty * WS;
            int Y = wy * WS + ty + shift, X = wx * WS + tx + shift;
            if (Y >= H) Y -= H;
            if (X >= W) X -= W;
            return ((long)t * H + Y) * W + X;
        };
        auto region = [&](int n) -> int {                // calculate_mask's region id of window token n (rolled coordinates)
            const int ty = n / WS, tx = n - ty * WS;
            const int Y = wy * WS + ty, X = wx * WS + tx;
            const int ry = (Y < H - WS) ? 0 : (Y < H - shift ? 1 : 2);
            const int rx = (X < W - WS) ? 0 : (X < W - shift ? 1 : 2);
            return 3 * ry + rx;
        };

        // V^T into LDS: 64 tokens x 4 chunks of 8 channels
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i, n = c >> 2, dp = c & 3;
            half8 v = zero8();
            if (n < N) v = *(const half8*)(qkv + pix(n) * CS + (2 * heads + hd) * 32 + 8 * dp);
#pragma unroll
            for (int j = 0; j < 8; ++j) vt[(8 * dp + j) * VT_STRIDE + n] = v[j];
        }
        // K as A fragments (rows = keys through rowperm), all 64 keys
        half8 ka[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int n = 32 * kt + rowperm(r);
#pragma unroll
            for (int s = 0; s < 2; ++s)
                ka[kt][s] = (n < N) ? *(const half8*)(qkv + pix(n) * CS + (heads + hd) * 32 + 16 * s + 8 * h) : zero8();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own V^T writes, read back below (no other wave touches them)

#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int nq = 32 * qt + r;
            const bool qvalid = nq < N;
            const long qpx = pix(qvalid ? nq : 0);
            half8 qb[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) qb[s] = qvalid ? *(const half8*)(qkv + qpx * CS + hd * 32 + 16 * s + 8 * h) : zero8();
            f32x16 st[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                st[kt] = zero16();
#pragma unroll
                for (int s = 0; s < 2; ++s) st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka[kt][s], qb[s], st[kt], 0, 0, 0);
            }
            // + relative position bias, + shift mask; keys >= N do not exist
            const int rq = (shift > 0) ? region(qvalid ? nq : 0) : 0;
            float mx = -3.0e38f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int k0 = 32 * kt + 16 * u + 8 * h;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int key = k0 + j;
                        float v = st[kt][8 * u + j] + bq[qt][kt][8 * u + j];
                        if (shift > 0 && key < N && region(key) != rq) v += -100.0f;
                        if (key >= N) v = -3.0e38f;
                        st[kt][8 * u + j] = v;
                        mx = fmaxf(mx, v);
                    }
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int j = 0; j < 16; ++j) { const float e = __expf(st[kt][j] - mx); st[kt][j] = e; sum += e; }
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
            // O^T[d][query] = sum_key V^T[d][key] P^T[key][query]
            f32x16 o = zero16();
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    half8 pb;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pb[j] = (f16)(st[kt][8 * u + j] * inv);
                    const half8 va = *(const half8*)(vt + rowperm(r) * VT_STRIDE + 32 * kt + 16 * u + 8 * h);
                    o = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, pb, o, 0, 0, 0);
                }
            if (qvalid) {
                f16* dst = out + qpx * (heads * 32) + hd * 32 + 8 * h;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    half8 ov;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ov[j] = (f16)o[8 * u + j];
                    *(half8*)(dst + 16 * u) = ov;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this window's V^T reads are done before the next window's writes
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm alone (patch_embed.norm :524-528 with patch_norm, the final self.norm :800): fp32 -> fp32 over the first C of
// Cp channels (padded channels stay zero: gain and bias are zero there).  One wave per pixel.
__global__ __launch_bounds__(256) void k_ln_affine(const float* __restrict__ x, const float* __restrict__ g,
                                                   const float* __restrict__ b, float* __restrict__ y, long npx, int C,
                                                   int Cp, float eps) {
    const int lane = threadIdx.x & 63;
    const long px = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (px >= npx) return;
    const float* src = x + px * Cp;
    float v[8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int c = lane + 64 * i; v[i] = (c < Cp) ? src[c] : 0.f; sum += v[i]; }
    sum = wave_sum(sum);
    const float mu = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int c = lane + 64 * i; if (c < C) { const float d = v[i] - mu; sq += d * d; } }
    sq = wave_sum(sq);
    const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
#pragma unroll
    for (int i = 0; i < 8; ++i) { const int c = lane + 64 * i; if (c < Cp) y[px * Cp + c] = (c < C) ? (v[i] - mu) * rstd * g[c] + b[c] : 0.f; }
}

// check_image_size (reflect pad to a multiple of the window, :783-788) + (x - mean) * img_range (:809-810)
__global__ __launch_bounds__(256) void k_swin_pre(const float* __restrict__ in, float* __restrict__ out, int T, int H, int W,
                                                  int Hp, int Wp, int c, f32x4 mean, float range) {
    const long total = (long)T * Hp * Wp * c;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % c);
        long rest = idx / c;
        int x = (int)(rest % Wp); rest /= Wp;
        int y = (int)(rest % Hp);
        const int t = (int)(rest / Hp);
        if (y >= H) y = 2 * (H - 1) - y;
        if (x >= W) x = 2 * (W - 1) - x;
        out[idx] = (in[(((long)t * H + y) * W + x) * c + ch] - mean[ch]) * range;
    }
}

// x / img_range + mean, crop to (H*s, W*s) (:838-840)
__global__ __launch_bounds__(256) void k_swin_post(const float* __restrict__ in, float* __restrict__ out, int T, int Hi, int Wi,
                                                   int Ho, int Wo, int c, f32x4 mean, float range) {
    const long total = (long)T * Ho * Wo * c;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % c);
        long rest = idx / c;
        const int x = (int)(rest % Wo); rest /= Wo;
        const int y = (int)(rest % Ho);
        const int t = (int)(rest / Ho);
        out[idx] = in[(((long)t * Hi + y) * Wi + x) * c + ch] / range + mean[ch];
    }
}

// F.interpolate(scale_factor=2, mode='nearest') (:830,832) on fp32 NHWC, C a multiple of 4
__global__ __launch_bounds__(256) void k_nearest2x(const float* __restrict__ in, float* __restrict__ out, int T, int H, int W, int C) {
    const int cv = C >> 2;
    const long total = (long)T * 2 * H * 2 * W * cv;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % cv);
        long rest = idx / cv;
        const int x = (int)(rest % (2 * W)); rest /= 2 * W;
        const int y = (int)(rest % (2 * H));
        const int t = (int)(rest / (2 * H));
        ((f32x4*)out)[idx] = ((const f32x4*)in)[(((long)t * H + (y >> 1)) * W + (x >> 1)) * cv + c4];
    }
}

__global__ __launch_bounds__(256) void k_lrelu(float* __restrict__ x, long n4, float slope) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n4; idx += (long)gridDim.x * blockDim.x) {
        f32x4 v = ((f32x4*)x)[idx];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : slope * v[j];
        ((f32x4*)x)[idx] = v;
    }
}

// =============================================================================================
static unsigned grid_cap(long total, int block) {
    long g = (total + block - 1) / block;
    if (g > 65536) g = 65536;
    if (g < 1) g = 1;
    return (unsigned)g;
}

hipError_t launch_swin_attn(hipStream_t st, const void* qkv, const float* bias, void* out, int T, int H, int W, int heads,
                            int ws, int shift) {
    if ((ws != 7 && ws != 8) || H % ws != 0 || W % ws != 0 || shift < 0 || shift >= ws || heads <= 0) return hipErrorInvalidValue;
    const long waves = (long)T * (H / ws) * (W / ws) * heads;
    // a wave walks windows with ONE head: the stride (workgroups x 4 waves) must be a multiple of heads.  As many workgroups as are
    // resident at once (occupancy x CUs), rounded up to that multiple; fewer when the launch is small.
    int gcd = heads, four = 4;
    while (four) { const int tmp = gcd % four; gcd = four; four = tmp; }
    const long mult = heads / gcd;
    static std::atomic<int> occ8{0}, occ7{0}, cus{0};
    std::atomic<int>& occ = ws == 8 ? occ8 : occ7;
    if (occ.load() == 0) {
        int o = 0, dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        const hipError_t e = ws == 8 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, k_swin_attn<8>, 256, 0)
                                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, k_swin_attn<7>, 256, 0);
        if (e != hipSuccess || o < 1) o = 2;
        cus.store(n > 0 ? n : 256);
        occ.store(o);
    }
    long wgs = std::min<long>((waves + 3) / 4, (long)occ.load() * cus.load());
    wgs = (wgs + mult - 1) / mult * mult;
    const dim3 grid((unsigned)wgs), block(256);
    if (ws == 8) hipLaunchKernelGGL((k_swin_attn<8>), grid, block, 0, st, (const f16*)qkv, bias, (f16*)out, T, H, W, heads, shift);
    else hipLaunchKernelGGL((k_swin_attn<7>), grid, block, 0, st, (const f16*)qkv, bias, (f16*)out, T, H, W, heads, shift);
    return hipGetLastError();
}

hipError_t launch_ln_affine(hipStream_t st, const float* x, const float* g, const float* b, float* y, long npx, int C, int Cp,
                            float eps) {
    if (Cp > 512 || C > Cp || C <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_ln_affine, dim3((unsigned)((npx + 3) / 4)), dim3(256), 0, st, x, g, b, y, npx, C, Cp, eps);
    return hipGetLastError();
}

hipError_t launch_swin_pre(hipStream_t st, const float* in, float* out, int T, int H, int W, int Hp, int Wp, int c,
                           const float* mean3, float range) {
    if (c > 4 || Hp - H >= H || Wp - W >= W) return hipErrorInvalidValue;
    f32x4 m = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < c; ++i) m[i] = mean3[i];
    hipLaunchKernelGGL(k_swin_pre, dim3(grid_cap((long)T * Hp * Wp * c, 256)), dim3(256), 0, st, in, out, T, H, W, Hp, Wp, c, m, range);
    return hipGetLastError();
}

hipError_t launch_swin_post(hipStream_t st, const float* in, float* out, int T, int Hi, int Wi, int Ho, int Wo, int c,
                            const float* mean3, float range) {
    if (c > 4) return hipErrorInvalidValue;
    f32x4 m = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < c; ++i) m[i] = mean3[i];
    hipLaunchKernelGGL(k_swin_post, dim3(grid_cap((long)T * Ho * Wo * c, 256)), dim3(256), 0, st, in, out, T, Hi, Wi, Ho, Wo, c, m, range);
    return hipGetLastError();
}

hipError_t launch_nearest2x(hipStream_t st, const float* in, float* out, int T, int H, int W, int C) {
    if (C & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_nearest2x, dim3(grid_cap((long)T * 4 * H * W * (C / 4), 256)), dim3(256), 0, st, in, out, T, H, W, C);
    return hipGetLastError();
}

hipError_t launch_lrelu(hipStream_t st, float* x, long n, float slope) {
    if (n & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_lrelu, dim3(grid_cap(n / 4, 256)), dim3(256), 0, st, x, n / 4, slope);
    return hipGetLastError();
}

}  // namespace ira
