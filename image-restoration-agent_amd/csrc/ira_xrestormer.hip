// gfx950 kernel specific to X-Restormer: the overlapping cross-attention of OCAB
// (X-Restormer/xrestormer/archs/xrestormer_arch.py:190-236 with RelPosEmb :49-82).  Everything else of an X-Restormer block is
// Restormer's (channel attention, GDFN) and runs on the kernels of ira_kernels.hip / ira_ffn_fused.hip / ira_qkv_fused.hip.
//
// One wave per (image, 8x8 query window, head).  Keys / values are the 12x12 window around it (nn.Unfold with stride 8 and
// padding 2, :200): positions outside the image are ZERO vectors that still take part in the softmax (their logit is the
// relative-position term alone).  logits[q][k] = q.k + q.rel_w[xk - xq + 11] + q.rel_h[yk - yq + 11] (q already scaled by
// dim_head^-0.5 through the packed weights; the skew trick of rel_to_abs :24-36 is this index map).
// MFMA mapping as in k_swin_attn: S^T[key][query] with keys as rows (rowperm) and one query per lane; head_dim 16 is exactly
// one k-step of v_mfma_f32_32x32x16_f16.  The 2 x 23 relative logits of a query come from one more MFMA pair (rel tables as
// the A operand, hi + lo planes) and travel through LDS because a lane needs entries that the D layout put in its partner lane.
#include "ira_kernels.h"
#include "ira_device.h"

namespace ira {

namespace {
constexpr int OC_WS = 8, OC_R = 12, OC_NK = OC_R * OC_R, OC_KT = 5;     // 144 keys in 5 tiles of 32
constexpr int OC_VTS = 168;                                              // halfs per d-row of V^T: 160 keys + 8 pad
}

//   qkv fp16 [T][H][W][3*inner], inner = heads*16: q | k | v, head hd at channels hd*16
//   rel fp32 [2][32][16]: rows 0..22 of tile 0 = rel_width, of tile 1 = rel_height (rest zero)
//   out fp16 [T][H][W][inner]
__global__ __launch_bounds__(256) void k_ocab_attn(const f16* __restrict__ qkv, const float* __restrict__ rel,
                                                   f16* __restrict__ out, int T, int H, int W, int heads) {
    __shared__ __attribute__((aligned(16))) f16 vt_all[4][16 * OC_VTS];
    __shared__ __attribute__((aligned(16))) float rl_all[4][32][48];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int nwx = W / OC_WS, nwy = H / OC_WS;
    const long total = (long)T * nwy * nwx * heads;
    long wv = (long)blockIdx.x * 4 + wave;
    const bool live = wv < total;
    if (!live) wv = total - 1;
    const int hd = (int)(wv % heads);
    long rest = wv / heads;
    const int wx = (int)(rest % nwx); rest /= nwx;
    const int wy = (int)(rest % nwy);
    const int t = (int)(rest / nwy);
    const int inner = heads * 16, CS = 3 * inner;
    f16* vt = vt_all[wave];
    float (*rl)[48] = rl_all[wave];
    const int y0 = wy * OC_WS - (OC_R - OC_WS) / 2, x0 = wx * OC_WS - (OC_R - OC_WS) / 2;

    auto kpix = [&](int key, bool& inside) -> long {      // pixel of key `key` of the unfolded 12x12 window
        const int ky = key / OC_R, kx = key - ky * OC_R;
        const int Y = y0 + ky, X = x0 + kx;
        inside = key < OC_NK && Y >= 0 && Y < H && X >= 0 && X < W;
        return ((long)t * H + (inside ? Y : 0)) * W + (inside ? X : 0);
    };

    // V^T into LDS: 160 key slots x 2 chunks of 8 channels (zeros outside the image and for the 16 padding slots)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int c = lane + 64 * i, key = c >> 1, dp = c & 1;
        bool in;
        const long px = kpix(key, in);
        half8 v = zero8();
        if (in) v = *(const half8*)(qkv + px * CS + 2 * inner + hd * 16 + 8 * dp);
#pragma unroll
        for (int j = 0; j < 8; ++j) vt[(8 * dp + j) * OC_VTS + key] = v[j];
    }
    // K as A fragments (rows = keys through rowperm)
    half8 ka[OC_KT];
#pragma unroll
    for (int kt = 0; kt < OC_KT; ++kt) {
        bool in;
        const long px = kpix(32 * kt + rowperm(r), in);
        ka[kt] = in ? *(const half8*)(qkv + px * CS + inner + hd * 16 + 8 * h) : zero8();
    }
    // relative-position tables as A fragments, hi + lo planes (natural row order: they go through LDS)
    half8 ra[2], ral[2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        const float* src = rel + (tl * 32 + r) * 16 + 8 * h;
        const f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
        ra[tl] = cvt8(a, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) { ral[tl][j] = (f16)(a[j] - (float)ra[tl][j]); ral[tl][4 + j] = (f16)(b[j] - (float)ra[tl][4 + j]); }
    }
    __syncthreads();

#pragma unroll 1
    for (int qt = 0; qt < 2; ++qt) {
        const int nq = 32 * qt + r, ty = nq >> 3, tx = nq & 7;
        const long qpx = ((long)t * H + wy * OC_WS + ty) * W + wx * OC_WS + tx;
        const half8 qb = *(const half8*)(qkv + qpx * CS + hd * 16 + 8 * h);
        // 2 x 23 relative logits of every query of this tile -> LDS [query][tile*24 + index]
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            f32x16 d = zero16();
            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ra[tl], qb, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ral[tl], qb, d, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int row = (j & 3) + 8 * (j >> 2) + 4 * h;
                if (row < 24) rl[r][tl * 24 + row] = d[j];
            }
        }
        f32x16 st[OC_KT];
#pragma unroll
        for (int kt = 0; kt < OC_KT; ++kt) st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka[kt], qb, zero16(), 0, 0, 0);
        // the LDS writes above and the reads below belong to the same wave: LDS executes them in order
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < OC_KT; ++kt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k0 = 32 * kt + 16 * u + 8 * h;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int key = k0 + j;
                    float v = -3.0e38f;
                    if (key < OC_NK) {
                        const int ky = key / OC_R, kx = key - ky * OC_R;
                        v = st[kt][8 * u + j] + rl[r][kx - tx + OC_R - 1] + rl[r][24 + ky - ty + OC_R - 1];
                    }
                    st[kt][8 * u + j] = v;
                    mx = fmaxf(mx, v);
                }
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < OC_KT; ++kt)
#pragma unroll
            for (int j = 0; j < 16; ++j) { const float e = __expf(st[kt][j] - mx); st[kt][j] = e; sum += e; }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        // O^T[d][query] = sum_key V^T[d][key] P^T[key][query]; MFMA rows 16..31 are padding (head_dim 16)
        f32x16 o = zero16();
#pragma unroll
        for (int kt = 0; kt < OC_KT; ++kt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                half8 pb;
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[j] = (f16)(st[kt][8 * u + j] * inv);
                const half8 va = (r < 16) ? *(const half8*)(vt + rowperm(r) * OC_VTS + 32 * kt + 16 * u + 8 * h) : zero8();
                o = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, pb, o, 0, 0, 0);
            }
        if (live) {
            half8 ov;
#pragma unroll
            for (int j = 0; j < 8; ++j) ov[j] = (f16)o[j];
            *(half8*)(out + qpx * inner + hd * 16 + 8 * h) = ov;
        }
    }
}

// F.interpolate(scale_factor=s, mode='bilinear', align_corners=False) on fp32 NHWC (the SR preset's first step, xrestormer_arch.py:378-379):
// source coordinate (dst + 0.5) / s - 0.5 clamped at 0, neighbours clamped at the border, as ATen's upsample_bilinear2d does.
__global__ __launch_bounds__(256) void k_bilinear_up(const float* __restrict__ in, float* __restrict__ out, int T, int H, int W, int C, int s) {
    const int Ho = H * s, Wo = W * s;
    const long total = (long)T * Ho * Wo * C;
    const float rs = 1.0f / (float)s;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % C);
        long rest = idx / C;
        const int x = (int)(rest % Wo); rest /= Wo;
        const int y = (int)(rest % Ho);
        const int t = (int)(rest / Ho);
        const float sy = fmaxf(((float)y + 0.5f) * rs - 0.5f, 0.f), sx = fmaxf(((float)x + 0.5f) * rs - 0.5f, 0.f);
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
        const float* b = in + (long)t * H * W * C + ch;
        out[idx] = hy * (hx * b[((long)y0 * W + x0) * C] + lx * b[((long)y0 * W + x1) * C]) +
                   ly * (hx * b[((long)y1 * W + x0) * C] + lx * b[((long)y1 * W + x1) * C]);
    }
}

hipError_t launch_bilinear_up(hipStream_t st, const float* in, float* out, int T, int H, int W, int C, int s) {
    if (s < 1) return hipErrorInvalidValue;
    long g = ((long)T * H * s * W * s * C + 255) / 256;
    if (g > 65536) g = 65536;
    hipLaunchKernelGGL(k_bilinear_up, dim3((unsigned)g), dim3(256), 0, st, in, out, T, H, W, C, s);
    return hipGetLastError();
}

hipError_t launch_ocab_attn(hipStream_t st, const void* qkv, const float* rel, void* out, int T, int H, int W, int heads) {
    if (H % OC_WS != 0 || W % OC_WS != 0 || heads <= 0) return hipErrorInvalidValue;
    const long waves = (long)T * (H / OC_WS) * (W / OC_WS) * heads;
    hipLaunchKernelGGL(k_ocab_attn, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, (const f16*)qkv, rel, (f16*)out, T, H, W, heads);
    return hipGetLastError();
}

}  // namespace ira
