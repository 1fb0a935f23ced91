// gfx950 kernels specific to MPRNet (reference MPRNet/Deblurring/MPRNet.py): everything that is not a dense 3x3
// or 1x1 convolution (those run on k_conv3x3 / k_naf_pw with the PReLU as the LeakyReLU epilogue).  fp32 NHWC throughout, channel
// counts padded to a multiple of 16 (padded channels are zero and stay zero).
#include "ira_kernels.h"
#include "ira_device.h"

#include <algorithm>

namespace ira {

// CALayer (:21-37), step 1: per-workgroup partial channel sums of a fp32 NHWC tensor.  part [T][bpi][Cp]; fixed reduction order.
__global__ __launch_bounds__(256) void k_mpr_chan_partial(const float* __restrict__ x, float* __restrict__ part, int P, int Cp, int bpi,
                                                          int px_per_block) {
    const int t = blockIdx.x / bpi, blk = blockIdx.x % bpi;
    const int p0 = blk * px_per_block, p1 = min(P, p0 + px_per_block);
    const int cv = Cp >> 2;                                   // float4 columns
    __shared__ f32x4 sm[256];
    const int col = threadIdx.x % cv, row = threadIdx.x / cv, rows = 256 / cv;    // cv <= 64 (Cp <= 256)
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (row < rows)
        for (int p = p0 + row; p < p1; p += rows) a += *(const f32x4*)(x + ((long)t * P + p) * Cp + 4 * col);
    sm[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < cv) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < rows; ++r) s += sm[r * cv + threadIdx.x];
        *(f32x4*)(part + ((long)t * bpi + blk) * Cp + 4 * threadIdx.x) = s;
    }
}

// CALayer step 2: mean -> 1x1 (C -> C/r) -> ReLU -> 1x1 (C/r -> C) -> sigmoid.  One workgroup of 1024 threads per image: the bpi partial
// records are summed by 1024 / Cp thread groups over fixed slices (four fixed chains each), then across groups in group order.  z [T][Cp].
__global__ __launch_bounds__(1024) void k_mpr_ca(const float* __restrict__ part, int bpi, float inv_px, const float* __restrict__ w1,
                                                 const float* __restrict__ w2, float* __restrict__ z, int C, int Cp, int Cr) {
    extern __shared__ float lds[];                             // mean[Cp] + hid[Cr] + red[G][Cp]
    float* mean = lds;
    float* hid = lds + Cp;
    float* red = hid + Cr;
    const int t = blockIdx.x, tid = threadIdx.x;
    const int G = 1024 / Cp, g = tid / Cp, c = tid - g * Cp;
    if (g < G) {
        const int per = (bpi + G - 1) / G, b0 = g * per, b1 = min(bpi, b0 + per);
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        const float* pp = part + (long)t * bpi * Cp + c;
        int b = b0;
        for (; b + 4 <= b1; b += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] += pp[(long)(b + u) * Cp];
        }
        for (; b < b1; ++b) a[0] += pp[(long)b * Cp];
        red[g * Cp + c] = (a[0] + a[1]) + (a[2] + a[3]);
    }
    __syncthreads();
    for (int cc = tid; cc < Cp; cc += 1024) {
        float a = 0.f;
        for (int gg = 0; gg < G; ++gg) a += red[gg * Cp + cc];
        mean[cc] = a * inv_px;
    }
    __syncthreads();
    for (int j = tid; j < Cr; j += 1024) {
        float a = 0.f;
        for (int cc = 0; cc < C; ++cc) a = fmaf(w1[(long)j * C + cc], mean[cc], a);
        hid[j] = fmaxf(a, 0.f);
    }
    __syncthreads();
    for (int cc = tid; cc < Cp; cc += 1024) {
        float a = 0.f;
        if (cc < C) {
            for (int j = 0; j < Cr; ++j) a = fmaf(w2[(long)cc * Cr + j], hid[j], a);
            a = 1.0f / (1.0f + __expf(-a));
        }
        z[(long)t * Cp + cc] = a;
    }
}

// CAB tail (:55-57): out = res * z[t] + x
__global__ __launch_bounds__(256) void k_mpr_scale_add(const float* __restrict__ res, const float* __restrict__ z, const float* __restrict__ x,
                                                       float* __restrict__ out, long P, int Cp, long total4) {
    const int cv = Cp >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long px = i / cv;
        const int c4 = (int)(i - px * cv);
        const long t = px / P;
        const f32x4 r = ((const f32x4*)res)[i], s = *(const f32x4*)(z + t * Cp + 4 * c4), a = ((const f32x4*)x)[i];
        ((f32x4*)out)[i] = r * s + a;
    }
}

// SAM tail (:68-74, kernel_size 1): img = conv2(x) + x_img (y8: conv2's 3 outputs in rows of 8 floats), out = x1 * sigmoid(conv3(img)) + x,
// conv3 = [C][3]
__global__ __launch_bounds__(256) void k_mpr_sam(const float* __restrict__ x1, const float* __restrict__ y8, const float* __restrict__ ximg,
                                                 const float* __restrict__ w3, const float* __restrict__ x, float* __restrict__ out,
                                                 float* __restrict__ img_out, int C, int Cp, long npx) {
    const long total = npx * Cp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long px = i / Cp;
        const int c = (int)(i - px * Cp);
        const float i0 = y8[px * 8] + ximg[px * 3], i1 = y8[px * 8 + 1] + ximg[px * 3 + 1], i2 = y8[px * 8 + 2] + ximg[px * 3 + 2];
        if (c == 0) { img_out[px * 3] = i0; img_out[px * 3 + 1] = i1; img_out[px * 3 + 2] = i2; }
        float v = 0.f;
        if (c < C) {
            const float a = w3[c * 3] * i0 + w3[c * 3 + 1] * i1 + w3[c * 3 + 2] * i2;
            v = x1[i] * (1.0f / (1.0f + __expf(-a))) + x[i];
        }
        out[i] = v;
    }
}

// left | right split (:268-277) and its inverse, the feature concatenation along the width (:294-295): wide [N][h][2w][C] <-> halves
// [2N][h][w][C], half image 2n + lr.  (The top / bottom split and concatenation are free: consecutive images of half the height.)
__global__ __launch_bounds__(256) void k_mpr_lr(float* __restrict__ wide, float* __restrict__ halves, int N, int h, int w, int C, int merge) {
    const long total = (long)N * h * 2 * w * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long r = i / C;
        const int x = (int)(r % (2 * w)); r /= 2 * w;
        const int y = (int)(r % h);
        const int n = (int)(r / h);
        const int lr = x >= w ? 1 : 0;
        const long j = ((((long)(2 * n + lr)) * h + y) * w + (x - lr * w)) * C + c;
        if (merge) wide[i] = halves[j]; else halves[j] = wide[i];
    }
}

// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False) (:166-184) (+ add: SkipUpSample's x + y)
__global__ __launch_bounds__(256) void k_mpr_up2(const float* __restrict__ in, const float* __restrict__ add, float* __restrict__ out, int T,
                                                 int H, int W, int C) {
    const int Ho = 2 * H, Wo = 2 * W, cv = C >> 2;
    const long total = (long)T * Ho * Wo * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % cv);
        long r = i / cv;
        const int x = (int)(r % Wo); r /= Wo;
        const int y = (int)(r % Ho);
        const int t = (int)(r / Ho);
        const float sy = fmaxf(((float)y + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf(((float)x + 0.5f) * 0.5f - 0.5f, 0.f);
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
        const f32x4* b = (const f32x4*)in + (long)t * H * W * cv + c4;
        f32x4 v = hy * (hx * b[((long)y0 * W + x0) * cv] + lx * b[((long)y0 * W + x1) * cv]) +
                  ly * (hx * b[((long)y1 * W + x0) * cv] + lx * b[((long)y1 * W + x1) * cv]);
        if (add != nullptr) v += ((const f32x4*)add)[i];
        ((f32x4*)out)[i] = v;
    }
}

static unsigned mpr_grid(long total, int block = 256) {
    long g = (total + block - 1) / block;
    if (g > 65536) g = 65536;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// z[T][Cp] of CALayer; part must hold T * mpr_ca_blocks(P) * Cp floats
int mpr_ca_blocks(long P) { return (int)std::min<long>(2048, (P + 255) / 256); }

hipError_t launch_mpr_ca(hipStream_t st, const float* x, float* part, const float* w1, const float* w2, float* z, int T, long P, int C, int Cp,
                         int Cr) {
    if (Cp % 16 != 0 || Cp > 256 || C > Cp || Cr < 1) return hipErrorInvalidValue;
    const int bpi = mpr_ca_blocks(P);
    const int ppb = (int)((P + bpi - 1) / bpi);
    hipLaunchKernelGGL(k_mpr_chan_partial, dim3((unsigned)(T * bpi)), dim3(256), 0, st, x, part, (int)P, Cp, bpi, ppb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_mpr_ca, dim3(T), dim3(1024), (size_t)(Cp + Cr + (1024 / Cp) * Cp) * sizeof(float), st, part, bpi, 1.0f / (float)P, w1, w2, z, C, Cp, Cr);
    return hipGetLastError();
}

hipError_t launch_mpr_scale_add(hipStream_t st, const float* res, const float* z, const float* x, float* out, int T, long P, int Cp) {
    const long total4 = (long)T * P * (Cp / 4);
    hipLaunchKernelGGL(k_mpr_scale_add, dim3(mpr_grid(total4)), dim3(256), 0, st, res, z, x, out, P, Cp, total4);
    return hipGetLastError();
}

hipError_t launch_mpr_sam(hipStream_t st, const float* x1, const float* y8, const float* ximg, const float* w3, const float* x, float* out,
                          float* img_out, int C, int Cp, long npx) {
    hipLaunchKernelGGL(k_mpr_sam, dim3(mpr_grid(npx * Cp)), dim3(256), 0, st, x1, y8, ximg, w3, x, out, img_out, C, Cp, npx);
    return hipGetLastError();
}

hipError_t launch_mpr_lr(hipStream_t st, float* wide, float* halves, int N, int h, int w, int C, int merge) {
    hipLaunchKernelGGL(k_mpr_lr, dim3(mpr_grid((long)N * h * 2 * w * C)), dim3(256), 0, st, wide, halves, N, h, w, C, merge);
    return hipGetLastError();
}

hipError_t launch_mpr_up2(hipStream_t st, const float* in, const float* add, float* out, int T, int H, int W, int C) {
    if (C & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_mpr_up2, dim3(mpr_grid((long)T * 4 * H * W * (C / 4))), dim3(256), 0, st, in, add, out, T, H, W, C);
    return hipGetLastError();
}

}  // namespace ira
