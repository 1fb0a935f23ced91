// The brightening tools as HIP kernels: RGB -> HSV (8-bit, OpenCV's integer arithmetic), a look-up on V, HSV -> RGB
// (8-bit through float32).  replaces BrighteningTool._invoke's cv2.cvtColor / _update_v / cv2.cvtColor chain
// (AgenticIR/executor/brightening/__init__.py:19-31) for ConstantShift (:34-41), GammaCorrection (:44-51) and
// HistogramEqualization = CLAHE(2.0, 8x8) (:54-63).  The arithmetic is OpenCV 4.8's (color_hsv.simd.hpp, clahe.cpp), restated in
// oracle/brightening_oracle.py, where every step is cited; that oracle is PARITY UNPINNED (no OpenCV in this image), and so is this file.
//
// HBM-bound byte work: 3 B read + 3 B written per pixel (CLAHE: 6 + 3, the V plane is recomputed rather than stored).  One thread
// converts 4 consecutive pixels = three aligned dwords in, three out, so a wave moves 768 contiguous bytes per access.  The two division
// tables of the forward transform (2 x 256 int) are rebuilt in LDS by every block (one double division per thread), the V look-up
// travels as a kernel argument.  This file is compiled with -ffp-contract=off: OpenCV's float steps round one operation at a time.
#include "ira_kernels.h"
#include <cmath>
#include <mutex>

namespace ira {
namespace {

#pragma clang fp contract(off)

constexpr int HSV_SHIFT = 12;
constexpr int CLAHE_TILES = 8;
struct VLut { uint8_t v[256]; };

__device__ __forceinline__ void build_div_tables(int* sdiv, int* hdiv) {
    // color_hsv.simd.hpp RGB2HSV_b: sdiv_table[i] = saturate_cast<int>((255 << hsv_shift) / (1. * i)), hdiv_table180[i] = ... (180 << hsv_shift) / (6. * i)
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        sdiv[i] = i ? __double2int_rn((double)(255 << HSV_SHIFT) / (1.0 * i)) : 0;
        hdiv[i] = i ? __double2int_rn((double)(180 << HSV_SHIFT) / (6.0 * i)) : 0;
    }
    __syncthreads();
}

__device__ __forceinline__ void rgb2hsv(int r, int g, int b, const int* sdiv, const int* hdiv, int& h, int& s, int& v) {
    v = max(max(r, g), b);
    const int vmin = min(min(r, g), b);
    const int diff = v - vmin;
    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    s = (diff * sdiv[v] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT;
    h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    h = (h * hdiv[diff] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT;
    h += h < 0 ? 180 : 0;
    h = min(max(h, 0), 255);
}

__device__ __forceinline__ int sat_u8(float x) { return min(max(__float2int_rn(x), 0), 255); }

__device__ __forceinline__ void hsv2rgb(int hi, int si, int vi, int& r, int& g, int& b) {
    // HSV2RGB_b -> HSV2RGB_simd: h * (6 / 180), s / 255, v / 255; tab1 = v - v s, tab2 = v - (v s) h, tab3 = tab1 + (v s) h
    const float hscale = 6.0f / 180.0f;
    float h = (float)hi * hscale;
    const float s = (float)si * (1.0f / 255.0f), v = (float)vi * (1.0f / 255.0f);
    const float pre = truncf(h);
    h = h - pre;
    const float vs = v * s;
    const float t0 = v, t1 = v - vs, t2 = v - vs * h, t3 = t1 + vs * h;
    const int sector = (int)(pre - truncf(pre * (1.0f / 6.0f)) * 6.0f);
    float fb, fg, fr;
    switch (sector) {
        case 0:  fb = t1; fg = t3; fr = t0; break;
        case 1:  fb = t1; fg = t0; fr = t2; break;
        case 2:  fb = t3; fg = t0; fr = t1; break;
        case 3:  fb = t0; fg = t2; fr = t1; break;
        case 4:  fb = t0; fg = t1; fr = t3; break;
        default: fb = t2; fg = t1; fr = t0; break;
    }
    r = sat_u8(fr * 255.0f); g = sat_u8(fg * 255.0f); b = sat_u8(fb * 255.0f);
}

// Four pixels per thread: bytes [12 q, 12 q + 12) of the interleaved frame as three dwords (the frame base is 4-byte aligned).
struct Quad { uint32_t w[3]; };
__device__ __forceinline__ Quad load_quad(const uint8_t* in, long q, long npx) {
    Quad d;
    if (4 * q + 4 <= npx) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(in) + 3 * q;
        d.w[0] = p[0]; d.w[1] = p[1]; d.w[2] = p[2];
    } else {
        d.w[0] = d.w[1] = d.w[2] = 0;
        for (long i = 12 * q; i < 3 * npx; ++i) d.w[(i - 12 * q) >> 2] |= (uint32_t)in[i] << (8 * ((i - 12 * q) & 3));
    }
    return d;
}
__device__ __forceinline__ void store_quad(uint8_t* out, long q, long npx, const Quad& d) {
    if (4 * q + 4 <= npx) {
        uint32_t* p = reinterpret_cast<uint32_t*>(out) + 3 * q;
        p[0] = d.w[0]; p[1] = d.w[1]; p[2] = d.w[2];
    } else {
        for (long i = 12 * q; i < 3 * npx; ++i) out[i] = (uint8_t)(d.w[(i - 12 * q) >> 2] >> (8 * ((i - 12 * q) & 3)));
    }
}
__device__ __forceinline__ int quad_byte(const Quad& d, int i) { return (d.w[i >> 2] >> (8 * (i & 3))) & 255; }
__device__ __forceinline__ void quad_set(Quad& d, int i, int v) { d.w[i >> 2] |= (uint32_t)v << (8 * (i & 3)); }

__global__ __launch_bounds__(256) void k_brighten_lut(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, long npx, const VLut lut) {
    __shared__ int sdiv[256], hdiv[256];
    __shared__ uint8_t vl[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) vl[i] = lut.v[i];
    build_div_tables(sdiv, hdiv);
    const long nq = (npx + 3) >> 2;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (long)gridDim.x * blockDim.x) {
        const Quad d = load_quad(in, q, npx);
        Quad o; o.w[0] = o.w[1] = o.w[2] = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int h, s, v, r, g, b;
            rgb2hsv(quad_byte(d, 3 * p), quad_byte(d, 3 * p + 1), quad_byte(d, 3 * p + 2), sdiv, hdiv, h, s, v);
            hsv2rgb(h, s, vl[v], r, g, b);
            quad_set(o, 3 * p, r); quad_set(o, 3 * p + 1, g); quad_set(o, 3 * p + 2, b);
        }
        store_quad(out, q, npx, o);
    }
}

// ---- CLAHE (clahe.cpp): per-tile histograms of V over the BORDER_REFLECT_101-extended plane
__device__ __forceinline__ int reflect101(int i, int n) { return i < n ? i : 2 * n - 2 - i; }      // i < 2 n - 1 (the host checks the pad < size)

// grid (chunks, 64): block (c, t) counts rows [c * rows_per, ...) of tile t into LDS, then adds its 256 bins to hist[t].
__global__ __launch_bounds__(256) void k_clahe_hist(const uint8_t* __restrict__ in, uint32_t* __restrict__ hist, int h, int w, int th, int tw, int rows_per) {
    __shared__ uint32_t bins[256];
    bins[threadIdx.x] = 0;
    __syncthreads();
    const int t = blockIdx.y, ty = t / CLAHE_TILES, tx = t % CLAHE_TILES;
    const int y0 = blockIdx.x * rows_per, y1 = min(y0 + rows_per, th);
    const int n = (y1 - y0) * tw;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int yy = y0 + i / tw, xx = i % tw;
        const int sy = reflect101(ty * th + yy, h), sx = reflect101(tx * tw + xx, w);
        const uint8_t* p = in + 3 * ((long)sy * w + sx);
        atomicAdd(&bins[max(max((int)p[0], (int)p[1]), (int)p[2])], 1u);
    }
    __syncthreads();
    if (bins[threadIdx.x]) atomicAdd(&hist[t * 256 + threadIdx.x], bins[threadIdx.x]);
}

// One block of 256 threads per tile: clip, redistribute, cumulate, scale (CLAHE_CalcLut_Body).
__global__ __launch_bounds__(256) void k_clahe_lut(const uint32_t* __restrict__ hist, uint8_t* __restrict__ luts, int clip, float lut_scale) {
    __shared__ int sc[256];
    const int i = threadIdx.x, t = blockIdx.x;
    int hv = (int)hist[t * 256 + i];
    if (clip > 0) {
        sc[i] = max(hv - clip, 0);
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (i < o) sc[i] += sc[i + o]; __syncthreads(); }
        const int clipped = sc[0];
        __syncthreads();
        hv = min(hv, clip);
        const int batch = clipped / 256;
        int residual = clipped - batch * 256;
        hv += batch;
        if (residual != 0) {
            const int step = max(256 / residual, 1);
            if (i % step == 0 && i / step < residual) hv += 1;        // for (i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++
        }
    }
    sc[i] = hv;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {                               // inclusive scan
        const int add = i >= o ? sc[i - o] : 0;
        __syncthreads();
        sc[i] += add;
        __syncthreads();
    }
    luts[t * 256 + i] = (uint8_t)sat_u8((float)sc[i] * lut_scale);
}

__global__ __launch_bounds__(256) void k_brighten_clahe(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, const uint8_t* __restrict__ luts,
                                                       int h, int w, int th, int tw) {
    __shared__ int sdiv[256], hdiv[256];
    build_div_tables(sdiv, hdiv);
    const long npx = (long)h * w, nq = (npx + 3) >> 2;
    const float inv_tw = 1.0f / (float)tw, inv_th = 1.0f / (float)th;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (long)gridDim.x * blockDim.x) {
        const Quad d = load_quad(in, q, npx);
        Quad o; o.w[0] = o.w[1] = o.w[2] = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const long px = min(4 * q + p, npx - 1);
            const int y = (int)(px / w), x = (int)(px - (long)y * w);
            int hh, s, v, r, g, b;
            rgb2hsv(quad_byte(d, 3 * p), quad_byte(d, 3 * p + 1), quad_byte(d, 3 * p + 2), sdiv, hdiv, hh, s, v);
            // CLAHE_Interpolation_Body
            const float txf = (float)x * inv_tw - 0.5f, tyf = (float)y * inv_th - 0.5f;
            int tx1 = (int)floorf(txf), ty1 = (int)floorf(tyf);
            const float xa = txf - (float)tx1, ya = tyf - (float)ty1;
            const float xa1 = 1.0f - xa, ya1 = 1.0f - ya;
            const int tx2 = min(tx1 + 1, CLAHE_TILES - 1), ty2 = min(ty1 + 1, CLAHE_TILES - 1);
            tx1 = max(tx1, 0); ty1 = max(ty1, 0);
            const float a = (float)luts[(ty1 * CLAHE_TILES + tx1) * 256 + v], bb = (float)luts[(ty1 * CLAHE_TILES + tx2) * 256 + v];
            const float c = (float)luts[(ty2 * CLAHE_TILES + tx1) * 256 + v], dd = (float)luts[(ty2 * CLAHE_TILES + tx2) * 256 + v];
            const float res = (a * xa1 + bb * xa) * ya1 + (c * xa1 + dd * xa) * ya;
            hsv2rgb(hh, s, sat_u8(res), r, g, b);
            quad_set(o, 3 * p, r); quad_set(o, 3 * p + 1, g); quad_set(o, 3 * p + 2, b);
        }
        store_quad(out, q, npx, o);
    }
}

}  // namespace

hipError_t launch_brighten(hipStream_t st, int tool, const uint8_t* in, uint8_t* out, int h, int w, uint32_t* hist_scratch, uint8_t* lut_scratch) {
    const long npx = (long)h * w, nq = (npx + 3) / 4;
    const int grid = (int)std::min<long>((nq + 255) / 256, 256L * 16);
    if (tool == 0 || tool == 1) {
        VLut lut;
        for (int v = 0; v < 256; ++v) {
            if (tool == 0) lut.v[v] = (uint8_t)std::min(v + 40, 255);                                          // ConstantShift :39-41
            else lut.v[v] = (uint8_t)std::nearbyint(std::min(std::max(std::pow(v / 255.0, 1.0 / 1.5) * 255, 0.0), 255.0));   // GammaCorrection :49-51
        }
        hipLaunchKernelGGL(k_brighten_lut, dim3(grid), dim3(256), 0, st, in, out, npx, lut);
        return hipGetLastError();
    }
    // clahe.cpp CLAHE_Impl::apply: tiles 8 x 8; a frame the grid does not divide is extended by 8 - (n % 8) on BOTH axes
    int he = h, we = w;
    if (h % CLAHE_TILES || w % CLAHE_TILES) { he = h + CLAHE_TILES - h % CLAHE_TILES; we = w + CLAHE_TILES - w % CLAHE_TILES; }
    const int th = he / CLAHE_TILES, tw = we / CLAHE_TILES, area = th * tw;
    const int clip = std::max((int)(2.0 * area / 256), 1);
    const float lut_scale = 255.0f / (float)area;
    hipError_t e = hipMemsetAsync(hist_scratch, 0, 64 * 256 * sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    const int rows_per = std::max(1, std::min(th, (4096 + tw - 1) / tw));
    hipLaunchKernelGGL(k_clahe_hist, dim3((th + rows_per - 1) / rows_per, 64), dim3(256), 0, st, in, hist_scratch, h, w, th, tw, rows_per);
    hipLaunchKernelGGL(k_clahe_lut, dim3(64), dim3(256), 0, st, hist_scratch, lut_scratch, clip, lut_scale);
    hipLaunchKernelGGL(k_brighten_clahe, dim3(grid), dim3(256), 0, st, in, out, lut_scratch, h, w, th, tw);
    return hipGetLastError();
}

}  // namespace ira
