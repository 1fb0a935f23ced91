// gfx950 glue kernels of RRDBNet / Real-ESRGAN (reference DiffBIR/utils/realesrgan/rrdbnet.py): everything else is
// k_conv3x3 reading / writing channel slices of one dense-block buffer.
#include "ira_kernels.h"
#include "ira_device.h"

namespace ira {

// pixel_unshuffle (:53-71): in [T][H][W][c] -> out [T][H/s][W/s][c*s*s], channel = ci*s*s + sy*s + sx
__global__ __launch_bounds__(256) void k_pixel_unshuffle(const float* __restrict__ in, float* __restrict__ out, int T, int H, int W, int c, int s) {
    const long total = (long)T * H * W * c;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % c);
        long r = i / c;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int t = (int)(r / H);
        const int oc = ci * s * s + (y % s) * s + (x % s);
        out[(((long)t * (H / s) + y / s) * (W / s) + x / s) * (c * s * s) + oc] = in[i];
    }
}

// y[px][0:C] = a * u[px][0:C] + v[px][0:C] on pixels of ldu / ldv / ldy floats (u == nullptr: a strided copy of v); C % 4 == 0
__global__ __launch_bounds__(256) void k_axpy_strided(const float* u, int ldu, const float* v, int ldv, float* y, int ldy, float a, int C, long npx) {
    const int cv = C >> 2;
    const long total = npx * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long px = i / cv;
        const int c4 = (int)(i - px * cv) * 4;
        f32x4 r = *(const f32x4*)(v + px * ldv + c4);
        if (u != nullptr) r += a * *(const f32x4*)(u + px * ldu + c4);
        *(f32x4*)(y + px * ldy + c4) = r;
    }
}

static unsigned rr_grid(long total) {
    long g = (total + 255) / 256;
    return (unsigned)(g > 65536 ? 65536 : (g < 1 ? 1 : g));
}

hipError_t launch_pixel_unshuffle(hipStream_t st, const float* in, float* out, int T, int H, int W, int c, int s) {
    if (s < 1 || H % s || W % s) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_pixel_unshuffle, dim3(rr_grid((long)T * H * W * c)), dim3(256), 0, st, in, out, T, H, W, c, s);
    return hipGetLastError();
}

hipError_t launch_axpy_strided(hipStream_t st, const float* u, int ldu, const float* v, int ldv, float* y, int ldy, float a, int C, long npx) {
    if ((C | ldu | ldv | ldy) & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_axpy_strided, dim3(rr_grid(npx * (C / 4))), dim3(256), 0, st, u, ldu, v, ldv, y, ldy, a, C, npx);
    return hipGetLastError();
}

}  // namespace ira
