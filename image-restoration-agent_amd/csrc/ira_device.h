// Device-side vocabulary shared by every kernel file of libira_hip.so (gfx950, wave64): vector types, the MFMA row
// permutation, small conversions, wave reductions, the XCD-aware block order, the LDS-only barrier, the mixed-precision
// depthwise tap and the exact-erf GELU.  Everything is __forceinline__, so sharing it changes no generated code.
#pragma once
#include <hip/hip_runtime.h>

namespace ira {

typedef _Float16 f16;
typedef f16 half8 __attribute__((ext_vector_type(8)));
typedef f16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define IRA_DEV __device__ __forceinline__

// XCD-aware work order (cdna_hip_programming.md T1): hardware deals consecutive block ids round-robin over the 8 XCDs, each
// with its own L2.  Remap so that every XCD walks one CONTIGUOUS chunk of the logical (tile, row, x) order: vertically
// adjacent patches then meet in the same L2 (the 3x3 halo re-reads cost 3-6x HBM/fabric traffic without this,
// profiles/r01_v2_*).  Bijective for any block count.
IRA_DEV long xcd_swizzle(long bid, long nblk) {
    const long q = nblk >> 3, rem = nblk & 7, x = bid & 7, j = bid >> 3;
    return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + j;
}

// MFMA row i of a weight fragment holds matrix row 32*nb + rowperm(i) (an involution: swaps bits 2 and 3), so that a
// lane's 16 accumulator registers are two runs of 8 consecutive channels: see the header of ira_kernels.hip.
IRA_DEV int rowperm(int i) { return 16 * (i >> 4) + 8 * ((i >> 2) & 1) + 4 * ((i >> 3) & 1) + (i & 3); }

IRA_DEV half8 cvt8(const f32x4 lo, const f32x4 hi) {
    half8 o;
    o[0] = (f16)lo[0]; o[1] = (f16)lo[1]; o[2] = (f16)lo[2]; o[3] = (f16)lo[3];
    o[4] = (f16)hi[0]; o[5] = (f16)hi[1]; o[6] = (f16)hi[2]; o[7] = (f16)hi[3];
    return o;
}

IRA_DEV half8 zero8() { half8 z; for (int j = 0; j < 8; ++j) z[j] = (f16)0.f; return z; }

IRA_DEV f32x16 zero16() { f32x16 z; for (int j = 0; j < 16; ++j) z[j] = 0.f; return z; }

IRA_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
IRA_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Exact-erf GELU (F.gelu default, restormer_arch.py:91; SwinIR Mlp): 0.5 x (1 + erf(x / sqrt 2)).
// erfc(|z|) by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7), used as 1+erf(z) = erfc(-z) so that the
// negative tail has no cancellation; branch-free: one v_rcp, one v_exp, six FMAs.
IRA_DEV float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    const float e = p * t * __expf(-z * z);            // erfc(|z|)
    const float one_plus_erf = (x >= 0.f) ? (2.0f - e) : e;
    return 0.5f * x * one_plus_erf;
}

// Workgroup barrier for LDS hand-offs only.  __syncthreads() also waits vmcnt(0): every chunk would then wait for the
// consumers' global STORES to be acknowledged and would drain the producers' weight prefetches (measured with
// s_memtime: 4.1k cycles per chunk for ~350 VALU instructions).  The LDS traffic only needs lgkmcnt(0) + s_barrier.
IRA_DEV void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// acc += (float)half(lo/hi 16 bits of `packed`) * w     -- v_fma_mix_f32: fp16 source 0, fp32 sources 1 and 2.
// One instruction instead of v_cvt_f32_f16 + v_fma_f32 (the converts were half of the VALU work of the depthwise
// conv and kept 32 extra registers live).
IRA_DEV void fma_mix_lo(float& acc, unsigned packed, float w) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(packed), "v"(w));
}
IRA_DEV void fma_mix_hi(float& acc, unsigned packed, float w) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(packed), "v"(w));
}
// one depthwise tap on 8 channels: v = 8 fp16 activations, wl/wh = their 8 fp32 weights
IRA_DEV void dw_tap(float (&acc)[8], const u32x4 v, const f32x4 wl, const f32x4 wh) {
    fma_mix_lo(acc[0], v[0], wl[0]); fma_mix_hi(acc[1], v[0], wl[1]);
    fma_mix_lo(acc[2], v[1], wl[2]); fma_mix_hi(acc[3], v[1], wl[3]);
    fma_mix_lo(acc[4], v[2], wh[0]); fma_mix_hi(acc[5], v[2], wh[1]);
    fma_mix_lo(acc[6], v[3], wh[2]); fma_mix_hi(acc[7], v[3], wh[3]);
}

}  // namespace ira
