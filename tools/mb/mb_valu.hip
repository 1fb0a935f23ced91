// VALU / MFMA issue-rate microbenchmark for gfx950 (one workgroup on one CU): what ONE wave's instruction stream sustains, and what two
// waves on the same SIMD sustain together, for the instruction kinds the fused depthwise kernels are made of.
//   hipcc --offload-arch=gfx950 -O3 -o mb_valu mb_valu.hip && ./mb_valu
// Prints cycles per instruction per wave (s_memtime around an unrolled loop, wave 0) for 1, 2 and 3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

// mode: 0 v_fma_f32, 1 v_fma_mix_f32 (vgpr weight), 2 v_fma_mix_f32 (sgpr weight), 3 v_pk_fma_f32, 4 v_dot2_f32_f16, 5 v_pk_fma_f16,
//       6 mfma 32x32x16 f16 only, 7 waves < 4: mfma only, waves >= 4: v_fma_f32 only (producer/consumer on one SIMD),
//       8 one wave interleaving 1 mfma : 6 v_fma_f32, 9 v_exp_f32, 10 ds_read_b128 (conflict-free, same address pattern as the dwconv)
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters, float sw) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = lane * 0.001f + i;
    unsigned p = 0x3c003c00u + lane;
    float w = 1.0001f;
    f32x2 pa[8];
    for (int i = 0; i < 8; ++i) { pa[i][0] = a[i]; pa[i][1] = a[i + 8]; }
    f32x2 pw = {1.0001f, 0.9999f};
    half8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(0.01f * (lane + i)); fb[i] = (_Float16)(0.02f * (lane - i)); }
    f32x16 acc = {0};
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
            REP4(for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(w));)
        } else if constexpr (MODE == 1) {
            REP4(for (int i = 0; i < 16; ++i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(p), "v"(w));)
        } else if constexpr (MODE == 2) {
            REP4(for (int i = 0; i < 16; ++i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(p), "s"(sw));)
        } else if constexpr (MODE == 3) {
            REP4(for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(pa[i]) : "v"(pw));
                 for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(pa[i]) : "v"(pw));)
        } else if constexpr (MODE == 4) {
            REP4(for (int i = 0; i < 16; ++i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(p), "v"(p));)
        } else if constexpr (MODE == 5) {
            REP4(for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(p), "v"(p));)
        } else if constexpr (MODE == 6) {
            REP16(acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);)
            REP16(acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);)
            REP16(acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);)
            REP16(acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);)
        } else if constexpr (MODE == 7) {
            if (__builtin_amdgcn_readfirstlane(wave) < 4) {
                REP16(acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);)
            } else {
                REP4(for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(w));)
                REP4(for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(w));)
            }
        } else if constexpr (MODE == 8) {
            REP16(asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_fma_f32 %3, %3, %9, %3\n\tv_fma_f32 %4, %4, %9, %4\n\tv_fma_f32 %5, %5, %9, %5\n\t"
                               "v_fma_f32 %6, %6, %9, %6\n\tv_fma_f32 %7, %7, %9, %7\n\tv_fma_f32 %8, %8, %9, %8"
                               : "+v"(acc) : "v"(fa), "v"(fb), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(w));)
        } else if constexpr (MODE == 11 || MODE == 12) {
            // the fused-GDFN slot: waves 0-3 ("P") a dependent chain of 4 MFMAs with 4 FMAs behind each; waves 4+ ("C") 2 dependent MFMAs + 16 FMAs
            // MODE 12: P uses two alternating accumulators
            if (__builtin_amdgcn_readfirstlane(wave) < 4) {
                f32x16 acc2 = acc;
                REP4(REP4(asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_fma_f32 %3, %3, %7, %3\n\tv_fma_f32 %4, %4, %7, %4\n\tv_fma_f32 %5, %5, %7, %5\n\tv_fma_f32 %6, %6, %7, %6"
                                   : "+v"(acc) : "v"(fa), "v"(fb), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(w));
                          if (MODE == 12) { f32x16 tmp = acc; acc = acc2; acc2 = tmp; }))
                acc += acc2;
            } else {
                REP4(asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb));
                     for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(w));
                     asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb));
                     for (int i = 8; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(w));)
            }
        } else if constexpr (MODE == 13 || MODE == 14) {
            // MFMA on fp16 SUBNORMAL A operands (the lo weight plane of the hi+lo split is mostly subnormal): same rate?
            half8 fd;
            for (int i = 0; i < 8; ++i) fd[i] = (_Float16)(MODE == 13 ? 3.0e-6f * (1 + (lane + i) % 7) : 0.f);
            REP16(acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fd, fb, acc, 0, 0, 0);)
            REP16(acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fd, fb, acc, 0, 0, 0);)
            REP16(acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fd, fb, acc, 0, 0, 0);)
            REP16(acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fd, fb, acc, 0, 0, 0);)
        } else if constexpr (MODE == 9) {
            REP4(for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));)
        } else if constexpr (MODE == 10) {
            const float* q = lds + (lane & 31) * 36 + (lane >> 5) * 4;     // 144-byte pixel stride, two octets
            REP4(for (int i = 0; i < 16; ++i) { f32x4v v = *(const f32x4v*)(q + 4 * i); asm volatile("" :: "v"(v)); })
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i] + acc[i];
    for (int i = 0; i < 8; ++i) s += pa[i][0] + pa[i][1];
    out[threadIdx.x] = s;
    if (lane == 0) cyc[wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter, float* out, unsigned long long* cyc) {
    const int iters = 200;
    for (int waves : {4, 8, 12, 16}) {
        k<MODE><<<1, 64 * waves>>>(out, cyc, iters, 1.0001f);
        hipDeviceSynchronize();
        unsigned long long h[16];
        hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        printf("%-44s %2d waves (%d/SIMD): %6.2f cyc/instr/wave (wave 0)  %6.2f (last wave)\n", name, waves, waves / 4, (double)h[0] / iters / per_iter,
               (double)h[waves - 1] / iters / per_iter);
    }
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 128);
    run<0>("v_fma_f32", 64, out, cyc);
    run<1>("v_fma_mix_f32 (f16 x f32 vgpr)", 64, out, cyc);
    run<2>("v_fma_mix_f32 (f16 x f32 sgpr)", 64, out, cyc);
    run<3>("v_pk_fma_f32 (2 fma each)", 64, out, cyc);
    run<4>("v_dot2_f32_f16 (2 mac each)", 64, out, cyc);
    run<5>("v_pk_fma_f16 (2 fma each)", 64, out, cyc);
    run<6>("v_mfma_f32_32x32x16_f16", 64, out, cyc);
    run<7>("waves 0-3 mfma x16 | waves 4+ v_fma x128", 16, out, cyc);
    run<8>("1 mfma + 6 v_fma interleaved (per group of 7)", 16, out, cyc);
    run<11>("slot: P 16x(mfma+4fma) | C 4x(2mfma+16fma) /16", 16, out, cyc);
    run<12>("slot, P on two accumulators /16", 16, out, cyc);
    run<13>("mfma, A = fp16 subnormals", 64, out, cyc);
    run<14>("mfma, A = zeros", 64, out, cyc);
    run<9>("v_exp_f32", 64, out, cyc);
    run<10>("ds_read_b128 (16 in flight)", 64, out, cyc);
    return 0;
}
