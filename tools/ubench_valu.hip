// Issue-rate microbenchmark for gfx950: cycles per instruction of the VALU / MFMA mixes the swin kernels use.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o gpurun_out/ubench_valu && gpurun_out/ubench_valu
// One workgroup of WAVES waves per CU-sized grid slot (grid = 1: a single CU), s_memtime around an unrolled loop.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define ITERS 256
__device__ __forceinline__ float fma1(float a, float m, float c) {
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(c));
    return a;
}
#define N_ACC 16

template <int MODE>
__global__ void __launch_bounds__(1024) k(float *out, long long *cyc, float seed) {
    float a[N_ACC];
#pragma unroll
    for (int i = 0; i < N_ACC; ++i) a[i] = seed + i + threadIdx.x;
    __shared__ f16x8 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = (f16x8){(f16)seed, 0, 0, 0, 0, 0, 0, 0};
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc16[2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc16[i][j] = seed;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){seed, seed, seed, seed};
    f16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (f16)(seed + i); fb[i] = (f16)(seed - i); }
    const float m = seed * 0.5f, c = seed * 0.25f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
        if constexpr (MODE == 0) {          // 16 independent v_fma_f32
#pragma unroll
            for (int i = 0; i < N_ACC; ++i) a[i] = fma1(a[i], m, c);
        } else if constexpr (MODE == 1) {   // 8 v_pk_fma_f32 (16 fmas)
#pragma unroll
            for (int i = 0; i < N_ACC; i += 2) {
                f32x2 v = {a[i], a[i + 1]};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"((f32x2){m, m}), "v"((f32x2){c, c}));
                a[i] = v[0]; a[i + 1] = v[1];
            }
        } else if constexpr (MODE == 2) {   // 16 v_exp_f32
#pragma unroll
            for (int i = 0; i < N_ACC; ++i) a[i] = __builtin_amdgcn_exp2f(a[i]);
        } else if constexpr (MODE == 3) {   // 16 v_cvt_pkrtz / cvt_pk_f16 (2 floats -> packed half)
#pragma unroll
            for (int i = 0; i < N_ACC; i += 2) {
                f16x2 h = {(f16)a[i], (f16)a[i + 1]};
                a[i] = (float)h[0] + m; a[i + 1] = (float)h[1];
            }
        } else if constexpr (MODE == 4) {   // 4 independent MFMA 16x16x32 f16
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[i], 0, 0, 0);
        } else if constexpr (MODE == 5) {   // 4 MFMA + 16 independent fma interleaved (same wave)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[i], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) a[4 * i + j] = fma1(a[4 * i + j], m, c);
            }
        } else if constexpr (MODE == 6) {   // 4 MFMA + 32 fma interleaved
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[i], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) { a[4 * i + j] = fma1(a[4 * i + j], m, c); a[4 * i + j] = fma1(a[4 * i + j], c, m); }
            }
        } else if constexpr (MODE == 7) {   // 16 v_max/min (med3 clamp)
#pragma unroll
            for (int i = 0; i < N_ACC; ++i) a[i] = __builtin_fminf(__builtin_fmaxf(a[i], -m), m + it);
        } else if constexpr (MODE == 8) {   // 16 v_pk_fma_f16 (32 half fmas)
#pragma unroll
            for (int i = 0; i < N_ACC; ++i) {
                f16x2 v = __builtin_bit_cast(f16x2, a[i]);
                v = v * (f16x2){(f16)m, (f16)m} + (f16x2){(f16)c, (f16)c};
                a[i] = __builtin_bit_cast(float, v);
            }
        } else if constexpr (MODE == 10) {  // waves 0-3: MFMA only; waves 4-7: fma only (wave w and w+4 share a SIMD)
            if (threadIdx.x < 256) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[i], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < N_ACC; ++i) a[i] = fma1(a[i], m, c);
            }
        } else if constexpr (MODE == 15 || MODE == 16 || MODE == 17) {  // as 10 with s_setprio on one side / 16 waves
            const bool mf = MODE == 17 ? (threadIdx.x / 64) % 8 < 4 : threadIdx.x < 256;
            if (it == 0) {
                if (MODE == 15 && !mf) __builtin_amdgcn_s_setprio(3);
                if (MODE == 16 && mf) __builtin_amdgcn_s_setprio(3);
            }
            if (mf) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[i], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < N_ACC; ++i) a[i] = fma1(a[i], m, c);
            }
        } else if constexpr (MODE == 11) {  // 4 x 32x32x16 MFMA
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc16[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc16[i], 0, 0, 0);
            }
        } else if constexpr (MODE == 12) {  // 2 x (32x32x16 MFMA + 8 fma)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc16[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc16[i], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[8 * i + j] = fma1(a[8 * i + j], m, c);
            }
        } else if constexpr (MODE == 13) {  // waves 0-3: MFMA only; waves 4-7: exp only
            if (threadIdx.x < 256) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[i], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < N_ACC; ++i) a[i] = __builtin_amdgcn_exp2f(a[i]);
            }
        } else if constexpr (MODE == 14) {  // 16 ds_read_b128 (LDS) + 4 MFMA in one wave
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[i], 0, 0, 0);
                f16x8 t = lds[(threadIdx.x + 64 * i + it) & 1023];
                fa[i] = t[0];
            }
        } else if constexpr (MODE == 9) {   // 8 dependent chains of 2 (dependent-issue latency): a = fma(fma(a))
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int r = 0; r < 8; ++r) a[i] = fma1(a[i], m, c);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < N_ACC; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) s += acc16[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char *name, int instr_per_iter, int waves) {
    float *out; long long *cyc;
    hipMalloc(&out, 1024 * 4); hipMalloc(&cyc, 16 * 8);
    k<MODE><<<1, waves * 64>>>(out, cyc, 1.0f);
    k<MODE><<<1, waves * 64>>>(out, cyc, 1.0f);
    hipDeviceSynchronize();
    std::vector<long long> h(16);
    hipMemcpy(h.data(), cyc, 16 * 8, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int i = 0; i < waves; ++i) mx = h[i] > mx ? h[i] : mx;
    if (MODE == 10 || MODE == 13 || MODE >= 15) { for (int i = 0; i < waves; ++i) printf(" %lld", h[i]); printf("\n"); }
    if (false) printf("   per-wave ticks: %lld %lld %lld %lld | %lld %lld %lld %lld\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    // s_memtime / readcyclecounter ticks at a constant 100 MHz on gfx9; convert with the shader clock if so
    printf("%-44s waves/WG=%d  ticks=%lld  ticks/instr/wave=%.3f  (x waves per SIMD = %.2f)\n", name, waves, mx,
           (double)mx / (ITERS * instr_per_iter), (double)mx / (ITERS * instr_per_iter) / ((waves + 3) / 4));
    hipFree(out); hipFree(cyc);
}

int main() {
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    int wclk = 0;
    hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0);
    printf("shader clock %d kHz, wall clock rate %d kHz\n", clk, wclk);
    for (int waves : {1, 4, 8}) {
        run<0>("16 x v_fma_f32", 16, waves);
        run<1>("8 x v_pk_fma_f32 (16 fma)", 8, waves);
        run<2>("16 x v_exp_f32", 16, waves);
        run<3>("8 x (cvt_pk_f16 + 2 cvt_f32_f16 + add)", 8, waves);
        run<4>("4 x mfma_16x16x32_f16", 4, waves);
        run<5>("4 x (mfma + 4 fma)", 4, waves);
        run<6>("4 x (mfma + 8 fma)", 4, waves);
        run<7>("16 x (max+min)", 16, waves);
        run<8>("16 x v_pk_fma_f16", 16, waves);
        run<9>("16 dependent fma (2 chains of 8)", 16, waves);
        run<11>("2 x mfma_32x32x16_f16", 2, waves);
        run<12>("2 x (mfma_32x32x16 + 8 fma)", 2, waves);
        run<14>("4 x (mfma + ds_read_b128)", 4, waves);
    }
    run<10>("waves0-3: 4 mfma | waves4-7: 16 fma", 1, 8);
    run<13>("waves0-3: 4 mfma | waves4-7: 16 exp", 1, 8);
    run<15>("mfma | fma, fma waves prio 3", 1, 8);
    run<16>("mfma | fma, mfma waves prio 3", 1, 8);
    run<17>("16 waves: w%8<4 mfma | else fma", 1, 16);
    run<4>("16 waves mfma only", 4, 16);
    run<0>("16 waves fma only", 16, 16);
    return 0;
}
