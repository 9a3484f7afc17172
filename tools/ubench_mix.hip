// MFMA / VALU co-issue microbenchmark for gfx950 (round 2): cycles per MFMA of a stream [MFMA ; K fillers] x 4 on
// independent accumulators, for both f16 MFMA shapes, several filler kinds and 1 / 2 / 4 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mix.hip -o gpurun_out/ubench_mix && gpurun_out/ubench_mix
// Every [MFMA ; fillers] group is ONE asm block, so the issue order is exactly what is written here.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ITERS 512

// filler kinds
enum { F_FMA3 = 0, F_FMAMK, F_MUL2, F_MULS, F_CVTPK, F_EXP, F_PKF16, F_DSREAD, F_MOV, F_NKINDS };
static const char *kFillName[] = {"v_fma_f32 v,v,v,v", "v_fmamk_f32 v,v,K,v", "v_mul_f32 v,v,v", "v_mul_f32 v,s,v",
                                  "v_cvt_pk_f16_f32", "v_exp_f32", "v_pk_fma_f16", "ds_read_b128", "v_mov_b32"};

template <int FILL>
__device__ __forceinline__ void filler(float &a, float &b, float m, float c, f32x4 &lv, unsigned lds_addr) {
    if constexpr (FILL == F_FMA3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(c));
    else if constexpr (FILL == F_FMAMK) asm volatile("v_fmamk_f32 %0, %0, 0x3f7fff00, %1" : "+v"(a) : "v"(c));
    else if constexpr (FILL == F_MUL2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(m));
    else if constexpr (FILL == F_MULS) asm volatile("v_mul_f32 %0, 1.0, %0" : "+v"(a));
    else if constexpr (FILL == F_CVTPK) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(b) : "v"(a), "v"(m));
    else if constexpr (FILL == F_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
    else if constexpr (FILL == F_PKF16) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(c));
    else if constexpr (FILL == F_DSREAD) asm volatile("ds_read_b128 %0, %1" : "=v"(lv) : "v"(lds_addr));
    else if constexpr (FILL == F_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(b) : "v"(a));
}

// SHAPE 0: v_mfma_f32_16x16x32_f16 (4 acc regs), 1: v_mfma_f32_32x32x16_f16 (16 acc regs); NACC independent accumulators
template <int SHAPE, int FILL, int K>
__global__ void __launch_bounds__(1024) kmix(float *out, long long *cyc, float seed) {
    constexpr int NACC = 4;
    __shared__ f16x8 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = (f16x8){(f16)seed, 0, 0, 0, 0, 0, 0, 0};
    float a[16], b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; b[i] = 0.f; }
    f32x4 lv[4];
    for (int i = 0; i < 4; ++i) lv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x16 acc16[NACC];
    f32x4 acc4[NACC];
    for (int i = 0; i < NACC; ++i) {
        for (int j = 0; j < 16; ++j) acc16[i][j] = seed;
        acc4[i] = (f32x4){seed, seed, seed, seed};
    }
    f16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (f16)(seed * 0.01f + i * 0.001f); fb[i] = (f16)(seed * 0.01f - i * 0.001f); }
    const float m = seed * 0.999f, c = seed * 0.0001f;
    const unsigned lds_addr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if constexpr (SHAPE == 0)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4[i]) : "v"(fa), "v"(fb));
            else
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc16[i]) : "v"(fa), "v"(fb));
#pragma unroll
            for (int j = 0; j < K; ++j) filler<FILL>(a[(i * K + j) & 15], b[(i * K + j) & 15], m, c, lv[j & 3], lds_addr);
        }
        if constexpr (FILL == F_DSREAD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + b[i];
    for (int i = 0; i < 4; ++i) s += lv[i][0] + lv[i][1] + lv[i][2] + lv[i][3];
    for (int i = 0; i < NACC; ++i) {
        for (int j = 0; j < 16; ++j) s += acc16[i][j];
        s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x / 64] = t1 - t0;
}

static float *g_out;
static long long *g_cyc;

template <int SHAPE, int FILL, int K>
void run_one() {
    printf("%-9s %-22s K=%d :", SHAPE ? "32x32x16" : "16x16x32", kFillName[FILL], K);
    for (int wps : {1, 2, 4}) {         // waves per SIMD
        const int waves = wps * 4;
        kmix<SHAPE, FILL, K><<<1, waves * 64>>>(g_out, g_cyc, 1.0f);
        kmix<SHAPE, FILL, K><<<1, waves * 64>>>(g_out, g_cyc, 1.0f);
        hipDeviceSynchronize();
        std::vector<long long> h(16);
        hipMemcpy(h.data(), g_cyc, 16 * 8, hipMemcpyDeviceToHost);
        long long mx = 0;
        for (int i = 0; i < waves; ++i) mx = h[i] > mx ? h[i] : mx;
        // cycles per MFMA per SIMD (all waves of that SIMD together issue wps * ITERS * 4 MFMAs)
        printf("  wps%d %6.1f", wps, (double)mx / (ITERS * 4.0 * wps));
    }
    printf("   [cycles per MFMA per SIMD]\n");
}

template <int SHAPE, int FILL>
void run_fill() {
    run_one<SHAPE, FILL, 1>();
    run_one<SHAPE, FILL, 2>();
    run_one<SHAPE, FILL, 3>();
    run_one<SHAPE, FILL, 4>();
    run_one<SHAPE, FILL, 6>();
    run_one<SHAPE, FILL, 8>();
    if (SHAPE == 1) run_one<SHAPE, FILL, 12>();
}

template <int SHAPE>
void run_shape() {
    run_one<SHAPE, F_FMA3, 0>();
    run_fill<SHAPE, F_FMA3>();
    run_fill<SHAPE, F_FMAMK>();
    run_fill<SHAPE, F_MUL2>();
    run_fill<SHAPE, F_MULS>();
    run_fill<SHAPE, F_CVTPK>();
    run_fill<SHAPE, F_EXP>();
    run_fill<SHAPE, F_PKF16>();
    run_fill<SHAPE, F_DSREAD>();
    run_fill<SHAPE, F_MOV>();
}

int main() {
    hipMalloc(&g_out, 1024 * 4);
    hipMalloc(&g_cyc, 16 * 8);
    run_shape<0>();
    run_shape<1>();
    return 0;
}
