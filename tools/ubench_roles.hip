// Role-split microbenchmark for gfx950 (round 6): what would the C = 192 swin tail (swin_block_tail_ws.hip) gain from
// v_mfma_f32_32x32x16_f16?  One 8-wave workgroup per CU, waves w and w + 4 share a SIMD, exactly the kernel's instruction MIX per
// 32 tokens and SIMD — no memory, no LDS, no barrier unless asked — in both MFMA shapes:
//     H wave (mlp.0 + GELU):  72 x 16x16x32  (or 36 x 32x32x16)  +  540 VALU (packed-fp16 GELU: v_pk_fma_f16 / v_cvt_pk_f16_f32)
//     P wave (proj + mlp.3): 108 x 16x16x32  (or 54 x 32x32x16)  +  200 VALU (v_fma_f32 / v_cvt_pk_f16_f32)
// The MFMAs of a wave rotate over NACC independent accumulators (the kernel's dependency distance is 2-3 MFMAs), the VALU fillers
// are spread evenly between them.  Also: the bare issue rate of v_mfma_f32_16x16x16_f16 (the K = 16 score MFMA of the head_dim-16
// attention) next to 16x16x32.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_roles.hip -o gpurun_out/ubench_roles && gpurun_out/ubench_roles
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TRIPS 256

template <int K>
__device__ __forceinline__ void fill_pk(float (&a)[8], float m, float c, int &r) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
        asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[r & 7]) : "v"(m), "v"(c));
        ++r;
    }
}
template <int K>
__device__ __forceinline__ void fill_f32(float (&a)[8], float m, float c, int &r) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[r & 7]) : "v"(m), "v"(c));
        ++r;
    }
}

// SHAPE 0: 16x16x32, 1: 32x32x16.  BAR: one s_barrier per trip (the kernel has one).  NACC: independent accumulators per wave.
template <int SHAPE, bool BAR, int NACC>
__global__ void __launch_bounds__(512) kroles(float *out, long long *cyc, float seed) {
    const int wave = threadIdx.x >> 6;
    const bool role_h = wave >= 4;
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;
    f32x4 acc4[NACC];
    f32x16 acc16[NACC];
    for (int i = 0; i < NACC; ++i) {
        acc4[i] = (f32x4){seed, seed, seed, seed};
        for (int j = 0; j < 16; ++j) acc16[i][j] = seed;
    }
    f16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (f16)(seed * 0.01f + i * 0.001f); fb[i] = (f16)(seed * 0.01f - i * 0.001f); }
    const float m = seed * 0.999f, c = seed * 0.0001f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < TRIPS; ++it) {
        int r = 0;
        if (role_h) {
            if constexpr (SHAPE == 0) {
#pragma unroll
                for (int i = 0; i < 72; ++i) {          // 540 / 72 = 7.5 fillers per MFMA
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4[i % NACC]) : "v"(fa), "v"(fb));
                    if (i & 1) fill_pk<8>(a, m, c, r); else fill_pk<7>(a, m, c, r);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 36; ++i) {          // 15 per MFMA
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc16[i % NACC]) : "v"(fa), "v"(fb));
                    fill_pk<15>(a, m, c, r);
                }
            }
        } else {
            if constexpr (SHAPE == 0) {
#pragma unroll
                for (int i = 0; i < 108; ++i) {         // 200 / 108: two fillers behind 92 of the 108
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4[i % NACC]) : "v"(fa), "v"(fb));
                    if (i % 27 < 23) fill_f32<2>(a, m, c, r); else fill_f32<1>(a, m, c, r);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 54; ++i) {
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc16[i % NACC]) : "v"(fa), "v"(fb));
                    if (i % 27 < 19) fill_f32<4>(a, m, c, r); else fill_f32<3>(a, m, c, r);
                }
            }
        }
        if constexpr (BAR) asm volatile("s_barrier" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < NACC; ++i) {
        for (int j = 0; j < 16; ++j) s += acc16[i][j];
        s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

// bare MFMA issue rate: KIND 0 = 16x16x32 f16, 1 = 16x16x16 f16, 2 = 32x32x16 f16; 4 independent accumulators, `waves` waves per workgroup
template <int KIND>
__global__ void __launch_bounds__(1024) kbare(float *out, long long *cyc, float seed) {
    f32x4 acc4[4];
    f32x16 acc16[4];
    for (int i = 0; i < 4; ++i) {
        acc4[i] = (f32x4){seed, seed, seed, seed};
        for (int j = 0; j < 16; ++j) acc16[i][j] = seed;
    }
    f16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (f16)(seed * 0.01f + i * 0.001f); fb[i] = (f16)(seed * 0.01f - i * 0.001f); }
    f16x4 ga = {fa[0], fa[1], fa[2], fa[3]}, gb = {fb[0], fb[1], fb[2], fb[3]};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 512; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (KIND == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4[i]) : "v"(fa), "v"(fb));
            else if constexpr (KIND == 1) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc4[i]) : "v"(ga), "v"(gb));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc16[i]) : "v"(fa), "v"(fb));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 16; ++j) s += acc16[i][j];
        s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x / 64] = t1 - t0;
}

static float *g_out;
static long long *g_cyc;

template <int SHAPE, bool BAR, int NACC>
void run_roles() {
    kroles<SHAPE, BAR, NACC><<<1, 512>>>(g_out, g_cyc, 1.0f);
    kroles<SHAPE, BAR, NACC><<<1, 512>>>(g_out, g_cyc, 1.0f);
    hipDeviceSynchronize();
    std::vector<long long> h(8);
    hipMemcpy(h.data(), g_cyc, 8 * 8, hipMemcpyDeviceToHost);
    long long hp = 0, hh = 0;
    for (int i = 0; i < 4; ++i) { hp = h[i] > hp ? h[i] : hp; hh = h[i + 4] > hh ? h[i + 4] : hh; }
    printf("roles %-9s barrier=%d nacc=%d : P wave %7.1f  H wave %7.1f   cycles per 32 tokens (MFMA pipe alone: 2880; real kernel ~7500)\n",
           SHAPE ? "32x32x16" : "16x16x32", (int)BAR, NACC, (double)hp / TRIPS, (double)hh / TRIPS);
}

template <int KIND>
void run_bare(const char *name) {
    printf("bare  %-9s :", name);
    for (int wps : {1, 2, 4}) {
        kbare<KIND><<<1, wps * 256>>>(g_out, g_cyc, 1.0f);
        kbare<KIND><<<1, wps * 256>>>(g_out, g_cyc, 1.0f);
        hipDeviceSynchronize();
        std::vector<long long> h(16);
        hipMemcpy(h.data(), g_cyc, 16 * 8, hipMemcpyDeviceToHost);
        long long mx = 0;
        for (int i = 0; i < wps * 4; ++i) mx = h[i] > mx ? h[i] : mx;
        printf("  wps%d %6.1f", wps, (double)mx / (512 * 4.0 * wps));
    }
    printf("   [cycles per MFMA per SIMD]\n");
}


// Generic two-role mix (16x16x32 only): wave >= 4 issues MH MFMAs + VH packed-fp16 VALU per trip, wave < 4 MP MFMAs + VP fp32 VALU, the
// VALU spread evenly behind the MFMAs.  What would BALANCING the two roles of a SIMD buy (the GELU split between H and P)?
template <int MH, int VH, int MP, int VP, bool BAR>
__global__ void __launch_bounds__(512) kmix2(float *out, long long *cyc, float seed) {
    const int wave = threadIdx.x >> 6;
    const bool role_h = wave >= 4;
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;
    f32x4 acc4[3];
    for (int i = 0; i < 3; ++i) acc4[i] = (f32x4){seed, seed, seed, seed};
    f16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (f16)(seed * 0.01f + i * 0.001f); fb[i] = (f16)(seed * 0.01f - i * 0.001f); }
    const float m = seed * 0.999f, c = seed * 0.0001f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < TRIPS; ++it) {
        int r = 0;
        if (role_h) {
#pragma unroll
            for (int i = 0; i < MH; ++i) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4[i % 3]) : "v"(fa), "v"(fb));
                constexpr int lo = VH / MH, extra = VH % MH;
                if (i < extra) fill_pk<lo + 1>(a, m, c, r); else fill_pk<lo>(a, m, c, r);
            }
        } else {
#pragma unroll
            for (int i = 0; i < MP; ++i) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4[i % 3]) : "v"(fa), "v"(fb));
                constexpr int lo = VP / MP, extra = VP % MP;
                if (i < extra) fill_pk<lo + 1>(a, m, c, r); else fill_pk<lo>(a, m, c, r);
            }
        }
        if constexpr (BAR) asm volatile("s_barrier" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 3; ++i) s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

template <int MH, int VH, int MP, int VP, bool BAR>
void run_mix2() {
    kmix2<MH, VH, MP, VP, BAR><<<1, 512>>>(g_out, g_cyc, 1.0f);
    kmix2<MH, VH, MP, VP, BAR><<<1, 512>>>(g_out, g_cyc, 1.0f);
    hipDeviceSynchronize();
    std::vector<long long> h(8);
    hipMemcpy(h.data(), g_cyc, 8 * 8, hipMemcpyDeviceToHost);
    long long hp = 0, hh = 0;
    for (int i = 0; i < 4; ++i) { hp = h[i] > hp ? h[i] : hp; hh = h[i + 4] > hh ? h[i + 4] : hh; }
    printf("mix2  H %3d MFMA + %3d VALU | P %3d MFMA + %3d VALU  barrier=%d : P wave %7.1f  H wave %7.1f  cycles per 32 tokens\n", MH, VH, MP, VP,
           (int)BAR, (double)hp / TRIPS, (double)hh / TRIPS);
}

// The kernel's PHASE structure instead of an even spread (per 16-token trip; two trips = 32 tokens per loop pass):
//   H wave: 12 bare MFMAs (pair 0) | 24 MFMAs with the GELU of the previous pair between them (7.5 VALU each) | 90 bare VALU (last GELU)
//   P wave: stage C 36 MFMAs with a few VALU | 50 VALU epilogue | stage A 18 MFMAs | 50 VALU epilogue
// PH = 1: these phases; PH = 0: the same counts spread evenly (kroles' form).  DEP = 1: each GELU block waits for "its" accumulators
// (an s_nop 7 x 2 in front of the block, the MFMA -> VALU wait states of a dependent read).
template <int PH, bool BAR>
__global__ void __launch_bounds__(512) kphase(float *out, long long *cyc, float seed) {
    const int wave = threadIdx.x >> 6;
    const bool role_h = wave >= 4;
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;
    f32x4 acc4[3];
    for (int i = 0; i < 3; ++i) acc4[i] = (f32x4){seed, seed, seed, seed};
    f16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (f16)(seed * 0.01f + i * 0.001f); fb[i] = (f16)(seed * 0.01f - i * 0.001f); }
    const float m = seed * 0.999f, c = seed * 0.0001f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 2 * TRIPS; ++it) {          // one pass = ONE 16-token trip
        int r = 0;
        auto mf = [&](int i) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4[i % 3]) : "v"(fa), "v"(fb)); };
        if (role_h) {
            if constexpr (PH == 1) {
#pragma unroll
                for (int i = 0; i < 12; ++i) mf(i);
#pragma unroll
                for (int i = 0; i < 24; ++i) { mf(i); if (i & 1) fill_pk<8>(a, m, c, r); else fill_pk<7>(a, m, c, r); }
                fill_pk<90>(a, m, c, r);
            } else {
#pragma unroll
                for (int i = 0; i < 36; ++i) { mf(i); if (i & 1) fill_pk<8>(a, m, c, r); else fill_pk<7>(a, m, c, r); }
            }
        } else {
            if constexpr (PH == 1) {
#pragma unroll
                for (int i = 0; i < 36; ++i) mf(i);
                fill_f32<50>(a, m, c, r);
#pragma unroll
                for (int i = 0; i < 18; ++i) mf(i);
                fill_f32<50>(a, m, c, r);
            } else {
#pragma unroll
                for (int i = 0; i < 54; ++i) { mf(i); if (i % 27 < 23) fill_f32<2>(a, m, c, r); else fill_f32<1>(a, m, c, r); }
            }
        }
        if constexpr (BAR) asm volatile("s_barrier" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 3; ++i) s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

template <int PH, bool BAR>
void run_phase() {
    kphase<PH, BAR><<<1, 512>>>(g_out, g_cyc, 1.0f);
    kphase<PH, BAR><<<1, 512>>>(g_out, g_cyc, 1.0f);
    hipDeviceSynchronize();
    std::vector<long long> h(8);
    hipMemcpy(h.data(), g_cyc, 8 * 8, hipMemcpyDeviceToHost);
    long long hp = 0, hh = 0;
    for (int i = 0; i < 4; ++i) { hp = h[i] > hp ? h[i] : hp; hh = h[i + 4] > hh ? h[i + 4] : hh; }
    printf("phase %s barrier=%d (16-token trips): P wave %7.1f  H wave %7.1f  cycles per 32 tokens\n", PH ? "kernel phases" : "even spread  ", (int)BAR,
           (double)hp / TRIPS, (double)hh / TRIPS);
}

int main() {
    hipMalloc(&g_out, 1024 * 4);
    hipMalloc(&g_cyc, 16 * 8);
    run_bare<0>("16x16x32");
    run_bare<1>("16x16x16");
    run_bare<2>("32x32x16");
    run_roles<0, false, 2>();
    run_roles<1, false, 2>();
    run_roles<0, false, 3>();
    run_roles<1, false, 3>();
    run_roles<0, true, 3>();
    run_roles<1, true, 3>();
    run_mix2<72, 540, 108, 200, true>();      // today's roles
    run_mix2<72, 370, 108, 370, true>();      // the GELU split evenly between the two waves of a SIMD
    run_mix2<90, 370, 90, 370, true>();       // MFMAs balanced too
    run_mix2<72, 270, 108, 470, true>();      // over-corrected (the P wave carries the surplus)
    run_mix2<72, 540, 108, 200, false>();
    run_mix2<72, 370, 108, 370, false>();
    run_phase<0, true>();
    run_phase<1, true>();
    run_phase<0, false>();
    run_phase<1, false>();
    return 0;
}
