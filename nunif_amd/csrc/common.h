// Shared host/device helpers for libnunif_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/nunif_hip.h"

namespace nunif {

void set_error(const char *fmt, ...);

#define NUNIF_HIP_CHECK(expr)                                                                     \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            ::nunif::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                               __LINE__);                                                         \
            return NUNIF_HIP_EHIP;                                                                \
        }                                                                                         \
    } while (0)

#define NUNIF_REQUIRE(cond, ...)              \
    do {                                      \
        if (!(cond)) {                        \
            ::nunif::set_error(__VA_ARGS__);  \
            return NUNIF_HIP_EINVAL;          \
        }                                     \
    } while (0)

#define NUNIF_LAUNCH_CHECK()                                                                      \
    do {                                                                                          \
        hipError_t _e = hipGetLastError();                                                        \
        if (_e != hipSuccess) {                                                                   \
            ::nunif::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),        \
                               __FILE__, __LINE__);                                               \
            return NUNIF_HIP_EHIP;                                                                \
        }                                                                                         \
    } while (0)

// ---- optional per-kernel-class timing (HIP events on the launch stream) ------------------------------------
struct ProfScope {
    ProfScope(const char *name, hipStream_t s, double flops, double bytes);
    ~ProfScope();
    int slot;
    hipStream_t stream;
};
bool profiling_enabled();
bool profile_tags_enabled();     // NUNIF_PROF_TAGS=1: profiler classes of the generic GEMMs are named after the call site, not the kernel symbol

typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#if defined(__HIPCC__)
// An MFMA accumulator lane (token l&15, lane group g = l>>4) holds channels 4g..4g+3 of a 16-channel tile: 8 bytes of
// fp16, and a wave-wide store of them touches 16 rows x 32 B — quarter cache lines.  Two v_permlane16_swap turn the
// chunks of two adjacent tiles (A: channels 4g.., B: channels 16+4g..) into ONE run of 8 consecutive channels per lane,
// starting at pair_run_channel(g) inside the 32-channel pair, so loads / stores are 16 B per lane and 64 B per row.
// (v_permlane16_swap a, b: a.row1 <-> b.row0, a.row3 <-> b.row2, rows = 16 lanes; verified on gfx950.)
__device__ __forceinline__ int pair_run_channel(int grp) { return ((grp & 1) << 4) | ((grp >> 1) << 3); }   // 0,16,8,24

__device__ __forceinline__ u32x2 lane16_swap(unsigned int a, unsigned int b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_permlane16_swap(a, b, false, false);
#else
    return (u32x2){a, b};          // host pass of the single-source compile only parses this
#endif
}

__device__ __forceinline__ f16x8 pair_to_run(f16x4 a, f16x4 b) {
    const u32x2 ai = __builtin_bit_cast(u32x2, a), bi = __builtin_bit_cast(u32x2, b);
    const u32x2 r0 = lane16_swap(ai[0], bi[0]);
    const u32x2 r1 = lane16_swap(ai[1], bi[1]);
    return __builtin_bit_cast(f16x8, ((u32x4){r0[0], r1[0], r0[1], r1[1]}));
}

// max over the four 16-lane rows of a wave (the lanes with equal lane & 15), result in every lane: two register swaps on the
// VALU (v_permlane16_swap / v_permlane32_swap of a value with its own copy) instead of two ds_bpermute round trips through
// the LDS crossbar, which sit in the middle of every softmax dependency chain
__device__ __forceinline__ float row_group_max(float m) {
#if defined(__HIP_DEVICE_COMPILE__)
    // NB: ``__builtin_bit_cast(float, v[1])`` on an ext-vector ELEMENT reads element 0 with hipcc of ROCm 7.2 (the cast takes
    // the address of the whole vector) — it silently reduced this to a max over lane group 0 only, a stabiliser all lanes still
    // share, so the softmax stayed right until another group's score exceeded it by 16 (fp16 P overflows -> NaN; one test input
    // did).  Elements are copied into scalars first.
    const unsigned u = __builtin_bit_cast(unsigned, m);
    const u32x2 a = __builtin_amdgcn_permlane16_swap(u, u, false, false);       // rows [0,0,2,2] / [1,1,3,3]
    const unsigned a0 = a[0], a1 = a[1];
    m = fmaxf(__builtin_bit_cast(float, a0), __builtin_bit_cast(float, a1));
    const unsigned v = __builtin_bit_cast(unsigned, m);
    const u32x2 b = __builtin_amdgcn_permlane32_swap(v, v, false, false);       // lower halves / upper halves
    const unsigned b0 = b[0], b1 = b[1];
    m = fmaxf(__builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1));
#endif
    return m;
}

__device__ __forceinline__ void run_to_pair(f16x8 v, f16x4 &a, f16x4 &b) {          // inverse (the swap is an involution)
    const u32x4 vi = __builtin_bit_cast(u32x4, v);
    const u32x2 r0 = lane16_swap(vi[0], vi[2]);
    const u32x2 r1 = lane16_swap(vi[1], vi[3]);
    a = __builtin_bit_cast(f16x4, ((u32x2){r0[0], r1[0]}));
    b = __builtin_bit_cast(f16x4, ((u32x2){r0[1], r1[1]}));
}
#endif

// Workgroups of a 1-D grid are dealt round-robin to the 8 XCDs (blockIdx.x % 8), and every XCD has its OWN L2: neighbours in
// blockIdx never share one.  xcd_contiguous() renumbers the grid so that XCD x owns a contiguous range of logical ids — use the
// result wherever neighbouring ids share operands (the channel blocks of one token block, the query blocks of one head), or each
// of the 8 L2s pulls its own copy of every shared tile through the fabric at the same moment.
#if defined(__HIPCC__)
__device__ __forceinline__ unsigned xcd_contiguous(unsigned b, unsigned n) {
    constexpr unsigned kXcd = 8;
    const unsigned x = b % kXcd, i = b / kXcd, base = n / kXcd, rem = n % kXcd;
    return x * base + (x < rem ? x : rem) + i;
}
#endif

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace nunif
