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

typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace nunif
