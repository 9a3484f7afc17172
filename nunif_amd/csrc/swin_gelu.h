// GELU(erf) of 8 MFMA accumulator values -> fp16, shared by the swin tail kernels (swin_block_tail.hip,
// swin_block_tail_ws.hip).  Phi(x) ~= clamp01(0.5 + x P(x^2)), P of degree 6 in u = x^2 (Lawson minimax fit on
// |x| <= 4, weight x^2): |gelu err| <= 1.9e-4, below the fp16 quantum of the hidden activation it is rounded to.
// The odd polynomial is monotone beyond the fit interval (-> +-inf), so the [0, 1] clamp of the LAST fma (a free
// output modifier) replaces the input clamp: 9 VALU ops per element + the convert.  Written stage-major over the 8
// values so that consecutive instructions are independent (a dependent VALU op issues ~2.5x slower on gfx950,
// tools/ubench_valu.hip); build with -fno-slp-vectorize (hipcc otherwise packs the chains into half-rate
// v_pk_fma_f32 with an s_nop after each).
#pragma once
#include "common.h"

namespace nunif {

__device__ __forceinline__ f16x8 gelu8(const f32x4 &a, const f32x4 &b) {
#ifdef NUNIF_GELU_DEG8
    constexpr int ND = 8;
    constexpr float kc[ND + 1] = {8.063430101e-11f, -7.003475758e-09f, 2.716159007e-07f, -6.295003997e-06f,
                                  9.890811950e-05f, -1.133922332e-03f, 9.877477530e-03f, -6.641059600e-02f,
                                  3.989227099e-01f};
#else
    constexpr int ND = 6;
    constexpr float kc[ND + 1] = {2.2779071073841806e-08f, -1.5984835499693872e-06f, 4.795320637640543e-05f,
                                  -0.0008139933925122023f, 0.008772282861173153f, -0.06457287818193436f,
                                  0.39788317680358887f};
#endif
    float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    float u[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = v[i] * v[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = fmaf(kc[0], u[i], kc[1]);
#pragma unroll
    for (int k = 2; k <= ND; ++k) {
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] = fmaf(q[i], u[i], kc[k]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) asm("v_fma_f32 %0, %1, %2, 0.5 clamp" : "=v"(q[i]) : "v"(v[i]), "v"(q[i]));
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= q[i];
    return (f16x8){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3], (f16)v[4], (f16)v[5], (f16)v[6], (f16)v[7]};
}

}  // namespace nunif
