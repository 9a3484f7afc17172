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

// ---- round 4: the same GELU in PACKED fp16 (v_pk_*_f16: two values per VALU slot) ------------------------------------
// gelu(x) = relu(x) - a q(t),  a = min(|x|, A), t = A - a, A = 3.5, q of degree 6: a q(t) = a Phi(-a) is the bump
// (<= 0.17) that separates GELU from ReLU, so the polynomial only ever produces a SMALL correction and fp16 rounding of
// its Horner steps costs <= ~3e-4 absolute (the x Phi(x) form loses 9e-3 to cancellation in fp16).  In
// the variable t the tail |x| -> A is the small-t end: every term vanishes there, no cancellation where the result is 0.
// Coefficients: weighted least squares on a q(t), then coordinate descent over fp16 neighbours under the exact packed-fma
// arithmetic (tools/gelu_fp16_fit.py): rms error 3.0e-4 for N(0, 0.4 / 1 / 2) inputs INCLUDING the rounding of the fp16
// result itself (the fp32 polynomial + rounding: 1.8e-4 .. 3.1e-4); end-to-end effect on the 2x net (CPU oracle with this
// arithmetic emulated): mse 4.4e-8 against 1.7e-8, i.e. 59.8 dB as the reference counts PSNR (tests hold >= 50).
// 11 packed operations + the convert per PAIR of values = 6 VALU slots per value instead of 9.5.
typedef f16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f16x8 gelu8h(const f32x4 &a, const f32x4 &b) {
    constexpr f16 kA = (f16)3.5f;
    constexpr f16 kq[7] = {(f16)0.00025725364685058594f, (f16)0.0005307197570800781f, (f16)0.0015268325805664062f,
                           (f16)0.00479888916015625f,    (f16)-0.004619598388671875f, (f16)0.004474639892578125f,
                           (f16)-0.0007538795471191406f};
    f16x2 x[4] = {{(f16)a[0], (f16)a[1]}, {(f16)a[2], (f16)a[3]}, {(f16)b[0], (f16)b[1]}, {(f16)b[2], (f16)b[3]}};
    f16x2 m[4], t[4], q[4], r[4];
    const f16x2 vA = {kA, kA}, z2 = {(f16)0.f, (f16)0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) m[i] = __builtin_elementwise_min(__builtin_elementwise_max(x[i], -x[i]), vA);
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = vA - m[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = __builtin_elementwise_fma(t[i], (f16x2){kq[6], kq[6]}, (f16x2){kq[5], kq[5]});
#pragma unroll
    for (int k = 4; k >= 0; --k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = __builtin_elementwise_fma(q[i], t[i], (f16x2){kq[k], kq[k]});
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = __builtin_elementwise_max(x[i], z2);
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = __builtin_elementwise_fma(-m[i], q[i], r[i]);
    return (f16x8){r[0][0], r[0][1], r[1][0], r[1][1], r[2][0], r[2][1], r[3][0], r[3][1]};
}

// The swin tails take the packed form; G32 = true (the 1x net, swin_unet.cpp) takes the fp32 polynomial.  Round-5 statistics over 8
// inputs per hot-regime case (tools/hot_regime_stats.py, profiles/r05a_hot_*.json): mean distance to the emulated fp16 reference
// with the packed / the fp32 form  2x -0.77 / -0.82 dB, 4x -0.86 / -0.56, 2x_chaos -2.81 / -2.76 — inside the +-0.3..0.4 dB standard
// error, no difference — but 1x -1.81 / -0.55: that net pays 1.3 dB for the packed form, so it does not get it.
// -DNUNIF_GELU_F32 forces the fp32 form everywhere (A/B builds).
template <bool G32 = false>
__device__ __forceinline__ f16x8 gelu8t(const f32x4 &a, const f32x4 &b) {
#ifdef NUNIF_GELU_F32
    return gelu8(a, b);
#else
    if constexpr (G32) return gelu8(a, b);
    else return gelu8h(a, b);
#endif
}

}  // namespace nunif
