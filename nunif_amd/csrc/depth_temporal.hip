// The temporal ("motion") modules of Video-Depth-Anything's DPT head in their STREAMING form, one frame per call — the network
// behind `model.infer_video_depth_one(frame)` / `model.reset_state()` of iw3/video_depth_anything_streaming_model.py:58-103 (loaded
// there with torch.hub; NOT in the reference tree, no second implementation offline: parity is against oracle/video_depth_anything_net.py
// only, which restates the published architecture — PARITY UNPINNED).
//
// A module works on one NHWC map [P = H*W][C] of the current frame: GroupNorm(32) -> proj_in -> two temporal attention blocks
// (x += to_out(attn(LayerNorm(x)))) -> GEGLU feed-forward -> proj_out -> + input.  The Linears run on the engine's GEMM kernels
// (depth_anything.hip run_tok); this file holds what is not a GEMM.
//
// Temporal attention of a pixel and head: ONE query (this frame) against at most 32 keys (the previous <= 31 frames + this one).
// The published form keeps the LayerNorm'ed hidden states h_j of the window and evaluates K_j = Wk (h_j + pe_j), V_j = Wv (h_j + pe_j)
// for the whole window on every frame, because the position code pe_j of a cached frame moves as the window slides: 32x the Linear
// work of one frame.  to_k / to_v have no bias, so K_j = Wk h_j + Wk pe_j: the engine caches K0_j = Wk h_j and V0_j = Wv h_j
// (fp16, [32 slots][P][C], a ring) ONCE per frame, and the position part is three [32][C] fp32 tables (Wq pe, Wk pe, Wv pe)
// computed when the weights are loaded.  Per frame the attention then reads the two caches once — 2 x 32 x P x C x 2 bytes, the
// algorithmic minimum for this step — and the Linears see P tokens, not 32 P.
#include <algorithm>

#include "swin_kernels.h"

namespace nunif {

// ---- GroupNorm(32 groups) over one frame's [P][C] map --------------------------------------------------------------------------
// pass 1: block b sums rows [b R, (b + 1) R) per channel (thread = channel: a wave reads 128 contiguous bytes of a row)
__global__ void vda_gn_partial_kernel(const f16 *__restrict__ x, float2 *__restrict__ part, int P, int C, int R) {
    const int c = threadIdx.x, b = blockIdx.x;
    const int p0 = b * R, p1 = min(P, p0 + R);
    float s = 0.f, q = 0.f;
    for (int p = p0; p < p1; ++p) {
        const float v = (float)x[(long)p * C + c];
        s += v;
        q = fmaf(v, v, q);
    }
    part[(long)b * C + c] = make_float2(s, q);
}
// pass 2: every block adds the NB partials in the same fixed order (deterministic, no atomics), in double — the group statistics are
// E[x^2] - mean^2 over P * C / 32 values — then normalises its rows: y = (x - mean_g) rstd_g gamma_c + beta_c
__global__ void vda_gn_apply_kernel(const f16 *__restrict__ x, const float2 *__restrict__ part, const float *__restrict__ gamma,
                                    const float *__restrict__ beta, f16 *__restrict__ y, int P, int C, int R, int NB, float eps) {
    __shared__ double cs[1024], cq[1024];
    const int c = threadIdx.x, b = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int i = 0; i < NB; ++i) {
        const float2 v = part[(long)i * C + c];
        s += (double)v.x;
        q += (double)v.y;
    }
    cs[c] = s;
    cq[c] = q;
    __syncthreads();
    const int cpg = C / 32, g0 = c / cpg * cpg;
    double gs = 0.0, gq = 0.0;
    for (int i = 0; i < cpg; ++i) { gs += cs[g0 + i]; gq += cq[g0 + i]; }
    const double n = (double)P * cpg, mean = gs / n;
    double var = gq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps)), mu = (float)mean;
    const float ga = gamma[c], be = beta[c];
    const int p0 = b * R, p1 = min(P, p0 + R);
    for (int p = p0; p < p1; ++p) {
        const float v = (float)x[(long)p * C + c];
        y[(long)p * C + c] = (f16)((v - mu) * rstd * ga + be);
    }
}

int launch_vda_groupnorm(const f16 *x, const float *gamma, const float *beta, f16 *y, float2 *part, int P, int C, float eps,
                         hipStream_t s) {
    NUNIF_REQUIRE(P > 0 && C % 32 == 0 && C >= 32 && C <= 1024, "vda_groupnorm: %d channels unsupported (a multiple of 32, <= 1024)", C);
    int NB = std::min(kVdaGnBlocks, (P + 15) / 16);
    const int R = (P + NB - 1) / NB;
    NB = (P + R - 1) / R;
    ProfScope ps("vda_groupnorm", s, 0.0, (double)P * C * 6.0);
    vda_gn_partial_kernel<<<NB, C, 0, s>>>(x, part, P, C, R);
    NUNIF_LAUNCH_CHECK();
    vda_gn_apply_kernel<<<NB, C, 0, s>>>(x, part, gamma, beta, y, P, C, R, NB, eps);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

// ---- LayerNorm over C (a multiple of 64), one wave per token; two-pass variance in fp32 ----------------------------------------
__global__ void __launch_bounds__(256) vda_layernorm_kernel(const f16 *__restrict__ x, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, f16 *__restrict__ y, long T, int C,
                                                            float eps) {
    const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= T) return;
    const int lane = threadIdx.x & 63;
    const f16 *row = x + tok * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += (float)row[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = (float)row[c] - mean; q = fmaf(d, d, q); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)C + eps);
    for (int c = lane; c < C; c += 64) y[tok * C + c] = (f16)(((float)row[c] - mean) * rstd * gamma[c] + beta[c]);
}

int launch_vda_layernorm(const f16 *x, const float *gamma, const float *beta, f16 *y, long T, int C, float eps, hipStream_t s) {
    NUNIF_REQUIRE(T > 0 && C % 64 == 0, "vda_layernorm: %d channels unsupported (a multiple of 64)", C);
    ProfScope ps("vda_layernorm_kernel", s, 0.0, (double)T * C * 4.0);
    vda_layernorm_kernel<<<(unsigned)((T + 3) / 4), 256, 0, s>>>(x, gamma, beta, y, T, C, eps);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

// ---- GEGLU: out[t][i] = h[t][i] * gelu_erf(h[t][I + i]), h = [T][2 I] (diffusers GEGLU: value half first, gate half second) -------
__global__ void __launch_bounds__(256) vda_geglu_kernel(const f16 *__restrict__ h, f16 *__restrict__ out, long T, int I) {
    const int oct = I >> 3;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= T * oct) return;
    const long t = id / oct;
    const int o = (int)(id - t * oct);
    const f16x8 a = *reinterpret_cast<const f16x8 *>(h + t * 2 * I + o * 8);
    const f16x8 g = *reinterpret_cast<const f16x8 *>(h + t * 2 * I + I + o * 8);
    f16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float gv = (float)g[e];
        r[e] = (f16)((float)a[e] * (0.5f * gv * (1.0f + erff(gv * 0.70710678118654752f))));
    }
    *reinterpret_cast<f16x8 *>(out + t * I + o * 8) = r;
}

int launch_vda_geglu(const f16 *h, f16 *out, long T, int I, hipStream_t s) {
    NUNIF_REQUIRE(T > 0 && I % 8 == 0, "vda_geglu: inner width %d unsupported", I);
    ProfScope ps("vda_geglu_kernel", s, 0.0, (double)T * I * 6.0);
    vda_geglu_kernel<<<(unsigned)((T * (I / 8) + 255) / 256), 256, 0, s>>>(h, out, T, I);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

// ---- temporal attention of the current frame -----------------------------------------------------------------------------------
// One thread = one (pixel, head).  Window position j = 0 .. idx: j < idx are cached frames (ring slot (start + j) & 31), j = idx is
// this frame, whose K0 / V0 (columns C .. 3 C of its qkv row) the thread also writes into slot (start + idx) & 31.
// score_j = (q0 + PQ[idx]) . (K0_j + PK[j])  — Wq and PQ carry hd^-1/2 log2(e), so the softmax is exp2 —
// out = sum_j p_j (V0_j + PV[j]) / sum_j p_j.   Two passes over the head's channels in chunks of 8 (16-byte loads): scores first
// (32 registers), then the weighted sum — any head width that is a multiple of 8 without per-width register arrays.
__global__ void __launch_bounds__(256) vda_tattn_kernel(VdaTattnArgs g) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long)g.P * 8) return;
    const int head = (int)(id & 7);
    const long p = id >> 3;
    const int C = g.C, hd = g.hd, nch = hd >> 3, idx = g.idx;
    const long row = p * C + (long)head * hd;              // this (pixel, head) inside a [P][C] map
    const long slot = (long)g.P * C;
    const f16 *qrow = g.qkv + p * 3 * C + (long)head * hd;
    const int cur = (g.start + idx) & 31;
    float s[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) s[j] = 0.f;
    for (int c = 0; c < nch; ++c) {
        const f16x8 qh = *reinterpret_cast<const f16x8 *>(qrow + c * 8);
        const f16x8 kcur = *reinterpret_cast<const f16x8 *>(qrow + C + c * 8);
        const f16x8 vcur = *reinterpret_cast<const f16x8 *>(qrow + 2 * C + c * 8);
        *reinterpret_cast<f16x8 *>(g.kc + cur * slot + row + c * 8) = kcur;
        *reinterpret_cast<f16x8 *>(g.vc + cur * slot + row + c * 8) = vcur;
        const float *pq = g.pq + (long)idx * C + head * hd + c * 8;
        float q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) q[e] = (float)qh[e] + pq[e];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            if (j <= idx) {                                   // uniform: a scalar branch
                f16x8 kh = kcur;
                if (j < idx) kh = *reinterpret_cast<const f16x8 *>(g.kc + (long)((g.start + j) & 31) * slot + row + c * 8);
                const float *pk = g.pk + (long)j * C + head * hd + c * 8;
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) d = fmaf(q[e], (float)kh[e] + pk[e], d);
                s[j] += d;
            }
        }
    }
    float m = s[0];
#pragma unroll
    for (int j = 1; j < 32; ++j) if (j <= idx) m = fmaxf(m, s[j]);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        s[j] = j <= idx ? exp2f(s[j] - m) : 0.f;
        sum += s[j];
    }
    const float inv = 1.0f / sum;
    for (int c = 0; c < nch; ++c) {
        const f16x8 vcur = *reinterpret_cast<const f16x8 *>(qrow + 2 * C + c * 8);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            if (j <= idx) {
                f16x8 vh = vcur;
                if (j < idx) vh = *reinterpret_cast<const f16x8 *>(g.vc + (long)((g.start + j) & 31) * slot + row + c * 8);
                const float *pv = g.pv + (long)j * C + head * hd + c * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(s[j], (float)vh[e] + pv[e], acc[e]);
            }
        }
        f16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (f16)(acc[e] * inv);
        *reinterpret_cast<f16x8 *>(g.att + row + c * 8) = r;
    }
}

int launch_vda_tattn(const VdaTattnArgs &g, hipStream_t s) {
    NUNIF_REQUIRE(g.P > 0 && g.C == 8 * g.hd && g.hd % 8 == 0 && g.idx >= 0 && g.idx < 32 && g.start >= 0 && g.start < 32,
                  "vda_tattn: C=%d hd=%d idx=%d start=%d unsupported (8 heads, head width a multiple of 8, window <= 32)", g.C, g.hd,
                  g.idx, g.start);
    // per launch: the window's K0 / V0 once, this frame's qkv row, the att row
    ProfScope ps("vda_tattn_kernel", s, 4.0 * g.P * (double)g.C * (g.idx + 1), (double)g.P * g.C * 2.0 * (2.0 * g.idx + 3.0 + 2.0 + 1.0));
    vda_tattn_kernel<<<(unsigned)(((long)g.P * 8 + 255) / 256), 256, 0, s>>>(g);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

}  // namespace nunif
