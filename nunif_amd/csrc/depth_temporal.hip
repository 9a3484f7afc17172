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
#include <cstdlib>

#include "swin_kernels.h"

namespace nunif {

// ---- GroupNorm(32 groups) over one frame's [P][C] map --------------------------------------------------------------------------
// Three small launches (the first form — 64 blocks whose threads walked 88 rows of 2-byte loads twice — took 35 us per module; a
// 256-block form whose single finalising block then added 256 partials per thread, 50):
// pass 1: a block of RL x C / 8 threads sums its R rows with 16-byte loads (thread = row lane x 8-channel octet), the row lanes meet
// in LDS in a fixed order, and the block writes one (sum, sum of squares) per channel
// (blockIdx.y = the frame of a batch: maps [F][P][C], partials / coefficients [F][kVdaGnBlocks + 1][C])
__global__ void __launch_bounds__(256) vda_gn_partial_kernel(const f16 *__restrict__ x, float2 *__restrict__ part, int P, int C, int R) {
    __shared__ float2 red[2048];                        // [RL][C], RL * C / 8 <= 256
    const int CL = C >> 3, RL = 256 / CL, tid = threadIdx.x, b = blockIdx.x;
    x += (long)blockIdx.y * P * C;
    part += (long)blockIdx.y * (kVdaGnBlocks + 1) * C;
    const int rl = tid / CL, cl = tid - rl * CL;
    const int p0 = b * R, p1 = min(P, p0 + R);
    if (rl < RL) {
        float s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
        for (int p = p0 + rl; p < p1; p += RL) {
            const f16x8 v = *reinterpret_cast<const f16x8 *>(x + (long)p * C + cl * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s[e] += f; q[e] = fmaf(f, f, q[e]); }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rl * C + cl * 8 + e] = make_float2(s[e], q[e]);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float ss = 0.f, qq = 0.f;
        for (int i = 0; i < RL; ++i) { const float2 v = red[i * C + c]; ss += v.x; qq += v.y; }
        part[(long)b * C + c] = make_float2(ss, qq);
    }
}
// pass 2 (one block): the NB partials of a channel added in a fixed order (deterministic, no atomics), in double — the group
// statistics are E[x^2] - mean^2 over P * C / 32 values — then per CHANNEL the affine map of the norm: y = x a_c + b_c with
// a_c = rstd_g gamma_c, b_c = beta_c - mean_g a_c
__global__ void vda_gn_finalize_kernel(const float2 *part, const float *__restrict__ gamma, const float *__restrict__ beta,
                                       int P, int C, int NB, float eps) {
    __shared__ double cs[1024], cq[1024];
    const int c = threadIdx.x;
    part += (long)blockIdx.x * (kVdaGnBlocks + 1) * C;                       // blockIdx.x = frame
    float2 *coef = const_cast<float2 *>(part) + (long)kVdaGnBlocks * C;       // behind the frame's partials
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
    const float a = (float)(1.0 / sqrt(var + (double)eps)) * gamma[c];
    coef[c] = make_float2(a, beta[c] - (float)mean * a);
}
// pass 3: elementwise, 8 channels per thread
__global__ void __launch_bounds__(256) vda_gn_apply_kernel(const f16 *__restrict__ x, const float2 *__restrict__ part, f16 *__restrict__ y,
                                                           long total8, int C8) {
    long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total8) return;
    const int o = (int)(id % C8);
    const float2 *coef = part + ((long)blockIdx.y * (kVdaGnBlocks + 1) + kVdaGnBlocks) * C8 * 8;
    id += (long)blockIdx.y * total8;
    const f16x8 v = *reinterpret_cast<const f16x8 *>(x + id * 8);
    f16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float2 ab = coef[o * 8 + e];
        r[e] = (f16)fmaf((float)v[e], ab.x, ab.y);
    }
    *reinterpret_cast<f16x8 *>(y + id * 8) = r;
}

int launch_vda_groupnorm(const f16 *x, const float *gamma, const float *beta, f16 *y, float2 *part, int frames, int P, int C, float eps,
                         hipStream_t s) {
    NUNIF_REQUIRE(frames > 0 && frames <= 65535 && P > 0 && C % 32 == 0 && C >= 32 && C <= 1024,
                  "vda_groupnorm: %d channels unsupported (a multiple of 32, <= 1024)", C);
    const int RL = 256 / (C / 8);                              // row lanes of a block: ~4 rows per lane
    int NB = std::min(kVdaGnBlocks, (P + 4 * RL - 1) / (4 * RL));
    const int R = (P + NB - 1) / NB;
    NB = (P + R - 1) / R;
    ProfScope ps("vda_groupnorm", s, 0.0, (double)frames * P * C * 6.0);
    vda_gn_partial_kernel<<<dim3(NB, frames), 256, 0, s>>>(x, part, P, C, R);
    NUNIF_LAUNCH_CHECK();
    vda_gn_finalize_kernel<<<frames, C, 0, s>>>(part, gamma, beta, P, C, NB, eps);
    NUNIF_LAUNCH_CHECK();
    const long total8 = (long)P * (C / 8);
    vda_gn_apply_kernel<<<dim3((unsigned)((total8 + 255) / 256), frames), 256, 0, s>>>(x, part, y, total8, C / 8);
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

// ---- the same attention, laid out for memory-level parallelism (C <= 384) ----------------------------------------------------------
// The form above is one thread per (pixel, head) walking 2 x 32 x hd / 8 dependent 16-byte loads: P = 1 400 pixels are 175 waves for
// 1 024 SIMDs, every load waited for on its own — 100 us per launch for 17-70 MB (0.28 TB/s).  Here a thread is one (pixel, 8-channel
// CHUNK): hd / 8 times the threads, a pixel's C / 8 chunk-lanes read whole contiguous rows, the window is walked in batches of 8
// unconditional loads (positions beyond the window re-read a valid slot and are masked), and the position tables PK / PV sit in LDS
// (rows 0 .. idx, staged once per workgroup) instead of being re-read from global memory by every thread.  The chunk-lanes of a
// head exchange their partial scores through LDS.  A workgroup = NPX pixels x C / 8 lanes.
template <int C>
__global__ void __launch_bounds__(C <= 64 ? 256 : (C / 8) * (C <= 128 ? 16 : C <= 192 ? 8 : 4)) vda_tattn2_kernel(VdaTattnArgs g) {
    constexpr int CL = C / 8, NPX = C <= 64 ? 32 : C <= 128 ? 16 : C <= 192 ? 8 : 4, NT = CL * NPX, SP = 33;
    __shared__ __attribute__((aligned(16))) float pk[32 * C];
    __shared__ __attribute__((aligned(16))) float pv[32 * C];
    __shared__ float sp[NT * SP];
    const int tid = threadIdx.x, idx = g.idx, nch = g.hd >> 3;
    for (int i = tid; i < (idx + 1) * (C / 4); i += NT) {
        reinterpret_cast<f32x4 *>(pk)[i] = reinterpret_cast<const f32x4 *>(g.pk)[i];
        reinterpret_cast<f32x4 *>(pv)[i] = reinterpret_cast<const f32x4 *>(g.pv)[i];
    }
    const int px = tid / CL, cc = tid - px * CL;
    long p = (long)blockIdx.x * NPX + px;
    const bool live = p < g.P;
    if (!live) p = g.P - 1;
    const long row = p * C + cc * 8, slot = (long)g.P * C;
    const f16 *qrow = g.qkv + p * 3 * C + cc * 8;
    const f16x8 qh = *reinterpret_cast<const f16x8 *>(qrow);
    const f16x8 kcur = *reinterpret_cast<const f16x8 *>(qrow + C);
    const f16x8 vcur = *reinterpret_cast<const f16x8 *>(qrow + 2 * C);
    const int cur = (g.start + idx) & 31;
    if (live) {
        *reinterpret_cast<f16x8 *>(g.kc + cur * slot + row) = kcur;
        *reinterpret_cast<f16x8 *>(g.vc + cur * slot + row) = vcur;
    }
    float q[8];
    {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(g.pq + (long)idx * C + cc * 8);
        const f32x4 b = *reinterpret_cast<const f32x4 *>(g.pq + (long)idx * C + cc * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { q[e] = (float)qh[e] + a[e]; q[4 + e] = (float)qh[4 + e] + b[e]; }
    }
    __syncthreads();                                     // the tables are staged
    // The window is read with UNCONDITIONAL loads, all 32 positions of K in flight at once (P = 1 400 pixels x 8 chunk-lanes are 175
    // waves: the 12-46 MB of a launch only move at HBM speed with ~30 loads per lane outstanding), then all 32 of V, issued BEFORE the
    // score exchange and the softmax so that they fly behind them.  Positions beyond the window (j > idx: only in the first 31 frames
    // after a reset) re-read position jlast — an older valid slot; with an empty window the slot being written, whose value is never
    // used — and are masked by selects, never by arithmetic.
    const int jlast = idx > 0 ? idx - 1 : 0;
    const f16 *kbase = g.kc + row, *vbase = g.vc + row;
    f16x8 kh[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) kh[j] = *reinterpret_cast<const f16x8 *>(kbase + (long)((g.start + min(j, jlast)) & 31) * slot);
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const f16x8 kk = j == idx ? kcur : kh[j];
        const f32x4 ta = *reinterpret_cast<const f32x4 *>(pk + min(j, idx) * C + cc * 8);
        const f32x4 tb = *reinterpret_cast<const f32x4 *>(pk + min(j, idx) * C + cc * 8 + 4);
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc = fmaf(q[e], (float)kk[e] + ta[e], acc);
            acc = fmaf(q[4 + e], (float)kk[4 + e] + tb[e], acc);
        }
        sp[tid * SP + j] = acc;
    }
    f16x8 vh[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) vh[j] = *reinterpret_cast<const f16x8 *>(vbase + (long)((g.start + min(j, jlast)) & 31) * slot);
    __syncthreads();
    // the head's score = the sum over its nch chunk-lanes (every lane of the head computes the same softmax)
    const int base = (px * CL + cc / nch * nch) * SP;
    float s[32];
    float m = -3.0e38f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        float t = 0.f;
        for (int i = 0; i < nch; ++i) t += sp[base + i * SP + j];
        s[j] = t;
        if (j <= idx) m = fmaxf(m, t);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        s[j] = j <= idx ? exp2f(s[j] - m) : 0.f;
        sum += s[j];
    }
    const float inv = 1.0f / sum;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const f16x8 vv = j == idx ? vcur : vh[j];
        const f32x4 ta = *reinterpret_cast<const f32x4 *>(pv + min(j, idx) * C + cc * 8);
        const f32x4 tb = *reinterpret_cast<const f32x4 *>(pv + min(j, idx) * C + cc * 8 + 4);
        // (p_j = 0 for j > idx, but 0 x garbage could be NaN: select the operand as well)
        const float pj = s[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[e] = j <= idx ? fmaf(pj, (float)vv[e] + ta[e], acc[e]) : acc[e];
            acc[4 + e] = j <= idx ? fmaf(pj, (float)vv[4 + e] + tb[e], acc[4 + e]) : acc[4 + e];
        }
    }
    if (live) {
        f16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (f16)(acc[e] * inv);
        *reinterpret_cast<f16x8 *>(g.att + row) = r;
    }
}

template <int C>
static void launch_tattn2(const VdaTattnArgs &g, hipStream_t s) {
    constexpr int NPX = C <= 64 ? 32 : C <= 128 ? 16 : C <= 192 ? 8 : 4;
    vda_tattn2_kernel<C><<<(unsigned)((g.P + NPX - 1) / NPX), (C / 8) * NPX, 0, s>>>(g);
}

int launch_vda_tattn(const VdaTattnArgs &g, hipStream_t s) {
    NUNIF_REQUIRE(g.P > 0 && g.C == 8 * g.hd && g.hd % 8 == 0 && g.idx >= 0 && g.idx < 32 && g.start >= 0 && g.start < 32,
                  "vda_tattn: C=%d hd=%d idx=%d start=%d unsupported (8 heads, head width a multiple of 8, window <= 32)", g.C, g.hd,
                  g.idx, g.start);
    // per launch: the window's K0 / V0 once, this frame's qkv row, the att row
    ProfScope ps("vda_tattn_kernel", s, 4.0 * g.P * (double)g.C * (g.idx + 1), (double)g.P * g.C * 2.0 * (2.0 * g.idx + 3.0 + 2.0 + 1.0));
    // NUNIF_VDA_TATTN=1: the one-thread-per-head form for every width (A/B; the only form for C > 384, where the tables outgrow LDS)
    const bool v1 = getenv("NUNIF_VDA_TATTN") && atoi(getenv("NUNIF_VDA_TATTN")) == 1;      // read per call (tests A/B it)
    if (!v1 && g.C == 64) launch_tattn2<64>(g, s);
    else if (!v1 && g.C == 128) launch_tattn2<128>(g, s);
    else if (!v1 && g.C == 192) launch_tattn2<192>(g, s);
    else if (!v1 && g.C == 256) launch_tattn2<256>(g, s);
    else if (!v1 && g.C == 384) launch_tattn2<384>(g, s);
    else vda_tattn_kernel<<<(unsigned)(((long)g.P * 8 + 255) / 256), 256, 0, s>>>(g);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

}  // namespace nunif
