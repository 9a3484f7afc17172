// iw3 "inpaint.light_inpaint_v1" (the image inpaint net behind MLBWInpaintImage) for gfx950.
//
// Replaces iw3/models/light_inpaint_v1.py LightInpaintV1.infer :106-110 = preprocess :93-104 + forward(skip_i2i_offset) :130-161
// (_forward :112-128, GMLPBlock :37-50, GLUConvMLP :15-34), nunif/modules/attention.py WindowGMLP2d :654-693 / GMLP :621-651,
// nunif/modules/norm.py FastLayerNorm :78-101, nunif/modules/gaussian_filter.py SeparableGaussianFilter2d :52-73.
//
// Data layout: NHWC fp16 token maps (level 1: 1/4 resolution, C = 96, 16 x 16 windows; level 2: 1/8, C = 192, 8 x 8 windows).
// Every Linear / 1x1 / 2x2-stride-2 conv is gemm_kernel, the 3x3 convs are conv_kernel (replicate padding folded into the
// gather); what is new here is the glue a gMLP needs:
//   * the token-mixing step contracts over the TOKENS of a window, i.e. over the strided axis of a token-major map: the
//     LayerNorm of the gate half writes its result transposed per window ([window][channel][token], coalesced because the
//     lanes of a wave walk consecutive tokens), the mixing is then a plain GEMM with K = tokens, and the gate kernel
//     transposes back while it multiplies with the value half;
//   * a shifted block works on the zero-padded map (half a window on every side): the padded tokens take part in the
//     mixing (LayerNorm(0) = 0 but proj_in(0) = bias), so the padded map really exists between ln1 and proj_out.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "swin_kernels.h"

namespace nunif {

// ---- pre-processing -----------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) li_u8_to_f32_kernel(const uint8_t *__restrict__ in, float *__restrict__ out, long n) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id < n) out[id] = in[id] ? 1.f : 0.f;
}

__global__ void __launch_bounds__(256) li_morph_kernel(const float *__restrict__ in, float *__restrict__ out, int B, int H, int W,
                                                        int is_min) {
    const long n = (long)B * H * W, id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    const int x = (int)(id % W);
    const long t = id / W;
    const int y = (int)(t % H);
    const float *p = in + (t / H) * (long)H * W;
    float v = p[(long)y * W + x];
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= W) continue;
            const float u = p[(long)yy * W + xx];
            v = is_min ? fminf(v, u) : fmaxf(v, u);
        }
    }
    out[id] = v;
}

__global__ void __launch_bounds__(256) li_add_clamp_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                            float *__restrict__ out, long n) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id < n) out[id] = fminf(fmaxf(a[id] + b[id], 0.f), 1.f);
}

// mask[x] = max over [x - n_outer, x + n_inner] (dilate_inner then dilate_outer, iw3/dilation.py:74-103; zeros at the border)
__global__ void __launch_bounds__(256) li_dilate_kernel(const float *__restrict__ in, float *__restrict__ out, long rows, int W,
                                                         int n_inner, int n_outer) {
    const long n = rows * W, id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    const int X = (int)(id % W);
    const float *p = in + (id / W) * W;
    float m = 0.f;
    for (int x = max(X - n_outer, 0); x <= min(X + n_inner, W - 1); ++x) m = fmaxf(m, p[x]);
    out[id] = m;
}

struct Gauss15 { float w[15]; };

// one pass of the separable 15-tap gaussian with replicate padding; vertical pass: out = clamp(blur + hard mask, 0, 1)
__global__ void __launch_bounds__(256) li_blur_kernel(const float *__restrict__ in, const float *__restrict__ hard,
                                                       float *__restrict__ out, int B, int H, int W, int vertical, Gauss15 g) {
    const long n = (long)B * H * W, id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    const int x = (int)(id % W);
    const long t = id / W;
    const int y = (int)(t % H);
    const float *p = in + (t / H) * (long)H * W;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        const int yy = vertical ? min(max(y + k - 7, 0), H - 1) : y, xx = vertical ? x : min(max(x + k - 7, 0), W - 1);
        acc += p[(long)yy * W + xx] * g.w[k];
    }
    out[id] = vertical ? fminf(fmaxf(acc + hard[id], 0.f), 1.f) : acc;
}

// ---- patch embedding input: (x (1 - m) - 0.5) / 0.5, replicate padding to the 64-pixel grid, pixel_unshuffle(4) -----------
// out: [B,h1,w1,64] fp16 (48 real channels c*16 + iy*4 + ix, rest 0); mtok: the token is masked if any of its 16 pixels of the
// soft mask exceeds 0.99 (light_inpaint_v1.py:116)
// mirror: the picture x is read (and, in li_compose_kernel, written) at column W - 1 - xx — the net sees flip(x) without a flip pass;
// the masks are in the mirrored frame already
__global__ void __launch_bounds__(256) li_patch_in_kernel(const float *__restrict__ x, const float *__restrict__ hard,
                                                           const float *__restrict__ soft, f16 *__restrict__ out,
                                                           uint8_t *__restrict__ mtok, int B, int H, int W, int h1, int w1,
                                                           int mirror) {
    const long n = (long)B * h1 * w1, id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    const int tx = (int)(id % w1);
    const long t = id / w1;
    const int ty = (int)(t % h1), b = (int)(t / h1);
    f16 v[64];
#pragma unroll
    for (int i = 48; i < 64; ++i) v[i] = (f16)0.f;
    float mmax = 0.f;
    for (int iy = 0; iy < 4; ++iy) {
        const int yy = min(ty * 4 + iy, H - 1);
        for (int ix = 0; ix < 4; ++ix) {
            const int xx = min(tx * 4 + ix, W - 1);
            const long pix = ((long)b * H + yy) * W + xx;
            const float keep = 1.f - hard[pix];
            mmax = fmaxf(mmax, soft[pix]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float px = x[(((long)b * 3 + c) * H + yy) * W + (mirror ? W - 1 - xx : xx)] * keep;
                v[c * 16 + iy * 4 + ix] = (f16)((px - 0.5f) / 0.5f);
            }
        }
    }
    f16x8 *o = reinterpret_cast<f16x8 *>(out + id * 64);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (f16x8){v[8 * i], v[8 * i + 1], v[8 * i + 2], v[8 * i + 3], v[8 * i + 4], v[8 * i + 5],
                                               v[8 * i + 6], v[8 * i + 7]};
    mtok[id] = mmax > 0.99f ? 1 : 0;
}

__global__ void __launch_bounds__(256) li_mask_bias_kernel(f16 *__restrict__ x, const uint8_t *__restrict__ mtok,
                                                            const f16 *__restrict__ bias, long tokens, int C) {
    const int runs = C / 8;
    const long n = tokens * runs, id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    const long tok = id / runs;
    const int r = (int)(id - tok * runs);
    if (mtok[tok]) reinterpret_cast<f16x8 *>(x)[id] = reinterpret_cast<const f16x8 *>(bias)[r];
}

// ---- gMLP glue --------------------------------------------------------------------------------------------------------------
// LayerNorm(C, eps 1e-5, weight only) of every token, written into the zero-padded map (pad = half a window, or 0)
template <int C>
__global__ void __launch_bounds__(256) li_ln_pad_kernel(const f16 *__restrict__ x, const float *__restrict__ w,
                                                         f16 *__restrict__ out, int B, int h, int wd, int pad) {
    const int hp = h + 2 * pad, wp = wd + 2 * pad;
    const long n = (long)B * hp * wp, id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    const int px = (int)(id % wp);
    const long t = id / wp;
    const int py = (int)(t % hp), b = (int)(t / hp);
    const int y = py - pad, xx = px - pad;
    f16x8 *o = reinterpret_cast<f16x8 *>(out + id * C);
    if (y < 0 || y >= h || xx < 0 || xx >= wd) {
        const f16x8 z = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
#pragma unroll
        for (int i = 0; i < C / 8; ++i) o[i] = z;
        return;
    }
    const f16x8 *p = reinterpret_cast<const f16x8 *>(x + (((long)b * h + y) * wd + xx) * C);
    // HOLD: the token's channel vector stays in registers between the three passes; wider tokens (the medium / large video
    // nets) are re-read from the cache instead of spilling
    constexpr bool HOLD = C <= 256;
    constexpr int UNR = HOLD ? C / 8 : 8;          // HOLD needs the loops fully unrolled (register array)
    f16x8 v[HOLD ? C / 8 : 1];
    float sum = 0.f;
#pragma unroll UNR
    for (int i = 0; i < C / 8; ++i) {
        const f16x8 t = p[i];
        if constexpr (HOLD) v[i] = t;
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += (float)t[j];
    }
    const float mean = sum / (float)C;
    float var = 0.f;
#pragma unroll UNR
    for (int i = 0; i < C / 8; ++i) {
        const f16x8 t = HOLD ? v[HOLD ? i : 0] : p[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = (float)t[j] - mean; var += d * d; }
    }
    const float rs = rsqrtf(var / (float)C + 1e-5f);
#pragma unroll UNR
    for (int i = 0; i < C / 8; ++i) {
        const f16x8 t = HOLD ? v[HOLD ? i : 0] : p[i];
        f16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (f16)(((float)t[j] - mean) * rs * w[8 * i + j]);
        o[i] = r;
    }
}

// LayerNorm over the gate half v = a[:, C2:2*C2] of proj_in's output (C2 = 2C channels), written TRANSPOSED per window:
// vt[window][channel][token].  Thread id = (window, token in window): the 64 lanes of a wave hold consecutive tokens, so each
// channel's store is one contiguous 128-byte run.
template <int C2>
__global__ void __launch_bounds__(256) li_ln2_t_kernel(const f16 *__restrict__ a, const float *__restrict__ w, f16 *__restrict__ vt,
                                                        int B, int hp, int wp, int ws) {
    const int N = ws * ws, nwx = wp / ws, nwy = hp / ws;
    const long n = (long)B * hp * wp, id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    const long win = id / N;
    const int tn = (int)(id - win * N);
    const int wx = (int)(win % nwx);
    const long t2 = win / nwx;
    const int wy = (int)(t2 % nwy), b = (int)(t2 / nwy);
    const long tok = ((long)b * hp + wy * ws + tn / ws) * wp + wx * ws + tn % ws;
    const f16x8 *p = reinterpret_cast<const f16x8 *>(a + tok * (2 * C2) + C2);
    constexpr bool HOLD = C2 <= 256;
    constexpr int UNR = HOLD ? C2 / 8 : 8;
    f16x8 v[HOLD ? C2 / 8 : 1];
    float sum = 0.f;
#pragma unroll UNR
    for (int i = 0; i < C2 / 8; ++i) {
        const f16x8 t = p[i];
        if constexpr (HOLD) v[i] = t;
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += (float)t[j];
    }
    const float mean = sum / (float)C2;
    float var = 0.f;
#pragma unroll UNR
    for (int i = 0; i < C2 / 8; ++i) {
        const f16x8 t = HOLD ? v[HOLD ? i : 0] : p[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = (float)t[j] - mean; var += d * d; }
    }
    const float rs = rsqrtf(var / (float)C2 + 1e-5f);
    f16 *o = vt + (win * C2) * N + tn;
#pragma unroll UNR
    for (int i = 0; i < C2 / 8; ++i) {
        const f16x8 t = HOLD ? v[HOLD ? i : 0] : p[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[(long)(8 * i + j) * N] = (f16)(((float)t[j] - mean) * rs * w[8 * i + j]);
    }
}

// temporal block (light_video_inpaint_v1.py GMLP3DBlock, window (12,1,1)): LayerNorm over the gate half, token-major.
// 32 lanes per token, 16 bytes per lane: a token's V channels are one contiguous run of the row, so a wave's load / store
// instruction covers two whole runs.  (Rounds 1-3 gave every THREAD a token: the 64 lanes of a load touched 64 rows 4 V bytes
// apart — 0.55 TB/s = 7 % of the HBM peak, 4.5 ms per launch in the config-5 leg.)  Two-pass statistics in fp32 as before; the
// sums are now reduced across lanes, i.e. in another order.
template <int V>
__global__ void __launch_bounds__(256) li_ln2_kernel(const f16 *__restrict__ a, const float *__restrict__ w, f16 *__restrict__ vn,
                                                      long tokens) {
    constexpr int LPT = 32, SEGS = V / 8, PER = (SEGS + LPT - 1) / LPT;
    const long id = (long)blockIdx.x * (256 / LPT) + (threadIdx.x / LPT);
    const int l = threadIdx.x % LPT;
    if (id >= tokens) return;                                   // (a 32-lane group leaves together)
    const f16x8 *p = reinterpret_cast<const f16x8 *>(a + id * (2 * V) + V);
    f16x8 v[PER];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int sg = l + LPT * i;
        v[i] = sg < SEGS ? p[sg] : (f16x8){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += (float)v[i][j];
    }
#pragma unroll
    for (int m = LPT / 2; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
    const float mean = sum / (float)V;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        if (l + LPT * i < SEGS) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = (float)v[i][j] - mean; var += d * d; }
        }
    }
#pragma unroll
    for (int m = LPT / 2; m >= 1; m >>= 1) var += __shfl_xor(var, m);
    const float rs = rsqrtf(var / (float)V + 1e-5f);
    f16x8 *o = reinterpret_cast<f16x8 *>(vn + id * V);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int sg = l + LPT * i;
        if (sg < SEGS) {
            f16x8 r;
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = (f16)(((float)v[i][j] - mean) * rs * w[8 * sg + j]);
            o[sg] = r;
        }
    }
}

// g[t][p][c] = u[t][p][c] * (b[t] + sum_s W[t][s] vn[s][p][c]) over the T = 12 frames of pixel p; thread = (pixel, 8 channels)
struct TMix { float w[12][12]; float b[12]; };
template <int V>
__global__ void __launch_bounds__(256) li_tmix_gate_kernel(const f16 *__restrict__ a, const f16 *__restrict__ vn,
                                                            f16 *__restrict__ g, long pixels, TMix m) {
    constexpr int T = 12, runs = V / 8;
    const long n = pixels * runs, id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    const long p = id / runs;
    const int c0 = (int)(id - p * runs) * 8;
    float v[T][8];
#pragma unroll
    for (int s = 0; s < T; ++s) {
        const f16x8 x = *reinterpret_cast<const f16x8 *>(vn + ((long)s * pixels + p) * V + c0);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[s][j] = (float)x[j];
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = m.b[t];
#pragma unroll
        for (int s = 0; s < T; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(m.w[t][s], v[s][j], acc[j]);
        const long tok = (long)t * pixels + p;
        const f16x8 u = *reinterpret_cast<const f16x8 *>(a + tok * (2 * V) + c0);
        f16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (f16)((float)u[j] * acc[j]);
        *reinterpret_cast<f16x8 *>(g + tok * V + c0) = r;
    }
}

// g[token][c] = u[token][c] * st[window][c][token]  (u = first half of proj_in's output, st = the mixed gate, transposed)
template <int C2>
__global__ void __launch_bounds__(256) li_gate_kernel(const f16 *__restrict__ a, const f16 *__restrict__ st, f16 *__restrict__ g,
                                                       int B, int hp, int wp, int ws) {
    // g[token][c] = u[token][c] * s[window][c][token in window]: the gate s comes out of the token-mixing GEMM channel-major, the
    // tokens are channel-minor.  A workgroup owns (window, 64 channels): the 64 x N gate tile goes through LDS (coalesced in, read
    // transposed), so that u is read and g written as whole 128-byte runs per token.  A thread per token moved 16 bytes into 64
    // different lines per instruction (WRITE_SIZE 3.2x the output, 2 006 us); a thread per 16-byte run with the gate gathered
    // straight from L2 1 845 us (profiles/r04f_pmc_*).
    constexpr int CB = 64, NCB = C2 / CB;
    extern __shared__ __attribute__((aligned(16))) unsigned char li_gate_smem[];
    f16 *sl = reinterpret_cast<f16 *>(li_gate_smem);                       // [64][N + 2]: the + 2 spreads the transposed reads over banks
    const int N = ws * ws, nwx = wp / ws, nwy = hp / ws, ld = N + 2;
    const long win = blockIdx.x / NCB;
    const int cb = blockIdx.x % NCB;
    const int wx = (int)(win % nwx);
    const long t2 = win / nwx;
    const int wy = (int)(t2 % nwy), b = (int)(t2 / nwy);
    const int tid = threadIdx.x;
    const f16 *sg = st + (win * C2 + cb * CB) * N;
    for (int p = tid; p < CB * N / 8; p += 256) {                           // 16-byte pieces: channel p / (N / 8), tokens 8 (p % (N / 8)) ..
        const int c = p / (N / 8), off = p % (N / 8);
        const f16x8 v = *reinterpret_cast<const f16x8 *>(sg + (long)c * N + 8 * off);
        unsigned int *d = reinterpret_cast<unsigned int *>(sl + c * ld + 8 * off);     // ld is even: 4-byte aligned
        const unsigned int *w = reinterpret_cast<const unsigned int *>(&v);
        d[0] = w[0]; d[1] = w[1]; d[2] = w[2]; d[3] = w[3];
    }
    __syncthreads();
    for (int it = tid; it < N * (CB / 8); it += 256) {
        const int ch = it & 7, tn = it >> 3;                                // 8 lanes = the 128 bytes of one token's 64 channels
        const long tok = ((long)b * hp + wy * ws + tn / ws) * wp + wx * ws + tn % ws;
        const f16x8 uv = *reinterpret_cast<const f16x8 *>(a + tok * (2 * C2) + cb * CB + 8 * ch);
        f16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (f16)((float)uv[j] * (float)sl[(8 * ch + j) * ld + tn]);
        *reinterpret_cast<f16x8 *>(g + tok * C2 + cb * CB + 8 * ch) = r;
    }
}

// x = x + crop(proj_out(...) + shortcut) with shortcut = the (padded) block input: x <- 2 x + crop(po)
__global__ void __launch_bounds__(256) li_crop_add_kernel(f16 *__restrict__ x, const f16 *__restrict__ po, int B, int h, int wd,
                                                           int pad, int C) {
    const int runs = C / 8, wp = wd + 2 * pad, hp = h + 2 * pad;
    const long n = (long)B * h * wd * runs, id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    const long tok = id / runs;
    const int r = (int)(id - tok * runs);
    const int xx = (int)(tok % wd);
    const long t = tok / wd;
    const int y = (int)(t % h), b = (int)(t / h);
    const long ptok = ((long)b * hp + y + pad) * wp + xx + pad;
    const f16x8 a = reinterpret_cast<const f16x8 *>(x)[id], p = reinterpret_cast<const f16x8 *>(po + ptok * C)[r];
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)(2.f * (float)a[j] + (float)p[j]);
    reinterpret_cast<f16x8 *>(x)[id] = o;
}

// F.glu(y, dim = channels): z[c] = y[c] * sigmoid(y[C/2 + c]); z is stored padded to Cz channels (zeros) for the 3x3 conv
__global__ void __launch_bounds__(256) li_glu_kernel(const f16 *__restrict__ y, f16 *__restrict__ z, long tokens, int C, int Cz) {
    const int runs = Cz / 8, half = C / 2;
    const long n = tokens * runs, id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    const long tok = id / runs;
    const int c0 = (int)(id - tok * runs) * 8;
    f16x8 o = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    if (c0 < half) {
        const f16x8 a = *reinterpret_cast<const f16x8 *>(y + tok * C + c0), g = *reinterpret_cast<const f16x8 *>(y + tok * C + half + c0);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)((float)a[j] * (1.f / (1.f + expf(-(float)g[j]))));
    }
    reinterpret_cast<f16x8 *>(z)[id] = o;
}

// out = clamp(src (1 - m) + y m, 0, 1): src = x (1 - hard mask), m = soft mask, y = pixel_shuffle(4) of the net's 48 channels
__global__ void __launch_bounds__(256) li_compose_kernel(const float *__restrict__ x, const float *__restrict__ hard,
                                                          const float *__restrict__ soft, const f16 *__restrict__ ti,
                                                          float *__restrict__ out, int B, int H, int W, int h1, int w1,
                                                          int ti_stride, int mirror) {
    const long n = (long)B * H * W, id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    const int xx = (int)(id % W);
    const long t = id / W;
    const int yy = (int)(t % H), b = (int)(t / H);
    const float keep = 1.f - hard[id], m = soft[id];
    const f16 *tv = ti + (((long)b * h1 + yy / 4) * w1 + xx / 4) * ti_stride + (yy & 3) * 4 + (xx & 3);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const long o = (((long)b * 3 + c) * H + yy) * W + (mirror ? W - 1 - xx : xx);
        const float src = x[o] * keep;
        out[o] = fminf(fmaxf(src * (1.f - m) + (float)tv[c * 16] * m, 0.f), 1.f);
    }
}

}  // namespace nunif

using namespace nunif;

// =====================================================================================================================
// host side
// =====================================================================================================================
namespace {

struct HostT { const float *data; std::vector<int64_t> shape; int64_t numel; };
typedef std::map<std::string, HostT> TMap;

int find(const TMap &m, const std::string &key, const HostT **out) {
    auto it = m.find(key);
    if (it == m.end()) { set_error("state_dict is missing '%s'", key.c_str()); return NUNIF_HIP_EMISSING; }
    *out = &it->second;
    return NUNIF_HIP_OK;
}

struct Buf {
    void *p = nullptr; size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return NUNIF_HIP_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        if (hipMalloc(&p, bytes) != hipSuccess) { set_error("hipMalloc(%zu) failed", bytes); return NUNIF_HIP_ENOMEM; }
        cap = bytes;
        return NUNIF_HIP_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct Lin { f16 *w = nullptr; float *bias = nullptr; int N = 0, K = 0; };                 // gemm_kernel packing [nt][ks]
struct Conv3 {                                                  // conv_kernel stream [ks][nt]
    f16 *stream = nullptr; float *bias = nullptr; int N = 0, n_real = 0, Cin = 0;
    // the same weights as streams of OUTPUT-CHANNEL SLICES whose tile counts the LDS-staged convs take (conv3_dma: Cin 32 / 64,
    // 1 / 2 / 4 / 8 tiles; conv3_lds: Cin <= 128, the same counts): 96 = 64 + 32, 192 = 128 + 64.  A slice writes its channels
    // of the NHWC map through ConvArgs::ldo; the input patch is staged once per slice.
    f16 *slice_stream[2] = {nullptr, nullptr}; int slice_nt[2] = {0, 0}, n_slices = 0;
};
struct GBlock {
    int C = 0, V = 0, ws = 0, shift = 0, temporal = 0;      // V = gate / value width (mlp_ratio * C); temporal: window (12,1,1)
    TMix tmix;
    float *ln1 = nullptr, *ln2 = nullptr;
    Lin proj_in, spatial, proj_out, w1;
    Conv3 w2;
};

}  // namespace

struct nunif_light_inpaint {
    std::vector<void *> owned;
    f16 *mask_bias = nullptr;
    Lin patch, down, up;
    // video = inpaint.light_video_inpaint_v1: patch slope 0.1, enc1 unshifted, enc2 = [2-D, temporal, 2-D, temporal, 2-D] with
    // mlp_ratio 1 in the 2-D blocks, 1x1 to_image; exactly 12 frames per call
    int video = 0, n_enc2 = 4;
    int C = 96;               // base_dim: 96 (image net, video small), 128 (video medium), 192 (video large)
    GBlock enc1, enc2[5], dec1;
    Conv3 to_image;
    Lin to_image1;
    Gauss15 gauss;
    Buf x1, x2, a, pi, vt, st, g, po, y, z, ti, mtok, mf0, mf1, mf2, hard, soft;
};

namespace {

template <typename T>
int upload(nunif_light_inpaint *h, const std::vector<T> &host, T **dev) {
    void *p = nullptr;
    if (hipMalloc(&p, host.size() * sizeof(T)) != hipSuccess) { set_error("hipMalloc failed"); return NUNIF_HIP_ENOMEM; }
    h->owned.push_back(p);
    NUNIF_HIP_CHECK(hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *dev = reinterpret_cast<T *>(p);
    return NUNIF_HIP_OK;
}

// W[n][k] (n < n_real, k < K) -> MFMA A fragments in [n-tile][k-step] order (+ 16 KiB of zeros for the ring prefetch)
template <typename F>
int make_lin(nunif_light_inpaint *h, int n_real, int K, F wt, const std::vector<float> &bias, Lin *L) {
    const int N = (n_real + 31) / 32 * 32, KS = K / 32;
    std::vector<f16> packed((size_t)N * K + 8192, (f16)0.f);
    for (int nt = 0; nt < N / 16; ++nt)
        for (int ks = 0; ks < KS; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int n = nt * 16 + (l & 15), k = ks * 32 + (l >> 4) * 8 + j;
                    packed[(((size_t)nt * KS + ks) * 64 + l) * 8 + j] = (f16)(n < n_real ? wt(n, k) : 0.f);
                }
    std::vector<float> b(N, 0.f);
    std::copy(bias.begin(), bias.end(), b.begin());
    L->N = N; L->K = K;
    int rc = upload(h, packed, &L->w);
    return rc ? rc : upload(h, b, &L->bias);
}

// 3x3 conv weight [cout][cin_real][3][3] -> conv_kernel stream [k-step][n-tile], k = tap * cin + ci (cin = cin_real padded)
int make_conv3(nunif_light_inpaint *h, const TMap &m, const std::string &key, int cin_real, int cin, int cout, Conv3 *c) {
    const HostT *w, *b;
    int rc;
    if ((rc = find(m, key + ".weight", &w)) || (rc = find(m, key + ".bias", &b))) return rc;
    NUNIF_REQUIRE(w->numel == (int64_t)cout * cin_real * 9 && b->numel == cout, "%s: unexpected shape", key.c_str());
    const int N = (cout + 31) / 32 * 32, NT = N / 16, KS = 9 * cin / 32;
    std::vector<f16> stream((size_t)KS * NT * 512 + 8192, (f16)0.f);
    for (int ks = 0; ks < KS; ++ks)
        for (int nt = 0; nt < NT; ++nt)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int n = nt * 16 + (l & 15), kk = ks * 32 + (l >> 4) * 8 + j;
                    const int tap = kk / cin, ci = kk % cin;
                    const float v = (n < cout && ci < cin_real) ? w->data[((size_t)n * cin_real + ci) * 9 + tap] : 0.f;
                    stream[(((size_t)ks * NT + nt) * 64 + l) * 8 + j] = (f16)v;
                }
    std::vector<float> bias(N, 0.f);
    std::copy(b->data, b->data + cout, bias.begin());
    c->N = N; c->n_real = cout; c->Cin = cin;
    if ((rc = upload(h, stream, &c->stream))) return rc;
    if ((NT == 6 || NT == 12) && cout == N && cin <= 128) {
        const int parts[2] = {NT == 6 ? 4 : 8, NT == 6 ? 2 : 4};
        int nt0 = 0;
        for (int q = 0; q < 2; ++q) {
            const int nts = parts[q];
            std::vector<f16> sl((size_t)KS * nts * 512 + 8192, (f16)0.f);
            for (int ks = 0; ks < KS; ++ks)
                for (int nt = 0; nt < nts; ++nt)
                    std::copy(stream.begin() + ((size_t)ks * NT + nt0 + nt) * 512, stream.begin() + ((size_t)ks * NT + nt0 + nt + 1) * 512,
                              sl.begin() + ((size_t)ks * nts + nt) * 512);
            if ((rc = upload(h, sl, &c->slice_stream[q]))) return rc;
            c->slice_nt[q] = nts;
            nt0 += nts;
        }
        c->n_slices = 2;
    }
    return upload(h, bias, &c->bias);
}

int make_plain(nunif_light_inpaint *h, const TMap &m, const std::string &key, int n_real, int K, Lin *L) {
    const HostT *w, *b;
    int rc;
    if ((rc = find(m, key + ".weight", &w)) || (rc = find(m, key + ".bias", &b))) return rc;
    NUNIF_REQUIRE(w->numel == (int64_t)n_real * K && b->numel == n_real, "%s: unexpected shape", key.c_str());
    const float *wd = w->data;
    return make_lin(h, n_real, K, [=](int n, int k) { return wd[(size_t)n * K + k]; },
                    std::vector<float>(b->data, b->data + n_real), L);
}

int make_gblock(nunif_light_inpaint *h, const TMap &m, const std::string &p, int C, int ws, int shift, GBlock *g,
                int ratio = 2, int temporal = 0) {
    const HostT *n1, *n2;
    int rc;
    if ((rc = find(m, p + "norm1.weight", &n1)) || (rc = find(m, p + "norm2.weight", &n2))) return rc;
    // ratio <= 0: take the gate width from the checkpoint (lv2_mlp_ratio of the video nets: 1 small, 2 medium / large)
    const int V = ratio > 0 ? ratio * C : (int)n2->numel;
    g->C = C; g->V = V; g->ws = ws; g->shift = shift; g->temporal = temporal;
    const int N = temporal ? 12 : ws * ws;
    NUNIF_REQUIRE(n1->numel == C && n2->numel == V && (V == C || V == 2 * C), "%s: LayerNorm shapes", p.c_str());
    if ((rc = upload(h, std::vector<float>(n1->data, n1->data + C), &g->ln1)) ||
        (rc = upload(h, std::vector<float>(n2->data, n2->data + V), &g->ln2)))
        return rc;
    if (temporal) {
        const HostT *sw, *sb;
        if ((rc = find(m, p + "gmlp.gmlp.proj_spatial.weight", &sw)) || (rc = find(m, p + "gmlp.gmlp.proj_spatial.bias", &sb)))
            return rc;
        NUNIF_REQUIRE(sw->numel == 144 && sb->numel == 12, "%s: temporal mixing matrix must be 12 x 12", p.c_str());
        for (int t = 0; t < 12; ++t) {
            g->tmix.b[t] = sb->data[t];
            for (int s2 = 0; s2 < 12; ++s2) g->tmix.w[t][s2] = sw->data[t * 12 + s2];
        }
    } else if ((rc = make_plain(h, m, p + "gmlp.gmlp.proj_spatial", N, N, &g->spatial))) {   // Conv1d(N, N, 1): weight [N][N][1]
        return rc;
    }
    if ((rc = make_plain(h, m, p + "gmlp.gmlp.proj_in", 2 * V, C, &g->proj_in)) ||
        (rc = make_plain(h, m, p + "gmlp.gmlp.proj_out", C, V, &g->proj_out)) ||
        (rc = make_plain(h, m, p + "glu_conv.w1", C, C, &g->w1)))
        return rc;
    return make_conv3(h, m, p + "glu_conv.w2", C / 2, (C / 2 + 31) / 32 * 32, C, &g->w2);
}

// A 3x3 conv whose output width is not a tile count the LDS-staged convs take, as two launches over channel slices (Conv3).
static inline int li_conv_slices() { const char *e = getenv("NUNIF_LI_CONV_SLICES"); return e ? atoi(e) : 1; }
int launch_conv3_sliced(const Conv3 &c, const ConvArgs &cv, hipStream_t s) {
    if (!c.n_slices || !li_conv_slices() || cv.n_real != c.N) return launch_conv(cv, s);
    ConvArgs t = cv;
    t.N = c.slice_nt[0] * 16; t.n_real = t.N; t.wstream = c.slice_stream[0];
    if (!conv3_dma_applies(t) && !conv3_lds_applies(t)) return launch_conv(cv, s);
    int n0 = 0, rc;
    for (int q = 0; q < c.n_slices; ++q) {
        t = cv;
        t.N = c.slice_nt[q] * 16; t.n_real = t.N; t.wstream = c.slice_stream[q]; t.bias = cv.bias + n0;
        t.ldo = cv.ldo > 0 ? cv.ldo : cv.n_real;
        t.out = cv.out + n0;
        if (cv.res) t.res = cv.res + n0;
        if (cv.res2) t.res2 = cv.res2 + n0;
        if ((rc = launch_conv(t, s))) return rc;
        n0 += t.N;
    }
    return NUNIF_HIP_OK;
}

int lin(const Lin &L, const f16 *a, long rows, int n_real, int act, float slope, const f16 *res, f16 *out, hipStream_t s,
        const char *tag) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.a = a; g.B = 1; g.Hi = 1; g.Wi = (int)rows; g.Cin = L.K; g.Ho = 1; g.Wo = (int)rows; g.stride = 1; g.kw = 1;
    g.K = L.K; g.w = L.w; g.bias = L.bias; g.N = L.N; g.mode = 0; g.act = act; g.slope = slope; g.res = res; g.out = out;
    g.ldo = n_real; g.n_real = n_real; g.ps = 1;
    // K = 192 -> 768 over millions of tokens (proj_in of the 1/8-resolution blocks at 4K): 288 KiB of weights do not fit the
    // resident-weight GEMM's LDS, and the ring form runs it at 1.1 TB/s (2.8 ms).  Two launches over the output halves (the packed
    // weights are n-tile major: the second half is an offset) read the rows twice and still finish in less than half the time.
    const size_t wbytes = (size_t)(L.N / 16) * (L.K / 32) * 1024;
    if (L.K == 192 && rows >= gemm_big_m() && wbytes > 144 * 1024 && wbytes <= 288 * 1024 && L.N % 64 == 0 && n_real == L.N) {
        int rc;
        g.N = L.N / 2; g.n_real = L.N / 2;
        if ((rc = launch_gemm(g, s, tag))) return rc;
        g.w = L.w + (size_t)(L.N / 32) * (L.K / 32) * 512; g.bias = L.bias + L.N / 2; g.out = out + L.N / 2;
        if (res) g.res = res + L.N / 2;
        return launch_gemm(g, s, tag);
    }
    return launch_gemm(g, s, tag);
}

template <int C, int V>
int run_gblock(nunif_light_inpaint *h, const GBlock &g, f16 *x, int B, int hh, int ww, hipStream_t s) {
    const int pad = (g.shift && !g.temporal) ? g.ws / 2 : 0, hp = hh + 2 * pad, wp = ww + 2 * pad, N = g.ws * g.ws;
    const long tok = (long)B * hh * ww, tokp = (long)B * hp * wp, wins = g.temporal ? 0 : tokp / N;
    f16 *a = (f16 *)h->a.p, *pi = (f16 *)h->pi.p, *vt = (f16 *)h->vt.p, *st = (f16 *)h->st.p, *gg = (f16 *)h->g.p;
    f16 *po = (f16 *)h->po.p, *y = (f16 *)h->y.p, *z = (f16 *)h->z.p;
    const unsigned bp = (unsigned)((tokp + 255) / 256);
    int rc;
    {
        ProfScope ps("li_ln_pad_kernel", s, 0.0, (double)tokp * C * 4.0);
        li_ln_pad_kernel<C><<<bp, 256, 0, s>>>(x, g.ln1, a, B, hh, ww, pad);
    }
    if ((rc = lin(g.proj_in, a, tokp, 2 * V, 1, 0.f, nullptr, pi, s, "li_proj_in"))) return rc;
    if (g.temporal) {
        NUNIF_REQUIRE(B == 12, "light_inpaint: the temporal block needs exactly 12 frames (got %d)", B);
        const long pixels = (long)hh * ww;
        {
            ProfScope ps("li_ln2_kernel", s, 0.0, (double)tok * V * 4.0);
            li_ln2_kernel<V><<<(unsigned)((tok + 7) / 8), 256, 0, s>>>(pi, g.ln2, vt, tok);
        }
        const long n = pixels * (V / 8);
        ProfScope ps("li_tmix_gate_kernel", s, 2.0 * 144.0 * pixels * V, (double)tok * V * 6.0);
        li_tmix_gate_kernel<V><<<(unsigned)((n + 255) / 256), 256, 0, s>>>(pi, vt, gg, pixels, g.tmix);
    } else {
        {
            ProfScope ps("li_ln2_t_kernel", s, 0.0, (double)tokp * V * 4.0);
            li_ln2_t_kernel<V><<<bp, 256, 0, s>>>(pi, g.ln2, vt, B, hp, wp, g.ws);
        }
        // token mixing: rows = (window, channel), K = tokens of the window
        if ((rc = lin(g.spatial, vt, wins * V, N, 0, 0.f, nullptr, st, s, "li_spatial"))) return rc;
        ProfScope ps("li_gate_kernel", s, 0.0, (double)tokp * V * 6.0);
        li_gate_kernel<V><<<(unsigned)(wins * (V / 64)), 256, (size_t)64 * (N + 2) * sizeof(f16), s>>>(pi, st, gg, B, hp, wp, g.ws);
    }
    if ((rc = lin(g.proj_out, gg, tokp, C, 0, 0.f, nullptr, po, s, "li_proj_out"))) return rc;
    {
        const long n = tok * (C / 8);
        ProfScope ps("li_crop_add_kernel", s, 0.0, (double)tok * C * 6.0);
        li_crop_add_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(x, po, B, hh, ww, pad, C);
    }
    // GLUConvMLP: 1x1 -> GLU -> replicate pad + 3x3, residual
    if ((rc = lin(g.w1, x, tok, C, 0, 0.f, nullptr, y, s, "li_glu_w1"))) return rc;
    {
        const long n = tok * (g.w2.Cin / 8);
        ProfScope ps("li_glu_kernel", s, 0.0, (double)tok * C * 3.0);
        li_glu_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(y, z, tok, C, g.w2.Cin);
    }
    ConvArgs cv;
    memset(&cv, 0, sizeof(cv));
    cv.a = z; cv.B = B; cv.Hi = hh; cv.Wi = ww; cv.Cin = g.w2.Cin; cv.Ho = hh; cv.Wo = ww; cv.stride = 1; cv.kh = 3; cv.kw = 3;
    cv.wstream = g.w2.stream; cv.bias = g.w2.bias; cv.N = g.w2.N; cv.n_real = C; cv.act = 0; cv.out = x; cv.rpad = 1; cv.res = x;
    if ((rc = launch_conv3_sliced(g.w2, cv, s))) return rc;
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

// (C, V) pairs of the registered nets: image / video-small 96: (96,192) (192,192) (192,384); video-medium 128: (128,256)
// (256,512); video-large 192: (192,384) (384,768)
int run_gblock_any(nunif_light_inpaint *h, const GBlock &g, f16 *x, int B, int hh, int ww, hipStream_t s) {
    const int key = g.C * 1000 + g.V;
    switch (key) {
        case 96192: return run_gblock<96, 192>(h, g, x, B, hh, ww, s);
        case 192192: return run_gblock<192, 192>(h, g, x, B, hh, ww, s);
        case 192384: return run_gblock<192, 384>(h, g, x, B, hh, ww, s);
        case 128256: return run_gblock<128, 256>(h, g, x, B, hh, ww, s);
        case 256512: return run_gblock<256, 512>(h, g, x, B, hh, ww, s);
        case 384768: return run_gblock<384, 768>(h, g, x, B, hh, ww, s);
        default:
            set_error("light_inpaint: block with C = %d, V = %d is not a registered variant", g.C, g.V);
            return NUNIF_HIP_EUNSUPPORTED;
    }
}

}  // namespace

extern "C" int nunif_hip_light_inpaint_create(const nunif_tensor_desc *tensors, int32_t n_tensors, nunif_light_inpaint **handle) {
    NUNIF_REQUIRE(tensors && handle && n_tensors > 0, "light_inpaint_create: NULL argument");
    TMap m;
    for (int i = 0; i < n_tensors; ++i) {
        HostT t;
        t.data = tensors[i].data;
        t.numel = 1;
        for (int d = 0; d < tensors[i].ndim; ++d) { t.shape.push_back(tensors[i].shape[d]); t.numel *= tensors[i].shape[d]; }
        m[tensors[i].name] = t;
    }
    nunif_light_inpaint *h = new nunif_light_inpaint();
    int rc = NUNIF_HIP_OK;
    do {
        const HostT *mb, *pw, *pb, *dw, *db, *uw, *ub;
        // inpaint.light_video_inpaint_v1 has `patch` = Conv2d(3, 96, 4, 4) (same k = c*16 + ky*4 + kx order as the
        // pixel_unshuffle(4) + 1x1 conv of the image net) and a 1x1 `to_image`
        h->video = m.count("patch.weight") ? 1 : 0;
        const std::string pk = h->video ? "patch" : "patch.0";
        if ((rc = find(m, "mask_bias", &mb)) || (rc = find(m, pk + ".weight", &pw)) || (rc = find(m, pk + ".bias", &pb)) ||
            (rc = find(m, "down.weight", &dw)) || (rc = find(m, "down.bias", &db)) || (rc = find(m, "up.weight", &uw)) ||
            (rc = find(m, "up.bias", &ub)))
            break;
        const int C = (int)mb->numel, C2 = 2 * C;
        if ((C != 96 && !(h->video && (C == 128 || C == 192))) || pw->numel != (int64_t)C * 48 ||
            dw->numel != (int64_t)C2 * C * 4 || uw->numel != (int64_t)4 * C * C2) {
            set_error("light_inpaint: unexpected shapes (base_dim 96; 128 / 192 for the video net)");
            rc = NUNIF_HIP_EUNSUPPORTED;
            break;
        }
        h->C = C;
        std::vector<f16> mbh(C);
        for (int i = 0; i < C; ++i) mbh[i] = (f16)mb->data[i];
        if ((rc = upload(h, mbh, &h->mask_bias))) break;
        {   // patch: 1x1 conv 48 -> C on the pixel_unshuffle(4) channels (input padded to 64)
            const float *wd = pw->data;
            if ((rc = make_lin(h, C, 64, [=](int n, int k) { return k < 48 ? wd[(size_t)n * 48 + k] : 0.f; },
                               std::vector<float>(pb->data, pb->data + C), &h->patch)))
                break;
        }
        {   // down: Conv2d(C, 2C, 2, 2): weight [2C][C][2][2] -> k = (i*2 + j) * C + ci
            const float *wd = dw->data;
            if ((rc = make_lin(h, C2, 4 * C, [=](int n, int k) { return wd[((size_t)n * C + k % C) * 4 + k / C]; },
                               std::vector<float>(db->data, db->data + C2), &h->down)))
                break;
        }
        {   // up: 1x1 conv 2C -> 4C + F.pixel_shuffle(2): rows c*4 + q -> q*C + c (gemm mode 1 column order)
            const float *wd = uw->data;
            std::vector<float> bb(4 * C);
            for (int n = 0; n < 4 * C; ++n) bb[n] = ub->data[(n % C) * 4 + n / C];
            if ((rc = make_lin(h, 4 * C, C2, [=](int n, int k) { return wd[(size_t)((n % C) * 4 + n / C) * C2 + k]; }, bb, &h->up)))
                break;
        }
        if (h->video) {
            // light_video_inpaint_v1.py:109-119; lv2_mlp_ratio (1 small, 2 medium / large) is read off the checkpoint
            h->n_enc2 = 5;
            if ((rc = make_gblock(h, m, "enc1.", C, 16, 0, &h->enc1))) break;
            static const int shift[5] = {1, 0, 0, 0, 1}, temporal[5] = {0, 1, 0, 1, 0};
            for (int i = 0; i < 5 && !rc; ++i)
                rc = make_gblock(h, m, "enc2." + std::to_string(i) + ".", C2, 8, shift[i], &h->enc2[i], temporal[i] ? 2 : 0,
                                 temporal[i]);
            if (rc) break;
            if ((rc = make_gblock(h, m, "dec1.", C, 16, 0, &h->dec1))) break;
            const HostT *tw, *tb;
            if ((rc = find(m, "to_image.weight", &tw)) || (rc = find(m, "to_image.bias", &tb))) break;
            NUNIF_REQUIRE(tw->numel == (int64_t)48 * C && tb->numel == 48, "to_image: unexpected shape");
            const float *wd = tw->data;
            std::vector<float> bb(64, 0.f);
            std::copy(tb->data, tb->data + 48, bb.begin());
            if ((rc = make_lin(h, 64, C, [=](int n, int k) { return n < 48 ? wd[(size_t)n * C + k] : 0.f; }, bb, &h->to_image1)))
                break;
        } else {
            if ((rc = make_gblock(h, m, "enc1.", 96, 16, 1, &h->enc1))) break;
            for (int i = 0; i < 4 && !rc; ++i) rc = make_gblock(h, m, "enc2." + std::to_string(i) + ".", 192, 8, i & 1, &h->enc2[i]);
            if (rc) break;
            if ((rc = make_gblock(h, m, "dec1.", 96, 16, 0, &h->dec1))) break;
            if ((rc = make_conv3(h, m, "to_image.1", 96, 96, 48, &h->to_image))) break;
        }
        // get_gaussian_kernel1d(15) (gaussian_filter.py:8-19): sigma = 15 * 0.15 + 0.35, normalised, computed in fp32 like torch
        float g[15], sum = 0.f;
        const float sigma = 15 * 0.15f + 0.35f;
        for (int i = 0; i < 15; ++i) { const float xv = (float)(i - 7) / sigma; g[i] = expf(-0.5f * (xv * xv)); sum += g[i]; }
        for (int i = 0; i < 15; ++i) h->gauss.w[i] = g[i] / sum;
    } while (0);
    if (rc) { nunif_hip_light_inpaint_destroy(h); return rc; }
    *handle = h;
    return NUNIF_HIP_OK;
}

extern "C" void nunif_hip_light_inpaint_destroy(nunif_light_inpaint *h) {
    if (!h) return;
    for (void *p : h->owned) (void)hipFree(p);
    for (Buf *b : {&h->x1, &h->x2, &h->a, &h->pi, &h->vt, &h->st, &h->g, &h->po, &h->y, &h->z, &h->ti, &h->mtok, &h->mf0, &h->mf1,
                   &h->mf2, &h->hard, &h->soft})
        b->release();
    delete h;
}

extern "C" int nunif_hip_light_inpaint_infer(nunif_light_inpaint *h, const float *x, const uint8_t *mask, float *out, int32_t B,
                                             int32_t H, int32_t W, int32_t closing, int32_t inner_iter, int32_t outer_iter,
                                             void *stream) {
    return nunif_hip_light_inpaint_infer_ex(h, x, mask, out, B, H, W, closing, inner_iter, outer_iter, 0, stream);
}

extern "C" int nunif_hip_light_inpaint_infer_ex(nunif_light_inpaint *h, const float *x, const uint8_t *mask, float *out, int32_t B,
                                                int32_t H, int32_t W, int32_t closing, int32_t inner_iter, int32_t outer_iter,
                                                int32_t mirror_x, void *stream) {
    NUNIF_REQUIRE(h && x && mask && out && B > 0 && H > 0 && W > 0 && inner_iter >= 0 && outer_iter >= 0 && x != out,
                  "light_inpaint_infer: bad argument");
    NUNIF_REQUIRE(!h->video || B == 12, "light_inpaint_infer: the video net takes exactly 12 frames per call (got %d)", B);
    hipStream_t s = (hipStream_t)stream;
    const int Hp = H + 64 - H % 64, Wp = W + 64 - W % 64;            // light_inpaint_v1.py:135-137: always pads (1..64)
    const int h1 = Hp / 4, w1 = Wp / 4, h2 = h1 / 2, w2 = w1 / 2;
    const long px = (long)B * H * W, t1 = (long)B * h1 * w1, t2 = (long)B * h2 * w2;
    const long tp1 = (long)B * (h1 + 16) * (w1 + 16), tp2 = (long)B * (h2 + 8) * (w2 + 8);
    const int C = h->C, C2 = 2 * C;
    const size_t big = (size_t)std::max(tp1 * C, tp2 * C2) * sizeof(f16);
    int rc;
    if ((rc = h->x1.ensure((size_t)t1 * C * 2)) || (rc = h->x2.ensure((size_t)t2 * C2 * 2)) || (rc = h->a.ensure(big)) ||
        (rc = h->pi.ensure(big * 4)) || (rc = h->vt.ensure(big * 2)) || (rc = h->st.ensure(big * 2)) ||
        (rc = h->g.ensure(big * 2)) || (rc = h->po.ensure(big)) || (rc = h->y.ensure(big)) || (rc = h->z.ensure(big)) ||
        (rc = h->ti.ensure((size_t)t1 * 64 * 2)) || (rc = h->mtok.ensure((size_t)t1)) || (rc = h->mf0.ensure((size_t)px * 4)) ||
        (rc = h->mf1.ensure((size_t)px * 4)) || (rc = h->mf2.ensure((size_t)px * 4)) || (rc = h->hard.ensure((size_t)px * 4)) ||
        (rc = h->soft.ensure((size_t)px * 4)))
        return rc;
    float *mf0 = (float *)h->mf0.p, *mf1 = (float *)h->mf1.p, *mf2 = (float *)h->mf2.p, *hard = (float *)h->hard.p;
    float *soft = (float *)h->soft.p;
    const unsigned pb = (unsigned)((px + 255) / 256);
    {
        ProfScope ps("li_preprocess", s, 0.0, (double)px * 40.0);
        // preprocess :93-104: [mask_closing] -> dilate_inner / dilate_outer -> hard mask; soft = clamp(blur15(hard) + hard)
        li_u8_to_f32_kernel<<<pb, 256, 0, s>>>(mask, mf0, px);
        float *cur = mf0;
        if (closing) {          // mask_closing (iw3/dilation.py:145-152): closing(3x3, n_iter = 2) + the original, clamp
            li_morph_kernel<<<pb, 256, 0, s>>>(mf0, mf1, B, H, W, 0);
            li_morph_kernel<<<pb, 256, 0, s>>>(mf1, mf2, B, H, W, 0);
            li_morph_kernel<<<pb, 256, 0, s>>>(mf2, mf1, B, H, W, 1);
            li_morph_kernel<<<pb, 256, 0, s>>>(mf1, mf2, B, H, W, 1);
            li_add_clamp_kernel<<<pb, 256, 0, s>>>(mf2, mf0, mf1, px);
            cur = mf1;
        }
        if (inner_iter > 0 || outer_iter > 0) li_dilate_kernel<<<pb, 256, 0, s>>>(cur, hard, (long)B * H, W, inner_iter, outer_iter);
        else NUNIF_HIP_CHECK(hipMemcpyAsync(hard, cur, (size_t)px * 4, hipMemcpyDeviceToDevice, s));
        li_blur_kernel<<<pb, 256, 0, s>>>(hard, hard, mf2, B, H, W, 0, h->gauss);
        li_blur_kernel<<<pb, 256, 0, s>>>(mf2, hard, soft, B, H, W, 1, h->gauss);
    }
    f16 *x1 = (f16 *)h->x1.p, *x2 = (f16 *)h->x2.p, *a = (f16 *)h->a.p, *ti = (f16 *)h->ti.p;
    uint8_t *mtok = (uint8_t *)h->mtok.p;
    {
        ProfScope ps("li_patch_in_kernel", s, 0.0, (double)px * 20.0);
        li_patch_in_kernel<<<(unsigned)((t1 + 255) / 256), 256, 0, s>>>(x, hard, soft, a, mtok, B, H, W, h1, w1, mirror_x);
    }
    // NOTE: the replicate padding of the soft mask is the clamp of the pixel coordinate inside li_patch_in_kernel
    if ((rc = lin(h->patch, a, t1, C, 2, h->video ? 0.1f : 0.2f, nullptr, x1, s, "li_patch"))) return rc;
    {
        const long n = t1 * (C / 8);
        li_mask_bias_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(x1, mtok, h->mask_bias, t1, C);
    }
    if ((rc = run_gblock_any(h, h->enc1, x1, B, h1, w1, s))) return rc;
    {   // down: 2x2 stride-2 conv as a gather GEMM
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.a = x1; g.B = B; g.Hi = h1; g.Wi = w1; g.Cin = C; g.Ho = h2; g.Wo = w2; g.stride = 2; g.kw = 2;
        g.K = 4 * C; g.w = h->down.w; g.bias = h->down.bias; g.N = h->down.N; g.mode = 0; g.out = x2; g.ldo = C2; g.n_real = C2;
        g.ps = 1;
        if ((rc = launch_gemm(g, s, "li_down"))) return rc;
    }
    for (int i = 0; i < h->n_enc2; ++i) {
        if ((rc = run_gblock_any(h, h->enc2[i], x2, B, h2, w2, s))) return rc;
    }
    {   // x = x1 + pixel_shuffle(up(x2), 2), written over x1
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.a = x2; g.B = B; g.Hi = h2; g.Wi = w2; g.Cin = C2; g.Ho = h2; g.Wo = w2; g.stride = 1; g.kw = 1;
        g.K = C2; g.w = h->up.w; g.bias = h->up.bias; g.N = h->up.N; g.mode = 1; g.res = x1; g.out = x1; g.ldo = C; g.n_real = 4 * C;
        g.ps = 1;
        // K = 192 -> 4 x 96 / 4 x 192 with the skip map as residual is the swin PatchUp's shape: its resident-weight kernel with
        // the skip tiles four trips ahead (swin_patchup.hip) takes it — every launch of that shape
        PatchUpArgs pu = {x2, h->up.w, h->up.bias, x1, x1, B, h2, w2, C, 0};
        if (C2 == 192 && h->up.N == 4 * C && patchup_supported(pu)) {
            if ((rc = launch_patchup(pu, s))) return rc;
        } else if ((rc = launch_gemm(g, s, "li_up"))) return rc;
    }
    if ((rc = run_gblock_any(h, h->dec1, x1, B, h1, w1, s))) return rc;
    if (h->video) {
        if ((rc = lin(h->to_image1, x1, t1, 64, 0, 0.f, nullptr, ti, s, "li_to_image"))) return rc;
    } else {
        ConvArgs cv;
        memset(&cv, 0, sizeof(cv));
        cv.a = x1; cv.B = B; cv.Hi = h1; cv.Wi = w1; cv.Cin = 96; cv.Ho = h1; cv.Wo = w1; cv.stride = 1; cv.kh = 3; cv.kw = 3;
        cv.wstream = h->to_image.stream; cv.bias = h->to_image.bias; cv.N = h->to_image.N; cv.n_real = 48; cv.act = 0; cv.out = ti;
        cv.rpad = 1;
        if ((rc = launch_conv(cv, s))) return rc;
    }
    {
        ProfScope ps("li_compose_kernel", s, 0.0, (double)px * 36.0);
        li_compose_kernel<<<pb, 256, 0, s>>>(x, hard, soft, ti, out, B, H, W, h1, w1, h->video ? 64 : 48, mirror_x);
    }
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

// ---- stand-alone mask morphology (iw3/dilation.py dilate :41-46, erode :49-54, closing :57-64, mask_closing :145-153,
//      dilate_outer :67-81, dilate_inner :84-98) on fp32 0/1 masks [B,H,W]; the same kernels the inpaint pre-processing uses ----
// op: 0 dilate x n_a, 1 erode x n_a, 2 closing(n_iter = n_a), 3 mask_closing(n_iter = n_a), 4 horizontal OR-dilation: n_a steps
// towards +x sources (dilate_inner) and n_b steps towards -x sources (dilate_outer).  work: B*H*W floats.  in != out.
extern "C" int nunif_hip_mask_morphology(const float *in, float *out, float *work, int32_t B, int32_t H, int32_t W, int32_t op,
                                         int32_t n_a, int32_t n_b, void *stream) {
    using namespace nunif;
    NUNIF_REQUIRE(in && out && work && B > 0 && H > 0 && W > 0 && n_a >= 0 && n_b >= 0 && op >= 0 && op <= 4 && in != out,
                  "mask_morphology: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)B * H * W;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    ProfScope ps("mask_morphology", s, 0.0, (double)n * 8.0 * (op == 4 ? 1 : (op >= 2 ? 2 * n_a : n_a) + 1));
    if (op == 4) {
        li_dilate_kernel<<<blocks, 256, 0, s>>>(in, out, (long)B * H, W, n_a, n_b);
        NUNIF_LAUNCH_CHECK();
        return NUNIF_HIP_OK;
    }
    // passes: list of (is_min); ping-pong between out and work so that the last pass lands in out
    int passes[64];
    int np = 0;
    NUNIF_REQUIRE(n_a <= 16, "mask_morphology: at most 16 iterations");
    if (op == 0 || op >= 2) for (int i = 0; i < n_a; ++i) passes[np++] = 0;
    if (op == 1 || op >= 2) for (int i = 0; i < n_a; ++i) passes[np++] = 1;
    if (np == 0) {
        NUNIF_HIP_CHECK(hipMemcpyAsync(out, in, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        return NUNIF_HIP_OK;
    }
    const float *src = in;
    // op 3 adds the original mask at the end (elementwise, in place on out), so the last morph pass must land in `work`
    const bool last_in_out = op != 3;
    for (int i = 0; i < np; ++i) {
        const bool to_out = ((np - 1 - i) % 2 == 0) == last_in_out;
        float *dst = to_out ? out : work;
        li_morph_kernel<<<blocks, 256, 0, s>>>(src, dst, B, H, W, passes[i]);
        src = dst;
    }
    if (op == 3) li_add_clamp_kernel<<<blocks, 256, 0, s>>>(src, in, out, n);      // (closing + original).clamp(0, 1)
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}
