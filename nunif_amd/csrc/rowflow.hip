// iw3 "sbs.row_flow_v3" (the default --method) on gfx950: depth -> horizontal flow (delta) -> backward warp.
//
// Reference: iw3/models/row_flow_v3.py — RowFlowV3._forward :56-66 (replicate pad to multiples of 96 x 12,
// pixel_unshuffle (1,8), 1x1 conv 24->64, WABlock(4x4), WABlock(3x3), pixel_shuffle, crop, replicate-pad 3x3 conv 8->1),
// WABlock :14-30; nunif/modules/attention.py — WindowMHA2d :118-161, MHA :94-115, sliced_sdp :61-77 (2 heads of 32,
// float score bias), WindowScoreBias :375-419; iw3/backward_warp.py — backward_warp :67-83, make_grid :86-93,
// apply_divergence_nn_delta :191-236 (right eye = flipped inputs / flipped output).
//
// The net runs at DEPTH resolution (e.g. 392 x 686 -> 396 x 96 tokens of 64 channels, ~9 GFLOP per eye), so it is a
// handful of small launches; maps are NHWC fp16 like the other nets:
//   rf_input_kernel   pad + unshuffle + 1x1 conv 24->64                        (VALU, K = 24)
//   rf_wmha_kernel<WS> per window: qkv GEMM, 2-head attention with the learned score bias, head_proj, residual —
//                      everything after the x gather in registers (same operand tricks as swin_qkv_attn_r.hip)
//   gemm_kernel<2,4>  conv_mlp[0] 1x1 + GELU(erf)          conv_kernel<4,4>  replicate-pad 3x3 + LeakyReLU + residual
//   rf_output_kernel  pixel_shuffle + crop + replicate-pad 3x3 conv 8->1 -> delta fp32
//   delta_warp_kernel grid = linspace + delta*scale, bilinear resize to the image size, grid_sample(border) + clamp
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "swin_kernels.h"

namespace nunif {

#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

// ---------------------------------------------------------------------------------------------------------------------
struct RfInputArgs {
    const float *x;          // [B,3,h,w] planes: depth, divergence feature, convergence feature
    const float *w;          // [24][64] (k = c*8 + sw, transposed for broadcast reads) then bias[64]
    f16 *out;                // [B,Hp,Wq,64]
    int B, h, w_, Hp, Wq, flip;
};

__global__ void __launch_bounds__(256) rf_input_kernel(RfInputArgs a) {
    __shared__ float sw[25 * 64];
    for (int i = threadIdx.x; i < 25 * 64; i += 256) sw[i] = a.w[i];
    __syncthreads();
    const long total = (long)a.B * a.Hp * a.Wq;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int xq = (int)(id % a.Wq);
    const long t = id / a.Wq;
    const int y = (int)(t % a.Hp), b = (int)(t / a.Hp);
    const int yy = min(y, a.h - 1);                                   // replicate pad (bottom / right only)
    float in[24];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            int xx = min(xq * 8 + s, a.w_ - 1);
            if (a.flip) xx = a.w_ - 1 - xx;                            // the right eye runs on the mirrored depth
            in[c * 8 + s] = a.x[(((long)b * 3 + c) * a.h + yy) * a.w_ + xx];
        }
    f16 *o = a.out + id * 64;
    for (int c0 = 0; c0 < 64; c0 += 8) {
        f16x8 ov;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float acc = sw[24 * 64 + c0 + j];
#pragma unroll
            for (int k = 0; k < 24; ++k) acc = fmaf(in[k], sw[k * 64 + c0 + j], acc);
            ov[j] = (f16)acc;
        }
        *reinterpret_cast<f16x8 *>(o + c0) = ov;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
struct WmhaArgs {
    f16 *x;                  // [B,H,W,C], updated in place: x += head_proj(attention(x))
    const f16 *wfrag;        // 3*NT*KS qkv fragments (part, nt, ks) + NT*KS head_proj fragments (nt, ks; chained k order)
    const float *bqkv;       // [3C] (q part pre-scaled by hd^-0.5 * log2e)
    const float *bproj;      // [C]
    const float *btab;       // [16][16] log2e * score bias [query][key]; -1e30 for keys beyond the window
    int B, H, W, n_windows;
    int sy, sx;              // WindowMHA2d shift: the map is ZERO-padded by (sy, sx) on both sides, windows tile the
                             // padded map, padded positions act as (all-zero input) tokens and are cropped away again
};

__device__ __forceinline__ f16x8 cat8f(f16x4 lo, f16x4 hi) {
    return (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// One window (WS x WS <= 16 tokens = one MFMA tile) per wave; C channels = C/32 heads of 32; weights resident in LDS.
template <int WS, int C>
__global__ void __launch_bounds__(256) wmha_kernel(WmhaArgs a) {
    constexpr int N = WS * WS, KS = C / 32, NT = C / 16, HEADS = C / 32;
    constexpr int NF = 4 * NT * KS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
    f16x8 *wl = reinterpret_cast<f16x8 *>(smem_w);                                  // [NF][64]
    float *tb = reinterpret_cast<float *>(wl + NF * 64);                            // [256]
    float *bq = tb + 256;                                                           // [3C]
    float *bp = bq + 3 * C;                                                         // [C]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, grp = lane >> 4;
    for (int i = tid; i < NF * 64; i += 256) wl[i] = reinterpret_cast<const f16x8 *>(a.wfrag)[i];
    tb[tid] = a.btab[tid];
    for (int i = tid; i < 3 * C; i += 256) bq[i] = a.bqkv[i];
    for (int i = tid; i < C; i += 256) bp[i] = a.bproj[i];
    __syncthreads();
    const f16x4 zero4 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    const f16x8 zero8 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    const int nwx = (a.W + 2 * a.sx) / WS, nwy = (a.H + 2 * a.sy) / WS;
    const int tt = min(r16, N - 1);
    const int iy = tt / WS, ix = tt - iy * WS;
    const f16x8 *wq = wl + lane;

    for (int wi = blockIdx.x * 4 + wave; wi < a.n_windows; wi += gridDim.x * 4) {
        const int wx = wi % nwx, t2 = wi / nwx;
        const int wy = t2 % nwy, b = t2 / nwy;
        const int y = wy * WS + iy - a.sy, x = wx * WS + ix - a.sx;
        const bool inside = y >= 0 && y < a.H && x >= 0 && x < a.W;
        const long pix = ((long)b * a.H + min(max(y, 0), a.H - 1)) * a.W + min(max(x, 0), a.W - 1);
        f16x8 xf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            xf[ks] = inside ? *reinterpret_cast<const f16x8 *>(a.x + pix * C + ks * 32 + 8 * grp) : zero8;
        // q, k: [channel 4g+r of tile nt][token l&15];  v (operands swapped): [token 4g+r][channel l&15 of tile nt]
        f16x4 q4[NT], k4[NT], v4[NT];
#pragma unroll
        for (int part = 0; part < 3; ++part)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int ch0 = part * C + nt * 16;
                f32x4 acc;
                if (part == 2) { const float bv = bq[ch0 + r16]; acc = (f32x4){bv, bv, bv, bv}; }
                else acc = *reinterpret_cast<const f32x4 *>(bq + ch0 + 4 * grp);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const f16x8 w = wq[((part * NT + nt) * KS + ks) * 64];
                    acc = part == 2 ? MFMA_16x16x32(xf[ks], w, acc) : MFMA_16x16x32(w, xf[ks], acc);
                }
                const f16x4 v = {(f16)acc[0], (f16)acc[1], (f16)acc[2], (f16)acc[3]};
                if (part == 0) q4[nt] = v; else if (part == 1) k4[nt] = v; else v4[nt] = v;
            }
        // heads of 32 channels: S^T[key][query] = K Q^T + bias; softmax over keys; O^T = V^T P^T
        f16x4 o4[NT];
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(tb + r16 * 16 + 4 * grp);          // [query l&15][keys 4g..]
#pragma unroll
        for (int hh = 0; hh < HEADS; ++hh) {
            f32x4 s = MFMA_16x16x32(cat8f(k4[2 * hh], k4[2 * hh + 1]), cat8f(q4[2 * hh], q4[2 * hh + 1]), bias);
            float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float p0 = __builtin_amdgcn_exp2f(s[0] - mx), p1 = __builtin_amdgcn_exp2f(s[1] - mx);
            const float p2 = __builtin_amdgcn_exp2f(s[2] - mx), p3 = __builtin_amdgcn_exp2f(s[3] - mx);
            float sum = (p0 + p1) + (p2 + p3);
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
            const f16x8 pf = cat8f((f16x4){(f16)p0, (f16)p1, (f16)p2, (f16)p3}, zero4);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                o = MFMA_16x16x32(cat8f(v4[2 * hh + dt], zero4), pf, o);
                o4[2 * hh + dt] = (f16x4){(f16)(o[0] * inv), (f16)(o[1] * inv), (f16)(o[2] * inv), (f16)(o[3] * inv)};
            }
        }
        // head_proj (weights packed in the chained k order: two accumulator tiles = one 32-wide k-step) + residual
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 acc = *reinterpret_cast<const f32x4 *>(bp + nt * 16 + 4 * grp);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc = MFMA_16x16x32(wq[(3 * NT * KS + nt * KS + ks) * 64], cat8f(o4[2 * ks], o4[2 * ks + 1]), acc);
            if (inside && r16 < N) {
                f16 *px = a.x + pix * C + nt * 16 + 4 * grp;
                const f16x4 xr = *reinterpret_cast<const f16x4 *>(px);
                const f16x4 ov = {(f16)(acc[0] + (float)xr[0]), (f16)(acc[1] + (float)xr[1]), (f16)(acc[2] + (float)xr[2]),
                                  (f16)(acc[3] + (float)xr[3])};
                *reinterpret_cast<f16x4 *>(px) = ov;
            }
        }
    }
}

template <int WS, int C>
static int launch_wmha_t(const WmhaArgs &a, hipStream_t s) {
    constexpr size_t smem = (size_t)4 * (C / 16) * (C / 32) * 1024 + 256 * 4 + 4 * C * 4;
    static bool configured = false;
    if (!configured) {
        NUNIF_HIP_CHECK(hipFuncSetAttribute((const void *)wmha_kernel<WS, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        configured = true;
    }
    const int grid = std::min((a.n_windows + 3) / 4, C == 64 ? 2048 : 256);
    wmha_kernel<WS, C><<<grid, 256, smem, s>>>(a);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

int launch_wmha(WmhaArgs a, int window, int C, hipStream_t s) {
    NUNIF_REQUIRE((a.H + 2 * a.sy) % window == 0 && (a.W + 2 * a.sx) % window == 0, "window attention: %dx%d (+%d,%d) is not "
                  "a multiple of the %dx%d window", a.H, a.W, a.sy, a.sx, window, window);
    a.n_windows = a.B * ((a.H + 2 * a.sy) / window) * ((a.W + 2 * a.sx) / window);
    const double tok = (double)a.B * a.H * a.W;
    if (window == 4 && C == 64) { ProfScope ps("wmha_kernel<4,64>", s, tok * (8.0 * C * C + 4.0 * 16 * C), tok * C * 4.0); return launch_wmha_t<4, 64>(a, s); }
    if (window == 3 && C == 64) { ProfScope ps("wmha_kernel<3,64>", s, tok * (8.0 * C * C + 4.0 * 9 * C), tok * C * 4.0); return launch_wmha_t<3, 64>(a, s); }
    if (window == 4 && C == 128) { ProfScope ps("wmha_kernel<4,128>", s, tok * (8.0 * C * C + 4.0 * 16 * C), tok * C * 4.0); return launch_wmha_t<4, 128>(a, s); }
    set_error("window attention: window %d / %d channels unsupported", window, C);
    return NUNIF_HIP_EUNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------------------------------
struct RfOutputArgs {
    const f16 *f;            // [B,Hp,Wq,64]
    const float *w;          // [8][3][3] then bias
    float *delta;            // [B,1,h,w]
    int B, h, w_, Hp, Wq;
};

__global__ void __launch_bounds__(256) rf_output_kernel(RfOutputArgs a) {
    __shared__ float sw[73];
    if (threadIdx.x < 73) sw[threadIdx.x] = a.w[threadIdx.x];
    __syncthreads();
    const long total = (long)a.B * a.h * a.w_;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int x = (int)(id % a.w_);
    const long t = id / a.w_;
    const int y = (int)(t % a.h), b = (int)(t / a.h);
    float acc = sw[72];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            // pixel_shuffle (1,8) + crop + ReplicationPad2d(1): shuffled[c][yy][xx] = f[yy][xx/8][c*8 + xx%8]
            const int yy = min(max(y + dy - 1, 0), a.h - 1), xx = min(max(x + dx - 1, 0), a.w_ - 1);
            const f16 *p = a.f + (((long)b * a.Hp + yy) * a.Wq + (xx >> 3)) * 64 + (xx & 7);
#pragma unroll
            for (int c = 0; c < 8; ++c) acc = fmaf((float)p[c * 8], sw[(c * 3 + dy) * 3 + dx], acc);
        }
    a.delta[id] = acc;
}

// ---------------------------------------------------------------------------------------------------------------------
struct DeltaWarpArgs {
    const float *c;          // [B,C,H,W]
    const float *delta;      // [B,L,h,w] (horizontal flows at depth resolution, in the mirrored frame when flip)
    const float *weight;     // NULL (L = 1, weight 1) or [B,L,H,W] layer weights at IMAGE resolution (mirrored frame)
    float *out;              // [B,C,H,W]
    int B, C, H, W, h, w, flip, L;
    float delta_scale;
};

__device__ __forceinline__ float lin_pm1(int i, int n) {           // torch.linspace(-1, 1, n)[i]
    if (n <= 1) return -1.0f;
    const float step = 2.0f / (float)(n - 1);
    return i < n / 2 ? -1.0f + step * (float)i : 1.0f - step * (float)(n - 1 - i);
}

__global__ void __launch_bounds__(256) delta_warp_kernel(DeltaWarpArgs a) {
    const long total = (long)a.B * a.H * a.W;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int Xo = (int)(id % a.W);
    const long t = id / a.W;
    const int Y = (int)(t % a.H), b = (int)(t / a.H);
    const int X = a.flip ? a.W - 1 - Xo : Xo;                         // position in the (possibly mirrored) frame
    float accum[4] = {0.f, 0.f, 0.f, 0.f};
    for (int layer = 0; layer < a.L; ++layer) {
    const float *dmap = a.delta + ((long)b * a.L + layer) * a.h * a.w;
    auto gx_at = [&](int yy, int xx) -> float { return lin_pm1(xx, a.w) + dmap[(long)yy * a.w + xx] * a.delta_scale; };
    float gx, gy;
    if (a.h == a.H && a.w == a.W) {
        gx = gx_at(Y, X);
        gy = lin_pm1(Y, a.h);
    } else {
        // F.interpolate(grid, size=(H,W), bilinear, align_corners=True)  (backward_warp.py:69-71)
        const float ry = a.H > 1 ? (float)(a.h - 1) / (float)(a.H - 1) : 0.f;
        const float rx = a.W > 1 ? (float)(a.w - 1) / (float)(a.W - 1) : 0.f;
        const float sy = ry * (float)Y, sx = rx * (float)X;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < a.h - 1 ? 1 : 0), x1 = x0 + (x0 < a.w - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        gx = hy * (hx * gx_at(y0, x0) + lx * gx_at(y0, x1)) + ly * (hx * gx_at(y1, x0) + lx * gx_at(y1, x1));
        const float g0 = lin_pm1(y0, a.h), g1 = lin_pm1(y1, a.h);
        gy = hy * (hx * g0 + lx * g0) + ly * (hx * g1 + lx * g1);
    }
    // grid_sampler_2d, align_corners=True, padding_mode=border
    float ix = ((gx + 1.f) / 2.f) * (float)(a.W - 1);
    float iy = ((gy + 1.f) / 2.f) * (float)(a.H - 1);
    ix = fminf(fmaxf(ix, 0.f), (float)(a.W - 1));
    iy = fminf(fmaxf(iy, 0.f), (float)(a.H - 1));
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float tx = ix - fx, ty = iy - fy;
    const float w_nw = (1.f - tx) * (1.f - ty), w_ne = tx * (1.f - ty), w_sw = (1.f - tx) * ty, w_se = tx * ty;
    const bool xin = x1 <= a.W - 1, yin = y1 <= a.H - 1;
    // the sampled image is the mirrored one when flip: column x of it is column W-1-x of c
    const int cx0 = a.flip ? a.W - 1 - x0 : x0, cx1 = a.flip ? a.W - 1 - x1 : x1;
    const float lw = a.weight ? a.weight[(((long)b * a.L + layer) * a.H + Y) * a.W + X] : 1.0f;
    for (int ch = 0; ch < a.C && ch < 4; ++ch) {
        const float *p = a.c + ((long)b * a.C + ch) * a.H * a.W;
        float v = p[(long)y0 * a.W + cx0] * w_nw;
        if (xin) v += p[(long)y0 * a.W + cx1] * w_ne;
        if (yin) v += p[(long)y1 * a.W + cx0] * w_sw;
        if (xin && yin) v += p[(long)y1 * a.W + cx1] * w_se;
        accum[ch] += fminf(fmaxf(v, 0.f), 1.f) * lw;                  // backward_warp clamps, then the layer weight
    }
    }
    for (int ch = 0; ch < a.C && ch < 4; ++ch)
        a.out[(((long)b * a.C + ch) * a.H + Y) * a.W + Xo] = fminf(fmaxf(accum[ch], 0.f), 1.f);
}

// ---------------------------------------------------------------------------------------------------------------------
// MLBW (iw3/models/mlbw.py): lv1_in = centred replicate pad to multiples of 32 x 4, ReplicationPad (4,4,0,0) + 1x9 conv
// 3 -> C/8 + LeakyReLU(0.2), written directly in the pixel_unshuffle (1,8) layout (channel c*8 + x%8 of token x/8).
struct MlbwInArgs {
    const float *x;          // [B,3,h,w]
    const float *w;          // [27][Cs] (k = ci*9 + tap) then bias[Cs]
    f16 *out, *x1;           // [B,Hp,Wq,C] twice: the working map and the preserved skip (x + x1 before lv1_out)
    int B, h, w_, Hp, Wp, ph1, pw1, Cs, flip;
};

__global__ void __launch_bounds__(256) mlbw_in_kernel(MlbwInArgs a) {
    __shared__ float sw[28 * 16];
    for (int i = threadIdx.x; i < 28 * a.Cs; i += 256) sw[i] = a.w[i];
    __syncthreads();
    const long total = (long)a.B * a.Hp * a.Wp;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int xp = (int)(id % a.Wp);
    const long t = id / a.Wp;
    const int yp = (int)(t % a.Hp), b = (int)(t / a.Hp);
    const int y = min(max(yp - a.ph1, 0), a.h - 1);
    float in[27];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int xs = min(max(xp + k - 4, 0), a.Wp - 1);              // lv1_in's own replicate pad, on the padded map
        int xx = min(max(xs - a.pw1, 0), a.w_ - 1);
        if (a.flip) xx = a.w_ - 1 - xx;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) in[ci * 9 + k] = a.x[(((long)b * 3 + ci) * a.h + y) * a.w_ + xx];
    }
    const int C = a.Cs * 8;
    const long base = (((long)b * a.Hp + yp) * (a.Wp >> 3) + (xp >> 3)) * C + (xp & 7);
    for (int co = 0; co < a.Cs; ++co) {
        float acc = sw[27 * a.Cs + co];
#pragma unroll
        for (int k = 0; k < 27; ++k) acc = fmaf(in[k], sw[k * a.Cs + co], acc);
        acc = acc >= 0.f ? acc : acc * 0.2f;
        a.out[base + co * 8] = (f16)acc;
        a.x1[base + co * 8] = (f16)acc;
    }
}

// lv1_out: pixel_shuffle + (x + x1) + ReplicationPad (4,4,0,0) + 1x9 conv C/8 -> 2L, crop, softmax over the L logits
struct MlbwOutArgs {
    const f16 *f, *x1;       // [B,Hp,Wq,C]
    const float *w;          // [Cs*9][2L] (k = ci*9 + tap) then bias[2L]
    float *delta, *weight;   // [B,L,h,w] each
    float *mask;             // hole-mask variant (2L + 1 outputs): [B,1,h,w] logits in IMAGE coordinates, else NULL
    int B, h, w_, Hp, Wp, ph1, pw1, Cs, L, no, flip;
};

__global__ void __launch_bounds__(256) mlbw_out_kernel(MlbwOutArgs a) {
    __shared__ float sw[(16 * 9 + 1) * 8];
    const int no = a.no;         // 2L, or 2L + 1 with the hole-mask logit (mlbw.py:70-75, 104-106)
    for (int i = threadIdx.x; i < (a.Cs * 9 + 1) * no; i += 256) sw[i] = a.w[i];
    __syncthreads();
    const long total = (long)a.B * a.h * a.w_;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int x = (int)(id % a.w_);
    const long t = id / a.w_;
    const int y = (int)(t % a.h), b = (int)(t / a.h);
    const int yp = y + a.ph1, xp = x + a.pw1, C = a.Cs * 8;
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = o < no ? sw[a.Cs * 9 * no + o] : 0.f;
    for (int k = 0; k < 9; ++k) {
        const int xs = min(max(xp + k - 4, 0), a.Wp - 1);
        const long base = (((long)b * a.Hp + yp) * (a.Wp >> 3) + (xs >> 3)) * C + (xs & 7);
        for (int ci = 0; ci < a.Cs; ++ci) {
            const float v = (float)a.f[base + ci * 8] + (float)a.x1[base + ci * 8];
            const float *wr = sw + (ci * 9 + k) * no;
#pragma unroll
            for (int o = 0; o < 8; ++o) if (o < no) acc[o] = fmaf(v, wr[o], acc[o]);
        }
    }
    float mx = -3.0e38f;
    for (int i = 0; i < a.L; ++i) mx = fmaxf(mx, acc[a.L + i]);
    float e[4], sum = 0.f;
    for (int i = 0; i < a.L; ++i) { e[i] = expf(acc[a.L + i] - mx); sum += e[i]; }
    const long hw = (long)a.h * a.w_, o0 = (long)b * a.L * hw + (long)y * a.w_ + x;
    for (int i = 0; i < a.L; ++i) {
        a.delta[o0 + i * hw] = acc[i];
        a.weight[o0 + i * hw] = e[i] / sum;
    }
    // the reference flips the logits back for the right eye (backward_warp.py:325-327); delta / weight stay in model
    // coordinates because the warp kernel folds the flip
    if (a.mask) a.mask[((long)b * a.h + y) * a.w_ + (a.flip ? a.w_ - 1 - x : x)] = acc[2 * a.L];
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

struct WaBlock {
    f16 *wfrag = nullptr; float *bqkv = nullptr, *bproj = nullptr, *btab = nullptr;
    f16 *w1 = nullptr; float *b1 = nullptr;          // conv_mlp[0] 1x1, gemm_kernel packing [nt][ks]
    f16 *w3 = nullptr; float *b3 = nullptr;          // conv_mlp[3] 3x3, conv_kernel stream [ks][nt]
    int window = 4;
};

}  // namespace

struct nunif_row_flow {
    std::vector<void *> owned;
    float *w_in = nullptr, *w_out = nullptr;
    WaBlock blk[2];
    Buf f, t1, t2;
};

namespace {

template <typename H, typename T>
int upload(H *h, const std::vector<T> &host, T **dev) {
    void *p = nullptr;
    if (hipMalloc(&p, host.size() * sizeof(T)) != hipSuccess) { set_error("hipMalloc failed"); return NUNIF_HIP_ENOMEM; }
    h->owned.push_back(p);
    NUNIF_HIP_CHECK(hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *dev = reinterpret_cast<T *>(p);
    return NUNIF_HIP_OK;
}

// MFMA A fragment (nt, ks): lane l holds W[nt*16 + (l&15)][k(ks, l>>4, j)], j = 0..7
template <typename F>
void put_frag(std::vector<f16> &dst, size_t frag, int nt, int ks, bool chained, F wt) {
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
            const int g = l >> 4, n = nt * 16 + (l & 15);
            const int k = chained ? ks * 32 + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4)) : ks * 32 + g * 8 + j;
            dst[(frag * 64 + l) * 8 + j] = (f16)wt(n, k);
        }
}

double gelu_erf_d(double v) { return 0.5 * v * (1.0 + erf(v * 0.70710678118654752440)); }

template <typename H>
int make_block(H *h, const TMap &m, const std::string &p, int window, int C, WaBlock *bk) {
    const int NT = C / 16, KS = C / 32;
    const HostT *wqkv, *bqkv, *wp, *bp, *w1, *b1, *w3, *b3, *tw0, *tb0, *tw2, *tb2;
    int rc;
    if ((rc = find(m, p + "mha.mha.qkv_proj.weight", &wqkv)) || (rc = find(m, p + "mha.mha.qkv_proj.bias", &bqkv)) ||
        (rc = find(m, p + "mha.mha.head_proj.weight", &wp)) || (rc = find(m, p + "mha.mha.head_proj.bias", &bp)) ||
        (rc = find(m, p + "conv_mlp.0.weight", &w1)) || (rc = find(m, p + "conv_mlp.0.bias", &b1)) ||
        (rc = find(m, p + "conv_mlp.3.weight", &w3)) || (rc = find(m, p + "conv_mlp.3.bias", &b3)) ||
        (rc = find(m, p + "bias.to_bias.0.weight", &tw0)) || (rc = find(m, p + "bias.to_bias.0.bias", &tb0)) ||
        (rc = find(m, p + "bias.to_bias.2.weight", &tw2)) || (rc = find(m, p + "bias.to_bias.2.bias", &tb2)))
        return rc;
    NUNIF_REQUIRE(wqkv->numel == (int64_t)3 * C * C && wp->numel == (int64_t)C * C && w1->numel == (int64_t)C * C &&
                  w3->numel == (int64_t)C * C * 9, "%s: expected %d channels (heads of 32)", p.c_str(), C);
    bk->window = window;
    const float qs = (1.0f / sqrtf(32.0f)) * 1.4426950408889634f;       // head_dim^-0.5 * log2(e), folded into q
    {
        std::vector<f16> frags((size_t)4 * NT * KS * 512);
        const float *wd = wqkv->data;
        for (int part = 0; part < 3; ++part)
            for (int nt = 0; nt < NT; ++nt)
                for (int ks = 0; ks < KS; ++ks)
                    put_frag(frags, (size_t)(part * NT + nt) * KS + ks, nt, ks, false, [=](int n, int k) {
                        return wd[(size_t)(part * C + n) * C + k] * (part == 0 ? qs : 1.0f); });
        const float *pd = wp->data;
        for (int nt = 0; nt < NT; ++nt)
            for (int ks = 0; ks < KS; ++ks)
                put_frag(frags, (size_t)3 * NT * KS + nt * KS + ks, nt, ks, true, [=](int n, int k) { return pd[(size_t)n * C + k]; });
        if ((rc = upload(h, frags, &bk->wfrag))) return rc;
        std::vector<float> bq(3 * C), bpv(bp->data, bp->data + C);
        for (int n = 0; n < 3 * C; ++n) bq[n] = bqkv->data[n] * (n < C ? qs : 1.0f);
        if ((rc = upload(h, bq, &bk->bqkv)) || (rc = upload(h, bpv, &bk->bproj))) return rc;
    }
    {   // WindowScoreBias (attention.py:375-419): to_bias MLP on the normalised relative offsets, evaluated once here
        const int hidden = (int)tb0->numel, N = window * window;
        NUNIF_REQUIRE(tw0->numel == hidden * 2 && tw2->numel == hidden && tb2->numel == 1, "%s: score-bias MLP shape", p.c_str());
        const float dmax = (float)(window - 1);
        std::vector<float> tab(256, -1.0e30f);
        for (int q = 0; q < N; ++q)
            for (int k = 0; k < N; ++k) {
                const float dy = (float)(q / window - k / window) / dmax, dx = (float)(q % window - k % window) / dmax;
                double o = tb2->data[0];
                for (int j = 0; j < hidden; ++j)
                    o += (double)tw2->data[j] * gelu_erf_d((double)tw0->data[j * 2] * dy + (double)tw0->data[j * 2 + 1] * dx +
                                                            (double)tb0->data[j]);
                tab[q * 16 + k] = (float)o * 1.4426950408889634f;
            }
        for (int q = N; q < 16; ++q) for (int k = 0; k < N; ++k) tab[q * 16 + k] = 0.f;     // padded queries: any finite row
        if ((rc = upload(h, tab, &bk->btab))) return rc;
    }
    {   // conv_mlp[0]: 1x1 as a Linear, gemm_kernel packing [nt][ks] (+16 KiB pad for the ring prefetch)
        std::vector<f16> packed((size_t)C * C + 8192, (f16)0.f);
        const float *wd = w1->data;
        for (int nt = 0; nt < NT; ++nt)
            for (int ks = 0; ks < KS; ++ks) put_frag(packed, (size_t)nt * KS + ks, nt, ks, false, [=](int n, int k) { return wd[(size_t)n * C + k]; });
        std::vector<float> bb(b1->data, b1->data + C);
        if ((rc = upload(h, packed, &bk->w1)) || (rc = upload(h, bb, &bk->b1))) return rc;
    }
    {   // conv_mlp[3]: 3x3, conv_kernel stream [ks][nt], k = tap*C + ci
        const int KS3 = 9 * C / 32;
        std::vector<f16> stream((size_t)KS3 * NT * 512 + 8192, (f16)0.f);
        const float *wd = w3->data;
        for (int ks = 0; ks < KS3; ++ks)
            for (int nt = 0; nt < NT; ++nt)
                put_frag(stream, (size_t)ks * NT + nt, nt, ks, false, [=](int n, int k) {
                    const int tap = k / C, ci = k % C;
                    return wd[((size_t)n * C + ci) * 9 + tap]; });
        std::vector<float> bb(b3->data, b3->data + C);
        if ((rc = upload(h, stream, &bk->w3)) || (rc = upload(h, bb, &bk->b3))) return rc;
    }
    return NUNIF_HIP_OK;
}

}  // namespace

extern "C" int nunif_hip_row_flow_create(const nunif_tensor_desc *tensors, int32_t n_tensors, nunif_row_flow **handle) {
    NUNIF_REQUIRE(tensors && handle && n_tensors > 0, "row_flow_create: NULL argument");
    TMap m;
    for (int i = 0; i < n_tensors; ++i) {
        HostT t;
        t.data = tensors[i].data;
        t.numel = 1;
        for (int d = 0; d < tensors[i].ndim; ++d) { t.shape.push_back(tensors[i].shape[d]); t.numel *= tensors[i].shape[d]; }
        m[tensors[i].name] = t;
    }
    nunif_row_flow *h = new nunif_row_flow();
    int rc = NUNIF_HIP_OK;
    do {
        const HostT *w0, *b0, *wl, *bl;
        if ((rc = find(m, "blocks.0.weight", &w0)) || (rc = find(m, "blocks.0.bias", &b0)) ||
            (rc = find(m, "last_layer.1.weight", &wl)) || (rc = find(m, "last_layer.1.bias", &bl)))
            break;
        if (w0->numel != 64 * 24 || wl->numel != 72) { set_error("row_flow_v3: unexpected stem / head shape"); rc = NUNIF_HIP_EINVAL; break; }
        std::vector<float> win(25 * 64), wout(73);
        for (int k = 0; k < 24; ++k) for (int co = 0; co < 64; ++co) win[k * 64 + co] = w0->data[co * 24 + k];
        for (int co = 0; co < 64; ++co) win[24 * 64 + co] = b0->data[co];
        for (int i = 0; i < 72; ++i) wout[i] = wl->data[i];
        wout[72] = bl->data[0];
        if ((rc = upload(h, win, &h->w_in)) || (rc = upload(h, wout, &h->w_out))) break;
        if ((rc = make_block(h, m, "blocks.1.", 4, 64, &h->blk[0]))) break;
        if ((rc = make_block(h, m, "blocks.2.", 3, 64, &h->blk[1]))) break;
    } while (0);
    if (rc) { nunif_hip_row_flow_destroy(h); return rc; }
    *handle = h;
    return NUNIF_HIP_OK;
}

extern "C" void nunif_hip_row_flow_destroy(nunif_row_flow *h) {
    if (!h) return;
    for (void *p : h->owned) (void)hipFree(p);
    h->f.release(); h->t1.release(); h->t2.release();
    delete h;
}

extern "C" int nunif_hip_row_flow_delta(nunif_row_flow *h, const float *x, float *delta, int32_t B, int32_t hh,
                                        int32_t ww, int32_t flip, void *stream) {
    NUNIF_REQUIRE(h && x && delta && B > 0 && hh > 0 && ww > 0, "row_flow_delta: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int Hp = hh + (12 - hh % 12), Wp = ww + (96 - ww % 96), Wq = Wp / 8;          // row_flow_v3.py:58-59
    const size_t tok = (size_t)B * Hp * Wq;
    int rc;
    if ((rc = h->f.ensure(tok * 64 * sizeof(f16))) || (rc = h->t1.ensure(tok * 64 * sizeof(f16))) ||
        (rc = h->t2.ensure(tok * 64 * sizeof(f16))))
        return rc;
    f16 *f = (f16 *)h->f.p, *t1 = (f16 *)h->t1.p, *t2 = (f16 *)h->t2.p;
    {
        ProfScope ps("rf_input_kernel", s, 2.0 * 24 * 64 * (double)tok, (double)tok * (12.0 * 8 + 128.0));
        RfInputArgs a;
        a.x = x; a.w = h->w_in; a.out = f; a.B = B; a.h = hh; a.w_ = ww; a.Hp = Hp; a.Wq = Wq; a.flip = flip;
        rf_input_kernel<<<(unsigned)((tok + 255) / 256), 256, 0, s>>>(a);
        NUNIF_LAUNCH_CHECK();
    }
    f16 *cur = f, *other = t2;
    for (int bi = 0; bi < 2; ++bi) {
        const WaBlock &bk = h->blk[bi];
        {
            WmhaArgs a;
            memset(&a, 0, sizeof(a));
            a.x = cur; a.wfrag = bk.wfrag; a.bqkv = bk.bqkv; a.bproj = bk.bproj; a.btab = bk.btab;
            a.B = B; a.H = Hp; a.W = Wq;
            if ((rc = launch_wmha(a, bk.window, 64, s))) return rc;
        }
        {   // conv_mlp[0..1]: 1x1 + GELU(erf)
            GemmArgs g;
            memset(&g, 0, sizeof(g));
            g.a = cur; g.B = B; g.Hi = Hp; g.Wi = Wq; g.Cin = 64; g.Ho = Hp; g.Wo = Wq; g.stride = 1; g.kw = 1;
            g.K = 64; g.w = bk.w1; g.bias = bk.b1; g.N = 64; g.mode = 0; g.act = 1; g.out = t1; g.ldo = 64; g.n_real = 64; g.ps = 1;
            if ((rc = launch_gemm(g, s, "rowflow_mlp0"))) return rc;
        }
        {   // conv_mlp[2..4]: ReplicationPad2d(1) + 3x3 + LeakyReLU(0.1), then the block's residual
            ConvArgs c;
            memset(&c, 0, sizeof(c));
            c.a = t1; c.B = B; c.Hi = Hp; c.Wi = Wq; c.Cin = 64; c.Ho = Hp; c.Wo = Wq; c.stride = 1; c.kh = 3; c.kw = 3;
            c.wstream = bk.w3; c.bias = bk.b3; c.N = 64; c.n_real = 64; c.act = 2; c.slope = 0.1f; c.out = other;
            c.rpad = 1; c.res = cur;
            if ((rc = launch_conv(c, s))) return rc;
        }
        std::swap(cur, other);
    }
    {
        const long px = (long)B * hh * ww;
        ProfScope ps("rf_output_kernel", s, 2.0 * 72 * (double)px, (double)px * (4.0 + 16.0));
        RfOutputArgs a;
        a.f = cur; a.w = h->w_out; a.delta = delta; a.B = B; a.h = hh; a.w_ = ww; a.Hp = Hp; a.Wq = Wq;
        rf_output_kernel<<<(unsigned)((px + 255) / 256), 256, 0, s>>>(a);
        NUNIF_LAUNCH_CHECK();
    }
    return NUNIF_HIP_OK;
}

static int run_delta_warp(const float *c, const float *delta, const float *weight, float *out, int B, int C, int H,
                          int W, int dh, int dw, int L, double delta_scale, int flip, hipStream_t s) {
    DeltaWarpArgs a;
    a.c = c; a.delta = delta; a.weight = weight; a.out = out; a.B = B; a.C = C; a.H = H; a.W = W; a.h = dh; a.w = dw;
    a.flip = flip; a.L = L; a.delta_scale = (float)delta_scale;
    const long total = (long)B * H * W;
    ProfScope ps("delta_warp_kernel", s, 0.0, (double)total * (4.0 + 8.0 * C + (weight ? 4.0 * L : 0.0)));
    delta_warp_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(a);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

extern "C" int nunif_hip_delta_warp(const float *c, const float *delta, float *out, int32_t B, int32_t C, int32_t H,
                                    int32_t W, int32_t dh, int32_t dw, double delta_scale, int32_t flip, void *stream) {
    NUNIF_REQUIRE(c && delta && out && B > 0 && C > 0 && C <= 4 && H > 0 && W > 0 && dh > 0 && dw > 0, "delta_warp: bad argument");
    return run_delta_warp(c, delta, nullptr, out, B, C, H, W, dh, dw, 1, delta_scale, flip, (hipStream_t)stream);
}

extern "C" int nunif_hip_delta_weight_warp(const float *c, const float *delta, const float *weight, float *out, int32_t B,
                                           int32_t C, int32_t H, int32_t W, int32_t dh, int32_t dw, int32_t L,
                                           double delta_scale, int32_t flip, void *stream) {
    NUNIF_REQUIRE(c && delta && weight && out && B > 0 && C > 0 && C <= 4 && H > 0 && W > 0 && dh > 0 && dw > 0 && L > 0 && L <= 4,
                  "delta_weight_warp: bad argument");
    return run_delta_warp(c, delta, weight, out, B, C, H, W, dh, dw, L, delta_scale, flip, (hipStream_t)stream);
}

// ---- MLBW --------------------------------------------------------------------------------------------------------------
struct nunif_mlbw {
    std::vector<void *> owned;
    int L = 2, C = 64, n_blocks = 4, hole_mask = 0;
    float *w_in = nullptr, *w_out = nullptr;
    WaBlock blk[4];
    int sy[4] = {0, 0, 0, 0}, sx[4] = {0, 0, 0, 0};
    Buf f, x1, t1, t2;
};

extern "C" int nunif_hip_mlbw_create(const nunif_tensor_desc *tensors, int32_t n_tensors, nunif_mlbw **handle) {
    NUNIF_REQUIRE(tensors && handle && n_tensors > 0, "mlbw_create: NULL argument");
    TMap m;
    for (int i = 0; i < n_tensors; ++i) {
        HostT t;
        t.data = tensors[i].data;
        t.numel = 1;
        for (int d = 0; d < tensors[i].ndim; ++d) { t.shape.push_back(tensors[i].shape[d]); t.numel *= tensors[i].shape[d]; }
        m[tensors[i].name] = t;
    }
    nunif_mlbw *h = new nunif_mlbw();
    int rc = NUNIF_HIP_OK;
    do {
        const HostT *wi, *bi, *wo, *bo;
        if ((rc = find(m, "lv1_in.1.weight", &wi)) || (rc = find(m, "lv1_in.1.bias", &bi)) ||
            (rc = find(m, "lv1_out.1.weight", &wo)) || (rc = find(m, "lv1_out.1.bias", &bo)))
            break;
        const int Cs = (int)bi->numel;                  // C / 8
        h->C = Cs * 8; h->L = h->C / 32;
        h->hole_mask = bo->numel == 2 * h->L + 1 ? 1 : 0;             // sbs.mask_mlbw_l2 (mlbw.py:275-278)
        const int no = 2 * h->L + h->hole_mask;
        if ((Cs != 8 && Cs != 16) || wi->numel != (int64_t)Cs * 27 || bo->numel != no || no > 8 ||
            wo->numel != (int64_t)no * Cs * 9) {
            set_error("mlbw: unsupported shape (C/8 = %d, %lld outputs)", Cs,
                      (long long)bo->numel);
            rc = NUNIF_HIP_EUNSUPPORTED;
            break;
        }
        h->n_blocks = m.count("lv2.3.mha.mha.qkv_proj.weight") ? 4 : 2;
        // shifts (mlbw.py:57-68): full (T,T),(F,F),(T,T),(F,F); small (F,T),(F,F)
        if (h->n_blocks == 4) { h->sy[0] = h->sx[0] = 2; h->sy[2] = h->sx[2] = 2; }
        else { h->sx[0] = 2; }
        std::vector<float> win(28 * Cs), wout((size_t)(Cs * 9 + 1) * no);
        for (int k = 0; k < 27; ++k) for (int co = 0; co < Cs; ++co) win[k * Cs + co] = wi->data[co * 27 + k];
        for (int co = 0; co < Cs; ++co) win[27 * Cs + co] = bi->data[co];
        for (int k = 0; k < Cs * 9; ++k) for (int o = 0; o < no; ++o) wout[(size_t)k * no + o] = wo->data[(size_t)o * Cs * 9 + k];
        for (int o = 0; o < no; ++o) wout[(size_t)Cs * 9 * no + o] = bo->data[o];
        if ((rc = upload(h, win, &h->w_in)) || (rc = upload(h, wout, &h->w_out))) break;
        for (int i = 0; i < h->n_blocks && !rc; ++i)
            rc = make_block(h, m, "lv2." + std::to_string(i) + ".", 4, h->C, &h->blk[i]);
    } while (0);
    if (rc) { nunif_hip_mlbw_destroy(h); return rc; }
    *handle = h;
    return NUNIF_HIP_OK;
}

extern "C" void nunif_hip_mlbw_destroy(nunif_mlbw *h) {
    if (!h) return;
    for (void *p : h->owned) (void)hipFree(p);
    h->f.release(); h->x1.release(); h->t1.release(); h->t2.release();
    delete h;
}

extern "C" int32_t nunif_hip_mlbw_num_layers(const nunif_mlbw *h) { return h ? h->L : 0; }
extern "C" int32_t nunif_hip_mlbw_has_hole_mask(const nunif_mlbw *h) { return h ? h->hole_mask : 0; }

extern "C" int nunif_hip_mlbw_delta(nunif_mlbw *h, const float *x, float *delta, float *weight, int32_t B, int32_t hh,
                                    int32_t ww, int32_t flip, void *stream) {
    return nunif_hip_mlbw_delta_mask(h, x, delta, weight, nullptr, B, hh, ww, flip, stream);
}

extern "C" int nunif_hip_mlbw_delta_mask(nunif_mlbw *h, const float *x, float *delta, float *weight, float *mask_logits,
                                         int32_t B, int32_t hh, int32_t ww, int32_t flip, void *stream) {
    NUNIF_REQUIRE(h && x && delta && weight && B > 0 && hh > 0 && ww > 0, "mlbw_delta: bad argument");
    NUNIF_REQUIRE(!mask_logits || h->hole_mask, "mlbw_delta_mask: this model has no hole-mask output");
    hipStream_t s = (hipStream_t)stream;
    const int pad_w = 32 - ww % 32, pad_h = 4 - hh % 4;                 // mlbw.py:78-93 (eval: centred)
    const int pw1 = pad_w / 2, ph1 = pad_h / 2;
    const int Hp = hh + pad_h, Wp = ww + pad_w, Wq = Wp / 8, C = h->C, Cs = C / 8;
    const size_t tok = (size_t)B * Hp * Wq;
    int rc;
    if ((rc = h->f.ensure(tok * C * sizeof(f16))) || (rc = h->x1.ensure(tok * C * sizeof(f16))) ||
        (rc = h->t1.ensure(tok * C * sizeof(f16))) || (rc = h->t2.ensure(tok * C * sizeof(f16))))
        return rc;
    f16 *f = (f16 *)h->f.p, *x1 = (f16 *)h->x1.p, *t1 = (f16 *)h->t1.p, *t2 = (f16 *)h->t2.p;
    {
        const long px = (long)B * Hp * Wp;
        ProfScope ps("mlbw_in_kernel", s, 2.0 * 27 * Cs * (double)px, (double)px * (12.0 + 4.0 * Cs));
        MlbwInArgs a;
        a.x = x; a.w = h->w_in; a.out = f; a.x1 = x1; a.B = B; a.h = hh; a.w_ = ww; a.Hp = Hp; a.Wp = Wp; a.ph1 = ph1;
        a.pw1 = pw1; a.Cs = Cs; a.flip = flip;
        mlbw_in_kernel<<<(unsigned)((px + 255) / 256), 256, 0, s>>>(a);
        NUNIF_LAUNCH_CHECK();
    }
    f16 *cur = f, *other = t2;
    for (int bi = 0; bi < h->n_blocks; ++bi) {
        const WaBlock &bk = h->blk[bi];
        {
            WmhaArgs a;
            memset(&a, 0, sizeof(a));
            a.x = cur; a.wfrag = bk.wfrag; a.bqkv = bk.bqkv; a.bproj = bk.bproj; a.btab = bk.btab;
            a.B = B; a.H = Hp; a.W = Wq; a.sy = h->sy[bi]; a.sx = h->sx[bi];
            if ((rc = launch_wmha(a, 4, C, s))) return rc;
        }
        {
            GemmArgs g;
            memset(&g, 0, sizeof(g));
            g.a = cur; g.B = B; g.Hi = Hp; g.Wi = Wq; g.Cin = C; g.Ho = Hp; g.Wo = Wq; g.stride = 1; g.kw = 1;
            g.K = C; g.w = bk.w1; g.bias = bk.b1; g.N = C; g.mode = 0; g.act = 1; g.out = t1; g.ldo = C; g.n_real = C; g.ps = 1;
            if ((rc = launch_gemm(g, s, "mlbw_mlp0"))) return rc;
        }
        {   // ReplicationPad2d(1) + 3x3 (no activation), then the block's residual
            ConvArgs c;
            memset(&c, 0, sizeof(c));
            c.a = t1; c.B = B; c.Hi = Hp; c.Wi = Wq; c.Cin = C; c.Ho = Hp; c.Wo = Wq; c.stride = 1; c.kh = 3; c.kw = 3;
            c.wstream = bk.w3; c.bias = bk.b3; c.N = C; c.n_real = C; c.act = 0; c.out = other; c.rpad = 1; c.res = cur;
            if ((rc = launch_conv(c, s))) return rc;
        }
        std::swap(cur, other);
    }
    {
        const long px = (long)B * hh * ww;
        ProfScope ps("mlbw_out_kernel", s, 2.0 * 9 * Cs * 2 * h->L * (double)px, (double)px * (8.0 * h->L + 36.0 * Cs));
        MlbwOutArgs a;
        a.f = cur; a.x1 = x1; a.w = h->w_out; a.delta = delta; a.weight = weight; a.B = B; a.h = hh; a.w_ = ww; a.Hp = Hp;
        a.Wp = Wp; a.ph1 = ph1; a.pw1 = pw1; a.Cs = Cs; a.L = h->L; a.no = 2 * h->L + h->hole_mask; a.flip = flip;
        a.mask = mask_logits;
        mlbw_out_kernel<<<(unsigned)((px + 255) / 256), 256, 0, s>>>(a);
        NUNIF_LAUNCH_CHECK();
    }
    return NUNIF_HIP_OK;
}
