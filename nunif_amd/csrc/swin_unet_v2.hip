// waifu2x swin_unet_v2 ("winc_unet": WindowMHA + conv-MLP U-Net with an IR stem, shortcut PatchDown / PatchUp and a source-
// residual head) on the HIP engine: the registered names waifu2x.swin_unet_v2_1x / _2x / _4x.
//
// Reference: waifu2x/models/swin_unet_v2.py — GLUConvMLP :14-35, MLP :53-68, WACBlock :71-101, WACBlocks :104-129, IR :132-141,
// PatchDown :144-170, PatchUp :173-196, ToImage :199-213, SourceResidual :216-259, get_shift_config :262-269,
// SwinUNetV2Base :272-352 (_forward :337-352); nunif/modules/attention.py WindowMHA2d :118-161 (zero-padded shift by half a
// window, LayerNorm on the windowed tokens), MHA :94-115, WindowScoreBias :375-419.
//
// This family has no released checkpoint (it is the reference's experimental line), so the engine is a COMPOSITION of the
// kernels the other nets already run on, not a tuned fusion: every 1x1 conv / Linear / 2x2-s2 conv / pixel-shuffle head is
// gemm_kernel, every 3x3 conv is conv_kernel (replicate padding folded into the gather, LeakyReLU + the block residual in
// the epilogue), LayerNorm is layernorm_nobias_kernel.  New here: the IR input kernels, a window attention for 8 x 8 / 6 x 6
// windows with heads of 32 and the zero-padded shift (v2_wattn_kernel: one wave per (window, head), K / V of the head in
// LDS, a query per lane, fp32 VALU — correct first), GLU, the two shortcut kernels and the source-residual head.
// Maps are NHWC fp16, accumulation fp32.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "swin_kernels.h"

namespace nunif {

// ---- IR stem, path 1: 3x3 conv on the replicate-padded fp32 tile, 3 -> 16, LeakyReLU(0.2) -> ir[..., 0:16] ---------------------
__global__ void __launch_bounds__(256) v2_ir_path1_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                          const float *__restrict__ bias, f16 *__restrict__ ir, int B, int T) {
    __shared__ float ws[16 * 27 + 16];
    for (int i = threadIdx.x; i < 16 * 27; i += 256) ws[i] = w[i];
    if (threadIdx.x < 16) ws[16 * 27 + threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * T * T) return;
    const int px = (int)(i % T), py = (int)((i / T) % T), b = (int)(i / ((long)T * T));
    float v[27];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int yy = min(max(py + ky - 1, 0), T - 1), xx = min(max(px + kx - 1, 0), T - 1);
                v[ci * 9 + ky * 3 + kx] = x[(((long)b * 3 + ci) * T + yy) * T + xx];
            }
    f16 *o = ir + i * 32;
#pragma unroll
    for (int co = 0; co < 16; ++co) {
        float a = ws[16 * 27 + co];
#pragma unroll
        for (int k = 0; k < 27; ++k) a = fmaf(ws[co * 27 + k], v[k], a);
        o[co] = (f16)(a >= 0.f ? a : a * 0.2f);
    }
}

// ---- IR stem, path 2 input: pixel_unshuffle(2) + 1x1 conv 12 -> 64 -> f [B, T/2, T/2, 64] --------------------------------------
__global__ void __launch_bounds__(256) v2_ir_path2_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                          const float *__restrict__ bias, f16 *__restrict__ f, int B, int T) {
    __shared__ float ws[64 * 12 + 64];
    for (int i = threadIdx.x; i < 64 * 12; i += 256) ws[i] = w[i];
    if (threadIdx.x < 64) ws[64 * 12 + threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    const int T2 = T / 2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;          // (pixel, 16-channel group)
    if (i >= (long)B * T2 * T2 * 4) return;
    const int cg = (int)(i & 3);
    const long p = i >> 2;
    const int px = (int)(p % T2), py = (int)((p / T2) % T2), b = (int)(p / ((long)T2 * T2));
    float v[12];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int q = 0; q < 4; ++q)                                // unshuffled channel = ci * 4 + dy * 2 + dx
            v[ci * 4 + q] = x[(((long)b * 3 + ci) * T + 2 * py + (q >> 1)) * T + 2 * px + (q & 1)];
    f16 *o = f + p * 64 + cg * 16;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int co = cg * 16 + c;
        float a = ws[64 * 12 + co];
#pragma unroll
        for (int k = 0; k < 12; ++k) a = fmaf(ws[co * 12 + k], v[k], a);
        o[c] = (f16)a;
    }
}

// ---- IR stem, path 2 output: pixel_shuffle(2) of the half-resolution 64-channel map -> ir[..., 16:32] -----------------------------
__global__ void __launch_bounds__(256) v2_ir_shuffle_kernel(const f16 *__restrict__ f, f16 *__restrict__ ir, int B, int T) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * T * T) return;
    const int px = (int)(i % T), py = (int)((i / T) % T), b = (int)(i / ((long)T * T));
    const int T2 = T / 2;
    const f16 *src = f + (((long)b * T2 + py / 2) * T2 + px / 2) * 64 + (py & 1) * 2 + (px & 1);
    f16 *o = ir + i * 32 + 16;
#pragma unroll
    for (int c = 0; c < 16; ++c) o[c] = src[c * 4];
}

// ---- window attention: WS x WS tokens, heads of 32, zero-padded shift (WindowMHA2d), score bias shared by the heads ------------
// qkv: [B,H,W,3C] fp16 (q rows pre-multiplied by 32^-0.5 * log2 e), att: [B,H,W,C].  A token of a border window that lies in the
// zero padding went through LayerNorm (-> 0) and the qkv Linear (-> its bias): it is a key / value = bqkv like in the reference,
// and it is not written.  One wave per (window, head); lane = query token.
struct V2AttnArgs {
    const f16 *qkv; f16 *att; const float *btab; const float *bqkv;
    int B, H, W, C, heads, pad;       // pad = WS / 2 when shifted, else 0
};

template <int WS>
__global__ void __launch_bounds__(256) v2_wattn_kernel(V2AttnArgs a) {
    constexpr int N = WS * WS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_v2[];
    float *bt = reinterpret_cast<float *>(smem_v2);                 // [N][N] score bias (log2 e scaled)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *kl = bt + N * N + wave * (2 * N * 32);                   // [N][32] keys, then [N][32] values of this wave's head
    float *vl = kl + N * 32;
    for (int i = threadIdx.x; i < N * N; i += 256) bt[i] = a.btab[i];
    const int nwy = (a.H + 2 * a.pad) / WS, nwx = (a.W + 2 * a.pad) / WS;
    const long total = (long)a.B * nwy * nwx * a.heads;
    const long gid = (long)blockIdx.x * 4 + wave;
    const bool active = gid < total;
    const int head = active ? (int)(gid % a.heads) : 0;
    long wi = active ? gid / a.heads : 0;
    const int wx = (int)(wi % nwx);
    wi /= nwx;
    const int wy = (int)(wi % nwy), b = (int)(wi / nwy);
    const int C3 = 3 * a.C;
    const int ty = wy * WS + lane / WS - a.pad, tx = wx * WS + lane % WS - a.pad;
    const bool tok = active && lane < N;
    const bool inside = tok && ty >= 0 && ty < a.H && tx >= 0 && tx < a.W;
    const f16 *row = a.qkv + (((long)b * a.H + (inside ? ty : 0)) * a.W + (inside ? tx : 0)) * C3 + head * 32;
    float q[32];
    if (tok) {
#pragma unroll
        for (int d = 0; d < 32; d += 8) {
            f16x8 qq, kk, vv;
            if (inside) {
                qq = *reinterpret_cast<const f16x8 *>(row + d);
                kk = *reinterpret_cast<const f16x8 *>(row + a.C + d);
                vv = *reinterpret_cast<const f16x8 *>(row + 2 * a.C + d);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                q[d + j] = inside ? (float)qq[j] : a.bqkv[head * 32 + d + j];
                kl[lane * 32 + d + j] = inside ? (float)kk[j] : a.bqkv[a.C + head * 32 + d + j];
                vl[lane * 32 + d + j] = inside ? (float)vv[j] : a.bqkv[2 * a.C + head * 32 + d + j];
            }
        }
    }
    __syncthreads();
    if (!tok) return;
    float s[N];
    float m = -3.0e38f;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const f32x4 *kr = reinterpret_cast<const f32x4 *>(kl + j * 32);        // same address in every lane: LDS broadcast
        float acc = bt[lane * N + j];
#pragma unroll
        for (int d4 = 0; d4 < 8; ++d4) {
            const f32x4 k4 = kr[d4];
            acc = fmaf(q[4 * d4], k4[0], acc); acc = fmaf(q[4 * d4 + 1], k4[1], acc);
            acc = fmaf(q[4 * d4 + 2], k4[2], acc); acc = fmaf(q[4 * d4 + 3], k4[3], acc);
        }
        s[j] = acc;
        m = fmaxf(m, acc);
    }
    float o[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = 0.f;
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const float p = __builtin_amdgcn_exp2f(s[j] - m);
        l += p;
        const f32x4 *vr = reinterpret_cast<const f32x4 *>(vl + j * 32);
#pragma unroll
        for (int d4 = 0; d4 < 8; ++d4) {
            const f32x4 v4 = vr[d4];
            o[4 * d4] = fmaf(p, v4[0], o[4 * d4]); o[4 * d4 + 1] = fmaf(p, v4[1], o[4 * d4 + 1]);
            o[4 * d4 + 2] = fmaf(p, v4[2], o[4 * d4 + 2]); o[4 * d4 + 3] = fmaf(p, v4[3], o[4 * d4 + 3]);
        }
    }
    if (!inside) return;
    const float inv = 1.0f / l;
    f16 *dst = a.att + (((long)b * a.H + ty) * a.W + tx) * a.C + head * 32;
#pragma unroll
    for (int d = 0; d < 32; d += 8) {
        f16x8 ov;
#pragma unroll
        for (int j = 0; j < 8; ++j) ov[j] = (f16)(o[d + j] * inv);
        *reinterpret_cast<f16x8 *>(dst + d) = ov;
    }
}

// ---- F.glu(dim = channels): out[c] = in[c] * sigmoid(in[m + c]) -----------------------------------------------------------------
__global__ void __launch_bounds__(256) v2_glu_kernel(const f16 *__restrict__ in, f16 *__restrict__ out, long M, int m) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;           // (token, group of 8 channels)
    const int g8 = m / 8;
    if (i >= M * g8) return;
    const long t = i / g8;
    const int c = (int)(i % g8) * 8;
    const f16x8 a = *reinterpret_cast<const f16x8 *>(in + t * 2 * m + c);
    const f16x8 b = *reinterpret_cast<const f16x8 *>(in + t * 2 * m + m + c);
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)((float)a[j] / (1.0f + __expf(-(float)b[j])));
    *reinterpret_cast<f16x8 *>(out + t * m + c) = o;
}

// ---- PatchDown shortcut: pixel_unshuffle(2) + mean over groups of G channels -> out [B,H/2,W/2,C2] (:157-162) ---------------------
__global__ void __launch_bounds__(256) v2_down_shortcut_kernel(const f16 *__restrict__ x, f16 *__restrict__ out, int B, int H,
                                                               int W, int C, int C2) {
    const int G = 4 * C / C2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int H2 = H / 2, W2 = W / 2;
    if (i >= (long)B * H2 * W2 * C2) return;
    const int c2 = (int)(i % C2);
    const long p = i / C2;
    const int px = (int)(p % W2), py = (int)((p / W2) % H2), b = (int)(p / ((long)W2 * H2));
    float acc = 0.f;
    for (int g = 0; g < G; ++g) {
        const int k = c2 * G + g, c = k >> 2, q = k & 3;           // unshuffled channel k = c * 4 + dy * 2 + dx
        acc += (float)x[(((long)b * H + 2 * py + (q >> 1)) * W + 2 * px + (q & 1)) * C + c];
    }
    out[i] = (f16)(acc / (float)G);
}

// ---- PatchUp shortcut + U-Net skip: out[b,Y,X,c] = low[b,Y/2,X/2,(c*4 + (Y&1)*2 + (X&1)) / R] + skip[b,Y,X,c]  (:186-192,:348) --
__global__ void __launch_bounds__(256) v2_up_shortcut_kernel(const f16 *__restrict__ low, const f16 *__restrict__ skip,
                                                             f16 *__restrict__ out, int B, int H, int W, int C, int C2) {
    const int R = 4 * C / C2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * H * W * C) return;
    const int c = (int)(i % C);
    const long p = i / C;
    const int px = (int)(p % W), py = (int)((p / W) % H), b = (int)(p / ((long)W * H));
    const int k = c * 4 + (py & 1) * 2 + (px & 1);
    const float v = (float)low[(((long)b * (H / 2) + py / 2) * (W / 2) + px / 2) * C2 + k / R];
    out[i] = (f16)(v + (float)skip[i]);
}

// ---- SourceResidual: z = clamp(pixel_shuffle(conv3x3(replicate_pad(src)))[crop] + r * scale_bias)  (:244-259, wrapper clamp) ----
__global__ void __launch_bounds__(256) v2_source_residual_kernel(const float *__restrict__ src, const float *__restrict__ r,
                                                                 const float *__restrict__ w, const float *__restrict__ scale_bias,
                                                                 float *__restrict__ z, int B, int T, int s, int O, int clamp01) {
    // out pixel (Y, X) of the O x O plane sits at (Y + off, X + off) of the s*T plane, off = (s*T - O) / 2
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * 3 * O * O) return;
    const int X = (int)(i % O), Y = (int)((i / O) % O), c = (int)((i / ((long)O * O)) % 3), b = (int)(i / ((long)3 * O * O));
    const int off = (s * T - O) / 2;
    const int yy = Y + off, xx = X + off;
    const int sy = yy / s, sx = xx / s, n = c * s * s + (yy % s) * s + (xx % s);       // pixel_shuffle: channel n of the conv
    float acc = 0.f;
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int py = min(max(sy + ky - 1, 0), T - 1), px = min(max(sx + kx - 1, 0), T - 1);
                acc = fmaf(w[((n * 3 + ci) * 3 + ky) * 3 + kx], src[(((long)b * 3 + ci) * T + py) * T + px], acc);
            }
    const float v = acc + r[i] * scale_bias[0];
    z[i] = clamp01 ? fminf(fmaxf(v, 0.f), 1.f) : v;
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

struct Lin { f16 *w = nullptr; float *b = nullptr; int N = 0, n_real = 0, K = 0; };       // gemm_kernel packing [nt][ks]
struct Conv3 { f16 *w = nullptr; float *b = nullptr; int Cin = 0, N = 0; };                 // conv_kernel stream [ks][nt]

struct WacBlock {
    Lin qkv, proj, w1, w2lin;     // w2lin: the plain-MLP form of the LAST decoder block (1x1)
    Conv3 w2;                     // GLUConvMLP: 3x3 on mid / 2 channels
    float *norm = nullptr, *btab = nullptr, *bqkv = nullptr;
    int C = 0, heads = 0, ws = 8, shift = 0, mid = 0, glu = 1;
};

}  // namespace

struct nunif_swin_unet_v2 {
    int scale = 2, C = 96, C2 = 192;
    std::vector<void *> owned;
    float *ir1_w = nullptr, *ir1_b = nullptr, *ir2_w = nullptr, *ir2_b = nullptr, *res_w = nullptr, *scale_bias = nullptr;
    WacBlock ir_blk[2];
    Conv3 patch;
    std::vector<WacBlock> wac1, wac2, wac3;
    Lin down1, up1, to_image;
    Buf ir, irf, irf2, f1, f1b, skip, f2, f2b, tmpA, tmpB, qkv, rimg;
};

namespace {

template <typename T>
int upload(nunif_swin_unet_v2 *h, const std::vector<T> &host, T **dev) {
    void *p = nullptr;
    if (hipMalloc(&p, host.size() * sizeof(T)) != hipSuccess) { set_error("hipMalloc failed"); return NUNIF_HIP_ENOMEM; }
    h->owned.push_back(p);
    NUNIF_HIP_CHECK(hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *dev = reinterpret_cast<T *>(p);
    return NUNIF_HIP_OK;
}

// MFMA A fragment (nt, ks): lane l holds W[nt*16 + (l&15)][ks*32 + (l>>4)*8 + j], j = 0..7
template <typename F>
void put_frag(std::vector<f16> &dst, size_t frag, int nt, int ks, F wt) {
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j)
            dst[(frag * 64 + l) * 8 + j] = (f16)wt(nt * 16 + (l & 15), ks * 32 + (l >> 4) * 8 + j);
}

// rows n < n_real of a [n_real][K_real] matrix (wt), zero beyond; K padded to a multiple of 32
template <typename F, typename G>
int make_lin(nunif_swin_unet_v2 *h, int n_real, int K_real, F wt, G bias, Lin *L) {
    const int N = (n_real + 15) / 16 * 16, K = (K_real + 31) / 32 * 32;
    std::vector<f16> packed((size_t)N * K + 8192, (f16)0.f);      // + 16 KiB: the ring prefetches one chunk past the end
    for (int nt = 0; nt < N / 16; ++nt)
        for (int ks = 0; ks < K / 32; ++ks)
            put_frag(packed, (size_t)nt * (K / 32) + ks, nt, ks,
                     [&](int n, int k) { return (n < n_real && k < K_real) ? wt(n, k) : 0.f; });
    std::vector<float> b(N, 0.f);
    for (int n = 0; n < n_real; ++n) b[n] = bias(n);
    L->N = N; L->n_real = n_real; L->K = K;
    int rc = upload(h, packed, &L->w);
    return rc ? rc : upload(h, b, &L->b);
}

// 3x3 conv [N][Cin][3][3] -> conv_kernel stream, k = tap * Cin + ci
int make_conv3(nunif_swin_unet_v2 *h, const HostT *w, const HostT *b, int N, int Cin, Conv3 *cv) {
    NUNIF_REQUIRE(w->numel == (int64_t)N * Cin * 9 && b->numel == N && Cin % 32 == 0 && N % 16 == 0, "3x3 conv shape");
    const int NT = N / 16, KS = 9 * Cin / 32;
    std::vector<f16> stream((size_t)KS * NT * 512 + 8192, (f16)0.f);
    const float *wd = w->data;
    for (int ks = 0; ks < KS; ++ks)
        for (int nt = 0; nt < NT; ++nt)
            put_frag(stream, (size_t)ks * NT + nt, nt, ks, [=](int n, int k) {
                const int tap = k / Cin, ci = k % Cin;
                return wd[((size_t)n * Cin + ci) * 9 + tap]; });
    std::vector<float> bb(b->data, b->data + N);
    cv->Cin = Cin; cv->N = N;
    int rc = upload(h, stream, &cv->w);
    return rc ? rc : upload(h, bb, &cv->b);
}

double gelu_erf_d(double v) { return 0.5 * v * (1.0 + erf(v * 0.70710678118654752440)); }

int make_wac(nunif_swin_unet_v2 *h, const TMap &m, const std::string &p, int C, int heads, int ws, int shift, WacBlock *bk) {
    const HostT *wqkv, *bqkv, *wp, *bp, *nw, *w1, *b1, *w2, *b2, *tw0, *tb0, *tw2, *tb2;
    int rc;
    if ((rc = find(m, p + "mha.mha.qkv_proj.weight", &wqkv)) || (rc = find(m, p + "mha.mha.qkv_proj.bias", &bqkv)) ||
        (rc = find(m, p + "mha.mha.head_proj.weight", &wp)) || (rc = find(m, p + "mha.mha.head_proj.bias", &bp)) ||
        (rc = find(m, p + "norm.weight", &nw)) || (rc = find(m, p + "conv_mlp.w1.weight", &w1)) ||
        (rc = find(m, p + "conv_mlp.w1.bias", &b1)) || (rc = find(m, p + "conv_mlp.w2.weight", &w2)) ||
        (rc = find(m, p + "conv_mlp.w2.bias", &b2)) || (rc = find(m, p + "relative_bias.to_bias.0.weight", &tw0)) ||
        (rc = find(m, p + "relative_bias.to_bias.0.bias", &tb0)) || (rc = find(m, p + "relative_bias.to_bias.2.weight", &tw2)) ||
        (rc = find(m, p + "relative_bias.to_bias.2.bias", &tb2)))
        return rc;
    NUNIF_REQUIRE(wqkv->numel == (int64_t)3 * C * C && wp->numel == (int64_t)C * C && nw->numel == C && C == heads * 32,
                  "%s: expected %d channels in heads of 32", p.c_str(), C);
    bk->C = C; bk->heads = heads; bk->ws = ws; bk->shift = shift;
    const float qs = (1.0f / sqrtf(32.0f)) * 1.4426950408889634f;       // head_dim^-0.5 * log2(e), folded into q
    {
        const float *wd = wqkv->data, *bd = bqkv->data;
        if ((rc = make_lin(h, 3 * C, C, [=](int n, int k) { return wd[(size_t)n * C + k] * (n < C ? qs : 1.f); },
                           [=](int n) { return bd[n] * (n < C ? qs : 1.f); }, &bk->qkv)))
            return rc;
        std::vector<float> bq(3 * C);
        for (int n = 0; n < 3 * C; ++n) bq[n] = bd[n] * (n < C ? qs : 1.f);
        if ((rc = upload(h, bq, &bk->bqkv))) return rc;
        const float *pd = wp->data, *pb = bp->data;
        if ((rc = make_lin(h, C, C, [=](int n, int k) { return pd[(size_t)n * C + k]; }, [=](int n) { return pb[n]; }, &bk->proj)))
            return rc;
        std::vector<float> g(nw->data, nw->data + C);
        if ((rc = upload(h, g, &bk->norm))) return rc;
    }
    {   // WindowScoreBias (attention.py:375-419): the to_bias MLP on the normalised relative offsets, evaluated once here
        const int hidden = (int)tb0->numel, N = ws * ws;
        NUNIF_REQUIRE(tw0->numel == hidden * 2 && tw2->numel == hidden && tb2->numel == 1, "%s: score-bias MLP shape", p.c_str());
        const float dmax = (float)(ws - 1);
        std::vector<float> tab((size_t)N * N);
        for (int q = 0; q < N; ++q)
            for (int k = 0; k < N; ++k) {
                const float dy = (float)(q / ws - k / ws) / dmax, dx = (float)(q % ws - k % ws) / dmax;
                double o = tb2->data[0];
                for (int j = 0; j < hidden; ++j)
                    o += (double)tw2->data[j] * gelu_erf_d((double)tw0->data[j * 2] * dy + (double)tw0->data[j * 2 + 1] * dx +
                                                            (double)tb0->data[j]);
                tab[(size_t)q * N + k] = (float)o * 1.4426950408889634f;
            }
        if ((rc = upload(h, tab, &bk->btab))) return rc;
    }
    const int mid = (int)w1->shape[0];
    bk->mid = mid;
    bk->glu = w2->shape.size() == 4 && w2->shape[2] == 3;
    {
        const float *wd = w1->data, *bd = b1->data;
        NUNIF_REQUIRE(w1->numel == (int64_t)mid * C && mid % 32 == 0, "%s: conv_mlp.w1 shape", p.c_str());
        if ((rc = make_lin(h, mid, C, [=](int n, int k) { return wd[(size_t)n * C + k]; }, [=](int n) { return bd[n]; }, &bk->w1)))
            return rc;
    }
    if (bk->glu) {
        NUNIF_REQUIRE(mid % 64 == 0, "%s: GLU needs mid / 2 in multiples of 32", p.c_str());
        if ((rc = make_conv3(h, w2, b2, C, mid / 2, &bk->w2))) return rc;
    } else {
        const float *wd = w2->data, *bd = b2->data;
        NUNIF_REQUIRE(w2->numel == (int64_t)C * mid, "%s: conv_mlp.w2 (MLP) shape", p.c_str());
        if ((rc = make_lin(h, C, mid, [=](int n, int k) { return wd[(size_t)n * mid + k]; }, [=](int n) { return bd[n]; }, &bk->w2lin)))
            return rc;
    }
    return NUNIF_HIP_OK;
}

int run_lin(const Lin &L, const f16 *a, int B, int Hi, int Wi, int Cin, int Ho, int Wo, int stride, int kw, int mode, int act,
            float slope, const f16 *res, void *out, int ldo, hipStream_t s, const char *tag) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.a = a; g.B = B; g.Hi = Hi; g.Wi = Wi; g.Cin = Cin; g.Ho = Ho; g.Wo = Wo; g.stride = stride; g.kw = kw;
    g.K = L.K; g.w = L.w; g.bias = L.b; g.N = L.N; g.mode = mode; g.act = act; g.slope = slope; g.res = res; g.out = out;
    g.ldo = ldo; g.n_real = L.n_real; g.ps = 1;
    return launch_gemm(g, s, tag);
}

template <int WS>
int launch_wattn_t(const V2AttnArgs &a, hipStream_t s) {
    constexpr size_t smem = (size_t)(WS * WS * WS * WS + 4 * 2 * WS * WS * 32) * sizeof(float);
    static bool configured = false;
    if (!configured) {
        NUNIF_HIP_CHECK(hipFuncSetAttribute((const void *)v2_wattn_kernel<WS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        configured = true;
    }
    const long units = (long)a.B * ((a.H + 2 * a.pad) / WS) * ((a.W + 2 * a.pad) / WS) * a.heads;
    v2_wattn_kernel<WS><<<(unsigned)((units + 3) / 4), 256, smem, s>>>(a);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

// one WACBlock on the NHWC map *px ([B,S,S,C]); the result ends up in *px again (the buffers swap for the 3x3 form)
int run_wac(nunif_swin_unet_v2 *h, const WacBlock &bk, f16 **px, f16 **pother, int B, int S, hipStream_t s) {
    f16 *x = *px, *other = *pother;
    const int C = bk.C;
    const long M = (long)B * S * S;
    f16 *tmpA = (f16 *)h->tmpA.p, *tmpB = (f16 *)h->tmpB.p, *qkv = (f16 *)h->qkv.p;
    NUNIF_REQUIRE(S % bk.ws == 0 && (!bk.shift || bk.ws % 2 == 0), "swin_unet_v2: map %d not a multiple of window %d", S, bk.ws);
    int rc;
    // x = x + head_proj(window_mha(qkv_proj(layer_norm(x))))          (WACBlock.forward :95-96)
    if ((rc = launch_layernorm_nobias(x, tmpA, bk.norm, M, C, s))) return rc;
    if ((rc = run_lin(bk.qkv, tmpA, B, S, S, C, S, S, 1, 1, 0, 0, 0.f, nullptr, qkv, 3 * C, s, "v2_qkv"))) return rc;
    {
        V2AttnArgs a;
        a.qkv = qkv; a.att = tmpA; a.btab = bk.btab; a.bqkv = bk.bqkv; a.B = B; a.H = S; a.W = S; a.C = C; a.heads = bk.heads;
        a.pad = bk.shift ? bk.ws / 2 : 0;
        ProfScope ps("v2_wattn_kernel", s, 4.0 * (double)M * bk.ws * bk.ws * C, (double)M * C * 8.0);
        if ((rc = bk.ws == 8 ? launch_wattn_t<8>(a, s) : bk.ws == 6 ? launch_wattn_t<6>(a, s) : NUNIF_HIP_EUNSUPPORTED)) {
            if (rc == NUNIF_HIP_EUNSUPPORTED) set_error("swin_unet_v2: window %d unsupported (8, 6)", bk.ws);
            return rc;
        }
    }
    if ((rc = run_lin(bk.proj, tmpA, B, S, S, C, S, S, 1, 1, 0, 0, 0.f, x, x, C, s, "v2_proj"))) return rc;
    if (bk.glu) {
        // x = x + leaky_relu(conv3x3(replicate_pad(glu(w1 x))), 0.2)      (GLUConvMLP :27-35)
        if ((rc = run_lin(bk.w1, x, B, S, S, C, S, S, 1, 1, 0, 0, 0.f, nullptr, tmpB, bk.mid, s, "v2_mlp_w1"))) return rc;
        const int m2 = bk.mid / 2;
        v2_glu_kernel<<<(unsigned)((M * (m2 / 8) + 255) / 256), 256, 0, s>>>(tmpB, tmpA, M, m2);
        NUNIF_LAUNCH_CHECK();
        ConvArgs c;
        memset(&c, 0, sizeof(c));
        c.a = tmpA; c.B = B; c.Hi = S; c.Wi = S; c.Cin = m2; c.Ho = S; c.Wo = S; c.stride = 1; c.kh = 3; c.kw = 3;
        c.wstream = bk.w2.w; c.bias = bk.w2.b; c.N = C; c.n_real = C; c.act = 2; c.slope = 0.2f; c.out = other; c.rpad = 1; c.res = x;
        if ((rc = launch_conv(c, s))) return rc;
        *px = other; *pother = x;
    } else {
        // x = x + w2(leaky_relu(w1 x, 0.1))                                (MLP :62-68)
        if ((rc = run_lin(bk.w1, x, B, S, S, C, S, S, 1, 1, 0, 2, 0.1f, nullptr, tmpB, bk.mid, s, "v2_mlp_w1"))) return rc;
        if ((rc = run_lin(bk.w2lin, tmpB, B, S, S, bk.mid, S, S, 1, 1, 0, 0, 0.f, x, x, C, s, "v2_mlp_w2"))) return rc;
    }
    return NUNIF_HIP_OK;
}

}  // namespace

extern "C" int nunif_hip_swin_unet_v2_create(const nunif_tensor_desc *tensors, int32_t n_tensors, int32_t scale_factor,
                                             nunif_swin_unet_v2 **handle) {
    NUNIF_REQUIRE(tensors && handle && n_tensors > 0, "swin_unet_v2_create: NULL argument");
    NUNIF_REQUIRE(scale_factor == 1 || scale_factor == 2 || scale_factor == 4, "swin_unet_v2_create: scale_factor %d (1, 2, 4)",
                  scale_factor);
    TMap m;
    for (int i = 0; i < n_tensors; ++i) {
        HostT t;
        t.data = tensors[i].data;
        t.numel = 1;
        for (int d = 0; d < tensors[i].ndim; ++d) { t.shape.push_back(tensors[i].shape[d]); t.numel *= tensors[i].shape[d]; }
        m[tensors[i].name] = t;
    }
    nunif_swin_unet_v2 *h = new nunif_swin_unet_v2();
    h->scale = scale_factor;
    const std::string P = "unet.";
    int rc = NUNIF_HIP_OK;
    do {
        const HostT *w, *b;
        if ((rc = find(m, P + "ir.path1.0.weight", &w)) || (rc = find(m, P + "ir.path1.0.bias", &b))) break;
        if (w->numel != 16 * 27 || b->numel != 16) { set_error("swin_unet_v2: IR(3, 32) expected"); rc = NUNIF_HIP_EUNSUPPORTED; break; }
        {
            std::vector<float> wv(w->data, w->data + w->numel), bv(b->data, b->data + b->numel);
            if ((rc = upload(h, wv, &h->ir1_w)) || (rc = upload(h, bv, &h->ir1_b))) break;
        }
        if ((rc = find(m, P + "ir.path2.1.weight", &w)) || (rc = find(m, P + "ir.path2.1.bias", &b))) break;
        if (w->numel != 64 * 12 || b->numel != 64) { set_error("swin_unet_v2: IR path2 conv 12 -> 64 expected"); rc = NUNIF_HIP_EUNSUPPORTED; break; }
        {
            std::vector<float> wv(w->data, w->data + w->numel), bv(b->data, b->data + b->numel);
            if ((rc = upload(h, wv, &h->ir2_w)) || (rc = upload(h, bv, &h->ir2_b))) break;
        }
        if ((rc = make_wac(h, m, P + "ir.path2.2.", 64, 2, 8, 1, &h->ir_blk[0])) ||
            (rc = make_wac(h, m, P + "ir.path2.3.", 64, 2, 8, 0, &h->ir_blk[1])))
            break;
        if ((rc = find(m, P + "patch.weight", &w)) || (rc = find(m, P + "patch.bias", &b))) break;
        const int C = (int)w->shape[0];
        if ((rc = make_conv3(h, w, b, C, 32, &h->patch))) break;
        if ((rc = find(m, P + "down1.conv.weight", &w)) || (rc = find(m, P + "down1.conv.bias", &b))) break;
        const int C2 = (int)w->shape[0];
        h->C = C; h->C2 = C2;
        if (C % 32 || C2 % 32 || (4 * C) % C2) { set_error("swin_unet_v2: base_dim %d / %d unsupported", C, C2); rc = NUNIF_HIP_EUNSUPPORTED; break; }
        {   // down1: 2x2 stride-2 conv as a gather GEMM, k = (dy*2 + dx) * C + ci
            const float *wd = w->data, *bd = b->data;
            if ((rc = make_lin(h, C2, 4 * C, [=](int n, int k) { const int tap = k / C, ci = k % C; return wd[((size_t)n * C + ci) * 4 + tap]; },
                               [=](int n) { return bd[n]; }, &h->down1)))
                break;
        }
        if ((rc = find(m, P + "up1.proj.weight", &w)) || (rc = find(m, P + "up1.proj.bias", &b))) break;
        if (w->numel != (int64_t)4 * C * C2) { set_error("swin_unet_v2: up1.proj shape"); rc = NUNIF_HIP_EINVAL; break; }
        {   // up1: gemm mode 1 writes column n' = q * C + c to sub-pixel q: torch's pixel_shuffle takes channel c * 4 + q
            const float *wd = w->data, *bd = b->data;
            if ((rc = make_lin(h, 4 * C, C2, [=](int n, int k) { const int q = n / C, c = n % C; return wd[(size_t)(c * 4 + q) * C2 + k]; },
                               [=](int n) { const int q = n / C, c = n % C; return bd[c * 4 + q]; }, &h->up1)))
                break;
        }
        auto count_blocks = [&](const std::string &key) {
            int n = 0;
            while (m.find(key + "blocks." + std::to_string(n) + ".mha.mha.qkv_proj.weight") != m.end()) ++n;
            return n;
        };
        const int n1 = count_blocks(P + "wac1."), n2 = count_blocks(P + "wac2."), n3 = count_blocks(P + "wac3.");
        const int heads = std::max(C / 32, 2), heads2 = std::max(C2 / 32, 2);
        // get_shift_config(n) :262-269: reversed([i % 2 == 1]), i.e. block i is shifted iff (n - 1 - i) is odd
        auto shift_of = [](int n, int i) { return ((n - 1 - i) % 2) == 1 ? 1 : 0; };
        h->wac1.resize(n1); h->wac2.resize(n2); h->wac3.resize(n3);
        const int win1[2] = {8, 6};
        for (int i = 0; i < n1 && !rc; ++i) {
            if (i >= 2) { set_error("swin_unet_v2: more than 2 first_layers"); rc = NUNIF_HIP_EUNSUPPORTED; break; }
            rc = make_wac(h, m, P + "wac1.blocks." + std::to_string(i) + ".", C, heads, win1[i], shift_of(n1, i), &h->wac1[i]);
        }
        for (int i = 0; i < n2 && !rc; ++i)
            rc = make_wac(h, m, P + "wac2.blocks." + std::to_string(i) + ".", C2, heads2, 8, shift_of(n2, i), &h->wac2[i]);
        for (int i = 0; i < n3 && !rc; ++i)
            rc = make_wac(h, m, P + "wac3.blocks." + std::to_string(i) + ".", C, heads, 8, shift_of(n3, i), &h->wac3[i]);
        if (rc) break;
        if ((rc = find(m, P + "to_residual_image.proj.weight", &w)) || (rc = find(m, P + "to_residual_image.proj.bias", &b))) break;
        const int s2 = scale_factor * scale_factor;
        if (w->numel != (int64_t)3 * s2 * C) { set_error("swin_unet_v2: to_residual_image shape"); rc = NUNIF_HIP_EINVAL; break; }
        {
            const float *wd = w->data, *bd = b->data;
            if ((rc = make_lin(h, 3 * s2, C, [=](int n, int k) { return wd[(size_t)n * C + k]; }, [=](int n) { return bd[n]; }, &h->to_image)))
                break;
        }
        if ((rc = find(m, P + "to_image.resampling.weight", &w)) || (rc = find(m, P + "to_image.scale_bias", &b))) break;
        if (w->numel != (int64_t)3 * s2 * 27 || b->numel != 1) { set_error("swin_unet_v2: to_image shape"); rc = NUNIF_HIP_EINVAL; break; }
        {
            std::vector<float> wv(w->data, w->data + w->numel), bv(b->data, b->data + 1);
            if ((rc = upload(h, wv, &h->res_w)) || (rc = upload(h, bv, &h->scale_bias))) break;
        }
    } while (0);
    if (rc) { nunif_hip_swin_unet_v2_destroy(h); return rc; }
    *handle = h;
    return NUNIF_HIP_OK;
}

extern "C" void nunif_hip_swin_unet_v2_destroy(nunif_swin_unet_v2 *h) {
    if (!h) return;
    for (void *p : h->owned) (void)hipFree(p);
    for (Buf *b : {&h->ir, &h->irf, &h->irf2, &h->f1, &h->f1b, &h->skip, &h->f2, &h->f2b, &h->tmpA, &h->tmpB, &h->qkv, &h->rimg})
        b->release();
    delete h;
}

// x: [B,3,T,T] f32 -> z: [B,3,O,O] f32 with O = T*s - 2*offset (offset = 9 s): SwinUNet{1,2,4}xV2.forward in eval mode
extern "C" int nunif_hip_swin_unet_v2_forward(nunif_swin_unet_v2 *h, const float *x, float *z, int32_t B, int32_t T,
                                              int32_t clamp01, void *stream) {
    NUNIF_REQUIRE(h && x && z && B > 0, "swin_unet_v2_forward: bad argument");
    const int S = T - 16, C = h->C, C2 = h->C2, sc = h->scale;
    NUNIF_REQUIRE(T % 2 == 0 && (T / 2) % 8 == 0 && S > 0 && S % 24 == 0 && (S / 2) % 8 == 0,
                  "tile_size %d is not valid for swin_unet_v2 (T / 2 multiple of 8, T - 16 multiple of 48)", T);
    hipStream_t s = (hipStream_t)stream;
    const size_t e2 = sizeof(f16);
    const long M1 = (long)B * S * S, M2 = M1 / 4, Mt = (long)B * T * T, Mh = Mt / 4;
    const int midmax = std::max({C * 2, C2 * 2, 64});
    int rc;
    if ((rc = h->ir.ensure(Mt * 32 * e2)) || (rc = h->irf.ensure(Mh * 64 * e2)) || (rc = h->irf2.ensure(Mh * 64 * e2)) ||
        (rc = h->f1.ensure(M1 * C * e2)) || (rc = h->f1b.ensure(M1 * C * e2)) || (rc = h->skip.ensure(M1 * C * e2)) ||
        (rc = h->f2.ensure(M2 * C2 * e2)) || (rc = h->f2b.ensure(M2 * C2 * e2)) ||
        (rc = h->tmpA.ensure(std::max<size_t>({(size_t)M1 * C, (size_t)M2 * C2, (size_t)Mh * 64}) * e2)) ||
        (rc = h->tmpB.ensure(std::max<size_t>({(size_t)M1 * midmax, (size_t)Mh * 64}) * e2)) ||
        (rc = h->qkv.ensure(std::max<size_t>({(size_t)M1 * 3 * C, (size_t)M2 * 3 * C2, (size_t)Mh * 192}) * e2)) ||
        (rc = h->rimg.ensure((size_t)B * 3 * (S * sc) * (S * sc) * sizeof(float))))
        return rc;
    f16 *ir = (f16 *)h->ir.p, *irf = (f16 *)h->irf.p, *irf2 = (f16 *)h->irf2.p;
    // ---- IR (:132-141): path1 and the input of path2 from the fp32 tile, two WAC blocks at half resolution, shuffle back --------
    v2_ir_path1_kernel<<<(unsigned)((Mt + 255) / 256), 256, 0, s>>>(x, h->ir1_w, h->ir1_b, ir, B, T);
    v2_ir_path2_kernel<<<(unsigned)((Mh * 4 + 255) / 256), 256, 0, s>>>(x, h->ir2_w, h->ir2_b, irf, B, T);
    NUNIF_LAUNCH_CHECK();
    {
        f16 *cur = irf, *other = irf2;
        if ((rc = run_wac(h, h->ir_blk[0], &cur, &other, B, T / 2, s)) || (rc = run_wac(h, h->ir_blk[1], &cur, &other, B, T / 2, s)))
            return rc;
        v2_ir_shuffle_kernel<<<(unsigned)((Mt + 255) / 256), 256, 0, s>>>(cur, ir, B, T);
        NUNIF_LAUNCH_CHECK();
    }
    // ---- patch: 3x3 VALID 32 -> C, crop 7, LeakyReLU(0.2) (:339-341): conv rows / cols [7, 7 + S) of the (T - 2)^2 result ------
    f16 *f1 = (f16 *)h->f1.p, *f1b = (f16 *)h->f1b.p;
    {
        ConvArgs c;
        memset(&c, 0, sizeof(c));
        c.a = ir + ((long)7 * T + 7) * 32; c.B = B; c.Hi = T; c.Wi = T; c.Cin = 32; c.Ho = S; c.Wo = S; c.stride = 1; c.kh = 3; c.kw = 3;
        c.wstream = h->patch.w; c.bias = h->patch.b; c.N = C; c.n_real = C; c.act = 2; c.slope = 0.2f; c.out = f1;
        if ((rc = launch_conv(c, s))) return rc;
    }
    for (auto &bk : h->wac1)
        if ((rc = run_wac(h, bk, &f1, &f1b, B, S, s))) return rc;
    // the encoder's level-1 result is the U-Net skip: keep it where the decoder does not write
    f16 *skip = f1;
    f16 *dec = f1 == (f16 *)h->f1.p ? (f16 *)h->skip.p : (f16 *)h->f1.p;       // a level-1 buffer other than skip / f1b
    // ---- down1 (:157-163): shortcut first, then the 2x2-s2 conv with LeakyReLU(0.2) accumulates onto it --------------------------
    f16 *f2 = (f16 *)h->f2.p, *f2b = (f16 *)h->f2b.p;
    v2_down_shortcut_kernel<<<(unsigned)((M2 * C2 + 255) / 256), 256, 0, s>>>(skip, f2, B, S, S, C, C2);
    NUNIF_LAUNCH_CHECK();
    if ((rc = run_lin(h->down1, skip, B, S, S, C, S / 2, S / 2, 2, 2, 0, 2, 0.2f, f2, f2, C2, s, "v2_down1"))) return rc;
    for (auto &bk : h->wac2)
        if ((rc = run_wac(h, bk, &f2, &f2b, B, S / 2, s))) return rc;
    // ---- up1 (:186-192) + skip (:348): shortcut + skip first, then proj + LeakyReLU(0.2) + pixel shuffle accumulates onto it -------
    v2_up_shortcut_kernel<<<(unsigned)((M1 * C + 255) / 256), 256, 0, s>>>(f2, skip, dec, B, S, S, C, C2);
    NUNIF_LAUNCH_CHECK();
    if ((rc = run_lin(h->up1, f2, B, S / 2, S / 2, C2, S / 2, S / 2, 1, 1, 1, 2, 0.2f, dec, dec, C, s, "v2_up1"))) return rc;
    f16 *other = f1b;
    if (other == dec || other == skip) other = (f16 *)h->skip.p == dec || (f16 *)h->skip.p == skip ? (f16 *)h->f1.p : (f16 *)h->skip.p;
    for (auto &bk : h->wac3)
        if ((rc = run_wac(h, bk, &dec, &other, B, S, s))) return rc;
    // ---- to_residual_image (:199-213): 1x1 conv, pixel shuffle, crop `scale`; then the source residual + clamp --------------------
    const int O = S * sc - 2 * sc;
    {
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.a = dec; g.B = B; g.Hi = S; g.Wi = S; g.Cin = C; g.Ho = S; g.Wo = S; g.stride = 1; g.kw = 1;
        g.K = h->to_image.K; g.w = h->to_image.w; g.bias = h->to_image.b; g.N = h->to_image.N; g.mode = 2; g.out = h->rimg.p;
        g.n_real = h->to_image.n_real; g.ps = sc; g.oshift = -sc; g.OH = O; g.OW = O; g.no_clamp = 1;
        if ((rc = launch_gemm(g, s, "v2_to_image"))) return rc;
    }
    const long n_out = (long)B * 3 * O * O;
    v2_source_residual_kernel<<<(unsigned)((n_out + 255) / 256), 256, 0, s>>>(x, (const float *)h->rimg.p, h->res_w, h->scale_bias, z,
                                                                               B, T, sc, O, clamp01);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}
