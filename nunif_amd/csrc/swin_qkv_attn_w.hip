// Fused qkv projection + (shifted) 6x6 window attention, "one window per wave" form, for gfx950.
//
// Replaces torchvision shifted_window_attention steps 2-7 (roll, window partition, qkv Linear,
// q*scale, QK^T + relative-position bias + shift mask, softmax, PV), called from waifu2x/models/swin_unet.py:26-36.
// Covers both channel widths of swin_unet: C = 96 (6 heads x 16) and C = 192 (6 heads x 32).
//
// Why this shape (measured on MI355X): the first fused kernel (swin_qkv_attn.hip, one wave per
// HEAD, 4 windows per workgroup) is MFMA-issue bound on the two SIMDs that must host two of its six waves, needs the
// whole x tile in LDS and all barriers of a 6-wave workgroup.  Here every wave owns ONE window (36 tokens padded to
// 3 MFMA tiles) for all heads, so waves are independent and evenly spread over the 4 SIMDs; the only shared thing is
// the qkv weight matrix, which all 8 waves of the workgroup consume in the same order — it is streamed through the
// same 2 x 8 KiB LDS ring as the block tail (swin_block_tail.hip), one chunk ahead of use.
//
// Everything after the GEMM stays in registers (accumulator layout: lane = token l&15, 4 channels 4*(l>>4)+r):
//   * Q / K tiles are directly the B / A fragments of S^T = K Q^T;
//   * V is computed with the MFMA operands swapped = A fragment of O^T = V^T P^T;   exp(S^T) is directly P^T;
//   * relative-position bias, the padded-key mask and the shifted-window region mask are folded into the score
//     MFMA by extending its reduction dimension with 48 extra columns:
//         c < 36 : Rq = bias_h[q_loc][c]          Rk = [k_loc == c]           -> + bias_h[q_loc][k_loc]
//         c = 36 : Rq = BIG                        Rk = [key is a real token]  -> padded keys fall BIG below
//         40-43  : Rq = 100 [reg(q) == c-40]       Rk = [reg(k) == c-40]       -> other shift regions fall 100 below
//     (softmax is invariant to the per-row constant).
#include "swin_kernels.h"

namespace nunif {

#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

constexpr int kWavesW = 8;          // windows per workgroup round
constexpr int kChunkW = 8;          // fragments (KiB) per ring chunk: 512 threads x 16 B
constexpr float kBig = 1000.0f;
constexpr float kRegion = 100.0f;

struct QkvAttnWArgs {
    const f16 *x;            // [B,H,W,C]
    f16 *att;                // [B,H,W,C]
    const f16 *wstream;      // per head: Wq tiles, Wk tiles, Wv tiles, each (nt, ks) fragment-major; padded to x8
    int n_chunks;
    const float *bqkv;       // [3C]
    const float *bias;       // [heads][36][48] relative-position bias
    int B, H, W, shift, n_windows;
    float scale;
};

__device__ __forceinline__ f16x8 cat8(f16x4 lo, f16x4 hi) {
    return (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int C, int HD>
__global__ void __launch_bounds__(512)
qkv_attn_w_kernel(QkvAttnWArgs a) {
    constexpr int KS = C / 32;
    constexpr int HEADS = C / HD;
    constexpr int NTH = HD / 16;                      // 16-row weight tiles per head for each of q, k, v
    constexpr int CH = kChunkW;
    static_assert(HEADS == 6, "swin_unet uses 6 heads at every level");
    __shared__ __attribute__((aligned(16))) f16x8 ring[2][CH * 64];
    __shared__ __attribute__((aligned(16))) f16 bt[HEADS * 36 * 36];      // bias table fp16 [head][q][k]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r16 = lane & 15;
    const int grp = lane >> 4;
    const int nwx = a.W / 6, nwy = a.H / 6;
    const f16x4 zero4 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};

    for (int i = tid; i < HEADS * 36 * 36; i += 512) {
        const int hq = i / 36, col = i - hq * 36;
        bt[i] = (f16)a.bias[(long)hq * 48 + col];
    }

    // ---- weight ring (identical protocol to swin_block_tail.hip, stream is circular over rounds) --------------------
    const f16x8 *gsrc = reinterpret_cast<const f16x8 *>(a.wstream) + tid;
    f16x8 st = gsrc[0];
    int gc = 0;                                         // chunks consumed so far (ring parity)
    auto wfrag = [&](int fi) -> f16x8 {
        if (fi % CH == 0) {
            const int c = fi / CH;
            ring[gc & 1][tid] = st;
            __syncthreads();
            st = gsrc[(long)(c + 1 < a.n_chunks ? c + 1 : 0) * (CH * 64)];
            ++gc;
        }
        return ring[(gc - 1) & 1][(fi % CH) * 64 + lane];
    };

    const int rounds = (a.n_windows + kWavesW * gridDim.x - 1) / (kWavesW * gridDim.x);
    for (int round = 0; round < rounds; ++round) {
        int wi = (round * gridDim.x + blockIdx.x) * kWavesW + wave;
        const bool live = wi < a.n_windows;
        if (!live) wi = a.n_windows - 1;
        const int wx = wi % nwx;
        const int t2 = wi / nwx;
        const int wy = t2 % nwy;
        const int b = t2 / nwy;
        auto pix_of = [&](int t) -> long {               // window-local token -> pixel of the un-rolled map
            t = min(t, 35);
            const int iy = t / 6, ix = t - 6 * iy;
            int yy = wy * 6 + iy + a.shift, xx = wx * 6 + ix + a.shift;
            if (yy >= a.H) yy -= a.H;
            if (xx >= a.W) xx -= a.W;
            return ((long)b * a.H + yy) * a.W + xx;
        };
        auto reg_of = [&](int t) -> int {                // shift region inside the window (0..3), -1 when unshifted
            if (a.shift == 0) return -1;
            t = min(t, 35);
            const int iy = t / 6, ix = t - 6 * iy;
            return ((wy == nwy - 1 && iy >= 3) ? 2 : 0) + ((wx == nwx - 1 && ix >= 3) ? 1 : 0);
        };

        // ---- x of this window as B fragments, pixel indices, Rk one-hot fragments -----------------------------------
        f16x8 xf[3][KS];
        long pix[3];
        f16x4 rk[3][3];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
            const int tok = 16 * mt + r16;
            pix[mt] = pix_of(tok);
            const f16 *p = a.x + pix[mt] * C + 8 * grp;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xf[mt][ks] = *reinterpret_cast<const f16x8 *>(p + 32 * ks);
            const int regk = reg_of(tok);
#pragma unroll
            for (int js = 0; js < 3; ++js) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = 16 * js + 4 * grp + j;
                    bool one;
                    if (col < 36) one = tok == col;                       // tok < 36 only when it is a real token
                    else if (col == 36) one = tok < 36;                   // "real key" column
                    else if (col >= 40 && col < 44) one = regk == col - 40;
                    else one = false;
                    rk[mt][js][j] = (f16)(one ? 1.f : 0.f);
                }
            }
        }

        int fi = 0;
#pragma unroll 1
        for (int head = 0; head < HEADS; ++head) {
            // ---- q, k (channels x tokens) and v (tokens x channels: operands swapped) of this head ---------------------
            f16x4 qt4[NTH][3], kt4[NTH][3], vt4[NTH][3];
#pragma unroll
            for (int part = 0; part < 3; ++part) {
#pragma unroll
                for (int nt = 0; nt < NTH; ++nt) {
                    f32x4 acc[3];
#pragma unroll
                    for (int mt = 0; mt < 3; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const f16x8 w = wfrag(fi);
                        ++fi;
#pragma unroll
                        for (int mt = 0; mt < 3; ++mt)
                            acc[mt] = part == 2 ? MFMA_16x16x32(xf[mt][ks], w, acc[mt]) : MFMA_16x16x32(w, xf[mt][ks], acc[mt]);
                    }
                    const int ch0 = part * C + head * HD + nt * 16;
                    if (part == 2) {
                        const float bv = a.bqkv[ch0 + r16];
#pragma unroll
                        for (int mt = 0; mt < 3; ++mt)
                            vt4[nt][mt] = (f16x4){(f16)(acc[mt][0] + bv), (f16)(acc[mt][1] + bv), (f16)(acc[mt][2] + bv),
                                                  (f16)(acc[mt][3] + bv)};
                    } else {
                        const float4 bb = *reinterpret_cast<const float4 *>(a.bqkv + ch0 + 4 * grp);
                        const float sc = part == 0 ? a.scale : 1.0f;
#pragma unroll
                        for (int mt = 0; mt < 3; ++mt) {
                            const f16x4 v = {(f16)((acc[mt][0] + bb.x) * sc), (f16)((acc[mt][1] + bb.y) * sc),
                                             (f16)((acc[mt][2] + bb.z) * sc), (f16)((acc[mt][3] + bb.w) * sc)};
                            if (part == 0) qt4[nt][mt] = v; else kt4[nt][mt] = v;
                        }
                    }
                }
            }

            // ---- attention of this head: 3 q tiles x 3 key tiles -------------------------------------------------------
#pragma unroll
            for (int qt = 0; qt < 3; ++qt) {
                const int tokq = min(16 * qt + r16, 35);
                const f16 *brow = &bt[(head * 36 + tokq) * 36];
                const f16x4 rq0 = *reinterpret_cast<const f16x4 *>(brow + 4 * grp);
                const f16x4 rq1 = *reinterpret_cast<const f16x4 *>(brow + 16 + 4 * grp);
                f16x4 rq2 = zero4;
                if (grp == 0) {
                    rq2 = *reinterpret_cast<const f16x4 *>(brow + 32);
                } else if (grp == 1) {
                    rq2[0] = (f16)kBig;                                  // column 36: every real key gets +BIG
                } else if (grp == 2) {
                    const int regq = reg_of(16 * qt + r16);
#pragma unroll
                    for (int j = 0; j < 4; ++j) rq2[j] = (f16)(regq == j ? kRegion : 0.f);
                }
                f32x4 s[3];
                float mx = -3.0e38f;
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (HD == 16) {
                        acc = MFMA_16x16x32(cat8(kt4[0][kt], rk[kt][0]), cat8(qt4[0][qt], rq0), acc);
                        acc = MFMA_16x16x32(cat8(rk[kt][1], rk[kt][2]), cat8(rq1, rq2), acc);
                    } else {
                        acc = MFMA_16x16x32(cat8(kt4[0][kt], kt4[1][kt]), cat8(qt4[0][qt], qt4[1][qt]), acc);
                        acc = MFMA_16x16x32(cat8(rk[kt][0], rk[kt][1]), cat8(rq0, rq1), acc);
                        acc = MFMA_16x16x32(cat8(rk[kt][2], zero4), cat8(rq2, zero4), acc);
                    }
                    s[kt] = acc;
                    mx = fmaxf(mx, fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])));
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                float sum = 0.f;
                f16x4 pf[3];
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    const float p0 = __expf(s[kt][0] - mx), p1 = __expf(s[kt][1] - mx);
                    const float p2 = __expf(s[kt][2] - mx), p3 = __expf(s[kt][3] - mx);
                    sum += (p0 + p1) + (p2 + p3);
                    pf[kt] = (f16x4){(f16)p0, (f16)p1, (f16)p2, (f16)p3};
                }
                sum += __shfl_xor(sum, 16);
                sum += __shfl_xor(sum, 32);
                const float inv = 1.0f / sum;
                const bool store = live && (16 * qt + r16) < 36;
#pragma unroll
                for (int dt = 0; dt < NTH; ++dt) {
                    f32x4 o = {0.f, 0.f, 0.f, 0.f};
                    o = MFMA_16x16x32(cat8(vt4[dt][0], vt4[dt][1]), cat8(pf[0], pf[1]), o);
                    o = MFMA_16x16x32(cat8(vt4[dt][2], zero4), cat8(pf[2], zero4), o);
                    if (store) {
                        const f16x4 ov = {(f16)(o[0] * inv), (f16)(o[1] * inv), (f16)(o[2] * inv), (f16)(o[3] * inv)};
                        *reinterpret_cast<f16x4 *>(a.att + pix[qt] * C + head * HD + dt * 16 + 4 * grp) = ov;
                    }
                }
            }
        }
    }
}

int qkv_attn_w_stream_frags(int C) { return 3 * (C / 16) * (C / 32); }

int launch_qkv_attn_w(const f16 *x, f16 *att, const f16 *wstream, const float *bqkv, const float *bias, int B, int H,
                      int W, int C, int heads, int shift, hipStream_t s) {
    NUNIF_REQUIRE(H % 6 == 0 && W % 6 == 0, "qkv_attn: %dx%d not a multiple of the 6x6 window", H, W);
    NUNIF_REQUIRE(heads == 6 && (C == 96 || C == 192), "qkv_attn: C=%d heads=%d unsupported", C, heads);
    if (H <= 6) shift = 0;                 // torchvision disables the shift when the window covers the map
    QkvAttnWArgs a;
    a.x = x; a.att = att; a.wstream = wstream; a.bqkv = bqkv; a.bias = bias;
    a.n_chunks = (qkv_attn_w_stream_frags(C) + kChunkW - 1) / kChunkW;
    a.B = B; a.H = H; a.W = W; a.shift = shift;
    a.n_windows = B * (H / 6) * (W / 6);
    a.scale = 1.0f / sqrtf((float)(C / heads));
    const double tok = (double)B * H * W;
    const int wgs = (a.n_windows + kWavesW - 1) / kWavesW;
    const int grid = wgs < 256 ? wgs : 256;             // persistent: one 8-wave workgroup per CU
    if (C == 96) {
        ProfScope ps("qkv_attn_w_kernel<96,16>", s, 2.0 * tok * C * 3.0 * C + 4.0 * tok * 36.0 * C, tok * C * 4.0);
        qkv_attn_w_kernel<96, 16><<<grid, 512, 0, s>>>(a);
    } else {
        ProfScope ps("qkv_attn_w_kernel<192,32>", s, 2.0 * tok * C * 3.0 * C + 4.0 * tok * 36.0 * C, tok * C * 4.0);
        qkv_attn_w_kernel<192, 32><<<grid, 512, 0, s>>>(a);
    }
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

}  // namespace nunif
