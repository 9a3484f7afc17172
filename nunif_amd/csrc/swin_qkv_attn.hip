// Fused qkv projection + (shifted) 6x6 window attention for gfx950 — one wave per head, everything in registers.
//
// Replaces torchvision shifted_window_attention steps 2-7: roll, window partition, the qkv
// Linear, q*scale, QK^T, relative-position bias, shift mask, softmax, PV; called from
// waifu2x/models/swin_unet.py:26-36 via SwinTransformerBlock.  The output projection continues in
// swin_block_tail.hip.
//
// Decomposition.  A workgroup takes 4 consecutive windows = 144 tokens = exactly 9 MFMA token tiles of 16 (no
// padding waste in the qkv GEMM) and has one wave per head.  The 144 x C input tile is staged once in LDS and read
// by every wave as MFMA B fragments; each wave keeps the qkv weights of ITS head in registers for the whole
// (persistent) kernel, so HBM/L2 see x once and the weights ~once per CU.
//
// Everything after the GEMM stays in registers, exploiting the accumulator layout (lane = token l&15, 4 channels
// 4*(l>>4)+r):
//   * Q and K tiles are directly the B / A fragments of S^T = K Q^T (16x16x16, k = head dim);
//   * V is computed with the MFMA operands swapped, which yields V^T-shaped tiles = the A fragment of O^T = V^T P^T;
//   * S^T accumulators, after exp(), are directly the B fragment P^T of that product.
// Relative-position bias, the "different window" mask inside a token tile that straddles two windows, and the
// shifted-window region mask are all folded into the score MFMAs by extending the reduction dimension:
//     S[q][k] = sum_d Q[q][d] K[k][d] + sum_c Rq[q][c] Rk[k][c]
// with columns c<36: Rq = bias_h[q_loc][c], Rk = onehot(k_loc == c)      -> + bias_h[q_loc][k_loc]
//      columns 36-39: Rq = BIG*[win(q)==c-36], Rk = [win(k)==c-36]        -> + BIG iff same window
//      columns 40-43: Rq = 100*[reg(q)==c-40], Rk = [reg(k)==c-40]        -> + 100 iff same shift region
// Softmax is invariant to the per-row constant this adds; a key in another window is lower by BIG=1000 (exp -> 0),
// one in another shift region by exactly the reference's 100.  No index arithmetic, no table look-ups, no LDS
// traffic in the attention phase.
#include <type_traits>

#include "swin_kernels.h"

namespace nunif {

#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define MFMA_16x16x16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16f16((a), (b), (c), 0, 0, 0)

constexpr int kGroupTokens = 144;   // 4 windows x 36 tokens = 9 tiles of 16
constexpr int kTiles = 9;
constexpr float kSameWindow = 1000.0f;
constexpr float kSameRegion = 100.0f;

// key tiles that can share a window with q tile qt (compile-time after unrolling)
__device__ __forceinline__ constexpr int kt_lo(int qt) { return ((16 * qt) / 36 * 36) / 16; }
__device__ __forceinline__ constexpr int kt_hi(int qt) { return (((16 * qt + 15) / 36) * 36 + 35) / 16; }

struct QkvAttnArgs {
    const f16 *x;            // [B,H,W,C]
    f16 *att;                // [B,H,W,C]
    const f16 *wqkv;         // fragment-packed [3C/16][C/32][64][8]
    const float *bqkv;       // [3C]
    const float *bias;       // [heads][36][48] relative-position bias (cols >= 36 unused here)
    int B, H, W, shift, n_windows, n_groups;
    float scale;
};

template <int C>
// (measured: forcing 168 VGPRs for 2 workgroups per CU spills 51 registers and is 1.5x SLOWER than one
//  workgroup per CU at 256 VGPRs — round 1 keeps the latter; the planned fix is 2 waves per head)
__global__ void __launch_bounds__(384)
qkv_attn_hd16_kernel(QkvAttnArgs a) {
    constexpr int HD = 16;
    constexpr int KS = C / 32;
    constexpr int HEADS = C / HD;
    static_assert(HEADS == 6, "one wave per head, 6 waves");
    constexpr int LDX = C + 8;                  // halfs per staged row (+16 B: conflict-free ds_read_b128)
    __shared__ __attribute__((aligned(16))) f16 xs[kGroupTokens * LDX];
    __shared__ int pixtab[kGroupTokens];       // pixel index of each token in the un-rolled map, -1 = dummy
    __shared__ int regtab[kGroupTokens];       // shift region inside the window, 0..3
    // Rk (A-operand one-hot fragments) is the same for every head: shared in LDS instead of 54 VGPRs per wave
    __shared__ __attribute__((aligned(16))) f16x4 rks[kTiles][3][64];
    // relative-position bias of all heads, [head][q_loc][k_loc] fp16: Rq fragments are 8-byte reads of its rows
    __shared__ __attribute__((aligned(16))) f16 bt[HEADS * 36 * 36];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int head = tid >> 6;
    const int r16 = lane & 15;
    const int grp = lane >> 4;
    const int nwx = a.W / 6, nwy = a.H / 6;

    // ---- per-wave constants --------------------------------------------------------------------------------------
    f16x8 wq[KS], wk[KS], wv[KS];
    {
        const f16x8 *wb = reinterpret_cast<const f16x8 *>(a.wqkv) + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            wq[ks] = wb[((head)*KS + ks) * 64];
            wk[ks] = wb[((HEADS + head) * KS + ks) * 64];
            wv[ks] = wb[((2 * HEADS + head) * KS + ks) * 64];
        }
    }
    const float4 bq = *reinterpret_cast<const float4 *>(a.bqkv + head * HD + 4 * grp);
    const float4 bk = *reinterpret_cast<const float4 *>(a.bqkv + C + head * HD + 4 * grp);
    const float bvv = a.bqkv[2 * C + head * HD + r16];

    // Rk (A operand) one-hot fragments, [tile][column block js]; lane holds columns 16js + 4grp + j
    if (head == 0) {
#pragma unroll
        for (int t = 0; t < kTiles; ++t) {
            const int tok = 16 * t + r16;
            const int win = tok / 36, loc = tok - 36 * win;
#pragma unroll
            for (int js = 0; js < 3; ++js) {
                f16x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = 16 * js + 4 * grp + j;
                    v[j] = (f16)((col < 36 ? loc == col : (col < 40 ? win == col - 36 : false)) ? 1.f : 0.f);
                }
                rks[t][js][lane] = v;
            }
        }
    }
    for (int i = tid; i < HEADS * 36 * 36; i += 384) {
        const int hq = i / 36, col = i - hq * 36;
        bt[i] = (f16)a.bias[(long)hq * 48 + col];
    }
    // (the first __syncthreads of the group loop publishes rks / bt)

    for (int group = blockIdx.x; group < a.n_groups; group += gridDim.x) {
        // ---- token tables: roll + window partition folded into addressing ------------------------------------------
        if (tid < kGroupTokens) {
            const int wl = tid / 36, loc = tid - 36 * wl;
            int wi = group * 4 + wl;
            const bool dummy = wi >= a.n_windows;
            if (dummy) wi = a.n_windows - 1;
            const int wx = wi % nwx;
            const int t2 = wi / nwx;
            const int wy = t2 % nwy;
            const int b = t2 / nwy;
            const int iy = loc / 6, ix = loc - 6 * iy;
            int yy = wy * 6 + iy + a.shift, xx = wx * 6 + ix + a.shift;
            if (yy >= a.H) yy -= a.H;
            if (xx >= a.W) xx -= a.W;
            const int pix = (b * a.H + yy) * a.W + xx;
            // region of the rolled position (slices (0,-6),(-6,-3),(-3,None)): inside one window only the split
            // at -3 can separate tokens, and only in the last window row / column
            const int ry = (a.shift > 0 && wy == nwy - 1 && iy >= 3) ? 1 : 0;
            const int rx = (a.shift > 0 && wx == nwx - 1 && ix >= 3) ? 1 : 0;
            pixtab[tid] = dummy ? -1 - pix : pix;      // dummy tokens still load valid memory, never store
            const int reg = ry * 2 + rx;
            regtab[tid] = reg;
            // region one-hot columns 40-43 of Rk live in block js=2, lane group 2 (lanes 32-47) of the token's tile
            f16x4 oh;
#pragma unroll
            for (int j = 0; j < 4; ++j) oh[j] = (f16)((a.shift > 0 && reg == j) ? 1.f : 0.f);
            rks[tid >> 4][2][32 + (tid & 15)] = oh;
        }
        __syncthreads();
        // ---- stage x[144][C] into LDS ------------------------------------------------------------------------------
        for (int i = tid; i < kGroupTokens * (C / 8); i += 384) {
            const int row = i / (C / 8), ch = i - row * (C / 8);
            int pix = pixtab[row];
            if (pix < 0) pix = -1 - pix;
            const f16x8 v = *reinterpret_cast<const f16x8 *>(a.x + (long)pix * C + ch * 8);
            *reinterpret_cast<f16x8 *>(&xs[row * LDX + ch * 8]) = v;
        }
        __syncthreads();

        // ---- q, k, v of this head for the 9 token tiles -------------------------------------------------------------
        f16x4 qf[kTiles], kf[kTiles], vf[kTiles];
#pragma unroll
        for (int mt = 0; mt < kTiles; ++mt) {
            f32x4 aq = {0.f, 0.f, 0.f, 0.f}, ak = aq, av = aq;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const f16x8 xf = *reinterpret_cast<const f16x8 *>(&xs[(16 * mt + r16) * LDX + 32 * ks + 8 * grp]);
                aq = MFMA_16x16x32(wq[ks], xf, aq);
                ak = MFMA_16x16x32(wk[ks], xf, ak);
                av = MFMA_16x16x32(xf, wv[ks], av);      // operands swapped: av[r] = V[token 4grp+r][d = r16]
            }
            qf[mt] = (f16x4){(f16)((aq[0] + bq.x) * a.scale), (f16)((aq[1] + bq.y) * a.scale),
                             (f16)((aq[2] + bq.z) * a.scale), (f16)((aq[3] + bq.w) * a.scale)};
            kf[mt] = (f16x4){(f16)(ak[0] + bk.x), (f16)(ak[1] + bk.y), (f16)(ak[2] + bk.z), (f16)(ak[3] + bk.w)};
            vf[mt] = (f16x4){(f16)(av[0] + bvv), (f16)(av[1] + bvv), (f16)(av[2] + bvv), (f16)(av[3] + bvv)};
            __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from hoisting later tiles' loads (spills)
        }

        // ---- attention, one q tile at a time ---------------------------------------------------------------------------
        // All score / PV products run on the full-rate K=32 MFMA by pairing 16-wide k blocks into 8-slot fragments:
        //   S^T = [K | Rk0][Q | Rq0]^T + [Rk1 | Rk2][Rq1 | Rq2]^T           (2 MFMAs per tile pair)
        //   O^T = sum over PAIRS of key tiles [V^T(kt) | V^T(kt+1)] [P^T(kt) ; P^T(kt+1)]
        auto cat = [](f16x4 lo, f16x4 hi) -> f16x8 {
            return (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        const f16x4 zero4 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
#pragma unroll
        for (int qt = 0; qt < kTiles; ++qt) {
            constexpr int kMaxK = 5;
            const int lo = kt_lo(qt), hi = kt_hi(qt);
            // Rq^T fragments of this q tile: bias rows from LDS, window / region columns from coordinates
            const int tokq = 16 * qt + r16;
            const int winq = tokq / 36, locq = tokq - 36 * winq;
            const f16 *brow = &bt[(head * 36 + locq) * 36];
            const f16x4 rq0 = *reinterpret_cast<const f16x4 *>(brow + 4 * grp);
            const f16x4 rq1 = *reinterpret_cast<const f16x4 *>(brow + 16 + 4 * grp);
            f16x4 rq2 = zero4;
            if (grp == 0) {
                rq2 = *reinterpret_cast<const f16x4 *>(brow + 32);
            } else if (grp == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) rq2[j] = (f16)(winq == j ? kSameWindow : 0.f);
            } else if (grp == 2 && a.shift > 0) {
                const int reg = regtab[tokq];
#pragma unroll
                for (int j = 0; j < 4; ++j) rq2[j] = (f16)(reg == j ? kSameRegion : 0.f);
            }
            const f16x8 bq1 = cat(qf[qt], rq0);
            const f16x8 bq2 = cat(rq1, rq2);
            f32x4 s[kMaxK];
            float mx = -3.0e38f;
#pragma unroll
            for (int i = 0; i < kMaxK; ++i) {
                const int kt = lo + i;
                if (kt > hi) break;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = MFMA_16x16x32(cat(kf[kt], rks[kt][0][lane]), bq1, acc);
                acc = MFMA_16x16x32(cat(rks[kt][1][lane], rks[kt][2][lane]), bq2, acc);
                s[i] = acc;
                mx = fmaxf(mx, fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])));
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f;
            f16x4 pf[kMaxK + 1];
#pragma unroll
            for (int i = 0; i < kMaxK + 1; ++i) pf[i] = zero4;
#pragma unroll
            for (int i = 0; i < kMaxK; ++i) {
                const int kt = lo + i;
                if (kt > hi) break;
                const float p0 = __expf(s[i][0] - mx), p1 = __expf(s[i][1] - mx);
                const float p2 = __expf(s[i][2] - mx), p3 = __expf(s[i][3] - mx);
                sum += (p0 + p1) + (p2 + p3);
                pf[i] = (f16x4){(f16)p0, (f16)p1, (f16)p2, (f16)p3};
            }
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < kMaxK; i += 2) {
                const int kt = lo + i;
                if (kt > hi) break;
                const f16x4 v1 = (kt + 1 <= hi) ? vf[kt + 1 <= 8 ? kt + 1 : 8] : zero4;
                o = MFMA_16x16x32(cat(vf[kt], v1), cat(pf[i], pf[i + 1]), o);
            }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
            const int pix = pixtab[16 * qt + r16];
            if (pix >= 0) {
                const f16x4 ov = {(f16)(o[0] * inv), (f16)(o[1] * inv), (f16)(o[2] * inv), (f16)(o[3] * inv)};
                *reinterpret_cast<f16x4 *>(a.att + (long)pix * C + head * HD + 4 * grp) = ov;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();   // pixtab / xs are rewritten by the next group
    }
}

int launch_qkv_attn(const f16 *x, f16 *att, const f16 *wqkv, const float *bqkv, const float *bias, int B, int H,
                    int W, int C, int heads, int shift, hipStream_t s) {
    NUNIF_REQUIRE(H % 6 == 0 && W % 6 == 0, "qkv_attn: %dx%d not a multiple of the 6x6 window", H, W);
    NUNIF_REQUIRE(C == 96 && heads == 6, "qkv_attn: only C=96 / 6 heads is fused");
    if (H <= 6) shift = 0;
    QkvAttnArgs a;
    a.x = x; a.att = att; a.wqkv = wqkv; a.bqkv = bqkv; a.bias = bias;
    a.B = B; a.H = H; a.W = W; a.shift = shift;
    a.n_windows = B * (H / 6) * (W / 6);
    a.n_groups = (a.n_windows + 3) / 4;
    a.scale = 1.0f / sqrtf((float)(C / heads));
    const double tok = (double)B * H * W;
    ProfScope ps("qkv_attn_hd16_kernel<96>", s, 2.0 * tok * C * 3.0 * C + 4.0 * tok * 36.0 * C, tok * C * 2.0 * 2.0);
    const int grid = a.n_groups < 512 ? a.n_groups : 512;     // persistent: 2 workgroups per CU
    qkv_attn_hd16_kernel<96><<<grid, 384, 0, s>>>(a);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

}  // namespace nunif
