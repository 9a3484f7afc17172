// One kernel per C = 96 swin block on gfx950: qkv Linear + (shifted) 6x6 window attention + attn.proj + residual +
// mlp.0 + GELU + mlp.3 + residual, in place on x.  The pre-projection attention map never exists in memory.
//
// Replaces, per block, torchvision SwinTransformerBlock.forward as called from
// waifu2x/models/swin_unet.py:20-42 (level-1 stages swin1 / swin5 of the 1x / 2x nets, swin1 of the 4x net).
//
// Why (round-2 profile): the two-kernel form (swin_qkv_attn_r.hip + swin_block_tail.hip) is VALU-issue bound,
// not bandwidth bound, but its memory operations still cost a quarter of the attention kernel — 8-byte per-head stores of
// `att` (1.48x write amplification), a second read of x and a read of att in the tail, and one exposed HBM latency per
// window / token group.  Here a wave owns a WINDOW from its x rows to its x' rows:
//   * x is read once (it is the qkv GEMM's B operand AND, through an identity MFMA, the residual) and written once,
//     16 bytes per lane; HBM traffic per token is 2 x 192 B instead of 5 x 192 B (+ partial-line stores);
//   * the attention output of head pair (2s, 2s+1) IS the B fragment of attn.proj's k-step s (accumulator tile pairs =
//     chained k order, swin_unet.cpp pack_a_fragments), so attention -> proj -> mlp.0 -> mlp.3 is one register chain;
//   * all weights are resident in LDS: Wqkv 54 KiB + tail stream 90 KiB (+ 3 KiB ToImage head) = 147 KiB.  The fp32
//     bias table of the two-kernel form (6 x 36 x 52 floats = 45 KiB) does not fit next to them, so the relative position
//     bias is read from the RAW 11 x 11 table: window tokens are enumerated in 2 x 2 blocks (t = 4 b + e, block b = (by, bx)
//     of a 3 x 3 grid, e = (dy, dx)), which makes the four keys of one accumulator lane the offsets {0, 1, 11, 12} from a
//     lane-constant table index — two ds_read2_b32 from one address register per (query tile, key tile);
//   * no padding in the tail: a 36-token window is two full 16-token tiles + 4 tokens; the 4-token remainders of FOUR
//     consecutive windows of a wave are merged lane-wise (DPP row_shr into lanes 4j..4j+3) into one full tile that runs
//     the tail once per four windows.  (A per-window tail on three tiles would pay 33 % more GELU / MFMA work.)
// Token order inside a window is free: attention is permutation-equivariant over the window's tokens and the tail is
// per token; only the bias index, the shift-region id and the pixel address see the enumeration.
#include <algorithm>
#include <cstring>

#include "swin_gelu.h"
#include "swin_kernels.h"

namespace nunif {

#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

namespace {

constexpr int kC = 96, kKS = 3, kNT = 6, kSH = 6, kHeads = 6;
constexpr int kQkvFrags = 54;                                  // per head: Wq, Wk, Wv, each 3 k-steps
constexpr int kTailFrags = 2 * kKS * kKS + kSH * (2 * kKS + kNT);      // 90: proj (chained k) | per 32 hidden: mlp.0 x6, mlp.3 x6
constexpr int kHeadFrags = 3;                                  // ToImage head (chained k), last block of a 1x / 2x net
constexpr int kBtabStride = 144;                               // floats per head: 121 reversed table entries, 124.. = -1000
constexpr int kBtabPad = 124;
constexpr float kRegion = 100.0f;                              // added where query and key share a shift region
constexpr int kWaves = 8;

struct Block96Args {
    f16 *x;                  // [B,H,W,96], updated in place
    const f16 *wqkv;         // 54 fragments, per head Wq | Wk | Wv (q rows pre-multiplied by head_dim^-0.5 * log2 e)
    const f16 *wtail;        // 90 fragments
    const f16 *whead;        // 3 fragments or NULL
    const float *bqkv;       // [288], q part pre-scaled
    const float *btab;       // [6][144]: R[i] = log2e * table[120 - i], floats 124..136 = -1000
    const float *bp, *b0, *b3, *bhead;
    float *img;              // ToImage output (planar fp32), when whead != NULL
    int ps, n_real;
    int B, H, W, shift, n_windows, rev;
};

__device__ __forceinline__ f16x8 cat8(f16x4 lo, f16x4 hi) {
    return (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// window-local token t (0..35) -> (iy, ix) in the 6x6 window: 2x2 blocks, blocks row-major over a 3x3 grid
__device__ __forceinline__ void tok_yx(int t, int &iy, int &ix) {
    const int b = t >> 2, e = t & 3;
    const int by = b / 3, bx = b - 3 * by;
    iy = 2 * by + (e >> 1);
    ix = 2 * bx + (e & 1);
}

// lane-wise merge of lanes 0..3 of every 16-lane row of `src` into lanes 4 SLOT .. 4 SLOT + 3 of `old`
template <int SLOT>
__device__ __forceinline__ unsigned merge_u32(unsigned old, unsigned src) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int ctrl = SLOT == 0 ? 0xE4 : 0x110 + 4 * SLOT;      // quad_perm identity / row_shr:4 SLOT
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, ctrl, 0xf, 1 << SLOT, false);
#else
    return old;
#endif
}
template <int SLOT>
__device__ __forceinline__ f16x8 merge_f16x8(f16x8 old, f16x8 src) {
    const u32x4 o = __builtin_bit_cast(u32x4, old), s = __builtin_bit_cast(u32x4, src);
    return __builtin_bit_cast(f16x8, ((u32x4){merge_u32<SLOT>(o[0], s[0]), merge_u32<SLOT>(o[1], s[1]),
                                              merge_u32<SLOT>(o[2], s[2]), merge_u32<SLOT>(o[3], s[3])}));
}

// ---- the block tail on MF token tiles held in registers -----------------------------------------------------------------
//     y = x + Wp att + bp;  x' = y + W3 gelu(W0 y + b0) + b3
// of: att as B fragments in the CHAINED k order (slots 0-3 = head 2s, slots 4-7 = head 2s+1), xr: x in the PLAIN k order
// (8 consecutive channels per lane).  Same dataflow as proj_mlp_r_kernel (swin_block_tail.hip): bias = MFMA C operand,
// both residuals as identity MFMAs, hidden activation through gelu8, weights read lane-linear from LDS.
// The arrays have ROWS >= MF tiles; tiles 0 .. MF-1 are processed.
template <int MF, int ROWS, bool TI>
__device__ __forceinline__ void tail96(const f16x8 *wt, const f16x8 *wh, const float *bl, const f16x8 (&of)[ROWS][kKS],
                                       const f16x8 (&xr)[ROWS][kKS], const int (&pix)[ROWS], const bool (&valid)[ROWS],
                                       const Block96Args &a, const f16x8 *idl, int grp) {
    // an opaque zero OFFSET keeps LICM from hoisting the (loop-invariant) LDS bias reads out of the window loop into
    // registers; laundering the POINTER (as proj_mlp_r_kernel does) loses the LDS address space and turns every bias read
    // into a flat_load, whose in-order vmcnt wait then sits behind the next window's x prefetch
    int opq = 0;
    asm volatile("" : "+v"(opq));
    const float *lbp = bl + opq, *lb0 = bl + kC + opq, *lb3 = bl + 3 * kC + opq;
    f16x8 yf[MF][kKS];
#pragma unroll
    for (int s = 0; s < kKS; ++s) {
        const int n0 = 32 * s + 4 * grp;
        const f32x4 ba = *reinterpret_cast<const f32x4 *>(lbp + n0);
        const f32x4 bb = *reinterpret_cast<const f32x4 *>(lbp + n0 + 16);
        f32x4 a0[MF], a1[MF];
#pragma unroll
        for (int f = 0; f < MF; ++f) { a0[f] = ba; a1[f] = bb; }
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
            const f16x8 wa = wt[((s * kKS + ks) * 2) * 64];
            const f16x8 wb = wt[((s * kKS + ks) * 2 + 1) * 64];
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                a0[f] = MFMA_16x16x32(wa, of[f][ks], a0[f]);
                a1[f] = MFMA_16x16x32(wb, of[f][ks], a1[f]);
            }
        }
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            a0[f] = MFMA_16x16x32(idl[0], xr[f][s], a0[f]);        // + x (channels 32 s .. 32 s + 15)
            a1[f] = MFMA_16x16x32(idl[64], xr[f][s], a1[f]);       // + x (channels 32 s + 16 .. 32 s + 31)
        }
#pragma unroll
        for (int f = 0; f < MF; ++f)
            yf[f][s] = (f16x8){(f16)a0[f][0], (f16)a0[f][1], (f16)a0[f][2], (f16)a0[f][3],
                               (f16)a1[f][0], (f16)a1[f][1], (f16)a1[f][2], (f16)a1[f][3]};
    }
    f32x4 acc[kNT][MF];
#pragma unroll
    for (int s = 0; s < kKS; ++s) {
        const int n0 = 32 * s + 4 * grp;
        const f32x4 ca = *reinterpret_cast<const f32x4 *>(lb3 + n0);
        const f32x4 cb = *reinterpret_cast<const f32x4 *>(lb3 + n0 + 16);
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            acc[2 * s][f] = MFMA_16x16x32(idl[128], yf[f][s], ca);           // b3 + y
            acc[2 * s + 1][f] = MFMA_16x16x32(idl[192], yf[f][s], cb);
        }
    }
    constexpr int F_MLP = 2 * kKS * kKS;
    constexpr int F_STEP = 2 * kKS + kNT;
#pragma unroll 1
    for (int s = 0; s < kSH; ++s) {
        const int n0 = 32 * s + 4 * grp;
        const f32x4 ba = *reinterpret_cast<const f32x4 *>(lb0 + n0);
        const f32x4 bb = *reinterpret_cast<const f32x4 *>(lb0 + n0 + 16);
        f32x4 h0[MF], h1[MF];
#pragma unroll
        for (int f = 0; f < MF; ++f) { h0[f] = ba; h1[f] = bb; }
        const f16x8 *wf = wt + (F_MLP + s * F_STEP) * 64;
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
            const f16x8 wa = wf[(ks * 2) * 64];
            const f16x8 wb = wf[(ks * 2 + 1) * 64];
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                h0[f] = MFMA_16x16x32(wa, yf[f][ks], h0[f]);
                h1[f] = MFMA_16x16x32(wb, yf[f][ks], h1[f]);
            }
        }
        f16x8 hf[MF];
#pragma unroll
        for (int f = 0; f < MF; ++f) hf[f] = gelu8(h0[f], h1[f]);
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            const f16x8 wv = wf[(2 * kKS + nt) * 64];
#pragma unroll
            for (int f = 0; f < MF; ++f) acc[nt][f] = MFMA_16x16x32(wv, hf[f], acc[nt][f]);
        }
    }
    if constexpr (TI) {
        // fused image head (ToImage: Linear 96 -> 3 ps^2, pixel_shuffle, clamp; swin_unet.py:85-116) on the fp16-rounded x'
        const int s2 = a.ps * a.ps;
        const f32x4 tb = *reinterpret_cast<const f32x4 *>(bl + 4 * kC + 4 * grp);
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            f32x4 img = tb;
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) {
                const f16x8 bfrag = {(f16)acc[2 * ks][f][0], (f16)acc[2 * ks][f][1], (f16)acc[2 * ks][f][2], (f16)acc[2 * ks][f][3],
                                     (f16)acc[2 * ks + 1][f][0], (f16)acc[2 * ks + 1][f][1], (f16)acc[2 * ks + 1][f][2],
                                     (f16)acc[2 * ks + 1][f][3]};
                img = MFMA_16x16x32(wh[ks * 64], bfrag, img);
            }
            if (!valid[f]) continue;
            const int m = pix[f];
            const int px = m % a.W;
            const int t2 = m / a.W;
            const int py = t2 % a.H, pb = t2 / a.H;
            const int OC = a.n_real / s2;
            const long OH = (long)a.H * a.ps, OW = (long)a.W * a.ps;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 4 * grp + r;
                if (n < a.n_real) {
                    const int c = n / s2, rem = n - c * s2, i = rem / a.ps, j = rem - i * a.ps;
                    a.img[(((long)pb * OC + c) * OH + (long)py * a.ps + i) * OW + (long)px * a.ps + j] =
                        fminf(fmaxf(img[r], 0.f), 1.f);
                }
            }
        }
    } else {
#pragma unroll
        for (int p = 0; p < kNT / 2; ++p) {
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                const f16x4 oa = {(f16)acc[2 * p][f][0], (f16)acc[2 * p][f][1], (f16)acc[2 * p][f][2], (f16)acc[2 * p][f][3]};
                const f16x4 ob = {(f16)acc[2 * p + 1][f][0], (f16)acc[2 * p + 1][f][1], (f16)acc[2 * p + 1][f][2],
                                  (f16)acc[2 * p + 1][f][3]};
                const f16x8 o = pair_to_run(oa, ob);           // all lanes take part in the swap; only valid rows store
                if (valid[f]) *reinterpret_cast<f16x8 *>(a.x + (long)pix[f] * kC + 32 * p + pair_run_channel(grp)) = o;
            }
        }
    }
}

template <bool TI>
__global__ void __launch_bounds__(kWaves * 64)
swin_block96_kernel(Block96Args a) {
    constexpr int NTHR = kWaves * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    f16x8 *wq = reinterpret_cast<f16x8 *>(smem_b);                           // [54][64]
    f16x8 *wtl = wq + kQkvFrags * 64;                                        // [90][64]
    f16x8 *whl = wtl + kTailFrags * 64;                                      // [3][64]
    float *bq = reinterpret_cast<float *>(whl + kHeadFrags * 64);            // [288]
    float *tb = bq + 3 * kC;                                                 // bp[96] | b0[192] | b3[96] | head bias[16]
    float *bt = tb + 4 * kC + 16;                                            // [6][144]
    f16x8 *idf = reinterpret_cast<f16x8 *>(bt + kHeads * kBtabStride);       // [4][64]: identity fragments, see below

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r16 = lane & 15;
    const int grp = lane >> 4;
    const int nwx = a.W / 6, nwy = a.H / 6;
    const f16x4 zero4 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    const f16x8 ones8 = {(f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f};

    {
        const f16x8 *s0 = reinterpret_cast<const f16x8 *>(a.wqkv);
        for (int i = tid; i < kQkvFrags * 64; i += NTHR) wq[i] = s0[i];
        const f16x8 *s1 = reinterpret_cast<const f16x8 *>(a.wtail);
        for (int i = tid; i < kTailFrags * 64; i += NTHR) wtl[i] = s1[i];
        if constexpr (TI) {
            const f16x8 *s2 = reinterpret_cast<const f16x8 *>(a.whead);
            for (int i = tid; i < kHeadFrags * 64; i += NTHR) whl[i] = s2[i];
            if (tid < 16) tb[4 * kC + tid] = tid < a.n_real ? a.bhead[tid] : 0.f;
        }
        for (int i = tid; i < 3 * kC; i += NTHR) bq[i] = a.bqkv[i];
        for (int i = tid; i < kC; i += NTHR) { tb[i] = a.bp[i]; tb[3 * kC + i] = a.b3[i]; }
        for (int i = tid; i < 2 * kC; i += NTHR) tb[kC + i] = a.b0[i];
        for (int i = tid; i < kHeads * kBtabStride; i += NTHR) bt[i] = a.btab[i];
        // identity fragments for the two residual adds on the MFMA (see proj_mlp_r_kernel): [0] / [1] = rows of the even /
        // odd tile of a pair against a PLAIN-k-order B fragment (k = 8 grp + j), [2] / [3] against a CHAINED-k-order one.
        // Kept in LDS (4 KiB) rather than in 16 registers: the kernel runs at the 256-register limit.
        if (tid < 64) {
            f16x8 ie, io, je, jo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                ie[j] = (f16)((r16 == 8 * grp + j) ? 1.f : 0.f);
                io[j] = (f16)((r16 + 16 == 8 * grp + j) ? 1.f : 0.f);
                je[j] = (f16)((j < 4 && r16 == 4 * grp + j) ? 1.f : 0.f);
                jo[j] = (f16)((j >= 4 && r16 == 4 * grp + j - 4) ? 1.f : 0.f);
            }
            idf[tid] = ie; idf[64 + tid] = io; idf[128 + tid] = je; idf[192 + tid] = jo;
        }
    }
    __syncthreads();
    const f16x8 *idl = idf + lane;
    const f16x8 *wql = wq + lane, *wtlane = wtl + lane, *whlane = whl + lane;


    // relative-position-bias index of this lane's (query r16 of tile qt, key block 4 kt + grp): float offset into a head's
    // reversed table; the four keys of the block are at +0, +1, +11, +12.  Key blocks >= 9 are padding: -1000.
    int bidx[3][3];
#pragma unroll
    for (int qt = 0; qt < 3; ++qt) {
        int qy, qx;
        tok_yx(min(16 * qt + r16, 35), qy, qx);
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            const int b = 4 * kt + grp;
            const int by = b / 3, bx = b - 3 * by;
            bidx[qt][kt] = b < 9 ? (5 - qy + 2 * by) * 11 + (5 - qx + 2 * bx) : kBtabPad;
        }
    }

    auto pix_of = [&](int wq_, int t) -> int {         // window-local token -> pixel of the un-rolled map
        const int wx = wq_ % nwx, t2 = wq_ / nwx;
        const int wy = t2 % nwy, b = t2 / nwy;
        int iy, ix;
        tok_yx(min(t, 35), iy, ix);
        int yy = wy * 6 + iy + a.shift, xx = wx * 6 + ix + a.shift;
        if (yy >= a.H) yy -= a.H;
        if (xx >= a.W) xx -= a.W;
        return (b * a.H + yy) * a.W + xx;
    };
    auto load_x = [&](int wq_, f16x8 (&xf)[3][kKS], int (&pix)[3]) {
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
            pix[mt] = pix_of(wq_, 16 * mt + r16);
            const f16 *p = a.x + (long)pix[mt] * kC + 8 * grp;
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) xf[mt][ks] = *reinterpret_cast<const f16x8 *>(p + 32 * ks);
        }
    };
    auto wmap = [&](int wi) { return a.rev ? a.n_windows - 1 - wi : wi; };

    const int wstride = kWaves * gridDim.x;
    const int w0 = blockIdx.x * kWaves + wave;

    f16x8 xf[3][kKS];
    int pix[3];
    if (w0 < a.n_windows) load_x(wmap(w0), xf, pix);

    // remainder tile: tokens 32..35 of up to four consecutive windows of this wave, merged lane-wise
    f16x8 ofR[1][kKS], xfR[1][kKS];
    int pixR[1] = {0};
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) { ofR[0][ks] = (f16x8){}; xfR[0][ks] = (f16x8){}; }

    int k = 0;
#pragma unroll 1
    for (int wi = w0; wi < a.n_windows; wi += wstride, ++k) {
        // ---- shift regions of this window (only the last window row / column straddles two regions) ----------------
        f16x4 rkr[3], rqr[3];
        {
            const int wq_ = wmap(wi);
            const int wx = wq_ % nwx, wy = (wq_ / nwx) % nwy;
            const bool last_y = a.shift > 0 && wy == nwy - 1, last_x = a.shift > 0 && wx == nwx - 1;
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                int iy, ix;
                tok_yx(min(16 * mt + r16, 35), iy, ix);
                const int reg = ((last_y && iy >= 3) ? 2 : 0) + ((last_x && ix >= 3) ? 1 : 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool on = a.shift > 0 && grp == 2 && reg == j;       // k-slots 20..23 live in lane group 2
                    rkr[mt][j] = (f16)(on ? 1.f : 0.f);
                    rqr[mt][j] = (f16)(on ? kRegion : 0.f);
                }
            }
        }

        // ---- qkv + attention, head by head; head pair (2s, 2s+1) -> B fragment s of attn.proj ------------------------
        // (a real loop over head pairs: fully unrolled, hipcc moves LDS reads across heads and spills ~80 registers)
        f16x8 of[3][kKS];
#pragma unroll 1
        for (int hp = 0; hp < kKS; ++hp) {
            f16x4 ov[2][3];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int head = 2 * hp + h2;
                const f16x8 *wh = wql + (head * 9) * 64;
                f16x4 qt4[3], kt4[3], vt4[3];
#pragma unroll
                for (int part = 0; part < 3; ++part) {
                    const int ch0 = part * kC + head * 16;
                    f32x4 acc[3];
                    if (part == 2) {
                        const float bv = bq[ch0 + r16];
#pragma unroll
                        for (int mt = 0; mt < 3; ++mt) acc[mt] = (f32x4){bv, bv, bv, bv};
                    } else {
                        const f32x4 bb = *reinterpret_cast<const f32x4 *>(bq + ch0 + 4 * grp);
#pragma unroll
                        for (int mt = 0; mt < 3; ++mt) acc[mt] = bb;
                    }
#pragma unroll
                    for (int ks = 0; ks < kKS; ++ks) {
                        const f16x8 w = wh[(part * kKS + ks) * 64];
#pragma unroll
                        for (int mt = 0; mt < 3; ++mt)
                            acc[mt] = part == 2 ? MFMA_16x16x32(xf[mt][ks], w, acc[mt]) : MFMA_16x16x32(w, xf[mt][ks], acc[mt]);
                    }
#pragma unroll
                    for (int mt = 0; mt < 3; ++mt) {
                        const f16x4 v = {(f16)acc[mt][0], (f16)acc[mt][1], (f16)acc[mt][2], (f16)acc[mt][3]};
                        if (part == 0) qt4[mt] = v; else if (part == 1) kt4[mt] = v; else vt4[mt] = v;
                    }
                }
                const float *bth = bt + head * kBtabStride;
#pragma unroll
                for (int qt = 0; qt < 3; ++qt) {
                    f32x4 s[3];
#pragma unroll
                    for (int kt = 0; kt < 3; ++kt) {
                        const float *bp_ = bth + bidx[qt][kt];
                        f32x4 acc = {bp_[0], bp_[1], bp_[11], bp_[12]};
                        s[kt] = MFMA_16x16x32(cat8(kt4[kt], rkr[kt]), cat8(qt4[qt], rqr[qt]), acc);
                    }
                    float mx = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3])),
                                     fmaxf(fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])),
                                           fmaxf(fmaxf(s[2][0], s[2][1]), fmaxf(s[2][2], s[2][3]))));
                    mx = row_group_max(mx);
                    f16x4 pf[3];
#pragma unroll
                    for (int kt = 0; kt < 3; ++kt) {
                        const float p0 = __builtin_amdgcn_exp2f(s[kt][0] - mx), p1 = __builtin_amdgcn_exp2f(s[kt][1] - mx);
                        const float p2 = __builtin_amdgcn_exp2f(s[kt][2] - mx), p3 = __builtin_amdgcn_exp2f(s[kt][3] - mx);
                        pf[kt] = (f16x4){(f16)p0, (f16)p1, (f16)p2, (f16)p3};
                    }
                    // denominator = sum of the fp16 probabilities the PV product actually uses: ones x P on the MFMA
                    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                    const f32x4 sm = MFMA_16x16x32(ones8, cat8(pf[0] + pf[1], pf[2]), z4);
                    const float inv = __builtin_amdgcn_rcpf(sm[0]);
                    f32x4 o = MFMA_16x16x32(cat8(vt4[0], vt4[1]), cat8(pf[0], pf[1]), z4);
                    o = MFMA_16x16x32(cat8(vt4[2], zero4), cat8(pf[2], zero4), o);
                    ov[h2][qt] = (f16x4){(f16)(o[0] * inv), (f16)(o[1] * inv), (f16)(o[2] * inv), (f16)(o[3] * inv)};
                }
            }
            switch (hp) {                       // hp is wave-uniform; constant indices keep `of` in registers
            case 0:
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) of[mt][0] = cat8(ov[0][mt], ov[1][mt]);
                break;
            case 1:
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) of[mt][1] = cat8(ov[0][mt], ov[1][mt]);
                break;
            default:
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) of[mt][2] = cat8(ov[0][mt], ov[1][mt]);
                break;
            }
        }

        // ---- x of the next window is requested before the tail starts (different pixels: no hazard with the stores) ---
        const bool has_next = wi + wstride < a.n_windows;
        f16x8 xn[3][kKS];
        int pixn[3];
        load_x(wmap(has_next ? wi + wstride : wi), xn, pixn);

        // ---- remainder tile: tokens 32..35 (lanes r16 < 4 of tile 2) -> lanes 4 j .. 4 j + 3 of R ---------------------
        const int slot = k & 3;
        switch (slot) {
        case 0:
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) { ofR[0][ks] = merge_f16x8<0>(ofR[0][ks], of[2][ks]); xfR[0][ks] = merge_f16x8<0>(xfR[0][ks], xf[2][ks]); }
            pixR[0] = (int)merge_u32<0>((unsigned)pixR[0], (unsigned)pix[2]);
            break;
        case 1:
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) { ofR[0][ks] = merge_f16x8<1>(ofR[0][ks], of[2][ks]); xfR[0][ks] = merge_f16x8<1>(xfR[0][ks], xf[2][ks]); }
            pixR[0] = (int)merge_u32<1>((unsigned)pixR[0], (unsigned)pix[2]);
            break;
        case 2:
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) { ofR[0][ks] = merge_f16x8<2>(ofR[0][ks], of[2][ks]); xfR[0][ks] = merge_f16x8<2>(xfR[0][ks], xf[2][ks]); }
            pixR[0] = (int)merge_u32<2>((unsigned)pixR[0], (unsigned)pix[2]);
            break;
        default:
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) { ofR[0][ks] = merge_f16x8<3>(ofR[0][ks], of[2][ks]); xfR[0][ks] = merge_f16x8<3>(xfR[0][ks], xf[2][ks]); }
            pixR[0] = (int)merge_u32<3>((unsigned)pixR[0], (unsigned)pix[2]);
            break;
        }

        // ---- tail on the two full tiles of this window ----------------------------------------------------------------
        {
            const bool valid3[3] = {true, true, false};
            tail96<2, 3, TI>(wtlane, whlane, tb, of, xf, pix, valid3, a, idl, grp);
        }
        // ---- ... and on the merged remainder tile once it is full (or the wave runs out of windows) ------------------
        if (slot == 3 || !has_next) {
            const bool validR[1] = {(r16 >> 2) <= slot};
            tail96<1, 1, TI>(wtlane, whlane, tb, ofR, xfR, pixR, validR, a, idl, grp);
        }

#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
            pix[mt] = pixn[mt];
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) xf[mt][ks] = xn[mt][ks];
        }
    }
}

}  // namespace

int swin_block96_tail_frags() { return kTailFrags; }
int swin_block96_btab_floats() { return kHeads * kBtabStride; }

// btab: swin_block96_btab_floats() floats, wtail: swin_block96_tail_frags() fragments (make_stage, swin_unet.cpp)
int launch_swin_block96(f16 *x, const f16 *wqkv, const float *bqkv, const float *btab, const f16 *wtail, const float *bp,
                        const float *b0, const float *b3, int B, int H, int W, int shift, hipStream_t s,
                        const TailToImage *ti, int rev) {
    NUNIF_REQUIRE(H % 6 == 0 && W % 6 == 0, "swin_block96: %dx%d not a multiple of the 6x6 window", H, W);
    NUNIF_REQUIRE((long)B * H * W < (1L << 31), "swin_block96: map too large for 32-bit pixel indices");
    NUNIF_REQUIRE(!ti || (ti->n_real <= 16 && ti->H == H && ti->W == W), "swin_block96: image head geometry");
    if (H <= 6) shift = 0;                 // torchvision disables the shift when the window covers the map
    Block96Args a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.wqkv = wqkv; a.wtail = wtail; a.bqkv = bqkv; a.btab = btab; a.bp = bp; a.b0 = b0; a.b3 = b3;
    a.B = B; a.H = H; a.W = W; a.shift = shift; a.n_windows = B * (H / 6) * (W / 6); a.rev = rev;
    if (ti) { a.whead = ti->w; a.bhead = ti->bias; a.img = ti->out; a.ps = ti->ps; a.n_real = ti->n_real; }
    constexpr size_t smem = (size_t)(kQkvFrags + kTailFrags + kHeadFrags + 4) * 1024 +
                            (3 * kC + 4 * kC + 16 + kHeads * kBtabStride) * sizeof(float);
    static_assert(smem <= 160 * 1024, "swin_block96: LDS budget");
    static bool configured = false;
    if (!configured) {
        NUNIF_HIP_CHECK(hipFuncSetAttribute((const void *)swin_block96_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        NUNIF_HIP_CHECK(hipFuncSetAttribute((const void *)swin_block96_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        configured = true;
    }
    const double tok = (double)B * H * W;
    // algorithmic work: qkv + attention (as qkv_attn_r_kernel<96,16>) + tail (as proj_mlp_r_kernel<96>); bytes: read x, write x
    ProfScope ps("swin_block96_kernel", s, 2.0 * tok * kC * 3.0 * kC + 4.0 * tok * 36.0 * kC + 2.0 * tok * kC * kC * 5.0,
                 tok * kC * 2.0 * 2.0);
    const int wgs = (a.n_windows + kWaves - 1) / kWaves;
    const int grid = wgs < 256 ? wgs : 256;                   // persistent: one 8-wave workgroup per CU
    if (ti) swin_block96_kernel<true><<<grid, kWaves * 64, smem, s>>>(a);
    else swin_block96_kernel<false><<<grid, kWaves * 64, smem, s>>>(a);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

}  // namespace nunif
