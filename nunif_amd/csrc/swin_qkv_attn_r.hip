// Fused qkv projection + (shifted) 6x6 window attention with the qkv weights RESIDENT in LDS, for gfx950.
//
// Replaces torchvision shifted_window_attention steps 2-7 (roll, window partition, qkv Linear,
// q*scale, QK^T + relative-position bias + shift mask, softmax, PV), called from waifu2x/models/swin_unet.py:26-36.
//
// One window per wave, everything after the GEMM in registers, bias / padding / shift-region masks folded into the score
// MFMA.  (History, DESIGN.md §6: a round-1 version streamed the weights through an LDS ring with one workgroup barrier per
// 8 fragments, which kept the two waves of a SIMD in lock step — both in their MFMA phase or both in their softmax
// (VALU) phase — and 40 % of the wave cycles were waits.)  160 KB of LDS per CU holds the whole packed Wqkv of C = 96
// (54 KiB) and half of the heads of C = 192 (108 KiB), so a persistent workgroup copies the weights once (per pass of
// HPP heads), and the window loop has NO barrier: waves drift apart and one wave's exp/convert work overlaps the
// other's MFMAs.
//   * q weights / bias are pre-multiplied by head_dim^-0.5 * log2(e) on the host and the bias table by log2(e):
//     softmax = exp2(s - max) with a bare v_exp_f32, no multiply;
//   * the qkv biases initialise the accumulators (no epilogue add);
//   * the window-invariant one-hot key fragments are built once per wave;
//   * the next window's x is requested at the end of a window's trip (four waves per SIMD hide the rest).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "swin_kernels.h"

namespace nunif {

#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

// LDS operands requested one step ahead (round 6): 1 = weight fragments, 2 = qkv biases, 4 = score bias rows
#ifndef NUNIF_QKV_PF
#define NUNIF_QKV_PF 1
#endif
#ifndef NUNIF_QKV_PF192
#define NUNIF_QKV_PF192 0
#endif
// Round 6: compile-time switches of this kernel, each measured as a same-box A/B of build variants (tools/build_variant.py,
// tools/ab_multi.sh; DESIGN.md 6.000000, profiles/r06*_attn_*).  NUNIF_QKV_DIET is a bit mask; the default is what measured fastest.
//   ON   1  window-major stores as `SGPR base + 32-bit lane constant` (no 64-bit address arithmetic per tile)
//   ON   2  `o * inv` and its fp16 convert as one v_fma_mixlo / mixhi_f16 per value (one rounding instead of two)
//   ON   4  (the lone probability of key tile 2 built from an opaque zero: hipcc still emits v_cvt_f16_f32 + v_pack_b32_f16 — no effect)
//   ON   8  C = 96: the six heads unrolled behind sched_barrier(0): every LDS address an immediate (13 address adds per head gone)
//   ON  32  C = 96 window-major: the next window's x requested when the last head's q / k / v exist          338.5 -> 333.4 us
//   off 16  K = 16 score MFMAs (v_mfma_f32_16x16x16_f16) on a path of their own for the windows without shift regions   344.9 vs 342.1
//   off 64  that split with K = 32 MFMAs on both paths (no region registers on the common path)                  24 spilled registers
//   off 128 two passes over a wave's windows (plain, then the last row / column): no scratch, 353.0 vs 341.7 (a tail of 0-3 windows)
//   off NUNIF_QKV_BTAB_FRAG (swin_kernels.h): bias tables / qkv biases as lane-linear fragments: no bank conflicts, 353.3 vs 342.7
//   off NUNIF_QKV_UNROLL192 1 / 2: C = 192 heads unrolled / + early x: 137.5 / 140.7 vs 138.6
// With 1 + 2 + 8 + 32 the window-major C = 96 instance sits at exactly 128 registers and hipcc parks ~15 per-lane constants of the
// last-row / last-column path in scratch (prologue + that path; +7.4 % FETCH_SIZE); tests/test_isa_invariants.py pins that no scratch
// access sits inside the unrolled heads.
#ifndef NUNIF_QKV_UNROLL192
#define NUNIF_QKV_UNROLL192 0
#endif
#ifndef NUNIF_QKV_DIET
#define NUNIF_QKV_DIET 47
#endif

// the "real key" column 36 of the bias table carries 1000: padded keys end up 1000 (log2 units) below every real one
constexpr float kRegionR = 100.0f;     // added where query and key share a shift region

struct QkvAttnRArgs {
    const f16 *x;            // [B,H,W,C]
    f16 *att;                // [B,H,W,C]
    const f16 *wres;         // per head: Wq tiles, Wk tiles, Wv tiles, each (nt, ks) fragment-major; q pre-scaled
    const float *bqkv;       // [3C], q part pre-scaled
    const float *btab32;     // [heads][36][52] fp32: log2e * bias, key columns in win_token order (col 32 + 4 g = key 32 + g), 48..51 unused
    int B, H, W, shift, n_windows;
    int rev;                 // 1: walk the windows from the last to the first (snake order, see launch_qkv_attn_r)
};

__device__ __forceinline__ f16x8 cat8r(f16x4 lo, f16x4 hi) {
    return (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// o * inv -> fp16, the product rounded ONCE: (f16) fma(o, inv, 0) is v_fma_mixlo / mixhi_f16 (fp32 fma, converted on the way out) —
// one instruction per value instead of v_mul_f32 + half a v_cvt_pk_f16_f32.  (Written as C so that hipcc counts the wait states
// between the MFMA that produces `o` and this read: as inline asm it does not, and the first version read half-written accumulators.)
__device__ __forceinline__ f16x4 mul4_to_f16(const f32x4 &o, float inv) {
    if constexpr ((NUNIF_QKV_DIET & 2) != 0) {
        return (f16x4){(f16)__builtin_fmaf(o[0], inv, 0.0f), (f16)__builtin_fmaf(o[1], inv, 0.0f), (f16)__builtin_fmaf(o[2], inv, 0.0f),
                       (f16)__builtin_fmaf(o[3], inv, 0.0f)};
    } else {
        return (f16x4){(f16)(o[0] * inv), (f16)(o[1] * inv), (f16)(o[2] * inv), (f16)(o[3] * inv)};
    }
}
// {(f16) e, 0, 0, 0}: with a zero hipcc cannot see through (`zf`, an SGPR) the pair is ONE v_cvt_pk_f16_f32 instead of v_cvt_f16_f32 +
// v_pack_b32_f16
__device__ __forceinline__ f16x4 lone_to_f16x4(float e, float zf) {
    if constexpr ((NUNIF_QKV_DIET & 4) != 0) {
        return (f16x4){(f16)e, (f16)zf, (f16)0.f, (f16)0.f};
    } else {
        return (f16x4){(f16)e, (f16)0.f, (f16)0.f, (f16)0.f};
    }
}

constexpr int kBiasStride = 52;        // fp32 row stride of the CBIAS table: 16 lanes x 16 B land in 16 distinct bank quads


// Column r16 of token tile mt holds window token win_token(mt, r16): tiles 0 / 1 the tokens 0..31 in order, tile 2 the
// tokens 32..35 at columns 0, 4, 8, 12 (every other column of tile 2 repeats token 35 and is never stored).  The host packs
// the bias table's key columns in the same order (swin_unet.cpp: column 32 + 4 g = key 32 + g).
__device__ __host__ inline int win_token(int mt, int r16) { return mt < 2 ? 16 * mt + r16 : ((r16 & 3) == 0 ? 32 + (r16 >> 2) : 35); }

// The score accumulators are INITIALISED from an fp32 bias table in LDS (relative-position bias and
// the padded-key mask are the MFMA C operand), the softmax denominator comes from one MFMA against a ones fragment, and
// the shift-region term rides in the unused half of the K = 32 step (head_dim 16) or in one extra MFMA that only the
// windows of the last row / column issue (head_dim 32).  Per head this removes 9 (hd 16) / 18 (hd 32) one-hot MFMAs and
// ~45 VALU instructions.
//
// Round 4 — the instruction diet (the kernel is VALU-issue bound: 245 issue slots per head + 360 per window of address
// arithmetic against 45 MFMAs per head, profiles/r04_isa_counts.txt):
//   * the 36 tokens of a window sit in the three 16-column tiles as 16 + 16 + 4, and the 4 sit at COLUMNS 0, 4, 8, 12 of
//     tile 2 (win_token).  As KEYS they are then row 4 g of the score tile = accumulator register 0 of every lane
//     group g, and registers 1-3 of key tile 2 are padding in ALL lanes: their 3 subtractions, 3 exponentials and half of
//     the converts are not issued at all (the contiguous placement kept 4 real + 12 padded keys in one lane group, so
//     every VALU instruction still had to run).  25 % of the softmax;
//   * the window index is wave-uniform: (batch, window row, window column) live in SGPRs and advance by the decomposition
//     of the stride (add + carry), a lane's pixel offset inside a window is a constant (two divisions per lane per
//     LAUNCH instead of ~360 VALU instructions of index arithmetic per window), loads / stores are `base SGPR + 32-bit
//     lane offset`; only the windows of the last row / column of a shifted map (which wrap around) compute offsets;
//   * the region one-hot fragments are constants outside those windows.
template <int C, int HD, int HPP, int WAVES = 8, bool WM = false>
__global__ void __launch_bounds__(WAVES * 64)
qkv_attn_r_kernel(QkvAttnRArgs a) {
    constexpr int kWavesR = WAVES;
    constexpr int NTHR = WAVES * 64;
    constexpr int KS = C / 32;
    constexpr int HEADS = C / HD;
    constexpr int NTH = HD / 16;                      // 16-row weight tiles per head for each of q, k, v
    constexpr int FPH = 3 * NTH * KS;                 // weight fragments (KiB) per head
    constexpr int PASSES = HEADS / HPP;
    static_assert(HEADS == 6 && HEADS % HPP == 0, "swin_unet uses 6 heads at every level");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_r[];
    f16x8 *wl = reinterpret_cast<f16x8 *>(smem_r);                                   // [HPP*FPH][64]
    float *bt32 = reinterpret_cast<float *>(wl + HPP * FPH * 64);                    // [HPP][36][52]
    // qkv biases: kQkvBiasFragMajor: the accumulators' initial values as FRAGMENTS [HPP][3 NTH][64 lanes] f32x4 (q / k tiles: four
    // channels per lane group, v tiles — operands swapped — the column's channel in all four registers): one lane-linear read like
    // every other LDS operand of this kernel, whose single address register is lane * 16; otherwise [3C] floats
    float *bl = bt32 + HPP * kQkvBiasFloatsPerHead;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15;
    const int grp = lane >> 4;
    const int nwx = a.W / 6, nwy = a.H / 6;
    const f16x4 zero4 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    float zf = 0.f;
    asm("" : "+s"(zf));                     // a zero the compiler cannot fold (lone_to_f16x4)

    if constexpr (!kQkvBiasFragMajor) {
        for (int i = tid; i < 3 * C; i += NTHR) bl[i] = a.bqkv[i];
    }
    const f16x8 ones8 = {(f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f};

    // ---- per-lane constants of the launch: the window token of column r16 of tile mt, its pixel offset ----------------------
    int tokc[3], iyc[3], ixc[3];
    unsigned xoff[3];                       // byte offset of (token, channel 8 grp) from the window's first pixel, no wrap-around
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
        tokc[mt] = win_token(mt, r16);
        iyc[mt] = tokc[mt] / 6;
        ixc[mt] = tokc[mt] - 6 * iyc[mt];
        xoff[mt] = (unsigned)(((iyc[mt] * a.W + ixc[mt]) * C + 8 * grp) * 2);
    }
    unsigned wm_off[3];                     // window-major store offset of (token, channels 4 grp ..) inside a head's 36 x HD block
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) wm_off[mt] = (unsigned)((tokc[mt] * HD + 4 * grp) * 2);
    const bool store2 = (r16 & 3) == 0;                                              // tile 2: columns 0, 4, 8, 12 are real
    // store offset of a lane relative to its load offset (non-WM): channel run of the 16-byte epilogue instead of 8 grp
    const int st_delta = (NTH == 2 ? pair_run_channel(grp) : 4 * grp) * 2 - 16 * grp;
    // shift-region one-hots of a window that does not touch the last row / column: every token is in region 0
    f16x4 rk0 = zero4, rq0 = zero4;
    if (a.shift > 0 && grp == 2) { rk0[0] = (f16)1.f; rq0[0] = (f16)kRegionR; }

    // Head passes are spread over WORKGROUPS, not run back to back inside one: workgroup i serves pass i % PASSES for the
    // windows (i / PASSES) + k * gridDim.x / PASSES.  Every workgroup loads its weight slice exactly once, there is no
    // mid-kernel barrier, and the unit of work is (window, pass): with 4 500 windows on the 60 x 60 level a wave gets
    // 4.4 units instead of 2.2 windows x 2 passes, i.e. the last-round quantisation loss drops from 36 % to 12 %.
    const int n_wg = gridDim.x / PASSES;
    const int wstride = kWavesR * n_wg;
    // With only a few units per wave (4.4 on the 60 x 60 level) the waves that get one more must not all sit in the same
    // workgroups: numbering the slots wave-major gives every workgroup the same mix, so a SIMD (waves w and w + 4) runs 5 + 4
    // units instead of 5 + 5 next to CUs with 4 + 4.  Long launches keep the workgroup-major numbering (16 neighbouring
    // windows = one contiguous stretch of the window-major att map).
    const bool spread = a.n_windows < 8 * wstride;
    const int w0 = spread ? wave * n_wg + (int)(blockIdx.x / PASSES) : (int)(blockIdx.x / PASSES) * kWavesR + wave;
    const int pass = blockIdx.x % PASSES;
    {
        const f16x8 *src = reinterpret_cast<const f16x8 *>(a.wres) + (long)pass * HPP * FPH * 64;
        for (int i = tid; i < HPP * FPH * 64; i += NTHR) wl[i] = src[i];
        if constexpr (kQkvBiasFragMajor) {
            for (int i = tid; i < HPP * 3 * NTH * 64; i += NTHR) {
                const int ln = i & 63, t = i >> 6, hl = t / (3 * NTH), pn = t - hl * (3 * NTH), part = pn / NTH, nt = pn - part * NTH;
                const int ch0 = part * C + (pass * HPP + hl) * HD + nt * 16;
                f32x4 v;
                if (part == 2) { const float bv = a.bqkv[ch0 + (ln & 15)]; v = (f32x4){bv, bv, bv, bv}; }
                else v = *reinterpret_cast<const f32x4 *>(a.bqkv + ch0 + 4 * (ln >> 4));
                reinterpret_cast<f32x4 *>(bl)[i] = v;
            }
        }
        const f32x4 *bsrc = reinterpret_cast<const f32x4 *>(a.btab32 + (long)pass * HPP * kQkvBiasFloatsPerHead);
        for (int i = tid; i < HPP * kQkvBiasFloatsPerHead / 4; i += NTHR) reinterpret_cast<f32x4 *>(bt32)[i] = bsrc[i];
    }
    __syncthreads();

    // (cb, cy, cx): batch / window row / window column of window wi — all wave-uniform; (db, dy, dx): of the stride
    int cx = w0 % nwx, cy = (w0 / nwx) % nwy, cb = w0 / (nwx * nwy);
    const int dx = wstride % nwx, dy = (wstride / nwx) % nwy, db = wstride / (nwx * nwy);
    const long img_bytes = (long)a.H * a.W * C * 2;

    // x of window (wb, wy, wx) -> B fragments; vo[mt]: byte offset of this lane's 16 bytes from the image of batch wb
    auto load_x = [&](int wb, int wy, int wx, bool special, f16x8 (&xf)[3][KS], unsigned (&vo)[3]) {
        const int y0 = wy * 6 + a.shift, x0 = wx * 6 + a.shift;
        const char *xb = reinterpret_cast<const char *>(a.x) + (long)wb * img_bytes;
        if (special) {
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                // (token coordinates recomputed here, in the rare path, instead of six more per-lane constants live across the window loop)
                const int tk = win_token(mt, r16), iy = tk / 6, ix = tk - 6 * iy;
                int yy = y0 + iy, xx = x0 + ix;
                if (yy >= a.H) yy -= a.H;
                if (xx >= a.W) xx -= a.W;
                vo[mt] = (unsigned)(((yy * a.W + xx) * C + 8 * grp) * 2);
            }
        } else {
            const unsigned org = (unsigned)((y0 * a.W + x0) * C * 2);
            if constexpr ((NUNIF_QKV_DIET & 32) != 0) {
                // the lane's offsets rebuilt from the lane id (an opaque copy: hoisted out of the window loop they are three more
                // registers carried — and spilled — across it): ~15 VALU per window
                int rl = r16;
                asm("" : "+v"(rl));
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) {
                    const int tk = win_token(mt, rl), iy = (tk * 43) >> 8, ix = tk - 6 * iy;        // tk / 6 for tk < 36
                    vo[mt] = org + (unsigned)(((iy * a.W + ix) * C + 8 * grp) * 2);
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) vo[mt] = org + xoff[mt];
            }
        }
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xf[mt][ks] = *reinterpret_cast<const f16x8 *>(xb + vo[mt] + 64 * ks);
        }
    };
    auto decode = [&](int &wb, int &wy, int &wx, bool &special) {          // snake order: digit-wise complement of the index
        wb = a.rev ? a.B - 1 - cb : cb;
        wy = a.rev ? nwy - 1 - cy : cy;
        wx = a.rev ? nwx - 1 - cx : cx;
        special = a.shift > 0 && (wy == nwy - 1 || wx == nwx - 1);
    };
    auto advance = [&]() {
        cx += dx;
        if (cx >= nwx) { cx -= nwx; ++cy; }
        cy += dy;
        if (cy >= nwy) { cy -= nwy; ++cb; }
        cb += db;
    };

    // initial accumulator of output tile (part, nt) of local head hl (global head `head`)
    auto qkv_bias = [&](int hl, int head, int part, int nt) -> f32x4 {
        if constexpr (kQkvBiasFragMajor) {
            return reinterpret_cast<const f32x4 *>(bl)[(hl * 3 * NTH + part * NTH + nt) * 64 + lane];
        } else {
            const int ch0 = part * C + head * HD + nt * 16;
            if (part == 2) { const float bv = bl[ch0 + r16]; return (f32x4){bv, bv, bv, bv}; }
            return *reinterpret_cast<const f32x4 *>(bl + ch0 + 4 * grp);
        }
    };
    // C operand of the score tile (query tile qt, key tile kt) of local head hl
    auto bias_frag = [&](int hl, int qt, int kt) -> f32x4 {
        if constexpr (kQkvBiasFragMajor)
            return reinterpret_cast<const f32x4 *>(bt32)[((hl * 3 + qt) * 3 + kt) * 64 + lane];
        else
            return *reinterpret_cast<const f32x4 *>(bt32 + (hl * 36 + tokc[qt]) * kBiasStride + 4 * grp + 16 * kt);
    };
    f16x8 xf[3][KS];
    unsigned vo[3], von[3] = {0u, 0u, 0u};      // von: the next window's offsets while this window's pixel-major stores still read vo
    int wb, wy, wx;
    bool special;
    f16x8 wnx = wl[lane];                  // first weight fragment of the next head (see the head loop)
    // DIET & 128 (C = 96, window-major): TWO passes over the wave's windows — first the windows that need no shift-region term (all but
    // the last row / column of a shifted map) through the unrolled heads with no region registers and the early x request, then the
    // others through the rolled form.  One loop over both kinds made hipcc allocate for their union: 15 per-lane constants in
    // scratch, reloaded per window (+7.4 % FETCH_SIZE, profiles/r06n_attn_fetch_by_variant.txt).  MODE 0: one pass over all windows.
    constexpr bool kTwoLoops = (NUNIF_QKV_DIET & 128) && C == 96 && WM && HD == 16;
    const int cx0 = cx, cy0 = cy, cb0 = cb;
    auto run_windows = [&](auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;            // 0: every window, 1: the plain ones, 2: the ones with two shift regions
    int wi = w0;
    cx = cx0; cy = cy0; cb = cb0;
    decode(wb, wy, wx, special);
    // step(): on to the next window of this pass
    auto skip = [&]() {
        if constexpr (MODE != 0) {
            while (wi < a.n_windows && special != (MODE == 2)) { advance(); wi += wstride; decode(wb, wy, wx, special); }
        }
    };
    auto step = [&]() { advance(); wi += wstride; decode(wb, wy, wx, special); skip(); };
    skip();
    if (wi < a.n_windows) load_x(wb, wy, wx, special, xf, vo);
#pragma unroll 1
    while (wi < a.n_windows) {
        const int wq = (wb * nwy + wy) * nwx + wx;                         // index of this window in the window-major att map
        char *ab = reinterpret_cast<char *>(a.att) + (WM ? (long)wq * (HEADS * 36 * HD * 2) : (long)wb * img_bytes);
        // window-major map: a wave-uniform byte offset (the launcher keeps the map below 4 GB) + a lane constant per tile
        const unsigned wm_base = (unsigned)__builtin_amdgcn_readfirstlane(wq * (HEADS * 36 * HD * 2));

        // C = 96: the six heads unrolled (every LDS address an immediate); C = 192 does not fit that way (143 spilled registers).
        // DIET & 16 (head_dim 16): the windows that need no shift-region term — all but the last row / column of a shifted map — take
        // a path of their own (SP = false) with K = 16 score MFMAs; the others the K = 32 form with the region one-hots, rolled.
        // DIET & 64: the same split with K = 32 score MFMAs on both paths — the common path then carries NO region registers (the term is
        // a constant over every row there and drops out of the softmax; rkr / rqr were 12 registers live across all six heads, and at
        // 128 registers the early x request of DIET & 32 had pushed 15 per-lane constants into scratch)
        constexpr bool kSplitK16 = (NUNIF_QKV_DIET & 16) && HD == 16;
        constexpr bool kSplit = (((NUNIF_QKV_DIET & 16) || (NUNIF_QKV_DIET & 64)) && HD == 16) || MODE != 0;
        // (C = 192, pixel-major stores: they read vo, so the early request goes through a second offset set there)
        constexpr bool kEarlyX = (NUNIF_QKV_DIET & 32) && (NUNIF_QKV_DIET & 8) && ((C == 96 && WM) || (C == 192 && NUNIF_QKV_UNROLL192 == 2)) && !kSplitK16 && MODE != 2;
        auto run_heads = [&](auto sp_tag) {
        constexpr bool SP = decltype(sp_tag)::value;
        const bool sp_now = special;            // (kEarlyX decodes the NEXT window inside the last head)
        // shift regions of this window (only the last window row / column straddles two regions)
        f16x4 rkr[3], rqr[3];
        if (special && (SP || !kSplit)) {
            const bool last_y = wy == nwy - 1, last_x = wx == nwx - 1;
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                const int tk = win_token(mt, r16), iy = tk / 6, ix = tk - 6 * iy;
                const int reg = ((last_y && iy >= 3) ? 2 : 0) + ((last_x && ix >= 3) ? 1 : 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool on = grp == 2 && reg == j;                  // cols 40..43 live in lane group 2
                    rkr[mt][j] = (f16)(on ? 1.f : 0.f);
                    rqr[mt][j] = (f16)(on ? kRegionR : 0.f);
                }
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) { rkr[mt] = rk0; rqr[mt] = rq0; }
        }
        constexpr int kHeadUnroll = ((NUNIF_QKV_DIET & 8) && (C == 96 || NUNIF_QKV_UNROLL192) && !(kSplit && SP)) ? HPP : 1;
#pragma unroll kHeadUnroll
        for (int hl = 0; hl < HPP; ++hl) {
            const int head = pass * HPP + hl;
            const f16x8 *wh = wl + (hl * FPH) * 64 + lane;
            // ---- q, k (channels x tokens) and v (tokens x channels: operands swapped) of this head ----------------
            // Round 6: every LDS operand is requested one step before the MFMAs that consume it (weight fragment f + 1 and the
            // next part's bias behind the MFMAs of fragment f; the first fragment of the NEXT head — carried in wnx across the
            // head and the window loop — behind the last): the round-5 form issued each ds_read directly in front of its
            // s_waitcnt lgkmcnt(0), 18 exposed LDS round trips per head and wave (31 % of the wave cycles in s_waitcnt,
            // profiles/r05b_sq.txt SQ_WAIT_ANY).
            constexpr int PF = C == 96 ? NUNIF_QKV_PF : NUNIF_QKV_PF192;
            f16x4 qt4[NTH][3], kt4[NTH][3], vt4[NTH][3];
            f16x8 wq2[2];
            wq2[0] = (PF & 1) ? wnx : wh[0];
            f32x4 bnx = qkv_bias(hl, head, 0, 0);
#pragma unroll
            for (int part = 0; part < 3; ++part) {
#pragma unroll
                for (int nt = 0; nt < NTH; ++nt) {
                    const int ch0 = part * C + head * HD + nt * 16;
                    const int fidx = (part * NTH + nt) * KS;
                    f32x4 acc[3];
                    if constexpr (!(PF & 2)) bnx = qkv_bias(hl, head, part, nt);
#pragma unroll
                    for (int mt = 0; mt < 3; ++mt) acc[mt] = bnx;
                    // bias of the next output tile: parts 0 / 1 four channels per lane group, part 2 (operands swapped) one per column
                    if constexpr ((PF & 2) != 0) {
                        const int nf = part * NTH + nt + 1;
                        if (nf < 3 * NTH) {
                            bnx = qkv_bias(hl, head, nf / NTH, nf % NTH);
                        }
                    }
                    (void)ch0;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const int f = fidx + ks;
                        // the fragment after the head's last one is the next head's first (the next window's first after the last head)
                        if constexpr ((PF & 1) != 0)
                            wq2[(f + 1) & 1] = (f + 1 < FPH) ? wh[(f + 1) * 64] : (hl + 1 < HPP ? wh[FPH * 64] : wl[lane]);
                        else wq2[f & 1] = wh[f * 64];
                        const f16x8 w = wq2[f & 1];
#pragma unroll
                        for (int mt = 0; mt < 3; ++mt)
                            acc[mt] = part == 2 ? MFMA_16x16x32(xf[mt][ks], w, acc[mt]) : MFMA_16x16x32(w, xf[mt][ks], acc[mt]);
                    }
#pragma unroll
                    for (int mt = 0; mt < 3; ++mt) {
                        const f16x4 v = {(f16)acc[mt][0], (f16)acc[mt][1], (f16)acc[mt][2], (f16)acc[mt][3]};
                        if (part == 0) qt4[nt][mt] = v; else if (part == 1) kt4[nt][mt] = v; else vt4[nt][mt] = v;
                    }
                }
            }
            if constexpr ((PF & 1) != 0) wnx = wq2[FPH & 1];
            if constexpr (kEarlyX) {
                // DIET & 32: x is dead once the LAST head's q / k / v exist — the next window's x is requested here, one softmax phase
                // (~1 000 issue cycles x 4 waves) ahead of the top of the next trip, where it used to be requested AND awaited.
                // (Needs the unrolled head loop: `hl` is a constant here.  The window-major stores do not read vo.)
                if (hl == HPP - 1) {
                    step();
                    if (wi < a.n_windows) {
                        if constexpr (WM) load_x(wb, wy, wx, special, xf, vo);
                        else load_x(wb, wy, wx, special, xf, von);
                    }
                }
            }
            f32x4 sb[3];                                   // score accumulators' initial values (bias rows) of the coming query tile
            if constexpr ((PF & 4) != 0) {
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) sb[kt] = bias_frag(hl, 0, kt);
            }

            // ---- attention of this head: 3 q tiles x 3 key tiles ---------------------------------------------------
#pragma unroll
            for (int qt = 0; qt < 3; ++qt) {
                f32x4 s[3];
                {
                    if constexpr (!(PF & 4)) {
#pragma unroll
                        for (int kt = 0; kt < 3; ++kt) sb[kt] = bias_frag(hl, qt, kt);
                    }
#pragma unroll
                    for (int kt = 0; kt < 3; ++kt) {
                        f32x4 acc = sb[kt];
                        if constexpr (HD == 16) {
                            if constexpr (kSplit && !SP && !kSplitK16) {
                                acc = MFMA_16x16x32(cat8r(kt4[0][kt], zero4), cat8r(qt4[0][qt], zero4), acc);
                            } else if constexpr (kSplit && !SP) {
                                // K = 16 is the whole head: v_mfma_f32_16x16x16_f16 takes q and k as they leave the converts (no
                                // 4-register operands to assemble, ~10 v_mov per head); the shift-region term is a constant over
                                // every row of these windows
                                acc = __builtin_amdgcn_mfma_f32_16x16x16f16(kt4[0][kt], qt4[0][qt], acc, 0, 0, 0);
                            } else {
                                acc = MFMA_16x16x32(cat8r(kt4[0][kt], rkr[kt]), cat8r(qt4[0][qt], rqr[qt]), acc);
                            }
                        } else {
                            acc = MFMA_16x16x32(cat8r(kt4[0][kt], kt4[1][kt]), cat8r(qt4[0][qt], qt4[1][qt]), acc);
                            if (sp_now) acc = MFMA_16x16x32(cat8r(rkr[kt], zero4), cat8r(rqr[qt], zero4), acc);
                        }
                        s[kt] = acc;
                    }
                    if ((PF & 4) && qt + 1 < 3) {           // the next query tile's bias rows travel behind this tile's softmax
#pragma unroll
                        for (int kt = 0; kt < 3; ++kt) sb[kt] = bias_frag(hl, qt + 1, kt);
                    }
                }
                // 9 real keys per lane: tiles 0 and 1 whole, register 0 of tile 2 (key 32 + grp); registers 1-3 of tile 2 are padding
                float mx = fmaxf(fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]), fmaxf(fmaxf(s[0][3], s[1][0]), s[1][1])),
                                 fmaxf(fmaxf(s[1][2], s[1][3]), s[2][0]));
                mx = row_group_max(mx);
                f16x4 pf[3];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const float p0 = __builtin_amdgcn_exp2f(s[kt][0] - mx), p1 = __builtin_amdgcn_exp2f(s[kt][1] - mx);
                    const float p2 = __builtin_amdgcn_exp2f(s[kt][2] - mx), p3 = __builtin_amdgcn_exp2f(s[kt][3] - mx);
                    pf[kt] = (f16x4){(f16)p0, (f16)p1, (f16)p2, (f16)p3};
                }
                pf[2] = lone_to_f16x4(__builtin_amdgcn_exp2f(s[2][0] - mx), zf);
                // ONE fragment [p0 + p1 | p2] serves the denominator (ones x it = the sum of the fp16 probabilities the PV
                // product actually uses) AND the key-tile-2 part of PV (V of tile 2 sits in the HIGH k-slots of vz, zeros below)
                const f16x8 psum = cat8r(pf[0] + pf[1], pf[2]);
                const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                const f32x4 sm = MFMA_16x16x32(ones8, psum, z4);
                const float inv = __builtin_amdgcn_rcpf(sm[0]);
                const bool store = qt < 2 || store2;
                // 16-byte stores: two adjacent 16-channel tiles -> one 8-channel run per lane (common.h pair_to_run).
                // head_dim 32: the two tiles of this head.
                f16x4 ov[NTH];
#pragma unroll
                for (int dt = 0; dt < NTH; ++dt) {
                    f32x4 o = MFMA_16x16x32(cat8r(vt4[dt][0], vt4[dt][1]), cat8r(pf[0], pf[1]), z4);
                    o = MFMA_16x16x32(cat8r(zero4, vt4[dt][2]), psum, o);
                    ov[dt] = mul4_to_f16(o, inv);
                }
                if constexpr (NTH == 2) {
                    const f16x8 run = pair_to_run(ov[0], ov[1]);
                    if (store) *reinterpret_cast<f16x8 *>(ab + (vo[qt] + (unsigned)st_delta) + head * (HD * 2)) = run;
                } else if constexpr (WM) {
                    // head_dim 16, window-major map (att is [window][head][36 tokens][HD], the C = 96 tail reads it so): the 64 lanes
                    // of this store cover ONE contiguous 512-byte run (16 tokens x 32 B) instead of sixteen 8-byte pieces 192 B
                    // apart (1.48x write amplification in the r02 counters).  Pairing with the neighbouring head's tile (held
                    // across one trip of the head loop) was measured slower (504 vs 457 us): plain 8-byte stores
                    if constexpr ((NUNIF_QKV_DIET & 1) != 0) {
                        // the base is pinned in an SGPR pair (an opaque asm: hipcc otherwise re-associates the sum into 64-bit
                        // per-lane arithmetic) so that the store is `global_store_dwordx2 v_off, v_data, s[base]`
                        unsigned long hb = reinterpret_cast<unsigned long>(a.att) + (wm_base + (unsigned)(head * (36 * HD * 2)));
                        asm("" : "+s"(hb));
                        typedef __attribute__((address_space(1))) char gchar;        // (an integer -> pointer cast would be a FLAT store)
                        typedef __attribute__((address_space(1))) f16x4 gf16x4;
                        unsigned lo = wm_off[qt];
                        asm("" : "+v"(lo));         // keeps the zero-extension next to the add (hoisted, it becomes a 64-bit VGPR pair)
                        if (store) *reinterpret_cast<gf16x4 *>(reinterpret_cast<gchar *>(hb) + lo) = ov[0];
                    } else {
                        if (store) *reinterpret_cast<f16x4 *>(ab + ((head * 36 + tokc[qt]) * HD + 4 * grp) * 2) = ov[0];
                    }
                } else {
                    if (store) *reinterpret_cast<f16x4 *>(ab + (vo[qt] + (unsigned)st_delta) + head * (HD * 2)) = ov[0];
                }
            }
            if constexpr (kHeadUnroll > 1) __builtin_amdgcn_sched_barrier(0);   // unrolled heads stay apart: nothing of head h + 1 is scheduled into head h
        }
        };
        if constexpr (MODE == 1) {
            run_heads(std::false_type{});
        } else if constexpr (MODE == 2) {
            run_heads(std::true_type{});
        } else if constexpr (kSplit) {
            if (special) run_heads(std::true_type{}); else run_heads(std::false_type{});
        } else {
            run_heads(std::true_type{});
        }

        if constexpr (kEarlyX && !WM) {
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) vo[mt] = von[mt];
        }
        if constexpr (!kEarlyX) {
            step();
            if (wi < a.n_windows) load_x(wb, wy, wx, special, xf, vo);
        }
    }
    };
    if constexpr (kTwoLoops) {
        run_windows(std::integral_constant<int, 1>{});
        if (a.shift > 0) run_windows(std::integral_constant<int, 2>{});
    } else {
        run_windows(std::integral_constant<int, 0>{});
    }
}

int qkv_attn_r_frags(int C) { return 3 * (C / 16) * (C / 32); }

template <int C, int HD, int HPP, int WAVES = 8, bool WM = false>
static int launch_r(const QkvAttnRArgs &a, int grid, hipStream_t s) {
    constexpr size_t smem = (size_t)HPP * 3 * (HD / 16) * (C / 32) * 1024 + HPP * kQkvBiasFloatsPerHead * 4 +
                            (kQkvBiasFragMajor ? (size_t)HPP * 3 * (HD / 16) * 1024 : (size_t)3 * C * 4);
    static bool configured = false;
    if (!configured) {
        NUNIF_HIP_CHECK(hipFuncSetAttribute((const void *)qkv_attn_r_kernel<C, HD, HPP, WAVES, WM>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        configured = true;
    }
    qkv_attn_r_kernel<C, HD, HPP, WAVES, WM><<<grid, WAVES * 64, smem, s>>>(a);
    return NUNIF_HIP_OK;
}

int launch_qkv_attn_r(const f16 *x, f16 *att, const f16 *wres, const float *bqkv, const float *btab32,
                      int B, int H, int W, int C, int heads, int shift, hipStream_t s, int rev, int window_major) {
    NUNIF_REQUIRE(H % 6 == 0 && W % 6 == 0, "qkv_attn: %dx%d not a multiple of the 6x6 window", H, W);
    NUNIF_REQUIRE(heads == 6 && (C == 96 || C == 192), "qkv_attn: C=%d heads=%d unsupported", C, heads);
    if (H <= 6) shift = 0;                 // torchvision disables the shift when the window covers the map
    QkvAttnRArgs a;
    a.x = x; a.att = att; a.wres = wres; a.bqkv = bqkv; a.btab32 = btab32;
    a.B = B; a.H = H; a.W = W; a.shift = shift;
    a.n_windows = B * (H / 6) * (W / 6);
    a.rev = rev;            // snake order between consecutive kernels (swin_unet.cpp next_dir)
    NUNIF_REQUIRE(!window_major || C == 96, "qkv_attn: the window-major att map exists for C = 96 only");
    const double tok = (double)B * H * W;
    // C = 96: 16 waves per workgroup (120 registers, no register prefetch: four waves per SIMD hide the x loads; 8 waves with
    // prefetch measured 10 % slower, 12 waves in between); C = 192: 8 waves
    const int kWavesR = C == 96 ? 16 : 8;
    const int wgs = (a.n_windows + kWavesR - 1) / kWavesR;
    int grid = wgs < 256 ? wgs : 256;                   // persistent: one workgroup per CU
    if (C == 192) grid = std::max(2, grid & ~1);        // two head passes = two kinds of workgroup
    int rc;
    if (C == 96) {
        ProfScope ps("qkv_attn_r_kernel<96,16>", s, 2.0 * tok * C * 3.0 * C + 4.0 * tok * 36.0 * C, tok * C * 4.0);
        rc = window_major ? launch_r<96, 16, 6, 16, true>(a, grid, s) : launch_r<96, 16, 6, 16, false>(a, grid, s);
    } else {
        ProfScope ps("qkv_attn_r_kernel<192,32>", s, 2.0 * tok * C * 3.0 * C + 4.0 * tok * 36.0 * C, tok * C * 4.0);
        rc = launch_r<192, 32, 3>(a, grid, s);
    }
    if (rc) return rc;
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

}  // namespace nunif
