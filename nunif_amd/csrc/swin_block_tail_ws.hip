// Tail of a swin block at C = 192 on gfx950, WEIGHT-STATIONARY form: W0 and W3 live in the register file for the whole
// launch, Wp in LDS; the activations move through LDS.
//
//     y  = x + Wp att + bp                     (attn.proj + residual)
//     x' = y + W3 gelu(W0 y + b0) + b3         (mlp.0, GELU(erf), mlp.3 + residual)      in place on x
//
// (torchvision SwinTransformerBlock's `x = x + proj(...)`, `x = x + mlp(norm2(x))` with norm = Identity,
//  waifu2x/models/swin_unet.py:16-17,26-36 — same contract as proj_mlp_kernel in swin_block_tail.hip.)
//
// Why.  At C = 192 the three matrices are 360 KiB of fp16: more than the 160 KiB of LDS, so the round-1 kernel pulled
// them through an LDS ring once per 128 tokens (1.9 GB of L2 -> LDS traffic per launch on the 120 x 120 level, one
// workgroup barrier per 8 KiB) and ran at 25 % of the MFMA peak.  But a CU's register file is 512 KiB.  ONE persistent
// 8-wave workgroup per CU keeps the weights on chip for the whole launch, split by OUTPUT channel and by ROLE:
//     waves 0-3 ("P", slice w):  W3 rows 48w..48w+47 in 144 registers, Wp rows 48w..48w+47 read from LDS (72 KiB, all slices)
//     waves 4-7 ("H", slice w):  W0 rows 96w..96w+95 in 144 registers
// Waves w and w + 4 share a SIMD (a workgroup's waves are dealt 0,1,2,3,0,1,2,3 over the SIMDs), and the roles are
// complementary on purpose: a P wave issues 54 MFMAs and ~100 VALU operations per 16 tokens, an H wave 36 MFMAs and the
// ~270 VALU operations of GELU — tools/ubench_mix.hip (profiles/r02_ubench_mix.txt): a SIMD issues about one instruction
// per 4 cycles next to MFMAs and a lone wave only one per 5.5-6.3, so the GELU has to sit in a DIFFERENT wave than the
// bulk of the MFMAs to be hidden.
// What moves is the activation tile of 16 tokens; every wave needs the whole K extent of att / y / hidden, so those go
// through LDS in MFMA-fragment-major form (one conflict-free ds_read_b128 per fragment, each feeding 3 or 6 MFMAs):
//     stage A (P, tile i+2): att (LDS-DMA'd straight from HBM by the H waves: global_load_lds_dwordx4) -> y slice -> LDS
//     stage B (H, tile i+1): y (LDS) -> W0, GELU -> hidden slice                                                 -> LDS
//     stage C (P, tile i)  : hidden (LDS) -> W3, + y + b3 -> x' slice                                            -> HBM
// The three stages of one loop trip work on three DIFFERENT tiles, so a trip needs exactly one workgroup barrier
// (LDS: att x3, x x3, y x3, hidden x2 slots = 78 KiB, Wp 72 KiB, biases 3 KiB).  HBM latency under load is more than one
// trip, so the att and x tiles are requested two trips ahead, both by LDS-DMA.
//
// HBM traffic per token is unchanged: read att (2C B) + read x (2C B) + write x (2C B).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "swin_gelu.h"
#include "swin_kernels.h"

namespace nunif {

#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

typedef __attribute__((address_space(3))) void lds_void;

constexpr int kWsC = 192;
// weight stream (1-KiB fragments): Wp [slice w][nt 3][ks 6] (plain k order) | W0 [w][nt 6][ks 6] | W3 [w][nt 3][ks 12] (chained)
constexpr int kWsWpFrags = 4 * 18, kWsW0Frags = 4 * 36, kWsW3Frags = 4 * 36;

constexpr int kDmaAhead = 2;              // the att / x tiles of stage A are requested this many trips ahead
constexpr int kInSlots = kDmaAhead + 1;
constexpr int kWsLdsKiB = 2 * kInSlots * 6 + 3 * 6 + 2 * 12 + kWsWpFrags;     // 153 KiB (+ 3 KiB of biases) of 160

int proj_mlp_ws_stream_frags() { return kWsWpFrags + kWsW0Frags + kWsW3Frags; }

#ifdef NUNIF_ABLATIONS
// ABL & 256: per-trip s_memtime stamps of workgroup 0 into g_ws_trace[wave 8][trip 64][point 8] (tools/trace_tail_ws.py)
__device__ long long *g_ws_trace = nullptr;
extern "C" int nunif_dbg_ws_trace(void *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_ws_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#define WS_STAMP(pt)                                                                                          \
    do {                                                                                                      \
        if constexpr ((ABL & 256) != 0) {                                                                     \
            if (blockIdx.x == 0 && i >= 8 && i < 72 && g_ws_trace) {                                         \
                const long long t_ = __builtin_readcyclecounter();                                            \
                if (lane == 0) g_ws_trace[(wave * 64 + (i - 8)) * 8 + (pt)] = t_;                             \
            }                                                                                                 \
        }                                                                                                     \
    } while (0)
#else
#define WS_STAMP(pt) do { } while (0)
#endif

// ABL != 0: timing-only ablations (wrong results), compiled only with -DNUNIF_ABLATIONS (NUNIF_BUILD_ABL=1 python -m
// nunif_amd.build) and selected with NUNIF_TAIL_WS_ABL: 1 = no GELU polynomial, 2 = no MFMA, 4 = no HBM traffic inside the
// loop (no att DMA, no x loads, no x' stores), 8 = no barrier, 16 = no bias reads from LDS, 32 = no Wp reads from LDS,
// 64 = stage C reads only 4 of its 12 hidden fragments
template <int ABL, bool G32 = false>
__global__ void __launch_bounds__(512, 2)
proj_mlp_ws_kernel(const f16 *__restrict__ att, f16 *x, const f16 *__restrict__ wws, const float *__restrict__ bp,
                   const float *__restrict__ b0, const float *__restrict__ b3, long M, int rev) {
    constexpr int C = kWsC, KS = C / 32, HS = 2 * C / 32;   // k-steps of the C-wide and of the hidden (2C) contraction
    constexpr int PT = 3, HT = 6;                           // 16-row output tiles per wave: proj / mlp.3, mlp.0
    constexpr int TOK = 16;
    constexpr int ATT_SLOT = KS * 64, Y_SLOT = KS * 64, H_SLOT = HS * 64;   // in f16x8 (16 B) units
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ws[];
    f16x8 *att_l = reinterpret_cast<f16x8 *>(smem_ws);      // [kInSlots][KS][64]  plain k order (as stored in HBM)
    f16x8 *x_l = att_l + kInSlots * ATT_SLOT;               // [kInSlots][KS][64]  residual x, plain k order
    f16x8 *y_l = x_l + kInSlots * ATT_SLOT;                      // [3][KS][64]  chained k order (accumulator-tile pairs)
    f16x8 *h_l = y_l + 3 * Y_SLOT;                          // [2][HS][64]  chained k order
    f16x8 *wp_l = h_l + 2 * H_SLOT;                         // [4][PT][KS][64]
    float *bias_l = reinterpret_cast<float *>(wp_l + kWsWpFrags * 64);   // bp[C] | b0[2C] | b3[C]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slice = wave & 3;
    const bool role_h = wave >= 4;
    const int r16 = lane & 15;
    const int grp = lane >> 4;

    for (int i = tid; i < C; i += 512) { bias_l[i] = bp[i]; bias_l[3 * C + i] = b3[i]; }
    for (int i = tid; i < 2 * C; i += 512) bias_l[C + i] = b0[i];
    {
        const f16x8 *src = reinterpret_cast<const f16x8 *>(wws);
        for (int i = tid; i < kWsWpFrags * 64; i += 512) wp_l[i] = src[i];
    }
    // ---- this wave's register-resident slice: 36 fragments = 144 registers (W3 rows for P waves, W0 rows for H waves) ---
    f16x8 wr[36];
    {
        const f16x8 *src = reinterpret_cast<const f16x8 *>(wws) +
                           (long)(kWsWpFrags + (role_h ? 0 : kWsW0Frags) + slice * 36) * 64 + lane;
#pragma unroll
        for (int f = 0; f < 36; ++f) wr[f] = src[f * 64];
    }

    const long n_tiles = (M + TOK - 1) / TOK;
    const int n_mine = (int)((n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x);     // tiles blockIdx.x + k * gridDim.x
    auto tile_of = [&](int k) -> long {
        const long t = blockIdx.x + (long)k * gridDim.x;
        return rev ? n_tiles - 1 - t : t;                    // snake order between consecutive kernels
    };

    // The att AND the x tile of the k-th tile go straight from HBM into LDS slot k % kInSlots (LDS-DMA, no registers):
    // fragment ks = lane (r16, grp) holds row[token r16][32 ks + 8 grp .. + 7].  Issued by the H waves kDmaAhead tiles ahead
    // of stage A; slice w takes fragments w and w + 4 of both tiles.  Inline asm on purpose: hipcc orders every later
    // ds_read behind an LDS-DMA it can see with `s_waitcnt vmcnt(0)` (it cannot tell the slots apart), and any vector load
    // it CAN see in this loop drags vmcnt waits into both roles (shared registers); so the loop has no compiler-visible
    // vector loads at all and the vmcnt of the H waves is counted by hand (dma_wait).
    auto dma_frag = [&](const f16 *src, const f16x8 *dst) {
        const unsigned lds_addr = (unsigned)reinterpret_cast<size_t>(dst);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr) : "memory");
    };
    auto dma_in = [&](int k, int slot) {
        long m = tile_of(k) * TOK + r16;
        m = m < M ? m : M - 1;
        const long off = m * C + grp * 8 + slice * 32;
        dma_frag(att + off, att_l + slot * ATT_SLOT + slice * 64);
        dma_frag(x + off, x_l + slot * ATT_SLOT + slice * 64);
        if (slice + 4 < KS) {
            dma_frag(att + off + 128, att_l + slot * ATT_SLOT + (slice + 4) * 64);
            dma_frag(x + off + 128, x_l + slot * ATT_SLOT + (slice + 4) * 64);
        }
    };
    // tile k has landed once at most (kDmaAhead - 1) younger tiles' DMAs of this wave are outstanding (loads return in order)
    auto dma_wait = [&](bool full_pipe) {
        static_assert(kDmaAhead == 2, "the vmcnt immediates below are 4 * (kDmaAhead - 1) and 2 * (kDmaAhead - 1)");
        if (!full_pipe) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (slice < 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    };

    const int ch_p = 48 * slice + 4 * grp;      // first channel of this lane in the slice of a P wave (tile 0)
    const int ch_h = 96 * slice + 4 * grp;      // ... in the hidden slice of an H wave

    // every weight load has landed before the loop — and the compiler must KNOW it (an opaque use of each register): its
    // s_waitcnt for a first use inside the loop would be vmcnt(0) on every trip, which drains the hand-counted DMAs as well
#pragma unroll
    for (int f = 0; f < 36; ++f) asm volatile("" : "+v"(wr[f]));
    if (role_h) {
#pragma unroll
        for (int k = 0; k < kDmaAhead; ++k)
            if (k < n_mine) dma_in(k, k);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // rotating slot numbers (wave-uniform): in_* of the att / x tiles, y* of the y tiles, h* of the hidden tiles
    int in_a = 0;                        // slot of tile i + 2 (stage A of this trip); the DMA target is (in_a + kDmaAhead) % kInSlots
    int y_c = 1, y_b = 2, y_a = 0;       // y slots of tiles i (stage C), i + 1 (stage B), i + 2 (stage A): i = -2 -> tile 0 in slot 0
    int h_c = 0, h_b = 1;                // hidden slots of tiles i (stage C) and i + 1 (stage B)

#pragma unroll 1
    for (int i = -2; i < n_mine; ++i) {
        if (role_h) {
            // ---- H wave: DMA of tile i + 2 + kDmaAhead, stage B of tile i + 1 ---------------------------------------------
            WS_STAMP(0);
            const bool pipe = i + 2 + kDmaAhead < n_mine;
            if (!(ABL & 4) && pipe) {
                int slot = in_a + kDmaAhead;
                slot = slot >= kInSlots ? slot - kInSlots : slot;
                dma_in(i + 2 + kDmaAhead, slot);
            }
            WS_STAMP(1);
            if (i + 1 >= 0 && i + 1 < n_mine) {
                const f16x8 *ys = y_l + y_b * Y_SLOT + lane;
                f16x8 *hd = h_l + h_b * H_SLOT + lane;
                f16x8 bq[KS];                // all six operand fragments of the tile are requested up front (24 registers)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) bq[ks] = ys[ks * 64];
                f32x4 acc[HT];
#pragma unroll
                for (int nt = 0; nt < HT; ++nt) {
                    if constexpr (ABL & 16) { acc[nt] = (f32x4){0.1f, 0.2f, 0.3f, 0.4f}; continue; }
                    acc[nt] = *reinterpret_cast<const f32x4 *>(bias_l + C + ch_h + 16 * nt);
                }
                // The GELU of this wave is ~230 VALU operations per tile and a lone wave issues one per ~5.5 cycles, so it
                // must not queue up behind the MFMAs: the six hidden tiles go in three PAIRS (two adjacent 16-channel tiles
                // = one 8-slot fragment of the next contraction, chained k order), and the GELU of pair p is issued next to
                // the 12 MFMAs of pair p + 1.  (Pinning one row of 8 GELU operations behind each MFMA with sched_barrier
                // measured the same, 164-168 vs 161-163 us, so the placement is left to the compiler.)
                auto mfma_at = [&](int p, int j) {          // j-th of the 12 MFMAs of pair p: k-step j / 2, tile 2p + (j & 1)
                    const int ks = j >> 1, nt = 2 * p + (j & 1);
                    if constexpr (ABL & 2) { acc[nt][0] += (float)wr[nt * KS + ks][0] * (float)bq[ks][nt]; return; }
                    acc[nt] = MFMA_16x16x32(wr[nt * KS + ks], bq[ks], acc[nt]);
                };
                auto raw_pair = [&](int p) {
                    const f32x4 &u = acc[2 * p], &v = acc[2 * p + 1];
                    hd[(3 * slice + p) * 64] = (f16x8){(f16)u[0], (f16)u[1], (f16)u[2], (f16)u[3], (f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                };
#pragma unroll
                for (int j = 0; j < 2 * KS; ++j) mfma_at(0, j);
                WS_STAMP(2);
#pragma unroll
                for (int p = 1; p < HT / 2; ++p) {
                    if constexpr (ABL & 1) {
#pragma unroll
                        for (int j = 0; j < 2 * KS; ++j) mfma_at(p, j);
                        raw_pair(p - 1);
                    } else {
#pragma unroll
                        for (int j = 0; j < 2 * KS; ++j) mfma_at(p, j);
                        hd[(3 * slice + p - 1) * 64] = gelu8t<G32>(acc[2 * p - 2], acc[2 * p - 1]);
                    }
                }
                WS_STAMP(3);
                if constexpr (ABL & 1) raw_pair(HT / 2 - 1);
                else hd[(3 * slice + HT / 2 - 1) * 64] = gelu8t<G32>(acc[HT - 2], acc[HT - 1]);
                WS_STAMP(4);
            }
            // the next trip's stage A reads tile i + 3: requested kDmaAhead trips ago
            dma_wait(pipe);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            WS_STAMP(5);
        } else {
            WS_STAMP(0);
            // ---- P wave: stage C of tile i, stage A of tile i + 2 -------------------------------------------------------------
            const bool do_a = i + 2 < n_mine, do_c = i >= 0;
            const f16x8 *as = att_l + in_a * ATT_SLOT + lane;
            const f16x8 *ws = wp_l + (slice * PT * KS) * 64 + lane;
            // operand rings: fragments are requested kHAhead (stage C) / 2 (stage A) k-steps before the MFMAs that consume
            // them; stage A's first two steps are requested from inside stage C's loop
            constexpr int kHAhead = 4;
            f16x8 aq[3], wq[3][PT];
            auto req_a = [&](int ks) {
                aq[ks % 3] = as[ks * 64];
#pragma unroll
                for (int nt = 0; nt < PT; ++nt) {
                    if constexpr (ABL & 32) { wq[ks % 3][nt] = wr[nt * KS + ks]; continue; }
                    wq[ks % 3][nt] = ws[(nt * KS + ks) * 64];
                }
            };
            // ---- stage C: x' = y + b3 + W3 hidden ------------------------------------------------------------------------
            if (do_c) {
                const f16x8 *hs = h_l + h_c * H_SLOT + lane;
                const f16x8 *ys = y_l + y_c * Y_SLOT + lane;
                f16x8 hq[kHAhead + 1];
#pragma unroll
                for (int ks = 0; ks < kHAhead; ++ks) hq[ks] = hs[ks * 64];
                f32x4 acc[PT];
#pragma unroll
                for (int nt = 0; nt < PT; ++nt) {
                    const f32x4 b = (ABL & 16) ? (f32x4){0.1f, 0.2f, 0.3f, 0.4f} : *reinterpret_cast<const f32x4 *>(bias_l + 3 * C + ch_p + 16 * nt);
                    const int tile = 3 * slice + nt;
                    const f16x4 yo = *(reinterpret_cast<const f16x4 *>(ys + (tile >> 1) * 64) + (tile & 1));
                    acc[nt] = (f32x4){(float)yo[0] + b[0], (float)yo[1] + b[1], (float)yo[2] + b[2], (float)yo[3] + b[3]};
                }
#pragma unroll
                for (int ks = 0; ks < HS; ++ks) {
                    if (!(ABL & 64) && ks + kHAhead < HS) hq[(ks + kHAhead) % (kHAhead + 1)] = hs[(ks + kHAhead) * 64];
                    if (ks == HS - 4) req_a(0);
                    if (ks == HS - 2) req_a(1);
#pragma unroll
                    for (int nt = 0; nt < PT; ++nt) {
                        if constexpr (ABL & 2) { acc[nt][0] += (float)wr[nt * HS + ks][0] * (float)hq[ks % (kHAhead + 1)][nt]; continue; }
                        acc[nt] = MFMA_16x16x32(wr[nt * HS + ks], hq[ks % (kHAhead + 1)], acc[nt]);
                    }
                }
                WS_STAMP(2);
                const long m = tile_of(i) * TOK + r16;
                if (m < M && (!(ABL & 4) || acc[0][0] == 12345.f)) {
#pragma unroll
                    for (int nt = 0; nt < PT; ++nt) {
                        const f16x4 o = {(f16)acc[nt][0], (f16)acc[nt][1], (f16)acc[nt][2], (f16)acc[nt][3]};
                        *reinterpret_cast<f16x4 *>(x + m * C + ch_p + 16 * nt) = o;
                    }
                }
                WS_STAMP(3);
            } else {
                req_a(0);
                req_a(1);
            }
            // ---- stage A: y = x + bp + Wp att ---------------------------------------------------------------------------------
            if (do_a) {
                f16x8 *yd = y_l + y_a * Y_SLOT + lane;
                const f16x8 *xs = x_l + in_a * ATT_SLOT;
                f32x4 acc[PT];
#pragma unroll
                for (int nt = 0; nt < PT; ++nt) {
                    const f32x4 b = (ABL & 16) ? (f32x4){0.1f, 0.2f, 0.3f, 0.4f} : *reinterpret_cast<const f32x4 *>(bias_l + ch_p + 16 * nt);
                    // channels c .. c + 3 of token r16 in the plain layout: fragment c / 32, lane r16 + 16 ((c % 32) / 8)
                    const int c = ch_p + 16 * nt;
                    const f16x4 xv = *(reinterpret_cast<const f16x4 *>(xs + (c >> 5) * 64 + r16 + 16 * ((c & 31) >> 3)) + ((c >> 2) & 1));
                    acc[nt] = (f32x4){(float)xv[0] + b[0], (float)xv[1] + b[1], (float)xv[2] + b[2], (float)xv[3] + b[3]};
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks + 2 < KS) req_a(ks + 2);
#pragma unroll
                    for (int nt = 0; nt < PT; ++nt) {
                        if constexpr (ABL & 2) { acc[nt][0] += (float)wq[ks % 3][nt][0] * (float)aq[ks % 3][nt]; continue; }
                        acc[nt] = MFMA_16x16x32(wq[ks % 3][nt], aq[ks % 3], acc[nt]);
                    }
                }
#pragma unroll
                for (int nt = 0; nt < PT; ++nt) {
                    const int tile = 3 * slice + nt;
                    const f16x4 yv = {(f16)acc[nt][0], (f16)acc[nt][1], (f16)acc[nt][2], (f16)acc[nt][3]};
                    *(reinterpret_cast<f16x4 *>(yd + (tile >> 1) * 64) + (tile & 1)) = yv;
                }
            }
            WS_STAMP(4);
            // no vmcnt wait here: the x' stores of stage C are fire-and-forget
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            WS_STAMP(5);
        }
        // one rendezvous per trip: LDS writes of stages A / B (and the DMA'd tiles) become visible to the other waves
        if constexpr (!(ABL & 8)) asm volatile("s_barrier" ::: "memory");
        WS_STAMP(6);
        in_a = in_a + 1 == kInSlots ? 0 : in_a + 1;
        { const int t = y_c; y_c = y_b; y_b = y_a; y_a = t; }
        { const int t = h_c; h_c = h_b; h_b = t; }
    }
}

int launch_proj_mlp_ws(const f16 *att, f16 *x, const f16 *wws, const float *bp, const float *b0, const float *b3, long M,
                       hipStream_t s, int rev, int gelu32) {
    if (M == 0) return NUNIF_HIP_OK;
    ProfScope ps("proj_mlp_ws_kernel", s, 2.0 * (double)M * kWsC * kWsC * 5.0, (double)M * kWsC * 2.0 * 3.0);
    constexpr size_t smem = (size_t)kWsLdsKiB * 1024 + 4 * kWsC * 4;
    const long n_tiles = (M + 15) / 16;
    const unsigned blocks = (unsigned)std::min<long>(n_tiles, 256);
    auto go = [&](auto kern, int slot) -> int {
        static bool configured[1024] = {false};
        if (!configured[slot]) {
            NUNIF_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            configured[slot] = true;
        }
        kern<<<blocks, 512, smem, s>>>(att, x, wws, bp, b0, b3, M, rev);
        return NUNIF_HIP_OK;
    };
    int rc;
    rc = gelu32 ? go(proj_mlp_ws_kernel<0, true>, 1) : go(proj_mlp_ws_kernel<0, false>, 0);       // ABL != 0 instantiations are timing-only experiments (see the kernel header), never shipped
    if (rc) return rc;
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

}  // namespace nunif
