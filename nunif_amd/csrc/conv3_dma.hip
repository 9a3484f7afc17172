// 3x3 stride-1 conv for gfx950, PERSISTENT, with input halo AND weights moved into LDS by LDS-DMA (global_load_lds_dwordx4).
//
// Same contract and the same accumulation order over k as conv3_lds_kernel (conv3_lds.hip) for the shapes it takes — results are
// bit-identical — and the same callers: the VALID 3x3 convs of waifu2x cunet / upcunet / vgg_7 / upconv_7
// (waifu2x/models/cunet.py:10-28 UNetConv, :31-121), the padded 3x3 convs of the side nets.
//
// Why (profiles/r04c_sq.txt, cunet 1080p, 66 tiles per launch).  conv3_lds_kernel<4> ran at 10 % of the matrix pipe: a wave lived
// 17 us for 2.1 us of MFMAs, 52 % of its cycles in s_waitcnt / s_barrier, and issued 2 100 VALU instructions next to 255 MFMAs —
//   * every workgroup staged its halo through registers (11 loads + 11 ds_write_b128 per thread, ~60 VALU of index arithmetic
//     per item), and nothing else happened in that workgroup until the last store had landed;
//   * the weight ring went through registers too (a copy chain in rounds 1-3 that forced `vmcnt(0)` at every chunk boundary);
//   * one patch per workgroup: launch, weight-pipe fill, halo round trip and epilogue of a patch never overlapped with its own
//     MFMAs, only with the one other workgroup an LDS-limited CU holds.
// Here a workgroup is persistent and owns a sequence of 8 x 32 patches:
//   * HALO: (8+2) x (32+2) pixels x Cin channels go global -> LDS by DMA, no registers, no ds_write.  The LDS image is LINEAR in
//     (pixel, 16-byte segment) with the segment index XOR-swizzled by the pixel index (seg ^ ((pixel / PPR) % SEG), PPR = pixels
//     per 256 B): any 16 consecutive pixels then read their fragment from 64 distinct banks — conflict free for every tap shift,
//     with no padding (the round-3 kernel padded each pixel by 16 B, which a DMA cannot).  A lane's 36 fragment addresses
//     (4 token tiles x 9 taps) are constants of the launch; the second 32-channel part of a tap is `address ^ 64`.
//     The next patch's halo is requested as soon as the last k-step has read the current one, i.e. it travels under the epilogue;
//   * WEIGHTS: 8-KiB chunks through a 3-slot LDS ring, also by DMA, two chunks ahead.  The chunk sequence simply wraps from one
//     patch to the next (the trip count per patch is rounded up to a multiple of three so that slot = chunk % 3 holds across
//     patches), so the ring never drains;
//   * zero padding (Conv2d(padding=1)) = the lanes of out-of-map pixels fetch from the zero padding behind the weight stream;
//     replicate padding / VALID edges = clamped coordinates.
// LDS: 24 KiB ring + 340 x Cin x 2 B halo = 67 KiB at Cin = 64: two workgroups per CU, as before.
//
// vmcnt is counted by hand (the DMAs are inline asm, hipcc never sees a vector-memory instruction inside the k loop): loads return
// in order, a wave issues 2 DMA instructions per weight chunk, so `vmcnt(2)` at a chunk boundary = "my part of this chunk has
// landed, the next one may still be in flight"; the workgroup barrier behind it publishes all four waves' parts.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <cstdio>

#include "swin_kernels.h"

namespace nunif {

#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

namespace {
constexpr int kTH = 8, kTW = 32;                        // output patch of a workgroup (4 waves x 2 rows x 32 columns)
constexpr int kHH = kTH + 2, kHW = kTW + 2;             // halo
constexpr int kHaloPix = kHH * kHW;                     // 340
constexpr int kCH = 8;                                  // fragments (KiB) per weight chunk

__device__ __forceinline__ void dma16(const void *src, unsigned lds_byte_addr) {
    // each lane moves 16 B to LDS[m0 + 16 * lane]; m0 is wave-uniform
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_byte_addr) : "memory");
}
}  // namespace

// NTT output tiles in NPASS passes of NT = NTT / NPASS over the SAME halo (64 -> 128: two passes of four tiles; eight tiles x four
// token tiles of accumulators do not fit 256 registers)
#ifdef NUNIF_C3D_TRACE
__device__ unsigned long long g_c3d_trace[1024 * 8];
#define NUNIF_C3D_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_c3d_trace[blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define NUNIF_C3D_STAMP(i)
#endif

// D = how many weight chunks ahead of the MFMAs the DMAs run (ring of D + 1 slots; 2: 67 KiB of LDS, two workgroups per CU).
// A deeper ring for the launches of one patch per workgroup (the DPT head's 64 -> 64 layers: 32 / 112 / 392 patches) was measured
// and is NOT dispatched: D = 8 (the whole 72 KiB stream in flight) 0.274 vs 0.258 ms per 13 launches — phase stamps show the chunk
// loop at 8.6 k ticks either way (it is not waiting for weights), the halo wait 6-9 k and the EPILOGUE 11-21 k
// (profiles/r04_c3d_trace.txt), which is what the 16-byte-run epilogue below addresses.
// KS > 1: the input has KS x CIN channels and the contraction runs in KS halves over the SAME LDS halo space (Cin = 128 as 2 x 64:
// a 128-channel halo would be 87 KiB, one workgroup per CU).  Half kh stages channels CIN kh .. of the patch, walks the k-steps
// tap-major inside the half (global k-step = tap (CPT KS) + kh CPT + part) into the same accumulators; the epilogue follows the
// last half.  Between halves the halo is re-staged behind a barrier (nothing to overlap it with but the ring's two chunks).
template <int NTT, int CIN, bool RELU_IN, int NPASS = 1, int D = 2, int KS = 1>
__global__ void __launch_bounds__(256, 2) conv3_dma_kernel(ConvArgs g, int n_patches) {
    static_assert(KS == 1 || NPASS == 1, "K halves and N passes are not combined");
    constexpr int kSlots = D + 1;
    constexpr int MF = 4;
    constexpr int NT = NTT / NPASS;
    constexpr int SEG = CIN / 8;                         // 16-byte segments per pixel
    constexpr int PB = CIN * 2;                          // bytes per pixel
    constexpr int PPR = 256 / PB;                        // pixels per 256-byte LDS row
    constexpr int CPT = CIN / 32;                        // 32-channel parts per tap
    constexpr int KSTEPS = 9 * CPT;
    constexpr int KPC = kCH / NT;                        // k-steps per chunk
    constexpr int NCH = (KSTEPS * NT + kCH - 1) / kCH;   // chunks that carry data
    constexpr int NCH3 = (NCH + kSlots - 1) / kSlots * kSlots;   // trip count per patch: a multiple of the slot count
    static_assert(D >= 2 && D <= NCH3, "the prologue requests D distinct chunks of the first pass");
    constexpr int HITEMS = kHaloPix * SEG;               // 16-byte items of a halo
    constexpr int HDMA = (HITEMS + 255) / 256;           // DMA instructions per wave... per THREAD ROW: items tid + 256 u
    static_assert(CIN == 32 || CIN == 64, "segment swizzle / part addressing below are written for 32 and 64 input channels");
    static_assert(kCH % NT == 0, "a chunk holds whole k-steps");

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_d[];
    f16x8 *ring = reinterpret_cast<f16x8 *>(smem_d);                      // [kSlots][kCH * 64]
    unsigned char *halo = smem_d + kSlots * kCH * 1024;                   // [340 px][SEG] 16-byte items, swizzled
    const unsigned ring_lds = (unsigned)reinterpret_cast<size_t>(ring);
    const unsigned halo_lds = (unsigned)reinterpret_cast<size_t>(halo);

    const int tid = threadIdx.x;
    const int lane = tid & 63, r16 = lane & 15, grp = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pad = (g.zpad || g.rpad) ? 1 : 0;                           // 0: VALID 3x3 (Ho = Hi - 2)
    const int tiles_x = (g.Wo + kTW - 1) / kTW, tiles_y = (g.Ho + kTH - 1) / kTH;
    const f16x8 *gsrc = reinterpret_cast<const f16x8 *>(g.wstream);       // zero-padded by 16 KiB on the host
    const f16x8 *zero16 = gsrc + (long)KSTEPS * KS * NTT * 64;            // first 16 bytes of that padding: the "zero pixel"

    // ---- per-lane constants --------------------------------------------------------------------------------------------
    // halo item q = tid + 256 u lives at LDS byte 16 q = pixel (q / SEG), slot (q % SEG); it holds segment slot ^ swz(pixel)
    int hq[HDMA];                                                         // (y << 16) | (x << 8) | segment
#pragma unroll
    for (int u = 0; u < HDMA; ++u) {
        const int q = min(tid + 256 * u, HITEMS - 1);
        const int p = q / SEG, slot = q - p * SEG;
        const int hy = p / kHW, hx = p - hy * kHW;
        hq[u] = (hy << 16) | (hx << 8) | (slot ^ ((p / PPR) % SEG));
    }
    // B fragment of token tile f (row 2 wave + (f >> 1), columns 16 (f & 1) + r16) at tap (dy, dx), part 0: segment grp
    unsigned faddr[MF][9];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - 3 * dy;
            const int p = (2 * wave + (f >> 1) + dy) * kHW + 16 * (f & 1) + r16 + dx;
            faddr[f][tap] = (unsigned)(p * PB + ((grp ^ ((p / PPR) % SEG)) * 16));
        }

    auto patch_of = [&](int i, int &b, int &ty0, int &tx0) {
        tx0 = (i % tiles_x) * kTW;
        ty0 = ((i / tiles_x) % tiles_y) * kTH;
        b = i / (tiles_x * tiles_y);
    };
    auto halo_dma = [&](int b, int ty0, int tx0, int kh) {
        const long img = (long)b * g.Hi * g.Wi;
#pragma unroll
        for (int u = 0; u < HDMA; ++u) {
            if (256 * u + 64 * wave < HITEMS) {                           // wave-uniform: the last row of instructions is partial
                const int yy = ty0 + (hq[u] >> 16) - pad, xx = tx0 + ((hq[u] >> 8) & 255) - pad;
                const int yc = min(max(yy, 0), g.Hi - 1), xc = min(max(xx, 0), g.Wi - 1);
                const f16 *src = g.a + ((img + (long)yc * g.Wi + xc) * (CIN * KS) + kh * CIN + (hq[u] & 255) * 8);
                if (g.zpad && (yy != yc || xx != xc)) src = reinterpret_cast<const f16 *>(zero16);
                dma16(src, halo_lds + (unsigned)(256 * u + 64 * wave) * 16);
            }
        }
    };
    // weight chunk c of pass p -> ring slot: 2 DMA instructions per wave.  A chunk = KPC k-steps x the NT fragments of the
    // pass; the stream is [k-step][NTT fragments], so fragment i of the chunk is k-step c KPC + i / NT, tile p NT + i % NT
    // (contiguous when NPASS == 1).  Fragments past the last k-step come from the zero padding behind the stream.
    auto chunk_dma = [&](int c, int p, int slot, int kh) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = wave + 4 * h;                                   // fragment of the chunk this wave moves
            const int kl = min(c * KPC + i / NT, KSTEPS);                 // k-step inside the half; KSTEPS: the zero padding
            const int ks = kl < KSTEPS ? (kl / CPT) * (CPT * KS) + kh * CPT + kl % CPT : KSTEPS * KS;
            const f16x8 *src = gsrc + ((long)ks * NTT + (kl < KSTEPS ? p * NT + i % NT : i % NT)) * 64 + lane;
            dma16(src, ring_lds + (unsigned)(slot * kCH * 64 + i * 64) * 16);
        }
    };

    int pi = blockIdx.x;
    int b, ty0, tx0;
    NUNIF_C3D_STAMP(0);
    if (pi < n_patches) {
        patch_of(pi, b, ty0, tx0);
        halo_dma(b, ty0, tx0, 0);                                         // first: the top wait below releases it with chunk 0
#pragma unroll
        for (int c = 0; c < D; ++c) chunk_dma(c, 0, c, 0);
    }
    bool first = true;
#pragma unroll 1
    for (; pi < n_patches; pi += gridDim.x) {
        // first patch: the halo and chunk 0 have landed once at most my 2 (D - 1) newest DMAs are in flight; later patches: their
        // halo was requested AFTER their first chunks (and the previous patch's stores are in the count): everything.
        // The barrier publishes all waves' parts.
        if (first) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * (D - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        first = false;
        NUNIF_C3D_STAMP(1);
        if constexpr (RELU_IN && KS == 1) {
            // the pre-activation ReLU once, in place, on the landed halo (11 items per thread) instead of on every B fragment of
            // every k-step (288 packed max per thread and patch: the chunk loop of the relu_in launches was 13.4 k ticks against
            // 8.6 k, profiles/r04_c3d_trace.txt)
#pragma unroll 1
            for (int u = 0; u < HDMA; ++u) {
                const int q = tid + 256 * u;
                if (q < HITEMS) {
                    f16x8 v = *reinterpret_cast<f16x8 *>(halo + 16 * q);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] > (f16)0.f ? v[e] : (f16)0.f;
                    *reinterpret_cast<f16x8 *>(halo + 16 * q) = v;
                }
            }
            __syncthreads();
        }
        const int cb = b, cty0 = ty0, ctx0 = tx0;
        f32x4 acc[NT][MF];
#pragma unroll 1
        for (int ph = 0; ph < NPASS * KS; ++ph) {                         // (a real loop: unrolled, hipcc keeps two accumulator sets alive)
        const int pass = KS > 1 ? 0 : ph, kh = KS > 1 ? ph : 0;
        if (KS == 1 || kh == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int f = 0; f < MF; ++f) acc[nt][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // the chunk loop is fully unrolled: chunk index, ring slot, tap and 32-channel part of every k-step are compile-time
#pragma unroll
        for (int c = 0; c < NCH3; ++c) {
            // chunk boundary.  My DMAs in flight: chunks c + 1 .. c + D - 1 (2 instructions each) at most -> chunk c has landed at
            // vmcnt(2 (D - 1)).  Behind the barrier every wave has finished reading chunk c - 1, whose slot chunk c + D now takes.
            // (The first chunk of a later pass is a boundary like any other; an earlier pass's stores also count in vmcnt, they
            // are older than the chunk and only make the wait longer.)
            if (KS > 1 && c == 0 && ph > 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // this half's halo (requested last)
            else if (c > 0 || ph > 0) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * (D - 1)) : "memory");
            {
                // D chunks ahead; wraps into the next pass / half / patch (the stream again, or its other half)
                const int cn = c + D < NCH3 ? c + D : c + D - NCH3;
                const int phn = c + D < NCH3 ? ph : (ph + 1 < NPASS * KS ? ph + 1 : 0);
                chunk_dma(cn, KS > 1 ? 0 : phn, (c + D) % kSlots, KS > 1 ? phn : 0);
            }
#pragma unroll
            for (int q = 0; q < KPC; ++q) {
                const int ks = c * KPC + q;
                if (ks < KSTEPS) {
                    const int tap = ks / CPT, cc = ks % CPT;
                    const f16x8 *wsrc = ring + ((c % kSlots) * kCH + q * NT) * 64 + lane;
                    f16x8 xq[MF];
#pragma unroll
                    for (int f = 0; f < MF; ++f) {
                        // part cc of the tap: segment 4 cc + grp; the swizzle is an XOR, so part 1 = address ^ 64
                        xq[f] = *reinterpret_cast<const f16x8 *>(halo + (faddr[f][tap] ^ (unsigned)(cc * 64)));
                        if constexpr (RELU_IN && KS > 1) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) xq[f][e] = xq[f][e] > (f16)0.f ? xq[f][e] : (f16)0.f;
                        }
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const f16x8 w = wsrc[nt * 64];
#pragma unroll
                        for (int f = 0; f < MF; ++f) acc[nt][f] = MFMA_16x16x32(w, xq[f], acc[nt][f]);
                    }
                }
            }
        }
        NUNIF_C3D_STAMP(2);
        if (ph == NPASS * KS - 1) {
            // every wave is done with the halo: the next patch's may overwrite it while this one's epilogue runs
            asm volatile("s_barrier" ::: "memory");
            if (pi + (int)gridDim.x < n_patches) {
                patch_of(pi + gridDim.x, b, ty0, tx0);
                halo_dma(b, ty0, tx0, 0);
            }
        } else if (KS > 1) {
            asm volatile("s_barrier" ::: "memory");                      // the next half of the same patch, no epilogue yet
            halo_dma(cb, cty0, ctx0, kh + 1);
            continue;
        }

        // ---- epilogue (as conv3_lds_kernel): bias, activation, residuals / image head ---------------------------------------
        const int ldo = g.ldo > 0 ? g.ldo : g.n_real;
        if constexpr (NT == 1) {
            if (g.out32) {
                // image head (planar fp32, `+ crop(add32)`, clamp; cunet.py:183-196): every add32 value is requested before any
                // is used (conv3_lds.hip)
                const int n0 = grp * 4;
                const float4 bv = *reinterpret_cast<const float4 *>(g.bias + n0);
                const float bvr[4] = {bv.x, bv.y, bv.z, bv.w};
                float addv[MF][4];
#pragma unroll
                for (int f = 0; f < MF; ++f) {
                    const int oy = min(cty0 + 2 * wave + (f >> 1), g.Ho - 1), ox = min(ctx0 + 16 * (f & 1) + r16, g.Wo - 1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = min(n0 + r, g.n_real - 1);
                        addv[f][r] = g.add32 ? g.add32[(((long)cb * g.n_real + n) * g.addH + oy + g.add_crop) * g.addW + ox + g.add_crop] : 0.f;
                    }
                }
#pragma unroll
                for (int f = 0; f < MF; ++f) {
                    const int oy = cty0 + 2 * wave + (f >> 1), ox = ctx0 + 16 * (f & 1) + r16;
                    if (oy >= g.Ho || ox >= g.Wo) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = n0 + r;
                        if (n >= g.n_real) continue;
                        float o = acc[0][f][r] + bvr[r];
                        if (g.act == 2) o = o >= 0.f ? o : o * g.slope;
                        else if (g.act == 3) o = fmaxf(o, 0.f);
                        o += addv[f][r];
                        if (g.clamp01) o = fminf(fmaxf(o, 0.f), 1.f);
                        g.out32[(((long)cb * g.n_real + n) * g.Ho + oy) * g.Wo + ox] = o;
                    }
                }
                continue;
            }
        }
        // NHWC fp16.  An accumulator lane holds 4 channels of a tile = 8 bytes, and a wave-wide store of those touches 16 pixels x 32
        // bytes; this epilogue (residual loads + stores) measured 14-21 k of a single-patch workgroup's ~30-50 k ticks
        // (profiles/r04_c3d_trace.txt).  Two adjacent tiles are turned into one run of 8 consecutive channels per lane (common.h
        // pair_to_run, here on the fp32 values so that `fp16(conv + res)` rounds exactly as before): 16-byte loads and stores,
        // 64 bytes per pixel, half the instructions.
        // (the two-pass and the two-half instantiations have no registers to spare for it: 11 / 26 spills)
        const bool runs = (NT % 2 == 0) && NPASS == 1 && KS == 1 && g.n_real % 32 == 0 && ldo % 8 == 0;
        if (runs) {
            if constexpr (NT % 2 == 0 && NPASS == 1 && KS == 1) {
#pragma unroll
                for (int np = 0; np < NT / 2; ++np) {
                    const int nb = (pass * NT + 2 * np) * 16;                       // first channel of the 32-channel pair
                    const float4 b0 = *reinterpret_cast<const float4 *>(g.bias + nb + grp * 4);
                    const float4 b1 = *reinterpret_cast<const float4 *>(g.bias + nb + 16 + grp * 4);
                    const float bb[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
                    // residual values are requested for two fragments at a time BEFORE the swaps that use them (a load behind a
                    // run-time `if` is waited for at once: 71 x vmcnt(0) in the first form of this epilogue)
#pragma unroll
                    for (int f0 = 0; f0 < MF; f0 += 2) {
                        long off[2];
                        bool live[2];
                        f16x8 rv1[2], rv2[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int f = f0 + u;
                            const int oy = cty0 + 2 * wave + (f >> 1), ox = ctx0 + 16 * (f & 1) + r16;
                            live[u] = oy < g.Ho && ox < g.Wo && nb < g.n_real;
                            off[u] = (((long)cb * g.Ho + min(oy, g.Ho - 1)) * g.Wo + min(ox, g.Wo - 1)) * ldo + nb + pair_run_channel(grp);
                        }
                        if (g.res) {
#pragma unroll
                            for (int u = 0; u < 2; ++u) rv1[u] = *reinterpret_cast<const f16x8 *>(g.res + off[u]);
                        }
                        if (g.res2) {
#pragma unroll
                            for (int u = 0; u < 2; ++u) rv2[u] = *reinterpret_cast<const f16x8 *>(g.res2 + off[u]);
                        }
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int f = f0 + u;
                            float v[2][4];
#pragma unroll
                            for (int t = 0; t < 2; ++t)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    float x = acc[2 * np + t][f][r] + bb[t][r];
                                    if (g.act == 2) x = x >= 0.f ? x : x * g.slope;
                                    else if (g.act == 3) x = fmaxf(x, 0.f);
                                    v[t][r] = x;
                                }
                            float run[8];                                           // channels nb + pair_run_channel(grp) + 0..7
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const u32x2 sw = lane16_swap(__builtin_bit_cast(unsigned, v[0][r]), __builtin_bit_cast(unsigned, v[1][r]));
                                const unsigned lo = sw[0], hi = sw[1];
                                run[r] = __builtin_bit_cast(float, lo);
                                run[4 + r] = __builtin_bit_cast(float, hi);
                            }
                            if (g.res) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) run[e] += (float)rv1[u][e];
                            }
                            if (g.res2) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) run[e] += (float)rv2[u][e];
                            }
                            f16x8 o;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = (f16)run[e];
                            if (live[u]) *reinterpret_cast<f16x8 *>(g.out + off[u]) = o;   // (after the swap: it needs every lane)
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n0 = (pass * NT + nt) * 16 + grp * 4;
                const float4 bv = *reinterpret_cast<const float4 *>(g.bias + n0);
    #pragma unroll
                for (int f = 0; f < MF; ++f) {
                    const int oy = cty0 + 2 * wave + (f >> 1), ox = ctx0 + 16 * (f & 1) + r16;
                    if (oy >= g.Ho || ox >= g.Wo || n0 >= g.n_real) continue;
                    float v[4] = {acc[nt][f][0] + bv.x, acc[nt][f][1] + bv.y, acc[nt][f][2] + bv.z, acc[nt][f][3] + bv.w};
                    if (g.act == 2) {
    #pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = v[r] >= 0.f ? v[r] : v[r] * g.slope;
                    } else if (g.act == 3) {
    #pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                    const long off = (((long)cb * g.Ho + oy) * g.Wo + ox) * ldo + n0;
                    if (g.res) {
                        const f16x4 rv = *reinterpret_cast<const f16x4 *>(g.res + off);
    #pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
                    }
                    if (g.res2) {
                        const f16x4 rv = *reinterpret_cast<const f16x4 *>(g.res2 + off);
    #pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
                    }
                    *reinterpret_cast<f16x4 *>(g.out + off) = (f16x4){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                }
            }
        }
        NUNIF_C3D_STAMP(3);
        }   // pass
    }
    // drain the two weight chunks the last trip requested for a patch that does not exist (LDS must not be written after exit)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NUNIF_C3D_STAMP(4);
}

static inline int conv3_dma_enabled() { const char *e = getenv("NUNIF_CONV3_DMA"); return e ? atoi(e) : 1; }

bool conv3_dma_applies(const ConvArgs &g) {
    const int pad = (g.zpad || g.rpad) ? 1 : 0;
    const int nt = g.N / 16;
    if (!conv3_dma_enabled()) return false;
    if (!(g.kh == 3 && g.kw == 3 && g.stride == 1) || g.a2 || g.cmaj || (g.out32 && nt != 1) || g.N % 16 != 0) return false;
    if (g.Ho != g.Hi + 2 * pad - 2 || g.Wo != g.Wi + 2 * pad - 2 || g.zpad > 1 || g.rpad > 1 || (g.zpad && g.rpad)) return false;
    // Cin = 128 with 64 outputs: two K halves over one halo space (NUNIF_CONV3_DMA_KSPLIT=0: conv3_lds_kernel as before)
    // (round 5: Cin = 256 as four quarters: cunet's 256 -> 128 conv arrives as two 64-output slices, cunet.cpp run_conv)
    const bool ksplit = (g.Cin == 128 || g.Cin == 256) && nt == 4 &&
                        !(getenv("NUNIF_CONV3_DMA_KSPLIT") && atoi(getenv("NUNIF_CONV3_DMA_KSPLIT")) == 0);
    if (!ksplit && (!(g.Cin == 32 || g.Cin == 64) || !(nt == 1 || nt == 2 || nt == 4 || (nt == 8 && g.Cin == 64)))) return false;
    const long n_patches = (long)g.B * ((g.Ho + kTH - 1) / kTH) * ((g.Wo + kTW - 1) / kTW);
    // tiny launches keep the resident-weight form of conv3_lds_kernel.  Threshold sweep on the depth net (4 x 1080p, ViT-S):
    // 512 -> 2 033 fps, 256 -> 2 031, 96 -> 2 049, 24 -> 2 070 (one patch per workgroup from 24 to 512 patches)
    static const long min_patches = getenv("NUNIF_CONV3_DMA_MIN") ? atol(getenv("NUNIF_CONV3_DMA_MIN")) : 24;
    // (the two-half form sums in another order than conv3_lds_kernel: it takes EVERY launch of its shape, so that a result does not
    //  depend on how many tiles share a launch — tests/test_cunet.py renders with minibatches of 4 and 9 and compares bits)
    return (ksplit || n_patches > min_patches) && n_patches < (1L << 30) && (long)g.B * g.Hi * g.Wi * g.Cin < (1L << 40);
}

template <int NT, int CIN, bool RELU_IN, int NPASS = 1, int D = 2, int KS = 1>
static int launch_c3d(const ConvArgs &g, hipStream_t s, const char *name) {
    // the last DMA instruction of a halo is a full 1 KiB whatever the item count: round the halo up to a multiple of 64 items
    const size_t smem = (size_t)(D + 1) * kCH * 1024 + (size_t)((kHaloPix * (CIN / 8) + 63) / 64) * 1024;
    const long M = (long)g.B * g.Ho * g.Wo;
    ProfScope ps(name, s, 2.0 * (double)M * 9.0 * g.Cin * g.n_real, (double)M * (g.Cin + g.n_real) * 2.0);
    static bool configured = false;
    if (!configured) {
        NUNIF_HIP_CHECK(hipFuncSetAttribute((const void *)conv3_dma_kernel<NT, CIN, RELU_IN, NPASS, D, KS>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        configured = true;
    }
    const long n_patches = (long)g.B * ((g.Ho + kTH - 1) / kTH) * ((g.Wo + kTW - 1) / kTW);
    const unsigned grid = (unsigned)std::min<long>(n_patches, 512);         // two persistent workgroups per CU
    conv3_dma_kernel<NT, CIN, RELU_IN, NPASS, D, KS><<<grid, 256, smem, s>>>(g, (int)n_patches);
    NUNIF_LAUNCH_CHECK();
#ifdef NUNIF_C3D_TRACE
    if (NT == 4 && CIN == 64 && n_patches <= 512) {
        static int n = 0;
        if (++n > 60 && n <= 73) {
            static unsigned long long host[1024 * 8];
            NUNIF_HIP_CHECK(hipStreamSynchronize(s));
            NUNIF_HIP_CHECK(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_c3d_trace), sizeof(host)));
            const int nb = (int)std::min<long>(grid, 1024);
            double avg[5] = {0};
            for (int b = 0; b < nb; ++b)
                for (int i = 0; i < 5; ++i) avg[i] += (double)(long long)(host[b * 8 + i] - host[b * 8]) / nb;
            fprintf(stderr, "[c3d trace] D %d patches %ld relu %d res %d res2 %d: %.0f %.0f %.0f %.0f %.0f\n", D, n_patches, (int)RELU_IN, g.res != nullptr,
                    g.res2 != nullptr, avg[0], avg[1], avg[2], avg[3], avg[4]);
        }
    }
#endif
    return NUNIF_HIP_OK;
}

int launch_conv3_dma(const ConvArgs &g, hipStream_t s) {
    const int nt = g.N / 16;
    if (g.Cin == 64) {
        if (nt == 8) return g.relu_in ? launch_c3d<8, 64, true, 2>(g, s, "conv3_dma_kernel<8,64>") : launch_c3d<8, 64, false, 2>(g, s, "conv3_dma_kernel<8,64>");
        if (nt == 1) return g.relu_in ? launch_c3d<1, 64, true>(g, s, "conv3_dma_kernel<1,64>") : launch_c3d<1, 64, false>(g, s, "conv3_dma_kernel<1,64>");
        if (nt == 4) return g.relu_in ? launch_c3d<4, 64, true>(g, s, "conv3_dma_kernel<4,64>") : launch_c3d<4, 64, false>(g, s, "conv3_dma_kernel<4,64>");
        if (nt == 2) return g.relu_in ? launch_c3d<2, 64, true>(g, s, "conv3_dma_kernel<2,64>") : launch_c3d<2, 64, false>(g, s, "conv3_dma_kernel<2,64>");
    } else if (g.Cin == 128) {
        if (nt == 4) return g.relu_in ? launch_c3d<4, 64, true, 1, 2, 2>(g, s, "conv3_dma_kernel<4,128>") : launch_c3d<4, 64, false, 1, 2, 2>(g, s, "conv3_dma_kernel<4,128>");
    } else if (g.Cin == 256) {
        if (nt == 4) return g.relu_in ? launch_c3d<4, 64, true, 1, 2, 4>(g, s, "conv3_dma_kernel<4,256>") : launch_c3d<4, 64, false, 1, 2, 4>(g, s, "conv3_dma_kernel<4,256>");
    } else if (g.Cin == 32) {
        if (nt == 1) return g.relu_in ? launch_c3d<1, 32, true>(g, s, "conv3_dma_kernel<1,32>") : launch_c3d<1, 32, false>(g, s, "conv3_dma_kernel<1,32>");
        if (nt == 4) return g.relu_in ? launch_c3d<4, 32, true>(g, s, "conv3_dma_kernel<4,32>") : launch_c3d<4, 32, false>(g, s, "conv3_dma_kernel<4,32>");
        if (nt == 2) return g.relu_in ? launch_c3d<2, 32, true>(g, s, "conv3_dma_kernel<2,32>") : launch_c3d<2, 32, false>(g, s, "conv3_dma_kernel<2,32>");
    }
    set_error("conv3_dma: Cin=%d Cout=%d unsupported", g.Cin, g.N);
    return NUNIF_HIP_EUNSUPPORTED;
}

}  // namespace nunif
