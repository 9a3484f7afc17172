// The MLP of a DINOv2 ViT-S block in ONE kernel, for gfx950:   t += ls2 * fc2(gelu(fc1(LayerNorm2(t))))
// (Depth-Anything's encoder, iw3/depth_anything_model.py:113-119 -> the hub network's Block.forward; embed 384, hidden 1536).
//
// Why.  At the reference's own benchmark shape (4 x 1080p -> 5 492 tokens) fc1 and fc2 were two launches of 2 064 and 516
// workgroups that each live a few microseconds: 22 + 27 us per layer for 13 GFLOP (profiles/r03d_kernel_stats_iw3_sched.csv), most
// of it launch ramp, operand latency and the round trip of the 1536-wide hidden rows through HBM (2 x 16.9 MB per layer).
// Here ONE workgroup owns 32 tokens from their 384-wide rows to their updated 384-wide rows:
//   * 8 waves split the OUTPUT channels, so every weight fragment is private to one wave and goes straight from L2 into its
//     registers (no LDS, no barrier for weights): a ring of 24 fragments (96 registers) per wave, each slot reloaded right after
//     its MFMAs with the fragment the wave needs 24 fragments later, as inline asm with hand-counted s_waitcnt vmcnt(22 / 21)
//     (left to hipcc, the reloads sink to within two loads of their use);
//   * phase B (fc1): the 32 rows sit in LDS as B fragments (LDS-DMA'd from HBM), wave w computes hidden pairs w, w + 8, ...;
//     norm2 is folded into fc1 as in gemm_ws_kernel<LNF>: raw rows, then r (W x) - r mu wsum + b with mu / r from
//     the 12 partial sums per token that attn.proj wrote; GELU; the hidden tile pair (32 channels) is written to LDS as ONE
//     chained-order B fragment of fc2 (accumulator tiles ARE k-slots of the next contraction, swin_block_tail.hip);
//   * phase C (fc2): wave w computes output tiles 3 w .. 3 w + 2 over all 48 hidden fragments on top of bias + residual rows
//     (still in LDS), stores, and writes the per-token partial sums of what it stored for the NEXT block's norm1 (12 per token,
//     the format qkv consumes).
// The hidden activation never leaves the CU.  LDS: 24 (rows) + 14 (biases) + 96 (hidden) + 9 (statistics) KiB, one workgroup per
// CU; 172 workgroups for 4 x 1080p, one pass.  This form is the fallback; the hidden-split pair below is what runs while its grid
// fits the chip.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "swin_gelu.h"
#include "swin_kernels.h"

namespace nunif {

#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

namespace {
constexpr int kD = 384, kH = 1536;
constexpr int kKS1 = kD / 32;            // 12 k-steps of fc1
constexpr int kKS2 = kH / 32;            // 48 k-steps of fc2 (= hidden tile pairs)
constexpr int kNT2 = kD / 16;            // 24 output tiles
constexpr int kWavesM = 8, kTok = 32;
constexpr int kParts = 12;               // statistics partials per token (32-channel pairs of a 384-wide row)
constexpr int kRing = 24;                // weight fragments in flight per wave (96 registers)
// LDS map (bytes).  The first 38 KiB are filled by 38 LDS-DMA items of 1 KiB each, item i at i * 1024.
constexpr int kYOff = 0, kYBytes = kKS1 * 2 * 1024;                      // rows as B fragments [ks 12][mt 2][64]
constexpr int kB1Off = kYOff + kYBytes;                                  // fc1 bias, 6 KiB
constexpr int kWsOff = kB1Off + kH * 4;                                  // fc1 row sums (LayerNorm fold), 6 KiB
constexpr int kB2Off = kWsOff + kH * 4;                                  // fc2 bias (ls2 folded), 1.5 KiB in a 2 KiB slot
constexpr int kDmaItems = (kB2Off + 2048) / 1024;                        // 38
constexpr int kHOff = kB2Off + 2048, kHBytes = kKS2 * 2 * 1024;          // hidden as chained-order B fragments [ks 48][mt 2][64]
constexpr int kStOff = kHOff + kHBytes;                                  // [32][12] float2 statistics from attn.proj
constexpr int kMurOff = kStOff + kTok * kParts * 8;                      // [32] (mu, r)
constexpr int kOpOff = kMurOff + kTok * 8;                               // [32][24] float2 partials of the stored rows
constexpr int kSmemM = kOpOff + kTok * kNT2 * 8;

// A weight fragment goes L2 -> registers with no compiler bookkeeping: hipcc sinks its own loads to within two loads of their use
// (s_waitcnt vmcnt(1..2) before every MFMA pair, profiles/r04_isa_notes.md), which leaves ~2 KiB in flight per wave.  These are counted by hand instead; every other vector
// memory operation of the kernel is issued before the first one or after the last.
__device__ __forceinline__ void wload(f16x8 &dst, unsigned voff, const void *sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase));
}
#define NUNIF_VMW2(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(a), "+v"(b)); break
#define NUNIF_VMW3(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(a), "+v"(b), "+v"(c)); break
__device__ __forceinline__ void vm_wait2(int n, f16x8 &a, f16x8 &b) {
    switch (n) { NUNIF_VMW2(22); default: asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b)); }
}
__device__ __forceinline__ void vm_wait3(int n, f16x8 &a, f16x8 &b, f16x8 &c) {
    switch (n) {
        NUNIF_VMW3(21); NUNIF_VMW3(18); NUNIF_VMW3(15); NUNIF_VMW3(12); NUNIF_VMW3(9); NUNIF_VMW3(6); NUNIF_VMW3(3);
        default: asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c));
    }
}
}  // namespace

__global__ void __launch_bounds__(kWavesM * 64) da_mlp_kernel(DaMlpArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_m[];
    const f16x8 *yl = reinterpret_cast<const f16x8 *>(smem_m + kYOff);
    const float *b1l = reinterpret_cast<const float *>(smem_m + kB1Off);
    const float *wsl = reinterpret_cast<const float *>(smem_m + kWsOff);
    const float *b2l = reinterpret_cast<const float *>(smem_m + kB2Off);
    f16x8 *hl = reinterpret_cast<f16x8 *>(smem_m + kHOff);
    float2 *stl = reinterpret_cast<float2 *>(smem_m + kStOff);
    float2 *mur = reinterpret_cast<float2 *>(smem_m + kMurOff);
    float2 *opl = reinterpret_cast<float2 *>(smem_m + kOpOff);

    const int tid = threadIdx.x;
    const int lane = tid & 63, r16 = lane & 15, grp = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long m0 = (long)blockIdx.x * kTok;
    const unsigned voff = (unsigned)lane * 16u;

    // fragment (tile T, k-step ks) of fc1 / fc2 as packed by the host: [tile][ks][64 lanes] x 16 bytes
    // Which fragments a wave owns is chosen for the L2: the 16 channels of an XCD's L2 interleave at 4 KiB, and the 21 workgroups
    // of an XCD walk the same addresses in step, so the 16 (fc1) / 24 (fc2) streams of a workgroup's 8 waves must sit on DIFFERENT
    // channels at every instant.  fc1: wave w owns hidden pairs w, w + 8, ... (a pair = 24 KiB = 6 channels on, tiles 3 apart);
    // fc2: the host packs [k-step / 4][tile 24][k-step % 4], one tile's four k-steps = 4 KiB = one channel, wave w tiles 3 w + n.
    const unsigned char *w1w = reinterpret_cast<const unsigned char *>(a.w1) + (long)(2 * wave) * kKS1 * 1024;
    const unsigned char *w2w = reinterpret_cast<const unsigned char *>(a.w2c) + (long)(3 * wave) * 4096;
    auto w1_at = [&](int pair, int e, int ks) { return w1w + ((2 * kWavesM * pair + e) * kKS1 + ks) * 1024; };
    auto w2_at = [&](int n, int ks) { return w2w + (((ks >> 2) * kNT2 + n) * 4 + (ks & 3)) * 1024; };

    // ---- the ring's first tenants: pair 0 of fc1, slot j = 2 ks + e (use order) ---------------------------------------------
    f16x8 wq[kRing];
#pragma unroll
    for (int j = 0; j < kRing; ++j) wload(wq[j], voff, w1_at(0, j & 1, j >> 1));

    // ---- the 32 rows, the bias vectors -> LDS by DMA (38 items over 8 waves); the rows' statistics -> LDS ---------------------
    {
        const unsigned lds0 = (unsigned)reinterpret_cast<size_t>(smem_m);
#pragma unroll
        for (int u = 0; u < (kDmaItems + kWavesM - 1) / kWavesM; ++u) {
            const int i = wave + kWavesM * u;
            if (i < kDmaItems) {
                const void *src;
                if (i < 2 * kKS1) {
                    const int ks = i >> 1, mt = i & 1;                          // row fragment (ks, mt)
                    long m = m0 + 16 * mt + r16;
                    m = m < a.M ? m : a.M - 1;
                    src = a.t + m * kD + 32 * ks + 8 * grp;
                } else if (i < 2 * kKS1 + 6) {
                    src = a.b1 + (i - 2 * kKS1) * 256 + lane * 4;
                } else if (i < 2 * kKS1 + 12) {
                    src = a.ws1 + (i - 2 * kKS1 - 6) * 256 + lane * 4;
                } else {
                    const int q = (i - 2 * kKS1 - 12) * 256 + lane * 4;         // 384 floats in two items: the tail is clamped
                    src = a.b2 + (q < kD ? q : kD - 4);
                }
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds0 + (unsigned)i * 1024)
                             : "memory");
            }
        }
        if (tid < kTok * kParts) {
            long m = m0 + tid / kParts;
            m = m < a.M ? m : a.M - 1;
            stl[tid] = a.stats_in[m * kParts + tid % kParts];
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid < kTok) {
        // mu, r of a token exactly as gemm_ws_kernel<LNF> takes them (variance = E[x^2] - mu^2 from fp32 partials)
        float su = 0.f, sq = 0.f;
#pragma unroll
        for (int q = 0; q < kParts; ++q) { const float2 v = stl[tid * kParts + q]; su += v.x; sq += v.y; }
        const float mu = su * (1.0f / kD);
        const float var = fmaxf(sq * (1.0f / kD) - mu * mu, 0.f);
        mur[tid] = make_float2(mu, rsqrtf(var + a.ln_eps));
    }
    __syncthreads();

    // ---- phase B: hidden tiles 12 w .. 12 w + 11, a pair (= one 32-channel fragment of fc2's contraction) at a time -----------
    {
        float2 mr[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) mr[mt] = mur[16 * mt + r16];
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            const int T0 = 2 * (wave + kWavesM * p);
            f32x4 acc[2][2];
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[e][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < kKS1; ++ks) {
                const f16x8 y0 = yl[(ks * 2 + 0) * 64 + lane], y1 = yl[(ks * 2 + 1) * 64 + lane];
                vm_wait2(22, wq[2 * ks], wq[2 * ks + 1]);                        // the two oldest of <= 24 in flight
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    acc[e][0] = MFMA_16x16x32(wq[2 * ks + e], y0, acc[e][0]);
                    acc[e][1] = MFMA_16x16x32(wq[2 * ks + e], y1, acc[e][1]);
                }
                // the slots' next tenants: the same position of the next pair; in the last pair, fc2's first 24 fragments
                // (slot j = 3 k + n, again the order of use), so the ring is full across the phase boundary
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int j = 2 * ks + e;
                    if (p < 5) wload(wq[j], voff, w1_at(p + 1, e, ks));
                    else wload(wq[j], voff, w2_at(j % 3, j / 3));
                }
            }
            // norm2 folded in: r (W x) - r mu wsum + b; GELU; the pair is one chained-order fragment of fc2's B operand
            float4 bv[2], wv[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                bv[e] = *reinterpret_cast<const float4 *>(b1l + (T0 + e) * 16 + 4 * grp);
                wv[e] = *reinterpret_cast<const float4 *>(wsl + (T0 + e) * 16 + 4 * grp);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float r = mr[mt].y, rm = mr[mt].y * mr[mt].x;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    acc[e][mt][0] = fmaf(r, acc[e][mt][0], fmaf(-rm, wv[e].x, bv[e].x));
                    acc[e][mt][1] = fmaf(r, acc[e][mt][1], fmaf(-rm, wv[e].y, bv[e].y));
                    acc[e][mt][2] = fmaf(r, acc[e][mt][2], fmaf(-rm, wv[e].z, bv[e].z));
                    acc[e][mt][3] = fmaf(r, acc[e][mt][3], fmaf(-rm, wv[e].w, bv[e].w));
                }
                hl[((T0 >> 1) * 2 + mt) * 64 + lane] = gelu8t(acc[0][mt], acc[1][mt]);
            }
        }
    }
    __syncthreads();

    // ---- phase C: output tiles 3 w .. 3 w + 2 over the 48 hidden fragments ------------------------------------------------------
    {
        constexpr int NW = 3, GK = 8;                                            // tiles per wave, k-steps the ring holds
        // accumulators start from bias + residual row; the row is still in LDS (fp16 -> fp32, exact)
        f32x4 acc[NW][2];
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const int N = NW * wave + n;
            const float4 bb = *reinterpret_cast<const float4 *>(b2l + N * 16 + 4 * grp);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                // channels 16 N + 4 grp .. + 3 of token 16 mt + r16: fragment (N / 2, mt), lane group 2 (N & 1) + grp / 2, half grp & 1
                const f16x4 rv = *reinterpret_cast<const f16x4 *>(
                    reinterpret_cast<const f16 *>(yl + ((N >> 1) * 2 + mt) * 64 + (2 * (N & 1) + (grp >> 1)) * 16 + r16) + 4 * (grp & 1));
                acc[n][mt] = (f32x4){bb.x + (float)rv[0], bb.y + (float)rv[1], bb.z + (float)rv[2], bb.w + (float)rv[3]};
            }
        }
#pragma unroll
        for (int ks = 0; ks < kKS2; ++ks) {
            const int k = ks % GK;
            const f16x8 h0 = hl[(ks * 2 + 0) * 64 + lane], h1 = hl[(ks * 2 + 1) * 64 + lane];
            vm_wait3(ks + GK < kKS2 ? 21 : 21 - 3 * (ks + GK - kKS2), wq[3 * k], wq[3 * k + 1], wq[3 * k + 2]);
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                acc[n][0] = MFMA_16x16x32(wq[3 * k + n], h0, acc[n][0]);
                acc[n][1] = MFMA_16x16x32(wq[3 * k + n], h1, acc[n][1]);
            }
            if (ks + GK < kKS2) {
#pragma unroll
                for (int n = 0; n < NW; ++n) wload(wq[3 * k + n], voff, w2_at(n, ks + GK));   // the same tile, 8 k-steps on
            }
        }
        // store the rows; per-tile partial sums of the fp16 values as stored (for the next block's norm1)
#pragma unroll
        for (int n = 0; n < NW; ++n)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const long m = m0 + 16 * mt + r16;
                const f16x4 o = {(f16)acc[n][mt][0], (f16)acc[n][mt][1], (f16)acc[n][mt][2], (f16)acc[n][mt][3]};
                if (m < a.M) *reinterpret_cast<f16x4 *>(a.t + m * kD + (NW * wave + n) * 16 + 4 * grp) = o;
                if (a.stats_out) {
                    const float v0 = (float)o[0], v1 = (float)o[1], v2 = (float)o[2], v3 = (float)o[3];
                    float su = (v0 + v1) + (v2 + v3), sq = (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
                    su += __shfl_xor(su, 16); sq += __shfl_xor(sq, 16);
                    su += __shfl_xor(su, 32); sq += __shfl_xor(sq, 32);
                    if (grp == 0) opl[(16 * mt + r16) * kNT2 + NW * wave + n] = make_float2(su, sq);
                }
            }
    }
    if (a.stats_out) {
        __syncthreads();
        if (tid < kTok * kParts) {
            const int tok = tid / kParts, q = tid % kParts;
            const float2 u = opl[tok * kNT2 + 2 * q], v = opl[tok * kNT2 + 2 * q + 1];
            if (m0 + tok < a.M) a.stats_out[(m0 + tok) * kParts + q] = make_float2(u.x + v.x, u.y + v.y);
        }
    }
}

// ---- the same MLP with the HIDDEN dimension split over a pair of workgroups ---------------------------------------------------
// What bounds da_mlp_kernel: every CU pulls all 2.36 MB of weights through its own L1 at 30-50 B/clk (172 CUs re-reading the same
// lines from 8 L2s), behind a prologue in which all CUs burst at once — and the chip holds one such workgroup per CU, so the
// kernel's time is ONE workgroup's chain (38 us at 4 x 1080p; the two GEMM launches it replaces: 49).
// Here a workgroup owns 64 tokens and ONE HALF of the 1536 hidden channels: half of fc1's rows, half of fc2's contraction, 1.18 MB
// per CU for the same 576 MFMAs per wave.  Blocks 2 g and 2 g + 1 are the halves of token group g.  In fc2 a block's waves 0-3
// compute the partial sums of the PARTNER's 192 output channels (they are the older waves of their SIMDs and win its arbitration:
// they finish first, s_setprio changes nothing), store them fp32 in accumulator layout to scratch with write-through stores and
// raise flag[block] (the last of the four does); waves 4-7 compute the block's own 192 channels on top of bias + residual, poll
// the partner's flag, add its partial, and the finished rows go through LDS so that all 8 waves store whole 384-byte runs.  The
// receiver lowers the flag again: nothing depends on a launch counter, a replayed graph would behave the same.  Both blocks must
// be resident at once: the launcher takes this kernel only while the grid fits the chip (HIP promises no dispatch order).  With
// other kernels on the device at the same time (the depth net's own side streams, the stereo stream of the frame pipeline) the grid
// may be admitted piecemeal: what keeps it live then is that a kernel's workgroups are dispatched in index order (observed, not
// promised), so at most ONE pair straddles the dispatch frontier and every other waiting block's partner is running — none of the
// co-running kernels waits for anything.  The
// hand-off is the {sc0 sc1 stores, relaxed agent flag, sc0 sc1 loads} form (MI355X_MICROARCH, inter-workgroup visibility).
// Every vector memory operation is inline asm and counted by hand: the prologue is LDS-DMA (rows, biases, statistics: 62 KiB) with
// the first 24 weight fragments issued behind it, so that s_waitcnt vmcnt(24) releases the rows while the weights are in flight.
// Phases of one workgroup (s_memtime, -DNUNIF_MLP_TRACE, profiles/r04_mlp_trace.txt), 49.5 k ticks = 29 us per launch:
//   prologue 8.4 k | statistics 0.8 | fc1 + GELU 17.6 | barrier 2.3 | fc2 15.1 (senders 8.5) | flag 0.8 | partial in 2.2 | rows out 2.5
// SQ counters (profiles/r04_mlp_sq.txt): MFMA busy 18.4 k cycles per SIMD of ~64 k; 66 % of wave-cycles in s_waitcnt — on the
// weight ring: 64 tokens per fragment want 64 B/clk/CU at full MFMA rate, the L2s deliver about 40 with every CU asking.
namespace {
constexpr int kTok2 = 64, kMT2 = 4, kPairsHalf = kKS2 / 2;                // 24 hidden pairs (= fc2 k-steps) per half
constexpr int kY2Bytes = kKS1 * kMT2 * 1024;                             // 48 KiB of rows
constexpr int kB1Off2 = kY2Bytes, kWsOff2 = kB1Off2 + 3072, kB2Off2 = kWsOff2 + 3072, kSt2Off = kB2Off2 + 2048;
constexpr int kDmaItems2 = (kSt2Off + kTok2 * kParts * 8) / 1024;        // 62 items of 1 KiB
constexpr int kH2Off = kDmaItems2 * 1024, kH2Bytes = kPairsHalf * kMT2 * 1024;   // 96 KiB of hidden fragments
constexpr int kMur2Off = kH2Off + kH2Bytes;
constexpr int kSentOff = kMur2Off + kTok2 * 8;                               // count of sending waves that have stored
constexpr int kSmemM2 = kSentOff + 16;
constexpr int kOwnTiles = kNT2 / 2;                                      // 12 output tiles per half
constexpr int kORowBytes = kD + 16;                                      // finished rows in LDS: 192 fp16 + a 16-byte pad (banks)
constexpr int kPartialBytes = 2 * kOwnTiles * kMT2 * 1024;               // per token group: 48 KiB from each half
static_assert(kSmemM2 <= 160 * 1024, "LDS");
}  // namespace

// Phase timestamps of the split kernel (-DNUNIF_MLP_TRACE builds only): waves 0 and 4 of every workgroup stamp s_memtime.
#ifdef NUNIF_MLP_TRACE
__device__ unsigned long long g_mlp_trace[512 * 32];
#define NUNIF_MLP_STAMP(i) do { if (lane == 0 && (wave & 3) == 0 && blockIdx.x < 512) g_mlp_trace[blockIdx.x * 32 + (wave >> 2) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define NUNIF_MLP_STAMP(i)
#endif

__global__ void __launch_bounds__(kWavesM * 64) da_mlp_split_kernel(DaMlpArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_m[];
    const f16x8 *yl = reinterpret_cast<const f16x8 *>(smem_m);
    const float *b1l = reinterpret_cast<const float *>(smem_m + kB1Off2);
    const float *wsl = reinterpret_cast<const float *>(smem_m + kWsOff2);
    const float *b2l = reinterpret_cast<const float *>(smem_m + kB2Off2);
    f16x8 *hl = reinterpret_cast<f16x8 *>(smem_m + kH2Off);
    const float2 *stl = reinterpret_cast<const float2 *>(smem_m + kSt2Off);
    float2 *mur = reinterpret_cast<float2 *>(smem_m + kMur2Off);
    int *sent = reinterpret_cast<int *>(smem_m + kSentOff);

    const int tid = threadIdx.x;
    const int lane = tid & 63, r16 = lane & 15, grp = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x >> 1, half = blockIdx.x & 1;
    const long m0 = (long)g * kTok2;
    const unsigned voff = (unsigned)lane * 16u;
    NUNIF_MLP_STAMP(0);

    // fc1: this half's tiles are 48 half .. 48 half + 47; wave w owns the local pairs w, w + 8, w + 16 (L2 channels: as above).
    // fc2: this half's k-steps are 24 half .. + 23 = blocks 6 half .. 6 half + 5 of the packed [ks / 4][tile][ks % 4] order; waves
    // 0-3 take the partner's tiles, waves 4-7 this block's own, three each.
    const int role = wave >> 2, wl = wave & 3;                                // role 0: send, role 1: finish
    const int tile0 = kOwnTiles * (role ? half : 1 - half) + 3 * wl;          // first of the wave's three output tiles
    const unsigned char *w1w = reinterpret_cast<const unsigned char *>(a.w1) + (long)(48 * half + 2 * wave) * kKS1 * 1024;
    const unsigned char *w2w = reinterpret_cast<const unsigned char *>(a.w2c) + (long)(6 * half * kNT2 + tile0) * 4096;
    auto w1_at = [&](int pair, int e, int ks) { return w1w + ((2 * kWavesM * pair + e) * kKS1 + ks) * 1024; };
    auto w2_at = [&](int n, int k) { return w2w + (((k >> 2) * kNT2 + n) * 4 + (k & 3)) * 1024; };

    // ---- prologue: rows, bias vectors, statistics -> LDS by DMA (62 items over 8 waves), then the ring's first tenants -----------
    {
        const unsigned lds0 = (unsigned)reinterpret_cast<size_t>(smem_m);
#pragma unroll
        for (int u = 0; u < (kDmaItems2 + kWavesM - 1) / kWavesM; ++u) {
            const int i = wave + kWavesM * u;
            const void *src;
            if (i < kKS1 * kMT2) {
                const int ks = i >> 2, mt = i & 3;                              // row fragment (ks, mt)
                long m = m0 + 16 * mt + r16;
                m = m < a.M ? m : a.M - 1;
                src = a.t + m * kD + 32 * ks + 8 * grp;
            } else if (i < kKS1 * kMT2 + 3) {
                src = a.b1 + half * (kH / 2) + (i - kKS1 * kMT2) * 256 + lane * 4;
            } else if (i < kKS1 * kMT2 + 6) {
                src = a.ws1 + half * (kH / 2) + (i - kKS1 * kMT2 - 3) * 256 + lane * 4;
            } else if (i < kKS1 * kMT2 + 8) {
                const int q = (i - kKS1 * kMT2 - 6) * 256 + lane * 4;
                src = a.b2 + (q < kD ? q : kD - 4);
            } else {
                const int e = (i - kKS1 * kMT2 - 8) * 128 + 2 * lane;           // two of a token's 12 (sum, sum of squares) pairs
                long m = m0 + e / kParts;
                m = m < a.M ? m : a.M - 1;
                src = a.stats_in + m * kParts + e % kParts;
            }
            // waves 6, 7 have one item less; their counted wait below still covers what they issued
            if (i < kDmaItems2)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds0 + (unsigned)i * 1024)
                             : "memory");
        }
    }
    f16x8 wq[kRing];
#pragma unroll
    for (int j = 0; j < kRing; ++j) wload(wq[j], voff, w1_at(0, j & 1, j >> 1));
    asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory");
    NUNIF_MLP_STAMP(1);
    if (tid < kTok2) {
        float su = 0.f, sq = 0.f;
#pragma unroll
        for (int q = 0; q < kParts; ++q) { const float2 v = stl[tid * kParts + q]; su += v.x; sq += v.y; }
        const float mu = su * (1.0f / kD);
        const float var = fmaxf(sq * (1.0f / kD) - mu * mu, 0.f);
        mur[tid] = make_float2(mu, rsqrtf(var + a.ln_eps));
        if (tid == 0) *sent = 0;
    }
    __syncthreads();
    NUNIF_MLP_STAMP(2);

    // ---- phase B: three hidden pairs per wave over 64 tokens ---------------------------------------------------------------------
    {
        float2 mr[kMT2];
#pragma unroll
        for (int mt = 0; mt < kMT2; ++mt) mr[mt] = mur[16 * mt + r16];
        f16x8 yb[2][kMT2];
#pragma unroll
        for (int mt = 0; mt < kMT2; ++mt) yb[0][mt] = yl[mt * 64 + lane];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int Pl = wave + kWavesM * p;                                  // local pair = local k-step of fc2
            f32x4 acc[2][kMT2];
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int mt = 0; mt < kMT2; ++mt) acc[e][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < kKS1; ++ks) {
                // the NEXT k-step's rows are read before this one's MFMAs; the rows are the same for every pair.  (Same-box A/B
                // against reading and waiting at once: no difference — the waves wait on the weight ring, not on LDS.)
                f16x8 *yc = yb[ks & 1], *yn = yb[(ks & 1) ^ 1];
#pragma unroll
                for (int mt = 0; mt < kMT2; ++mt) yn[mt] = yl[(((ks + 1) % kKS1) * kMT2 + mt) * 64 + lane];
                vm_wait2(22, wq[2 * ks], wq[2 * ks + 1]);
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int mt = 0; mt < kMT2; ++mt) acc[e][mt] = MFMA_16x16x32(wq[2 * ks + e], yc[mt], acc[e][mt]);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int j = 2 * ks + e;
                    if (p < 2) wload(wq[j], voff, w1_at(p + 1, e, ks));
                    else wload(wq[j], voff, w2_at(j % 3, j / 3));
                }
            }
            float4 bv[2], wv[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                bv[e] = *reinterpret_cast<const float4 *>(b1l + (2 * Pl + e) * 16 + 4 * grp);
                wv[e] = *reinterpret_cast<const float4 *>(wsl + (2 * Pl + e) * 16 + 4 * grp);
            }
#pragma unroll
            for (int mt = 0; mt < kMT2; ++mt) {
                const float r = mr[mt].y, rm = mr[mt].y * mr[mt].x;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    acc[e][mt][0] = fmaf(r, acc[e][mt][0], fmaf(-rm, wv[e].x, bv[e].x));
                    acc[e][mt][1] = fmaf(r, acc[e][mt][1], fmaf(-rm, wv[e].y, bv[e].y));
                    acc[e][mt][2] = fmaf(r, acc[e][mt][2], fmaf(-rm, wv[e].z, bv[e].z));
                    acc[e][mt][3] = fmaf(r, acc[e][mt][3], fmaf(-rm, wv[e].w, bv[e].w));
                }
                hl[(Pl * kMT2 + mt) * 64 + lane] = gelu8t(acc[0][mt], acc[1][mt]);
            }
        }
    }
    NUNIF_MLP_STAMP(3);
    // ---- phase C: three output tiles per wave over this half's 24 hidden fragments ------------------------------------------------
    constexpr int NW = 3, GK = 8;
    f32x4 acc[NW][kMT2];
#pragma unroll
    for (int n = 0; n < NW; ++n) {
        const int N = tile0 + n;
        const float4 bb = *reinterpret_cast<const float4 *>(b2l + N * 16 + 4 * grp);
#pragma unroll
        for (int mt = 0; mt < kMT2; ++mt) {
            if (role) {     // bias + residual row ride on the waves that finish; read BEFORE the barrier: the rows' LDS is free after it
                const f16x4 rv = *reinterpret_cast<const f16x4 *>(
                    reinterpret_cast<const f16 *>(yl + ((N >> 1) * kMT2 + mt) * 64 + (2 * (N & 1) + (grp >> 1)) * 16 + r16) + 4 * (grp & 1));
                acc[n][mt] = (f32x4){bb.x + (float)rv[0], bb.y + (float)rv[1], bb.z + (float)rv[2], bb.w + (float)rv[3]};
            } else {
                acc[n][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    __syncthreads();
    NUNIF_MLP_STAMP(4);
    f16x8 hb[2][kMT2];
#pragma unroll
    for (int mt = 0; mt < kMT2; ++mt) hb[0][mt] = hl[mt * 64 + lane];
#pragma unroll
    for (int k = 0; k < kPairsHalf; ++k) {
        const int sl = k % GK;
        f16x8 *hf = hb[k & 1], *hn = hb[(k & 1) ^ 1];
        if (k + 1 < kPairsHalf) {
#pragma unroll
            for (int mt = 0; mt < kMT2; ++mt) hn[mt] = hl[((k + 1) * kMT2 + mt) * 64 + lane];
        }
        vm_wait3(k + GK < kPairsHalf ? 21 : 21 - 3 * (k + GK - kPairsHalf), wq[3 * sl], wq[3 * sl + 1], wq[3 * sl + 2]);
#pragma unroll
        for (int n = 0; n < NW; ++n)
#pragma unroll
            for (int mt = 0; mt < kMT2; ++mt) acc[n][mt] = MFMA_16x16x32(wq[3 * sl + n], hf[mt], acc[n][mt]);
        if (k + GK < kPairsHalf) {
#pragma unroll
            for (int n = 0; n < NW; ++n) wload(wq[3 * sl + n], voff, w2_at(n, k + GK));
        }
    }
    NUNIF_MLP_STAMP(5);

    // scratch: [token group][sending half][local tile 12][mt 4][lane] x 16 bytes
    unsigned char *pbase = reinterpret_cast<unsigned char *>(a.partial) + (long)g * kPartialBytes;
    unsigned char *orow = smem_m;                                               // [64 tokens][192 channels] fp16, rows kORowBytes apart
    if (!role) {
        unsigned char *part = pbase + (long)(half * kOwnTiles + 3 * wl) * kMT2 * 1024;
#pragma unroll
        for (int n = 0; n < NW; ++n)
#pragma unroll
            for (int mt = 0; mt < kMT2; ++mt)
                asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1" ::"v"(voff), "v"(acc[n][mt]), "s"(part + (n * kMT2 + mt) * 1024)
                             : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // the last of the four sending waves (its own stores and, by the LDS counter's order, the others' are complete) raises the flag
        if (lane == 0 && atomicAdd(sent, 1) == 3) __hip_atomic_store(a.flags + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        NUNIF_MLP_STAMP(8);
    } else {
        if (lane == 0) {
            // bounded: ~2^24 polls of >= 128 clocks (about a second) is four orders of magnitude beyond the partner's whole run.
            // A partner that never arrives (the two blocks not co-resident: see launch_da_mlp) must not hang the device: trap,
            // the launch faults and the host sees it at its next synchronisation
            unsigned polls = 0;
            while (__hip_atomic_load(a.flags + (blockIdx.x ^ 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                __builtin_amdgcn_s_sleep(2);
                if (++polls > (1u << 24)) __builtin_trap();
            }
        }
        __builtin_amdgcn_wave_barrier();
        NUNIF_MLP_STAMP(6);
        const unsigned char *part = pbase + (long)((1 - half) * kOwnTiles + 3 * wl) * kMT2 * 1024;
        f32x4 pv[NW][kMT2];
#pragma unroll
        for (int n = 0; n < NW; ++n)
#pragma unroll
            for (int mt = 0; mt < kMT2; ++mt)
                asm volatile("global_load_dwordx4 %0, %1, %2 sc0 sc1" : "=v"(pv[n][mt]) : "v"(voff), "s"(part + (n * kMT2 + mt) * 1024) : "memory");
#pragma unroll
        for (int n = 0; n < NW; ++n)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(pv[n][0]), "+v"(pv[n][1]), "+v"(pv[n][2]), "+v"(pv[n][3]));
        NUNIF_MLP_STAMP(9);
        // the finished rows go through LDS (the input rows' space) so that the global stores are whole 384-byte runs
#pragma unroll
        for (int n = 0; n < NW; ++n)
#pragma unroll
            for (int mt = 0; mt < kMT2; ++mt) {
                const f32x4 v = acc[n][mt] + pv[n][mt];
                const f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                *reinterpret_cast<f16x4 *>(orow + (16 * mt + r16) * kORowBytes + ((3 * wl + n) * 16 + 4 * grp) * 2) = o;
            }
    }
    NUNIF_MLP_STAMP(10);
    __syncthreads();
    // all four finishing waves have seen the partner's flag: lower it for the next launch (stream order makes that visible)
    if (tid == 0) __hip_atomic_store(a.flags + (blockIdx.x ^ 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int j = 0; j < kTok2 * 24 / (kWavesM * 64); ++j) {
        const int c = tid + kWavesM * 64 * j, tok = c / 24, ch = c % 24;            // 16-byte chunk ch of this half's 192 channels
        const f16x8 v = *reinterpret_cast<const f16x8 *>(orow + tok * kORowBytes + ch * 16);
        if (m0 + tok < a.M) *reinterpret_cast<f16x8 *>(a.t + (m0 + tok) * kD + half * (kD / 2) + ch * 8) = v;
    }
    if (a.stats_out && tid < kTok2 * (kParts / 2)) {
        // this block's six of a token's twelve partials (32-channel pairs of its own 192 channels), from the fp16 values as stored
        const int tok = tid / (kParts / 2), q = tid % (kParts / 2);
        float su = 0.f, sq = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f16x8 v = *reinterpret_cast<const f16x8 *>(orow + tok * kORowBytes + q * 64 + u * 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float x = (float)v[e]; su += x; sq = fmaf(x, x, sq); }
        }
        if (m0 + tok < a.M) a.stats_out[(m0 + tok) * kParts + half * (kParts / 2) + q] = make_float2(su, sq);
    }
    NUNIF_MLP_STAMP(7);
}

bool da_mlp_supported(int D, int hidden) { return D == kD && hidden == kH; }

long da_mlp_partial_bytes(long M) { return ((M + kTok2 - 1) / kTok2) * (long)kPartialBytes; }
long da_mlp_flag_count(long M) { return 2 * ((M + kTok2 - 1) / kTok2); }   // one per block, zero between launches

int launch_da_mlp(const DaMlpArgs &a, hipStream_t s) {
    NUNIF_REQUIRE(a.t && a.w1 && a.b1 && a.ws1 && a.w2c && a.b2 && a.stats_in && a.M > 0, "da_mlp: bad argument");
    ProfScope ps("da_mlp_kernel", s, 4.0 * (double)a.M * kD * kH, (double)a.M * kD * 4.0);
    static bool configured = false;
    static int max_pair_grid = 256;
    if (!configured) {
        NUNIF_HIP_CHECK(hipFuncSetAttribute((const void *)da_mlp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kSmemM));
        NUNIF_HIP_CHECK(hipFuncSetAttribute((const void *)da_mlp_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kSmemM2));
        int dev = 0, cus = 0;
        NUNIF_HIP_CHECK(hipGetDevice(&dev));
        NUNIF_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        max_pair_grid = cus;                                                     // one workgroup per CU (LDS): all of them resident
        if (const char *e = getenv("NUNIF_DA_MLP_SPLIT")) if (atoi(e) == 0) max_pair_grid = 0;
        configured = true;
    }
    const long groups = (a.M + kTok2 - 1) / kTok2;
    if (a.partial && a.flags && 2 * groups <= max_pair_grid) {
        da_mlp_split_kernel<<<(unsigned)(2 * groups), kWavesM * 64, kSmemM2, s>>>(a);
    } else {
        da_mlp_kernel<<<(unsigned)((a.M + kTok - 1) / kTok), kWavesM * 64, kSmemM, s>>>(a);
    }
    NUNIF_LAUNCH_CHECK();
#ifdef NUNIF_MLP_TRACE
    {
        static int dumped = 0;
        if (a.partial && dumped < 40 && ++dumped >= 38) {
            static unsigned long long host[512 * 32];
            NUNIF_HIP_CHECK(hipStreamSynchronize(s));
            NUNIF_HIP_CHECK(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_mlp_trace), sizeof(host)));
            const int nb = (int)std::min<long>(2 * groups, 512);
            for (int role = 0; role < 2; ++role) {
                double avg[12] = {0}, mx[12] = {0};
                for (int b = 0; b < nb; ++b)
                    for (int i = 0; i < 12; ++i) {
                        const double v = (double)(long long)(host[b * 32 + role * 16 + i] - host[b * 32]);
                        avg[i] += v / nb; mx[i] = std::max(mx[i], v);
                    }
                fprintf(stderr, "[mlp trace] role %d blocks %d  avg:", role, nb);
                for (int i = 0; i < 12; ++i) fprintf(stderr, " %.0f", avg[i]);
                fprintf(stderr, "  max:");
                for (int i = 0; i < 12; ++i) fprintf(stderr, " %.0f", mx[i]);
                fprintf(stderr, "\n");
            }
        }
    }
#endif
    return NUNIF_HIP_OK;
}

}  // namespace nunif
