// PatchUp of the swin U-Nets for gfx950:  out = pixel_shuffle_2(Linear(192 -> 4 Cq)(x)) + skip   (waifu2x/models/swin_unet.py:65-82 and
// the `x = up(x) + skip` lines of SwinUNetBase.forward :189-196), NHWC fp16, written in place over the skip map.
//
// Why a kernel of its own (round 5).  On `gemm_kernel<6,2>` (token-stationary, weights through an LDS ring, 3 workgroups per CU) the
// two PatchUps of the 2x net took 200 + 380 us for 0.56 + 1.25 GB of compulsory traffic: 3.2 TB/s at 0.10 of the MFMA peak.  PMC
// traffic was 1.04x algorithmic, so nothing was re-read — the kernel was bound by memory-level parallelism: a wave had ONE trip's skip
// tile (2 KiB) in flight, requested at the top of the trip and consumed at its end, ~24 KiB per CU with a duty cycle well under one.
// Asking for it a whole trip earlier cost the third resident workgroup; resident weights alone (gemm_res_kernel<6,2>)
// lost as well (381 vs 340 us) because the skip tile was still requested in the epilogue that consumes it.
//
// This form keeps what both lacked:
//   * weights RESIDENT in LDS: one persistent 8-wave workgroup per CU copies 384 output columns' fragments (24 tiles x 6 k-steps x
//     1 KiB = 144 KiB) in by LDS-DMA once, plus their bias; no ring, no barrier after the prologue.  Cq = 192 (768 columns) runs as
//     two column halves = the two sub-pixel ROWS, on alternating workgroups;
//   * with two waves per SIMD every wave owns 256 registers: the NEXT token group's 32 x 192 activations (48 registers) are requested
//     while this group computes, and the skip tiles travel through a register ring kD = 4 trips deep that runs across group
//     boundaries — 8 KiB + 12 KiB of loads in flight per wave, ~160 KiB per CU, against ~24 before.  `out` may alias `res`, so hipcc
//     keeps every load between the stores it is written between: the source order below IS the issue order;
//   * bias is the MFMA C operand (accumulators start from it), token coordinates are 32-bit divisions once per group.
// Operand orientation as everywhere (swin_kernels.hip): weights = A (rows = output channels), activations = B (columns = tokens); a
// lane's accumulator is 4 consecutive channels of one token, two adjacent tiles leave as one 16-byte run per lane (pair_to_run).
#include <algorithm>
#include <cstdlib>

#include "swin_kernels.h"

namespace nunif {

#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

namespace {
constexpr int kKS = 6;                               // K = 192
constexpr int kNT = 24;                              // 16-column tiles resident per workgroup (384 columns)
constexpr int kTrips = kNT / 2;                      // a trip = one 32-channel pair of tiles
constexpr int kWaves = 8;
constexpr int kMF = 2;                               // token tiles per wave and group: 32 tokens
#ifndef NUNIF_PU_KD
#define NUNIF_PU_KD 4
#endif
constexpr int kD = NUNIF_PU_KD;                      // depth of the skip-tile ring, in trips (A/B builds: -DNUNIF_PU_KD=6)
constexpr int kWBytes = kNT * kKS * 1024;            // 147 456
constexpr int kSmem = kWBytes + 384 * 4;             // + bias
static_assert(kSmem <= 160 * 1024, "LDS");
static_assert(kTrips % kD == 0, "the ring slot of a trip is static");

__device__ __forceinline__ void dma16(const void *src, unsigned lds_byte_addr) {
    // each lane moves 16 B to LDS[m0 + 16 * lane]; m0 is wave-uniform
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_byte_addr) : "memory");
}
}  // namespace

template <int CQ>
__global__ void __launch_bounds__(kWaves * 64) patchup_kernel(PatchUpArgs g) {
    constexpr int HALVES = 4 * CQ / (kNT * 16);      // 1 (Cq = 96) or 2 (Cq = 192: workgroup parity picks the sub-pixel row)
    static_assert(HALVES == 1 || HALVES == 2, "Cq is 96 or 192");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_pu[];
    const f16x8 *wres = reinterpret_cast<const f16x8 *>(smem_pu);
    const float4 *bres = reinterpret_cast<const float4 *>(smem_pu + kWBytes);
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, grp = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = HALVES == 2 ? (int)(blockIdx.x & 1) : 0;
    const unsigned wb = blockIdx.x / HALVES, nwb = gridDim.x / HALVES;

    {   // this half's fragments and bias -> LDS
        const unsigned char *src = reinterpret_cast<const unsigned char *>(g.w) + (size_t)half * kWBytes + lane * 16;
        const unsigned lds0 = (unsigned)reinterpret_cast<size_t>(smem_pu);
#pragma unroll
        for (int u = 0; u < kWBytes / 1024 / kWaves; ++u) {
            const int i = wave + kWaves * u;
            dma16(src + (size_t)i * 1024, lds0 + i * 1024);
        }
        if (tid < 96) reinterpret_cast<float4 *>(smem_pu + kWBytes)[tid] = reinterpret_cast<const float4 *>(g.bias + half * 384)[tid];
    }

    const unsigned Wo = (unsigned)g.Wo, Ho = (unsigned)g.Ho;
    const unsigned M = (unsigned)g.B * Ho * Wo;
    const unsigned n_groups = (M + kMF * 16 - 1) / (kMF * 16);
    const unsigned step = nwb * kWaves;
    unsigned gi = wb * kWaves + wave;
    const bool any = gi < n_groups;
    const int prc = pair_run_channel(grp);

    // element offset of trip t's 32-channel pair inside a token's 2 x 2 output block: column n = q Cq + c, q = (qi, qj)
    auto trip_off = [&](int t) -> unsigned {
        const int ql = (t * 32) / CQ, c = (t * 32) % CQ;
        const int q = HALVES == 2 ? 2 * half + ql : ql;
        return (unsigned)(((q >> 1) * 2 * Wo + (q & 1)) * CQ + c);
    };
    struct Grp { unsigned pix[kMF], xo[kMF]; };      // BYTE offsets (32 bits: launcher guard), added to a wave-uniform base
    auto coords = [&](unsigned gidx, Grp &c) {
        const unsigned gg = g.rev ? n_groups - 1 - gidx : gidx;
#pragma unroll
        for (int f = 0; f < kMF; ++f) {
            // rows beyond M are clamped to the last token: such a lane loads what the last token's own lane loads and stores the same
            // values to the same address in the same instruction — no mask, hence no branch around the stores (a conditional store
            // also hides itself from hipcc's vmcnt bookkeeping: every wait in front of a skip tile then drained the ring)
            unsigned m = gg * (kMF * 16) + f * 16 + r16;
            m = m < M ? m : M - 1;
            const unsigned t = m / Wo, x = m - t * Wo;
            const unsigned b = t / Ho, y = t - b * Ho;
            c.xo[f] = (m * (kKS * 32) + grp * 8) * 2;
            c.pix[f] = (((b * 2 * Ho + 2 * y) * (2 * Wo) + 2 * x) * CQ + prc) * 2;
        }
    };
    auto load_x = [&](const Grp &c, f16x8 (&xf)[kMF][kKS]) {
#pragma unroll
        for (int f = 0; f < kMF; ++f) {
            const unsigned char *p = reinterpret_cast<const unsigned char *>(g.a) + (size_t)c.xo[f];
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) xf[f][ks] = *reinterpret_cast<const f16x8 *>(p + ks * 64);
        }
    };
    auto load_res = [&](const Grp &c, int t, f16x8 (&r)[kMF]) {
        const unsigned char *base = reinterpret_cast<const unsigned char *>(g.res) + (size_t)trip_off(t) * 2;      // scalar
#pragma unroll
        for (int f = 0; f < kMF; ++f) r[f] = *reinterpret_cast<const f16x8 *>(base + (size_t)c.pix[f]);
    };

    Grp cur, nxt;
    f16x8 xc[kMF][kKS], xn[kMF][kKS];
    f16x8 rr[kD][kMF];
    coords(any ? gi : 0, cur);
    load_x(cur, xc);
#pragma unroll
    for (int t = 0; t < kD; ++t) load_res(cur, t, rr[t]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // hipcc must KNOW that nothing is pending here: it merges this path's wait counts into the loop header, and with the prologue
    // loads still on its books the first trips of every group would wait for younger loads than they need
#pragma unroll
    for (int f = 0; f < kMF; ++f) {
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) asm volatile("" : "+v"(xc[f][ks]));
#pragma unroll
        for (int t = 0; t < kD; ++t) asm volatile("" : "+v"(rr[t][f]));
    }
    __syncthreads();
    if (!any) return;

    while (true) {
        const unsigned gn = gi + step;
        const bool has_next = gn < n_groups;                       // wave-uniform
        coords(has_next ? gn : gi, nxt);                           // (no next group: harmless re-reads of this one)
        load_x(nxt, xn);
        // the fragments are the same for every group: without an opaque offset hipcc hoists all 144 LDS reads out of the loop
        // (576 registers, spilled)
        int lofs = lane;
        asm volatile("" : "+v"(lofs));
#pragma unroll
        for (int t = 0; t < kTrips; ++t) {
            f32x4 acc0[kMF], acc1[kMF];
            {
                const float4 b0 = bres[t * 8 + (lofs >> 4)], b1 = bres[t * 8 + 4 + (lofs >> 4)];
#pragma unroll
                for (int f = 0; f < kMF; ++f) {
                    acc0[f] = (f32x4){b0.x, b0.y, b0.z, b0.w};
                    acc1[f] = (f32x4){b1.x, b1.y, b1.z, b1.w};
                }
            }
            f16x8 wa[kKS], wb[kKS];
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) wa[ks] = wres[((2 * t) * kKS + ks) * 64 + lofs];
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) wb[ks] = wres[((2 * t + 1) * kKS + ks) * 64 + lofs];
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) {
#pragma unroll
                for (int f = 0; f < kMF; ++f) acc0[f] = MFMA_16x16x32(wa[ks], xc[f][ks], acc0[f]);
            }
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) {
#pragma unroll
                for (int f = 0; f < kMF; ++f) acc1[f] = MFMA_16x16x32(wb[ks], xc[f][ks], acc1[f]);
            }
            unsigned char *obase = reinterpret_cast<unsigned char *>(g.out) + (size_t)trip_off(t) * 2;                   // scalar
            // the skip tile is consumed HERE and not before: without the opaque pass hipcc converts / swaps the loaded registers right
            // behind their load (s_waitcnt vmcnt(1) in front of every trip = no ring at all)
#pragma unroll
            for (int f = 0; f < kMF; ++f) asm volatile("" : "+v"(rr[t % kD][f]));
#pragma unroll
            for (int f = 0; f < kMF; ++f) {
                // accumulators (tile pair: channels 4 grp .. of each tile) -> the run layout of the loads / stores (8 consecutive
                // channels per lane, pair_to_run's permutation applied to fp32 registers): the skip tile needs no shuffle at all
                float lo[4], hi[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // (elements copied into scalars first: __builtin_bit_cast on an ext-vector ELEMENT reads element 0 with hipcc
                    //  of ROCm 7.2 — common.h row_group_max)
                    const float a0 = acc0[f][r], a1 = acc1[f][r];
                    const u32x2 sw = lane16_swap(__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, a1));
                    const unsigned s0 = sw[0], s1 = sw[1];
                    lo[r] = __builtin_bit_cast(float, s0);
                    hi[r] = __builtin_bit_cast(float, s1);
                }
                const f16x8 rv = rr[t % kD][f];
                f16x8 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ov[r] = (f16)(lo[r] + (float)rv[r]);
                    ov[4 + r] = (f16)(hi[r] + (float)rv[4 + r]);
                }
                *reinterpret_cast<f16x8 *>(obase + (size_t)cur.pix[f]) = ov;
            }
            // the ring slot just consumed takes the tile of kD trips ahead: this group's, or the next group's first trips
            if (t + kD < kTrips) load_res(cur, t + kD, rr[t % kD]);
            else load_res(nxt, t + kD - kTrips, rr[t % kD]);
        }
        if (!has_next) break;
#pragma unroll
        for (int f = 0; f < kMF; ++f) {
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) xc[f][ks] = xn[f][ks];
        }
        cur = nxt;
        gi = gn;
    }
}

bool patchup_supported(const PatchUpArgs &g) {
    if (const char *e = getenv("NUNIF_PATCHUP")) if (atoi(e) == 0) return false;       // read per call (A/B runs)
    const long M = (long)g.B * g.Ho * g.Wo;
    // byte offsets into the activations (384 B per token) and the output map (8 Cq B per token) are 32-bit
    return (g.Cq == 96 || g.Cq == 192) && g.res && M > 0 && M * 384 < (1L << 32) && M * 8 * g.Cq < (1L << 32);
}

int launch_patchup(const PatchUpArgs &g, hipStream_t s) {
    NUNIF_REQUIRE(g.a && g.w && g.bias && g.res && g.out && patchup_supported(g), "patchup: bad argument");
    const long M = (long)g.B * g.Ho * g.Wo;
    static bool configured = false;
    static int cus = 256;
    if (!configured) {
        NUNIF_HIP_CHECK(hipFuncSetAttribute((const void *)patchup_kernel<96>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));
        NUNIF_HIP_CHECK(hipFuncSetAttribute((const void *)patchup_kernel<192>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));
        int dev = 0;
        NUNIF_HIP_CHECK(hipGetDevice(&dev));
        NUNIF_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        configured = true;
    }
    const long groups = (M + kMF * 16 - 1) / (kMF * 16);
    const int halves = g.Cq == 192 ? 2 : 1;
    const long per_half = std::max<long>(1, std::min<long>((groups + kWaves - 1) / kWaves, std::max(1, cus / halves)));
    const unsigned grid = (unsigned)(per_half * halves);
    const double flops = 2.0 * (double)M * 192.0 * 4.0 * g.Cq;
    const double bytes = (double)M * (192.0 * 2.0 + 4.0 * g.Cq * 2.0 * 2.0);
    if (g.Cq == 96) {
        ProfScope ps("patchup_kernel<96>", s, flops, bytes);
        patchup_kernel<96><<<grid, kWaves * 64, kSmem, s>>>(g);
    } else {
        ProfScope ps("patchup_kernel<192>", s, flops, bytes);
        patchup_kernel<192><<<grid, kWaves * 64, kSmem, s>>>(g);
    }
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

}  // namespace nunif
