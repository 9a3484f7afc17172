// cunet's up step for gfx950:  out = LeakyReLU(ConvTranspose2d(64, 64, 2, 2)(se_scale * x)) + crop(skip)   (waifu2x/models/cunet.py:
// 40-45,58-60 / 86-97,111-118 and nunif/modules/attention.py:29-44 for the scale), NHWC fp16 — as a pixel-shuffle GEMM 64 -> 4 x 64.
//
// The same problem as the swin PatchUp (swin_patchup.hip), at K = 64: 1.03 GB of compulsory traffic per launch of the 1080p render
// (activations 0.11, skip tiles 0.46, output 0.46) for 58 GFLOP — memory-bound by a wide margin.  gemm_kernel<2,4> (weights through an
// LDS ring, the skip tile of a trip requested at the top of that trip) moved it at 3.5-3.8 TB/s.  Here, as there:
//   * the 32 KiB of weights (16 tiles x 2 k-steps) and the bias are resident in LDS, loaded once per persistent workgroup by LDS-DMA;
//   * a wave owns 32 tokens per group; the NEXT group's activations are requested while this one computes, and the skip tiles travel
//     through a register ring kD = 4 trips deep that runs across group boundaries; stores are unconditional (clamped duplicate rows
//     store what the last token's own lane stores), so hipcc's vmcnt bookkeeping sees every memory operation;
//   * the squeeze-excitation scale of the input map rides on the activation fragments when they are taken over from the prefetch
//     registers: fp16(x * s), the arithmetic of a separate scale pass;
//   * bias is the MFMA C operand; LeakyReLU, then the skip (fp32) — gemm_kernel's order.
// The skip map is LARGER than the output (cunet crops it): a token's skip address and output address differ, both 32-bit byte offsets
// from wave-uniform bases.
#include <algorithm>
#include <cstdlib>

#include "swin_kernels.h"

namespace nunif {

#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

namespace {
constexpr int kC = 64;                               // Cin = Cq = 64
constexpr int kKS = 2;
constexpr int kNT = 16;                              // 4 sub-pixels x 64 channels
constexpr int kTrips = kNT / 2;
constexpr int kWaves = 8;
constexpr int kMF = 2;
constexpr int kD = 4;
constexpr int kWBytes = kNT * kKS * 1024;            // 32 768
constexpr int kSmem = kWBytes + kNT * 16 * 4;
static_assert(kTrips % kD == 0, "the ring slot of a trip is static");

__device__ __forceinline__ void dma16(const void *src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_byte_addr) : "memory");
}
}  // namespace

template <bool SCALE>
__global__ void __launch_bounds__(kWaves * 64, 4) cunet_up_kernel(CunetUpArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_cu[];
    const f16x8 *wres = reinterpret_cast<const f16x8 *>(smem_cu);
    const float4 *bres = reinterpret_cast<const float4 *>(smem_cu + kWBytes);
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, grp = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const unsigned char *src = reinterpret_cast<const unsigned char *>(g.w) + lane * 16;
        const unsigned lds0 = (unsigned)reinterpret_cast<size_t>(smem_cu);
#pragma unroll
        for (int u = 0; u < kWBytes / 1024 / kWaves; ++u) {
            const int i = wave + kWaves * u;
            dma16(src + (size_t)i * 1024, lds0 + i * 1024);
        }
        if (tid < kNT * 4) reinterpret_cast<float4 *>(smem_cu + kWBytes)[tid] = reinterpret_cast<const float4 *>(g.bias)[tid];
        // the scale table [B][64] goes to LDS as well: read per group through lgkmcnt — a global read here would sit at the young end
        // of the vmcnt queue and every group would start by draining the skip-tile ring
        if constexpr (SCALE)
            for (int i = tid; i < g.B * (kC / 4); i += kWaves * 64)
                reinterpret_cast<float4 *>(smem_cu + kSmem)[i] = reinterpret_cast<const float4 *>(g.in_scale)[i];
    }
    const unsigned S = (unsigned)g.S;                 // input side (square maps); output side 2 S; skip side g.res_S
    const unsigned M = (unsigned)g.B * S * S;
    const unsigned n_groups = (M + kMF * 16 - 1) / (kMF * 16);
    const unsigned step = gridDim.x * kWaves;
    unsigned gi = blockIdx.x * kWaves + wave;
    const bool any = gi < n_groups;
    const int prc = pair_run_channel(grp);
    const unsigned RS = (unsigned)g.res_S, crop = (unsigned)g.crop;

    // byte offsets of trip t's 32-channel pair inside a token's 2 x 2 output block: column n = q 64 + c, q = (qi, qj)
    auto out_off = [&](int t) -> unsigned { const int q = t / 2, c = (t % 2) * 32; return (((q >> 1) * 2 * S + (q & 1)) * kC + c) * 2; };
    auto res_off = [&](int t) -> unsigned { const int q = t / 2, c = (t % 2) * 32; return (((q >> 1) * RS + (q & 1)) * kC + c) * 2; };
    struct Grp { unsigned xo[kMF], po[kMF], pr[kMF], so[kMF]; };     // BYTE offsets: activations, output, skip; scale (floats) offset
    auto coords = [&](unsigned gidx, Grp &c) {
#pragma unroll
        for (int f = 0; f < kMF; ++f) {
            unsigned m = gidx * (kMF * 16) + f * 16 + r16;
            m = m < M ? m : M - 1;                                    // clamped duplicates: swin_patchup.hip
            const unsigned t = m / S, x = m - t * S;
            const unsigned b = t / S, y = t - b * S;
            c.xo[f] = (m * kC + grp * 8) * 2;
            c.po[f] = (((b * 2 * S + 2 * y) * (2 * S) + 2 * x) * kC + prc) * 2;
            c.pr[f] = (((b * RS + 2 * y + crop) * RS + 2 * x + crop) * kC + prc) * 2;
            c.so[f] = (b * kC + grp * 8) * 4;
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
        const unsigned char *base = reinterpret_cast<const unsigned char *>(g.res) + (size_t)res_off(t);          // scalar
#pragma unroll
        for (int f = 0; f < kMF; ++f) r[f] = *reinterpret_cast<const f16x8 *>(base + (size_t)c.pr[f]);
    };
    // x <- fp16(x * s[b][c]) on the fragments of group c (the separate pass's rounding)
    auto apply_scale = [&](const Grp &c, f16x8 (&xf)[kMF][kKS]) {
        if constexpr (SCALE) {
#pragma unroll
            for (int f = 0; f < kMF; ++f) {
                const unsigned char *sp = smem_cu + kSmem + c.so[f];
#pragma unroll
                for (int ks = 0; ks < kKS; ++ks) {
                    const float4 s0 = *reinterpret_cast<const float4 *>(sp + ks * 128), s1 = *reinterpret_cast<const float4 *>(sp + ks * 128 + 16);
                    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) xf[f][ks][j] = (f16)((float)xf[f][ks][j] * sc[j]);
                }
            }
        }
    };

    Grp cur, nxt;
    f16x8 xc[kMF][kKS], xn[kMF][kKS];
    f16x8 rr[kD][kMF];
    coords(any ? gi : 0, cur);
    load_x(cur, xc);
#pragma unroll
    for (int t = 0; t < kD; ++t) load_res(cur, t, rr[t]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (hipcc must know that nothing is pending here: swin_patchup.hip)
#pragma unroll
    for (int f = 0; f < kMF; ++f) {
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) asm volatile("" : "+v"(xc[f][ks]));
#pragma unroll
        for (int t = 0; t < kD; ++t) asm volatile("" : "+v"(rr[t][f]));
    }
    __syncthreads();
    if (!any) return;
    apply_scale(cur, xc);

    while (true) {
        const unsigned gn = gi + step;
        const bool has_next = gn < n_groups;                       // wave-uniform
        coords(has_next ? gn : gi, nxt);                           // (no next group: harmless re-reads of this one)
        load_x(nxt, xn);
        int lofs = lane;                                           // opaque: the LDS reads stay inside the loop
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
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) {
                const f16x8 wa = wres[((2 * t) * kKS + ks) * 64 + lofs], wb = wres[((2 * t + 1) * kKS + ks) * 64 + lofs];
#pragma unroll
                for (int f = 0; f < kMF; ++f) {
                    acc0[f] = MFMA_16x16x32(wa, xc[f][ks], acc0[f]);
                    acc1[f] = MFMA_16x16x32(wb, xc[f][ks], acc1[f]);
                }
            }
            unsigned char *obase = reinterpret_cast<unsigned char *>(g.out) + (size_t)out_off(t);                   // scalar
            // the skip tile is consumed HERE and not before (swin_patchup.hip)
#pragma unroll
            for (int f = 0; f < kMF; ++f) asm volatile("" : "+v"(rr[t % kD][f]));
#pragma unroll
            for (int f = 0; f < kMF; ++f) {
                float lo[4], hi[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // LeakyReLU on the accumulators, then the tile pair -> the run layout of loads / stores (fp32 permlane swaps;
                    // elements copied into scalars first: common.h row_group_max)
                    float a0 = acc0[f][r], a1 = acc1[f][r];
                    a0 = a0 >= 0.f ? a0 : a0 * g.slope;
                    a1 = a1 >= 0.f ? a1 : a1 * g.slope;
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
                *reinterpret_cast<f16x8 *>(obase + (size_t)cur.po[f]) = ov;
            }
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
        apply_scale(cur, xc);
    }
}

bool cunet_up_supported(const CunetUpArgs &g) {
    if (const char *e = getenv("NUNIF_CUNET_UP")) if (atoi(e) == 0) return false;       // read per call (A/B runs)
    const long M = (long)g.B * g.S * g.S;
    return g.a && g.w && g.bias && g.res && g.out && M > 0 && g.res_S >= 2 * g.S + 2 * g.crop && g.crop >= 0 &&
           (long)g.B * g.res_S * g.res_S * kC * 2 < (1L << 32) && 4 * M * kC * 2 < (1L << 32) && g.out != g.res &&
           (!g.in_scale || g.B <= 128);                                  // the scale table [B][64] fp32 rides in LDS
}

int launch_cunet_up(const CunetUpArgs &g, hipStream_t s) {
    NUNIF_REQUIRE(cunet_up_supported(g), "cunet_up: bad argument");
    const long M = (long)g.B * g.S * g.S;
    static bool configured = false;
    static int cus = 256;
    if (!configured) {
        NUNIF_HIP_CHECK(hipFuncSetAttribute((const void *)cunet_up_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem + 128 * kC * 4));
        NUNIF_HIP_CHECK(hipFuncSetAttribute((const void *)cunet_up_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));
        int dev = 0;
        NUNIF_HIP_CHECK(hipGetDevice(&dev));
        NUNIF_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        configured = true;
    }
    const long groups = (M + kMF * 16 - 1) / (kMF * 16);
    const unsigned grid = (unsigned)std::max<long>(1, std::min<long>((groups + kWaves - 1) / kWaves, 2L * cus));
    ProfScope ps("cunet_up_kernel", s, 2.0 * (double)M * kC * 4.0 * kC, (double)M * kC * 2.0 * (1.0 + 4.0 + 4.0));
    if (g.in_scale) cunet_up_kernel<true><<<grid, kWaves * 64, kSmem + (size_t)g.B * kC * 4, s>>>(g);
    else cunet_up_kernel<false><<<grid, kWaves * 64, kSmem, s>>>(g);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

}  // namespace nunif
