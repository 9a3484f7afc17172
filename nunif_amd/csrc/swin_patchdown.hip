// PatchDown of the swin U-Nets for gfx950:  out = Conv2d(Cin -> 192, kernel 2, stride 2)(x)   (waifu2x/models/swin_unet.py:45-62), NHWC
// fp16, as a gather GEMM with K = 384: down1 is the whole 2 x 2 x 96 patch of a token; down2 (2 x 2 x 192 = 768) runs as its two tap
// ROWS of K = 384, the second accumulating onto the first's output (swin_unet.cpp) — this kernel takes the passes WITHOUT a residual
// (down1, down2's first row), `gemm_res_kernel<12,2>` keeps the accumulating one.
//
// Why a kernel of its own (round 5).  `gemm_res_kernel<12,2>` already holds the 144 KiB of weights in LDS, but a wave loaded its 32
// tokens' 24 activation fragments, waited for them, multiplied, stored, and only then asked for the next group's: SQ counters of
// round 4 have 56 % of its wave-cycles in s_waitcnt, 238 us for down1 against a 119-us byte floor.  Prefetching the next group next
// to this one's needs 2 x 96 registers in the token-stationary loop order (all of a token's K is live until the last output tile).
// Here the loop order is K-OUTER: all 12 x 2 accumulator tiles of the group are live (96 registers) and k-step ks's activation
// fragment dies after its 24 MFMAs — its registers take the NEXT group's fragment ks right there.  One group of lead on every load,
// no second register set: 96 (activations) + 96 (accumulators).  Bias is the MFMA C operand; token coordinates are 32-bit.
// Rows beyond M are clamped (duplicate loads / stores of the last token's values: no masks, no branches — swin_patchup.hip).
#include <algorithm>
#include <cstdlib>

#include "swin_kernels.h"

namespace nunif {

#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

namespace {
constexpr int kWaves = 8;
constexpr int kMF = 2;

__device__ __forceinline__ void dma16(const void *src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_byte_addr) : "memory");
}
}  // namespace

// <96, 12, 12>: four taps (2 x 2) of 96 channels -> 192; <192, 12, 12>: the two taps of ONE row (g.oy) -> 192;
// <64, 8, 4>: cunet's Conv2d(64, 64, 2, 2) + LeakyReLU (waifu2x/models/cunet.py:37,78), 32 KiB of weights: two workgroups per CU
template <int CIN, int kKS, int kNT, bool LRELU>
__global__ void __launch_bounds__(kWaves * 64) patchdown_kernel(PatchDownArgs g) {
    static_assert(CIN == 64 || CIN == 96 || CIN == 192, "Cin");
    constexpr int kWBytes = kNT * kKS * 1024;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_pd[];
    const f16x8 *wres = reinterpret_cast<const f16x8 *>(smem_pd);
    const float4 *bres = reinterpret_cast<const float4 *>(smem_pd + kWBytes);
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, grp = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const unsigned char *src = reinterpret_cast<const unsigned char *>(g.w) + lane * 16;
        const unsigned lds0 = (unsigned)reinterpret_cast<size_t>(smem_pd);
#pragma unroll
        for (int u = 0; u < kWBytes / 1024 / kWaves; ++u) {
            const int i = wave + kWaves * u;
            dma16(src + (size_t)i * 1024, lds0 + i * 1024);
        }
        if (tid < kNT * 4) reinterpret_cast<float4 *>(smem_pd + kWBytes)[tid] = reinterpret_cast<const float4 *>(g.bias)[tid];
    }
    const unsigned Wo = (unsigned)g.Wo, Ho = (unsigned)g.Ho, Wi = 2 * Wo, Hi = 2 * Ho;
    const unsigned M = (unsigned)g.B * Ho * Wo;
    const unsigned n_groups = (M + kMF * 16 - 1) / (kMF * 16);
    const unsigned step = gridDim.x * kWaves;
    unsigned gi = blockIdx.x * kWaves + wave;
    const bool any = gi < n_groups;
    const int prc = pair_run_channel(grp);

    struct Grp { unsigned in[kMF], out[kMF]; };        // BYTE offsets (32 bits: launcher guard)
    auto coords = [&](unsigned gidx, Grp &c) {
        const unsigned gg = g.rev ? n_groups - 1 - gidx : gidx;
#pragma unroll
        for (int f = 0; f < kMF; ++f) {
            unsigned m = gg * (kMF * 16) + f * 16 + r16;
            m = m < M ? m : M - 1;
            const unsigned t = m / Wo, x = m - t * Wo;
            const unsigned b = t / Ho, y = t - b * Ho;
            c.in[f] = (((b * Hi + 2 * y + (unsigned)g.oy) * Wi + 2 * x) * CIN + grp * 8) * 2;
            c.out[f] = (m * (kNT * 16) + prc) * 2;
        }
    };
    // byte offset of k-step ks inside a token's gather: k = tap Cin + c, tap = (dy, dx) -> pixel (+dy, +dx)
    auto ks_off = [&](int ks) -> unsigned {
        const int k0 = ks * 32, tap = k0 / CIN, c0 = k0 % CIN;
        const int dy = tap / 2, dx = tap % 2;
        return (unsigned)((dy * Wi + dx) * CIN + c0) * 2;
    };
    auto load_frag = [&](const Grp &c, int f, int ks) -> f16x8 {
        const unsigned char *base = reinterpret_cast<const unsigned char *>(g.a) + (size_t)ks_off(ks);       // scalar
        return *reinterpret_cast<const f16x8 *>(base + (size_t)c.in[f]);
    };

    Grp cur, nxt;
    f16x8 xf[kMF][kKS];
    coords(any ? gi : 0, cur);
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) {
#pragma unroll
        for (int f = 0; f < kMF; ++f) xf[f][ks] = load_frag(cur, f, ks);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // hipcc must KNOW that nothing is pending here: it merges the wait counts of this path into the loop header, and with the 24
    // prologue loads still on its books every group would start by draining its whole prefetch (s_waitcnt vmcnt(11) at k-step 0)
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) {
#pragma unroll
        for (int f = 0; f < kMF; ++f) asm volatile("" : "+v"(xf[f][ks]));
    }
    __syncthreads();
    if (!any) return;

    while (true) {
        const unsigned gn = gi + step;
        const bool has_next = gn < n_groups;                       // wave-uniform
        coords(has_next ? gn : gi, nxt);                           // (no next group: harmless re-reads of this one)
        // the fragments are the same for every group: an opaque offset keeps hipcc from hoisting the LDS reads out of the loop
        int lofs = lane;
        asm volatile("" : "+v"(lofs));
        f32x4 acc[kNT][kMF];
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            const float4 bv = bres[nt * 4 + (lofs >> 4)];
#pragma unroll
            for (int f = 0; f < kMF; ++f) acc[nt][f] = (f32x4){bv.x, bv.y, bv.z, bv.w};
        }
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) {
                const f16x8 wv = wres[(nt * kKS + ks) * 64 + lofs];
#pragma unroll
                for (int f = 0; f < kMF; ++f) acc[nt][f] = MFMA_16x16x32(wv, xf[f][ks], acc[nt][f]);
            }
            // k-step ks of this group is done: its registers take the next group's fragment
#pragma unroll
            for (int f = 0; f < kMF; ++f) xf[f][ks] = load_frag(nxt, f, ks);
            // ... HERE: left alone, hipcc's scheduler sinks all 24 loads behind the last MFMA of the group (nothing between them and
            // the stores they may alias), and the lead shrinks from a group to an epilogue
            __builtin_amdgcn_sched_barrier(0);
        }
        unsigned char *obase = reinterpret_cast<unsigned char *>(g.out);
#pragma unroll
        for (int p = 0; p < kNT / 2; ++p) {
#pragma unroll
            for (int f = 0; f < kMF; ++f) {
                f32x4 a0 = acc[2 * p][f], a1 = acc[2 * p + 1][f];
                if constexpr (LRELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        a0[r] = a0[r] >= 0.f ? a0[r] : a0[r] * g.slope;
                        a1[r] = a1[r] >= 0.f ? a1[r] : a1[r] * g.slope;
                    }
                }
                const f16x8 ov = pair_to_run((f16x4){(f16)a0[0], (f16)a0[1], (f16)a0[2], (f16)a0[3]},
                                             (f16x4){(f16)a1[0], (f16)a1[1], (f16)a1[2], (f16)a1[3]});
                *reinterpret_cast<f16x8 *>(obase + (size_t)cur.out[f] + p * 64) = ov;
            }
        }
        if (!has_next) break;
        cur = nxt;
        gi = gn;
    }
}

bool patchdown_supported(const PatchDownArgs &g) {
    if (const char *e = getenv("NUNIF_PATCHDOWN")) if (atoi(e) == 0) return false;      // read per call (A/B runs)
    const long M = (long)g.B * g.Ho * g.Wo;
    const bool shape = (g.Cin == 96 && g.N == 192 && g.oy == 0) || (g.Cin == 192 && g.N == 192 && (g.oy == 0 || g.oy == 1)) ||
                       (g.Cin == 64 && g.N == 64 && g.oy == 0);
    const bool act_ok = g.Cin == 64 ? g.act == 2 : g.act == 0;               // (the activation is a template parameter)
    // byte offsets into the input map (4 M pixels of Cin channels) and the output map (2 N bytes per token) are 32-bit
    return shape && M > 0 && 4 * M * g.Cin * 2 < (1L << 32) && M * g.N * 2 < (1L << 32) && act_ok;
}

template <int CIN, int KS, int NT, bool LRELU>
static int launch_pd(const PatchDownArgs &g, hipStream_t s, const char *name, int wgs_per_cu) {
    constexpr int smem = NT * KS * 1024 + NT * 16 * 4;
    static_assert(smem <= 160 * 1024, "LDS");
    const long M = (long)g.B * g.Ho * g.Wo;
    static bool configured = false;
    static int cus = 256;
    if (!configured) {
        NUNIF_HIP_CHECK(hipFuncSetAttribute((const void *)patchdown_kernel<CIN, KS, NT, LRELU>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        int dev = 0;
        NUNIF_HIP_CHECK(hipGetDevice(&dev));
        NUNIF_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        configured = true;
    }
    const long groups = (M + kMF * 16 - 1) / (kMF * 16);
    const unsigned grid = (unsigned)std::max<long>(1, std::min<long>((groups + kWaves - 1) / kWaves, (long)cus * wgs_per_cu));
    ProfScope ps(name, s, 2.0 * (double)M * (KS * 32.0) * (NT * 16.0), (double)M * (KS * 32.0 + NT * 16.0) * 2.0);
    patchdown_kernel<CIN, KS, NT, LRELU><<<grid, kWaves * 64, smem, s>>>(g);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

int launch_patchdown(const PatchDownArgs &g, hipStream_t s) {
    NUNIF_REQUIRE(g.a && g.w && g.bias && g.out && patchdown_supported(g), "patchdown: bad argument");
    if (g.Cin == 96) return launch_pd<96, 12, 12, false>(g, s, "patchdown_kernel<96>", 1);
    if (g.Cin == 192) return launch_pd<192, 12, 12, false>(g, s, "patchdown_kernel<192>", 1);
    return launch_pd<64, 8, 4, true>(g, s, "patchdown_kernel<64>", 2);
}

}  // namespace nunif
