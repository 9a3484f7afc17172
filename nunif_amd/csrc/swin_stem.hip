// Fused swin_unet stem for gfx950: conv1 3x3 (3 -> 48) + LeakyReLU + conv2 3x3 (48 -> 96) + LeakyReLU + crop, one kernel.
//
// Replaces PatchDown's conv stack as SwinUNetBase.forward runs it (waifu2x/models/swin_unet.py:135-137 `patch`, :182 the
// F.pad(x, [-6] * 4) crop) and, in frame mode, the replicate-pad + tile slicing of nunif/utils/seam_blending.py:82,90.
//
// The two-kernel stem (stem1_kernel on the VALU + the gather GEMM) cost 0.84 ms of an 8.5 ms frame: 0.37 GB of conv1
// activations written to HBM, then gathered nine times through the L2 with 64-bit index math per 16-byte piece, at 23 %
// MFMA issue.  Here a persistent 8-wave workgroup owns one 16 x 16 output tile at a time and nothing but the input image
// and the 96-channel result touches memory:
//   1. the 20 x 20 x 3 input patch goes to LDS as fp16 (the next tile's patch is fetched into registers during step 3);
//   2. conv1 runs on the MFMA: K = 27 taps + a constant-one column carrying the bias (28 of 32), 18 x 18 pixels = 21 token
//      tiles, result (LeakyReLU, fp16) written to an 18 x 18 x 48 LDS tile with a 112-byte pixel stride (odd multiple
//      of 16 B: 16 consecutive pixels hit 16 distinct bank quads for ds_read_b128);
//   3. conv2 reads its B operands straight from that tile — k = tap * 48 + c, so K = 432 -> 14 k-steps instead of the 18
//      of the channel-padded gather GEMM — against W2 resident in LDS (84 KiB), wave w owning output rows 2w, 2w + 1.
// LDS: 84 (W2) + 3 (W1) + 35.4 (conv1 tile) + 2.4 (input) KiB = 125 KiB, one workgroup per CU.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "swin_kernels.h"

namespace nunif {

#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

namespace {
constexpr int kTile = 16;                 // output tile WIDTH; its height is 8 waves x ROWS rows (ROWS = 2: 16 x 16, ROWS = 3: 24 x 16)
constexpr int kWaves = 8;

// Geometry of one instantiation.  <48, 96, 6>: the swin_unet stem (crop 6 = F.pad(x, [-6] * 4), swin_unet.py:182).
// <32, 64, 0>: UNetConv(3, 32, 64) at the head of both cunet U-Nets (waifu2x/models/cunet.py:10-28,36,77), no crop.
// ROWS (round 5): output rows per wave.  With 2 every W2 fragment read from LDS feeds 2 MFMAs and a tile's eight waves read the
// whole 84-KiB W2 once each for 256 pixels: LDS reads (7.2 k cycles per tile and CU) cost more than the MFMAs (5.7 k per SIMD) and
// the two barely overlap.  With 3 the tile is 24 x 16: a third less W2 traffic and a third fewer barriers per pixel for 16 KiB more
// of conv1 tile (143 KiB of LDS for the swin stem).
template <int C1, int C, int CROP, int ROWS>
struct StemGeom {
    static constexpr int kTH = kWaves * ROWS;                // output tile height
    static constexpr int kS1H = kTH + 2, kS1W = kTile + 2;   // conv1 tile
    static constexpr int kInH = kTH + 4, kInW = kTile + 4;   // input patch
    static constexpr int kInElems = 3 * kInH * kInW;         // + [kInElems] = 1.0, [kInElems + 1] = 0.0
    static constexpr int kInBytes = (kInElems + 8) * 2;
    static constexpr int kPix = C1 * 2 + 16;                 // bytes per conv1 pixel in LDS: an odd multiple of 16 B
    static constexpr int kKS2 = (9 * C1 + 31) / 32;
    static constexpr int kNT1 = C1 / 16, kNT2 = C / 16;
    static constexpr int kW2Bytes = kKS2 * kNT2 * 1024, kW1Bytes = kNT1 * 1024, kS1Bytes = kS1H * kS1W * kPix;
    static constexpr int kSmem = kW2Bytes + kW1Bytes + kS1Bytes + kInBytes + C * 4;
    static_assert(C1 % 16 == 0 && C % 32 == 0 && (kPix / 16) % 2 == 1 && C1 % 8 == 0, "stem geometry");
    static_assert(kSmem <= 160 * 1024, "LDS");
};
}  // namespace

template <int C1, int C, int CROP, int ROWS>
__global__ void __launch_bounds__(kWaves * 64) stem_fused_kernel(StemFusedArgs a) {
    typedef StemGeom<C1, C, CROP, ROWS> G;
    constexpr int kC1 = C1, kC = C, kPix = G::kPix, kKS2 = G::kKS2, kNT1 = G::kNT1, kNT2 = G::kNT2;
    constexpr int kW2Bytes = G::kW2Bytes, kW1Bytes = G::kW1Bytes, kS1Bytes = G::kS1Bytes;
    constexpr int kTH = G::kTH, kS1H = G::kS1H, kS1W = G::kS1W, kInH = G::kInH, kInW = G::kInW, kInElems = G::kInElems;
    constexpr int kInBytes = G::kInBytes;
    constexpr int kNIn = (kInElems + kWaves * 64 - 1) / (kWaves * 64);        // input elements per thread
    constexpr int kT1 = (kS1H * kS1W + 15) / 16;                               // conv1 token tiles
    constexpr int kR1 = (kT1 + kWaves - 1) / kWaves;                           // ... rounds over the waves
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
    const f16x8 *w2l = reinterpret_cast<const f16x8 *>(smem_s);
    unsigned char *s1l = smem_s + kW2Bytes + kW1Bytes;
    f16 *inl = reinterpret_cast<f16 *>(s1l + kS1Bytes);
    float *b2l = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(inl) + kInBytes);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r16 = lane & 15;
    const int grp = lane >> 4;

    {
        f16x8 *dst = reinterpret_cast<f16x8 *>(smem_s);
        const f16x8 *src2 = reinterpret_cast<const f16x8 *>(a.w2);
        for (int i = tid; i < kW2Bytes / 16; i += kWaves * 64) dst[i] = src2[i];
        if (tid < kC) b2l[tid] = a.b2[tid];
        if (tid == 0) { inl[kInElems] = (f16)1.0f; inl[kInElems + 1] = (f16)0.0f; }
    }

    // conv1 B operand: k = 8 grp + j -> (ci, ky, kx) = (k / 9, (k % 9) / 3, k % 3); k = 27: the constant one; k > 27: zero
    // element = ioff + (pixel offset where the tap is a real one): the two constants (k = 27: one, beyond: zero) do not move.
    // Only lane group 3 holds constants (k = 24 + j: j = 3 is the one, j > 3 zeros): `cmask` bit j set = slot j does not move
    int ioff[8];
    unsigned cmask = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * grp + j;
        const int ci = k / 9, r = k - 9 * ci;
        ioff[j] = k < 27 ? ci * (kInH * kInW) + (r / 3) * kInW + (r % 3) : (k == 27 ? kInElems : kInElems + 1);
        if (k >= 27) cmask |= 1u << j;
    }
    // conv2 B operand: k0 = 32 ks + 8 grp is one 8-channel run of tap k0 / 48 (48 % 8 == 0: a run never straddles taps);
    // the zero-weight tail k >= 432 re-reads tap 0 (finite data x 0)
    int koff[kKS2];
#pragma unroll
    for (int ks = 0; ks < kKS2; ++ks) {
        int k0 = 32 * ks + 8 * grp;
        if (k0 >= 9 * kC1) k0 = 0;
        const int tap = k0 / kC1, c = k0 - tap * kC1;
        koff[ks] = ((tap / 3) * kS1W + tap % 3) * kPix + c * 2;
    }
    const f16x8 *w1g = reinterpret_cast<const f16x8 *>(a.w1);

    const int S = a.T - 4 - 2 * CROP;
    const int ntx = (S + kTile - 1) / kTile, nty = (S + kTH - 1) / kTH;
    const int n_tiles = a.B * ntx * nty;

    // input patch element e = tid + 512 i of tile t -> fp32 value (replicate clamp at the frame border in frame mode)
    auto fetch_in = [&](int t, float (&v)[kNIn]) {
        const int txt = t % ntx, t2 = t / ntx;
        const int tyt = t2 % nty, b = t2 / nty;
        int fy = 0, fx = 0;
        if (a.frame_mode) {
            const int k = a.tile_begin + b;
            const int ti = k / a.wb, tj = k - ti * a.wb;
            fy = ti * a.istep - a.pad_t;
            fx = tj * a.istep - a.pad_l;
        }
#pragma unroll
        for (int i = 0; i < kNIn; ++i) {
            const int e = tid + kWaves * 64 * i;
            v[i] = 0.f;
            if (e < kInElems) {
                const int ci = e / (kInH * kInW), r = e - ci * (kInH * kInW);
                const int iy = r / kInW, ix = r - iy * kInW;
                const int yy = min(kTH * tyt + CROP + iy, a.T - 1), xx = min(kTile * txt + CROP + ix, a.T - 1);
                if (a.frame_mode) {
                    const int sy = min(max(fy + yy, 0), a.H - 1), sx = min(max(fx + xx, 0), a.W - 1);
                    v[i] = a.x[((long)ci * a.H + sy) * a.W + sx];
                } else {
                    v[i] = a.x[(((long)b * 3 + ci) * a.T + yy) * a.T + xx];
                }
            }
        }
    };
    auto tmap = [&](int t) { return a.rev ? n_tiles - 1 - t : t; };

    float pin[kNIn];
    if ((int)blockIdx.x < n_tiles) fetch_in(tmap(blockIdx.x), pin);
    __syncthreads();

#pragma unroll 1
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int tm = tmap(t);
        const int txt = tm % ntx, t2 = tm / ntx;
        const int tyt = t2 % nty, b = t2 / nty;
        // conv1's three weight fragments (3 KiB, L2-resident) are re-read per tile — requested here, in front of the staging barrier —
        // instead of living in 12 registers through conv2 (with ROWS = 3 the kernel sits at the 256-register limit)
        f16x8 w1f[kNT1];
#pragma unroll
        for (int nt = 0; nt < kNT1; ++nt) w1f[nt] = w1g[nt * 64 + lane];
        // ---- 1. input patch -> LDS (fp16) ---------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < kNIn; ++i) {
            const int e = tid + kWaves * 64 * i;
            if (e < kInElems) inl[e] = (f16)pin[i];
        }
        __syncthreads();
        // ---- 2. conv1 on the MFMA: 21 tiles of 16 pixels x 48 channels ---------------------------------------------------
#pragma unroll
        for (int i = 0; i < kR1; ++i) {
            const int mt = wave + kWaves * i;
            if (mt * 16 < kS1H * kS1W) {
                const int p = min(mt * 16 + r16, kS1H * kS1W - 1);
                const int py = p / kS1W, px = p - py * kS1W;
                const int pb = py * kInW + px;
                f16x8 bfrag;
#pragma unroll
                for (int j = 0; j < 8; ++j) bfrag[j] = inl[ioff[j] + ((cmask >> j) & 1u ? 0 : pb)];
                const bool wr = mt * 16 + r16 < kS1H * kS1W;
#pragma unroll
                for (int nt = 0; nt < kNT1; ++nt) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    acc = MFMA_16x16x32(w1f[nt], bfrag, acc);
                    f16x4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = (f16)(acc[q] >= 0.f ? acc[q] : acc[q] * a.slope);
                    if (wr) *reinterpret_cast<f16x4 *>(s1l + p * kPix + (16 * nt + 4 * grp) * 2) = o;
                }
            }
        }
        __syncthreads();
        // the next tile's input travels while conv2 runs
        if (t + (int)gridDim.x < n_tiles) fetch_in(tmap(t + gridDim.x), pin);
        // ---- 3. conv2: rows ROWS wave .. of the tile; B fragments from the conv1 tile, W2 resident ------------------------------
        f32x4 acc[kNT2][ROWS];
#pragma unroll
        for (int nt = 0; nt < kNT2; ++nt) {
            const f32x4 bb = *reinterpret_cast<const f32x4 *>(b2l + 16 * nt + 4 * grp);
#pragma unroll
            for (int f = 0; f < ROWS; ++f) acc[nt][f] = bb;
        }
        const unsigned char *bp0 = s1l + ((ROWS * wave) * kS1W + r16) * kPix;
#pragma unroll
        for (int ks = 0; ks < kKS2; ++ks) {
            f16x8 bf[ROWS];
#pragma unroll
            for (int f = 0; f < ROWS; ++f) bf[f] = *reinterpret_cast<const f16x8 *>(bp0 + f * (kS1W * kPix) + koff[ks]);
#pragma unroll
            for (int nt = 0; nt < kNT2; ++nt) {
                const f16x8 w = w2l[(ks * kNT2 + nt) * 64 + lane];
#pragma unroll
                for (int f = 0; f < ROWS; ++f) acc[nt][f] = MFMA_16x16x32(w, bf[f], acc[nt][f]);
            }
        }
        const int ox = kTile * txt + r16;
#pragma unroll
        for (int f = 0; f < ROWS; ++f) {
            const int oy = kTH * tyt + ROWS * wave + f;
            const bool ok = oy < S && ox < S;
            f16 *op = a.out + (((long)b * S + min(oy, S - 1)) * S + min(ox, S - 1)) * kC + pair_run_channel(grp);
#pragma unroll
            for (int p = 0; p < kNT2 / 2; ++p) {
                f16x4 oa, ob;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float u = acc[2 * p][f][q], v = acc[2 * p + 1][f][q];
                    oa[q] = (f16)(u >= 0.f ? u : u * a.slope);
                    ob[q] = (f16)(v >= 0.f ? v : v * a.slope);
                }
                const f16x8 o = pair_to_run(oa, ob);
                if (ok) *reinterpret_cast<f16x8 *>(op + 32 * p) = o;
            }
        }
        __syncthreads();          // the conv1 tile and the input patch are rewritten by the next trip
    }
}

bool stem_fused_supported(int C1, int C) { return (C1 == 48 && C == 96) || (C1 == 32 && C == 64); }

template <int C1, int C, int CROP, int ROWS>
static int launch_stem_t(const StemFusedArgs &a, hipStream_t s, const char *name) {
    typedef StemGeom<C1, C, CROP, ROWS> G;
    const int S = a.T - 4 - 2 * CROP;
    NUNIF_REQUIRE(S > 0, "stem: tile size %d too small", a.T);
    const int ntx = (S + kTile - 1) / kTile, nty = (S + G::kTH - 1) / G::kTH;
    const long n_tiles = (long)a.B * ntx * nty;
    if (n_tiles == 0) return NUNIF_HIP_OK;
    const double px = (double)a.B * S * S;
    ProfScope ps(name, s, 2.0 * px * (27.0 * C1 + 9.0 * C1 * C), px * (12.0 + C * 2.0));
    static bool configured = false;
    if (!configured) {
        NUNIF_HIP_CHECK(hipFuncSetAttribute((const void *)stem_fused_kernel<C1, C, CROP, ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            G::kSmem));
        configured = true;
    }
    // persistent: one workgroup per CU, two where the LDS image allows it (the <32, 64> form is 67 KiB)
    const unsigned grid = (unsigned)std::min<long>(n_tiles, G::kSmem <= 80 * 1024 ? 512 : 256);
    stem_fused_kernel<C1, C, CROP, ROWS><<<grid, kWaves * 64, G::kSmem, s>>>(a);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

int launch_stem_fused(const StemFusedArgs &a, hipStream_t s) {
    // rows per wave: 3 (24 x 16 tiles) unless NUNIF_STEM_ROWS=2 (round 4's 16 x 16 tiles; A/B runs): 305 -> 267 us on the 1080p
    // frame's 45 tiles (profiles/r05u_ab_*).  Same products in the same order per output pixel: the two forms give the same bits.
    static const int rows = getenv("NUNIF_STEM_ROWS") && atoi(getenv("NUNIF_STEM_ROWS")) == 2 ? 2 : 3;
    // cunet's <32, 64> form keeps 2: with 3 it needs 176 registers, i.e. one workgroup per CU instead of the two its 79 KiB of LDS
    // allow — measured 0.386 vs 0.354 ms per frame (profiles/r05u_cunet_*); the swin stem is one workgroup per CU either way
    static const int rows_c = getenv("NUNIF_STEM_ROWS_CUNET") && atoi(getenv("NUNIF_STEM_ROWS_CUNET")) == 3 ? 3 : 2;
    if (a.C1 == 32 && a.C == 64 && a.crop == 0)
        return rows_c == 3 ? launch_stem_t<32, 64, 0, 3>(a, s, "stem_fused_kernel<32,64>") : launch_stem_t<32, 64, 0, 2>(a, s, "stem_fused_kernel<32,64>");
    NUNIF_REQUIRE((a.C1 == 48 || a.C1 == 0) && (a.C == 96 || a.C == 0), "stem: C1=%d C=%d unsupported", a.C1, a.C);
    return rows == 3 ? launch_stem_t<48, 96, 6, 3>(a, s, "stem_fused_kernel") : launch_stem_t<48, 96, 6, 2>(a, s, "stem_fused_kernel");
}

}  // namespace nunif
