// The image heads of cunet (`conv_bottom`: Conv2d(64, 3, 3) VALID -> fp32 planar, `+ crop(z1)`, clamp; waifu2x/models/cunet.py:62,120,
// 183-196) in TAP-SCATTER form.
//
// As a conv (conv3_dma_kernel<1, 64>: N = 16 with 3 real outputs) every output pixel pulls 9 taps x 64 channels = 1 152 bytes of MFMA
// operands out of LDS for 18 MFMAs of which 3 / 16 of the rows are used: 0.16 ms per launch at 2.5 TB/s, LDS-read bound.  Here the
// contraction over the 64 channels comes FIRST: T[p][tap, c] = sum_ci W[c][ci][tap] x[p][ci] for every INPUT pixel p of a tile —
// one 64 -> 27 (padded 32) Linear, 4 MFMAs per 16 pixels, operands straight from global memory (a pixel is 128 contiguous bytes), no
// halo staging — and the output is the sum of nine of those values, out[y][x][c] = sum_tap T[(y + dy, x + dx)][tap, c], gathered from
// an fp32 LDS image of T (written once: 128 B per input pixel; read: 108 B per output pixel).  The same 576 products per output in
// fp32, in another order (channels first, then taps).
// Workgroup = 16 x 16 (or 16 x 32) outputs = 18 x 18 (18 x 34) input pixels (halo re-reads 1.27x / 1.19x, neighbours' — L2), 4 waves; T
// is kept channel-major [28][336 / 624] so that phase-1 writes (16 pixels x 4 lane groups) and phase-2 reads (consecutive pixels) are
// conflict-free.
#include <cstdlib>

#include "swin_kernels.h"

namespace nunif {

#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

template <int TW>                                        // tile = 16 x TW outputs (TW 16 or 32)
struct HeadGeom {
    static constexpr int kTH = 16, kIH = kTH + 2, kIW = TW + 2, kPix = kIH * kIW;             // 324 / 612 input pixels
    static constexpr int kGroups = (kPix + 15) / 16, kGW = (kGroups + 3) / 4;                   // 21 / 39 pixel groups, 6 / 10 per wave
    static constexpr int kRow = TW == 16 ? 336 : 624;                                           // >= 16 kGroups, = 16 or 48 (mod 64) words
    static constexpr int kNR = 28;                                                              // rows of T kept (27 used)
    static_assert(kRow >= 16 * kGroups && kRow % 32 == 16, "T rows: every group's 16 pixels fit, lane groups on different banks");
};

template <int TW>
__global__ void __launch_bounds__(256) cunet_head_kernel(CunetHeadArgs g) {
    using G = HeadGeom<TW>;
    constexpr int kTH = G::kTH, kIW = G::kIW, kPix = G::kPix, kGroups = G::kGroups, kGW = G::kGW, kRow = G::kRow;
    __shared__ float T[G::kNR * kRow];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, grp = lane >> 4;
    const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * kTH, b = blockIdx.z;
    const f16 *img = g.a + (long)b * g.Hi * g.Wi * 64;
    // the wave's pixel groups: all their operand loads are issued before the first MFMA (2 x 16 bytes per lane and group)
    f16x8 bf[kGW][2];
#pragma unroll
    for (int q = 0; q < kGW; ++q) {
        const int pi = min((wave + 4 * q) * 16 + r16, kPix - 1);
        const int ty = pi / kIW, tx = pi - ty * kIW;
        const int iy = min(ty0 + ty, g.Hi - 1), ix = min(tx0 + tx, g.Wi - 1);
        const f16 *src = img + ((long)iy * g.Wi + ix) * 64 + grp * 8;
        bf[q][0] = *reinterpret_cast<const f16x8 *>(src);
        bf[q][1] = *reinterpret_cast<const f16x8 *>(src + 32);
    }
    f16x8 wf[2][2];                                                  // [n-tile][k-step], make_conv's head packing
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wf[nt][ks] = *reinterpret_cast<const f16x8 *>(g.w + ((nt * 2 + ks) * 64 + lane) * 8);
#pragma unroll
    for (int q = 0; q < kGW; ++q) {
        const int gi = wave + 4 * q;
        if (gi >= kGroups) break;
        f32x4 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[nt] = MFMA_16x16x32(wf[nt][0], bf[q][0], acc[nt]);
            acc[nt] = MFMA_16x16x32(wf[nt][1], bf[q][1], acc[nt]);
        }
        // lane (pixel r16, group grp) holds T[n = 16 nt + 4 grp + r][pixel]; rows 28 .. 31 are padding
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(grp * 4 + r) * kRow + gi * 16 + r16] = acc[0][r];
        if (grp < 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(16 + grp * 4 + r) * kRow + gi * 16 + r16] = acc[1][r];
        }
    }
    __syncthreads();
    constexpr int kPer = kTH * TW / 256, kRows = 256 / TW;          // outputs per thread, tile rows covered by one pass of the threads
    const int ox = threadIdx.x % TW, oyb = threadIdx.x / TW;
    const float bb[3] = {g.bias[0], g.bias[1], g.bias[2]};
#pragma unroll
    for (int hh = 0; hh < kPer; ++hh) {
        const int oy = oyb + kRows * hh;
        const int gy = ty0 + oy, gx = tx0 + ox;
        const bool live = gy < g.Ho && gx < g.Wo;
        float add[3] = {0.f, 0.f, 0.f};
        if (g.add32 && live) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                add[c] = g.add32[(((long)b * 3 + c) * g.addH + gy + g.add_crop) * g.addW + gx + g.add_crop];
        }
        float o[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int pi = (oy + tap / 3) * kIW + ox + tap % 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] += T[(tap * 3 + c) * kRow + pi];
        }
        if (!live) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = o[c] + bb[c] + add[c];
            if (g.clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
            g.out32[(((long)b * 3 + c) * g.Ho + gy) * g.Wo + gx] = v;
        }
    }
}

bool cunet_head_supported(const CunetHeadArgs &g) {
    const bool off = getenv("NUNIF_CUNET_HEAD") && atoi(getenv("NUNIF_CUNET_HEAD")) == 0;          // read per call (tests A/B it)
    return !off && g.w && g.a && g.out32 && g.B > 0 && g.Ho == g.Hi - 2 && g.Wo == g.Wi - 2 && g.Ho > 0 && g.Wo > 0 && g.B <= 65535 &&
           (g.Ho + 15) / 16 <= 65535;
}

int launch_cunet_head(const CunetHeadArgs &g, hipStream_t s) {
    NUNIF_REQUIRE(cunet_head_supported(g), "cunet_head: unsupported shape");
    // 16 x 16 tiles: 37 KiB of LDS, four workgroups per CU; NUNIF_CUNET_HEAD_TW=32: 16 x 32 tiles (68 KiB, two per CU; halo 1.19x
    // instead of 1.27x)
    const int tw = getenv("NUNIF_CUNET_HEAD_TW") && atoi(getenv("NUNIF_CUNET_HEAD_TW")) == 32 ? 32 : 16;
    const dim3 grid((g.Wo + tw - 1) / tw, (g.Ho + 15) / 16, g.B);
    // per output pixel: 2 * 576 * 3 flops; bytes: the 64-channel input once + 12 output bytes (+ 12 of the added map)
    ProfScope ps("cunet_head_kernel", s, 2.0 * 576 * 3 * g.B * (double)g.Ho * g.Wo,
                 (double)g.B * ((double)g.Hi * g.Wi * 128.0 + (double)g.Ho * g.Wo * (g.add32 ? 24.0 : 12.0)));
    if (tw == 32) cunet_head_kernel<32><<<grid, 256, 0, s>>>(g);
    else cunet_head_kernel<16><<<grid, 256, 0, s>>>(g);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

}  // namespace nunif
