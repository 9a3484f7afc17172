// Tile grid, tile gather and the single-pass overlap-blend stitcher.
//
// Reference: nunif/utils/seam_blending.py — create_config :109-143, create_blend_filter :146-153,
// tiled_render's F.pad(replicate) + slicing :82,:90, update :156-174, get_output :39-40.
//
// MI355X design: HBM-bound.  The reference keeps two frame-sized fp32 accumulators (pixels, weights) and
// touches them ~22 tile-sized passes per tile.  Here every tile output of the frame stays resident (288 GB HBM)
// and ONE kernel writes each output pixel once: it finds the <=4 covering tiles from integer grid math and
// replays the reference's running-mean recurrence over them in the reference's row-major tile order, with
// un-contracted fp32 mul/add/div, so for identical tile inputs the result is bit-identical to the reference's
// cumulative update.  Traffic: read each tile pixel that lands in the frame once + write each output once
// (~24.5 B per output pixel at 3 channels fp32).
#include <cstdlib>

#include "common.h"

namespace nunif {

__global__ void __launch_bounds__(256)
gather_tiles_kernel(const float *__restrict__ x, float *__restrict__ tiles, int C, int H, int W, int T,
                    int wb, int istep, int pad_t, int pad_l, int tile_begin, long total) {
    // one thread per destination element; destination-coalesced, source rows are contiguous runs too
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int tx = idx % T;
    long r = idx / T;
    int ty = r % T;
    r /= T;
    int c = r % C;
    int k = (int)(r / C) + tile_begin;
    int ti = k / wb, tj = k % wb;
    int sy = min(max(ti * istep + ty - pad_t, 0), H - 1);   // replicate padding == clamp
    int sx = min(max(tj * istep + tx - pad_l, 0), W - 1);
    tiles[idx] = x[((long)c * H + sy) * W + sx];
}

struct StitchParams {
    int C, y_h, y_w, hb, wb, ostep, To, blend;
    int fast1;       // 1: pixel groups with one covering tile skip the recurrence (NUNIF_STITCH_FAST=0 turns it off: A/B runs)
    int y0;          // first output row of this launch (tile-row sharding: a band of the image), 0 for the whole image
    int dst_rows;    // rows per channel plane of the destination: y_h for the whole image, the band height for a compact band
    float ramp[64];
};

__device__ __forceinline__ float ramp_at(const StitchParams &p, int t) {
    int d = min(t, p.To - 1 - t);
    return d < p.blend ? p.ramp[d] : 1.0f;
}

// CC > 0: the channel count is a compile-time constant and the channels of a pixel group travel TOGETHER — all CC tile reads of a
// covering tile are in flight at once and the blend weights (which do not depend on the channel) are computed once.  Round 4's form
// walked the channels one after the other, one 16-byte load in flight per thread: 204 MB in 47 us = 0.53 of the HBM peak, bound by
// memory-level parallelism (a thread's whole life was three dependent load -> store round trips).  Per channel the arithmetic and
// its order are unchanged (bit-identical to the reference's cumulative update, tests/test_gpu_stitch.py).  CC = 0: any C, the old walk.
template <int VEC, int CC>
__global__ void __launch_bounds__(256)
stitch_kernel(const float *__restrict__ tiles, float *__restrict__ y, StitchParams p) {
    // HIP's __fmul_rn/__fadd_rn are plain operators: without this the compiler contracts P*a + t*b into an FMA
    // and the result is 1 ulp off the reference's separately rounded mul/mul/add (seen on gfx950).
#pragma clang fp contract(off)
    const int xg = blockIdx.x * blockDim.x + threadIdx.x;   // group of VEC pixels along x
    const int Y = blockIdx.y + p.y0;                        // (y0 > 0: a row band of the image, tile-row sharding)
    const int X0 = xg * VEC;
    if (X0 >= p.y_w) return;
    const int i_hi = min(Y / p.ostep, p.hb - 1);
    const int i_lo = Y < p.To ? 0 : (Y - p.To) / p.ostep + 1;
    const int j_hi = min(X0 / p.ostep, p.wb - 1);
    const int j_lo = X0 < p.To ? 0 : (X0 - p.To) / p.ostep + 1;
    const long plane = (long)p.To * p.To;
    constexpr int NC = CC > 0 ? CC : 1;
    const int c_outer = CC > 0 ? 1 : p.C;

    auto load_vec = [&](const float *src, float (&t)[VEC]) {
        if constexpr (VEC >= 4) {
#pragma unroll
            for (int u = 0; u < VEC / 4; ++u) {
                const float4 q = reinterpret_cast<const float4 *>(src)[u];
                t[4 * u] = q.x; t[4 * u + 1] = q.y; t[4 * u + 2] = q.z; t[4 * u + 3] = q.w;
            }
        } else {
            t[0] = src[0];
        }
    };
    for (int c0 = 0; c0 < c_outer; ++c0) {
        float P[NC][VEC], Wt[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            Wt[v] = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) P[c][v] = 0.f;
        }
        if (p.fast1 && p.blend > 0 && i_lo == i_hi && j_lo == j_hi) {
            // ONE covering tile (> 95 % of a frame): the recurrence starts from P = 0, W = 0, so a = 0 / F = 0, b = 1 and
            // P = 0 * 0 + t * 1 = t whatever the ramp value F > 0 is — the tile's pixel, without the four divisions
            const int ty = Y - p.ostep * i_lo, tx = X0 - p.ostep * j_lo;
            const float *src = tiles + ((long)(i_lo * p.wb + j_lo) * p.C + c0) * plane + (long)ty * p.To + tx;
#pragma unroll
            for (int c = 0; c < NC; ++c) load_vec(src + c * plane, P[c]);
        } else if (p.blend > 0) {
            for (int i = i_lo; i <= i_hi; ++i) {
                const int ty = Y - p.ostep * i;
                const float ry = ramp_at(p, ty);
                for (int j = j_lo; j <= j_hi; ++j) {
                    const int tx = X0 - p.ostep * j;
                    const float *src = tiles + ((long)(i * p.wb + j) * p.C + c0) * plane + (long)ty * p.To + tx;
                    float t[NC][VEC];
#pragma unroll
                    for (int c = 0; c < NC; ++c) load_vec(src + c * plane, t[c]);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        // seam_blending.py:163-168, same operation order, no FMA contraction
                        const float F = fminf(ry, ramp_at(p, tx + v));
                        // plain operators on purpose: they are compiled under contract(off) above, whereas the
                        // __fmul_rn/__fadd_rn header wrappers carry their own 'contract' flag and still fuse
                        const float w_new = Wt[v] + F;
                        const float a = Wt[v] / w_new;
                        const float b = 1.0f - a;
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            const float pa = P[c][v] * a;
                            const float tb = t[c][v] * b;
                            P[c][v] = pa + tb;
                        }
                        Wt[v] = w_new;
                    }
                }
            }
        } else {
            // blend_size == 0: plain overwrite in row-major order -> the last covering tile wins (:170-172)
            const int ty = Y - p.ostep * i_hi, tx = X0 - p.ostep * j_hi;
            const float *src = tiles + ((long)(i_hi * p.wb + j_hi) * p.C + c0) * plane + (long)ty * p.To + tx;
#pragma unroll
            for (int c = 0; c < NC; ++c) load_vec(src + c * plane, P[c]);
        }
        float *dst = y + ((long)c0 * p.dst_rows + (Y - (p.dst_rows == p.y_h ? 0 : p.y0))) * p.y_w + X0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float *d = dst + (long)c * p.dst_rows * p.y_w;
            if constexpr (VEC >= 4) {
#pragma unroll
                for (int u = 0; u < VEC / 4; ++u) {
                    float4 o;
                    o.x = fminf(fmaxf(P[c][4 * u], 0.f), 1.f); o.y = fminf(fmaxf(P[c][4 * u + 1], 0.f), 1.f);
                    o.z = fminf(fmaxf(P[c][4 * u + 2], 0.f), 1.f); o.w = fminf(fmaxf(P[c][4 * u + 3], 0.f), 1.f);
                    reinterpret_cast<float4 *>(d)[u] = o;
                }
            } else {
                d[0] = fminf(fmaxf(P[c][0], 0.f), 1.f);
            }
        }
    }
}

int launch_stitch(const float *tile_out, float *y, const nunif_tile_grid *g, int C, hipStream_t s, int y0, int rows,
                  int compact) {
    StitchParams p;
    if (rows < 0) rows = g->y_h - y0;
    p.dst_rows = compact ? rows : g->y_h;
    NUNIF_REQUIRE(y0 >= 0 && rows >= 0 && y0 + rows <= g->y_h, "stitch: row band [%d, %d) outside the image", y0, y0 + rows);
    if (rows == 0) return NUNIF_HIP_OK;
    p.y0 = y0;
    p.fast1 = !(getenv("NUNIF_STITCH_FAST") && atoi(getenv("NUNIF_STITCH_FAST")) == 0);
    p.C = C; p.y_h = g->y_h; p.y_w = g->y_w; p.hb = g->h_blocks; p.wb = g->w_blocks;
    p.ostep = g->output_tile_step; p.To = g->out_tile_size; p.blend = g->blend_size;
    NUNIF_REQUIRE(g->blend_size <= 64, "blend_size %d > 64 unsupported", g->blend_size);
    NUNIF_REQUIRE(p.ostep > 0 && p.To > 0, "bad tile grid");
    nunif_hip_blend_ramp(g->blend_size, p.ramp);
    // with blend==0 and overlapping tiles the "last tile wins" rule differs per pixel inside a VEC group only
    // if a tile boundary is not VEC-aligned; the alignment test below covers that too.
    const bool vec = (p.To % 4 == 0) && (p.ostep % 4 == 0) && (p.y_w % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(tile_out) | reinterpret_cast<uintptr_t>(y)) % 16 == 0);
    const double bytes = (double)C * rows * p.y_w * 4.0 * 2.0;
    ProfScope ps(vec ? "stitch_kernel<4>" : "stitch_kernel<1>", s, 0.0, bytes);
    static const bool together = !(getenv("NUNIF_STITCH_TOGETHER") && atoi(getenv("NUNIF_STITCH_TOGETHER")) == 0);   // A/B runs
    // (8 pixels per thread measured SLOWER than 4 — 44.1 vs 37.6 us on the 1080p 2x frame, profiles/r05e_*: half the waves, and
    //  the waves were what kept the loads in flight; kept for A/B runs only)
    static const int vec8_on = getenv("NUNIF_STITCH_VEC8") ? atoi(getenv("NUNIF_STITCH_VEC8")) : 0;
    const bool vec8 = vec && vec8_on && C == 3 && together && (p.To % 8 == 0) && (p.ostep % 8 == 0) && (p.y_w % 8 == 0) &&
                      ((reinterpret_cast<uintptr_t>(tile_out) | reinterpret_cast<uintptr_t>(y)) % 32 == 0);
    if (vec8) {
        // 8 pixels x 3 channels per thread: six 16-byte loads in flight
        dim3 grid(cdiv(p.y_w / 8, 128), rows);
        stitch_kernel<8, 3><<<grid, 128, 0, s>>>(tile_out, y, p);
    } else if (vec) {
        // a row is y_w / 4 pixel groups: 960 for a 1080p 2x frame = 3.75 workgroups of 256 (a quarter of every fourth one idle), but
        // exactly 5 of 192 — take the largest wave multiple that divides the row (NUNIF_STITCH_BS forces one; A/B runs)
        static const int bs_env = getenv("NUNIF_STITCH_BS") ? atoi(getenv("NUNIF_STITCH_BS")) : 0;
        int bs = 256;
        const int groups = p.y_w / 4;
        if (bs_env == 64 || bs_env == 128 || bs_env == 192 || bs_env == 256) bs = bs_env;
        else if (groups % 256 != 0) { if (groups % 192 == 0) bs = 192; else if (groups % 128 == 0) bs = 128; }
        dim3 grid(cdiv(groups, bs), rows);
        if (C == 3 && together) stitch_kernel<4, 3><<<grid, bs, 0, s>>>(tile_out, y, p);
        else stitch_kernel<4, 0><<<grid, bs, 0, s>>>(tile_out, y, p);
    } else {
        dim3 grid(cdiv(p.y_w, 256), rows);
        if (C == 3 && together) stitch_kernel<1, 3><<<grid, 256, 0, s>>>(tile_out, y, p);
        else stitch_kernel<1, 0><<<grid, 256, 0, s>>>(tile_out, y, p);
    }
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

}  // namespace nunif

using namespace nunif;

extern "C" int nunif_hip_tile_grid_init(int32_t x_h, int32_t x_w, int32_t scale, int32_t offset,
                                        int32_t tile_size, int32_t blend_size, nunif_tile_grid *g) {
    NUNIF_REQUIRE(g != nullptr, "grid is NULL");
    NUNIF_REQUIRE(x_h > 0 && x_w > 0 && scale > 0 && offset >= 0 && tile_size > 0 && blend_size >= 0,
                  "tile_grid_init: bad arguments");
    const int io = (offset + scale - 1) / scale;        // math.ceil(offset / scale)
    const int ib = (blend_size + scale - 1) / scale;    // math.ceil(blend_size / scale)
    const int step = tile_size - (io * 2 + ib);
    NUNIF_REQUIRE(step > 0, "tile_size %d too small for offset %d / blend %d", tile_size, offset, blend_size);
    // the reference's while-loops (:119-124), literally
    int hb = 0, wb = 0, in_h = 0, in_w = 0;
    while (in_h < x_h + io * 2) { in_h = hb * step + tile_size; hb++; }
    while (in_w < x_w + io * 2) { in_w = wb * step + tile_size; wb++; }
    g->x_h = x_h; g->x_w = x_w; g->scale = scale; g->offset = offset; g->tile_size = tile_size;
    g->blend_size = blend_size;
    g->y_h = x_h * scale; g->y_w = x_w * scale;         // floor(x*scale), integer scale
    g->h_blocks = hb; g->w_blocks = wb;
    g->pad_l = io; g->pad_r = in_w - (x_w + io); g->pad_t = io; g->pad_b = in_h - (x_h + io);
    g->y_buffer_h = in_h * scale; g->y_buffer_w = in_w * scale;
    g->input_tile_step = step; g->output_tile_step = step * scale;
    g->out_tile_size = tile_size * scale - offset * 2;
    return NUNIF_HIP_OK;
}

extern "C" int nunif_hip_blend_ramp(int32_t blend_size, float *ramp) {
    // create_blend_filter :150-152 — pad ring i (0 = innermost) carries 1 - (1/(b+1))*(i+1) evaluated in double;
    // ring i ends up (blend_size-1-i) pixels from the border.
    for (int d = 0; d < blend_size; ++d) {
        const int i = blend_size - 1 - d;
        const double value = 1.0 - (1.0 / (double)(blend_size + 1)) * (double)(i + 1);
        ramp[d] = (float)value;
    }
    return NUNIF_HIP_OK;
}

extern "C" int nunif_hip_gather_tiles(const float *x, float *tiles, const nunif_tile_grid *g, int32_t C,
                                      int32_t tile_begin, int32_t n_tiles, void *stream) {
    NUNIF_REQUIRE(x && tiles && g, "gather_tiles: NULL pointer");
    NUNIF_REQUIRE(tile_begin >= 0 && n_tiles >= 0 && tile_begin + n_tiles <= g->h_blocks * g->w_blocks,
                  "gather_tiles: tile range out of grid");
    if (n_tiles == 0) return NUNIF_HIP_OK;
    const int T = g->tile_size;
    const long total = (long)n_tiles * C * T * T;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("gather_tiles_kernel", s, 0.0, (double)total * 8.0);
    gather_tiles_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(
        x, tiles, C, g->x_h, g->x_w, T, g->w_blocks, g->input_tile_step, g->pad_t, g->pad_l, tile_begin, total);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

extern "C" int nunif_hip_stitch_tiles(const float *tile_out, float *y, const nunif_tile_grid *g, int32_t C,
                                      void *stream) {
    NUNIF_REQUIRE(tile_out && y && g, "stitch_tiles: NULL pointer");
    return launch_stitch(tile_out, y, g, C, (hipStream_t)stream, 0, -1, 0);
}
