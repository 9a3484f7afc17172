// iw3 depth-side kernels for gfx950: antialiased separable resize, edge-weighted depth dilation, min-max normalise.
//
// Reference: iw3/dilation.py — dilate_edge :116-142, edge_weight :101-113, gaussian_blur :30-38, dilate :41-46;
// iw3/depth_anything_model.py batch_preprocess :69-110 (bilinear antialias resize + clamp + ImageNet normalise);
// iw3/forward_warp.py:147-148 (depth -> frame size, bilinear + align_corners + antialias);
// iw3/depth_scaler.py minmax_normalize :4-17;  ATen upsample_bilinear2d_aa / upsample_bicubic2d_aa
// (UpSampleKernel.cpp, _compute_indices_min_size_weights_aa) for the resampling semantics.
//
// All HBM-bound.  dilate_edge: the reference runs ~12 full-frame ATen ops per iteration; here an iteration is one
// statistics pass (range mean / variance / min / max per image, fp64 block-reduced atomics) and one apply pass that
// recomputes the 3x3 range, the Gaussian and the max-pool from a 5x5 neighbourhood in registers.
#include <cstdlib>

#include "common.h"

namespace nunif {

// ---- antialiased separable resize -----------------------------------------------------------------------------------
__device__ __forceinline__ float aa_filter(float x, int bicubic) {
    if (x < 0.f) x = -x;
    if (!bicubic) return x < 1.f ? 1.f - x : 0.f;
    const float a = -0.5f;                                   // antialiased bicubic uses a = -0.5 (non-aa: -0.75)
    if (x < 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
    if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * a;
    return 0.f;
}

// one axis: out[n][o][i] = sum_j w_j in[n][xmin+j][i]  (stride_in/stride_out select row- or column-wise)
__global__ void __launch_bounds__(256)
resize_aa_axis_kernel(const float *__restrict__ in, float *__restrict__ out, long planes, int len_in, int len_out,
                      int other, int axis_is_x, float scale, int bicubic, int do_clamp, int norm_channels,
                      float m0, float m1, float m2, float s0, float s1, float s2) {
    // thread = one output element; index layout [plane][y][x]
    const long total = planes * (long)len_out * other;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    int o, q;           // o = output index along the resized axis, q = index along the other axis
    long plane;
    if (axis_is_x) { o = (int)(id % len_out); const long t = id / len_out; q = (int)(t % other); plane = t / other; }
    else { q = (int)(id % other); const long t = id / other; o = (int)(t % len_out); plane = t / len_out; }
    const float interp = bicubic ? 4.f : 2.f;
    const float support = scale >= 1.f ? (interp * 0.5f) * scale : interp * 0.5f;
    const float invscale = scale >= 1.f ? 1.f / scale : 1.f;
    const float center = scale * ((float)o + 0.5f);          // align_corners_delta = 0 whenever antialias is on
    int xmin = (int)(center - support + 0.5f);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5f);
    if (xmax > len_in) xmax = len_in;
    const int xsize = xmax - xmin;
    float total_w = 0.f;
    for (int j = 0; j < xsize; ++j) total_w += aa_filter(((float)(j + xmin) - center + 0.5f) * invscale, bicubic);
    const float norm = total_w != 0.f ? 1.f / total_w : 0.f;
    const float *src = in + plane * (long)len_in * other;
    float acc = 0.f;
    for (int j = 0; j < xsize; ++j) {
        const float w = aa_filter(((float)(j + xmin) - center + 0.5f) * invscale, bicubic) * norm;
        const long off = axis_is_x ? (long)q * len_in + (xmin + j) : (long)(xmin + j) * other + q;
        acc += w * src[off];
    }
    if (do_clamp) acc = fminf(fmaxf(acc, 0.f), 1.f);
    if (norm_channels > 0) {    // x.sub_(mean).div_(std), depth_anything_model.py:107-109
        const int c = (int)(plane % norm_channels);
        acc = (acc - (c == 0 ? m0 : (c == 1 ? m1 : m2))) / (c == 0 ? s0 : (c == 1 ? s1 : s2));
    }
    out[id] = acc;
}

// The generic kernel above evaluates the filter twice per tap and output (once for the normaliser, once for the weight): at
// 1080p -> 392 x 686 bicubic that is 24 cubic evaluations for each of 8.9 M intermediate pixels — ~70 us of pure VALU work for
// weights that only depend on the output COLUMN — and its 4-byte reads at a 2.8-element lane stride touch six cache lines per
// wave-load.  The two kernels below do the same arithmetic in the same order (bit-identical results) with the weights
// computed once: a persistent workgroup of the row pass builds the [len_out][K] weight table in LDS and then streams input rows
// through LDS; a workgroup of the column pass owns one output row, whose <= K filter values are evaluated once, by K threads.
__global__ void __launch_bounds__(256)
resize_aa_rows_kernel(const float *__restrict__ in, float *__restrict__ out, long rows, int len_in, int len_out, float scale,
                      int bicubic, int K) {
    extern __shared__ float aa_sm[];
    float *wtab = aa_sm;                                      // [len_out][K]
    int *xm = reinterpret_cast<int *>(wtab + (long)len_out * K), *xs = xm + len_out;
    float *row = reinterpret_cast<float *>(xs + len_out);     // [len_in]
    const int tid = threadIdx.x;
    const float interp = bicubic ? 4.f : 2.f;
    const float support = scale >= 1.f ? (interp * 0.5f) * scale : interp * 0.5f;
    const float invscale = scale >= 1.f ? 1.f / scale : 1.f;
    for (int o = tid; o < len_out; o += 256) {
        const float center = scale * ((float)o + 0.5f);
        int xmin = (int)(center - support + 0.5f);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5f);
        if (xmax > len_in) xmax = len_in;
        const int xsize = xmax - xmin;
        float total_w = 0.f;
        for (int j = 0; j < xsize; ++j) total_w += aa_filter(((float)(j + xmin) - center + 0.5f) * invscale, bicubic);
        const float norm = total_w != 0.f ? 1.f / total_w : 0.f;
        for (int j = 0; j < xsize; ++j) wtab[o * K + j] = aa_filter(((float)(j + xmin) - center + 0.5f) * invscale, bicubic) * norm;
        xm[o] = xmin; xs[o] = xsize;
    }
    // the next row travels in registers (len_in <= 8 x 256, checked by the launcher) while this one is being filtered from LDS
    float nxt[8];
    auto fetch = [&](long r) {
        const float *src = in + r * len_in;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + 256 * u;
            nxt[u] = src[i < len_in ? i : len_in - 1];
        }
    };
    long r = blockIdx.x;
    if (r < rows) fetch(r);
    while (r < rows) {
        __syncthreads();                                      // table ready / previous row consumed
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (tid + 256 * u < len_in) row[tid + 256 * u] = nxt[u];
        __syncthreads();
        const long rn = r + gridDim.x;
        fetch(rn < rows ? rn : r);
        float *dst = out + r * len_out;
        for (int o = tid; o < len_out; o += 256) {
            const float *w = wtab + o * K, *x = row + xm[o];
            const int n = xs[o];
            float acc = 0.f;
            for (int j = 0; j < n; ++j) acc += w[j] * x[j];
            dst[o] = acc;
        }
        r = rn;
    }
}

__global__ void __launch_bounds__(256)
resize_aa_cols_kernel(const float *__restrict__ in, float *__restrict__ out, int len_in, int len_out, int other, float scale,
                      int bicubic, int do_clamp, int norm_channels, float m0, float m1, float m2, float s0, float s1, float s2) {
    __shared__ float f[256];
    const int o = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
    const long plane = blockIdx.z;
    const float interp = bicubic ? 4.f : 2.f;
    const float support = scale >= 1.f ? (interp * 0.5f) * scale : interp * 0.5f;
    const float invscale = scale >= 1.f ? 1.f / scale : 1.f;
    const float center = scale * ((float)o + 0.5f);
    int xmin = (int)(center - support + 0.5f);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5f);
    if (xmax > len_in) xmax = len_in;
    const int xsize = xmax - xmin;                            // <= 256: checked by the launcher
    if ((int)threadIdx.x < xsize) f[threadIdx.x] = aa_filter(((float)((int)threadIdx.x + xmin) - center + 0.5f) * invscale, bicubic);
    __syncthreads();
    if (q >= other) return;
    float total_w = 0.f;
    for (int j = 0; j < xsize; ++j) total_w += f[j];
    const float norm = total_w != 0.f ? 1.f / total_w : 0.f;
    const float *src = in + (plane * len_in + xmin) * (long)other + q;
    float acc = 0.f;
    for (int j = 0; j < xsize; ++j) {
        const float w = f[j] * norm;
        acc += w * src[(long)j * other];
    }
    if (do_clamp) acc = fminf(fmaxf(acc, 0.f), 1.f);
    if (norm_channels > 0) {    // x.sub_(mean).div_(std), depth_anything_model.py:107-109
        const int c = (int)(plane % norm_channels);
        acc = (acc - (c == 0 ? m0 : (c == 1 ? m1 : m2))) / (c == 0 ? s0 : (c == 1 ? s1 : s2));
    }
    out[(plane * len_out + o) * (long)other + q] = acc;
}

// ---- dilate_edge ------------------------------------------------------------------------------------------------------
// Statistics of the 3x3 range map (edge_weight :101-113 needs its mean, std, min and max over the whole image) travel as
// one PARTIAL per workgroup, reduced in a fixed order by every consumer workgroup: no atomics (round 1: 4 same-line atomics
// per workgroup serialised in L2 and WERE the kernel), no init launch, deterministic sums.
struct RangePartial { double sum, sumsq; float rmin, rmax; };

__device__ __forceinline__ float range3x3(const float *__restrict__ img, int H, int W, int y, int x) {
    float mx = -3.0e38f, mn = 3.0e38f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {       // max_pool2d pads with -inf: outside is ignored
                const float v = img[(long)yy * W + xx];
                mx = fmaxf(mx, v);
                mn = fminf(mn, v);
            }
        }
    return mx - mn;
}

// 256 threads -> one RangePartial in sh[0..3] order (wave shuffles, then the four waves in order); all threads get the result
__device__ __forceinline__ RangePartial block_range_reduce(double s, double s2, float mn, float mx) {
    __shared__ RangePartial sh[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o); s2 += __shfl_xor(s2, o);
        mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    __syncthreads();                                           // sh may still be read from a previous call
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = RangePartial{s, s2, mn, mx};
    __syncthreads();
    RangePartial r = sh[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        r.sum += sh[w].sum; r.sumsq += sh[w].sumsq; r.rmin = fminf(r.rmin, sh[w].rmin); r.rmax = fmaxf(r.rmax, sh[w].rmax);
    }
    return r;
}

__global__ void minmax_init_kernel(unsigned int *mm, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) { mm[2 * b] = 0xFFFFFFFFu; mm[2 * b + 1] = 0u; }
}

// statistics of the INPUT of the first iteration: grid (blocks, B), one partial per workgroup
__global__ void __launch_bounds__(256)
range_partials_kernel(const float *__restrict__ x, RangePartial *__restrict__ part, int H, int W) {
    const int b = blockIdx.y;
    const float *img = x + (long)b * H * W;
    double s = 0.0, s2 = 0.0;
    float mn = 3.0e38f, mx = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)H * W; i += (long)gridDim.x * 256) {
        const float r = range3x3(img, H, W, (int)(i / W), (int)(i % W));
        s += r; s2 += (double)r * r;
        mn = fminf(mn, r); mx = fmaxf(mx, r);
    }
    const RangePartial r = block_range_reduce(s, s2, mn, mx);
    if (threadIdx.x == 0) part[(long)b * gridDim.x + blockIdx.x] = r;
}

// One iteration of dilate_edge for a 64 x 8 output patch per workgroup — AND the range statistics of its own output, which
// are the next iteration's edge-weight normalisation: n iterations are n + 1 launches (round 1: 3 n, 12 us each on a
// 392 x 686 map, launch-bound).  The patch's input halo (3 px: blur 1 + pool 1 + one more ring) is staged in LDS once
// (replicate-clamped coordinates, which is exactly the blur's padding); the blur is evaluated once per position of the
// patch + 2 ring, the output on the patch + 1 ring (the ring is recomputed by the neighbours bit-identically: same code, same
// inputs), and the 3x3 range of the OUTPUT is taken on the patch itself.  ~1.5 global loads per pixel instead of 90.
constexpr int kDilTW = 64, kDilTH = 8;
__global__ void __launch_bounds__(256)
dilate_fused_kernel(const float *__restrict__ x, float *__restrict__ y, const RangePartial *__restrict__ pin, int n_in,
                    RangePartial *__restrict__ pout, int H, int W, int ky, int kx) {
    __shared__ float tin[kDilTH + 6][kDilTW + 6];
    __shared__ float tbl[kDilTH + 4][kDilTW + 4];
    __shared__ float tyv[kDilTH + 2][kDilTW + 2];
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * kDilTW, y0 = blockIdx.y * kDilTH;
    const float *img = x + (long)b * H * W;
    for (int t = threadIdx.x; t < (kDilTH + 6) * (kDilTW + 6); t += 256) {
        const int ty = t / (kDilTW + 6), tx = t % (kDilTW + 6);
        const int yy = min(max(y0 + ty - 3, 0), H - 1), xx = min(max(x0 + tx - 3, 0), W - 1);
        tin[ty][tx] = img[(long)yy * W + xx];
    }
    // the whole image's range statistics from the producer's partials (fixed order: thread-strided, then block_range_reduce)
    float mean, denom, w_min, w_scale;
    {
        double s = 0.0, s2 = 0.0;
        float mn = 3.0e38f, mx = 0.f;
        const RangePartial *pp = pin + (long)b * n_in;
        for (int i = threadIdx.x; i < n_in; i += 256) {
            const RangePartial q = pp[i];
            s += q.sum; s2 += q.sumsq; mn = fminf(mn, q.rmin); mx = fmaxf(mx, q.rmax);
        }
        const RangePartial st = block_range_reduce(s, s2, mn, mx);          // (its barriers also publish tin)
        const double n = (double)H * W;
        const double meand = st.sum / n;
        mean = (float)meand;
        double var = st.sumsq / n - meand * meand;
        if (var < 0.0) var = 0.0;
        denom = (float)sqrt(var) + 1e-6f;                                                  // dilation.py:107-108
        const float wa = fminf(fmaxf((st.rmin - mean) / denom, -3.f), 3.f), wb = fminf(fmaxf((st.rmax - mean) / denom, -3.f), 3.f);
        w_min = wa;
        w_scale = (wb - wa) + 1e-6f;                                                       // :109-110
    }
    for (int t = threadIdx.x; t < (kDilTH + 4) * (kDilTW + 4); t += 256) {
        const int ty = t / (kDilTW + 4), tx = t % (kDilTW + 4);
        const int cy = y0 + ty - 2, cx = x0 + tx - 2;
        float g = -3.0e38f;                                   // outside the image: the pool's -inf padding
        if (cy >= 0 && cy < H && cx >= 0 && cx < W) {
            g = 0.f;
#pragma unroll
            for (int gy = 0; gy < 3; ++gy)
#pragma unroll
                for (int gx = 0; gx < 3; ++gx) {
                    const float kw = (gy == 1 && gx == 1) ? 48.f / 256.f
                                     : ((gy == 1 || gx == 1) ? 31.f / 256.f : 21.f / 256.f);
                    // a clamped coordinate of an in-image centre stays inside the staged halo, and the halo itself
                    // was loaded with clamped coordinates: tin[ty+gy][tx+gx] IS img[clamp(cy+gy-1)][clamp(cx+gx-1)]
                    g += kw * tin[ty + gy][tx + gx];
                }
        }
        tbl[ty][tx] = g;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < (kDilTH + 2) * (kDilTW + 2); t += 256) {
        const int ty = t / (kDilTW + 2), tx = t % (kDilTW + 2);
        const int py = y0 + ty - 1, px = x0 + tx - 1;
        if (py < 0 || py >= H || px < 0 || px >= W) continue;          // never read: range3x3 skips out-of-image taps
        float rmx = -3.0e38f, rmn = 3.0e38f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = py + dy, xx = px + dx;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                    const float v = tin[ty + 2 + dy][tx + 2 + dx];
                    rmx = fmaxf(rmx, v);
                    rmn = fminf(rmn, v);
                }
            }
        const float wr = fminf(fmaxf(((rmx - rmn) - mean) / denom, -3.f), 3.f);
        const float w = (wr - w_min) / w_scale;
        // x2 = max_pool(gaussian_blur(x)) over a ky x kx window (blur: replicate pad; pool: -inf pad)
        float x2 = -3.0e38f;
        for (int dy = -(ky / 2); dy <= ky / 2; ++dy)
            for (int dx = -(kx / 2); dx <= kx / 2; ++dx) x2 = fmaxf(x2, tbl[ty + 1 + dy][tx + 1 + dx]);
        const float v = tin[ty + 2][tx + 2];
        const float o = (v * (1.f - w)) + (x2 * w);                                       // :121
        tyv[ty][tx] = o;
        if (ty >= 1 && ty <= kDilTH && tx >= 1 && tx <= kDilTW) y[(long)b * H * W + (long)py * W + px] = o;
    }
    if (!pout) return;
    __syncthreads();
    double s = 0.0, s2 = 0.0;
    float mn = 3.0e38f, mx = 0.f;
    for (int t = threadIdx.x; t < kDilTH * kDilTW; t += 256) {
        const int ly = t / kDilTW, lx = t % kDilTW;
        const int py = y0 + ly, px = x0 + lx;
        if (py >= H || px >= W) continue;
        float rmx = -3.0e38f, rmn = 3.0e38f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = py + dy, xx = px + dx;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                    const float v = tyv[ly + 1 + dy][lx + 1 + dx];
                    rmx = fmaxf(rmx, v);
                    rmn = fminf(rmn, v);
                }
            }
        const float r = rmx - rmn;
        s += r; s2 += (double)r * r;
        mn = fminf(mn, r); mx = fmaxf(mx, r);
    }
    const RangePartial r = block_range_reduce(s, s2, mn, mx);
    if (threadIdx.x == 0) pout[(long)b * gridDim.x * gridDim.y + (long)blockIdx.y * gridDim.x + blockIdx.x] = r;
}

// ---- per-image min-max normalise (depth_scaler.py:4-17, reset path: ema disabled) ----------------------------------------
__global__ void __launch_bounds__(256) minmax_stats_kernel(const float *__restrict__ x, float *mm, long n_per) {
    const int b = blockIdx.y;
    const float *img = x + (long)b * n_per;
    float mn = 3.0e38f, mx = -3.0e38f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_per; i += (long)gridDim.x * 256) {
        const float v = img[i];
        mn = fminf(mn, v); mx = fmaxf(mx, v);
    }
    __shared__ float sh_mn[256], sh_mx[256];
    sh_mn[threadIdx.x] = mn; sh_mx[threadIdx.x] = mx;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
            sh_mn[threadIdx.x] = fminf(sh_mn[threadIdx.x], sh_mn[threadIdx.x + st]);
            sh_mx[threadIdx.x] = fmaxf(sh_mx[threadIdx.x], sh_mx[threadIdx.x + st]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // order-preserving float -> uint so that atomicMin/Max work for negative values too
        auto key = [](float v) { unsigned int u = __float_as_uint(v); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
        atomicMin(reinterpret_cast<unsigned int *>(mm) + 2 * b, key(sh_mn[0]));
        atomicMax(reinterpret_cast<unsigned int *>(mm) + 2 * b + 1, key(sh_mx[0]));
    }
}

__global__ void __launch_bounds__(256) minmax_apply_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                           const float *mm, long n_per) {
    const int b = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_per) return;
    auto unkey = [](unsigned int k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); };
    const unsigned int *u = reinterpret_cast<const unsigned int *>(mm);
    const float mn = unkey(u[2 * b]), mx = unkey(u[2 * b + 1]);
    const float v = x[(long)b * n_per + i];
    // depth_scaler.py minmax_normalize: (x - min) / (max - min) with a guard for a flat map, clamped to [0,1]
    const float range = mx - mn;
    const float o = range > 0.f ? (v - mn) / range : v;
    y[(long)b * n_per + i] = fminf(fmaxf(o, 0.f), 1.f);
}

// ---- EMAMinMaxScaler on the device (iw3/depth_scaler.py:33-142) ---------------------------------------------------------------
// The reference keeps the look-ahead ring, the running extrema and the frame's own min / max as 0-dim device tensors and
// runs ~14 tiny ATen kernels + one host sync (`if scale > 0`) per frame.  Here the SAME arithmetic (fp32, separately
// rounded: this file is built with -ffp-contract=off) is one single-thread kernel on a small state block, and the
// normalisation reads lo / hi from that block — no host round trip.
//   state: [0 .. 2N)  MinMaxBuffer.data      [2N] min_value   [2N + 1] max_value
// The bookkeeping that is data-independent (count, whether the ring is filled, whether an EMA value exists) stays on the
// host, exactly as the reference's Python does it.
__global__ void ema_scaler_push_kernel(float *state, const float *mm_keys, int size, long count, int filled, int first,
                                       float decay, float one_minus_decay) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    auto unkey = [](unsigned int k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); };
    const unsigned int *u = reinterpret_cast<const unsigned int *>(mm_keys);
    const float mn = unkey(u[0]), mx = unkey(u[1]);
    if (count == 0) {                                           // MinMaxBuffer.add :41-45: the first sample fills the ring
        for (int i = 0; i < size; i += 2) { state[i] = mn; state[i + 1] = mx; }
    } else {
        state[count % size] = mn;
        state[(count + 1) % size] = mx;
    }
    if (!filled) return;
    float lo = state[0], hi = state[0];                         // get_minmax :60-61: amin / amax over the whole ring
    for (int i = 1; i < size; ++i) { lo = fminf(lo, state[i]); hi = fmaxf(hi, state[i]); }
    if (first) {
        state[size] = lo; state[size + 1] = hi;
    } else {                                                    // :110-111
        state[size] = decay * state[size] + one_minus_decay * lo;
        state[size + 1] = decay * state[size + 1] + one_minus_decay * hi;
    }
}

// flush() before any EMA value exists (:124-125): lo / hi = the ring's extrema
__global__ void ema_scaler_ring_minmax_kernel(float *state, int size) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float lo = state[0], hi = state[0];
    for (int i = 1; i < size; ++i) { lo = fminf(lo, state[i]); hi = fmaxf(hi, state[i]); }
    state[size] = lo; state[size + 1] = hi;
}

// minmax_normalize / max_normalize with given extrema (depth_scaler.py:4-30); lohi on the device
__global__ void __launch_bounds__(256) range_normalize_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                              const float *lohi, long n, int max_mode) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float lo = lohi[0], hi = lohi[1];
    const float v = x[i];
    float o;
    if (max_mode) o = hi > 0.f ? v / hi : v;
    else { const float scale = hi - lo; o = scale > 0.f ? (v - lo) / scale : v; }
    y[i] = o != o ? o : fminf(fmaxf(o, 0.f), 1.f);             // torch.clamp keeps NaN
}

// make_input_tensor (iw3/backward_warp.py:33-64, c = None): [depth | divergence plane | convergence plane] with the
// optional screen-border taper (linspace(0, 1, n) on the left, linspace(1, 0, n) on the right, multiplied in fp32)
__device__ __forceinline__ float linspace_at(float start, float end, int i, int n) {
    // torch.linspace(start, end, n)[i] in fp32, ATen's two-sided form: the first half counts up from `start`, the second
    // half down from `end` (RangeFactories: step = (end - start) / (n - 1))
    if (n == 1) return start;
    const float step = (end - start) / (float)(n - 1);
    return i < n / 2 ? start + step * (float)i : end - step * (float)(n - 1 - i);
}
__global__ void __launch_bounds__(256) make_input_planes_kernel(const float *__restrict__ depth, float *__restrict__ out,
                                                                int B, int H, int W, float dv, float cv, int border) {
    const long hw = (long)H * W;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * hw) return;
    const long b = i / hw, p = i - b * hw;
    const int x = (int)(p % W);
    float d = dv, c = cv;
    if (border > 0) {
        // left strip first, then the right strip on the result (a narrow map's strips may overlap), like the reference's
        // two in-place slice multiplications
        if (x < border) { const float w = linspace_at(0.0f, 1.0f, x, border); d = w * d; c = w * c; }
        if (x >= W - border) { const float w = linspace_at(1.0f, 0.0f, x - (W - border), border); d = w * d; c = w * c; }
    }
    float *o = out + b * 3 * hw;
    o[p] = depth[i];
    o[hw + p] = d;
    o[2 * hw + p] = c;
}

// torch.stack of up to 16 equally sized device buffers (the per-frame tensors of a batch) in ONE launch
struct StackArgs { const void *src[16]; };
__global__ void __launch_bounds__(256) stack_kernel(StackArgs a, uint4 *__restrict__ dst, long vec_each, int n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= vec_each) return;
    for (int k = 0; k < n; ++k) dst[(long)k * vec_each + i] = reinterpret_cast<const uint4 *>(a.src[k])[i];
}
__global__ void __launch_bounds__(256) stack_bytes_kernel(StackArgs a, unsigned char *__restrict__ dst, long bytes_each, int n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= bytes_each) return;
    for (int k = 0; k < n; ++k) dst[(long)k * bytes_each + i] = reinterpret_cast<const unsigned char *>(a.src[k])[i];
}

}  // namespace nunif

using namespace nunif;

extern "C" int nunif_hip_resize_aa(const float *x, float *y, float *tmp, int64_t planes, int32_t h_in, int32_t w_in,
                                   int32_t h_out, int32_t w_out, int32_t bicubic, int32_t align_corners,
                                   int32_t clamp01, const float *mean3, const float *std3, void *stream) {
    NUNIF_REQUIRE(x && y && tmp && planes > 0 && h_in > 0 && w_in > 0 && h_out > 0 && w_out > 0, "resize_aa: bad argument");
    // area_pixel_compute_scale: align_corners -> (in-1)/(out-1) (0 when out == 1), else in/out
    auto scale_of = [&](int in, int out) -> float {
        if (align_corners) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
        return (float)in / (float)out;
    };
    hipStream_t s = (hipStream_t)stream;
    const double bytes = (double)planes * 4.0 * ((double)h_in * w_in + 2.0 * h_in * w_out + (double)h_out * w_out);
    ProfScope ps("resize_aa", s, 0.0, bytes);
    auto taps = [&](float scale) {                               // upper bound of xsize = xmax - xmin
        const float support = (bicubic ? 2.f : 1.f) * (scale >= 1.f ? scale : 1.f);
        return (int)(2.f * support) + 3;
    };
    {   // width first: [planes][h_in][w_in] -> tmp [planes][h_in][w_out]
        const float sc = scale_of(w_in, w_out);
        const int K = taps(sc);
        const size_t smem = ((size_t)w_out * K + 2 * (size_t)w_out + (size_t)w_in) * 4;
        const long rows = planes * (long)h_in;
        if (smem <= 64 * 1024 && rows >= 2048 && w_in <= 2048) {                 // the table is worth building when a workgroup streams several rows
            resize_aa_rows_kernel<<<768, 256, smem, s>>>(x, tmp, rows, w_in, w_out, sc, bicubic, K);
        } else {
            const long total = planes * (long)h_in * w_out;
            resize_aa_axis_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(
                x, tmp, planes, w_in, w_out, h_in, 1, sc, bicubic, 0, 0, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f);
        }
        NUNIF_LAUNCH_CHECK();
    }
    if (taps(scale_of(h_in, h_out)) <= 256 && planes <= 65535 && h_out <= 65535) {   // then height: tmp -> y [planes][h_out][w_out]
        resize_aa_cols_kernel<<<dim3((unsigned)((w_out + 255) / 256), (unsigned)h_out, (unsigned)planes), 256, 0, s>>>(
            tmp, y, h_in, h_out, w_out, scale_of(h_in, h_out), bicubic, clamp01, mean3 ? 3 : 0,
            mean3 ? mean3[0] : 0.f, mean3 ? mean3[1] : 0.f, mean3 ? mean3[2] : 0.f, std3 ? std3[0] : 1.f,
            std3 ? std3[1] : 1.f, std3 ? std3[2] : 1.f);
        NUNIF_LAUNCH_CHECK();
    } else {
        const long total = planes * (long)h_out * w_out;
        resize_aa_axis_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(
            tmp, y, planes, h_in, h_out, w_out, 0, scale_of(h_in, h_out), bicubic, clamp01, mean3 ? 3 : 0,
            mean3 ? mean3[0] : 0.f, mean3 ? mean3[1] : 0.f, mean3 ? mean3[2] : 0.f, std3 ? std3[0] : 1.f,
            std3 ? std3[1] : 1.f, std3 ? std3[2] : 1.f);
        NUNIF_LAUNCH_CHECK();
    }
    return NUNIF_HIP_OK;
}

static const long kStatBlocks = 96L;      // workgroups of the min / max statistics kernels (r01d: 96 beat 32 and 256)

namespace {
constexpr long kFirstStatBlocks = 512;
long dilate_blocks(int H, int W) { return (long)((W + kDilTW - 1) / kDilTW) * ((H + kDilTH - 1) / kDilTH); }
long dilate_partials(int H, int W) { return std::max(dilate_blocks(H, W), std::min<long>(((long)H * W + 255) / 256, kFirstStatBlocks)); }
}  // namespace

// floats of scratch nunif_hip_dilate_edge needs: one ping-pong image + two sets of per-workgroup statistics partials
extern "C" int64_t nunif_hip_dilate_edge_work_floats(int32_t B, int32_t H, int32_t W) {
    const long n = (long)B * H * W;
    return ((n + 3) / 4) * 4 + 4 + 2 * (long)B * dilate_partials(H, W) * (long)(sizeof(RangePartial) / sizeof(float));
}

extern "C" int nunif_hip_dilate_edge(const float *x, float *y, float *work, int32_t B, int32_t H, int32_t W,
                                     int32_t n_x, int32_t n_y, void *stream) {
    NUNIF_REQUIRE(x && y && work && B > 0 && H > 0 && W > 0 && n_x >= 0 && n_y >= 0, "dilate_edge: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)B * H * W;
    // work: [n floats ping-pong][2 x B x partials RangePartial]
    float *bufs[2] = {y, work};
    const long np = dilate_partials(H, W);
    RangePartial *parts[2];
    parts[0] = reinterpret_cast<RangePartial *>(work + ((n + 3) / 4) * 4 + 4);
    parts[1] = parts[0] + (long)B * np;
    const int xy = n_x < n_y ? n_x : n_y;                       // dilation.py:118-120
    const int total_iters = xy + (n_y - xy) + (n_x - xy);
    if (total_iters == 0) {
        if (x != y) NUNIF_HIP_CHECK(hipMemcpyAsync(y, x, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        return NUNIF_HIP_OK;
    }
    ProfScope ps("dilate_edge", s, 0.0, (double)n * 16.0 * total_iters);
    const float *src = x;
    int it = 0;
    // the last iteration must land in y: choose the starting buffer by parity
    int dst_i = (total_iters % 2 == 1) ? 0 : 1;
    int n_in = (int)std::min<long>(((long)H * W + 255) / 256, kFirstStatBlocks), pi = 0;
    range_partials_kernel<<<dim3((unsigned)n_in, B), 256, 0, s>>>(src, parts[pi], H, W);
    const dim3 g2((unsigned)((W + kDilTW - 1) / kDilTW), (unsigned)((H + kDilTH - 1) / kDilTH), B);
    auto run = [&](int ky, int kx) -> int {
        const bool last = it + 1 == total_iters;
        dilate_fused_kernel<<<g2, 256, 0, s>>>(src, bufs[dst_i], parts[pi], n_in, last ? nullptr : parts[pi ^ 1], H, W, ky, kx);
        n_in = (int)dilate_blocks(H, W);
        pi ^= 1;
        src = bufs[dst_i];
        dst_i ^= 1;
        ++it;
        return NUNIF_HIP_OK;
    };
    int rc;
    for (int i = 0; i < xy; ++i) if ((rc = run(3, 3))) return rc;
    for (int i = 0; i < n_y - xy; ++i) if ((rc = run(3, 1))) return rc;       // kernel_size=(3,1): rows
    for (int i = 0; i < n_x - xy; ++i) if ((rc = run(1, 3))) return rc;
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

extern "C" int nunif_hip_minmax_normalize(const float *x, float *y, float *minmax, int32_t B, int64_t n_per,
                                          void *stream) {
    NUNIF_REQUIRE(x && y && minmax && B > 0 && n_per > 0, "minmax_normalize: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("minmax_normalize", s, 0.0, (double)B * n_per * 12.0);
    minmax_init_kernel<<<(B + 63) / 64, 64, 0, s>>>(reinterpret_cast<unsigned int *>(minmax), B);
    dim3 g1((unsigned)std::min<long>((n_per + 255) / 256, kStatBlocks), B);       // same-address atomics: see dilate_edge
    minmax_stats_kernel<<<g1, 256, 0, s>>>(x, minmax, n_per);
    dim3 g2((unsigned)((n_per + 255) / 256), B);
    minmax_apply_kernel<<<g2, 256, 0, s>>>(x, y, minmax, n_per);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

// ---- VideoDepthAnything pre/post glue (iw3/video_depth_anything_model.py:51-91) ---------------------------------------------
// reflection_pad2d_naive (nunif/modules/reflection_pad2d.py:13-48): reflect WITHOUT repeating the edge pixel for positive
// pads, crop for negative ones (F.pad(out, (-14,) * 4), video_depth_anything_model.py:79); one gather, planar fp32.
__global__ void __launch_bounds__(256) reflection_pad_kernel(const float *__restrict__ x, float *__restrict__ y, long planes,
                                                             int H, int W, int Ho, int Wo, int left, int top) {
    const long n = planes * Ho * Wo;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int xo = (int)(i % Wo);
    const long t = i / Wo;
    const int yo = (int)(t % Ho);
    const long p = t / Ho;
    int sx = xo - left, sy = yo - top;
    sx = sx < 0 ? -sx : (sx >= W ? 2 * (W - 1) - sx : sx);
    sy = sy < 0 ? -sy : (sy >= H ? 2 * (H - 1) - sy : sy);
    y[i] = x[(p * H + sy) * W + sx];
}

extern "C" int nunif_hip_reflection_pad2d(const float *x, float *y, int64_t planes, int32_t H, int32_t W, int32_t left,
                                          int32_t right, int32_t top, int32_t bottom, void *stream) {
    NUNIF_REQUIRE(x && y && planes > 0 && H > 0 && W > 0, "reflection_pad2d: bad argument");
    // the reference asserts padding <= size (reflection_pad2d.py:15-16); a reflected index must stay inside the map
    NUNIF_REQUIRE(left < W && right < W && top < H && bottom < H, "reflection_pad2d: padding must be smaller than the map");
    const int Ho = H + top + bottom, Wo = W + left + right;
    NUNIF_REQUIRE(Ho > 0 && Wo > 0, "reflection_pad2d: empty result");
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)planes * Ho * Wo;
    ProfScope ps("reflection_pad_kernel", s, 0.0, (double)n * 8.0);
    reflection_pad_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(x, y, planes, H, W, Ho, Wo, left, top);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

// _postprocess :66-76: nan_to_num, optional clamp(max=max_dist), metric depth -> disparity 1 / (d + eps), optional sign flip
__global__ void __launch_bounds__(256) depth_post_kernel(const float *__restrict__ x, float *__restrict__ y, long n,
                                                         float max_dist, int to_disparity, float eps, int negate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = x[i];
    if (v != v) v = 0.0f;                                   // torch.nan_to_num defaults: nan -> 0, +-inf -> +-FLT_MAX
    v = fminf(fmaxf(v, -3.4028234663852886e38f), 3.4028234663852886e38f);
    if (max_dist > 0.0f) v = fminf(v, max_dist);
    if (to_disparity) v = 1.0f / (v + eps);
    y[i] = negate ? -v : v;
}

extern "C" int nunif_hip_depth_postprocess(const float *x, float *y, int64_t n, float max_dist, int32_t to_disparity,
                                           float eps, int32_t negate, void *stream) {
    NUNIF_REQUIRE(x && y && n > 0, "depth_postprocess: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("depth_post_kernel", s, 0.0, (double)n * 8.0);
    depth_post_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(x, y, (long)n, max_dist, to_disparity, eps, negate);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

// ---- EMAMinMaxScaler on the device --------------------------------------------------------------------------------------------
extern "C" int nunif_hip_minmax(const float *x, float *minmax_keys, int32_t B, int64_t n_per, void *stream) {
    NUNIF_REQUIRE(x && minmax_keys && B > 0 && n_per > 0, "minmax: bad argument");
    hipStream_t s = (hipStream_t)stream;
    minmax_init_kernel<<<(B + 63) / 64, 64, 0, s>>>(reinterpret_cast<unsigned int *>(minmax_keys), B);
    dim3 g1((unsigned)std::min<long>((n_per + 255) / 256, kStatBlocks), B);
    minmax_stats_kernel<<<g1, 256, 0, s>>>(x, minmax_keys, n_per);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

extern "C" int nunif_hip_ema_scaler_push(float *state, const float *minmax_keys, int32_t ring_size, int64_t count,
                                         int32_t filled, int32_t first, double decay, void *stream) {
    NUNIF_REQUIRE(state && minmax_keys && ring_size >= 2 && ring_size % 2 == 0 && count >= 0, "ema_scaler_push: bad argument");
    // the reference multiplies fp32 0-dim tensors by Python floats: `decay` and `1. - decay` are rounded to fp32 separately
    ema_scaler_push_kernel<<<1, 64, 0, (hipStream_t)stream>>>(state, minmax_keys, ring_size, (long)count, filled, first,
                                                              (float)decay, (float)(1.0 - decay));
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

extern "C" int nunif_hip_ema_scaler_ring_minmax(float *state, int32_t ring_size, void *stream) {
    NUNIF_REQUIRE(state && ring_size >= 2, "ema_scaler_ring_minmax: bad argument");
    ema_scaler_ring_minmax_kernel<<<1, 64, 0, (hipStream_t)stream>>>(state, ring_size);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

extern "C" int nunif_hip_range_normalize(const float *x, float *y, const float *lohi, int64_t n, int32_t max_mode,
                                         void *stream) {
    NUNIF_REQUIRE(x && y && lohi && n > 0, "range_normalize: bad argument");
    range_normalize_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, y, lohi, (long)n, max_mode);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

extern "C" int nunif_hip_make_input_planes(const float *depth, float *out, int32_t B, int32_t H, int32_t W,
                                           double divergence_value, double convergence_value, int32_t border_pix,
                                           void *stream) {
    NUNIF_REQUIRE(depth && out && B > 0 && H > 0 && W > 0 && border_pix >= 0, "make_input_planes: bad argument");
    const long n = (long)B * H * W;
    make_input_planes_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        depth, out, B, H, W, (float)divergence_value, (float)convergence_value, border_pix);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

extern "C" int nunif_hip_stack(const void *const *srcs, int32_t n, int64_t bytes_each, void *dst, void *stream) {
    NUNIF_REQUIRE(srcs && dst && n > 0 && n <= 16 && bytes_each > 0, "stack: 1..16 buffers of equal size");
    StackArgs a;
    bool aligned = (bytes_each % 16 == 0) && ((uintptr_t)dst % 16 == 0);
    for (int k = 0; k < 16; ++k) {
        a.src[k] = k < n ? srcs[k] : nullptr;
        if (k < n) { NUNIF_REQUIRE(srcs[k], "stack: NULL source"); aligned = aligned && ((uintptr_t)srcs[k] % 16 == 0); }
    }
    hipStream_t s = (hipStream_t)stream;
    if (aligned) {
        const long v = bytes_each / 16;
        stack_kernel<<<(unsigned)((v + 255) / 256), 256, 0, s>>>(a, reinterpret_cast<uint4 *>(dst), v, n);
    } else {
        stack_bytes_kernel<<<(unsigned)((bytes_each + 255) / 256), 256, 0, s>>>(a, reinterpret_cast<unsigned char *>(dst), bytes_each, n);
    }
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}
