// waifu2x image-side helpers on gfx950: 8-way dihedral TTA views / merge and alpha border padding.
//
// Reference: nunif/transforms/tta.py — tta_split :20-34 (x, hflip, vflip, vflip+hflip, rot90 and its three flips),
// tta_merge :37-48 (inverse transforms, sum in that order, * 1/8, clamp); nunif/utils/alpha.py — AlphaBorderPadding
// :32-57 (offset iterations of: 3x3 zero-padded box sums of mask and rgb, divide, write into the transparent pixels, grow
// the mask), ChannelWiseSum :5-29.  All HBM-bound single-pass kernels (-ffp-contract=off: the merge adds the eight
// views in the reference's order, so it is bit-exact for identical inputs).
#include "common.h"

namespace nunif {

// view v = t*4 + vflip*2 + hflip (the reference's tuple order).  Forward: y = hflip?(vflip?(rot90?(x))).
// rot90(x, 1, (1,2)): r[i][j] = x[j][W-1-i], r has shape [W, H].
__device__ __forceinline__ void view_src(int v, int H, int W, int i, int j, int &sy, int &sx) {
    // (i, j) index the VIEW (shape [Hv, Wv]); returns the source pixel of x (shape [H, W])
    const int Hv = (v & 4) ? W : H, Wv = (v & 4) ? H : W;
    if (v & 1) j = Wv - 1 - j;
    if (v & 2) i = Hv - 1 - i;
    if (v & 4) { sy = j; sx = W - 1 - i; } else { sy = i; sx = j; }
}

__global__ void __launch_bounds__(256) tta_view_kernel(const float *__restrict__ x, float *__restrict__ y, int C, int H, int W,
                                                       int view) {
    const int Hv = (view & 4) ? W : H, Wv = (view & 4) ? H : W;
    const long total = (long)C * Hv * Wv;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int j = (int)(id % Wv);
    const long t = id / Wv;
    const int i = (int)(t % Hv), c = (int)(t / Hv);
    int sy, sx;
    view_src(view, H, W, i, j, sy, sx);
    y[id] = x[((long)c * H + sy) * W + sx];
}

struct TtaMergeArgs { const float *v[8]; float *out; int C, H, W; };

__global__ void __launch_bounds__(256) tta_merge_kernel(TtaMergeArgs a) {
    const long total = (long)a.C * a.H * a.W;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int sx = (int)(id % a.W);
    const long t = id / a.W;
    const int sy = (int)(t % a.H), c = (int)(t / a.H);
    float acc = 0.f;
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        // invert view_src: the view pixel (i, j) that shows source pixel (sy, sx)
        const int Hv = (v & 4) ? a.W : a.H, Wv = (v & 4) ? a.H : a.W;
        int i, j;
        if (v & 4) { j = sy; i = a.W - 1 - sx; } else { i = sy; j = sx; }
        if (v & 2) i = Hv - 1 - i;
        if (v & 1) j = Wv - 1 - j;
        const float val = a.v[v][((long)c * Hv + i) * Wv + j];
        acc = v == 0 ? val : acc + val;
    }
    a.out[id] = fminf(fmaxf(acc * 0.125f, 0.f), 1.f);
}

// one AlphaBorderPadding iteration; iteration 0 also applies the initial "rgb[:, mask < 1] = 0" (first != 0: mask_in is alpha)
__global__ void __launch_bounds__(256) alpha_pad_iter_kernel(const float *__restrict__ rgb_in, const float *__restrict__ mask_in,
                                                             float *__restrict__ rgb_out, float *__restrict__ mask_out, int H,
                                                             int W, int first, int last) {
    const long hw = (long)H * W;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= hw) return;
    const int x = (int)(id % W), y = (int)(id / W);
    auto m_at = [&](int yy, int xx) -> float {
        const float m = mask_in[(long)yy * W + xx];
        return first ? (m > 0.f ? 1.f : 0.f) : m;
    };
    float wsum = 0.f, s[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const float m = m_at(yy, xx);
            wsum += m;
            // rgb of a transparent pixel is 0 from the start; afterwards it holds the bled colour
            const bool zeroed = first && m < 1.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) s[c] += zeroed ? 0.f : rgb_in[c * hw + (long)yy * W + xx];
        }
    const float m0 = m_at(y, x);
    const bool hole = m0 < 1.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = hole ? s[c] / (wsum + 1e-7f) : rgb_in[c * hw + id];
        if (last) v = fminf(fmaxf(v, 0.f), 1.f);
        rgb_out[c * hw + id] = v;
    }
    mask_out[id] = wsum > 0.f ? 1.f : 0.f;
}

__global__ void __launch_bounds__(256) alpha_pad_zero_kernel(const float *__restrict__ rgb, const float *__restrict__ alpha,
                                                             float *__restrict__ out, long hw) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= hw) return;
    const bool hole = !(alpha[id] > 0.f);
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c * hw + id] = fminf(fmaxf(hole ? 0.f : rgb[c * hw + id], 0.f), 1.f);
}


// ---- hole-mask post-processing (iw3/backward_warp.py postprocess_hole_mask :382-393) --------------------------------------
// 3x3 max / min filter with the window clipped at the border (F.max_pool2d pads with -inf; erode = -max_pool(-x))
__global__ void __launch_bounds__(256) hm_morph_kernel(const float *__restrict__ in, float *__restrict__ out, int B, int h,
                                                        int w, int is_min) {
    const long n = (long)B * h * w, id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    const int x = (int)(id % w);
    const long t = id / w;
    const int y = (int)(t % h);
    const float *p = in + (t / h) * (long)h * w;
    float v = p[(long)y * w + x];
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= h) continue;
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= w) continue;
            const float u = p[(long)yy * w + xx];
            v = is_min ? fminf(v, u) : fmaxf(v, u);
        }
    }
    out[id] = v;
}

// bilinear resize with align_corners=True exactly as ATen's upsample_bilinear2d computes it (area_pixel_compute_scale,
// guard_index_and_lambda; value = wy0 (wx0 v00 + wx1 v01) + wy1 (wx0 v10 + wx1 v11)), then sigmoid(v) > threshold
__global__ void __launch_bounds__(256) hm_resize_thr_kernel(const float *__restrict__ in, uint8_t *__restrict__ out, int B,
                                                             int h, int w, int H, int W, float thr) {
    const long n = (long)B * H * W, id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    const int X = (int)(id % W);
    const long t = id / W;
    const int Y = (int)(t % H);
    const float *p = in + (t / H) * (long)h * w;
    float v;
    if (H == h && W == w) {
        v = p[(long)Y * w + X];
    } else {
        const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
        const float ry = sy * (float)Y, rx = sx * (float)X;
        const int y0 = min((int)ry, h - 1), x0 = min((int)rx, w - 1);
        const float ly = fminf(fmaxf(ry - (float)y0, 0.f), 1.f), lx = fminf(fmaxf(rx - (float)x0, 0.f), 1.f);
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float wy0 = 1.f - ly, wx0 = 1.f - lx;
        const float a = wx0 * p[(long)y0 * w + x0] + lx * p[(long)y0 * w + x1];
        const float b = wx0 * p[(long)y1 * w + x0] + lx * p[(long)y1 * w + x1];
        v = wy0 * a + ly * b;
    }
    const float sg = 1.f / (1.f + expf(-v));
    out[id] = sg > thr ? 1 : 0;
}

// mask[x] = OR t[x - n_outer .. x + n_inner] (dilate_inner: mask |= mask shifted left, then dilate_outer: |= shifted right,
// zeros flowing in at the border); optional z[b,c,y,x] *= 1 - mask
__global__ void __launch_bounds__(256) hm_dilate_kernel(const uint8_t *__restrict__ t, uint8_t *__restrict__ mask, int B, int H,
                                                         int W, int n_inner, int n_outer, float *z, int C) {
    const long n = (long)B * H * W, id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    const int X = (int)(id % W);
    const long row = id / W;
    const uint8_t *p = t + row * W;
    const int lo = max(X - n_outer, 0), hi = min(X + n_inner, W - 1);
    uint8_t m = 0;
    for (int x = lo; x <= hi; ++x) m |= p[x];
    mask[id] = m;
    if (z && m) {
        const long b = row / H, y = row % H;
        for (int c = 0; c < C; ++c) z[((b * C + c) * H + y) * W + X] = 0.f;
    }
}


// ---- iw3 output formats ---------------------------------------------------------------------------------------------------
// iw3/anaglyph.py :4-93.  mode: 0 color, 1 gray, 2 half-color, 3 wimmer, 4 wimmer2, 5 dubois (clip before the sum), 6 dubois2
__device__ __forceinline__ float clamp01f(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float bt601(float r, float g, float b) { return r * 0.299f + g * 0.587f + b * 0.114f; }
__device__ __forceinline__ float srgb_to_linear(float x) { return x <= 0.04045f ? x / 12.92f : powf((x + 0.055f) / 1.055f, 2.4f); }
__device__ __forceinline__ float linear_to_srgb(float x) {
    return x <= 0.0031308f ? x * 12.92f : 1.055f * powf(x, 1.0f / 2.4f) - 0.055f;
}

__global__ void __launch_bounds__(256) anaglyph_kernel(const float *__restrict__ l, const float *__restrict__ r,
                                                        float *__restrict__ out, long hw, int mode) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= hw) return;
    const float lr = l[id], lg = l[hw + id], lb = l[2 * hw + id];
    const float rr = r[id], rg = r[hw + id], rb = r[2 * hw + id];
    float o0, o1, o2;
    switch (mode) {
        case 0: o0 = lr; o1 = rg; o2 = rb; break;                                            // no clamp (anaglyph.py:9-11)
        case 1: { const float ry = bt601(rr, rg, rb); o0 = clamp01f(bt601(lr, lg, lb)); o1 = clamp01f(ry); o2 = clamp01f(ry); break; }
        case 2: o0 = clamp01f(bt601(lr, lg, lb)); o1 = clamp01f(rg); o2 = clamp01f(rb); break;
        case 3: o0 = clamp01f(lg * 0.7f + lb * 0.3f); o1 = clamp01f(rg); o2 = clamp01f(rb); break;
        case 4: {
            const float g_l = lg + 0.45f * fmaxf(lr - lg, 0.f), b_l = lb + 0.25f * fmaxf(lr - lb, 0.f);
            const float g_r = rg + 0.45f * fmaxf(rr - rg, 0.f), b_r = rb + 0.25f * fmaxf(rr - rb, 0.f);
            o0 = clamp01f(powf(0.75f * g_l + 0.25f * b_l, 1.0f / 1.6f)); o1 = clamp01f(g_r); o2 = clamp01f(b_r);
            break;
        }
        default: {
            const bool clip = mode == 5;
            const float L0 = srgb_to_linear(lr), L1 = srgb_to_linear(lg), L2 = srgb_to_linear(lb);
            const float R0 = srgb_to_linear(rr), R1 = srgb_to_linear(rg), R2 = srgb_to_linear(rb);
            const float lm[3][3] = {{0.437f, 0.449f, 0.164f}, {-0.062f, -0.062f, -0.024f}, {-0.048f, -0.050f, -0.017f}};
            const float rm[3][3] = {{-0.011f, -0.032f, -0.007f}, {0.377f, 0.761f, 0.009f}, {-0.026f, -0.093f, 1.234f}};
            float o[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float a = L0 * lm[c][0] + L1 * lm[c][1] + L2 * lm[c][2];
                float b = R0 * rm[c][0] + R1 * rm[c][1] + R2 * rm[c][2];
                if (clip) { a = clamp01f(a); b = clamp01f(b); }
                o[c] = clamp01f(linear_to_srgb(clamp01f(a + b)));
            }
            o0 = o[0]; o1 = o[1]; o2 = o[2];
        }
    }
    out[id] = o0; out[hw + id] = o1; out[2 * hw + id] = o2;
}

// iw3/equirectangular.py :7-40: zero-pad to (max_edge * 3 / 2)^2-ish, then bicubic grid_sample (zeros, align_corners) on the
// grid x' = k tan(az), y' = k tan(el) / cos(az), k = max_edge / output_size; the padding is folded into the sampler
__device__ __forceinline__ float cubic1(float x) { return ((-0.75f + 2.f) * x - (-0.75f + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x) { return ((-0.75f * x - 5.f * -0.75f) * x + 8.f * -0.75f) * x - 4.f * -0.75f; }
__device__ __forceinline__ float linspace_m1_1(int i, int n) {
    if (n == 1) return -1.f;
    const float step = 2.f / (float)(n - 1);
    return i < n / 2 ? -1.f + step * (float)i : 1.f - step * (float)(n - 1 - i);
}

__global__ void __launch_bounds__(256) equirect_kernel(const float *__restrict__ c, float *__restrict__ out, int C, int h, int w,
                                                        int Hp, int Wp, int pad_h, int pad_w, float k) {
    const long n = (long)Hp * Wp, id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    const int X = (int)(id % Wp), Y = (int)(id / Wp);
    const float az = linspace_m1_1(X, Wp) * 1.5707963267948966f, el = linspace_m1_1(Y, Hp) * 1.5707963267948966f;
    const float gx = k * tanf(az), gy = k * (tanf(el) / cosf(az));
    const float ix = ((gx + 1.f) / 2.f) * (float)(Wp - 1), iy = ((gy + 1.f) / 2.f) * (float)(Hp - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const float tx = ix - fx, ty = iy - fy;
    const float cx[4] = {cubic2(tx + 1.f), cubic1(tx), cubic1(1.f - tx), cubic2(2.f - tx)};
    const float cy[4] = {cubic2(ty + 1.f), cubic1(ty), cubic1(1.f - ty), cubic2(2.f - ty)};
    // coordinates far outside (tan near the poles) must not overflow the int conversion
    const bool far = !(fabsf(ix) < 1.0e8f && fabsf(iy) < 1.0e8f);
    const int x0 = far ? -100000 : (int)fx - 1 - pad_w, y0 = far ? -100000 : (int)fy - 1 - pad_h;
    for (int ch = 0; ch < C; ++ch) {
        const float *p = c + (long)ch * h * w;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int yy = y0 + j;
            float row = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int xx = x0 + i;
                const float v = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? p[(long)yy * w + xx] : 0.f;
                row += v * cx[i];
            }
            acc += row * cy[j];
        }
        out[(long)ch * n + id] = clamp01f(acc);
    }
}

}  // namespace nunif

using namespace nunif;

extern "C" int nunif_hip_tta_view(const float *x, float *y, int32_t C, int32_t H, int32_t W, int32_t view, void *stream) {
    NUNIF_REQUIRE(x && y && C > 0 && H > 0 && W > 0 && view >= 0 && view < 8, "tta_view: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)C * H * W;
    ProfScope ps("tta_view_kernel", s, 0.0, (double)n * 8.0);
    tta_view_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(x, y, C, H, W, view);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

extern "C" int nunif_hip_tta_merge(const float *const *views, float *out, int32_t C, int32_t H, int32_t W, void *stream) {
    NUNIF_REQUIRE(views && out && C > 0 && H > 0 && W > 0, "tta_merge: bad argument");
    TtaMergeArgs a;
    for (int v = 0; v < 8; ++v) { NUNIF_REQUIRE(views[v], "tta_merge: view %d is NULL", v); a.v[v] = views[v]; }
    a.out = out; a.C = C; a.H = H; a.W = W;
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)C * H * W;
    ProfScope ps("tta_merge_kernel", s, 0.0, (double)n * 36.0);
    tta_merge_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(a);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

// work: 8*H*W floats (two rgb + two mask planes for the ping-pong)
extern "C" int nunif_hip_alpha_border_padding(const float *rgb, const float *alpha, float *out, float *work, int32_t H,
                                              int32_t W, int32_t offset, void *stream) {
    NUNIF_REQUIRE(rgb && alpha && out && work && H > 0 && W > 0 && offset >= 0, "alpha_border_padding: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long hw = (long)H * W;
    const unsigned blocks = (unsigned)((hw + 255) / 256);
    ProfScope ps("alpha_border_padding", s, 0.0, (double)hw * 32.0 * (offset > 0 ? offset : 1));
    if (offset == 0) {
        alpha_pad_zero_kernel<<<blocks, 256, 0, s>>>(rgb, alpha, out, hw);
        NUNIF_LAUNCH_CHECK();
        return NUNIF_HIP_OK;
    }
    float *rbuf[2] = {work, work + 3 * hw}, *mbuf[2] = {work + 6 * hw, work + 7 * hw};
    const float *rin = rgb, *min_ = alpha;
    for (int i = 0; i < offset; ++i) {
        const bool last = i == offset - 1;
        float *rout = last ? out : rbuf[i & 1], *mout = mbuf[i & 1];
        alpha_pad_iter_kernel<<<blocks, 256, 0, s>>>(rin, min_, rout, mout, H, W, i == 0 ? 1 : 0, last ? 1 : 0);
        rin = rout; min_ = mout;
    }
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

// work: 2*B*h*w floats (the two morphology passes) followed by B*H*W bytes (the thresholded map before the dilations)
extern "C" int nunif_hip_hole_mask_postprocess(const float *logits, uint8_t *mask, float *work, int32_t B, int32_t h, int32_t w,
                                               int32_t H, int32_t W, float threshold, int32_t inner_iter, int32_t outer_iter,
                                               float *z, int32_t C, void *stream) {
    NUNIF_REQUIRE(logits && mask && work && B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && inner_iter >= 0 && outer_iter >= 0 &&
                  (!z || C > 0), "hole_mask_postprocess: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)B * h * w, N = (long)B * H * W;
    ProfScope ps("hole_mask_postprocess", s, 0.0, (double)n * 16.0 + (double)N * 3.0);
    float *t0 = work, *t1 = work + n;
    uint8_t *thr = reinterpret_cast<uint8_t *>(work + 2 * n);
    hm_morph_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(logits, t0, B, h, w, 0);       // closing(n_iter = 1): dilate
    hm_morph_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(t0, t1, B, h, w, 1);           //                      erode
    const bool direct = inner_iter == 0 && outer_iter == 0 && !z;
    hm_resize_thr_kernel<<<(unsigned)((N + 255) / 256), 256, 0, s>>>(t1, direct ? mask : thr, B, h, w, H, W, threshold);
    if (!direct)
        hm_dilate_kernel<<<(unsigned)((N + 255) / 256), 256, 0, s>>>(thr, mask, B, H, W, inner_iter, outer_iter, z, C);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

extern "C" int nunif_hip_anaglyph(const float *left, const float *right, float *out, int32_t H, int32_t W, int32_t mode,
                                  void *stream) {
    NUNIF_REQUIRE(left && right && out && H > 0 && W > 0 && mode >= 0 && mode <= 6, "anaglyph: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long hw = (long)H * W;
    ProfScope ps("anaglyph_kernel", s, 0.0, (double)hw * 36.0);
    anaglyph_kernel<<<(unsigned)((hw + 255) / 256), 256, 0, s>>>(left, right, out, hw, mode);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

extern "C" int nunif_hip_equirectangular(const float *c, float *out, int32_t C, int32_t h, int32_t w, void *stream) {
    NUNIF_REQUIRE(c && out && C > 0 && h > 0 && w > 0, "equirectangular: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int max_edge = h > w ? h : w, output_size = max_edge + max_edge / 2;
    const int pad_w = (output_size - w) / 2, pad_h = (output_size - h) / 2;
    const int Hp = h + 2 * pad_h, Wp = w + 2 * pad_w;
    const long n = (long)Hp * Wp;
    ProfScope ps("equirect_kernel", s, 0.0, (double)n * C * 8.0);
    equirect_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(c, out, C, h, w, Hp, Wp, pad_h, pad_w,
                                                                 (float)((double)max_edge / (double)output_size));
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}
