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
