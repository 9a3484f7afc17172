// iw3 frame edge + pointwise kernels for gfx950: uint8/uint16 HWC <-> fp32 CHW, stereo compose + quantise, depth mappers.
//
// Reference: nunif/utils/video.py to_tensor :218-223 / from_tensor :236-245 (x/max; (x*max).round().to(uint)),
// iw3/utils.py postprocess_image :430-487 (cat(left,right) + clamp; top-bottom; cross-eyed), iw3/mapper.py :7-118
// (softplus01 / inv_softplus01 / distance_to_disparity / shift_relative_depth / pow2 / softplus legacy).
//
// All HBM-bound streaming kernels.  The compose kernel fuses the reference's cat + clamp + permute + *255 + round +
// cast chain into one pass: read 24 B, write 6 B per SBS pixel pair (30 B / px).
#include "common.h"

namespace nunif {

template <typename T>
__global__ void __launch_bounds__(256)
frame_to_tensor_kernel(const T *__restrict__ in, float *__restrict__ out, long hw, float maxv) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= hw) return;
    // x.permute(2,0,1) / iinfo.max  (true division, fp32)
    out[i] = (float)in[i * 3] / maxv;
    out[hw + i] = (float)in[i * 3 + 1] / maxv;
    out[2 * hw + i] = (float)in[i * 3 + 2] / maxv;
}

// out HWC [Ho, Wo, 3]; layout 0: left|right, 1: right|left (cross-eyed), 2: left over right (top-bottom),
// 3: the left image alone (VU.to_frame of a single frame, video.py:236-245)
template <typename T>
__global__ void __launch_bounds__(256)
stereo_to_frame_kernel(const float *__restrict__ left, const float *__restrict__ right, T *__restrict__ out, int H,
                       int W, int layout, float maxv) {
    const int Ho = layout == 2 ? 2 * H : H, Wo = (layout == 2 || layout == 3) ? W : 2 * W;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)Ho * Wo) return;
    const int x = (int)(i % Wo), y = (int)(i / Wo);
    const float *src;
    int sx = x, sy = y;
    if (layout == 3) src = left;
    else if (layout == 2) { src = y < H ? left : right; if (y >= H) sy = y - H; }
    else {
        const bool first = x < W;
        src = (first == (layout == 0)) ? left : right;
        if (!first) sx = x - W;
    }
    const long hw = (long)H * W, p = (long)sy * W + sx;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = fminf(fmaxf(src[c * hw + p], 0.f), 1.f);          // torch.clamp(sbs, 0, 1)
        out[i * 3 + c] = (T)rintf(v * maxv);                               // (x * max).round_(): half to even
    }
}

__global__ void __launch_bounds__(256)
stereo_compose_kernel(const float *__restrict__ left, const float *__restrict__ right, float *__restrict__ out, int H,
                      int W, int layout) {
    const int Ho = layout == 2 ? 2 * H : H, Wo = layout == 2 ? W : 2 * W;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)Ho * Wo) return;
    const int x = (int)(i % Wo), y = (int)(i / Wo);
    const float *src;
    int sx = x, sy = y;
    if (layout == 2) { src = y < H ? left : right; if (y >= H) sy = y - H; }
    else {
        const bool first = x < W;
        src = (first == (layout == 0)) ? left : right;
        if (!first) sx = x - W;
    }
    const long hw = (long)H * W, p = (long)sy * W + sx, ohw = (long)Ho * Wo;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c * ohw + i] = fminf(fmaxf(src[c * hw + p], 0.f), 1.f);
}

// kind: 1 pow2, 2 softplus01_legacy(c=p0), 3 softplus01(bias=p0, scale=p1), 4 inv_softplus01(bias=p0, scale=p1),
//       5 distance_to_disparity(c=p0), 6 shift_relative_depth(min_distance=p0, max_distance=p1)
__global__ void __launch_bounds__(256)
map_depth_kernel(const float *__restrict__ x, float *__restrict__ y, long n, int kind, float p0, float p1, float k0,
                 float k1) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    float o = v;
    switch (kind) {
        case 1: o = v * v; break;
        case 2: o = (logf(1.f + expf(v * 12.0f - p0)) / (12.f - p0) - k0) / (k1 - k0); break;
        case 3: o = (logf(1.f + expf((v - p0) * p1)) - k0) / (k1 - k0); break;
        case 4: o = (logf(fmaxf(expm1f((v - p0) * p1), 1e-6f)) - k0) / (k1 - k0); break;
        case 5: { const float c1 = 1.0f + p0, mn = p0 / c1; o = ((p0 / (c1 - v)) - mn) / (1.0f - mn); break; }
        case 6: {
            const float pmax = p0 + p1;
            const float A = 1.0f / pmax, B = (1.0f / p0) - (1.0f / pmax);
            float dist = 1.f / (A + B * v);
            dist = (1.0f - p0) + dist;
            const float nx = 1.0f / dist;
            const float mn = 1.0f / (p1 + 1.f), range = 1.0f - 1.0f / (p1 + 1.f);
            o = (nx - mn) / range;
            break;
        }
        default: break;
    }
    y[i] = o;
}

}  // namespace nunif

using namespace nunif;

extern "C" int nunif_hip_frame_to_tensor(const void *frame, float *chw, int32_t H, int32_t W, int32_t bits,
                                         void *stream) {
    NUNIF_REQUIRE(frame && chw && H > 0 && W > 0 && (bits == 8 || bits == 16), "frame_to_tensor: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long hw = (long)H * W;
    ProfScope ps("frame_to_tensor_kernel", s, 0.0, (double)hw * (3.0 * bits / 8 + 12.0));
    const unsigned blocks = (unsigned)((hw + 255) / 256);
    if (bits == 8) frame_to_tensor_kernel<uint8_t><<<blocks, 256, 0, s>>>((const uint8_t *)frame, chw, hw, 255.0f);
    else frame_to_tensor_kernel<uint16_t><<<blocks, 256, 0, s>>>((const uint16_t *)frame, chw, hw, 65535.0f);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

extern "C" int nunif_hip_stereo_to_frame(const float *left, const float *right, void *frame, int32_t H, int32_t W,
                                         int32_t layout, int32_t bits, void *stream) {
    NUNIF_REQUIRE(left && (right || layout == 3) && frame && H > 0 && W > 0 && layout >= 0 && layout <= 3 &&
                  (bits == 8 || bits == 16), "stereo_to_frame: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long n = (layout == 3 ? 1L : 2L) * H * W;
    ProfScope ps("stereo_to_frame_kernel", s, 0.0, (double)n * (12.0 + 3.0 * bits / 8));
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (bits == 8) stereo_to_frame_kernel<uint8_t><<<blocks, 256, 0, s>>>(left, right, (uint8_t *)frame, H, W, layout, 255.0f);
    else stereo_to_frame_kernel<uint16_t><<<blocks, 256, 0, s>>>(left, right, (uint16_t *)frame, H, W, layout, 65535.0f);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

extern "C" int nunif_hip_stereo_compose(const float *left, const float *right, float *out, int32_t H, int32_t W,
                                        int32_t layout, void *stream) {
    NUNIF_REQUIRE(left && right && out && H > 0 && W > 0 && layout >= 0 && layout <= 2, "stereo_compose: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long n = 2L * H * W;
    ProfScope ps("stereo_compose_kernel", s, 0.0, (double)n * 24.0);
    stereo_compose_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(left, right, out, H, W, layout);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

extern "C" int nunif_hip_map_depth(const float *x, float *y, int64_t n, int32_t kind, double p0, double p1,
                                   void *stream) {
    NUNIF_REQUIRE(x && y && n > 0 && kind >= 0 && kind <= 6, "map_depth: bad argument");
    // the reference evaluates the normalisation constants min_v / max_v on the host in double (math.log/exp) or on a
    // 1-element fp32 tensor (inv_softplus01); both are reproduced here
    double k0 = 0.0, k1 = 1.0;
    if (kind == 2) {
        k0 = log(1.0 + exp(0.0 * 12.0 - p0)) / (12.0 - p0);
        k1 = log(1.0 + exp(1.0 * 12.0 - p0)) / (12.0 - p0);
    } else if (kind == 3) {
        k0 = log(1.0 + exp((0.0 - p0) * p1));
        k1 = log(1.0 + exp((1.0 - p0) * p1));
    } else if (kind == 4) {
        const float f0 = logf(fmaxf(expm1f((0.0f - (float)p0) * (float)p1), 1e-6f));
        const float f1 = logf(fmaxf(expm1f((1.0f - (float)p0) * (float)p1), 1e-6f));
        k0 = f0; k1 = f1;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps("map_depth_kernel", s, 0.0, (double)n * 8.0);
    map_depth_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(x, y, n, kind, (float)p0, (float)p1, (float)k0, (float)k1);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}
