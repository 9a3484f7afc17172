// iw3 "iw3.depth_aa" (depth anti-aliasing net, --depth-aa) on gfx950.
//
// Reference: iw3/models/depth_aa.py — DepthAA.infer :46-56 (tensor-wide min-max normalise, forward(clamp=False),
// de-normalise), DepthAA.forward :59-85 (centred replicate pad to multiples of 16, pixel_unshuffle 2, proj_in 4->32,
// three WABlocks with zero-pad shift True/False/True, proj_out 32->4, pixel_shuffle 2, crop, residual), WABlock :11-26;
// nunif/modules/attention.py WindowMHA2d :118-161 (window 8x8 = 64 tokens, 2 heads of 16), WindowScoreBias :375-419.
//
// Maps are NHWC fp16 [B, H/2, W/2, 32].  Kernels:
//   daa_minmax_kernel  tensor-wide min / max (order-preserving uint keys + atomics)
//   daa_in_kernel      normalise + pad + unshuffle + proj_in                               (VALU, K = 4)
//   wmha8_kernel       one 8x8 window per wave: qkv GEMM, 2 heads x (4x4 score tiles with the learned 64x64 bias as
//                      the MFMA C operand, softmax over 64 keys, PV), head_proj, residual — all in registers
//   gemm_kernel<1,4>   conv_mlp[0] 1x1 + GELU(erf)     conv_kernel<2,4>   replicate-pad 3x3 + LeakyReLU + residual
//   daa_out_kernel     proj_out + pixel_shuffle + crop + residual + de-normalise           (VALU)
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "swin_kernels.h"

namespace nunif {

#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ unsigned int daa_key(float v) {
    const unsigned int u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float daa_unkey(unsigned int k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void daa_minmax_init_kernel(unsigned int *mm) { mm[0] = 0xFFFFFFFFu; mm[1] = 0u; }

__global__ void __launch_bounds__(256) daa_minmax_kernel(const float *__restrict__ x, unsigned int *mm, long n) {
    float mn = 3.0e38f, mx = -3.0e38f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = x[i];
        mn = fminf(mn, v); mx = fmaxf(mx, v);
    }
    __shared__ float smn[256], smx[256];
    smn[threadIdx.x] = mn; smx[threadIdx.x] = mx;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
            smn[threadIdx.x] = fminf(smn[threadIdx.x], smn[threadIdx.x + st]);
            smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + st]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { atomicMin(mm, daa_key(smn[0])); atomicMax(mm + 1, daa_key(smx[0])); }
}

struct DaaInArgs {
    const float *x;           // [B,1,h,w]
    const unsigned int *mm;   // min / max keys, or NULL: no normalisation (plain forward)
    const float *w;           // [4][32] then bias[32]   (k = i*2 + j of the 2x2 unshuffle)
    f16 *out;                 // [B,Hq,Wq,32]
    int B, h, w_, Hq, Wq, ph1, pw1;
};

__global__ void __launch_bounds__(256) daa_in_kernel(DaaInArgs a) {
    __shared__ float sw[5 * 32];
    if (threadIdx.x < 160) sw[threadIdx.x] = a.w[threadIdx.x];
    __syncthreads();
    const long total = (long)a.B * a.Hq * a.Wq;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int xq = (int)(id % a.Wq);
    const long t = id / a.Wq;
    const int yq = (int)(t % a.Hq), b = (int)(t / a.Hq);
    float mn = 0.f, scale = 1.f;
    if (a.mm) { mn = daa_unkey(a.mm[0]); scale = daa_unkey(a.mm[1]) - mn; }
    float in[4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int y = min(max(yq * 2 + i - a.ph1, 0), a.h - 1), x = min(max(xq * 2 + j - a.pw1, 0), a.w_ - 1);
            float v = a.x[((long)b * a.h + y) * a.w_ + x];
            if (a.mm) {
                v = (v - mn) / scale;                                  // depth_aa.py:49-51, nan_to_num
                if (v != v) v = 0.f;
                else if (v > 3.4028235e38f) v = 3.4028235e38f;
                else if (v < -3.4028235e38f) v = -3.4028235e38f;
            }
            in[i * 2 + j] = v;
        }
    f16 *o = a.out + id * 32;
    for (int c0 = 0; c0 < 32; c0 += 8) {
        f16x8 ov;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float acc = sw[4 * 32 + c0 + j];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc = fmaf(in[k], sw[k * 32 + c0 + j], acc);
            ov[j] = (f16)acc;
        }
        *reinterpret_cast<f16x8 *>(o + c0) = ov;
    }
}

struct Wmha8Args {
    f16 *x;                   // [B,H,W,32] in place
    const f16 *wfrag;         // 6 qkv fragments (part*2 + head) + 2 head_proj fragments (chained k order)
    const float *bqkv;        // [96]  (q pre-scaled by 16^-0.5 * log2e)
    const float *bproj;       // [32]
    const float *btab;        // [64][64] log2e * score bias [query][key]
    int B, H, W, n_windows, shift;   // shift: 0 or 4 (zero padding on all four sides)
};

__device__ __forceinline__ f16x8 cat8d(f16x4 lo, f16x4 hi) {
    return (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__global__ void __launch_bounds__(256) wmha8_kernel(Wmha8Args a) {
    __shared__ __attribute__((aligned(16))) f16x8 wl[8 * 64];
    __shared__ __attribute__((aligned(16))) float tb[64 * 64], bq[96], bp[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, grp = lane >> 4;
    for (int i = tid; i < 8 * 64; i += 256) wl[i] = reinterpret_cast<const f16x8 *>(a.wfrag)[i];
    for (int i = tid; i < 64 * 64; i += 256) tb[i] = a.btab[i];
    if (tid < 96) bq[tid] = a.bqkv[tid];
    if (tid < 32) bp[tid] = a.bproj[tid];
    __syncthreads();
    const f16x4 zero4 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    const f16x8 zero8 = cat8d(zero4, zero4);
    const int nwx = (a.W + 2 * a.shift) / 8, nwy = (a.H + 2 * a.shift) / 8;
    const f16x8 *wq = wl + lane;

    for (int wi = blockIdx.x * 4 + wave; wi < a.n_windows; wi += gridDim.x * 4) {
        const int wx = wi % nwx, t2 = wi / nwx;
        const int wy = t2 % nwy, b = t2 / nwy;
        // token tile mt holds window tokens 16 mt + r16 = rows 2 mt, 2 mt + 1 of the 8x8 window
        long pix[4];
        bool inside[4];
        f16x8 xf[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int t = 16 * mt + r16;
            const int y = wy * 8 + (t >> 3) - a.shift, x = wx * 8 + (t & 7) - a.shift;
            inside[mt] = y >= 0 && y < a.H && x >= 0 && x < a.W;
            pix[mt] = ((long)b * a.H + min(max(y, 0), a.H - 1)) * a.W + min(max(x, 0), a.W - 1);
            xf[mt] = inside[mt] ? *reinterpret_cast<const f16x8 *>(a.x + pix[mt] * 32 + 8 * grp) : zero8;
        }
        f16x4 o4[2][4];                                         // [head][query tile]: channels 4g+r of the head
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            // q, k: [channel 4g+r][token];  v (operands swapped): [token 4g+r][channel l&15]
            f16x4 q4[4], k4[4], v4[4];
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                const int ch0 = part * 32 + hh * 16;
                const f16x8 w = wq[(part * 2 + hh) * 64];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    f32x4 acc;
                    if (part == 2) { const float bv = bq[ch0 + r16]; acc = (f32x4){bv, bv, bv, bv}; }
                    else acc = *reinterpret_cast<const f32x4 *>(bq + ch0 + 4 * grp);
                    acc = part == 2 ? MFMA_16x16x32(xf[mt], w, acc) : MFMA_16x16x32(w, xf[mt], acc);
                    const f16x4 v = {(f16)acc[0], (f16)acc[1], (f16)acc[2], (f16)acc[3]};
                    if (part == 0) q4[mt] = v; else if (part == 1) k4[mt] = v; else v4[mt] = v;
                }
            }
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                // S^T[key][query] for the 4 key tiles; the learned bias [query][key] is the accumulator's initial value
                f32x4 s[4];
                float mx = -3.0e38f;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const f32x4 bias = *reinterpret_cast<const f32x4 *>(tb + (16 * qt + r16) * 64 + 16 * kt + 4 * grp);
                    s[kt] = MFMA_16x16x32(cat8d(k4[kt], zero4), cat8d(q4[qt], zero4), bias);
                    mx = fmaxf(mx, fmaxf(fmaxf(s[kt][0], s[kt][1]), fmaxf(s[kt][2], s[kt][3])));
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                float sum = 0.f;
                f16x4 p[4];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const float p0 = __builtin_amdgcn_exp2f(s[kt][0] - mx), p1 = __builtin_amdgcn_exp2f(s[kt][1] - mx);
                    const float p2 = __builtin_amdgcn_exp2f(s[kt][2] - mx), p3 = __builtin_amdgcn_exp2f(s[kt][3] - mx);
                    sum += (p0 + p1) + (p2 + p3);
                    p[kt] = (f16x4){(f16)p0, (f16)p1, (f16)p2, (f16)p3};
                }
                sum += __shfl_xor(sum, 16);
                sum += __shfl_xor(sum, 32);
                const float inv = 1.0f / sum;
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                o = MFMA_16x16x32(cat8d(v4[0], v4[1]), cat8d(p[0], p[1]), o);
                o = MFMA_16x16x32(cat8d(v4[2], v4[3]), cat8d(p[2], p[3]), o);
                o4[hh][qt] = (f16x4){(f16)(o[0] * inv), (f16)(o[1] * inv), (f16)(o[2] * inv), (f16)(o[3] * inv)};
            }
        }
        // head_proj: K = 32 = [head 0's 16 channels | head 1's 16 channels] in the chained slot order
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                f32x4 acc = *reinterpret_cast<const f32x4 *>(bp + nt * 16 + 4 * grp);
                acc = MFMA_16x16x32(wq[(6 + nt) * 64], cat8d(o4[0][mt], o4[1][mt]), acc);
                if (inside[mt]) {
                    f16 *px = a.x + pix[mt] * 32 + nt * 16 + 4 * grp;
                    const f16x4 xr = *reinterpret_cast<const f16x4 *>(px);
                    *reinterpret_cast<f16x4 *>(px) = (f16x4){(f16)(acc[0] + (float)xr[0]), (f16)(acc[1] + (float)xr[1]),
                                                             (f16)(acc[2] + (float)xr[2]), (f16)(acc[3] + (float)xr[3])};
                }
            }
    }
}

struct DaaOutArgs {
    const f16 *f;             // [B,Hq,Wq,32]
    const float *src;         // [B,1,h,w] the un-normalised input
    const unsigned int *mm;   // or NULL
    const float *w;           // [32][4] then bias[4]
    float *out;               // [B,1,h,w]
    int B, h, w_, Hq, Wq, ph1, pw1, clamp01;
};

__global__ void __launch_bounds__(256) daa_out_kernel(DaaOutArgs a) {
    __shared__ float sw[33 * 4];
    if (threadIdx.x < 132) sw[threadIdx.x] = a.w[threadIdx.x];
    __syncthreads();
    const long total = (long)a.B * a.h * a.w_;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int x = (int)(id % a.w_);
    const long t = id / a.w_;
    const int y = (int)(t % a.h), b = (int)(t / a.h);
    const int yp = y + a.ph1, xp = x + a.pw1;
    const int n = (yp & 1) * 2 + (xp & 1);                            // pixel_shuffle 2: channel i*2 + j
    const f16 *p = a.f + (((long)b * a.Hq + (yp >> 1)) * a.Wq + (xp >> 1)) * 32;
    float acc = sw[32 * 4 + n];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc = fmaf((float)p[k], sw[k * 4 + n], acc);
    float v = a.src[id];
    if (a.mm) {
        const float mn = daa_unkey(a.mm[0]), scale = daa_unkey(a.mm[1]) - mn;
        float z = (v - mn) / scale;
        if (z != z) z = 0.f;
        v = (z + acc) * scale + mn;                                   // forward(clamp=False) * scale + min
    } else {
        v = v + acc;
        if (a.clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
    }
    a.out[id] = v;
}

}  // namespace nunif

using namespace nunif;

namespace {
struct HostT { const float *data; std::vector<int64_t> shape; int64_t numel; };
typedef std::map<std::string, HostT> TMap;
int find(const TMap &m, const std::string &key, const HostT **out) {
    auto it = m.find(key);
    if (it == m.end()) { set_error("state_dict is missing '%s'", key.c_str()); return NUNIF_HIP_EMISSING; }
    *out = &it->second;
    return NUNIF_HIP_OK;
}
struct Buf {
    void *p = nullptr; size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return NUNIF_HIP_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        if (hipMalloc(&p, bytes) != hipSuccess) { set_error("hipMalloc(%zu) failed", bytes); return NUNIF_HIP_ENOMEM; }
        cap = bytes;
        return NUNIF_HIP_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
struct DaaBlock { f16 *wfrag = nullptr, *w1 = nullptr, *w3 = nullptr; float *bqkv = nullptr, *bproj = nullptr, *btab = nullptr, *b1 = nullptr, *b3 = nullptr; };
double gelu_d(double v) { return 0.5 * v * (1.0 + erf(v * 0.70710678118654752440)); }
template <typename F>
void put_frag(std::vector<f16> &dst, size_t frag, int nt, int ks, bool chained, F wt) {
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
            const int g = l >> 4, n = nt * 16 + (l & 15);
            const int k = chained ? ks * 32 + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4)) : ks * 32 + g * 8 + j;
            dst[(frag * 64 + l) * 8 + j] = (f16)wt(n, k);
        }
}
}  // namespace

struct nunif_depth_aa {
    std::vector<void *> owned;
    float *w_in = nullptr, *w_out = nullptr;
    DaaBlock blk[3];
    Buf f, t1, t2, mm;
};

namespace {
template <typename T>
int upload(nunif_depth_aa *h, const std::vector<T> &host, T **dev) {
    void *p = nullptr;
    if (hipMalloc(&p, host.size() * sizeof(T)) != hipSuccess) { set_error("hipMalloc failed"); return NUNIF_HIP_ENOMEM; }
    h->owned.push_back(p);
    NUNIF_HIP_CHECK(hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *dev = reinterpret_cast<T *>(p);
    return NUNIF_HIP_OK;
}

int make_daa_block(nunif_depth_aa *h, const TMap &m, const std::string &p, DaaBlock *bk) {
    const HostT *wqkv, *bqkv, *wp, *bp, *w1, *b1, *w3, *b3, *tw0, *tb0, *tw2, *tb2;
    int rc;
    if ((rc = find(m, p + "mha.mha.qkv_proj.weight", &wqkv)) || (rc = find(m, p + "mha.mha.qkv_proj.bias", &bqkv)) ||
        (rc = find(m, p + "mha.mha.head_proj.weight", &wp)) || (rc = find(m, p + "mha.mha.head_proj.bias", &bp)) ||
        (rc = find(m, p + "conv_mlp.0.weight", &w1)) || (rc = find(m, p + "conv_mlp.0.bias", &b1)) ||
        (rc = find(m, p + "conv_mlp.3.weight", &w3)) || (rc = find(m, p + "conv_mlp.3.bias", &b3)) ||
        (rc = find(m, p + "bias.to_bias.0.weight", &tw0)) || (rc = find(m, p + "bias.to_bias.0.bias", &tb0)) ||
        (rc = find(m, p + "bias.to_bias.2.weight", &tw2)) || (rc = find(m, p + "bias.to_bias.2.bias", &tb2)))
        return rc;
    NUNIF_REQUIRE(wqkv->numel == 96 * 32 && wp->numel == 32 * 32 && w1->numel == 32 * 32 && w3->numel == 32 * 32 * 9,
                  "%s: depth_aa expects 32 channels", p.c_str());
    const float qs = 0.25f * 1.4426950408889634f;                       // 16^-0.5 * log2(e)
    {
        std::vector<f16> frags((size_t)8 * 512);
        const float *wd = wqkv->data;
        for (int part = 0; part < 3; ++part)
            for (int hh = 0; hh < 2; ++hh)       // one 16-row tile per (part, head); K = 32 = one k-step
                put_frag(frags, (size_t)part * 2 + hh, 0, 0, false, [=](int n, int k) {
                    return wd[(size_t)(part * 32 + hh * 16 + n) * 32 + k] * (part == 0 ? qs : 1.0f); });
        const float *pd = wp->data;
        for (int nt = 0; nt < 2; ++nt)
            put_frag(frags, (size_t)6 + nt, nt, 0, true, [=](int n, int k) { return pd[(size_t)n * 32 + k]; });
        std::vector<float> bq(96), bpv(bp->data, bp->data + 32);
        for (int n = 0; n < 96; ++n) bq[n] = bqkv->data[n] * (n < 32 ? qs : 1.0f);
        if ((rc = upload(h, frags, &bk->wfrag)) || (rc = upload(h, bq, &bk->bqkv)) || (rc = upload(h, bpv, &bk->bproj))) return rc;
    }
    {   // WindowScoreBias(8): 64 x 64 table from the to_bias MLP on (dy, dx) / 7
        const int hidden = (int)tb0->numel;
        NUNIF_REQUIRE(tw0->numel == hidden * 2 && tw2->numel == hidden && tb2->numel == 1, "%s: score-bias MLP shape", p.c_str());
        std::vector<float> tab(64 * 64);
        for (int q = 0; q < 64; ++q)
            for (int k = 0; k < 64; ++k) {
                const float dy = (float)(q / 8 - k / 8) / 7.0f, dx = (float)(q % 8 - k % 8) / 7.0f;
                double o = tb2->data[0];
                for (int j = 0; j < hidden; ++j)
                    o += (double)tw2->data[j] * gelu_d((double)tw0->data[j * 2] * dy + (double)tw0->data[j * 2 + 1] * dx + (double)tb0->data[j]);
                tab[q * 64 + k] = (float)o * 1.4426950408889634f;
            }
        if ((rc = upload(h, tab, &bk->btab))) return rc;
    }
    {
        std::vector<f16> packed((size_t)32 * 32 + 8192, (f16)0.f);
        const float *wd = w1->data;
        for (int nt = 0; nt < 2; ++nt) put_frag(packed, (size_t)nt, nt, 0, false, [=](int n, int k) { return wd[(size_t)n * 32 + k]; });
        std::vector<float> bb(b1->data, b1->data + 32);
        if ((rc = upload(h, packed, &bk->w1)) || (rc = upload(h, bb, &bk->b1))) return rc;
    }
    {
        std::vector<f16> stream((size_t)9 * 2 * 512 + 8192, (f16)0.f);
        const float *wd = w3->data;
        for (int ks = 0; ks < 9; ++ks)
            for (int nt = 0; nt < 2; ++nt)
                put_frag(stream, (size_t)ks * 2 + nt, nt, ks, false, [=](int n, int k) {
                    const int tap = k / 32, ci = k % 32;
                    return wd[((size_t)n * 32 + ci) * 9 + tap]; });
        std::vector<float> bb(b3->data, b3->data + 32);
        if ((rc = upload(h, stream, &bk->w3)) || (rc = upload(h, bb, &bk->b3))) return rc;
    }
    return NUNIF_HIP_OK;
}
}  // namespace

extern "C" int nunif_hip_depth_aa_create(const nunif_tensor_desc *tensors, int32_t n_tensors, nunif_depth_aa **handle) {
    NUNIF_REQUIRE(tensors && handle && n_tensors > 0, "depth_aa_create: NULL argument");
    TMap m;
    for (int i = 0; i < n_tensors; ++i) {
        HostT t;
        t.data = tensors[i].data;
        t.numel = 1;
        for (int d = 0; d < tensors[i].ndim; ++d) { t.shape.push_back(tensors[i].shape[d]); t.numel *= tensors[i].shape[d]; }
        m[tensors[i].name] = t;
    }
    nunif_depth_aa *h = new nunif_depth_aa();
    int rc = NUNIF_HIP_OK;
    do {
        const HostT *wi, *bi, *wo, *bo;
        if ((rc = find(m, "proj_in.weight", &wi)) || (rc = find(m, "proj_in.bias", &bi)) ||
            (rc = find(m, "proj_out.weight", &wo)) || (rc = find(m, "proj_out.bias", &bo)))
            break;
        if (wi->numel != 32 * 4 || wo->numel != 4 * 32) { set_error("depth_aa: unexpected proj shapes"); rc = NUNIF_HIP_EINVAL; break; }
        std::vector<float> win(5 * 32), wout(33 * 4);
        for (int k = 0; k < 4; ++k) for (int co = 0; co < 32; ++co) win[k * 32 + co] = wi->data[co * 4 + k];
        for (int co = 0; co < 32; ++co) win[4 * 32 + co] = bi->data[co];
        for (int k = 0; k < 32; ++k) for (int n = 0; n < 4; ++n) wout[k * 4 + n] = wo->data[n * 32 + k];
        for (int n = 0; n < 4; ++n) wout[32 * 4 + n] = bo->data[n];
        if ((rc = upload(h, win, &h->w_in)) || (rc = upload(h, wout, &h->w_out))) break;
        for (int i = 0; i < 3 && !rc; ++i) rc = make_daa_block(h, m, "blocks." + std::to_string(i) + ".", &h->blk[i]);
    } while (0);
    if (rc) { nunif_hip_depth_aa_destroy(h); return rc; }
    *handle = h;
    return NUNIF_HIP_OK;
}

extern "C" void nunif_hip_depth_aa_destroy(nunif_depth_aa *h) {
    if (!h) return;
    for (void *p : h->owned) (void)hipFree(p);
    h->f.release(); h->t1.release(); h->t2.release(); h->mm.release();
    delete h;
}

extern "C" int nunif_hip_depth_aa_forward(nunif_depth_aa *h, const float *x, float *y, int32_t B, int32_t hh, int32_t ww,
                                          int32_t mode, void *stream) {
    NUNIF_REQUIRE(h && x && y && B > 0 && hh > 0 && ww > 0 && mode >= 0 && mode <= 2, "depth_aa_forward: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int pad_w = 16 - ww % 16, pad_h = 16 - hh % 16;               // depth_aa.py:61-66
    const int pw1 = pad_w / 2, ph1 = pad_h / 2;
    const int Hq = (hh + pad_h) / 2, Wq = (ww + pad_w) / 2;
    const size_t tok = (size_t)B * Hq * Wq;
    int rc;
    if ((rc = h->f.ensure(tok * 32 * sizeof(f16))) || (rc = h->t1.ensure(tok * 32 * sizeof(f16))) ||
        (rc = h->t2.ensure(tok * 32 * sizeof(f16))) || (rc = h->mm.ensure(16)))
        return rc;
    f16 *f = (f16 *)h->f.p, *t1 = (f16 *)h->t1.p, *t2 = (f16 *)h->t2.p;
    unsigned int *mm = mode == 2 ? (unsigned int *)h->mm.p : nullptr;
    const long px = (long)B * hh * ww;
    if (mm) {
        daa_minmax_init_kernel<<<1, 1, 0, s>>>(mm);
        daa_minmax_kernel<<<(unsigned)std::min<long>((px + 255) / 256, 1024), 256, 0, s>>>(x, mm, px);
        NUNIF_LAUNCH_CHECK();
    }
    {
        ProfScope ps("daa_in_kernel", s, 2.0 * 4 * 32 * (double)tok, (double)tok * (16.0 + 64.0));
        DaaInArgs a;
        a.x = x; a.mm = mm; a.w = h->w_in; a.out = f; a.B = B; a.h = hh; a.w_ = ww; a.Hq = Hq; a.Wq = Wq; a.ph1 = ph1; a.pw1 = pw1;
        daa_in_kernel<<<(unsigned)((tok + 255) / 256), 256, 0, s>>>(a);
        NUNIF_LAUNCH_CHECK();
    }
    f16 *cur = f, *other = t2;
    for (int bi = 0; bi < 3; ++bi) {
        const DaaBlock &bk = h->blk[bi];
        {
            Wmha8Args a;
            a.x = cur; a.wfrag = bk.wfrag; a.bqkv = bk.bqkv; a.bproj = bk.bproj; a.btab = bk.btab;
            a.B = B; a.H = Hq; a.W = Wq; a.shift = (bi % 2 == 0) ? 4 : 0;       // shift True / False / True
            a.n_windows = B * ((Hq + 2 * a.shift) / 8) * ((Wq + 2 * a.shift) / 8);
            ProfScope ps("wmha8_kernel", s, (double)tok * (8.0 * 32 * 32 + 4.0 * 64 * 32), (double)tok * 128.0);
            wmha8_kernel<<<std::min((a.n_windows + 3) / 4, 2048), 256, 0, s>>>(a);
            NUNIF_LAUNCH_CHECK();
        }
        {
            GemmArgs g;
            memset(&g, 0, sizeof(g));
            g.a = cur; g.B = B; g.Hi = Hq; g.Wi = Wq; g.Cin = 32; g.Ho = Hq; g.Wo = Wq; g.stride = 1; g.kw = 1;
            g.K = 32; g.w = bk.w1; g.bias = bk.b1; g.N = 32; g.mode = 0; g.act = 1; g.out = t1; g.ldo = 32; g.n_real = 32; g.ps = 1;
            if ((rc = launch_gemm(g, s, "depth_aa_mlp0"))) return rc;
        }
        {
            ConvArgs c;
            memset(&c, 0, sizeof(c));
            c.a = t1; c.B = B; c.Hi = Hq; c.Wi = Wq; c.Cin = 32; c.Ho = Hq; c.Wo = Wq; c.stride = 1; c.kh = 3; c.kw = 3;
            c.wstream = bk.w3; c.bias = bk.b3; c.N = 32; c.n_real = 32; c.act = 2; c.slope = 0.1f; c.out = other;
            c.rpad = 1; c.res = cur;
            if ((rc = launch_conv(c, s))) return rc;
        }
        std::swap(cur, other);
    }
    {
        ProfScope ps("daa_out_kernel", s, 2.0 * 32 * (double)px, (double)px * (8.0 + 64.0));
        DaaOutArgs a;
        a.f = cur; a.src = x; a.mm = mm; a.w = h->w_out; a.out = y; a.B = B; a.h = hh; a.w_ = ww; a.Hq = Hq; a.Wq = Wq;
        a.ph1 = ph1; a.pw1 = pw1; a.clamp01 = mode == 1 ? 1 : 0;
        daa_out_kernel<<<(unsigned)((px + 255) / 256), 256, 0, s>>>(a);
        NUNIF_LAUNCH_CHECK();
    }
    return NUNIF_HIP_OK;
}
