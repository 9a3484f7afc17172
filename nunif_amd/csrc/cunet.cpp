// Host side of the waifu2x CUNet engine: weight repacking, workspace, launch sequence, whole-frame render.
//
// Reference: waifu2x/models/cunet.py — UNet1 :31-67, UNet2 :70-121, CUNet :172-203 (layer inventory, crops, the
// cascade z = crop(z1, 20) + unet2(z1)); nunif/utils/seam_blending.py tiled_render :48-106 (frame loop; CUNet has
// blend_size None -> plain overwrite).  State-dict keys are the reference's.
#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "swin_kernels.h"

namespace nunif {
int launch_stitch(const float *tile_out, float *y, const nunif_tile_grid *g, int C, hipStream_t s, int y0 = 0, int rows = -1,
                  int compact = 0);
}

using namespace nunif;

namespace {

struct HostT { const float *data; std::vector<int64_t> shape; int64_t numel; };
typedef std::map<std::string, HostT> TMap;

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

struct ConvW { f16 *stream = nullptr; float *bias = nullptr; int N = 0, n_real = 0, Cin = 0, k = 3, stride = 1;
               f16 *gemm_w = nullptr;           // 2x2 stride-2 convs: the same weights in gemm_kernel's [n-tile][k-step] order
               // 3x3 convs with 128 / 256 inputs and more than 64 outputs (the two bottom convs of unet2): the SAME weights as
               // streams of 64 output channels each, so that every slice runs on the LDS-DMA conv with K halves
               // (conv3_dma_kernel<4, 64, ., 1, 2, KS>), writing its channels of the NHWC map through ConvArgs::ldo
               std::vector<f16 *> slice;
               f16 *head_w = nullptr; };        // 64 -> 3 image heads: the tap-scatter form's fragments (cunet_head.hip), row n = 3 tap + c
struct UpW { f16 *w = nullptr; float *bias = nullptr; int N = 0, K = 0, cq = 0; };       // ConvTranspose2d 2x2 s2
struct SEW { float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr; int C = 0; };
struct C3W { float *w = nullptr, *b = nullptr; int C = 0; f16 *frag = nullptr; };   // frag: MFMA A fragments of the fused stem (K = 27 taps + bias)

}  // namespace

struct nunif_cunet {
    // kind 0 / 1: CUNet / UpCUNet; 2: waifu2x.vgg_7 (vgg_7.py:6-30: seven 3x3 VALID convs, scale 1, offset 7);
    // 3: waifu2x.upconv_7 (upconv_7.py:6-35: six 3x3 VALID convs + ConvTranspose2d(256, 3, 4, 2, 3), scale 2, offset 14)
    int kind = 0;
    C3W st_first; ConvW st_conv[6]; UpW st_deconv; int st_n = 0;      // the plain conv stacks
    int no_clip = 0;
    int up = 0;                 // 1: UpCUNet (unet1 ends in ConvTranspose2d(64, 3, 4, 2, 3); scale 2, offset 36)
    UpW u1bottom_up;            // that head as a 2x2-window gather GEMM (K = 4*64) with a pixel-shuffle store
    std::vector<void *> owned;
    // unet1
    C3W u1c1a; ConvW u1c1b, u1down, u1c2a, u1c2b, u1c3, u1bottom; SEW u1se2; UpW u1up;
    // unet2
    C3W u2c1a; ConvW u2c1b, u2down1, u2c2a, u2c2b, u2down2, u2c3a, u2c3b, u2c4a, u2c4b, u2c5, u2bottom;
    SEW u2se2, u2se3, u2se4; UpW u2up3, u2up4;
    Buf t[12], z1, sums, scale, tile_out;
};

namespace {

int find(const TMap &m, const std::string &key, const HostT **out) {
    auto it = m.find(key);
    if (it == m.end()) { set_error("state_dict is missing '%s'", key.c_str()); return NUNIF_HIP_EMISSING; }
    *out = &it->second;
    return NUNIF_HIP_OK;
}

template <typename T>
int upload(nunif_cunet *h, const std::vector<T> &host, T **dev) {
    void *p = nullptr;
    if (hipMalloc(&p, host.size() * sizeof(T)) != hipSuccess) { set_error("hipMalloc failed"); return NUNIF_HIP_ENOMEM; }
    h->owned.push_back(p);
    NUNIF_HIP_CHECK(hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *dev = reinterpret_cast<T *>(p);
    return NUNIF_HIP_OK;
}

int upload_f32(nunif_cunet *h, const HostT *t, float **dev) {
    std::vector<float> v(t->data, t->data + t->numel);
    return upload(h, v, dev);
}

// Conv2d weight [Cout][Cin][k][k] -> MFMA A fragments in [k-step][n-tile] order, reduction index = tap*Cin + ci
// cin_real < cin: the producer stores its cin_real channels padded with zeros to cin (a multiple of 32)
int make_conv(nunif_cunet *h, const TMap &m, const std::string &key, int cin, int cout, int k, int stride, ConvW *c,
              int cin_real = 0) {
    const HostT *w, *b;
    int rc;
    if (cin_real <= 0) cin_real = cin;
    if ((rc = find(m, key + ".weight", &w)) || (rc = find(m, key + ".bias", &b))) return rc;
    NUNIF_REQUIRE(w->numel == (int64_t)cout * cin_real * k * k && b->numel == cout, "%s: unexpected shape", key.c_str());
    NUNIF_REQUIRE(cin % 32 == 0, "%s: Cin=%d must be a multiple of 32", key.c_str(), cin);
    const int N = (cout + 15) / 16 * 16, NT = N / 16, KS = k * k * cin / 32;
    std::vector<f16> stream((size_t)KS * NT * 512 + 8192, (f16)0.0f);
    for (int ks = 0; ks < KS; ++ks)
        for (int nt = 0; nt < NT; ++nt)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int n = nt * 16 + (l & 15), kk = ks * 32 + (l >> 4) * 8 + j;
                    const int tap = kk / cin, ci = kk % cin;
                    const float v = (n < cout && ci < cin_real) ? w->data[((size_t)n * cin_real + ci) * k * k + tap] : 0.0f;
                    stream[(((size_t)ks * NT + nt) * 64 + l) * 8 + j] = (f16)v;
                }
    std::vector<float> bias(N, 0.0f);
    for (int n = 0; n < cout; ++n) bias[n] = b->data[n];
    c->N = N; c->n_real = cout; c->Cin = cin; c->k = k; c->stride = stride;
    if ((rc = upload(h, stream, &c->stream))) return rc;
    if (k == 3 && stride == 1 && cin_real == cin && (cin == 128 || cin == 256) && cout > 64 && cout % 64 == 0) {
        for (int sl = 0; sl < cout / 64; ++sl) {
            std::vector<f16> part((size_t)KS * 4 * 512 + 8192, (f16)0.0f);
            for (int ks = 0; ks < KS; ++ks)
                std::copy(stream.begin() + ((size_t)ks * NT + 4 * sl) * 512, stream.begin() + ((size_t)ks * NT + 4 * sl + 4) * 512,
                          part.begin() + (size_t)ks * 4 * 512);
            f16 *dev = nullptr;
            if ((rc = upload(h, part, &dev))) return rc;
            c->slice.push_back(dev);
        }
    }
    if (k == 3 && stride == 1 && cin_real == 64 && cin == 64 && cout == 3) {
        std::vector<f16> hw((size_t)4 * 512 + 8192, (f16)0.0f);
        for (int nt = 0; nt < 2; ++nt)
            for (int ks = 0; ks < 2; ++ks)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int n = nt * 16 + (l & 15), ci = ks * 32 + (l >> 4) * 8 + j, tap = n / 3, co = n % 3;
                        hw[(((size_t)nt * 2 + ks) * 64 + l) * 8 + j] = n < 27 ? (f16)w->data[((size_t)co * 64 + ci) * 9 + tap] : (f16)0.0f;
                    }
        if ((rc = upload(h, hw, &c->head_w))) return rc;
    }
    if (k == 2 && stride == 2 && cin_real == cin && N == cout && N % 32 == 0) {
        // k = stride: a 2 x 2 gather GEMM (the PatchDown form of gemm_kernel: every input pixel is read exactly once, the token
        // tile's whole K extent sits in registers) instead of the K-looped conv_kernel with its nine-tap machinery
        std::vector<f16> packed((size_t)N * k * k * cin + 8192, (f16)0.0f);
        for (int nt = 0; nt < NT; ++nt)
            for (int ks = 0; ks < KS; ++ks)
                std::copy(stream.begin() + ((size_t)ks * NT + nt) * 512, stream.begin() + ((size_t)ks * NT + nt + 1) * 512,
                          packed.begin() + ((size_t)nt * KS + ks) * 512);
        if ((rc = upload(h, packed, &c->gemm_w))) return rc;
    }
    return upload(h, bias, &c->bias);
}

// ConvTranspose2d(cin, cout, 2, 2) weight [cin][cout][2][2] -> Linear cin -> 4*cout with pixel-shuffle column order
// n = q*cout + co, q = i*2 + j  (gemm_kernel mode 1), fragments in [n-tile][k-step] order
int make_up(nunif_cunet *h, const TMap &m, const std::string &key, int cin, int cout, UpW *u) {
    const HostT *w, *b;
    int rc;
    if ((rc = find(m, key + ".weight", &w)) || (rc = find(m, key + ".bias", &b))) return rc;
    NUNIF_REQUIRE(w->numel == (int64_t)cin * cout * 4 && b->numel == cout, "%s: unexpected shape", key.c_str());
    const int N = 4 * cout, NT = N / 16, KS = cin / 32;
    std::vector<f16> packed((size_t)N * cin + 8192, (f16)0.0f);
    for (int nt = 0; nt < NT; ++nt)
        for (int ks = 0; ks < KS; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int n = nt * 16 + (l & 15), ci = ks * 32 + (l >> 4) * 8 + j;
                    const int q = n / cout, co = n % cout;
                    packed[(((size_t)nt * KS + ks) * 64 + l) * 8 + j] = (f16)w->data[((size_t)ci * cout + co) * 4 + q];
                }
    std::vector<float> bias(N);
    for (int n = 0; n < N; ++n) bias[n] = b->data[n % cout];
    u->N = N; u->K = cin; u->cq = cout;
    if ((rc = upload(h, packed, &u->w))) return rc;
    return upload(h, bias, &u->bias);
}

// ConvTranspose2d(cin, cout, 4, 2, 3), weight [cin][cout][4][4]:  out[oy][ox] = b + sum in[iy][ix] W[ky][kx] over
// oy = 2 iy - 3 + ky.  The 2x2 input window (rows j, j+1; cols i, i+1) produces exactly the 2x2 output pixels
// (2j-1 .. 2j) x (2i-1 .. 2i): window row dy contributes to output row parity a through ky = a + 2 (1 - dy) (same for
// columns).  So the head is a gather GEMM over the (H-1)^2 windows, K = (dy*2+dx)*cin + ci (gemm_kernel taps with
// kw = 2), N = co*4 + a*2 + b, stored by gemm_kernel mode 2 with oshift = -1 into the (2H-4)^2 plane.
int make_deconv4(nunif_cunet *h, const TMap &m, const std::string &key, int cin, int cout, UpW *u) {
    const HostT *w, *b;
    int rc;
    if ((rc = find(m, key + ".weight", &w)) || (rc = find(m, key + ".bias", &b))) return rc;
    NUNIF_REQUIRE(w->numel == (int64_t)cin * cout * 16 && b->numel == cout, "%s: unexpected shape", key.c_str());
    const int n_real = 4 * cout, N = (n_real + 15) / 16 * 16, NT = N / 16, K = 4 * cin, KS = K / 32;
    std::vector<f16> packed((size_t)N * K + 8192, (f16)0.0f);
    for (int nt = 0; nt < NT; ++nt)
        for (int ks = 0; ks < KS; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int n = nt * 16 + (l & 15), k = ks * 32 + (l >> 4) * 8 + j;
                    if (n >= n_real) continue;
                    const int co = n / 4, a = (n >> 1) & 1, bb = n & 1;
                    const int tap = k / cin, ci = k % cin, dy = tap >> 1, dx = tap & 1;
                    const int ky = a + 2 * (1 - dy), kx = bb + 2 * (1 - dx);
                    packed[(((size_t)nt * KS + ks) * 64 + l) * 8 + j] = (f16)w->data[(((size_t)ci * cout + co) * 4 + ky) * 4 + kx];
                }
    std::vector<float> bias(N, 0.0f);
    for (int n = 0; n < n_real; ++n) bias[n] = b->data[n / 4];
    u->N = N; u->K = K; u->cq = cout;
    if ((rc = upload(h, packed, &u->w))) return rc;
    return upload(h, bias, &u->bias);
}

// cout_pad > cout: the extra output channels get zero weights / bias (LeakyReLU(0) = 0), so the next conv sees Cin % 32 == 0
int make_c3(nunif_cunet *h, const TMap &m, const std::string &key, int cout, C3W *c, int cout_pad = 0) {
    const HostT *w, *b;
    int rc;
    if ((rc = find(m, key + ".weight", &w)) || (rc = find(m, key + ".bias", &b))) return rc;
    NUNIF_REQUIRE(w->numel == (int64_t)cout * 27 && b->numel == cout, "%s: unexpected shape (3 input channels)", key.c_str());
    if (cout_pad <= cout) {
        c->C = cout;
        if (cout % 16 == 0) {
            // the same conv as cout / 16 MFMA A fragments for stem_fused_kernel (swin_stem.hip): row n, k = ci*9 + ky*3 + kx
            // (the weight's own layout), k = 27: the bias (the B operand carries a constant one there), k > 27: zero
            std::vector<f16> fr((size_t)cout / 16 * 512);
            for (int nt = 0; nt < cout / 16; ++nt)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int n = nt * 16 + (l & 15), k = (l >> 4) * 8 + j;
                        fr[((size_t)nt * 64 + l) * 8 + j] = (f16)(k < 27 ? w->data[(size_t)n * 27 + k] : (k == 27 ? b->data[n] : 0.0f));
                    }
            if ((rc = upload(h, fr, &c->frag))) return rc;
        }
        if ((rc = upload_f32(h, w, &c->w))) return rc;
        return upload_f32(h, b, &c->b);
    }
    std::vector<float> wp((size_t)cout_pad * 27, 0.0f), bp(cout_pad, 0.0f);
    std::copy(w->data, w->data + (size_t)cout * 27, wp.begin());
    std::copy(b->data, b->data + cout, bp.begin());
    c->C = cout_pad;
    if ((rc = upload(h, wp, &c->w))) return rc;
    return upload(h, bp, &c->b);
}

int make_se(nunif_cunet *h, const TMap &m, const std::string &key, int C, SEW *s) {
    const HostT *w1, *b1, *w2, *b2;
    int rc;
    if ((rc = find(m, key + ".conv1.weight", &w1)) || (rc = find(m, key + ".conv1.bias", &b1)) ||
        (rc = find(m, key + ".conv2.weight", &w2)) || (rc = find(m, key + ".conv2.bias", &b2)))
        return rc;
    NUNIF_REQUIRE(w1->numel == (int64_t)C * C / 8 && w2->numel == (int64_t)C * C / 8, "%s: unexpected shape", key.c_str());
    s->C = C;
    if ((rc = upload_f32(h, w1, &s->w1)) || (rc = upload_f32(h, b1, &s->b1)) || (rc = upload_f32(h, w2, &s->w2)) ||
        (rc = upload_f32(h, b2, &s->b2)))
        return rc;
    return NUNIF_HIP_OK;
}

int run_conv(const ConvW &c, const f16 *a, const f16 *a2, int H2, int crop2, int B, int Hi, f16 *out, float *out32,
             const float *add32, int addH, int add_crop, int clamp01, int act, hipStream_t s) {
    ConvArgs g;
    memset(&g, 0, sizeof(g));
    g.a = a; g.a2 = a2; g.H2 = H2; g.W2 = H2; g.crop2 = crop2;
    g.B = B; g.Hi = Hi; g.Wi = Hi; g.Cin = c.Cin; g.stride = c.stride; g.kh = c.k; g.kw = c.k;
    g.Ho = (Hi - c.k) / c.stride + 1; g.Wo = g.Ho;
    g.wstream = c.stream; g.bias = c.bias; g.N = c.N; g.n_real = c.n_real;
    g.act = act; g.slope = 0.1f;
    g.out = out; g.out32 = out32; g.add32 = add32; g.addH = addH; g.addW = addH; g.add_crop = add_crop;
    g.clamp01 = clamp01;
    if (c.head_w && out32 && !out && !a2 && act == 0) {
        // the 64 -> 3 image heads in tap-scatter form (cunet_head.hip), every launch of that shape
        CunetHeadArgs hg = {a, c.head_w, c.bias, out32, add32, B, Hi, Hi, g.Ho, g.Wo, addH, addH, add_crop, clamp01};
        if (cunet_head_supported(hg)) return launch_cunet_head(hg, s);
    }
    // (every launch of such a conv takes the sliced form or none does: a result must not depend on the tile minibatch)
    static const bool sliced = !(getenv("NUNIF_CUNET_SLICED") && atoi(getenv("NUNIF_CUNET_SLICED")) == 0);
    if (sliced && !c.slice.empty() && out && !out32 && !a2) {
        for (size_t sl = 0; sl < c.slice.size(); ++sl) {
            ConvArgs q = g;
            q.wstream = c.slice[sl]; q.bias = c.bias + 64 * sl; q.N = 64; q.n_real = 64;
            q.out = out + 64 * sl; q.ldo = c.n_real;
            int rc = launch_conv(q, s);
            if (rc) return rc;
        }
        return NUNIF_HIP_OK;
    }
    return launch_conv(g, s);
}

static inline bool down_gemm_enabled() { const char *e = getenv("NUNIF_CUNET_DOWN_GEMM"); return e ? atoi(e) != 0 : true; }

// Conv2d(k = stride = 2) + LeakyReLU (cunet.py:37,78,81) as a gather GEMM
int run_down(const ConvW &c, const f16 *a, int B, int Hi, f16 *out, hipStream_t s) {
    if (!c.gemm_w || !down_gemm_enabled()) return run_conv(c, a, nullptr, 0, 0, B, Hi, out, nullptr, nullptr, 0, 0, 0, 2, s);
    {   // 64 -> 64: the K-outer prefetching kernel of the swin PatchDown (swin_patchdown.hip), every launch of that shape
        PatchDownArgs p;
        p.a = a; p.w = c.gemm_w; p.bias = c.bias; p.out = out; p.B = B; p.Ho = Hi / 2; p.Wo = Hi / 2; p.Cin = c.Cin; p.oy = 0;
        p.rev = 0; p.N = c.N; p.act = 2; p.slope = 0.1f;
        if (Hi % 2 == 0 && patchdown_supported(p)) return launch_patchdown(p, s);
    }
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.a = a; g.B = B; g.Hi = Hi; g.Wi = Hi; g.Cin = c.Cin; g.Ho = Hi / 2; g.Wo = Hi / 2; g.stride = 2; g.kw = 2;
    g.K = 4 * c.Cin; g.w = c.gemm_w; g.bias = c.bias; g.N = c.N; g.mode = 0; g.act = 2; g.slope = 0.1f;
    g.out = out; g.ldo = c.N; g.n_real = c.N; g.ps = 1;
    return launch_gemm(g, s, "cunet_down");
}

// ConvTranspose2d(k = stride = 2) + LeakyReLU as a pixel-shuffle GEMM; with `skip`: + crop(skip, crop) in the epilogue, so that
// the next conv reads ONE input (`conv3(crop(x1) + x2)`, cunet.py:58-60,111-118 — the add used to ride in that conv's staging,
// which a DMA-staged conv cannot do)
int run_up(const UpW &u, const f16 *a, int B, int Hi, f16 *out, hipStream_t s, const f16 *skip = nullptr, int skip_side = 0,
           int crop = 0, const float *in_scale = nullptr) {
    if (u.K == 64 && u.cq == 64 && skip) {
        // 64 -> 4 x 64 with a cropped skip: the resident-weight prefetching kernel (cunet_up.hip), every launch of that shape
        CunetUpArgs cu = {a, u.w, u.bias, skip, out, in_scale, B, Hi, skip_side, crop, 0.1f};
        if (cunet_up_supported(cu)) return launch_cunet_up(cu, s);
    }
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.a = a; g.B = B; g.Hi = Hi; g.Wi = Hi; g.Cin = u.K; g.Ho = Hi; g.Wo = Hi; g.stride = 1; g.kw = 1;
    g.K = u.K; g.w = u.w; g.bias = u.bias; g.N = u.N; g.mode = 1; g.act = 2; g.slope = 0.1f;
    g.out = out; g.ldo = u.cq; g.n_real = u.N; g.ps = 1;
    if (skip) { g.res = skip; g.res_H = skip_side; g.res_W = skip_side; g.res_crop = crop; }
    g.in_scale = in_scale;                // the squeeze-excitation scale of `a`, applied as the fragments are loaded
    return launch_gemm(g, s, "cunet_up");
}

int run_deconv4(const UpW &u, const f16 *a, int B, int Hi, float *out, int no_clamp, hipStream_t s) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.a = a; g.B = B; g.Hi = Hi; g.Wi = Hi; g.Cin = u.K / 4; g.Ho = Hi - 1; g.Wo = Hi - 1; g.stride = 1; g.kw = 2;
    g.K = u.K; g.w = u.w; g.bias = u.bias; g.N = u.N; g.mode = 2; g.act = 0;
    g.out = out; g.n_real = 4 * u.cq; g.ps = 2; g.oshift = -1; g.OH = 2 * Hi - 4; g.OW = 2 * Hi - 4; g.no_clamp = no_clamp;
    return launch_gemm(g, s, "upcunet_bottom");
}

// SE blocks whose map has ONE consumer, a transposed-conv GEMM: pooled + MLP here, the channel scale rides in that GEMM's loads
static inline bool se_fuse_enabled() { const char *e = getenv("NUNIF_CUNET_SE_FUSE"); return e ? atoi(e) != 0 : true; }

static inline bool stem_fused_enabled() { const char *e = getenv("NUNIF_CUNET_STEM"); return e ? atoi(e) != 0 : true; }

// first UNetConv of a U-Net through stem_fused_kernel<32, 64, 0>: c3 carries the input geometry (tile / frame mode), c2.stream is
// already in the kernel's [k-step][n-tile] order with k = tap * 32 + ci (make_conv)
int run_stem(const C3ConvArgs &c3, const C3W &c1, const ConvW &c2, f16 *out, hipStream_t s) {
    StemFusedArgs sf;
    memset(&sf, 0, sizeof(sf));
    sf.x = c3.x; sf.frame_mode = c3.frame_mode; sf.H = c3.H; sf.W = c3.W; sf.wb = c3.wb; sf.istep = c3.istep;
    sf.pad_t = c3.pad_t; sf.pad_l = c3.pad_l; sf.tile_begin = c3.tile_begin;
    sf.B = c3.B; sf.T = c3.T; sf.w1 = c1.frag; sf.w2 = c2.stream; sf.b2 = c2.bias; sf.out = out; sf.slope = c3.slope;
    sf.C1 = 32; sf.C = 64; sf.crop = 0;
    return launch_stem_fused(sf, s);
}

// waifu2x.vgg_7 / waifu2x.upconv_7: first conv on the VALU (3 input channels), the 3x3 VALID convs on conv_kernel with
// LeakyReLU(0.1), image head = last conv (fp32 planar, clamp) or the 4x4 s2 p3 ConvTranspose as a gather GEMM.
// x: tile mode [B,3,T,T] or frame + grid; z: [B,3,T-14,T-14] (vgg_7) / [B,3,2T-28,2T-28] (upconv_7)
int forward_stack(nunif_cunet *h, const float *x, const float *frame, const nunif_tile_grid *grid, int tile_begin, float *z,
                  int B, int T, hipStream_t s) {
    NUNIF_REQUIRE(T > 14, "tile_size %d is too small (7 convolutions of 3x3)", T);
    int rc, cmax = 0;
    for (int i = 0; i < h->st_n; ++i) cmax = std::max(cmax, h->st_conv[i].N);
    const size_t bytes = (size_t)B * (T - 2) * (T - 2) * std::max(cmax, h->st_first.C) * sizeof(f16);
    if ((rc = h->t[0].ensure(bytes)) || (rc = h->t[1].ensure(bytes))) return rc;
    f16 *cur = (f16 *)h->t[0].p, *nxt = (f16 *)h->t[1].p;
    C3ConvArgs c3;
    memset(&c3, 0, sizeof(c3));
    if (frame) {
        c3.x = frame; c3.frame_mode = 1; c3.H = grid->x_h; c3.W = grid->x_w; c3.wb = grid->w_blocks;
        c3.istep = grid->input_tile_step; c3.pad_t = grid->pad_t; c3.pad_l = grid->pad_l; c3.tile_begin = tile_begin;
    } else {
        c3.x = x;
    }
    c3.B = B; c3.T = T; c3.w = h->st_first.w; c3.bias = h->st_first.b; c3.C = h->st_first.C; c3.out = cur; c3.slope = 0.1f;
    if ((rc = launch_c3_conv(c3, s))) return rc;
    int side = T - 2;
    const int n_mid = h->kind == 2 ? h->st_n - 1 : h->st_n;          // vgg_7: the last conv is the image head
    for (int i = 0; i < n_mid; ++i) {
        if ((rc = run_conv(h->st_conv[i], cur, nullptr, 0, 0, B, side, nxt, nullptr, nullptr, 0, 0, 0, 2, s))) return rc;
        std::swap(cur, nxt);
        side -= 2;
    }
    if (h->kind == 2)
        return run_conv(h->st_conv[h->st_n - 1], cur, nullptr, 0, 0, B, side, nullptr, z, nullptr, 0, 0, 1, 0, s);
    return run_deconv4(h->st_deconv, cur, B, side, z, 0, s);
}

// x: tile mode [B,3,T,T] or (frame != NULL) frame + grid; z: [B,3,T-56,T-56] (CUNet) / [B,3,2T-72,2T-72] (UpCUNet)
int forward_impl(nunif_cunet *h, const float *x, const float *frame, const nunif_tile_grid *grid, int tile_begin,
                 float *z, int B, int T, hipStream_t s) {
    if (h->kind >= 2) return forward_stack(h, x, frame, grid, tile_begin, z, B, T, s);
    NUNIF_REQUIRE(T % 4 == 0 && T >= 64, "tile_size %d is not valid for cunet (multiple of 4, >= 64)", T);
    const size_t b = B;
    const int a1 = T - 2, x1 = T - 4, d1 = x1 / 2, e1 = d1 - 2, f1 = d1 - 4, g1 = 2 * f1, h1 = g1 - 2;
    const int T2 = h->up ? 2 * h1 - 4 : g1 - 4;
    const int a2 = T2 - 2, y1 = T2 - 4, d2 = y1 / 2, e2 = d2 - 2, y2 = d2 - 4, d3 = y2 / 2, e3 = d3 - 2, f3 = d3 - 4;
    const int g3 = 2 * f3, e4 = g3 - 2, f4 = g3 - 4, g4 = 2 * f4, h5 = g4 - 2, To = g4 - 4;
    NUNIF_REQUIRE(g1 == x1 - 8 && g3 == y2 - 8 && g4 == y1 - 32 && To == (h->up ? 2 * T - 72 : T - 56),
                  "internal: cunet geometry");
    auto sz = [&](int side, int ch) { return b * side * side * ch * sizeof(f16); };
    int rc;
    // live-range packing of the fp16 maps onto 12 buffers
    enum { A = 0, X1, D, E, F, G, H, Y1, X2, U3, P, Q };
    if ((rc = h->t[A].ensure(std::max(sz(a1, 32), sz(a2, 32)))) || (rc = h->t[X1].ensure(sz(x1, 64))) ||
        (rc = h->t[D].ensure(std::max({sz(d1, 64), sz(d2, 64), sz(d3, 128)}))) ||
        (rc = h->t[E].ensure(std::max({sz(e1, 128), sz(e2, 64), sz(e3, 256), sz(e4, 64)}))) ||
        (rc = h->t[F].ensure(std::max({sz(f1, 64), sz(f3, 128), sz(f4, 64)}))) ||
        (rc = h->t[G].ensure(std::max({sz(g1, 64), sz(g3, 128), sz(g4, 64)}))) ||
        (rc = h->t[H].ensure(std::max(sz(h1, 64), sz(h5, 64)))) || (rc = h->t[Y1].ensure(sz(y1, 64))) ||
        (rc = h->t[X2].ensure(sz(y2, 128))) || (rc = h->z1.ensure(b * 3 * T2 * T2 * sizeof(float))) ||
        (rc = h->sums.ensure(b * 128 * 256 * sizeof(float))) || (rc = h->scale.ensure(b * 256 * sizeof(float))))
        return rc;
    f16 *tA = (f16 *)h->t[A].p, *tX1 = (f16 *)h->t[X1].p, *tD = (f16 *)h->t[D].p, *tE = (f16 *)h->t[E].p;
    f16 *tF = (f16 *)h->t[F].p, *tG = (f16 *)h->t[G].p, *tH = (f16 *)h->t[H].p, *tY1 = (f16 *)h->t[Y1].p;
    f16 *tX2 = (f16 *)h->t[X2].p;
    float *z1 = (float *)h->z1.p, *sums = (float *)h->sums.p, *scale = (float *)h->scale.p;

    // ---------------- unet1 (cunet.py:52-67) ----------------
    C3ConvArgs c3;
    memset(&c3, 0, sizeof(c3));
    if (frame) {
        c3.x = frame; c3.frame_mode = 1; c3.H = grid->x_h; c3.W = grid->x_w; c3.wb = grid->w_blocks;
        c3.istep = grid->input_tile_step; c3.pad_t = grid->pad_t; c3.pad_l = grid->pad_l; c3.tile_begin = tile_begin;
    } else {
        c3.x = x;
    }
    c3.B = B; c3.T = T; c3.w = h->u1c1a.w; c3.bias = h->u1c1a.b; c3.C = h->u1c1a.C; c3.out = tA; c3.slope = 0.1f;
    // UNetConv(3, 32, 64) as ONE kernel (conv + LeakyReLU + conv + LeakyReLU, the 32-channel map never leaves LDS): the VALU
    // first conv (0.2 ms per launch at 1.3 TB/s, 27 FMAs per output) and the 32 -> 64 conv's read of its 0.27-GB output are gone
    const bool fused_stem = stem_fused_enabled() && h->u1c1a.frag && h->u2c1a.frag && h->u1c1b.Cin == 32 && h->u1c1b.N == 64;
    if (fused_stem) {
        if ((rc = run_stem(c3, h->u1c1a, h->u1c1b, tX1, s))) return rc;                                                // x1
    } else {
        if ((rc = launch_c3_conv(c3, s))) return rc;
        if ((rc = run_conv(h->u1c1b, tA, nullptr, 0, 0, B, a1, tX1, nullptr, nullptr, 0, 0, 0, 2, s))) return rc;   // x1
    }
    if ((rc = run_down(h->u1down, tX1, B, x1, tD, s))) return rc;
    if ((rc = run_conv(h->u1c2a, tD, nullptr, 0, 0, B, d1, tE, nullptr, nullptr, 0, 0, 0, 2, s))) return rc;
    if ((rc = run_conv(h->u1c2b, tE, nullptr, 0, 0, B, e1, tF, nullptr, nullptr, 0, 0, 0, 2, s))) return rc;
    const bool sef = se_fuse_enabled();
    if ((rc = launch_se(tF, sums, scale, h->u1se2.w1, h->u1se2.b1, h->u1se2.w2, h->u1se2.b2, B, (long)f1 * f1, 64, s, sef))) return rc;
    // conv3(crop(x1, 4) + x2): the add happens in the up-GEMM's epilogue
    if ((rc = run_up(h->u1up, tF, B, f1, tG, s, tX1, x1, 4, sef ? scale : nullptr))) return rc;
    if ((rc = run_conv(h->u1c3, tG, nullptr, 0, 0, B, g1, tH, nullptr, nullptr, 0, 0, 0, 2, s))) return rc;
    // conv_bottom -> z1 (clamped unless no_clip, cunet.py:185-186)
    if (h->up) {
        if ((rc = run_deconv4(h->u1bottom_up, tH, B, h1, z1, h->no_clip ? 1 : 0, s))) return rc;
    } else if ((rc = run_conv(h->u1bottom, tH, nullptr, 0, 0, B, h1, nullptr, z1, nullptr, 0, 0, h->no_clip ? 0 : 1, 0, s)))
        return rc;

    // ---------------- unet2 (cunet.py:99-121) ----------------
    memset(&c3, 0, sizeof(c3));
    c3.x = z1; c3.B = B; c3.T = T2; c3.w = h->u2c1a.w; c3.bias = h->u2c1a.b; c3.C = h->u2c1a.C; c3.out = tA; c3.slope = 0.1f;
    if (fused_stem) {
        if ((rc = run_stem(c3, h->u2c1a, h->u2c1b, tY1, s))) return rc;                                                // x1
    } else {
        if ((rc = launch_c3_conv(c3, s))) return rc;
        if ((rc = run_conv(h->u2c1b, tA, nullptr, 0, 0, B, a2, tY1, nullptr, nullptr, 0, 0, 0, 2, s))) return rc;   // x1
    }
    if ((rc = run_down(h->u2down1, tY1, B, y1, tD, s))) return rc;
    if ((rc = run_conv(h->u2c2a, tD, nullptr, 0, 0, B, d2, tE, nullptr, nullptr, 0, 0, 0, 2, s))) return rc;
    if ((rc = run_conv(h->u2c2b, tE, nullptr, 0, 0, B, e2, tX2, nullptr, nullptr, 0, 0, 0, 2, s))) return rc;       // x2
    if ((rc = launch_se(tX2, sums, scale, h->u2se2.w1, h->u2se2.b1, h->u2se2.w2, h->u2se2.b2, B, (long)y2 * y2, 128, s))) return rc;
    if ((rc = run_down(h->u2down2, tX2, B, y2, tD, s))) return rc;
    if ((rc = run_conv(h->u2c3a, tD, nullptr, 0, 0, B, d3, tE, nullptr, nullptr, 0, 0, 0, 2, s))) return rc;
    if ((rc = run_conv(h->u2c3b, tE, nullptr, 0, 0, B, e3, tF, nullptr, nullptr, 0, 0, 0, 2, s))) return rc;
    if ((rc = launch_se(tF, sums, scale, h->u2se3.w1, h->u2se3.b1, h->u2se3.w2, h->u2se3.b2, B, (long)f3 * f3, 128, s, sef))) return rc;
    // conv4(crop(x2, 4) + x3)
    if ((rc = run_up(h->u2up3, tF, B, f3, tG, s, tX2, y2, 4, sef ? scale : nullptr))) return rc;
    if ((rc = run_conv(h->u2c4a, tG, nullptr, 0, 0, B, g3, tE, nullptr, nullptr, 0, 0, 0, 2, s))) return rc;
    if ((rc = run_conv(h->u2c4b, tE, nullptr, 0, 0, B, e4, tF, nullptr, nullptr, 0, 0, 0, 2, s))) return rc;
    if ((rc = launch_se(tF, sums, scale, h->u2se4.w1, h->u2se4.b1, h->u2se4.w2, h->u2se4.b2, B, (long)f4 * f4, 64, s, sef))) return rc;
    // conv5(crop(x1, 16) + x4)
    if ((rc = run_up(h->u2up4, tF, B, f4, tG, s, tY1, y1, 16, sef ? scale : nullptr))) return rc;
    if ((rc = run_conv(h->u2c5, tG, nullptr, 0, 0, B, g4, tH, nullptr, nullptr, 0, 0, 0, 2, s))) return rc;
    // z = clamp(crop(z1, 20) + conv_bottom(x5), 0, 1)
    if ((rc = run_conv(h->u2bottom, tH, nullptr, 0, 0, B, h5, nullptr, z, z1, T2, 20, 1, 0, s))) return rc;
    return NUNIF_HIP_OK;
}

}  // namespace

extern "C" int nunif_hip_cunet_create(const nunif_tensor_desc *tensors, int32_t n_tensors, int32_t no_clip,
                                      nunif_cunet **handle) {
    NUNIF_REQUIRE(tensors && handle && n_tensors > 0, "cunet_create: NULL argument");
    TMap m;
    for (int i = 0; i < n_tensors; ++i) {
        HostT t;
        t.data = tensors[i].data;
        t.numel = 1;
        for (int d = 0; d < tensors[i].ndim; ++d) { t.shape.push_back(tensors[i].shape[d]); t.numel *= tensors[i].shape[d]; }
        m[tensors[i].name] = t;
    }
    nunif_cunet *h = new nunif_cunet();
    h->no_clip = no_clip;
    int rc = NUNIF_HIP_OK;
    do {
        const HostT *bw;
        if (m.count("net.12.weight")) {                  // nn.Sequential key layout of vgg_7 / upconv_7
            bw = &m["net.12.weight"];
            const bool upconv = bw->shape.size() == 4 && bw->shape[2] == 4;
            h->kind = upconv ? 3 : 2;
            static const int vgg_c[8] = {3, 32, 32, 64, 64, 128, 128, 3}, up_c[7] = {3, 16, 32, 64, 128, 128, 256};
            const int *ch = upconv ? up_c : vgg_c;
            const int first_pad = (ch[1] + 31) / 32 * 32;
            if ((rc = make_c3(h, m, "net.0", ch[1], &h->st_first, first_pad))) break;
            h->st_n = upconv ? 5 : 6;
            for (int i = 0; i < h->st_n && !rc; ++i) {
                const int cin = i == 0 ? first_pad : ch[i + 1];
                rc = make_conv(h, m, "net." + std::to_string(2 * (i + 1)), cin, ch[i + 2], 3, 1, &h->st_conv[i], ch[i + 1]);
            }
            if (rc) break;
            if (upconv && (rc = make_deconv4(h, m, "net.12", 256, 3, &h->st_deconv))) break;
            break;
        }
        if ((rc = find(m, "unet1.conv_bottom.weight", &bw))) break;
        h->up = (bw->shape.size() == 4 && bw->shape[2] == 4) ? 1 : 0;
        h->kind = h->up;
        const std::string a = "unet1.", b = "unet2.";
        if ((rc = make_c3(h, m, a + "conv1.conv.0", 32, &h->u1c1a))) break;
        if ((rc = make_conv(h, m, a + "conv1.conv.2", 32, 64, 3, 1, &h->u1c1b))) break;
        if ((rc = make_conv(h, m, a + "conv1_down", 64, 64, 2, 2, &h->u1down))) break;
        if ((rc = make_conv(h, m, a + "conv2.conv.0", 64, 128, 3, 1, &h->u1c2a))) break;
        if ((rc = make_conv(h, m, a + "conv2.conv.2", 128, 64, 3, 1, &h->u1c2b))) break;
        if ((rc = make_se(h, m, a + "conv2.seblock", 64, &h->u1se2))) break;
        if ((rc = make_up(h, m, a + "conv2_up", 64, 64, &h->u1up))) break;
        if ((rc = make_conv(h, m, a + "conv3", 64, 64, 3, 1, &h->u1c3))) break;
        if (h->up) {
            if ((rc = make_deconv4(h, m, a + "conv_bottom", 64, 3, &h->u1bottom_up))) break;
        } else if ((rc = make_conv(h, m, a + "conv_bottom", 64, 3, 3, 1, &h->u1bottom))) break;
        if ((rc = make_c3(h, m, b + "conv1.conv.0", 32, &h->u2c1a))) break;
        if ((rc = make_conv(h, m, b + "conv1.conv.2", 32, 64, 3, 1, &h->u2c1b))) break;
        if ((rc = make_conv(h, m, b + "conv1_down", 64, 64, 2, 2, &h->u2down1))) break;
        if ((rc = make_conv(h, m, b + "conv2.conv.0", 64, 64, 3, 1, &h->u2c2a))) break;
        if ((rc = make_conv(h, m, b + "conv2.conv.2", 64, 128, 3, 1, &h->u2c2b))) break;
        if ((rc = make_se(h, m, b + "conv2.seblock", 128, &h->u2se2))) break;
        if ((rc = make_conv(h, m, b + "conv2_down", 128, 128, 2, 2, &h->u2down2))) break;
        if ((rc = make_conv(h, m, b + "conv3.conv.0", 128, 256, 3, 1, &h->u2c3a))) break;
        if ((rc = make_conv(h, m, b + "conv3.conv.2", 256, 128, 3, 1, &h->u2c3b))) break;
        if ((rc = make_se(h, m, b + "conv3.seblock", 128, &h->u2se3))) break;
        if ((rc = make_up(h, m, b + "conv3_up", 128, 128, &h->u2up3))) break;
        if ((rc = make_conv(h, m, b + "conv4.conv.0", 128, 64, 3, 1, &h->u2c4a))) break;
        if ((rc = make_conv(h, m, b + "conv4.conv.2", 64, 64, 3, 1, &h->u2c4b))) break;
        if ((rc = make_se(h, m, b + "conv4.seblock", 64, &h->u2se4))) break;
        if ((rc = make_up(h, m, b + "conv4_up", 64, 64, &h->u2up4))) break;
        if ((rc = make_conv(h, m, b + "conv5", 64, 64, 3, 1, &h->u2c5))) break;
        if ((rc = make_conv(h, m, b + "conv_bottom", 64, 3, 3, 1, &h->u2bottom))) break;
    } while (0);
    if (rc) { nunif_hip_cunet_destroy(h); return rc; }
    *handle = h;
    return NUNIF_HIP_OK;
}

extern "C" void nunif_hip_cunet_destroy(nunif_cunet *h) {
    if (!h) return;
    for (void *p : h->owned) (void)hipFree(p);
    for (auto &b : h->t) b.release();
    h->z1.release(); h->sums.release(); h->scale.release(); h->tile_out.release();
    delete h;
}

extern "C" int nunif_hip_cunet_forward(nunif_cunet *h, const float *x, float *z, int32_t batch, int32_t tile_size,
                                       void *stream) {
    NUNIF_REQUIRE(h && x && z && batch > 0, "cunet_forward: bad argument");
    return forward_impl(h, x, nullptr, nullptr, 0, z, batch, tile_size, (hipStream_t)stream);
}

extern "C" int nunif_hip_cunet_render(nunif_cunet *h, const float *x, float *y, int32_t x_h, int32_t x_w,
                                      int32_t tile_size, int32_t batch_size, void *stream) {
    NUNIF_REQUIRE(h && x && y && batch_size > 0, "cunet_render: bad argument");
    nunif_tile_grid g;
    // CUNet: scale 1, offset 28; UpCUNet: scale 2, offset 36 (cunet.py:143,177); vgg_7: 1, 7; upconv_7: 2, 14;
    // blend_size None -> plain overwrite
    static const int kScale[4] = {1, 2, 1, 2}, kOffset[4] = {28, 36, 7, 14};
    int rc = nunif_hip_tile_grid_init(x_h, x_w, kScale[h->kind], kOffset[h->kind], tile_size, 0, &g);
    if (rc) return rc;
    const int n_tiles = g.h_blocks * g.w_blocks;
    const size_t To = g.out_tile_size;
    if ((rc = h->tile_out.ensure((size_t)n_tiles * 3 * To * To * sizeof(float)))) return rc;
    float *tile_out = (float *)h->tile_out.p;
    hipStream_t st = (hipStream_t)stream;
    for (int t0 = 0; t0 < n_tiles; t0 += batch_size) {
        const int nb = std::min(batch_size, n_tiles - t0);
        if ((rc = forward_impl(h, nullptr, x, &g, t0, tile_out + (size_t)t0 * 3 * To * To, nb, tile_size, st))) return rc;
    }
    return launch_stitch(tile_out, y, &g, 3, st);
}
