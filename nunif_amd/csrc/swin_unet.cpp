// Host side of the waifu2x swin_unet engine: weight repacking, workspace, launch sequence, whole-frame render.
//
// Reference: waifu2x/models/swin_unet.py SwinUNetBase.__init__/forward :119-199 (layer inventory and data flow),
// nunif/utils/seam_blending.py tiled_render :48-106 (frame loop).  State-dict keys are the reference's.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "swin_kernels.h"

namespace nunif {

int launch_stitch(const float *tile_out, float *y, const nunif_tile_grid *g, int C, hipStream_t s, int y0 = 0, int rows = -1,
                  int compact = 0);

namespace {

struct HostTensor { const float *data; std::vector<int64_t> shape; int64_t numel; };

struct Linear {          // fragment-packed fp16 weight + fp32 bias, device memory
    f16 *w = nullptr;
    float *bias = nullptr;
    int N = 0, n_real = 0, K = 0;
};

struct Block {
    Linear qkv, proj, mlp0, mlp3;
    // generic path (LayerNorm variants, head counts other than 6: swin_unet_4xl): mlp.0 / mlp.3 in the PLAIN k order for
    // gemm_kernel, LayerNormNoBias weights (nunif/modules/norm.py:17-22)
    Linear mlp0p, mlp3p;
    float *norm1 = nullptr, *norm2 = nullptr;
    bool fast = true;             // fused qkv+attention / register-chained tail kernels apply
    float *attn_bias = nullptr;   // [heads][36][48]
    f16 *tail_stream = nullptr;   // proj | mlp.0 | mlp.3 fragments in proj_mlp_kernel's consumption order
    f16 *tail_ws = nullptr;       // C = 192: per-slice fragments of the weight-stationary tail (swin_block_tail_ws.hip)
    // LDS-resident attention (swin_qkv_attn_r.hip): per head Wq | Wk | Wv fragments, q rows / q bias pre-multiplied by
    // head_dim^-0.5 * log2(e); fp32 bias table [heads][36][52] * log2(e) read as the MFMA C operand, padded keys = -1000
    f16 *qkv_res = nullptr;
    float *qkv_rbias = nullptr;
    float *attn_btab32 = nullptr;
    // C = 96: one kernel per block (swin_block96.hip): proj (chained k) | mlp fragments, compact reversed bias table
    f16 *b96_tail = nullptr;
    float *b96_btab = nullptr;
};

struct DeviceBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return NUNIF_HIP_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        if (hipMalloc(&p, bytes) != hipSuccess) {
            set_error("hipMalloc(%zu) failed", bytes);
            return NUNIF_HIP_ENOMEM;
        }
        cap = bytes;
        return NUNIF_HIP_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace
}  // namespace nunif

using namespace nunif;

struct nunif_swin_unet {
    int scale_factor = 2;
    int C = 96, heads = 6;
    int C1 = 48, C1P = 64;            // stem conv1 channels (real / padded to a multiple of 32)
    int top_dim = 96;                 // channels of level 1 on the decoder side (C for 1x/2x, 2C for 4x)
    float *stem1_w = nullptr, *stem1_b = nullptr;
    Linear stem2, down1, down2, up2, up1, proj2, to_image, to_image_pre;
    Linear down2b;                    // second tap row of down2 when 4 * Cin > 1024 (swin_unet_4xl: K = 1536 as two K = 768 passes)
    bool down2_split = false;
    f16 *to_image_chained = nullptr;  // ToImage weights in the chained k order, for the fused head of the last C = 96 block
    f16 *stemf_w1 = nullptr, *stemf_w2 = nullptr;   // fused stem (swin_stem.hip)
    int dir = 0;                      // direction of the next kernel (snake order); next_dir() flips it
    int att_wm = 1;                   // NUNIF_ATT_WM=0: C = 96 att map pixel-major (8-byte partial-line stores) as in round 2
    int gelu32 = 0;                   // tails with the fp32-polynomial GELU: the 1x net (swin_gelu.h); NUNIF_SWIN_GELU32=0/1 overrides
    int block96 = 0;                  // NUNIF_BLOCK96=1: C = 96 blocks as ONE kernel (swin_block96.hip) instead of attention + tail
    bool has_proj2 = false;
    std::vector<Block> swin[5];
    std::vector<void *> owned;        // every device allocation made at create time
    DeviceBuf s1, f1, f2, f3, g1, qkv, att, hid, tile_out;
    // token -> pixel tables of the window-major C = 96 tail (launch_winmap_build), one per shift, for the geometry last seen
    DeviceBuf winmap[2];
    int winmap_B = 0, winmap_S = 0;
    int device = 0;
    // debug taps (tests only): when on, every stage's fp16 output is snapshotted device-side
    struct Tap { std::string name; void *dev; size_t bytes; };
    bool taps_on = false;
    std::vector<Tap> taps;
    void clear_taps() { for (auto &t : taps) (void)hipFree(t.dev); taps.clear(); }
};

namespace nunif {
namespace {

typedef std::map<std::string, HostTensor> TensorMap;

int find(const TensorMap &m, const std::string &key, const HostTensor **out) {
    auto it = m.find(key);
    if (it == m.end()) {
        set_error("state_dict is missing '%s'", key.c_str());
        return NUNIF_HIP_EMISSING;
    }
    *out = &it->second;
    return NUNIF_HIP_OK;
}

template <typename T>
int upload(nunif_swin_unet *h, const std::vector<T> &host, T **dev) {
    void *p = nullptr;
    if (hipMalloc(&p, host.size() * sizeof(T)) != hipSuccess) {
        set_error("hipMalloc(%zu) failed", host.size() * sizeof(T));
        return NUNIF_HIP_ENOMEM;
    }
    h->owned.push_back(p);
    NUNIF_HIP_CHECK(hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *dev = reinterpret_cast<T *>(p);
    return NUNIF_HIP_OK;
}

// Fragment-major packing for the MFMA A operand (v_mfma_f32_16x16x32_f16): fragment (nt, ks) is 64 lanes x 8 halfs,
// lane l holds W[nt*16 + (l&15)][ks*32 + (l>>4)*8 + 0..7]; rows beyond n_real are zero.  `wt(n,k)` returns fp32.
//
// `chained` packing is for a GEMM whose B operand is not loaded from memory but taken straight from the fp32
// accumulators of the previous GEMM (proj -> mlp.0 -> mlp.3 in proj_mlp_kernel): a lane of a 16x16 accumulator
// tile holds channels 4*(l>>4)+r, so the 8 k-slots of lane group g in K-chunk ks are the channels
// {32ks + 4g + 0..3} (tile 2ks) and {32ks + 16 + 4g + 0..3} (tile 2ks+1).  The reduction order over k is free, so
// the permutation is absorbed here at zero run-time cost.
template <typename F>
std::vector<f16> pack_a_fragments(int n_real, int K, F wt, bool chained) {
    const int N = (n_real + 15) / 16 * 16;
    // + 16 KiB of zeros: the LDS-ring kernels prefetch one 8-KiB chunk past the last fragment they consume
    std::vector<f16> packed((size_t)N * K + 8192, (f16)0.0f);
    const int KS = K / 32;
    for (int nt = 0; nt < N / 16; ++nt)
        for (int ks = 0; ks < KS; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int g = l >> 4;
                    const int n = nt * 16 + (l & 15);
                    const int k = chained ? ks * 32 + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4)) : ks * 32 + g * 8 + j;
                    packed[(((size_t)nt * KS + ks) * 64 + l) * 8 + j] = (f16)(n < n_real ? wt(n, k) : 0.0f);
                }
    return packed;
}

template <typename F>
int make_linear(nunif_swin_unet *h, int n_real, int K, F wt, const float *bias, Linear *L, bool chained = false,
                std::vector<f16> *keep = nullptr) {
    const int N = (n_real + 15) / 16 * 16;
    std::vector<f16> packed = pack_a_fragments(n_real, K, wt, chained);
    std::vector<float> b(N, 0.0f);
    for (int n = 0; n < n_real; ++n) b[n] = bias[n];
    L->N = N; L->n_real = n_real; L->K = K;
    if (keep) *keep = packed;
    int rc = upload(h, packed, &L->w);
    if (rc) return rc;
    return upload(h, b, &L->bias);
}

int make_plain_linear(nunif_swin_unet *h, const TensorMap &m, const std::string &key, int n_real, int K, Linear *L,
                      bool chained = false, std::vector<f16> *keep = nullptr) {
    const HostTensor *w, *b;
    int rc;
    if ((rc = find(m, key + ".weight", &w)) || (rc = find(m, key + ".bias", &b))) return rc;
    NUNIF_REQUIRE(w->numel == (int64_t)n_real * K && b->numel == n_real, "%s: unexpected shape", key.c_str());
    const float *wd = w->data;
    return make_linear(h, n_real, K, [=](int n, int k) { return wd[(size_t)n * K + k]; }, b->data, L, chained, keep);
}

int make_stage(nunif_swin_unet *h, const TensorMap &m, const std::string &key, int dim, int layers,
               std::vector<Block> *blocks) {
    blocks->resize(layers);
    const int heads = h->heads;
    for (int i = 0; i < layers; ++i) {
        const std::string p = key + ".block." + std::to_string(i) + ".";
        Block &bl = (*blocks)[i];
        int rc;
        std::vector<f16> hq;
        if ((rc = make_plain_linear(h, m, p + "attn.qkv", 3 * dim, dim, &bl.qkv, false, &hq))) return rc;
        const bool has_norm = m.find(p + "norm1.weight") != m.end();
        bl.fast = heads == 6 && (dim == 96 || dim == 192) && !has_norm;
        if (has_norm) {
            // LayerNormNoBias (swin_unet_4xl, swin_unet.py:390-391): x = x + attn(norm1(x)); x = x + mlp(norm2(x))
            const HostTensor *n1, *n2;
            if ((rc = find(m, p + "norm1.weight", &n1)) || (rc = find(m, p + "norm2.weight", &n2))) return rc;
            NUNIF_REQUIRE(n1->numel == dim && n2->numel == dim, "%snorm: shape", p.c_str());
            NUNIF_REQUIRE(m.find(p + "norm1.bias") == m.end(), "%snorm1: LayerNorm with bias is not a reference variant", p.c_str());
            std::vector<float> g1(n1->data, n1->data + dim), g2(n2->data, n2->data + dim);
            if ((rc = upload(h, g1, &bl.norm1)) || (rc = upload(h, g2, &bl.norm2))) return rc;
        }
        if (!bl.fast) {
            if ((rc = make_plain_linear(h, m, p + "attn.proj", dim, dim, &bl.proj)) ||
                (rc = make_plain_linear(h, m, p + "mlp.0", 2 * dim, dim, &bl.mlp0p)) ||
                (rc = make_plain_linear(h, m, p + "mlp.3", dim, 2 * dim, &bl.mlp3p)))
                return rc;
        }
        if (bl.fast) {
            const HostTensor *w, *b;
            if ((rc = find(m, p + "attn.qkv.weight", &w)) || (rc = find(m, p + "attn.qkv.bias", &b))) return rc;
            const float qs = (1.0f / sqrtf((float)(dim / heads))) * 1.4426950408889634f;
            const float *wd = w->data;
            std::vector<f16> hs = pack_a_fragments(3 * dim, dim, [=](int n, int k) {
                return wd[(size_t)n * dim + k] * (n < dim ? qs : 1.0f); }, false);
            // per head: Wq tiles, Wk tiles, Wv tiles, each (nt, ks) fragment-major — the order qkv_attn_r_kernel reads them
            const int KS = dim / 32, NTH = (dim / heads) / 16, nf = 3 * (dim / 16) * KS;
            std::vector<f16> stream((size_t)nf * 512, (f16)0.0f);
            size_t fi = 0;
            for (int hh = 0; hh < heads; ++hh)
                for (int part = 0; part < 3; ++part)
                    for (int nt = 0; nt < NTH; ++nt)
                        for (int ks = 0; ks < KS; ++ks) {
                            const size_t frag = (size_t)(part * (dim / 16) + hh * NTH + nt) * KS + ks;
                            std::copy(hs.begin() + frag * 512, hs.begin() + (frag + 1) * 512, stream.begin() + fi * 512);
                            ++fi;
                        }
            if ((rc = upload(h, stream, &bl.qkv_res))) return rc;
            std::vector<float> rb(3 * dim);
            for (int n = 0; n < 3 * dim; ++n) rb[n] = b->data[n] * (n < dim ? qs : 1.0f);
            if ((rc = upload(h, rb, &bl.qkv_rbias))) return rc;
        }
        std::vector<f16> hp, h0, h3;
        if (bl.fast) {
        if ((rc = make_plain_linear(h, m, p + "attn.proj", dim, dim, &bl.proj, false, &hp))) return rc;
        if ((rc = make_plain_linear(h, m, p + "mlp.0", 2 * dim, dim, &bl.mlp0, true, &h0))) return rc;   // chained
        if ((rc = make_plain_linear(h, m, p + "mlp.3", dim, 2 * dim, &bl.mlp3, true, &h3))) return rc;   // chained
        }
        if (bl.fast) {   // one linear stream of 1-KiB fragments in the order proj_mlp_kernel consumes them (swin_block_tail.hip)
            const int KS = dim / 32, NT = dim / 16, SH = 2 * dim / 32;
            const int nf = proj_mlp_stream_frags(dim);
            std::vector<f16> stream((size_t)(nf + 15) / 16 * 16 * 512, (f16)0.0f);     // whole 16-fragment chunks
            size_t fi = 0;
            auto put = [&](const std::vector<f16> &src, int frag) {
                std::copy(src.begin() + (size_t)frag * 512, src.begin() + (size_t)(frag + 1) * 512,
                          stream.begin() + fi * 512);
                ++fi;
            };
            for (int sx = 0; sx < KS; ++sx)
                for (int ks = 0; ks < KS; ++ks) { put(hp, (2 * sx) * KS + ks); put(hp, (2 * sx + 1) * KS + ks); }
            for (int sx = 0; sx < SH; ++sx) {
                for (int ks = 0; ks < KS; ++ks) { put(h0, (2 * sx) * KS + ks); put(h0, (2 * sx + 1) * KS + ks); }
                for (int nt = 0; nt < NT; ++nt) put(h3, nt * SH + sx);
            }
            NUNIF_REQUIRE((int)fi == nf, "internal: tail stream has %zu fragments, expected %d", fi, nf);
            if ((rc = upload(h, stream, &bl.tail_stream))) return rc;
            if (dim == 192) {
                // weight-stationary tail: output-channel slice w of 4 owns proj / mlp.3 tiles 3w..3w+2 and mlp.0 tiles 6w..6w+5
                //   Wp [w][nt 3][ks 6] (plain k order) | W0 [w][nt 6][ks 6] | W3 [w][nt 3][ks 12] (both chained)
                std::vector<f16> ws((size_t)proj_mlp_ws_stream_frags() * 512, (f16)0.0f);
                size_t wi = 0;
                auto putw = [&](const std::vector<f16> &src, int frag) {
                    std::copy(src.begin() + (size_t)frag * 512, src.begin() + (size_t)(frag + 1) * 512, ws.begin() + wi * 512);
                    ++wi;
                };
                for (int w = 0; w < 4; ++w)
                    for (int nt = 0; nt < 3; ++nt)
                        for (int ks = 0; ks < KS; ++ks) putw(hp, (3 * w + nt) * KS + ks);
                for (int w = 0; w < 4; ++w)
                    for (int nt = 0; nt < 6; ++nt)
                        for (int ks = 0; ks < KS; ++ks) putw(h0, (6 * w + nt) * KS + ks);
                for (int w = 0; w < 4; ++w)
                    for (int nt = 0; nt < 3; ++nt)
                        for (int ks = 0; ks < SH; ++ks) putw(h3, (3 * w + nt) * SH + ks);
                NUNIF_REQUIRE((int)wi == proj_mlp_ws_stream_frags(), "internal: ws tail stream has %zu fragments", wi);
                if ((rc = upload(h, ws, &bl.tail_ws))) return rc;
            }
        }
        const HostTensor *tab;
        if ((rc = find(m, p + "attn.relative_position_bias_table", &tab))) return rc;
        NUNIF_REQUIRE(tab->numel == 121 * heads, "%s: bias table shape", p.c_str());
        if (bl.fast && dim == 96) {
            // fused block kernel: attn.proj re-packed in the chained k order (its B operand is the attention output taken
            // straight from the accumulators of head pairs), followed by the mlp part of the tail stream unchanged
            const HostTensor *pw;
            if ((rc = find(m, p + "attn.proj.weight", &pw))) return rc;
            const float *pd = pw->data;
            std::vector<f16> hpc = pack_a_fragments(dim, dim, [=](int n, int k) { return pd[(size_t)n * dim + k]; }, true);
            const int KS = dim / 32, NT = dim / 16, SH = 2 * dim / 32;
            std::vector<f16> stream((size_t)swin_block96_tail_frags() * 512, (f16)0.0f);
            size_t fi = 0;
            auto put = [&](const std::vector<f16> &src, int frag) {
                std::copy(src.begin() + (size_t)frag * 512, src.begin() + (size_t)(frag + 1) * 512, stream.begin() + fi * 512);
                ++fi;
            };
            for (int sx = 0; sx < KS; ++sx)
                for (int ks = 0; ks < KS; ++ks) { put(hpc, (2 * sx) * KS + ks); put(hpc, (2 * sx + 1) * KS + ks); }
            for (int sx = 0; sx < SH; ++sx) {
                for (int ks = 0; ks < KS; ++ks) { put(h0, (2 * sx) * KS + ks); put(h0, (2 * sx + 1) * KS + ks); }
                for (int nt = 0; nt < NT; ++nt) put(h3, nt * SH + sx);
            }
            NUNIF_REQUIRE((int)fi == swin_block96_tail_frags(), "internal: block96 stream has %zu fragments", fi);
            if ((rc = upload(h, stream, &bl.b96_tail))) return rc;
            // R[h][i] = log2(e) * table[120 - i][h]: the keys of a 2 x 2 block are then at +0, +1, +11, +12 from the
            // lane's index (swin_block96.hip); floats 124.. of a head = -1000 (padded keys)
            const int stride = swin_block96_btab_floats() / heads;
            std::vector<float> rt((size_t)heads * stride, -1000.0f);
            for (int hh = 0; hh < heads; ++hh) {
                for (int i2 = 0; i2 < 121; ++i2) rt[(size_t)hh * stride + i2] = tab->data[(size_t)(120 - i2) * heads + hh] * 1.4426950408889634f;
                rt[(size_t)hh * stride + 121] = rt[(size_t)hh * stride + 122] = rt[(size_t)hh * stride + 123] = 0.0f;
            }
            if ((rc = upload(h, rt, &bl.b96_btab))) return rc;
        }
        // bias[h][q][key] = table[(yq-yk+5)*11 + (xq-xk+5)][h]  (torchvision relative_position_index, window 6x6);
        // the 12 padding key columns get -1e30 so that exp() makes them exactly 0.
        std::vector<float> bias((size_t)heads * 36 * 48);
        for (int hh = 0; hh < heads; ++hh)
            for (int q = 0; q < 36; ++q)
                for (int k = 0; k < 48; ++k) {
                    float v = -1.0e30f;
                    if (k < 36) {
                        const int idx = (q / 6 - k / 6 + 5) * 11 + (q % 6 - k % 6 + 5);
                        v = tab->data[(size_t)idx * heads + hh];
                    }
                    bias[((size_t)hh * 36 + q) * 48 + k] = v;
                }
        if ((rc = upload(h, bias, &bl.attn_bias))) return rc;
        // fp32 table read as the MFMA C operand: [heads][36 queries][52], log2(e) * bias.  Key COLUMNS follow the token
        // placement of qkv_attn_r_kernel (win_token, swin_qkv_attn_r.hip): columns 0..31 = keys 0..31, column 32 + 4 g =
        // key 32 + g (register 0 of lane group g in key tile 2); the other columns of tile 2 are padding the kernel never
        // exponentiates (-1000 all the same)
        std::vector<float> btab32((size_t)heads * 36 * 52, 0.0f);
        for (int hh = 0; hh < heads; ++hh)
            for (int q = 0; q < 36; ++q)
                for (int k = 0; k < 48; ++k) {
                    const int key = k < 32 ? k : (((k - 32) & 3) == 0 ? 32 + ((k - 32) >> 2) : -1);
                    btab32[((size_t)hh * 36 + q) * 52 + k] =
                        key >= 0 ? bias[((size_t)hh * 36 + q) * 48 + key] * 1.4426950408889634f : -1000.0f;
                }
        if (kQkvBiasFragMajor) {
            // fragment-major: lane (r16, grp) of the score tile (qt, kt) holds rows 4 grp .. + 3 (keys) of column r16 (query win_token(qt, r16))
            std::vector<float> frag((size_t)heads * kQkvBiasFloatsPerHead, 0.0f);
            for (int hh = 0; hh < heads; ++hh)
                for (int qt = 0; qt < 3; ++qt)
                    for (int kt = 0; kt < 3; ++kt)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int r16 = lane & 15, grp = lane >> 4;
                            const int q = qt < 2 ? 16 * qt + r16 : ((r16 & 3) == 0 ? 32 + (r16 >> 2) : 35);      // win_token
                            for (int j = 0; j < 4; ++j)
                                frag[((((size_t)hh * 3 + qt) * 3 + kt) * 64 + lane) * 4 + j] =
                                    btab32[((size_t)hh * 36 + q) * 52 + 16 * kt + 4 * grp + j];
                        }
            btab32.swap(frag);
        }
        if ((rc = upload(h, btab32, &bl.attn_btab32))) return rc;
    }
    return NUNIF_HIP_OK;
}

int run_gemm(const Linear &L, const f16 *a, int B, int Hi, int Wi, int Cin, int Ho, int Wo, int stride, int oy,
             int ox, int kw, int mode, int act, float slope, const f16 *res, void *out, int ldo, int ps,
             hipStream_t s, const char *tag, int rev = 0) {
    GemmArgs g;
    g.a = a; g.B = B; g.Hi = Hi; g.Wi = Wi; g.Cin = Cin;
    g.Ho = Ho; g.Wo = Wo; g.stride = stride; g.oy = oy; g.ox = ox; g.kw = kw;
    g.K = L.K; g.w = L.w; g.bias = L.bias; g.N = L.N;
    g.mode = mode; g.act = act; g.slope = slope; g.res = res; g.out = out; g.ldo = ldo;
    g.n_real = L.n_real; g.ps = ps; g.oshift = 0; g.OH = 0; g.OW = 0; g.no_clamp = 0; g.lda = 0; g.nt_chunk = 0; g.rev = rev;
    return launch_gemm(g, s, tag);
}

// x = PatchUp(a) + skip, in place over the skip map (swin_unet.py:65-82,189-196): the resident-weight kernel when it takes the
// shape (every launch of that shape, whatever the batch: results must not depend on the tile minibatch), else the generic GEMM
int run_patchup(const Linear &L, const f16 *a, int B, int H, int W, int Cq, f16 *skip, hipStream_t s, const char *tag, int rev) {
    PatchUpArgs p = {a, L.w, L.bias, skip, skip, B, H, W, Cq, rev};
    if (L.K == 192 && L.n_real == 4 * Cq && patchup_supported(p)) return launch_patchup(p, s);
    return run_gemm(L, a, B, H, W, L.K, H, W, 1, 0, 0, 1, 1, 0, 0.f, skip, skip, Cq, 1, s, tag, rev);
}

// PatchDown pass without a residual (swin_unet.py:45-62): a [B, 2 Ho, 2 Wo, Cin] -> out [B, Ho, Wo, 192], K = 384 = the taps of
// input row(s) 2 y + oy ..; the K-outer prefetching kernel when it takes the shape (every launch of that shape), else the generic GEMM
int run_patchdown(const Linear &L, const f16 *a, int B, int Ho, int Wo, int Cin, int oy, f16 *out, hipStream_t s, const char *tag,
                  int rev) {
    PatchDownArgs p;
    p.a = a; p.w = L.w; p.bias = L.bias; p.out = out; p.B = B; p.Ho = Ho; p.Wo = Wo; p.Cin = Cin; p.oy = oy; p.rev = rev;
    if (L.K == 384 && L.n_real == 192 && patchdown_supported(p)) return launch_patchdown(p, s);
    return run_gemm(L, a, B, 2 * Ho, 2 * Wo, Cin, Ho, Wo, 2, oy, 0, 2, 0, 0, 0.f, nullptr, out, L.n_real, 1, s, tag, rev);
}

int run_linear(const Linear &L, const f16 *a, int B, int H, int W, int act, const f16 *res, f16 *out,
               hipStream_t s, const char *tag, int rev = 0) {
    return run_gemm(L, a, B, H, W, L.K, H, W, 1, 0, 0, 1, 0, act, 0.f, res, out, L.n_real, 1, s, tag, rev);
}

int tap(nunif_swin_unet *h, const std::string &name, const void *src, size_t bytes, hipStream_t s) {
    if (!h->taps_on) return NUNIF_HIP_OK;
    void *p = nullptr;
    NUNIF_HIP_CHECK(hipMalloc(&p, bytes));
    NUNIF_HIP_CHECK(hipMemcpyAsync(p, src, bytes, hipMemcpyDeviceToDevice, s));
    h->taps.push_back({name, p, bytes});
    return NUNIF_HIP_OK;
}

// Snake order: consecutive kernels walk their tokens in opposite directions, so each one starts on the lines its
// predecessor touched last (still in the 256-MB memory-side cache) — measured +1.3 % on the 1080p 2x frame with only the
// attention kernels reversed.  Results do not depend on the walk order.
static int next_dir(nunif_swin_unet *h) {
    const int d = h->dir;
    h->dir ^= 1;
    return d;
}

int run_stage(nunif_swin_unet *h, std::vector<Block> &blocks, f16 *x, int B, int S, int dim, hipStream_t s,
              const char *name, const TailToImage *to_image = nullptr) {
    f16 *qkv = (f16 *)h->qkv.p, *att = (f16 *)h->att.p;
    const size_t tok = (size_t)B * S * S;
    int rc;
    for (size_t i = 0; i < blocks.size(); ++i) {
        Block &bl = blocks[i];
        const int shift = (i % 2 == 1) ? 3 : 0;      // swin_unet.py:30
        const std::string tn = std::string(name) + ".b" + std::to_string(i);
        // fast blocks (6 heads, no norm): qkv Linear + attention in one kernel (one window per wave, weights resident in LDS);
        // the 3C-wide qkv map never exists in HBM.  Everything else takes the generic composition below.
        if (!bl.fast) {
            // generic block (swin_unet_4xl: 12 heads, LayerNormNoBias): every Linear on gemm_kernel, attention on the
            // fused-qkv-map kernel, the two LayerNorms as their own pass.  att / hid double as the normalised maps.
            f16 *hid = (f16 *)h->hid.p;
            const f16 *xin = x;
            if (bl.norm1) {
                if ((rc = launch_layernorm_nobias(x, hid, bl.norm1, (long)tok, dim, s))) return rc;
                xin = hid;
            }
            if ((rc = run_linear(bl.qkv, xin, B, S, S, 0, nullptr, qkv, s, "gemm_qkv", next_dir(h)))) return rc;
            if ((rc = tap(h, tn + ".qkv", qkv, tok * 3 * dim * 2, s))) return rc;
            if ((rc = launch_window_attn(qkv, att, bl.attn_bias, B, S, S, h->heads, dim / h->heads, shift, s))) return rc;
            if ((rc = tap(h, tn + ".attn", att, tok * dim * 2, s))) return rc;
            if ((rc = run_linear(bl.proj, att, B, S, S, 0, x, x, s, "gemm_proj", next_dir(h)))) return rc;
            xin = x;
            if (bl.norm2) {
                if ((rc = launch_layernorm_nobias(x, att, bl.norm2, (long)tok, dim, s))) return rc;
                xin = att;
            }
            if ((rc = run_linear(bl.mlp0p, xin, B, S, S, 1, nullptr, hid, s, "gemm_mlp0", next_dir(h)))) return rc;
            if ((rc = run_linear(bl.mlp3p, hid, B, S, S, 0, x, x, s, "gemm_mlp3", next_dir(h)))) return rc;
            if ((rc = tap(h, tn + ".out", x, tok * dim * 2, s))) return rc;
            continue;
        }
        const bool last_blk = i + 1 == blocks.size();
        if (dim == 96 && bl.b96_tail && h->block96 && !h->taps_on) {
            // C = 96: the whole block in one kernel, x updated in place (swin_block96.hip)
            if ((rc = launch_swin_block96(x, bl.qkv_res, bl.qkv_rbias, bl.b96_btab, bl.b96_tail, bl.proj.bias, bl.mlp0.bias,
                                          bl.mlp3.bias, B, S, S, shift, s, last_blk ? to_image : nullptr, next_dir(h))))
                return rc;
            if (last_blk && to_image) break;
            continue;
        }
        // C = 96: att travels in window-major order (full-line stores in the attention kernel, the tail walks its tokens in
        // window order); debug taps want the pixel-major map the oracle has
        // (the tail indexes the window-major map in 32 bits: 96 tok elements; larger launches take the pixel-major map)
        const bool att_wm = dim == 96 && h->att_wm && !h->taps_on && 96L * (long)tok < (1L << 31);
        if ((rc = launch_qkv_attn_r(x, att, bl.qkv_res, bl.qkv_rbias, bl.attn_btab32, B, S, S, dim, h->heads, shift, s, next_dir(h),
                                    att_wm ? 1 : 0)))
            return rc;
        if ((rc = tap(h, tn + ".attn", att, tok * dim * 2, s))) return rc;
        // x = y + mlp(y), y = x + proj(attn): three GEMMs chained through registers, one read + one write of x
        const bool last = i + 1 == blocks.size();
        if (dim == 192) {
            // C = 192: weights stationary in registers / LDS (swin_block_tail_ws.hip)
            if ((rc = launch_proj_mlp_ws(att, x, bl.tail_ws, bl.proj.bias, bl.mlp0.bias, bl.mlp3.bias, (long)tok, s, next_dir(h), h->gelu32)))
                return rc;
        } else {
            const int eff_shift = S <= 6 ? 0 : shift;
            WinMap wmap = {att_wm ? 1 : 0, S, S, eff_shift, nullptr};
            if (att_wm) {
                if (h->winmap_S != S || B > h->winmap_B) {
                    // (re)build both tables on this stream: every launch that reads them is ordered behind it.  The table of a
                    // larger batch contains the smaller one's as its prefix (token n -> batch n / (36 windows per image))
                    for (int k = 0; k < 2; ++k) {
                        if ((rc = h->winmap[k].ensure(tok * sizeof(int)))) return rc;
                        if ((rc = launch_winmap_build((int *)h->winmap[k].p, B, S, S, k ? 3 : 0, s))) return rc;
                    }
                    h->winmap_B = B; h->winmap_S = S;
                }
                wmap.pixmap = (const int *)h->winmap[eff_shift ? 1 : 0].p;
            }
            if ((rc = launch_proj_mlp(att, x, bl.tail_stream, bl.proj.bias, bl.mlp0.bias, bl.mlp3.bias, (long)tok, dim, s,
                                      last ? to_image : nullptr, next_dir(h), &wmap, h->gelu32)))
                return rc;
        }
        if (last && to_image) break;               // x of the last block is not materialised
        if ((rc = tap(h, tn + ".out", x, tok * dim * 2, s))) return rc;
    }
    return NUNIF_HIP_OK;
}

int ensure_workspace(nunif_swin_unet *h, int B, int T) {
    const size_t S = T - 16, S2 = S / 2, S4 = S / 4, C = h->C;
    const size_t t1 = (size_t)B * S * S, t2 = (size_t)B * S2 * S2, t3 = (size_t)B * S4 * S4;
    const size_t top = h->top_dim;
    int rc;
    if ((rc = h->s1.ensure((size_t)B * (T - 14) * (T - 14) * h->C1P * 2))) return rc;
    if ((rc = h->f1.ensure(t1 * C * 2))) return rc;
    if ((rc = h->f2.ensure(t2 * 2 * C * 2))) return rc;
    if ((rc = h->f3.ensure(t3 * 2 * C * 2))) return rc;
    if (h->has_proj2 && (rc = h->g1.ensure(t1 * top * 2))) return rc;
    const size_t qkv_elems = std::max(t1 * 3 * std::max(C, top), t2 * 6 * C);
    if ((rc = h->qkv.ensure(qkv_elems * 2))) return rc;
    if ((rc = h->att.ensure(std::max(t1 * std::max(C, top), t2 * 2 * C) * 2))) return rc;
    if ((rc = h->hid.ensure(std::max(t1 * 2 * std::max(C, top), t2 * 4 * C) * 2))) return rc;
    return NUNIF_HIP_OK;
}

// x: tile mode [B,3,T,T] or (frame != NULL) the frame + grid; z: [B,3,S*s,S*s]
int forward_impl(nunif_swin_unet *h, const float *x, const float *frame, const nunif_tile_grid *grid,
                 int tile_begin, float *z, int B, int T, hipStream_t s) {
    const int S = T - 16, C = h->C;
    NUNIF_REQUIRE(T > 16 && S % 12 == 0 && S % 16 == 0, "tile_size %d is not valid for swin_unet", T);
    int rc;
    if ((rc = ensure_workspace(h, B, T))) return rc;
    f16 *s1 = (f16 *)h->s1.p, *f1 = (f16 *)h->f1.p, *f2 = (f16 *)h->f2.p, *f3 = (f16 *)h->f3.p;

    if (h->stemf_w2 && stem_fused_supported(h->C1, (int)C)) {
        StemFusedArgs sf;
        memset(&sf, 0, sizeof(sf));
        if (frame) {
            sf.x = frame; sf.frame_mode = 1; sf.H = grid->x_h; sf.W = grid->x_w; sf.wb = grid->w_blocks;
            sf.istep = grid->input_tile_step; sf.pad_t = grid->pad_t; sf.pad_l = grid->pad_l; sf.tile_begin = tile_begin;
        } else {
            sf.x = x;
        }
        sf.B = B; sf.T = T; sf.w1 = h->stemf_w1; sf.w2 = h->stemf_w2; sf.b2 = h->stem2.bias; sf.out = f1; sf.slope = 0.1f;
        h->dir = 1;
        if ((rc = launch_stem_fused(sf, s))) return rc;
        goto stem_done;
    }
    Stem1Args a1;
    memset(&a1, 0, sizeof(a1));
    if (frame) {
        a1.x = frame; a1.frame_mode = 1; a1.H = grid->x_h; a1.W = grid->x_w; a1.wb = grid->w_blocks;
        a1.istep = grid->input_tile_step; a1.pad_t = grid->pad_t; a1.pad_l = grid->pad_l; a1.tile_begin = tile_begin;
    } else {
        a1.x = x;
    }
    a1.B = B; a1.T = T; a1.w = h->stem1_w; a1.bias = h->stem1_b; a1.C1 = h->C1; a1.C1P = h->C1P; a1.out = s1;
    a1.slope = 0.1f;
    h->dir = 1;                                    // stem1 walks upwards; everything after it alternates
    if ((rc = launch_stem1(a1, s))) return rc;
    // conv2 3x3 VALID + LeakyReLU(0.1) + crop 6 (swin_unet.py:135-137,182): s1 already starts at conv1 row/col 6
    if ((rc = run_gemm(h->stem2, s1, B, T - 14, T - 14, h->C1P, S, S, 1, 0, 0, 3, 0, 2, 0.1f, nullptr, f1, C, 1, s,
                              "gemm_stem2", next_dir(h))))
        return rc;
stem_done:
    if ((rc = tap(h, "stem", f1, (size_t)B * S * S * C * 2, s))) return rc;
    if ((rc = run_stage(h, h->swin[0], f1, B, S, C, s, "swin1"))) return rc;                          // swin1 -> x3
    if ((rc = run_patchdown(h->down1, f1, B, S / 2, S / 2, C, 0, f2, s, "gemm_down1", next_dir(h)))) return rc;
    if ((rc = tap(h, "down1", f2, (size_t)B * (S / 2) * (S / 2) * 2 * C * 2, s))) return rc;
    if ((rc = run_stage(h, h->swin[1], f2, B, S / 2, 2 * C, s, "swin2"))) return rc;                  // swin2 -> x4
    if (h->down2_split) {
        if ((rc = run_patchdown(h->down2, f2, B, S / 4, S / 4, 2 * C, 0, f3, s, "gemm_down2", next_dir(h)))) return rc;
    } else if ((rc = run_gemm(h->down2, f2, B, S / 2, S / 2, 2 * C, S / 4, S / 4, 2, 0, 0, 2, 0, 0, 0.f, nullptr, f3,
                              2 * C, 1, s, "gemm_down2", next_dir(h))))
        return rc;
    if (h->down2_split && (rc = run_gemm(h->down2b, f2, B, S / 2, S / 2, 2 * C, S / 4, S / 4, 2, 1, 0, 2, 0, 0, 0.f, f3, f3,
                                         2 * C, 1, s, "gemm_down2b", next_dir(h))))
        return rc;
    if ((rc = tap(h, "down2", f3, (size_t)B * (S / 4) * (S / 4) * 2 * C * 2, s))) return rc;
    if ((rc = run_stage(h, h->swin[2], f3, B, S / 4, 2 * C, s, "swin3"))) return rc;                  // swin3
    // x = up2(x5) + x4, written in place over x4 (each lane reads then writes its own 8 bytes)
    if ((rc = run_patchup(h->up2, f3, B, S / 4, S / 4, 2 * C, f2, s, "gemm_up2", next_dir(h)))) return rc;
    if ((rc = tap(h, "up2", f2, (size_t)B * (S / 2) * (S / 2) * 2 * C * 2, s))) return rc;
    if ((rc = run_stage(h, h->swin[3], f2, B, S / 2, 2 * C, s, "swin4"))) return rc;                  // swin4
    f16 *top = f1;
    if (h->has_proj2) {
        // 4x: x = up1(x) + proj2(x3)   (swin_unet.py:166,195-196)
        top = (f16 *)h->g1.p;
        if ((rc = run_linear(h->proj2, f1, B, S, S, 0, nullptr, top, s, "gemm_proj2", next_dir(h)))) return rc;
    }
    if ((rc = run_patchup(h->up1, f2, B, S / 2, S / 2, h->top_dim, top, s, "gemm_up1", next_dir(h)))) return rc;
    if ((rc = tap(h, "up1", top, (size_t)B * S * S * h->top_dim * 2, s))) return rc;
    // to_image + pixel_shuffle + clamp(0,1) (ToImage.forward :110-116, wrapper eval clamp :225-226): fused into the
    // last block's tail kernel when that block runs on the resident C = 96 kernel (1x / 2x nets)
    if (h->top_dim == 96 && h->swin[4].back().fast && h->to_image_chained && !h->taps_on) {
        TailToImage ti;
        ti.w = h->to_image_chained; ti.bias = h->to_image.bias; ti.out = z; ti.H = S; ti.W = S; ti.ps = h->scale_factor;
        ti.n_real = h->to_image.n_real;
        return run_stage(h, h->swin[4], top, B, S, h->top_dim, s, "swin5", &ti);
    }
    if ((rc = run_stage(h, h->swin[4], top, B, S, h->top_dim, s, "swin5"))) return rc;                // swin5
    if (h->scale_factor == 8) {
        f16 *pre = (f16 *)h->att.p;                 // the attention scratch map is free again: [B,S,S,192]
        if ((rc = run_gemm(h->to_image_pre, top, B, S, S, h->top_dim, S, S, 1, 0, 0, 1, 0, 2, 0.2f, nullptr, pre, 192, 1, s,
                           "gemm_to_image_pre", next_dir(h))))
            return rc;
        top = pre;
    }
    if ((rc = run_gemm(h->to_image, top, B, S, S, h->top_dim, S, S, 1, 0, 0, 1, 2, 0, 0.f, nullptr, z, 0,
                       h->scale_factor, s, "gemm_to_image", next_dir(h))))
        return rc;
    return NUNIF_HIP_OK;
}

}  // namespace
}  // namespace nunif

extern "C" int nunif_hip_swin_unet_create(const nunif_tensor_desc *tensors, int32_t n_tensors,
                                          int32_t scale_factor, nunif_swin_unet **handle) {
    NUNIF_REQUIRE(tensors && handle && n_tensors > 0, "swin_unet_create: NULL argument");
    NUNIF_REQUIRE(scale_factor == 1 || scale_factor == 2 || scale_factor == 4 || scale_factor == 8,
                  "swin_unet_create: scale_factor %d unsupported (1, 2, 4, 8)", scale_factor);
    TensorMap m;
    for (int i = 0; i < n_tensors; ++i) {
        HostTensor t;
        t.data = tensors[i].data;
        t.numel = 1;
        for (int d = 0; d < tensors[i].ndim; ++d) {
            t.shape.push_back(tensors[i].shape[d]);
            t.numel *= tensors[i].shape[d];
        }
        m[tensors[i].name] = t;
    }
    nunif_swin_unet *h = new nunif_swin_unet();
    (void)hipGetDevice(&h->device);
    if (const char *v = getenv("NUNIF_BLOCK96")) h->block96 = atoi(v);
    if (const char *v = getenv("NUNIF_ATT_WM")) h->att_wm = atoi(v);
    h->gelu32 = scale_factor == 1;
    if (const char *v = getenv("NUNIF_SWIN_GELU32")) h->gelu32 = atoi(v);
    h->scale_factor = scale_factor;
    const std::string P = "unet.";
    int rc = NUNIF_HIP_OK;
    do {
        const HostTensor *w0, *b0, *w2, *b2;
        if ((rc = find(m, P + "patch.0.weight", &w0)) || (rc = find(m, P + "patch.0.bias", &b0)) ||
            (rc = find(m, P + "patch.2.weight", &w2)) || (rc = find(m, P + "patch.2.bias", &b2)))
            break;
        if (w2->shape.size() != 4 || w0->shape.size() != 4 || w0->shape[1] != 3) {
            set_error("patch conv: unexpected shape"); rc = NUNIF_HIP_EINVAL; break;
        }
        const int C = (int)w2->shape[0];
        const int C1 = (int)w0->shape[0];
        if (C % 96 != 0 || C1 * 2 != C) { set_error("base_dim %d unsupported", C); rc = NUNIF_HIP_EUNSUPPORTED; break; }
        h->C = C; h->heads = C / 16; h->C1 = C1; h->C1P = (C1 + 31) / 32 * 32;
        h->has_proj2 = scale_factor >= 4;
        h->top_dim = h->has_proj2 ? 2 * C : C;
        {
            std::vector<float> w(w0->data, w0->data + w0->numel), b(b0->data, b0->data + b0->numel);
            if ((rc = upload(h, w, &h->stem1_w)) || (rc = upload(h, b, &h->stem1_b))) break;
        }
        {   // conv2 [C][C1][3][3] -> W[n][k = (dy*3+dx)*C1P + ci], zero for the padded input channels
            const float *wd = w2->data;
            const int C1P = h->C1P;
            rc = make_linear(h, C, 9 * C1P, [=](int n, int k) {
                const int tap = k / C1P, ci = k % C1P;
                return ci < C1 ? wd[((size_t)n * C1 + ci) * 9 + tap] : 0.0f;
            }, b2->data, &h->stem2);
            if (rc) break;
        }
        if (stem_fused_supported(C1, C)) {
            const float *w0d = w0->data, *b0d = b0->data, *wd = w2->data;
            std::vector<f16> p1 = pack_a_fragments(C1, 32, [=](int n, int k) {
                return k < 27 ? w0d[(size_t)n * 27 + k] : (k == 27 ? b0d[n] : 0.0f); }, false);
            const int K2 = 448, KS = K2 / 32, NT = C / 16;
            std::vector<f16> p2 = pack_a_fragments(C, K2, [=](int n, int k) {
                const int tap = k / C1, ci = k % C1;
                return k < 9 * C1 ? wd[((size_t)n * C1 + ci) * 9 + tap] : 0.0f; }, false);
            std::vector<f16> stream((size_t)KS * NT * 512, (f16)0.0f);
            for (int ks = 0; ks < KS; ++ks)
                for (int nt = 0; nt < NT; ++nt)
                    std::copy(p2.begin() + ((size_t)nt * KS + ks) * 512, p2.begin() + ((size_t)nt * KS + ks + 1) * 512,
                              stream.begin() + ((size_t)ks * NT + nt) * 512);
            if ((rc = upload(h, p1, &h->stemf_w1)) || (rc = upload(h, stream, &h->stemf_w2))) break;
        }
        if ((rc = make_stage(h, m, P + "swin1", C, 2, &h->swin[0]))) break;
        if ((rc = make_stage(h, m, P + "swin2", 2 * C, 2, &h->swin[1]))) break;
        if ((rc = make_stage(h, m, P + "swin3", 2 * C, 6, &h->swin[2]))) break;
        if ((rc = make_stage(h, m, P + "swin4", 2 * C, 2, &h->swin[3]))) break;
        if ((rc = make_stage(h, m, P + "swin5", h->top_dim, 2, &h->swin[4]))) break;
        auto make_down = [&](const std::string &key, int cin, int cout, Linear *L) -> int {
            const HostTensor *w, *b;
            int r;
            if ((r = find(m, key + ".conv.weight", &w)) || (r = find(m, key + ".conv.bias", &b))) return r;
            NUNIF_REQUIRE(w->numel == (int64_t)cout * cin * 4, "%s: shape", key.c_str());
            const float *wd = w->data;   // [cout][cin][2][2] -> k = (i*2+j)*cin + ci
            return make_linear(h, cout, 4 * cin, [=](int n, int k) {
                const int tap = k / cin, ci = k % cin;
                return wd[((size_t)n * cin + ci) * 4 + tap];
            }, b->data, L);
        };
        auto make_up = [&](const std::string &key, int cin, int cq, Linear *L) -> int {
            const HostTensor *w, *b;
            int r;
            if ((r = find(m, key + ".proj.weight", &w)) || (r = find(m, key + ".proj.bias", &b))) return r;
            NUNIF_REQUIRE(w->numel == (int64_t)4 * cq * cin, "%s: shape", key.c_str());
            const float *wd = w->data;   // rows c*4 + q  ->  q*cq + c   (pixel_shuffle(2) channel order)
            std::vector<float> bb(4 * cq);
            for (int n = 0; n < 4 * cq; ++n) bb[n] = b->data[(n % cq) * 4 + n / cq];
            return make_linear(h, 4 * cq, cin, [=](int n, int k) {
                const int q = n / cq, c = n % cq;
                return wd[((size_t)(c * 4 + q)) * cin + k];
            }, bb.data(), L);
        };
        if ((rc = make_down(P + "down1", C, 2 * C, &h->down1))) break;
        // the base_dim-96 nets' down2 runs the same way (K = 768, one token tile per wave on gemm_kernel<24,1>)
        // as two K = 384 passes on the resident-weight gemm_res_kernel<12,2>
        // (measured on the 2x net: gemm_kernel<24,1> 172 us -> 2 x 114 us - the 200 us of down1 stay = 30 us less per frame)
        {
            // gemm_kernel keeps a token's whole K extent in registers (K <= 1024): the 2x2 stride-2 conv over 2C = 384
            // channels runs as its two tap ROWS, the second accumulating onto the first's output
            const HostTensor *w, *b;
            if ((rc = find(m, P + "down2.conv.weight", &w)) || (rc = find(m, P + "down2.conv.bias", &b))) break;
            const int cin = 2 * C, cout = 2 * C;
            if (w->numel != (int64_t)cout * cin * 4) { set_error("down2: shape"); rc = NUNIF_HIP_EINVAL; break; }
            const float *wd = w->data;
            std::vector<float> zero(cout, 0.0f);
            if ((rc = make_linear(h, cout, 2 * cin, [=](int n, int k) {
                    const int dx = k / cin, ci = k % cin; return wd[((size_t)n * cin + ci) * 4 + dx]; }, b->data, &h->down2)) ||
                (rc = make_linear(h, cout, 2 * cin, [=](int n, int k) {
                    const int dx = k / cin, ci = k % cin; return wd[((size_t)n * cin + ci) * 4 + 2 + dx]; }, zero.data(), &h->down2b)))
                break;
            h->down2_split = true;
        }
        if ((rc = make_up(P + "up2", 2 * C, 2 * C, &h->up2))) break;
        if ((rc = make_up(P + "up1", 2 * C, h->top_dim, &h->up1))) break;
        if (h->has_proj2 && (rc = make_plain_linear(h, m, P + "proj2", 2 * C, C, &h->proj2))) break;
        if (scale_factor == 8) {
            // ToImage of the 8x net (swin_unet.py:96-101): Linear -> LeakyReLU(0.2) -> Linear, both 192 wide
            if ((rc = make_plain_linear(h, m, P + "to_image.proj.0", 192, h->top_dim, &h->to_image_pre)) ||
                (rc = make_plain_linear(h, m, P + "to_image.proj.2", 192, 192, &h->to_image)))
                break;
        } else if ((rc = make_plain_linear(h, m, P + "to_image.proj", 3 * scale_factor * scale_factor, h->top_dim,
                                           &h->to_image)))
            break;
        if (h->top_dim == 96 && 3 * scale_factor * scale_factor <= 16) {
            const HostTensor *tw;
            if ((rc = find(m, P + "to_image.proj.weight", &tw))) break;
            const float *wd = tw->data;
            const int nr = 3 * scale_factor * scale_factor, K = h->top_dim;
            std::vector<f16> ch = pack_a_fragments(nr, K, [=](int n, int k) { return wd[(size_t)n * K + k]; }, true);
            if ((rc = upload(h, ch, &h->to_image_chained))) break;
        }
    } while (0);
    if (rc) {
        nunif_hip_swin_unet_destroy(h);
        return rc;
    }
    *handle = h;
    return NUNIF_HIP_OK;
}

extern "C" int nunif_hip_swin_unet_debug_taps(nunif_swin_unet *h, int32_t enable) {
    NUNIF_REQUIRE(h, "debug_taps: NULL handle");
    h->clear_taps();
    h->taps_on = enable != 0;
    return NUNIF_HIP_OK;
}

extern "C" int nunif_hip_swin_unet_get_tap(nunif_swin_unet *h, int32_t index, char *name, int32_t name_cap,
                                           void *host_dst, int64_t cap_bytes, int64_t *nbytes) {
    NUNIF_REQUIRE(h && nbytes, "get_tap: NULL argument");
    if (index < 0 || index >= (int)h->taps.size()) return 1;   // end of list
    const auto &t = h->taps[index];
    if (name && name_cap > 0) { strncpy(name, t.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    *nbytes = (int64_t)t.bytes;
    if (host_dst) {
        NUNIF_REQUIRE(cap_bytes >= (int64_t)t.bytes, "get_tap: buffer too small");
        NUNIF_HIP_CHECK(hipDeviceSynchronize());
        NUNIF_HIP_CHECK(hipMemcpy(host_dst, t.dev, t.bytes, hipMemcpyDeviceToHost));
    }
    return NUNIF_HIP_OK;
}

extern "C" void nunif_hip_swin_unet_destroy(nunif_swin_unet *h) {
    if (!h) return;
    h->clear_taps();
    for (void *p : h->owned) (void)hipFree(p);
    for (DeviceBuf *b : {&h->s1, &h->f1, &h->f2, &h->f3, &h->g1, &h->qkv, &h->att, &h->hid, &h->tile_out, &h->winmap[0], &h->winmap[1]}) b->release();
    delete h;
}

extern "C" int nunif_hip_swin_unet_forward(nunif_swin_unet *h, const float *x, float *z, int32_t batch,
                                           int32_t tile_size, void *stream) {
    NUNIF_REQUIRE(h && x && z && batch > 0, "swin_unet_forward: bad argument");
    return forward_impl(h, x, nullptr, nullptr, 0, z, batch, tile_size, (hipStream_t)stream);
}

extern "C" int nunif_hip_swin_unet_render(nunif_swin_unet *h, const float *x, float *y, int32_t x_h, int32_t x_w,
                                          int32_t tile_size, int32_t batch_size, void *stream) {
    NUNIF_REQUIRE(h && x && y && batch_size > 0, "swin_unet_render: bad argument");
    // (the reference registers its experimental 8x net with scale = 4 / offset = 64, swin_unet.py:309, which no tile grid
    //  can serve: only the per-tile forward exists for it)
    NUNIF_REQUIRE(h->scale_factor != 8, "swin_unet_render: the 8x net has no consistent tile geometry; use forward");
    const int s = h->scale_factor;
    nunif_tile_grid g;
    int rc = nunif_hip_tile_grid_init(x_h, x_w, s, 8 * s, tile_size, 4 * s, &g);   // offsets/blend: swin_unet.py:213,234,267
    if (rc) return rc;
    const int n_tiles = g.h_blocks * g.w_blocks;
    const size_t To = g.out_tile_size;
    if ((rc = h->tile_out.ensure((size_t)n_tiles * 3 * To * To * sizeof(float)))) return rc;
    float *tile_out = (float *)h->tile_out.p;
    hipStream_t st = (hipStream_t)stream;
    for (int t0 = 0; t0 < n_tiles; t0 += batch_size) {
        const int nb = std::min(batch_size, n_tiles - t0);
        if ((rc = forward_impl(h, nullptr, x, &g, t0, tile_out + (size_t)t0 * 3 * To * To, nb, tile_size, st)))
            return rc;
    }
    return launch_stitch(tile_out, y, &g, 3, st);
}

// ---- tile-ROW sharding of ONE huge image (SURVEY.md §8e fallback): a rank renders the tiles of its tile rows, neighbours
// exchange the (out_tile_size - output_tile_step)-row overlap band of their boundary tile rows, every rank stitches its own
// band of output rows.  The tile store of the handle is addressed like the whole-frame render's ([tile][3][To][To]), so the
// stitch kernel — and therefore the blend recurrence — is the same one: a sharded render is bit-identical to a whole one.
static int rows_grid(nunif_swin_unet *h, int32_t x_h, int32_t x_w, int32_t tile_size, nunif_tile_grid *g) {
    NUNIF_REQUIRE(h->scale_factor != 8, "swin_unet: the 8x net has no consistent tile geometry");
    const int s = h->scale_factor;
    int rc = nunif_hip_tile_grid_init(x_h, x_w, s, 8 * s, tile_size, 4 * s, g);
    if (rc) return rc;
    const size_t To = g->out_tile_size;
    return h->tile_out.ensure((size_t)g->h_blocks * g->w_blocks * 3 * To * To * sizeof(float));
}

extern "C" int nunif_hip_swin_unet_render_tile_rows(nunif_swin_unet *h, const float *x, int32_t x_h, int32_t x_w,
                                                    int32_t tile_size, int32_t batch_size, int32_t row_begin,
                                                    int32_t row_end, void *stream) {
    NUNIF_REQUIRE(h && x && batch_size > 0, "swin_unet_render_tile_rows: bad argument");
    nunif_tile_grid g;
    int rc = rows_grid(h, x_h, x_w, tile_size, &g);
    if (rc) return rc;
    NUNIF_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= g.h_blocks, "tile rows [%d, %d) outside the grid of %d rows",
                  row_begin, row_end, g.h_blocks);
    const size_t To = g.out_tile_size;
    float *tile_out = (float *)h->tile_out.p;
    const int t_end = row_end * g.w_blocks;
    for (int t0 = row_begin * g.w_blocks; t0 < t_end; t0 += batch_size) {
        const int nb = std::min(batch_size, t_end - t0);
        if ((rc = forward_impl(h, nullptr, x, &g, t0, tile_out + (size_t)t0 * 3 * To * To, nb, tile_size, (hipStream_t)stream)))
            return rc;
    }
    return NUNIF_HIP_OK;
}

// band: [w_blocks][3][n_rows][To] fp32 = output rows [row0, row0 + n_rows) of every tile of tile row `tile_row`
extern "C" int nunif_hip_swin_unet_tile_row_band(nunif_swin_unet *h, int32_t x_h, int32_t x_w, int32_t tile_size,
                                                 int32_t tile_row, int32_t row0, int32_t n_rows, float *band,
                                                 int32_t import_band, void *stream) {
    NUNIF_REQUIRE(h && band, "swin_unet_tile_row_band: bad argument");
    nunif_tile_grid g;
    int rc = rows_grid(h, x_h, x_w, tile_size, &g);
    if (rc) return rc;
    const int To = g.out_tile_size;
    NUNIF_REQUIRE(0 <= tile_row && tile_row < g.h_blocks && row0 >= 0 && n_rows > 0 && row0 + n_rows <= To, "tile_row_band: bad window");
    float *store = (float *)h->tile_out.p + (size_t)tile_row * g.w_blocks * 3 * To * To + (size_t)row0 * To;
    // (tile, channel) planes: To * To apart in the store, n_rows * To apart in the band
    const size_t planes = (size_t)g.w_blocks * 3;
    if (import_band)
        NUNIF_HIP_CHECK(hipMemcpy2DAsync(store, (size_t)To * To * 4, band, (size_t)n_rows * To * 4, (size_t)n_rows * To * 4, planes,
                                         hipMemcpyDeviceToDevice, (hipStream_t)stream));
    else
        NUNIF_HIP_CHECK(hipMemcpy2DAsync(band, (size_t)n_rows * To * 4, store, (size_t)To * To * 4, (size_t)n_rows * To * 4, planes,
                                         hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return NUNIF_HIP_OK;
}

extern "C" int nunif_hip_swin_unet_stitch_rows(nunif_swin_unet *h, float *y_band, int32_t x_h, int32_t x_w, int32_t tile_size,
                                               int32_t y_row_begin, int32_t y_row_end, void *stream) {
    NUNIF_REQUIRE(h && y_band, "swin_unet_stitch_rows: bad argument");
    nunif_tile_grid g;
    int rc = rows_grid(h, x_h, x_w, tile_size, &g);
    if (rc) return rc;
    return launch_stitch((const float *)h->tile_out.p, y_band, &g, 3, (hipStream_t)stream, y_row_begin, y_row_end - y_row_begin, 1);
}
