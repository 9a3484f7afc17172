// Launch wrappers of the swin_unet device kernels (swin_kernels.hip) used by the host model (swin_unet.cpp).
#pragma once
#include "common.h"

namespace nunif {

// ---- implicit-GEMM linear / conv:  D[m][n] = act(sum_k X[m][k] W[n][k] + bias[n]) (+ res) -------------------
// X is gathered from an NHWC fp16 map: output token m=(b,y,x) over [B,Ho,Wo]; K = taps*Cin where tap t=(dy,dx),
// dy=t/kw, dx=t%kw reads pixel (y*stride+oy+dy, x*stride+ox+dx).  A plain Linear is taps=1, stride=1.
struct GemmArgs {
    const f16 *a; int B, Hi, Wi, Cin;
    int Ho, Wo, stride, oy, ox, kw;
    int K;                    // taps*Cin, multiple of 32
    const f16 *w;             // packed [N/16][K/32][64 lanes][8] (see pack_weight_frag in swin_unet.cpp)
    const float *bias;        // [N] (padded)
    int N;                    // padded to a multiple of 16
    int mode;                 // 0: NHWC fp16 [M][ldo]   1: 2x2 pixel-shuffle NHWC fp16   2: to_image NCHW f32 clamp
    int act;                  // 0 none, 1 GELU(erf), 2 LeakyReLU(slope)
    float slope;
    const f16 *res;           // optional residual, indexed like out (modes 0,1) — or, with res_W > 0, pixel (y + res_crop, x + res_crop)
                              // of a LARGER map [B, res_H, res_W, ldo] (the cropped U-Net skip of cunet: cunet.py:58-60,111-118)
    void *out;
    int ldo;                  // channels per output pixel (mode 0: N_real, mode 1: N/4)
    int n_real;               // number of valid output columns
    int ps;                   // mode 2: pixel-shuffle factor s (out channels = n_real/(s*s))
    int oshift, OH, OW;       // mode 2: output pixel (y*s+i+oshift, x*s+j+oshift) inside an OH x OW plane (0: Ho*s x Wo*s);
                              //         positions outside the plane are dropped (ConvTranspose2d 4x4 s2 p3 head of UpCUNet)
    int no_clamp;             // mode 2: 1 = no clamp(0,1)
    int lda;                  // element stride between input pixels (0: Cin) — lets a GEMM read a K-slice of wider rows
    int rev;                  // 1: walk the token groups downwards (snake order, swin_unet.cpp next_dir)
    int nt_chunk;             // set by the launcher: output tiles per workgroup column (blockIdx.y) for small-M GEMMs
    int res_H = 0, res_W = 0, res_crop = 0;
    // optional per-IMAGE channel scale of the input (squeeze-excitation, nunif/modules/attention.py:29-44): a value read from
    // channel c of image b is replaced by fp16(x * in_scale[b * Cin + c]) before it is multiplied — the arithmetic of a separate
    // scale pass over the map, without the pass (ring form only)
    const float *in_scale = nullptr;
};
int launch_gemm(const GemmArgs &g, hipStream_t s, const char *tag);

// ---- PatchUp of the swin U-Nets (swin_patchup.hip): out = pixel_shuffle_2(Linear(192 -> 4 Cq)(a)) + res, NHWC fp16 ---------------
// a: [B, Ho, Wo, 192]; w / bias: make_linear's packing with columns n = q Cq + c (q = 2 i + j the sub-pixel); res / out:
// [B, 2 Ho, 2 Wo, Cq] (out may alias res).  Cq = 96 or 192.  Weights resident in LDS, persistent workgroups, the next token
// group's activations and the skip tiles of the next four trips in flight.
struct PatchUpArgs { const f16 *a, *w; const float *bias; const f16 *res; f16 *out; int B, Ho, Wo, Cq, rev; };
bool patchup_supported(const PatchUpArgs &g);
int launch_patchup(const PatchUpArgs &g, hipStream_t s);

// ---- PatchDown of the swin U-Nets (swin_patchdown.hip): out[b, y, x, 192] = bias + sum over the taps of a 2 x 2 stride-2 patch --------
// a: [B, 2 Ho, 2 Wo, Cin]; Cin = 96: K = 384 = the four taps (dy, dx) of a token, k = (2 dy + dx) Cin + c; Cin = 192: K = 384 = the
// two taps of input row 2 y + oy (the other row is a second, accumulating pass on gemm_res_kernel); Cin = 64, N = 64: cunet's
// Conv2d(64, 64, 2, 2) + LeakyReLU.  w / bias: make_linear's packing ([n-tile][k-step]).
struct PatchDownArgs { const f16 *a, *w; const float *bias; f16 *out; int B, Ho, Wo, Cin, oy, rev; int N = 192; int act = 0; float slope = 0.f; };     // act 2: LeakyReLU(slope)
bool patchdown_supported(const PatchDownArgs &g);
int launch_patchdown(const PatchDownArgs &g, hipStream_t s);

// ---- cunet's up step (cunet_up.hip): out[B, 2S, 2S, 64] = LeakyReLU(pixel_shuffle_2(Linear(64 -> 256)(in_scale * a))) + crop(res) ---
// a: [B, S, S, 64]; w / bias: make_up's packing (cunet.cpp: columns n = q 64 + c, [n-tile][k-step]); res: [B, res_S, res_S, 64], the
// skip map, read at (y + crop, x + crop); in_scale: optional [B][64] squeeze-excitation scale of a.
struct CunetUpArgs { const f16 *a, *w; const float *bias; const f16 *res; f16 *out; const float *in_scale; int B, S, res_S, crop;
                     float slope; };
bool cunet_up_supported(const CunetUpArgs &g);
int launch_cunet_up(const CunetUpArgs &g, hipStream_t s);

// ---- output-stationary Linear for TOKEN matrices of a few thousand rows (the ViT encoders of the depth nets) -----------------
// out[m][n] = act(sum_k a[m][k] W[n][k] + bias[n]) (+ res[m][n]); a: [M][lda] fp16, W in gemm_kernel's packing [nt][ks].
// gemm_kernel is token-stationary (a wave keeps 16-32 tokens' whole K extent and sweeps all of N through the LDS ring): with
// K >= 384 that is ONE MFMA per fragment read.  The token Linears of the depth ViTs use one of two other forms (swin_kernels.hip):
// gemm_os_kernel<MT, PF, NB> — output-stationary: a workgroup owns MT x 16 tokens x 128 channels, the activations of a k-group go
// through LDS once for its four waves, the weights are direct loads PF k-steps ahead — and, for K = 384 with N >= 768,
// gemm_ws_kernel — weight-stationary and persistent: 128 channels' weights stay in registers, token tiles stream through an LDS
// ring by LDS-DMA two tiles ahead.  Needs N % 128 == 0 and K % 128 == 0.
struct GemmOsArgs {
    const f16 *a; long M; int lda, K;
    const f16 *w; const float *bias; int N;
    int act;                  // 0 none, 1 GELU(erf)
    const f16 *res;           // optional residual [M][ldo] (may alias out)
    f16 *out; int ldo;
    // LayerNorm without a LayerNorm kernel.  A PRODUCER (stats_out != nullptr) also writes, per token and per 32 output
    // channels, the (sum, sum of squares) of the fp16 values it stored: stats_out[m][N / 32] float2.  A CONSUMER (stats_in != nullptr,
    // weight-stationary launches only: gemm_os_consumes_stats()) multiplies the RAW rows and finishes with
    //   out = r (W x - mu wsum) + bias,   mu / r from the stats_parts partials of its token, wsum[n] = sum_k W[n][k]
    // which is W ((x - mu) r) + bias exactly; the caller folds the affine part of the norm into W / bias / wsum.
    float2 *stats_out;
    const float2 *stats_in; int stats_parts; const float *wsum; float ln_eps;
};
bool gemm_os_supported(long M, int N, int K);
bool gemm_os_consumes_stats(long M, int N, int K);
int launch_gemm_os(const GemmOsArgs &g, hipStream_t s, const char *tag);

// ---- the MLP of a ViT-S block in one kernel (depth_mlp.hip): t += fc2'(gelu(fc1'(norm2(t)))), embed 384, hidden 1536 ---------
// w1 / b1 / ws1: fc1 with norm2 folded in (gemm_kernel packing [n-tile 96][k-step 12], bias, row sums — the GemmOsArgs::stats_in
// convention); w2c: fc2 with LayerScale folded in, CHAINED k order, [n-tile 24][k-step 48]; b2: its bias.  stats_in: the 12
// (sum, sum of squares) partials per token that the producer of t wrote; stats_out (optional): the same for the rows stored here.
struct DaMlpArgs {
    f16 *t; long M;
    const f16 *w1; const float *b1, *ws1;
    const f16 *w2c; const float *b2;
    const float2 *stats_in; float2 *stats_out; float ln_eps;
    // hidden-split form (optional): scratch of da_mlp_partial_bytes(M) and da_mlp_flag_count(M) flags, zeroed once
    void *partial; unsigned *flags;
};
bool da_mlp_supported(int D, int hidden);

// ---- cunet's image heads (cunet_head.hip): out32[b][c][y][x] = bias[c] + sum over the 3 x 3 taps and 64 channels (+ crop(add32), clamp) --------
// a: [B, Hi, Wi, 64] NHWC fp16; w: 4 MFMA A fragments [n-tile 2][k-step 2][64 lanes][8], row n = 3 tap + c (27 of 32), k = ci;
// out32: [B, 3, Ho = Hi - 2, Wo = Wi - 2] fp32; add32: optional [B, 3, addH, addW], read at (y + add_crop, x + add_crop)
struct CunetHeadArgs { const f16 *a, *w; const float *bias; float *out32; const float *add32; int B, Hi, Wi, Ho, Wo, addH, addW, add_crop, clamp01; };
bool cunet_head_supported(const CunetHeadArgs &g);
int launch_cunet_head(const CunetHeadArgs &g, hipStream_t s);

// ---- the temporal modules of Video-Depth-Anything's head, streaming form (depth_temporal.hip) ----------------------------------------
constexpr int kVdaGnBlocks = 256;                 // GroupNorm partial-sum blocks: `part` holds (kVdaGnBlocks + 1) * C float2 (the last row: per-channel coefficients)
int launch_vda_groupnorm(const f16 *x, const float *gamma, const float *beta, f16 *y, float2 *part, int frames, int P, int C, float eps, hipStream_t s);   // x, y: [frames][P][C]; part: frames * (kVdaGnBlocks + 1) * C float2
int launch_vda_layernorm(const f16 *x, const float *gamma, const float *beta, f16 *y, long T, int C, float eps, hipStream_t s);
int launch_vda_geglu(const f16 *h, f16 *out, long T, int I, hipStream_t s);
// qkv: this frame's [P][3 C] rows (q0 | K0 | V0; Wq pre-scaled by hd^-1/2 log2 e); kc / vc: [32][P][C] ring caches, logical window
// position j lives in slot (start + j) & 31; idx = this frame's position (= number of cached frames, <= 31); pq / pk / pv: [32][C]
// fp32 = Wq pe_j (scaled like Wq), Wk pe_j, Wv pe_j; att: [P][C]
struct VdaTattnArgs { const f16 *qkv; f16 *kc, *vc; const float *pq, *pk, *pv; f16 *att; int P, C, hd, start, idx; };
int launch_vda_tattn(const VdaTattnArgs &g, hipStream_t s);
long da_mlp_partial_bytes(long M);
long da_mlp_flag_count(long M);
int launch_da_mlp(const DaMlpArgs &a, hipStream_t s);

// ---- first conv of the stem (3 -> C1 real channels, stored padded to C1P), VALU ----------------------------------
struct Stem1Args {
    const float *x;           // tile mode: [B,3,T,T]; frame mode: [3,H,W]
    int frame_mode, H, W, wb, istep, pad_t, pad_l, tile_begin;   // frame-mode tile origin (seam_blending.py:82,90)
    int B, T;
    const float *w;           // [C1][3][3][3] fp32 (reference layout)
    const float *bias;        // [C1]
    int C1, C1P;
    f16 *out;                 // [B, T-14, T-14, C1P]  (conv1 coordinates [6, T-8), the part conv2+crop consumes)
    float slope;
};
int launch_stem1(const Stem1Args &a, hipStream_t s);

// ---- fused stem (swin_stem.hip): conv1 + LeakyReLU + conv2 + LeakyReLU + crop in one kernel, C1 = 48 / C = 96 ----------
struct StemFusedArgs {
    const float *x;           // tile mode: [B,3,T,T]; frame mode: [3,H,W]  (as Stem1Args)
    int frame_mode, H, W, wb, istep, pad_t, pad_l, tile_begin;
    int B, T;
    const f16 *w1;            // 3 A fragments: rows = conv1 channels, k = ci*9 + ky*3 + kx (27), k = 27: bias, rest 0
    const f16 *w2;            // 14 x 6 A fragments in [k-step][n-tile] order, k = (dy*3 + dx) * 48 + ci, zero beyond 432
    const float *b2;          // [96]
    f16 *out;                 // [B, T-16, T-16, 96]
    float slope;
    int rev;                  // snake order flag
    int C1, C, crop;          // 0 / 0 / 0 = the swin_unet stem (48, 96, crop 6); 32 / 64 / 0 = cunet's UNetConv(3, 32, 64)
};
bool stem_fused_supported(int C1, int C);
int launch_stem_fused(const StemFusedArgs &a, hipStream_t s);

// ---- tail of a swin block: x = y + W3 gelu(W0 y + b0) + b3 with y = x + Wp att + bp, in place on x ------------------
// wstream: proj (plain packed) | mlp.0 | mlp.3 ("chained" packed) fragments in consumption order, padded to a
// multiple of 8 fragments (swin_block_tail.hip; assembled in make_stage, swin_unet.cpp).
int proj_mlp_stream_frags(int C);
// Optional fused image head for the LAST block of the net (C = 96): instead of storing x the kernel applies
// ToImage (Linear C -> 3*ps*ps, pixel_shuffle, clamp(0,1); swin_unet.py:85-116) to the fp16-rounded result and writes
// planar fp32.  w: KS fragments of a 16-row tile in the chained k order; tokens are (b, y, x) over [B, H, W].
struct TailToImage { const f16 *w; const float *bias; float *out; int H, W, ps, n_real; };
// WinMap (C = 96 resident kernel only): tokens are walked in WINDOW order (n = window * 36 + t, the shifted 6x6 windows of
// launch_qkv_attn_r) and att is the window-major map that kernel writes with window_major = 1; x rows / image pixels are
// addressed through the same window -> pixel map.  The per-token arithmetic is unchanged (bit-identical results).
struct WinMap { int on, H, W, shift; const int *pixmap; };     // pixmap: token -> pixel table of launch_winmap_build (B H W ints)
int launch_winmap_build(int *pixmap, int B, int H, int W, int shift, hipStream_t s);
int launch_proj_mlp(const f16 *att, f16 *x, const f16 *wstream, const float *bp, const float *b0, const float *b3,
                    long M, int C, hipStream_t s, const TailToImage *to_image = nullptr, int rev = 0,
                    const WinMap *wm = nullptr, int gelu32 = 0);      // gelu32: the fp32-polynomial GELU (swin_gelu.h gelu8t)

// ---- one kernel per C = 96 swin block: qkv + window attention + proj + MLP, in place on x (swin_block96.hip) -----------
// wqkv / bqkv: the LDS-resident attention's packing (q pre-scaled); btab: swin_block96_btab_floats() floats, per head the
// REVERSED 11 x 11 relative-position table * log2(e) (R[i] = table[120 - i]) then -1000 from float 124 on; wtail:
// swin_block96_tail_frags() fragments = attn.proj in the CHAINED k order, then the mlp part of proj_mlp_kernel's stream.
int swin_block96_tail_frags();
int swin_block96_btab_floats();
int launch_swin_block96(f16 *x, const f16 *wqkv, const float *bqkv, const float *btab, const f16 *wtail, const float *bp,
                        const float *b0, const float *b3, int B, int H, int W, int shift, hipStream_t s,
                        const TailToImage *ti = nullptr, int rev = 0);

// ---- the same tail at C = 192 with the weights stationary on chip (swin_block_tail_ws.hip) --------------------------
// wws: Wp [slice 4][nt 3][ks 6] (plain k order) | W0 [4][nt 6][ks 6] | W3 [4][nt 3][ks 12] (chained k order), 1-KiB fragments
int proj_mlp_ws_stream_frags();
int launch_proj_mlp_ws(const f16 *att, f16 *x, const f16 *wws, const float *bp, const float *b0, const float *b3, long M,
                       hipStream_t s, int rev = 0, int gelu32 = 0);

// ---- fused qkv Linear + (shifted) 6x6 window attention, one window per wave, qkv weights resident in LDS, no barrier in the
// window loop (swin_qkv_attn_r.hip); C = 96 (6 heads of 16) or 192 (6 heads of 32).  x: [B,H,W,C] -> att: [B,H,W,C]
// (pre-projection attention output at the un-rolled positions); wres: per head Wq | Wk | Wv fragments, q pre-scaled;
// btab32: fp32 C-operand table of the score MFMAs, log2(e) * bias, key columns in win_token order, padded keys -1000:
// kQkvBiasFragMajor = 1 (round 6): [heads][query tile 3][key tile 3][64 lanes][4] — the C fragment of lane (r16, grp) for (qt, kt) is one
// lane-linear 16-byte read like the weight fragments (no bank conflicts: the [36][52] row form cost 24 conflict cycles per head and
// window, SQ_LDS_BANK_CONFLICT 10.37 M per launch in profiles/r04b_sq.txt .. r06b_sq.txt, and a per-lane row address per tile);
// 0: [heads][36][52] rows
// window_major = 1 (C = 96): att is written as [window][head][36][16] — every store instruction covers one contiguous
// 512-byte run; launch_proj_mlp's WinMap reads it back in the same order
#ifndef NUNIF_QKV_BTAB_FRAG
#define NUNIF_QKV_BTAB_FRAG 0
#endif
constexpr int kQkvBiasFragMajor = NUNIF_QKV_BTAB_FRAG;
constexpr int kQkvBiasFloatsPerHead = kQkvBiasFragMajor ? 9 * 64 * 4 : 36 * 52;
int launch_qkv_attn_r(const f16 *x, f16 *att, const f16 *wres, const float *bqkv, const float *btab32,
                      int B, int H, int W, int C, int heads, int shift, hipStream_t s, int rev = 0, int window_major = 0);

// ---- (shifted) 6x6 window attention on a fused qkv map ----------------------------------------------------------
// qkv: [B,H,W,3C] fp16 (q | k | v, each heads x hd), out: [B,H,W,C]; bias: [heads][36][48] fp32 with the
// relative-position bias gathered per (q,key) and -1e30 in the 12 padding key columns.
int launch_window_attn(const f16 *qkv, f16 *out, const float *bias, int B, int H, int W, int heads, int hd,
                       int shift, hipStream_t s);

// ---- LayerNormNoBias over the channels of an NHWC fp16 map (swin_unet_4xl blocks; C = 96 / 192 / 384) ----------------------
int launch_layernorm_nobias(const f16 *x, f16 *y, const float *gamma, long M, int C, hipStream_t s);

// ---- CUNet kernels (cunet_kernels.hip) ---------------------------------------------------------------------------------
// K-looped implicit-GEMM conv, NHWC fp16.  Output pixel (y,x), tap (dy,dx) reads a[(y*stride+dy), (x*stride+dx)];
// optional second input a2 is added element-wise at (y*stride+dy+crop2, x*stride+dx+crop2) (the cropped U-Net skip).
struct ConvArgs {
    const f16 *a; const f16 *a2;
    int H2, W2, crop2;
    int B, Hi, Wi, Cin, Ho, Wo, stride, kh, kw;
    const f16 *wstream;       // fragments in [k-step][n-tile] order, k = tap*Cin + ci, zero-padded by 16 KiB
    const float *bias;        // [N]
    int N, n_real;
    int act; float slope;     // 0 none, 2 LeakyReLU
    f16 *out;                 // NHWC [B,Ho,Wo,n_real] (when out32 == NULL)
    float *out32;             // image head: planar fp32 [B,n_real,Ho,Wo]
    const float *add32; int addH, addW, add_crop;   // optional + add32[b][n][y+add_crop][x+add_crop]
    int clamp01;
    int rpad;                 // > 0: ReplicationPad2d(rpad) folded into the gather (tap coordinates clamped; the
                              //      caller passes Ho = Hi + 2*rpad - k + 1); stride must be 1, no second input
    const f16 *res;           // optional residual added AFTER the activation, NHWC [B,Ho,Wo,n_real]
    int zpad;                 // > 0: zero padding `zpad` (Conv2d(padding=zpad)): taps outside the map read 0; any stride;
                              //      the caller passes Ho = (Hi + 2*zpad - k) / stride + 1
    int relu_in;              // 1: ReLU applied to the input as it is loaded (pre-activation residual units)
    const f16 *res2;          // optional second residual (same indexing as res)
    int ldo;                  // > 0: channel stride of the NHWC output / residual pixels (default n_real): lets a conv write a
                              //      channel slice of a wider map (out pre-offset by the slice's first channel)
    int cmaj;                 // 64 / 32: wstream is packed CHUNK-MAJOR ([cmaj-channel chunk][tap][32-channel part][n-tile]) and the
                              //     contraction of a 3x3 stride-1 conv with Cin > 128 is split over workgroups (conv3_lds.hip);
                              //     0: tap-major (k = tap * Cin + ci)
    float *part32;            // cmaj: scratch for the fp32 partials, (Cin / cmaj) * B * Ho * Wo * N floats
};
long gemm_big_m();          // swin_kernels.hip: M from which plain K = 192 Linears run on the resident-weight GEMM
int launch_conv(const ConvArgs &g, hipStream_t s);
// 3x3 stride-1 same conv with the input tile staged in LDS (conv3_lds.hip); launch_conv takes it when it applies
bool conv3_lds_applies(const ConvArgs &g);
int launch_conv3_lds(const ConvArgs &g, hipStream_t s);
// the same conv, persistent, with halo and weights moved into LDS by LDS-DMA (conv3_dma.hip): launches of more than 512 patches,
// Cin 32 / 64, Cout 32 / 64, one input, NHWC fp16 output
bool conv3_dma_applies(const ConvArgs &g);
int launch_conv3_dma(const ConvArgs &g, hipStream_t s);

struct C3ConvArgs {
    const float *x; int frame_mode, H, W, wb, istep, pad_t, pad_l, tile_begin;
    int B, T;                 // tile mode input [B,3,T,T]; output [B,T-2,T-2,C]
    const float *w, *bias;    // [C][3][3][3], [C]
    int C;
    f16 *out;
    float slope;
};
int launch_c3_conv(const C3ConvArgs &a, hipStream_t s);

// x[b,:,:,c] *= sigmoid(W2 relu(W1 mean_hw(x[b]) + b1) + b2)   (nunif/modules/attention.py SEBlock :29-44)
int launch_se(f16 *x, float *sums, float *scale, const float *w1, const float *b1, const float *w2, const float *b2,
              int B, long hw, int C, hipStream_t s, int scale_in_consumer = 0);

}  // namespace nunif
