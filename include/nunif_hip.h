/*
 * nunif_hip.h — C ABI of libnunif_hip.so, the MI355X (gfx950) engine under nunif's tiled-inference hot path.
 *
 * The reference (nagadomi/nunif) is pure Python on PyTorch and has no FFI of its own: the
 * drop-in boundary is a set of Python call signatures.  Each entry point below names the reference function
 * (file:line under the reference tree) whose work it replaces; nunif_amd/ mirrors the Python signatures on top
 * of these calls and INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *  - plain C: pointers + sizes, no torch / C++ types.  `stream` is a hipStream_t passed as void* (NULL = the
 *    null stream).  All device pointers are caller-owned (e.g. torch tensor.data_ptr()); the library owns only
 *    the weights/workspace that live inside a model handle.
 *  - every function returns 0 on success or a negative nunif_hip_status; nunif_hip_last_error() returns a
 *    thread-local message for the last failure.  No function synchronises the device unless it says so.
 *  - image tensors are planar CHW / NCHW float32 in [0,1] at the boundary, exactly like the reference.
 */
#ifndef NUNIF_HIP_H
#define NUNIF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NUNIF_HIP_ABI_VERSION 1

typedef enum {
    NUNIF_HIP_OK = 0,
    NUNIF_HIP_EINVAL = -1,   /* bad argument (the Python layer raises ValueError / AssertionError) */
    NUNIF_HIP_ENOMEM = -2,   /* hipMalloc failed */
    NUNIF_HIP_EHIP = -3,     /* a HIP runtime call or kernel launch failed */
    NUNIF_HIP_EMISSING = -4, /* a required weight tensor is missing from the state dict */
    NUNIF_HIP_EUNSUPPORTED = -5
} nunif_hip_status;

int nunif_hip_abi_version(void);
const char *nunif_hip_last_error(void);

/* ------------------------------------------------------------------------------------------------------------
 * Tile grid + stitcher.  Replaces nunif/utils/seam_blending.py: create_config :109-143, create_blend_filter
 * :146-153, the F.pad + tile slicing of tiled_render :82,:90, update :156-174 and get_output :39-40.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t x_h, x_w, scale, offset, tile_size, blend_size;                 /* inputs */
    int32_t y_h, y_w, h_blocks, w_blocks;                                   /* create_config outputs */
    int32_t pad_l, pad_r, pad_t, pad_b, y_buffer_h, y_buffer_w;
    int32_t input_tile_step, output_tile_step;
    int32_t out_tile_size;                                                  /* tile_size*scale - 2*offset */
} nunif_tile_grid;

/* Host-only integer math; bit-exact with create_config. */
int nunif_hip_tile_grid_init(int32_t x_h, int32_t x_w, int32_t scale, int32_t offset, int32_t tile_size,
                             int32_t blend_size, nunif_tile_grid *grid);

/* Host-only: the 1-D edge ramp r[0..blend_size) (fp32) with filter F[y,x] = min(r'[y], r'[x]);
 * values are the reference's Python-double `1 - (1/(b+1))*(i+1)` rounded to fp32. */
int nunif_hip_blend_ramp(int32_t blend_size, float *ramp /* [blend_size] host */);

/* tiles[k] = replicate-padded x[:, i:i+T, j:j+T] for tile indices tile_begin .. tile_begin+n_tiles (row-major
 * over the grid).  x: [C,x_h,x_w] f32 device; tiles: [n_tiles,C,T,T] f32 device. */
int nunif_hip_gather_tiles(const float *x, float *tiles, const nunif_tile_grid *grid, int32_t channels,
                           int32_t tile_begin, int32_t n_tiles, void *stream);

/* Single-pass stitch of ALL tile outputs of a frame: for each output pixel replay the reference's running-mean
 * update over the (<=4) covering tiles in row-major tile order, crop to [y_h,y_w] and clamp to [0,1].
 * tile_out: [h_blocks*w_blocks, C, To, To] f32 device; y: [C, y_h, y_w] f32 device. */
int nunif_hip_stitch_tiles(const float *tile_out, float *y, const nunif_tile_grid *grid, int32_t channels,
                           void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * waifu2x swin_unet.  Replaces waifu2x/models/swin_unet.py SwinUNetBase.forward :180-199 (+ the eval clamp of
 * the SwinUNet/SwinUNet2x/SwinUNet4x wrappers :221-226,:244-249,:281-290) and torchvision's
 * SwinTransformerBlock (imported at :9-12).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    const char *name;      /* reference state_dict key, e.g. "unet.swin1.block.0.attn.qkv.weight" */
    const float *data;     /* HOST pointer, contiguous fp32 */
    int32_t ndim;
    int64_t shape[4];
} nunif_tensor_desc;

typedef struct nunif_swin_unet nunif_swin_unet;

/* Build a model from a reference-format state dict (host fp32).  Weights are repacked to MFMA-fragment-major
 * fp16 in device memory owned by the handle.  scale_factor in {1,2,4}.  The current HIP device is bound. */
int nunif_hip_swin_unet_create(const nunif_tensor_desc *tensors, int32_t n_tensors, int32_t scale_factor,
                               nunif_swin_unet **handle);
void nunif_hip_swin_unet_destroy(nunif_swin_unet *handle);

/* z = clamp(unet(x), 0, 1).  x: [B,3,T,T] f32 device, z: [B,3,(T-16)*s,(T-16)*s] f32 device; T must satisfy the
 * reference's tile_size_validator (swin_unet.py:202-205).  Workspace grows inside the handle as needed. */
int nunif_hip_swin_unet_forward(nunif_swin_unet *handle, const float *x, float *z, int32_t batch,
                                int32_t tile_size, void *stream);

/* Whole-frame render = SeamBlending.tiled_render (seam_blending.py:48-106) with this model:
 * x: [3,H,W] f32 device -> y: [3,H*s,W*s] f32 device.  The tile gather is fused into the first conv, tiles run
 * in minibatches of `batch_size`, the stitch is the single-pass kernel above. */
int nunif_hip_swin_unet_render(nunif_swin_unet *handle, const float *x, float *y, int32_t x_h, int32_t x_w,
                               int32_t tile_size, int32_t batch_size, void *stream);

/* Tile-ROW sharding of one huge image (the multi-GPU fallback of SURVEY.md §8e; the reference has no counterpart — its
 * SeamBlending.tiled_render, seam_blending.py:48-106, runs every tile on one model): the same render cut into
 *   render_tile_rows  the tiles of tile rows [row_begin, row_end) of the grid into the handle's tile store,
 *   tile_row_band     export (import_band = 0) / import (1) of output rows [row0, row0 + n_rows) of every tile of one tile row,
 *                     band = [w_blocks][3][n_rows][out_tile_size] f32 device — what neighbouring ranks exchange,
 *   stitch_rows       the single-pass stitch of output rows [y_row_begin, y_row_end) into a compact [3][rows][y_w] band.
 * Same tile grid, same tile store layout and the same stitch kernel as nunif_hip_swin_unet_render: bit-identical results. */
int nunif_hip_swin_unet_render_tile_rows(nunif_swin_unet *handle, const float *x, int32_t x_h, int32_t x_w,
                                         int32_t tile_size, int32_t batch_size, int32_t row_begin, int32_t row_end,
                                         void *stream);
int nunif_hip_swin_unet_tile_row_band(nunif_swin_unet *handle, int32_t x_h, int32_t x_w, int32_t tile_size,
                                      int32_t tile_row, int32_t row0, int32_t n_rows, float *band, int32_t import_band,
                                      void *stream);
int nunif_hip_swin_unet_stitch_rows(nunif_swin_unet *handle, float *y_band, int32_t x_h, int32_t x_w, int32_t tile_size,
                                    int32_t y_row_begin, int32_t y_row_end, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * waifu2x CUNet.  Replaces waifu2x/models/cunet.py CUNet.forward :183-196 (UNet1 :52-67, UNet2 :99-121, SEBlock
 * nunif/modules/attention.py:29-44).  Geometry: scale 1, offset 28, no blending (cunet.py:177).
 * ---------------------------------------------------------------------------------------------------------- */
/* waifu2x "waifu2x.swin_unet_v2_1x / _2x / _4x" (aliases winc_unet_*; waifu2x/models/swin_unet_v2.py SwinUNetV2Base :272-352,
 * the model classes :361-466): IR stem, window-MHA + conv-MLP blocks (zero-padded half-window shift, learned score bias),
 * shortcut PatchDown / PatchUp, ToImage + SourceResidual.  State dict in the reference's key layout ("unet." prefix); base_dim
 * (64 / 96 / 128), lv2_ratio and the block counts are read from the tensor shapes.
 * forward: x [B,3,T,T] f32 device -> z [B,3,T*s - 18*s, ...] f32 device (offset 9 s), clamp01 != 0 = eval-mode clamp; T must
 * satisfy the reference's tile_size_validator :355-358 ((T - 16) % 48 == 0). */
typedef struct nunif_swin_unet_v2 nunif_swin_unet_v2;
int nunif_hip_swin_unet_v2_create(const nunif_tensor_desc *tensors, int32_t n_tensors, int32_t scale_factor,
                                  nunif_swin_unet_v2 **handle);
void nunif_hip_swin_unet_v2_destroy(nunif_swin_unet_v2 *handle);
int nunif_hip_swin_unet_v2_forward(nunif_swin_unet_v2 *handle, const float *x, float *z, int32_t batch, int32_t tile_size,
                                   int32_t clamp01, void *stream);

typedef struct nunif_cunet nunif_cunet;
int nunif_hip_cunet_create(const nunif_tensor_desc *tensors, int32_t n_tensors, int32_t no_clip, nunif_cunet **handle);
void nunif_hip_cunet_destroy(nunif_cunet *handle);
/* z = clamp(crop(z1,20) + unet2(z1), 0, 1), z1 = clamp(unet1(x)).  x: [B,3,T,T] f32, z: [B,3,T-56,T-56] f32; T % 4 == 0. */
int nunif_hip_cunet_forward(nunif_cunet *handle, const float *x, float *z, int32_t batch, int32_t tile_size,
                            void *stream);
/* Whole-frame tiled render (x: [3,H,W] -> y: [3,H,W]); tile gather fused into the first conv, overwrite stitch. */
int nunif_hip_cunet_render(nunif_cunet *handle, const float *x, float *y, int32_t x_h, int32_t x_w,
                           int32_t tile_size, int32_t batch_size, void *stream);

/* waifu2x image-side helpers.
 * tta_view: one of the 8 dihedral views of nunif/transforms/tta.py tta_split :20-34; view = rot90*4 + vflip*2 + hflip in
 * the reference's tuple order; x: [C,H,W] -> y: [C,H,W] (views 0-3) or [C,W,H] (views 4-7).
 * tta_merge :37-48: views[k]: the model's output for view k ([C,H,W] / [C,W,H]); out = clamp(sum of the inverse-transformed
 * views * 1/8) added in the reference's order (bit-exact).
 * alpha_border_padding: nunif/utils/alpha.py AlphaBorderPadding :32-57; rgb [3,H,W], alpha [1,H,W], work: 8*H*W floats. */
int nunif_hip_tta_view(const float *x, float *y, int32_t C, int32_t H, int32_t W, int32_t view, void *stream);
int nunif_hip_tta_merge(const float *const *views, float *out, int32_t C, int32_t H, int32_t W, void *stream);
int nunif_hip_alpha_border_padding(const float *rgb, const float *alpha, float *out, float *work, int32_t H, int32_t W,
                                   int32_t offset, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * iw3 stereo synthesis + depth post-processing.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t B, H, W;          /* c: [B,3,H,W], depth: [B,1,H,W] (already at image resolution) */
    double divergence;        /* percent of the base size, as in the reference (doubles: host math is done in double) */
    double convergence;
    int32_t fill;             /* 1 = method "forward_fill" (shift_fill inpaint), 0 = "forward" (clamp) */
    int32_t synthetic_view;   /* 0 both, 1 left, 2 right */
    int32_t width_base;       /* base_size = W if 1 else max(H,W) */
} nunif_forward_warp_params;

/* Replaces iw3/forward_warp.py apply_divergence_forward_warp :246-256 -> depth_order_bilinear_forward_warp
 * :140-243 (inconsistent_shift=False).  left/right: [B,3,H,W]; lmask/rmask: optional [B,1,H,W] (gen_mask2).
 * For a single-eye view the other output pointer is ignored (the caller returns the source image). */
int nunif_hip_forward_warp(const float *c, const float *depth, float *left, float *right, float *lmask,
                           float *rmask, const nunif_forward_warp_params *params, void *stream);

/* Depth-Anything-V2 ViT-S (DINOv2 ViT-S/14 + DPT head): the depth backbone that iw3/depth_anything_model.py:200-230 loads
 * through torch.hub and calls in _forward :113-119.  The network is NOT part of the reference tree; this engine follows
 * the published architecture (state-dict keys of the public checkpoint: pretrained.*, depth_head.*), see
 * oracle/depth_anything_v2.py — parity is against that restatement only.
 * x: [B,3,h,w] f32 device, ImageNet-normalised (batch_preprocess), h and w multiples of 14;
 * pos: [1 + (h/14)*(w/14)][384] f32 device = the position embedding already interpolated to this grid (host side,
 * once per resolution); depth: [B,h,w] f32 device (relu'd inverse depth, larger = nearer). */
typedef struct nunif_depth_anything nunif_depth_anything;
int nunif_hip_depth_anything_create(const nunif_tensor_desc *tensors, int32_t n_tensors, nunif_depth_anything **handle);
/* The same with the two things a checkpoint does not say: taps = the four encoder blocks that feed the DPT head (NULL: the V2
 * layout of the checkpoint's depth — 2, 5, 8, 11 of 12 blocks, 4, 11, 17, 23 of 24; Depth-Anything V1 takes the last four),
 * and max_depth > 0 for the V2 metric heads (Sigmoid x max_depth: 20 hypersim, 80 vkitti) instead of ReLU.  The geometry —
 * ViT-S / B / L (embed 384 / 768 / 1024, 12 / 24 blocks), DPT out_channels and fusion width — is read from the tensors. */
int nunif_hip_depth_anything_create_ex(const nunif_tensor_desc *tensors, int32_t n_tensors, const int32_t *taps, float max_depth,
                                       nunif_depth_anything **handle);
void nunif_hip_depth_anything_destroy(nunif_depth_anything *handle);
int nunif_hip_depth_anything_forward(nunif_depth_anything *handle, const float *x, const float *pos, float *depth,
                                     int32_t B, int32_t h, int32_t w, void *stream);
/* Video-Depth-Anything, streaming (iw3/video_depth_anything_streaming_model.py:58-103: torch.hub
 * "nagadomi/Video-Depth-Anything_iw3" `VideoDepthAnythingStreaming`; `model.infer_video_depth_one(frame)` :94 and
 * `model.reset_state()` :75).  When the tensors given to _create_ex carry `depth_head.motion_modules.{0..3}.*` (the published
 * checkpoint's `head.motion_modules.*`; the caller renames `head.` to `depth_head.`) the engine is TEMPORAL: _forward with B = 1 is
 * `infer_video_depth_one` — each of the 8 temporal attention blocks attends to the previous <= 31 frames through K / V caches the
 * engine owns — and with B > 1 it is B such calls on CONSECUTIVE frames of the stream in one pass (the reference's per-frame loop
 * :92-95 collapsed: encoder, convs and Linears run over the B frames at once, only the temporal attention steps frame by frame);
 * _reset_state starts a new window.  A change of resolution also does.
 * The network is not in the reference tree; oracle/video_depth_anything_net.py restates the published architecture, PARITY UNPINNED. */
int nunif_hip_depth_anything_reset_state(nunif_depth_anything *handle);
int nunif_hip_depth_anything_is_temporal(const nunif_depth_anything *handle);

/* iw3 "iw3.depth_aa" (iw3/models/depth_aa.py :29-95, --depth-aa): depth anti-aliasing net.  x, y: [B,1,h,w] f32 device.
 * mode 0: forward(clamp=False); 1: forward(clamp=True) (eval default); 2: infer() = tensor-wide min-max normalise,
 * forward(clamp=False), de-normalise (what BaseDepthModel.infer(depth_aa=True) calls). */
typedef struct nunif_depth_aa nunif_depth_aa;
int nunif_hip_depth_aa_create(const nunif_tensor_desc *tensors, int32_t n_tensors, nunif_depth_aa **handle);
void nunif_hip_depth_aa_destroy(nunif_depth_aa *handle);
int nunif_hip_depth_aa_forward(nunif_depth_aa *handle, const float *x, float *y, int32_t B, int32_t h, int32_t w,
                               int32_t mode, void *stream);

/* iw3 "sbs.row_flow_v3" (iw3/models/row_flow_v3.py :33-107, the default --method): a small window-attention net that
 * turns the 3-plane feature map [depth | divergence feature | convergence feature] (iw3/backward_warp.py
 * make_input_tensor :17-64) into a horizontal flow `delta` at depth resolution.  create() takes the reference
 * state dict (keys blocks.*, last_layer.1.*).  x: [B,3,h,w] f32 device, delta: [B,1,h,w] f32 device.  flip = 1 runs the
 * net on the horizontally mirrored planes (the right eye of apply_divergence_nn_delta :191-236); delta is then in
 * the mirrored frame, which is what nunif_hip_delta_warp(flip = 1) expects. */
typedef struct nunif_row_flow nunif_row_flow;
int nunif_hip_row_flow_create(const nunif_tensor_desc *tensors, int32_t n_tensors, nunif_row_flow **handle);
void nunif_hip_row_flow_destroy(nunif_row_flow *handle);
int nunif_hip_row_flow_delta(nunif_row_flow *handle, const float *x, float *delta, int32_t B, int32_t h, int32_t w,
                             int32_t flip, void *stream);

/* Replaces iw3/backward_warp.py backward_warp :67-83 for a horizontal flow map: grid = make_grid + [delta, 0] *
 * delta_scale at (dh, dw), bilinear (align_corners) resize to (H, W), grid_sample(bilinear, border, align_corners) +
 * clamp(0,1).  flip = 1: out = flip(backward_warp(flip(c), ...)) without materialising the flips. */
int nunif_hip_delta_warp(const float *c, const float *delta, float *out, int32_t B, int32_t C, int32_t H, int32_t W,
                         int32_t dh, int32_t dw, double delta_scale, int32_t flip, void *stream);

/* iw3 "inpaint.light_inpaint_v1" (iw3/models/light_inpaint_v1.py LightInpaintV1 :53-161), the image inpaint net behind
 * MLBWInpaintImage (iw3/mlbw_inpaint.py:78-157).  create() takes the reference state dict (mask_bias, patch.0, enc1.*, down,
 * enc2.N.*, up, dec1.*, to_image.1).  infer = LightInpaintV1.infer :106-110: preprocess (optional mask_closing, horizontal
 * dilations with FINAL iteration counts, x * (1 - mask), soft mask = clamp(gaussian15(mask) + mask)) + forward with
 * skip_i2i_offset=True.  x, out: [B,3,H,W] f32 in [0,1]; mask: [B,1,H,W] uint8 (0 / 1 hole mask).
 * The same entry points serve "inpaint.light_video_inpaint_v1" (iw3/models/light_video_inpaint_v1.py LightVideoInpaintV1
 * :92-229, base_dim 96 / lv2_mlp_ratio 1; recognised by its `patch.weight` / `to_image.weight` keys): level 2 alternates
 * 8x8-window gMLPs with temporal gMLPs over the 12 frames of each pixel, so infer takes EXACTLY B = 12 consecutive frames
 * (the host pads shorter batches by repeating the first / last frame, :141-150). */
typedef struct nunif_light_inpaint nunif_light_inpaint;
int nunif_hip_light_inpaint_create(const nunif_tensor_desc *tensors, int32_t n_tensors, nunif_light_inpaint **handle);
void nunif_hip_light_inpaint_destroy(nunif_light_inpaint *handle);
int nunif_hip_light_inpaint_infer(nunif_light_inpaint *handle, const float *x, const uint8_t *mask, float *out, int32_t B,
                                  int32_t H, int32_t W, int32_t closing, int32_t inner_iter, int32_t outer_iter, void *stream);
/* mirror_x = 1: out = flip(infer(flip(x), mask)) along the width, with `mask` given in the FLIPPED frame — the nets are trained on
 * the right view, so the reference feeds the left eye mirrored (forward_left: iw3/forward_inpaint.py:29-40, iw3/mlbw_inpaint.py:
 * 60-76 flip -> infer -> flip); here the picture is read and written at column W - 1 - x instead of two flip passes over it. */
int nunif_hip_light_inpaint_infer_ex(nunif_light_inpaint *handle, const float *x, const uint8_t *mask, float *out, int32_t B,
                                     int32_t H, int32_t W, int32_t closing, int32_t inner_iter, int32_t outer_iter,
                                     int32_t mirror_x, void *stream);

/* iw3 output formats.
 * anaglyph: iw3/anaglyph.py apply_anaglyph_redcyan :96-110; left, right, out: [3,H,W] f32; mode 0 color, 1 gray, 2 half-color,
 * 3 wimmer, 4 wimmer2, 5 dubois, 6 dubois2.
 * equirectangular: iw3/equirectangular.py equirectangular_projection :7-40 (VR180); c: [C,h,w] f32 -> out: [C,Hp,Wp] with
 * Hp = h + 2*((S - h)/2), Wp = w + 2*((S - w)/2), S = max(h,w) + max(h,w)/2 (the zero padding is folded into the sampler). */
int nunif_hip_anaglyph(const float *left, const float *right, float *out, int32_t H, int32_t W, int32_t mode, void *stream);
int nunif_hip_equirectangular(const float *c, float *out, int32_t C, int32_t h, int32_t w, void *stream);

/* iw3 "sbs.mlbw" (iw3/models/mlbw.py :37-247; --method mlbw_l2 / mlbw_l4 / mlbw_l2s / mlbw_l4s): multi-layer backward
 * warp.  create() takes the reference state dict (lv1_in.1, lv2.N.*, lv1_out.1); the layer count (2 / 4) and the
 * small / full block stack are read from the tensor shapes.  x: [B,3,h,w] feature planes; delta, weight: [B,L,h,w] f32
 * (weight = softmax over the L layer logits).  flip as in nunif_hip_row_flow_delta. */
typedef struct nunif_mlbw nunif_mlbw;
int nunif_hip_mlbw_create(const nunif_tensor_desc *tensors, int32_t n_tensors, nunif_mlbw **handle);
void nunif_hip_mlbw_destroy(nunif_mlbw *handle);
int32_t nunif_hip_mlbw_num_layers(const nunif_mlbw *handle);
int nunif_hip_mlbw_delta(nunif_mlbw *handle, const float *x, float *delta, float *weight, int32_t B, int32_t h, int32_t w,
                         int32_t flip, void *stream);
/* "sbs.mask_mlbw_l2" (mlbw.py:275-278; MLBW(hole_mask=True): lv1_out has 2L + 1 channels): the extra channel is the hole
 * logit map.  mask_logits: [B,1,h,w] f32, already flipped back to image coordinates when flip != 0
 * (backward_warp.py:325-327); NULL = skip it.  has_hole_mask: 1 when the loaded weights carry the extra channel. */
int32_t nunif_hip_mlbw_has_hole_mask(const nunif_mlbw *handle);
int nunif_hip_mlbw_delta_mask(nunif_mlbw *handle, const float *x, float *delta, float *weight, float *mask_logits, int32_t B,
                              int32_t h, int32_t w, int32_t flip, void *stream);
/* iw3/backward_warp.py postprocess_hole_mask :382-393 with iw3/dilation.py closing :64-71 (n_iter = 1, 3x3),
 * dilate_inner :91-103 / dilate_outer :74-88: closing of the logits at model resolution, bilinear (align_corners) resize to
 * H x W, sigmoid > threshold, then the horizontal OR-dilations.  logits: [B,1,h,w] f32; mask: [B,1,H,W] uint8 (0 / 1);
 * work: 2*B*h*w floats followed by B*H*W bytes.  inner_iter / outer_iter are the FINAL iteration counts (the caller applies the reference's
 * max(round(W / base_width * n), 1) scaling); mask[x] = OR of the thresholded map over [x - outer_iter, x + inner_iter].
 * z (optional, [B,C,H,W] f32): the "hole fill for visualize" of apply_divergence_nn_delta_weight :333-339, z *= 1 - mask. */
int nunif_hip_hole_mask_postprocess(const float *logits, uint8_t *mask, float *work, int32_t B, int32_t h, int32_t w,
                                    int32_t H, int32_t W, float threshold, int32_t inner_iter, int32_t outer_iter, float *z,
                                    int32_t C, void *stream);

/* The composite of iw3/backward_warp.py apply_divergence_nn_delta_weight :300-321: out = clamp(sum_i
 * backward_warp(c, delta_i) * weight_i).  delta: [B,L,dh,dw]; weight: [B,L,H,W] already at image resolution (the
 * reference resizes it with bilinear + antialias: nunif_hip_resize_aa).  L <= 4, C <= 4. */
int nunif_hip_delta_weight_warp(const float *c, const float *delta, const float *weight, float *out, int32_t B, int32_t C,
                                int32_t H, int32_t W, int32_t dh, int32_t dw, int32_t L, double delta_scale, int32_t flip,
                                void *stream);

/* Replaces iw3/backward_warp.py apply_divergence_grid_sample :96-121 (make_grid + backward_warp + grid_sample
 * bilinear/border/align_corners=True + clamp).  c: [B,C,H,W]; depth: [B,1,dh,dw] (grid is built at depth
 * resolution and bilinearly resized, as the reference does). */
int nunif_hip_backward_warp(const float *c, const float *depth, float *left, float *right, int32_t B, int32_t C,
                            int32_t H, int32_t W, int32_t dh, int32_t dw, double divergence, double convergence,
                            int32_t synthetic_view, void *stream);

/* F.interpolate(..., antialias=True) for planar fp32 maps: bilinear (bicubic=0) or bicubic a=-0.5 (bicubic=1),
 * with ATen's coordinate rules (align_corners changes only the scale when antialias is on — SURVEY.md App. C).
 * Optional fused clamp to [0,1] and (x-mean[c])/std[c] over 3-channel groups (iw3/depth_anything_model.py
 * batch_preprocess :103-109; mean3/std3 are HOST pointers or NULL).  tmp: [planes,h_in,w_out] floats. */
int nunif_hip_resize_aa(const float *x, float *y, float *tmp, int64_t planes, int32_t h_in, int32_t w_in,
                        int32_t h_out, int32_t w_out, int32_t bicubic, int32_t align_corners, int32_t clamp01,
                        const float *mean3, const float *std3, void *stream);

/* Replaces iw3/dilation.py dilate_edge :116-142.  x,y: [B,1,H,W]; work: nunif_hip_dilate_edge_work_floats(B,H,W) floats of
 * scratch (a ping-pong image + the per-workgroup range statistics that carry edge_weight's mean / std / min / max from one
 * iteration's kernel to the next: n iterations are n + 1 launches). */
int64_t nunif_hip_dilate_edge_work_floats(int32_t B, int32_t H, int32_t W);
int nunif_hip_dilate_edge(const float *x, float *y, float *work, int32_t B, int32_t H, int32_t W, int32_t n_x,
                          int32_t n_y, void *stream);

/* Per-image (x-min)/(max-min) clamped to [0,1] (iw3/depth_scaler.py minmax_normalize :4-17 with the frame's own
 * min/max, i.e. EMA off).  minmax: [B,2] device scratch that receives the order-keyed min/max. */
int nunif_hip_minmax_normalize(const float *x, float *y, float *minmax, int32_t B, int64_t n_per, void *stream);

/* EMAMinMaxScaler on the device (iw3/depth_scaler.py MinMaxBuffer :33-61, EMAMinMaxScaler.update :95-114, flush :116-133):
 * the reference's 0-dim-tensor arithmetic (amin / amax of the frame, ring insert, ring amin / amax, the EMA recurrence, the
 * `if scale > 0` host sync, (x - lo) / scale, clamp) as four kernels on a small device state block, no host round trip.
 *   nunif_hip_minmax: minmax_keys[B][2] <- order-keyed min / max of every item of x [B, n_per]
 *   nunif_hip_ema_scaler_push: one frame's keys into state = [ring 2N | min_value | max_value]; `count` is
 *     MinMaxBuffer.count before the add, `filled` whether the ring is full after it, `first` whether no EMA value exists yet
 *   nunif_hip_ema_scaler_ring_minmax: state[2N], state[2N+1] <- extrema of the ring (flush before the first EMA value)
 *   nunif_hip_range_normalize: y = clamp((x - lo) / (hi - lo)) (max_mode: clamp(x / hi)) with lohi[2] on the device */
int nunif_hip_minmax(const float *x, float *minmax_keys, int32_t B, int64_t n_per, void *stream);
int nunif_hip_ema_scaler_push(float *state, const float *minmax_keys, int32_t ring_size, int64_t count, int32_t filled,
                              int32_t first, double decay, void *stream);
int nunif_hip_ema_scaler_ring_minmax(float *state, int32_t ring_size, void *stream);
int nunif_hip_range_normalize(const float *x, float *y, const float *lohi, int64_t n, int32_t max_mode, void *stream);

/* iw3/backward_warp.py make_input_tensor :33-64 with c = None for a whole batch: out [B,3,H,W] = depth | divergence plane |
 * convergence plane, with the screen-border taper of `border_pix` columns (0: none). */
int nunif_hip_make_input_planes(const float *depth, float *out, int32_t B, int32_t H, int32_t W, double divergence_value,
                                double convergence_value, int32_t border_pix, void *stream);

/* torch.stack of n <= 16 equally sized device buffers into dst (one launch; the per-frame tensors of a batch). */
int nunif_hip_stack(const void *const *srcs, int32_t n, int64_t bytes_each, void *dst, void *stream);

/* Stand-alone mask morphology on fp32 0/1 masks [B,H,W] (iw3/dilation.py dilate :41-46, erode :49-54, closing :57-64,
 * mask_closing :145-153, dilate_outer :67-81, dilate_inner :84-98).  op: 0 dilate x n_a (3x3 max, window clipped at the border),
 * 1 erode x n_a, 2 closing(n_iter = n_a), 3 mask_closing(n_iter = n_a) = clamp(closing + mask), 4 horizontal OR-dilation with
 * n_a steps of dilate_inner (pixel x takes x+1) and n_b steps of dilate_outer (pixel x takes x-1).  work: B*H*W floats. */
int nunif_hip_mask_morphology(const float *in, float *out, float *work, int32_t B, int32_t H, int32_t W, int32_t op,
                              int32_t n_a, int32_t n_b, void *stream);

/* VideoDepthAnything pre/post glue (iw3/video_depth_anything_model.py:51-91).
 * reflection_pad2d: nunif/modules/reflection_pad2d.py reflection_pad2d_naive :13-48 on planar fp32 [planes,H,W] -> [planes,
 *   H+top+bottom, W+left+right]; positive pads reflect without repeating the edge, negative pads crop (the F.pad(out, (-14,)*4)
 *   of _postprocess :79).
 * depth_postprocess: _postprocess :66-76 — nan_to_num, clamp(max=max_dist) when max_dist > 0, metric depth -> disparity
 *   1/(d+eps) when to_disparity, optional sign flip (the "-out" for non-disparity outputs :88-90).  In place allowed. */
int nunif_hip_reflection_pad2d(const float *x, float *y, int64_t planes, int32_t H, int32_t W, int32_t left, int32_t right,
                               int32_t top, int32_t bottom, void *stream);
int nunif_hip_depth_postprocess(const float *x, float *y, int64_t n, float max_dist, int32_t to_disparity, float eps,
                                int32_t negate, void *stream);

/* Frame edge.  frame: HWC [H,W,3] uint8 (bits=8) or uint16 (bits=16) device memory.
 * frame_to_tensor replaces nunif/utils/video.py to_tensor :218-223 (x.permute(2,0,1) / iinfo.max).
 * stereo_to_frame fuses iw3/utils.py postprocess_image :468-479 (cat + clamp) with video.py from_tensor :236-245
 * ((x*max).round().to(uint)); layout 0 = left|right, 1 = right|left (cross-eyed), 2 = left over right (top-bottom),
 * 3 = the left image alone (right may be NULL): clamp + quantise of a single frame.
 * stereo_compose is the same composition with a planar fp32 result [3,Ho,Wo]. */
int nunif_hip_frame_to_tensor(const void *frame, float *chw, int32_t H, int32_t W, int32_t bits, void *stream);
int nunif_hip_stereo_to_frame(const float *left, const float *right, void *frame, int32_t H, int32_t W,
                              int32_t layout, int32_t bits, void *stream);
int nunif_hip_stereo_compose(const float *left, const float *right, float *out, int32_t H, int32_t W,
                             int32_t layout, void *stream);

/* Pointwise depth -> disparity mappers of iw3/mapper.py :7-118.  kind: 0 identity, 1 pow2, 2 softplus01_legacy(c=p0),
 * 3 softplus01(bias=p0, scale=p1), 4 inv_softplus01(bias=p0, scale=p1), 5 distance_to_disparity(c=p0),
 * 6 shift_relative_depth(min_distance=p0, max_distance=p1). */
int nunif_hip_map_depth(const float *x, float *y, int64_t n, int32_t kind, double p0, double p1, void *stream);

/* Test hooks (tests/ only): snapshot every stage's NHWC fp16 output during the next forward calls, then read
 * them back one by one (returns 1 past the last tap).  Names match oracle.swin_unet.unet_forward(taps=...). */
int nunif_hip_swin_unet_debug_taps(nunif_swin_unet *handle, int32_t enable);
int nunif_hip_swin_unet_get_tap(nunif_swin_unet *handle, int32_t index, char *name, int32_t name_cap,
                                void *host_dst, int64_t cap_bytes, int64_t *nbytes);

/* Timing hooks for bench.py: per-kernel-class HIP-event accumulation on the launch stream. */
int nunif_hip_profile_enable(int32_t on);
/* Writes up to `cap` (name,total_ms,launches) records; returns the count. Synchronises the events. */
typedef struct { char name[48]; double total_ms; int64_t launches; double flops; double bytes; } nunif_prof_record;
int nunif_hip_profile_read(nunif_prof_record *out, int32_t cap, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* NUNIF_HIP_H */
