// Depth-Anything (V1 / V2 / V2-metric; DINOv2 ViT-S / B / L /14 encoder + DPT head) on gfx950 — the depth backbone behind
// BaseDepthModel.infer (iw3/depth_anything_model.py:113-119,200-230 loads it through torch.hub; the network itself is NOT in the
// reference tree).  Parity: oracle/depth_anything_v2.py restates the published architecture and is pinned against HuggingFace
// transformers' DepthAnythingForDepthEstimation (tests/test_depth_anything_vs_hf.py); the GPU tests read HF-produced fixtures
// (tests/golden/depth_anything_hf.npz).
//
// Geometry is read from the checkpoint (embed 384 / 768 / 1024 = 6 / 12 / 16 heads of 64, 12 / 24 blocks, DPT out_channels and
// fusion width from the head's own tensors); which four blocks feed the head and the metric head's max_depth are arguments.
// Layout: tokens [B][1 + gh*gw][D] fp16 (class token first), DPT maps NHWC fp16.  GEMM-shaped work reuses the engine's
// kernels: the token Linears on gemm_os_kernel / gemm_ws_kernel (swin_kernels.hip: output-stationary, and weight-stationary +
// persistent for K = 384 with N >= 768), gemm_kernel for the patch embedding, the 1x1 convs and the k = stride ConvTranspose2d
// resize layers (pixel-shuffle GEMMs), conv3_lds_kernel / conv_kernel for every 3x3 of the head (zero padding, pre-activation
// ReLU and up to two residuals fused).  LayerScale is folded into proj / fc2 on the host, the softmax scale * log2(e) into Wq.
// ViT-S: norm1 (from the second block on) and norm2 have no kernel — proj / fc2 write per-token partial sums of what they
// store, qkv / fc1 multiply the raw rows and finish with r (W x - mu wsum) + b (GemmOsArgs::stats_out / stats_in).
// Round 4: the ViT-S MLP (fc1 + GELU + fc2 + residual + the next norm1's statistics) is ONE kernel over a pair of workgroups per
// 64 tokens (depth_mlp.hip); the 3x3 convs of the head with 32 / 64 / 128 input channels run on conv3_dma_kernel (persistent,
// halo + weights by LDS-DMA, 16-byte-run epilogue); a fusion block's 1x1 out_conv runs before the resize it commutes with; the
// reassemble branch of every tap but the last runs on a side stream beside the encoder (forward()).
// New kernels here:
//   da_layernorm_kernel   one wave per token, fp32 statistics (the first norm1, the four tap norms; every norm of ViT-B / L)
//   da_attn_kernel<WAVES> global softmax attention over 1 + gh*gw tokens, heads of 64: one wave per 16 queries, 8 or 12 query tiles
//                         per workgroup sharing K / V through a three-slot LDS ring fetched two steps ahead, 32 keys per step,
//                         online softmax whose stabiliser moves only as an overflow guard; S^T = K Q^T so that exp2(S^T) is directly the P^T operand of
//                         O^T = V^T P^T; V is read in its natural layout and transposed on the way into LDS; the key -> MFMA-row
//                         permutation that makes P^T's k-slots contiguous keys is free because the K fragment is a row gather
//   da_upsample_kernel    bilinear, align_corners=True, NHWC
//   im2col / assemble / final 1x1 + ReLU
#include <algorithm>
#include <type_traits>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "swin_kernels.h"

namespace nunif {

#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

constexpr int kHd = 64, kPatch = 14, kKp = 608;   // 3*14*14 = 588 padded to 19 k-steps

__global__ void __launch_bounds__(256) da_im2col_kernel(const float *__restrict__ x, f16 *__restrict__ a, int B, int h,
                                                        int w, int gh, int gw) {
    const long total = (long)B * gh * gw * kKp;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int k = (int)(id % kKp);
    const long m = id / kKp;
    float v = 0.f;
    if (k < 3 * kPatch * kPatch) {
        const int gx = (int)(m % gw);
        const long t = m / gw;
        const int gy = (int)(t % gh), b = (int)(t / gh);
        const int ci = k / (kPatch * kPatch), r = k - ci * kPatch * kPatch, ky = r / kPatch, kx = r - ky * kPatch;
        v = x[(((long)b * 3 + ci) * h + gy * kPatch + ky) * w + gx * kPatch + kx];
    }
    a[id] = (f16)v;
}

// t[b][0] = cls + pos[0];  t[b][1+m] = patch[b][m] + pos[1+m]
__global__ void __launch_bounds__(256) da_assemble_kernel(const f16 *__restrict__ pe, const float *__restrict__ cls,
                                                          const float *__restrict__ pos, f16 *__restrict__ t, int B, int Np,
                                                          int kD) {
    const long total = (long)B * Np * kD;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int c = (int)(id % kD);
    const long r = id / kD;
    const int n = (int)(r % Np), b = (int)(r / Np);
    const float v = n == 0 ? cls[c] : (float)pe[((long)b * (Np - 1) + n - 1) * kD + c];
    t[id] = (f16)(v + pos[(long)n * kD + c]);
}

template <int NP>                                   // D = 64 * NP channels, one wave per token
__global__ void __launch_bounds__(256) da_layernorm_kernel(const f16 *__restrict__ x, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, f16 *__restrict__ y, long T) {
    constexpr int kD = 64 * NP;
    const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= T) return;
    const int lane = threadIdx.x & 63;
    float v[NP];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) { v[i] = (float)x[tok * kD + lane + 64 * i]; s += v[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / kD);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) { const float d = v[i] - mean; q += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q * (1.0f / kD) + 1e-6f);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c = lane + 64 * i;
        y[tok * kD + c] = (f16)((v[i] - mean) * rstd * gamma[c] + beta[c]);
    }
}

// (Rounds 1-2 went through two earlier forms, both deleted: one wave per 16-query tile reading K / V straight from L2 (59.7 us
//  per launch for B = 2 at 392 x 686: 1 032 waves, two dependent load latencies per key step) and a split-key form that gave one
//  query tile to a workgroup of 4 waves (-4.5 % on the iw3 frame).  Both pulled 8 KiB of K / V per wave and key step through the
//  vector memory pipe for 8 MFMAs: that pipe, not the matrix pipe, was the bound — 155 TFLOP/s.)
// K / V shared through LDS: a workgroup of 8 waves = 8 query tiles (128 queries) walks the keys together; the 8 operand
// fragments of a 32-key step (K rows in MFMA row order for both key tiles x 2 k-steps, V^T for the 4 channel tiles) are loaded
// ONCE per workgroup — one 16-byte load per thread — into an 8-KiB LDS slot, and every wave reads them from there.
// grid (ceil(Np / 128), heads, B).
//
// Round 3, second form.  The first LDS form fetched step i + 1 while computing step i behind a __syncthreads(), which drains
// vmcnt: one exposed L2 / MALL latency per 32 keys — 1.0 us per step for ~0.3 us of issue work.  Now the slots are a ring of
// THREE and the fetch runs two steps ahead in two register sets (the loop is unrolled by two so that they are static): the
// load of step i + 3 is issued in front of step i's MFMAs and lands in LDS behind step i + 1's; the barrier is
// `s_waitcnt lgkmcnt(0); s_barrier` — it must not wait for the loads in flight.
// V is read in its natural [key][channel] layout (128 contiguous bytes per key and head) and transposed on the way INTO LDS
// (8 ds_write_b16 per staging thread and step), which removed the separate V^T kernel and its 2-byte gathers (11 us per layer);
// the 16 lanes of a V fragment are stored at r16 ^ 2f so that the 8 channels x 4 fragments of one write fall into different banks.
// (QT query tiles per wave were measured on the first form, ms per 12 launches at B = 4: <1, 8 waves> 0.52, <2, 4> 0.69, <2, 8>
//  0.58 — with about one workgroup per CU, fewer waves or fewer workgroups cost more than the halved LDS reads give.)
template <int WAVES>                                // query tiles (= waves) per workgroup; waves 0..7 stage K / V for all of them
__global__ void __launch_bounds__(WAVES * 64) da_attn_kernel(const f16 *__restrict__ qkv, f16 *__restrict__ att, int Np, int kD,
                                                             int kHeads) {
    __shared__ __attribute__((aligned(16))) f16x8 kv[3][8][64];
    // the wave index as an SGPR: the K / V staging arms become scalar branches, so that both carry their vmcnt wait on every path
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r16 = lane & 15, grp = lane >> 4;
    // 1-D grid, renumbered so that the query blocks of one (frame, head) — which all walk the same K / V — sit on one XCD's L2
    const unsigned bid = xcd_contiguous(blockIdx.x, gridDim.x);
    const int qblocks = (Np + 16 * WAVES - 1) / (16 * WAVES);
    const int qt0 = (int)(bid % qblocks) * WAVES + wave;
    const bool has_q = qt0 * 16 < Np;
    const int hh = (int)(bid / qblocks) % kHeads, b = (int)(bid / qblocks) / kHeads;
    const f16 *base = qkv + (long)b * Np * (3 * kD);
    const int steps = (Np + 31) >> 5;
    f16x8 qf[2];
    {
        const int q = min(qt0 * 16 + r16, Np - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[ks] = *reinterpret_cast<const f16x8 *>(base + (long)q * (3 * kD) + hh * kHd + 32 * ks + 8 * grp);
    }
    // staging piece of this thread.  Waves 0..3: K fragment f = wave (key tile f >> 1, k-step f & 1); MFMA row i of key tile 0 / 1
    // <-> key k0 + 8*(i>>2) + (i&3) [+ 4], so that a lane's 8 P^T slots are keys k0 + 8g + 0..7.  Waves 4..7: 8 keys x 128 B of V,
    // lane = 8 * (key & 7) + channel block.
    const bool is_k = wave < 4;
    const int key_off = is_k ? 8 * (r16 >> 2) + (r16 & 3) + (wave >= 2 ? 4 : 0) : 8 * (wave - 4) + (lane >> 3);
    const f16 *src = base + (is_k ? kD + hh * kHd + 32 * (wave & 1) + 8 * grp : 2 * kD + hh * kHd + 8 * (lane & 7));
    auto stage = [&](int step) -> f16x8 {
        const int key = min(step * 32 + key_off, Np - 1);              // clamped keys are masked (K) / meet P = 0 (V)
        return *reinterpret_cast<const f16x8 *>(src + (long)key * (3 * kD));
    };
    // V element i of this thread: channel 8c + i = 16 f + r16v -> fragment f = c >> 1, lane (r16v, grp = wave - 4), slot e = key & 7
    const int vc = lane & 7;
    f16 *vdst = reinterpret_cast<f16 *>(&kv[0][4 + (vc >> 1)][(wave & 3) * 16]) + (lane >> 3);
    const int vr0 = 8 * (vc & 1), vx = 2 * (vc >> 1);
    auto publish = [&](int slot, f16x8 v) {
        if (WAVES > 8 && wave >= 8) return;
        if (is_k) {
            kv[slot][wave][lane] = v;
        } else {
            f16 *d = vdst + slot * (8 * 64 * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) d[((vr0 + i) ^ vx) * 8] = v[i];
        }
    };
    f32x4 o[4];
    float m_run = -1.0e30f, l_run = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int vrd = grp * 16;                               // reader side of the V swizzle
    auto compute = [&](int step, int slot) {
        const int k0 = step * 32;
        f16x8 kf[4], vf[4];                        // V fragments are read HERE: their LDS latency hides beneath the softmax
#pragma unroll
        for (int f = 0; f < 4; ++f) kf[f] = kv[slot][f][lane];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vf[dt] = kv[slot][4 + dt][vrd + (r16 ^ (2 * dt))];
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            s0 = MFMA_16x16x32(kf[ks], qf[ks], s0);
            s1 = MFMA_16x16x32(kf[2 + ks], qf[ks], s1);
        }
        // accumulator row 4g+r of tile 0 is key k0 + 8g + r, of tile 1 key k0 + 8g + 4 + r
        if (k0 + 32 > Np) {                      // only the last step has keys beyond the sequence (wave-uniform branch)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (k0 + 8 * grp + r >= Np) s0[r] = -1.0e30f;
                if (k0 + 8 * grp + 4 + r >= Np) s1[r] = -1.0e30f;
            }
        }
        float mx = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
        float m_new = m_run;
        // The stabiliser only has to keep exp2(s - m) inside fp16, it need not be the maximum: m_run is left alone until some logit
        // of the tile exceeds it by more than 8 (P <= 256), and only then is the column maximum looked up across the four lane groups
        // (which must share one m) and the running sums rescaled — after the first key steps: never.  Against a maximum taken on
        // every step (row_group_max + compare: ~10 of the step's 83 VALU instructions) 0.433 -> 0.389 ms per 12 launches, same box.
        if (__any(mx > m_run + 8.0f)) {
            m_new = fmaxf(m_run, row_group_max(mx));
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){o[dt][0] * alpha, o[dt][1] * alpha, o[dt][2] * alpha, o[dt][3] * alpha};
            l_run *= alpha;
        }
        float p[8], sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            p[r] = __builtin_amdgcn_exp2f(s0[r] - m_new);
            p[4 + r] = __builtin_amdgcn_exp2f(s1[r] - m_new);
            sum += p[r] + p[4 + r];
        }
        // the denominator stays a per-LANE partial (alpha is the same in all four lane groups of a column): one
        // cross-group reduction after the key loop instead of two shuffles through the LDS crossbar per 32-key step
        l_run += sum;
        m_run = m_new;
        const f16x8 pf = {(f16)p[0], (f16)p[1], (f16)p[2], (f16)p[3], (f16)p[4], (f16)p[5], (f16)p[6], (f16)p[7]};
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = MFMA_16x16x32(vf[dt], pf, o[dt]);
    };
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // one third of the unrolled loop: computes `step` from `slot`; st_old holds step + 2 (fetched one trip earlier), st_new takes step + 3.
    // Round 6: the SLOT is a compile-time constant (the loop is unrolled by 6 = lcm(3 slots, 2 register sets)): with a run-time slot
    // hipcc rebuilt the LDS address of each of the 8 fragment reads per step (13 v_lshl_add_u32 + 4 v_add_u32 + 3 v_lshl_or_b32 of the
    // step's ~83 VALU instructions in a kernel that is 3 : 1 VALU-bound); now every read is `base VGPR + immediate`.
    auto third_step = [&](int step, auto slot_c, f16x8 &st_new, const f16x8 &st_old) {
        constexpr int slot = decltype(slot_c)::value;
        st_new = stage(step + 3);                 // unconditional (keys are clamped): a conditional load makes hipcc wait vmcnt(0)
        if (has_q) compute(step, slot);
        // slot (slot + 2) % 3; unconditional as well (behind the last steps it rewrites a slot nobody reads again)
        publish(slot >= 1 ? slot - 1 : 2, st_old);
        lds_barrier();
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    f16x8 sa, sb;
    publish(0, stage(0));
    if (steps > 1) publish(1, stage(1));
    sb = stage(2);
    lds_barrier();
    // `steps` is uniform over the workgroup, so every wave leaves at the same barrier count; leaving abandons the loads in flight
#pragma unroll 1
    for (int step = 0; step < steps; step += 6) {
        third_step(step, S0{}, sa, sb);
        if (step + 1 >= steps) break;
        third_step(step + 1, S1{}, sb, sa);
        if (step + 2 >= steps) break;
        third_step(step + 2, S2{}, sa, sb);
        if (step + 3 >= steps) break;
        third_step(step + 3, S0{}, sb, sa);
        if (step + 4 >= steps) break;
        third_step(step + 4, S1{}, sa, sb);
        if (step + 5 >= steps) break;
        third_step(step + 5, S2{}, sb, sa);
    }
    float l = l_run;
    l += __shfl_xor(l, 16);                      // all lanes take part (the exchange partners share r16, not the branch below)
    l += __shfl_xor(l, 32);
    const int qrow = qt0 * 16 + r16;
    if (has_q && qrow < Np) {
        const float inv = 1.0f / l;
        f16 *dst = att + ((long)b * Np + qrow) * kD + hh * kHd + 4 * grp;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<f16x4 *>(dst + dt * 16) = (f16x4){(f16)(o[dt][0] * inv), (f16)(o[dt][1] * inv),
                                                                (f16)(o[dt][2] * inv), (f16)(o[dt][3] * inv)};
    }
}

// resize_layers.3 (3x3, stride 2, padding 1, C -> C on the patch grid: 364 output pixels per 4 x 1080p batch, K = 9 C up to 9 216):
// as a conv it is a chain of 108+ k-steps for a handful of workgroups (160 us); as im2col + the output-stationary Linear the
// same contraction is spread over M / 32 x C / 128 workgroups.  a: [B,H,W,C] -> col: [B*Ho*Wo][9*C], k = tap * C + c
__global__ void __launch_bounds__(256) da_im2col_s2_kernel(const f16 *__restrict__ a, f16 *__restrict__ col, int B, int H, int W,
                                                           int C, int Ho, int Wo) {
    const int c8 = C / 8;
    const long total = (long)B * Ho * Wo * 9 * c8;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int sg = (int)(id % c8);
    long t = id / c8;
    const int tap = (int)(t % 9); t /= 9;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho), b = (int)(t / Ho);
    const int yy = 2 * oy + tap / 3 - 1, xx = 2 * ox + tap % 3 - 1;
    f16x8 v = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = *reinterpret_cast<const f16x8 *>(a + (((long)b * H + yy) * W + xx) * C + sg * 8);
    *reinterpret_cast<f16x8 *>(col + (((long)b * Ho + oy) * Wo + ox) * 9 * C + (long)tap * C + sg * 8) = v;
}

// F.interpolate(bilinear, align_corners=True) on NHWC fp16; one thread = 8 channels of one output pixel
// add (optional, laid out like y): y = fp16(interp + add) — the refinenet's `path + RCU1(skip)` when RCU1 ran beside the encoder
__global__ void __launch_bounds__(256) da_upsample_kernel(const f16 *__restrict__ x, f16 *__restrict__ y, int B, int Hi,
                                                          int Wi, int Ho, int Wo, int C, const f16 *__restrict__ add) {
    const int cq = C / 8;
    const long total = (long)B * Ho * Wo * cq;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int c8 = (int)(id % cq);
    long t = id / cq;
    const int X = (int)(t % Wo); t /= Wo;
    const int Y = (int)(t % Ho), b = (int)(t / Ho);
    const float ry = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, rx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    const float sy = ry * (float)Y, sx = rx * (float)X;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0), x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    auto ld = [&](int yy, int xx) { return *reinterpret_cast<const f16x8 *>(x + (((long)b * Hi + yy) * Wi + xx) * C + c8 * 8); };
    const f16x8 a = ld(y0, x0), bq = ld(y0, x1), c = ld(y1, x0), d = ld(y1, x1);
    f16x8 o;
    const long oidx = (((long)b * Ho + Y) * Wo + X) * C + c8 * 8;
    if (add) {
        const f16x8 e = *reinterpret_cast<const f16x8 *>(add + oidx);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            o[j] = (f16)(hy * (hx * (float)a[j] + lx * (float)bq[j]) + ly * (hx * (float)c[j] + lx * (float)d[j]) + (float)e[j]);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            o[j] = (f16)(hy * (hx * (float)a[j] + lx * (float)bq[j]) + ly * (hx * (float)c[j] + lx * (float)d[j]));
    }
    *reinterpret_cast<f16x8 *>(y + oidx) = o;
}

// relu(conv1x1 32 -> 1) (+ the model's final relu, idempotent) -> fp32 [B,h,w]; metric heads (max_depth > 0) end in a
// Sigmoid instead and the model scales by max_depth (Depth-Anything-V2 metric_depth dpt.py)
__global__ void __launch_bounds__(256) da_final_kernel(const f16 *__restrict__ x, const float *__restrict__ w, float *__restrict__ y,
                                                       long n, float max_depth) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= n) return;
    float acc = w[32];
    const f16 *p = x + id * 32;
#pragma unroll
    for (int k = 0; k < 32; ++k) acc = fmaf((float)p[k], w[k], acc);
    y[id] = max_depth > 0.f ? max_depth / (1.0f + __expf(-acc)) : fmaxf(acc, 0.f);
}

}  // namespace nunif

using namespace nunif;

// =====================================================================================================================
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
struct Lin { f16 *w = nullptr; float *b = nullptr; int N = 0, K = 0; float *ws = nullptr; };     // gemm_kernel packing [nt][ks]; ws: row sums (LayerNorm-folded Linears)
struct Cnv { f16 *w = nullptr; float *b = nullptr; int N = 0, Cin = 0, k = 3, cmaj = 0; };     // conv_kernel stream [ks][nt]
struct Blk { float *g1, *b1, *g2, *b2; Lin qkv, proj, fc1, fc2[4], fc2_full, qkv_ln, fc1_ln; int n_fc2; f16 *fc2c = nullptr; };   // fc2c: fc2_full in the chained k order (depth_mlp.hip)   // *_ln: norm1 / norm2 folded in (ViT-S)   // fc2 (K = 4 D) = 2 or 4 GEMMs of K = 768 / 1024
struct Rcu { Cnv c1, c2; };
struct Fus { Rcu r1, r2; Lin out; };
// Video-Depth-Anything's temporal modules (depth_temporal.hip): a temporal attention block = LayerNorm + [Wq | Wk | Wv] (no bias, Wq
// scaled by hd^-1/2 log2 e) + to_out, the three position tables, the K0 / V0 ring caches of the window; a module = GroupNorm,
// proj_in, two attention blocks, GEGLU feed-forward (ff1: C -> 8 C, ff2: 4 C -> C), proj_out
struct TAtt { Lin qkv, out; float *g = nullptr, *b = nullptr, *pq = nullptr, *pk = nullptr, *pv = nullptr; Buf kc, vc; };
struct TMod { int C = 0; float *gn_g = nullptr, *gn_b = nullptr, *ff_g = nullptr, *ff_b = nullptr; Lin proj_in, proj_out, ff1, ff2; TAtt at[2]; };
constexpr int kTLen = 32;                // temporal_max_len: the attention window (this frame + 31 cached)
}  // namespace

namespace {
// The hidden-split MLP kernel (depth_mlp.hip `da_mlp_split_kernel`) spin-waits on its partner workgroup: both must be resident.
// One such grid at a time always is (2 groups <= CUs, in-order dispatch); two of them on different streams could interleave on
// the XCD dispatchers into blocks that all wait for partners the other grid keeps out.  So the split form is LEASED to one stream
// at a time: a forward() on another stream takes it only once the previous holder's forward has completed on the device
// (event query), otherwise it runs the one-block-per-tile form.  Same stream = stream order = never two grids at once.
// (Another PROCESS on the same GPU is not covered: the kernel's bounded spin traps instead of hanging; NUNIF_DA_MLP_SPLIT=0.)
struct SplitLeaseState {
    std::mutex mu;
    hipStream_t owner = nullptr;
    hipEvent_t done = nullptr;
    bool has_owner = false, enqueueing = false, recorded = false;
};
SplitLeaseState g_split;
struct SplitLease {
    hipStream_t s;
    bool held = false;
    explicit SplitLease(hipStream_t stream) : s(stream) {
        std::lock_guard<std::mutex> lk(g_split.mu);
        bool free_now = !g_split.has_owner;
        if (!free_now && !g_split.enqueueing) {
            if (g_split.owner == s) free_now = true;
            else free_now = !g_split.recorded || hipEventQuery(g_split.done) == hipSuccess;
        }
        if (!free_now) return;
        if (!g_split.done && hipEventCreateWithFlags(&g_split.done, hipEventDisableTiming) != hipSuccess) { g_split.done = nullptr; return; }
        g_split.owner = s; g_split.has_owner = true; g_split.enqueueing = true;
        held = true;
    }
    ~SplitLease() {
        if (!held) return;
        std::lock_guard<std::mutex> lk(g_split.mu);
        g_split.recorded = hipEventRecord(g_split.done, s) == hipSuccess;
        if (!g_split.recorded) g_split.has_owner = false;            // nothing to wait for: the next caller re-evaluates
        g_split.enqueueing = false;
    }
};
}  // namespace

struct nunif_depth_anything {
    std::vector<void *> owned;
    int D = 384, heads = 6, depth = 12, taps[4] = {2, 5, 8, 11};
    int OC[4] = {48, 96, 192, 384}, OCP[4] = {64, 96, 192, 384}, feat_ch = 64;
    float max_depth = 0.f;                 // > 0: metric head (Sigmoid * max_depth)
    Lin patch; float *cls = nullptr, *norm_g = nullptr, *norm_b = nullptr;
    std::vector<Blk> blk;
    Lin proj[4], rs0, rs1, rs3g; std::vector<Cnv> rs3; Cnv rn[4]; Fus fus[4]; Cnv oc1, oc2; float *w_final = nullptr;
    Buf a_col, pe, t, y, qkv, att, hid, lnstats, mlp_flags, feat[4], rnb[4], m1, m2, m3, m4, m5, part, col;
    // the reassemble branch of tap k (project -> resize -> layer_rn conv) on a stream of its own, beside the encoder layers that
    // follow the tap: per-branch temporaries, fork / join events (forward())
    Buf bm1[4], bm2[4], bpart[4];
    Buf bt1[3], br1[3];               // RCU1 of refinenets 1-3 beside the encoder: conv1's map, RCU1(skip) (forward())
    hipStream_t side[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork[3] = {nullptr, nullptr, nullptr}, ev_join[3] = {nullptr, nullptr, nullptr};
    // Video-Depth-Anything (streaming): four temporal modules (on layer_3, layer_4, path_4, path_3) and the state of the window —
    // t_len cached frames (<= 31) whose K0 / V0 sit in ring slots t_start .. of every attention block's caches; t_P: the pixel
    // counts the caches were laid out for (a resolution change starts a new window)
    bool temporal = false;
    TMod tm[4];
    int t_len = 0, t_start = 0, t_P[4] = {0, 0, 0, 0};
    Buf ta, th, tqkv, tatt, thid, tgg, tpart;
};

namespace {
template <typename T>
int upload(nunif_depth_anything *h, const std::vector<T> &host, T **dev) {
    void *p = nullptr;
    if (hipMalloc(&p, host.size() * sizeof(T)) != hipSuccess) { set_error("hipMalloc failed"); return NUNIF_HIP_ENOMEM; }
    h->owned.push_back(p);
    NUNIF_HIP_CHECK(hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *dev = reinterpret_cast<T *>(p);
    return NUNIF_HIP_OK;
}
int up_f32(nunif_depth_anything *h, const HostT *t, float **dev) {
    std::vector<float> v(t->data, t->data + t->numel);
    return upload(h, v, dev);
}
template <typename F>
void put_frag(std::vector<f16> &dst, size_t frag, int nt, int ks, F wt) {
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j)
            dst[(frag * 64 + l) * 8 + j] = (f16)wt(nt * 16 + (l & 15), ks * 32 + (l >> 4) * 8 + j);
}
// Linear / 1x1 / pixel-shuffle GEMM: W'(n, k) given by wt (zero outside the real range), N and K already padded
template <typename F, typename G>
int make_lin(nunif_depth_anything *h, int N, int K, F wt, G bias, Lin *L) {
    const int NT = N / 16, KS = K / 32;
    std::vector<f16> packed((size_t)N * K + 8192, (f16)0.f);
    for (int nt = 0; nt < NT; ++nt)
        for (int ks = 0; ks < KS; ++ks) put_frag(packed, (size_t)nt * KS + ks, nt, ks, wt);
    std::vector<float> b(N);
    for (int n = 0; n < N; ++n) b[n] = bias(n);
    L->N = N; L->K = K;
    int rc = upload(h, packed, &L->w);
    return rc ? rc : upload(h, b, &L->b);
}
// k x k conv, stream [ks][nt], reduction index = tap*Cin + ci (Cin padded), wt(n, tap, ci)
template <typename F, typename G>
int make_cnv(nunif_depth_anything *h, int N, int Cin, int k, F wt, G bias, Cnv *C, int cmaj = 0) {
    const int NT = N / 16, KS = k * k * Cin / 32;
    std::vector<f16> stream((size_t)KS * NT * 512 + 8192, (f16)0.f);
    if (cmaj) {
        // chunk-major (conv3_lds_cm_kernel): [chunk of `cmaj` channels][tap][32-channel half][n-tile]
        size_t frag = 0;
        for (int q = 0; q < Cin / cmaj; ++q)
            for (int tap = 0; tap < k * k; ++tap)
                for (int c = 0; c < cmaj / 32; ++c)
                    for (int nt = 0; nt < NT; ++nt, ++frag)
                        for (int l = 0; l < 64; ++l)
                            for (int j = 0; j < 8; ++j)
                                stream[(frag * 64 + l) * 8 + j] = (f16)wt(nt * 16 + (l & 15), tap, q * cmaj + c * 32 + (l >> 4) * 8 + j);
    } else {
        for (int ks = 0; ks < KS; ++ks)
            for (int nt = 0; nt < NT; ++nt)
                put_frag(stream, (size_t)ks * NT + nt, nt, ks, [&](int n, int kk) { return wt(n, kk / Cin, kk % Cin); });
    }
    std::vector<float> b(N);
    for (int n = 0; n < N; ++n) b[n] = bias(n);
    C->N = N; C->Cin = Cin; C->k = k; C->cmaj = cmaj;
    int rc = upload(h, stream, &C->w);
    return rc ? rc : upload(h, b, &C->b);
}
int conv_from(nunif_depth_anything *h, const TMap &m, const std::string &key, int cout, int cin, int cin_pad, int k, bool has_bias,
              Cnv *C, int cmaj = 0) {
    const HostT *w, *b = nullptr;
    int rc;
    if ((rc = find(m, key + ".weight", &w)) || (has_bias && (rc = find(m, key + ".bias", &b)))) return rc;
    NUNIF_REQUIRE(w->numel == (int64_t)cout * cin * k * k, "%s: unexpected shape", key.c_str());
    const float *wd = w->data, *bd = b ? b->data : nullptr;
    return make_cnv(h, cout, cin_pad, k, [=](int n, int tap, int ci) { return ci < cin ? wd[((size_t)n * cin + ci) * k * k + tap] : 0.f; },
                    [=](int n) { return bd ? bd[n] : 0.f; }, C, cmaj);
}

int run_lin(const Lin &L, const f16 *a, int B, int Wi, int Wo, int ox, int act, const f16 *res, f16 *out, hipStream_t s,
            const char *tag, int mode = 0, int ldo = 0, int ps = 1, int Hi = 1, int lda = 0) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.a = a; g.B = B; g.Hi = Hi; g.Wi = Wi; g.Cin = L.K; g.Ho = Hi; g.Wo = Wo; g.stride = 1; g.ox = ox; g.kw = 1;
    g.K = L.K; g.w = L.w; g.bias = L.b; g.N = L.N; g.mode = mode; g.act = act; g.res = res; g.out = out;
    g.ldo = ldo ? ldo : L.N; g.n_real = L.N; g.ps = ps; g.lda = lda;
    return launch_gemm(g, s, tag);
}
// token-matrix Linear [T][K] -> [T][N]: the output-stationary kernel when the shape allows, else gemm_kernel
int run_tok(const Lin &L, const f16 *a, long T, int act, const f16 *res, f16 *out, hipStream_t s, const char *tag,
            float2 *stats_out = nullptr, const float2 *stats_in = nullptr) {
    if (gemm_os_supported(T, L.N, L.K)) {
        GemmOsArgs g;
        memset(&g, 0, sizeof(g));
        g.a = a; g.M = T; g.lda = L.K; g.K = L.K; g.w = L.w; g.bias = L.b; g.N = L.N; g.act = act; g.res = res; g.out = out;
        g.ldo = L.N; g.stats_out = stats_out; g.stats_in = stats_in; g.stats_parts = 12; g.wsum = L.ws; g.ln_eps = 1e-6f;
        return launch_gemm_os(g, s, tag);
    }
    NUNIF_REQUIRE(!stats_out && !stats_in, "%s: LayerNorm statistics need the output-stationary Linear", tag);
    return run_lin(L, a, 1, (int)T, (int)T, 0, act, res, out, s, tag);
}
int run_cnv(const Cnv &C, const f16 *a, int B, int Hi, int Wi, int stride, int zpad, int relu_in, int act, const f16 *res,
            const f16 *res2, f16 *out, hipStream_t s, int ldo = 0, float *part32 = nullptr) {
    ConvArgs c;
    memset(&c, 0, sizeof(c));
    c.a = a; c.B = B; c.Hi = Hi; c.Wi = Wi; c.Cin = C.Cin; c.stride = stride; c.kh = C.k; c.kw = C.k;
    c.Ho = (Hi + 2 * zpad - C.k) / stride + 1; c.Wo = (Wi + 2 * zpad - C.k) / stride + 1;
    c.wstream = C.w; c.bias = C.b; c.N = C.N; c.n_real = C.N; c.act = act; c.out = out; c.zpad = zpad; c.relu_in = relu_in;
    c.res = res; c.res2 = res2; c.ldo = ldo; c.cmaj = C.cmaj; c.part32 = part32;
    return launch_conv(c, s);
}
}  // namespace

namespace {
int launch_da_layernorm(const f16 *x, const float *g, const float *b, f16 *y, long T, int D, hipStream_t s) {
    const unsigned grid = (unsigned)((T + 3) / 4);
    if (D == 384) da_layernorm_kernel<6><<<grid, 256, 0, s>>>(x, g, b, y, T);
    else if (D == 768) da_layernorm_kernel<12><<<grid, 256, 0, s>>>(x, g, b, y, T);
    else if (D == 1024) da_layernorm_kernel<16><<<grid, 256, 0, s>>>(x, g, b, y, T);
    else { set_error("depth_anything: embed dim %d unsupported (384, 768, 1024)", D); return NUNIF_HIP_EUNSUPPORTED; }
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}
}  // namespace

// taps: the four encoder blocks whose (final-norm'ed) tokens feed the DPT head, or NULL for the V2 defaults of the checkpoint's
// depth (12 blocks: 2, 5, 8, 11; 24 blocks: 4, 11, 17, 23; V1 checkpoints pass the LAST four blocks).  max_depth > 0 selects the
// metric head (Sigmoid, x max_depth); 0 the relative head (ReLU).
extern "C" int nunif_hip_depth_anything_create_ex(const nunif_tensor_desc *tensors, int32_t n_tensors, const int32_t *taps,
                                                  float max_depth, nunif_depth_anything **handle) {
    NUNIF_REQUIRE(tensors && handle && n_tensors > 0, "depth_anything_create: NULL argument");
    TMap m;
    for (int i = 0; i < n_tensors; ++i) {
        HostT t;
        t.data = tensors[i].data;
        t.numel = 1;
        for (int d = 0; d < tensors[i].ndim; ++d) { t.shape.push_back(tensors[i].shape[d]); t.numel *= tensors[i].shape[d]; }
        m[tensors[i].name] = t;
    }
    nunif_depth_anything *h = new nunif_depth_anything();
    h->max_depth = max_depth > 0.f ? max_depth : 0.f;
    int rc = NUNIF_HIP_OK;
    do {
        const std::string P = "pretrained.", H = "depth_head.";
        const HostT *w, *b, *t1, *t2;
        if ((rc = find(m, P + "patch_embed.proj.weight", &w)) || (rc = find(m, P + "patch_embed.proj.bias", &b))) break;
        const int kD = (int)b->numel;
        if ((kD != 384 && kD != 768 && kD != 1024) || w->numel != (int64_t)kD * 588) {
            set_error("depth_anything: ViT-S / B / L with 14 x 14 patches expected (embed 384 / 768 / 1024), got embed %d", kD);
            rc = NUNIF_HIP_EUNSUPPORTED; break;
        }
        h->D = kD; h->heads = kD / kHd;
        int depth = 0;
        while (m.find(P + "blocks." + std::to_string(depth) + ".attn.qkv.weight") != m.end()) ++depth;
        if (depth != 12 && depth != 24) { set_error("depth_anything: %d encoder blocks (12 or 24 expected)", depth); rc = NUNIF_HIP_EUNSUPPORTED; break; }
        h->depth = depth;
        for (int i = 0; i < 4; ++i) {
            static const int t12[4] = {2, 5, 8, 11}, t24[4] = {4, 11, 17, 23};
            h->taps[i] = taps ? taps[i] : (depth == 12 ? t12[i] : t24[i]);
            if (h->taps[i] < 0 || h->taps[i] >= depth || (i > 0 && h->taps[i] <= h->taps[i - 1])) {
                set_error("depth_anything: taps must be 4 increasing block indices < %d", depth); rc = NUNIF_HIP_EINVAL; break;
            }
        }
        if (rc) break;
        {
            const float *wd = w->data, *bd = b->data;
            if ((rc = make_lin(h, kD, kKp, [=](int n, int k) { return k < 588 ? wd[(size_t)n * 588 + k] : 0.f; },
                               [=](int n) { return bd[n]; }, &h->patch))) break;
        }
        if ((rc = find(m, P + "cls_token", &t1)) || (rc = up_f32(h, t1, &h->cls))) break;
        if ((rc = find(m, P + "norm.weight", &t1)) || (rc = find(m, P + "norm.bias", &t2)) || (rc = up_f32(h, t1, &h->norm_g)) ||
            (rc = up_f32(h, t2, &h->norm_b)))
            break;
        const float qs = (1.0f / sqrtf((float)kHd)) * 1.4426950408889634f;
        h->blk.resize(depth);
        // fc2 contracts over 4 D = 1536 / 3072 / 4096 hidden channels: gemm_kernel takes K <= 1024, so it runs as 2 or 4 K-slices
        // that accumulate into the residual stream
        const int kpiece = kD == 384 ? 768 : kD, npiece = 4 * kD / kpiece;
        for (int i = 0; i < depth && !rc; ++i) {
            const std::string bp = P + "blocks." + std::to_string(i) + ".";
            Blk &bk = h->blk[i];
            const HostT *g1, *b1, *g2, *b2, *wq, *bq, *wp, *bpj, *w1, *bb1, *w2, *bb2, *ls1, *ls2;
            if ((rc = find(m, bp + "norm1.weight", &g1)) || (rc = find(m, bp + "norm1.bias", &b1)) ||
                (rc = find(m, bp + "norm2.weight", &g2)) || (rc = find(m, bp + "norm2.bias", &b2)) ||
                (rc = find(m, bp + "attn.qkv.weight", &wq)) || (rc = find(m, bp + "attn.qkv.bias", &bq)) ||
                (rc = find(m, bp + "attn.proj.weight", &wp)) || (rc = find(m, bp + "attn.proj.bias", &bpj)) ||
                (rc = find(m, bp + "mlp.fc1.weight", &w1)) || (rc = find(m, bp + "mlp.fc1.bias", &bb1)) ||
                (rc = find(m, bp + "mlp.fc2.weight", &w2)) || (rc = find(m, bp + "mlp.fc2.bias", &bb2)) ||
                (rc = find(m, bp + "ls1.gamma", &ls1)) || (rc = find(m, bp + "ls2.gamma", &ls2)))
                break;
            if (wq->numel != (int64_t)3 * kD * kD || w1->numel != (int64_t)4 * kD * kD || w2->numel != (int64_t)4 * kD * kD) {
                set_error("%s: unexpected shapes for embed %d", bp.c_str(), kD); rc = NUNIF_HIP_EINVAL; break;
            }
            if ((rc = up_f32(h, g1, &bk.g1)) || (rc = up_f32(h, b1, &bk.b1)) || (rc = up_f32(h, g2, &bk.g2)) || (rc = up_f32(h, b2, &bk.b2))) break;
            const float *d;
            const float *e;
            d = wq->data; e = bq->data;
            if ((rc = make_lin(h, 3 * kD, kD, [=](int n, int k) { return d[(size_t)n * kD + k] * (n < kD ? qs : 1.f); },
                               [=](int n) { return e[n] * (n < kD ? qs : 1.f); }, &bk.qkv))) break;
            {   // LayerScale folded: ls * (W x + b)
                const float *wd = wp->data, *bd = bpj->data, *ls = ls1->data;
                if ((rc = make_lin(h, kD, kD, [=](int n, int k) { return wd[(size_t)n * kD + k] * ls[n]; },
                                   [=](int n) { return bd[n] * ls[n]; }, &bk.proj))) break;
            }
            d = w1->data; e = bb1->data;
            if ((rc = make_lin(h, 4 * kD, kD, [=](int n, int k) { return d[(size_t)n * kD + k]; }, [=](int n) { return e[n]; }, &bk.fc1))) break;
            if (gemm_os_consumes_stats(1, 3 * kD, kD) && kD / 32 == 12) {
                // norm1 / norm2 folded into their consumers: W (gamma xhat + beta) + b = (W diag(gamma)) xhat + (W beta + b), and
                // xhat = (x - mu) r enters as r (W' x) - r mu wsum with wsum[n] = sum_k of the fp16 weights the MFMA really uses
                auto fold = [&](const float *w, const float *b, const float *ga, const float *be, int N, float scale_upto, Lin *L) -> int {
                    int rc2 = make_lin(h, N, kD, [=](int n, int k) { return w[(size_t)n * kD + k] * ga[k] * (n < kD ? scale_upto : 1.f); },
                                       [=](int n) {
                                           double acc = b[n];
                                           for (int k = 0; k < kD; ++k) acc += (double)(float)(f16)w[(size_t)n * kD + k] * be[k];
                                           return (float)acc * (n < kD ? scale_upto : 1.f);
                                       }, L);
                    if (rc2) return rc2;
                    std::vector<float> ws(N);
                    for (int n = 0; n < N; ++n) {
                        double acc = 0.0;
                        for (int k = 0; k < kD; ++k) acc += (double)(float)(f16)(w[(size_t)n * kD + k] * ga[k] * (n < kD ? scale_upto : 1.f));
                        ws[n] = (float)acc;
                    }
                    return upload(h, ws, &L->ws);
                };
                if ((rc = fold(wq->data, bq->data, g1->data, b1->data, 3 * kD, qs, &bk.qkv_ln))) break;
                if ((rc = fold(w1->data, bb1->data, g2->data, b2->data, 4 * kD, 1.f, &bk.fc1_ln))) break;
            }
            {   // fc2 with LayerScale folded, split along K; the bias rides on the first slice
                const float *wd = w2->data, *bd = bb2->data, *ls = ls2->data;
                bk.n_fc2 = npiece;
                if (gemm_os_supported(1, kD, 4 * kD)) {     // output-stationary GEMM: the whole K = 4 D contraction in one launch
                    bk.n_fc2 = 0;
                    rc = make_lin(h, kD, 4 * kD, [=](int n, int k) { return wd[(size_t)n * 4 * kD + k] * ls[n]; },
                                  [=](int n) { return bd[n] * ls[n]; }, &bk.fc2_full);
                    if (!rc && da_mlp_supported(kD, 4 * kD) && bk.fc1_ln.w) {
                        // the same matrix for the fused MLP kernel: its B operand is the hidden tile PAIR taken straight from
                        // fc1's accumulators, so the 8 k-slots of lane group g in k-chunk ks are the channels 32 ks + 4 g + 0..3
                        // (tile 2 ks) and 32 ks + 16 + 4 g + 0..3 (tile 2 ks + 1).  Fragment order [ks / 4][tile][ks % 4]: the
                        // kernel's 24 concurrent streams then sit on different L2 channels (depth_mlp.hip)
                        const int NT = kD / 16, KS = 4 * kD / 32;
                        std::vector<f16> pc((size_t)NT * KS * 512 + 8192, (f16)0.f);
                        for (int nt = 0; nt < NT; ++nt)
                            for (int ks = 0; ks < KS; ++ks)
                                for (int l = 0; l < 64; ++l)
                                    for (int j = 0; j < 8; ++j) {
                                        const int g = l >> 4, n = nt * 16 + (l & 15);
                                        const int k = ks * 32 + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
                                        pc[((((size_t)(ks >> 2) * NT + nt) * 4 + (ks & 3)) * 64 + l) * 8 + j] = (f16)(wd[(size_t)n * 4 * kD + k] * ls[n]);
                                    }
                        rc = upload(h, pc, &bk.fc2c);
                    }
                }
                for (int q = 0; q < bk.n_fc2 && !rc; ++q)
                    rc = make_lin(h, kD, kpiece, [=](int n, int k) { return wd[(size_t)n * 4 * kD + (size_t)q * kpiece + k] * ls[n]; },
                                  [=](int n) { return q == 0 ? bd[n] * ls[n] : 0.f; }, &bk.fc2[q]);
                if (rc) break;
            }
        }
        if (rc) break;
        // ---- DPT head.  out_channels from projects.{i}, fusion width from layer1_rn; channel counts are stored padded to 32
        const HostT *rnw;
        if ((rc = find(m, H + "scratch.layer1_rn.weight", &rnw))) break;
        const int F = (int)rnw->shape[0];
        if (F != 64 && F != 128 && F != 256) { set_error("depth_anything: DPT features %d unsupported (64, 128, 256)", F); rc = NUNIF_HIP_EUNSUPPORTED; break; }
        h->feat_ch = F;
        for (int i = 0; i < 4 && !rc; ++i) {
            const HostT *pw, *pb;
            if ((rc = find(m, H + "projects." + std::to_string(i) + ".weight", &pw)) ||
                (rc = find(m, H + "projects." + std::to_string(i) + ".bias", &pb)))
                break;
            const float *wd = pw->data, *bd = pb->data;
            const int oc = (int)pb->numel, ocp = (oc + 31) / 32 * 32;
            if (pw->numel != (int64_t)oc * kD) { set_error("depth_head.projects.%d: unexpected shape", i); rc = NUNIF_HIP_EINVAL; break; }
            h->OC[i] = oc; h->OCP[i] = ocp;
            if ((rc = make_lin(h, ocp, kD, [=](int n, int k) { return n < oc ? wd[(size_t)n * kD + k] : 0.f; },
                               [=](int n) { return n < oc ? bd[n] : 0.f; }, &h->proj[i])))
                break;
            // wide inputs (> 128 channels, a multiple of 64): chunk-major stream for conv3_lds_cm_kernel
            const bool cm_off = (getenv("NUNIF_CONV3_LDS") && atoi(getenv("NUNIF_CONV3_LDS")) == 0) ||
                                (getenv("NUNIF_CONV3_CM") && atoi(getenv("NUNIF_CONV3_CM")) == 0);      // read per engine (tests)
            // (chunks of 64 channels; 32 for the 128-wide fusion maps: 72 KiB of weights per workgroup either way)
            const int cmaj = (!cm_off && ocp > 128 && ocp % 64 == 0 && (F == 64 || F == 128)) ? (F == 128 ? 32 : 64) : 0;
            rc = conv_from(h, m, H + "scratch.layer" + std::to_string(i + 1) + "_rn", F, oc, ocp, 3, false, &h->rn[i], cmaj);
        }
        if (rc) break;
        {   // resize_layers.0: ConvTranspose2d(oc0, oc0, 4, 4): N = (i*4+j)*ocp0 + co, K = ci (padded); weight [ci][co][4][4]
            if ((rc = find(m, H + "resize_layers.0.weight", &w)) || (rc = find(m, H + "resize_layers.0.bias", &b))) break;
            const float *wd = w->data, *bd = b->data;
            const int o0 = h->OC[0], p0 = h->OCP[0];
            if (w->numel != (int64_t)o0 * o0 * 16) { set_error("depth_head.resize_layers.0: unexpected shape"); rc = NUNIF_HIP_EINVAL; break; }
            if ((rc = make_lin(h, 16 * p0, p0, [=](int n, int k) {
                    const int q = n / p0, co = n % p0;
                    return (co < o0 && k < o0) ? wd[((size_t)k * o0 + co) * 16 + q] : 0.f; },
                    [=](int n) { return n % p0 < o0 ? bd[n % p0] : 0.f; }, &h->rs0))) break;
            // resize_layers.1: ConvTranspose2d(oc1, oc1, 2, 2)
            if ((rc = find(m, H + "resize_layers.1.weight", &w)) || (rc = find(m, H + "resize_layers.1.bias", &b))) break;
            const float *w1d = w->data, *b1d = b->data;
            const int o1 = h->OC[1];
            if (o1 % 32 || w->numel != (int64_t)o1 * o1 * 4) { set_error("depth_head.resize_layers.1: unexpected shape"); rc = NUNIF_HIP_EINVAL; break; }
            if ((rc = make_lin(h, 4 * o1, o1, [=](int n, int k) { return w1d[((size_t)k * o1 + n % o1) * 4 + n / o1]; },
                               [=](int n) { return b1d[n % o1]; }, &h->rs1))) break;
            // resize_layers.3: Conv2d(oc3, oc3, 3, stride 2, pad 1); conv_kernel holds <= 384 output channels per launch: chunks
            if ((rc = find(m, H + "resize_layers.3.weight", &w)) || (rc = find(m, H + "resize_layers.3.bias", &b))) break;
            const int o3 = h->OC[3];
            if (o3 % 32 || h->OC[2] % 32 || w->numel != (int64_t)o3 * o3 * 9) { set_error("depth_head.resize_layers.3: unexpected shape"); rc = NUNIF_HIP_EINVAL; break; }
            if (gemm_os_supported(1, o3, 9 * o3)) {         // im2col + output-stationary Linear (k = tap * C + ci)
                const float *w3g = w->data, *b3g = b->data;
                if ((rc = make_lin(h, o3, 9 * o3, [=](int n, int k) { const int tap = k / o3, ci = k % o3; return w3g[((size_t)n * o3 + ci) * 9 + tap]; },
                                   [=](int n) { return b3g[n]; }, &h->rs3g))) break;
            }
            const int chunk = o3 <= 384 ? o3 : 256;
            if (o3 % chunk) { set_error("depth_head.resize_layers.3: %d channels unsupported", o3); rc = NUNIF_HIP_EUNSUPPORTED; break; }
            if (!h->rs3g.w) h->rs3.resize(o3 / chunk);
            const float *w3 = w->data, *b3 = b->data;
            for (int c = 0; c < (int)h->rs3.size() && !rc; ++c)
                rc = make_cnv(h, chunk, o3, 3, [=](int n, int tap, int ci) { return w3[((size_t)(c * chunk + n) * o3 + ci) * 9 + tap]; },
                              [=](int n) { return b3[c * chunk + n]; }, &h->rs3[c]);
            if (rc) break;
        }
        for (int k = 0; k < 4 && !rc; ++k) {
            const std::string r = H + "scratch.refinenet" + std::to_string(k + 1) + ".";
            Fus &f = h->fus[k];
            if ((rc = conv_from(h, m, r + "resConfUnit1.conv1", F, F, F, 3, true, &f.r1.c1)) ||
                (rc = conv_from(h, m, r + "resConfUnit1.conv2", F, F, F, 3, true, &f.r1.c2)) ||
                (rc = conv_from(h, m, r + "resConfUnit2.conv1", F, F, F, 3, true, &f.r2.c1)) ||
                (rc = conv_from(h, m, r + "resConfUnit2.conv2", F, F, F, 3, true, &f.r2.c2)))
                break;
            const HostT *ow, *ob;
            if ((rc = find(m, r + "out_conv.weight", &ow)) || (rc = find(m, r + "out_conv.bias", &ob))) break;
            const float *wd = ow->data, *bd = ob->data;
            rc = make_lin(h, F, F, [=](int n, int kk) { return wd[(size_t)n * F + kk]; }, [=](int n) { return bd[n]; }, &f.out);
        }
        if (rc) break;
        if ((rc = conv_from(h, m, H + "scratch.output_conv1", F / 2, F, F, 3, true, &h->oc1)) ||
            (rc = conv_from(h, m, H + "scratch.output_conv2.0", 32, F / 2, F / 2, 3, true, &h->oc2)))
            break;
        if ((rc = find(m, H + "scratch.output_conv2.2.weight", &w)) || (rc = find(m, H + "scratch.output_conv2.2.bias", &b))) break;
        std::vector<float> wf(33);
        for (int k = 0; k < 32; ++k) wf[k] = w->data[k];
        wf[32] = b->data[0];
        if ((rc = upload(h, wf, &h->w_final))) break;
        // ---- Video-Depth-Anything: head.motion_modules.{0..3} (published key layout; the caller maps `head.` to `depth_head.`)
        if (m.find(H + "motion_modules.0.temporal_transformer.proj_in.weight") == m.end()) break;
        h->temporal = true;
        const int chans[4] = {h->OC[2], h->OC[3], F, F};
        for (int i = 0; i < 4 && !rc; ++i) {
            const int C = chans[i], hd = C / 8;
            TMod &tmod = h->tm[i];
            tmod.C = C;
            if (C % 64 || hd % 8) { set_error("motion_modules.%d: %d channels unsupported (a multiple of 64)", i, C); rc = NUNIF_HIP_EUNSUPPORTED; break; }
            const std::string T = H + "motion_modules." + std::to_string(i) + ".temporal_transformer.", B0 = T + "transformer_blocks.0.";
            const HostT *g1, *b1, *wi, *bi, *wo, *bo, *fg, *fb, *w1, *bb1, *w2, *bb2;
            if ((rc = find(m, T + "norm.weight", &g1)) || (rc = find(m, T + "norm.bias", &b1)) ||
                (rc = find(m, T + "proj_in.weight", &wi)) || (rc = find(m, T + "proj_in.bias", &bi)) ||
                (rc = find(m, T + "proj_out.weight", &wo)) || (rc = find(m, T + "proj_out.bias", &bo)) ||
                (rc = find(m, B0 + "ff_norm.weight", &fg)) || (rc = find(m, B0 + "ff_norm.bias", &fb)) ||
                (rc = find(m, B0 + "ff.net.0.proj.weight", &w1)) || (rc = find(m, B0 + "ff.net.0.proj.bias", &bb1)) ||
                (rc = find(m, B0 + "ff.net.2.weight", &w2)) || (rc = find(m, B0 + "ff.net.2.bias", &bb2)))
                break;
            if (g1->numel != C || wi->numel != (int64_t)C * C || wo->numel != (int64_t)C * C || w1->numel != (int64_t)8 * C * C ||
                w2->numel != (int64_t)4 * C * C) {
                set_error("motion_modules.%d: unexpected shapes for %d channels", i, C); rc = NUNIF_HIP_EINVAL; break;
            }
            if ((rc = up_f32(h, g1, &tmod.gn_g)) || (rc = up_f32(h, b1, &tmod.gn_b)) || (rc = up_f32(h, fg, &tmod.ff_g)) ||
                (rc = up_f32(h, fb, &tmod.ff_b)))
                break;
            auto plain = [&](const HostT *w, const HostT *b, int N, int K, Lin *L) {
                const float *wd = w->data, *bd = b ? b->data : nullptr;
                return make_lin(h, N, K, [=](int n, int k) { return wd[(size_t)n * K + k]; }, [=](int n) { return bd ? bd[n] : 0.f; }, L);
            };
            if ((rc = plain(wi, bi, C, C, &tmod.proj_in)) || (rc = plain(wo, bo, C, C, &tmod.proj_out)) ||
                (rc = plain(w1, bb1, 8 * C, C, &tmod.ff1)) || (rc = plain(w2, bb2, C, 4 * C, &tmod.ff2)))
                break;
            const float qs = (1.0f / sqrtf((float)hd)) * 1.4426950408889634f;
            for (int a = 0; a < 2 && !rc; ++a) {
                TAtt &at = tmod.at[a];
                const std::string A = B0 + "attention_blocks." + std::to_string(a) + ".";
                const HostT *ng, *nb, *wq, *wk, *wv, *wo2, *bo2;
                if ((rc = find(m, B0 + "norms." + std::to_string(a) + ".weight", &ng)) ||
                    (rc = find(m, B0 + "norms." + std::to_string(a) + ".bias", &nb)) || (rc = find(m, A + "to_q.weight", &wq)) ||
                    (rc = find(m, A + "to_k.weight", &wk)) || (rc = find(m, A + "to_v.weight", &wv)) ||
                    (rc = find(m, A + "to_out.0.weight", &wo2)) || (rc = find(m, A + "to_out.0.bias", &bo2)))
                    break;
                if (wq->numel != (int64_t)C * C || wk->numel != (int64_t)C * C || wv->numel != (int64_t)C * C || wo2->numel != (int64_t)C * C) {
                    set_error("%s: unexpected shapes for %d channels", A.c_str(), C); rc = NUNIF_HIP_EINVAL; break;
                }
                if ((rc = up_f32(h, ng, &at.g)) || (rc = up_f32(h, nb, &at.b)) || (rc = plain(wo2, bo2, C, C, &at.out))) break;
                const float *q = wq->data, *k = wk->data, *v = wv->data;
                if ((rc = make_lin(h, 3 * C, C, [=](int n, int kk) {
                        return n < C ? q[(size_t)n * C + kk] * qs : n < 2 * C ? k[(size_t)(n - C) * C + kk] : v[(size_t)(n - 2 * C) * C + kk]; },
                        [](int) { return 0.f; }, &at.qkv)))
                    break;
                // the sinusoidal code of window position j (AnimateDiff PositionalEncoding; the checkpoint's `pos_encoder.pe` buffer
                // when it is there) through the three projections: [32][C] fp32 tables
                std::vector<double> pe((size_t)kTLen * C);
                auto it = m.find(A + "pos_encoder.pe");
                if (it != m.end() && it->second.numel >= (int64_t)kTLen * C) {
                    for (size_t e = 0; e < pe.size(); ++e) pe[e] = it->second.data[e];
                } else {
                    for (int j = 0; j < kTLen; ++j)
                        for (int c2 = 0; c2 < C; c2 += 2) {
                            // float arithmetic like torch.exp(arange * (-ln 1e4 / d)) in fp32
                            const float div = expf((float)c2 * (-logf(10000.0f) / (float)C));
                            pe[(size_t)j * C + c2] = sinf((float)j * div);
                            pe[(size_t)j * C + c2 + 1] = cosf((float)j * div);
                        }
                }
                std::vector<float> tq((size_t)kTLen * C), tk((size_t)kTLen * C), tv((size_t)kTLen * C);
                for (int j = 0; j < kTLen; ++j)
                    for (int n = 0; n < C; ++n) {
                        double aq = 0.0, ak = 0.0, av = 0.0;
                        for (int kk = 0; kk < C; ++kk) {
                            const double e = pe[(size_t)j * C + kk];
                            aq += (double)q[(size_t)n * C + kk] * e;
                            ak += (double)k[(size_t)n * C + kk] * e;
                            av += (double)v[(size_t)n * C + kk] * e;
                        }
                        tq[(size_t)j * C + n] = (float)(aq * qs);
                        tk[(size_t)j * C + n] = (float)ak;
                        tv[(size_t)j * C + n] = (float)av;
                    }
                if ((rc = upload(h, tq, &at.pq)) || (rc = upload(h, tk, &at.pk)) || (rc = upload(h, tv, &at.pv))) break;
            }
        }
    } while (0);
    if (rc) { nunif_hip_depth_anything_destroy(h); return rc; }
    *handle = h;
    return NUNIF_HIP_OK;
}

extern "C" int nunif_hip_depth_anything_create(const nunif_tensor_desc *tensors, int32_t n_tensors,
                                               nunif_depth_anything **handle) {
    return nunif_hip_depth_anything_create_ex(tensors, n_tensors, nullptr, 0.f, handle);
}

extern "C" void nunif_hip_depth_anything_destroy(nunif_depth_anything *h) {
    if (!h) return;
    for (void *p : h->owned) (void)hipFree(p);
    Buf *bufs[] = {&h->a_col, &h->pe, &h->t, &h->y, &h->qkv, &h->att, &h->hid, &h->lnstats, &h->mlp_flags, &h->feat[0], &h->feat[1], &h->feat[2],
                   &h->feat[3], &h->rnb[0], &h->rnb[1], &h->rnb[2], &h->rnb[3], &h->m1, &h->m2, &h->m3, &h->m4, &h->m5, &h->part, &h->col};
    for (Buf *b : bufs) b->release();
    for (int i = 0; i < 4; ++i) { h->bm1[i].release(); h->bm2[i].release(); h->bpart[i].release(); }
    for (int i = 0; i < 3; ++i) { h->bt1[i].release(); h->br1[i].release(); }
    for (TMod &tmod : h->tm)
        for (TAtt &at : tmod.at) { at.kc.release(); at.vc.release(); }
    for (Buf *b : {&h->ta, &h->th, &h->tqkv, &h->tatt, &h->thid, &h->tgg, &h->tpart}) b->release();
    for (int i = 0; i < 3; ++i) {
        if (h->side[i]) (void)hipStreamDestroy(h->side[i]);
        if (h->ev_fork[i]) (void)hipEventDestroy(h->ev_fork[i]);
        if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
    }
    delete h;
}

// x: [B,3,h,w] f32 (ImageNet-normalised), pos: [1 + gh*gw][D] f32 (interpolated position embedding, device),
// depth: [B,h,w] f32
extern "C" int nunif_hip_depth_anything_forward(nunif_depth_anything *h, const float *x, const float *pos, float *depth,
                                                int32_t B, int32_t hh, int32_t ww, void *stream) {
    NUNIF_REQUIRE(h && x && pos && depth && B > 0 && hh >= 28 && ww >= 28 && hh % kPatch == 0 && ww % kPatch == 0,
                  "depth_anything_forward: h, w must be multiples of 14 (>= 28)");
    hipStream_t s = (hipStream_t)stream;
    const int kD = h->D, kHeads = h->heads, F = h->feat_ch;
    const int gh = hh / kPatch, gw = ww / kPatch, N = gh * gw, Np = N + 1;
    const long T = (long)B * Np;
    const size_t e2 = sizeof(f16);
    // DPT map sizes
    const int H1 = gh * 4, W1 = gw * 4, H2 = gh * 2, W2 = gw * 2, H3 = gh, W3 = gw, H4 = (gh + 2 - 3) / 2 + 1, W4 = (gw + 2 - 3) / 2 + 1;
    const int HF = 2 * H1, WF = 2 * W1;
    const int Hs[4] = {H1, H2, H3, H4}, Ws[4] = {W1, W2, W3, W4};
    // (a Video-Depth-Anything engine takes the B frames as CONSECUTIVE frames of its stream: the same results as B calls with one)
    const int ocp_max = std::max(std::max(h->OCP[0], h->OCP[1]), std::max(h->OCP[2], h->OCP[3]));
    size_t big = std::max<size_t>((size_t)B * HF * WF * F, (size_t)B * hh * ww * std::max(32, F / 2));
    big = std::max<size_t>(big, (size_t)B * N * ocp_max);
    big = std::max<size_t>(big, (size_t)B * H1 * W1 * h->OCP[0]);
    big = std::max<size_t>(big, (size_t)B * H2 * W2 * h->OCP[1]);
    int rc;
    if ((rc = h->a_col.ensure((size_t)B * N * kKp * e2)) || (rc = h->pe.ensure((size_t)B * N * kD * e2)) ||
        (rc = h->t.ensure(T * kD * e2)) || (rc = h->y.ensure(T * kD * e2)) || (rc = h->qkv.ensure(T * 3 * kD * e2)) ||
        (rc = h->att.ensure(T * kD * e2)) ||
        (rc = h->hid.ensure(T * 4 * kD * e2)) || (rc = h->m1.ensure(big * e2)) || (rc = h->m2.ensure(big * e2)) ||
        (rc = h->m3.ensure(big * e2)) || (rc = h->m4.ensure(big * e2)) || (rc = h->m5.ensure(big * e2)))
        return rc;
    {
        // LayerNorm partial sums: producers write live rows only, the consumer's last 32-token tile reads its pad rows as well
        // (results discarded by the m < M store guard): zeroed once per allocation so that nothing uninitialised is ever read
        const size_t ln_bytes = (size_t)((T + 31) / 32) * 32 * 12 * sizeof(float2);
        const bool fresh = ln_bytes > h->lnstats.cap;
        if ((rc = h->lnstats.ensure(ln_bytes))) return rc;
        if (fresh) NUNIF_HIP_CHECK(hipMemsetAsync(h->lnstats.p, 0, h->lnstats.cap, s));
        // the hidden-split MLP kernel's hand-off flags (one per workgroup, raised by the sender and lowered by the receiver: zeroed once)
        const size_t fl_bytes = (size_t)da_mlp_flag_count(T) * sizeof(unsigned);
        // ... and again at the start of every forward: a launch that faulted or was aborted half-way leaves flags raised, and the
        // next forward must not consume stale partial sums behind them
        if ((rc = h->mlp_flags.ensure(fl_bytes))) return rc;
        NUNIF_HIP_CHECK(hipMemsetAsync(h->mlp_flags.p, 0, h->mlp_flags.cap, s));
    }
    for (int i = 0; i < 4; ++i)
        if ((rc = h->feat[i].ensure(T * kD * e2)) || (rc = h->rnb[i].ensure((size_t)B * Hs[i] * Ws[i] * F * e2))) return rc;
    // Reassemble branches beside the encoder.  Tap k (after layers 2 / 5 / 8 of ViT-S) feeds project -> resize -> layer_rn conv,
    // three small launches (43 / 22 / up to 392 workgroups) that nothing needs before the refinenets: they run on side streams
    // while the encoder goes on (its launches leave a quarter to a third of the CUs idle).  Everything is allocated here, before
    // the first fork (hipMalloc synchronises the device).  NUNIF_DA_BRANCH_STREAMS=0: everything on the caller's stream.
    // (a temporal engine keeps everything on the caller's stream: its modules share scratch buffers and sit inside the branches)
    const bool side_streams = !h->temporal && !(getenv("NUNIF_DA_BRANCH_STREAMS") && atoi(getenv("NUNIF_DA_BRANCH_STREAMS")) == 0);
    for (int i = 0; i < 4; ++i) {
        if ((rc = h->bm1[i].ensure((size_t)B * N * h->OCP[i] * e2)) || (rc = h->bm2[i].ensure((size_t)B * Hs[i] * Ws[i] * h->OCP[i] * e2))) return rc;
        if (h->rn[i].cmaj && (rc = h->bpart[i].ensure((size_t)(h->rn[i].Cin / h->rn[i].cmaj) * B * Hs[i] * Ws[i] * h->rn[i].N * sizeof(float)))) return rc;
    }
    // RCU1 of refinenet k + 1 (`residual_layer1` on layer{k+1}_rn, DPT FeatureFusionBlock) depends on the reassembled map only, not
    // on the path coming down the pyramid: its two 3x3 convs join the branch of tap k and leave the head's critical path; the head
    // then adds the result where it resizes the path into that stage (da_upsample_kernel `add`).  Needs the out_conv-first order
    // (the resize is then the last op in front of the stage).  NUNIF_DA_RCU1_BRANCH=0: RCU1 in the head, as before.
    static const bool conv_first_g = !(getenv("NUNIF_DA_OUTCONV_FIRST") && atoi(getenv("NUNIF_DA_OUTCONV_FIRST")) == 0);
    // (a temporal module works on path_4 / path_3 BEFORE the skip joins them: RCU1 stays in the head there)
    const bool rcu1_branch = !h->temporal && conv_first_g && !(getenv("NUNIF_DA_RCU1_BRANCH") && atoi(getenv("NUNIF_DA_RCU1_BRANCH")) == 0);
    if (rcu1_branch)
        for (int i = 0; i < 3; ++i)
            if ((rc = h->bt1[i].ensure((size_t)B * Hs[i] * Ws[i] * F * e2)) || (rc = h->br1[i].ensure((size_t)B * Hs[i] * Ws[i] * F * e2)))
                return rc;
    if (h->rs3g.w && (rc = h->col.ensure((size_t)B * H4 * W4 * 9 * h->OCP[3] * e2))) return rc;
    if (side_streams && !h->side[0]) {
        for (int i = 0; i < 3; ++i) {
            NUNIF_HIP_CHECK(hipStreamCreateWithFlags(&h->side[i], hipStreamNonBlocking));
            NUNIF_HIP_CHECK(hipEventCreateWithFlags(&h->ev_fork[i], hipEventDisableTiming));
            NUNIF_HIP_CHECK(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
        }
    }
    f16 *a_col = (f16 *)h->a_col.p, *pe = (f16 *)h->pe.p, *t = (f16 *)h->t.p, *y = (f16 *)h->y.p, *qkv = (f16 *)h->qkv.p;
    f16 *att = (f16 *)h->att.p, *hid = (f16 *)h->hid.p;
    float2 *lnstats = (float2 *)h->lnstats.p;
    bool fuse_ln = kD / 32 == 12 && gemm_os_consumes_stats(T, 3 * kD, kD) && gemm_os_supported(T, kD, kD) && gemm_os_supported(T, kD, 4 * kD);
    for (const Blk &bk : h->blk) fuse_ln = fuse_ln && bk.qkv_ln.w && bk.fc1_ln.w && bk.n_fc2 == 0;
    const bool use_mlp = !(getenv("NUNIF_DA_MLP") && atoi(getenv("NUNIF_DA_MLP")) == 0);      // read per call (tests A/B it)
    SplitLease lease(s);
    auto blocks = [](long n) { return (unsigned)((n + 255) / 256); };

    // ---- Video-Depth-Anything: the window state and one temporal module on one frame's map (depth_temporal.hip) ------------------
    const int tP[4] = {H3 * W3, H4 * W4, H3 * W3, H2 * W2};          // layer_3, layer_4, path_4, path_3
    // the window position / ring start of each frame of the batch, and the state behind the last one
    std::vector<int> t_idx(B, 0), t_st(B, 0);
    int t_len_end = h->t_len, t_start_end = h->t_start;
    if (h->temporal) {
        bool same = true;
        for (int i = 0; i < 4; ++i) same = same && h->t_P[i] == tP[i];
        if (!same) { h->t_len = 0; h->t_start = 0; for (int i = 0; i < 4; ++i) h->t_P[i] = tP[i]; }     // another resolution: a new window
        t_len_end = h->t_len; t_start_end = h->t_start;
        for (int f = 0; f < B; ++f) {
            // frame f joins the window at position t_len; beyond 31 cached frames the oldest leaves (its slot is the next one written)
            t_idx[f] = t_len_end; t_st[f] = t_start_end;
            if (t_len_end + 1 > kTLen - 1) t_start_end = (t_start_end + 1) & (kTLen - 1);
            else ++t_len_end;
        }
        size_t pc = 0;
        for (int i = 0; i < 4; ++i) {
            pc = std::max(pc, (size_t)B * tP[i] * h->tm[i].C);
            for (TAtt &at : h->tm[i].at)
                if ((rc = at.kc.ensure((size_t)kTLen * tP[i] * h->tm[i].C * e2)) || (rc = at.vc.ensure((size_t)kTLen * tP[i] * h->tm[i].C * e2))) return rc;
        }
        if ((rc = h->ta.ensure(pc * e2)) || (rc = h->th.ensure(pc * e2)) || (rc = h->tatt.ensure(pc * e2)) || (rc = h->tqkv.ensure(3 * pc * e2)) ||
            (rc = h->thid.ensure(8 * pc * e2)) || (rc = h->tgg.ensure(4 * pc * e2)) ||
            (rc = h->tpart.ensure((size_t)B * (kVdaGnBlocks + 1) * 1024 * sizeof(float2))))
            return rc;
    }
    // x: the B frames' maps [B][tP[i]][C], updated in place.  Everything per token runs once over the B * P tokens; GroupNorm is per
    // frame (blockIdx.y); the attention is the one sequential step — frame f attends to the window as frames 0 .. f - 1 left it
    auto run_tmod = [&](int i, f16 *x, hipStream_t st) -> int {
        TMod &tm = h->tm[i];
        const int C = tm.C, P = tP[i];
        const long T = (long)B * P;
        f16 *a = (f16 *)h->ta.p, *hs = (f16 *)h->th.p, *tq = (f16 *)h->tqkv.p, *ta = (f16 *)h->tatt.p, *hid = (f16 *)h->thid.p, *gg = (f16 *)h->tgg.p;
        int rc;
        if ((rc = launch_vda_groupnorm(x, tm.gn_g, tm.gn_b, a, (float2 *)h->tpart.p, B, P, C, 1e-6f, st))) return rc;
        if ((rc = run_tok(tm.proj_in, a, T, 0, nullptr, hs, st, "vda_proj_in"))) return rc;
        for (TAtt &at : tm.at) {
            if ((rc = launch_vda_layernorm(hs, at.g, at.b, a, T, C, 1e-5f, st))) return rc;
            if ((rc = run_tok(at.qkv, a, T, 0, nullptr, tq, st, "vda_qkv"))) return rc;
            for (int f = 0; f < B; ++f) {
                VdaTattnArgs ga;
                ga.qkv = tq + (size_t)f * P * 3 * C; ga.kc = (f16 *)at.kc.p; ga.vc = (f16 *)at.vc.p; ga.pq = at.pq; ga.pk = at.pk; ga.pv = at.pv;
                ga.att = ta + (size_t)f * P * C; ga.P = P; ga.C = C; ga.hd = C / 8; ga.start = t_st[f]; ga.idx = t_idx[f];
                if ((rc = launch_vda_tattn(ga, st))) return rc;
            }
            if ((rc = run_tok(at.out, ta, T, 0, hs, hs, st, "vda_to_out"))) return rc;                 // hs += to_out(att)
        }
        if ((rc = launch_vda_layernorm(hs, tm.ff_g, tm.ff_b, a, T, C, 1e-5f, st))) return rc;
        if ((rc = run_tok(tm.ff1, a, T, 0, nullptr, hid, st, "vda_ff1"))) return rc;
        if ((rc = launch_vda_geglu(hid, gg, T, 4 * C, st))) return rc;
        if ((rc = run_tok(tm.ff2, gg, T, 0, hs, hs, st, "vda_ff2"))) return rc;                        // hs += ff2(geglu(ff1(norm(hs))))
        return run_tok(tm.proj_out, hs, T, 0, x, x, st, "vda_proj_out");                               // x += proj_out(hs)
    };

    {   // patch embedding
        ProfScope ps("da_im2col_kernel", s, 0.0, (double)B * N * (kKp * 2.0 + 588 * 4.0));
        da_im2col_kernel<<<blocks((long)B * N * kKp), 256, 0, s>>>(x, a_col, B, hh, ww, gh, gw);
        NUNIF_LAUNCH_CHECK();
    }
    if ((rc = run_lin(h->patch, a_col, 1, B * N, B * N, 0, 0, nullptr, pe, s, "da_patch"))) return rc;
    da_assemble_kernel<<<blocks(T * kD), 256, 0, s>>>(pe, h->cls, pos, t, B, Np, kD);
    NUNIF_LAUNCH_CHECK();

    f16 *rn[4];
    for (int i = 0; i < 4; ++i) rn[i] = (f16 *)h->rnb[i].p;
    // reassemble branch i: projects[i] (1x1 on the patch tokens) -> resize_layers[i] -> layer{i+1}_rn, on stream st
    auto branch = [&](int i, hipStream_t st) -> int {
        int rc;
        f16 *m1 = (f16 *)h->bm1[i].p, *m2 = (f16 *)h->bm2[i].p;
        const f16 *feat = (const f16 *)h->feat[i].p;
        // projects[i]: 1x1 on the patch tokens (row 0 of every image = class token is skipped: Wi = Np, ox = 1)
        if ((rc = run_lin(h->proj[i], feat, B, Np, N, 1, 0, nullptr, m1, st, "da_project"))) return rc;
        const f16 *src = m1;
        if (i == 0) { if ((rc = run_lin(h->rs0, m1, B, gw, gw, 0, 0, nullptr, m2, st, "da_resize0", 1, h->OCP[0], 4, gh))) return rc; src = m2; }
        else if (i == 1) { if ((rc = run_lin(h->rs1, m1, B, gw, gw, 0, 0, nullptr, m2, st, "da_resize1", 1, h->OCP[1], 2, gh))) return rc; src = m2; }
        else if (i == 3) {
            if (h->rs3g.w) {
                const int C3 = h->OCP[3];
                const long M4 = (long)B * H4 * W4;
                da_im2col_s2_kernel<<<blocks(M4 * 9 * (C3 / 8)), 256, 0, st>>>(m1, (f16 *)h->col.p, B, gh, gw, C3, H4, W4);
                NUNIF_LAUNCH_CHECK();
                if ((rc = run_tok(h->rs3g, (const f16 *)h->col.p, M4, 0, nullptr, m2, st, "da_resize3"))) return rc;
            } else {
                const int chunk = h->rs3[0].N;
                for (size_t c = 0; c < h->rs3.size(); ++c)
                    if ((rc = run_cnv(h->rs3[c], m1, B, gh, gw, 2, 1, 0, 0, nullptr, nullptr, m2 + c * chunk, st, h->OCP[3]))) return rc;
            }
            src = m2;
        }
        // Video-Depth-Anything: motion_modules[0] on layer_3, [1] on layer_4 (after resize_layers, before layer{3,4}_rn)
        if (h->temporal && i >= 2 && (rc = run_tmod(i - 2, i == 2 ? m1 : m2, st))) return rc;
        float *part = h->rn[i].cmaj ? (float *)h->bpart[i].p : nullptr;
        if ((rc = run_cnv(h->rn[i], src, B, Hs[i], Ws[i], 1, 1, 0, 0, nullptr, nullptr, rn[i], st, 0, part))) return rc;
        if (rcu1_branch && i < 3) {
            // br1 = conv2(relu(conv1(relu(rn)))) + rn
            const Rcu &r = h->fus[i].r1;
            f16 *t1 = (f16 *)h->bt1[i].p;
            if ((rc = run_cnv(r.c1, rn[i], B, Hs[i], Ws[i], 1, 1, 1, 3, nullptr, nullptr, t1, st))) return rc;
            if ((rc = run_cnv(r.c2, t1, B, Hs[i], Ws[i], 1, 1, 0, 0, rn[i], nullptr, (f16 *)h->br1[i].p, st))) return rc;
        }
        return NUNIF_HIP_OK;
    };
    bool forked[4] = {false, false, false, false};
    auto fork = [&](int k) -> int {              // tap k's map is complete on s: its branch starts now, on its own stream
        if (!side_streams || k >= 3) return NUNIF_HIP_OK;
        NUNIF_HIP_CHECK(hipEventRecord(h->ev_fork[k], s));
        NUNIF_HIP_CHECK(hipStreamWaitEvent(h->side[k], h->ev_fork[k], 0));
        int rc = branch(k, h->side[k]);
        if (rc) return rc;
        NUNIF_HIP_CHECK(hipEventRecord(h->ev_join[k], h->side[k]));
        forked[k] = true;
        return NUNIF_HIP_OK;
    };

    int tap = 0;
    for (int i = 0; i < h->depth; ++i) {
        const Blk &bk = h->blk[i];
        // ViT-S: norm1 (from the second block on) and norm2 have no kernel of their own — the Linear in front of them (fc2 of the
        // previous block, proj) writes per-token partial sums of what it stores, the Linear behind them (qkv, fc1) multiplies the
        // raw rows and finishes with r (W x - mu wsum) + b (GemmOsArgs::stats_out / stats_in)
        const bool ln2 = fuse_ln, ln1 = fuse_ln && i > 0, ln1_next = fuse_ln && i + 1 < h->depth;
        if (ln1) {
            if ((rc = run_tok(bk.qkv_ln, t, T, 0, nullptr, qkv, s, "da_qkv", nullptr, lnstats))) return rc;
        } else {
            {
                ProfScope ps("da_layernorm_kernel", s, 0.0, (double)T * kD * 4.0);
                if ((rc = launch_da_layernorm(t, bk.g1, bk.b1, y, T, kD, s))) return rc;
            }
            if ((rc = run_tok(bk.qkv, y, T, 0, nullptr, qkv, s, "da_qkv"))) return rc;
        }
        {
            ProfScope ps("da_attn_kernel", s, 4.0 * B * (double)Np * Np * kD, (double)T * kD * 8.0);
            // One wave = one 16-query tile over ALL keys, so the chip's time is (workgroups per CU) x (waves per SIMD) wave-passes:
            // ViT-S at B = 4 is 24 (frame, head) pairs x 86 tiles = 264 workgroups of 8 — eight more than CUs, and the CUs that
            // get two run four waves per SIMD for twice as long while the other 248 idle (SQ_WAVE_CYCLES: 1.34 waves per SIMD on
            // average).  12 tiles per workgroup are 192 workgroups of 3 waves per SIMD: 3 passes instead of 4 (measured on one box:
            // 0.530 -> 0.435 ms per 12 launches; 16 tiles: 0.506).
            const int tiles = (Np + 15) / 16, pairs = kHeads * B;
            auto passes = [&](int w) { return (long)((pairs * ((tiles + w - 1) / w) + 255) / 256) * ((w + 3) / 4); };
            int w = passes(12) < passes(8) ? 12 : 8;
            if (passes(16) < passes(w)) w = 16;
            const unsigned grid = (unsigned)((tiles + w - 1) / w) * kHeads * B;
            if (w == 8) da_attn_kernel<8><<<grid, 512, 0, s>>>(qkv, att, Np, kD, kHeads);
            else if (w == 12) da_attn_kernel<12><<<grid, 768, 0, s>>>(qkv, att, Np, kD, kHeads);
            else da_attn_kernel<16><<<grid, 1024, 0, s>>>(qkv, att, Np, kD, kHeads);
            NUNIF_LAUNCH_CHECK();
        }
        if ((rc = run_tok(bk.proj, att, T, 0, t, t, s, "da_proj", ln2 ? lnstats : nullptr))) return rc;        // t += ls1 * proj(att)
        if (ln2 && bk.fc2c && use_mlp) {
            // fc1 + GELU + fc2 + residual (+ the next norm1's statistics) in one kernel: the hidden rows never leave the CU
            DaMlpArgs ma;
            memset(&ma, 0, sizeof(ma));
            ma.t = t; ma.M = T; ma.w1 = bk.fc1_ln.w; ma.b1 = bk.fc1_ln.b; ma.ws1 = bk.fc1_ln.ws; ma.w2c = bk.fc2c; ma.b2 = bk.fc2_full.b;
            ma.stats_in = lnstats; ma.stats_out = ln1_next ? lnstats : nullptr; ma.ln_eps = 1e-6f;
            // the split form needs BOTH blocks of a pair resident; that holds while it is the only spin-waiting grid on the
            // device (depth_mlp.hip launch_da_mlp) — `lease`, above
            if (h->mlp_flags.p && lease.held && (size_t)da_mlp_partial_bytes(T) <= h->hid.cap) {   // the hidden rows' buffer is free on this path
                ma.partial = hid; ma.flags = (unsigned *)h->mlp_flags.p;
            }
            if ((rc = launch_da_mlp(ma, s))) return rc;
            if (tap < 4 && i == h->taps[tap]) {
                if ((rc = launch_da_layernorm(t, h->norm_g, h->norm_b, (f16 *)h->feat[tap].p, T, kD, s))) return rc;
                if ((rc = fork(tap))) return rc;
                ++tap;
            }
            continue;
        }
        if (ln2) {
            if ((rc = run_tok(bk.fc1_ln, t, T, 1, nullptr, hid, s, "da_fc1", nullptr, lnstats))) return rc;
        } else {
            if ((rc = launch_da_layernorm(t, bk.g2, bk.b2, y, T, kD, s))) return rc;
            if ((rc = run_tok(bk.fc1, y, T, 1, nullptr, hid, s, "da_fc1"))) return rc;      // GELU(erf)
        }
        // t += ls2 * fc2(.): K-slices of the 4 D-wide hidden rows (lda = 4 D)
        if (bk.n_fc2 == 0 && (rc = run_tok(bk.fc2_full, hid, T, 0, t, t, s, "da_fc2", ln1_next ? lnstats : nullptr))) return rc;
        for (int q = 0; q < bk.n_fc2; ++q)
            if ((rc = run_lin(bk.fc2[q], hid + (size_t)q * bk.fc2[q].K, 1, (int)T, (int)T, 0, 0, t, t, s, "da_fc2", 0, 0, 1, 1, 4 * kD)))
                return rc;
        if (tap < 4 && i == h->taps[tap]) {
            if ((rc = launch_da_layernorm(t, h->norm_g, h->norm_b, (f16 *)h->feat[tap].p, T, kD, s))) return rc;
            if ((rc = fork(tap))) return rc;
            ++tap;
        }
    }

    // ---- DPT head ----------------------------------------------------------------------------------------------------
    f16 *m1 = (f16 *)h->m1.p, *m2 = (f16 *)h->m2.p, *m3 = (f16 *)h->m3.p, *m4 = (f16 *)h->m4.p, *m5 = (f16 *)h->m5.p;
    // branches that were not forked beside the encoder (the last tap's always) run here; the others are joined where their map is
    // first needed
    for (int i = 0; i < 4; ++i)
        if (!forked[i] && (rc = branch(i, s))) return rc;
    for (int i = 0; i < 3; ++i)
        if (forked[i]) NUNIF_HIP_CHECK(hipStreamWaitEvent(s, h->ev_join[i], 0));
    // refinenet k: x = path (+ RCU1(skip)); x = RCU2(x); upsample; out_conv
    auto rcu = [&](const Rcu &r, const f16 *in, int Hc, int Wc, const f16 *extra, f16 *tmp, f16 *out) -> int {
        // out = conv2(relu(conv1(relu(in)))) + in (+ extra)
        int e = run_cnv(r.c1, in, B, Hc, Wc, 1, 1, 1, 3, nullptr, nullptr, tmp, s);
        return e ? e : run_cnv(r.c2, tmp, B, Hc, Wc, 1, 1, 0, 0, in, extra, out, s);
    };
    auto upsample = [&](const f16 *in, int Hi_, int Wi_, int Ho_, int Wo_, int C, f16 *out, const f16 *add = nullptr) -> int {
        ProfScope ps("da_upsample_kernel", s, 0.0, (double)B * Ho_ * Wo_ * C * 2.0 * 5.0);
        da_upsample_kernel<<<blocks((long)B * Ho_ * Wo_ * (C / 8)), 256, 0, s>>>(in, out, B, Hi_, Wi_, Ho_, Wo_, C, add);
        NUNIF_LAUNCH_CHECK();
        return NUNIF_HIP_OK;
    };
    // path4
    if ((rc = rcu(h->fus[3].r2, rn[3], H4, W4, nullptr, m1, m2))) return rc;
    // A refinenet ends with `interpolate(x, bilinear, align_corners=True)` then the 1x1 `out_conv` (DPT FeatureFusionBlock).  Both
    // are linear and the interpolation weights of a pixel sum to 1, so the 1x1 conv (with its bias) commutes with the resize: it
    // runs on the map BEFORE the resize, a quarter of the pixels (the largest of the four was a 351 k-pixel GEMM).  Same result
    // in real arithmetic; the fp16 rounding of the intermediate map moves (tests: unchanged PSNR against the oracle and the
    // HuggingFace fixture).  NUNIF_DA_OUTCONV_FIRST=0 keeps the reference's order.
    static const bool conv_first = !(getenv("NUNIF_DA_OUTCONV_FIRST") && atoi(getenv("NUNIF_DA_OUTCONV_FIRST")) == 0);
    if (conv_first) {
        if ((rc = run_lin(h->fus[3].out, m2, B, W4, W4, 0, 0, nullptr, m3, s, "da_out_conv", 0, 0, 1, H4))) return rc;
        // (rcu1_branch: m4 = path4 + RCU1(layer3_rn) right away)
        if ((rc = upsample(m3, H4, W4, H3, W3, F, m4, rcu1_branch ? (const f16 *)h->br1[2].p : nullptr))) return rc;       // path4 in m4
    } else {
        if ((rc = upsample(m2, H4, W4, H3, W3, F, m3))) return rc;
        if ((rc = run_lin(h->fus[3].out, m3, B, W3, W3, 0, 0, nullptr, m4, s, "da_out_conv", 0, 0, 1, H3))) return rc;      // path4 in m4
    }
    if (h->temporal && (rc = run_tmod(2, m4, s))) return rc;                     // motion_modules[2] on path_4
    // path3 .. path1
    const f16 *path = m4;
    f16 *pout = m5;
    for (int k = 2; k >= 0; --k) {
        const int Hc = Hs[k], Wc = Ws[k];
        const int Hn = k > 0 ? Hs[k - 1] : HF, Wn = k > 0 ? Ws[k - 1] : WF;
        const f16 *xin = path;                                                         // rcu1_branch: path + RCU1(skip) already
        if (!rcu1_branch) {
            if ((rc = rcu(h->fus[k].r1, rn[k], Hc, Wc, path, m1, m2))) return rc;    // m2 = path + RCU1(skip)
            xin = m2;
        }
        if ((rc = rcu(h->fus[k].r2, xin, Hc, Wc, nullptr, m1, m3))) return rc;       // m3 = RCU2(.)
        if (conv_first) {
            if ((rc = run_lin(h->fus[k].out, m3, B, Wc, Wc, 0, 0, nullptr, m1, s, "da_out_conv", 0, 0, 1, Hc))) return rc;
            if ((rc = upsample(m1, Hc, Wc, Hn, Wn, F, pout, rcu1_branch && k > 0 ? (const f16 *)h->br1[k - 1].p : nullptr))) return rc;
        } else {
            if ((rc = upsample(m3, Hc, Wc, Hn, Wn, F, m2))) return rc;
            if ((rc = run_lin(h->fus[k].out, m2, B, Wn, Wn, 0, 0, nullptr, pout, s, "da_out_conv", 0, 0, 1, Hn))) return rc;
        }
        if (h->temporal && k == 2 && (rc = run_tmod(3, pout, s))) return rc;      // motion_modules[3] on path_3
        path = pout;
        pout = (pout == m5) ? m4 : m5;
    }
    // output_conv1 (F -> F / 2) at 8x the patch grid, resize to the input size, output_conv2
    if ((rc = run_cnv(h->oc1, path, B, HF, WF, 1, 1, 0, 0, nullptr, nullptr, m1, s))) return rc;
    if ((rc = upsample(m1, HF, WF, hh, ww, F / 2, m2))) return rc;
    if ((rc = run_cnv(h->oc2, m2, B, hh, ww, 1, 1, 0, 3, nullptr, nullptr, m3, s))) return rc;
    {
        const long n = (long)B * hh * ww;
        ProfScope ps("da_final_kernel", s, 64.0 * n, (double)n * 68.0);
        da_final_kernel<<<blocks(n), 256, 0, s>>>(m3, h->w_final, depth, n, h->max_depth);
        NUNIF_LAUNCH_CHECK();
    }
    if (h->temporal) { h->t_len = t_len_end; h->t_start = t_start_end; }                    // the B frames joined the window
    return NUNIF_HIP_OK;
}

// `model.reset_state()` (iw3/video_depth_anything_streaming_model.py:74-75, at scene cuts): the next frame starts a new window.
// Host state only — the caches are overwritten slot by slot — so it is ordered with the forwards by the CALLER's program order.
extern "C" int nunif_hip_depth_anything_reset_state(nunif_depth_anything *h) {
    NUNIF_REQUIRE(h, "depth_anything_reset_state: NULL handle");
    h->t_len = 0;
    h->t_start = 0;
    return NUNIF_HIP_OK;
}

// 1: the checkpoint carried head.motion_modules (Video-Depth-Anything), 0: a per-frame Depth-Anything engine
extern "C" int nunif_hip_depth_anything_is_temporal(const nunif_depth_anything *h) { return h && h->temporal ? 1 : 0; }
