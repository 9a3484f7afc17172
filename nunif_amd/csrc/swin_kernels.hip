// Device kernels of the waifu2x swin_unet forward for gfx950 (CDNA4, wave64, MFMA).
//
// Reference ops replaced: waifu2x/models/swin_unet.py patch :132-137, PatchDown :45-62,
// PatchUp :65-82, ToImage :85-116, and torchvision SwinTransformerBlock (qkv/proj/MLP Linear, window attention).
//
// Layout: every feature map is NHWC fp16 in HBM; accumulation is fp32 in MFMA accumulators.
// All GEMM-shaped work runs on v_mfma_f32_16x16x32_f16 with the *weights* as the A operand (rows = output
// channel n) and the *activations* as the B operand (cols = token m).  The accumulator of a lane then holds 4
// consecutive output channels of ONE token (D[n = 4*(lane>>4)+r][m = lane&15]) so the epilogue stores 8 bytes of
// contiguous NHWC per lane and bias/residual are plain 4-wide loads — no LDS transpose anywhere.
#include <algorithm>
#include <cstdlib>

#include "swin_kernels.h"
#include "swin_gelu.h"

namespace nunif {

#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define MFMA_16x16x16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16f16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// =================================================================================================================
// Implicit-GEMM linear / conv.  One wave owns MF x 16 tokens and keeps their whole K extent in registers (the
// activations are the streaming operand: read once from HBM); it then sweeps all N/16 output-channel tiles.
// The packed weight array [nt][ks] is already the consumption order, so the 4 waves of a workgroup pull it through
// the same 2 x 8 KiB LDS ring as the block kernels (global -> registers one chunk ahead, registers -> LDS at the
// chunk boundary, one __syncthreads per chunk).  v1 loaded every fragment straight from L2 and exposed its latency.
// =================================================================================================================
template <int KS, int MF>
__global__ void __launch_bounds__(256) gemm_kernel(GemmArgs g) {
    constexpr int CH = 8;
    __shared__ __attribute__((aligned(16))) f16x8 ring[2][CH * 64];
    const int tid = threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r16 = lane & 15;
    const int grp = lane >> 4;
    const long M = (long)g.B * g.Ho * g.Wo;
    // g.rev: the token groups are walked from the last to the first ("snake" order between consecutive kernels: a kernel
    // starts on the lines its predecessor touched last, which the 256-MB memory-side cache still holds)
    const long m_base = ((long)(g.rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * 4 + wave) * (MF * 16);
    // no early exit: every wave joins every chunk barrier; rows beyond M are clamped on load and masked on store
    // (the host pads every packed weight array with 16 KiB of zeros, so the one-chunk-ahead prefetch never
    //  leaves the allocation — make_linear in swin_unet.cpp)
    // small-M GEMMs (ViT token matrices) split the output tiles over blockIdx.y so that the grid still fills the chip
    const int NT_all = g.N >> 4;
    const int nt_lo = g.nt_chunk ? (int)blockIdx.y * g.nt_chunk : 0;
    const int nt_hi = g.nt_chunk ? min(NT_all, nt_lo + g.nt_chunk) : NT_all;
    const f16x8 *gsrc = reinterpret_cast<const f16x8 *>(g.w) + (long)nt_lo * KS * 64;
    auto fetch = [&](int e) -> f16x8 { return gsrc[e]; };
    f16x8 st0 = fetch(tid), st1 = fetch(tid + 256);
    auto wfrag = [&](int fi) -> f16x8 {
        const int c = fi / CH;
        if (fi % CH == 0) {
            ring[c & 1][tid] = st0;
            ring[c & 1][tid + 256] = st1;
            __syncthreads();
            st0 = fetch((c + 1) * (CH * 64) + tid);
            st1 = fetch((c + 1) * (CH * 64) + tid + 256);
        }
        return ring[c & 1][(fi % CH) * 64 + lane];
    };

    f16x8 xf[MF][KS];
    int tb[MF], ty[MF], tx[MF];
    bool valid[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        long m = m_base + f * 16 + r16;
        valid[f] = m < M;
        if (m >= M) m = M - 1;
        // (a wave-uniform division + per-lane carry loops, and 32-bit division, were both tried: slower / faulting)
        const int x = (int)(m % g.Wo);
        const long t = m / g.Wo;
        const int y = (int)(t % g.Ho);
        const int b = (int)(t / g.Ho);
        tb[f] = b; ty[f] = y; tx[f] = x;
        const long pix0 = ((long)b * g.Hi + (long)y * g.stride + g.oy) * g.Wi + (long)x * g.stride + g.ox;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k0 = ks * 32;
            const int tap = k0 / g.Cin;
            const int c0 = k0 - tap * g.Cin;
            const int dy = tap / g.kw;
            const int dx = tap - dy * g.kw;
            const f16 *p = g.a + (pix0 + (long)dy * g.Wi + dx) * (g.lda ? g.lda : g.Cin) + c0 + grp * 8;
            xf[f][ks] = *reinterpret_cast<const f16x8 *>(p);
        }
    }
    if (g.in_scale) {
        // squeeze-excitation scale of the input map, applied to the fragments as they arrive: (f16)((float)x * s), the rounding
        // of the separate scale pass
#pragma unroll
        for (int f = 0; f < MF; ++f) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int c0 = (ks * 32) % g.Cin;
                const float4 *sp = reinterpret_cast<const float4 *>(g.in_scale + (long)tb[f] * g.Cin + c0 + grp * 8);
                const float4 s0 = sp[0], s1 = sp[1];
                const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) xf[f][ks][j] = (f16)((float)xf[f][ks][j] * sc[j]);
            }
        }
    }

    const int NT = nt_hi - nt_lo;
    if (g.mode == 2) {
        // ToImage: column n = c*s*s + i*s + j -> out[b][c][y*s+i][x*s+j], clamp(0,1)  (swin_unet.py:110-116)
#pragma unroll 1
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 acc[MF];
#pragma unroll
            for (int f = 0; f < MF; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const f16x8 wv = wfrag(nt * KS + ks);
#pragma unroll
                for (int f = 0; f < MF; ++f) acc[f] = MFMA_16x16x32(wv, xf[f][ks], acc[f]);
            }
            const int n0 = (nt_lo + nt) * 16 + grp * 4;
            const float4 bv = *reinterpret_cast<const float4 *>(g.bias + n0);
            const int s = g.ps, s2 = s * s;
            const int OC = g.n_real / s2;
            float *o = reinterpret_cast<float *>(g.out);
            const long OH = g.OH ? g.OH : (long)g.Ho * s, OW = g.OW ? g.OW : (long)g.Wo * s;
            if ((s == 4 || s == 8) && g.oshift == 0 && (OW & 3) == 0) {
                // Round 6 (the 4x nets' head: 560 us per launch at 2.0 TB/s, 7 % of a 4K frame, profiles/r06k_kernel_stats_4k.csv):
                // with s = 4 a lane's four columns n0 .. n0 + 3 are the four sub-pixels j of ONE output row (c = n0 / 16,
                // i = n0 / 4 % 4) — one 16-byte store, 256 contiguous bytes per lane group, and no division by a run-time s
                // (the general form below issued 8 integer divisions and 16 four-byte stores per output tile).
                // (s = 8, the 8x net: the four columns are sub-pixels j0 .. j0 + 3, j0 = n0 % 8, of row i = n0 / 8 % 8 of c = n0 / 64)
                const int ls = s == 4 ? 2 : 3;
                const int c = n0 >> (2 * ls), i = (n0 >> ls) & (s - 1), j0 = n0 & (s - 1);
                if (n0 < g.n_real) {
#pragma unroll
                    for (int f = 0; f < MF; ++f) {
                        if (!valid[f]) continue;
                        float4 v = {acc[f][0] + bv.x, acc[f][1] + bv.y, acc[f][2] + bv.z, acc[f][3] + bv.w};
                        if (!g.no_clamp) {
                            v.x = fminf(fmaxf(v.x, 0.f), 1.f); v.y = fminf(fmaxf(v.y, 0.f), 1.f);
                            v.z = fminf(fmaxf(v.z, 0.f), 1.f); v.w = fminf(fmaxf(v.w, 0.f), 1.f);
                        }
                        const long oy = (long)ty[f] * s + i, ox = (long)tx[f] * s + j0;
                        if (oy < OH && ox + 3 < OW) *reinterpret_cast<float4 *>(o + (((long)tb[f] * OC + c) * OH + oy) * OW + ox) = v;
                    }
                }
                continue;
            }
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                if (!valid[f]) continue;
                const float v[4] = {acc[f][0] + bv.x, acc[f][1] + bv.y, acc[f][2] + bv.z, acc[f][3] + bv.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + r;
                    if (n < g.n_real) {
                        const int c = n / s2, rem = n - c * s2;
                        const int i = rem / s, j = rem - i * s;
                        const long oy = (long)ty[f] * s + i + g.oshift, ox = (long)tx[f] * s + j + g.oshift;
                        if (oy >= 0 && oy < OH && ox >= 0 && ox < OW)
                            o[(((long)tb[f] * OC + c) * OH + oy) * OW + ox] = g.no_clamp ? v[r] : fminf(fmaxf(v[r], 0.f), 1.f);
                    }
                }
            }
        }
        return;
    }
    // NHWC fp16 outputs: two adjacent 16-channel tiles per trip, so that a lane owns a run of 8 consecutive channels
    // after pair_to_run() (common.h) and the residual read / store are 16 B per lane, 64 B per pixel row.
    auto out_off = [&](int f, int np) -> long {
        long off;
        if (g.mode == 0) {
            off = (((long)tb[f] * g.Ho + ty[f]) * g.Wo + tx[f]) * g.ldo + np;
        } else {
            // PatchUp: column n = q*Cq + c (repacked), q=(i,j) -> pixel (2y+i, 2x+j)  (swin_unet.py:76-82);
            // Cq is a multiple of 32, so a tile pair never straddles two sub-pixels
            const int q = np / g.ldo, c = np - q * g.ldo;
            const int sf = g.ps > 1 ? g.ps : 2;                       // ConvTranspose2d(k = stride = sf): q = i*sf + j
            const int qi = q / sf, qj = q - qi * sf;
            off = (((long)tb[f] * (sf * g.Ho) + sf * ty[f] + qi) * (sf * g.Wo) + sf * tx[f] + qj) * g.ldo + c;
        }
        return off + pair_run_channel(grp);
    };
    // residual position: out_off, or (res_W > 0, mode 1) the same output pixel shifted by res_crop inside a larger map
    auto res_off = [&](int f, int np) -> long {
        if (g.res_W == 0) return out_off(f, np);
        const int q = np / g.ldo, c = np - q * g.ldo;
        const int sf = g.ps > 1 ? g.ps : 2;
        const int qi = q / sf, qj = q - qi * sf;
        return (((long)tb[f] * g.res_H + sf * ty[f] + qi + g.res_crop) * g.res_W + sf * tx[f] + qj + g.res_crop) * g.ldo + c +
               pair_run_channel(grp);
    };
    // The residual of a tile pair is requested at the TOP of its trip, in front of the trip's MFMAs: the chunk barriers of the
    // weight ring otherwise pin the load right in front of its use in the epilogue, and with the U-Net skip as residual (PatchUp)
    // every trip then exposed one HBM latency.  (Requesting it a whole trip ahead costs 16 more registers and the third resident
    // workgroup per CU: 312 vs 265 us on the two PatchUps, measured.)
#pragma unroll 1
    for (int nt = 0; nt < NT; nt += 2) {
        f32x4 acc0[MF], acc1[MF];
        f16x8 rcur[MF];
        if (g.res) {
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                const int npn = (nt_lo + nt) * 16;
                const bool live = valid[f] && npn < g.n_real;
                rcur[f] = *reinterpret_cast<const f16x8 *>(g.res + (live ? res_off(f, npn) : 0));
            }
        }
#pragma unroll
        for (int f = 0; f < MF; ++f) { acc0[f] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[f] = acc0[f]; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f16x8 wv = wfrag(nt * KS + ks);
#pragma unroll
            for (int f = 0; f < MF; ++f) acc0[f] = MFMA_16x16x32(wv, xf[f][ks], acc0[f]);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f16x8 wv = wfrag((nt + 1) * KS + ks);
#pragma unroll
            for (int f = 0; f < MF; ++f) acc1[f] = MFMA_16x16x32(wv, xf[f][ks], acc1[f]);
        }
        const int n0 = (nt_lo + nt) * 16 + grp * 4;
        const float4 bv0 = *reinterpret_cast<const float4 *>(g.bias + n0);
        const float4 bv1 = *reinterpret_cast<const float4 *>(g.bias + n0 + 16);
        const int np = (nt_lo + nt) * 16;               // first channel of the 32-channel pair
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            float v0[4] = {acc0[f][0] + bv0.x, acc0[f][1] + bv0.y, acc0[f][2] + bv0.z, acc0[f][3] + bv0.w};
            float v1[4] = {acc1[f][0] + bv1.x, acc1[f][1] + bv1.y, acc1[f][2] + bv1.z, acc1[f][3] + bv1.w};
            if (g.act == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { v0[r] = gelu_erf(v0[r]); v1[r] = gelu_erf(v1[r]); }
            } else if (g.act == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v0[r] = v0[r] >= 0.f ? v0[r] : v0[r] * g.slope;
                    v1[r] = v1[r] >= 0.f ? v1[r] : v1[r] * g.slope;
                }
            }
            const long off = out_off(f, np);
            const bool live = valid[f] && np < g.n_real;
            if (g.res) {
                f16x4 ra, rb;
                run_to_pair(rcur[f], ra, rb);
#pragma unroll
                for (int r = 0; r < 4; ++r) { v0[r] += (float)ra[r]; v1[r] += (float)rb[r]; }
            }
            const f16x8 ov = pair_to_run((f16x4){(f16)v0[0], (f16)v0[1], (f16)v0[2], (f16)v0[3]},
                                         (f16x4){(f16)v1[0], (f16)v1[1], (f16)v1[2], (f16)v1[3]});
            if (live) *reinterpret_cast<f16x8 *>(reinterpret_cast<f16 *>(g.out) + off) = ov;
        }
    }
}

// gemm_res_kernel: the same GEMM with the weights RESIDENT in LDS: a persistent 8-wave workgroup per CU copies the WHOLE packed weight matrix into LDS once (launcher
//   checks it fits in 160 KB) and every wave loops over its own token groups with no barrier.  With few MFMAs per
//   fragment (MF = 2) the ring's one-chunk-ahead weight prefetch covers only ~256 cycles, less than the L2 latency, and
//   every chunk boundary stalled; measured on the stem conv (K = 576): 770 us with the ring.
template <int KS, int MF>
__global__ void __launch_bounds__(512) gemm_res_kernel(GemmArgs g) {
    constexpr bool RES = true;
    constexpr int CH = 8;
    constexpr int WAVES = RES ? 8 : 4;
    __shared__ __attribute__((aligned(16))) f16x8 ring[RES ? 1 : 2][RES ? 1 : CH * 64];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_g[];
    const f16x8 *wres = reinterpret_cast<const f16x8 *>(smem_g);
    const int tid = threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r16 = lane & 15;
    const int grp = lane >> 4;
    const long M = (long)g.B * g.Ho * g.Wo;
    const int NT = g.N >> 4;
    // ring mode: no early exit — every wave joins every chunk barrier; rows beyond M are clamped on load and masked on
    // store (the host pads every packed weight array with 16 KiB of zeros, so the one-chunk-ahead prefetch never
    //  leaves the allocation — make_linear in swin_unet.cpp)
    const f16x8 *gsrc = reinterpret_cast<const f16x8 *>(g.w);
    f16x8 st0, st1;
    if constexpr (RES) {
        f16x8 *dst = reinterpret_cast<f16x8 *>(smem_g);
        for (int i = tid; i < NT * KS * 64; i += 512) dst[i] = gsrc[i];
        __syncthreads();
    } else {
        st0 = gsrc[tid]; st1 = gsrc[tid + 256];
    }
    auto wfrag = [&](int fi) -> f16x8 {
        if constexpr (RES) {
            return wres[fi * 64 + lane];
        } else {
            const int c = fi / CH;
            if (fi % CH == 0) {
                ring[c & 1][tid] = st0;
                ring[c & 1][tid + 256] = st1;
                __syncthreads();
                st0 = gsrc[(c + 1) * (CH * 64) + tid];
                st1 = gsrc[(c + 1) * (CH * 64) + tid + 256];
            }
            return ring[c & 1][(fi % CH) * 64 + lane];
        }
    };

    const long n_groups = (M + MF * 16 - 1) / (MF * 16);
    const long g_first = (long)blockIdx.x * WAVES + wave;
    const long g_step = RES ? (long)gridDim.x * WAVES : n_groups + WAVES;      // ring mode: exactly one trip
#pragma unroll 1
    for (long gi = g_first; RES ? gi < n_groups : gi == g_first; gi += g_step) {
    const long m_base = (g.rev ? n_groups - 1 - gi : gi) * (MF * 16);
    f16x8 xf[MF][KS];
    int tb[MF], ty[MF], tx[MF];
    bool valid[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        long m = m_base + f * 16 + r16;
        valid[f] = m < M;
        if (m >= M) m = M - 1;
        // (a wave-uniform division + per-lane carry loops, and 32-bit division, were both tried: slower / faulting)
        const int x = (int)(m % g.Wo);
        const long t = m / g.Wo;
        const int y = (int)(t % g.Ho);
        const int b = (int)(t / g.Ho);
        tb[f] = b; ty[f] = y; tx[f] = x;
        const long pix0 = ((long)b * g.Hi + (long)y * g.stride + g.oy) * g.Wi + (long)x * g.stride + g.ox;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k0 = ks * 32;
            const int tap = k0 / g.Cin;
            const int c0 = k0 - tap * g.Cin;
            const int dy = tap / g.kw;
            const int dx = tap - dy * g.kw;
            const f16 *p = g.a + (pix0 + (long)dy * g.Wi + dx) * (g.lda ? g.lda : g.Cin) + c0 + grp * 8;
            xf[f][ks] = *reinterpret_cast<const f16x8 *>(p);
        }
    }

    if (g.mode == 2) {
        // ToImage: column n = c*s*s + i*s + j -> out[b][c][y*s+i][x*s+j], clamp(0,1)  (swin_unet.py:110-116)
#pragma unroll 1
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 acc[MF];
#pragma unroll
            for (int f = 0; f < MF; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const f16x8 wv = wfrag(nt * KS + ks);
#pragma unroll
                for (int f = 0; f < MF; ++f) acc[f] = MFMA_16x16x32(wv, xf[f][ks], acc[f]);
            }
            const int n0 = nt * 16 + grp * 4;
            const float4 bv = *reinterpret_cast<const float4 *>(g.bias + n0);
            const int s = g.ps, s2 = s * s;
            const int OC = g.n_real / s2;
            float *o = reinterpret_cast<float *>(g.out);
            const long OH = g.OH ? g.OH : (long)g.Ho * s, OW = g.OW ? g.OW : (long)g.Wo * s;
            if ((s == 4 || s == 8) && g.oshift == 0 && (OW & 3) == 0) {          // as in gemm_kernel: one 16-byte store per lane and tile
                const int ls = s == 4 ? 2 : 3;
                const int c = n0 >> (2 * ls), i = (n0 >> ls) & (s - 1), j0 = n0 & (s - 1);
                if (n0 < g.n_real) {
#pragma unroll
                    for (int f = 0; f < MF; ++f) {
                        if (!valid[f]) continue;
                        float4 v = {acc[f][0] + bv.x, acc[f][1] + bv.y, acc[f][2] + bv.z, acc[f][3] + bv.w};
                        if (!g.no_clamp) {
                            v.x = fminf(fmaxf(v.x, 0.f), 1.f); v.y = fminf(fmaxf(v.y, 0.f), 1.f);
                            v.z = fminf(fmaxf(v.z, 0.f), 1.f); v.w = fminf(fmaxf(v.w, 0.f), 1.f);
                        }
                        const long oy = (long)ty[f] * s + i, ox = (long)tx[f] * s + j0;
                        if (oy < OH && ox + 3 < OW) *reinterpret_cast<float4 *>(o + (((long)tb[f] * OC + c) * OH + oy) * OW + ox) = v;
                    }
                }
                continue;
            }
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                if (!valid[f]) continue;
                const float v[4] = {acc[f][0] + bv.x, acc[f][1] + bv.y, acc[f][2] + bv.z, acc[f][3] + bv.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + r;
                    if (n < g.n_real) {
                        const int c = n / s2, rem = n - c * s2;
                        const int i = rem / s, j = rem - i * s;
                        const long oy = (long)ty[f] * s + i + g.oshift, ox = (long)tx[f] * s + j + g.oshift;
                        if (oy >= 0 && oy < OH && ox >= 0 && ox < OW)
                            o[(((long)tb[f] * OC + c) * OH + oy) * OW + ox] = g.no_clamp ? v[r] : fminf(fmaxf(v[r], 0.f), 1.f);
                    }
                }
            }
        }
        continue;
    }
    // NHWC fp16 outputs: two adjacent 16-channel tiles per trip, so that a lane owns a run of 8 consecutive channels
    // after pair_to_run() (common.h) and the residual read / store are 16 B per lane, 64 B per pixel row.
#pragma unroll 1
    for (int nt = 0; nt < NT; nt += 2) {
        f32x4 acc0[MF], acc1[MF];
#pragma unroll
        for (int f = 0; f < MF; ++f) { acc0[f] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[f] = acc0[f]; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f16x8 wv = wfrag(nt * KS + ks);
#pragma unroll
            for (int f = 0; f < MF; ++f) acc0[f] = MFMA_16x16x32(wv, xf[f][ks], acc0[f]);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f16x8 wv = wfrag((nt + 1) * KS + ks);
#pragma unroll
            for (int f = 0; f < MF; ++f) acc1[f] = MFMA_16x16x32(wv, xf[f][ks], acc1[f]);
        }
        const int n0 = nt * 16 + grp * 4;
        const float4 bv0 = *reinterpret_cast<const float4 *>(g.bias + n0);
        const float4 bv1 = *reinterpret_cast<const float4 *>(g.bias + n0 + 16);
        const int np = nt * 16;                         // first channel of the 32-channel pair
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            float v0[4] = {acc0[f][0] + bv0.x, acc0[f][1] + bv0.y, acc0[f][2] + bv0.z, acc0[f][3] + bv0.w};
            float v1[4] = {acc1[f][0] + bv1.x, acc1[f][1] + bv1.y, acc1[f][2] + bv1.z, acc1[f][3] + bv1.w};
            if (g.act == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { v0[r] = gelu_erf(v0[r]); v1[r] = gelu_erf(v1[r]); }
            } else if (g.act == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v0[r] = v0[r] >= 0.f ? v0[r] : v0[r] * g.slope;
                    v1[r] = v1[r] >= 0.f ? v1[r] : v1[r] * g.slope;
                }
            }
            long off;
            if (g.mode == 0) {
                off = (((long)tb[f] * g.Ho + ty[f]) * g.Wo + tx[f]) * g.ldo + np;
            } else {
                // PatchUp: column n = q*Cq + c (repacked), q=(i,j) -> pixel (2y+i, 2x+j)  (swin_unet.py:76-82);
                // Cq is a multiple of 32, so a tile pair never straddles two sub-pixels
                const int q = np / g.ldo, c = np - q * g.ldo;
                const int sf = g.ps > 1 ? g.ps : 2;                       // ConvTranspose2d(k = stride = sf): q = i*sf + j
                const int qi = q / sf, qj = q - qi * sf;
                off = (((long)tb[f] * (sf * g.Ho) + sf * ty[f] + qi) * (sf * g.Wo) + sf * tx[f] + qj) * g.ldo + c;
            }
            off += pair_run_channel(grp);
            const bool live = valid[f] && np < g.n_real;
            if (g.res) {
                f16x4 ra, rb;
                run_to_pair(*reinterpret_cast<const f16x8 *>(g.res + (live ? off : 0)), ra, rb);
#pragma unroll
                for (int r = 0; r < 4; ++r) { v0[r] += (float)ra[r]; v1[r] += (float)rb[r]; }
            }
            const f16x8 ov = pair_to_run((f16x4){(f16)v0[0], (f16)v0[1], (f16)v0[2], (f16)v0[3]},
                                         (f16x4){(f16)v1[0], (f16)v1[1], (f16)v1[2], (f16)v1[3]});
            if (live) *reinterpret_cast<f16x8 *>(reinterpret_cast<f16 *>(g.out) + off) = ov;
        }
    }
    }   // token-group loop
}

static thread_local const char *g_prof_tag = nullptr;

// token count from which the plain K = 192 Linears take the resident-weight form (NUNIF_GEMM_BIG_M: tests lower it to cover that
// path at small sizes; read per launch)
long gemm_big_m() { const char *e = getenv("NUNIF_GEMM_BIG_M"); return e ? atol(e) : (1L << 20); }

template <int KS, int MF>
static int launch_gemm_t(const GemmArgs &g, hipStream_t s, const char *ring_sym, const char *res_sym, double flops,
                         double bytes, bool ring_only = false) {
    const long M = (long)g.B * g.Ho * g.Wo;
    const size_t wbytes = (size_t)(g.N / 16) * KS * 1024;
    // resident weights pay where the ring's one-chunk-ahead prefetch is too short (MF = 2: K = 384, 576); measured
    // slower for the MF = 4 shapes (K = 96, 192), which keep the ring (re-checked at the end of round 2: PatchDown through the
    // ring 134 us, resident 112 us).
    const bool fits = wbytes <= 144 * 1024 && M >= 8 * MF * 16 * 64;
    // ... and for the plain K = 192 Linears over millions of tokens (the inpaint net's proj_out at 4K, config 5: 3.4 M tokens x
    // (384 B in + 192 B out)): the ring form is one 256-token tile per workgroup and all prologue, 1 040 us = 1.9 TB/s; resident
    // 543 us = 3.2 TB/s.  K = 96 measured equal (stays on the ring).
    // (round 6: and the image head of the 4x / 8x nets, mode 2 — NUNIF_GEMM_RES_IMAGE=0 keeps it on the ring for A/B runs)
    static const bool res_image = !(getenv("NUNIF_GEMM_RES_IMAGE") && atoi(getenv("NUNIF_GEMM_RES_IMAGE")) == 0);
    const bool big_plain = MF == 4 && KS == 6 && (g.mode == 0 || (g.mode == 2 && res_image)) && M >= gemm_big_m();
    const bool res = fits && !ring_only && (MF == 2 || big_plain) && g.res_W == 0 && !g.in_scale;     // (the cropped residual and the input scale exist in the ring form only)
    // profiler classes are named after the kernel symbol so that they line up with rocprofv3's kernel stats
    // (NUNIF_PROF_TAGS=1 names the class after the call site instead: separates e.g. the two gemm_kernel<6,4> users)
    ProfScope ps(g_prof_tag ? g_prof_tag : res ? res_sym : ring_sym, s, flops, bytes);
    if (res) {
        static bool configured = false;
        if (!configured) {
            NUNIF_HIP_CHECK(hipFuncSetAttribute((const void *)gemm_res_kernel<KS, MF>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
            configured = true;
        }
        const long groups = (M + MF * 16 - 1) / (MF * 16);
        const unsigned blocks = (unsigned)std::min<long>((groups + 7) / 8, 256);
        gemm_res_kernel<KS, MF><<<blocks, 512, wbytes, s>>>(g);
    } else {
        const long rows_per_block = 4 * MF * 16;
        const unsigned blocks = (unsigned)((M + rows_per_block - 1) / rows_per_block);
        GemmArgs gg = g;
        unsigned ny = 1;
        const int NT = g.N / 16;
        if (blocks < 128 && NT >= 4 && g.mode != 2) {       // few token groups: also split the output tiles (even chunks)
            const unsigned want = (512 + blocks - 1) / blocks;
            int chunk = std::max(2, (int)((NT + want - 1) / want));
            chunk += chunk & 1;
            ny = (unsigned)((NT + chunk - 1) / chunk);
            gg.nt_chunk = ny > 1 ? chunk : 0;
            if (ny <= 1) ny = 1;
        }
        gemm_kernel<KS, MF><<<dim3(blocks, ny), 256, 0, s>>>(gg);
    }
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

int launch_gemm(const GemmArgs &g, hipStream_t s, const char *tag) {
    NUNIF_REQUIRE(g.K % 32 == 0 && g.N % 16 == 0 && g.Cin % 32 == 0, "gemm %s: K=%d N=%d Cin=%d not aligned", tag,
                  g.K, g.N, g.Cin);
    NUNIF_REQUIRE(g.mode == 2 || (g.N % 32 == 0 && g.n_real % 32 == 0 && g.ldo % 32 == 0),
                  "gemm %s: NHWC outputs are written in 32-channel pairs (N=%d n_real=%d ldo=%d)", tag, g.N, g.n_real, g.ldo);
    const long M = (long)g.B * g.Ho * g.Wo;
    if (M == 0) return NUNIF_HIP_OK;
    g_prof_tag = profile_tags_enabled() ? tag : nullptr;
    const double flops = 2.0 * (double)M * g.K * g.n_real;
    const double bytes = (double)M * (g.Cin * 2.0 * (g.K / g.Cin > 1 ? 1.0 : 1.0) + g.n_real * (g.mode == 2 ? 4.0 : 2.0) +
                                      (g.res ? g.n_real * 2.0 : 0.0));
    switch (g.K / 32) {
        case 1: return launch_gemm_t<1, 4>(g, s, "gemm_kernel<1,4>", "gemm_res_kernel<1,4>", flops, bytes);
        case 2: return launch_gemm_t<2, 4>(g, s, "gemm_kernel<2,4>", "gemm_res_kernel<2,4>", flops, bytes);
        case 4: return launch_gemm_t<4, 4>(g, s, "gemm_kernel<4,4>", "gemm_res_kernel<4,4>", flops, bytes);
        case 3: return launch_gemm_t<3, 4>(g, s, "gemm_kernel<3,4>", "gemm_res_kernel<3,4>", flops, bytes);
        case 6: {
            // PatchUp (mode 1 with the skip as residual) is HBM-bound and exposes the latency of its residual reads: with 2 token
            // tiles per wave the kernel needs 156 instead of 252 registers, so 3 workgroups per CU are resident and hide it
            // (measured, both PatchUps of the 2x net: 604 -> 529 us; the resident-weight form of the same shape was slower: 381 vs
            // 340 us on PatchUp 1).
            if (g.mode == 1 && g.res)
                return launch_gemm_t<6, 2>(g, s, "gemm_kernel<6,2>", "gemm_res_kernel<6,2>", flops, bytes, true);
            return launch_gemm_t<6, 4>(g, s, "gemm_kernel<6,4>", "gemm_res_kernel<6,4>", flops, bytes);
        }
        case 8: return launch_gemm_t<8, 4>(g, s, "gemm_kernel<8,4>", "gemm_res_kernel<8,4>", flops, bytes);
        case 12: return launch_gemm_t<12, 2>(g, s, "gemm_kernel<12,2>", "gemm_res_kernel<12,2>", flops, bytes);
        case 16: return launch_gemm_t<16, 2>(g, s, "gemm_kernel<16,2>", "gemm_res_kernel<16,2>", flops, bytes);
        case 18: return launch_gemm_t<18, 2>(g, s, "gemm_kernel<18,2>", "gemm_res_kernel<18,2>", flops, bytes);
        case 19: return launch_gemm_t<19, 1>(g, s, "gemm_kernel<19,1>", "gemm_res_kernel<19,1>", flops, bytes);
        case 24: return launch_gemm_t<24, 1>(g, s, "gemm_kernel<24,1>", "gemm_res_kernel<24,1>", flops, bytes);
        case 27: return launch_gemm_t<27, 1>(g, s, "gemm_kernel<27,1>", "gemm_res_kernel<27,1>", flops, bytes);
        case 32: return launch_gemm_t<32, 1>(g, s, "gemm_kernel<32,1>", "gemm_res_kernel<32,1>", flops, bytes);
        default:
            set_error("gemm %s: unsupported K=%d", tag, g.K);
            return NUNIF_HIP_EUNSUPPORTED;
    }
}

// =================================================================================================================
// Output-stationary Linear (swin_kernels.h GemmOsArgs).  Workgroup = 4 waves = MT x 16 tokens x 128 channels; wave w owns channel
// tiles 2w, 2w + 1 of the workgroup's 8 and all MT token tiles: acc[MT][2].  The K loop runs in groups of PF k-steps over PF
// register buffers (refilled PF k-steps ahead, right behind the MFMAs that read them) for the weights (direct loads, static indices) and an LDS image of the group's activations (shared by the
// four waves: once through the vector memory pipe instead of four times).
//
// These are SHORT-K products (the depth ViT: 5 492 tokens, K = 384 or 1 536): a workgroup lives for a few thousand cycles, and
// round 2's form (PF = 4: weights three k-steps = ~200 MFMA-cycles ahead, the residual read in the epilogue, libm erff for the
// GELU) was a chain of exposed memory latencies — SQ counters of the ViT-S run: 53 % of all wave-cycles in s_waitcnt, 10 VALU
// per MFMA, 2.05 waves per SIMD on average.  Now: PF = 12 (8 where K is not a multiple of 384) puts a whole K = 384 product's
// loads in flight at once — one latency per workgroup; the residual tile is fetched with them; every load of the loop is
// unconditional and the barrier is lgkmcnt-only (a conditional load or __syncthreads() makes hipcc drain vmcnt, which
// serialises the prefetch); the last group is peeled so that nothing is fetched twice; GELU is swin_gelu.h's polynomial.
// =================================================================================================================
template <int MT, int PF, int NB>                   // NB = 2 LDS buffers; 1 when the whole K is one group (KS == PF)
__global__ void __launch_bounds__(256) gemm_os_kernel(GemmOsArgs g) {
    constexpr int NTW = 2, SW = PF / 4;              // a wave stages k-steps w, w + 4, .. of every group
    static_assert(PF % 4 == 0, "four waves share the staging of a group");
    __shared__ __attribute__((aligned(16))) f16x8 act[NB][PF][MT][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, grp = lane >> 4;
    const int KS = g.K >> 5;
    const int n_blocks = g.N >> 7;
    const unsigned bid = xcd_contiguous(blockIdx.x, gridDim.x);  // the n_blocks channel blocks of a token block on ONE XCD's L2
    const int nb = bid % n_blocks;
    const long mb = bid / n_blocks;
    const long m0 = mb * (MT * 16);
    const int nt0 = nb * 8 + wave * NTW;                          // first 16-channel tile of this wave
    const int np = nt0 * 16;                                      // the wave's two tiles = one 32-channel pair
    const f16 *arow[MT];
    f16x8 rres[MT];
    bool live[MT];
    long ooff[MT];
#pragma unroll
    for (int f = 0; f < MT; ++f) {
        const long m = m0 + f * 16 + r16;
        live[f] = m < g.M;
        const long mc = live[f] ? m : g.M - 1;
        arow[f] = g.a + mc * g.lda + grp * 8 + wave * 32;
        ooff[f] = mc * g.ldo + np + pair_run_channel(grp);
        if (g.res) rres[f] = *reinterpret_cast<const f16x8 *>(g.res + ooff[f]);
    }
    const f16x8 *wbase = reinterpret_cast<const f16x8 *>(g.w) + (long)nt0 * KS * 64 + lane;
    f16x8 st[SW][MT], aq[PF][NTW];
    auto load_act = [&](int k0) {
#pragma unroll
        for (int i = 0; i < SW; ++i)
#pragma unroll
            for (int f = 0; f < MT; ++f) st[i][f] = *reinterpret_cast<const f16x8 *>(arow[f] + (k0 + 4 * i) * 32);
    };
    auto store_act = [&](int buf) {
#pragma unroll
        for (int i = 0; i < SW; ++i)
#pragma unroll
            for (int f = 0; f < MT; ++f) act[buf][wave + 4 * i][f][lane] = st[i][f];
    };
    auto load_w = [&](int ks, int buf) {
#pragma unroll
        for (int n = 0; n < NTW; ++n) aq[buf][n] = wbase[((long)n * KS + ks) * 64];
    };
    f32x4 acc[MT][NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const float4 bv = *reinterpret_cast<const float4 *>(g.bias + (nt0 + n) * 16 + grp * 4);
#pragma unroll
        for (int f = 0; f < MT; ++f) acc[f][n] = (f32x4){bv.x, bv.y, bv.z, bv.w};
    }
    load_act(0);
#pragma unroll
    for (int j = 0; j < PF; ++j) load_w(j, j);
    store_act(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int buf = 0;
    auto mfma_step = [&](int j) {
        f16x8 bq[MT];
#pragma unroll
        for (int f = 0; f < MT; ++f) bq[f] = act[NB == 1 ? 0 : buf][j][f][lane];
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int f = 0; f < MT; ++f) acc[f][n] = MFMA_16x16x32(aq[j][n], bq[f], acc[f][n]);
    };
    if constexpr (NB == 2) {
#pragma unroll 1
        for (int k0 = 0; k0 + PF < KS; k0 += PF) {               // every group but the last: the next group is fetched behind it
            load_act(k0 + PF);
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                mfma_step(j);
                load_w(k0 + PF + j, j);                           // buffer j is free again: PF k-steps ahead
            }
            store_act(buf ^ 1);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            buf ^= 1;
        }
    }
#pragma unroll
    for (int j = 0; j < PF; ++j) mfma_step(j);
#pragma unroll
    for (int f = 0; f < MT; ++f) {
        f16x4 o0, o1;
        if (g.act == 1) {
            const f16x8 h = gelu8(acc[f][0], acc[f][1]);
            o0 = (f16x4){h[0], h[1], h[2], h[3]};
            o1 = (f16x4){h[4], h[5], h[6], h[7]};
            if (g.res) {                                          // (no caller today: GELU is fc1, the residual proj / fc2)
                f16x4 ra, rb;
                run_to_pair(rres[f], ra, rb);
#pragma unroll
                for (int r = 0; r < 4; ++r) { o0[r] = (f16)((float)o0[r] + (float)ra[r]); o1[r] = (f16)((float)o1[r] + (float)rb[r]); }
            }
        } else {
            float v0[4] = {acc[f][0][0], acc[f][0][1], acc[f][0][2], acc[f][0][3]};
            float v1[4] = {acc[f][1][0], acc[f][1][1], acc[f][1][2], acc[f][1][3]};
            if (g.res) {
                f16x4 ra, rb;
                run_to_pair(rres[f], ra, rb);
#pragma unroll
                for (int r = 0; r < 4; ++r) { v0[r] += (float)ra[r]; v1[r] += (float)rb[r]; }
            }
            o0 = (f16x4){(f16)v0[0], (f16)v0[1], (f16)v0[2], (f16)v0[3]};
            o1 = (f16x4){(f16)v1[0], (f16)v1[1], (f16)v1[2], (f16)v1[3]};
        }
        const f16x8 ov = pair_to_run(o0, o1);
        if (live[f]) *reinterpret_cast<f16x8 *>(g.out + ooff[f]) = ov;
        if (g.stats_out) {
            // LayerNorm statistics of the NEXT norm, from the values as stored: this lane holds 8 of its token's channels, the four
            // lane groups the wave's 32 — one (sum, sum of squares) partial per token and 32-channel pair
            float su = 0.f, sq = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float v = (float)ov[e]; su += v; sq += v * v; }
            su += __shfl_xor(su, 16); sq += __shfl_xor(sq, 16);
            su += __shfl_xor(su, 32); sq += __shfl_xor(sq, 32);
            if (grp == 0 && live[f]) g.stats_out[(m0 + f * 16 + r16) * (g.N >> 5) + (nt0 >> 1)] = make_float2(su, sq);
        }
    }
}

bool gemm_os_supported(long M, int N, int K) {
    return M > 0 && N % 128 == 0 && K % 128 == 0 && K >= 128;
}

template <int MT, int PF>
static void launch_gemm_os_t(const GemmOsArgs &g, hipStream_t s) {
    const unsigned blocks = (unsigned)(((g.M + MT * 16 - 1) / (MT * 16)) * (g.N / 128));
    if (g.K == PF * 32) gemm_os_kernel<MT, PF, 1><<<blocks, 256, 0, s>>>(g);
    else gemm_os_kernel<MT, PF, 2><<<blocks, 256, 0, s>>>(g);
}

// ---- the same Linear with the weights STATIONARY in registers (K = 384, two channel tiles per wave, no residual) ----------------
// A short-K product is all prologue and epilogue when every 32 x 128 output tile is its own workgroup (gemm_os_kernel above: 2 064
// workgroups that each fetch 96 KiB of weights to run 48 MFMAs per wave).  Here a workgroup keeps its 128 channels'
// weights in registers (KS x NTW fragments per wave) for its whole life and walks the token blocks g, g + G, ..: per tile it
// fetches only the 32 tokens' rows, by LDS-DMA (global_load_lds_dwordx4: no staging registers) into a ring of three LDS images,
// TWO TILES AHEAD of the MFMAs.  The loop has no compiler-visible vector load at all, so the only vmcnt wait in it is the
// hand-counted one: `vmcnt(MT * SW)` behind the MFMAs = everything but the newest tile's DMAs has landed (loads return in order;
// the stores of the previous tile, which may retire in any order, are a tile old by then and are simply waited for as well).
// The grid is sized to what is RESIDENT (occupancy x CUs): a persistent kernel with a few workgroups too many runs a second pass.
template <int KS, int NTW, int MT, bool GELU, bool LNF>          // LNF: rows normalised through GemmOsArgs::stats_in (12 partials per token)
__global__ void __launch_bounds__(256) gemm_ws_kernel(GemmOsArgs g, int tiles, int G) {
    constexpr int SW = KS / 4;                                    // a wave stages k-steps w, w + 4, ..
    static_assert(KS % 4 == 0 && NTW == 2 && MT * SW == 6, "vmcnt(6 / 7) below; the epilogue writes 32-channel pairs");
    constexpr int kParts = 12, kStatTile = MT * 16 * kParts;      // float2 per tile: 3 KiB, moved by one more DMA per wave
    static_assert(!LNF || kStatTile * 8 == 3 * 1024, "the statistics tile is three 1-KiB DMAs (+ a fourth, discarded, so that every wave issues one)");
    __shared__ __attribute__((aligned(16))) f16x8 act[3][KS][MT][64];
    __shared__ __attribute__((aligned(16))) float2 stl[LNF ? 2 : 1][LNF ? 3 * 128 : 1];      // 78 KiB in all: two workgroups per CU
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, grp = lane >> 4;
    const int n_blocks = g.N / (64 * NTW);
    const unsigned bid = xcd_contiguous(blockIdx.x, gridDim.x);  // neighbours in `bid` share their token blocks: same XCD, same L2
    const int nb = bid % n_blocks;
    const int g0 = bid / n_blocks;
    const int nt0 = nb * 4 * NTW + wave * NTW;
    const int ocol = nt0 * 16 + pair_run_channel(grp);
    const int acol = grp * 8 + wave * 32;
    f16x8 aq[KS][NTW];
    {
        const f16x8 *wbase = reinterpret_cast<const f16x8 *>(g.w) + (long)nt0 * KS * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int n = 0; n < NTW; ++n) aq[ks][n] = wbase[((long)n * KS + ks) * 64];
    }
    float4 bv[NTW], wsv[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        bv[n] = *reinterpret_cast<const float4 *>(g.bias + (nt0 + n) * 16 + grp * 4);
        if constexpr (LNF) wsv[n] = *reinterpret_cast<const float4 *>(g.wsum + (nt0 + n) * 16 + grp * 4);
    }
    // a tile's statistics: 32 tokens x 12 partials = 3 072 contiguous bytes (the buffer is padded to whole tiles), one DMA per wave
    // (wave 3 repeats wave 2's piece onto the same KiB, so that every wave's vmcnt sees the same number of operations).  They travel
    // ONE tile ahead in a ring of two: behind the next trip's vmcnt wait they are older than that trip's seven DMAs.
    auto dma_stats = [&](int t, int sslot) {
        const int tc = t < tiles ? t : tiles - 1, piece = wave < 3 ? wave : 2;
        const char *src = reinterpret_cast<const char *>(g.stats_in) + (long)tc * (kStatTile * 8) + piece * 1024 + lane * 16;
        const unsigned lds_addr = (unsigned)reinterpret_cast<size_t>(&stl[sslot][piece * 128]);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr) : "memory");
    };
    auto dma_tile = [&](int t, int slot) {                        // rows clamped, never conditional: always MT * SW DMAs
#pragma unroll
        for (int f = 0; f < MT; ++f) {
            long m = (long)t * (MT * 16) + f * 16 + r16;
            m = m < g.M ? m : g.M - 1;
            const f16 *row = g.a + m * g.lda + acol;
#pragma unroll
            for (int i = 0; i < SW; ++i) {
                const unsigned lds_addr = (unsigned)reinterpret_cast<size_t>(&act[slot][wave + 4 * i][f][0]);
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(row + 128 * i), "s"(lds_addr) : "memory");
            }
        }
    };
    dma_tile(g0, 0);
    dma_tile(g0 + G, 1);
    if constexpr (LNF) dma_stats(g0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the weights and the bias have landed, and the compiler must KNOW it (an opaque use of each register): its own wait for a
    // first use inside the loop would be vmcnt(0) on every trip, which drains the hand-counted DMAs as well
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int n = 0; n < NTW; ++n) asm volatile("" : "+v"(aq[ks][n]));
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        asm volatile("" : "+v"(bv[n].x), "+v"(bv[n].y), "+v"(bv[n].z), "+v"(bv[n].w));
        if constexpr (LNF) asm volatile("" : "+v"(wsv[n].x), "+v"(wsv[n].y), "+v"(wsv[n].z), "+v"(wsv[n].w));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int slot = 0, sslot = 0;
#pragma unroll 1
    for (int t = g0; t < tiles; t += G) {
        if constexpr (LNF) dma_stats(t + G, sslot ^ 1);
        dma_tile(t + 2 * G, slot == 0 ? 2 : slot - 1);           // (slot + 2) % 3: read last in the previous trip, behind its barrier
        f32x4 acc[MT][NTW];
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int f = 0; f < MT; ++f)
                acc[f][n] = LNF ? (f32x4){0.f, 0.f, 0.f, 0.f} : (f32x4){bv[n].x, bv[n].y, bv[n].z, bv[n].w};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            f16x8 bq[MT];
#pragma unroll
            for (int f = 0; f < MT; ++f) bq[f] = act[slot][ks][f][lane];
#pragma unroll
            for (int n = 0; n < NTW; ++n)
#pragma unroll
                for (int f = 0; f < MT; ++f) acc[f][n] = MFMA_16x16x32(aq[ks][n], bq[f], acc[f][n]);
        }
        // tile t + G is in LDS; only tile t + 2G's DMAs (six, seven with the statistics) may be in flight
        if constexpr (LNF) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
#pragma unroll
        for (int f = 0; f < MT; ++f) {
            const long m = (long)t * (MT * 16) + f * 16 + r16;
            if constexpr (LNF) {
                // W ((x - mu) r) + b = r (W x) + (b - r mu wsum): two FMAs per output, mu and r from the token's 12 partials
                const float4 *pp = reinterpret_cast<const float4 *>(&stl[sslot][(f * 16 + r16) * kParts]);
                float su = 0.f, sq = 0.f;
#pragma unroll
                for (int q = 0; q < kParts / 2; ++q) { const float4 v = pp[q]; su += v.x + v.z; sq += v.y + v.w; }
                const float mu = su * (1.0f / (KS * 32));
                const float var = fmaxf(sq * (1.0f / (KS * 32)) - mu * mu, 0.f);
                const float r = rsqrtf(var + g.ln_eps), rm = r * mu;
#pragma unroll
                for (int n = 0; n < NTW; ++n) {
                    acc[f][n][0] = fmaf(r, acc[f][n][0], fmaf(-rm, wsv[n].x, bv[n].x));
                    acc[f][n][1] = fmaf(r, acc[f][n][1], fmaf(-rm, wsv[n].y, bv[n].y));
                    acc[f][n][2] = fmaf(r, acc[f][n][2], fmaf(-rm, wsv[n].z, bv[n].z));
                    acc[f][n][3] = fmaf(r, acc[f][n][3], fmaf(-rm, wsv[n].w, bv[n].w));
                }
            }
            f16x4 o0, o1;
            if constexpr (GELU) {
                const f16x8 h = gelu8(acc[f][0], acc[f][1]);
                o0 = (f16x4){h[0], h[1], h[2], h[3]};
                o1 = (f16x4){h[4], h[5], h[6], h[7]};
            } else {
                o0 = (f16x4){(f16)acc[f][0][0], (f16)acc[f][0][1], (f16)acc[f][0][2], (f16)acc[f][0][3]};
                o1 = (f16x4){(f16)acc[f][1][0], (f16)acc[f][1][1], (f16)acc[f][1][2], (f16)acc[f][1][3]};
            }
            const f16x8 ov = pair_to_run(o0, o1);
            if (m < g.M) *reinterpret_cast<f16x8 *>(g.out + m * g.ldo + ocol) = ov;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        slot = slot == 2 ? 0 : slot + 1;
        sslot ^= 1;
    }
}

// Measured on ViT-S (5 492 tokens, ms per 12 launches, against gemm_os_kernel<2, 12, 1>; register-staged first form of this kernel):
// fc1 (N = 1 536) 0.266 -> 0.226, qkv (N = 1 152) 0.215 -> 0.204, proj (N = 384: 516 tiles are 258 workgroups of two tiles, one
// wave per SIMD) 0.139 -> 0.159 — so only N >= 768, which also means no residual variant is needed.
static bool gemm_ws_shape(const GemmOsArgs &g) { return g.K == 384 && g.N % 128 == 0 && g.N >= 768 && !g.res; }

bool gemm_os_consumes_stats(long M, int N, int K) { return M > 0 && K == 384 && N % 128 == 0 && N >= 768; }

template <bool GELU, bool LNF>
static int launch_gemm_ws_t(const GemmOsArgs &g, hipStream_t s) {
    static int resident = 0;                                      // workgroups the chip holds at once
    if (!resident) {
        int dev = 0, per_cu = 0;
        hipDeviceProp_t prop;
        NUNIF_HIP_CHECK(hipGetDevice(&dev));
        NUNIF_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        NUNIF_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)gemm_ws_kernel<12, 2, 2, GELU, LNF>, 256, 0));
        resident = std::max(1, per_cu) * prop.multiProcessorCount;
    }
    const int tiles = (int)((g.M + 31) / 32), n_blocks = g.N / 128;
    const int per_wg = (int)(((long)tiles * n_blocks + resident - 1) / resident);        // tiles per workgroup
    const int G = (tiles + per_wg - 1) / per_wg;
    gemm_ws_kernel<12, 2, 2, GELU, LNF><<<(unsigned)(G * n_blocks), 256, 0, s>>>(g, tiles, G);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

static int launch_gemm_ws(const GemmOsArgs &g, hipStream_t s) {
    if (g.stats_in) return g.act == 1 ? launch_gemm_ws_t<true, true>(g, s) : launch_gemm_ws_t<false, true>(g, s);
    return g.act == 1 ? launch_gemm_ws_t<true, false>(g, s) : launch_gemm_ws_t<false, false>(g, s);
}

int launch_gemm_os(const GemmOsArgs &g, hipStream_t s, const char *tag) {
    NUNIF_REQUIRE(gemm_os_supported(g.M, g.N, g.K) && g.lda % 8 == 0 && g.ldo % 8 == 0, "gemm_os %s: M=%ld N=%d K=%d unsupported",
                  tag, g.M, g.N, g.K);
    ProfScope ps(profile_tags_enabled() ? tag : "gemm_os_kernel", s, 2.0 * (double)g.M * g.K * g.N,
                 (double)g.M * (g.K * 2.0 + g.N * 2.0 * (g.res ? 2.0 : 1.0)));
    NUNIF_REQUIRE(!g.stats_in || (gemm_ws_shape(g) && g.stats_parts == 12 && g.wsum),
                  "gemm_os %s: statistics-normalised rows need the weight-stationary launch (K = 384, N >= 768, 12 partials)", tag);
    NUNIF_REQUIRE(!g.stats_out || (!gemm_ws_shape(g) && g.N % 32 == 0), "gemm_os %s: this launch does not write statistics", tag);
    if (gemm_ws_shape(g)) return launch_gemm_ws(g, s);
    // 32-token workgroups (fewer registers, more resident waves) unless 64-token ones already give the chip four workgroups per CU
    // (those keep the round-2 shape: 64 tokens, four k-steps per group)
    const bool wide = ((g.M + 63) / 64) * (g.N / 128) >= 1100;
    // (64-token tiles with PF = 12 — 40 % less operand traffic — measured on ViT-S: fc1 0.272 -> 0.262, qkv 0.219 -> 0.220, fc2
    //  0.334 -> 0.404 ms per 12 launches: these launches are not bound by operand bytes either)
    if (wide) launch_gemm_os_t<4, 4>(g, s);
    else if (g.K % 384 == 0) launch_gemm_os_t<2, 12>(g, s);
    else if (g.K % 256 == 0) launch_gemm_os_t<2, 8>(g, s);
    else launch_gemm_os_t<2, 4>(g, s);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

// =================================================================================================================
// Stem conv1: 3 -> C1 3x3 VALID + LeakyReLU on the VALU (K = 27 is too thin for MFMA; 0.15 % of the FLOPs).
// Fuses the reference's replicate-pad + tile slicing (seam_blending.py:82,90) when reading from the frame.
// One thread = one output pixel, all channels; weights broadcast from LDS.
// =================================================================================================================
__global__ void __launch_bounds__(256) stem1_kernel(Stem1Args a) {
    extern __shared__ float sw[];   // [27][C1] then bias[C1]
    const int C1 = a.C1;
    for (int i = threadIdx.x; i < 27 * C1 + C1; i += blockDim.x) {
        if (i < 27 * C1) {
            const int co = i % C1, t = i / C1;        // t = ci*9 + ky*3 + kx
            sw[i] = a.w[co * 27 + t];
        } else {
            sw[i] = a.bias[i - 27 * C1];
        }
    }
    __syncthreads();
    const int S = a.T - 14;
    const long total = (long)a.B * S * S;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % S);
    const long t = idx / S;
    const int y = (int)(t % S);
    const int b = (int)(t / S);

    float in[27];
    if (a.frame_mode) {
        const int k = a.tile_begin + b;
        const int ti = k / a.wb, tj = k - ti * a.wb;
        const int y0 = ti * a.istep - a.pad_t + y + 6, x0 = tj * a.istep - a.pad_l + x + 6;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int sy = min(max(y0 + ky, 0), a.H - 1), sx = min(max(x0 + kx, 0), a.W - 1);
                    in[ci * 9 + ky * 3 + kx] = a.x[((long)ci * a.H + sy) * a.W + sx];
                }
    } else {
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    in[ci * 9 + ky * 3 + kx] = a.x[(((long)b * 3 + ci) * a.T + (y + 6 + ky)) * a.T + (x + 6 + kx)];
    }
    f16 *o = a.out + idx * a.C1P;
    for (int c0 = 0; c0 < a.C1P; c0 += 8) {
        f16x8 ov;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int co = c0 + j;
            float acc = 0.f;
            if (co < C1) {
                acc = sw[27 * C1 + co];
#pragma unroll
                for (int t2 = 0; t2 < 27; ++t2) acc = fmaf(in[t2], sw[t2 * C1 + co], acc);
                acc = acc >= 0.f ? acc : acc * a.slope;
            }
            ov[j] = (f16)acc;
        }
        *reinterpret_cast<f16x8 *>(o + c0) = ov;
    }
}

int launch_stem1(const Stem1Args &a, hipStream_t s) {
    NUNIF_REQUIRE(a.C1P % 8 == 0 && a.C1 <= a.C1P && a.T > 14, "stem1: bad shape");
    const int S = a.T - 14;
    const long total = (long)a.B * S * S;
    ProfScope ps("stem1_kernel", s, 2.0 * 27 * a.C1 * (double)total, (double)total * (a.C1P * 2.0 + 12.0));
    const size_t smem = (size_t)(28 * a.C1) * sizeof(float);
    stem1_kernel<<<(unsigned)((total + 255) / 256), 256, smem, s>>>(a);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

// =================================================================================================================
// LayerNormNoBias over the channel axis of an NHWC fp16 map (nunif/modules/norm.py:17-22 = nn.LayerNorm(bias=False),
// eps 1e-5; the norm1 / norm2 of swin_unet_4xl's blocks, swin_unet.py:390-391).  One wave per token, two-pass in
// registers (mean, then the centred second moment, like ATen's CPU kernel), fp32 statistics, fp16 result.
// =================================================================================================================
template <int C>
__global__ void __launch_bounds__(256) layernorm_nobias_kernel(const f16 *__restrict__ x, f16 *__restrict__ y,
                                                               const float *__restrict__ gamma, long M, float eps) {
    static_assert(C % 32 == 0, "C must be a multiple of 32");
    constexpr int NP = (C / 2 + 63) / 64;            // half2 pieces per lane
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const f16 *px = x + m * C;
    float v[NP][2];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c2 = i * 64 + lane;
        if (c2 < C / 2) {
            const __half2 h2 = *reinterpret_cast<const __half2 *>(px + 2 * c2);
            v[i][0] = __low2float(h2); v[i][1] = __high2float(h2);
        } else {
            v[i][0] = 0.f; v[i][1] = 0.f;
        }
        sum += v[i][0] + v[i][1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum * (1.0f / C);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c2 = i * 64 + lane;
        if (c2 < C / 2) {
            const float d0 = v[i][0] - mean, d1 = v[i][1] - mean;
            sq += d0 * d0 + d1 * d1;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = rsqrtf(sq * (1.0f / C) + eps);
    f16 *py = y + m * C;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c2 = i * 64 + lane;
        if (c2 < C / 2) {
            const float2 g2 = *reinterpret_cast<const float2 *>(gamma + 2 * c2);
            *reinterpret_cast<__half2 *>(py + 2 * c2) =
                __floats2half2_rn((v[i][0] - mean) * rstd * g2.x, (v[i][1] - mean) * rstd * g2.y);
        }
    }
}

int launch_layernorm_nobias(const f16 *x, f16 *y, const float *gamma, long M, int C, hipStream_t s) {
    if (M == 0) return NUNIF_HIP_OK;
    ProfScope ps("layernorm_nobias_kernel", s, 8.0 * (double)M * C, (double)M * C * 4.0);
    const unsigned blocks = (unsigned)((M + 3) / 4);
    if (C == 96) layernorm_nobias_kernel<96><<<blocks, 256, 0, s>>>(x, y, gamma, M, 1e-5f);
    else if (C == 192) layernorm_nobias_kernel<192><<<blocks, 256, 0, s>>>(x, y, gamma, M, 1e-5f);
    else if (C == 384) layernorm_nobias_kernel<384><<<blocks, 256, 0, s>>>(x, y, gamma, M, 1e-5f);
    else if (C == 64) layernorm_nobias_kernel<64><<<blocks, 256, 0, s>>>(x, y, gamma, M, 1e-5f);
    else if (C == 128) layernorm_nobias_kernel<128><<<blocks, 256, 0, s>>>(x, y, gamma, M, 1e-5f);
    else if (C == 256) layernorm_nobias_kernel<256><<<blocks, 256, 0, s>>>(x, y, gamma, M, 1e-5f);
    else {
        set_error("layernorm: channel count %d unsupported (64, 96, 128, 192, 256, 384)", C);
        return NUNIF_HIP_EUNSUPPORTED;
    }
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

// =================================================================================================================
// Window attention, one wave per (window, head).  36 tokens are padded to 48 = 3 MFMA tiles.
//   S^T[key][q] = K Q^T          (A = K rows, B = Q rows)         -> lane holds (q = lane&15, 4 keys)
//   softmax over keys            = over the 12 in-lane values and the 4 lane groups (2 xor-shuffles)
//   O^T[d][q]  = V^T P^T         (A = V^T gathered, B = P^T)      -> P^T is already in B-operand layout
// The cyclic shift (torch.roll) and the window partition are folded into the token addressing; the shift mask is
// computed from coordinates (torchvision shifted_window_attention).
// =================================================================================================================
template <int HD>
__global__ void __launch_bounds__(256)
window_attn_kernel(const f16 *__restrict__ qkv, f16 *__restrict__ out, const float *__restrict__ bias, int B, int H,
                   int W, int heads, int shift, float scale) {
    typedef typename std::conditional<HD == 32, f16x8, f16x4>::type frag_t;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r16 = lane & 15;
    const int grp = lane >> 4;
    const int C = heads * HD, C3 = 3 * C;
    const int nwx = W / 6, nwy = H / 6;
    const long total = (long)B * nwy * nwx * heads;
    const long gid = (long)blockIdx.x * 4 + wave;
    if (gid >= total) return;
    const int head = (int)(gid % heads);
    long wi = gid / heads;
    const int wx = (int)(wi % nwx);
    wi /= nwx;
    const int wy = (int)(wi % nwy);
    const int b = (int)(wi / nwy);

    auto pix = [&](int t) -> long {   // window-local token -> pixel index in the un-rolled map
        t = min(t, 35);
        const int iy = t / 6, ix = t - iy * 6;
        int yy = wy * 6 + iy + shift, xx = wx * 6 + ix + shift;
        if (yy >= H) yy -= H;
        if (xx >= W) xx -= W;
        return ((long)b * H + yy) * W + xx;
    };
    auto region = [&](int t) -> int {  // 9-region id of the rolled position (slices (0,-6),(-6,-3),(-3,None))
        t = min(t, 35);
        const int iy = t / 6, ix = t - iy * 6;
        const int py = wy * 6 + iy, px = wx * 6 + ix;
        const int ry = py < H - 6 ? 0 : (py < H - 3 ? 1 : 2);
        const int rx = px < W - 6 ? 0 : (px < W - 3 ? 1 : 2);
        return ry * 3 + rx;
    };

    long pq[3];
    frag_t qf[3], kf[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        pq[t] = pix(t * 16 + r16);
        const f16 *p = qkv + pq[t] * C3 + head * HD + grp * (HD / 4);
        qf[t] = *reinterpret_cast<const frag_t *>(p);
        kf[t] = *reinterpret_cast<const frag_t *>(p + C);
    }

    // scores: sc[kt][qt][r] is S[q = qt*16 + r16][key = kt*16 + grp*4 + r]
    float sc[3][3][4];
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int qt = 0; qt < 3; ++qt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (HD == 32) acc = MFMA_16x16x32(kf[kt], qf[qt], acc);
            else acc = MFMA_16x16x16(kf[kt], qf[qt], acc);
            const int qi = min(qt * 16 + r16, 35);
            const float4 bv = *reinterpret_cast<const float4 *>(bias + ((long)head * 36 + qi) * 48 + kt * 16 + grp * 4);
            sc[kt][qt][0] = acc[0] * scale + bv.x;
            sc[kt][qt][1] = acc[1] * scale + bv.y;
            sc[kt][qt][2] = acc[2] * scale + bv.z;
            sc[kt][qt][3] = acc[3] * scale + bv.w;
        }
    if (shift > 0) {
        int idq[3], idk[3][4];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            idq[t] = region(t * 16 + r16);
#pragma unroll
            for (int r = 0; r < 4; ++r) idk[t][r] = region(t * 16 + grp * 4 + r);
        }
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int qt = 0; qt < 3; ++qt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (idq[qt] != idk[kt][r]) sc[kt][qt][r] += -100.0f;
    }

    // softmax over keys (un-normalised p in fp16 feeds the MFMA; 1/sum is applied to the fp32 output)
    float inv_sum[3];
    f16x4 pf[3][3];
#pragma unroll
    for (int qt = 0; qt < 3; ++qt) {
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[kt][qt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(sc[kt][qt][r] - mx);
                sum += p;
                pf[kt][qt][r] = (f16)p;
            }
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        inv_sum[qt] = 1.0f / sum;
    }

    // O^T = V^T P^T
    const f16 *vbase = qkv + 2 * C + head * HD;
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) {
        f16x4 vf[3];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) vf[kt][j] = vbase[pix(kt * 16 + grp * 4 + j) * C3 + dt * 16 + r16];
#pragma unroll
        for (int qt = 0; qt < 3; ++qt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) acc = MFMA_16x16x16(vf[kt], pf[kt][qt], acc);
            if (qt * 16 + r16 < 36) {
                const float s = inv_sum[qt];
                f16x4 ov = {(f16)(acc[0] * s), (f16)(acc[1] * s), (f16)(acc[2] * s), (f16)(acc[3] * s)};
                *reinterpret_cast<f16x4 *>(out + pq[qt] * C + head * HD + dt * 16 + grp * 4) = ov;
            }
        }
    }
}

int launch_window_attn(const f16 *qkv, f16 *out, const float *bias, int B, int H, int W, int heads, int hd,
                       int shift, hipStream_t s) {
    NUNIF_REQUIRE(H % 6 == 0 && W % 6 == 0, "window_attn: %dx%d not a multiple of the 6x6 window", H, W);
    NUNIF_REQUIRE(hd == 16 || hd == 32, "window_attn: head_dim %d unsupported", hd);
    // torchvision disables the shift on an axis whose size is <= the window; both axes are equal here
    if (H <= 6) shift = 0;
    const long total = (long)B * (H / 6) * (W / 6) * heads;
    const double tok = (double)B * H * W;
    ProfScope ps(hd == 16 ? "window_attn_kernel<16>" : "window_attn_kernel<32>", s, 4.0 * tok * 36.0 * heads * hd, tok * heads * hd * 2.0 * 4.0);
    const unsigned blocks = (unsigned)((total + 3) / 4);
    const float scale = 1.0f / sqrtf((float)hd);
    if (hd == 16) window_attn_kernel<16><<<blocks, 256, 0, s>>>(qkv, out, bias, B, H, W, heads, shift, scale);
    else window_attn_kernel<32><<<blocks, 256, 0, s>>>(qkv, out, bias, B, H, W, heads, shift, scale);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

}  // namespace nunif
