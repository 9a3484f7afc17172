// Tail of a C = 96 swin block on gfx950: three GEMMs chained through registers, weights resident in LDS.
//
//     y  = x + Wp att + bp                     (attn.proj + residual)
//     x' = y + W3 gelu(W0 y + b0) + b3         (mlp.0, GELU(erf), mlp.3 + residual)      in place on x
//
// Replaces, per block, torchvision SwinTransformerBlock's `x = x + proj(...)` and `x = x + mlp(norm2(x))`
// (norm = Identity for the waifu2x nets, waifu2x/models/swin_unet.py:16-17,26-36).  C = 192 blocks run the
// weight-stationary form of the same math (swin_block_tail_ws.hip).
//
// Register chaining.  The weights are the MFMA A operand (rows = output channel), the activations the B operand
// (cols = token).  A 16x16 accumulator tile then holds, in lane l, channels 4*(l>>4)+r of token l&15 — which is
// exactly a B-operand fragment of the NEXT GEMM once two tiles are paired into 8 k-slots.  The k-slot permutation
// that implies is baked into the "chained" weight packing on the host (make_linear in swin_unet.cpp), so y and the
// 2C-wide hidden activation never leave registers: no LDS transpose, no HBM round trip.
//
// Weights.  The three matrices form ONE stream of 1-KiB fragments in consumption order (90 KiB at C = 96); a persistent
// 8-wave workgroup copies it into LDS once and every wave then loops over its own 32-token groups with NO barrier
// (round 1 pulled the stream through a 2 x 8 KiB LDS ring with a barrier per chunk; that kernel is gone).
// Fragments are read back with lane-linear ds_read_b128 (conflict free).
//
// HBM traffic per token: read att (2C B) + read x (2C B) + write x (2C B).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "swin_gelu.h"
#include "swin_kernels.h"

namespace nunif {

#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

// Waves drift apart (no barrier in the group loop), so one wave's GELU / convert (VALU) phase overlaps its SIMD partner's
// MFMA phase (same finding as swin_qkv_attn_r.hip).  MF = 2 token tiles per group, 8 waves, next-group prefetch: the fastest
// of the eight (MF, waves, prefetch) combinations measured in rounds 1-2.
template <int C, int MF, int WAVES, bool PF = false, bool WM = false, bool G32 = false>
__global__ void __launch_bounds__(WAVES * 64)
proj_mlp_r_kernel(const f16 *__restrict__ att, f16 *x, const f16 *__restrict__ wstream, const float *__restrict__ bp,
                  const float *__restrict__ b0, const float *__restrict__ b3, long M, TailToImage ti, int rev, WinMap wm) {
    constexpr int KS = C / 32, NT = C / 16, SH = 2 * C / 32;
    constexpr int NF = 2 * KS * KS + SH * (2 * KS + NT);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_t[];
    f16x8 *wl = reinterpret_cast<f16x8 *>(smem_t);                 // [NF + KS][64] (KS image-head fragments at the end)
    float *bl = reinterpret_cast<float *>(wl + (NF + KS) * 64);    // bp[C] | b0[2C] | b3[C] | image-head bias[16]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r16 = lane & 15;
    const int grp = lane >> 4;
    {
        const f16x8 *src = reinterpret_cast<const f16x8 *>(wstream);
        for (int i = tid; i < NF * 64; i += WAVES * 64) wl[i] = src[i];
        for (int i = tid; i < C; i += WAVES * 64) { bl[i] = bp[i]; bl[3 * C + i] = b3[i]; }
        for (int i = tid; i < 2 * C; i += WAVES * 64) bl[C + i] = b0[i];
        if (ti.w) {
            for (int i = tid; i < KS * 64; i += WAVES * 64) wl[NF * 64 + i] = reinterpret_cast<const f16x8 *>(ti.w)[i];
            if (tid < 16) bl[4 * C + tid] = tid < ti.n_real ? ti.bias[tid] : 0.f;
        }
    }
    __syncthreads();
    const f16x8 *wlane = wl + lane;
    auto wfrag = [&](int fi) -> f16x8 { return wlane[fi * 64]; };
    // Both residual adds ride the matrix pipe: this kernel is VALU-issue bound (GELU alone is 9.5 operations per hidden
    // element, tools/ubench_mix.hip) while the MFMA pipe is a quarter busy, so `y = x + ...` and `x' = y + ...` are one
    // extra MFMA per 16-channel tile against a constant IDENTITY fragment instead of 48 + 48 v_cvt_f32_f16 / v_add_f32
    // (+ 24 v_permlane) per 32 tokens.  fp16 x 1.0 is exact and the accumulation stays fp32, so only the summation order
    // changes.  ie / io: identity rows of the even / odd tile of a pair for a PLAIN-k-order B fragment (k = 8 grp + j);
    // je / jo: the same for a CHAINED-k-order fragment (slots 0-3 = tile 2s channels 4 grp + j, slots 4-7 = tile 2s + 1).
    f16x8 ie, io, je, jo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ie[j] = (f16)((r16 == 8 * grp + j) ? 1.f : 0.f);
        io[j] = (f16)((r16 + 16 == 8 * grp + j) ? 1.f : 0.f);
        je[j] = (f16)((j < 4 && r16 == 4 * grp + j) ? 1.f : 0.f);
        jo[j] = (f16)((j >= 4 && r16 == 4 * grp + j - 4) ? 1.f : 0.f);
    }

    const long n_groups = (M + MF * 16 - 1) / (MF * 16);
    const long gstride = (long)gridDim.x * WAVES;
    // PF: the att / x tiles of the NEXT group are requested before this group's GEMMs start.  Without it every
    // wave exposes one full HBM latency per group and the kernel sits at ~3.3 TB/s however cheap the math is
    // (bytes in flight per CU ~ 20 KB; Little's law wants >= 40 KB for 5 TB/s).
    auto gmap = [&](long g) { return rev ? n_groups - 1 - g : g; };          // snake order between kernels
    // pixn[f]: row of x (and of the image head) that token f*16 + r16 of the group lives at.  Plain map: the token index.
    // Window map (WM): token n = window * 36 + t of the SHIFTED 6x6 windows, t row-major in the window — the order
    // qkv_attn_r_kernel writes its window-major att map in; att is then [window][head 6][36][16].  The token -> pixel map
    // is a TABLE (wm.pixmap, one int per token, built once per geometry by winmap_build_kernel): round 3 recomputed it per
    // lane and group with five integer divisions, ~150 of the 1 380 issue slots of a group in a kernel that is issue-bound
    // (profiles/r04_isa_counts.txt).  A group's table entries are requested one group EARLIER than its att / x tiles
    // (pm_next), so the dependent loads never wait on each other.
    auto token_of = [&](long g, int f) -> long {
        const long m = gmap(g) * (MF * 16) + f * 16 + r16;
        return m < M ? m : M - 1;
    };
    auto load_pixmap = [&](long g, int (&pm)[MF]) {
#pragma unroll
        for (int f = 0; f < MF; ++f) pm[f] = wm.pixmap[(int)token_of(g, f)];
    };
    auto load_group = [&](long g, const int (&pm)[MF], f16x8 (&of)[MF][KS], f16x8 (&xr)[MF][KS], long (&pixn)[MF]) {
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const long r = token_of(g, f);
            const f16 *p;
            if constexpr (WM) {
                // lane group g holds channels 32 ks + 8 g .. + 7 = head 2 ks + (g >> 1), dims 8 (g & 1) ..:
                // ((w 6 + head) 36 + t) 16 = 16 r + 2880 w + 576 head   (r = 36 w + t; 96 M < 2^31 checked by the launcher)
                const int ri = (int)r, w = ri / 36;
                pixn[f] = pm[f];
                p = att + (ri * 16 + w * 2880 + (grp >> 1) * 576 + 8 * (grp & 1));
            } else {
                pixn[f] = r;
                p = att + r * C + grp * 8;
            }
            const f16 *px = x + pixn[f] * C + grp * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                of[f][ks] = *reinterpret_cast<const f16x8 *>(p + (WM ? ks * (2 * 36 * 16) : ks * 32));
                xr[f][ks] = *reinterpret_cast<const f16x8 *>(px + ks * 32);      // B-operand layout: the residual rides an MFMA
            }
        }
    };
    f16x8 ofn[PF ? MF : 1][PF ? KS : 1];
    f16x8 xrn[PF ? MF : 1][PF ? KS : 1];
    long pixnn[MF];
    const long g_first = (long)blockIdx.x * WAVES + wave;
    int pm_next[MF] = {};
    auto g_clamp = [&](long g) { return g < n_groups ? g : n_groups - 1; };
    if constexpr (PF) {
        if (g_first < n_groups) {
            if constexpr (WM) load_pixmap(g_first, pm_next);
            load_group(g_first, pm_next, ofn, xrn, pixnn);
            if constexpr (WM) load_pixmap(g_clamp(g_first + gstride), pm_next);
        }
    }
#pragma unroll 1
    for (long g = g_first; g < n_groups; g += gstride) {
        const long m_base = gmap(g) * (MF * 16);
        // an opaque per-iteration OFFSET keeps LICM from hoisting 48 registers of (loop-invariant) LDS bias reads.  (Rounds 1-3
        // laundered the POINTERS: hipcc then no longer knows they are LDS addresses and reads the biases with flat_load,
        // which counts on vmcnt — every `s_waitcnt vmcnt(0)` in front of a bias use, i.e. the top of each hidden-slice trip,
        // also drained the next group's att / x prefetch issued just before the loop: one HBM latency exposed per group.)
        int lofs = 0;
        asm volatile("" : "+v"(lofs));
        const float *lbp = bl + lofs, *lb0 = bl + C + lofs, *lb3 = bl + 3 * C + lofs;
        long row[MF];
        bool valid[MF];
        f16x8 yf[MF][KS];
        {
            f16x8 of[MF][KS];
            f16x8 xr[MF][KS];
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                const long m = m_base + f * 16 + r16;
                valid[f] = m < M;
            }
            if constexpr (PF) {
#pragma unroll
                for (int f = 0; f < MF; ++f) {
                    row[f] = pixnn[f];
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) { of[f][ks] = ofn[f][ks]; xr[f][ks] = xrn[f][ks]; }
                }
                load_group(g_clamp(g + gstride), pm_next, ofn, xrn, pixnn);
                if constexpr (WM) load_pixmap(g_clamp(g + 2 * gstride), pm_next);
            } else {
                if constexpr (WM) load_pixmap(g, pm_next);
                load_group(g, pm_next, of, xr, row);
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int n0 = 32 * s + 4 * grp;
                const f32x4 ba = *reinterpret_cast<const f32x4 *>(lbp + n0);
                const f32x4 bb = *reinterpret_cast<const f32x4 *>(lbp + n0 + 16);
                f32x4 a0[MF], a1[MF];
#pragma unroll
                for (int f = 0; f < MF; ++f) { a0[f] = ba; a1[f] = bb; }        // accumulators start from the bias
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const f16x8 wa = wfrag((s * KS + ks) * 2);
                    const f16x8 wb = wfrag((s * KS + ks) * 2 + 1);
#pragma unroll
                    for (int f = 0; f < MF; ++f) {
                        a0[f] = MFMA_16x16x32(wa, of[f][ks], a0[f]);
                        a1[f] = MFMA_16x16x32(wb, of[f][ks], a1[f]);
                    }
                }
#pragma unroll
                for (int f = 0; f < MF; ++f) {
                    a0[f] = MFMA_16x16x32(ie, xr[f][s], a0[f]);          // + x (channels 32 s .. 32 s + 15)
                    a1[f] = MFMA_16x16x32(io, xr[f][s], a1[f]);          // + x (channels 32 s + 16 .. 32 s + 31)
                }
#pragma unroll
                for (int f = 0; f < MF; ++f)
                    yf[f][s] = (f16x8){(f16)a0[f][0], (f16)a0[f][1], (f16)a0[f][2], (f16)a0[f][3],
                                       (f16)a1[f][0], (f16)a1[f][1], (f16)a1[f][2], (f16)a1[f][3]};
            }
        }
        f32x4 acc[NT][MF];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int n0 = 32 * s + 4 * grp;
            const f32x4 ca = *reinterpret_cast<const f32x4 *>(lb3 + n0);
            const f32x4 cb = *reinterpret_cast<const f32x4 *>(lb3 + n0 + 16);
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                acc[2 * s][f] = MFMA_16x16x32(je, yf[f][s], ca);             // b3 + y
                acc[2 * s + 1][f] = MFMA_16x16x32(jo, yf[f][s], cb);
            }
        }
        constexpr int F_MLP = 2 * KS * KS;
        constexpr int F_STEP = 2 * KS + NT;
#pragma unroll 1
        for (int s = 0; s < SH; ++s) {
            const int n0 = 32 * s + 4 * grp;
            const f32x4 ba = *reinterpret_cast<const f32x4 *>(lb0 + n0);
            const f32x4 bb = *reinterpret_cast<const f32x4 *>(lb0 + n0 + 16);
            f32x4 h0[MF], h1[MF];
#pragma unroll
            for (int f = 0; f < MF; ++f) { h0[f] = ba; h1[f] = bb; }
            const f16x8 *wf = wlane + (F_MLP + s * F_STEP) * 64;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const f16x8 wa = wf[(ks * 2) * 64];
                const f16x8 wb = wf[(ks * 2 + 1) * 64];
#pragma unroll
                for (int f = 0; f < MF; ++f) {
                    h0[f] = MFMA_16x16x32(wa, yf[f][ks], h0[f]);
                    h1[f] = MFMA_16x16x32(wb, yf[f][ks], h1[f]);
                }
            }
            f16x8 hf[MF];
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                hf[f] = gelu8t<G32>(h0[f], h1[f]);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f16x8 wv = wf[(2 * KS + nt) * 64];
#pragma unroll
                for (int f = 0; f < MF; ++f) acc[nt][f] = MFMA_16x16x32(wv, hf[f], acc[nt][f]);
            }
        }
        if (ti.w) {
            // fused image head: img[n][token] = sum_c Wt[n][c] x'[c] with x' rounded to fp16 exactly as the stored map
            // would be; the accumulator tile pairs are the B fragments (chained k order), n = c*ps*ps + i*ps + j
            const int s2 = ti.ps * ti.ps;
            const f32x4 tb = *reinterpret_cast<const f32x4 *>(bl + 4 * C + 4 * grp);
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                f32x4 img = tb;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const f16x8 bfrag = {(f16)acc[2 * ks][f][0], (f16)acc[2 * ks][f][1], (f16)acc[2 * ks][f][2], (f16)acc[2 * ks][f][3],
                                         (f16)acc[2 * ks + 1][f][0], (f16)acc[2 * ks + 1][f][1], (f16)acc[2 * ks + 1][f][2],
                                         (f16)acc[2 * ks + 1][f][3]};
                    img = MFMA_16x16x32(wlane[(NF + ks) * 64], bfrag, img);
                }
                if (!valid[f]) continue;
                // window-map launches address < 2^31 pixels (checked by the launcher): 32-bit divisions
                using pix_t = std::conditional_t<WM, unsigned, long>;
                const pix_t m = (pix_t)row[f];
                const int px = (int)(m % (pix_t)ti.W);
                const pix_t t2 = m / (pix_t)ti.W;
                const int py = (int)(t2 % (pix_t)ti.H), pb = (int)(t2 / (pix_t)ti.H);
                const int OC = ti.n_real / s2;
                const long OH = (long)ti.H * ti.ps, OW = (long)ti.W * ti.ps;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = 4 * grp + r;
                    if (n < ti.n_real) {
                        const int c = n / s2, rem = n - c * s2, i = rem / ti.ps, j = rem - i * ti.ps;
                        ti.out[(((long)pb * OC + c) * OH + (long)py * ti.ps + i) * OW + (long)px * ti.ps + j] =
                            fminf(fmaxf(img[r], 0.f), 1.f);
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int p = 0; p < NT / 2; ++p) {
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                const f16x4 oa = {(f16)acc[2 * p][f][0], (f16)acc[2 * p][f][1], (f16)acc[2 * p][f][2], (f16)acc[2 * p][f][3]};
                const f16x4 ob = {(f16)acc[2 * p + 1][f][0], (f16)acc[2 * p + 1][f][1], (f16)acc[2 * p + 1][f][2],
                                  (f16)acc[2 * p + 1][f][3]};
                const f16x8 o = pair_to_run(oa, ob);
                if (valid[f]) *reinterpret_cast<f16x8 *>(x + row[f] * C + 32 * p + pair_run_channel(grp)) = o;
            }
        }
    }
}

// pixmap[n] = pixel (b H + y) W + x of token n = window * 36 + t in the order of the SHIFTED 6x6 windows (torchvision rolls the
// map by -shift, partitions, and rolls back: window (wy, wx) token (iy, ix) is pixel ((6 wy + iy + shift) mod H, ...)).
__global__ void winmap_build_kernel(int *__restrict__ pixmap, int B, int H, int W, int shift) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int nwx = W / 6, nwy = H / 6;
    if (n >= B * H * W) return;
    const int w = n / 36, t = n - 36 * w;
    const int wx = w % nwx, t2 = w / nwx;
    const int wy = t2 % nwy, b = t2 / nwy;
    const int iy = t / 6, ix = t - 6 * iy;
    int yy = wy * 6 + iy + shift, xx = wx * 6 + ix + shift;
    if (yy >= H) yy -= H;
    if (xx >= W) xx -= W;
    pixmap[n] = (b * H + yy) * W + xx;
}

int launch_winmap_build(int *pixmap, int B, int H, int W, int shift, hipStream_t s) {
    NUNIF_REQUIRE(H % 6 == 0 && W % 6 == 0 && (long)B * H * W < (1L << 31), "winmap: geometry %d x %d x %d", B, H, W);
    const long n = (long)B * H * W;
    if (n == 0) return NUNIF_HIP_OK;
    winmap_build_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(pixmap, B, H, W, shift);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

constexpr int proj_mlp_stream_frags_c(int C) { return 2 * (C / 32) * (C / 32) + (2 * C / 32) * (2 * (C / 32) + C / 16); }

int proj_mlp_stream_frags(int C) {
    const int KS = C / 32, NT = C / 16, SH = 2 * C / 32;
    return 2 * KS * KS + SH * (2 * KS + NT);
}

int launch_proj_mlp(const f16 *att, f16 *x, const f16 *wstream, const float *bp, const float *b0, const float *b3,
                    long M, int C, hipStream_t s, const TailToImage *to_image, int rev, const WinMap *wmap, int gelu32) {
    if (M == 0) return NUNIF_HIP_OK;
    NUNIF_REQUIRE(C == 96, "proj_mlp: channel count %d unsupported (C = 192 runs launch_proj_mlp_ws)", C);
    NUNIF_REQUIRE(!to_image || to_image->n_real <= 16, "proj_mlp: the fused image head takes at most 16 output channels");
    // the profiler class is named after the kernel symbol so that it lines up with rocprofv3's kernel stats
    ProfScope ps("proj_mlp_r_kernel<96,2,8,true>", s, 2.0 * (double)M * C * C * 5.0, (double)M * C * 2.0 * 3.0);
    constexpr size_t smem = (size_t)(proj_mlp_stream_frags_c(96) + 3) * 1024 + (4 * 96 + 16) * 4;
    TailToImage ti;
    memset(&ti, 0, sizeof(ti));
    if (to_image) ti = *to_image;
    WinMap wm;
    memset(&wm, 0, sizeof(wm));
    if (wmap) wm = *wmap;
    NUNIF_REQUIRE(!wm.on || (wm.pixmap && wm.H % 6 == 0 && wm.W % 6 == 0 && M % ((long)wm.H * wm.W) == 0 && 96L * M < (1L << 31)),
                  "proj_mlp: window map geometry");        // 16 M + 2880 (M / 36) = 96 M att elements are indexed in 32 bits (swin_unet.cpp
                                                           // hands larger launches the pixel-major map)
    constexpr int MF = 2, WAVES = 8;
    auto go = [&](auto kern, int slot) -> int {
        static bool configured[4] = {false, false, false, false};
        if (!configured[slot]) {
            NUNIF_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            configured[slot] = true;
        }
        const long groups = (M + MF * 16 - 1) / (MF * 16);
        const unsigned blocks = (unsigned)std::min<long>((groups + WAVES - 1) / WAVES, 256);
        kern<<<blocks, WAVES * 64, smem, s>>>(att, x, wstream, bp, b0, b3, M, ti, rev, wm);
        return NUNIF_HIP_OK;
    };
    int rc;
    if (gelu32) rc = wm.on ? go(proj_mlp_r_kernel<96, MF, WAVES, true, true, true>, 3) : go(proj_mlp_r_kernel<96, MF, WAVES, true, false, true>, 2);
    else rc = wm.on ? go(proj_mlp_r_kernel<96, MF, WAVES, true, true, false>, 1) : go(proj_mlp_r_kernel<96, MF, WAVES, true, false, false>, 0);
    if (rc) return rc;
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

}  // namespace nunif
