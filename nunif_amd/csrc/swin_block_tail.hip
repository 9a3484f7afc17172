// Tail of a swin block on gfx950: three GEMMs chained through registers, weights streamed through an LDS ring.
//
//     y  = x + Wp att + bp                     (attn.proj + residual)
//     x' = y + W3 gelu(W0 y + b0) + b3         (mlp.0, GELU(erf), mlp.3 + residual)      in place on x
//
// Replaces, per block, torchvision SwinTransformerBlock's `x = x + proj(...)` and `x = x + mlp(norm2(x))`
// (norm = Identity for the waifu2x nets, waifu2x/models/swin_unet.py:16-17,26-36).
//
// Register chaining.  The weights are the MFMA A operand (rows = output channel), the activations the B operand
// (cols = token).  A 16x16 accumulator tile then holds, in lane l, channels 4*(l>>4)+r of token l&15 — which is
// exactly a B-operand fragment of the NEXT GEMM once two tiles are paired into 8 k-slots.  The k-slot permutation
// that implies is baked into the "chained" weight packing on the host (make_linear in swin_unet.cpp), so y and the
// 2C-wide hidden activation never leave registers: no LDS transpose, no HBM round trip.
//
// Weight ring.  All 4 waves of a workgroup walk the SAME fragment sequence (each wave owns MF x 16 other tokens),
// so the host lays the three weight matrices out as ONE stream of 1-KiB fragments in consumption order.  The
// workgroup pulls it in 8-KiB chunks: global -> registers is issued one chunk ahead (in flight during a whole
// chunk of MFMAs), registers -> LDS happens at the chunk boundary, one __syncthreads per chunk, two LDS buffers.
// Fragments are read back with lane-linear ds_read_b128 (conflict free).  L2 weight traffic drops 4x and the
// ~500-cycle L2 latency that v1 exposed on every fragment is hidden.
//
// HBM traffic per token: read att (2C B) + read x (2C B) + write x (2C B).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "swin_gelu.h"
#include "swin_kernels.h"

namespace nunif {

#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

// GELU(erf) = x * Phi(x).  libm erff is ~40 instructions; a first version used Abramowitz & Stegun 7.1.26
// (1 rcp + 1 exp + 6 fma).  Round-1 profile (profiles/r01_pmc_sq.txt): the kernel issued 14 VALU instructions per
// MFMA and was VALU-bound — the quarter-rate v_rcp + v_exp per element dominate.  Now: Phi(x) - 0.5 = xc * Q(xc^2) with
// xc = clamp(x, -4, 4) and Q a degree-8 minimax polynomial (Lawson fit on [0,4]; |Phi err| <= 4e-6, gelu abs err
// <= 1.1e-5 for |x| < 3 and <= 3e-5 * |x| beyond the clamp) — 12 full-rate VALU ops, no transcendental.
__device__ __forceinline__ float gelu_fast(float v) {
    const float xc = __builtin_amdgcn_fmed3f(v, -4.0f, 4.0f);      // one v_med3_f32, no canonicalising v_max
    const float u = xc * xc;
    float q = 8.063430101e-11f;
    q = fmaf(q, u, -7.003475758e-09f);
    q = fmaf(q, u, 2.716159007e-07f);
    q = fmaf(q, u, -6.295003997e-06f);
    q = fmaf(q, u, 9.890811950e-05f);
    q = fmaf(q, u, -1.133922332e-03f);
    q = fmaf(q, u, 9.877477530e-03f);
    q = fmaf(q, u, -6.641059600e-02f);
    q = fmaf(q, u, 3.989227099e-01f);
    return v * fmaf(xc, q, 0.5f);
}

constexpr int kChunkFrags = 8;   // 8 KiB per chunk: 256 threads x 2 x 16 B

// ABL != 0 are timing-only ablations used to find what bounds the kernel (NUNIF_TAIL_ABL in a NUNIF_BUILD_ABL=1 build;
// results are wrong; the shipping library instantiates ABL = 0 only):
// 1 = no GELU polynomial, 2 = no stores, 4 = no residual read, 8 = no MFMA in the MLP loop, 16 = no ring barrier
template <int C, int MF, int ABL = 0, int WAVES = 4, int CHF = kChunkFrags>
__global__ void __launch_bounds__(WAVES * 64, WAVES == 4 ? 2 : 1)
proj_mlp_kernel(const f16 *__restrict__ att, f16 *x, const f16 *__restrict__ wstream, int n_chunks,
                const float *__restrict__ bp, const float *__restrict__ b0, const float *__restrict__ b3, long M, int rev) {
    constexpr int KS = C / 32;       // K chunks of the C-wide GEMMs (proj, mlp.0)
    constexpr int NT = C / 16;       // 16-channel output tiles of a C-wide result
    constexpr int SH = 2 * C / 32;   // K chunks of the hidden (2C) dimension
    constexpr int CH = CHF;          // fragments (KiB) per ring chunk; 16 doubles the prefetch distance (4 staging regs)
    __shared__ f16x8 ring[2][CH * 64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r16 = lane & 15;
    const int grp = lane >> 4;
    const long m_base = ((long)(rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * WAVES + wave) * (MF * 16);   // snake order
    // NOTE: no early exit — every wave takes part in every chunk barrier; out-of-range rows are clamped + masked.

    // WAVES = 8: the workgroup covers twice the tokens per pass over the weight stream — at C = 192 the stream is
    // 360 KiB per workgroup, i.e. 1.8 GB of L2 -> LDS traffic per launch with 128-token workgroups (more than twice
    // the kernel's HBM traffic); 512 threads load one 16-byte piece of a chunk each
    const f16x8 *gsrc = reinterpret_cast<const f16x8 *>(wstream) + tid;
    f16x8 st0 = gsrc[0], st1, st2, st3;
    if constexpr (WAVES == 4) st1 = gsrc[256];       // chunk 0 in flight
    if constexpr (CH == 16) { st2 = gsrc[512]; st3 = gsrc[768]; }

    // fragment `fi` of the stream (fi is a compile-time constant at every call site after unrolling, and call
    // sites are in increasing fi order)
    auto wfrag = [&](int fi) -> f16x8 {
        const int c = fi / CH;
        if (fi % CH == 0) {
            ring[c & 1][tid] = st0;
            if constexpr (WAVES == 4) ring[c & 1][tid + 256] = st1;
            if constexpr (CH == 16) { ring[c & 1][tid + 512] = st2; ring[c & 1][tid + 768] = st3; }
            if constexpr (!(ABL & 16)) __syncthreads();
            if (c + 1 < n_chunks) {
                st0 = gsrc[(long)(c + 1) * (CH * 64)];
                if constexpr (WAVES == 4) st1 = gsrc[(long)(c + 1) * (CH * 64) + 256];
                if constexpr (CH == 16) { st2 = gsrc[(long)(c + 1) * (CH * 64) + 512]; st3 = gsrc[(long)(c + 1) * (CH * 64) + 768]; }
            }
        }
        return ring[c & 1][(fi % CH) * 64 + lane];
    };

    long row[MF];
    bool valid[MF];
    f16x8 yf[MF][KS];        // y (fp16) as B-operand fragments: slots 0-3 = tile 2s, slots 4-7 = tile 2s+1
    {
        f16x8 of[MF][KS];
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const long m = m_base + f * 16 + r16;
            valid[f] = m < M;
            row[f] = m < M ? m : M - 1;
            const f16 *p = att + row[f] * C + grp * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) of[f][ks] = *reinterpret_cast<const f16x8 *>(p + ks * 32);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {           // output-tile pair (2s, 2s+1) -> yf[.][s]
            f32x4 a0[MF], a1[MF];
#pragma unroll
            for (int f = 0; f < MF; ++f) { a0[f] = (f32x4){0.f, 0.f, 0.f, 0.f}; a1[f] = a0[f]; }
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
            const int n0 = 32 * s + 4 * grp;
            const float4 ba = *reinterpret_cast<const float4 *>(bp + n0);
            const float4 bb = *reinterpret_cast<const float4 *>(bp + n0 + 16);
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                f16x4 xa = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f}, xb = xa;
                if constexpr (!(ABL & 4))
                    run_to_pair(*reinterpret_cast<const f16x8 *>(x + row[f] * C + 32 * s + pair_run_channel(grp)), xa, xb);
                yf[f][s] = (f16x8){(f16)(a0[f][0] + ba.x + (float)xa[0]), (f16)(a0[f][1] + ba.y + (float)xa[1]),
                                   (f16)(a0[f][2] + ba.z + (float)xa[2]), (f16)(a0[f][3] + ba.w + (float)xa[3]),
                                   (f16)(a1[f][0] + bb.x + (float)xb[0]), (f16)(a1[f][1] + bb.y + (float)xb[1]),
                                   (f16)(a1[f][2] + bb.z + (float)xb[2]), (f16)(a1[f][3] + bb.w + (float)xb[3])};
            }
        }
    }
    // mlp.3 accumulators start from the residual y (+ b3): y needs no second copy
    f32x4 acc[NT][MF];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int n0 = 32 * s + 4 * grp;
        const float4 ca = *reinterpret_cast<const float4 *>(b3 + n0);
        const float4 cb = *reinterpret_cast<const float4 *>(b3 + n0 + 16);
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            acc[2 * s][f] = (f32x4){(float)yf[f][s][0] + ca.x, (float)yf[f][s][1] + ca.y,
                                    (float)yf[f][s][2] + ca.z, (float)yf[f][s][3] + ca.w};
            acc[2 * s + 1][f] = (f32x4){(float)yf[f][s][4] + cb.x, (float)yf[f][s][5] + cb.y,
                                        (float)yf[f][s][6] + cb.z, (float)yf[f][s][7] + cb.w};
        }
    }

    constexpr int F_MLP = 2 * KS * KS;            // first fragment of the mlp part of the stream
    constexpr int F_STEP = 2 * KS + NT;           // fragments per 32 hidden channels
    // a real loop (not unrolled): full unrolling lets the scheduler hoist ~100 loads and spill; the chunk-boundary
    // test inside wfrag() is wave-uniform, so a run-time fragment index costs one scalar branch
    // mlp.0 bias = C operand of the first MFMA of each hidden slice; the next slice's 8 values are fetched one trip ahead
    float4 ba_n = *reinterpret_cast<const float4 *>(b0 + 4 * grp);
    float4 bb_n = *reinterpret_cast<const float4 *>(b0 + 4 * grp + 16);
#pragma unroll 1
    for (int s = 0; s < SH; ++s) {               // 32 hidden channels at a time
        const float4 ba0 = ba_n, bb0 = bb_n;
        {
            const int sn = s + 1 < SH ? s + 1 : s;
            ba_n = *reinterpret_cast<const float4 *>(b0 + 32 * sn + 4 * grp);
            bb_n = *reinterpret_cast<const float4 *>(b0 + 32 * sn + 4 * grp + 16);
        }
        f32x4 h0[MF], h1[MF];
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            h0[f] = (f32x4){ba0.x, ba0.y, ba0.z, ba0.w};
            h1[f] = (f32x4){bb0.x, bb0.y, bb0.z, bb0.w};
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f16x8 wa = wfrag(F_MLP + s * F_STEP + ks * 2);
            const f16x8 wb = wfrag(F_MLP + s * F_STEP + ks * 2 + 1);
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                if constexpr (ABL & 8) { h0[f][0] += (float)wa[0]; h1[f][0] += (float)wb[0]; continue; }
                h0[f] = MFMA_16x16x32(wa, yf[f][ks], h0[f]);
                h1[f] = MFMA_16x16x32(wb, yf[f][ks], h1[f]);
            }
        }
        f16x8 hf[MF];
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            if constexpr (ABL & 1)
                hf[f] = (f16x8){(f16)h0[f][0], (f16)h0[f][1], (f16)h0[f][2], (f16)h0[f][3],
                                (f16)h1[f][0], (f16)h1[f][1], (f16)h1[f][2], (f16)h1[f][3]};
            else
                hf[f] = gelu8(h0[f], h1[f]);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f16x8 wv = wfrag(F_MLP + s * F_STEP + 2 * KS + nt);
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                if constexpr (ABL & 8) { acc[nt][f][0] += (float)wv[0] + (float)hf[f][nt & 7]; continue; }
                acc[nt][f] = MFMA_16x16x32(wv, hf[f], acc[nt][f]);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < NT / 2; ++p) {
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const f16x4 oa = {(f16)acc[2 * p][f][0], (f16)acc[2 * p][f][1], (f16)acc[2 * p][f][2], (f16)acc[2 * p][f][3]};
            const f16x4 ob = {(f16)acc[2 * p + 1][f][0], (f16)acc[2 * p + 1][f][1], (f16)acc[2 * p + 1][f][2],
                              (f16)acc[2 * p + 1][f][3]};
            const f16x8 o = pair_to_run(oa, ob);           // all lanes take part in the swap; only valid rows store
            if constexpr (ABL & 2) { if (o[0] != (f16)12345.f) continue; }
            if (valid[f]) *reinterpret_cast<f16x8 *>(x + row[f] * C + 32 * p + pair_run_channel(grp)) = o;
        }
    }
}

// ---- LDS-resident form (C = 96: the whole 90-KiB stream fits) ----------------------------------------------------------
// Same dataflow, but a persistent 8-wave workgroup copies the weight stream into LDS once and every wave then loops
// over its own 64-token groups with NO barrier: waves drift apart, so one wave's GELU / convert (VALU) phase overlaps
// its SIMD partner's MFMA phase instead of both stalling at a chunk barrier (same finding as swin_qkv_attn_r.hip).
template <int C, int MF, int WAVES, bool PF = false>
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
    // Window map (wm.on): token n = window * 36 + t of the SHIFTED 6x6 windows, t row-major in the window — the order
    // qkv_attn_r_kernel writes its window-major att map in; att is then [window][head 6][36][16].
    const int nwx = wm.on ? wm.W / 6 : 1, nwy = wm.on ? wm.H / 6 : 1;
    auto load_group = [&](long g, f16x8 (&of)[MF][KS], f16x8 (&xr)[MF][KS], long (&pixn)[MF]) {
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const long m = gmap(g) * (MF * 16) + f * 16 + r16;
            const long r = m < M ? m : M - 1;
            const f16 *p;
            if (wm.on) {
                const int w = (int)(r / 36), t = (int)(r - 36L * w);
                const int wx = w % nwx, t2 = w / nwx;
                const int wy = t2 % nwy, b = t2 / nwy;
                const int iy = t / 6, ix = t - 6 * iy;
                int yy = wy * 6 + iy + wm.shift, xx = wx * 6 + ix + wm.shift;
                if (yy >= wm.H) yy -= wm.H;
                if (xx >= wm.W) xx -= wm.W;
                pixn[f] = ((long)b * wm.H + yy) * wm.W + xx;
                // lane group g holds channels 32 ks + 8 g .. + 7 = head 2 ks + (g >> 1), dims 8 (g & 1) ..
                p = att + (((long)w * 6 + (grp >> 1)) * 36 + t) * 16 + 8 * (grp & 1);
            } else {
                pixn[f] = r;
                p = att + r * C + grp * 8;
            }
            const f16 *px = x + pixn[f] * C + grp * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                of[f][ks] = *reinterpret_cast<const f16x8 *>(p + (wm.on ? ks * (2 * 36 * 16) : ks * 32));
                xr[f][ks] = *reinterpret_cast<const f16x8 *>(px + ks * 32);      // B-operand layout: the residual rides an MFMA
            }
        }
    };
    f16x8 ofn[PF ? MF : 1][PF ? KS : 1];
    f16x8 xrn[PF ? MF : 1][PF ? KS : 1];
    long pixnn[MF];
    const long g_first = (long)blockIdx.x * WAVES + wave;
    if constexpr (PF) {
        if (g_first < n_groups) load_group(g_first, ofn, xrn, pixnn);
    }
#pragma unroll 1
    for (long g = g_first; g < n_groups; g += gstride) {
        const long m_base = gmap(g) * (MF * 16);
        // opaque per-iteration copies: keeps LICM from hoisting 48 registers of (loop-invariant) LDS bias reads
        const float *lbp = bl, *lb0 = bl + C, *lb3 = bl + 3 * C;
        asm volatile("" : "+v"(lbp), "+v"(lb0), "+v"(lb3));
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
                load_group(g + gstride < n_groups ? g + gstride : g, ofn, xrn, pixnn);
            } else {
                load_group(g, of, xr, row);
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
                hf[f] = gelu8(h0[f], h1[f]);
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
                const long m = row[f];
                const int px = (int)(m % ti.W);
                const long t2 = m / ti.W;
                const int py = (int)(t2 % ti.H), pb = (int)(t2 / ti.H);
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

constexpr int proj_mlp_stream_frags_c(int C) { return 2 * (C / 32) * (C / 32) + (2 * C / 32) * (2 * (C / 32) + C / 16); }

int proj_mlp_stream_frags(int C) {
    const int KS = C / 32, NT = C / 16, SH = 2 * C / 32;
    return 2 * KS * KS + SH * (2 * KS + NT);
}

int launch_proj_mlp(const f16 *att, f16 *x, const f16 *wstream, const float *bp, const float *b0, const float *b3,
                    long M, int C, hipStream_t s, const TailToImage *to_image, int rev, const WinMap *wmap) {
    if (M == 0) return NUNIF_HIP_OK;
    NUNIF_REQUIRE(!to_image || (C == 96 && !getenv("NUNIF_TAIL_RING") && to_image->n_real <= 16),
                  "proj_mlp: the fused image head needs the resident C = 96 kernel");
    NUNIF_REQUIRE(!(wmap && wmap->on) || (C == 96 && !getenv("NUNIF_TAIL_RING")), "proj_mlp: the window map needs the resident C = 96 kernel");
    static const bool ring96 = getenv("NUNIF_TAIL_RING") != nullptr;     // A/B switch: the round-1 ring version
    static const int variant = getenv("NUNIF_TAIL_VARIANT") ? atoi(getenv("NUNIF_TAIL_VARIANT")) : 6;
    // profiler classes are named after the kernel symbol so that they line up with rocprofv3's kernel stats
    const char *sym = C == 192 ? "proj_mlp_kernel<192,2,0>" : ring96 ? "proj_mlp_kernel<96,4,0>"
                      : variant == 6 ? "proj_mlp_r_kernel<96,2,8,true>" : "proj_mlp_r_kernel<96,*>";
    ProfScope ps(sym, s, 2.0 * (double)M * C * C * 5.0, (double)M * C * 2.0 * 3.0);
    const int n_chunks = (proj_mlp_stream_frags(C) + kChunkFrags - 1) / kChunkFrags;
    if (C == 96 && !ring96) {
        constexpr size_t smem = (size_t)(proj_mlp_stream_frags_c(96) + 3) * 1024 + (4 * 96 + 16) * 4;
        TailToImage ti;
        memset(&ti, 0, sizeof(ti));
        if (to_image) ti = *to_image;
        WinMap wm;
        memset(&wm, 0, sizeof(wm));
        if (wmap) wm = *wmap;
        NUNIF_REQUIRE(!wm.on || (wm.H % 6 == 0 && wm.W % 6 == 0 && M % ((long)wm.H * wm.W) == 0), "proj_mlp: window map geometry");
        auto go = [&](auto kern, int mf, int waves) -> int {
            static bool configured[8] = {false};
            if (!configured[variant & 7]) {
                NUNIF_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
                configured[variant & 7] = true;
            }
            const long groups = (M + mf * 16 - 1) / (mf * 16);
            const unsigned blocks = (unsigned)std::min<long>((groups + waves - 1) / waves, 256);
            kern<<<blocks, waves * 64, smem, s>>>(att, x, wstream, bp, b0, b3, M, ti, rev, wm);
            return NUNIF_HIP_OK;
        };
        int rc;
        switch (variant) {
            case 1: rc = go(proj_mlp_r_kernel<96, 2, 8>, 2, 8); break;
            case 2: rc = go(proj_mlp_r_kernel<96, 2, 12>, 2, 12); break;
            case 3: rc = go(proj_mlp_r_kernel<96, 2, 16>, 2, 16); break;
            case 4: rc = go(proj_mlp_r_kernel<96, 3, 8>, 3, 8); break;
            case 5: rc = go(proj_mlp_r_kernel<96, 3, 12>, 3, 12); break;
            case 6: rc = go(proj_mlp_r_kernel<96, 2, 8, true>, 2, 8); break;
            case 7: rc = go(proj_mlp_r_kernel<96, 3, 8, true>, 3, 8); break;
            default: rc = go(proj_mlp_r_kernel<96, 4, 8>, 4, 8); break;
        }
        if (rc) return rc;
    } else if (C == 96) {
        constexpr int MF = 4;
        const unsigned blocks = (unsigned)((M + 4 * MF * 16 - 1) / (4 * MF * 16));
        proj_mlp_kernel<96, MF><<<blocks, 256, 0, s>>>(att, x, wstream, n_chunks, bp, b0, b3, M, rev);
    } else if (C == 192) {
        constexpr int MF = 2;
        const unsigned blocks = (unsigned)((M + 4 * MF * 16 - 1) / (4 * MF * 16));
#ifdef NUNIF_ABLATIONS
        // timing-only ablations (wrong results): only in a NUNIF_BUILD_ABL=1 build, never in the shipping library
        static const int abl = getenv("NUNIF_TAIL_ABL") ? atoi(getenv("NUNIF_TAIL_ABL")) : 0;
        switch (abl) {
            case 1: proj_mlp_kernel<192, MF, 1><<<blocks, 256, 0, s>>>(att, x, wstream, n_chunks, bp, b0, b3, M, rev); break;
            case 2: proj_mlp_kernel<192, MF, 2><<<blocks, 256, 0, s>>>(att, x, wstream, n_chunks, bp, b0, b3, M, rev); break;
            case 4: proj_mlp_kernel<192, MF, 4><<<blocks, 256, 0, s>>>(att, x, wstream, n_chunks, bp, b0, b3, M, rev); break;
            case 8: proj_mlp_kernel<192, MF, 8><<<blocks, 256, 0, s>>>(att, x, wstream, n_chunks, bp, b0, b3, M, rev); break;
            case 9: proj_mlp_kernel<192, MF, 9><<<blocks, 256, 0, s>>>(att, x, wstream, n_chunks, bp, b0, b3, M, rev); break;
            case 16: proj_mlp_kernel<192, MF, 16><<<blocks, 256, 0, s>>>(att, x, wstream, n_chunks, bp, b0, b3, M, rev); break;
            case 6: proj_mlp_kernel<192, MF, 6><<<blocks, 256, 0, s>>>(att, x, wstream, n_chunks, bp, b0, b3, M, rev); break;
            default: proj_mlp_kernel<192, MF><<<blocks, 256, 0, s>>>(att, x, wstream, n_chunks, bp, b0, b3, M, rev); break;
        }
#else
        proj_mlp_kernel<192, MF><<<blocks, 256, 0, s>>>(att, x, wstream, n_chunks, bp, b0, b3, M, rev);
#endif
    } else {
        set_error("proj_mlp: channel count %d unsupported (96, 192)", C);
        return NUNIF_HIP_EUNSUPPORTED;
    }
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

}  // namespace nunif
