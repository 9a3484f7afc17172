// Device kernels of the waifu2x CUNet forward for gfx950.
//
// Reference ops replaced: waifu2x/models/cunet.py — UNetConv :10-28 (3x3 VALID conv +
// LeakyReLU(0.1) x2), conv 2x2 stride 2 :34,:76,:78, the cropped skip additions :63,:111,:116, conv_bottom + the
// cascade sum `crop(z1,20) + unet2(z1)` + clamp :183-196; nunif/modules/attention.py SEBlock :29-44.
// (ConvTranspose2d 2x2 stride 2 runs on gemm_kernel's pixel-shuffle mode, swin_kernels.hip.)
//
// conv_kernel<NT,MF>: implicit-GEMM conv on v_mfma_f32_16x16x32_f16, NHWC fp16 maps, fp32 accumulation.  Same operand
// orientation as the swin kernels (weights = A, activations = B => a lane's accumulator is 4 consecutive output
// channels of one pixel).  Cout <= 256, so ALL output-channel tiles of a pixel group stay in accumulators while the
// kernel walks K = taps x Cin: activations stream from HBM exactly once (next k-step prefetched into registers),
// weights stream [k-step][n-tile] through the 2 x 8 KiB LDS ring shared by the 4 waves.
#include <algorithm>
#include <cstdlib>

#include "swin_kernels.h"

namespace nunif {

#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

// RES = true: the WHOLE weight stream (ksteps x NT fragments <= 80 KiB) is copied into LDS once and the workgroup is persistent
// over pixel groups — no barrier and no weight latency inside the k loop.  With NT x MF <= 16 MFMAs per k-step the ring's
// one-chunk prefetch distance (2 k-steps, ~500 cycles) is shorter than an L2 round trip and every chunk boundary stalled
// all four waves (DPT-head 64 -> 64 3x3 convs: 81 TFLOP/s, profiles/r01d_kernel_stats_iw3_sched.csv).
template <int NT, int MF, bool RES, bool A2 = false>      // A2: a second input is added element-wise (cunet skip)
__global__ void __launch_bounds__(256) conv_kernel(ConvArgs g) {
    constexpr int CH = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_conv[];
    f16x8 *ring = reinterpret_cast<f16x8 *>(smem_conv);                  // ring form: [2][CH * 64]; RES: [ksteps * NT][64]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r16 = lane & 15;
    const int grp = lane >> 4;
    const long M = (long)g.B * g.Ho * g.Wo;
    const long n_groups = (M + 4 * MF * 16 - 1) / (4 * MF * 16);
    const f16x8 *gsrc = reinterpret_cast<const f16x8 *>(g.wstream);      // zero-padded by 16 KiB on the host
    if (RES) {
        const int nfrag = g.kh * g.kw * (g.Cin >> 5) * NT;
        for (int i = tid; i < nfrag * 64; i += 256) ring[i] = gsrc[i];
        __syncthreads();
    }
    int gi = blockIdx.x;                                                 // ring form: gridDim.x == n_groups, ONE trip (the
    do {                                                                 // loop condition folds to false at compile time)
    const long m_base = ((long)gi * 4 + wave) * (MF * 16);

    // ring form: the weight chunks (8 KiB) are requested THREE chunks ahead into a register queue (st / sq / sr) and written to
    // the LDS ring at their own chunk boundary.  One chunk ahead (round 1) is ~500 cycles; the weights of a launch are read once
    // and come from HBM, so on the small maps of the DPT head (6-22 workgroups, nothing else to switch to) every chunk boundary
    // waited a full memory round trip: 1.3 us per k-step on the 14 x 25 map (profiles/r02 kernel trace).
    f16x8 st0, st1, sq0, sq1, sr0, sr1;
    const int n_chunks = (g.kh * g.kw * (g.Cin >> 5) * NT + CH - 1) / CH;
    if (!RES) {                                             // (chunk n_chunks is the last one that lies inside the 16-KiB zero padding)
        const int c1 = min(1, n_chunks), c2 = min(2, n_chunks);
        st0 = gsrc[tid]; st1 = gsrc[tid + 256];
        sq0 = gsrc[c1 * CH * 64 + tid]; sq1 = gsrc[c1 * CH * 64 + tid + 256];
        sr0 = gsrc[c2 * CH * 64 + tid]; sr1 = gsrc[c2 * CH * 64 + tid + 256];
    }
    auto wfrag = [&](int fi) -> f16x8 {
        if (RES) return ring[fi * 64 + lane];
        const int c = fi / CH;
        if (fi % CH == 0) {
            ring[(c & 1) * (CH * 64) + tid] = st0;
            ring[(c & 1) * (CH * 64) + tid + 256] = st1;
            __syncthreads();
            st0 = sq0; st1 = sq1; sq0 = sr0; sq1 = sr1;
            const int cn = min(c + 3, n_chunks);            // the stream is zero-padded by two chunks (16 KiB) on the host
            sr0 = gsrc[cn * (CH * 64) + tid];
            sr1 = gsrc[cn * (CH * 64) + tid + 256];
        }
        return ring[(c & 1) * (CH * 64) + (fi % CH) * 64 + lane];
    };

    // per-lane pixel bases (element offsets) of the MF tiles
    long base[MF], base2[MF];
    int pb[MF], py[MF], px[MF];
    bool valid[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        long m = m_base + f * 16 + r16;
        valid[f] = m < M;
        if (m >= M) m = M - 1;
        px[f] = (int)(m % g.Wo);
        const long t = m / g.Wo;
        py[f] = (int)(t % g.Ho);
        pb[f] = (int)(t / g.Ho);
        base[f] = (((long)pb[f] * g.Hi + (long)py[f] * g.stride) * g.Wi + (long)px[f] * g.stride) * g.Cin + 8 * grp;
        base2[f] = g.a2 ? (((long)pb[f] * g.H2 + (long)py[f] * g.stride + g.crop2) * g.W2 + (long)px[f] * g.stride +
                           g.crop2) * g.Cin + 8 * grp : 0;
    }
    const int cpt = g.Cin >> 5;                     // 32-channel chunks per tap
    const int ksteps = g.kh * g.kw * cpt;
    // load_x only ISSUES the gathers (clamped coordinates, so that every lane loads); what has to happen to the loaded values —
    // zeroing the taps that fell into the zero padding, the pre-activation ReLU — is done by finish_x right in front of the
    // MFMAs that consume them.  Doing it at load time (round 1) put an s_waitcnt vmcnt(0) behind every load: the prefetch
    // buffers never had more than one load in flight.
    // ONE branch-free address path for the three padding modes (a uniform branch per mode inside the unrolled loops made hipcc
    // reuse load-destination registers across the arms and guard every arm with vmcnt(0)): tap coordinate = pixel * stride +
    // tap - pad, clamped into the map; replicate padding keeps the clamped value, zero padding remembers which lanes were outside.
    const int pad = g.rpad ? g.rpad : g.zpad;
    auto load_x = [&](int ks, f16x8 (&dst)[MF], f16x8 (&dst2)[A2 ? MF : 1], unsigned &inb_mask) {
        const int tap = ks / cpt, c0 = (ks - tap * cpt) << 5;
        const int dy = tap / g.kw, dx = tap - dy * g.kw;
        inb_mask = 0u;
        if constexpr (A2) {         // second input (cropped U-Net skip; VALID, stride as given): both raw, added by finish_x
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                dst[f] = *reinterpret_cast<const f16x8 *>(g.a + base[f] + ((long)dy * g.Wi + dx) * g.Cin + c0);
                dst2[f] = *reinterpret_cast<const f16x8 *>(g.a2 + base2[f] + ((long)dy * g.W2 + dx) * g.Cin + c0);
            }
            return;
        }
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int yy = py[f] * g.stride + dy - pad, xx = px[f] * g.stride + dx - pad;
            const int yc = min(max(yy, 0), g.Hi - 1), xc = min(max(xx, 0), g.Wi - 1);
            dst[f] = *reinterpret_cast<const f16x8 *>(g.a + (((long)pb[f] * g.Hi + yc) * g.Wi + xc) * g.Cin + 8 * grp + c0);
            inb_mask |= ((yy == yc && xx == xc) ? 1u : 0u) << f;
        }
    };
    auto finish_x = [&](f16x8 (&x)[MF], const f16x8 (&x2)[A2 ? MF : 1], unsigned inb_mask) {
        if constexpr (A2) {
#pragma unroll
            for (int f = 0; f < MF; ++f) x[f] += x2[f];
            return;
        }
        if (!(g.zpad || g.relu_in) || g.rpad) return;
        const f16x8 z8 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            f16x8 v = ((inb_mask >> f) & 1u) ? x[f] : z8;
            if (g.relu_in) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = v[j] > (f16)0.f ? v[j] : (f16)0.f;
            }
            x[f] = v;
        }
    };

    f32x4 acc[NT][MF];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int f = 0; f < MF; ++f) acc[nt][f] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // NB register buffers of activation fragments: the gathers of k-step ks + NB - 1 are issued in front of the MFMAs of step ks.
    // With NT x MF <= 16 MFMAs per k-step a one-step distance is ~256 cycles — far less than an L2 / HBM round trip, and the
    // small maps these shapes run on (DPT head, side nets) leave 1-2 waves per SIMD to hide it: three steps ahead there.
    constexpr int NB = (NT * MF <= 16) ? 4 : 2;
    f16x8 xq[NB][MF], xq2[NB][A2 ? MF : 1];
    unsigned xm[NB];
#pragma unroll
    for (int j = 0; j < NB - 1; ++j)
        if (j < ksteps) load_x(j, xq[j], xq2[j], xm[j]);
#pragma unroll 1
    for (int ks = 0; ks < ksteps; ks += NB) {        // NB k-steps per trip: statically named register buffers
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (ks + j < ksteps) {
                if (ks + j + NB - 1 < ksteps)
                    load_x(ks + j + NB - 1, xq[(j + NB - 1) % NB], xq2[(j + NB - 1) % NB], xm[(j + NB - 1) % NB]);
                finish_x(xq[j], xq2[j], xm[j]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f16x8 w = wfrag((ks + j) * NT + nt);
#pragma unroll
                    for (int f = 0; f < MF; ++f) acc[nt][f] = MFMA_16x16x32(w, xq[j][f], acc[nt][f]);
                }
            }
        }
    }

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n0 = nt * 16 + grp * 4;
        const float4 bv = *reinterpret_cast<const float4 *>(g.bias + n0);
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            if (!valid[f]) continue;
            float v[4] = {acc[nt][f][0] + bv.x, acc[nt][f][1] + bv.y, acc[nt][f][2] + bv.z, acc[nt][f][3] + bv.w};
            if (g.act == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] >= 0.f ? v[r] : v[r] * g.slope;
            } else if (g.act == 3) {                       // ReLU
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (g.out32) {
                // image head: planar fp32, optional `+ crop(add32)` and clamp  (cunet.py:183-196)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + r;
                    if (n >= g.n_real) continue;
                    float o = v[r];
                    if (g.add32)
                        o += g.add32[(((long)pb[f] * g.n_real + n) * g.addH + py[f] + g.add_crop) * g.addW + px[f] + g.add_crop];
                    if (g.clamp01) o = fminf(fmaxf(o, 0.f), 1.f);
                    g.out32[(((long)pb[f] * g.n_real + n) * g.Ho + py[f]) * g.Wo + px[f]] = o;
                }
            } else if (n0 < g.n_real) {
                const long off = (((long)pb[f] * g.Ho + py[f]) * g.Wo + px[f]) * (g.ldo > 0 ? g.ldo : g.n_real) + n0;
                if (g.res) {
                    const f16x4 rv = *reinterpret_cast<const f16x4 *>(g.res + off);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
                }
                if (g.res2) {
                    const f16x4 rv = *reinterpret_cast<const f16x4 *>(g.res2 + off);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
                }
                const f16x4 ov = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                *reinterpret_cast<f16x4 *>(g.out + off) = ov;
            }
        }
    }
    gi += gridDim.x;
    } while (RES && gi < (int)n_groups);   // pixel groups
}

// read per launch (the lookup is noise next to a launch) so that a test can compare both forms inside one process
// (taking the resident form for EVERY launch size measured 3.8 % slower on the iw3 frame: profiles/r01d_ab_conv_res.txt)
static inline long conv_res_min_groups() { return 512L; }
static inline int conv_res_enabled() { const char *e = getenv("NUNIF_CONV_RES"); return e ? atoi(e) : 1; }

template <int NT, int MF>
static int launch_conv_t(const ConvArgs &g, hipStream_t s) {
    const long M = (long)g.B * g.Ho * g.Wo;
    const long rows = 4 * MF * 16;
    const long n_groups = (M + rows - 1) / rows;
    const size_t res_bytes = (size_t)g.kh * g.kw * (g.Cin >> 5) * NT * 1024;
    // resident weights pay when the ring cannot hide its refills (few MFMAs per k-step) AND every workgroup has many pixel groups
    // to amortise its 36-72 KiB copy over.  Measured (profiles/r01d_ab_conv_res.txt, whole iw3 frame, same box): >= 512 groups
    // only: -0.8 %; all launch sizes (NUNIF_CONV_RES_MIN_GROUPS=1): +3.8 % — for the small DPT-head maps one trip per workgroup
    // does not pay for the copy, although both forms are bit-identical on every size (the conv tests pass either way).
    if (conv_res_enabled() && !g.a2 && NT * MF <= 16 && res_bytes <= 80 * 1024 && n_groups >= conv_res_min_groups()) {
        static bool configured = false;
        if (!configured) {
            NUNIF_HIP_CHECK(hipFuncSetAttribute((const void *)conv_kernel<NT, MF, true>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
            configured = true;
        }
        conv_kernel<NT, MF, true><<<(unsigned)std::min<long>(n_groups, 512), 256, res_bytes, s>>>(g);
    } else if (g.a2) {
        conv_kernel<NT, MF, false, true><<<(unsigned)n_groups, 256, 16 * 1024, s>>>(g);
    } else {
        conv_kernel<NT, MF, false><<<(unsigned)n_groups, 256, 16 * 1024, s>>>(g);
    }
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

int launch_conv(const ConvArgs &g, hipStream_t s) {
    NUNIF_REQUIRE(g.Cin % 32 == 0 && g.N % 16 == 0, "conv: Cin=%d N=%d not aligned", g.Cin, g.N);
    NUNIF_REQUIRE(!g.rpad || (g.stride == 1 && !g.a2), "conv: replicate padding needs stride 1 and a single input");
    NUNIF_REQUIRE(!(g.zpad || g.relu_in) || (!g.a2 && !g.rpad), "conv: zero padding / relu_in need a single input");
    const long M = (long)g.B * g.Ho * g.Wo;
    if (M == 0) return NUNIF_HIP_OK;
    if (conv3_dma_applies(g)) return launch_conv3_dma(g, s);          // large grids: persistent, halo + weights by LDS-DMA
    if (conv3_lds_applies(g)) return launch_conv3_lds(g, s);
    NUNIF_REQUIRE(!g.cmaj, "conv: a chunk-major weight stream needs the 3x3 stride-1 LDS kernel (NUNIF_CONV3_LDS=0 set?)");
    const double K = (double)g.kh * g.kw * g.Cin;
    const double flops = 2.0 * (double)M * K * g.n_real;
    const double bytes = (double)g.B * g.Hi * g.Wi * g.Cin * 2.0 * (g.a2 ? 2.0 : 1.0) +
                         (double)M * g.n_real * (g.out32 ? 4.0 : 2.0);
    switch (g.N / 16) {
        case 1: { ProfScope ps("conv_kernel<1,4>", s, flops, bytes); return launch_conv_t<1, 4>(g, s); }
        case 2: { ProfScope ps("conv_kernel<2,4>", s, flops, bytes); return launch_conv_t<2, 4>(g, s); }
        case 4: { ProfScope ps("conv_kernel<4,4>", s, flops, bytes); return launch_conv_t<4, 4>(g, s); }
        case 6: { ProfScope ps("conv_kernel<6,4>", s, flops, bytes); return launch_conv_t<6, 4>(g, s); }
        case 8: { ProfScope ps("conv_kernel<8,4>", s, flops, bytes); return launch_conv_t<8, 4>(g, s); }
        case 12: { ProfScope ps("conv_kernel<12,2>", s, flops, bytes); return launch_conv_t<12, 2>(g, s); }
        case 16: { ProfScope ps("conv_kernel<16,2>", s, flops, bytes); return launch_conv_t<16, 2>(g, s); }
        case 24: { ProfScope ps("conv_kernel<24,1>", s, flops, bytes); return launch_conv_t<24, 1>(g, s); }
        default:
            set_error("conv: unsupported Cout=%d", g.N);
            return NUNIF_HIP_EUNSUPPORTED;
    }
}

// ---- first conv of a UNet: 3 -> Cout (<= 64) 3x3 VALID + LeakyReLU on the VALU (K = 27) ---------------------------------
// in: tile mode [B,3,T,T] fp32 or, frame mode, the frame [3,H,W] with replicate-pad + tile slicing folded in.
__global__ void __launch_bounds__(256) c3_conv_kernel(C3ConvArgs a) {
    extern __shared__ float sw[];   // [27][C] then bias[C]
    const int C = a.C;
    for (int i = threadIdx.x; i < 28 * C; i += blockDim.x) {
        if (i < 27 * C) { const int co = i % C, t = i / C; sw[i] = a.w[co * 27 + t]; }
        else sw[i] = a.bias[i - 27 * C];
    }
    __syncthreads();
    const int S = a.T - 2;
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
        const int y0 = ti * a.istep - a.pad_t + y, x0 = tj * a.istep - a.pad_l + x;
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
                    in[ci * 9 + ky * 3 + kx] = a.x[(((long)b * 3 + ci) * a.T + (y + ky)) * a.T + (x + kx)];
    }
    f16 *o = a.out + idx * C;
    for (int c0 = 0; c0 < C; c0 += 8) {
        f16x8 ov;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float acc = sw[27 * C + c0 + j];
#pragma unroll
            for (int t2 = 0; t2 < 27; ++t2) acc = fmaf(in[t2], sw[t2 * C + c0 + j], acc);
            ov[j] = (f16)(acc >= 0.f ? acc : acc * a.slope);
        }
        *reinterpret_cast<f16x8 *>(o + c0) = ov;
    }
}

int launch_c3_conv(const C3ConvArgs &a, hipStream_t s) {
    NUNIF_REQUIRE(a.C % 8 == 0 && a.C <= 64 && a.T > 2, "c3_conv: bad shape");
    const int S = a.T - 2;
    const long total = (long)a.B * S * S;
    ProfScope ps("c3_conv_kernel", s, 2.0 * 27 * a.C * (double)total, (double)total * (a.C * 2.0 + 12.0));
    c3_conv_kernel<<<(unsigned)((total + 255) / 256), 256, (size_t)(28 * a.C) * sizeof(float), s>>>(a);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

// ---- squeeze-excitation: global average pool -> 1x1 -> ReLU -> 1x1 -> sigmoid -> channel scale ------------------------------
constexpr int kSePoolBlocks = 128;

// Deterministic two-stage pooling (no float atomics: the result must not depend on launch order or tile batch).
// A thread owns 8 consecutive channels of every (256 / (C / 8))-th pixel of its block's share: 16-byte loads, two in flight
// (round 4 read one fp16 per lane per trip — 128 bytes per wave-load — and the pool, a third of the block's bytes, took most of
// its time).  The fixed per-thread order and the fixed order of the LDS reduction keep the sums independent of everything but
// the image.
__global__ void __launch_bounds__(256) se_pool_kernel(const f16 *__restrict__ x, float *partial, long hw, int C) {
    const int b = blockIdx.y;
    const int C8 = C / 8, ppb = 256 / C8;                   // pixels per block and trip
    const int c8 = threadIdx.x % C8, pl = threadIdx.x / C8;
    const f16 *img = x + (long)b * hw * C + c8 * 8;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const long step = (long)gridDim.x * ppb;
    long p = (long)blockIdx.x * ppb + pl;
    for (; p + step < hw; p += 2 * step) {
        const f16x8 v0 = *reinterpret_cast<const f16x8 *>(img + p * C);
        const f16x8 v1 = *reinterpret_cast<const f16x8 *>(img + (p + step) * C);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += (float)v0[j];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += (float)v1[j];
    }
    if (p < hw) {
        const f16x8 v0 = *reinterpret_cast<const f16x8 *>(img + p * C);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += (float)v0[j];
    }
    __shared__ float sh[256 * 8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sh[threadIdx.x * 8 + j] = s[j];
    __syncthreads();
    if (threadIdx.x < C) {
        const int c = threadIdx.x, cc8 = c / 8, j = c % 8;
        float t = 0.f;
        for (int k = 0; k < ppb; ++k) t += sh[(k * C8 + cc8) * 8 + j];
        partial[((long)b * kSePoolBlocks + blockIdx.x) * C + c] = t;
    }
}

__global__ void __launch_bounds__(256)
se_mlp_kernel(const float *__restrict__ partial, float *scale, const float *__restrict__ w1,
              const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ b2, float inv_hw,
              int C) {
    __shared__ float mean[256], hid[32];
    const int b = blockIdx.x, t = threadIdx.x, R = C / 8;
    if (t < C) {
        float s = 0.f;
        for (int k = 0; k < kSePoolBlocks; ++k) s += partial[((long)b * kSePoolBlocks + k) * C + t];
        mean[t] = s * inv_hw;
    }
    __syncthreads();
    if (t < R) {
        float a = b1[t];
        for (int c = 0; c < C; ++c) a += w1[t * C + c] * mean[c];
        hid[t] = fmaxf(a, 0.f);
    }
    __syncthreads();
    if (t < C) {
        float a = b2[t];
        for (int j = 0; j < R; ++j) a += w2[t * R + j] * hid[j];
        scale[b * C + t] = 1.0f / (1.0f + __expf(-a));
    }
}

__global__ void __launch_bounds__(256) se_scale_kernel(f16 *x, const float *__restrict__ scale, long hw, int C, long total8) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;      // one f16x8 per thread
    if (i >= total8) return;
    const long e = i * 8;
    const int c = (int)(e % C);
    const int b = (int)(e / (hw * C));
    f16x8 v = *reinterpret_cast<f16x8 *>(x + e);
    const float *sc = scale + b * C + c;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (f16)((float)v[j] * sc[j]);
    *reinterpret_cast<f16x8 *>(x + e) = v;
}

int launch_se(f16 *x, float *sums, float *scale, const float *w1, const float *b1, const float *w2, const float *b2,
              int B, long hw, int C, hipStream_t s, int scale_in_consumer) {
    NUNIF_REQUIRE(C % 8 == 0 && C <= 256 && 256 % C == 0, "se: C=%d unsupported", C);
    ProfScope ps("se_block", s, 0.0, (double)B * hw * C * 2.0 * 3.0);
    dim3 g1(kSePoolBlocks, B);                    // sums: [B][kSePoolBlocks][C] partials
    se_pool_kernel<<<g1, 256, 0, s>>>(x, sums, hw, C);
    se_mlp_kernel<<<B, 256, 0, s>>>(sums, scale, w1, b1, w2, b2, 1.0f / (float)hw, C);
    // scale_in_consumer: the one consumer of the map multiplies its fragments by `scale` as it loads them (GemmArgs::in_scale):
    // no read + write of the whole map here
    if (scale_in_consumer) { NUNIF_LAUNCH_CHECK(); return NUNIF_HIP_OK; }
    const long total8 = (long)B * hw * C / 8;
    se_scale_kernel<<<(unsigned)((total8 + 255) / 256), 256, 0, s>>>(x, scale, hw, C, total8);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

}  // namespace nunif
