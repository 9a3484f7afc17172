// 3x3 stride-1 "same" conv (zero or replicate padding 1) with the INPUT TILE STAGED IN LDS, for gfx950.
//
// conv_kernel (cunet_kernels.hip) gathers the B operand of every tap straight from global memory: each input pixel is
// fetched nine times and a k-step's MFMAs wait for gathers that were issued at most three k-steps earlier.  On the maps the
// DPT head of the depth nets runs on (14 x 25 ... 224 x 392, iw3/depth_anything_model.py's external network; 64 / 128
// channels) that is a chain of memory round trips: 27-47 us per launch regardless of the map size (kernel trace, DESIGN.md
// 4.10).  Here a workgroup owns an 8 x 32 output patch: its (8+2) x (32+2) input halo with ALL Cin channels is loaded once,
// every load in flight at the same time (one round trip), padded / ReLU'd while it is written to LDS, and the 9 x Cin/32
// k-steps read their B fragments from LDS only (pixel stride Cin*2 + 16 bytes: the 16 pixels of a fragment hit 16 distinct
// bank quads).  Weights come through the same 2 x 8-KiB ring as conv_kernel, requested three chunks ahead.
// Same contract as conv_kernel for its case (ConvArgs: kh = kw = 3, stride 1, zpad = 1 or rpad = 1, one input, NHWC fp16
// output with bias, LeakyReLU / ReLU, up to two residuals, ldo), same accumulation order over k — results are identical.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "swin_kernels.h"

namespace nunif {

#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

constexpr int kC3TH = 8, kC3TW = 32;                     // output patch of a workgroup (4 waves x 2 rows x 32 columns)
constexpr int kC3HH = kC3TH + 2, kC3HW = kC3TW + 2;      // halo

// RESW: the whole weight stream (9 x Cin/32 x NT KiB) fits next to the halo and is copied into LDS up front, all loads in flight
// together with the halo's: the weights of a launch are read once, so they come from HBM, and through the ring every 8-KiB
// chunk boundary waited for a request only three chunks (~0.7 us of MFMAs) old — 23-26 us per launch whatever the map size.
template <int NT, bool RESW>
__global__ void __launch_bounds__(256) conv3_lds_kernel(ConvArgs g) {
    constexpr int CH = 8, MF = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_c3[];
    f16x8 *ring = reinterpret_cast<f16x8 *>(smem_c3);                    // ring: [2][CH * 64]; RESW: [9 * Cin/32 * NT][64]
    unsigned char *halo = smem_c3 + (RESW ? 9 * ((g.cmaj ? g.cmaj : g.Cin) >> 5) * NT : 2 * CH) * 1024;   // [kC3HH * kC3HW pixels][cin * 2 + 16 B]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, r16 = lane & 15, grp = lane >> 4;
    // split contraction (g.cmaj, grid.y = chunks): this workgroup convolves channels [cmaj * blockIdx.y, + cmaj) of rows that are
    // lda channels wide and writes an fp32 partial; otherwise cin = lda = g.Cin
    const int cin = g.cmaj ? g.cmaj : g.Cin, lda = g.Cin;
    const int pad = (g.zpad || g.rpad) ? 1 : 0;                          // 0: VALID 3x3 (Ho = Hi - 2), the waifu2x conv nets
    const int pstride = cin * 2 + 16;
    const int tiles_x = (g.Wo + kC3TW - 1) / kC3TW, tiles_y = (g.Ho + kC3TH - 1) / kC3TH;
    const int cpt = cin >> 5;
    const f16 *a_in = g.a + (g.cmaj ? (int)blockIdx.y * cin : 0);
    const f16x8 *gsrc = reinterpret_cast<const f16x8 *>(g.wstream) +    // zero-padded by 16 KiB on the host
                        (g.cmaj ? (long)blockIdx.y * 9 * cpt * NT * 64 : 0);
    const int ksteps = 9 * cpt;
    const int n_chunks = (ksteps * NT + CH - 1) / CH;
    // loads in flight per thread and batch: weights, halo.  UH = 11 covers the whole 10 x 34 x 64-channel halo (2 720 16-byte
    // items) in ONE batch: round 3 staged it in three batches of four, three dependent memory round trips before the first MFMA
    // of a workgroup that then computes for 2 us (only two workgroups per CU overlap each other)
    constexpr int UW = 4, UH1 = 11, UH2 = 6;                             // UH2: the two-input form (twice the registers per item)
    const int w_total = ksteps * NT * 64;                                // 16-byte items of the weight stream
    const int segs = cin >> 3;                                           // 16-byte segments per pixel
    const int h_items = kC3HH * kC3HW * segs;
    auto w_load = [&](int i0, f16x8 (&wv)[UW]) {
#pragma unroll
        for (int u = 0; u < UW; ++u) wv[u] = gsrc[min(i0 + u * 256 + tid, w_total - 1)];
    };
    auto w_store = [&](int i0, const f16x8 (&wv)[UW]) {
#pragma unroll
        for (int u = 0; u < UW; ++u)
            if (i0 + u * 256 + tid < w_total) ring[i0 + u * 256 + tid] = wv[u];
    };
    // halo: (pixel, 16-byte segment) = work item; zero / replicate padding and the pre-activation ReLU are applied while it is
    // written to LDS, once per element
    auto h_load = [&](auto a2_tag, int i0, int b, int ty0, int tx0, auto &v, auto &v2, unsigned &inb) {
        constexpr bool A2 = decltype(a2_tag)::value;
        constexpr int UH = std::extent_v<std::remove_reference_t<decltype(v)>>;
        inb = 0u;
#pragma unroll
        for (int u = 0; u < UH; ++u) {
            const int i = min(i0 + u * 256 + tid, h_items - 1);
            const int p = i / segs, sg = i - p * segs;
            const int hy = p / kC3HW, hx = p - hy * kC3HW;
            const int yy = ty0 + hy - pad, xx = tx0 + hx - pad;         // pad 0 (VALID): rows beyond the map only feed masked outputs
            const int yc = min(max(yy, 0), g.Hi - 1), xc = min(max(xx, 0), g.Wi - 1);
            inb |= ((!g.zpad || (yy == yc && xx == xc)) ? 1u : 0u) << u;
            v[u] = *reinterpret_cast<const f16x8 *>(a_in + (((long)b * g.Hi + yc) * g.Wi + xc) * lda + sg * 8);
            if constexpr (A2)       // second input (cropped U-Net skip, VALID convs only): added while staging
                v2[u] = *reinterpret_cast<const f16x8 *>(g.a2 + (((long)b * g.H2 + yc + g.crop2) * g.W2 + xc + g.crop2) * lda + sg * 8);
        }
    };
    auto h_store = [&](auto a2_tag, int i0, const auto &v, const auto &v2, unsigned inb) {
        constexpr bool A2 = decltype(a2_tag)::value;
        constexpr int UH = std::extent_v<std::remove_reference_t<decltype(v)>>;
        const f16x8 z8 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
#pragma unroll
        for (int u = 0; u < UH; ++u) {
            const int i = i0 + u * 256 + tid;
            if (i >= h_items) continue;
            const int p = i / segs, sg = i - p * segs;
            f16x8 w = ((inb >> u) & 1u) ? v[u] : z8;
            if constexpr (A2) w += v2[u];
            if (g.relu_in) {
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] = w[j] > (f16)0.f ? w[j] : (f16)0.f;
            }
            *reinterpret_cast<f16x8 *>(halo + (long)p * pstride + sg * 16) = w;
        }
    };
    // One patch per workgroup.  (A persistent loop over patches — resident weights copied once — was not faster on large grids
    // and its loop-carried state cost 70 VGPRs, i.e. two resident waves per SIMD: 155 vs 84 registers for NT = 4.)
    const int tile = blockIdx.x;
    const int tx0 = (tile % tiles_x) * kC3TW;
    const int ty0 = ((tile / tiles_x) % tiles_y) * kC3TH;
    const int b = tile / (tiles_x * tiles_y);
    // ring staging: chunk c travels HBM / L2 -> stage[c % 3] (requested three chunks ahead) -> LDS slot c & 1.  The stage index is
    // STATIC (the chunk loop is unrolled by three): rounds 1-3 rotated three named registers with copies (st = sq; sq = sr; ...),
    // and a copy of a register whose load is still in flight has to wait for it — hipcc put `s_waitcnt vmcnt(0)` right behind the
    // two loads it had just issued, so every chunk boundary (9 per 64 -> 64 patch) exposed a full L2 round trip
    // (tools/ring_trace.py on the 64 -> 64 layer: 12 us per workgroup for 2.1 us of MFMAs; profiles/r04_isa_notes.md item 3).
    f16x8 stage[3][2];
    // the second-input form is its own instantiation of the staging code: its eleven extra registers per load batch would
    // otherwise be reserved in every launch (the staging arrays are the kernel's register high-water mark)
    auto stage_in = [&](auto a2_tag) {
        constexpr int UH = decltype(a2_tag)::value ? UH2 : UH1;
        f16x8 hv[UH], hv2[decltype(a2_tag)::value ? UH : 1];
        unsigned hin;
        if (RESW) {
            // the first batches of BOTH streams are requested before anything is waited for: one round trip
            f16x8 wv[UW];
            w_load(0, wv);
            h_load(a2_tag, 0, b, ty0, tx0, hv, hv2, hin);
            w_store(0, wv);
            h_store(a2_tag, 0, hv, hv2, hin);
            for (int i0 = 256 * UW; i0 < w_total; i0 += 256 * UW) { w_load(i0, wv); w_store(i0, wv); }
        } else {
            // the first three weight chunks travel together with the halo
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int cj = min(j, n_chunks);                         // (the stream is zero-padded by 16 KiB on the host)
                stage[j][0] = gsrc[cj * (CH * 64) + tid];
                stage[j][1] = gsrc[cj * (CH * 64) + tid + 256];
            }
            h_load(a2_tag, 0, b, ty0, tx0, hv, hv2, hin);
            h_store(a2_tag, 0, hv, hv2, hin);
        }
        for (int i0 = 256 * UH; i0 < h_items; i0 += 256 * UH) {
            h_load(a2_tag, i0, b, ty0, tx0, hv, hv2, hin);
            h_store(a2_tag, i0, hv, hv2, hin);
        }
    };
    if (g.a2) stage_in(std::true_type{});
    else stage_in(std::false_type{});
    if constexpr (RESW) __syncthreads();
    // token tile f of wave w: output row ty0 + 2w + (f >> 1), columns tx0 + 16 (f & 1) + r16
    int hoff[MF];                                                        // byte offsets into the halo (tap (0, 0), channel 8 grp)
#pragma unroll
    for (int f = 0; f < MF; ++f) hoff[f] = ((2 * wave + (f >> 1)) * kC3HW + 16 * (f & 1) + r16) * pstride + grp * 16;

    f32x4 acc[NT][MF];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int f = 0; f < MF; ++f) acc[nt][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // one k-step = tap (dy, dx), 32-channel part c: four B fragments from the halo against the NT weight fragments `wsrc[nt * 64]`
    auto kstep = [&](int tap, int c, const f16x8 *wsrc) {
        const int dy = tap / 3, dx = tap - 3 * dy;
        const int toff = (dy * kC3HW + dx) * pstride;
        f16x8 xq[MF];
#pragma unroll
        for (int f = 0; f < MF; ++f) xq[f] = *reinterpret_cast<const f16x8 *>(halo + hoff[f] + toff + c * 64);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f16x8 w = wsrc[nt * 64];
#pragma unroll
            for (int f = 0; f < MF; ++f) acc[nt][f] = MFMA_16x16x32(w, xq[f], acc[nt][f]);
        }
    };
    if constexpr (RESW) {
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll 1
            for (int c = 0; c < cpt; ++c) kstep(tap, c, ring + (tap * cpt + c) * NT * 64 + lane);
        }
    } else {
        constexpr int KPC = CH / NT;                                     // k-steps per 8-KiB chunk
        int tap = 0, cc = 0;                                             // (tap, part) of the next k-step, advanced incrementally
        // The trip count is rounded up to a multiple of three and NOTHING that touches vmcnt sits under a branch (stores to the
        // ring, barrier and refill are unconditional; past the end they move zero padding): with a conditional boundary hipcc
        // no longer knows how many loads are pending at the next one and waits for all of them again.
#pragma unroll 1
        for (int c0 = 0; c0 < n_chunks; c0 += 3) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int c = c0 + j;
                // chunk boundary: stage j (requested three chunks ago) -> LDS slot c & 1, then refill the stage
                ring[(c & 1) * (CH * 64) + tid] = stage[j][0];
                ring[(c & 1) * (CH * 64) + tid + 256] = stage[j][1];
                __syncthreads();                                         // (the first one also publishes the halo)
                const int cn = min(c + 3, n_chunks);                     // <= n_chunks: inside the host's 16-KiB zero padding
                stage[j][0] = gsrc[cn * (CH * 64) + tid];
                stage[j][1] = gsrc[cn * (CH * 64) + tid + 256];
#pragma unroll
                for (int q = 0; q < KPC; ++q) {
                    if (c * KPC + q < ksteps) {
                        kstep(tap, cc, ring + (c & 1) * (CH * 64) + q * NT * 64 + lane);
                        if (++cc == cpt) { cc = 0; ++tap; }
                    }
                }
            }
        }
    }

    if (g.cmaj) {                                                        // fp32 partial [chunk][pixel][N], no epilogue
        float *part = g.part32 + (long)blockIdx.y * ((long)g.B * g.Ho * g.Wo) * g.N;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                const int oy = ty0 + 2 * wave + (f >> 1), ox = tx0 + 16 * (f & 1) + r16;
                if (oy >= g.Ho || ox >= g.Wo) continue;
                *reinterpret_cast<f32x4 *>(part + (((long)b * g.Ho + oy) * g.Wo + ox) * g.N + nt * 16 + grp * 4) = acc[nt][f];
            }
        return;
    }
    const int ldo = g.ldo > 0 ? g.ldo : g.n_real;
    if constexpr (NT == 1) {
        if (g.out32) {
            // Image head (planar fp32, `+ crop(add32)`, clamp; cunet.py:183-196).  All `add32` reads of a lane are requested
            // BEFORE any of them is used — unconditionally, from clamped addresses: as a load under `if` next to its use every
            // one of the 12 was followed by `s_waitcnt vmcnt(0)`, twelve dependent memory round trips per workgroup in a
            // kernel that otherwise moves 130 B per pixel (conv3_lds_kernel<1>: 340 us per launch at 1.1 TB/s).
            const int n0 = grp * 4;
            const float4 bv = *reinterpret_cast<const float4 *>(g.bias + n0);
            const float bvr[4] = {bv.x, bv.y, bv.z, bv.w};
            float addv[MF][4];
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                const int oy = min(ty0 + 2 * wave + (f >> 1), g.Ho - 1), ox = min(tx0 + 16 * (f & 1) + r16, g.Wo - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = min(n0 + r, g.n_real - 1);
                    addv[f][r] = g.add32 ? g.add32[(((long)b * g.n_real + n) * g.addH + oy + g.add_crop) * g.addW + ox + g.add_crop] : 0.f;
                }
            }
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                const int oy = ty0 + 2 * wave + (f >> 1), ox = tx0 + 16 * (f & 1) + r16;
                if (oy >= g.Ho || ox >= g.Wo) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + r;
                    if (n >= g.n_real) continue;
                    float o = acc[0][f][r] + bvr[r];
                    if (g.act == 2) o = o >= 0.f ? o : o * g.slope;
                    else if (g.act == 3) o = fmaxf(o, 0.f);
                    o += addv[f][r];
                    if (g.clamp01) o = fminf(fmaxf(o, 0.f), 1.f);
                    g.out32[(((long)b * g.n_real + n) * g.Ho + oy) * g.Wo + ox] = o;
                }
            }
            return;
        }
    }
    if constexpr (NT % 2 == 0) {
        // NHWC fp16 in 16-byte runs: two adjacent tiles -> 8 consecutive channels per lane (fp32 permlane swap, so that
        // `fp16(conv + res)` rounds exactly as in the 8-byte form below), residual values requested two fragments at a time
        // before the swaps that use them.  As in conv3_dma_kernel (profiles/r04_c3d_trace.txt).
        if (!g.out32 && g.n_real % 32 == 0 && ldo % 8 == 0) {
#pragma unroll
            for (int np = 0; np < NT / 2; ++np) {
                const int nb = 2 * np * 16;
                const float4 b0 = *reinterpret_cast<const float4 *>(g.bias + nb + grp * 4);
                const float4 b1 = *reinterpret_cast<const float4 *>(g.bias + nb + 16 + grp * 4);
                const float bb[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
                for (int f0 = 0; f0 < MF; f0 += 2) {
                    long off[2];
                    bool live[2];
                    f16x8 rv1[2], rv2[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int f = f0 + u;
                        const int oy = ty0 + 2 * wave + (f >> 1), ox = tx0 + 16 * (f & 1) + r16;
                        live[u] = oy < g.Ho && ox < g.Wo && nb < g.n_real;
                        off[u] = (((long)b * g.Ho + min(oy, g.Ho - 1)) * g.Wo + min(ox, g.Wo - 1)) * ldo + nb + pair_run_channel(grp);
                    }
                    if (g.res) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) rv1[u] = *reinterpret_cast<const f16x8 *>(g.res + off[u]);
                    }
                    if (g.res2) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) rv2[u] = *reinterpret_cast<const f16x8 *>(g.res2 + off[u]);
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int f = f0 + u;
                        float v[2][4];
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float x = acc[2 * np + t][f][r] + bb[t][r];
                                if (g.act == 2) x = x >= 0.f ? x : x * g.slope;
                                else if (g.act == 3) x = fmaxf(x, 0.f);
                                v[t][r] = x;
                            }
                        float run[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const u32x2 sw = lane16_swap(__builtin_bit_cast(unsigned, v[0][r]), __builtin_bit_cast(unsigned, v[1][r]));
                            const unsigned lo = sw[0], hi = sw[1];
                            run[r] = __builtin_bit_cast(float, lo);
                            run[4 + r] = __builtin_bit_cast(float, hi);
                        }
                        if (g.res) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) run[e] += (float)rv1[u][e];
                        }
                        if (g.res2) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) run[e] += (float)rv2[u][e];
                        }
                        f16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (f16)run[e];
                        if (live[u]) *reinterpret_cast<f16x8 *>(g.out + off[u]) = o;      // (after the swap: it needs every lane)
                    }
                }
            }
            return;
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n0 = nt * 16 + grp * 4;
        const float4 bv = *reinterpret_cast<const float4 *>(g.bias + n0);
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            const int oy = ty0 + 2 * wave + (f >> 1), ox = tx0 + 16 * (f & 1) + r16;
            if (oy >= g.Ho || ox >= g.Wo || n0 >= g.n_real) continue;
            float v[4] = {acc[nt][f][0] + bv.x, acc[nt][f][1] + bv.y, acc[nt][f][2] + bv.z, acc[nt][f][3] + bv.w};
            if (g.act == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] >= 0.f ? v[r] : v[r] * g.slope;
            } else if (g.act == 3) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (g.out32) {
                // image head: planar fp32, optional `+ crop(add32)` and clamp  (cunet.py:183-196), as in conv_kernel
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + r;
                    if (n >= g.n_real) continue;
                    float o = v[r];
                    if (g.add32) o += g.add32[(((long)b * g.n_real + n) * g.addH + oy + g.add_crop) * g.addW + ox + g.add_crop];
                    if (g.clamp01) o = fminf(fmaxf(o, 0.f), 1.f);
                    g.out32[(((long)b * g.n_real + n) * g.Ho + oy) * g.Wo + ox] = o;
                }
                continue;
            }
            const long off = (((long)b * g.Ho + oy) * g.Wo + ox) * ldo + n0;
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
            *reinterpret_cast<f16x4 *>(g.out + off) = (f16x4){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
        }
    }
}

// Wide inputs (Cin > 128: the layer{3,4}_rn convs of the DPT head read 192 ... 1024 channels on 14 x 25 / 28 x 49 maps, 6-22
// patches).  One workgroup per patch walks 54-108 k-steps with at most 24 KiB of weight requests in flight — ~10 GB/s against a
// 2.5-us round trip, 51-113 us per launch — and staging the input in LDS does not change that (measured: 90 us).  The
// contraction is therefore SPLIT ACROSS WORKGROUPS: the owner packs the stream chunk-major (ConvArgs.cmaj = 64:
// [64-channel chunk][tap][32-channel half][n-tile], i.e. chunk q is a complete tap-major stream of a 64-channel conv), grid.y =
// chunks, every workgroup runs conv3_lds_kernel<NT, true> on its channel slice (72 KiB of weights + the halo, everything
// requested at once) and writes an fp32 partial; conv_partial_sum_kernel adds the partials in chunk order (deterministic),
// then bias, activation, residuals as usual.
__global__ void __launch_bounds__(256) conv_partial_sum_kernel(ConvArgs g, const float *__restrict__ part, int n_q) {
    const long M = (long)g.B * g.Ho * g.Wo;
    const int nq4 = g.N / 4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * nq4) return;
    const long p = i / nq4;
    const int n0 = (int)(i - p * nq4) * 4;
    if (n0 >= g.n_real) return;
    f32x4 v = *reinterpret_cast<const f32x4 *>(part + p * g.N + n0);
    for (int q = 1; q < n_q; ++q) {
        const f32x4 w = *reinterpret_cast<const f32x4 *>(part + ((long)q * M + p) * g.N + n0);
        v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
    }
    const float4 bv = *reinterpret_cast<const float4 *>(g.bias + n0);
    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
    if (g.act == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] >= 0.f ? v[r] : v[r] * g.slope;
    } else if (g.act == 3) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    }
    const long off = p * (g.ldo > 0 ? g.ldo : g.n_real) + n0;
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
    *reinterpret_cast<f16x4 *>(g.out + off) = (f16x4){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
}

static inline int conv3_lds_enabled() { const char *e = getenv("NUNIF_CONV3_LDS"); return e ? atoi(e) : 1; }

bool conv3_lds_applies(const ConvArgs &g) {
    const int nt = g.N / 16;
    const int pad = (g.zpad || g.rpad) ? 1 : 0;
    const bool shape = g.kh == 3 && g.kw == 3 && g.stride == 1 && (!g.a2 || (!pad && !g.cmaj)) && (!g.out32 || !g.cmaj) && g.Ho == g.Hi + 2 * pad - 2 &&
                       g.Wo == g.Wi + 2 * pad - 2 && g.zpad <= 1 && g.rpad <= 1 && !(g.zpad && g.rpad) && g.N % 16 == 0 &&
                       (nt == 1 || nt == 2 || nt == 4 || nt == 8);
    if (!conv3_lds_enabled() || !shape) return false;
    if (g.cmaj) return nt != 1 && (g.cmaj == 64 || g.cmaj == 32) && g.Cin % g.cmaj == 0;     // a chunk-major stream: the split form only
    return g.Cin % 32 == 0 && g.Cin <= 128;
}

template <int NT, bool RESW>
static int launch_c3r(const ConvArgs &g, hipStream_t s, size_t smem) {
    static bool configured = false;
    if (!configured) {
        NUNIF_HIP_CHECK(hipFuncSetAttribute((const void *)conv3_lds_kernel<NT, RESW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            150 * 1024));
        configured = true;
    }
    long blocks = (long)g.B * ((g.Ho + kC3TH - 1) / kC3TH) * ((g.Wo + kC3TW - 1) / kC3TW);
    conv3_lds_kernel<NT, RESW><<<(unsigned)blocks, 256, smem, s>>>(g);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

template <int NT>
static int launch_c3(const ConvArgs &g, hipStream_t s, const char *name) {
    const size_t halo = (size_t)kC3HH * kC3HW * (g.Cin * 2 + 16);
    const size_t wres = (size_t)9 * (g.Cin >> 5) * NT * 1024;
    const long M = (long)g.B * g.Ho * g.Wo;
    ProfScope ps(name, s, 2.0 * (double)M * 9.0 * g.Cin * g.n_real, (double)M * (g.Cin + g.n_real) * 2.0);
    // resident weights for launches of at most one workgroup per CU (the small maps, where the ring's chunk boundaries are a chain
    // of memory round trips: 23 -> 17 us); on larger grids the 72-KiB copy per workgroup and one resident workgroup per CU cost
    // more than they save (392 patches: 34 vs 26 us), and a persistent form with 18 + 12 loads in flight was no better either
    const long n_tiles = (long)g.B * ((g.Ho + kC3TH - 1) / kC3TH) * ((g.Wo + kC3TW - 1) / kC3TW);
    if (n_tiles <= 256 && halo + wres <= 150 * 1024) return launch_c3r<NT, true>(g, s, halo + wres);
    return launch_c3r<NT, false>(g, s, halo + 2 * 8 * 1024);
}

template <int NT>
static int launch_c3cm(const ConvArgs &g, hipStream_t s) {
    const int n_q = g.Cin / g.cmaj;
    const size_t smem = (size_t)9 * (g.cmaj >> 5) * NT * 1024 + (size_t)kC3HH * kC3HW * (g.cmaj * 2 + 16);
    NUNIF_REQUIRE(g.part32 && smem <= 150 * 1024, "conv3_lds: split contraction needs ConvArgs.part32 (B*Ho*Wo*N*Cin/cmaj floats)");
    static bool configured = false;
    if (!configured) {
        NUNIF_HIP_CHECK(hipFuncSetAttribute((const void *)conv3_lds_kernel<NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            150 * 1024));
        configured = true;
    }
    const long M = (long)g.B * g.Ho * g.Wo;
    ProfScope ps("conv3_lds_kernel<split-K>", s, 2.0 * (double)M * 9.0 * g.Cin * g.n_real, (double)M * (g.Cin + g.n_real) * 2.0);
    const long tiles = (long)g.B * ((g.Ho + kC3TH - 1) / kC3TH) * ((g.Wo + kC3TW - 1) / kC3TW);
    conv3_lds_kernel<NT, true><<<dim3((unsigned)tiles, (unsigned)n_q), 256, smem, s>>>(g);
    conv_partial_sum_kernel<<<(unsigned)((M * (g.N / 4) + 255) / 256), 256, 0, s>>>(g, g.part32, n_q);
    NUNIF_LAUNCH_CHECK();
    return NUNIF_HIP_OK;
}

int launch_conv3_lds(const ConvArgs &g, hipStream_t s) {
    if (g.cmaj) {
        switch (g.N / 16) {
            case 2: return launch_c3cm<2>(g, s);
            case 4: return launch_c3cm<4>(g, s);
            case 8: return launch_c3cm<8>(g, s);
            default: set_error("conv3_lds: Cout=%d unsupported", g.N); return NUNIF_HIP_EUNSUPPORTED;
        }
    }
    switch (g.N / 16) {
        case 1: return launch_c3<1>(g, s, "conv3_lds_kernel<1>");
        case 2: return launch_c3<2>(g, s, "conv3_lds_kernel<2>");
        case 4: return launch_c3<4>(g, s, "conv3_lds_kernel<4>");
        case 8: return launch_c3<8>(g, s, "conv3_lds_kernel<8>");
        default: set_error("conv3_lds: Cout=%d unsupported", g.N); return NUNIF_HIP_EUNSUPPORTED;
    }
}

}  // namespace nunif
