// dl4ds_amd -- 3x3 convolutions with few channels (Cin <= 16, Cout <= 16) on the f32 matrix cores.
//
// ConvBlock_att / the first residual blocks (blocks.py:87-103, 210-230 with n_filters = 8) are 8->8 and 16->16 3x3
// layers; on the HR grid of cfg2 (64 x 512 x 512 pixels) one of them moves 1.07 GB and needs 18 16x16x4 MFMAs per 16
// pixels, i.e. HBM time and matrix-core time are both ~0.25 ms.  The general implicit-GEMM kernel reached 0.67 ms
// there: its K loop is generic (runtime channel chunks, filter slices re-staged through LDS for every tile, one
// ds_read_b32 per MFMA with computed addresses) and for K = 72 that bookkeeping outweighs the MFMAs.  This variant
// fixes everything at compile time:
//   * the whole filter lives in registers for the lifetime of a persistent block (18 or 36 VGPRs);
//   * the K axis is ordered (tap, cin) with a lane's MFMA k-slot q owning the cin group [E*q, E*q+E), E = CI/4, so
//     ONE ds_read_b64/b128 at an immediate offset feeds E consecutive MFMAs, and 16 pixels x CI floats are read as
//     one contiguous conflict-free LDS segment;
//   * an input row segment read once is used by the three output rows it contributes to (30 LDS reads for 144 MFMAs
//     per wave and tile at CI = 8);
//   * 16 x 32-pixel tiles, 4 waves x 8 rows, persistent over tiles, several blocks per CU so that staging and the
//     epilogue of one block overlap with another block's MFMAs.
// Forward and dgrad (flipped/transposed filter) both run through it; the epilogue is the shared conv_epilogue.
#include "ops.h"
#include "prof.h"
#include "conv_kernels.h"
#include "launch.h"
#include <algorithm>
#include <vector>
#include <cstdlib>

namespace {

constexpr int NTW = 16, NTH = 32, NROWS = 8;      // tile width / height, rows per wave
#ifndef NARROW_PAIR_ROWS
#define NARROW_PAIR_ROWS 4
#endif

#ifndef NARROW16_LB
#define NARROW16_LB 2
#endif
template <int CI>
__global__ void __launch_bounds__(256, CI == 8 ? 4 : NARROW16_LB) conv_narrow_kernel(const ConvParams a) {
    constexpr int E = CI / 4;                      // cin values per lane and tap
    constexpr int TWH = NTW + 2, THH = NTH + 2, HPIX = TWH * THH;
    constexpr int Q4 = CI / 4;                     // float4s per pixel
    constexpr int TOTAL = HPIX * Q4;
    constexpr int ITERS = (TOTAL + 255) / 256;
    __shared__ __attribute__((aligned(16))) float tile[HPIX * CI];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;

    // filter fragments: MFMA first operand = W^T[cout = l15][k-slot lq]; k-step (tap, e) uses cin = E*lq + e
    float wr[9][E];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int ci = E * lq + e;
            const bool ok = ci < a.Cin && l15 < a.Cout;
            const float v = a.w[((size_t)t * a.Cin + (ok ? ci : 0)) * a.Cout + (ok ? l15 : 0)];
            wr[t][e] = ok ? v : 0.f;
        }

    const float* rd = tile + ((wave * NROWS) * TWH + l15) * CI + E * lq;
    const size_t in_total = (size_t)(a.in.N - 1) * a.in.nstride + (size_t)a.H * a.W * a.in.ld;      // floats in the input view
    const bool ragged = (a.Cin & 3) != 0;           // channel quads that reach into the next pixel (and, at the very end, beyond the view)

    for (int t = blockIdx.x; t < a.tiles_x * a.tiles_y * a.in.N; t += gridDim.x) {
        const int q = fast_div(t, a.m_txy[0]);
        const int bx = t - q * a.tiles_x;
        const int n = fast_div(q, a.m_txy[1]);
        const int by = q - n * a.tiles_y;
        const int x0 = bx * NTW, y0 = by * NTH;

        // ---- stage the halo tile (all loads issued first, masked when written to LDS)
        {
            float4 r[ITERS];
            unsigned m[ITERS];
#pragma unroll
            for (int u = 0; u < ITERS; ++u) {
                const int e = tid + u * 256;
                const int pix = e / Q4, c4 = e - pix * Q4;
                const int hy = pix / TWH, hx = pix - hy * TWH;
                const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                const bool ok = e < TOTAL && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                // plain 16-byte-aligned view (checked by the dispatcher): always-issued load from a clamped address
                const bool cok = ok && c4 * 4 < a.Cin;
                const size_t off = (size_t)n * a.in.nstride + ((size_t)(ok ? gy : 0) * a.W + (ok ? gx : 0)) * a.in.ld +
                                   (cok ? c4 * 4 : 0);
                if (!ragged || off + 4 <= in_total) {
                    r[u] = *reinterpret_cast<const float4*>(a.in.p + off);      // (dword-aligned when the channel count is not a multiple of 4)
                } else {
                    // the last quad of the tensor's last pixel when Cin % 4 != 0: nothing may be read beyond the buffer
                    const float* q = a.in.p + off;
                    r[u] = make_float4(q[0], off + 1 < in_total ? q[1] : 0.f, off + 2 < in_total ? q[2] : 0.f, 0.f);
                }
                m[u] = valid4(c4 * 4, a.Cin, ok);
            }
#pragma unroll
            for (int u = 0; u < ITERS; ++u) {
                const int e = tid + u * 256;
                if (e < TOTAL) *reinterpret_cast<float4*>(tile + (size_t)e * 4) = mask4(r[u], m[u]);
            }
        }
        __syncthreads();

        f32x4 acc[NROWS];
#pragma unroll
        for (int i = 0; i < NROWS; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rho = 0; rho < NROWS + 2; ++rho) {           // input row of the wave's 10-row band
            if (rho % 2 == 0) __builtin_amdgcn_sched_barrier(0);   // bound how far the LDS reads are hoisted (VGPRs)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                float v[E];
                const float* src = rd + (rho * TWH + dx) * CI;
                if (E == 2) {
                    const float2 t2 = *reinterpret_cast<const float2*>(src);
                    v[0] = t2.x; v[1 % E] = t2.y;
                } else {
                    const float4 t4 = *reinterpret_cast<const float4*>(src);
                    v[0] = t4.x; v[1 % E] = t4.y; v[2 % E] = t4.z; v[3 % E] = t4.w;
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int r = rho - dy;               // output row fed through tap (dy, dx)
                        if (r >= 0 && r < NROWS)
                            acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[dy * 3 + dx][e], v[e], acc[r], 0, 0, 0);
                    }
                }
            }
        }
        {
            AccPack<NROWS, 1> accp;
#pragma unroll
            for (int i = 0; i < NROWS; ++i) accp.v[i][0] = acc[i];
            conv_epilogue<NROWS, 1>(a, accp, n, x0, y0, 0, wave, 0, l15, lq);
        }
        __syncthreads();
    }
}

template <int CI>
void launch_narrow(hipStream_t s, ConvParams& p, int N) {
    p.tiles_x = cdiv(p.W, NTW);
    p.tiles_y = cdiv(p.H, NTH);
    p.m_txy[0] = div_magic(p.tiles_x);
    p.m_txy[1] = div_magic(p.tiles_y);
    const int ntiles = p.tiles_x * p.tiles_y * N;
    if (ntiles == 0) return;
    // persistent blocks: exactly one residency round (equal work per block, so a partial second round would
    // cost a full extra round)
    const int blocks = std::min(ntiles, resident_blocks<conv_narrow_kernel<CI>>(256));
    const double px = (double)N * p.H * p.W;
    ProfScope ps(s, "conv_narrow<" + std::to_string(CI) + ">", 2.0 * px * 9 * p.Cin * p.Cout,
                 4.0 * (px * (p.Cin + p.Cout * (1 + (p.add.p ? 1 : 0) + (p.mask.p ? 1 : 0) + (p.accumulate ? 1 : 0))) + 9.0 * p.Cin * p.Cout));
    DL4DS_LAUNCH((conv_narrow_kernel<CI>), dim3(blocks), dim3(256), 0, s, p);
    HIP_CHECK(hipGetLastError());
}

// --------------------------------------------------------------------------------------------
// Cin <= 8, Cout <= 8: two output pixels per MFMA column.  With 8 couts only half of the 16 MFMA rows carry data; here
// rows are (h, cout): h = 0 computes pixel (y, 2j), h = 1 pixel (y, 2j+1) of column j's pixel PAIR.  Both share the
// second operand x[(y+dy), 2j+ux][cin] with ux = 0..3 (the union of the two 3-wide windows); the filter operand of row
// (h, co) at (dy, ux) is W[dy][ux-h][cin][co] (zero outside the 3 taps).  K grows from 72 to 96 but a tile of 16 columns
// now covers 32 pixels: 24 MFMAs per 32 pixels instead of 36, and after the MFMA every lane holds four couts of one
// pixel, so all 64 lanes store (the 16-pixel variant idles half of them).
template <int NR>
#ifndef NARROW_PAIR_LB
#define NARROW_PAIR_LB 3
#endif
__global__ void __launch_bounds__(256, NARROW_PAIR_LB) conv_narrow_pair_kernel(const ConvParams a) {
    constexpr int PTW = 32, PTH = 4 * NR;            // tile: 32 x (4 waves x NR rows)
    constexpr int TWH = PTW + 2, THH = PTH + 2, HPIX = TWH * THH;
    constexpr int P = 10;                            // LDS pixel pitch (floats): lanes 2 pixels apart -> all 32 banks
    constexpr int TOTAL = HPIX * 2, ITERS = (TOTAL + 255) / 256;
    __shared__ __attribute__((aligned(16))) float tile[HPIX * P];
    __shared__ __attribute__((aligned(16))) float pool_red[2][32];     // a.pool: [parity][wave][channel]
    int it = 0;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;

    // filter fragments: row l15 = (h, co), k-slot lq owns cin 2*lq, 2*lq+1
    float wr[3][4][2];
    {
        const int h = l15 >> 3, co = l15 & 7;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int ux = 0; ux < 4; ++ux)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int kx = ux - h, ci = 2 * lq + e;
                    const bool ok = kx >= 0 && kx <= 2 && ci < a.Cin && co < a.Cout;
                    const float v = a.w[((size_t)(dy * 3 + (ok ? kx : 0)) * a.Cin + (ok ? ci : 0)) * a.Cout + (ok ? co : 0)];
                    wr[dy][ux][e] = ok ? v : 0.f;
                }
    }
    const float* rd = tile + ((wave * NR) * TWH + 2 * l15) * P + 2 * lq;
    const bool vec_in = a.in.vec != 0;               // uniform

    // epilogue constants: lane (pair column l15, k-slot lq) holds rows 4*lq + r = (h = lq >> 1, couts 4*(lq & 1) + r)
    const int eh = lq >> 1, ec = 4 * (lq & 1);
    const bool c_ok = ec < a.Cout;
    const size_t q_out = view_chan_off(a.out, c_ok ? ec : 0);
    const size_t q_add = a.add.p ? view_chan_off(a.add, c_ok ? ec : 0) : 0;
    const size_t q_mask = a.mask.p ? view_chan_off(a.mask, c_ok ? ec : 0) : 0;
    const float4 bias_v = (a.bias && c_ok) ? *reinterpret_cast<const float4*>(a.bias + ec) : make_float4(0.f, 0.f, 0.f, 0.f);

    // The halo tile of the NEXT tile is requested while this one is computed and stored: the kernel is HBM-bound, and with
    // load -> LDS -> MFMA -> store phases per tile only a third of the resident workgroups had loads in flight at any time
    // (3.0-3.4 TB/s).  The five float4 staging registers and the packed validity bits stay live across the MFMA phase.
    const int ntiles_all = a.tiles_x * a.tiles_y * a.in.N;
    float4 r[ITERS];
    unsigned mbits = 0;
    float4 s4 = make_float4(1.f, 1.f, 1.f, 1.f), h4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto issue = [&](int t) __attribute__((always_inline)) {
        const int q = fast_div(t, a.m_txy[0]);
        const int bx = t - q * a.tiles_x;
        const int n = fast_div(q, a.m_txy[1]);
        const int by = q - n * a.tiles_y;
        const int x0 = bx * PTW, y0 = by * PTH;
        mbits = 0;
        // the vec / scalar choice is made OUTSIDE the unrolled load loops: a (uniform) branch per load keeps the
        // loads in separate basic blocks and cost 13 % on the aligned case
        if (vec_in) {
#pragma unroll
            for (int u = 0; u < ITERS; ++u) {
                const int e = tid + u * 256;
                const int pix = e >> 1, c4 = e & 1;
                const int hy = pix / TWH, hx = pix - hy * TWH;
                const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                const bool ok = e < TOTAL && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                const bool cok = ok && c4 * 4 < a.Cin;
                const size_t off = (size_t)n * a.in.nstride + ((size_t)(ok ? gy : 0) * a.W + (ok ? gx : 0)) * a.in.ld +
                                   (cok ? c4 * 4 : 0);
                r[u] = *reinterpret_cast<const float4*>(a.in.p + off);
                mbits |= valid4(c4 * 4, a.Cin, ok) << (4 * u);
            }
            // channel affine of the view (ChannelAttention2D's scale / its backward, see TView): the thread's channel quad
            // is fixed (e & 1 == tid & 1), the image is fixed for the tile; applied when the tile is written to LDS
            if (a.in.sc) view_affine4(a.in, n, (tid & 1) * 4, s4, h4);
        } else {
            // Cin not a multiple of 4 (e.g. the 5 + 1 input channels of the U-Net): four clamped scalar loads
#pragma unroll
            for (int u = 0; u < ITERS; ++u) {
                const int e = tid + u * 256;
                const int pix = e >> 1, c4 = e & 1;
                const int hy = pix / TWH, hx = pix - hy * TWH;
                const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                const bool ok = e < TOTAL && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                const float* px = a.in.p + (size_t)n * a.in.nstride + ((size_t)(ok ? gy : 0) * a.W + (ok ? gx : 0)) * a.in.ld;
                const int cm = a.Cin - 1;
                r[u] = make_float4(px[min(c4 * 4, cm)], px[min(c4 * 4 + 1, cm)], px[min(c4 * 4 + 2, cm)], px[min(c4 * 4 + 3, cm)]);
                mbits |= valid4(c4 * 4, a.Cin, ok) << (4 * u);
            }
        }
    };
#ifndef NARROW_PAIR_NOPF
    if ((int)blockIdx.x < ntiles_all) issue(blockIdx.x);
#endif
    for (int t = blockIdx.x; t < ntiles_all; t += gridDim.x) {
#ifdef NARROW_PAIR_NOPF
        issue(t);
#endif
        const int q = fast_div(t, a.m_txy[0]);
        const int bx = t - q * a.tiles_x;
        const int n = fast_div(q, a.m_txy[1]);
        const int by = q - n * a.tiles_y;
        const int x0 = bx * PTW, y0 = by * PTH;
        {
            if (vec_in && a.in.sc) {
#pragma unroll
                for (int u = 0; u < ITERS; ++u) r[u] = affine4(r[u], s4, h4);
            }
#pragma unroll
            for (int u = 0; u < ITERS; ++u) {
                const int e = tid + u * 256;
                if (e < TOTAL) {
                    const float4 v = mask4(r[u], (mbits >> (4 * u)) & 15u);
                    float2* d = reinterpret_cast<float2*>(tile + (size_t)(e >> 1) * P + (e & 1) * 4);
                    d[0] = make_float2(v.x, v.y);
                    d[1] = make_float2(v.z, v.w);
                }
            }
        }
        __syncthreads();
#ifndef NARROW_PAIR_NOPF
        if (t + (int)gridDim.x < ntiles_all) issue(t + gridDim.x);
#endif

        f32x4 acc[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rho = 0; rho < NR + 2; ++rho) {
            if (rho % 2 == 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ux = 0; ux < 4; ++ux) {
                const float2 v = *reinterpret_cast<const float2*>(rd + (rho * TWH + ux) * P);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int r = rho - dy;
                    if (r >= 0 && r < NR) {
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[dy][ux][0], v.x, acc[r], 0, 0, 0);
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[dy][ux][1], v.y, acc[r], 0, 0, 0);
                    }
                }
            }
        }
        const int gx = x0 + 2 * l15 + eh;
        float4 psum = make_float4(0.f, 0.f, 0.f, 0.f);      // a.pool: this lane's share of the tile's channel sums
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int gy = y0 + wave * NR + i;
            if (c_ok && gy < a.H && gx < a.W) {
                float4 v = make_float4(acc[i][0] + bias_v.x, acc[i][1] + bias_v.y, acc[i][2] + bias_v.z, acc[i][3] + bias_v.w);
                if (a.add.p) {
                    const float4 ad = *reinterpret_cast<const float4*>(a.add.p + view_pix_base(a.add, n, gy, gx) + q_add);
                    v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
                }
                if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (a.mask.p) {
                    const float4 mk = *reinterpret_cast<const float4*>(a.mask.p + view_pix_base(a.mask, n, gy, gx) + q_mask);
                    v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
                    v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
                }
                float4* dst = reinterpret_cast<float4*>(a.out.p + view_pix_base(a.out, n, gy, gx) + q_out);
                if (a.accumulate) {
                    const float4 old = *dst;
                    v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
                }
                *dst = v;
                psum.x += v.x; psum.y += v.y; psum.z += v.z; psum.w += v.w;
            }
        }
        if (a.pool) {
            // GlobalAveragePooling of ChannelAttention2D (blocks.py:585-588) without a pass over the tensor just written:
            // per-tile channel sums in a fixed order -- 16 pair columns (xor shuffles), the two pixels of a pair (lanes 32
            // apart), the four waves through LDS -- one 8-float record per tile, summed per image by colsum_finish_kernel
#pragma unroll
            for (int mk = 1; mk < 16; mk <<= 1) {
                psum.x += __shfl_xor(psum.x, mk, 64); psum.y += __shfl_xor(psum.y, mk, 64);
                psum.z += __shfl_xor(psum.z, mk, 64); psum.w += __shfl_xor(psum.w, mk, 64);
            }
            psum.x += __shfl_xor(psum.x, 32, 64); psum.y += __shfl_xor(psum.y, 32, 64);
            psum.z += __shfl_xor(psum.z, 32, 64); psum.w += __shfl_xor(psum.w, 32, 64);
            float* red = pool_red[it & 1];
            if (lane == 0 || lane == 16) *reinterpret_cast<float4*>(red + wave * 8 + (lane >> 4) * 4) = psum;
        }
        __syncthreads();
        if (a.pool && tid < 8) {
            const float* red = pool_red[it & 1];
            a.pool[(size_t)t * 8 + tid] = (red[tid] + red[8 + tid]) + (red[16 + tid] + red[24 + tid]);
        }
        ++it;
    }
}

// Producer / consumer form of the pair kernel (float4-loadable plain inputs, no pooling partials): 8-wave workgroups, two per
// CU.  Waves 4-7 do nothing but move halo tiles: buffer loads three tiles ahead (three register sets; out-of-range offsets
// return the zero padding), LDS writes one tile ahead into the other of two buffers.  Waves 0-3 run the 96 MFMAs of a tile
// (accumulators start at the bias through the C operand of each row's first MFMA) and store their four float4 per lane
// through precomputed offsets.  One barrier per tile.  With load -> LDS -> MFMA -> store phases inside every wave the
// kernel kept neither the matrix pipe (55 % busy) nor HBM (3.2 TB/s) occupied.
#ifdef PAIR_WS_TRACE
__device__ unsigned long long pair_ws_trace[4096 * 8];      // diagnostics build: per-workgroup phase sums (tools/variant_build.sh)
#define PT_TIC() do { pt_t = wall_clock64(); } while (0)
#define PT_TOC(slot_) do { const unsigned long long n_ = wall_clock64(); pt_acc[slot_] += n_ - pt_t; pt_t = n_; } while (0)
#else
#define PT_TIC()
#define PT_TOC(slot_)
#endif
// EPI: the epilogue form as a compile-time bit set (PAIR_EPI_*), or -1 = decided at run time from the parameters.  Every
// run-time switch costs the MFMA waves vector instructions (selects between the variants), and a vector instruction of one
// wave is starved to ONE issue per MFMA of the other workgroup's wave on the same SIMD (profiles/mfma_ubench_r02.txt): the
// generic epilogue's ~45 vector instructions per row were as long as the K loop itself (PAIR_WS_TRACE: K loop 150 us,
// epilogue 133 us per workgroup at 64 x 512^2).
enum : int { PAIR_EPI_ADD = 1, PAIR_EPI_RELU = 2, PAIR_EPI_MASK = 4, PAIR_EPI_ACC = 8, PAIR_EPI_POOL = 16, PAIR_EPI_AFF = 32 };
// C16 (round 5): the same kernel for 9 .. 16 INPUT channels and <= 8 outputs (13 -> 8: ConvBlock_att's first layer in the recurrent nets;
// 16 -> 8: the U-Net decoder's first 512^2 layer) -- conv_narrow16_ws left rows 8 .. 15 of every MFMA idle there (36 MFMAs per 16
// pixels).  k-slot lq carries channels 4 lq .. 4 lq + 3 (one ds_read_b128 per tap column), four MFMAs per tap position: 48 per 32 pixels.
// Pixels are staged 20 floats apart: the pair columns are then 10 sixteen-byte slots apart = 2 (mod 4), the conflict-free pitch.
// SPLIT (experiments builds only): the K loop on the 16-bit matrix pipe.  The loaders split every staged value
// into bf16 parts (hi = bf16(x), lo = bf16(x - hi), lo2 = bf16(x - hi - lo): three 16-byte planes of eight channels per pixel), the
// filter is split the same way in registers, and ONE v_mfma_f32_16x16x32_bf16 covers the four tap columns x eight channels of a halo row
// for one term: SPLIT = 3 issues (hi,lo) (lo,hi) (hi,hi), SPLIT = 6 adds (lo,lo) (hi,lo2) (lo2,hi) -- fp32 accumulation, the C layout of the
// fp32 MFMA, so everything outside the K loop is unchanged.  Not part of the product: it exists to MEASURE what the form buys in a kernel.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
template <int NR, int EPI, bool C16 = false, int SPLIT = 0>
__global__ void __launch_bounds__(512, 4) conv_narrow_pair_ws_kernel(const ConvParams a) {      // (4 waves per SIMD = two workgroups per CU)
    static_assert(SPLIT == 0 || !C16, "the split form is built for <= 8 input channels");
    const bool f_add = EPI < 0 ? a.add.p != nullptr : (EPI & PAIR_EPI_ADD) != 0;
    const bool f_relu = EPI < 0 ? a.relu != 0 : (EPI & PAIR_EPI_RELU) != 0;
    const bool f_mask = EPI < 0 ? a.mask.p != nullptr : (EPI & PAIR_EPI_MASK) != 0;
    const bool f_acc = EPI < 0 ? a.accumulate != 0 : (EPI & PAIR_EPI_ACC) != 0;
    const bool f_pool = EPI < 0 ? a.pool != nullptr : (EPI & PAIR_EPI_POOL) != 0;
    const bool f_aff = EPI < 0 ? a.in.sc != nullptr : (EPI & PAIR_EPI_AFF) != 0;       // channel affine of the input view (loaders)
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    constexpr int PTW = 32, PTH = 4 * NR;
    constexpr int TWH = PTW + 2, THH = PTH + 2, HPIX = TWH * THH;
    constexpr int P = SPLIT ? 12 : (C16 ? 20 : 10);              // floats per staged pixel (SPLIT: three 16-byte bf16 planes)
    constexpr int QPP = C16 ? 4 : 2;                              // channel quads staged per pixel
    constexpr int CPK = C16 ? 4 : 2;                              // channels per k-slot
    constexpr int PSTEP = 256 / QPP;                              // halo pixels a pass of the 256 loader threads covers
    constexpr int TOTAL = HPIX * QPP, ITERS = (TOTAL + 255) / 256;
    constexpr int TILE = HPIX * P;
    constexpr int OOB = (int)0xffffff00u, RSRC3 = 0x00020000;
    __shared__ __attribute__((aligned(16))) float lds[2 * TILE];
    // per tile, written by the loaders one tile ahead: byte addresses of the tile's origin in out / add / mask + the (rows, columns)
    // of the tile that exist -- the MFMA waves' epilogue used to spend 40 % of their time on this arithmetic, starved by the other
    // workgroup's MFMAs, while the loader waves idle 80 % of theirs (PAIR_WS_TRACE build)
    __shared__ __attribute__((aligned(16))) unsigned dsc[2][8];
    __shared__ __attribute__((aligned(16))) float pool_red[2][32];     // a.pool: [parity][wave][channel]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave8 = tid >> 6;
    // XCD-aware walk (a.CK != 0, grid a multiple of 8): workgroup b runs on XCD b % 8; each XCD takes one contiguous eighth of the tiles
    // and its workgroups walk it with the XCD's stride, so the tiles in flight on an XCD are neighbours (consecutive tile rows of
    // one image) and their halo rows meet in that XCD's L2 instead of being fetched once per XCD
    const int ntiles_all = a.tiles_x * a.tiles_y * a.in.N;
    int T0 = blockIdx.x, G = gridDim.x, ntiles = ntiles_all;
    if (a.CK && (G & 7) == 0) {
        const int per = (ntiles_all + 7) >> 3, xcd = blockIdx.x & 7;
        T0 = xcd * per + (blockIdx.x >> 3);
        G >>= 3;
        ntiles = min(ntiles_all, (xcd + 1) * per);
    }
#ifdef PAIR_WS_TRACE
    unsigned long long pt_t = 0, pt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    auto origin = [&](int t, int& n, int& y0, int& x0) {
        const int q = fast_div(t, a.m_txy[0]);
        const int bx = t - q * a.tiles_x;
        n = fast_div(q, a.m_txy[1]);
        const int by = q - n * a.tiles_y;
        x0 = bx * PTW; y0 = by * PTH;
    };

#ifdef PAIR_WS_DESYNC
    // (experiment: the two workgroups of a CU start in phase and stay in phase -- both in their K loops, then both in their
    //  epilogues; delay one of them by part of a tile time.  1: odd workgroups, 2: the upper half of the grid)
    if (PAIR_WS_DESYNC == 1 ? (blockIdx.x & 1) : ((int)blockIdx.x >= ((int)gridDim.x >> 1))) {
#pragma unroll
        for (int q_ = 0; q_ < PAIR_WS_DESYNC_N; ++q_) __builtin_amdgcn_s_sleep(32);
    }
#endif
    if (wave8 >= 4) {
#ifdef PAIR_WS_LOADER_PRIO
        __builtin_amdgcn_s_setprio(PAIR_WS_LOADER_PRIO);
#endif
        // ---- loaders: element e = htid + 256 u = (halo pixel (htid >> 1) + 128 u, channel quad htid & 1)
        const int htid = tid & 255;
        const int c4 = htid & (QPP - 1), p0 = htid / QPP;
        const size_t isx = a.in.ld, isy = (size_t)a.W * a.in.ld;
        const bool ragged = (a.Cin & 3) != 0 || (a.in.ld & 3) != 0;     // quads may reach beyond the end of the view
        const long in_total = (long)((size_t)(a.in.N - 1) * a.in.nstride + (size_t)a.H * a.W * a.in.ld);
        int rel[ITERS], soff[ITERS], hyx[ITERS];
#pragma unroll
        for (int u = 0; u < ITERS; ++u) {
            const int pix = p0 + PSTEP * u;
            const int hy = pix / TWH, hx = pix - hy * TWH;
            const bool live = pix < HPIX && c4 * 4 < a.Cin;
            hyx[u] = pix < HPIX ? ((hy << 8) | hx) : 0x7f7f;
            rel[u] = live ? (int)((hy * isy + hx * isx + (size_t)c4 * 4) * 4) : OOB;
            soff[u] = rel[u];
        }
        int sig_cur = (THH << 8) | TWH;
        i32x4_t r[3][ITERS];
        unsigned vm[3] = {0u, 0u, 0u};      // a.in.sc: which elements of a register set lie inside the image (the affine must not touch the padding)
        // ... and the channel affine of the NEXT tile to be written to LDS for this thread's quad, requested right after the previous
        // tile's LDS writes and BEFORE the next batch of tile loads (vector memory returns in order: requested at write time it
        // would wait for the two tiles requested in between)
        float4 as4 = make_float4(1.f, 1.f, 1.f, 1.f), ah4 = make_float4(0.f, 0.f, 0.f, 0.f);
        auto load_affine = [&](int t) __attribute__((always_inline)) {
            if (f_aff && t < ntiles) { int n, y0, x0; origin(t, n, y0, x0); view_affine4(a.in, n, c4 * 4, as4, ah4); }
        };
        auto issue = [&](int t, i32x4_t (&dst)[ITERS], unsigned& vmask) __attribute__((always_inline)) {
            int n, y0, x0;
            origin(t, n, y0, x0);
            const int ylo = max(0, 1 - y0), yhi = min(THH, a.H + 1 - y0);
            const int xlo = max(0, 1 - x0), xhi = min(TWH, a.W + 1 - x0);
            const int sig = (ylo << 24) | (xlo << 16) | (yhi << 8) | xhi;
            if (sig != sig_cur) {
                sig_cur = sig;
#pragma unroll
                for (int u = 0; u < ITERS; ++u) {
                    const int hy = hyx[u] >> 8, hx = hyx[u] & 0xff;
                    soff[u] = (hy >= ylo && hy < yhi && hx >= xlo && hx < xhi) ? rel[u] : OOB;
                }
            }
            const long org = (long)((size_t)n * a.in.nstride) + (long)(y0 - 1) * (long)isy + (long)(x0 - 1) * (long)isx;
            int nrec = 0x7fffff00;
            if (ragged) { const long rem = (in_total - org) * 4; nrec = rem < 0x7fffff00l ? (int)rem : 0x7fffff00; }
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(reinterpret_cast<const char*>(a.in.p)) + org * 4, 0, nrec, RSRC3);
            if (f_aff) {
                vmask = 0u;
#pragma unroll
                for (int u = 0; u < ITERS; ++u) vmask |= (soff[u] != OOB ? 1u : 0u) << u;
            }
#pragma unroll
            for (int u = 0; u < ITERS; ++u) {
#if defined(PAIR_WS_ABL) && PAIR_WS_ABL == 3
                dst[u] = (i32x4_t){0, 0, 0, 0};
#else
                dst[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, soff[u], 0, 0);
#endif
            }
        };
        auto put = [&](const i32x4_t (&src)[ITERS], float* tile, unsigned vmask) __attribute__((always_inline)) {
            const float4 s4 = as4, h4 = ah4;
            float* d0 = tile + p0 * P + c4 * 4;
            // channel affine of the view (ChannelAttention2D's scale / its backward, see TView): this thread's channel quad is
            // fixed, the image is the tile's; applied on the way into LDS, inside the image only
#pragma unroll
            for (int u = 0; u < ITERS; ++u) {
                if (u + 1 < ITERS || (hyx[u] >> 8) < THH) {
                    i32x4_t v = src[u];
                    if (f_aff && ((vmask >> u) & 1u)) {
                        const float4 f = affine4(make_float4(__int_as_float(v[0]), __int_as_float(v[1]), __int_as_float(v[2]), __int_as_float(v[3])), s4, h4);
                        v = (i32x4_t){__float_as_int(f.x), __float_as_int(f.y), __float_as_int(f.z), __float_as_int(f.w)};
                    }
                    if constexpr (SPLIT != 0) {
                        // this thread's channel quad c4 of the pixel: 8 bytes at offset 8 c4 of each plane
                        bf16x4_t hi, lo, lo2;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float x = __int_as_float(v[q]);
                            hi[q] = (__bf16)x;
                            const float r1 = x - (float)hi[q];
                            lo[q] = (__bf16)r1;
                            lo2[q] = (__bf16)(r1 - (float)lo[q]);
                        }
                        char* dp = reinterpret_cast<char*>(tile) + (size_t)(p0 + u * PSTEP) * 48 + c4 * 8;
                        *reinterpret_cast<bf16x4_t*>(dp) = hi;
                        *reinterpret_cast<bf16x4_t*>(dp + 16) = lo;
                        *reinterpret_cast<bf16x4_t*>(dp + 32) = lo2;
                    } else {
                    int2* d = reinterpret_cast<int2*>(d0 + u * (PSTEP * P));
                    d[0] = make_int2(v[0], v[1]);
                    d[1] = make_int2(v[2], v[3]);
                    }
                }
            }
        };
        size_t hosx, hosy, hasx = 0, hasy = 0, hmsx = 0, hmsy = 0;
        auto strides_of = [](const TView& v, size_t& sx, size_t& sy) {
            const int rr = v.d2s > 1 ? v.d2s : 1;
            sx = (size_t)rr * v.ld;
            sy = (size_t)rr * (size_t)(v.W * rr) * v.ld;
        };
        strides_of(a.out, hosx, hosy);
        if (f_add) strides_of(a.add, hasx, hasy);
        if (f_mask) strides_of(a.mask, hmsx, hmsy);
        auto describe = [&](int t, unsigned* d) __attribute__((always_inline)) {
            int n, y0, x0;
            origin(t, n, y0, x0);
            // (every operand through its OWN pixel strides: the output may live inside a Concatenate's buffer -- pixel pitch 16 --
            //  while the residual operand or the ReLU mask is a dense 8-channel tensor, and the other way round)
            const size_t pb = (size_t)y0 * hosy + (size_t)x0 * hosx;
            const unsigned long long ob = (unsigned long long)(uintptr_t)a.out.p + ((size_t)n * a.out.nstride + pb) * 4;
            const unsigned long long ab = (unsigned long long)(uintptr_t)a.add.p + (f_add ? ((size_t)n * a.add.nstride + (size_t)y0 * hasy + (size_t)x0 * hasx) * 4 : 0);
            const unsigned long long mb = (unsigned long long)(uintptr_t)a.mask.p + (f_mask ? ((size_t)n * a.mask.nstride + (size_t)y0 * hmsy + (size_t)x0 * hmsx) * 4 : 0);
            const int ymax = min(PTH, a.H - y0), xmax = min(PTW, a.W - x0);
            if (tid == 256) {
                *reinterpret_cast<uint4*>(d) = make_uint4((unsigned)ob, (unsigned)(ob >> 32), (unsigned)ab, (unsigned)(ab >> 32));
                *reinterpret_cast<uint4*>(d + 4) = make_uint4((unsigned)mb, (unsigned)(mb >> 32), (unsigned)((ymax << 8) | xmax), 0u);
            }
        };
        const int t0 = T0;
        if (t0 < ntiles) describe(t0, dsc[0]);
        load_affine(t0);
        if (t0 < ntiles) issue(t0, r[0], vm[0]);
        if (t0 + G < ntiles) issue(t0 + G, r[1], vm[1]);
        if (t0 < ntiles) put(r[0], lds, vm[0]);
        load_affine(t0 + G);
        if (t0 + 2 * G < ntiles) issue(t0 + 2 * G, r[2], vm[2]);
        __syncthreads();                                          // S0: tile 0 staged
        // iteration k (beside the MFMAs of tile k): write tile k+1, request tile k+3 into the set tile k used
        int k = 0;
        for (int t = t0; t < ntiles; t += 3 * G) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int tk = t + j * G;
                if (tk < ntiles) {
                    PT_TIC();
                    if (tk + G < ntiles) { put(r[(j + 1) % 3], lds + ((k + 1) & 1) * TILE, vm[(j + 1) % 3]); describe(tk + G, dsc[(k + 1) & 1]); }
                    PT_TOC(4);
                    load_affine(tk + 2 * G);
                    if (tk + 3 * G < ntiles) issue(tk + 3 * G, r[j], vm[j]);
                    PT_TOC(5);
                    __syncthreads();                              // X: tile k consumed, tile k+1 staged
                    PT_TOC(6);
                    ++k;
                }
            }
        }
#ifdef PAIR_WS_TRACE
        if (tid == 256) for (int q_ = 4; q_ < 8; ++q_) pair_ws_trace[(size_t)blockIdx.x * 8 + q_] = pt_acc[q_];
#endif
        if (f_pool) __syncthreads();                               // P: the MFMA waves' last pooling record (below)
        return;
    }

    // ---- MFMA waves
    const int wave = wave8 & 3;
    const int l15 = lane & 15, lq = lane >> 4;
    bf16x8_t ws[SPLIT ? 3 : 1][3];                                // SPLIT: [dy][plane], row (h, co) x k-group ux = lq: channels 0 .. 7
    if constexpr (SPLIT != 0) {
        const int h = l15 >> 3, co = l15 & 7, kx = lq - h;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) {
                const bool ok = kx >= 0 && kx <= 2 && ci < a.Cin && co < a.Cout;
                const float xv = a.w[((size_t)(dy * 3 + (ok ? kx : 0)) * a.Cin + (ok ? ci : 0)) * a.Cout + (ok ? co : 0)];
                const float x = ok ? xv : 0.f;
                ws[dy][0][ci] = (__bf16)x;
                const float r1 = x - (float)ws[dy][0][ci];
                ws[dy][1][ci] = (__bf16)r1;
                ws[dy][2][ci] = (__bf16)(r1 - (float)ws[dy][1][ci]);
            }
    }
    float wr[3][4][CPK];
    if constexpr (SPLIT == 0) {
        const int h = l15 >> 3, co = l15 & 7;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int ux = 0; ux < 4; ++ux)
#pragma unroll
                for (int e = 0; e < CPK; ++e) {
                    const int kx = ux - h, ci = CPK * lq + e;
                    const bool ok = kx >= 0 && kx <= 2 && ci < a.Cin && co < a.Cout;
                    const float v = a.w[((size_t)(dy * 3 + (ok ? kx : 0)) * a.Cin + (ok ? ci : 0)) * a.Cout + (ok ? co : 0)];
                    wr[dy][ux][e] = ok ? v : 0.f;
                }
    }
    const int rd_off = ((wave * NR) * TWH + 2 * l15) * P + CPK * lq;
    // lane (pair column l15, k-slot lq) holds rows 4*lq + r = (h = lq >> 1, couts 4*(lq & 1) + r)
    const int eh = lq >> 1, ec = 4 * (lq & 1);
    const bool c_ok = ec < a.Cout;
    const float4 bias_v = (a.bias && c_ok) ? *reinterpret_cast<const float4*>(a.bias + ec) : make_float4(0.f, 0.f, 0.f, 0.f);
    const f32x4 bias_c = {bias_v.x, bias_v.y, bias_v.z, bias_v.w};
    size_t osx, osy;
    {
        const int r = a.out.d2s > 1 ? a.out.d2s : 1;
        osx = (size_t)r * a.out.ld;
        osy = (size_t)r * (size_t)(a.out.W * r) * a.out.ld;
    }
    // per-lane offsets of the four output rows, for the output and -- through their own strides -- for the residual operand and the
    // mask; recomputed (a handful of multiplications) only when the tile's border signature changes
    size_t asx = 0, asy = 0, msx = 0, msy = 0;
    auto strides_of = [](const TView& v, size_t& sx, size_t& sy) {
        const int rr = v.d2s > 1 ? v.d2s : 1;
        sx = (size_t)rr * v.ld;
        sy = (size_t)rr * (size_t)(v.W * rr) * v.ld;
    };
    if (f_add) strides_of(a.add, asx, asy);
    if (f_mask) strides_of(a.mask, msx, msy);
    const int co_ch = (int)view_chan_off(a.out, c_ok ? ec : 0) * 4;
    const int ca_ch = f_add ? (int)view_chan_off(a.add, c_ok ? ec : 0) * 4 : 0;
    const int cm_ch = f_mask ? (int)view_chan_off(a.mask, c_ok ? ec : 0) * 4 : 0;
    int eoff[NR], aoff[NR], moff[NR];
    auto set_offsets = [&](int ymax, int xmax) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const bool in = c_ok && wave * NR + i < ymax && 2 * l15 + eh < xmax;
            eoff[i] = in ? (int)(((wave * NR + i) * osy + (2 * l15 + eh) * osx) * 4) + co_ch : OOB;
            aoff[i] = (f_add && in) ? (int)(((wave * NR + i) * asy + (2 * l15 + eh) * asx) * 4) + ca_ch : OOB;
            moff[i] = (f_mask && in) ? (int)(((wave * NR + i) * msy + (2 * l15 + eh) * msx) * 4) + cm_ch : OOB;
        }
    };
    set_offsets(PTH, PTW);
    int esig = (PTH << 8) | PTW;

    __syncthreads();                                              // S0
    int k = 0;
    for (int t = T0; t < ntiles; t += G, ++k) {
        const float* rd = lds + (k & 1) * TILE + rd_off;
        PT_TIC();
        // the tile's descriptors come from the loaders (dsc[k & 1]: written during the previous step, overwritten for tile k + 2
        // during the NEXT one -- so they are taken into scalar registers before this step's barrier)
        const uint4 d0 = *reinterpret_cast<const uint4*>(dsc[k & 1]);
        const uint4 d1 = *reinterpret_cast<const uint4*>(dsc[k & 1] + 4);
        auto sg = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
        const unsigned long long ob = ((unsigned long long)sg(d0.y) << 32) | sg(d0.x);
        const unsigned long long ab = ((unsigned long long)sg(d0.w) << 32) | sg(d0.z);
        const unsigned long long mb = ((unsigned long long)sg(d1.y) << 32) | sg(d1.x);
        const int sig = (int)sg(d1.z);
        if (sig != esig) {
            esig = sig;
            set_offsets(sig >> 8, sig & 0xff);
        }
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>((uintptr_t)ob), 0, 0x7fffff00, RSRC3);
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>((uintptr_t)ab), 0, 0x7fffff00, RSRC3);
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>((uintptr_t)mb), 0, 0x7fffff00, RSRC3);
        // the epilogue's operands are requested BEFORE the K loop (compiled forms): requested in the epilogue their latency was
        // exposed once per tile (cfg5: the residual + ReLU form 155 us against 122 us for the plain ReLU form with 1.5 x the bytes).
        // Two operands at once (32 registers) only where the budget of 128 holds them.
        constexpr bool PRE = EPI >= 0;
        // (all three operands at once do not fit 128 registers beside their offsets: the old value is then requested after the K loop)
        constexpr bool PRE_OLD = PRE && !((EPI & PAIR_EPI_ADD) && (EPI & PAIR_EPI_MASK) && (EPI & PAIR_EPI_ACC));
        i32x4_t ad[NR], mk[NR], old[NR];
        if (PRE) {
            if (f_add) {
#pragma unroll
                for (int i = 0; i < NR; ++i) ad[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, aoff[i], 0, 0);
            }
            if (f_mask) {
#pragma unroll
                for (int i = 0; i < NR; ++i) mk[i] = __builtin_amdgcn_raw_buffer_load_b128(rm, moff[i], 0, 0);
            }
            if (f_acc && PRE_OLD) {
#pragma unroll
                for (int i = 0; i < NR; ++i) old[i] = __builtin_amdgcn_raw_buffer_load_b128(ro, eoff[i], 0, 0);
            }
        }
        f32x4 acc[NR];
        // pixel fragments ONE halo row ahead of the MFMAs that use them: without it every pair of rows waited for its eight LDS
        // reads (ablation: no loads / no stores change nothing, no MFMAs -> 26 us of 79; the K loop itself ran at 65 %).  Two rows
        // ahead costs 16 more registers and with them the second workgroup per CU (78 -> 87 us).
        float pv[2][4][CPK];
        auto fetch = [&](float (&dst)[CPK], const float* src) __attribute__((always_inline)) {
            if constexpr (C16) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src);
                dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
            } else {
                const float2 v = *reinterpret_cast<const float2*>(src);
                dst[0] = v.x; dst[1] = v.y;
            }
        };
        if constexpr (SPLIT != 0) {
            // lane (pair column l15, k-group lq = tap column ux): the eight channels of halo pixel 2 l15 + lq, one 16-byte read per plane
            constexpr int NPL = SPLIT >= 6 ? 3 : 2;
            const char* rb = reinterpret_cast<const char*>(lds + (k & 1) * TILE) + (size_t)((wave * NR) * TWH + 2 * l15 + lq) * 48;
            bf16x8_t pb[2][NPL];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) pb[0][pl] = *reinterpret_cast<const bf16x8_t*>(rb + pl * 16);
#pragma unroll
            for (int rho = 0; rho < NR + 2; ++rho) {
                if (rho + 1 < NR + 2) {
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) pb[(rho + 1) & 1][pl] = *reinterpret_cast<const bf16x8_t*>(rb + (size_t)(rho + 1) * TWH * 48 + pl * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
                // terms (filter plane, pixel plane), small ones first; consecutive MFMAs go to different accumulators
                constexpr int NT_ = SPLIT >= 6 ? 6 : 3;
                constexpr int TA[6] = {1, 0, 2, 0, 1, 0}, TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
                for (int tm = 6 - NT_; tm < 6; ++tm) {
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int r = rho - dy;
                        if (r >= 0 && r < NR) {
                            const bool first = dy == 0 && tm == 6 - NT_;        // this row's first MFMA: C = bias
                            acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws[dy][TA[tm]], pb[rho & 1][TB[tm]], first ? bias_c : acc[r], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#if defined(PAIR_WS_ABL) && PAIR_WS_ABL == 1
        for (int i_ = 0; i_ < NR; ++i_) acc[i_] = bias_c;
        for (int rho = 0; rho < 0; ++rho) {
#else
#pragma unroll
        for (int ux = 0; ux < 4; ++ux) fetch(pv[0][ux], rd + ux * P);
#pragma unroll
        for (int rho = 0; rho < NR + 2; ++rho) {
#endif
            if (rho + 1 < NR + 2) {
#pragma unroll
                for (int ux = 0; ux < 4; ++ux) fetch(pv[(rho + 1) & 1][ux], rd + ((rho + 1) * TWH + ux) * P);
            }
            __builtin_amdgcn_sched_barrier(0);
            // (consecutive MFMAs go to different accumulators wherever a halo row feeds more than one output row)
#pragma unroll
            for (int ux = 0; ux < 4; ++ux) {
#pragma unroll
                for (int e = 0; e < CPK; ++e) {
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int r = rho - dy;
                        if (r >= 0 && r < NR) {
                            const bool first = dy == 0 && ux == 0 && e == 0;    // this row's first MFMA: C = bias
                            acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[dy][ux][e], pv[rho & 1][ux][e], first ? bias_c : acc[r], 0, 0, 0);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        PT_TOC(0);
        __syncthreads();                                          // X
        PT_TOC(1);
        // a.pool: the previous tile's per-wave channel sums are complete behind this barrier: one 8-float record per tile
        if (f_pool && k > 0 && tid < 8) {
            const float* red = pool_red[(k - 1) & 1];
            a.pool[(size_t)(t - G) * 8 + tid] = (red[tid] + red[8 + tid]) + (red[16 + tid] + red[24 + tid]);
        }
        f32x4 psum = {0.f, 0.f, 0.f, 0.f};          // a.pool: this lane's share of the tile's channel sums
        if (PRE && !PRE_OLD && f_acc) {
#pragma unroll
            for (int i = 0; i < NR; ++i) old[i] = __builtin_amdgcn_raw_buffer_load_b128(ro, eoff[i], 0, 0);
        }
        if (!PRE) {
            if (f_add) {
#pragma unroll
                for (int i = 0; i < NR; ++i) ad[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, aoff[i], 0, 0);
            }
            if (f_mask) {
#pragma unroll
                for (int i = 0; i < NR; ++i) mk[i] = __builtin_amdgcn_raw_buffer_load_b128(rm, moff[i], 0, 0);
            }
            if (f_acc) {
#pragma unroll
                for (int i = 0; i < NR; ++i) old[i] = __builtin_amdgcn_raw_buffer_load_b128(ro, eoff[i], 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            f32x4 v = acc[i];
            if (f_add) v += __builtin_bit_cast(f32x4, ad[i]);
            if (f_relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            if (f_mask) {
                const f32x4 m = __builtin_bit_cast(f32x4, mk[i]);
                v[0] = m[0] > 0.f ? v[0] : 0.f; v[1] = m[1] > 0.f ? v[1] : 0.f;
                v[2] = m[2] > 0.f ? v[2] : 0.f; v[3] = m[3] > 0.f ? v[3] : 0.f;
            }
            if (f_acc) v += __builtin_bit_cast(f32x4, old[i]);
#if defined(PAIR_WS_ABL) && PAIR_WS_ABL == 2
            if (v[0] == 12345.678f)
#endif
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), ro, eoff[i], 0, 0);
            if (f_pool && eoff[i] != OOB) psum += v;
        }
        if (f_pool) {
            // GlobalAveragePooling of ChannelAttention2D (blocks.py:585-588) from this epilogue: per-tile channel sums in the
            // fixed order of conv_narrow_pair_kernel (16 pair columns, the two pixels of a pair, the four waves through LDS)
#pragma unroll
            for (int mk_ = 1; mk_ < 16; mk_ <<= 1) {
                psum[0] += __shfl_xor(psum[0], mk_, 64); psum[1] += __shfl_xor(psum[1], mk_, 64);
                psum[2] += __shfl_xor(psum[2], mk_, 64); psum[3] += __shfl_xor(psum[3], mk_, 64);
            }
            psum[0] += __shfl_xor(psum[0], 32, 64); psum[1] += __shfl_xor(psum[1], 32, 64);
            psum[2] += __shfl_xor(psum[2], 32, 64); psum[3] += __shfl_xor(psum[3], 32, 64);
            if (lane == 0 || lane == 16) *reinterpret_cast<f32x4*>(pool_red[k & 1] + wave * 8 + (lane >> 4) * 4) = psum;
        }
        PT_TOC(2);
#ifdef PAIR_WS_TRACE
        pt_acc[3] += 1;
#endif
    }
    if (f_pool) {
        __syncthreads();                                          // P (the loaders arrive here too)
        if (k > 0 && tid < 8) {
            const float* red = pool_red[(k - 1) & 1];
            const int tl = T0 + (k - 1) * G;
            a.pool[(size_t)tl * 8 + tid] = (red[tid] + red[8 + tid]) + (red[16 + tid] + red[24 + tid]);
        }
    }
#ifdef PAIR_WS_TRACE
    if (tid == 0) for (int q_ = 0; q_ < 4; ++q_) pair_ws_trace[(size_t)blockIdx.x * 8 + q_] = pt_acc[q_];
#endif
}

// Producer / consumer form for 9..16 input channels (the 16-channel levels of the U-Net at 512^2 / 256^2, the recurrent nets'
// 16-channel upsampling convolution): the same division of labour as conv_narrow_pair_ws_kernel -- waves 4-7 move halo tiles
// (zero-filling buffer loads three tiles ahead, LDS writes one tile ahead into the other of two buffers) and prepare the tile's
// store descriptors, waves 0-3 hold the 16 x 16 x 9 filter in 36 registers and issue nothing but the 36 MFMAs per halo row
// (pixel fragments as one ds_read_b128 per tap column, one halo row ahead; accumulators start at the bias) plus one 16-byte
// store per output row.  conv_narrow_kernel<16> -- load, LDS, MFMA and store phases inside every wave -- held the matrix
// pipe 43 % busy (profiles/pmc_mfma_r02.txt) at 74 TFLOP/s stand-alone, 58-63 inside the models.
// Waves per SIMD: the compiled forms with at most two epilogue operands fit 128 registers = TWO workgroups per CU (62 KB of
// LDS each), so that one workgroup's K loop runs beside the other's epilogue; the run-time form and the three-operand forms
// need 144 and keep one workgroup per CU (capped at 128 they spill and run at half the rate).
constexpr int narrow16_wps(int epi) {
#ifdef NARROW16_WPS
    return NARROW16_WPS;
#else
    return (epi >= 0 && ((epi & 1) + ((epi >> 2) & 1) + ((epi >> 3) & 1)) <= 2) ? 4 : 2;
#endif
}
// C8 (round 5): the form for <= 8 INPUT channels (8 -> 13: the dgrad of ConvBlock_att's first layer in the recurrent nets).  With the
// 16-channel K layout half of every MFMA's k-slots met zero pixels (36 MFMAs per output row for 9 x 8 channels of work).  Here k-slots
// 0, 1 carry the two channel quads of tap 2j and k-slots 2, 3 those of tap 2j + 1: the nine taps take FIVE groups of four MFMAs
// per output row (the tenth half-group has a zero filter), each lane reading the pixel of ITS tap (one ds_read_b128 per group at a
// per-lane offset); pixels are staged 8 floats apart (two 16-byte slots: conflict-free by the same rule as the pitch of 24).
template <int NR, int EPI, bool C8 = false>      // EPI: compiled epilogue form (PAIR_EPI_* bits) or -1 = run-time, as in conv_narrow_pair_ws_kernel
__global__ void __launch_bounds__(512, narrow16_wps(EPI)) conv_narrow16_ws_kernel(const ConvParams a) {
    const bool f_add = EPI < 0 ? a.add.p != nullptr : (EPI & PAIR_EPI_ADD) != 0;
    const bool f_relu = EPI < 0 ? a.relu != 0 : (EPI & PAIR_EPI_RELU) != 0;
    const bool f_mask = EPI < 0 ? a.mask.p != nullptr : (EPI & PAIR_EPI_MASK) != 0;
    const bool f_acc = EPI < 0 ? a.accumulate != 0 : (EPI & PAIR_EPI_ACC) != 0;    // (2 waves per SIMD: ONE workgroup per CU -- 144 registers; capped at 128 for two workgroups it spills and runs at half the rate)
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    constexpr int PTW = 16, PTH = 4 * NR;
    constexpr int TWH = PTW + 2, THH = PTH + 2, HPIX = TWH * THH;
#ifndef NARROW16_PITCH
#define NARROW16_PITCH 24
#endif
    // floats per staged pixel: 6 x 16 bytes.  ds_read_b128 is serviced in four non-contiguous 16-lane groups that mix eight
    // pixels of one k-slot with the other eight of the next (MI355X_MICROARCH.md, LDS): with the k-slots one 16-byte slot apart
    // the 16 lanes cover all 64 banks iff the pitch is 2 (mod 4) slots; the odd pitch (20 floats) lost 0.46 of the LDS cycles
    constexpr int P = C8 ? 8 : NARROW16_PITCH;
    constexpr int QPP = C8 ? 2 : 4;                               // channel quads staged per pixel
    constexpr int TOTAL = HPIX * QPP, ITERS = (TOTAL + 255) / 256;
    constexpr int TILE = HPIX * P;
    constexpr int OOB = (int)0xffffff00u, RSRC3 = 0x00020000;
    __shared__ __attribute__((aligned(16))) float lds[2 * TILE];
    __shared__ __attribute__((aligned(16))) unsigned dsc[2][8];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave8 = tid >> 6;
    // XCD-aware walk (a.CK != 0, grid a multiple of 8): workgroup b runs on XCD b % 8; each XCD takes one contiguous eighth of the tiles
    // and its workgroups walk it with the XCD's stride, so the tiles in flight on an XCD are neighbours (consecutive tile rows of
    // one image) and their halo rows meet in that XCD's L2 instead of being fetched once per XCD
    const int ntiles_all = a.tiles_x * a.tiles_y * a.in.N;
    int T0 = blockIdx.x, G = gridDim.x, ntiles = ntiles_all;
    if (a.CK && (G & 7) == 0) {
        const int per = (ntiles_all + 7) >> 3, xcd = blockIdx.x & 7;
        T0 = xcd * per + (blockIdx.x >> 3);
        G >>= 3;
        ntiles = min(ntiles_all, (xcd + 1) * per);
    }
    auto origin = [&](int t, int& n, int& y0, int& x0) {
        const int q = fast_div(t, a.m_txy[0]);
        const int bx = t - q * a.tiles_x;
        n = fast_div(q, a.m_txy[1]);
        const int by = q - n * a.tiles_y;
        x0 = bx * PTW; y0 = by * PTH;
    };

    if (wave8 >= 4) {
#ifdef PAIR_WS_LOADER_PRIO
        __builtin_amdgcn_s_setprio(PAIR_WS_LOADER_PRIO);
#endif
        // ---- loaders: element e = htid + 256 u = (halo pixel (htid >> 2) + 64 u, channel quad htid & 3)   [C8: htid >> 1, + 128 u, htid & 1]
        const int htid = tid & 255;
        constexpr int PSTEP = 256 / QPP;
        const int c4 = htid & (QPP - 1), p0 = htid / QPP;
        const size_t isx = a.in.ld, isy = (size_t)a.W * a.in.ld;
        // channel counts / pixel pitches that are not multiples of four (the 13-channel layer behind TransitionLast 26 -> 13,
        // 16-channel slices of a 26-channel concatenation): the 16-byte buffer loads only need dword alignment, what a quad
        // picks up beyond Cin meets zero filter entries, the descriptor's exact size zero-fills at the very end of the view
        const bool ragged = (a.Cin & 3) != 0 || (a.in.ld & 3) != 0;
        const long in_total = (long)((size_t)(a.in.N - 1) * a.in.nstride + (size_t)a.H * a.W * a.in.ld);
        int rel[ITERS], soff[ITERS], hyx[ITERS];
#pragma unroll
        for (int u = 0; u < ITERS; ++u) {
            const int pix = p0 + PSTEP * u;
            const int hy = pix / TWH, hx = pix - hy * TWH;
            const bool live = pix < HPIX && c4 * 4 < a.Cin;
            hyx[u] = pix < HPIX ? ((hy << 8) | hx) : 0x7f7f;
            rel[u] = live ? (int)((hy * isy + hx * isx + (size_t)c4 * 4) * 4) : OOB;
            soff[u] = rel[u];
        }
        int sig_cur = (THH << 8) | TWH;
        i32x4_t r[3][ITERS];
        auto issue = [&](int t, i32x4_t (&dst)[ITERS]) __attribute__((always_inline)) {
            int n, y0, x0;
            origin(t, n, y0, x0);
            const int ylo = max(0, 1 - y0), yhi = min(THH, a.H + 1 - y0);
            const int xlo = max(0, 1 - x0), xhi = min(TWH, a.W + 1 - x0);
            const int sig = (ylo << 24) | (xlo << 16) | (yhi << 8) | xhi;
            if (sig != sig_cur) {
                sig_cur = sig;
#pragma unroll
                for (int u = 0; u < ITERS; ++u) {
                    const int hy = hyx[u] >> 8, hx = hyx[u] & 0xff;
                    soff[u] = (hy >= ylo && hy < yhi && hx >= xlo && hx < xhi) ? rel[u] : OOB;
                }
            }
            const long org = (long)((size_t)n * a.in.nstride) + (long)(y0 - 1) * (long)isy + (long)(x0 - 1) * (long)isx;
            int nrec = 0x7fffff00;
            if (ragged) { const long rem = (in_total - org) * 4; nrec = rem < 0x7fffff00l ? (int)rem : 0x7fffff00; }
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(reinterpret_cast<const char*>(a.in.p)) + org * 4, 0, nrec, RSRC3);
#pragma unroll
            for (int u = 0; u < ITERS; ++u) dst[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, soff[u], 0, 0);
        };
        auto put = [&](const i32x4_t (&src)[ITERS], float* tile) __attribute__((always_inline)) {
            float* d0 = tile + p0 * P + c4 * 4;
#pragma unroll
            for (int u = 0; u < ITERS; ++u)
                if (u + 1 < ITERS || (hyx[u] >> 8) < THH) *reinterpret_cast<i32x4_t*>(d0 + u * (PSTEP * P)) = src[u];
        };
        size_t hosx, hosy;
        {
            const int rr = a.out.d2s > 1 ? a.out.d2s : 1;
            hosx = (size_t)rr * a.out.ld;
            hosy = (size_t)rr * (size_t)(a.out.W * rr) * a.out.ld;
        }
        auto describe = [&](int t, unsigned* d) __attribute__((always_inline)) {
            int n, y0, x0;
            origin(t, n, y0, x0);
            const size_t pb = (size_t)y0 * hosy + (size_t)x0 * hosx;
            const unsigned long long ob = (unsigned long long)(uintptr_t)a.out.p + ((size_t)n * a.out.nstride + pb) * 4;
            const unsigned long long ab = (unsigned long long)(uintptr_t)a.add.p + (f_add ? ((size_t)n * a.add.nstride + pb) * 4 : 0);
            const unsigned long long mb = (unsigned long long)(uintptr_t)a.mask.p + (f_mask ? ((size_t)n * a.mask.nstride + pb) * 4 : 0);
            const int ymax = min(PTH, a.H - y0), xmax = min(PTW, a.W - x0);
            if (tid == 256) {
                *reinterpret_cast<uint4*>(d) = make_uint4((unsigned)ob, (unsigned)(ob >> 32), (unsigned)ab, (unsigned)(ab >> 32));
                *reinterpret_cast<uint4*>(d + 4) = make_uint4((unsigned)mb, (unsigned)(mb >> 32), (unsigned)((ymax << 8) | xmax), 0u);
            }
        };
        const int t0 = T0;
        if (t0 < ntiles) describe(t0, dsc[0]);
        if (t0 < ntiles) issue(t0, r[0]);
        if (t0 + G < ntiles) issue(t0 + G, r[1]);
        if (t0 < ntiles) put(r[0], lds);
        if (t0 + 2 * G < ntiles) issue(t0 + 2 * G, r[2]);
        __syncthreads();                                          // S0: tile 0 staged
        int k = 0;
        for (int t = t0; t < ntiles; t += 3 * G) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int tk = t + j * G;
                if (tk < ntiles) {
                    if (tk + G < ntiles) { put(r[(j + 1) % 3], lds + ((k + 1) & 1) * TILE); describe(tk + G, dsc[(k + 1) & 1]); }
                    if (tk + 3 * G < ntiles) issue(tk + 3 * G, r[j]);
                    __syncthreads();                              // X: tile k consumed, tile k+1 staged
                    ++k;
                }
            }
        }
        return;
    }

    // ---- MFMA waves: filter fragment (tap, e) = W[tap][cin = 4 lq + e][cout = l15]
    const int wave = wave8 & 3;
    const int l15 = lane & 15, lq = lane >> 4;
    constexpr int NWR = C8 ? 5 : 9;
    float wr[NWR][4];
    int toff[C8 ? 5 : 1];                                         // C8: float offset of this lane's tap of pair j inside the halo tile
    if constexpr (C8) {
        const int hi = lq >> 1, cq = lq & 1;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int tap = 2 * j + hi;
            toff[j] = tap < 9 ? ((tap / 3) * TWH + (tap % 3)) * P : 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ci = 4 * cq + e;
                const bool ok = tap < 9 && ci < a.Cin && l15 < a.Cout;
                const float v = a.w[((size_t)(ok ? tap : 0) * a.Cin + (ok ? ci : 0)) * a.Cout + (ok ? l15 : 0)];
                wr[j][e] = ok ? v : 0.f;
            }
        }
    } else {
        toff[0] = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ci = 4 * lq + e;
                const bool ok = ci < a.Cin && l15 < a.Cout;
                const float v = a.w[((size_t)t * a.Cin + (ok ? ci : 0)) * a.Cout + (ok ? l15 : 0)];
                wr[t][e] = ok ? v : 0.f;
            }
    }
    const int rd_off = ((wave * NR) * TWH + l15) * P + 4 * (C8 ? (lq & 1) : lq);
    // lane (pixel column l15, k-slot lq) ends with rows 4 lq + r = couts 4 lq .. 4 lq + 3 of its pixel
    const int ec = 4 * lq;
    const bool c_ok = ec < a.Cout;
    const int nvalid = min(max(a.Cout - ec, 0), 4);              // channels of this lane's quad that exist
    const float4 bias_v = (a.bias && c_ok) ? *reinterpret_cast<const float4*>(a.bias + ec) : make_float4(0.f, 0.f, 0.f, 0.f);
    const f32x4 bias_c = {bias_v.x, bias_v.y, bias_v.z, bias_v.w};
    size_t osx, osy;
    {
        const int r = a.out.d2s > 1 ? a.out.d2s : 1;
        osx = (size_t)r * a.out.ld;
        osy = (size_t)r * (size_t)(a.out.W * r) * a.out.ld;
    }
    int eo[NR], eoff[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        eo[i] = c_ok ? (int)(((wave * NR + i) * osy + l15 * osx + view_chan_off(a.out, c_ok ? ec : 0)) * 4) : OOB;
        eoff[i] = eo[i];
    }
    int esig = (PTH << 8) | PTW;

    __syncthreads();                                              // S0
    int k = 0;
    for (int t = T0; t < ntiles; t += G, ++k) {
        const float* rd = lds + (k & 1) * TILE + rd_off;
        const uint4 d0 = *reinterpret_cast<const uint4*>(dsc[k & 1]);
        const uint4 d1 = *reinterpret_cast<const uint4*>(dsc[k & 1] + 4);
        auto sg = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
        const unsigned long long ob = ((unsigned long long)sg(d0.y) << 32) | sg(d0.x);
        const unsigned long long ab = ((unsigned long long)sg(d0.w) << 32) | sg(d0.z);
        const unsigned long long mb = ((unsigned long long)sg(d1.y) << 32) | sg(d1.x);
        const int sig = (int)sg(d1.z);
        if (sig != esig) {
            esig = sig;
            const int ymax = sig >> 8, xmax = sig & 0xff;
#pragma unroll
            for (int i = 0; i < NR; ++i) eoff[i] = (wave * NR + i < ymax && l15 < xmax) ? eo[i] : OOB;
        }
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>((uintptr_t)ob), 0, 0x7fffff00, RSRC3);
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>((uintptr_t)ab), 0, 0x7fffff00, RSRC3);
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>((uintptr_t)mb), 0, 0x7fffff00, RSRC3);
        // compiled forms request the epilogue's operands before the K loop (see conv_narrow_pair_ws_kernel)
        constexpr bool PRE = EPI >= 0;
        i32x4_t ad[NR], mk[NR], old[NR];
        if (PRE) {
            if (f_add) {
#pragma unroll
                for (int i = 0; i < NR; ++i) ad[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, eoff[i], 0, 0);
            }
            if (f_mask) {
#pragma unroll
                for (int i = 0; i < NR; ++i) mk[i] = __builtin_amdgcn_raw_buffer_load_b128(rm, eoff[i], 0, 0);
            }
            if (f_acc) {
#pragma unroll
                for (int i = 0; i < NR; ++i) old[i] = __builtin_amdgcn_raw_buffer_load_b128(ro, eoff[i], 0, 0);
            }
        }
        f32x4 acc[NR];
        if constexpr (C8) {
            f32x4 pw[2][5];                                       // pixel fragments of the five tap pairs, ONE output row ahead
#pragma unroll
            for (int j = 0; j < 5; ++j) pw[0][j] = *reinterpret_cast<const f32x4*>(rd + toff[j]);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (r + 1 < NR) {
#pragma unroll
                    for (int j = 0; j < 5; ++j) pw[(r + 1) & 1][j] = *reinterpret_cast<const f32x4*>(rd + (r + 1) * TWH * P + toff[j]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const f32x4 v = pw[r & 1][j];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[j][e], v[e], (j == 0 && e == 0) ? bias_c : acc[r], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
        f32x4 pv[2][3];                                           // pixel fragments, ONE halo row ahead of the MFMAs that use them
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) pv[0][dx] = *reinterpret_cast<const f32x4*>(rd + dx * P);
#pragma unroll
        for (int rho = 0; rho < NR + 2; ++rho) {
            if (rho + 1 < NR + 2) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) pv[(rho + 1) & 1][dx] = *reinterpret_cast<const f32x4*>(rd + ((rho + 1) * TWH + dx) * P);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const f32x4 v = pv[rho & 1][dx];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int r = rho - dy;
                        if (r >= 0 && r < NR) {
                            const bool first = dy == 0 && dx == 0 && e == 0;    // this row's first MFMA: C = bias
                            acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[dy * 3 + dx][e], v[e], first ? bias_c : acc[r], 0, 0, 0);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        __syncthreads();                                          // X
        if (!PRE) {
            if (f_add) {
#pragma unroll
                for (int i = 0; i < NR; ++i) ad[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, eoff[i], 0, 0);
            }
            if (f_mask) {
#pragma unroll
                for (int i = 0; i < NR; ++i) mk[i] = __builtin_amdgcn_raw_buffer_load_b128(rm, eoff[i], 0, 0);
            }
            if (f_acc) {
#pragma unroll
                for (int i = 0; i < NR; ++i) old[i] = __builtin_amdgcn_raw_buffer_load_b128(ro, eoff[i], 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            f32x4 v = acc[i];
            if (f_add) v += __builtin_bit_cast(f32x4, ad[i]);
            if (f_relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            if (f_mask) {
                const f32x4 m = __builtin_bit_cast(f32x4, mk[i]);
                v[0] = m[0] > 0.f ? v[0] : 0.f; v[1] = m[1] > 0.f ? v[1] : 0.f;
                v[2] = m[2] > 0.f ? v[2] : 0.f; v[3] = m[3] > 0.f ? v[3] : 0.f;
            }
            if (f_acc) v += __builtin_bit_cast(f32x4, old[i]);
            if (nvalid == 4) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), ro, eoff[i], 0, 0);
            } else {
                // Cout % 4 != 0 (8 -> 13): the last quad of a pixel ends in the next pixel -- its valid channels go out one by one
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (j < nvalid) __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v[j]), ro, eoff[i] + 4 * j, 0, 0);
            }
        }
    }
}

bool narrow16_ws_ok(const ConvParams& p) {
    static const bool off = exp_env("DL4DS_NO_NARROW16_WS") != nullptr;
    auto same_layout = [](const TView& u, const TView& v) { return u.ld == v.ld && u.d2s == v.d2s && u.W == v.W && u.cp == v.cp; };
    static const bool no_ragged = exp_env("DL4DS_NO_NARROW16_WS_RAGGED") != nullptr;      // (A/B)
    if (off || p.pool || p.in.sc || p.in.d2s > 1) return false;
    // <= 8 input channels with 9..16 outputs (8 -> 13: the dgrad of ConvBlock_att's first layer in the recurrent nets) come here too
    // since the epilogue forms are compiled in: conv_narrow_kernel<8> issues half the MFMAs but 0.53 ms at 128 x 256^2 against
    // 0.35 here (first half of round 3, run-time epilogue: 0.54 there, 0.59 here).  DL4DS_NARROW16_NO_SMALL_CIN=1 for A/B.
    static const bool no_small_cin = exp_env("DL4DS_NARROW16_NO_SMALL_CIN") != nullptr;
    if (p.Cin <= 8 && no_small_cin) return false;
    if ((!p.in.vec || (p.Cin & 3)) && no_ragged) return false;
    if ((((uintptr_t)p.in.p) & 3) != 0) return false;
    // outputs (and the epilogue's operands, which share the output's layout) in a channel slice of a wider buffer -- pixel pitch
    // not a multiple of 16 bytes -- are fine: buffer_store / buffer_load_dwordx4 only need dword alignment
    auto quad_ok = [&](const TView& v) { return v.vec || (!no_ragged && v.d2s <= 1 && ((((uintptr_t)v.p) & 3) == 0)); };
    if (((p.Cout & 3) && (no_ragged || p.out.d2s > 1)) || !quad_ok(p.out) || (p.add.p && !quad_ok(p.add)) || (p.mask.p && !quad_ok(p.mask))) return false;
    if ((((uintptr_t)p.bias) & 15) != 0) return false;
    if (p.add.p && !same_layout(p.add, p.out)) return false;
    if (p.mask.p && !same_layout(p.mask, p.out)) return false;
    const size_t r = p.out.d2s > 1 ? p.out.d2s : 1;
    if ((size_t)20 * p.out.W * r * r * p.out.ld * 4 >= (1ull << 31) || (size_t)20 * p.W * p.in.ld * 4 >= (1ull << 31)) return false;
    return true;
}

template <int NR>
bool launch_narrow16_ws(hipStream_t s, ConvParams& p, int N) {
    if (!narrow16_ws_ok(p)) {
        if (exp_env("DL4DS_NARROW_DEBUG"))
            fprintf(stderr, "narrow16_ws declined: N=%d H=%d W=%d Cin=%d (ld %d vec %d d2s %d sc %d) Cout=%d (ld %d vec %d d2s %d) add=%d(ld %d vec %d) mask=%d(ld %d vec %d) "
                            "acc=%d pool=%d bias&15=%d\n", N, p.H, p.W, p.Cin, p.in.ld, p.in.vec, p.in.d2s, p.in.sc != nullptr, p.Cout, p.out.ld, p.out.vec,
                    p.out.d2s, p.add.p != nullptr, p.add.ld, p.add.vec, p.mask.p != nullptr, p.mask.ld, p.mask.vec, p.accumulate, p.pool != nullptr,
                    (int)(((uintptr_t)p.bias) & 15));
        return false;
    }
    p.tiles_x = cdiv(p.W, 16);
    p.tiles_y = cdiv(p.H, 4 * NR);
    p.m_txy[0] = div_magic(p.tiles_x);
    p.m_txy[1] = div_magic(p.tiles_y);
    const long nt = (long)p.tiles_x * p.tiles_y * N;
    if (nt == 0 || nt >= (1l << 20)) return false;
    const int ntiles = (int)nt;
    static const bool generic_only = exp_env("DL4DS_NARROW16_WS_GENERIC") != nullptr;      // (A/B)
    const int epi = generic_only ? -1 : ((p.add.p ? PAIR_EPI_ADD : 0) | (p.relu ? PAIR_EPI_RELU : 0) | (p.mask.p ? PAIR_EPI_MASK : 0) |
                                         (p.accumulate ? PAIR_EPI_ACC : 0));
    const double px = (double)N * p.H * p.W;
    ProfScope ps(s, "conv_narrow16_ws<" + std::to_string(NR) + ">", 2.0 * px * 9 * p.Cin * p.Cout,
                 4.0 * (px * (p.Cin + p.Cout * (1 + (p.add.p ? 1 : 0) + (p.mask.p ? 1 : 0) + (p.accumulate ? 1 : 0))) + 9.0 * p.Cin * p.Cout));
    static const bool no_xcd = exp_env("DL4DS_NO_XCD_WALK") != nullptr;                     // (A/B)
    p.CK = no_xcd ? 0 : 1;
    auto grid_of = [&](int resident) { const int b = std::min(ntiles, resident); return b >= 8 ? (b & ~7) : b; };
    // <= 8 input channels: the five-group form (round 5), compiled for the epilogue forms such layers have -- plain, ReLU, ReLU mask,
    // accumulate and their pairs; the others take the run-time form of it.  DL4DS_NARROW16_NO_C8=1 for A/B.
    static const bool no_c8 = exp_env("DL4DS_NARROW16_NO_C8") != nullptr;
    if (p.Cin <= 8 && !no_c8) {
#define NARROW16_C8_FORM(E_) case E_: DL4DS_LAUNCH((conv_narrow16_ws_kernel<NR, E_, true>), \
        dim3(grid_of(resident_blocks<conv_narrow16_ws_kernel<NR, E_, true>>(512))), dim3(512), 0, s, p); break;
        switch (epi) {
            NARROW16_C8_FORM(0) NARROW16_C8_FORM(2) NARROW16_C8_FORM(4) NARROW16_C8_FORM(6) NARROW16_C8_FORM(8) NARROW16_C8_FORM(12)
            default: DL4DS_LAUNCH((conv_narrow16_ws_kernel<NR, -1, true>), dim3(grid_of(resident_blocks<conv_narrow16_ws_kernel<NR, -1, true>>(512))),
                                  dim3(512), 0, s, p); break;
        }
#undef NARROW16_C8_FORM
        HIP_CHECK(hipGetLastError());
        return true;
    }
#define NARROW16_FORM(E_) case E_: DL4DS_LAUNCH((conv_narrow16_ws_kernel<NR, E_>), \
        dim3(grid_of(resident_blocks<conv_narrow16_ws_kernel<NR, E_>>(512))), dim3(512), 0, s, p); break;
    switch (epi) {
        NARROW16_FORM(0) NARROW16_FORM(1) NARROW16_FORM(2) NARROW16_FORM(3) NARROW16_FORM(4) NARROW16_FORM(5) NARROW16_FORM(6) NARROW16_FORM(7)
        NARROW16_FORM(8) NARROW16_FORM(9) NARROW16_FORM(10) NARROW16_FORM(11) NARROW16_FORM(12) NARROW16_FORM(13) NARROW16_FORM(14) NARROW16_FORM(15)
        NARROW16_FORM(-1)
    }
#undef NARROW16_FORM
    HIP_CHECK(hipGetLastError());
    return true;
}

bool narrow_pair_ws_ok(const ConvParams& p) {
    static const bool off = exp_env("DL4DS_NO_PAIR_WS") != nullptr;
    auto same_layout = [](const TView& u, const TView& v) { return u.ld == v.ld && u.d2s == v.d2s && u.W == v.W && u.cp == v.cp; };
    // inputs whose channel count / pixel pitch is not a multiple of four (the discriminator's 5-channel first layer): the loaders'
    // 16-byte buffer loads only need dword alignment, what a quad picks up beyond Cin meets zero filter entries, and the
    // descriptor's exact size makes the buffer unit zero-fill at the very end of the view
    static const bool no_ragged = exp_env("DL4DS_NO_PAIR_WS_RAGGED") != nullptr;
    if (exp_env("DL4DS_NARROW_DEBUG"))
        fprintf(stderr, "pair_ws?: N=%d H=%d W=%d Cin=%d (ld %d vec %d d2s %d sc %d) Cout=%d (ld %d vec %d d2s %d) add=%d(ld %d d2s %d) mask=%d(ld %d d2s %d) acc=%d pool=%d\n",
                p.in.N, p.H, p.W, p.Cin, p.in.ld, p.in.vec, p.in.d2s, p.in.sc != nullptr, p.Cout, p.out.ld, p.out.vec, p.out.d2s, p.add.p != nullptr,
                p.add.ld, p.add.d2s, p.mask.p != nullptr, p.mask.ld, p.mask.d2s, p.accumulate, p.pool != nullptr);
    if (off || p.in.d2s > 1) return false;
    if ((p.pool || p.in.sc) && exp_env("DL4DS_NO_PAIR_WS_FUSED")) return false;      // (A/B: attention pieces through the older kernel)
    if (p.in.sc && (!p.in.vec || (p.Cin & 3))) return false;
    if ((!p.in.vec || (p.Cin & 3)) && no_ragged) return false;
    // the residual operand and the ReLU mask are addressed through their own strides: any plain layout of the output's grid (a
    // dense tensor beside an output that lives inside a Concatenate's buffer, ...); depth_to_space operands like the output only
    auto operand_ok = [&](const TView& v) { return same_layout(v, p.out) || (v.d2s <= 1 && p.out.d2s <= 1 && v.vec); };
    if (p.add.p && !operand_ok(p.add)) return false;
    if (p.mask.p && !operand_ok(p.mask)) return false;
    auto span_ok = [](const TView& v) { const size_t r = v.d2s > 1 ? v.d2s : 1; return (size_t)20 * v.W * r * r * v.ld * 4 < (1ull << 31); };
    if (!span_ok(p.out) || (p.add.p && !span_ok(p.add)) || (p.mask.p && !span_ok(p.mask)) || (size_t)20 * p.W * p.in.ld * 4 >= (1ull << 31)) return false;
    return true;
}

template <int NR>
void launch_narrow_pair(hipStream_t s, ConvParams& p, int N) {
    p.tiles_x = cdiv(p.W, 32);
    p.tiles_y = cdiv(p.H, 4 * NR);
    p.m_txy[0] = div_magic(p.tiles_x);
    p.m_txy[1] = div_magic(p.tiles_y);
    const int ntiles = p.tiles_x * p.tiles_y * N;
    if (ntiles == 0) return;
    const double px = (double)N * p.H * p.W;
    if (narrow_pair_ws_ok(p)) {
        // the epilogue forms the models use are compiled in (see the kernel's comment); anything else takes the run-time form
        static const bool generic_only = exp_env("DL4DS_PAIR_WS_GENERIC") != nullptr;      // (A/B)
        const int epi = (p.add.p ? PAIR_EPI_ADD : 0) | (p.relu ? PAIR_EPI_RELU : 0) | (p.mask.p ? PAIR_EPI_MASK : 0) |
                        (p.accumulate ? PAIR_EPI_ACC : 0) | (p.pool ? PAIR_EPI_POOL : 0) | (p.in.sc ? PAIR_EPI_AFF : 0);
        int blocks = std::min(ntiles, resident_blocks<conv_narrow_pair_ws_kernel<NR, -1>>(512));
        static const bool no_xcd = exp_env("DL4DS_NO_XCD_WALK") != nullptr;                 // (A/B)
        if (blocks >= 8) blocks &= ~7;
        p.CK = no_xcd ? 0 : 1;
        ProfScope ps(s, "conv_narrow_pair_ws<" + std::to_string(NR) + ">", 2.0 * px * 9 * p.Cin * p.Cout,
                     4.0 * (px * (p.Cin + p.Cout * (1 + (p.add.p ? 1 : 0) + (p.mask.p ? 1 : 0) + (p.accumulate ? 1 : 0))) + 9.0 * p.Cin * p.Cout));
        bool split_done = false;
#ifdef DL4DS_EXPERIMENTS
        // DL4DS_PAIR_SPLIT=3|6: the K loop as split-bf16 MFMAs (measurement only, see the kernel's comment); plain / ReLU / mask forms
        static const int split = exp_env("DL4DS_PAIR_SPLIT") ? atoi(exp_env("DL4DS_PAIR_SPLIT")) : 0;
        if ((split == 3 || split == 6) && !p.pool && !p.in.sc && (epi == 0 || epi == 2 || epi == 4)) {
#define PAIR_SPLIT_FORM(E_, S_) DL4DS_LAUNCH((conv_narrow_pair_ws_kernel<NR, E_, false, S_>), dim3(blocks), dim3(512), 0, s, p)
            if (split == 3) { if (epi == 0) PAIR_SPLIT_FORM(0, 3); else if (epi == 2) PAIR_SPLIT_FORM(2, 3); else PAIR_SPLIT_FORM(4, 3); }
            else { if (epi == 0) PAIR_SPLIT_FORM(0, 6); else if (epi == 2) PAIR_SPLIT_FORM(2, 6); else PAIR_SPLIT_FORM(4, 6); }
#undef PAIR_SPLIT_FORM
            split_done = true;
        }
#endif
#define PAIR_WS_FORM(E_) case E_: DL4DS_LAUNCH((conv_narrow_pair_ws_kernel<NR, E_>), dim3(blocks), dim3(512), 0, s, p); break;
        if (!split_done) switch (generic_only ? -1 : epi) {
            PAIR_WS_FORM(0) PAIR_WS_FORM(1) PAIR_WS_FORM(2) PAIR_WS_FORM(3) PAIR_WS_FORM(4) PAIR_WS_FORM(5) PAIR_WS_FORM(6) PAIR_WS_FORM(7)
            PAIR_WS_FORM(8) PAIR_WS_FORM(9) PAIR_WS_FORM(10) PAIR_WS_FORM(11) PAIR_WS_FORM(12) PAIR_WS_FORM(13) PAIR_WS_FORM(14) PAIR_WS_FORM(15)
            PAIR_WS_FORM(PAIR_EPI_POOL)
            PAIR_WS_FORM(PAIR_EPI_POOL | PAIR_EPI_RELU)
            PAIR_WS_FORM(PAIR_EPI_AFF) PAIR_WS_FORM(PAIR_EPI_AFF | PAIR_EPI_MASK) PAIR_WS_FORM(PAIR_EPI_AFF | PAIR_EPI_ACC)
            PAIR_WS_FORM(PAIR_EPI_AFF | PAIR_EPI_MASK | PAIR_EPI_ACC) PAIR_WS_FORM(PAIR_EPI_AFF | PAIR_EPI_RELU)
            default: DL4DS_LAUNCH((conv_narrow_pair_ws_kernel<NR, -1>), dim3(blocks), dim3(512), 0, s, p); break;
        }
#undef PAIR_WS_FORM
        HIP_CHECK(hipGetLastError());
#ifdef PAIR_WS_TRACE
        {
            HIP_CHECK(hipStreamSynchronize(s));
            std::vector<unsigned long long> h((size_t)blocks * 8);
            HIP_CHECK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(pair_ws_trace), h.size() * 8));
            double a_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int b_ = 0; b_ < blocks; ++b_) for (int q_ = 0; q_ < 8; ++q_) a_[q_] += (double)h[(size_t)b_ * 8 + q_];
            fprintf(stderr, "pair_ws trace (%d workgroups, %.1f tiles each; us per workgroup): MFMA waves K loop %.1f | barrier %.1f | epilogue %.1f"
                            " || loaders LDS writes %.1f | issue loads %.1f | barrier %.1f\n", blocks, a_[3] / blocks, a_[0] / blocks / 100,
                    a_[1] / blocks / 100, a_[2] / blocks / 100, a_[4] / blocks / 100, a_[5] / blocks / 100, a_[6] / blocks / 100);
        }
#endif
        return;
    }
    const int blocks = std::min(ntiles, resident_blocks<conv_narrow_pair_kernel<NR>>(256));
    ProfScope ps(s, "conv_narrow_pair<" + std::to_string(NR) + ">", 2.0 * px * 9 * p.Cin * p.Cout,
                 4.0 * (px * (p.Cin + p.Cout * (1 + (p.add.p ? 1 : 0) + (p.mask.p ? 1 : 0) + (p.accumulate ? 1 : 0))) + 9.0 * p.Cin * p.Cout));
    DL4DS_LAUNCH((conv_narrow_pair_kernel<NR>), dim3(blocks), dim3(256), 0, s, p);
    HIP_CHECK(hipGetLastError());
}

// 9 .. 16 input channels, <= 8 outputs: the two-pixels-per-column kernel with 16-channel k-slots (round 5).  Two tile rows per wave so
// that two workgroups (2 x 27 KB of halo tiles each) share a CU; epilogue forms of the layers that come here compiled in.
bool launch_narrow_pair16(hipStream_t s, ConvParams& p, int N) {
    constexpr int NR = 2;
    static const bool off = exp_env("DL4DS_NO_PAIR16") != nullptr;                    // (A/B)
    if (off || p.pool || p.in.sc || p.Cin <= 8 || p.Cin > 16 || p.Cout > 8 || (p.Cout & 3) || !narrow_pair_ws_ok(p)) return false;
    if (!p.out.vec || (p.add.p && !p.add.vec) || (p.mask.p && !p.mask.vec) || ((((uintptr_t)p.bias) & 15) != 0)) return false;
    if ((((uintptr_t)p.in.p) & 3) != 0) return false;
    p.tiles_x = cdiv(p.W, 32);
    p.tiles_y = cdiv(p.H, 4 * NR);
    p.m_txy[0] = div_magic(p.tiles_x);
    p.m_txy[1] = div_magic(p.tiles_y);
    const long nt = (long)p.tiles_x * p.tiles_y * N;
    if (nt == 0 || nt >= (1l << 20)) return false;
    const int ntiles = (int)nt;
    const int epi = (p.add.p ? PAIR_EPI_ADD : 0) | (p.relu ? PAIR_EPI_RELU : 0) | (p.mask.p ? PAIR_EPI_MASK : 0) | (p.accumulate ? PAIR_EPI_ACC : 0);
    int blocks = std::min(ntiles, resident_blocks<conv_narrow_pair_ws_kernel<NR, -1, true>>(512));
    if (blocks >= 8) blocks &= ~7;
    p.CK = 1;
    const double px = (double)N * p.H * p.W;
    ProfScope ps(s, "conv_narrow_pair16_ws<" + std::to_string(NR) + ">", 2.0 * px * 9 * p.Cin * p.Cout,
                 4.0 * (px * (p.Cin + p.Cout * (1 + (p.add.p ? 1 : 0) + (p.mask.p ? 1 : 0) + (p.accumulate ? 1 : 0))) + 9.0 * p.Cin * p.Cout));
#define PAIR16_FORM(E_) case E_: DL4DS_LAUNCH((conv_narrow_pair_ws_kernel<NR, E_, true>), dim3(blocks), dim3(512), 0, s, p); break;
    switch (epi) {
        PAIR16_FORM(0) PAIR16_FORM(1) PAIR16_FORM(2) PAIR16_FORM(3) PAIR16_FORM(4) PAIR16_FORM(8) PAIR16_FORM(12)
        default: DL4DS_LAUNCH((conv_narrow_pair_ws_kernel<NR, -1, true>), dim3(blocks), dim3(512), 0, s, p); break;
    }
#undef PAIR16_FORM
    HIP_CHECK(hipGetLastError());
    return true;
}

// --------------------------------------------------------------------------------------------
// Weight gradient for Cin <= 8: dW[tap][ci][co] = sum_p x[p + tap][ci] * dz[p][co].
// GEMM view per MFMA: rows = cout (first operand, dz), columns = (tap-of-a-pair, cin) (second operand, x), K = 4
// consecutive pixels of a tile row.  With 8 cins two taps share the 16 columns, so the nine taps take five MFMAs per
// pixel quad -- pairs (0,1) (3,4) (6,7) are horizontally adjacent (one contiguous 16-float LDS segment per k-slot),
// pair (2,5) is vertically adjacent (second half one tile row further), tap 8 is alone -- instead of the nine the
// general kernel issues (one 16-cin tile per tap, half of it padding).  All LDS reads use immediate offsets from
// two per-lane bases; the dz fragment of a quad is read once and reused by the five MFMAs; the bias gradient is the
// running sum of those dz fragments.  Persistent blocks, one partial slab per block, fixed reduction order.
// Cout <= 8 (PZ == 8) leaves half of the MFMA's 16 ROWS empty as well; they take the dz of the pixel ONE ROW BELOW:
//   D[(s, co)][(j, ci)] += dz[p + s * (1,0)][co] * x[p + o_j][ci]   ==   dW[o_j - s * (1,0)][ci][co],
// four taps per MFMA.  Column pairs o = (1,0)|(1,1), (2,1)|(2,2), (2,0)|(1,2) give, with s = 0 / 1, the taps
// {(1,0),(1,1),(0,0),(0,1)}, {(2,1),(2,2),(1,1)*,(1,2)}, {(2,0),(1,2)*,(1,0)*,(0,2)} (* = duplicate, discarded): all nine taps
// in THREE MFMAs per pixel quad, all three with the same dz fragment.  The shifted half sees every image row but the
// first (tile row 32 = the next tile's first row is staged as well); tiles on the top edge take one extra row step
// (r = -1) with the unshifted half zeroed.
struct NarrowWgradParams {
    TView x, dz;
    float* partial;
    int Cin, Cout, H, W;
    int tiles_x, tiles_y, ntiles;
    unsigned m_tx, m_ty;
};

template <int PZ, bool XVEC>     // PZ: dz channels per pixel in LDS: 8 (Cout <= 8) or 16; XVEC: x is float4-loadable
__global__ void __launch_bounds__(256, PZ == 8 ? 4 : 2) conv_narrow_wgrad_kernel(const NarrowWgradParams a) {
    constexpr int TWH = NTW + 2, THH = NTH + 2, HPIX = TWH * THH;
    constexpr int XQ = HPIX * 2;                        // float4s in the x halo tile (8 channels per pixel)
#ifdef NARROW_WGRAD_NO_P3                               // (variant builds: the five-MFMA form for A/B runs)
    constexpr bool P3 = false;
#else
    constexpr bool P3 = (PZ == 8);                      // three-MFMA form: dz rows of two vertically adjacent pixels per MFMA
#endif
    constexpr int ZROWS = P3 ? NTH + 1 : NTH;           // (the shifted half needs the row below the tile)
    constexpr int ZQ4 = PZ / 4, ZQ = NTW * ZROWS * ZQ4; // float4s in the dz tile
    constexpr int XIT = (XQ + 255) / 256, ZIT = (ZQ + 255) / 256;
    constexpr int XF = HPIX * 8 + 32;                   // + slack: the unused half of the lone tap reads one pixel on
    constexpr int NG = P3 ? 3 : 5, NV = NG * 4 + 1;     // accumulator groups; values per lane in the final reduction
    __shared__ __attribute__((aligned(16))) float smem[XF + NTW * ZROWS * PZ];
    static_assert(XF + NTW * ZROWS * PZ >= NV * 256, "reduction buffer does not fit");
    float* xt = smem;
    float* zt = smem + XF;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    if (tid < 32) xt[HPIX * 8 + tid] = 0.f;

    f32x4 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const bool zlane = (PZ == 16) || l15 < 8;           // PZ == 8: lanes 8..15 hold the pixel one row below
    const float* zrd = P3 ? zt + ((wave * NROWS + (l15 >> 3)) * NTW + lq) * PZ + (l15 & 7)
                          : zt + ((wave * NROWS) * NTW + lq) * PZ + (zlane ? l15 : 0);
    const float* xa = xt + ((wave * NROWS) * TWH + lq) * 8 + l15;                        // halves one pixel apart
    // second half one row apart (five-MFMA form: taps (0,2)|(1,2)) / one row up and two pixels on (three-MFMA form: (2,0)|(1,2))
    const float* xb = xt + ((wave * NROWS) * TWH + lq) * 8 + (l15 & 7) + (l15 >> 3) * (P3 ? (2 - TWH) * 8 : TWH * 8);

    // float4-loadable x: ONE buffer load per element and nothing else per element.  The generic staging code spent ~35 VALU
    // instructions per element on index arithmetic -- ~300 per tile and thread next to 160 MFMAs per wave, on the same issue
    // port.  Offsets relative to the tile origin are per-thread constants; whatever lies outside the image (or the channel
    // count) gets an out-of-range offset, for which the buffer unit returns the zero padding; they are recomputed only when
    // the tile's border signature changes.
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    constexpr int OOB = (int)0xffffff00u, RSRC3 = 0x00020000;
    int xoff[XVEC ? XIT : 1], xhyx[XVEC ? XIT : 1], zoff[XVEC ? ZIT : 1], zyx[XVEC ? ZIT : 1];
    int xsig = 0, zsig = 0;
    if constexpr (XVEC) {
#pragma unroll
        for (int u = 0; u < XIT; ++u) {
            const int e = tid + u * 256;
            const int pix = e >> 1, c4 = e & 1;
            const int hy = pix / TWH, hx = pix - hy * TWH;
            const bool live = e < XQ && c4 * 4 < a.Cin;
            xhyx[u] = live ? ((hy << 8) | hx) : 0x7f7f;
            xoff[u] = live ? (int)((((size_t)hy * a.W + hx) * a.x.ld + c4 * 4) * 4) : OOB;
        }
#pragma unroll
        for (int u = 0; u < ZIT; ++u) {
            const int e = tid + u * 256;
            const int pix = e / ZQ4, c4 = e - pix * ZQ4;
            const int ry = pix / NTW, rx = pix - ry * NTW;
            const bool live = e < ZQ && c4 * 4 < a.Cout;
            zyx[u] = live ? ((ry << 8) | rx) : 0x7f7f;
            zoff[u] = live ? (int)((((size_t)ry * a.W + rx) * a.dz.ld + c4 * 4) * 4) : OOB;
        }
        xsig = (THH << 8) | TWH;                        // rows [0, THH) x columns [0, TWH) of the halo inside the image
        zsig = (ZROWS << 8) | NTW;
    }

    for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
        const int q0 = fast_div(t, a.m_tx);
        const int bx = t - q0 * a.tiles_x;
        const int n = fast_div(q0, a.m_ty);
        const int by = q0 - n * a.tiles_y;
        const int x0 = bx * NTW, y0 = by * NTH;
        if constexpr (XVEC) {
            const int ylo = max(0, 1 - y0), yhi = min(THH, a.H + 1 - y0), xlo = max(0, 1 - x0), xhi = min(TWH, a.W + 1 - x0);
            const int sx = (ylo << 24) | (xlo << 16) | (yhi << 8) | xhi;
            if (sx != xsig) {
                xsig = sx;
#pragma unroll
                for (int u = 0; u < XIT; ++u) {
                    const int hy = xhyx[u] >> 8, hx = xhyx[u] & 0xff, c4 = (tid + u * 256) & 1;
                    xoff[u] = (hy >= ylo && hy < yhi && hx >= xlo && hx < xhi) ? (int)((((size_t)hy * a.W + hx) * a.x.ld + c4 * 4) * 4) : OOB;
                }
            }
            const int ymax = min(ZROWS, a.H - y0), xmax = min(NTW, a.W - x0);
            const int sz = (ymax << 8) | xmax;
            if (sz != zsig) {
                zsig = sz;
#pragma unroll
                for (int u = 0; u < ZIT; ++u) {
                    const int ry = zyx[u] >> 8, rx = zyx[u] & 0xff, c4 = (tid + u * 256) % ZQ4;
                    zoff[u] = (ry < ymax && rx < xmax) ? (int)((((size_t)ry * a.W + rx) * a.dz.ld + c4 * 4) * 4) : OOB;
                }
            }
            const long xorg = (long)((size_t)n * a.x.nstride) + ((long)(y0 - 1) * a.W + (x0 - 1)) * (long)a.x.ld;
            const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(reinterpret_cast<const char*>(a.x.p)) + xorg * 4, 0, 0x7fffff00, RSRC3);
            const size_t zorg = (size_t)n * a.dz.nstride + ((size_t)y0 * a.W + x0) * a.dz.ld;
            const __amdgpu_buffer_rsrc_t rsz = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(reinterpret_cast<const char*>(a.dz.p)) + zorg * 4, 0, 0x7fffff00, RSRC3);
            i32x4_t xr[XIT], zr[ZIT];
#pragma unroll
            for (int u = 0; u < XIT; ++u) xr[u] = __builtin_amdgcn_raw_buffer_load_b128(rsx, xoff[u], 0, 0);
#pragma unroll
            for (int u = 0; u < ZIT; ++u) zr[u] = __builtin_amdgcn_raw_buffer_load_b128(rsz, zoff[u], 0, 0);
            if (a.dz.sc) {
                // dz carries a channel affine (ChannelAttention2D backward: dX = dY * scale + dmean): ZQ4 divides 256, so
                // the thread's channel quad is fixed; the image is fixed for the tile.  Only INSIDE the tile: padding stays 0
                float4 s4, h4;
                view_affine4(a.dz, n, (tid % ZQ4) * 4, s4, h4);
#pragma unroll
                for (int u = 0; u < ZIT; ++u) {
                    if (zoff[u] != OOB) {
                        const float4 v = affine4(make_float4(__int_as_float(zr[u][0]), __int_as_float(zr[u][1]), __int_as_float(zr[u][2]),
                                                             __int_as_float(zr[u][3])), s4, h4);
                        zr[u] = (i32x4_t){__float_as_int(v.x), __float_as_int(v.y), __float_as_int(v.z), __float_as_int(v.w)};
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < XIT; ++u) {
                const int e = tid + u * 256;
                if (e < XQ) *reinterpret_cast<i32x4_t*>(xt + (size_t)e * 4) = xr[u];
            }
#pragma unroll
            for (int u = 0; u < ZIT; ++u) {
                const int e = tid + u * 256;
                if (e < ZQ) *reinterpret_cast<i32x4_t*>(zt + (size_t)e * 4) = zr[u];
            }
        } else {
            float4 xr[XIT], zr[ZIT];
            unsigned xm[XIT], zm[ZIT];
            if constexpr (XVEC) {
#pragma unroll
                for (int u = 0; u < XIT; ++u) {
                    const int e = tid + u * 256;
                    const int pix = e >> 1, c4 = e & 1;
                    const int hy = pix / TWH, hx = pix - hy * TWH;
                    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                    const bool ok = e < XQ && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                    const bool cok = ok && c4 * 4 < a.Cin;
                    const size_t off = (size_t)n * a.x.nstride + ((size_t)(ok ? gy : 0) * a.W + (ok ? gx : 0)) * a.x.ld +
                                       (cok ? c4 * 4 : 0);
                    xr[u] = *reinterpret_cast<const float4*>(a.x.p + off);
                    xm[u] = valid4(c4 * 4, a.Cin, ok);
                }
#pragma unroll
                for (int u = 0; u < ZIT; ++u) {
                    const int e = tid + u * 256;
                    const int pix = e / ZQ4, c4 = e - pix * ZQ4;
                    const int ry = pix / NTW, rx = pix - ry * NTW;
                    const int gy = y0 + ry, gx = x0 + rx;
                    const bool ok = e < ZQ && gy < a.H && gx < a.W;
                    const bool cok = ok && c4 * 4 < a.Cout;
                    const size_t off = (size_t)n * a.dz.nstride + ((size_t)(ok ? gy : 0) * a.W + (ok ? gx : 0)) * a.dz.ld +
                                       (cok ? c4 * 4 : 0);
                    zr[u] = *reinterpret_cast<const float4*>(a.dz.p + off);
                    zm[u] = valid4(c4 * 4, a.Cout, ok);
                }
            } else {
#pragma unroll
                for (int u = 0; u < XIT; ++u) {
                    const int e = tid + u * 256;
                    const int pix = e >> 1, c4 = e & 1;
                    const int hy = pix / TWH, hx = pix - hy * TWH;
                    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                    const bool ok = e < XQ && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                    const float* px = a.x.p + (size_t)n * a.x.nstride + ((size_t)(ok ? gy : 0) * a.W + (ok ? gx : 0)) * a.x.ld;
                    const int cm = a.Cin - 1;
                    xr[u] = make_float4(px[min(c4 * 4, cm)], px[min(c4 * 4 + 1, cm)], px[min(c4 * 4 + 2, cm)], px[min(c4 * 4 + 3, cm)]);
                    xm[u] = valid4(c4 * 4, a.Cin, ok);
                }
#pragma unroll
                for (int u = 0; u < ZIT; ++u) {
                    const int e = tid + u * 256;
                    const int pix = e / ZQ4, c4 = e - pix * ZQ4;
                    const int ry = pix / NTW, rx = pix - ry * NTW;
                    const int gy = y0 + ry, gx = x0 + rx;
                    const bool ok = e < ZQ && gy < a.H && gx < a.W;
                    const bool cok = ok && c4 * 4 < a.Cout;
                    const size_t off = (size_t)n * a.dz.nstride + ((size_t)(ok ? gy : 0) * a.W + (ok ? gx : 0)) * a.dz.ld +
                                       (cok ? c4 * 4 : 0);
                    zr[u] = *reinterpret_cast<const float4*>(a.dz.p + off);
                    zm[u] = valid4(c4 * 4, a.Cout, ok);
                }
            }
            if (a.dz.sc) {
                // dz carries a channel affine (ChannelAttention2D backward: dX = dY * scale + dmean): ZQ4 divides 256, so
                // the thread's channel quad is fixed; the image is fixed for the tile
                float4 s4, h4;
                view_affine4(a.dz, n, (tid % ZQ4) * 4, s4, h4);
#pragma unroll
                for (int u = 0; u < ZIT; ++u) zr[u] = affine4(zr[u], s4, h4);
            }
#pragma unroll
            for (int u = 0; u < XIT; ++u) {
                const int e = tid + u * 256;
                if (e < XQ) *reinterpret_cast<float4*>(xt + (size_t)e * 4) = mask4(xr[u], xm[u]);
            }
#pragma unroll
            for (int u = 0; u < ZIT; ++u) {
                const int e = tid + u * 256;
                if (e < ZQ) *reinterpret_cast<float4*>(zt + (size_t)e * 4) = mask4(zr[u], zm[u]);
            }
        }
        __syncthreads();
        if constexpr (P3) {
            if (y0 == 0 && wave == 0) {
                // top edge: the shifted half has no tile above that covers image row 0 -- one extra step at r = -1
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float dzv = zrd[(-1 * NTW + q * 4) * PZ];                  // (lanes >= 8: tile row 0; lanes < 8: out of the tile)
                    dzv = zlane ? 0.f : dzv;
                    const float x0v = xa[((-1 + 1) * TWH + q * 4 + 0) * 8];
                    const float x1v = xa[((-1 + 2) * TWH + q * 4 + 1) * 8];
                    const float x2v = xb[((-1 + 2) * TWH + q * 4 + 0) * 8];
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(dzv, x0v, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(dzv, x1v, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(dzv, x2v, acc[2], 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < NROWS; ++r) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float dzv = zrd[(r * NTW + q * 4) * PZ];             // rows 0-7: dz[r], rows 8-15: dz[r + 1]
                    bsum += dzv;                                               // (only lanes < 8 are kept)
                    const float x0v = xa[((r + 1) * TWH + q * 4 + 0) * 8];     // (1,0) | (1,1)   -> shifted: (0,0) | (0,1)
                    const float x1v = xa[((r + 2) * TWH + q * 4 + 1) * 8];     // (2,1) | (2,2)   -> shifted: (1,1)* | (1,2)
                    const float x2v = xb[((r + 2) * TWH + q * 4 + 0) * 8];     // (2,0) | (1,2)*  -> shifted: (1,0)* | (0,2)
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(dzv, x0v, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(dzv, x1v, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(dzv, x2v, acc[2], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
        for (int r = 0; r < NROWS; ++r) {
            __builtin_amdgcn_sched_barrier(0);          // one row's fragments in flight at a time (VGPR bound)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float dzv = zrd[(r * NTW + q * 4) * PZ];
                if (PZ == 8) dzv = zlane ? dzv : 0.f;
                bsum += dzv;
                const float x0v = xa[((r + 0) * TWH + q * 4 + 0) * 8];     // taps (0,0) | (0,1)
                const float x1v = xa[((r + 1) * TWH + q * 4 + 0) * 8];     // taps (1,0) | (1,1)
                const float x2v = xa[((r + 2) * TWH + q * 4 + 0) * 8];     // taps (2,0) | (2,1)
                const float x3v = xb[((r + 0) * TWH + q * 4 + 2) * 8];     // taps (0,2) | (1,2)
                const float x4v = xa[((r + 2) * TWH + q * 4 + 2) * 8];     // tap  (2,2) | unused
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(dzv, x0v, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(dzv, x1v, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(dzv, x2v, acc[2], 0, 0, 0);
                acc[NG - 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(dzv, x3v, acc[NG - 2], 0, 0, 0);
                acc[NG - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(dzv, x4v, acc[NG - 1], 0, 0, 0);
            }
        }
        }
        __syncthreads();
    }

    // ---- block reduction (waves in fixed order) and slab write by wave 0
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) smem[(g * 4 + rg) * 256 + tid] = acc[g][rg];
    smem[(NG * 4) * 256 + tid] = bsum;
    __syncthreads();
    if (wave != 0) return;
    const size_t nw = (size_t)9 * a.Cin * a.Cout;
    float* slab = a.partial + (size_t)blockIdx.x * (nw + a.Cout);
    const int half = l15 >> 3, ci = l15 & 7;
    if constexpr (P3) {
        // rows (s, co) = (lq >> 1, (lq & 1) * 4 + rg), columns (j, ci) = (half, ci); tap[g][s][j], -1 = duplicate
        const int s1 = lq >> 1;
        const int tap3[3] = {s1 ? (half ? 1 : 0) : (half ? 4 : 3), s1 ? (half ? 5 : -1) : (half ? 8 : 7),
                             s1 ? (half ? 2 : -1) : (half ? -1 : 6)};
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float* src = smem + (g * 4 + rg) * 256 + lane;
                const float v = (src[0] + src[64]) + (src[128] + src[192]);
                const int co = (lq & 1) * 4 + rg;
                if (tap3[g] >= 0 && ci < a.Cin && co < a.Cout) slab[((size_t)tap3[g] * a.Cin + ci) * a.Cout + co] = v;
            }
        }
    } else {
    const int tap_a[5] = {0, 3, 6, 2, 8}, tap_b[5] = {1, 4, 7, 5, -1};
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int tap = half ? tap_b[g] : tap_a[g];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const float* src = smem + (g * 4 + rg) * 256 + lane;
            const float v = (src[0] + src[64]) + (src[128] + src[192]);
            const int co = lq * 4 + rg;
            if (tap >= 0 && ci < a.Cin && co < a.Cout) slab[((size_t)tap * a.Cin + ci) * a.Cout + co] = v;
        }
    }
    }
    {
        const float* src = smem + (NG * 4) * 256 + lane;
        const float v = (src[0] + src[64]) + (src[128] + src[192]);
        if (lq == 0 && l15 < a.Cout) slab[nw + l15] = v;
    }
}

bool narrow_wgrad_eligible(const TView& x, const TView& dz, int KS) {
    if (x.sc) return false;                                   // only the dz operand takes a channel affine here
    if (KS != 3 || x.C > 8 || dz.C > 16 || x.d2s > 1 || dz.d2s > 1 || !dz.vec) return false;    // x may be unaligned / Cin % 4 != 0
    return (long)cdiv(x.W, NTW) * cdiv(x.H, NTH) * x.N < (1l << 20);                   // fast_div range
}

}  // namespace

int conv2d_narrow_wgrad_slabs(const TView& x, const TView& dz, int KS) {
    if (!narrow_wgrad_eligible(x, dz, KS)) return 0;
    const int ntiles = cdiv(x.W, NTW) * cdiv(x.H, NTH) * x.N;
    return std::max(1, std::min(ntiles, 1024));
}

int conv2d_narrow_wgrad(hipStream_t s, const TView& x, const TView& dz, int KS, float* partial, int max_slabs) {
    NarrowWgradParams p;
    p.x = x; p.dz = dz; p.partial = partial;
    p.Cin = x.C; p.Cout = dz.C; p.H = x.H; p.W = x.W;
    p.tiles_x = cdiv(p.W, NTW);
    p.tiles_y = cdiv(p.H, NTH);
    p.ntiles = p.tiles_x * p.tiles_y * x.N;
    p.m_tx = div_magic(p.tiles_x);
    p.m_ty = div_magic(p.tiles_y);
    const bool wide = dz.C > 8;
    // The 16-byte buffer loads of the staging code only need dword alignment: a plain view whose channel count or pixel pitch is
    // not a multiple of four (channel slices of a 13-channel tensor) takes the same path -- what the last quad of a pixel picks up
    // beyond its channels (the next pixel's first values) sits in MFMA columns ci >= Cin, which are never written out.
    static const bool no_unaligned = exp_env("DL4DS_NO_NARROW_WGRAD_UNALIGNED") != nullptr;
    const bool xv = x.vec || (x.d2s <= 1 && !no_unaligned);
    void (*kern)(const NarrowWgradParams) =
        wide ? (xv ? conv_narrow_wgrad_kernel<16, true> : conv_narrow_wgrad_kernel<16, false>)
             : (xv ? conv_narrow_wgrad_kernel<8, true> : conv_narrow_wgrad_kernel<8, false>);
    const int resident = wide ? (xv ? resident_blocks<conv_narrow_wgrad_kernel<16, true>>(256) : resident_blocks<conv_narrow_wgrad_kernel<16, false>>(256))
                              : (xv ? resident_blocks<conv_narrow_wgrad_kernel<8, true>>(256) : resident_blocks<conv_narrow_wgrad_kernel<8, false>>(256));
    const int blocks = std::max(1, std::min(std::min(max_slabs, p.ntiles), resident));
    const double px = (double)x.N * p.H * p.W;
    ProfScope ps(s, std::string("conv_narrow_wgrad<") + (wide ? "16>" : "8>"), 2.0 * px * 9 * p.Cin * p.Cout,
                 4.0 * px * (p.Cin + p.Cout));
    DL4DS_LAUNCH(kern, dim3(blocks), dim3(256), 0, s, p);
    HIP_CHECK(hipGetLastError());
    return blocks;
}

// <= 8 x <= 8 channels with float4-able outputs: two pixels per MFMA column (1.5x fewer MFMAs, all lanes store);
// the only variant that also takes inputs whose channel count is not a multiple of 4.  Also the only narrow variant that
// reads a view with a channel affine (float4-loadable inputs only) and emits the pooling partial sums.
bool conv2d_narrow_pair_ok(const TView& in, const TView& out, int KS, const ConvEpilogue& ep) {
    if (KS != 3 || in.d2s > 1 || exp_env("DL4DS_NO_PAIR") || exp_env("DL4DS_NO_NARROW")) return false;
    if ((long)cdiv(in.W, NTW) * cdiv(in.H, NTH) * in.N >= (1l << 20)) return false;    // fast_div range
    return in.C <= 8 && out.C <= 8 && (out.C & 3) == 0 && out.vec && (!ep.add.p || ep.add.vec) &&
           (!ep.mask.p || ep.mask.vec) && ((((uintptr_t)ep.bias) & 15) == 0) && (!in.sc || (in.vec && (in.C & 3) == 0));
}
int conv2d_narrow_pair_tiles_per_image(int H, int W) { return cdiv(W, 32) * cdiv(H, 4 * NARROW_PAIR_ROWS); }

bool conv2d_narrow_forward(hipStream_t s, const TView& in, const float* w, int KS, const TView& out,
                           const ConvEpilogue& ep) {
    if (KS != 3 || in.C > 16 || out.C > 16 || in.d2s > 1) return false;
    const bool pair_ok = conv2d_narrow_pair_ok(in, out, KS, ep);
    // 9..16 input channels that are not a multiple of four (the 13-channel ConvBlock behind TransitionLast 26 -> 13): the halo
    // staging's float4 loads only need dword alignment; channels beyond Cin are masked when the tile is written to LDS
    const bool unaligned_ok = in.C > 8 && in.d2s <= 1 && !in.sc && !ep.pool && !exp_env("DL4DS_NO_NARROW_UNALIGNED");
    if (!in.vec && !pair_ok && !unaligned_ok) return false;
    if ((long)cdiv(in.W, NTW) * cdiv(in.H, NTH) * in.N >= (1l << 20)) return false;    // fast_div range
    DL4DS_REQUIRE(pair_ok || (!in.sc && !ep.pool), "conv_narrow: channel-affine input / pooling partials need the pair kernel");
    ConvParams p;
    p.in = in; p.out = out; p.add = ep.add; p.mask = ep.mask;
    p.w = w; p.bias = ep.bias;
    p.Cin = in.C; p.Cout = out.C; p.H = in.H; p.W = in.W;
    p.relu = ep.relu; p.accumulate = ep.accumulate;
    p.wvec = 0; p.CK = 0; p.TPS = 1;
    p.pool = ep.pool;
    if (pair_ok) {
        launch_narrow_pair<NARROW_PAIR_ROWS>(s, p, in.N);
        return true;
    }
    if (in.C > 8 && out.C <= 8 && launch_narrow_pair16(s, p, in.N)) return true;
    {
        const bool done = launch_narrow16_ws<4>(s, p, in.N);
        if (!done) { if (in.C <= 8) launch_narrow<8>(s, p, in.N); else launch_narrow<16>(s, p, in.N); }
    }
    return true;
}
