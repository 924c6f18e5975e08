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

namespace {

constexpr int NTW = 16, NTH = 32, NROWS = 8;      // tile width / height, rows per wave

template <int CI>
__global__ void __launch_bounds__(256, CI == 8 ? 4 : 2) conv_narrow_kernel(const ConvParams a) {
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
                r[u] = *reinterpret_cast<const float4*>(a.in.p + off);
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
                 4.0 * (px * (p.Cin + p.Cout) + 9.0 * p.Cin * p.Cout));
    hipLaunchKernelGGL((conv_narrow_kernel<CI>), dim3(blocks), dim3(256), 0, s, p);
    HIP_CHECK(hipGetLastError());
}

}  // namespace

bool conv2d_narrow_forward(hipStream_t s, const TView& in, const float* w, int KS, const TView& out,
                           const ConvEpilogue& ep) {
    if (KS != 3 || in.C > 16 || out.C > 16 || in.d2s > 1 || !in.vec) return false;
    if ((long)cdiv(in.W, NTW) * cdiv(in.H, NTH) * in.N >= (1l << 20)) return false;    // fast_div range
    ConvParams p;
    p.in = in; p.out = out; p.add = ep.add; p.mask = ep.mask;
    p.w = w; p.bias = ep.bias;
    p.Cin = in.C; p.Cout = out.C; p.H = in.H; p.W = in.W;
    p.relu = ep.relu; p.accumulate = ep.accumulate;
    p.wvec = 0; p.dbg = 0; p.CK = 0; p.TPS = 1;
    if (in.C <= 8) launch_narrow<8>(s, p, in.N);
    else launch_narrow<16>(s, p, in.N);
    return true;
}
