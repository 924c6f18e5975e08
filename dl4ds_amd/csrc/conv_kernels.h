// Device-side pieces shared by the implicit-GEMM convolution kernels (conv.hip, conv_narrow.hip).
#pragma once
#include "common.h"

// issue U independent 16-byte loads before the first LDS store so the memory latency is paid once per
// batch, not once per element
template <int U, int NTHR, class LoadF, class ValidF, class StoreF>
__device__ __forceinline__ void staged_copy(int total, int tid, LoadF ld, ValidF valid, StoreF st) {
    for (int base = tid; base < total; base += NTHR * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * NTHR;
            const bool ok = idx < total;
            v[u] = ld(ok ? idx : 0, ok);           // always issued (clamped address); masked at store time
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * NTHR;
            if (idx < total) st(idx, mask4(v[u], valid(idx)));
        }
    }
}

struct ConvParams {
    TView in, out, add, mask;
    const float* w;
    const float* bias;
    int Cin, Cout, H, W;
    int CK, TPS;
    int tiles_x, tiles_y;
    int relu, accumulate;
    int wvec;
    unsigned m_txy[2];      // magic dividers for tiles_x, tiles_y
    // split-K (conv_igemm_kernel on grids with too few tiles to fill the chip): blockIdx.z owns `kchunks` consecutive
    // channel chunks and writes its partial result as image (z * nimg + n) of a plain slab buffer; 0 = off
    int kchunks = 0, nimg = 0;
    float* pool = nullptr;  // conv_narrow_pair only: [tile][8] channel sums of the values stored (ChannelAttention2D pooling)
};

// Epilogue shared by the forward/dgrad kernels.  The MFMA is issued as D = W^T-fragment x pixel-fragment, so with the
// 16x16x4 C/D layout (col = lane&15, row = (lane>>4)*4 + reg) every lane owns FOUR CONSECUTIVE output channels
// (cout = tile*16 + lq*4 + reg) of ONE pixel (column l15 of the m-tile's row): one 16-byte store per accumulator tile
// instead of four scattered dword stores -- the epilogue is store-issue bound, so this is worth ~1.5x on the whole
// kernel.  Offsets are separable (pixel base + channel offset), which also covers depth_to_space views.
template <int MT, int NT>
struct AccPack { f32x4 v[MT][NT]; };

template <int MT, int NT>
__device__ __forceinline__ void conv_epilogue(const ConvParams& a, const AccPack<MT, NT> accp, int n, int x0, int y0, int n0,
                                              int wm, int wn, int l15, int lq) {
    const bool vec_ok = a.out.vec && (!a.add.p || a.add.vec) && (!a.mask.p || a.mask.vec) && ((a.Cout & 3) == 0);
    const int gx = x0 + l15;
    // Two separately unrolled nests (vector / scalar) keep each body under clang's pragma-unroll size cap; a rolled
    // loop would index the accumulators dynamically and push all of them through scratch memory.
    if (vec_ok) {
        // Cout % 4 == 0 and co % 4 == 0: a lane's four channels are all valid or all out of range
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int co = n0 + (wn * NT + j) * 16 + lq * 4;
            const bool co_ok = co < a.Cout;
            const int cs = co_ok ? co : 0;
            const size_t q_out = view_chan_off(a.out, cs);
            const size_t q_add = a.add.p ? view_chan_off(a.add, cs) : 0;
            const size_t q_mask = a.mask.p ? view_chan_off(a.mask, cs) : 0;
            float4 bias_v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias && co_ok) bias_v = *reinterpret_cast<const float4*>(a.bias + cs);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int gy = y0 + wm * MT + i;
                if (co_ok && gy < a.H && gx < a.W) {
                    float4 v = make_float4(accp.v[i][j][0] + bias_v.x, accp.v[i][j][1] + bias_v.y,
                                           accp.v[i][j][2] + bias_v.z, accp.v[i][j][3] + bias_v.w);
                    if (a.add.p) {
                        const float4 r = *reinterpret_cast<const float4*>(a.add.p + view_pix_base(a.add, n, gy, gx) + q_add);
                        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                    }
                    if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if (a.mask.p) {
                        const float4 m = *reinterpret_cast<const float4*>(a.mask.p + view_pix_base(a.mask, n, gy, gx) + q_mask);
                        v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
                        v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
                    }
                    float4* dst = reinterpret_cast<float4*>(a.out.p + view_pix_base(a.out, n, gy, gx) + q_out);
                    if (a.accumulate) {
                        const float4 o = *dst;
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                    *dst = v;
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int co = n0 + (wn * NT + j) * 16 + lq * 4;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int gy = y0 + wm * MT + i;
                const bool pix_ok = gy < a.H && gx < a.W;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    if (pix_ok && co + rg < a.Cout) {
                        float t = accp.v[i][j][rg] + (a.bias ? a.bias[co + rg] : 0.f);
                        if (a.add.p) t += a.add.p[view_off(a.add, n, gy, gx, co + rg)];
                        if (a.relu) t = fmaxf(t, 0.f);
                        if (a.mask.p) t = (a.mask.p[view_off(a.mask, n, gy, gx, co + rg)] > 0.f) ? t : 0.f;
                        const size_t o = view_off(a.out, n, gy, gx, co + rg);
                        if (a.accumulate) t += a.out.p[o];
                        a.out.p[o] = t;
                    }
                }
            }
        }
    }
}

// split-K plumbing of conv.hip, also used by conv_gemm.hip: grow-only slab scratch per stream; out = epilogue(sum of S slabs)
float* conv_splitk_scratch(hipStream_t s, size_t floats);
void conv_splitk_combine(hipStream_t s, const float* slabs, int S, size_t slab_stride, const ConvParams& p, int N);
