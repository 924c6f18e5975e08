// dl4ds_amd -- direct (VALU) 3x3 convolutions for layers with a handful of channels.
//
// The tail of every dl4ds generator is ConvBlock_out (blocks.py:87-103 called from sp_postups.py:209-212): a
// C -> 1 and a 1 -> 1 3x3 convolution on the FULL-RESOLUTION grid, plus the C_in -> 8 stem on the LR grid.  With
// Cin*Cout <= 8 there is no GEMM to speak of: a 16x16x4 MFMA tile would be >= 15/16 padding, and the implicit-GEMM
// kernels spent 0.5-1.6 ms per launch on layers that move 0.13-0.6 GB (cfg2, batch 64).  These layers are pure HBM
// streaming, so they are written as stencils:
//   * one output pixel per thread, 8x32-pixel tile per 256-thread block, persistent over tiles;
//   * the halo'd input tile is staged in LDS as float4 planes ([c/4][row][col][4]) so that consecutive lanes read
//     consecutive 16-byte words (conflict-free ds_read_b128);
//   * the <= 72 weights live in SGPRs (staged zero-padded through LDS once per block, v_readfirstlane), so every
//     FMA takes its weight as a scalar operand;
//   * the same epilogue semantics as the MFMA kernels (bias, residual add, ReLU, ReLU-mask, accumulate).
// dgrad is the same kernel run on the flipped/transposed weights (conv2d_dgrad_weights), exactly as for the MFMA path.
// The weight gradient accumulates the KK*CI*CO (+CO bias) sums per thread over a persistent loop and writes one
// partial slab per block in the layout conv2d_wgrad's deterministic slab reduction already consumes.
#include "ops.h"
#include "prof.h"
#include "launch.h"
#include <algorithm>

namespace {

constexpr int DTX = 32, DTY = 8;        // output tile (one pixel per thread, 256 threads)

struct DirectParams {
    TView in, out, add, mask;
    const float* w;
    const float* bias;
    float* partial;                      // wgrad only: [gridDim.x][KK*Cin*Cout + Cout]
    int Cin, Cout, H, W;
    int tiles_x, tiles_y, ntiles;
    unsigned m_tx, m_ty;
    int relu, accumulate;
    int bpi = 0;                         // wgrad only: > 0 = per-image slabs, `bpi` blocks per image (slab = blockIdx.x = n*bpi + b)
    int xcd = 0;                         // second generation: XCD-aware tile walk
};

// Stage the (DTY+KS-1) x (DTX+KS-1) halo tile of `v` (CI padded channels) around (n, y0, x0) into LDS planes.
template <int KS, int CI>
__device__ __forceinline__ void stage_tile(const TView& v, int Cin, int H, int W, int n, int y0, int x0, int tid,
                                           float* __restrict__ tile) {
    constexpr int R = KS / 2, HWD = DTX + KS - 1, HHT = DTY + KS - 1, HPIX = HWD * HHT;
    constexpr int VEC = CI < 4 ? CI : 4, NPL = CI / VEC;
    constexpr int TOTAL = HPIX * NPL;
    constexpr int ITERS = (TOTAL + 255) / 256;
    if (VEC == 4 && v.vec) {
        float4 r[ITERS];
        unsigned m[ITERS];
#pragma unroll
        for (int u = 0; u < ITERS; ++u) {
            const int e = tid + u * 256;
            const int hp = e / NPL, pl = e - hp * NPL;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const int gy = y0 - R + hy, gx = x0 - R + hx;
            const bool ok = e < TOTAL && gy >= 0 && gy < H && gx >= 0 && gx < W;
            r[u] = view_load4_raw(v, n, gy, gx, pl * 4, ok);
            m[u] = valid4(pl * 4, Cin, ok);
        }
        if (v.sc) {
            // channel affine of the view (ChannelAttention2D's scale folded into this load, see TView): NPL divides 256,
            // so the thread's plane (channel quad) is fixed; the image is fixed for the tile
            float4 s4, h4;
            view_affine4(v, n, (tid % NPL) * 4, s4, h4);
#pragma unroll
            for (int u = 0; u < ITERS; ++u) r[u] = affine4(r[u], s4, h4);
        }
#pragma unroll
        for (int u = 0; u < ITERS; ++u) {
            const int e = tid + u * 256;
            if (e < TOTAL) {
                const int hp = e / NPL, pl = e - hp * NPL;
                *reinterpret_cast<float4*>(tile + ((size_t)pl * HPIX + hp) * 4) = mask4(r[u], m[u]);
            }
        }
    } else {
        // scalar path (Cin < 4, or a view without 16-byte alignment): one (pixel, channel) float per element
        constexpr int TS = HPIX * CI;
        constexpr int ITS = (TS + 255) / 256;
        float r[ITS];
        unsigned m[ITS];
#pragma unroll
        for (int u = 0; u < ITS; ++u) {
            const int e = tid + u * 256;
            const int hp = e / CI, c = e - hp * CI;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const int gy = y0 - R + hy, gx = x0 - R + hx;
            const bool ok = e < TS && c < Cin && gy >= 0 && gy < H && gx >= 0 && gx < W;
            r[u] = v.p[view_off(v, n, ok ? gy : 0, ok ? gx : 0, ok ? c : 0)];
            m[u] = ok ? 0xffffffffu : 0u;
        }
#pragma unroll
        for (int u = 0; u < ITS; ++u) {
            const int e = tid + u * 256;
            if (e < TS) {
                const int hp = e / CI, c = e - hp * CI;
                unsigned mm = m[u];
                asm volatile("" : "+v"(mm));
                tile[((size_t)(c / VEC) * HPIX + hp) * VEC + (c % VEC)] = __uint_as_float(__float_as_uint(r[u]) & mm);
            }
        }
    }
}

// The same staging with everything per-element hoisted out of the tile loop (float4-loadable plain views): offsets relative to
// the halo origin are per-thread constants, positions outside the image or the channel count carry an out-of-range offset for
// which the buffer unit returns the zero padding, and they are recomputed only when the tile's border signature changes.  One
// buffer load + one LDS store per element instead of ~35 VALU instructions of index arithmetic -- these kernels execute 72 FMAs
// per pixel, so that arithmetic was more than half of their instruction stream.
template <int KS, int CI>
struct TileStager {
    static constexpr int R = KS / 2, HWD = DTX + KS - 1, HHT = DTY + KS - 1, HPIX = HWD * HHT;
    static constexpr int VEC = CI < 4 ? CI : 4, NPL = CI / VEC;
    static constexpr int TOTAL = HPIX * NPL, ITERS = (TOTAL + 255) / 256;
    static constexpr int OOB = (int)0xffffff00u, RSRC3 = 0x00020000;
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    int soff[ITERS], hyx[ITERS];
    int sig;
    bool lean;
    __device__ __forceinline__ int rel(const TView& v, int hy, int hx, int pl) const {
        return (int)((((size_t)hy * v.W + hx) * v.ld + pl * 4) * 4);
    }
    __device__ __forceinline__ void init(const TView& v, int Cin, int tid) {
        lean = VEC == 4 && 256 % NPL == 0 && v.vec && v.d2s <= 1 && (size_t)(HHT + 1) * v.W * v.ld * 4 < (1ull << 31);
        sig = (HHT << 8) | HWD;
#pragma unroll
        for (int u = 0; u < ITERS; ++u) {
            const int e = tid + u * 256;
            const int hp = e / NPL, pl = e - hp * NPL;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const bool live = e < TOTAL && pl * 4 < Cin;
            hyx[u] = live ? ((hy << 8) | hx) : 0x7f7f;
            soff[u] = live ? rel(v, hy, hx, pl) : OOB;
        }
    }
    __device__ __forceinline__ void stage(const TView& v, int Cin, int H, int W, int n, int y0, int x0, int tid,
                                          float* __restrict__ tile) {
        if (!lean) { stage_tile<KS, CI>(v, Cin, H, W, n, y0, x0, tid, tile); return; }
        const int ylo = max(0, R - y0), yhi = min(HHT, H + R - y0), xlo = max(0, R - x0), xhi = min(HWD, W + R - x0);
        const int sg = (ylo << 24) | (xlo << 16) | (yhi << 8) | xhi;
        if (sg != sig) {
            sig = sg;
#pragma unroll
            for (int u = 0; u < ITERS; ++u) {
                const int hy = hyx[u] >> 8, hx = hyx[u] & 0xff, pl = (tid + u * 256) % NPL;
                soff[u] = (hy >= ylo && hy < yhi && hx >= xlo && hx < xhi) ? rel(v, hy, hx, pl) : OOB;
            }
        }
        const long org = (long)((size_t)n * v.nstride) + ((long)(y0 - R) * v.W + (x0 - R)) * (long)v.ld;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(v.p)) + org * 4, 0, 0x7fffff00, RSRC3);
        i32x4_t r[ITERS];
#pragma unroll
        for (int u = 0; u < ITERS; ++u) r[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, soff[u], 0, 0);
        if (v.sc) {
            // channel affine of the view (see stage_tile), inside the image only: the padding stays zero
            float4 s4, h4;
            view_affine4(v, n, (tid % NPL) * 4, s4, h4);
#pragma unroll
            for (int u = 0; u < ITERS; ++u) {
                if (soff[u] != OOB) {
                    const float4 t4 = affine4(make_float4(__int_as_float(r[u][0]), __int_as_float(r[u][1]), __int_as_float(r[u][2]),
                                                          __int_as_float(r[u][3])), s4, h4);
                    r[u] = (i32x4_t){__float_as_int(t4.x), __float_as_int(t4.y), __float_as_int(t4.z), __float_as_int(t4.w)};
                }
            }
        }
#pragma unroll
        for (int u = 0; u < ITERS; ++u) {
            const int e = tid + u * 256;
            if (e < TOTAL) {
                const int hp = e / NPL, pl = e - hp * NPL;
                *reinterpret_cast<i32x4_t*>(tile + ((size_t)pl * HPIX + hp) * 4) = r[u];
            }
        }
    }
};

template <int KS, int CI, int CO, int PLAIN>
__global__ void __launch_bounds__(256) conv_direct_kernel(const DirectParams a_) {
    // PLAIN: bias (+ ReLU) only, plain input view, output without depth_to_space.  The KK * CI * CO filter values live in scalar
    // registers; with the residual / mask / accumulate views and the input's channel affine live across the tile loop as well the
    // kernels spill scalar registers into vector lanes (v_readlane before the FMAs that use them); nulled at compile time the
    // 1 -> 1 layer has none left (0.117 -> 0.069 ms for two launches at 64 x 512^2).  The 72-value filters of 8 -> 1 / 1 -> 8 still
    // spill ~40; reading them with just-in-time scalar loads (constant address space, pointer made opaque per tile) and four
    // blocks per CU were measured: no change -- those two wait on HBM, not on issue slots.
    DirectParams a = a_;
    if constexpr (PLAIN) {
        a.add.p = nullptr; a.mask.p = nullptr; a.accumulate = 0;
        if constexpr (PLAIN == 1) { a.in.sc = nullptr; a.in.sh = nullptr; }      // (2: the input view keeps its channel affine)
        a.out.d2s = 1; a.in.d2s = 1;
    }
    constexpr int HWD = DTX + KS - 1, HHT = DTY + KS - 1, HPIX = HWD * HHT;
    constexpr int VEC = CI < 4 ? CI : 4, NPL = CI / VEC;
    constexpr int KK = KS * KS, NWT = KK * CI * CO;
    __shared__ __attribute__((aligned(16))) float tile[NPL * HPIX * VEC];
    __shared__ float wl[NWT];
    const int tid = threadIdx.x;
    for (int e = tid; e < NWT; e += 256) {
        const int co = e % CO, ci = (e / CO) % CI, tap = e / (CO * CI);
        wl[e] = (ci < a.Cin && co < a.Cout) ? a.w[((size_t)tap * a.Cin + ci) * a.Cout + co] : 0.f;
    }
    __syncthreads();
    float ws[NWT];
#pragma unroll
    for (int e = 0; e < NWT; ++e) ws[e] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wl[e])));
    float bs[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) bs[co] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;

    const int ty = tid / DTX, tx = tid % DTX;
    const bool vec_out = (CO % 4 == 0) && a.Cout == CO && a.out.vec && (!a.add.p || a.add.vec) && (!a.mask.p || a.mask.vec);
    TileStager<KS, CI> stager;
    stager.init(a.in, a.Cin, tid);
    for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
        const int q = fast_div(t, a.m_tx);
        const int bx = t - q * a.tiles_x;
        const int n = fast_div(q, a.m_ty);
        const int by = q - n * a.tiles_y;
        const int x0 = bx * DTX, y0 = by * DTY;
        stager.stage(a.in, a.Cin, a.H, a.W, n, y0, x0, tid, tile);
        __syncthreads();
        float acc[CO];
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] = 0.f;
#pragma unroll
        for (int dy = 0; dy < KS; ++dy) {
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    float v[VEC];
                    const float* src = tile + ((size_t)pl * HPIX + (ty + dy) * HWD + tx + dx) * VEC;
                    if (VEC == 4) {
                        const float4 t4 = *reinterpret_cast<const float4*>(src);
                        v[0] = t4.x; v[1] = t4.y; v[2 % VEC] = t4.z; v[3 % VEC] = t4.w;
                    } else if (VEC == 2) {
                        const float2 t2 = *reinterpret_cast<const float2*>(src);
                        v[0] = t2.x; v[1 % VEC] = t2.y;
                    } else {
                        v[0] = src[0];
                    }
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
#pragma unroll
                        for (int co = 0; co < CO; ++co)
                            acc[co] = fmaf(v[k], ws[((dy * KS + dx) * CI + pl * VEC + k) * CO + co], acc[co]);
                    }
                }
            }
        }
        const int gy = y0 + ty, gx = x0 + tx;
        if (gy < a.H && gx < a.W) {
            if (vec_out) {
#pragma unroll
                for (int c4 = 0; c4 < CO / 4; ++c4) {
                    float4 v = make_float4(acc[(c4 * 4) % CO] + bs[(c4 * 4) % CO], acc[(c4 * 4 + 1) % CO] + bs[(c4 * 4 + 1) % CO],
                                           acc[(c4 * 4 + 2) % CO] + bs[(c4 * 4 + 2) % CO], acc[(c4 * 4 + 3) % CO] + bs[(c4 * 4 + 3) % CO]);
                    if (a.add.p) {
                        const float4 r = *reinterpret_cast<const float4*>(a.add.p + view_off(a.add, n, gy, gx, c4 * 4));
                        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                    }
                    if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    if (a.mask.p) {
                        const float4 m = *reinterpret_cast<const float4*>(a.mask.p + view_off(a.mask, n, gy, gx, c4 * 4));
                        v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
                        v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
                    }
                    float4* dst = reinterpret_cast<float4*>(a.out.p + view_off(a.out, n, gy, gx, c4 * 4));
                    if (a.accumulate) {
                        const float4 o = *dst;
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                    *dst = v;
                }
            } else {
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    if (co < a.Cout) {
                        float tv = acc[co] + bs[co];
                        if (a.add.p) tv += a.add.p[view_off(a.add, n, gy, gx, co)];
                        if (a.relu) tv = fmaxf(tv, 0.f);
                        if (a.mask.p) tv = (a.mask.p[view_off(a.mask, n, gy, gx, co)] > 0.f) ? tv : 0.f;
                        const size_t o = view_off(a.out, n, gy, gx, co);
                        if (a.accumulate) tv += a.out.p[o];
                        a.out.p[o] = tv;
                    }
                }
            }
        }
        __syncthreads();
    }
}

// dW[tap][ci][co] = sum_p x[p + tap][ci] * dz[p][co], db[co] = sum_p dz[p][co]; a.in = x, a.out = dz.
template <int KS, int CI, int CO>
__global__ void __launch_bounds__(256) conv_direct_wgrad_kernel(const DirectParams a) {
    constexpr int HWD = DTX + KS - 1, HHT = DTY + KS - 1, HPIX = HWD * HHT;
    constexpr int VEC = CI < 4 ? CI : 4, NPL = CI / VEC;
    constexpr int KK = KS * KS, NWT = KK * CI * CO, NACC = NWT + CO;
    __shared__ __attribute__((aligned(16))) float tile[NPL * HPIX * VEC];
    __shared__ float red[4][NACC];
    const int tid = threadIdx.x;
    const int ty = tid / DTX, tx = tid % DTX;
    float acc[NACC];
#pragma unroll
    for (int e = 0; e < NACC; ++e) acc[e] = 0.f;
    const bool vec_dz = (CO % 4 == 0) && a.Cout == CO && a.out.vec;
    // per-image mode (a.bpi > 0): block blockIdx.x = n*bpi + b walks tiles b, b + bpi, ... of image n only, so that its slab
    // is the weight gradient of ONE sample (the attention backward needs those: conv2d_direct_wgrad_attention)
    const int tpi = a.tiles_x * a.tiles_y;
    const int img = a.bpi > 0 ? (int)blockIdx.x / a.bpi : 0;
    const int t_first = a.bpi > 0 ? img * tpi + ((int)blockIdx.x - img * a.bpi) : (int)blockIdx.x;
    const int t_end = a.bpi > 0 ? (img + 1) * tpi : a.ntiles;
    const int t_step = a.bpi > 0 ? a.bpi : (int)gridDim.x;
    TileStager<KS, CI> stager;
    stager.init(a.in, a.Cin, tid);
    for (int t = t_first; t < t_end; t += t_step) {
        const int q = fast_div(t, a.m_tx);
        const int bx = t - q * a.tiles_x;
        const int n = fast_div(q, a.m_ty);
        const int by = q - n * a.tiles_y;
        const int x0 = bx * DTX, y0 = by * DTY;
        const int gy = y0 + ty, gx = x0 + tx;
        const bool pok = gy < a.H && gx < a.W;
        // dz of this thread's pixel: issued before the staging so that both are in flight together
        float dz[CO];
        if (vec_dz) {
#pragma unroll
            for (int c4 = 0; c4 < CO / 4; ++c4) {
                const float4 r4 = mask4(view_load4_raw(a.out, n, gy, gx, c4 * 4, pok), valid4(c4 * 4, a.Cout, pok));
                dz[(c4 * 4) % CO] = r4.x; dz[(c4 * 4 + 1) % CO] = r4.y; dz[(c4 * 4 + 2) % CO] = r4.z; dz[(c4 * 4 + 3) % CO] = r4.w;
            }
        } else {
#pragma unroll
            for (int co = 0; co < CO; ++co) {
                const bool ok = pok && co < a.Cout;
                const float r = a.out.p[view_off(a.out, n, ok ? gy : 0, ok ? gx : 0, ok ? co : 0)];
                unsigned mm = ok ? 0xffffffffu : 0u;
                asm volatile("" : "+v"(mm));
                dz[co] = __uint_as_float(__float_as_uint(r) & mm);
            }
        }
        stager.stage(a.in, a.Cin, a.H, a.W, n, y0, x0, tid, tile);
        __syncthreads();
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[NWT + co] += dz[co];
#pragma unroll
        for (int dy = 0; dy < KS; ++dy) {
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    float v[VEC];
                    const float* src = tile + ((size_t)pl * HPIX + (ty + dy) * HWD + tx + dx) * VEC;
                    if (VEC == 4) {
                        const float4 t4 = *reinterpret_cast<const float4*>(src);
                        v[0] = t4.x; v[1] = t4.y; v[2 % VEC] = t4.z; v[3 % VEC] = t4.w;
                    } else if (VEC == 2) {
                        const float2 t2 = *reinterpret_cast<const float2*>(src);
                        v[0] = t2.x; v[1 % VEC] = t2.y;
                    } else {
                        v[0] = src[0];
                    }
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
#pragma unroll
                        for (int co = 0; co < CO; ++co) {
                            const int e = ((dy * KS + dx) * CI + pl * VEC + k) * CO + co;
                            acc[e] = fmaf(v[k], dz[co], acc[e]);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    // block reduction in a fixed order: lanes (butterfly) -> waves -> slab
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int e = 0; e < NACC; ++e) {
        const float s = wave_sum(acc[e]);
        if (lane == 0) red[wave][e] = s;
    }
    __syncthreads();
    const size_t nw = (size_t)KK * a.Cin * a.Cout;
    float* slab = a.partial + (size_t)blockIdx.x * (nw + a.Cout);
    for (int e = tid; e < NACC; e += 256) {
        const float s = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
        if (e < NWT) {
            const int co = e % CO, ci = (e / CO) % CI, tap = e / (CO * CI);
            if (ci < a.Cin && co < a.Cout) slab[((size_t)tap * a.Cin + ci) * a.Cout + co] = s;
        } else if (e - NWT < a.Cout) {
            slab[nw + (e - NWT)] = s;
        }
    }
}


// --------------------------------------------------------------------------------------------
// Second generation (round 3) of the two kernels above, for plain (not depth_to_space) inputs:
//   * 32 x 16 tile, every thread owns TWO vertically adjacent pixels: the halo shrinks from 1.33 to 1.19 x the tile and the four
//     halo rows of a pixel pair are read from LDS once for both (24 instead of 36 reads per pair with 8 input channels);
//   * the NEXT tile's buffer loads are issued right after this tile is in LDS and stay in flight during the 72+ FMAs per
//     pixel (register staging: 5 x 16 bytes per thread) -- the first generation waited for its loads between two barriers;
//   * the epilogue's operands (residual, ReLU mask, old value) are requested before the FMAs, too;
//   * XCD-aware walk: workgroup b runs on XCD b % 8; each XCD takes one contiguous eighth of the tiles, so vertically
//     neighbouring tiles meet in one L2 (the halo rows and the partial 128-byte lines at the tile's left and right edge
//     were fetched once per XCD before: 8 -> 1 at 64 x 512^2 read 1.3 x its input).
// Same arithmetic per output, same epilogue semantics, same slab layout; DL4DS_NO_DIRECT2=1 restores the first generation.
constexpr int D2Y = 16;

template <int CI>
struct Stager2 {
    static constexpr int HWD = DTX + 2, HHT = D2Y + 2, HPIX = HWD * HHT;
    static constexpr int VEC = CI < 4 ? CI : 4, NPL = CI / VEC;
    static constexpr int TOTAL = HPIX * NPL, ITERS = (TOTAL + 255) / 256;
    static constexpr int OOB = (int)0xffffff00u, RSRC3 = 0x00020000;
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    typedef int i32x2_t __attribute__((ext_vector_type(2)));
    int soff[ITERS], hyx[ITERS], sig;
    int r[ITERS][VEC];
    unsigned inside;                     // (channel affine: which elements of `r` lie inside the image)
    float4 aff_s, aff_h;                 // (channel affine of the requested tile's image, this thread's channel quad)
    __device__ __forceinline__ int rel(const TView& v, int hy, int hx, int pl) const {
        return (int)((((size_t)hy * v.W + hx) * v.ld + pl * 4) * 4);
    }
    __device__ __forceinline__ void init(const TView& v, int Cin, int tid) {
        sig = (HHT << 8) | HWD;
        inside = 0u;
#pragma unroll
        for (int u = 0; u < ITERS; ++u) {
            const int e = tid + u * 256;
            const int hp = e / NPL, pl = e - hp * NPL;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const bool live = e < TOTAL && pl * 4 < Cin;
            hyx[u] = live ? ((hy << 8) | hx) : 0x7f7f;
            soff[u] = live ? rel(v, hy, hx, pl) : OOB;
        }
    }
    // request the halo of the tile at (n, y0, x0): zero padding = out-of-range offsets
    __device__ __forceinline__ void issue(const TView& v, int H, int W, int n, int y0, int x0, int tid) {
        const int ylo = max(0, 1 - y0), yhi = min(HHT, H + 1 - y0), xlo = max(0, 1 - x0), xhi = min(HWD, W + 1 - x0);
        const int sg = (ylo << 24) | (xlo << 16) | (yhi << 8) | xhi;
        if (sg != sig) {
            sig = sg;
#pragma unroll
            for (int u = 0; u < ITERS; ++u) {
                const int hy = hyx[u] >> 8, hx = hyx[u] & 0xff, pl = (tid + u * 256) % NPL;
                soff[u] = (hy >= ylo && hy < yhi && hx >= xlo && hx < xhi) ? rel(v, hy, hx, pl) : OOB;
            }
        }
        const long org = (long)((size_t)n * v.nstride) + ((long)(y0 - 1) * v.W + (x0 - 1)) * (long)v.ld;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(v.p)) + org * 4, 0, 0x7fffff00, RSRC3);
        if (v.sc) {
            inside = 0u;
#pragma unroll
            for (int u = 0; u < ITERS; ++u) inside |= (soff[u] != OOB ? 1u : 0u) << u;
        }
#pragma unroll
        for (int u = 0; u < ITERS; ++u) {
            if constexpr (VEC == 4) {
                const i32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(rs, soff[u], 0, 0);
                r[u][0] = t[0]; r[u][1] = t[1]; r[u][2 % VEC] = t[2]; r[u][3 % VEC] = t[3];
            } else if constexpr (VEC == 2) {
                const i32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(rs, soff[u], 0, 0);
                r[u][0] = t[0]; r[u][1 % VEC] = t[1];
            } else {
                r[u][0] = __builtin_amdgcn_raw_buffer_load_b32(rs, soff[u], 0, 0);
            }
        }
        // the view's channel affine for this tile's image travels with the tile: requested here, behind the tile's loads (vector
        // memory returns in order), it is there when put() needs it.  Requested in put() every tile waited a full memory round
        // trip for it (8 -> 1 behind ChannelAttention2D: 0.164 ms in the step against 0.116 ms for the same layer without affine)
        if constexpr (VEC == 4) {
            if (v.sc) view_affine4(v, n, (tid % NPL) * 4, aff_s, aff_h);
        }
    }
    // the requested halo -> LDS planes [pl][halo pixel][VEC]; the channel affine of the view (image n_of_tile) on the way
    __device__ __forceinline__ void put(const TView& v, int n_of_tile, int tid, float* __restrict__ tile) {
        if constexpr (VEC == 4) {
            if (v.sc) {
                const float4 s4 = aff_s, h4 = aff_h;      // (requested by issue(), right behind the tile's loads)
#pragma unroll
                for (int u = 0; u < ITERS; ++u) {
                    if ((inside >> u) & 1u) {
                        const float4 t4 = affine4(make_float4(__int_as_float(r[u][0]), __int_as_float(r[u][1]), __int_as_float(r[u][2 % VEC]),
                                                              __int_as_float(r[u][3 % VEC])), s4, h4);
                        r[u][0] = __float_as_int(t4.x); r[u][1] = __float_as_int(t4.y);
                        r[u][2 % VEC] = __float_as_int(t4.z); r[u][3 % VEC] = __float_as_int(t4.w);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < ITERS; ++u) {
            const int e = tid + u * 256;
            if (e < TOTAL) {
                const int hp = e / NPL, pl = e - hp * NPL;
                float* d = tile + ((size_t)pl * HPIX + hp) * VEC;
                if constexpr (VEC == 4) *reinterpret_cast<i32x4_t*>(d) = (i32x4_t){r[u][0], r[u][1], r[u][2 % VEC], r[u][3 % VEC]};
                else if constexpr (VEC == 2) *reinterpret_cast<i32x2_t*>(d) = (i32x2_t){r[u][0], r[u][1 % VEC]};
                else d[0] = __int_as_float(r[u][0]);
            }
        }
    }
};

// which tiles a workgroup takes: first, end, stride (XCD-aware when the grid is a multiple of 8; per-image mode for the
// attention weight gradient: block n * bpi + b walks tiles b, b + bpi, ... of image n only)
struct Walk2 { int t0, t1, ts; };
__device__ __forceinline__ Walk2 walk2(const DirectParams& a) {
    Walk2 w;
    const int G = gridDim.x;
    if (a.bpi > 0) {
        const int tpi = a.tiles_x * a.tiles_y, img = (int)blockIdx.x / a.bpi;
        w.t0 = img * tpi + ((int)blockIdx.x - img * a.bpi); w.t1 = (img + 1) * tpi; w.ts = a.bpi;
    } else if (a.xcd && (G & 7) == 0) {
        const int per = (a.ntiles + 7) >> 3, xcd = blockIdx.x & 7;
        w.t0 = xcd * per + (int)(blockIdx.x >> 3); w.t1 = min(a.ntiles, (xcd + 1) * per); w.ts = G >> 3;
    } else {
        w.t0 = blockIdx.x; w.t1 = a.ntiles; w.ts = G;
    }
    return w;
}

template <int KS, int CI, int CO, int PLAIN>
__global__ void __launch_bounds__(256) conv_direct2_kernel(const DirectParams a_) {
    // PLAIN: bias (+ ReLU) only, plain input view, output without depth_to_space.  The KK * CI * CO filter values live in scalar
    // registers; with the residual / mask / accumulate views and the input's channel affine live across the tile loop as well the
    // kernels spill scalar registers into vector lanes (v_readlane before the FMAs that use them); nulled at compile time the
    // 1 -> 1 layer has none left (0.117 -> 0.069 ms for two launches at 64 x 512^2).  The 72-value filters of 8 -> 1 / 1 -> 8 still
    // spill ~40; reading them with just-in-time scalar loads (constant address space, pointer made opaque per tile) and four
    // blocks per CU were measured: no change -- those two wait on HBM, not on issue slots.
    DirectParams a = a_;
    if constexpr (PLAIN) {
        a.add.p = nullptr; a.mask.p = nullptr; a.accumulate = 0;
        if constexpr (PLAIN == 1) { a.in.sc = nullptr; a.in.sh = nullptr; }      // (2: the input view keeps its channel affine)
        a.out.d2s = 1; a.in.d2s = 1;
    }
    typedef Stager2<CI> ST;
    constexpr int HWD = ST::HWD, HPIX = ST::HPIX, VEC = ST::VEC, NPL = ST::NPL;
    constexpr int KK = KS * KS, NWT = KK * CI * CO;
    __shared__ __attribute__((aligned(16))) float tile[NPL * HPIX * VEC];
    __shared__ float wl[NWT];
    const int tid = threadIdx.x;
    for (int e = tid; e < NWT; e += 256) {
        const int co = e % CO, ci = (e / CO) % CI, tap = e / (CO * CI);
        wl[e] = (ci < a.Cin && co < a.Cout) ? a.w[((size_t)tap * a.Cin + ci) * a.Cout + co] : 0.f;
    }
    __syncthreads();
    float ws[NWT];
#pragma unroll
    for (int e = 0; e < NWT; ++e) ws[e] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wl[e])));
    float bs[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) bs[co] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;

    const int tx = tid & (DTX - 1), ty = tid / DTX;          // pixels (2 ty, tx) and (2 ty + 1, tx) of the tile
    const bool vec_out = (CO % 4 == 0) && a.Cout == CO && a.out.vec && (!a.add.p || a.add.vec) && (!a.mask.p || a.mask.vec);
    const Walk2 wk = walk2(a);
    auto decode = [&](int t, int& n, int& y0, int& x0) {
        const int q = fast_div(t, a.m_tx);
        const int bx = t - q * a.tiles_x;
        n = fast_div(q, a.m_ty);
        const int by = q - n * a.tiles_y;
        x0 = bx * DTX; y0 = by * D2Y;
    };
    ST st;
    st.init(a.in, a.Cin, tid);
    if (wk.t0 < wk.t1) { int n, y0, x0; decode(wk.t0, n, y0, x0); st.issue(a.in, a.H, a.W, n, y0, x0, tid); }
    for (int t = wk.t0; t < wk.t1; t += wk.ts) {
        int n, y0, x0;
        decode(t, n, y0, x0);
        st.put(a.in, n, tid, tile);
        __syncthreads();
        if (t + wk.ts < wk.t1) { int n2, y2, x2; decode(t + wk.ts, n2, y2, x2); st.issue(a.in, a.H, a.W, n2, y2, x2, tid); }
        const int gx = x0 + tx;
        // the epilogue's operands, requested before the arithmetic where that is cheap in registers (<= 2 outputs per pixel;
        // with 8 the 48 registers cost a wave per SIMD and the 1 -> 8 layer ran 15 % slower than the first generation)
        constexpr bool PRE = CO <= 2;
        float e_add[2][CO], e_mask[2][CO], e_old[2][CO];
        bool pok[2];
        auto fetch_operands = [&](int i) __attribute__((always_inline)) {
            const int gy = y0 + 2 * ty + i;
            if (vec_out) {
#pragma unroll
                for (int c4 = 0; c4 < CO / 4; ++c4) {
                    if (a.add.p) {
                        const float4 q = mask4(view_load4_raw(a.add, n, gy, gx, c4 * 4, pok[i]), pok[i] ? 0xfu : 0u);
                        e_add[i][(c4 * 4) % CO] = q.x; e_add[i][(c4 * 4 + 1) % CO] = q.y; e_add[i][(c4 * 4 + 2) % CO] = q.z; e_add[i][(c4 * 4 + 3) % CO] = q.w;
                    }
                    if (a.mask.p) {
                        const float4 q = mask4(view_load4_raw(a.mask, n, gy, gx, c4 * 4, pok[i]), pok[i] ? 0xfu : 0u);
                        e_mask[i][(c4 * 4) % CO] = q.x; e_mask[i][(c4 * 4 + 1) % CO] = q.y; e_mask[i][(c4 * 4 + 2) % CO] = q.z; e_mask[i][(c4 * 4 + 3) % CO] = q.w;
                    }
                    if (a.accumulate) {
                        const float4 q = mask4(view_load4_raw(a.out, n, gy, gx, c4 * 4, pok[i]), pok[i] ? 0xfu : 0u);
                        e_old[i][(c4 * 4) % CO] = q.x; e_old[i][(c4 * 4 + 1) % CO] = q.y; e_old[i][(c4 * 4 + 2) % CO] = q.z; e_old[i][(c4 * 4 + 3) % CO] = q.w;
                    }
                }
            } else {
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    const bool ok = pok[i] && co < a.Cout;
                    e_add[i][co] = (a.add.p && ok) ? a.add.p[view_off(a.add, n, gy, gx, co)] : 0.f;
                    e_mask[i][co] = (a.mask.p && ok) ? a.mask.p[view_off(a.mask, n, gy, gx, co)] : 0.f;
                    e_old[i][co] = (a.accumulate && ok) ? a.out.p[view_off(a.out, n, gy, gx, co)] : 0.f;
                }
            }
        };
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            pok[i] = (y0 + 2 * ty + i) < a.H && gx < a.W;
            if (PRE) fetch_operands(i);
        }
        float acc[2][CO];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int co = 0; co < CO; ++co) acc[i][co] = 0.f;
#pragma unroll
        for (int rr = 0; rr < KS + 1; ++rr) {
            const int row_off = (2 * ty + rr) * HWD + tx;
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    float v[VEC];
                    const float* src = tile + ((size_t)pl * HPIX + row_off + dx) * VEC;
                    if (VEC == 4) {
                        const float4 t4 = *reinterpret_cast<const float4*>(src);
                        v[0] = t4.x; v[1] = t4.y; v[2 % VEC] = t4.z; v[3 % VEC] = t4.w;
                    } else if (VEC == 2) {
                        const float2 t2 = *reinterpret_cast<const float2*>(src);
                        v[0] = t2.x; v[1 % VEC] = t2.y;
                    } else {
                        v[0] = src[0];
                    }
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
#pragma unroll
                        for (int co = 0; co < CO; ++co) {
                            if (rr < KS) acc[0][co] = fmaf(v[k], ws[((rr * KS + dx) * CI + pl * VEC + k) * CO + co], acc[0][co]);
                            if (rr >= 1) acc[1][co] = fmaf(v[k], ws[(((rr - 1) * KS + dx) * CI + pl * VEC + k) * CO + co], acc[1][co]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int gy = y0 + 2 * ty + i;
            if (!pok[i]) continue;
            if (!PRE) fetch_operands(i);
            float o[CO];
#pragma unroll
            for (int co = 0; co < CO; ++co) {
                float tv = acc[i][co] + bs[co];
                if (a.add.p) tv += e_add[i][co];
                if (a.relu) tv = fmaxf(tv, 0.f);
                if (a.mask.p) tv = e_mask[i][co] > 0.f ? tv : 0.f;
                if (a.accumulate) tv += e_old[i][co];
                o[co] = tv;
            }
            if (vec_out) {
#pragma unroll
                for (int c4 = 0; c4 < CO / 4; ++c4)
                    *reinterpret_cast<float4*>(a.out.p + view_off(a.out, n, gy, gx, c4 * 4)) =
                        make_float4(o[(c4 * 4) % CO], o[(c4 * 4 + 1) % CO], o[(c4 * 4 + 2) % CO], o[(c4 * 4 + 3) % CO]);
            } else {
#pragma unroll
                for (int co = 0; co < CO; ++co)
                    if (co < a.Cout) a.out.p[view_off(a.out, n, gy, gx, co)] = o[co];
            }
        }
        __syncthreads();
    }
}

template <int KS, int CI, int CO>
__global__ void __launch_bounds__(256) conv_direct2_wgrad_kernel(const DirectParams a) {
    typedef Stager2<CI> ST;
    constexpr int HWD = ST::HWD, HPIX = ST::HPIX, VEC = ST::VEC, NPL = ST::NPL;
    constexpr int KK = KS * KS, NWT = KK * CI * CO, NACC = NWT + CO;
    __shared__ __attribute__((aligned(16))) float tile[NPL * HPIX * VEC];
    __shared__ float red[4][NACC];
    const int tid = threadIdx.x;
    const int tx = tid & (DTX - 1), ty = tid / DTX;
    float acc[NACC];
#pragma unroll
    for (int e = 0; e < NACC; ++e) acc[e] = 0.f;
    const bool vec_dz = (CO % 4 == 0) && a.Cout == CO && a.out.vec;
    const Walk2 wk = walk2(a);
    auto decode = [&](int t, int& n, int& y0, int& x0) {
        const int q = fast_div(t, a.m_tx);
        const int bx = t - q * a.tiles_x;
        n = fast_div(q, a.m_ty);
        const int by = q - n * a.tiles_y;
        x0 = bx * DTX; y0 = by * D2Y;
    };
    // dz of this thread's two pixels of the tile at (n, y0, x0) (zero outside the image)
    auto load_dz = [&](int n, int y0, int x0, float (&dz)[2][CO]) __attribute__((always_inline)) {
        const int gx = x0 + tx;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int gy = y0 + 2 * ty + i;
            const bool pok = gy < a.H && gx < a.W;
            if (vec_dz) {
#pragma unroll
                for (int c4 = 0; c4 < CO / 4; ++c4) {
                    const float4 r4 = mask4(view_load4_raw(a.out, n, gy, gx, c4 * 4, pok), valid4(c4 * 4, a.Cout, pok));
                    dz[i][(c4 * 4) % CO] = r4.x; dz[i][(c4 * 4 + 1) % CO] = r4.y; dz[i][(c4 * 4 + 2) % CO] = r4.z; dz[i][(c4 * 4 + 3) % CO] = r4.w;
                }
            } else {
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    const bool ok = pok && co < a.Cout;
                    const float rv = a.out.p[view_off(a.out, n, ok ? gy : 0, ok ? gx : 0, ok ? co : 0)];
                    unsigned mm = ok ? 0xffffffffu : 0u;
                    asm volatile("" : "+v"(mm));
                    dz[i][co] = __uint_as_float(__float_as_uint(rv) & mm);
                }
            }
        }
    };
    ST st;
    st.init(a.in, a.Cin, tid);
    float dzn[2][CO];
    if (wk.t0 < wk.t1) {
        int n, y0, x0;
        decode(wk.t0, n, y0, x0);
        st.issue(a.in, a.H, a.W, n, y0, x0, tid);
        load_dz(n, y0, x0, dzn);
    }
    for (int t = wk.t0; t < wk.t1; t += wk.ts) {
        int n, y0, x0;
        decode(t, n, y0, x0);
        st.put(a.in, n, tid, tile);
        float dz[2][CO];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int co = 0; co < CO; ++co) dz[i][co] = dzn[i][co];
        __syncthreads();
        if (t + wk.ts < wk.t1) {
            int n2, y2, x2;
            decode(t + wk.ts, n2, y2, x2);
            st.issue(a.in, a.H, a.W, n2, y2, x2, tid);
            load_dz(n2, y2, x2, dzn);
        }
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[NWT + co] += dz[0][co] + dz[1][co];
#pragma unroll
        for (int rr = 0; rr < KS + 1; ++rr) {
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    float v[VEC];
                    const float* src = tile + ((size_t)pl * HPIX + (2 * ty + rr) * HWD + tx + dx) * VEC;
                    if (VEC == 4) {
                        const float4 t4 = *reinterpret_cast<const float4*>(src);
                        v[0] = t4.x; v[1] = t4.y; v[2 % VEC] = t4.z; v[3 % VEC] = t4.w;
                    } else if (VEC == 2) {
                        const float2 t2 = *reinterpret_cast<const float2*>(src);
                        v[0] = t2.x; v[1 % VEC] = t2.y;
                    } else {
                        v[0] = src[0];
                    }
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
#pragma unroll
                        for (int co = 0; co < CO; ++co) {
                            if (rr < KS) {
                                const int e = ((rr * KS + dx) * CI + pl * VEC + k) * CO + co;
                                acc[e] = fmaf(v[k], dz[0][co], acc[e]);
                            }
                            if (rr >= 1) {
                                const int e = (((rr - 1) * KS + dx) * CI + pl * VEC + k) * CO + co;
                                acc[e] = fmaf(v[k], dz[1][co], acc[e]);
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    // block reduction in a fixed order: lanes (butterfly) -> waves -> slab
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int e = 0; e < NACC; ++e) {
        const float sm = wave_sum(acc[e]);
        if (lane == 0) red[wave][e] = sm;
    }
    __syncthreads();
    const size_t nw = (size_t)KK * a.Cin * a.Cout;
    float* slab = a.partial + (size_t)blockIdx.x * (nw + a.Cout);
    for (int e = tid; e < NACC; e += 256) {
        const float sm = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
        if (e < NWT) {
            const int co = e % CO, ci = (e / CO) % CI, tap = e / (CO * CI);
            if (ci < a.Cin && co < a.Cout) slab[((size_t)tap * a.Cin + ci) * a.Cout + co] = sm;
        } else if (e - NWT < a.Cout) {
            slab[nw + (e - NWT)] = sm;
        }
    }
}

// the second generation takes plain inputs whose channels its loads cover exactly
bool direct2_ok(const DirectParams& p, int ci) {
    static const bool off = exp_env("DL4DS_NO_DIRECT2") != nullptr;
    if (off || p.in.d2s > 1) return false;
    if ((((uintptr_t)p.in.p) & 3) != 0) return false;
    if (ci >= 4 ? (!p.in.vec || (p.Cin & 3) != 0) : (p.Cin != ci)) return false;
    if (ci < 4 && p.in.sc) return false;
    if ((size_t)(D2Y + 3) * p.W * p.in.ld * 4 >= (1ull << 31)) return false;
    return (long)cdiv(p.W, DTX) * cdiv(p.H, D2Y) * p.in.N < (1l << 20);
}

inline int pad_pow2(int c) { int p = 1; while (p < c) p <<= 1; return p; }

bool eligible(const TView& in, const TView& out, int KS) {
    if (KS != 3) return false;
    if (in.d2s > 1 || out.d2s > 1) return false;
    if (in.C < 1 || out.C < 1) return false;
    if ((long)cdiv(in.W, DTX) * cdiv(in.H, DTY) * in.N >= (1l << 20)) return false;      // fast_div range
    return pad_pow2(in.C) * pad_pow2(out.C) <= 8;
}

void fill_tiles(DirectParams& p, int N) {
    p.tiles_x = cdiv(p.W, DTX);
    p.tiles_y = cdiv(p.H, DTY);
    p.ntiles = p.tiles_x * p.tiles_y * N;
    p.m_tx = div_magic(p.tiles_x);
    p.m_ty = div_magic(p.tiles_y);
}

template <int KS, int CI, int CO>
int launch_direct(hipStream_t s, const DirectParams& p0, bool wgrad, int max_blocks, bool exact_grid = false) {
    DirectParams p = p0;
    // forward with more than two outputs per pixel stays on the first generation (1 -> 8 at 64 x 512^2: 0.196 ms there, 0.229 here:
    // 142 registers against 45 and the layer is bound by its stores); every weight gradient measured is faster here
    static const bool no_plain = exp_env("DL4DS_DIRECT_NO_PLAIN") != nullptr;      // (A/B)
    const int plain = (!wgrad && !no_plain && !p.add.p && !p.mask.p && !p.accumulate && p.out.d2s <= 1 && p.in.d2s <= 1) ? (p.in.sc ? 2 : 1) : 0;
    static const bool no_wide2 = exp_env("DL4DS_DIRECT2_NO_WIDE") != nullptr;      // (A/B)
    // ... but in the PLAIN form (no epilogue operands to hold) the second generation wins there too: 1 -> 8 0.218 -> 0.171 ms
    const bool gen2 = direct2_ok(p, CI) && (wgrad || CO <= 2 || (plain == 1 && !no_wide2));
    if (gen2) {
        p.tiles_y = cdiv(p.H, D2Y);
        p.ntiles = p.tiles_x * p.tiles_y * p.in.N;
        p.m_ty = div_magic(p.tiles_y);
        static const bool no_xcd = exp_env("DL4DS_NO_XCD_WALK") != nullptr;
        p.xcd = no_xcd ? 0 : 1;
    }
    // persistent kernels: one residency round (blocks do equal work); exact_grid: per-image slabs need exactly that many blocks
    int blocks = max_blocks;
    if (!exact_grid) {
        const int resident = gen2 ? (wgrad ? resident_blocks<conv_direct2_wgrad_kernel<KS, CI, CO>>(256) : (plain == 2 ? resident_blocks<conv_direct2_kernel<KS, CI, CO, 2>>(256) : plain ? resident_blocks<conv_direct2_kernel<KS, CI, CO, 1>>(256) : resident_blocks<conv_direct2_kernel<KS, CI, CO, 0>>(256)))
                                  : (wgrad ? resident_blocks<conv_direct_wgrad_kernel<KS, CI, CO>>(256) : (plain == 2 ? resident_blocks<conv_direct_kernel<KS, CI, CO, 2>>(256) : plain ? resident_blocks<conv_direct_kernel<KS, CI, CO, 1>>(256) : resident_blocks<conv_direct_kernel<KS, CI, CO, 0>>(256)));
        blocks = std::max(1, std::min(std::min(max_blocks, resident), p.ntiles));
        if (gen2 && blocks >= 8) blocks &= ~7;
    }
    const double px = (double)p.in.N * p.H * p.W;
    const std::string tag = std::string(wgrad ? "conv_direct_wgrad<" : "conv_direct<") + std::to_string(KS) + "," +
                            std::to_string(CI) + "," + std::to_string(CO) + ">";
    // (forward / dgrad: the epilogue's operands -- residual, ReLU mask, old value -- are part of the layer's traffic)
    ProfScope ps(s, tag, 2.0 * px * KS * KS * p.Cin * p.Cout,
                 4.0 * px * (p.Cin + p.Cout * (wgrad ? 1 : 1 + (p.add.p ? 1 : 0) + (p.mask.p ? 1 : 0) + (p.accumulate ? 1 : 0))));
    if (gen2) {
        if (wgrad) DL4DS_LAUNCH((conv_direct2_wgrad_kernel<KS, CI, CO>), dim3(blocks), dim3(256), 0, s, p);
        else if (plain == 2) DL4DS_LAUNCH((conv_direct2_kernel<KS, CI, CO, 2>), dim3(blocks), dim3(256), 0, s, p);
        else if (plain) DL4DS_LAUNCH((conv_direct2_kernel<KS, CI, CO, 1>), dim3(blocks), dim3(256), 0, s, p);
        else DL4DS_LAUNCH((conv_direct2_kernel<KS, CI, CO, 0>), dim3(blocks), dim3(256), 0, s, p);
    } else {
        if (wgrad) DL4DS_LAUNCH((conv_direct_wgrad_kernel<KS, CI, CO>), dim3(blocks), dim3(256), 0, s, p);
        else if (plain == 2) DL4DS_LAUNCH((conv_direct_kernel<KS, CI, CO, 2>), dim3(blocks), dim3(256), 0, s, p);
        else if (plain) DL4DS_LAUNCH((conv_direct_kernel<KS, CI, CO, 1>), dim3(blocks), dim3(256), 0, s, p);
        else DL4DS_LAUNCH((conv_direct_kernel<KS, CI, CO, 0>), dim3(blocks), dim3(256), 0, s, p);
    }
    HIP_CHECK(hipGetLastError());
    return blocks;
}

// returns the number of blocks launched (= partial slabs written, for wgrad)
int dispatch_direct(hipStream_t s, const DirectParams& p, bool wgrad, int blocks, bool exact_grid = false) {
    const int ci = pad_pow2(p.Cin), co = pad_pow2(p.Cout);
    switch (ci * 16 + co) {
        case 1 * 16 + 1: return launch_direct<3, 1, 1>(s, p, wgrad, blocks, exact_grid);
        case 1 * 16 + 2: return launch_direct<3, 1, 2>(s, p, wgrad, blocks, exact_grid);
        case 1 * 16 + 4: return launch_direct<3, 1, 4>(s, p, wgrad, blocks, exact_grid);
        case 1 * 16 + 8: return launch_direct<3, 1, 8>(s, p, wgrad, blocks, exact_grid);
        case 2 * 16 + 1: return launch_direct<3, 2, 1>(s, p, wgrad, blocks, exact_grid);
        case 2 * 16 + 2: return launch_direct<3, 2, 2>(s, p, wgrad, blocks, exact_grid);
        case 2 * 16 + 4: return launch_direct<3, 2, 4>(s, p, wgrad, blocks, exact_grid);
        case 4 * 16 + 1: return launch_direct<3, 4, 1>(s, p, wgrad, blocks, exact_grid);
        case 4 * 16 + 2: return launch_direct<3, 4, 2>(s, p, wgrad, blocks, exact_grid);
        case 8 * 16 + 1: return launch_direct<3, 8, 1>(s, p, wgrad, blocks, exact_grid);
        default: throw Dl4dsError("conv_direct: unsupported channel combination");
    }
}

}  // namespace

// a view with a channel affine is only read by the float4 staging path (Cin % 4 == 0, 16-byte aligned)
static bool affine_ok(const TView& v) { return !v.sc || (v.vec && (v.C & 3) == 0 && v.C >= 4); }

bool conv2d_direct_eligible(const TView& in, const TView& out, int KS) { return eligible(in, out, KS) && affine_ok(in); }

bool conv2d_direct_forward(hipStream_t s, const TView& in, const float* w, int KS, const TView& out,
                           const ConvEpilogue& ep) {
    if (!eligible(in, out, KS)) return false;
    DL4DS_REQUIRE(affine_ok(in) && !ep.pool, "conv_direct: channel-affine input needs float4-loadable channels; no pooling partials");
    if ((ep.add.p && ep.add.d2s > 1) || (ep.mask.p && ep.mask.d2s > 1)) return false;
    DirectParams p;
    p.in = in; p.out = out; p.add = ep.add; p.mask = ep.mask;
    p.w = w; p.bias = ep.bias; p.partial = nullptr;
    p.Cin = in.C; p.Cout = out.C; p.H = in.H; p.W = in.W;
    p.relu = ep.relu; p.accumulate = ep.accumulate;
    fill_tiles(p, in.N);
    if (p.ntiles == 0) return true;
    dispatch_direct(s, p, false, p.ntiles);
    return true;
}

int conv2d_direct_wgrad_slabs(const TView& x, const TView& dz, int KS) {
    if (!eligible(x, dz, KS)) return 0;
    DL4DS_REQUIRE(affine_ok(x) && !dz.sc, "conv_direct wgrad: only a float4-loadable x operand may carry a channel affine");
    const int ntiles = cdiv(x.W, DTX) * cdiv(x.H, DTY) * x.N;
    return std::max(1, std::min(ntiles, 1024));
}

namespace {
// per image n: sum of its bpi slabs -> perimg[n][.]; ds[n][ci] = sum_{tap,co} w[tap][ci][co] * perimg[n][(tap,ci,co)]
__global__ void __launch_bounds__(256) att_wgrad_per_image_kernel(const float* __restrict__ partial, float* __restrict__ perimg,
                                                                  const float* __restrict__ w, float* __restrict__ ds, int bpi,
                                                                  int n_el, int nw, int KK, int Cin, int Cout) {
    extern __shared__ float sh[];                 // [n_el]
    const int n = blockIdx.x;
    for (int e = threadIdx.x; e < n_el; e += blockDim.x) {
        float s = 0.f;
        for (int b = 0; b < bpi; ++b) s += partial[((size_t)n * bpi + b) * n_el + e];      // fixed order
        sh[e] = s;
        perimg[(size_t)n * n_el + e] = s;
    }
    __syncthreads();
    for (int ci = threadIdx.x; ci < Cin; ci += blockDim.x) {
        float d = 0.f;
        for (int tap = 0; tap < KK; ++tap)
            for (int co = 0; co < Cout; ++co) {
                const int e = (tap * Cin + ci) * Cout + co;
                d += w[e] * sh[e];
            }
        ds[(size_t)n * Cin + ci] = d;
    }
}
// dW[e] (+)= sum_n scale[n][ci(e)] * perimg[n][e] ; db[co] (+)= sum_n perimg[n][nw + co].  Four threads per element (images
// n = j (mod 4)), combined as (p0 + p1) + (p2 + p3): fixed order, a quarter of the dependent chain (22 us for 64 images before)
__global__ void __launch_bounds__(256) att_wgrad_combine_kernel(const float* __restrict__ perimg, const float* __restrict__ scale,
                                                                float* __restrict__ dw, float* __restrict__ db, int N, int n_el,
                                                                int nw, int Cin, int Cout, int acc_w, int acc_b) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = t >> 2, j = t & 3;
    const bool live = e < n_el;
    float s = 0.f;
    if (live && e < nw) {
        const int ci = (e / Cout) % Cin;
        for (int n = j; n < N; n += 4) s += scale[(size_t)n * Cin + ci] * perimg[(size_t)n * n_el + e];
    } else if (live) {
        for (int n = j; n < N; n += 4) s += perimg[(size_t)n * n_el + e];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (!live || j != 0) return;
    if (e < nw) dw[e] = acc_w ? dw[e] + s : s;
    else if (db) db[e - nw] = acc_b ? db[e - nw] + s : s;
}
}  // namespace

// Weight gradient of a stencil convolution that reads its input through a ChannelAttention2D scale (y = conv(x * s_n), s_n
// per image and channel) TOGETHER with the attention's d(loss)/d(scale):
//   dW[tap,ci,co]  = sum_n s[n,ci] * R_n[tap,ci,co],   R_n = wgrad(x_n, dz_n)  (the RAW, un-scaled input)
//   ds[n,ci]       = sum_p x_n[p,ci] * dgrad(dz_n)[p,ci] = sum_{tap,co} W[tap,ci,co] * R_n[tap,ci,co]
// so the per-image raw weight gradients give both, and the attention backward needs no pass over (dy * x) at all.
// workspace: (N*bpi + N) * (KK*Cin*Cout + Cout) floats.  Returns false if the layer is not eligible (caller falls back).
bool conv2d_direct_wgrad_attention(hipStream_t s, const TView& x_raw, const TView& dz, int KS, const float* scale, const float* w,
                                   float* dw, int accumulate, float* db, int accumulate_db, float* ds, float* workspace,
                                   size_t workspace_bytes) {
    if (!eligible(x_raw, dz, KS) || x_raw.sc || dz.sc || !affine_ok(x_raw)) return false;
    const int N = x_raw.N;
    const int nw = KS * KS * x_raw.C * dz.C, n_el = nw + dz.C;
    const int bpi = std::max(1, std::min(1024 / std::max(N, 1), cdiv(x_raw.W, DTX) * cdiv(x_raw.H, DTY)));
    if (N > 1024 || (size_t)(N * bpi + N) * n_el * sizeof(float) > workspace_bytes || n_el * sizeof(float) > 48 * 1024) return false;
    DirectParams p;
    p.in = x_raw; p.out = dz; p.add = TView{nullptr, 0, 0, 0, 0, 0, 0, 0}; p.mask = p.add;
    p.w = nullptr; p.bias = nullptr; p.partial = workspace;
    p.Cin = x_raw.C; p.Cout = dz.C; p.H = x_raw.H; p.W = x_raw.W;
    p.relu = 0; p.accumulate = 0;
    p.bpi = bpi;
    fill_tiles(p, N);
    dispatch_direct(s, p, true, N * bpi, /*exact_grid=*/true);
    float* perimg = workspace + (size_t)N * bpi * n_el;
    DL4DS_LAUNCH(att_wgrad_per_image_kernel, dim3(N), dim3(256), n_el * sizeof(float), s, workspace, perimg, w, ds, bpi, n_el,
                       nw, KS * KS, x_raw.C, dz.C);
    HIP_CHECK(hipGetLastError());
    DL4DS_LAUNCH(att_wgrad_combine_kernel, dim3(cdiv(4 * n_el, 256)), dim3(256), 0, s, perimg, scale, dw, db, N, n_el, nw,
                       x_raw.C, dz.C, accumulate, accumulate_db);
    HIP_CHECK(hipGetLastError());
    return true;
}

int conv2d_direct_wgrad(hipStream_t s, const TView& x, const TView& dz, int KS, float* partial, int slabs) {
    DirectParams p;
    p.in = x; p.out = dz; p.add = TView{nullptr, 0, 0, 0, 0, 0, 0, 0}; p.mask = p.add;
    p.w = nullptr; p.bias = nullptr; p.partial = partial;
    p.Cin = x.C; p.Cout = dz.C; p.H = x.H; p.W = x.W;
    p.relu = 0; p.accumulate = 0;
    fill_tiles(p, x.N);
    return dispatch_direct(s, p, true, slabs);
}
