// NHWC fp32 convolution for gfx950 as implicit GEMM on the f32 matrix cores
// (v_mfma_f32_16x16x4_f32: exact f32 fma chains at the 157 TFLOP/s vector rate).
//
//   forward / dgrad : M = output pixels (16-wide row segments of a 2-D spatial tile so the KSxKS halo
//                     is reused from LDS), N = Cout, K = KS*KS*Cin.  Input halo tile and the filter
//                     slice of the current (channel-chunk, tap-group) are staged in LDS; each wave owns
//                     MT x NT accumulator tiles.  Epilogue fuses bias, residual add, ReLU, ReLU-mask,
//                     gradient accumulation and the depth_to_space store (through TView).
//   wgrad           : M = Cin, N = Cout, K = pixels.  Each block walks a strip of spatial tiles with
//                     all KS*KS taps accumulated in registers and writes partial slabs; the bias gradient
//                     (column sums of dz) rides along for free; a second kernel reduces the slabs
//                     deterministically.  Waves split Cout (wide layers) or the pixel/K axis (Cout <= 32).
//
// Replaces (third-party in the reference): tf.keras.layers.Conv2D forward and the Conv2DBackpropInput /
// Conv2DBackpropFilter TF autodiff would have produced -- call sites dl4ds/models/blocks.py:49-61,208,
// 249-259,299,414-416,479,582-583; sp_postups.py:134,156; discriminator.py:35-65.
#include "ops.h"
#include "prof.h"
#include "conv_kernels.h"
#include <algorithm>
#include <mutex>
#include <vector>
#include <cstdlib>

namespace {

constexpr int kLdsBudget = 80 * 1024;   // 2 workgroups per CU (160 KiB LDS)

// KSP = 2: a second set of WM x WN waves takes every other tap of the staged filter slice and the two partial accumulator
// sets are added through LDS before the epilogue -- two waves per SIMD for the launches that cannot fill the chip with
// blocks (ConvLSTM2D's per-time-step recurrent convolutions: 128 blocks of 1600 MFMAs each, LDS latency exposed).
template <int KS, int MT, int NT, int WM, int WN, int KSP = 1>
__global__ void __launch_bounds__(64 * WM * WN * KSP, KSP > 2 ? 1 : 2) conv_igemm_kernel(const ConvParams a) {
    constexpr int TW = 16;
    constexpr int NTHR = 64 * WM * WN * KSP;
    constexpr int BM = WM * MT * 16;
    constexpr int TH = BM / TW;
    constexpr int BN = WN * NT * 16;
    constexpr int BN4 = BN / 4;
    constexpr int PAD = KS / 2;
    constexpr int TWH = TW + KS - 1;
    constexpr int THH = TH + KS - 1;
    constexpr int HPIX = TWH * THH;
    constexpr int NP = (BN % 32 == 0) ? BN + 16 : BN;   // rows k and k+1 land on disjoint bank halves
    constexpr int KK = KS * KS;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int CK = a.CK;
    const int P = CK + 2;                                // P = 2*odd -> conflict-free A-fragment reads
    float* in_tile = smem;
    float* w_tile = smem + ((HPIX * P + 3) & ~3);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = (tid >> 6) % (WM * WN), ksl = (tid >> 6) / (WM * WN);      // ksl: which tap subset (KSP = 2)
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, lq = lane >> 4;

    int t = blockIdx.x;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;
    const int n0 = blockIdx.y * BN;

    int a_base[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a_base[i] = ((wm * MT + i) * TWH + l15) * P + lq;
    const int b_base = lq * NP + wn * NT * 16 + l15;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // split-K: this block's share of the channel chunks (the whole range when kchunks == 0)
    const int c_begin = a.kchunks ? (int)blockIdx.z * a.kchunks * CK : 0;
    const int c_end = a.kchunks ? min(a.Cin, c_begin + a.kchunks * CK) : a.Cin;
    for (int c0 = c_begin; c0 < c_end; c0 += CK) {
        const int ck = min(CK, a.Cin - c0);
        const int ck4 = (ck + 3) >> 2;
        const unsigned m4 = div_magic(ck4);
        __syncthreads();
        // ---- stage the input halo tile, channels [c0, c0+ck) (zero outside the image / beyond Cin)
        staged_copy<4, NTHR>(
            HPIX * ck4, tid,
            [&](int idx, bool ok) {
                const int pix = fast_div(idx, m4);
                const int q = idx - pix * ck4;
                const int r = pix / TWH;
                const int c = pix - r * TWH;
                const int gy = y0 + r - PAD, gx = x0 + c - PAD;
                return view_load4_raw(a.in, n, gy, gx, c0 + q * 4, ok && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W);
            },
            [&](int idx) {
                const int pix = fast_div(idx, m4);
                const int q = idx - pix * ck4;
                const int r = pix / TWH;
                const int c = pix - r * TWH;
                const int gy = y0 + r - PAD, gx = x0 + c - PAD;
                return valid4(c0 + q * 4, a.Cin, gy >= 0 && gy < a.H && gx >= 0 && gx < a.W);
            },
            [&](int idx, float4 v) {
                const int pix = fast_div(idx, m4);
                const int q = idx - pix * ck4;
                float2* d = reinterpret_cast<float2*>(in_tile + pix * P + q * 4);
                d[0] = make_float2(v.x, v.y);
                d[1] = make_float2(v.z, v.w);
            });
        for (int tg = 0; tg < KK; tg += a.TPS) {
            const int ntap = min(a.TPS, KK - tg);
            if (tg > 0) __syncthreads();
            // ---- stage the filter slice [ntap][ck4*4][BN] (zero rows/cols beyond Cin/Cout)
            const int rows = ck4 * 4;
            const unsigned mrows = div_magic(rows);
            {
                const float* wsrc = a.w + ((size_t)tg * a.Cin + c0) * a.Cout;
                staged_copy<5, NTHR>(
                    ntap * rows * BN4, tid,
                    [&](int idx, bool ok) {
                        const int row = idx / BN4;
                        const int q = idx - row * BN4;
                        const int tl = fast_div(row, mrows);
                        const int r = row - tl * rows;
                        const bool rok = ok && r < ck;
                        const float* src = wsrc + ((size_t)tl * a.Cin + (rok ? r : 0)) * a.Cout;
                        return row_load4_raw(src, n0 + q * 4, a.Cout, a.wvec != 0, rok);
                    },
                    [&](int idx) {
                        const int row = idx / BN4;
                        const int q = idx - row * BN4;
                        const int tl = fast_div(row, mrows);
                        const int r = row - tl * rows;
                        return valid4(n0 + q * 4, a.Cout, r < ck);
                    },
                    [&](int idx, float4 v) {
                        const int row = idx / BN4;
                        const int q = idx - row * BN4;
                        const int tl = fast_div(row, mrows);
                        const int r = row - tl * rows;
                        *reinterpret_cast<float4*>(w_tile + (tl * CK + r) * NP + q * 4) = v;
                    });
            }
            __syncthreads();
            // ---- MFMA over the staged taps
            for (int tl = ksl; tl < ntap; tl += KSP) {
                const int tap = tg + tl;
                const int ky = tap / KS, kx = tap - ky * KS;
                const float* ap = in_tile + (ky * TWH + kx) * P;
                const float* bp = w_tile + tl * CK * NP + b_base;
                // software pipeline: fragments of step kk+1 are in flight while step kk's MFMAs issue
                float av[MT], bv[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) av[i] = ap[a_base[i]];
#pragma unroll
                for (int j = 0; j < NT; ++j) bv[j] = bp[j * 16];
                for (int kk = 0; kk < ck4; ++kk) {
                    const int kn = (kk + 1 < ck4) ? kk + 1 : kk;
                    float an[MT], bn[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) an[i] = ap[a_base[i] + kn * 4];
#pragma unroll
                    for (int j = 0; j < NT; ++j) bn[j] = bp[kn * 4 * NP + j * 16];
                    __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ahead of the MFMA block
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j], av[i], acc[i][j], 0, 0, 0);   // D[cout][pixel]
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < MT; ++i) av[i] = an[i];
#pragma unroll
                    for (int j = 0; j < NT; ++j) bv[j] = bn[j];
                }
            }
        }
    }

    if constexpr (KSP > 1) {
        // add the second tap subset's accumulators (through the now idle tile memory), first subset runs the epilogue
        __syncthreads();
        constexpr int SET = WM * WN * 64 * MT * NT * 4;             // floats per accumulator set
        float* red = smem + (size_t)(wave * 64 + lane) * (MT * NT * 4);
        if (ksl >= 1) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(red + (size_t)(ksl - 1) * SET + (i * NT + j) * 4) = acc[i][j];
        }
        __syncthreads();
        if (ksl != 0) return;
#pragma unroll
        for (int k = 0; k < KSP - 1; ++k)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] += *reinterpret_cast<const f32x4*>(red + (size_t)k * SET + (i * NT + j) * 4);
    }
    {
        AccPack<MT, NT> accp;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) accp.v[i][j] = acc[i][j];
        conv_epilogue<MT, NT>(a, accp, a.kchunks ? (int)blockIdx.z * a.nimg + n : n, x0, y0, n0, wm, wn, l15, lq);
    }
}

// out = epilogue(sum of the S split-K slabs): bias, residual add, ReLU, ReLU-backward mask, accumulate -- through the
// real output view (plain or depth_to_space).  One thread per output element; the tensors here are tiny.
__global__ void splitk_combine_kernel(const float* __restrict__ slabs, int S, size_t slab_stride, const ConvParams a, int N) {
    const size_t total = (size_t)N * a.H * a.W * a.Cout;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % a.Cout);
        size_t r = e / a.Cout;
        const int x = (int)(r % a.W); r /= a.W;
        const int y = (int)(r % a.H);
        const int n = (int)(r / a.H);
        float v = a.bias ? a.bias[c] : 0.f;
        for (int z = 0; z < S; ++z) v += slabs[(size_t)z * slab_stride + e];
        if (a.add.p) v += a.add.p[view_off(a.add, n, y, x, c)];
        if (a.relu) v = fmaxf(v, 0.f);
        if (a.mask.p && !(a.mask.p[view_off(a.mask, n, y, x, c)] > 0.f)) v = 0.f;
        const size_t o = view_off(a.out, n, y, x, c);
        a.out.p[o] = a.accumulate ? a.out.p[o] + v : v;
    }
}

struct StreamScratch { hipStream_t stream; float* buf; size_t floats; };
float* splitk_scratch(hipStream_t s, size_t floats) {     // grow-only, one buffer per stream (launches on a stream are ordered)
    static std::mutex mu;
    static std::vector<StreamScratch> all;
    std::lock_guard<std::mutex> lk(mu);
    for (auto& e : all) {
        if (e.stream != s) continue;
        if (e.floats < floats) {
            HIP_CHECK(hipStreamSynchronize(s));
            HIP_CHECK(hipFree(e.buf));
            HIP_CHECK(hipMalloc((void**)&e.buf, floats * sizeof(float)));
            e.floats = floats;
        }
        return e.buf;
    }
    StreamScratch e{s, nullptr, std::max<size_t>(floats, (size_t)1 << 20)};
    HIP_CHECK(hipMalloc((void**)&e.buf, e.floats * sizeof(float)));
    all.push_back(e);
    return e.buf;
}

// --------------------------------------------------------------------------------------------
// Double-buffered ("db") variant for the MFMA-bound layers.  The K loop is flattened into stages
// s = (channel chunk, tap group); while stage s's MFMAs issue, the filter slice of stage s+1 (and, at a chunk
// boundary, the next chunk's input halo tile) is already in flight from L2/HBM into registers and is written to
// the alternate LDS buffers after the MFMA block -> one barrier per stage, and the memory system sees a smooth
// stream instead of every CU staging at the same instant (measured: in-phase staging bursts cost ~40 % of the
// kernel in the single-buffered version).  Single-chunk layers (e.g. 48->192) keep one input buffer.
//   IR / WR : float4 registers per thread reserved for the input / filter prefetch.
//   MC      : multi-chunk capable (keeps the input prefetch registers live across the stage loop).
template <int KS, int MT, int NT, int WM, int WN, int IR, int WR, bool MC>
__global__ void __launch_bounds__(64 * WM * WN, 2) conv_igemm_db_kernel(const ConvParams a) {
    constexpr int TW = 16;
    constexpr int NTHR = 64 * WM * WN;
    constexpr int BM = WM * MT * 16;
    constexpr int TH = BM / TW;
    constexpr int BN = WN * NT * 16;
    constexpr int BN4 = BN / 4;
    constexpr int PAD = KS / 2;
    constexpr int TWH = TW + KS - 1;
    constexpr int THH = TH + KS - 1;
    constexpr int HPIX = TWH * THH;
    constexpr int NP = (BN % 32 == 0) ? BN + 16 : BN;
    constexpr int KK = KS * KS;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int CK = a.CK, TPS = a.TPS;
    const int P = CK + 2;
    const int nchunks = MC ? (a.Cin + CK - 1) / CK : 1;
    const int ngroups = (KK + TPS - 1) / TPS;
    const int nstages = nchunks * ngroups;
    const int in_floats = (HPIX * P + 3) & ~3;
    const int w_floats = TPS * CK * NP;
    float* in_buf = smem;                                            // [1 or 2][in_floats]
    float* w_buf = smem + (nchunks > 1 ? 2 : 1) * in_floats;         // [2][w_floats]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, lq = lane >> 4;

    int t = blockIdx.x;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;
    const int n0 = blockIdx.y * BN;

    int a_base[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a_base[i] = ((wm * MT + i) * TWH + l15) * P + lq;
    const int b_base = lq * NP + wn * NT * 16 + l15;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 ireg[IR], wreg[WR];
    const int ckq = CK >> 2;                       // float4 groups per pixel of a full chunk
    const unsigned mckq = div_magic(ckq);
    const int in_total = HPIX * ckq;
    const int rows_w = CK;                         // filter rows staged per tap (zero beyond Cin)
    const unsigned mrows = div_magic(rows_w);

    auto load_in = [&](int chunk) {
        const int c0 = chunk * CK;
#pragma unroll
        for (int u = 0; u < IR; ++u) {
            const int idx0 = tid + u * NTHR;
            const bool ok = idx0 < in_total;
            const int idx = ok ? idx0 : 0;
            const int pix = fast_div(idx, mckq);
            const int q = idx - pix * ckq;
            const int r = pix / TWH;
            const int c = pix - r * TWH;
            const int gy = y0 + r - PAD, gx = x0 + c - PAD;
            ireg[u] = view_load4_raw(a.in, n, gy, gx, c0 + q * 4,
                                     ok && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && c0 + q * 4 < a.Cin);
        }
    };
    auto store_in = [&](int chunk) {
        float* dst = in_buf + (chunk & 1) * in_floats;
        const int c0 = chunk * CK;
#pragma unroll
        for (int u = 0; u < IR; ++u) {
            const int idx = tid + u * NTHR;
            if (idx < in_total) {
                const int pix = fast_div(idx, mckq);
                const int q = idx - pix * ckq;
                const int r = pix / TWH;
                const int c = pix - r * TWH;
                const int gy = y0 + r - PAD, gx = x0 + c - PAD;
                const float4 v = mask4(ireg[u], valid4(c0 + q * 4, a.Cin, gy >= 0 && gy < a.H && gx >= 0 && gx < a.W));
                float2* d = reinterpret_cast<float2*>(dst + pix * P + q * 4);
                d[0] = make_float2(v.x, v.y);
                d[1] = make_float2(v.z, v.w);
            }
        }
    };
    auto load_w = [&](int stage) {
        const int chunk = stage / ngroups;
        const int tg = (stage - chunk * ngroups) * TPS;
        const int ntap = min(TPS, KK - tg);
        const int c0 = chunk * CK;
        const float* wsrc = a.w + ((size_t)tg * a.Cin + c0) * a.Cout;
        const int total = ntap * rows_w * BN4;
#pragma unroll
        for (int u = 0; u < WR; ++u) {
            const int idx0 = tid + u * NTHR;
            const bool ok = idx0 < total;
            const int idx = ok ? idx0 : 0;
            const int row = idx / BN4;
            const int q = idx - row * BN4;
            const int tl = fast_div(row, mrows);
            const int r = row - tl * rows_w;
            const bool rok = ok && (c0 + r < a.Cin);
            const float* src = wsrc + ((size_t)tl * a.Cin + (rok ? r : 0)) * a.Cout;
            wreg[u] = row_load4_raw(src, n0 + q * 4, a.Cout, a.wvec != 0, rok);
        }
    };
    auto store_w = [&](int stage) {
        const int chunk = stage / ngroups;
        const int tg = (stage - chunk * ngroups) * TPS;
        const int ntap = min(TPS, KK - tg);
        const int total = ntap * rows_w * BN4;
        float* dst = w_buf + (stage & 1) * w_floats;
#pragma unroll
        for (int u = 0; u < WR; ++u) {
            const int idx = tid + u * NTHR;
            if (idx < total) {
                const int row = idx / BN4;
                const int q = idx - row * BN4;
                const int tl = fast_div(row, mrows);
                const int r = row - tl * rows_w;
                *reinterpret_cast<float4*>(dst + (tl * CK + r) * NP + q * 4) =
                    mask4(wreg[u], valid4(n0 + q * 4, a.Cout, chunk * CK + r < a.Cin));
            }
        }
    };

    load_in(0);
    load_w(0);
    store_in(0);
    store_w(0);
    __syncthreads();

    for (int s = 0; s < nstages; ++s) {
        const int chunk = s / ngroups;
        const int gi = s - chunk * ngroups;
        const int nxt = s + 1;
        const bool new_chunk = MC && (nxt < nstages) && (gi == ngroups - 1);
        if (nxt < nstages) {
            if (new_chunk) load_in(chunk + 1);
            load_w(nxt);
        }
        // ---- MFMA over this stage's taps
        const int tg = gi * TPS;
        const int ntap = min(TPS, KK - tg);
        const int ck_here = min(CK, a.Cin - chunk * CK);
        const int ck4 = (ck_here + 3) >> 2;
        const float* inb = in_buf + ((nchunks > 1) ? (chunk & 1) * in_floats : 0);
        const float* wb = w_buf + (s & 1) * w_floats + b_base;
        for (int tl = 0; tl < ntap; ++tl) {
            const int tap = tg + tl;
            const int ky = tap / KS, kx = tap - ky * KS;
            const float* ap = inb + (ky * TWH + kx) * P;
            const float* bp = wb + tl * CK * NP;
            float av[MT], bv[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) av[i] = ap[a_base[i]];
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = bp[j * 16];
            for (int kk = 0; kk < ck4; ++kk) {
                const int kn = (kk + 1 < ck4) ? kk + 1 : kk;
                float an[MT], bn[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) an[i] = ap[a_base[i] + kn * 4];
#pragma unroll
                for (int j = 0; j < NT; ++j) bn[j] = bp[kn * 4 * NP + j * 16];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j], av[i], acc[i][j], 0, 0, 0);   // D[cout][pixel]
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MT; ++i) av[i] = an[i];
#pragma unroll
                for (int j = 0; j < NT; ++j) bv[j] = bn[j];
            }
        }
        // ---- park the prefetched data in the alternate buffers (last read >= one barrier ago)
        if (nxt < nstages) {
            if (new_chunk) store_in(chunk + 1);
            store_w(nxt);
        }
        __syncthreads();
    }

    {
        AccPack<MT, NT> accp;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) accp.v[i][j] = acc[i][j];
        conv_epilogue<MT, NT>(a, accp, n, x0, y0, n0, wm, wn, l15, lq);
    }
}

constexpr int kLdsMax = 160 * 1024;

// Picks (CK, TPS) for the db kernel: largest stage that fits the prefetch registers and the LDS limit
// (80 KB -> two workgroups per CU; single-chunk layouts may take the whole 160 KB with 8 waves).
template <int KS, int MT, int NT, int WM, int WN, int IR, int WR, bool MC>
bool try_launch_db(hipStream_t s, ConvParams& p, int N, size_t lds_limit) {
    constexpr int NTHR = 64 * WM * WN;
    constexpr int BM = WM * MT * 16, TH = BM / 16, BN = WN * NT * 16;
    constexpr int TWH = 16 + KS - 1, THH = TH + KS - 1, HPIX = TWH * THH;
    constexpr int NP = (BN % 32 == 0) ? BN + 16 : BN;
    constexpr int KK = KS * KS;
    const int cin4 = (p.Cin + 3) & ~3;
    int bestCK = 0, bestTPS = 0;
    size_t best_lds = 0;
    long best_work = -1;
    const int ck_cand[] = {cin4, 64, 48, 32, 24, 16, 8};
    const int tps_cand[] = {KK, KS, 1};
    for (int ck : ck_cand) {
        if (ck > cin4 || (ck & 3)) continue;
        if (HPIX * (ck / 4) > IR * NTHR) continue;
        const int nchunks = (p.Cin + ck - 1) / ck;
        if (!MC && nchunks > 1) continue;
        for (int tps : tps_cand) {
            if (tps * ck * (BN / 4) > WR * NTHR) continue;
            const size_t lds = (size_t)((nchunks > 1 ? 2 : 1) * ((HPIX * (ck + 2) + 3) & ~3) + 2 * tps * ck * NP) * sizeof(float);
            if (lds > lds_limit) continue;
            // padded channel work (ceil(Cin/ck)*ck) first, then the largest stage (fewest barriers)
            const long waste = (long)nchunks * ck - p.Cin;
            const long work = (long)tps * ck * 1000 - waste * 100000;
            if (work > best_work) { best_work = work; bestCK = ck; bestTPS = tps; best_lds = lds; }
        }
    }
    if (bestCK == 0) return false;
    auto kern = conv_igemm_db_kernel<KS, MT, NT, WM, WN, IR, WR, MC>;
    static std::once_flag once;
    std::call_once(once, [&]() {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kLdsMax));
    });
    p.CK = bestCK;
    p.TPS = bestTPS;
    p.tiles_x = cdiv(p.W, 16);
    p.tiles_y = cdiv(p.H, TH);
    dim3 grid((unsigned)(p.tiles_x * p.tiles_y * N), (unsigned)cdiv(p.Cout, BN));
    const double px = (double)N * p.H * p.W;
    ProfScope ps(s, "conv_igemm_db<" + std::to_string(KS) + "," + std::to_string(MT) + "," + std::to_string(NT) + "," +
                        std::to_string(WM) + "," + std::to_string(WN) + ">",
                 2.0 * px * KS * KS * p.Cin * p.Cout, 4.0 * (px * (p.Cin + p.Cout * (1 + (p.add.p ? 1 : 0) + (p.mask.p ? 1 : 0) + (p.accumulate ? 1 : 0))) + (double)KS * KS * p.Cin * p.Cout));
    DL4DS_LAUNCH(kern, grid, dim3(NTHR), best_lds, s, p);
    HIP_CHECK(hipGetLastError());
    return true;
}

// --------------------------------------------------------------------------------------------
template <int KS, int MT, int NT, int WM, int WN, int KSP = 1>
void launch_fwd(hipStream_t s, ConvParams& p, int N) {
    constexpr int BM = WM * MT * 16, TH = BM / 16, BN = WN * NT * 16;
    constexpr int TWH = 16 + KS - 1, THH = TH + KS - 1, HPIX = TWH * THH;
    constexpr int NP = (BN % 32 == 0) ? BN + 16 : BN;
    constexpr int KK = KS * KS;
    auto kern = conv_igemm_kernel<KS, MT, NT, WM, WN, KSP>;
    // pick the channel chunk CK (multiple of 4) and taps-per-stage under the LDS budget
    const int cin4 = (p.Cin + 3) & ~3;
    int CK = 0, TPS = 1;
    const int cand[] = {cin4, 64, 48, 32, 16, 8, 4};
    auto bytes = [&](int ck, int tps) {
        return (size_t)(((HPIX * (ck + 2) + 3) & ~3) + tps * ck * NP) * sizeof(float);
    };
    for (int c : cand) {
        if (c > cin4) continue;
        if (bytes(c, 1) <= (size_t)kLdsBudget) { CK = c; break; }
    }
    DL4DS_REQUIRE(CK > 0, "conv tile does not fit in LDS");
    const int tps_cand[] = {KK, KS, 1};
    for (int tps : tps_cand) {
        if (bytes(CK, tps) <= (size_t)kLdsBudget) { TPS = tps; break; }
    }
    p.CK = CK;
    p.TPS = TPS;
    p.tiles_x = cdiv(p.W, 16);
    p.tiles_y = cdiv(p.H, TH);
    const size_t lds = std::max(bytes(CK, TPS), KSP > 1 ? (size_t)(KSP - 1) * WM * WN * 64 * MT * NT * 4 * sizeof(float) : (size_t)0);
    static std::once_flag once;
    std::call_once(once, [&]() {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
    });
    dim3 grid((unsigned)(p.tiles_x * p.tiles_y * N), (unsigned)cdiv(p.Cout, BN));
    const double px = (double)N * p.H * p.W;
    // Deep U-Net levels (256 -> 256 at 8x8, the 9x9 transposed convolutions as 5x5 convolutions with 1024 couts): a few
    // dozen blocks each walking thousands of k-steps.  Split the channel chunks over blockIdx.z into plain slabs and let
    // a small kernel sum them and apply the epilogue (fixed order: deterministic).
    static const bool no_splitk = exp_env("DL4DS_NO_SPLITK") != nullptr;
    const long blocks = (long)grid.x * grid.y;
    const int nchunks = cdiv(p.Cin, CK);
    int S = 1;
    static const long sk_target = exp_env("DL4DS_SPLITK_TARGET") ? atol(exp_env("DL4DS_SPLITK_TARGET")) : 768;
    if (!no_splitk && blocks < 256 && nchunks >= 2) S = (int)std::min<long>(nchunks, std::max<long>(2, sk_target / blocks));
    ProfScope ps(s, "conv_igemm<" + std::to_string(KS) + "," + std::to_string(MT) + "," + std::to_string(NT) + "," +
                        std::to_string(WM) + "," + std::to_string(WN) + (S > 1 ? ",splitk>" : ">"),
                 2.0 * px * KK * p.Cin * p.Cout, 4.0 * (px * (p.Cin + p.Cout * (1 + (p.add.p ? 1 : 0) + (p.mask.p ? 1 : 0) + (p.accumulate ? 1 : 0))) + (double)KK * p.Cin * p.Cout));
    p.kchunks = 0; p.nimg = N;
    if (S > 1) {
        const int cps = cdiv(nchunks, S);
        S = cdiv(nchunks, cps);
        const size_t slab = (size_t)N * p.H * p.W * p.Cout;
        float* slabs = splitk_scratch(s, slab * S);
        ConvParams q = p;
        q.out = make_view(slabs, N * S, p.H, p.W, p.Cout);
        q.bias = nullptr; q.add.p = nullptr; q.mask.p = nullptr; q.relu = 0; q.accumulate = 0;
        q.kchunks = cps;
        grid.z = (unsigned)S;
        DL4DS_LAUNCH(kern, grid, dim3(64 * WM * WN * KSP), lds, s, q);
        HIP_CHECK(hipGetLastError());
        const size_t total = slab;
        DL4DS_LAUNCH(splitk_combine_kernel, dim3((unsigned)std::min<size_t>(cdivz(total, 256), 4096)), dim3(256), 0, s, slabs, S,
                           slab, p, N);
        HIP_CHECK(hipGetLastError());
        return;
    }
    DL4DS_LAUNCH(kern, grid, dim3(64 * WM * WN * KSP), lds, s, p);
    HIP_CHECK(hipGetLastError());
}

template <int KS>
void dispatch_fwd(hipStream_t s, ConvParams& p, int N) {
    // choose the cout tile minimising padded work (ties -> larger tile)
    const int bns[] = {192, 128, 96, 48, 32, 16};
    int best = 16;
    long bestw = -1;
    for (int bn : bns) {
        long wk = (long)cdiv(p.Cout, bn) * bn;
        if (bestw < 0 || wk < bestw) { bestw = wk; best = bn; }
    }
    // few pixels, many channels (deep U-Net / deconvolution levels): the grid is tiles x n-blocks, so take the widest
    // cout tile that still gives every CU a block (and at most 1.5x the minimal padded work)
    {
        auto blocks_for = [&](int bn) {
            const int th = (bn >= 96) ? 8 : 16;
            return (long)cdiv(p.W, 16) * cdiv(p.H, th) * N * cdiv(p.Cout, bn);
        };
        // ... unless the reduction is long enough for split-K (launch_fwd) to supply the blocks: a wide cout tile reads 0.5 LDS
        // fragments per MFMA where the 16-cout tile reads 1.25, and the slabs of these layers are a few MB
        const long kparts = std::max(1, cdiv(p.Cin, 64));
        const bool wide_splitk = kparts >= 2 && blocks_for(best) * kparts >= (exp_env("DL4DS_WIDE_SPLITK_MIN") ? atol(exp_env("DL4DS_WIDE_SPLITK_MIN")) : 128) && !exp_env("DL4DS_NO_WIDE_SPLITK");
        if (blocks_for(best) < 256 && !wide_splitk) {
            int pick = best;
            for (int bn : bns) {
                if (bn >= pick) continue;
                const long wk = (long)cdiv(p.Cout, bn) * bn;
                if (wk * 2 > bestw * 3) continue;
                pick = bn;
                if (blocks_for(bn) >= 256) break;
            }
            best = pick;
        }
    }
    // MFMA-bound layers: double-buffered kernel (8 waves for wide Cout, 4 waves x two workgroups per CU otherwise)
    const bool db_ok = p.Cin >= 16 && (long)p.H * p.W >= 256;
    if (db_ok) {
        switch (best) {
            case 192: if (try_launch_db<KS, 4, 6, 4, 2, 8, 6, false>(s, p, N, kLdsMax)) return; break;
            case 128: if (try_launch_db<KS, 4, 4, 4, 2, 8, 6, false>(s, p, N, kLdsMax)) return; break;
            case 96:  if (try_launch_db<KS, 4, 3, 4, 2, 8, 6, true>(s, p, N, kLdsBudget)) return; break;
            case 48:  if (try_launch_db<KS, 4, 3, 4, 1, 6, 3, true>(s, p, N, kLdsBudget)) return; break;
            case 32:  if (try_launch_db<KS, 4, 2, 4, 1, 6, 3, true>(s, p, N, kLdsBudget)) return; break;
            default: break;
        }
    }
    switch (best) {
        case 192: launch_fwd<KS, 4, 6, 2, 2>(s, p, N); break;   // 8x16 pixels x 192 couts
        case 128: launch_fwd<KS, 4, 4, 2, 2>(s, p, N); break;   // 8x16 x 128
        case 96:  launch_fwd<KS, 4, 3, 2, 2>(s, p, N); break;   // 8x16 x 96
        case 48:  launch_fwd<KS, 4, 3, 4, 1>(s, p, N); break;   // 16x16 x 48
        case 32:  launch_fwd<KS, 4, 2, 4, 1>(s, p, N); break;   // 16x16 x 32 (a second wave set measured no better here)
        default:                                                // 16x16 x 16; under-filled multi-tap launches: 2 waves/SIMD
            if (KS > 1 && (long)cdiv(p.W, 16) * cdiv(p.H, 16) * N * cdiv(p.Cout, 16) <= (exp_env("DL4DS_KSP_MAX") ? atol(exp_env("DL4DS_KSP_MAX")) : 256) && !exp_env("DL4DS_NO_KSP"))
                launch_fwd<KS, 4, 1, 4, 1, 2>(s, p, N);       // (four wave sets measured no better than two)
            else
                launch_fwd<KS, 4, 1, 4, 1>(s, p, N);
            break;
    }
}

// --------------------------------------------------------------------------------------------
__global__ void dgrad_weights_kernel(const float* __restrict__ w, float* __restrict__ wt, int KK, int Cin,
                                     int Cout) {
    const size_t total = (size_t)KK * Cin * Cout;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        // destination index e = (tap', co, ci)
        const int ci = (int)(e % Cin);
        const size_t r = e / Cin;
        const int co = (int)(r % Cout);
        const int tp = (int)(r / Cout);
        wt[e] = w[((size_t)(KK - 1 - tp) * Cin + ci) * Cout + co];
    }
}

// --------------------------------------------------------------------------------------------
struct WgradParams {
    TView x, dz;
    float* partial;     // [S*WK][KK*Cin*Cout + Cout]  (weight slab followed by the bias-gradient slab)
    int Cin, Cout, H, W;
    int tiles_x, tiles_y, ntiles, S;
    unsigned m_tx, m_ty;        // div_magic(tiles_x / tiles_y): tile -> (n, ty, tx) on the scalar unit (ntiles < 2^20)
};

// block = 4 waves arranged WCO (cout tiles) x WK (pixel/K split).  A wave owns cout tiles
// [wco*COT,(wco+1)*COT) of the block's 16*COT*WCO couts, ALL (tap, cin-tile) combinations of the block's
// 16*CIT cins, and every WK-th group of 4 pixels of each spatial tile.
template <int KS, int CIT, int COT, int WCO, bool PF>
__global__ void __launch_bounds__(256, 2) conv_wgrad_kernel(const WgradParams a) {
    constexpr int TW = 16, TH = 8;
    constexpr int WK = 4 / WCO;
    constexpr int PAD = KS / 2;
    constexpr int TWH = TW + KS - 1, THH = TH + KS - 1, HPIX = TWH * THH;
    constexpr int KK = KS * KS;
    constexpr int CIB = 16 * CIT;
    constexpr int COB = 16 * COT * WCO;
    constexpr int PX = (CIB % 32 == 16) ? CIB : CIB + 16;   // pixel k and k+1 on disjoint bank halves
    constexpr int PZ = (COB % 32 == 16) ? COB : COB + 16;
    constexpr int NPIX = TW * TH;

    // PF (small-channel variants): the next tile's x / dz are fetched into registers while this tile's MFMAs issue and
    // parked in the second LDS buffer afterwards -> one barrier per tile and no exposed HBM latency.
    constexpr int XR = (HPIX * (CIB / 4) + 255) / 256, ZR = (NPIX * (COB / 4) + 255) / 256;
    constexpr int TILE_FLOATS = HPIX * PX + NPIX * PZ;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* x_tile = smem;                       // [HPIX][PX]
    float* z_tile = smem + HPIX * PX;           // [NPIX][PZ]   (second copy of both at +TILE_FLOATS when PF)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wco = wave % WCO, wk = wave / WCO;
    const int l15 = lane & 15, lq = lane >> 4;
    const int ci0 = blockIdx.z * CIB;
    const int co0 = blockIdx.y * COB;

    f32x4 acc[KK][CIT][COT];
    float bsum[COT];
#pragma unroll
    for (int t = 0; t < KK; ++t)
#pragma unroll
        for (int i = 0; i < CIT; ++i)
#pragma unroll
            for (int j = 0; j < COT; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < COT; ++j) bsum[j] = 0.f;

    float4 xr[PF ? XR : 1], zr[PF ? ZR : 1];
    auto tile_xyn = [&](int tile, int& n, int& y0, int& x0) {
        int t = tile;
        const int tx = t % a.tiles_x;
        t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        n = t / a.tiles_y;
        x0 = tx * TW;
        y0 = ty * TH;
    };
    auto pf_load = [&](int tile) {
        int n, y0, x0;
        tile_xyn(tile, n, y0, x0);
#pragma unroll
        for (int u = 0; u < (PF ? XR : 0); ++u) {
            const int idx0 = tid + u * 256;
            const bool ok = idx0 < HPIX * (CIB / 4);
            const int idx = ok ? idx0 : 0;
            const int pix = idx / (CIB / 4), q = idx - pix * (CIB / 4);
            const int r = pix / TWH, c = pix - r * TWH;
            const int gy = y0 + r - PAD, gx = x0 + c - PAD;
            xr[u] = view_load4_raw(a.x, n, gy, gx, ci0 + q * 4,
                                   ok && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && ci0 + q * 4 < a.Cin);
        }
#pragma unroll
        for (int u = 0; u < (PF ? ZR : 0); ++u) {
            const int idx0 = tid + u * 256;
            const bool ok = idx0 < NPIX * (COB / 4);
            const int idx = ok ? idx0 : 0;
            const int pix = idx / (COB / 4), q = idx - pix * (COB / 4);
            const int r = pix / TW, c = pix - r * TW;
            const int gy = y0 + r, gx = x0 + c;
            zr[u] = view_load4_raw(a.dz, n, gy, gx, co0 + q * 4, ok && gy < a.H && gx < a.W && co0 + q * 4 < a.Cout);
        }
    };
    auto pf_store = [&](int tile, int buf) {
        int n, y0, x0;
        tile_xyn(tile, n, y0, x0);
        float* xt = x_tile + buf * TILE_FLOATS;
        float* zt = z_tile + buf * TILE_FLOATS;
#pragma unroll
        for (int u = 0; u < (PF ? XR : 0); ++u) {
            const int idx = tid + u * 256;
            if (idx < HPIX * (CIB / 4)) {
                const int pix = idx / (CIB / 4), q = idx - pix * (CIB / 4);
                const int r = pix / TWH, c = pix - r * TWH;
                const int gy = y0 + r - PAD, gx = x0 + c - PAD;
                *reinterpret_cast<float4*>(xt + pix * PX + q * 4) =
                    mask4(xr[u], valid4(ci0 + q * 4, a.Cin, gy >= 0 && gy < a.H && gx >= 0 && gx < a.W));
            }
        }
#pragma unroll
        for (int u = 0; u < (PF ? ZR : 0); ++u) {
            const int idx = tid + u * 256;
            if (idx < NPIX * (COB / 4)) {
                const int pix = idx / (COB / 4), q = idx - pix * (COB / 4);
                const int r = pix / TW, c = pix - r * TW;
                *reinterpret_cast<float4*>(zt + pix * PZ + q * 4) =
                    mask4(zr[u], valid4(co0 + q * 4, a.Cout, y0 + r < a.H && x0 + c < a.W));
            }
        }
    };
    if (PF && (int)blockIdx.x < a.ntiles) {
        pf_load(blockIdx.x);
        pf_store(blockIdx.x, 0);
        __syncthreads();
    }
    int pbuf = 0;

    for (int tile = blockIdx.x; tile < a.ntiles; tile += a.S) {
        int n, y0, x0;
        tile_xyn(tile, n, y0, x0);
        const float* xt = x_tile + (PF ? pbuf * TILE_FLOATS : 0);
        const float* zt = z_tile + (PF ? pbuf * TILE_FLOATS : 0);
        if (PF) {
            if (tile + a.S < a.ntiles) pf_load(tile + a.S);
        } else {
        __syncthreads();
        // stage x halo tile (channels ci0 .. ci0+CIB)
        staged_copy<(KS == 1 && WCO == 1 ? 8 : 4), 256>(
            HPIX * (CIB / 4), tid,
            [&](int idx, bool ok) {
                const int pix = idx / (CIB / 4);
                const int q = idx - pix * (CIB / 4);
                const int r = pix / TWH, c = pix - r * TWH;
                const int gy = y0 + r - PAD, gx = x0 + c - PAD;
                return view_load4_raw(a.x, n, gy, gx, ci0 + q * 4,
                                      ok && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && ci0 + q * 4 < a.Cin);
            },
            [&](int idx) {
                const int pix = idx / (CIB / 4);
                const int q = idx - pix * (CIB / 4);
                const int r = pix / TWH, c = pix - r * TWH;
                const int gy = y0 + r - PAD, gx = x0 + c - PAD;
                return valid4(ci0 + q * 4, a.Cin, gy >= 0 && gy < a.H && gx >= 0 && gx < a.W);
            },
            [&](int idx, float4 v) {
                const int pix = idx / (CIB / 4);
                const int q = idx - pix * (CIB / 4);
                *reinterpret_cast<float4*>(x_tile + pix * PX + q * 4) = v;
            });
        // stage dz tile (channels co0 .. co0+COB); zero outside the image so padded pixels add nothing
        staged_copy<(KS == 1 && WCO == 1 ? 8 : 4), 256>(
            NPIX * (COB / 4), tid,
            [&](int idx, bool ok) {
                const int pix = idx / (COB / 4);
                const int q = idx - pix * (COB / 4);
                const int r = pix / TW, c = pix - r * TW;
                const int gy = y0 + r, gx = x0 + c;
                return view_load4_raw(a.dz, n, gy, gx, co0 + q * 4, ok && gy < a.H && gx < a.W && co0 + q * 4 < a.Cout);
            },
            [&](int idx) {
                const int pix = idx / (COB / 4);
                const int q = idx - pix * (COB / 4);
                const int r = pix / TW, c = pix - r * TW;
                return valid4(co0 + q * 4, a.Cout, y0 + r < a.H && x0 + c < a.W);
            },
            [&](int idx, float4 v) {
                const int pix = idx / (COB / 4);
                const int q = idx - pix * (COB / 4);
                *reinterpret_cast<float4*>(z_tile + pix * PZ + q * 4) = v;
            });
        __syncthreads();
        }
        // K loop over the 128 pixels, 4 per MFMA: pixel pk = kk*4 + lq -> (row kk>>2, col (kk&3)*4+lq)
        // software pipeline: the KK*CIT + COT fragments of step kk+WK are in flight while step kk's MFMAs issue
        float av[KK][CIT], bv[COT];
        {
            const int pr = wk >> 2, pc = (wk & 3) * 4 + lq;
#pragma unroll
            for (int j = 0; j < COT; ++j) bv[j] = zt[(pr * TW + pc) * PZ + (wco * COT + j) * 16 + l15];
#pragma unroll
            for (int tp = 0; tp < KK; ++tp)
#pragma unroll
                for (int i = 0; i < CIT; ++i)
                    av[tp][i] = xt[((pr + tp / KS) * TWH + (pc + tp % KS)) * PX + l15 + i * 16];
        }
        for (int kk = wk; kk < NPIX / 4; kk += WK) {
            const int kn = (kk + WK < NPIX / 4) ? kk + WK : kk;
            const int pr = kn >> 2, pc = (kn & 3) * 4 + lq;
            float an[KK][CIT], bn[COT];
#pragma unroll
            for (int j = 0; j < COT; ++j) bn[j] = zt[(pr * TW + pc) * PZ + (wco * COT + j) * 16 + l15];
#pragma unroll
            for (int tp = 0; tp < KK; ++tp)
#pragma unroll
                for (int i = 0; i < CIT; ++i)
                    an[tp][i] = xt[((pr + tp / KS) * TWH + (pc + tp % KS)) * PX + l15 + i * 16];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < COT; ++j) bsum[j] += bv[j];
#pragma unroll
            for (int tp = 0; tp < KK; ++tp)
#pragma unroll
                for (int i = 0; i < CIT; ++i)
#pragma unroll
                    for (int j = 0; j < COT; ++j)
                        acc[tp][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tp][i], bv[j], acc[tp][i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < COT; ++j) bv[j] = bn[j];
#pragma unroll
            for (int tp = 0; tp < KK; ++tp)
#pragma unroll
                for (int i = 0; i < CIT; ++i) av[tp][i] = an[tp][i];
        }
        if (PF) {
            if (tile + a.S < a.ntiles) pf_store(tile + a.S, pbuf ^ 1);     // that buffer was last read one barrier ago
            pbuf ^= 1;
            __syncthreads();
        }
    }
    // K-split waves: pairwise tree reduction through LDS so the block emits ONE slab
    if (WK > 1) {
        constexpr int SLOTF = (KK * CIT * COT * 4 + COT) * 64;      // floats per wave image
        for (int half = WK / 2; half >= 1; half >>= 1) {
            __syncthreads();
            if (wk >= half && wk < 2 * half) {
                float* dst = smem + (size_t)((wk - half) * WCO + wco) * SLOTF + lane;
                int o = 0;
#pragma unroll
                for (int tp = 0; tp < KK; ++tp)
#pragma unroll
                    for (int i = 0; i < CIT; ++i)
#pragma unroll
                        for (int j = 0; j < COT; ++j)
#pragma unroll
                            for (int rg = 0; rg < 4; ++rg) { dst[o * 64] = acc[tp][i][j][rg]; ++o; }
#pragma unroll
                for (int j = 0; j < COT; ++j) { dst[o * 64] = bsum[j]; ++o; }
            }
            __syncthreads();
            if (wk < half) {
                const float* src = smem + (size_t)(wk * WCO + wco) * SLOTF + lane;
                int o = 0;
#pragma unroll
                for (int tp = 0; tp < KK; ++tp)
#pragma unroll
                    for (int i = 0; i < CIT; ++i)
#pragma unroll
                        for (int j = 0; j < COT; ++j)
#pragma unroll
                            for (int rg = 0; rg < 4; ++rg) { acc[tp][i][j][rg] += src[o * 64]; ++o; }
#pragma unroll
                for (int j = 0; j < COT; ++j) { bsum[j] += src[o * 64]; ++o; }
            }
        }
        if (wk != 0) return;
    }
    // write the partial slab: D row = ci (lq*4+reg), col = co (l15)
    const size_t nw = (size_t)KK * a.Cin * a.Cout;
    float* slab = a.partial + (size_t)blockIdx.x * (nw + a.Cout);
#pragma unroll
    for (int tp = 0; tp < KK; ++tp)
#pragma unroll
        for (int i = 0; i < CIT; ++i)
#pragma unroll
            for (int j = 0; j < COT; ++j) {
                const int co = co0 + (wco * COT + j) * 16 + l15;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int ci = ci0 + i * 16 + lq * 4 + rg;
                    if (co < a.Cout && ci < a.Cin) slab[((size_t)tp * a.Cin + ci) * a.Cout + co] = acc[tp][i][j][rg];
                }
            }
    // bias gradient: sum the 4 pixel sub-lanes of each cout column; only the first cin block writes it
    if (blockIdx.z == 0) {
#pragma unroll
        for (int j = 0; j < COT; ++j) {
            float v = bsum[j];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int co = co0 + (wco * COT + j) * 16 + l15;
            if (lq == 0 && co < a.Cout) slab[nw + co] = v;
        }
    }
}

// Row-walking variant of the 3x3 weight gradient (all layers except the small-channel prefetch variants).
// The k-slot of an MFMA is mapped to pixel column q + 4*slot of a 16-pixel row segment (q = 0..3) instead of 4q + slot.
// With that mapping the x fragment of (quad q, tap column kx) is the fragment of column group c = q + kx, so ONE input
// row needs 6 fragment reads per cin tile (c = 0..5) for its 4 quads x 3 tap columns, and it is used by the three
// output rows it feeds (ky = 0..2); a dz row fragment is read once and kept for three input rows.  22 LDS reads per 108
// MFMAs (CIT = 3) instead of 28 per 27: the general kernel above was LDS-read bound, this one is MFMA bound.  All LDS
// addresses are immediates off two per-lane bases; pitches are 4 mod 8 floats so the four k-slots of a fragment read
// (4 pixels apart) fall on two disjoint bank halves.
template <int CIT, int WCO>
__global__ void __launch_bounds__(256, 2) conv_wgrad_rows_kernel(const WgradParams a) {
    [[maybe_unused]] constexpr int KS = 3;
    constexpr int COT = 1;
    constexpr int TW = 16, TH = 8;
    constexpr int WK = 4 / WCO, RW = TH / WK;            // output rows per wave
    constexpr int PAD = 1;
    constexpr int TWH = TW + 2, THH = TH + 2, HPIX = TWH * THH;
    constexpr int KK = 9;
    constexpr int CIB = 16 * CIT, COB = 16 * WCO;
    constexpr int PX = CIB + 4, PZ = COB + 4;
    constexpr int NPIX = TW * TH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* x_tile = smem;                       // [HPIX][PX]
    float* z_tile = smem + HPIX * PX;           // [NPIX][PZ]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wco = wave % WCO, wk = wave / WCO;
    const int l15 = lane & 15, lq = lane >> 4;
    const int ci0 = blockIdx.z * CIB;
    const int co0 = blockIdx.y * COB;

    f32x4 acc[KK][CIT][COT];
    float bsum[COT];
#pragma unroll
    for (int t = 0; t < KK; ++t)
#pragma unroll
        for (int i = 0; i < CIT; ++i) acc[t][i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bsum[0] = 0.f;

    const float* xf = x_tile + ((wk * RW) * TWH + lq * 4) * PX + l15;
    const float* zf = z_tile + ((wk * RW) * TW + lq * 4) * PZ + wco * 16 + l15;

    for (int tile = blockIdx.x; tile < a.ntiles; tile += a.S) {
        int t = tile;
        const int tx = t % a.tiles_x;
        t /= a.tiles_x;
        const int ty = t % a.tiles_y;
        const int n = t / a.tiles_y;
        const int x0 = tx * TW, y0 = ty * TH;
        __syncthreads();
        staged_copy<4, 256>(
            HPIX * (CIB / 4), tid,
            [&](int idx, bool ok) {
                const int pix = idx / (CIB / 4);
                const int q = idx - pix * (CIB / 4);
                const int r = pix / TWH, c = pix - r * TWH;
                const int gy = y0 + r - PAD, gx = x0 + c - PAD;
                return view_load4_raw(a.x, n, gy, gx, ci0 + q * 4,
                                      ok && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && ci0 + q * 4 < a.Cin);
            },
            [&](int idx) {
                const int pix = idx / (CIB / 4);
                const int q = idx - pix * (CIB / 4);
                const int r = pix / TWH, c = pix - r * TWH;
                const int gy = y0 + r - PAD, gx = x0 + c - PAD;
                return valid4(ci0 + q * 4, a.Cin, gy >= 0 && gy < a.H && gx >= 0 && gx < a.W);
            },
            [&](int idx, float4 v) {
                const int pix = idx / (CIB / 4);
                const int q = idx - pix * (CIB / 4);
                *reinterpret_cast<float4*>(x_tile + pix * PX + q * 4) = v;
            });
        staged_copy<4, 256>(
            NPIX * (COB / 4), tid,
            [&](int idx, bool ok) {
                const int pix = idx / (COB / 4);
                const int q = idx - pix * (COB / 4);
                const int r = pix / TW, c = pix - r * TW;
                const int gy = y0 + r, gx = x0 + c;
                return view_load4_raw(a.dz, n, gy, gx, co0 + q * 4, ok && gy < a.H && gx < a.W && co0 + q * 4 < a.Cout);
            },
            [&](int idx) {
                const int pix = idx / (COB / 4);
                const int q = idx - pix * (COB / 4);
                const int r = pix / TW, c = pix - r * TW;
                return valid4(co0 + q * 4, a.Cout, y0 + r < a.H && x0 + c < a.W);
            },
            [&](int idx, float4 v) {
                const int pix = idx / (COB / 4);
                const int q = idx - pix * (COB / 4);
                *reinterpret_cast<float4*>(z_tile + pix * PZ + q * 4) = v;
            });
        __syncthreads();

        float zr[3][4];                 // dz fragments of the last three output rows (ring)
#pragma unroll
        for (int rr = 0; rr < RW + 2; ++rr) {          // input row wk*RW + rr of the halo tile
            float fx[6][CIT];
#pragma unroll
            for (int c = 0; c < 6; ++c)
#pragma unroll
                for (int i = 0; i < CIT; ++i) fx[c][i] = xf[(rr * TWH + c) * PX + i * 16];
            if (rr < RW) {
#pragma unroll
                for (int q = 0; q < 4; ++q) zr[rr % 3][q] = zf[(rr * TW + q) * PZ];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (rr < RW) {
#pragma unroll
                for (int q = 0; q < 4; ++q) bsum[0] += zr[rr % 3][q];
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int r = rr - ky;                  // output row fed through taps (ky, *)
                if (r >= 0 && r < RW) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                            for (int i = 0; i < CIT; ++i)
                                acc[ky * 3 + kx][i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fx[q + kx][i], zr[r % 3][q],
                                                                                              acc[ky * 3 + kx][i][0], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // K-split waves: pairwise tree reduction through LDS so the block emits ONE slab
    if (WK > 1) {
        constexpr int SLOTF = (KK * CIT * COT * 4 + COT) * 64;      // floats per wave image
        for (int half = WK / 2; half >= 1; half >>= 1) {
            __syncthreads();
            if (wk >= half && wk < 2 * half) {
                float* dst = smem + (size_t)((wk - half) * WCO + wco) * SLOTF + lane;
                int o = 0;
#pragma unroll
                for (int tp = 0; tp < KK; ++tp)
#pragma unroll
                    for (int i = 0; i < CIT; ++i)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) { dst[o * 64] = acc[tp][i][0][rg]; ++o; }
                dst[o * 64] = bsum[0];
            }
            __syncthreads();
            if (wk < half) {
                const float* src = smem + (size_t)(wk * WCO + wco) * SLOTF + lane;
                int o = 0;
#pragma unroll
                for (int tp = 0; tp < KK; ++tp)
#pragma unroll
                    for (int i = 0; i < CIT; ++i)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) { acc[tp][i][0][rg] += src[o * 64]; ++o; }
                bsum[0] += src[o * 64];
            }
        }
        if (wk != 0) return;
    }
    // write the partial slab: D row = ci (lq*4+reg), col = co (l15)
    const size_t nw = (size_t)KK * a.Cin * a.Cout;
    float* slab = a.partial + (size_t)blockIdx.x * (nw + a.Cout);
    const int co = co0 + wco * 16 + l15;
#pragma unroll
    for (int tp = 0; tp < KK; ++tp)
#pragma unroll
        for (int i = 0; i < CIT; ++i) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int ci = ci0 + i * 16 + lq * 4 + rg;
                if (co < a.Cout && ci < a.Cin) slab[((size_t)tp * a.Cin + ci) * a.Cout + co] = acc[tp][i][0][rg];
            }
        }
    if (blockIdx.z == 0) {
        float v = bsum[0];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lq == 0 && co < a.Cout) slab[nw + co] = v;
    }
}

// Producer / consumer ("warp-specialised") form of the row-walking kernel: ONE 8-wave workgroup per CU; waves 0-3 run the
// MFMAs of tile t out of LDS buffer t&1 while waves 4-7 stage tile t+1 (x halo + dz) into the other buffer; one barrier
// per tile.  With two independent 4-wave workgroups per CU the staging phases were NOT hidden (measured: 1.72 ms as is,
// 1.26 ms with the loads removed, and no re-ordering of the loads changed the sum) -- here the overlap is by
// construction and the MFMA waves never issue a global load.
// PACK (CIT == 1, Cin <= 8: the ConvLSTM kernels of the recurrent nets, 8 -> 32 gate channels): rows 8..15 of the MFMA's first
// operand would be zero padding; they take the SAME eight input channels one pixel to the right instead, so one MFMA accumulates
// the taps (ky, kx) and (ky, kx + 1): 3 instead of 5 MFMAs per tap row for 5x5, 2 instead of 3 for 3x3.
template <int CIT, int WCO, int KS = 3, bool PACK = false>
__global__ void __launch_bounds__(512, 1) conv_wgrad_rows_ws_kernel(const WgradParams a) {
    static_assert(!PACK || CIT == 1, "tap packing: one 16-row block of which eight are channels");
    constexpr int KXS = PACK ? 2 : 1, NKX = (KS + KXS - 1) / KXS, NACC = KS * NKX;     // tap columns per MFMA, MFMAs per tap row
    constexpr int COT = 1;
    constexpr int TW = 16, TH = 8;
    constexpr int WK = 4 / WCO, RW = TH / WK;            // output rows per wave
    constexpr int PAD = KS / 2;
    constexpr int TWH = TW + KS - 1, THH = TH + KS - 1, HPIX = TWH * THH;
    constexpr int KK = KS * KS;
    constexpr int CIB = 16 * CIT, COB = 16 * WCO;
    constexpr int PX = CIB + 4, PZ = COB + 4;
    constexpr int NPIX = TW * TH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TILE_FLOATS = HPIX * PX + NPIX * PZ;          // one buffer: x halo tile [HPIX][PX] + dz tile [NPIX][PZ]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave8 = tid >> 6;
    const bool producer = wave8 >= 4;            // waves 4-7: staging only
    const int wave = wave8 & 3, stid = tid & 255;
    const int wco = wave % WCO, wk = wave / WCO;
    const int l15 = lane & 15, lq = lane >> 4;
    const int ci0 = blockIdx.z * CIB;
    const int co0 = blockIdx.y * COB;

    const float* xf0 = PACK ? smem + ((wk * RW) * TWH + lq * 4 + (l15 >> 3)) * PX + (l15 & 7)
                            : smem + ((wk * RW) * TWH + lq * 4) * PX + l15;
    const float* zf0 = smem + HPIX * PX + ((wk * RW) * TW + lq * 4) * PZ + wco * 16 + l15;

    if (producer) {
        // Staging waves.  A wave that shares its SIMD with an MFMA wave gets an instruction issued only every ~100 cycles
        // (tools/ws_trace.py on conv_stream_ws), so a tile costs ONE memory instruction per element and a handful of others:
        //   * thread = (tile column, channel quad), element u = tile row u, so a wave's load covers one row: whether that
        //     row lies inside the image is wave-uniform and goes into the buffer descriptor (0 records = every lane out
        //     of range), whether the thread's column does is one compare per tile and goes into its offset;
        //   * out-of-range buffer loads return zeros: exactly the zero padding the halo needs;
        //   * LDS addresses are thread base + immediate.
        typedef int i32x4_t __attribute__((ext_vector_type(4)));
        constexpr int OOB = (int)0xffffff00u, RSRC3 = 0x00020000;
        constexpr int QX = CIB / 4, QZ = COB / 4;
        static_assert(TWH * QX <= 256 && TW * QZ <= 256 && 256 % (TW * QZ) == 0, "one (column, quad) pair per staging thread");
        constexpr int RP = 256 / (TW * QZ), ZR = TH / RP;          // dz: tile rows per pass, passes
        const TView& vx = a.x;
        const TView& vz = a.dz;
        const int rx = vx.d2s > 1 ? vx.d2s : 1, rz = vz.d2s > 1 ? vz.d2s : 1;
        const size_t xsx = (size_t)rx * vx.ld, xsy = (size_t)rx * (size_t)(vx.W * rx) * vx.ld;
        const size_t zsx = (size_t)rz * vz.ld, zsy = (size_t)rz * (size_t)(vz.W * rz) * vz.ld;
        const int xc = stid / QX, xq = stid - xc * QX;
        const bool x_act = stid < TWH * QX;
        const int xv0 = (x_act && ci0 + 4 * xq < a.Cin) ? (int)((xc * xsx + view_chan_off(vx, ci0 + 4 * xq)) * 4) : OOB;
        const int zp = stid % (TW * QZ);
        const int zr0 = __builtin_amdgcn_readfirstlane(stid / (TW * QZ));          // (wave-uniform: TW * QZ is a multiple of 64)
        const int zc = zp / QZ, zq = zp - zc * QZ;
        const int zv0 = (co0 + 4 * zq < a.Cout) ? (int)((zc * zsx + view_chan_off(vz, co0 + 4 * zq)) * 4) : OOB;
        const int x_dst = xc * PX + xq * 4, z_dst = (zr0 * TW + zc) * PZ + zq * 4;
        // views that are not float4-loadable (channel count / pitch not a multiple of four): the last quad of a pixel reaches into
        // the next pixel (harmless, see plan_wgrad) and, for the very last pixel, past the tensor -- there the descriptor carries
        // the exact number of bytes left, and the buffer unit zero-fills dword by dword (measured on gfx950)
        const bool x_exact = !vx.vec, z_exact = !vz.vec;
        const char* x_end = reinterpret_cast<const char*>(vx.p) + ((size_t)(vx.N - 1) * vx.nstride + (size_t)vx.H * vx.W * vx.ld) * 4;
        const char* z_end = reinterpret_cast<const char*>(vz.p) + ((size_t)(vz.N - 1) * vz.nstride + (size_t)vz.H * vz.W * vz.ld) * 4;
        auto stage = [&](int tile, int buf) __attribute__((always_inline)) {
            const int tq = fast_div(tile, a.m_tx);
            const int tx = tile - tq * a.tiles_x;
            const int n = fast_div(tq, a.m_ty);
            const int ty = tq - n * a.tiles_y;
            const int x0 = tx * TW, y0 = ty * TH;
            char* xb = reinterpret_cast<char*>(vx.p) + ((long)((size_t)n * vx.nstride) + (long)(y0 - PAD) * (long)xsy + (long)(x0 - PAD) * (long)xsx) * 4;
            char* zb = reinterpret_cast<char*>(vz.p) + ((size_t)n * vz.nstride + y0 * zsy + x0 * zsx) * 4;
            const int xv = (x0 - PAD + xc >= 0 && x0 - PAD + xc < a.W) ? xv0 : OOB;
            const int zv = (x0 + zc < a.W) ? zv0 : OOB;
            i32x4_t xr[THH], zq4[ZR];
            int nrx = 0x7fffff00, nrz = 0x7fffff00;
            if (x_exact) { const long rem = x_end - xb; nrx = rem < 0x7fffff00l ? (int)rem : 0x7fffff00; }
            if (z_exact) { const long rem = z_end - zb; nrz = rem < 0x7fffff00l ? (int)rem : 0x7fffff00; }
            if (y0 >= PAD && y0 + TH + PAD <= a.H) {              // every row of the halo inside the image: one descriptor each
                const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(xb, 0, nrx, RSRC3);
                const __amdgpu_buffer_rsrc_t rsz = __builtin_amdgcn_make_buffer_rsrc(zb, 0, nrz, RSRC3);
#pragma unroll
                for (int u = 0; u < THH; ++u) xr[u] = __builtin_amdgcn_raw_buffer_load_b128(rsx, xv, (int)(u * xsy * 4), 0);
#pragma unroll
                for (int u = 0; u < ZR; ++u) zq4[u] = __builtin_amdgcn_raw_buffer_load_b128(rsz, zv, (int)((u * RP + zr0) * zsy * 4), 0);
            } else {
#pragma unroll
                for (int u = 0; u < THH; ++u) {
                    const bool row_ok = y0 - PAD + u >= 0 && y0 - PAD + u < a.H;
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xb, 0, row_ok ? nrx : 0, RSRC3);
                    xr[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, xv, (int)(u * xsy * 4), 0);
                }
#pragma unroll
                for (int u = 0; u < ZR; ++u) {
                    const int row = u * RP + zr0;
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(zb, 0, (y0 + row < a.H) ? nrz : 0, RSRC3);
                    zq4[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, zv, (int)(row * zsy * 4), 0);
                }
            }
            float* x_tile = smem + buf * TILE_FLOATS;
            float* z_tile = x_tile + HPIX * PX;
            if (x_act) {
#pragma unroll
                for (int u = 0; u < THH; ++u) *reinterpret_cast<i32x4_t*>(x_tile + x_dst + u * (TWH * PX)) = xr[u];
            }
#pragma unroll
            for (int u = 0; u < ZR; ++u) *reinterpret_cast<i32x4_t*>(z_tile + z_dst + u * (RP * TW * PZ)) = zq4[u];
        };
        if ((int)blockIdx.x < a.ntiles) stage(blockIdx.x, 0);
        __syncthreads();
        int itp = 0;
        for (int tile = blockIdx.x; tile < a.ntiles; tile += a.S, ++itp) {
            if (tile + a.S < a.ntiles) stage(tile + a.S, (itp & 1) ^ 1);
            __syncthreads();
        }
        return;                                        // (finished waves no longer take part in barriers)
    }
    f32x4 acc[NACC][CIT][COT];
    float bsum[COT];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int i = 0; i < CIT; ++i) acc[t][i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bsum[0] = 0.f;

    __syncthreads();                                   // tile 0 is staged
    int it = 0;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += a.S, ++it) {
        const int buf = it & 1;
        const float* xf = xf0 + buf * TILE_FLOATS;
        const float* zf = zf0 + buf * TILE_FLOATS;
        float zr[KS][4];                // dz fragments of the last KS output rows (ring)
#pragma unroll
        for (int rr = 0; rr < RW + KS - 1; ++rr) {     // input row wk*RW + rr of the halo tile
            float fx[4 + KS - 1][CIT];             // (PACK: the upper half-rows read one pixel further: column c + 1)
#pragma unroll
            for (int c = 0; c < 4 + KS - 1; ++c)
#pragma unroll
                for (int i = 0; i < CIT; ++i) fx[c][i] = xf[(rr * TWH + c) * PX + i * 16];
            if (rr < RW) {
#pragma unroll
                for (int q = 0; q < 4; ++q) zr[rr % KS][q] = zf[(rr * TW + q) * PZ];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (rr < RW) {
#pragma unroll
                for (int q = 0; q < 4; ++q) bsum[0] += zr[rr % KS][q];
            }
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int r = rr - ky;                  // output row fed through taps (ky, *)
                if (r >= 0 && r < RW) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int kx = 0; kx < KS; kx += KXS)
#pragma unroll
                            for (int i = 0; i < CIT; ++i)
                                acc[ky * NKX + kx / KXS][i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fx[q + kx][i], zr[r % KS][q],
                                                                                                      acc[ky * NKX + kx / KXS][i][0], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                               // buffer `buf` is free, buffer `buf ^ 1` is filled
    }
    // K-split waves: pairwise tree reduction through LDS so the block emits ONE slab
    if (WK > 1) {
        constexpr int SLOTF = (NACC * CIT * COT * 4 + COT) * 64;      // floats per wave image
        for (int half = WK / 2; half >= 1; half >>= 1) {
            __syncthreads();
            if (wk >= half && wk < 2 * half) {
                float* dst = smem + (size_t)((wk - half) * WCO + wco) * SLOTF + lane;
                int o = 0;
#pragma unroll
                for (int tp = 0; tp < NACC; ++tp)
#pragma unroll
                    for (int i = 0; i < CIT; ++i)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) { dst[o * 64] = acc[tp][i][0][rg]; ++o; }
                dst[o * 64] = bsum[0];
            }
            __syncthreads();
            if (wk < half) {
                const float* src = smem + (size_t)(wk * WCO + wco) * SLOTF + lane;
                int o = 0;
#pragma unroll
                for (int tp = 0; tp < NACC; ++tp)
#pragma unroll
                    for (int i = 0; i < CIT; ++i)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) { acc[tp][i][0][rg] += src[o * 64]; ++o; }
                bsum[0] += src[o * 64];
            }
        }
        if (wk != 0) return;
    }
    // write the partial slab: D row = ci (lq*4+reg), col = co (l15)
    const size_t nw = (size_t)KK * a.Cin * a.Cout;
    float* slab = a.partial + (size_t)blockIdx.x * (nw + a.Cout);
    const int co = co0 + wco * 16 + l15;
    if constexpr (PACK) {
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
            for (int kp = 0; kp < NKX; ++kp)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int row = lq * 4 + rg, ci = ci0 + (row & 7), kx = kp * 2 + (row >> 3);
                    if (co < a.Cout && ci < a.Cin && kx < KS) slab[((size_t)(ky * KS + kx) * a.Cin + ci) * a.Cout + co] = acc[ky * NKX + kp][0][0][rg];
                }
    } else {
#pragma unroll
    for (int tp = 0; tp < KK; ++tp)
#pragma unroll
        for (int i = 0; i < CIT; ++i) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int ci = ci0 + i * 16 + lq * 4 + rg;
                if (co < a.Cout && ci < a.Cin) slab[((size_t)tp * a.Cin + ci) * a.Cout + co] = acc[tp][i][0][rg];
            }
        }
    }
    if (blockIdx.z == 0) {
        float v = bsum[0];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lq == 0 && co < a.Cout) slab[nw + co] = v;
    }
}

// out0[e] (+)= sum_k partial[k*n + e] for e < n0 ; out1[e-n0] likewise for e >= n0.
// One block reduces 16 consecutive elements: thread = (slab sub-index 0..15, element 0..15) -> 64-byte row
// segments per slab, 16 slabs in flight per block, fixed summation order (deterministic).
// rowlen > 0: out0 is a channel SLICE of a wider filter -- element e of a slab lands at (e / rowlen) * dstride + doff + e % rowlen
// (the 9..16-input-channel weight gradients computed as two 8-channel slices: rowlen = Cs * Cout per tap, dstride = C * Cout).
template <int EL>      // elements per block: 16 (64-byte row segments per slab, 16 slabs in flight) or 4 (small filters: 64 slabs in flight)
__global__ void __launch_bounds__(256) reduce_slabs_kernel(const float* __restrict__ partial, float* __restrict__ out0,
                                                           float* __restrict__ out1, size_t n0, size_t n, int S,
                                                           int acc0, int acc1, int rowlen, int dstride, int doff) {
    constexpr int SL = 256 / EL;
    __shared__ float red[SL][EL + 1];
    const int el = threadIdx.x % EL, sl = threadIdx.x / EL;
    const size_t ngroups = (n + EL - 1) / EL;
    for (size_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const size_t e = grp * EL + el;
        float s = 0.f;
        if (e < n) {
            int k = sl;
            for (; k + 3 * SL < S; k += 4 * SL) {       // 4 independent loads in flight
                const float a0 = partial[(size_t)k * n + e], a1 = partial[(size_t)(k + SL) * n + e];
                const float a2 = partial[(size_t)(k + 2 * SL) * n + e], a3 = partial[(size_t)(k + 3 * SL) * n + e];
                s += (a0 + a1) + (a2 + a3);
            }
            for (; k < S; k += SL) s += partial[(size_t)k * n + e];
        }
        red[sl][el] = s;
        __syncthreads();
        if (sl == 0 && e < n) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < SL; ++k) t += red[k][el];
            if (e < n0) {
                const size_t d = rowlen ? (e / (size_t)rowlen) * (size_t)dstride + doff + e % (size_t)rowlen : e;
                out0[d] = acc0 ? out0[d] + t : t;
            } else if (out1) out1[e - n0] = acc1 ? out1[e - n0] + t : t;
        }
        __syncthreads();
    }
}

struct WgradPlan { int S, CIT, WCO, WK, tiles_x, tiles_y, ntiles; bool ws; };

WgradPlan plan_wgrad(const TView& x, const TView& dz, int KS) {
    WgradPlan p;
    // The producer / consumer kernel stages with 16-byte buffer loads.  Plain views whose channel count or pixel pitch is not a
    // multiple of four are fine too: the loads only need dword alignment, and what the last quad of a pixel picks up beyond
    // its channels (the next pixel's first channels) lands in rows / columns of the MFMA tile that are never written out
    // (ci >= Cin, co >= Cout) -- 13-, 26- and 2-channel layers (densenet transitions, LocalizedConvBlock) used to fall back to
    // the scalar-staged kernel at 1.5 TB/s
    auto ws_ok = [](const TView& v) { return v.vec || (v.d2s <= 1 && !exp_env("DL4DS_NO_WGRAD_WS_UNALIGNED")); };
    p.tiles_x = cdiv(x.W, 16);
    p.tiles_y = cdiv(x.H, 8);
    p.ntiles = p.tiles_x * p.tiles_y * x.N;
    p.CIT = (x.C > 32) ? 3 : (x.C > 16 ? 2 : 1);
    if (KS >= 5) p.CIT = 1;
    // cout tiles per block (16 * WCO couts; the other 4 / WCO waves split the tile's rows): the least padded one.  48 and 40
    // couts used to run as one 64-cout block with a quarter of the MFMAs on zero columns (93 TFLOP/s); as three 16-cout
    // blocks whose four waves split the rows they reach 110
    p.WCO = 4;
    for (int wco = 4; wco >= 1; wco >>= 1)
        if (cdiv(dz.C, 16 * wco) * 16 * wco < cdiv(dz.C, 16 * p.WCO) * 16 * p.WCO) p.WCO = wco;
    if (dz.C <= 16) p.WCO = 1;
    else if (dz.C <= 32 && p.WCO == 4) p.WCO = 2;
    // 1x1 layers are HBM-bound: every cout block re-reads the x tile, so the fewest blocks win there (padded MFMAs are free)
    if (KS == 1) p.WCO = (dz.C <= 16) ? 1 : (dz.C <= 32 ? 2 : 4);
    if (const char* e = exp_env("DL4DS_WGRAD_WCO")) p.WCO = atoi(e);      // (experiments)
    p.WK = 4 / p.WCO;
    const int cob = cdiv(dz.C, 16 * p.WCO), cib = cdiv(x.C, 16 * p.CIT);
    // every block does the same amount of work, so the grid should be exactly one residency round:
    // 256 CUs x 2 workgroups (LDS / VGPR limited) = 512 blocks.  768 blocks ran as 1.5 rounds (+33 % time).
    // (the producer/consumer rows kernel runs ONE 8-wave workgroup per CU)
    // (1x1 layers are HBM streaming: the same producer / consumer kernel with one tap)
    // (<= 16 x <= 16 channels used to keep the register-prefetch kernel: 373 us vs 196 us for 16 -> 16 at 16 x 512^2)
    p.ws = (KS == 3 || (KS == 1 && !exp_env("DL4DS_NO_WGRAD_WS1")) || (KS == 5 && !exp_env("DL4DS_NO_WGRAD_WS5"))) && (!(p.CIT == 1 && p.WCO == 1) || !exp_env("DL4DS_NO_WGRAD_WS11")) && ws_ok(x) && ws_ok(dz) &&
           p.ntiles < (1 << 20) && !exp_env("DL4DS_NO_WGRAD_ROWS") && !exp_env("DL4DS_NO_WGRAD_WS");
    int target = std::max(1, (p.ws ? 256 : 512) / (cob * cib));
    const size_t slab = (size_t)KS * KS * x.C * dz.C + dz.C;
    const size_t cap = std::max<size_t>(1, ((size_t)192 << 20) / (slab * sizeof(float)));
    p.S = (int)std::min<size_t>(std::min<size_t>(target, p.ntiles), cap);
    if (p.S < 1) p.S = 1;
    return p;
}

template <int KS, int CIT, int COT, int WCO>
void launch_wgrad(hipStream_t s, WgradParams& p, const WgradPlan& pl) {
    constexpr bool PF = (CIT == 1 && WCO == 1 && KS <= 3);     // prefetch registers fit only for the small-channel variants
    constexpr int TWH = 16 + KS - 1, THH = 8 + KS - 1, HPIX = TWH * THH;
    constexpr int CIB = 16 * CIT, COB = 16 * COT * WCO;
    constexpr int PX = (CIB % 32 == 16) ? CIB : CIB + 16;
    constexpr int PZ = (COB % 32 == 16) ? COB : COB + 16;
    constexpr int WK = 4 / WCO;
    constexpr size_t red_bytes = (size_t)(WK / 2) * WCO * (KS * KS * CIT * COT * 4 + COT) * 64 * sizeof(float);
    // 3x3 layers beyond the small-channel prefetch variants: row-walking kernel (MFMA bound instead of LDS-read bound)
    static const bool no_rows = exp_env("DL4DS_NO_WGRAD_ROWS") != nullptr;
    // (KS == 1, KS == 5 and the small-channel variants: only the producer / consumer form; 5x5 plans always have CIT == 1)
    constexpr bool ROWS_OK = (KS == 3 || KS == 1 || (KS == 5 && CIT == 1)) && COT == 1;
    const bool rows = ROWS_OK && !no_rows && ((KS == 3 && !PF) || pl.ws);
    const bool ws = rows && pl.ws;            // producer/consumer form: float4-loadable views, one workgroup per CU
    const size_t lds = rows ? std::max((size_t)(ws ? 2 : 1) * (HPIX * (CIB + 4) + 128 * (COB + 4)) * sizeof(float), red_bytes)
                            : std::max((size_t)(PF ? 2 : 1) * (HPIX * PX + 128 * PZ) * sizeof(float), red_bytes);
    void (*kern)(const WgradParams) = conv_wgrad_kernel<KS, CIT, COT, WCO, PF>;
    // <= 8 input channels on the producer / consumer kernel: two taps per MFMA (see the kernel's PACK comment); DL4DS_NO_WGRAD_PACK=1 for A/B
    static const bool no_pack = exp_env("DL4DS_NO_WGRAD_PACK") != nullptr;
    constexpr bool PACK_OK = ROWS_OK && CIT == 1 && (KS == 3 || KS == 5);
    const bool pack = PACK_OK && ws && p.Cin <= 8 && !no_pack;
    if constexpr (ROWS_OK) {
        if constexpr (KS == 3) {
            if constexpr (PF) { if (rows) kern = conv_wgrad_rows_ws_kernel<CIT, WCO, 3>; }
            else if (rows) kern = ws ? conv_wgrad_rows_ws_kernel<CIT, WCO, 3> : conv_wgrad_rows_kernel<CIT, WCO>;
        } else {
            if (rows) kern = conv_wgrad_rows_ws_kernel<CIT, WCO, KS>;
        }
    }
    if constexpr (PACK_OK) {
        if (pack) kern = conv_wgrad_rows_ws_kernel<1, WCO, KS, true>;
    }
    static std::once_flag once;
    std::call_once(once, [&]() {
        // 7x7 (ConvNext stem / tail): the cross-wave reduction buffer alone is 100 KB -> one workgroup per CU
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<KS, CIT, COT, WCO, PF>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, KS >= 7 ? kLdsMax : kLdsBudget));
        if constexpr (PACK_OK)
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_rows_ws_kernel<1, WCO, KS, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kLdsMax));
        if constexpr (ROWS_OK) {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_rows_ws_kernel<CIT, WCO, KS>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kLdsMax));
            if constexpr (KS == 3 && !PF)
                HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_rows_kernel<CIT, WCO>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
        }
    });
    DL4DS_REQUIRE(lds <= (size_t)((ws || KS >= 7) ? kLdsMax : kLdsBudget), "wgrad tile does not fit in LDS");
    dim3 grid((unsigned)pl.S, (unsigned)cdiv(p.Cout, COB), (unsigned)cdiv(p.Cin, CIB));
    const double px = (double)p.x.N * p.H * p.W;
    ProfScope ps(s, std::string(rows ? "conv_wgrad_rows<" : "conv_wgrad<") + std::to_string(KS) + "," + std::to_string(CIT) +
                        "," + std::to_string(COT) + "," + std::to_string(WCO) + ">",
                 2.0 * px * KS * KS * p.Cin * p.Cout, 4.0 * (px * (p.Cin + p.Cout) + (double)KS * KS * p.Cin * p.Cout));
    DL4DS_LAUNCH(kern, grid, dim3(ws ? 512 : 256), lds, s, p);
    HIP_CHECK(hipGetLastError());
}

template <int KS, int CIT>
void dispatch_wgrad_wco(hipStream_t s, WgradParams& p, const WgradPlan& pl) {
    switch (pl.WCO) {
        case 1: launch_wgrad<KS, CIT, 1, 1>(s, p, pl); break;
        case 2: launch_wgrad<KS, CIT, 1, 2>(s, p, pl); break;
        default: launch_wgrad<KS, CIT, 1, 4>(s, p, pl); break;
    }
}

template <int KS>
void dispatch_wgrad(hipStream_t s, WgradParams& p, const WgradPlan& pl) {
    switch (pl.CIT) {
        case 3: dispatch_wgrad_wco<KS, 3>(s, p, pl); break;
        case 2: dispatch_wgrad_wco<KS, 2>(s, p, pl); break;
        default: dispatch_wgrad_wco<KS, 1>(s, p, pl); break;
    }
}

}  // namespace

// split-K plumbing shared with conv_gemm.hip: the per-stream slab scratch and the combine + epilogue launch
float* conv_splitk_scratch(hipStream_t s, size_t floats) { return splitk_scratch(s, floats); }
void conv_splitk_combine(hipStream_t s, const float* slabs, int S, size_t slab_stride, const ConvParams& p, int N) {
    const size_t total = (size_t)N * p.H * p.W * p.Cout;
    DL4DS_LAUNCH(splitk_combine_kernel, dim3((unsigned)std::min<size_t>(cdivz(total, 256), 4096)), dim3(256), 0, s, slabs, S, slab_stride, p, N);
    HIP_CHECK(hipGetLastError());
}

// =============================================================================================
void conv2d_forward(hipStream_t s, const TView& in, const float* w, int KS, const TView& out,
                    const ConvEpilogue& ep) {
    DL4DS_REQUIRE(in.N == out.N && in.H == out.H && in.W == out.W, "conv2d: stride-1 SAME shapes differ");
    if (conv2d_direct_forward(s, in, w, KS, out, ep)) return;      // a handful of channels: HBM-bound stencil
    if (!exp_env("DL4DS_NO_NARROW") && conv2d_narrow_forward(s, in, w, KS, out, ep)) return;
    DL4DS_REQUIRE(!in.sc && !ep.pool, "conv2d: channel-affine input / pooling partials are only implemented by the direct "
                                     "and narrow-pair kernels (the caller must check conv2d_direct_eligible / conv2d_narrow_pair_ok)");
    if (KS == 3 && conv2d_split_forward(s, in, w, out, ep)) return; // 40 / 48-channel 3x3 layers: fp32 products as six bf16 MFMA terms
    if (KS == 3 && conv2d_wino_forward(s, in, w, out, ep)) return;  // MFMA-bound 3x3 layers: Winograd F(2x2, 3x3)
    // small grids, many channels: GEMM over the flattened pixels of the batch -- up to 16 x 16 ahead of the streaming kernels (which
    // need two tiles per workgroup), up to 32 x 32 for what they decline (measured on cfg5: 106 vs 128-133 TFLOP/s where both apply)
    if (conv2d_gemm_forward(s, in, w, KS, out, ep, 16 * 16)) return;
    if (KS == 1 && exp_env("DL4DS_POINT_FIRST") && conv2d_point_forward(s, in, w, KS, out, ep)) return;     // (experiment)
    if (!exp_env("DL4DS_NO_STREAM") && conv2d_stream_forward(s, in, w, KS, out, ep)) return;
    if (conv2d_point_forward(s, in, w, KS, out, ep)) return;      // 1x1 with channel counts that are not multiples of four
    if (conv2d_gemm_forward(s, in, w, KS, out, ep, 32 * 32)) return;
    ConvParams p;
    p.in = in; p.out = out; p.add = ep.add; p.mask = ep.mask;
    p.w = w; p.bias = ep.bias;
    p.Cin = in.C; p.Cout = out.C; p.H = in.H; p.W = in.W;
    p.relu = ep.relu; p.accumulate = ep.accumulate;
    p.wvec = ((out.C & 3) == 0) && ((((uintptr_t)w) & 15) == 0);
    p.CK = 0; p.TPS = 1; p.tiles_x = p.tiles_y = 0;
    switch (KS) {
        case 1: dispatch_fwd<1>(s, p, in.N); break;
        case 3: dispatch_fwd<3>(s, p, in.N); break;
        case 5: dispatch_fwd<5>(s, p, in.N); break;
        case 7: dispatch_fwd<7>(s, p, in.N); break;       // ConvNext stem / tail (sp_postups.py:121,205-210)
        default: throw Dl4dsError("conv2d: kernel size " + std::to_string(KS) + " not supported (1,3,5,7)");
    }
}

// the same map for large filters (deep U-Net levels: 25 x 256 x 1024): 32 x 32 tiles through LDS so that both the read
// (rows of Cout floats) and the write (rows of Cin floats) are coalesced
__global__ void __launch_bounds__(256) dgrad_weights_tiled_kernel(const float* __restrict__ w, float* __restrict__ wt, int KK, int Cin,
                                                                  int Cout) {
    __shared__ float tile[32][33];
    const int tp = blockIdx.z;                                   // destination tap
    const float* src = w + (size_t)(KK - 1 - tp) * Cin * Cout;   // [Cin][Cout]
    float* dst = wt + (size_t)tp * Cout * Cin;                   // [Cout][Cin]
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        tile[r][tx] = (ci < Cin && co < Cout) ? src[(size_t)ci * Cout + co] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        if (co < Cout && ci < Cin) dst[(size_t)co * Cin + ci] = tile[tx][r];
    }
}

// one launch for a list of filters: block b belongs to the job with block0 <= b < next job's block0 and transposes a
// 32 x 32 (cin x cout) tile of one tap
__global__ void __launch_bounds__(256) dgrad_weights_batched_kernel(const DgradWeightsJob* __restrict__ jobs, int nj) {
    __shared__ float tile[32][33];
    int j = 0;
    while (j + 1 < nj && (int)blockIdx.x >= jobs[j + 1].block0) ++j;         // a few dozen jobs: linear scan
    const DgradWeightsJob job = jobs[j];
    int b = (int)blockIdx.x - job.block0;
    const int nci = (job.Cin + 31) / 32, nco = (job.Cout + 31) / 32;
    const int bi = b % nci; b /= nci;
    const int bo = b % nco;
    const int tp = b / nco;
    const float* src = job.src + (size_t)(job.KK - 1 - tp) * job.Cin * job.Cout;
    float* dst = job.dst + (size_t)tp * job.Cout * job.Cin;
    const int ci0 = bi * 32, co0 = bo * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        tile[r][tx] = (ci < job.Cin && co < job.Cout) ? src[(size_t)ci * job.Cout + co] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        if (co < job.Cout && ci < job.Cin) dst[(size_t)co * job.Cin + ci] = tile[tx][r];
    }
}
int dgrad_weights_job_blocks(int KK, int Cin, int Cout) { return cdiv(Cin, 32) * cdiv(Cout, 32) * KK; }
void conv2d_dgrad_weights_batched(hipStream_t s, const DgradWeightsJob* jobs_dev, int nj, int blocks) {
    if (nj == 0 || blocks == 0) return;
    ProfScope ps(s, "dgrad_weights_batched", 0.0, 0.0);
    DL4DS_LAUNCH(dgrad_weights_batched_kernel, dim3(blocks), dim3(256), 0, s, jobs_dev, nj);
    HIP_CHECK(hipGetLastError());
}

void conv2d_dgrad_weights(hipStream_t s, const float* w, float* wt, int KS, int Cin, int Cout) {
    const size_t total = (size_t)KS * KS * Cin * Cout;
    ProfScope ps(s, "dgrad_weights", 0.0, 8.0 * (double)total);
    if ((size_t)Cin * Cout >= 4096 && Cin >= 16 && Cout >= 16) {
        DL4DS_LAUNCH(dgrad_weights_tiled_kernel, dim3(cdiv(Cin, 32), cdiv(Cout, 32), KS * KS), dim3(256), 0, s, w, wt, KS * KS,
                           Cin, Cout);
    } else {
        const int blocks = (int)std::min<size_t>(cdivz(total, 256), 2048);
        DL4DS_LAUNCH(dgrad_weights_kernel, dim3(blocks), dim3(256), 0, s, w, wt, KS * KS, Cin, Cout);
    }
    HIP_CHECK(hipGetLastError());
}

// 9 .. 16 input channels with <= 8 outputs (13 -> 8: ConvBlock_att's first layer in the recurrent nets; 16 -> 8 U-Net decoder
// layers): the general kernel runs these with the eight output channels on half of the MFMA's rows (nine MFMAs per pixel quad);
// conv_narrow_wgrad_kernel<8> on the channel slices [0, 8) and [8, C) takes 2 x 3 (rows 8-15 carry the pixel below).  The slices are
// plain views of the same pixels (pitch = the tensor's); the slab sums scatter into the [9][C][Cout] gradient.
static TView channel_slice(const TView& x, int c0, int c1) {
    TView v = x;
    v.p = x.p + c0; v.C = c1 - c0; v.cp = v.C;
    v.vec = ((v.C & 3) == 0) && ((x.ld & 3) == 0) && ((((uintptr_t)v.p) & 15) == 0);
    return v;
}
static int narrow_wgrad_split_slabs(const TView& x, const TView& dz, int KS) {
    static const bool off = exp_env("DL4DS_NO_WGRAD_SPLIT") != nullptr;                 // (A/B)
    if (off || KS != 3 || x.C <= 8 || x.C > 16 || dz.C > 8 || x.sc || x.d2s > 1 || dz.d2s > 1 || !dz.vec) return 0;
    if (conv2d_direct_wgrad_slabs(x, dz, KS) || exp_env("DL4DS_NO_NARROW")) return 0;
    return conv2d_narrow_wgrad_slabs(channel_slice(x, 0, 8), dz, KS);
}

size_t conv2d_wgrad_workspace_bytes(const TView& x, const TView& dz, int KS) {
    const size_t slab = ((size_t)KS * KS * x.C * dz.C + dz.C) * sizeof(float);
    if (const int ds = conv2d_direct_wgrad_slabs(x, dz, KS)) return (size_t)ds * slab;
    // (the narrow path can be switched off for A/B runs: size for whichever of the two plans needs more)
    const size_t general = (size_t)plan_wgrad(x, dz, KS).S * slab;
    if (const int ns = conv2d_narrow_wgrad_slabs(x, dz, KS)) return std::max((size_t)ns * slab, general);
    if (const int ns = narrow_wgrad_split_slabs(x, dz, KS)) return std::max((size_t)2 * ns * slab, general);     // (two slices, each < slab)
    return general;
}

void conv2d_wgrad(hipStream_t s, const TView& x, const TView& dz, int KS, float* dw, int accumulate, float* db,
                  int accumulate_db, float* workspace, size_t workspace_bytes) {
    DL4DS_REQUIRE(x.N == dz.N && x.H == dz.H && x.W == dz.W, "wgrad: shapes differ");
    if (KS == 3 && conv2d_wino_wgrad(s, x, dz, dw, accumulate, db, accumulate_db)) return;     // MFMA-bound 3x3 layers: Winograd
    if (const int ns = narrow_wgrad_split_slabs(x, dz, KS)) {
        float* ws = workspace;
        for (int c0 = 0; c0 < x.C; c0 += 8) {
            const TView xs = channel_slice(x, c0, std::min(c0 + 8, x.C));
            const size_t nws = (size_t)9 * xs.C * dz.C, ns_n = nws + dz.C;
            DL4DS_REQUIRE((size_t)(ws - workspace) * sizeof(float) + (size_t)ns * ns_n * sizeof(float) <= workspace_bytes, "wgrad: workspace too small");
            const int got = conv2d_narrow_wgrad(s, xs, dz, KS, ws, ns);
            ProfScope ps(s, "wgrad_reduce_slabs", 0.0, 4.0 * (double)ns_n * (got + 1));
            float* dbs = c0 == 0 ? db : nullptr;                                           // (every slice sums the same dz)
            if (got >= 128) {
                DL4DS_LAUNCH(reduce_slabs_kernel<4>, dim3((int)cdivz(ns_n, 4)), dim3(256), 0, s, ws, dw, dbs, nws, ns_n, got, accumulate,
                             accumulate_db, xs.C * dz.C, x.C * dz.C, c0 * dz.C);
            } else {
                DL4DS_LAUNCH(reduce_slabs_kernel<16>, dim3((int)cdivz(ns_n, 16)), dim3(256), 0, s, ws, dw, dbs, nws, ns_n, got, accumulate,
                             accumulate_db, xs.C * dz.C, x.C * dz.C, c0 * dz.C);
            }
            HIP_CHECK(hipGetLastError());
            ws += (size_t)ns * ns_n;
        }
        return;
    }
    WgradPlan pl = plan_wgrad(x, dz, KS);
    const size_t nw = (size_t)KS * KS * x.C * dz.C;
    const size_t n = nw + dz.C;
    const int direct_slabs = conv2d_direct_wgrad_slabs(x, dz, KS);
    const int narrow_slabs = (direct_slabs || exp_env("DL4DS_NO_NARROW")) ? 0 : conv2d_narrow_wgrad_slabs(x, dz, KS);
    DL4DS_REQUIRE(direct_slabs || ((!x.sc) && (narrow_slabs || !dz.sc)),
                  "wgrad: channel-affine operands are only implemented by the direct (x) and narrow (dz) kernels");
    int nslabs = direct_slabs ? direct_slabs : (narrow_slabs ? narrow_slabs : pl.S);
    DL4DS_REQUIRE(workspace_bytes >= (size_t)nslabs * n * sizeof(float), "wgrad: workspace too small");
    WgradParams p;
    p.x = x; p.dz = dz; p.partial = workspace;
    p.Cin = x.C; p.Cout = dz.C; p.H = x.H; p.W = x.W;
    p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.ntiles = pl.ntiles; p.S = pl.S;
    p.m_tx = div_magic(p.tiles_x); p.m_ty = div_magic(p.tiles_y);
    if (direct_slabs) {
        nslabs = conv2d_direct_wgrad(s, x, dz, KS, workspace, direct_slabs);
    } else if (narrow_slabs) {
        nslabs = conv2d_narrow_wgrad(s, x, dz, KS, workspace, narrow_slabs);
    } else {
        switch (KS) {
            case 1: dispatch_wgrad<1>(s, p, pl); break;
            case 3: dispatch_wgrad<3>(s, p, pl); break;
            case 5: dispatch_wgrad_wco<5, 1>(s, p, pl); break;
            case 7: dispatch_wgrad_wco<7, 1>(s, p, pl); break;
            default: throw Dl4dsError("wgrad: kernel size not supported (1,3,5,7)");
        }
    }
    ProfScope ps(s, "wgrad_reduce_slabs", 0.0, 4.0 * (double)n * (nslabs + 1));
    // small filters (a 3x3 8 -> 8 layer has 584 values in up to 1024 slabs): 16 elements per block would leave 37 blocks walking
    // 64 slabs per thread one latency after the other (13 us); 4 elements per block put 64 slabs in flight per block
    if (n < 4096 && nslabs >= 128 && !exp_env("DL4DS_REDUCE16")) {
        const int blocks = (int)std::max<size_t>(1, cdivz(n, 4));
        DL4DS_LAUNCH(reduce_slabs_kernel<4>, dim3(blocks), dim3(256), 0, s, workspace, dw, db, nw, n, nslabs, accumulate,
                           accumulate_db, 0, 0, 0);
    } else {
        const int blocks = (int)std::max<size_t>(1, std::min<size_t>(cdivz(n, 16), 8192));
        DL4DS_LAUNCH(reduce_slabs_kernel<16>, dim3(blocks), dim3(256), 0, s, workspace, dw, db, nw, n, nslabs, accumulate,
                           accumulate_db, 0, 0, 0);
    }
    HIP_CHECK(hipGetLastError());
}
