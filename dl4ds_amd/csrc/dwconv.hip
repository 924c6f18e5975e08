// dl4ds_amd -- DepthwiseConv2D(kernel_size=7, padding='same', depth_multiplier=1) of ConvNextBlock
// (dl4ds/models/blocks.py:143-144,176): y[n,h,w,c] = b[c] + sum_{ky,kx} x[n,h+ky-3,w+kx-3,c] * k[ky,kx,c].
//
// 49 multiply-adds per output element and no reduction over channels: this is VALU + cache work, not MFMA work.  A
// thread owns a channel pack (float4 when C % 4 == 0) and PW consecutive pixels of one row, so a row of PW + K - 1
// inputs and K weights feeds PW * K FMAs; the 7 x 7 halo re-reads stay in L1/L2, HBM sees x once and y once.
//   forward : taps as stored;        dgrad : the same kernel with the taps mirrored (flip = 1), optional accumulate;
//   wgrad   : thread = (channel pack, ky, pixel lane) keeps the K taps of row ky in registers and walks a chunk of
//             PW-pixel row segments (K + PW - 1 inputs and PW gradients per K * PW FMAs); lanes are combined through LDS, chunks through fixed-order partial sums (bit-reproducible).
#include "ops.h"
#include "prof.h"
#include <algorithm>

namespace {

constexpr int DW_PW = 4;            // output pixels per thread (along W)
constexpr int DW_MAX_CHUNKS = 512;  // wgrad partial slabs

template <int V> struct Pk { float v[V]; };
template <int V> __device__ __forceinline__ Pk<V> ldp(const float* p) {
    Pk<V> r;
    if constexpr (V == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    } else {
        r.v[0] = *p;
    }
    return r;
}
template <int V> __device__ __forceinline__ void stp(float* p, const Pk<V>& r) {
    if constexpr (V == 4) *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
    else *p = r.v[0];
}

template <int K, int V>
__global__ void __launch_bounds__(256) dwconv_kernel(const float* __restrict__ x, const float* __restrict__ k,
                                                     const float* __restrict__ bias, float* __restrict__ y, int N, int H, int W,
                                                     int C, int flip, int accumulate) {
    constexpr int R = K / 2;
    const int CP = C / V;
    const int WG = (W + DW_PW - 1) / DW_PW;
    const size_t total = (size_t)N * H * WG * CP;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int cp = (int)(e % CP);
        size_t r = e / CP;
        const int wg = (int)(r % WG); r /= WG;
        const int h = (int)(r % H);
        const size_t n = r / H;
        const int w0 = wg * DW_PW;
        Pk<V> acc[DW_PW];
        const Pk<V> b = bias ? ldp<V>(bias + cp * V) : Pk<V>{};
#pragma unroll
        for (int p = 0; p < DW_PW; ++p)
#pragma unroll
            for (int i = 0; i < V; ++i) acc[p].v[i] = bias ? b.v[i] : 0.f;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int yy = h + ky - R;
            if (yy < 0 || yy >= H) continue;
            const float* row = x + ((n * H + yy) * (size_t)W) * C + cp * V;
            Pk<V> xv[DW_PW + K - 1];
#pragma unroll
            for (int j = 0; j < DW_PW + K - 1; ++j) {
                const int xx = w0 + j - R;
                if (xx >= 0 && xx < W) {
                    xv[j] = ldp<V>(row + (size_t)xx * C);
                } else {
#pragma unroll
                    for (int i = 0; i < V; ++i) xv[j].v[i] = 0.f;
                }
            }
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int tap = flip ? (K * K - 1 - (ky * K + kx)) : (ky * K + kx);
                const Pk<V> wv = ldp<V>(k + (size_t)tap * C + cp * V);
#pragma unroll
                for (int p = 0; p < DW_PW; ++p)
#pragma unroll
                    for (int i = 0; i < V; ++i) acc[p].v[i] = fmaf(xv[p + kx].v[i], wv.v[i], acc[p].v[i]);
            }
        }
#pragma unroll
        for (int p = 0; p < DW_PW; ++p) {
            if (w0 + p >= W) break;
            float* dst = y + ((n * H + h) * (size_t)W + w0 + p) * C + cp * V;
            if (accumulate) {
                const Pk<V> o = ldp<V>(dst);
#pragma unroll
                for (int i = 0; i < V; ++i) acc[p].v[i] += o.v[i];
            }
            stp<V>(dst, acc[p]);
        }
    }
}

// Thread t < T = LANES * K * CPB: channel pack cb = t % CPB (of the block's channel group), tap row ky = (t / CPB) % K,
// pixel lane = t / (CPB * K).  partial[chunk][tap][c] (+ partial[chunk][K*K][c] = sum dy for the bias).
template <int K, int V>
__global__ void __launch_bounds__(256) dwconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ partial, int N, int H, int W, int C, int CPB,
                                                           int LANES, size_t chunk) {
    constexpr int R = K / 2;
    __shared__ float red[256 * V];
    const int t = threadIdx.x;
    const int cb = t % CPB, ky = (t / CPB) % K, lane = t / (CPB * K);
    const int cp = blockIdx.y * CPB + cb;
    const int CP = C / V;
    const bool active = lane < LANES && cp < CP;
    const size_t nitems = (size_t)N * H * ((W + DW_PW - 1) / DW_PW);
    const size_t p0 = (size_t)blockIdx.x * chunk, p1 = min(p0 + chunk, nitems);
    float acc[K][V], bsum[V];
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (int i = 0; i < V; ++i) acc[j][i] = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) bsum[i] = 0.f;
    if (active) {
        // work item = DW_PW consecutive pixels of one row: K + DW_PW - 1 inputs and DW_PW gradients feed K * DW_PW FMAs
        const int WG = (W + DW_PW - 1) / DW_PW;
        for (size_t it = p0 + lane; it < p1; it += LANES) {
            const int wg = (int)(it % WG);
            const size_t q = it / WG;                     // n * H + h
            const int h = (int)(q % H);
            const int w0 = wg * DW_PW;
            Pk<V> d[DW_PW];
#pragma unroll
            for (int p = 0; p < DW_PW; ++p) {
                if (w0 + p < W) {
                    d[p] = ldp<V>(dy + (q * W + w0 + p) * C + cp * V);
                } else {
#pragma unroll
                    for (int i = 0; i < V; ++i) d[p].v[i] = 0.f;
                }
            }
            if (ky == 0) {
#pragma unroll
                for (int p = 0; p < DW_PW; ++p)
#pragma unroll
                    for (int i = 0; i < V; ++i) bsum[i] += d[p].v[i];
            }
            const int yy = h + ky - R;
            if (yy < 0 || yy >= H) continue;
            const float* row = x + ((q - h + yy) * (size_t)W) * C + cp * V;     // (n*H + yy) * W
            Pk<V> a[DW_PW + K - 1];
#pragma unroll
            for (int j = 0; j < DW_PW + K - 1; ++j) {
                const int xx = w0 + j - R;
                if (xx >= 0 && xx < W) {
                    a[j] = ldp<V>(row + (size_t)xx * C);
                } else {
#pragma unroll
                    for (int i = 0; i < V; ++i) a[j].v[i] = 0.f;
                }
            }
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
#pragma unroll
                for (int p = 0; p < DW_PW; ++p)
#pragma unroll
                    for (int i = 0; i < V; ++i) acc[kx][i] = fmaf(a[p + kx].v[i], d[p].v[i], acc[kx][i]);
        }
    }
    float* out = partial + (size_t)blockIdx.x * (K * K + 1) * C;
#pragma unroll
    for (int j = 0; j <= K; ++j) {            // j == K: the bias sums held by the ky == 0 threads
        __syncthreads();
#pragma unroll
        for (int i = 0; i < V; ++i) red[t * V + i] = (j < K) ? acc[j < K ? j : 0][i] : bsum[i];
        __syncthreads();
        if (lane == 0 && cp < CP && (j < K || ky == 0)) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                float s = 0.f;
                for (int l = 0; l < LANES; ++l) s += red[((l * K + ky) * CPB + cb) * V + i];
                const int tap = (j < K) ? ky * K + j : K * K;
                out[(size_t)tap * C + cp * V + i] = s;
            }
        }
    }
}

__global__ void dwconv_reduce_kernel(const float* __restrict__ partial, int nchunks, int KK, int C, float* __restrict__ dk,
                                     float* __restrict__ db, int accumulate) {
    const int n = (KK + 1) * C;
    for (int e = blockIdx.x * 4 + (threadIdx.x >> 6); e < n; e += gridDim.x * 4) {       // one wavefront per output
        double s = 0.0;
        for (int b = threadIdx.x & 63; b < nchunks; b += 64) s += (double)partial[(size_t)b * n + e];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (threadIdx.x & 63) continue;
        float* d = e < KK * C ? dk + e : (db ? db + (e - KK * C) : nullptr);
        if (d) *d = accumulate ? *d + (float)s : (float)s;
    }
}

inline bool vec_ok(int C, std::initializer_list<const void*> ptrs) {
    if (C & 3) return false;
    for (const void* p : ptrs)
        if (p && ((uintptr_t)p & 15)) return false;
    return true;
}

struct WgradGeom { int CPB, LANES, groups, nchunks; size_t chunk; };
inline WgradGeom wgrad_geom(size_t npix, int C, int V, int K) {     // npix: work items (DW_PW-pixel row segments)
    WgradGeom g;
    const int CP = C / V;
    g.CPB = std::min(CP, 256 / K);                       // channel packs per block
    g.LANES = std::max(1, 256 / (g.CPB * K));
    g.groups = (CP + g.CPB - 1) / g.CPB;
    const size_t want = std::max<size_t>(1, std::min<size_t>(DW_MAX_CHUNKS, npix / (size_t)(g.LANES * 4) + 1));
    g.chunk = (npix + want - 1) / want;
    g.nchunks = (int)((npix + g.chunk - 1) / g.chunk);
    return g;
}

}  // namespace

void dwconv_forward(hipStream_t s, const float* x, const float* k, const float* bias, float* y, int N, int H, int W, int C,
                    int KS, int flip, int accumulate) {
    DL4DS_REQUIRE(KS == 7, "depthwise conv: kernel size must be 7");
    const size_t npix = (size_t)N * H * W;
    if (npix == 0) return;
    ProfScope ps(s, flip ? "dwconv_dgrad" : "dwconv_fwd", 2.0 * KS * KS * (double)npix * C, 8.0 * (double)npix * C);
    const bool v4 = vec_ok(C, {x, k, bias, y});
    const size_t total = (size_t)N * H * ((W + DW_PW - 1) / DW_PW) * (C / (v4 ? 4 : 1));
    const int blocks = (int)std::max<size_t>(1, std::min<size_t>((total + 255) / 256, 1 << 20));
    if (v4) DL4DS_LAUNCH((dwconv_kernel<7, 4>), dim3(blocks), dim3(256), 0, s, x, k, bias, y, N, H, W, C, flip, accumulate);
    else DL4DS_LAUNCH((dwconv_kernel<7, 1>), dim3(blocks), dim3(256), 0, s, x, k, bias, y, N, H, W, C, flip, accumulate);
    HIP_CHECK(hipGetLastError());
}

size_t dwconv_wgrad_workspace_bytes(int C, int KS) { return (size_t)DW_MAX_CHUNKS * (KS * KS + 1) * C * sizeof(float); }

void dwconv_wgrad(hipStream_t s, const float* x, const float* dy, float* dk, float* db, int accumulate, int N, int H, int W, int C,
                  int KS, float* ws, size_t ws_bytes) {
    DL4DS_REQUIRE(KS == 7, "depthwise conv: kernel size must be 7");
    DL4DS_REQUIRE(ws_bytes >= dwconv_wgrad_workspace_bytes(C, KS), "depthwise wgrad: workspace too small");
    const size_t npix = (size_t)N * H * W;
    if (npix == 0) return;
    ProfScope ps(s, "dwconv_wgrad", 2.0 * KS * KS * (double)npix * C, 8.0 * (double)npix * C);
    const bool v4 = vec_ok(C, {x, dy, ws});
    const WgradGeom g = wgrad_geom((size_t)N * H * ((W + DW_PW - 1) / DW_PW), C, v4 ? 4 : 1, KS);
    if (v4) DL4DS_LAUNCH((dwconv_wgrad_kernel<7, 4>), dim3(g.nchunks, g.groups), dim3(256), 0, s, x, dy, ws, N, H, W, C, g.CPB, g.LANES, g.chunk);
    else DL4DS_LAUNCH((dwconv_wgrad_kernel<7, 1>), dim3(g.nchunks, g.groups), dim3(256), 0, s, x, dy, ws, N, H, W, C, g.CPB, g.LANES, g.chunk);
    HIP_CHECK(hipGetLastError());
    const int n = (KS * KS + 1) * C;
    DL4DS_LAUNCH(dwconv_reduce_kernel, dim3((n + 3) / 4), dim3(256), 0, s, ws, g.nchunks, KS * KS, C, dk, db, accumulate);
    HIP_CHECK(hipGetLastError());
}
