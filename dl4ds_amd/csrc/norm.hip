// dl4ds_amd -- LayerNormalization / BatchNormalization of the block variants ("next" row f2).
//
// dl4ds/models/blocks.py:63-71 (ConvBlock), :151-159 (ConvNextBlock), :293-309 (TransitionBlock) instantiate the Keras
// layers with their defaults: both normalise over the LAST axis of an NHWC tensor,
//   LayerNormalization(axis=-1, epsilon=1e-3 | 1e-6 in ConvNextBlock): per pixel over the C channels,
//   BatchNormalization(axis=-1, momentum=0.99, epsilon=1e-3): per channel over N*H*W, batch statistics when training
//     (moving averages updated: mean with the batch mean, variance with the Bessel-corrected batch variance, as the fused
//     Keras kernel does), moving statistics at inference.
// In every block the normalisation is followed by the activation, so an optional ReLU is fused into the store and its
// mask into the backward read.  All kernels are HBM streaming: x is read once (twice for BN: statistics, then apply),
// y written once; per-channel sums go through fixed-order per-block partials, so results are bit-reproducible.
#include "ops.h"
#include "prof.h"
#include <algorithm>

namespace {

constexpr int NORM_THREADS = 256;
constexpr int NORM_KMAX = 4;           // channel slots per lane (LayerNorm): C <= 64 * KMAX * V
constexpr int NORM_MAX_BLOCKS = 1024;

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
__device__ __forceinline__ float group_sum(float v, int L) {
    for (int m = L >> 1; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// sum over b < nb of partial[b * stride + col], computed by one full wavefront (lane-strided loads, then a fixed
// shuffle tree: the order of additions depends only on nb, so the result is reproducible); every lane returns it
__device__ __forceinline__ double wave_col_sum(const float* __restrict__ partial, int nb, size_t stride, int col) {
    double t = 0.0;
    for (int b = threadIdx.x & 63; b < nb; b += 64) t += (double)partial[(size_t)b * stride + col];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    return t;
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// L lanes (power of two <= 64) share one pixel; lane j owns the channel packs j, j+L, j+2L, ... (V floats each).
template <int V>
__global__ void __launch_bounds__(NORM_THREADS) ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ y,
                                                             size_t npix, int C, int L, float eps, int relu) {
    const int CP = C / V;
    const int j = threadIdx.x & (L - 1);
    const size_t gpb = NORM_THREADS / L;                         // pixel groups per block
    const float inv_c = 1.f / (float)C;
    for (size_t p = (size_t)blockIdx.x * gpb + threadIdx.x / L; p < npix; p += (size_t)gridDim.x * gpb) {
        const float* xp = x + p * C;
        float s = 0.f;
        for (int k = j; k < CP; k += L) {
            const Pk<V> a = ldp<V>(xp + k * V);
#pragma unroll
            for (int i = 0; i < V; ++i) s += a.v[i];
        }
        const float mu = group_sum(s, L) * inv_c;
        float q = 0.f;
        for (int k = j; k < CP; k += L) {
            const Pk<V> a = ldp<V>(xp + k * V);
#pragma unroll
            for (int i = 0; i < V; ++i) q += (a.v[i] - mu) * (a.v[i] - mu);
        }
        const float r = rsqrtf(group_sum(q, L) * inv_c + eps);
        for (int k = j; k < CP; k += L) {
            const Pk<V> a = ldp<V>(xp + k * V), g = ldp<V>(gamma + k * V), b = ldp<V>(beta + k * V);
            Pk<V> o;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float t = (a.v[i] - mu) * r * g.v[i] + b.v[i];
                o.v[i] = relu ? fmaxf(t, 0.f) : t;
            }
            stp<V>(y + p * C + k * V, o);
        }
    }
}

// dx = r * (g - mean(g) - xhat * mean(g * xhat)),  g = dy_eff * gamma;  per-block partial sums of dgamma / dbeta
template <int V>
__global__ void __launch_bounds__(NORM_THREADS) ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ dy, const float* __restrict__ gamma,
                                                             float* __restrict__ dx, int acc_dx, float* __restrict__ partial,
                                                             size_t npix, int C, int L, float eps, int relu) {
    __shared__ float red[NORM_THREADS * V];
    const int CP = C / V;
    const int j = threadIdx.x & (L - 1);
    const size_t gpb = NORM_THREADS / L;
    const float inv_c = 1.f / (float)C;
    float dg[NORM_KMAX][V], db[NORM_KMAX][V];
#pragma unroll
    for (int s = 0; s < NORM_KMAX; ++s)
#pragma unroll
        for (int i = 0; i < V; ++i) dg[s][i] = db[s][i] = 0.f;
    for (size_t p = (size_t)blockIdx.x * gpb + threadIdx.x / L; p < npix; p += (size_t)gridDim.x * gpb) {
        const float* xp = x + p * C;
        float s = 0.f;
        for (int k = j; k < CP; k += L) {
            const Pk<V> a = ldp<V>(xp + k * V);
#pragma unroll
            for (int i = 0; i < V; ++i) s += a.v[i];
        }
        const float mu = group_sum(s, L) * inv_c;
        float q = 0.f;
        for (int k = j; k < CP; k += L) {
            const Pk<V> a = ldp<V>(xp + k * V);
#pragma unroll
            for (int i = 0; i < V; ++i) q += (a.v[i] - mu) * (a.v[i] - mu);
        }
        const float r = rsqrtf(group_sum(q, L) * inv_c + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int sl = 0; sl < NORM_KMAX; ++sl) {
            const int k = j + sl * L;
            if (k < CP) {
                const Pk<V> a = ldp<V>(xp + k * V), d = ldp<V>(dy + p * C + k * V), g = ldp<V>(gamma + k * V);
                Pk<V> o;
                if (relu) o = ldp<V>(y + p * C + k * V);
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const float de = (relu && !(o.v[i] > 0.f)) ? 0.f : d.v[i];
                    const float xh = (a.v[i] - mu) * r;
                    dg[sl][i] += de * xh;
                    db[sl][i] += de;
                    s1 += de * g.v[i];
                    s2 += de * g.v[i] * xh;
                }
            }
        }
        const float m1 = group_sum(s1, L) * inv_c, m2 = group_sum(s2, L) * inv_c;
        if (dx) {
#pragma unroll
            for (int sl = 0; sl < NORM_KMAX; ++sl) {
                const int k = j + sl * L;
                if (k < CP) {
                    const Pk<V> a = ldp<V>(xp + k * V), d = ldp<V>(dy + p * C + k * V), g = ldp<V>(gamma + k * V);
                    Pk<V> o, w;
                    if (relu) o = ldp<V>(y + p * C + k * V);
                    if (acc_dx) w = ldp<V>(dx + p * C + k * V);
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        const float de = (relu && !(o.v[i] > 0.f)) ? 0.f : d.v[i];
                        const float xh = (a.v[i] - mu) * r;
                        const float t = r * (de * g.v[i] - m1 - xh * m2);
                        w.v[i] = acc_dx ? w.v[i] + t : t;
                    }
                    stp<V>(dx + p * C + k * V, w);
                }
            }
        }
    }
    // block partials: lanes with the same j (stride L) hold the same channels
    float* out = partial + (size_t)blockIdx.x * 2 * C;
    for (int which = 0; which < 2; ++which) {
#pragma unroll
        for (int sl = 0; sl < NORM_KMAX; ++sl) {
            const int k = j + sl * L;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < V; ++i) red[threadIdx.x * V + i] = which ? db[sl][i] : dg[sl][i];
            __syncthreads();
            if (threadIdx.x < L && k < CP) {
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    float t = 0.f;
                    for (int g = 0; g < NORM_THREADS / L; ++g) t += red[(g * L + j) * V + i];
                    out[which * C + k * V + i] = t;
                }
            }
        }
    }
}

// dst[c] (+)= sum over blocks of partial[b][c], c in [0, n); one wavefront per output
__global__ void __launch_bounds__(256) partial_reduce_kernel(const float* __restrict__ partial, int nb, int n, float* __restrict__ d0,
                                                             float* __restrict__ d1, int half, int acc) {
    if (d0 == nullptr) return;
    for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < n; c += gridDim.x * 4) {
        const double t = wave_col_sum(partial, nb, (size_t)n, c);
        if ((threadIdx.x & 63) == 0) {
            float* d = c < half ? d0 + c : d1 + (c - half);
            *d = acc ? *d + (float)t : (float)t;
        }
    }
}

// ------------------------------------------------------------------------------------------------ BatchNorm
// Thread t < T = R * CP owns channel pack t % CP and walks pixels t / CP, t / CP + R, ... of the block's chunk.
struct BnGeom { int CP, R, T, nb; size_t chunk; };

inline BnGeom bn_geom(size_t npix, int C, int V) {
    BnGeom g;
    g.CP = C / V;
    g.R = NORM_THREADS / g.CP;
    g.T = g.R * g.CP;
    const size_t want = std::max<size_t>(1, std::min<size_t>(NORM_MAX_BLOCKS, npix / (size_t)(g.R * 8) + 1));
    g.chunk = (npix + want - 1) / want;
    g.nb = (int)((npix + g.chunk - 1) / g.chunk);
    return g;
}

// partial[b][0][c] = sum (x - pilot_c), partial[b][1][c] = sum (x - pilot_c)^2, pilot_c = x[0][c] (shifted sums: no
// cancellation when |mean| >> std)
template <int V>
__global__ void __launch_bounds__(NORM_THREADS) bn_stats_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                               size_t npix, int C, BnGeom g) {
    __shared__ float red[2][NORM_THREADS * V];
    const int t = threadIdx.x;
    const int cp = t % g.CP, row = t / g.CP;
    float s1[V], s2[V];
#pragma unroll
    for (int i = 0; i < V; ++i) s1[i] = s2[i] = 0.f;
    if (t < g.T) {
        const Pk<V> pilot = ldp<V>(x + cp * V);
        const size_t p0 = (size_t)blockIdx.x * g.chunk, p1 = min(p0 + g.chunk, npix);
        for (size_t p = p0 + row; p < p1; p += g.R) {
            const Pk<V> a = ldp<V>(x + p * C + cp * V);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float d = a.v[i] - pilot.v[i];
                s1[i] += d;
                s2[i] += d * d;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) { red[0][t * V + i] = s1[i]; red[1][t * V + i] = s2[i]; }
    __syncthreads();
    if (t < g.CP) {
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int i = 0; i < V; ++i) {
                float a = 0.f;
                for (int r = 0; r < g.R; ++r) a += red[w][(r * g.CP + t) * V + i];
                partial[((size_t)blockIdx.x * 2 + w) * C + t * V + i] = a;
            }
    }
}

// stats[0][c] = a_c (scale), stats[1][c] = b_c (shift) of y = a*x + b;  saved[0][c] = mean, saved[1][c] = invstd
__global__ void bn_finalize_kernel(const float* __restrict__ x, const float* __restrict__ partial, int nb, size_t npix, int C,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ mov_mean,
                                   float* __restrict__ mov_var, float eps, float momentum, int training,
                                   float* __restrict__ stats, float* __restrict__ saved) {
    for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < C; c += gridDim.x * 4) {      // one wavefront per channel
        float mean, invstd;
        if (training) {
            const double s1 = wave_col_sum(partial, nb, (size_t)2 * C, c);
            const double s2 = wave_col_sum(partial, nb, (size_t)2 * C, C + c);
            if (threadIdx.x & 63) continue;
            const double n = (double)npix, m = s1 / n;
            const double var = fmax(s2 / n - m * m, 0.0);
            mean = (float)((double)x[c] + m);
            invstd = (float)(1.0 / sqrt(var + (double)eps));
            const double unbiased = npix > 1 ? var * n / (n - 1.0) : var;
            mov_mean[c] = mov_mean[c] * momentum + mean * (1.f - momentum);
            mov_var[c] = mov_var[c] * momentum + (float)unbiased * (1.f - momentum);
            saved[c] = mean;
            saved[C + c] = invstd;
        } else {
            if (threadIdx.x & 63) continue;
            mean = mov_mean[c];
            invstd = rsqrtf(mov_var[c] + eps);
        }
        const float a = gamma[c] * invstd;
        stats[c] = a;
        stats[C + c] = beta[c] - mean * a;
    }
}

template <int V>
__global__ void __launch_bounds__(NORM_THREADS) bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                               float* __restrict__ y, size_t npix, int C, BnGeom g, int relu) {
    const int t = threadIdx.x;
    if (t >= g.T) return;
    const int cp = t % g.CP, row = t / g.CP;
    const Pk<V> a = ldp<V>(stats + cp * V), b = ldp<V>(stats + C + cp * V);
    const size_t p0 = (size_t)blockIdx.x * g.chunk, p1 = min(p0 + g.chunk, npix);
    for (size_t p = p0 + row; p < p1; p += g.R) {
        const Pk<V> v = ldp<V>(x + p * C + cp * V);
        Pk<V> o;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float u = v.v[i] * a.v[i] + b.v[i];
            o.v[i] = relu ? fmaxf(u, 0.f) : u;
        }
        stp<V>(y + p * C + cp * V, o);
    }
}

// partial[b][0][c] = sum dy_eff * xhat, partial[b][1][c] = sum dy_eff
template <int V>
__global__ void __launch_bounds__(NORM_THREADS) bn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                    const float* __restrict__ dy, const float* __restrict__ saved,
                                                                    float* __restrict__ partial, size_t npix, int C, BnGeom g,
                                                                    int relu) {
    __shared__ float red[2][NORM_THREADS * V];
    const int t = threadIdx.x;
    const int cp = t % g.CP, row = t / g.CP;
    float s1[V], s2[V];
#pragma unroll
    for (int i = 0; i < V; ++i) s1[i] = s2[i] = 0.f;
    if (t < g.T) {
        const Pk<V> mean = ldp<V>(saved + cp * V), inv = ldp<V>(saved + C + cp * V);
        const size_t p0 = (size_t)blockIdx.x * g.chunk, p1 = min(p0 + g.chunk, npix);
        for (size_t p = p0 + row; p < p1; p += g.R) {
            const Pk<V> a = ldp<V>(x + p * C + cp * V), d = ldp<V>(dy + p * C + cp * V);
            Pk<V> o;
            if (relu) o = ldp<V>(y + p * C + cp * V);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float de = (relu && !(o.v[i] > 0.f)) ? 0.f : d.v[i];
                s1[i] += de * (a.v[i] - mean.v[i]) * inv.v[i];
                s2[i] += de;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) { red[0][t * V + i] = s1[i]; red[1][t * V + i] = s2[i]; }
    __syncthreads();
    if (t < g.CP) {
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int i = 0; i < V; ++i) {
                float a = 0.f;
                for (int r = 0; r < g.R; ++r) a += red[w][(r * g.CP + t) * V + i];
                partial[((size_t)blockIdx.x * 2 + w) * C + t * V + i] = a;
            }
    }
}

// sums[0][c] = sum dy*xhat / n, sums[1][c] = sum dy / n;  dgamma / dbeta written or accumulated
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nb, size_t npix, int C, float* __restrict__ sums,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int acc) {
    for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < C; c += gridDim.x * 4) {      // one wavefront per channel
        const double s1 = wave_col_sum(partial, nb, (size_t)2 * C, c);
        const double s2 = wave_col_sum(partial, nb, (size_t)2 * C, C + c);
        if (threadIdx.x & 63) continue;
        sums[c] = (float)(s1 / (double)npix);
        sums[C + c] = (float)(s2 / (double)npix);
        if (dgamma) {
            dgamma[c] = acc ? dgamma[c] + (float)s1 : (float)s1;
            dbeta[c] = acc ? dbeta[c] + (float)s2 : (float)s2;
        }
    }
}

template <int V>
__global__ void __launch_bounds__(NORM_THREADS) bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                   const float* __restrict__ dy, const float* __restrict__ gamma,
                                                                   const float* __restrict__ saved, const float* __restrict__ sums,
                                                                   float* __restrict__ dx, int acc_dx, size_t npix, int C,
                                                                   BnGeom g, int relu) {
    const int t = threadIdx.x;
    if (t >= g.T) return;
    const int cp = t % g.CP, row = t / g.CP;
    const Pk<V> mean = ldp<V>(saved + cp * V), inv = ldp<V>(saved + C + cp * V), gm = ldp<V>(gamma + cp * V);
    const Pk<V> c1 = ldp<V>(sums + cp * V), c2 = ldp<V>(sums + C + cp * V);
    const size_t p0 = (size_t)blockIdx.x * g.chunk, p1 = min(p0 + g.chunk, npix);
    for (size_t p = p0 + row; p < p1; p += g.R) {
        const Pk<V> a = ldp<V>(x + p * C + cp * V), d = ldp<V>(dy + p * C + cp * V);
        Pk<V> o, w;
        if (relu) o = ldp<V>(y + p * C + cp * V);
        if (acc_dx) w = ldp<V>(dx + p * C + cp * V);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float de = (relu && !(o.v[i] > 0.f)) ? 0.f : d.v[i];
            const float xh = (a.v[i] - mean.v[i]) * inv.v[i];
            const float u = gm.v[i] * inv.v[i] * (de - c2.v[i] - xh * c1.v[i]);
            w.v[i] = acc_dx ? w.v[i] + u : u;
        }
        stp<V>(dx + p * C + cp * V, w);
    }
}

inline bool vec_ok(int C, std::initializer_list<const void*> ptrs) {
    if (C & 3) return false;
    for (const void* p : ptrs)
        if (p && ((uintptr_t)p & 15)) return false;
    return true;
}
inline int ln_lanes(int CP) {
    int L = 1;
    while (L < CP && L < 64) L <<= 1;
    return L;
}
inline int ln_blocks(size_t npix, int L) {
    const size_t gpb = NORM_THREADS / L;
    return (int)std::max<size_t>(1, std::min<size_t>(NORM_MAX_BLOCKS, (npix + gpb - 1) / gpb));
}

}  // namespace

size_t norm_workspace_bytes(int C) {
    // per-block partials [blocks][2][C] + two small [2][C] tables (scale/shift or the backward means)
    return ((size_t)NORM_MAX_BLOCKS * 2 * C + 4 * (size_t)C) * sizeof(float);
}

static void check_channels(int C, int V, const char* what) {
    DL4DS_REQUIRE(C >= 1 && C / V <= 64 * NORM_KMAX && C / V <= NORM_THREADS,
                  std::string(what) + ": channel count not supported (at most 256 channels, or 1024 when a multiple of 4)");
}

void layernorm_forward(hipStream_t s, const float* x, const float* gamma, const float* beta, float* y, size_t npix, int C,
                       float eps, int relu) {
    if (npix == 0) return;
    ProfScope ps(s, "layernorm_fwd", 0.0, 8.0 * (double)npix * C);
    const bool v4 = vec_ok(C, {x, gamma, beta, y});
    check_channels(C, v4 ? 4 : 1, "layernorm");
    const int L = ln_lanes(v4 ? C / 4 : C);
    if (v4) DL4DS_LAUNCH(ln_fwd_kernel<4>, dim3(ln_blocks(npix, L)), dim3(NORM_THREADS), 0, s, x, gamma, beta, y, npix, C, L, eps, relu);
    else DL4DS_LAUNCH(ln_fwd_kernel<1>, dim3(ln_blocks(npix, L)), dim3(NORM_THREADS), 0, s, x, gamma, beta, y, npix, C, L, eps, relu);
    HIP_CHECK(hipGetLastError());
}

void layernorm_backward(hipStream_t s, const float* x, const float* y, const float* dy, const float* gamma, float* dx, int acc_dx,
                        float* dgamma, float* dbeta, int acc_dw, size_t npix, int C, float eps, int relu, float* ws,
                        size_t ws_bytes) {
    if (npix == 0) return;
    DL4DS_REQUIRE(ws_bytes >= norm_workspace_bytes(C), "layernorm: workspace too small");
    ProfScope ps(s, "layernorm_bwd", 0.0, 4.0 * (double)npix * C * (dx ? 3 + (relu ? 1 : 0) + (acc_dx ? 1 : 0) : 2));
    const bool v4 = vec_ok(C, {x, y, dy, gamma, dx});
    check_channels(C, v4 ? 4 : 1, "layernorm");
    const int L = ln_lanes(v4 ? C / 4 : C);
    const int nb = ln_blocks(npix, L);
    if (v4) DL4DS_LAUNCH(ln_bwd_kernel<4>, dim3(nb), dim3(NORM_THREADS), 0, s, x, y, dy, gamma, dx, acc_dx, ws, npix, C, L, eps, relu);
    else DL4DS_LAUNCH(ln_bwd_kernel<1>, dim3(nb), dim3(NORM_THREADS), 0, s, x, y, dy, gamma, dx, acc_dx, ws, npix, C, L, eps, relu);
    HIP_CHECK(hipGetLastError());
    DL4DS_LAUNCH(partial_reduce_kernel, dim3((2 * C + 3) / 4), dim3(256), 0, s, ws, nb, 2 * C, dgamma, dbeta, C, acc_dw);
    HIP_CHECK(hipGetLastError());
}

void batchnorm_forward(hipStream_t s, const float* x, const float* gamma, const float* beta, float* mov_mean, float* mov_var,
                       float* y, float* saved, size_t npix, int C, float eps, float momentum, int training, int relu, float* ws,
                       size_t ws_bytes) {
    if (npix == 0) return;
    DL4DS_REQUIRE(ws_bytes >= norm_workspace_bytes(C), "batchnorm: workspace too small");
    ProfScope ps(s, "batchnorm_fwd", 0.0, 4.0 * (double)npix * C * (training ? 3 : 2));
    const bool v4 = vec_ok(C, {x, y, ws});
    const int V = v4 ? 4 : 1;
    check_channels(C, V, "batchnorm");
    const BnGeom g = bn_geom(npix, C, V);
    float* partial = ws;
    float* stats = ws + (size_t)NORM_MAX_BLOCKS * 2 * C;
    if (training) {
        if (v4) DL4DS_LAUNCH(bn_stats_kernel<4>, dim3(g.nb), dim3(NORM_THREADS), 0, s, x, partial, npix, C, g);
        else DL4DS_LAUNCH(bn_stats_kernel<1>, dim3(g.nb), dim3(NORM_THREADS), 0, s, x, partial, npix, C, g);
        HIP_CHECK(hipGetLastError());
    }
    DL4DS_LAUNCH(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, s, x, partial, g.nb, npix, C, gamma, beta, mov_mean,
                       mov_var, eps, momentum, training, stats, saved);
    HIP_CHECK(hipGetLastError());
    if (v4) DL4DS_LAUNCH(bn_apply_kernel<4>, dim3(g.nb), dim3(NORM_THREADS), 0, s, x, stats, y, npix, C, g, relu);
    else DL4DS_LAUNCH(bn_apply_kernel<1>, dim3(g.nb), dim3(NORM_THREADS), 0, s, x, stats, y, npix, C, g, relu);
    HIP_CHECK(hipGetLastError());
}

void batchnorm_backward(hipStream_t s, const float* x, const float* y, const float* dy, const float* gamma, const float* saved,
                        float* dx, int acc_dx, float* dgamma, float* dbeta, int acc_dw, size_t npix, int C, int relu, float* ws,
                        size_t ws_bytes) {
    if (npix == 0) return;
    DL4DS_REQUIRE(ws_bytes >= norm_workspace_bytes(C), "batchnorm: workspace too small");
    ProfScope ps(s, "batchnorm_bwd", 0.0, 4.0 * (double)npix * C * (dx ? 5 : 2));
    const bool v4 = vec_ok(C, {x, y, dy, dx, ws, saved, gamma});
    const int V = v4 ? 4 : 1;
    check_channels(C, V, "batchnorm");
    const BnGeom g = bn_geom(npix, C, V);
    float* partial = ws;
    float* sums = ws + (size_t)NORM_MAX_BLOCKS * 2 * C;
    if (v4) DL4DS_LAUNCH(bn_bwd_reduce_kernel<4>, dim3(g.nb), dim3(NORM_THREADS), 0, s, x, y, dy, saved, partial, npix, C, g, relu);
    else DL4DS_LAUNCH(bn_bwd_reduce_kernel<1>, dim3(g.nb), dim3(NORM_THREADS), 0, s, x, y, dy, saved, partial, npix, C, g, relu);
    HIP_CHECK(hipGetLastError());
    DL4DS_LAUNCH(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, s, partial, g.nb, npix, C, sums, dgamma, dbeta, acc_dw);
    HIP_CHECK(hipGetLastError());
    if (dx) {
        if (v4) DL4DS_LAUNCH(bn_bwd_apply_kernel<4>, dim3(g.nb), dim3(NORM_THREADS), 0, s, x, y, dy, gamma, saved, sums, dx, acc_dx, npix, C, g, relu);
        else DL4DS_LAUNCH(bn_bwd_apply_kernel<1>, dim3(g.nb), dim3(NORM_THREADS), 0, s, x, y, dy, gamma, saved, sums, dx, acc_dx, npix, C, g, relu);
        HIP_CHECK(hipGetLastError());
    }
}
