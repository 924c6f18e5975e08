// Discriminator head of dl4ds/models/discriminator.py:72-79: GlobalAveragePooling2D -> Dropout(0.4) ->
// Dense(32, sigmoid) -> Dense(1, sigmoid).  Tiny tensors ((B, C) with C <= a few hundred): one reduction
// kernel for the pooling, single-wave-per-row kernels for the dense layers; deterministic sums.
#include "ops.h"
#include "prof.h"
#include "head.h"
#include <algorithm>

namespace {

// out[n][c] = mean over the HW pixels of x[n, :, :, c]
__global__ void __launch_bounds__(256) gap_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int HW, int C) {
    __shared__ float red[256];
    const int n = blockIdx.x, c = blockIdx.y;
    const float* p = x + (size_t)n * HW * C + c;
    float s = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) s += p[(size_t)i * C];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(size_t)n * C + c] = red[0] / (float)HW;
}

// Coalesced form for feature maps (the discriminator's merge tensor is 2B x 512 x 512 x 16): a block owns a chunk of
// consecutive pixels of one image, thread t the channel pack t % CP on the pixel rows t / CP, t / CP + R, ...; rows are
// combined through LDS, chunks by gap_finish_kernel in a fixed order.  (The per-(n, c) kernel above reads one float per
// 4*C-byte stride: 0.56 ms for that tensor vs 0.03 ms here.)
constexpr int GAP_CHUNKS = 128;
template <int V>
__global__ void __launch_bounds__(256) gap_partial_kernel(const float* __restrict__ x, float* __restrict__ partial, int HW, int C) {
    __shared__ float red[256 * V];
    const int CP = C / V, R = 256 / CP, T = R * CP;
    const int t = threadIdx.x, cp = t % CP, row = t / CP;
    const int n = blockIdx.y;
    const int chunk = (HW + (int)gridDim.x - 1) / (int)gridDim.x;
    const int p0 = blockIdx.x * chunk, p1 = min(p0 + chunk, HW);
    float acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0.f;
    if (t < T) {
        const float* base = x + (size_t)n * HW * C + cp * V;
        for (int p = p0 + row; p < p1; p += R) {
            if constexpr (V == 4) {
                const float4 v = *reinterpret_cast<const float4*>(base + (size_t)p * C);
                acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
            } else {
                acc[0] += base[(size_t)p * C];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) red[t * V + i] = acc[i];
    __syncthreads();
    if (t < CP) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float a = 0.f;
            for (int r = 0; r < R; ++r) a += red[(r * CP + t) * V + i];
            partial[((size_t)n * gridDim.x + blockIdx.x) * C + t * V + i] = a;
        }
    }
}
__global__ void gap_finish_kernel(const float* __restrict__ partial, float* __restrict__ out, int nchunks, int C, int total, float inv) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int n = e / C, c = e - n * C;
    double a = 0.0;
    int k = 0;
    for (; k + 8 <= nchunks; k += 8) {          // (eight loads in flight, added in the order of k)
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = partial[((size_t)n * nchunks + k + u) * C + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += (double)t[u];
    }
    for (; k < nchunks; ++k) a += (double)partial[((size_t)n * nchunks + k) * C + c];
    out[e] = (float)(a * (double)inv);
}

__global__ void gap_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int HW, int C, size_t total,
                               int accumulate) {
    const float inv = 1.f / (float)HW;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const size_t n = e / ((size_t)HW * C);
        const float v = dy[n * C + c] * inv;
        dx[e] = accumulate ? dx[e] + v : v;
    }
}

// y[b][f] = act(b[f] + sum_c x[b][c] w[c][f]); one thread per output
__global__ void dense_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                 float* __restrict__ y, int B, int Cin, int F, int act) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * F) return;
    const int bi = e / F, f = e - bi * F;
    float s = b ? b[f] : 0.f;
    for (int c = 0; c < Cin; ++c) s += x[(size_t)bi * Cin + c] * w[(size_t)c * F + f];
    if (act == ACT_SIGMOID) s = 1.f / (1.f + expf(-s));
    else if (act == ACT_RELU) s = fmaxf(s, 0.f);
    else if (act == ACT_TANH) s = tanhf(s);
    y[e] = s;
}

// dz = dy * act'(y) (in place over dy); then dW, db, dx -- single block, deterministic loops over the batch
__global__ void __launch_bounds__(256) dense_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ y, float* __restrict__ dy,
                                                        float* __restrict__ dx, int acc_dx, float* __restrict__ dw,
                                                        float* __restrict__ db, int acc_dw, int want_dw, int b0, int B,
                                                        int Cin, int F, int act) {
    for (int e = threadIdx.x; e < B * F; e += blockDim.x) {
        const size_t i = (size_t)b0 * F + e;
        float g = dy[i];
        const float yy = y[i];
        if (act == ACT_SIGMOID) g *= yy * (1.f - yy);
        else if (act == ACT_RELU) g = yy > 0.f ? g : 0.f;
        else if (act == ACT_TANH) g *= 1.f - yy * yy;
        dy[i] = g;
    }
    __syncthreads();
    if (want_dw) {
        for (int e = threadIdx.x; e < Cin * F; e += blockDim.x) {
            const int c = e / F, f = e - c * F;
            float s = 0.f;
            for (int bi = 0; bi < B; ++bi) s += x[(size_t)(b0 + bi) * Cin + c] * dy[(size_t)(b0 + bi) * F + f];
            dw[e] = acc_dw ? dw[e] + s : s;
        }
        if (db)
            for (int f = threadIdx.x; f < F; f += blockDim.x) {
                float s = 0.f;
                for (int bi = 0; bi < B; ++bi) s += dy[(size_t)(b0 + bi) * F + f];
                db[f] = acc_dw ? db[f] + s : s;
            }
    }
    if (dx)
        for (int e = threadIdx.x; e < B * Cin; e += blockDim.x) {
            const int bi = e / Cin, c = e - bi * Cin;
            float s = 0.f;
            for (int f = 0; f < F; ++f) s += dy[(size_t)(b0 + bi) * F + f] * w[(size_t)c * F + f];
            const size_t o = (size_t)(b0 + bi) * Cin + c;
            dx[o] = acc_dx ? dx[o] + s : s;
        }
}

// counter-based noise (splitmix-style hash of (seed, index)).  gaussian == 0: keep mask, 1 with probability 1-rate
// (Dropout / SpatialDropout, blocks.py:679-701); gaussian == 1: the multiplicative noise of GaussianDropout,
// N(1, rate / (1 - rate)) by Box-Muller on the two 24-bit halves of the hash.
__global__ void dropout_mask_kernel(float* __restrict__ mask, size_t n, float rate, unsigned long long seed, int gaussian) {
    const float sigma = sqrtf(rate / (1.f - rate));
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (e + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const float u = (float)(z >> 40) * (1.f / 16777216.f);
        if (gaussian) {
            const float u2 = (float)((z >> 16) & 0xFFFFFFull) * (1.f / 16777216.f);
            const float r = sqrtf(-2.f * logf(1.f - u));                      // 1 - u in (0, 1]
            mask[e] = 1.f + sigma * r * cospif(2.f * u2);
        } else {
            mask[e] = (u >= rate) ? 1.f : 0.f;
        }
    }
}
// mask broadcast over `inner` positions: element e of sample block q = e / (inner * C) uses mask[q * C + e % C]
__global__ void dropout_apply_bcast_kernel(const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ y,
                                           size_t n, float scale, int accumulate, int C, size_t inner) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t q = e / (inner * C);
        const float v = x[e] * mask[q * C + e % C] * scale;
        y[e] = accumulate ? y[e] + v : v;
    }
}
__global__ void dropout_apply_kernel(const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ y,
                                     size_t n, float scale, int accumulate) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const float v = x[e] * mask[e] * scale;
        y[e] = accumulate ? y[e] + v : v;
    }
}

inline int ew_blocks(size_t n) { return (int)std::max<size_t>(1, std::min<size_t>(cdivz(n, 256), 4096)); }
}  // namespace

size_t gap_workspace_bytes(int N, int C) { return (size_t)N * GAP_CHUNKS * C * sizeof(float); }

void gap_forward(hipStream_t s, const float* x, float* out, int N, int HW, int C, float* ws, size_t ws_bytes) {
    ProfScope ps(s, "gap_fwd", 0.0, 4.0 * (double)N * HW * C);
    const bool v4 = (C & 3) == 0 && (((uintptr_t)x) & 15) == 0;
    const int CP = v4 ? C / 4 : C;
    if (ws == nullptr || ws_bytes < gap_workspace_bytes(N, C) || CP > 256 || HW < 4096) {
        DL4DS_LAUNCH(gap_fwd_kernel, dim3(N, C), dim3(256), 0, s, x, out, HW, C);
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (v4) DL4DS_LAUNCH(gap_partial_kernel<4>, dim3(GAP_CHUNKS, N), dim3(256), 0, s, x, ws, HW, C);
    else DL4DS_LAUNCH(gap_partial_kernel<1>, dim3(GAP_CHUNKS, N), dim3(256), 0, s, x, ws, HW, C);
    HIP_CHECK(hipGetLastError());
    DL4DS_LAUNCH(gap_finish_kernel, dim3((N * C + 255) / 256), dim3(256), 0, s, ws, out, GAP_CHUNKS, C, N * C, 1.f / (float)HW);
    HIP_CHECK(hipGetLastError());
}
// ... with the ReLU backward of the pooled tensor on the way (mask = that tensor, or null): the residual block in front of the
// discriminator's GlobalAveragePooling used to get its own pass over the HR gradient (0.33 ms at 32 x 512^2 x 16)
__global__ void gap_bwd4_kernel(const float* __restrict__ dy, float4* __restrict__ dx, const float4* __restrict__ mask, int HW, int C4,
                                size_t total4, int accumulate) {
    const float inv = 1.f / (float)HW;
    const size_t per = (size_t)HW * C4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (size_t)gridDim.x * blockDim.x) {
        const size_t n = e / per;
        const int c4 = (int)((e - n * per) % (size_t)C4);
        const float4 g = *reinterpret_cast<const float4*>(dy + n * (size_t)(4 * C4) + 4 * c4);
        float4 v = make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
        if (mask) {
            const float4 m = mask[e];
            v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        }
        if (accumulate) { const float4 o = dx[e]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        dx[e] = v;
    }
}
__global__ void gap_bwd_masked_kernel(const float* __restrict__ dy, float* __restrict__ dx, const float* __restrict__ mask, int HW, int C,
                                      size_t total, int accumulate) {
    const float inv = 1.f / (float)HW;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const size_t n = e / ((size_t)HW * C);
        float v = dy[n * C + c] * inv;
        v = mask[e] > 0.f ? v : 0.f;
        dx[e] = accumulate ? dx[e] + v : v;
    }
}
void gap_backward(hipStream_t s, const float* dy, float* dx, int N, int HW, int C, int accumulate, const float* mask) {
    const size_t total = (size_t)N * HW * C;
    ProfScope ps(s, "gap_bwd", 0.0, 4.0 * (double)total * (1 + (mask ? 1 : 0) + (accumulate ? 1 : 0)));
    const bool v4 = (C & 3) == 0 && ((((uintptr_t)dy) | ((uintptr_t)dx) | ((uintptr_t)mask)) & 15) == 0;
    if (v4)
        DL4DS_LAUNCH(gap_bwd4_kernel, dim3(ew_blocks(total / 4)), dim3(256), 0, s, dy, reinterpret_cast<float4*>(dx),
                           reinterpret_cast<const float4*>(mask), HW, C / 4, total / 4, accumulate);
    else if (mask)
        DL4DS_LAUNCH(gap_bwd_masked_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, dy, dx, mask, HW, C, total, accumulate);
    else
        DL4DS_LAUNCH(gap_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, dy, dx, HW, C, total, accumulate);
    HIP_CHECK(hipGetLastError());
}
void dense_forward(hipStream_t s, const float* x, const float* w, const float* b, float* y, int B, int Cin, int F, int act) {
    DL4DS_LAUNCH(dense_fwd_kernel, dim3(cdiv(B * F, 256)), dim3(256), 0, s, x, w, b, y, B, Cin, F, act);
    HIP_CHECK(hipGetLastError());
}
void dense_backward(hipStream_t s, const float* x, const float* w, const float* y, float* dy, float* dx, int acc_dx,
                    float* dw, float* db, int acc_dw, int want_dw, int b0, int B, int Cin, int F, int act) {
    DL4DS_LAUNCH(dense_bwd_kernel, dim3(1), dim3(256), 0, s, x, w, y, dy, dx, acc_dx, dw, db, acc_dw, want_dw, b0, B,
                       Cin, F, act);
    HIP_CHECK(hipGetLastError());
}
void dropout_make_mask(hipStream_t s, float* mask, size_t n, float rate, unsigned long long seed, int gaussian) {
    DL4DS_LAUNCH(dropout_mask_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, mask, n, rate, seed, gaussian);
    HIP_CHECK(hipGetLastError());
}
void dropout_apply_bcast(hipStream_t s, const float* x, const float* mask, float* y, size_t n, float scale, int accumulate, int C,
                         size_t inner) {
    DL4DS_LAUNCH(dropout_apply_bcast_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, mask, y, n, scale, accumulate, C, inner);
    HIP_CHECK(hipGetLastError());
}
void dropout_apply(hipStream_t s, const float* x, const float* mask, float* y, size_t n, float scale, int accumulate) {
    DL4DS_LAUNCH(dropout_apply_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, mask, y, n, scale, accumulate);
    HIP_CHECK(hipGetLastError());
}
