// ChannelAttention2D (squeeze-excite, r=4) for gfx950 -- dl4ds/models/blocks.py:537-593:
//     y = x * sigmoid(W2^T relu(W1^T mean_{axes 1,2}(x) + b1) + b2)
// The tensor is viewed as [G][R][P*C]: the mean runs over R, the tiny MLP over the last C channels of
// each of the G*P instances.  4-D (B,H,W,C): G=B, R=H*W, P=1.  5-D (B,T,H,W,C) as the reference reaches
// it through spt_postups.py:153-154 reduces axes [1,2] = (T,H): G=B, R=T*H, P=W.
// Kernels: wave/LDS column-sum reduction (HBM-bound, one read of x), one-block MLP forward/backward,
// broadcast-scale.  Backward needs sum_r(dy*x) -- a second column-sum with a fused product.
#include "ops.h"
#include "prof.h"
#include <algorithm>

namespace {

// partial[(g*nb + blockIdx.x)*Q + q] = sum over this block's rows of a[g][r][q] (* b[g][r][q])
template <int TX>
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     float* __restrict__ partial, int R, int Q) {
    constexpr int TY = 256 / TX;
    __shared__ float red[TY][TX + 1];
    const int tx = threadIdx.x % TX, tyi = threadIdx.x / TX;
    const int q = blockIdx.y * TX + tx;
    const int g = blockIdx.z;
    const float* ap = a + (size_t)g * R * Q;
    const float* bp = b ? b + (size_t)g * R * Q : nullptr;
    float sum = 0.f;
    if (q < Q) {
        for (int r = blockIdx.x * TY + tyi; r < R; r += gridDim.x * TY) {
            float v = ap[(size_t)r * Q + q];
            if (bp) v *= bp[(size_t)r * Q + q];
            sum += v;
        }
    }
    red[tyi][tx] = sum;
    __syncthreads();
    if (tyi == 0 && q < Q) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < TY; ++k) s += red[k][tx];
        partial[((size_t)g * gridDim.x + blockIdx.x) * Q + q] = s;
    }
}

// the same with four consecutive columns per thread (16-byte loads; round 5: the scalar form moved 2.6 TB/s on cfg4's 5-D attention,
// 268 MB in 103 us): block = 64 column quads x 4 row phases, 16 rows per thread in flight as two batches of eight
__global__ void __launch_bounds__(256) colsum4_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ partial,
                                                      int R, int Q) {
    constexpr int TX = 64, TY = 4;
    __shared__ float4 red[TY][TX];
    const int tx = threadIdx.x % TX, tyi = threadIdx.x / TX;
    const int q = (blockIdx.y * TX + tx) * 4;
    const int g = blockIdx.z;
    const float* ap = a + (size_t)g * R * Q;
    const float* bp = b ? b + (size_t)g * R * Q : nullptr;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < Q) {
        const int step = gridDim.x * TY;
        int r = blockIdx.x * TY + tyi;
        for (; r + 7 * step < R; r += 8 * step) {
            float4 v[8], w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(ap + (size_t)(r + u * step) * Q + q);
            if (bp) {
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] = *reinterpret_cast<const float4*>(bp + (size_t)(r + u * step) * Q + q);
#pragma unroll
                for (int u = 0; u < 8; ++u) { v[u].x *= w[u].x; v[u].y *= w[u].y; v[u].z *= w[u].z; v[u].w *= w[u].w; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { sum.x += v[u].x; sum.y += v[u].y; sum.z += v[u].z; sum.w += v[u].w; }
        }
        for (; r < R; r += step) {
            float4 v = *reinterpret_cast<const float4*>(ap + (size_t)r * Q + q);
            if (bp) { const float4 w = *reinterpret_cast<const float4*>(bp + (size_t)r * Q + q); v.x *= w.x; v.y *= w.y; v.z *= w.z; v.w *= w.w; }
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
    }
    red[tyi][tx] = sum;
    __syncthreads();
    if (tyi == 0 && q < Q) {
        float4 s = red[0][tx];
#pragma unroll
        for (int k = 1; k < TY; ++k) { s.x += red[k][tx].x; s.y += red[k][tx].y; s.z += red[k][tx].z; s.w += red[k][tx].w; }
        *reinterpret_cast<float4*>(partial + ((size_t)g * gridDim.x + blockIdx.x) * Q + q) = s;
    }
}

// out[g*Q+q] = scale * sum_k partial[(g*nb+k)*Q+q]
__global__ void colsum_finish_kernel(const float* __restrict__ partial, float* __restrict__ out, int nb, int Q,
                                     int GQ, float scale) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= GQ) return;
    const int g = e / Q, q = e - g * Q;
    // eight independent partial sums keep eight loads in flight (a single dependent chain took 40 us for nb = 256);
    // fixed association order -> deterministic
    const float* p = partial + (size_t)g * nb * Q + q;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= nb; k += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] += p[(size_t)(k + u) * Q];
    }
    for (; k < nb; ++k) s[0] += p[(size_t)k * Q];
    out[e] = (((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]))) * scale;
}

// mean[g*C + c] = scale * sum_k partial[(g*tiles + k)*8 + c]: the per-tile channel sums the narrow pair convolution
// wrote from its epilogue (ConvEpilogue::pool), same eight-way fixed association order as colsum_finish_kernel
// One wave per instance g: lane (u = lane >> 3, c = lane & 7) sums the tiles k = u (mod 8) -- the eight partial sums one thread
// used to keep (512 dependent loads per thread, 32 us at 64 x 512^2) -- and three shuffles combine them in the same order.
__global__ void __launch_bounds__(256) pool_finish_kernel(const float* __restrict__ partial, float* __restrict__ out, int tiles, int C,
                                                          int G, float scale) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;                                   // (whole waves)
    const int lane = threadIdx.x & 63, u = lane >> 3, c = lane & 7;
    const float* p = partial + (size_t)g * tiles * 8 + c;
    const int full = tiles & ~7;
    float s = 0.f;
#pragma unroll 8
    for (int k = u; k < full; k += 8) s += p[(size_t)k * 8];                // (unrolled: the loads of eight steps in flight, the additions in order)
    if (u == 0) for (int k = full; k < tiles; ++k) s += p[(size_t)k * 8];
    s += __shfl_xor(s, 8, 64);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (u == 0 && c < C) out[(size_t)g * C + c] = s * scale;
}

// one thread per (instance, c)
__global__ void chatt_mlp_fwd_kernel(const float* __restrict__ mean, const float* __restrict__ w1,
                                     const float* __restrict__ b1, const float* __restrict__ w2,
                                     const float* __restrict__ b2, float* __restrict__ hidden,
                                     float* __restrict__ scale, int ninst, int C, int Cr) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ninst * C) return;
    const int inst = e / C, c = e - inst * C;
    const float* m = mean + (size_t)inst * C;
    float z = b2[c];
    for (int j = 0; j < Cr; ++j) {
        float h = b1[j];
        for (int k = 0; k < C; ++k) h += m[k] * w1[k * Cr + j];
        h = fmaxf(h, 0.f);
        if (c == 0) hidden[(size_t)inst * Cr + j] = h;
        z += h * w2[j * C + c];
    }
    scale[e] = 1.f / (1.f + expf(-z));
}

// y = x * scale[g, q] (forward) and dx (+)= dy * scale[g, q] + dmean[g, q] (backward) over [G][R][Q]: blockIdx.y = g, so the
// channel index is one 32-bit remainder per thread and advances by a constant per grid stride (the flat kernels spent two
// 64-bit divisions per element: 2.2 TB/s); four consecutive elements per thread where R*Q allows it
template <int V, bool BWD>
__global__ void chatt_apply_kernel(const float* __restrict__ a, const float* __restrict__ scale, const float* __restrict__ dmean,
                                   float* __restrict__ out, unsigned RQv, int Q, int accumulate) {
    const size_t base = (size_t)blockIdx.y * RQv * V;
    const float* sc = scale + (size_t)blockIdx.y * Q;
    const float* dm = BWD ? dmean + (size_t)blockIdx.y * Q : nullptr;
    const unsigned stride = gridDim.x * blockDim.x;
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    int q = (int)(((unsigned long long)i * V) % (unsigned)Q);
    const int dq = (int)(((unsigned long long)stride * V) % (unsigned)Q);
    for (; i < RQv; i += stride) {
        float v[V], o[V];
        if constexpr (V == 4) {
            const float4 t = reinterpret_cast<const float4*>(a + base)[i];
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            if (BWD && accumulate) { const float4 p = reinterpret_cast<const float4*>(out + base)[i]; o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = p.w; }
        } else {
            v[0] = a[base + i];
            if (BWD && accumulate) o[0] = out[base + i];
        }
        int qq = q;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float r = v[j] * sc[qq];
            if (BWD) { r += dm[qq]; if (accumulate) r += o[j]; }
            v[j] = r;
            qq = (qq + 1 == Q) ? 0 : qq + 1;
        }
        if constexpr (V == 4) reinterpret_cast<float4*>(out + base)[i] = make_float4(v[0], v[1], v[2], v[3]);
        else out[base + i] = v[0];
        q += dq;
        if (q >= Q) q -= Q;
    }
}

template <bool BWD>
void chatt_apply(hipStream_t s, const float* a, const float* scale, const float* dmean, float* out, int G, int R, int Q, int accumulate) {
    const size_t RQ = (size_t)R * Q;
    DL4DS_REQUIRE(RQ < (1ull << 32), "chatt: one sample has 2^32 elements or more");
    const bool v4 = (RQ & 3) == 0 && ((((uintptr_t)a) | ((uintptr_t)out)) & 15) == 0;
    const size_t n = v4 ? RQ / 4 : RQ;
    const unsigned bx = (unsigned)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, std::max<size_t>(1, 8192 / (size_t)G)));
    if (v4) DL4DS_LAUNCH((chatt_apply_kernel<4, BWD>), dim3(bx, (unsigned)G), dim3(256), 0, s, a, scale, dmean, out, (unsigned)n, Q, accumulate);
    else DL4DS_LAUNCH((chatt_apply_kernel<1, BWD>), dim3(bx, (unsigned)G), dim3(256), 0, s, a, scale, dmean, out, (unsigned)n, Q, accumulate);
    HIP_CHECK(hipGetLastError());
}

// phase A per instance (one thread each), phase B deterministic parameter-gradient sums (one wavefront per output element,
// lanes stride the instances, fixed shuffle tree).  Two launches over as many blocks as there is work: as ONE block (round 2)
// the 5-D form of cfg4 (4 096 instances) took 113 us per attention layer.
__global__ void __launch_bounds__(256) chatt_mlp_bwd_inst_kernel(
    const float* __restrict__ ds, const float* __restrict__ hidden, const float* __restrict__ scale, const float* __restrict__ w1,
    const float* __restrict__ w2, float* __restrict__ dpre1, float* __restrict__ dpre2, float* __restrict__ dmean, int ninst, int C,
    int Cr, float inv_r) {
    const int inst = blockIdx.x * blockDim.x + threadIdx.x;
    if (inst >= ninst) return;
    const float* s = scale + (size_t)inst * C;
    const float* h = hidden + (size_t)inst * Cr;
    float* p2 = dpre2 + (size_t)inst * C;
    float* p1 = dpre1 + (size_t)inst * Cr;
    for (int c = 0; c < C; ++c) p2[c] = ds[(size_t)inst * C + c] * s[c] * (1.f - s[c]);
    for (int j = 0; j < Cr; ++j) {
        float dh = 0.f;
        for (int c = 0; c < C; ++c) dh += p2[c] * w2[j * C + c];
        p1[j] = (h[j] > 0.f) ? dh : 0.f;
    }
    for (int c = 0; c < C; ++c) {
        float dm = 0.f;
        for (int j = 0; j < Cr; ++j) dm += p1[j] * w1[c * Cr + j];
        dmean[(size_t)inst * C + c] = dm * inv_r;
    }
}
// dW1[c][j] = sum_inst mean[c]*dpre1[j] ; dW2[j][c] = sum_inst h[j]*dpre2[c] ; db1 = sum dpre1 ; db2 = sum dpre2
__global__ void __launch_bounds__(256) chatt_mlp_bwd_param_kernel(
    const float* __restrict__ mean, const float* __restrict__ hidden, const float* __restrict__ dpre1, const float* __restrict__ dpre2,
    float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2, int ninst, int C, int Cr,
    int accumulate) {
    const int lane = threadIdx.x & 63;
    const int nout = 2 * C * Cr + Cr + C;
    const int o = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (o >= nout) return;                                  // (whole waves)
    float a = 0.f;
    float* dst;
    if (o < C * Cr) {
        const int c = o / Cr, j = o - c * Cr;
        for (int inst = lane; inst < ninst; inst += 64) a += mean[(size_t)inst * C + c] * dpre1[(size_t)inst * Cr + j];
        dst = dw1 + c * Cr + j;
    } else if (o < 2 * C * Cr) {
        const int e = o - C * Cr, c = e / Cr, j = e - c * Cr;
        for (int inst = lane; inst < ninst; inst += 64) a += hidden[(size_t)inst * Cr + j] * dpre2[(size_t)inst * C + c];
        dst = dw2 + j * C + c;
    } else if (o < 2 * C * Cr + Cr) {
        const int j = o - 2 * C * Cr;
        for (int inst = lane; inst < ninst; inst += 64) a += dpre1[(size_t)inst * Cr + j];
        dst = db1 + j;
    } else {
        const int c = o - 2 * C * Cr - Cr;
        for (int inst = lane; inst < ninst; inst += 64) a += dpre2[(size_t)inst * C + c];
        dst = db2 + c;
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) a += __shfl_xor(a, m, 64);
    if (lane == 0) *dst = accumulate ? *dst + a : a;
}

int pick_tx(int Q) { return Q <= 8 ? 8 : (Q <= 16 ? 16 : (Q <= 32 ? 32 : 64)); }
int colsum_blocks(int R, int TY) { return std::max(1, std::min(cdiv(R, TY * 8), 256)); }

void colsum(hipStream_t s, const float* a, const float* b, float* partial, float* out, int G, int R, int Q, float scale) {
    const int TX = pick_tx(Q), TY = 256 / TX;
    // wide instances (5-D attention: Q = W C) with 16-byte-aligned rows: the float4 form
    if (Q >= 256 && (Q & 3) == 0 && ((uintptr_t)a & 15) == 0 && (!b || ((uintptr_t)b & 15) == 0) && ((uintptr_t)partial & 15) == 0 &&
        !exp_env("DL4DS_NO_COLSUM4")) {
        const int nb4 = std::max(1, std::min(cdiv(R, 4 * 8), 256));
        dim3 grid4((unsigned)nb4, (unsigned)cdiv(Q, 256), (unsigned)G);
        DL4DS_LAUNCH(colsum4_kernel, grid4, dim3(256), 0, s, a, b, partial, R, Q);
        HIP_CHECK(hipGetLastError());
        DL4DS_LAUNCH(colsum_finish_kernel, dim3(cdiv(G * Q, 256)), dim3(256), 0, s, partial, out, nb4, Q, G * Q, scale);
        HIP_CHECK(hipGetLastError());
        return;
    }
    const int nb = colsum_blocks(R, TY);
    dim3 grid((unsigned)nb, (unsigned)cdiv(Q, TX), (unsigned)G);
    switch (TX) {
        case 8: DL4DS_LAUNCH(colsum_kernel<8>, grid, dim3(256), 0, s, a, b, partial, R, Q); break;
        case 16: DL4DS_LAUNCH(colsum_kernel<16>, grid, dim3(256), 0, s, a, b, partial, R, Q); break;
        case 32: DL4DS_LAUNCH(colsum_kernel<32>, grid, dim3(256), 0, s, a, b, partial, R, Q); break;
        default: DL4DS_LAUNCH(colsum_kernel<64>, grid, dim3(256), 0, s, a, b, partial, R, Q); break;
    }
    HIP_CHECK(hipGetLastError());
    DL4DS_LAUNCH(colsum_finish_kernel, dim3(cdiv(G * Q, 256)), dim3(256), 0, s, partial, out, nb, Q, G * Q, scale);
    HIP_CHECK(hipGetLastError());
}
inline int ew_blocks(size_t n) { return (int)std::max<size_t>(1, std::min<size_t>(cdivz(n, 256), 8192)); }

}  // namespace

// workspace layout (floats): [partial: G*256*Q][ds: G*Q][dmean: G*Q][dpre1: G*P*Cr][dpre2: G*P*C]
size_t chatt_workspace_bytes(const AttShape& sh) {
    const size_t Q = (size_t)sh.P * sh.C;
    return ((size_t)sh.G * 256 * Q + 2 * (size_t)sh.G * Q + (size_t)sh.G * sh.P * (sh.Cr + sh.C)) * sizeof(float);
}

void chatt_forward(hipStream_t s, const float* x, float* y, const AttShape& sh, const float* w1, const float* b1,
                   const float* w2, const float* b2, float* mean, float* hidden, float* scale, float* workspace,
                   const float* pool_partial, int pool_tiles) {
    const int Q = sh.P * sh.C;
    const int ninst = sh.G * sh.P;
    // algorithmic traffic: one read of x for the pooling unless the producer supplied it, one read + one write for the scale
    // unless the consumer applies it while loading (y == nullptr)
    ProfScope ps(s, "chatt_fwd", 0.0, 4.0 * (double)sh.G * sh.R * Q * ((pool_partial ? 0 : 1) + (y ? 2 : 0)));
    if (pool_partial) {
        DL4DS_REQUIRE(sh.P == 1 && sh.C <= 8, "chatt: pooling partials come from the 8-channel pair convolution");
        DL4DS_LAUNCH(pool_finish_kernel, dim3(cdiv(sh.G, 4)), dim3(256), 0, s, pool_partial, mean, pool_tiles,
                           sh.C, sh.G, 1.f / (float)sh.R);
        HIP_CHECK(hipGetLastError());
    } else {
        colsum(s, x, nullptr, workspace, mean, sh.G, sh.R, Q, 1.f / (float)sh.R);
    }
    DL4DS_LAUNCH(chatt_mlp_fwd_kernel, dim3(cdiv(ninst * sh.C, 256)), dim3(256), 0, s, mean, w1, b1, w2, b2,
                       hidden, scale, ninst, sh.C, sh.Cr);
    HIP_CHECK(hipGetLastError());
    if (y == nullptr) return;                  // the consumer reads x through a view carrying `scale` (TView::sc)
    chatt_apply<false>(s, x, scale, nullptr, y, sh.G, sh.R, Q, 0);
}

void chatt_backward(hipStream_t s, const float* x, const float* dy, float* dx, int accumulate_dx, const AttShape& sh,
                    const float* w1, const float* w2, const float* mean, const float* hidden, const float* scale,
                    float* dw1, float* db1, float* dw2, float* db2, int accumulate_dw, float* workspace, float* dmean_out,
                    const float* ds_given) {
    const int Q = sh.P * sh.C;
    const int ninst = sh.G * sh.P;
    float* partial = workspace;
    float* ds = partial + (size_t)sh.G * 256 * Q;
    float* dmean = ds + (size_t)sh.G * Q;
    float* dpre1 = dmean + (size_t)sh.G * Q;
    float* dpre2 = dpre1 + (size_t)ninst * sh.Cr;
    if (dmean_out) dmean = dmean_out;           // kept by the caller: the producer reads dX = dY * scale + dmean lazily
    ProfScope ps(s, "chatt_bwd", 0.0, 4.0 * (double)sh.G * sh.R * Q * ((ds_given ? 0 : 2) + (dx ? 2 + (accumulate_dx ? 1 : 0) : 0)));
    if (ds_given == nullptr) colsum(s, dy, x, partial, ds, sh.G, sh.R, Q, 1.f);
    DL4DS_LAUNCH(chatt_mlp_bwd_inst_kernel, dim3(cdiv(ninst, 256)), dim3(256), 0, s, ds_given ? ds_given : ds, hidden, scale, w1, w2,
                       dpre1, dpre2, dmean, ninst, sh.C, sh.Cr, 1.f / (float)sh.R);
    HIP_CHECK(hipGetLastError());
    if (dw1) {              // (null: input gradient only -- the CGAN generator pass through the discriminator)
        const int nout = 2 * sh.C * sh.Cr + sh.Cr + sh.C;
        DL4DS_LAUNCH(chatt_mlp_bwd_param_kernel, dim3(cdiv(nout, 4)), dim3(256), 0, s, mean, hidden, dpre1, dpre2, dw1, db1, dw2, db2,
                           ninst, sh.C, sh.Cr, accumulate_dw);
        HIP_CHECK(hipGetLastError());
    }
    if (dx == nullptr) return;                  // dX is not materialised (TView::sc / sh on the producer's dY view)
    chatt_apply<true>(s, dy, scale, dmean, dx, sh.G, sh.R, Q, accumulate_dx);
}
